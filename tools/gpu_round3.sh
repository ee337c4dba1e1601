#!/bin/bash
# One pass that produces the numbers DESIGN.md / profiles/ quote for round 3.  PART=A: tests, smoke, bench lines, rehearsals.
# PART=B: rocprofv3 kernel stats (render, training graph / eager), PMC traffic passes, lookup sweep, scatter counters.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/final3"; mkdir -p "$O"
export TMPDIR=/tmp
cd "$R"
run() { local out="$1"; shift; timeout -k 10 400 "$@" > "$O/$out" 2> "$O/${out%.json}.err" || { echo "FAILED: $*"; tail -5 "$O/${out%.json}.err"; exit 1; }; }
if [ "${PART:-A}" = "A" ]; then
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout -k 10 1000 python -m pytest tests -q -m gpu > "$O/pytest_gpu.log" 2>&1; rc=$?; tail -3 "$O/pytest_gpu.log"; [ $rc -eq 0 ] || exit $rc
fi
[ "${ONLY_TESTS:-0}" = 1 ] && exit 0
timeout -k 10 300 python __graft_entry__.py smoke > "$O/smoke.log" 2>&1 || { tail -5 "$O/smoke.log"; exit 1; }; tail -2 "$O/smoke.log"
run bench_hash19_f32.json python bench.py
run bench_hash19_f32_driver_shape.json python bench.py --steps 20 --warmup 5
for v in "tiled16 f32" "hash19 f32x2" "hash19 f16" "tiled16 f16"; do set -- $v
  run bench_$1_$2.json python bench.py --grid $1 --mlp $2 --no-cpu-baseline --no-train-record
done
run bench_hash19_f16_half_tables.json python bench.py --mlp f16 --half-tables --no-cpu-baseline --no-train-record
run bench_hash19_f32_regimeA.json python bench.py --regime A --no-cpu-baseline --no-train-record
run bench_hash19_f32_streams2.json python bench.py --streams 2 --no-cpu-baseline --no-train-record
run bench_tile1024.json python bench.py --workload tile --size 1024 --steps 60 --warmup 10 --no-cpu-baseline
for i in 1 2 3 4 5; do run bench_train_run$i.json python bench.py --workload train --steps 128; done
run bench_train_eager.json python bench.py --workload train --steps 128 --train-engine eager
RN_TRAIN_OVERLAP=0 run bench_train_no_overlap.json python bench.py --workload train --steps 128
RN_SCATTER=lbc run bench_train_scatter_line_merge_only.json python bench.py --workload train --steps 128
RN_TRAIN_HEAD=ops RN_TRAIN_LOSS=torch run bench_train_round2_operator_path.json python bench.py --workload train --steps 128
timeout -k 10 400 python tools/train_step_launches.py > "$O/train_step_launches.json" 2> /dev/null || exit 1
for w in "render" "tile --size 1024"; do tag=${w%% *}
  RN_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --workload $w --steps 40 --warmup 10 --no-cpu-baseline 2> "$O/rehearse4_$tag.err" | tail -1 > "$O/rehearse4_gloo_one_gpu_$tag.json" || { tail "$O/rehearse4_$tag.err"; exit 1; }
done
fi
if [ "${PART:-A}" = "C" ]; then   # the training lines again (after a change that only touches the training step)
for i in 1 2 3 4 5; do run bench_train_run$i.json python bench.py --workload train --steps 128; done
run bench_train_eager.json python bench.py --workload train --steps 128 --train-engine eager
RN_TRAIN_OVERLAP=0 run bench_train_no_overlap.json python bench.py --workload train --steps 128
RN_SCATTER=binned run bench_train_scatter_binned.json python bench.py --workload train --steps 128
RN_SCATTER_DIRECT=0 run bench_train_scatter_line_merge_only.json python bench.py --workload train --steps 128
RN_TRAIN_HEAD=ops RN_TRAIN_LOSS=torch run bench_train_round2_operator_path.json python bench.py --workload train --steps 128
run bench_hash19_f32_driver_shape.json python bench.py --steps 20 --warmup 5
run bench_hash19_f32.json python bench.py
timeout -k 10 400 python tools/train_step_launches.py > "$O/train_step_launches.json" 2> /dev/null || exit 1
timeout -k 10 120 python tools/bench_train_head.py > "$O/train_head_kernels.json" 2>/dev/null || exit 1
cd /tmp
rm -rf "$O/trace_train_graph" "$O/trace_train_eager"; mkdir -p "$O/trace_train_graph" "$O/trace_train_eager"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_train_graph" -- python3 "$R/bench.py" --workload train --steps 128 > "$O/trace_train_graph/bench.json" 2> "$O/trace_train_graph/err.log" || { tail "$O/trace_train_graph/err.log"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_train_eager" -- python3 "$R/bench.py" --workload train --steps 64 --train-engine eager > "$O/trace_train_eager/bench.json" 2> "$O/trace_train_eager/err.log" || { tail "$O/trace_train_eager/err.log"; exit 1; }
cd "$R"
bash tools/gpu_grid_bwd_counters.sh > "$O/grid_bwd_counters.log" 2>&1 || { tail "$O/grid_bwd_counters.log"; exit 1; }
cp gpurun_out/grid_bwd/counters.json "$O/grid_backward_counters.json"
cp gpurun_out/grid_bwd/timing.json "$O/grid_backward_timing.json"
f=$(ls -t gpurun_out/grid_bwd/pmc/runc/*_counter_collection.csv | head -1); python3 - "$f" > "$O/grid_backward_requests_by_point_set.json" <<PY
import csv, sys, collections, json
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" and ("scatter" in r["Kernel_Name"] or "k_grid_bwd" in r["Kernel_Name"] or "k_grid_bin" in r["Kernel_Name"])]
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
print(json.dumps({k: [v for _, v in sorted(vs)] for k, vs in by.items()}, indent=0))
PY
fi
if [ "${PART:-A}" = "B" ]; then
cd /tmp
mkdir -p "$O/trace_f32" "$O/trace_train_graph" "$O/trace_train_eager" "$O/pmc_fetch_f32" "$O/pmc_write_f32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_f32" -- python3 "$R/bench.py" --steps 64 --warmup 20 --no-cpu-baseline --no-train-record > "$O/trace_f32/bench.json" 2> "$O/trace_f32/err.log" || { tail "$O/trace_f32/err.log"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_train_graph" -- python3 "$R/bench.py" --workload train --steps 128 > "$O/trace_train_graph/bench.json" 2> "$O/trace_train_graph/err.log" || { tail "$O/trace_train_graph/err.log"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_train_eager" -- python3 "$R/bench.py" --workload train --steps 64 --train-engine eager > "$O/trace_train_eager/bench.json" 2> "$O/trace_train_eager/err.log" || { tail "$O/trace_train_eager/err.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_f32" -- python3 "$R/bench.py" --steps 8 --warmup 20 --no-cpu-baseline --no-train-record > "$O/pmc_fetch_f32/bench.json" 2> "$O/pmc_fetch_f32/err.log" || { tail "$O/pmc_fetch_f32/err.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_f32" -- python3 "$R/bench.py" --steps 8 --warmup 20 --no-cpu-baseline --no-train-record > "$O/pmc_write_f32/bench.json" 2> "$O/pmc_write_f32/err.log" || { tail "$O/pmc_write_f32/err.log"; exit 1; }
cd "$R"
mkdir -p "$O/lookup"
for c in 19 22; do RN_GRID_CHUNK_LOG2=$c timeout -k 10 200 python tools/bench_lookup.py --points frame,bundle --layouts lbc,blc,module --rounds 20 --out "$O/lookup/lookup_hash19_chunk$c.json" > /dev/null 2>&1 || exit 1; done
cd /tmp
for p in frame bundle; do mkdir -p "$O/lookup/trace_$p"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/lookup/trace_$p" -- python3 "$R/tools/bench_lookup.py" --points $p --layouts lbc,blc --rounds 10 > "$O/lookup/trace_$p/bench.log" 2> "$O/lookup/trace_$p/err.log" || { tail "$O/lookup/trace_$p/err.log"; exit 1; }
done
cd "$R"
bash tools/gpu_grid_bwd_counters.sh > "$O/grid_bwd_counters.log" 2>&1 || { tail "$O/grid_bwd_counters.log"; exit 1; }
cp gpurun_out/grid_bwd/counters.json "$O/grid_backward_counters.json"
fi
find "$O" -name "*.db" -delete; find "$O" -name "*agent_info*" -delete
find "$O" -name "*kernel_trace.csv" -size +8M -delete
du -sh "$O"; ls "$O"
