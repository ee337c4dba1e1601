#!/bin/bash
# One pass that produces every number DESIGN.md / profiles/ quote for round 2: full GPU test suite, smoke, bench lines (all
# workloads and variants), rocprofv3 kernel traces, PMC traffic passes, the standalone lookup profile, training, rehearsals.
set -o pipefail
# PART=A: tests, smoke, bench lines, rehearsals.  PART=B: micro-benchmarks, rocprofv3 traces, counter passes, training profile,
# lookup profile.  PART=C: only the tail of B (training micro-benchmarks + profile, lookup profile).
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/final2"; mkdir -p "$O"
export TMPDIR=/tmp
cd "$R"
if [ "${PART:-A}" = "A" ]; then
timeout -k 10 1000 python -m pytest tests -q -m gpu > "$O/pytest_gpu.log" 2>&1; rc=$?; tail -3 "$O/pytest_gpu.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python __graft_entry__.py smoke > "$O/smoke.log" 2>&1 || { tail -5 "$O/smoke.log"; exit 1; }; tail -2 "$O/smoke.log"
timeout -k 10 400 python bench.py > "$O/bench_hash19_f32.json" 2> "$O/bench_hash19_f32.err" || { tail "$O/bench_hash19_f32.err"; exit 1; }
for v in "tiled16 f32" "hash19 f32x2" "tiled16 f32x2" "hash19 f16" "tiled16 f16"; do set -- $v
  timeout -k 10 300 python bench.py --grid $1 --mlp $2 --no-cpu-baseline > "$O/bench_$1_$2.json" 2> "$O/bench_$1_$2.err" || { tail "$O/bench_$1_$2.err"; exit 1; }
done
timeout -k 10 300 python bench.py --streams 2 --no-cpu-baseline > "$O/bench_hash19_f32_streams2.json" 2> "$O/bench_s2.err" || { tail "$O/bench_s2.err"; exit 1; }
timeout -k 10 300 python bench.py --regime A --no-cpu-baseline > "$O/bench_hash19_f32_regimeA.json" 2> "$O/bench_A.err" || { tail "$O/bench_A.err"; exit 1; }
timeout -k 10 300 python bench.py --audio-batch 0 --no-cpu-baseline > "$O/bench_hash19_f32_live_audio.json" 2> "$O/bench_la.err" || { tail "$O/bench_la.err"; exit 1; }
timeout -k 10 300 python bench.py --engine ops --steps 40 --warmup 5 --no-cpu-baseline > "$O/bench_hash19_ops_engine.json" 2> "$O/bench_ops.err" || { tail "$O/bench_ops.err"; exit 1; }
timeout -k 10 300 python bench.py --workload train --steps 128 > "$O/bench_train.json" 2> "$O/bench_train.err" || { tail "$O/bench_train.err"; exit 1; }
for i in 2 3 4 5; do timeout -k 10 300 python bench.py --workload train --steps 128 > "$O/bench_train_run$i.json" 2>/dev/null || exit 1; done
timeout -k 10 300 python bench.py --loop-launch coop --no-cpu-baseline > "$O/bench_hash19_f32_loop_coop.json" 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload train --steps 128 --train-engine eager > "$O/bench_train_eager.json" 2> "$O/bench_train_e.err" || { tail "$O/bench_train_e.err"; exit 1; }
timeout -k 10 300 python bench.py --workload tile --size 1024 --steps 60 --warmup 10 --no-cpu-baseline > "$O/bench_tile1024.json" 2> "$O/bench_tile.err" || { tail "$O/bench_tile.err"; exit 1; }
# multi-rank rehearsals on this one GPU (gloo; RCCL needs a GPU per rank): 4 ranks each
for w in "render" "tile --size 1024"; do tag=${w%% *}
  RN_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --workload $w --steps 40 --warmup 10 --no-cpu-baseline 2> "$O/rehearse4_$tag.err" | tail -1 > "$O/rehearse4_gloo_one_gpu_$tag.json" || { tail "$O/rehearse4_$tag.err"; exit 1; }
done
fi
if [ "${PART:-A}" = "B" ]; then
timeout -k 10 400 python tools/bench_kernels.py --rounds 10 --out "$O/kernels.json" > "$O/kernels.log" 2>&1 || { tail "$O/kernels.log"; exit 1; }
for m in f32 f32x2 f16; do for g in hash19 tiled16; do timeout -k 10 120 python tools/bench_fused.py --mlp $m --grid $g >> "$O/fused_kernel.jsonl" 2>/dev/null || exit 1; done; done
cd /tmp
for m in f32 f32x2 f16; do
  mkdir -p "$O/trace_$m" "$O/pmc_fetch_$m" "$O/pmc_write_$m"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$m" -- python3 "$R/bench.py" --mlp $m --steps 64 --warmup 20 --no-cpu-baseline > "$O/trace_$m/bench.json" 2> "$O/trace_$m/err.log" || { tail "$O/trace_$m/err.log"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$m" -- python3 "$R/bench.py" --mlp $m --steps 8 --warmup 20 --no-cpu-baseline > "$O/pmc_fetch_$m/bench.json" 2> "$O/pmc_fetch_$m/err.log" || { tail "$O/pmc_fetch_$m/err.log"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$m" -- python3 "$R/bench.py" --mlp $m --steps 8 --warmup 20 --no-cpu-baseline > "$O/pmc_write_$m/bench.json" 2> "$O/pmc_write_$m/err.log" || { tail "$O/pmc_write_$m/err.log"; exit 1; }
done
fi
if [ "${PART:-A}" = "B" ] || [ "${PART:-A}" = "C" ]; then
cd /tmp
timeout -k 10 200 python "$R/tools/bench_mlp.py" > "$O/mlp_kernels.json" 2> "$O/mlp_kernels.err" || { tail "$O/mlp_kernels.err"; exit 1; }
timeout -k 10 300 python "$R/tools/train_probe.py" > "$O/train_probe.json" 2> "$O/train_probe.err" || { tail "$O/train_probe.err"; exit 1; }
mkdir -p "$O/trace_train"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_train" -- python3 "$R/bench.py" --workload train --steps 64 --train-engine eager > "$O/trace_train/bench.json" 2> "$O/trace_train/err.log" || { tail "$O/trace_train/err.log"; exit 1; }
cd "$R"
bash tools/gpu_lookup_profile.sh > "$O/lookup_profile.log" 2>&1 || { tail "$O/lookup_profile.log"; exit 1; }
fi
# keep the merged-back payload small: drop everything but the csv summaries
find "$O" "$R/gpurun_out/lookup" -name "*.db" -delete; find "$O" "$R/gpurun_out/lookup" -name "*agent_info*" -delete
find "$O" "$R/gpurun_out/lookup" -name "*kernel_trace.csv" -size +8M -delete
du -sh "$O" "$R/gpurun_out/lookup"; ls "$O"
