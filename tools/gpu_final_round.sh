#!/bin/bash
# One pass that produces every number DESIGN.md / profiles/ quote for this round:
# full GPU test suite, smoke, bench lines (all workloads), rocprofv3 kernel traces, PMC traffic passes, micro-benchmarks.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/final"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd "$R"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$O/pytest_gpu.log" 2>&1; rc=$?; tail -3 "$O/pytest_gpu.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python __graft_entry__.py smoke > "$O/smoke.log" 2>&1 || { tail -5 "$O/smoke.log"; exit 1; }; tail -2 "$O/smoke.log"
timeout -k 10 400 python bench.py > "$O/bench_hash19_f32.json" 2> "$O/bench_hash19_f32.err" || { tail "$O/bench_hash19_f32.err"; exit 1; }
for v in "tiled16 f32" "hash19 f32x2" "tiled16 f32x2" "hash19 f16" "tiled16 f16"; do set -- $v
  timeout -k 10 300 python bench.py --grid $1 --mlp $2 --no-cpu-baseline > "$O/bench_$1_$2.json" 2> "$O/bench_$1_$2.err" || { tail "$O/bench_$1_$2.err"; exit 1; }
done
timeout -k 10 300 python bench.py --engine ops --no-cpu-baseline > "$O/bench_hash19_ops_engine.json" 2> "$O/bench_ops.err" || { tail "$O/bench_ops.err"; exit 1; }
timeout -k 10 300 python bench.py --workload train --steps 100 --warmup 20 > "$O/bench_train.json" 2> "$O/bench_train.err" || { tail "$O/bench_train.err"; exit 1; }
timeout -k 10 300 python bench.py --workload tile --size 1024 --steps 30 --no-cpu-baseline > "$O/bench_tile1024.json" 2> "$O/bench_tile.err" || { tail "$O/bench_tile.err"; exit 1; }
timeout -k 10 400 python tools/bench_kernels.py --rounds 10 --out "$O/kernels.json" > "$O/kernels.log" 2>&1 || { tail "$O/kernels.log"; exit 1; }
for m in f32 f32x2 f16; do for g in hash19 tiled16; do timeout -k 10 120 python tools/bench_fused.py --mlp $m --grid $g >> "$O/fused_kernel.jsonl" 2>/dev/null || exit 1; done; done
cd /tmp
for m in f32 f32x2 f16; do
  mkdir -p "$O/trace_$m" "$O/pmc_fetch_$m" "$O/pmc_write_$m"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$m" -- python3 "$R/bench.py" --mlp $m --steps 20 --warmup 3 --no-cpu-baseline > "$O/trace_$m/bench.json" 2> "$O/trace_$m/err.log" || { tail "$O/trace_$m/err.log"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$m" -- python3 "$R/bench.py" --mlp $m --steps 6 --warmup 2 --no-cpu-baseline > "$O/pmc_fetch_$m/bench.json" 2> "$O/pmc_fetch_$m/err.log" || { tail "$O/pmc_fetch_$m/err.log"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$m" -- python3 "$R/bench.py" --mlp $m --steps 6 --warmup 2 --no-cpu-baseline > "$O/pmc_write_$m/bench.json" 2> "$O/pmc_write_$m/err.log" || { tail "$O/pmc_write_$m/err.log"; exit 1; }
done
# keep the merged-back payload small: drop everything but the csv summaries
find "$O" -name "*.db" -delete; find "$O" -name "*agent_info*" -delete
du -sh "$O"; ls "$O"
