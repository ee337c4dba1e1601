#!/bin/bash
# GPU round for the hash-grid workload: parity tests, kernel microbench, bench lines for both xyz grids.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_kernels.py --rounds 10 --out gpurun_out/kernels.json > gpurun_out/kernels.log 2>&1 || { tail -20 gpurun_out/kernels.log; exit 1; }
grep -E '"B": 4194304' gpurun_out/kernels.log | grep ray-ordered | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['table'][:12], r['dtype'], r['layout'][:18], round(r['median_ms'],3), round(r['frac_of_hbm_peak'],3))"
tail -1 gpurun_out/kernels.log
timeout -k 10 300 python bench.py --grid hash19 > gpurun_out/bench_hash19.json 2> gpurun_out/bench_hash19.err && cat gpurun_out/bench_hash19.json &&
timeout -k 10 300 python bench.py --grid tiled16 --no-cpu-baseline > gpurun_out/bench_tiled16.json 2> gpurun_out/bench_tiled16.err && cat gpurun_out/bench_tiled16.json
