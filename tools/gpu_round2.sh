#!/bin/bash
# GPU round: new tests first, then the secondary workloads (train, tile 1024^2).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_tile.py -x -q -m gpu > gpurun_out/pytest_new.log 2>&1; rc=$?
tail -15 gpurun_out/pytest_new.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload train --steps 100 --warmup 20 > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err || { tail -20 gpurun_out/bench_train.err; exit 1; }
cat gpurun_out/bench_train.json
timeout -k 10 300 python bench.py --workload tile --size 1024 --steps 30 --no-cpu-baseline > gpurun_out/bench_tile1024.json 2> gpurun_out/bench_tile.err || { tail -20 gpurun_out/bench_tile.err; exit 1; }
cat gpurun_out/bench_tile1024.json
