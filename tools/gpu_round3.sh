#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -x -q -m gpu > gpurun_out/pytest_new.log 2>&1; rc=$?
tail -15 gpurun_out/pytest_new.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload train --steps 100 --warmup 20 > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err || { tail -20 gpurun_out/bench_train.err; exit 1; }
cat gpurun_out/bench_train.json
