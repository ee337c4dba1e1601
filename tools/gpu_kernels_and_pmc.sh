#!/bin/bash
# Kernel micro-benchmarks + rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) of the bench command.
mkdir -p gpurun_out/pmc_fetch gpurun_out/pmc_write
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
timeout -k 10 600 python tools/bench_kernels.py --rounds 10 --out gpurun_out/kernels.json > gpurun_out/kernels.log 2>&1 || { tail -n 20 gpurun_out/kernels.log; exit 1; }
tail -n 3 gpurun_out/kernels.log | cut -c1-400
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fetch" -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/pmc_fetch/bench.json" 2> "$R/gpurun_out/pmc_fetch/err.log" || { tail -n 20 "$R/gpurun_out/pmc_fetch/err.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_write" -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/pmc_write/bench.json" 2> "$R/gpurun_out/pmc_write/err.log" || { tail -n 20 "$R/gpurun_out/pmc_write/err.log"; exit 1; }
find "$R/gpurun_out/pmc_fetch" "$R/gpurun_out/pmc_write" -name "*.csv" | head
