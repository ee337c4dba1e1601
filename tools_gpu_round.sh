#!/bin/bash
# One gpurun round: GPU parity tests, smoke, bench, rocprof kernel trace. Output under gpurun_out/.
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/test_gpu.log 2>&1
rc=$?
tail -n 25 gpurun_out/test_gpu.log
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -n 30 gpurun_out/smoke.log; exit 1; }
tail -n 5 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -n 30 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
