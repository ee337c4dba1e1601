#!/bin/bash
# rocprofv3 kernel trace of the bench command (summary copied to profiles/ by hand afterwards)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof/bench_under_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof/err.log"
echo rc=$?
find "$GRAFT_REPO_ROOT/gpurun_out/prof" -name "*stats*" | head
