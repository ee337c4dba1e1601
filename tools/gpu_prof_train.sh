#!/bin/bash
# rocprofv3 kernel trace of the training workload
rm -rf gpurun_out/prof_train; mkdir -p gpurun_out/prof_train
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_train" -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload train --steps 48 --warmup 20 > "$GRAFT_REPO_ROOT/gpurun_out/prof_train/bench_under_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_train/err.log"
echo rc=$?
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof_train" -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-200
