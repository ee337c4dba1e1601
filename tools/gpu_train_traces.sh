#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training bench (graph and eager), summaries only.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/final3"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/trace_train_graph" "$O/trace_train_eager"; mkdir -p "$O/trace_train_graph" "$O/trace_train_eager"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_train_graph" -- python3 "$R/bench.py" --workload train --steps 128 > "$O/trace_train_graph/bench.json" 2> "$O/trace_train_graph/err.log" || { tail "$O/trace_train_graph/err.log"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_train_eager" -- python3 "$R/bench.py" --workload train --steps 64 --train-engine eager > "$O/trace_train_eager/bench.json" 2> "$O/trace_train_eager/err.log" || { tail "$O/trace_train_eager/err.log"; exit 1; }
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*agent_info*" -delete; find "$O" -name "*kernel_trace.csv" -delete
RN_SCATTER=binned timeout -k 10 200 python bench.py --workload train --steps 128 > "$O/bench_train_scatter_binned.json" 2>/dev/null
timeout -k 10 120 python tools/bench_train_head.py > "$O/train_head_kernels.json" 2>/dev/null
ls "$O"/trace_train_*/*/ | head
