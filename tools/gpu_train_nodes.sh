#!/bin/bash
# Kernel time per replayed training step, by kernel (rocprofv3 --kernel-trace --stats over bench.py --workload train --steps 128:
# 33 warm-up + 128 timed + probe steps = 171 steps), then two plain bench lines.
#   gpurun --timeout 600 -- bash tools/gpu_train_nodes.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/tg"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 "$R/bench.py" --workload train --steps 128 > "$O/bench.json" 2> "$O/err.log" || { tail "$O/err.log"; exit 1; }
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info*" -delete
python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps, tot = 171, 0.0
for r in rows:
    per = float(r["TotalDurationNs"]) / steps / 1e3
    tot += per
    if per > 2.5:
        print("%7.1f us/step  calls/step %5.1f  %s" % (per, int(r["Calls"]) / steps, r["Name"][:84]))
print("total %.1f us/step, %.1f kernels/step" % (tot, sum(int(r["Calls"]) for r in rows) / steps))
PY
tail -1 "$O/bench.json" | cut -c1-160
for i in 1 2; do timeout -k 10 200 python bench.py --workload train --steps 128 2>/dev/null | tail -1 | cut -c1-140; done
