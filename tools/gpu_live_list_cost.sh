#!/bin/bash
# What the live-sample list costs the loop's small kernels: rocprofv3 stats of the default bench with and without it.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/livelist"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do d="$O/ll$v"; rm -rf "$d"; mkdir -p "$d"
  RN_LIVE_LIST=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$R/bench.py" --steps 64 --warmup 20 --no-cpu-baseline --no-train-record > "$d/bench.json" 2> "$d/err.log" || { tail "$d/err.log"; exit 1; }
done
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*agent_info*" -delete; find "$O" -name "*kernel_trace.csv" -delete
python - "$O" <<'PY'
import csv, glob, sys, json
for v in ("1", "0"):
    f = glob.glob(sys.argv[1] + f"/ll{v}/**/*kernel_stats.csv", recursive=True)[0]
    print("live list", v, json.loads(open(sys.argv[1] + f"/ll{v}/bench.json").read().strip().splitlines()[-1])["value"])
    for r in list(csv.DictReader(open(f)))[:7]:
        print("   %-46s calls %5s avg %7.1f us" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
