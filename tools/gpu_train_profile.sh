#!/bin/bash
# Training step (BASELINE config 2): bench line + rocprofv3 kernel stats of the same command.
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/train_prof"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --workload train --steps 64 --warmup 20 "$@" > "$O/bench.json" 2> "$O/err.log" || { tail "$O/err.log"; exit 1; }
find "$O" -name "*.db" -delete
cut -c1-300 "$O/bench.json"
python3 - <<PY
import csv, glob, json
f = glob.glob("$O/trace/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
steps = json.load(open("$O/bench.json"))["steps"] + 20
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e3 / steps
print("GPU busy per step (us):", round(tot, 1), " kernels/step:", round(sum(int(r["Calls"]) for r in rows) / steps, 1))
for r in rows[:30]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/steps:6.1f}/step {float(r['TotalDurationNs'])/1e3/steps:8.1f} us/step")
PY
