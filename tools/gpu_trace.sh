#!/bin/bash
# rocprofv3 kernel trace of the default bench (args: extra bench flags); prints the per-frame kernel table
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/trace"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$O/bench.json" 2> "$O/err.log" || { tail "$O/err.log"; exit 1; }
find "$O" -name "*.db" -delete
python3 - <<PY
import csv,glob,os
fs=sorted(glob.glob("$O/**/*kernel_stats.csv",recursive=True), key=os.path.getmtime)
rows=list(csv.DictReader(open(fs[-1])))
frames=23
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel us/frame", round(tot/1e3/frames,1), "kernels/frame", round(sum(int(r['Calls']) for r in rows)/frames,1))
for r in rows[:28]:
    print(f"{int(r['TotalDurationNs'])/1e3/frames:8.1f} us/frame {int(r['Calls'])/frames:6.1f} calls {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
PY
