#!/bin/bash
# Counters of the table-gradient kernel at a training step's size: how many atomic requests leave the CUs per launch.
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/grid_bwd"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 python3 "$R/tools/bench_grid_backward.py" > "$O/timing.json" 2> "$O/timing.err" || { tail "$O/timing.err"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum SQ_INSTS_LDS_ATOMIC GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d "$O/pmc" -- python3 "$R/tools/bench_grid_backward.py" > "$O/pmc.log" 2> "$O/pmc.err" || { tail "$O/pmc.err"; exit 1; }
find "$O" -name "*.db" -delete
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_grid_bwd" in r["Kernel_Name"] or "k_grid_scatter" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} | {"launches": max(len(v) for v in cs.values())} for k, cs in agg.items()}
out["timing"] = json.load(open("$O/timing.json"))
json.dump(out, open("$O/counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
