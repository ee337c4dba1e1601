#!/bin/bash
# Standalone grid lookup at hash T=2^19, B=2^22: HIP-event timings, rocprofv3 --kernel-trace --stats of the same command, and
# PMC passes (FETCH_SIZE, WRITE_SIZE, L1/L2 request counters) -- each counter set in its own run, no trace domain besides
# --kernel-trace.  Output: gpurun_out/lookup/ ; tools/refresh_lookup_profiles.py copies the summaries to profiles/.
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/lookup"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
POINTS=${POINTS:-frame,bundle}
LAYOUTS=${LAYOUTS:-lbc0,lbc,blc0,blc,module}
PMC_POINTS=${PMC_POINTS:-frame}   # the counter passes see ONE point set, so per-kernel means are not a mix
cd "$R"
timeout -k 10 300 python tools/bench_lookup.py --points $POINTS --layouts $LAYOUTS --per-level --out "$O/lookup_hash19.json" > "$O/lookup_hash19.log" 2>&1 || { tail -20 "$O/lookup_hash19.log"; exit 1; }
timeout -k 10 200 python tools/bench_lookup.py --table tiled16 --points $POINTS --layouts $LAYOUTS --out "$O/lookup_tiled16.json" > "$O/lookup_tiled16.log" 2>&1 || { tail -20 "$O/lookup_tiled16.log"; exit 1; }
cd /tmp
mkdir -p "$O/trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/tools/bench_lookup.py" --points $POINTS --layouts $LAYOUTS --rounds 10 > "$O/trace/bench.log" 2> "$O/trace/err.log" || { tail "$O/trace/err.log"; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  mkdir -p "$O/pmc_$i"
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc_$i" -- python3 "$R/tools/bench_lookup.py" --points $PMC_POINTS --layouts $LAYOUTS --rounds 3 > "$O/pmc_$i/bench.log" 2> "$O/pmc_$i/err.log" || { tail "$O/pmc_$i/err.log"; exit 1; }
done
find "$O" -name "*.db" -delete; find "$O" -name "*agent_info*" -delete
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_grid_" in k and int(r["Grid_Size"]) >= (1 << 18):
            agg[k.split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} | {"launches": max(len(v) for v in cs.values())} for k, cs in agg.items()}
json.dump(out, open("$O/counters_mean_per_launch.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
du -sh "$O"
