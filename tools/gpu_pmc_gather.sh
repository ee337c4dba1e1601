#!/bin/bash
# TLB / L1 / L2 counters of the isolated f16 fused kernel, hash T=2^19 vs tiled T=2^16 xyz grid (separate --pmc passes).
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmc_gather"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
for g in hash19 tiled16; do
  timeout -k 10 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/a_$g" -- python3 "$R/tools/bench_fused.py" --mlp f16 --grid $g --rounds 4 > "$O/a_$g.log" 2>&1 || { tail -5 "$O/a_$g.log"; }
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY TA_TA_BUSY_sum --kernel-trace --output-format csv -d "$O/b_$g" -- python3 "$R/tools/bench_fused.py" --mlp f16 --grid $g --rounds 4 > "$O/b_$g.log" 2>&1 || { tail -5 "$O/b_$g.log"; }
done
find "$O" -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
for g in ("hash19", "tiled16"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$O/?_%s/**/*counter_collection.csv" % g, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_nerf_fused_h16" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 256 * 512:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(g, {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())}, "launches", {k: len(v) for k, v in agg.items()}.get("GRBM_GUI_ACTIVE"))
PY
