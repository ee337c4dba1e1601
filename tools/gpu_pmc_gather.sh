#!/bin/bash
# Address-path / L1 / L2 / issue counters of one isolated fused kernel (MLP=f32|f32x2|f16), hash T=2^19 vs tiled T=2^16
# xyz grid, 2^20 samples per launch (separate --pmc passes, no trace domains besides --kernel-trace).
R="$GRAFT_REPO_ROOT"; MLP=${MLP:-f32}; O="$R/gpurun_out/pmc_gather_$MLP"; rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
for g in hash19 tiled16; do
  i=0
  for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
             "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/${i}_$g" -- python3 "$R/tools/bench_fused.py" --mlp $MLP --grid $g --rounds 3 --sweep 1048576 > "$O/${i}_$g.log" 2>&1 || { tail -5 "$O/${i}_$g.log"; }
  done
done
find "$O" -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
for g in ("hash19", "tiled16"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$O/?_%s/**/*counter_collection.csv" % g, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_nerf_fused" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 256 * 256:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(g, "$MLP", {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())}, "launches", max(len(v) for v in agg.values()) if agg else 0)
PY
