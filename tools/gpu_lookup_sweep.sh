#!/bin/bash
# Planned grid forward: ablations (no LDS staging / no coarse pass) and the chunk size of the [B, L*C] path.
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/lookup_sweep"; rm -rf "$O"; mkdir -p "$O"
cd "$R"
for t in hash19 tiled16; do
  for v in "base" "RN_GRID_NO_LDS=1" "RN_GRID_NO_COARSE=1" "RN_GRID_CHUNK_LOG2=18" "RN_GRID_CHUNK_LOG2=19" "RN_GRID_CHUNK_LOG2=21" "RN_GRID_CHUNK_LOG2=22"; do
    echo "== $t $v" >> "$O/sweep.log"
    if [ "$v" = base ]; then timeout -k 10 120 python tools/bench_lookup.py --table $t --points frame,bundle --layouts lbc,blc --rounds 15 >> "$O/sweep.log" 2>&1 || exit 1
    else env $v timeout -k 10 120 python tools/bench_lookup.py --table $t --points frame,bundle --layouts lbc,blc --rounds 15 >> "$O/sweep.log" 2>&1 || exit 1; fi
  done
done
python3 - <<PY
import json
for line in open("$O/sweep.log"):
    if line.startswith("=="): print(line.strip())
    elif line.startswith("{"):
        r = json.loads(line); print(f"   {r['points']:7s} {r['kernel']:42s} {r['median_ms']:.3f} ms {100*r['frac_of_hbm_peak']:.1f}%")
PY
