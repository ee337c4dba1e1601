#!/bin/bash
# [B, L*C] lookup with the transposition folded into the last level launch(es) (k_grid_fwd_level_rows): parity tests, then
# the lookup bench with the transposition pass (RN_GRID_ROWS=0), the default split and the other splits.
#   gpurun --timeout 900 -- bash tools/gpu_lookup_rows.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
O=gpurun_out/rows; mkdir -p "$O"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "planned or module" > "$O/pytest.txt" 2>&1 || { tail -30 "$O/pytest.txt"; exit 1; }
tail -1 "$O/pytest.txt"
show() { python - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        r = json.loads(line)
        print("   %-48s %-7s %.4f ms (min %.4f)" % (r["kernel"], r["points"], r["median_ms"], r.get("min_ms", 0)))
PY
}
run() { name=$1; shift; echo "== $name"; env "$@" timeout -k 10 200 python tools/bench_lookup.py --points ${POINTS:-frame,bundle} --layouts ${LAYOUTS:-lbc,blc,module} --dtype ${DTYPE:-f32} --rounds 20 --out "$O/lookup_$name.json" > "$O/lookup_$name.log" 2>&1 || { tail "$O/lookup_$name.log"; return 1; }; show "$O/lookup_$name.log"; }
if [ "${SWEEP:-0}" = 1 ]; then
  POINTS=frame LAYOUTS=blc run transpose RN_GRID_ROWS=0 || exit 1
  for l in ${LEVELS:-5 6 7 8 15}; do POINTS=frame LAYOUTS=blc run own$l RN_GRID_ROWS_OWN=$l || exit 1; done
  POINTS=frame LAYOUTS=blc run transpose_again RN_GRID_ROWS=0 || exit 1
  exit 0
fi
run transpose RN_GRID_ROWS=0 && run rows1 RN_GRID_ROWS=1 && run rows2 RN_GRID_ROWS_SEGS=2 && run rows1_chunk19 RN_GRID_CHUNK_LOG2=19 && run transpose_chunk19 RN_GRID_ROWS=0 RN_GRID_CHUNK_LOG2=19 &&
DTYPE=f16 POINTS=frame LAYOUTS=lbc,blc run f16_rows RN_GRID_ROWS=1 && DTYPE=f16 POINTS=frame LAYOUTS=lbc,blc run f16_transpose RN_GRID_ROWS=0
# profiler's view: one point set and one layout per CSV
cd /tmp
for p in frame bundle; do for m in rows transpose; do d="$R/$O/trace_${p}_$m"; mkdir -p "$d"
  if [ $m = transpose ]; then export RN_GRID_ROWS=0; else unset RN_GRID_ROWS; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$R/tools/bench_lookup.py" --points $p --layouts blc --rounds 10 > "$d/bench.log" 2> "$d/err.log" || { tail "$d/err.log"; exit 1; }
done; done
unset RN_GRID_ROWS
cd "$R"; find "$O" -name "*.db" -delete; find "$O" -name "*agent_info*" -delete; find "$O" -name "*kernel_trace.csv" -delete
for f in $(find "$O" -name "*kernel_stats.csv"); do echo "$f"; head -6 "$f" | cut -c1-160; done
