#!/bin/bash
# Build variants of rn_grid.hip on the GPU box and run the lookup micro-benchmark with each (the tree's .so is restored).
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v rn_grid.o)
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $v -c rad-nerf_amd/csrc/rn_grid.hip -o /tmp/v.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/v.o -o $SO || exit 1
  echo "variant $v"
  timeout -k 10 300 python tools/bench_kernels.py --rounds 10 --out /tmp/k.json 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if r.get('B') == 4194304 and r.get('points') == 'ray-ordered' and 'L,B,C' in r.get('layout', ''):
        print('  ', r['table'][:14], r['dtype'], round(r['median_ms'], 3), round(r['frac_of_hbm_peak'], 3))"
done
cp /tmp/orig.so $SO
