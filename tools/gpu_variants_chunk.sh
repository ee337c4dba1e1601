#!/bin/bash
# Tile chunk size of the XCD-aware schedule (RN_TILE_CHUNK) for the three fused kernels: isolated kernel time and bench FPS.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v "rn_fused.o\|rn_fused_h16.o\|rn_fused_x2.o")
for v in ${CHUNKS:-8 4 2 1}; do
  for f in rn_fused rn_fused_h16 rn_fused_x2; do /opt/rocm/bin/hipcc $FLAGS -DRN_TILE_CHUNK=$v $EXTRA -c rad-nerf_amd/csrc/$f.hip -o /tmp/$f.o || exit 1; done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/rn_fused.o /tmp/rn_fused_h16.o /tmp/rn_fused_x2.o -o $SO || exit 1
  echo "RN_TILE_CHUNK=$v"
  for m in f32 f32x2 f16; do for g in hash19; do
    timeout -k 10 120 python tools/bench_fused.py --mlp $m --grid $g 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', d['mlp'], d['grid'], d['M1048576_ms'], d['M206000_ms'])"
  done; done
  for m in f32 f32x2 f16; do python bench.py --mlp $m --no-cpu-baseline | python3 -c "import sys,json; d=json.load(sys.stdin); print('   bench $m fps', round(d['value'],1))"; done
done
cp /tmp/orig.so $SO
