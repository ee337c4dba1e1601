#!/bin/bash
# Build variants of rn_fused_h16.hip on the GPU box and time each (scratch experiment; the tree's .so is restored at the end).
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v rn_fused_h16.o)
: > gpurun_out/variants_h16.log
for v in "1 2 4" "0 2 4" "1 4 4" "0 4 4" "1 2 8" "1 1 2"; do
  set -- $v
  /opt/rocm/bin/hipcc $FLAGS -DRN_FUSED_PAIR_HASHED=$1 -DRN_XYZ_GROUP=$2 -DRN_AMB_GROUP=$3 -c rad-nerf_amd/csrc/rn_fused_h16.hip -o /tmp/h16_v.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/h16_v.o -o $SO || exit 1
  for grid in hash19 tiled16; do
    timeout -k 10 120 python tools/bench_fused.py --mlp f16 --grid $grid --tag "pair=$1 xyz=$2 amb=$3" >> gpurun_out/variants_h16.log 2>/dev/null || exit 1
  done
done
cp /tmp/orig.so $SO
cat gpurun_out/variants_h16.log
