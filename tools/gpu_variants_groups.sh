#!/bin/bash
# Build gather-depth variants of the two 16-bit fused kernels on the GPU box and time each (the tree's .so is restored).
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
: > gpurun_out/variants_groups.log
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v rn_fused_h16.o)
for v in "1 2" "2 4" "1 4" "2 2"; do set -- $v
  /opt/rocm/bin/hipcc $FLAGS -DRN_XYZ_GROUP=$1 -DRN_AMB_GROUP=$2 -c rad-nerf_amd/csrc/rn_fused_h16.hip -o /tmp/v.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/v.o -o $SO || exit 1
  for grid in hash19 tiled16; do timeout -k 10 120 python tools/bench_fused.py --mlp f16 --grid $grid --tag "h16 xyz=$1 amb=$2" >> gpurun_out/variants_groups.log 2>/dev/null || exit 1; done
done
cp /tmp/orig.so $SO
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v rn_fused_x2.o)
for v in 1 2; do
  /opt/rocm/bin/hipcc $FLAGS -DRN_X2_XYZ_GROUP=$v -c rad-nerf_amd/csrc/rn_fused_x2.hip -o /tmp/v.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/v.o -o $SO || exit 1
  for grid in hash19 tiled16; do timeout -k 10 120 python tools/bench_fused.py --mlp f32x2 --grid $grid --tag "x2 xyz=$v" >> gpurun_out/variants_groups.log 2>/dev/null || exit 1; done
done
cp /tmp/orig.so $SO
python3 -c "
import json
for l in open('gpurun_out/variants_groups.log'):
    d=json.loads(l); print(d['tag'], d['grid'], d['M1048576_ms'], d['M206000_ms'])"
