#!/bin/bash
# Per-phase time of one tile inside k_nerf_fused (fp32): builds the kernel with -DRN_PHASE_CLOCK, runs tools/phase_clock.py.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v "rn_fused.o")
for extra in "" $VARIANTS; do
  echo "== variant: $extra"
  /opt/rocm/bin/hipcc $FLAGS -DRN_PHASE_CLOCK $(echo $extra | tr "," " ") -c rad-nerf_amd/csrc/rn_fused.hip -o /tmp/rn_fused.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/rn_fused.o -o $SO || exit 1
  timeout -k 10 300 python tools/phase_clock.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/orig.so $SO
