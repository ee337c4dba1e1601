#!/bin/bash
# Build variants of rn_fused.hip and print the per-frame time of the loop kernels (rocprofv3 trace of the f16 bench).
set -o pipefail
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v "rn_fused.o")
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $v -c rad-nerf_amd/csrc/rn_fused.hip -o /tmp/v.o 2>/dev/null || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/v.o -o $SO || exit 1
  echo "variant $v"
  bash tools/gpu_trace.sh --mlp f16 2>/dev/null | grep "k_head_\|total kernel" | cut -c1-80
done
cp /tmp/orig.so $SO
