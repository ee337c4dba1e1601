#!/bin/bash
# Build variants of rn_fused.hip (fp32 kernel) on the GPU box and time each; the tree's .so is restored.
# usage: tools/gpu_variants_fused.sh "-DFLAG=1" "-DFLAG=2" ...
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v "rn_fused.o")
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $v -c rad-nerf_amd/csrc/rn_fused.hip -o /tmp/v.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A14 "k_nerf_fusedIffE" | grep "VGPRs Spill" | head -1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/v.o -o $SO || exit 1
  for grid in hash19 tiled16; do
    timeout -k 10 120 python tools/bench_fused.py --mlp f32 --grid $grid --tag="$v" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['tag'], d['grid'], d['M1048576_ms'], d['M206000_ms'])"
  done
done
cp /tmp/orig.so $SO
