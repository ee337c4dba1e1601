#!/bin/bash
# Standalone grid encoder after a change: bit-exactness tests, then the lookup micro-benchmark with / without hashed x-pairs.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
[ -n "$SKIP_TESTS" ] || { timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3 || exit 1; }
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v "rn_grid.o")
for v in 1 0; do
  /opt/rocm/bin/hipcc $FLAGS -DRN_GRID_PAIR_HASHED=$v -c rad-nerf_amd/csrc/rn_grid.hip -o /tmp/rn_grid.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/rn_grid.o -o $SO || exit 1
  echo "== RN_GRID_PAIR_HASHED=$v"
  timeout -k 10 300 python tools/bench_kernels.py --rounds 10 --out gpurun_out/kernels_pair$v.json 2>&1 | grep -i "grid_encode_forward" | grep "level-major \[L,B,C\]" | grep "\"f32\"" | cut -c1-400
done
cp /tmp/orig.so $SO
