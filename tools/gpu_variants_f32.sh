#!/bin/bash
# Build variants of rn_fused.hip on the GPU box and time each (scratch experiment; the tree's .so is restored at the end).
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
SO=rad-nerf_amd/lib/libradnerf_hip.so
cp $SO /tmp/orig.so
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude"
objs=$(ls rad-nerf_amd/csrc/*.o | grep -v "rn_fused.o")
: > gpurun_out/variants_f32.log
for v in "1" "0"; do
  /opt/rocm/bin/hipcc $FLAGS -DRN_F32_PIPELINE=$v -c rad-nerf_amd/csrc/rn_fused.hip -o /tmp/f32_v.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/f32_v.o -o $SO || exit 1
  for grid in hash19 tiled16; do
    timeout -k 10 120 python tools/bench_fused.py --mlp f32 --grid $grid --tag "pipeline=$v" >> gpurun_out/variants_f32.log 2>/dev/null || exit 1
    timeout -k 10 200 python bench.py --grid $grid --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('pipeline=$v $grid fps', round(d['value'],1), 'work_ms', round(d['roofline']['avg_launch_ms_with_work'],4))" >> gpurun_out/variants_f32.log || exit 1
  done
done
cp /tmp/orig.so $SO
cat gpurun_out/variants_f32.log
