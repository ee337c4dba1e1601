#!/bin/bash
# After a change of the fused kernels: parity tests of the fused path, isolated kernel times, per-phase clocks, bench FPS.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_tile.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for m in f32 f32x2 f16; do
  timeout -k 10 120 python tools/bench_fused.py --mlp $m --grid hash19 --sweep 8192,131072,206016,1048576 2>/dev/null
done
for m in f32 f32x2 f16; do for g in hash19 tiled16; do python bench.py --mlp $m --grid $g --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('bench $m $g fps', round(d['value'],1), 'roofline', d['roofline']['frac'])"; done; done
bash tools/gpu_phase_clock.sh
