#!/bin/bash
# Quick check after a host-side change: fused/tile/audio/render GPU tests + the three bench lines.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_tile.py tests/test_gpu_audio.py tests/test_gpu_render.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for m in f32 f32x2 f16; do python bench.py --mlp $m --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('bench $m fps', round(d['value'],1))"; done
