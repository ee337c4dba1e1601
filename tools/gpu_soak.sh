#!/bin/bash
# Repeat the GPU suite and the determinism check a few times (flakiness soak).
set -o pipefail
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -1 || exit 1
  timeout -k 10 120 python tools/check_determinism.py --launches 24 || exit 1
done
for i in 1 2 3; do python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.load(sys.stdin); print(round(d['value'],1))"; done
