#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > gpurun_out/pytest_new.log 2>&1; rc=$?
tail -15 gpurun_out/pytest_new.log
[ $rc -eq 0 ] || exit $rc
for g in hash19 tiled16; do timeout -k 10 120 python tools/bench_fused.py --mlp f32x2 --grid $g; done
for mlp in f32 f32x2 f16; do
timeout -k 10 300 python bench.py --mlp $mlp --no-cpu-baseline > gpurun_out/bench_$mlp.json 2> gpurun_out/bench_$mlp.err || { tail -20 gpurun_out/bench_$mlp.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$mlp.json"))
r=d["roofline"]
print("$mlp", "fps", round(d["value"],1), "ms", round(d["ms_per_step"],3), "bound", r["bound"], "frac", round(r["frac"],3), "avg_with_work_ms", round(r["avg_launch_ms_with_work"],4), "share", round(r["share_of_step"],3))
PY
done
