#!/bin/bash
# Idle time between the kernels of the frame loop, with and without bench.py's in-region kernel timing.
R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; cd /tmp
for prof in 1 0; do
  rm -rf $R/gpurun_out/trace_gap_$prof
  RN_BENCH_PROF=$prof rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_gap_$prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "== in-bench kernel timing: $prof"
  python3 $R/tools/trace_gaps.py $(ls $R/gpurun_out/trace_gap_$prof/*/*kernel_trace.csv | head -1)
  find $R/gpurun_out/trace_gap_$prof -name "*.db" -delete
done
cd $R
for prof in 1 0; do RN_BENCH_PROF=$prof python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); print('prof $prof fps', round(d['value'],1))"; done
