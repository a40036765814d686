#!/bin/bash
# steady-state backbone kernel time per forward with and without the fused 1x1 convolutions (rocprofv3 --kernel-trace of bench.py --streams 1, B=8)
export TMPDIR=/tmp
for v in 1 0; do
  rm -rf gpurun_out/prof
  SCENEEGO_CONV1X1=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --batch ${1:-8} --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras --no-repeats --streams 1 > gpurun_out/prof_run.log 2>&1
  t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
  echo "== SCENEEGO_CONV1X1=$v"
  python3 tools/diag/backbone_time.py "$t" 5
done
rm -rf gpurun_out/prof
