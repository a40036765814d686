#!/bin/bash
# per-kernel-name listing of the backbone's launches per forward (steady state, B=8)
export TMPDIR=/tmp
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --batch ${1:-8} --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras --no-repeats --streams 1 > gpurun_out/prof_run.log 2>&1
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/diag/backbone_time.py "$t" 5 -v
rm -rf gpurun_out/prof
