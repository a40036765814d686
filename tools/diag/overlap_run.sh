#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-parity --no-extras --no-repeats --no-kernel-events > gpurun_out/prof_run.log 2>&1
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/diag/overlap_stats.py "$t" 12
rm -rf gpurun_out/prof
