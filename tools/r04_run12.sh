#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extras --no-repeats --streams 1 > gpurun_out/r04l_prof_b1.log 2>&1
t=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python3 tools/per_dispatch_table.py "$t" 2 > gpurun_out/r04l_b1_per_dispatch_table.txt && cat gpurun_out/r04l_b1_per_dispatch_table.txt | sed -n 1,100p
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04l_b1_kernel_stats.csv
rm -rf gpurun_out/prof
