#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "nan_behind or front_layer_64 or k3_64 or k3_128" > gpurun_out/r04h_tests.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04h_tests.txt
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extras --no-repeats --streams 1 > gpurun_out/r04h_prof_b1.log 2>&1
tail -2 gpurun_out/r04h_prof_b1.log | cut -c1-400
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04h_b1_kernel_stats.csv && head -45 "$f" | cut -c1-200
t=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python3 tools/per_dispatch_table.py "$t" 2 > gpurun_out/r04h_b1_per_dispatch_table.txt && cat gpurun_out/r04h_b1_per_dispatch_table.txt
rm -rf gpurun_out/prof
