#!/bin/bash
# One GPU-box session: full gpu test suite, bench, rocprofv3 kernel-trace stats of the bench.  Run via gpurun.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1
tail -25 gpurun_out/gpu_tests.log
python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
tail -2 gpurun_out/bench.log
export TMPDIR=/tmp
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --streams 1 > gpurun_out/prof_run.log 2>&1
tail -2 gpurun_out/prof_run.log
find gpurun_out/prof -name '*kernel_stats*' | head
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && head -30 "$f"
# keep the merged-back volume small
find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete
