#!/bin/bash
# One GPU-box session: full gpu test suite, bench, rocprofv3 kernel-trace stats of the bench + per-dispatch table.  Run via gpurun.
# usage: tools/gpu_round.sh [tag]   (outputs under gpurun_out/<tag>_*)
tag=${1:-round}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -25 gpurun_out/${tag}_gpu_tests.txt
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc $?"; tail -c 3000 gpurun_out/${tag}_bench.json; tail -5 gpurun_out/${tag}_bench.err
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras --streams 1 > gpurun_out/${tag}_prof_run.log 2>&1
tail -2 gpurun_out/${tag}_prof_run.log
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv && head -30 "$f"
t=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python3 tools/per_dispatch_table.py "$t" 2 > gpurun_out/${tag}_per_dispatch_table.txt && tail -30 gpurun_out/${tag}_per_dispatch_table.txt
rm -rf gpurun_out/prof
