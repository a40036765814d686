#!/bin/bash
# One GPU-box session: full gpu test suite, bench, rocprofv3 kernel-trace stats of the bench + per-dispatch tables (B=8 and B=1), and
# (with a third argument) the per-dispatch counter passes.  Run via gpurun.
# usage: tools/gpu_round.sh [tag] [pmc]   (outputs under gpurun_out/<tag>_*)
tag=${1:-round}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -6 gpurun_out/${tag}_gpu_tests.txt
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc $?"; tail -c 1500 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err
for b in 8 1; do
  rm -rf gpurun_out/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --batch $b --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras --no-repeats --streams 1 > gpurun_out/${tag}_prof_run_b$b.log 2>&1
  tail -1 gpurun_out/${tag}_prof_run_b$b.log | cut -c1-300
  f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_b${b}_kernel_stats.csv
  t=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python3 tools/per_dispatch_table.py "$t" 2 > gpurun_out/${tag}_b${b}_per_dispatch_table.txt && tail -28 gpurun_out/${tag}_b${b}_per_dispatch_table.txt
  rm -rf gpurun_out/prof
done
if [ -n "$2" ]; then python3 tools/pmc_round.py 8 64 > gpurun_out/${tag}_pmc_round.log 2>&1; tail -3 gpurun_out/${tag}_pmc_round.log | cut -c1-600; fi
