tag=$1
mkdir -p gpurun_out; export TMPDIR=/tmp
for b in 8 1; do
  rm -rf gpurun_out/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --batch $b --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras --no-repeats --streams 1 > gpurun_out/${tag}_prof_run_b$b.log 2>&1
  tail -1 gpurun_out/${tag}_prof_run_b$b.log | cut -c1-200
  f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${tag}_b${b}_kernel_stats.csv
  t=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python3 tools/per_dispatch_table.py "$t" 2 > gpurun_out/${tag}_b${b}_per_dispatch_table.txt && awk 'NR>2 && $NF+0 > 3*$(NF-2)+50' gpurun_out/${tag}_b${b}_per_dispatch_table.txt | head -12
  rm -rf gpurun_out/prof
done
