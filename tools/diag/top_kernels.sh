export TMPDIR=/tmp
rm -rf gpurun_out/prof
SCENEEGO_CONV1X1=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras --no-repeats --streams 1 > gpurun_out/prof_run.log 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-200
rm -rf gpurun_out/prof
