#!/bin/bash
# one GPU-box session for the frequency-domain front layer: check, HIP-event time, per-pass kernel stats (usage: tools/fft7_gpu.sh <tag> [batch])
tag=${1:-fft7}; b=${2:-8}
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/bench_fft7.py --check > gpurun_out/${tag}_check.txt 2>&1; echo "check rc $?"; grep -c "max|fft" gpurun_out/${tag}_check.txt; tail -2 gpurun_out/${tag}_check.txt
timeout 300 python tools/bench_fft7.py --time --batch $b > gpurun_out/${tag}_time.txt 2>&1; grep -v amdgpu.ids gpurun_out/${tag}_time.txt
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/bench_fft7.py --time --batch $b > gpurun_out/${tag}_prof.log 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${tag}_kernel_stats.csv
grep -E "fft7_(fwd|gemm|inv)|wino67" gpurun_out/${tag}_kernel_stats.csv | awk -F'","' '{print $1, $2, $4}' | cut -c1-160
rm -rf gpurun_out/prof
