#!/bin/bash
# round 4, GPU session 2: first run of the F(4,3) x F(4,3) ping-pong kernel: parity, then A/B timing against the F(4,3) x F(2,3) kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3d or conv7" > gpurun_out/r04b_kernel_tests.txt 2>&1; echo "pytest rc $?"
tail -15 gpurun_out/r04b_kernel_tests.txt
export SCENEEGO_HIP_LIB=$PWD/sceneego_amd/libsceneego_hip_dev.so
for o in 3 0; do
  echo "== octet $o, with skip tensor"; timeout 300 python tools/bench_conv.py --variants 0,64 --octet $o --only 0
  timeout 300 python tools/bench_conv.py --variants 0,64 --octet $o --only 1
  timeout 300 python tools/bench_conv.py --variants 0,64 --octet $o --only 3
  timeout 300 python tools/bench_conv.py --variants 0,64 --octet $o --only 4
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04b_bench_conv.txt
echo "== octet 3, no skip tensor"; for i in 0 3 6; do timeout 300 python tools/bench_conv.py --variants 0,64 --octet 3 --only $i --no-res; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04b_bench_conv.txt
unset SCENEEGO_HIP_LIB
timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04b_bench.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['value'], d['single_stream_value'], d['parity'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['stage'], d['roofline']['stage_ms'])
PY
tail -3 gpurun_out/r04b_bench.err
