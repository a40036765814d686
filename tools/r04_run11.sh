#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=sceneego_amd
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04k_tests.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04k_tests.txt
timeout 900 python tools/ab_libs.py $L/libse_rs148.so $L/libse_rs132.so $L/libse_rs156.so $L/libse_rs164.so $L/libse_rs180.so --shapes 0,3 --rounds 12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04k_ab.txt
python bench.py --no-cpu-baseline > gpurun_out/r04k_bench.json 2> gpurun_out/r04k_bench.err; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04k_bench.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['value'], d['single_stream_value'], d['step_ms'], d['parity']['max_joint_err_m'], d['roofline']['avg_launch_ms'], d['roofline']['stage_ms'])
    for k,v in d['extra'].items(): print(k, {a:b for a,b in v.items() if a in ('value','ms_per_step','hipgraph','pipelined','error')})
PY
tail -3 gpurun_out/r04k_bench.err
