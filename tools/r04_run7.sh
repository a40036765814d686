#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/r04g_gpu_tests.txt 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r04g_gpu_tests.txt
python bench.py > gpurun_out/r04g_bench.json 2> gpurun_out/r04g_bench.err; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04g_bench.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['value'], d['single_stream_value'], d['step_ms'], d['repeat_values']['values'], d['parity']['max_joint_err_m'], d['parity']['max_joint_err_vs_f32_softargmax_m'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['stage'], d['roofline']['stage_ms'])
    for k,v in d['extra'].items(): print(k, {a:b for a,b in v.items() if a in ('value','ms_per_step','hipgraph','pipelined','error')})
PY
tail -3 gpurun_out/r04g_bench.err
