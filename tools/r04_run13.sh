#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04m_tests.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r04m_tests.txt
python bench.py --no-cpu-baseline > gpurun_out/r04m_bench.json 2> gpurun_out/r04m_bench.err; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r04m_bench.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['value'], d['single_stream_value'], d['step_ms'], d['parity']['max_joint_err_m'], d['roofline']['avg_launch_ms'], d['roofline']['stage_ms'])
    for k,v in d['extra'].items(): print(k, {a:b for a,b in v.items() if a in ('value','ms_per_step','hipgraph','pipelined','error')})
PY
tail -3 gpurun_out/r04m_bench.err
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extras --no-repeats --streams 1 > gpurun_out/r04m_prof_b1.log 2>&1
t=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python3 tools/per_dispatch_table.py "$t" 2 > gpurun_out/r04m_b1_per_dispatch_table.txt && tail -28 gpurun_out/r04m_b1_per_dispatch_table.txt
rm -rf gpurun_out/prof
