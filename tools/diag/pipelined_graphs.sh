#!/bin/bash
# headline mode (three streams) eager vs graph replay, B=8
export TMPDIR=/tmp
for rep in 1 2; do
  for g in "" "--graphs"; do
    python bench.py $g --no-extras --no-cpu-baseline --no-repeats --no-kernel-events --steps 40 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('three streams $g', d['value'], 'ms', d['ms_per_step'], 'single', d.get('single_stream_value'))"
  done
done
