#!/bin/bash
# batch 1 on one stream: eager and as a graph, alternating
export TMPDIR=/tmp
for rep in 1 2 3; do
  for g in "" "--graphs"; do
    python bench.py --batch 1 --streams 1 $g --no-extras --no-cpu-baseline --no-repeats --no-kernel-events --steps 300 --warmup 30 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch 1 one stream $g', d['value'], 'ms', d['ms_per_step'])"
  done
done
