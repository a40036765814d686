#!/bin/bash
# headline at 1..5 streams (B=8, eager)
export TMPDIR=/tmp
for n in 1 2 3 4 5; do
  python bench.py --streams $n --no-extras --no-cpu-baseline --no-repeats --no-kernel-events --steps 30 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $n', d['value'], 'ms', d['ms_per_step'])"
done
