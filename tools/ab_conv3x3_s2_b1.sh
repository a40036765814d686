#!/bin/bash
export TMPDIR=/tmp
run() {
  python bench.py --batch ${B:-1} --streams 1 --graphs --no-extras --no-cpu-baseline --no-repeats --no-kernel-events --steps 200 --warmup 20 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=${B:-1} graph $1', d['value'], 'ms', d['ms_per_step'])"
}
for rep in 1 2; do
  run "default"
  SCENEEGO_CONV3X3_S2_MAX_PIXELS=0 run "stride-2 3x3 on MIOpen"
  SCENEEGO_CONV3X3_S2_MAX_PIXELS=256 run "stride-2 3x3: only <= 256 output pixels direct"
done
