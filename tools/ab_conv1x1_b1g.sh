#!/bin/bash
export TMPDIR=/tmp
run() {
  python bench.py --batch ${B:-1} --streams 1 --graphs --no-extras --no-cpu-baseline --no-repeats --no-kernel-events --steps 200 --warmup 20 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=${B:-1} graph $1', d['value'], 'ms', d['ms_per_step'])"
}
for rep in 1 2; do
  run "default (1x1 min wg 256)"
  SCENEEGO_CONV1X1_MIN_WG=128 run "1x1 min wg 128"
  SCENEEGO_CONV1X1_MIN_WG=64 run "1x1 min wg 64"
  SCENEEGO_CONV1X1_MIN_WG=0 run "1x1 min wg 0"
done
