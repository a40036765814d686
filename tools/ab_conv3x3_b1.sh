#!/bin/bash
# batch-1 A/B of the direct 3x3 convolution's routing rule: one stream, as a captured graph
export TMPDIR=/tmp
run() {
  python bench.py --batch ${B:-1} --streams 1 --graphs --no-extras --no-cpu-baseline --no-repeats --no-kernel-events --steps 200 --warmup 20 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity']['max_joint_err_m'])"
}
for rep in 1 2; do
  SCENEEGO_CONV3X3=0 run "3x3 off      "
  SCENEEGO_CONV3X3_MIN_WG=0 run "3x3 minwg 0  "
  SCENEEGO_CONV3X3_MIN_WG=128 run "3x3 minwg 128"
  run "default      "
done
