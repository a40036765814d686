#!/bin/bash
# batch-1 A/B of the fused 1x1 convolutions' routing rule: one stream, eager and as a captured graph
export TMPDIR=/tmp
run() {
  python bench.py --batch 1 --streams 1 $2 --no-extras --no-cpu-baseline --no-repeats --no-kernel-events --steps 200 --warmup 20 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2', d['value'], 'ms', d['ms_per_step'], 'parity', d['parity']['max_joint_err_m'])"
}
for rep in 1 2; do
  for g in "" "--graphs"; do
    SCENEEGO_CONV1X1=0 run "off            " "$g"
    SCENEEGO_CONV1X1_MIN_WG=0 run "minwg0 cin<=512" "$g"
    SCENEEGO_CONV1X1_MIN_WG=0 SCENEEGO_CONV1X1_MAX_CIN=256 run "minwg0 cin<=256" "$g"
    SCENEEGO_CONV1X1_MIN_WG=64 run "minwg64        " "$g"
  done
done
