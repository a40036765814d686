#!/bin/bash
# whole-forward A/B of the fused 1x1 convolutions (SCENEEGO_CONV1X1=1 / 0): headline, single stream, stage times, parity
export TMPDIR=/tmp
for v in 1 0 1 0; do
  env SCENEEGO_${KNOB:-CONV1X1}=$v python bench.py --no-extras --no-cpu-baseline --no-repeats --steps 20 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('SCENEEGO_${KNOB:-CONV1X1}=$v', d['value'], 'single', d['extra']['single_stream']['value'], r['stage_ms'], 'parity', d['parity']['max_joint_err_m'])"
done
