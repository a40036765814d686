#!/bin/bash
# the direct 3x3 kernel also on the wide maps of layer1 / layer2 (SCENEEGO_CONV3X3_MAX_PIXELS=4096) against MIOpen there (256): B=8 and B=1
export TMPDIR=/tmp
for rep in 1 2; do
  for mp in 256 4096; do
    SCENEEGO_CONV3X3_MAX_PIXELS=$mp python bench.py --no-extras --no-cpu-baseline --no-repeats --steps 30 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('B=8 max pixels $mp', d['value'], 'single', d['extra']['single_stream']['value'], 'backbone', r['stage_ms']['backbone'], 'parity', d['parity']['max_joint_err_m'])"
    SCENEEGO_CONV3X3_MAX_PIXELS=$mp python bench.py --batch 1 --streams 1 --graphs --no-extras --no-cpu-baseline --no-repeats --no-kernel-events --steps 200 --warmup 20 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=1 graph max pixels $mp', d['value'], 'ms', d['ms_per_step'])"
  done
done
