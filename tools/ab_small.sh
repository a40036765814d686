#!/bin/bash
# the small-M 1x1 form: cap on its workgroup count (0 = off) at batch 8 / 2 / 4
export TMPDIR=/tmp
for rep in 1 2; do
  for cap in 0 512 256; do
    SCENEEGO_CONV1X1_SMALL_MAX_WG=$cap python bench.py --no-extras --no-cpu-baseline --no-repeats --steps 30 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('B=8 cap $cap', d['value'], 'single', d['extra']['single_stream']['value'], 'backbone', r['stage_ms']['backbone'])"
    for b in 2 4; do
    SCENEEGO_CONV1X1_SMALL_MAX_WG=$cap python bench.py --batch $b --streams 1 --graphs --no-extras --no-cpu-baseline --no-repeats --no-kernel-events --steps 200 --warmup 20 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=$b graph cap $cap', d['value'], 'ms', d['ms_per_step'])"
    done
  done
done
