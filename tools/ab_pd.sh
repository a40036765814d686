#!/bin/bash
# prefetch distance of the unrolled 1x1 GEMM: 2 (production), 3, 4 (variant builds)
for l in libsceneego_hip_dev libse_pd3 libse_pd4; do
  echo "== $l"
  SCENEEGO_HIP_LIB=$PWD/sceneego_amd/$l.so python tools/bench_conv1x1.py 2>&1 | tail -14 | cut -c1-52,118-240
done
