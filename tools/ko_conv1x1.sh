#!/bin/bash
# knock-out builds of the 1x1 GEMM (scratch copies of the source with -DC1_KO=bits: 1 no loads in the k loop, 2 no MFMA phase, 4 no LDS writes,
# 8 no output stores, 16 no residual read), per backbone shape
for l in libsceneego_hip_dev libse_ko1 libse_ko2 libse_ko4 libse_ko5 libse_ko7 libse_ko8 libse_ko16 libse_ko24; do
  echo "== $l"
  SCENEEGO_HIP_LIB=$PWD/sceneego_amd/$l.so python tools/bench_conv1x1.py 2>&1 | grep "64->  256 @64^2 x3\|128->  512\|256-> 1024\|512-> 2048\|256->   64\|512->  128" | cut -c1-48
done
