#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=sceneego_amd
for env in "" "NO_RES=1" "SHAPE=32,64,64" "CL=1"; do
  env $env SCENEEGO_HIP_LIB=$PWD/$L/libse_stamp.so timeout 300 python tools/stamp_k44p.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04c_stamps.txt
timeout 900 python tools/ab_libs.py $L/libsceneego_hip_dev.so $L/libse_x1.so $L/libse_x2.so $L/libse_x4.so $L/libse_x8.so $L/libse_x16.so $L/libse_x32.so $L/libse_x15.so $L/libse_x63.so --shapes 0,3 --rounds 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c_attribution.txt
