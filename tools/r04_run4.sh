#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=sceneego_amd
for env in "" "SHAPE=32,64,64"; do
  env $env SCENEEGO_HIP_LIB=$PWD/$L/libse_stamp4.so timeout 300 python tools/stamp_k44p.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04d_stamps.txt
timeout 900 python tools/ab_libs.py $L/libse_qa5.so $L/libse_qa4.so $L/libse_qa4k0.so --shapes 0,1,3,6 --rounds 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04d_ab.txt
timeout 900 python tools/ab_libs.py $L/libse_qa5.so $L/libse_qa4.so $L/libse_qa4k0.so --shapes 0,3 --rounds 10 --no-res 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04d_ab.txt
