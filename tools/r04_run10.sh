#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=sceneego_amd
timeout 900 python tools/ab_libs.py $L/libse_v2.so $L/libse_np.so $L/libse_p1.so $L/libse_d0.so $L/libse_qa5.so $L/libse_r7.so --shapes 0,3 --rounds 12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04j_ab.txt
timeout 900 python tools/ab_libs.py $L/libse_v2.so $L/libse_np.so $L/libse_p1.so $L/libse_d0.so $L/libse_qa5.so $L/libse_r7.so --shapes 0 --rounds 12 --no-res 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04j_ab.txt
