#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=sceneego_amd
timeout 900 python tools/ab_libs.py $L/libse_pk.so $L/libse_pkt.so --shapes 0,1,3,6 --rounds 12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04f_ab.txt
timeout 900 python tools/ab_libs.py $L/libse_pk.so $L/libse_pkt.so --shapes 0,3 --rounds 12 --no-res 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04f_ab.txt
