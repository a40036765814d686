#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=sceneego_amd
timeout 900 python tools/ab_libs.py $L/libse_base.so $L/libse_ep2.so $L/libse_ep3.so $L/libse_pk.so $L/libse_pkep.so --shapes 0,1,3 --rounds 12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04e_ab.txt
timeout 900 python tools/ab_libs.py $L/libse_base.so $L/libse_ep2.so $L/libse_ep3.so $L/libse_pk.so $L/libse_pkep.so --shapes 0,3 --rounds 12 --no-res 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04e_ab.txt
