#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
L=sceneego_amd
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3d" > gpurun_out/r04i_tests.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r04i_tests.txt
timeout 900 python tools/ab_libs.py $L/libse_v1.so $L/libse_v2.so --shapes 0,1,3,6 --rounds 12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04i_ab.txt
timeout 900 python tools/ab_libs.py $L/libse_v1.so $L/libse_v2.so --shapes 0,3 --rounds 12 --no-res 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04i_ab.txt
for env in "" "NO_RES=1"; do env $env SCENEEGO_HIP_LIB=$PWD/$L/libse_v2stamp.so timeout 300 python tools/stamp_k44p.py 2>&1 | grep -v amdgpu.ids | grep -v "^wave [1235679]"; done | tee gpurun_out/r04i_stamps.txt
