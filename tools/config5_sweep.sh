#!/bin/bash
# BASELINE configs[4]: 128^3 grid, batch 8 - kernel-form / LDS-tile sweep of the 3x3x3 convolution 32 -> 32 with the CURRENT kernel set
# (development build: se_debug_set_variant selects the form), float32 and bf16 legs, plus the 7^3 front layer in both forms.
# usage (GPU box): tools/config5_sweep.sh > gpurun_out/r06_config5_sweep_raw.txt
export TMPDIR=/tmp SCENEEGO_HIP_LIB=$PWD/sceneego_amd/libsceneego_hip_dev.so
echo "== f32, quad-planar in / out / skip (production layout at 128^3): variant 0 = production dispatch (F(4,3) x F(4,3) ping-pong)"
python tools/bench_conv.py --only 6 --variants 0 --quad 7 --rounds 5 2>&1 | grep "^k"
echo "== f32, channels-last: 0 = production dispatch (F(4,3) x F(2,3)), 30 = 1-D F(4,3), 4 = 1-D F(2,3), 2 = direct persistent, 21 / 22 / 23 = direct LDS tiles 8x8x4 / 8x8x8 / 8x8x16"
python tools/bench_conv.py --only 6 --variants 0,30,4,2,21,22,23 --rounds 5 2>&1 | grep "^k"
echo "== f32, octet-planar in / out: 0 = F(4,3) x F(2,3)"
python tools/bench_conv.py --only 6 --variants 0 --octet 3 --rounds 5 2>&1 | grep "^k"
echo "== bf16 storage: 0 = production (tile 8(y) rows, 2 workgroups per CU), 4 = tile 4x4x16 (3 per CU), 3 = persistent form"
python tools/bench_conv.py --only 6 --variants 0,4,3 --bf16 --rounds 5 2>&1 | grep "^k"
echo "== 7^3 front layer at 128^3: frequency-domain form vs F(6,7) Winograd"
python tools/bench_fft7.py --time --batch 8 --dim 128 --chunk 2 2>&1 | grep -E "fft7|wino67"
