#!/bin/bash
# LDS bank-conflict attribution of conv3d_k3_wino2d_kernel (32->32 @64^3, B=8, octet-planar in/out): SQ_LDS_BANK_CONFLICT /
# SQ_LDS_IDX_ACTIVE per launch for the production kernel and the attribution variants of the development build
# (51 no weight LDS writes, 52 no V-tile LDS writes, 48 no MFMA stream = no operand reads).  Run on the GPU box.
export TMPDIR=/tmp
export SCENEEGO_HIP_LIB=$PWD/sceneego_amd/libsceneego_hip_dev.so
mkdir -p gpurun_out
for v in 0 51 52 48; do
  rm -rf gpurun_out/ldsattr_$v
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/ldsattr_$v -- python3 tools/bench_conv.py --variants $v --rounds 3 --only 0 --octet 3 > gpurun_out/ldsattr_$v.log 2>&1
  f=$(find gpurun_out/ldsattr_$v -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" $v <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "wino2d" in r["Kernel_Name"]:
        agg[r["Counter_Name"]][0] += 1; agg[r["Counter_Name"]][1] += float(r["Counter_Value"])
print("variant", sys.argv[2], {k: round(v[1] / v[0]) for k, v in agg.items()}, "per launch;  per step and CU:",
      {k: round(v[1] / v[0] / 256 / 64) for k, v in agg.items()})
PY
  rm -rf gpurun_out/ldsattr_$v
done
