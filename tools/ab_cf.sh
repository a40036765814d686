#!/bin/bash
# LDS writes of the next step in front of / behind the MFMAs of this one (variant builds), kernel durations by rocprofv3
export TMPDIR=/tmp
for shape in "256 1024 16 1" "64 256 64 1" "512 128 32 0" "128 512 32 1" "256 64 64 0" "512 2048 8 1"; do
  echo "== $shape"
  for l in libsceneego_hip_dev libse_cf2 libse_cf3; do
    rm -rf gpurun_out/prof
    SCENEEGO_HIP_LIB=$PWD/sceneego_amd/$l.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/diag/one_conv1x1.py $shape > /dev/null 2>&1
    f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
    python3 - "$f" "$l" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "conv1x1" in r["Name"]:
        print(f"{sys.argv[2]:22s} {float(r['AverageNs']) / 1e3:7.2f} us  {r['Name'][30:70]}")
PY
  done
done
rm -rf gpurun_out/prof
