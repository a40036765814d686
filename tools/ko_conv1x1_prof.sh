#!/bin/bash
# kernel durations (rocprofv3) of the knock-out builds on one shape: usage ko_conv1x1_prof.sh cin cout H residual
export TMPDIR=/tmp
for l in libsceneego_hip_dev libse_ko1 libse_ko2 libse_ko4 libse_ko5 libse_ko7 libse_ko8 libse_ko16 libse_ko24; do
  rm -rf gpurun_out/prof
  SCENEEGO_HIP_LIB=$PWD/sceneego_amd/$l.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/diag/one_conv1x1.py $1 $2 $3 $4 > /dev/null 2>&1
  f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$l" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "conv1x1" in r["Name"]:
        print(f"{sys.argv[2]:22s} {float(r['AverageNs']) / 1e3:7.2f} us  ({r['Calls']} calls)  {r['Name'][:60]}")
PY
done
rm -rf gpurun_out/prof
