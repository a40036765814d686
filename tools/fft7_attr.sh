#!/bin/bash
# attribution of the three passes of the frequency-domain front layer: per-kernel times (rocprofv3 --kernel-trace --stats) of the
# knock-out builds libse_fx<bits>.so (tools/build_variant.sh fx<bits> conv3d_fft7 -DSE_FFT7_EXP=<bits>) next to the production library
# usage: tools/fft7_attr.sh <tag> <bits...>
tag=$1; shift
export TMPDIR=/tmp; mkdir -p gpurun_out
out=gpurun_out/${tag}_attr.txt; : > $out
for v in prod "$@"; do
  lib=sceneego_amd/libsceneego_hip.so; [ "$v" != prod ] && lib=sceneego_amd/libse_fx$v.so
  rm -rf gpurun_out/prof
  SCENEEGO_HIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/bench_fft7.py --time --quick --batch 8 > gpurun_out/attr_run.log 2>&1
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  echo "== $v" >> $out
  python3 - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "fft7_" in n and "pack" not in n:
        print("%-24s calls %4s avg_us %8.1f" % (n[n.index("fft7_"):].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  grep "per call" gpurun_out/attr_run.log >> $out
done
rm -rf gpurun_out/prof
cat $out
