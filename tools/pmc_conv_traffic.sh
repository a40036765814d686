#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant conv shape via the conv micro-benchmark.
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmct_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmct_$tag -- python3 tools/bench_conv.py --variants 0 --rounds 3 --only 0 > gpurun_out/pmct_$tag.log 2>&1
  f=$(find gpurun_out/pmct_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    if "wino43" in n:
        agg[r["Counter_Name"]][0] += 1; agg[r["Counter_Name"]][1] += float(r["Counter_Value"])
for c, (n, v) in sorted(agg.items()):
    print(f"conv3d_k3_wino43_kernel 32->32@64^3 B=8 (with residual): {c:28s} launches {n:3d} per-launch {v / n:16.1f}")
PY
  rm -rf gpurun_out/pmct_$tag
done
