#!/bin/bash
# BASELINE config 3: rocprofv3 PMC counters (separate passes) on the bf16 V2V conv kernels via the conv micro-benchmark:
# HBM bytes (FETCH_SIZE / WRITE_SIZE, KiB), MFMA busy / CU busy cycles, LDS bank conflicts.  $1 = index into SHAPES (0: 3^3
# 32->32 @64^3, 2: 7^3 front layer).
export TMPDIR=/tmp
IDX=${1:-0}
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmcb_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmcb_$tag -- python3 tools/bench_conv.py --bf16 --variants 0 --rounds 3 --only $IDX > gpurun_out/pmcb_$tag.log 2>&1
  f=$(find gpurun_out/pmcb_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    if "conv_bf16_k" in n:
        short = n.split("(anonymous namespace)::")[-1].split("(")[0]
        agg[(short, r["Counter_Name"])][0] += 1
        agg[(short, r["Counter_Name"])][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:28s} B=8 (bench_conv --bf16): {c:28s} launches {n:3d} per-launch {v / n:16.1f}")
PY
  rm -rf gpurun_out/pmcb_$tag
done
