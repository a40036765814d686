#!/bin/bash
# clock and MFMA-busy of the 7^3 Winograd kernels under the conv micro-benchmark ($1 = variant: 0 = F(4,7), 17 = F(2,7))
export TMPDIR=/tmp
V=${1:-0}
for pass in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmck_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmck_$tag -- python3 tools/bench_conv.py --variants $V --rounds 3 --only 2 > gpurun_out/pmck_$tag.log 2>&1
  f=$(find gpurun_out/pmck_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    if "k7_wino" in n:
        k = n.split("::")[-1].split("(")[0]
        agg[(k, r["Counter_Name"])][0] += 1; agg[(k, r["Counter_Name"])][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:28s} {c:28s} launches {n:3d} per-launch {v / n:16.1f}")
PY
  rm -rf gpurun_out/pmck_$tag
done
