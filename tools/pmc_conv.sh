#!/bin/bash
# PMC counters for the conv micro-benchmark (variant given as $1), kernels of interest only.
export TMPDIR=/tmp
V=${1:-0}
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_WAVE32_LDS SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmcc_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmcc_$tag -- python3 tools/bench_conv.py --variants $V --rounds 2 --only 0 > gpurun_out/pmcc_$tag.log 2>&1
  f=$(find gpurun_out/pmcc_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r.get("Kernel_Name", "")
    if "wino" not in name and "persistent" not in name and "tiled" not in name:
        continue
    short = name.split("(anonymous namespace)::")[-1].split("(")[0]
    agg[(short, r["Counter_Name"])][0] += 1
    agg[(short, r["Counter_Name"])][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:45s} {c:30s} n {n:3d}  per-launch {v / n:16.1f}")
PY
  rm -rf gpurun_out/pmcc_$tag
done
