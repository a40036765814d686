#!/bin/bash
# BASELINE configs[2] (bf16 storage, B=32) through bench.py under rocprofv3 --pmc, one counter group per run (--kernel-trace only).
export TMPDIR=/tmp
mkdir -p gpurun_out
agg() {
python3 - "$1" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    if "anonymous namespace" in n or "conv_bf16" in n or "bf16" in n:
        short = n.split("(anonymous namespace)::")[-1].split("(")[0][:56]
        k = (short, r["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"   {k:56s} {c:28s} launches {n:4d} per-launch {v / n:16.1f}")
PY
}
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  echo "== $pass (bench.py --batch 32 --v2v-dtype bf16 --backbone-dtype bf16 --steps 2 --warmup 1)"
  rm -rf gpurun_out/p3_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/p3_$tag -- python3 bench.py --batch 32 --v2v-dtype bf16 --backbone-dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extras > gpurun_out/p3_$tag.log 2>&1
  tail -1 gpurun_out/p3_$tag.log | cut -c1-200
  f=$(find gpurun_out/p3_$tag -name '*counter_collection.csv' | head -1); [ -n "$f" ] && agg "$f"
  rm -rf gpurun_out/p3_$tag
done
