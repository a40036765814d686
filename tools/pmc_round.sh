#!/bin/bash
# PMC passes (separate runs, --kernel-trace only; never combined with sys/hip traces): HBM bytes, MFMA busy, LDS conflicts.
mkdir -p gpurun_out
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf gpurun_out/pmc_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > gpurun_out/pmc_$tag.log 2>&1
  f=$(find gpurun_out/pmc_$tag -name '*counter_collection.csv' | head -1)
  echo "== $pass -> $f"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r.get("Kernel_Name", "")
    if "anonymous namespace" not in name or "at::" in name:
        continue
    short = name.split("(anonymous namespace)::")[-1].split("(")[0]
    key = (short, r["Counter_Name"])
    agg[key][0] += 1
    agg[key][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:55s} {c:32s} launches {n:4d}  per-launch {v / n:16.1f}")
PY
  # keep only the small summaries
  find gpurun_out/pmc_$tag -type f ! -name '*counter_collection.csv' -delete
done
