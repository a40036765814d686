#!/bin/bash
# clock and MFMA-busy of the F(4,3) kernels under the conv micro-benchmark ($1 = variant)
export TMPDIR=/tmp
V=${1:-0}
for pass in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmcp_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmcp_$tag -- python3 tools/bench_conv.py --variants $V --rounds 3 --only 0 > gpurun_out/pmcp_$tag.log 2>&1
  f=$(find gpurun_out/pmcp_$tag -name '*counter_collection.csv' | head -1)
  k=$(find gpurun_out/pmcp_$tag -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" "$k" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    if "wino43" in n:
        agg[(n.split("::")[-1].split("(")[0], r["Counter_Name"])][0] += 1; agg[(n.split("::")[-1].split("(")[0], r["Counter_Name"])][1] += float(r["Counter_Value"])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    n = r.get("Kernel_Name", "")
    if "wino43" in n:
        dur[n.split("::")[-1].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, c), (n, v) in sorted(agg.items()):
    d = sum(dur[k]) / len(dur[k])
    extra = f"  -> {v / n / 8 / d / 1e3:.3f} GHz" if c == "GRBM_GUI_ACTIVE" else ""
    print(f"{k:30s} {c:28s} launches {n:3d} per-launch {v / n:16.1f}  avg duration {d:8.1f} us{extra}")
PY
  rm -rf gpurun_out/pmcp_$tag
done
