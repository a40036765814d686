#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the 7^3 F(4,7) kernel as it runs inside bench.py (planar input).
export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf gpurun_out/pmck7t_$pass
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmck7t_$pass -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > gpurun_out/pmck7t_$pass.log 2>&1
  f=$(find gpurun_out/pmck7t_$pass -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    for key in ("wino47p3", "gather_planar3", "wino43pp"):
        if key in n:
            agg[(key, r["Counter_Name"])][0] += 1; agg[(key, r["Counter_Name"])][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:16s} {c:12s} launches {n:3d} per-launch {v / n:16.1f} KB")
PY
  rm -rf gpurun_out/pmck7t_$pass
done
