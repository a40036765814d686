#!/bin/bash
# FETCH_SIZE per launch of one kernel for several builds of the library (one rocprofv3 --pmc pass per build over tools/ab_libs.py).
# usage (GPU box): tools/diag/fetch_ab.sh "<ab_libs args>" <kernel substring> lib1.so lib2.so ...
export TMPDIR=/tmp
args=$1; kern=$2; shift 2
for lib in "$@"; do
  d=gpurun_out/fetch_ab_$(basename $lib .so)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -- python3 tools/ab_libs.py $lib $args > $d.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$kern" "$lib" <<'PY'
import csv, sys, collections
per = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
v = sorted(per.values())
print(f"{sys.argv[3]:40s} {sys.argv[2]}: {len(v)} launches, FETCH_SIZE median {v[len(v)//2]:.0f} KiB (x2 = {2*v[len(v)//2]*1024/1e6:.1f} MB read), min {v[0]:.0f} max {v[-1]:.0f}")
PY
  tail -3 $d.log | grep med
  rm -rf $d
done
