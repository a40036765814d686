#!/bin/bash
# Round-2 PMC passes (separate runs; --kernel-trace only).  1) FETCH_SIZE / WRITE_SIZE calibration on copies of known size,
# 2) HBM bytes, MFMA busy, LDS conflicts of the 2-D Winograd 3x3x3 kernel at its headline shape (32->32 @64^3, B=8, with residual),
#    channels-last and octet-planar input, 3) per-kernel aggregates over one bench step.
export TMPDIR=/tmp
mkdir -p gpurun_out
agg() {  # $1 csv, $2 kernel substring
python3 - "$1" "$2" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    if sys.argv[2] in n:
        short = n.split("(anonymous namespace)::")[-1].split("(")[0][:60]
        k = (short, r["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print(f"   {k:60s} {c:30s} launches {n:3d} per-launch {v / n:18.1f}")
PY
}
[ -x tools/diag/copy_calib ] || hipcc -O3 --offload-arch=gfx950 tools/diag/copy_calib.hip -o tools/diag/copy_calib
echo "== calibration (tools/diag/copy_calib: 1 GiB read + 1 GiB written per copy launch; gather reads 32 of every 128 B)"
for pass in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pc_$pass
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pc_$pass -- ./tools/diag/copy_calib > gpurun_out/pc_$pass.log 2>&1
  f=$(find gpurun_out/pc_$pass -name '*counter_collection.csv' | head -1); [ -n "$f" ] && agg "$f" "copy" && agg "$f" "gather"
  rm -rf gpurun_out/pc_$pass
done
for oct in 0 1; do
echo "== conv3d_k3_wino2d_kernel 32->32 @64^3 B=8 with residual, input layout octet=$oct (tools/bench_conv.py --only 0)"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pk_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pk_$tag -- python3 tools/bench_conv.py --variants 0 --rounds 3 --only 0 --octet $oct > gpurun_out/pk_$tag.log 2>&1
  f=$(find gpurun_out/pk_$tag -name '*counter_collection.csv' | head -1); [ -n "$f" ] && agg "$f" "wino2d"
  rm -rf gpurun_out/pk_$tag
done
done
echo "== 7^3 front layer (conv3d_k7_wino47) and whole step: bench.py, per kernel"
for pass in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pb_$pass
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pb_$pass -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extras > gpurun_out/pb_$pass.log 2>&1
  f=$(find gpurun_out/pb_$pass -name '*counter_collection.csv' | head -1); [ -n "$f" ] && agg "$f" "anonymous namespace"
  rm -rf gpurun_out/pb_$pass
done
