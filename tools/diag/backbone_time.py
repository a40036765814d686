#!/usr/bin/env python3
"""Steady-state kernel time of the 2-D backbone per forward from a rocprofv3 --kernel-trace CSV of `bench.py --streams 1`: the trace is cut at
the gather kernel that follows every backbone, the last N forwards are kept, and the kernels between two gathers that are not V2V-side
launches are summed by class.  usage: backbone_time.py <kernel_trace.csv> [N]"""
import csv
import sys

rows = sorted((r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"), key=lambda r: int(r["Start_Timestamp"]))
N = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 5
V2V = ("conv3d_", "deconv3d_", "maxpool2", "pointwise_chain3", "softargmax_", "voxelize_", "splitk_reduce", "fft7_")
steps, cur = [], []
for r in rows:
    n = r["Kernel_Name"]
    if "::gather_" in n:
        steps.append(cur)
        cur = []
    elif not any(k in n for k in V2V):
        cur.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
steps = steps[-N:]
tot = {}
for s in steps:
    for n, us in s:
        k = ("conv1x1 (fused GEMM)" if ("conv1x1_kernel" in n or "conv1x1_small_kernel" in n) else "conv3x3 (direct MFMA)" if "conv3x3_kernel" in n else "stem tail" if "bias_relu_maxpool" in n else "bias_act" if "bias_act" in n else "deconv assemble" if "assemble" in n
             else "Tensile GEMM" if n.startswith("Cijk") else "MIOpen asm / igemm / CK" if ("miopen" in n.lower() or "igemm" in n or "ck" in n[:8]) else "torch / other")
        e = tot.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += us
print(f"{len(steps)} forwards; per forward:")
for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} {c / len(steps):6.1f} launches {us / len(steps):9.1f} us")
if "-v" in sys.argv:
    by = {}
    for st in steps:
        for n, us in st:
            e = by.setdefault(n[:110], [0, 0.0]); e[0] += 1; e[1] += us
    for n, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"      {c / len(steps):5.1f} x {us / c:7.1f} us  {n}")
print(f"  {'backbone kernels, total':28s} {sum(c for c, _ in tot.values()) / len(steps):6.1f} launches {sum(u for _, u in tot.values()) / len(steps):9.1f} us")
