"""Per-dispatch table of the HIP kernels of one V2V forward from a rocprofv3 --kernel-trace CSV.

rocprofv3's --stats averages a kernel template over all shapes it ran on (the 2-D Winograd kernel runs on 22 layers of 7 shapes per
step); this splits the trace into steps at the 7^3 front-layer kernel (or pass 1 of its frequency-domain form) and prints every dispatch POSITION inside a step with its
mean / min / max duration over the steps.  The positions follow V2VModel's layer order (sceneego_amd/v2v.py: front 7^3, front_res,
encoder 32^3 / 16^3 / 8^3 / 4^3 / 2^3, decoder back up, back_res, tail), so a position identifies the layer and its shape.

usage: python tools/per_dispatch_table.py <kernel_trace.csv> [steps-to-skip]
"""
import collections
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
OURS = ("conv3d_", "deconv3d_", "maxpool2", "pointwise_chain3", "softargmax_", "::gather_", "voxelize_", "splitk_reduce", "fft7_")
FRONT = ("conv3d_k7", "fft7_fwd")       # the 7^3 front layer: Winograd kernel, or pass 1 of the frequency-domain form (round 6)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:44]


steps, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "::gather_" in n or "voxelize_kernel" in n:      # first HIP kernel of a forward
        if cur is None or any(any(f in x[0] for f in FRONT) for x in cur):
            cur = []
            steps.append(cur)
    if cur is not None and any(k in n for k in OURS) and "pack" not in n:
        cur.append((short(n), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
                    (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
steps = [s for s in steps if any(any(f in x[0] for f in FRONT) for x in s)][skip:]
n = min(len(s) for s in steps)
print(f"{len(steps)} steps, {n} dispatches of this library per step (MIOpen / Tensile / torch kernels of the 2-D backbone left out)")
print(f"{'pos':>3s}  {'kernel':44s} {'workgroups (x,y,z)':>20s} {'mean us':>9s} {'min':>8s} {'max':>8s}")
tot = collections.defaultdict(float)
for i in range(n):
    d = [s[i][4] for s in steps]
    k = steps[0][i]
    print(f"{i:3d}  {k[0]:44s} {str((k[1], k[2], k[3])):>20s} {sum(d) / len(d):9.1f} {min(d):8.1f} {max(d):8.1f}")
    tot[k[0]] += sum(d) / len(d)
print("per kernel, us per step:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"   {k:44s} {v:9.1f}")
print(f"   {'total':44s} {sum(tot.values()):9.1f}")
