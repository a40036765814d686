"""Which torch operator launches which device kernel in one backbone forward (37 copyBuffer + 10
batched_transpose + 9 direct_copy + 2 scatter_gather launches per step that are not convolutions).

  python tools/diag/backbone_glue.py [--batch 8]

torch.profiler (kineto / roctracer) attributes every kernel to the CPU operator that issued it; kernels launched through the C
ABI (ctypes) show up without a parent operator and are listed as "<C ABI>".
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    from torch.profiler import ProfilerActivity, profile
    from sceneego_amd import _lib, pose_resnet
    _lib.load()
    dev = "cuda:0"
    net = pose_resnet.get_pose_net(None).to(dev).eval()
    fb = pose_resnet.FoldedBackbone(net)
    x = torch.randn(args.batch, 3, 256, 256, device=dev)
    with torch.no_grad():
        for _ in range(3):
            fb(x)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            fb(x)
            torch.cuda.synchronize()
    per_op = collections.OrderedDict()
    n_k = 0
    for e in prof.events():
        ks = getattr(e, "kernels", None) or []
        if not ks:
            continue
        # only leaf attribution: an event whose children also carry the kernels would double count
        if any(getattr(c, "kernels", None) for c in (e.cpu_children or [])):
            continue
        key = (e.name, str(e.input_shapes)[:110])
        d = per_op.setdefault(key, collections.Counter())
        for k in ks:
            d[(k.name[:70])] += 1
            d[("__us__", k.name[:70])] += k.duration
            n_k += 1
    print(f"{n_k} kernels attributed to {len(per_op)} (operator, shapes) pairs")
    agg = collections.Counter()
    agg_us = collections.Counter()
    for (op, shp), d in per_op.items():
        names = [k for k in d if not (isinstance(k, tuple) and k[0] == "__us__")]
        for n in names:
            agg[(op, n)] += d[n]
            agg_us[(op, n)] += d[("__us__", n)]
    print(f"{'operator':34s} {'kernel':72s} {'launches':>8s} {'us':>9s}")
    for (op, n), c in sorted(agg.items(), key=lambda kv: -agg_us[kv[0]]):
        print(f"{op[:34]:34s} {n:72s} {c:8d} {agg_us[(op, n)]:9.1f}")
    print("\nnon-convolution kernels by (operator, input shapes):")
    for (op, shp), d in per_op.items():
        names = [k for k in d if not (isinstance(k, tuple) and k[0] == "__us__")]
        if any(("copy" in n.lower() or "transpose" in n.lower() or "gather" in n.lower() or "elementwise" in n.lower()) for n in names):
            print(f"  {op[:40]:40s} {shp}")
            for n in names:
                print(f"      {d[n]:3d} x {n}  ({d[('__us__', n)]:.1f} us)")


if __name__ == "__main__":
    main()
