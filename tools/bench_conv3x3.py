#!/usr/bin/env python3
"""The backbone's stride-1 3x3 convolutions: se_conv2d_3x3_f32 (direct float32 MFMA product, raw sums) against MIOpen's convolution, per
distinct shape of the ResNet-50 pose backbone (reference network/pose_resnet.py:52-90), with the max |difference| of the two results.
usage: python tools/bench_conv3x3.py [--batch 8]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sceneego_amd import _lib      # noqa: E402

# (channels, H, count per forward)
SHAPES = [(64, 64, 3), (128, 32, 3), (256, 16, 5), (512, 8, 2)]


def timeit(fn, n=400, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    B, dev = a.batch, "cuda:0"
    lib = _lib.load()
    for c, H, count in SHAPES:
        x = torch.randn(B, c, H, H, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5
        old = lambda: F.conv2d(x, w, padding=1)
        t_old = timeit(old)
        tile = _lib.conv2d_3x3_tile(B, c, c, H, H)
        if not tile:
            print(f"{c:4d}->{c:4d} @{H:2d}^2: not covered; MIOpen {t_old:7.1f} us")
            continue
        wp = _lib.conv2d_3x3_pack(w, tile)
        new = lambda: _lib.conv2d_3x3(x, wp, None, False)
        t_new = timeit(new)
        extra = ""
        if hasattr(lib, "se_debug_set_variant"):
            lib.se_debug_set_variant(76)
            extra = f"   [one wave group {timeit(new):6.1f}]"
            lib.se_debug_set_variant(0)
        diff = float((new() - old()).abs().max())
        flop = 2.0 * 9 * B * H * H * c * c
        print(f"{c:4d}->{c:4d} @{H:2d}^2 x{count}: direct MFMA {t_new:7.1f} us ({flop / t_new / 1e6:6.1f} TF/s, tile {tile}, "
              f"{(B * H * H // 64) * (c // tile)} workgroups)   MIOpen {t_old:7.1f} us (all its launches)   maxdiff {diff:.2e}" + extra, flush=True)


if __name__ == "__main__":
    main()
