#!/usr/bin/env python3
"""The backbone's stride-1 1x1 convolutions: se_conv2d_1x1_f32 (one MFMA GEMM with bias / residual / ReLU fused) against MIOpen's
convolution + se_bias_act_nchw_f32, per distinct shape of the ResNet-50 pose backbone (reference network/pose_resnet.py:52-90), with the
max |difference| of the two results.  usage: python tools/bench_conv1x1.py [--batch 8]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sceneego_amd import _lib      # noqa: E402

# (cin, cout, H, residual, count per forward)
SHAPES = [(64, 64, 64, False, 1), (64, 256, 64, True, 3), (64, 256, 64, False, 1), (256, 64, 64, False, 2), (256, 128, 64, False, 1),
          (128, 512, 32, True, 4), (512, 128, 32, False, 3), (512, 256, 32, False, 1), (256, 1024, 16, True, 6), (1024, 256, 16, False, 5),
          (1024, 512, 16, False, 1), (512, 2048, 8, True, 3), (2048, 512, 8, False, 2)]


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
    tot_new = tot_old = 0.0
    for cin, cout, H, has_res, count in SHAPES:
        x = torch.randn(B, cin, H, H, device=dev)
        w = torch.randn(cout, cin, 1, 1, device=dev) * (2.0 / cin) ** 0.5
        b = torch.randn(cout, device=dev)
        res = torch.randn(B, cout, H, H, device=dev) if has_res else None
        old = lambda: _lib.bias_act_nchw(F.conv2d(x, w), b, res, True)
        t_old = timeit(old)
        tile = _lib.conv2d_1x1_tile(B, cin, cout, H * H)
        if not tile:
            print(f"{cin:5d}->{cout:5d} @{H:2d}^2: not covered; MIOpen + epilogue {t_old:7.1f} us")
            continue
        wp = _lib.conv2d_1x1_pack(w.reshape(cout, cin), tile)
        new = lambda: _lib.conv2d_1x1(x, wp, b, res, True)
        t_new = timeit(new)
        extra = ""
        lib = _lib.load()
        if hasattr(lib, "se_debug_set_variant"):          # development library: force the 64- / 128-pixel tile
            ts = []
            ds = []
            for v in (73, 74, 78):
                lib.se_debug_set_variant(v)
                ts.append(timeit(new))
                ds.append(float((new() - old()).abs().max()))
            lib.se_debug_set_variant(0)
            extra = f"   [k split over 1 / 2 wave groups, run-time loop: {ts[0]:6.1f} {ts[1]:6.1f} {ts[2]:6.1f} (maxdiff {max(ds):.1e})]  workgroups {(B * H * H // 64) * (cout // tile)}"
        diff = float((new() - old()).abs().max())
        flop = 2.0 * B * H * H * cin * cout
        byts = 4.0 * B * H * H * (cin + cout * (2 if has_res else 1))
        print(f"{cin:5d}->{cout:5d} @{H:2d}^2 x{count}: fused GEMM {t_new:7.1f} us ({flop / t_new / 1e6:6.1f} TF/s, {byts / t_new / 1e3:6.0f} GB/s)   "
              f"MIOpen + epilogue {t_old:7.1f} us   maxdiff {diff:.2e}" + extra, flush=True)
        tot_new += count * min(t_new, t_old) if False else count * t_new
        tot_old += count * t_old
    print(f"sum over one forward (B={B}): fused {tot_new:.0f} us, MIOpen + epilogue {tot_old:.0f} us")


if __name__ == "__main__":
    main()
