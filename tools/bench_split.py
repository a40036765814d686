"""Launch time of the experimental split-bf16 3x3x3 convolution (se_conv3d_k3_split3_f32) beside the float32 2-D Winograd kernel on
the same tensors, interleaved rounds in one process.  usage: python tools/bench_split.py [--batch 8] [--rounds 12]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sceneego_amd import _lib  # noqa: E402
from sceneego_amd.v2v import _PackedConv  # noqa: E402

SHAPES = [(64, 32, 32), (64, 16, 32), (32, 64, 64), (32, 32, 64), (16, 128, 128), (16, 64, 128), (128, 32, 32)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--octet", type=int, default=3, help="1 IN_OCTET, 2 OUT_OCTET (both kernels interpret the same buffers alike)")
    args = ap.parse_args()
    dev = "cuda:0"
    _lib.load()
    B = args.batch
    for dim, cin, cout in SHAPES:
        if dim == 128 and B > 8:
            continue
        conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
        pc = _PackedConv(conv, None, None, torch.float32, split3=True)
        x = torch.randn(B, dim, dim, dim, cin, device=dev)
        res = torch.randn(B, dim, dim, dim, cout, device=dev)
        o32 = torch.empty_like(res)
        osp = torch.empty_like(res)
        flags = _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU | (_lib.IN_OCTET if args.octet & 1 else 0) | (_lib.OUT_OCTET if args.octet & 2 else 0)
        t32, tsp = [], []
        for r in range(args.rounds + 2):
            for which in (0, 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if which == 0:
                    _lib.conv3d(x, pc.w, pc.b, res, o32, B, dim, cin, pc.cin_pad, cout, 3, flags, None)
                else:
                    _lib.conv3d_k3_split3(x, pc.w_split, pc.b, res, osp, B, dim, pc.cin_pad, cout, flags)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    (t32 if which == 0 else tsp).append(e0.elapsed_time(e1))
        t32.sort(); tsp.sort()
        flop = 2.0 * B * dim ** 3 * 27 * cin * cout
        nbytes = 4.0 * B * dim ** 3 * (cin + 2 * cout)
        d = float((o32 - osp).abs().max() / o32.abs().max())
        print(f"3x3x3 {cin:3d}->{cout:3d} @{dim}^3 B={B}: f32 Winograd med {t32[len(t32) // 2]:.4f} ms | split-bf16 med {tsp[len(tsp) // 2]:.4f} ms "
              f"min {tsp[0]:.4f} = {flop / tsp[len(tsp) // 2] / 1e9:.0f} TF/s direct-equivalent ({3 * flop / tsp[len(tsp) // 2] / 1e9:.0f} executed bf16 TF/s), "
              f"{nbytes / tsp[len(tsp) // 2] / 1e6:.0f} GB/s algorithmic; max|diff| / max|y| {d:.1e}", flush=True)


if __name__ == "__main__":
    main()
