"""A/B timing of se_conv3d_f32 variants on the GPU (interleaved rounds in one process, guide rule 24).

usage: python tools/bench_conv.py [--batch 8] [--rounds 10]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sceneego_amd import _lib  # noqa: E402
from sceneego_amd.v2v import _PackedConv  # noqa: E402

SHAPES = [  # (dim, cin, cin_pad, cout, k)
    (64, 32, 32, 32, 3),
    (64, 16, 16, 32, 3),
    (64, 33, 48, 16, 7),
    (32, 64, 64, 64, 3),
    (32, 32, 32, 64, 3),
    (16, 128, 128, 128, 3),
    (128, 32, 32, 32, 3),      # BASELINE config 5: 128^3 grid
    (128, 33, 48, 16, 7),
    (8, 128, 128, 128, 3),     # deep pyramid levels
    (4, 128, 128, 128, 3),
    (2, 128, 128, 128, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--no-res", action="store_true")
    ap.add_argument("--only", type=int, default=-1, help="index into SHAPES")
    ap.add_argument("--octet", type=int, default=0, help="extra flags for 2-D Winograd shapes: 1 = IN_OCTET, 2 = OUT_OCTET, 3 = both (timing only)")
    ap.add_argument("--quad", type=int, default=0, help="extra flags for shapes of the F(4,3) x F(4,3) kernel: 1 = IN_QUAD, 2 = OUT_QUAD, 4 = RES_QUAD (timing only: the tensors are random, the bytes moved are the same)")
    ap.add_argument("--bf16", action="store_true", help="time se_conv3d_bf16 (bf16 storage) instead")
    args = ap.parse_args()
    lib = _lib.load()
    dev = "cuda:0"
    B = args.batch
    variants = [int(v) for v in args.variants.split(",")]
    ws = torch.empty(32 << 20, device=dev)
    dt = torch.bfloat16 if args.bf16 else torch.float32
    for dim, cin, cin_pad, cout, k in (SHAPES if args.only < 0 else SHAPES[args.only:args.only + 1]):
        if args.bf16:
            cin_pad = (cin + 7) // 8 * 8
        conv = torch.nn.Conv3d(cin, cout, k, padding=(k - 1) // 2).to(dev)
        pc = _PackedConv(conv, None, cin_pad, dt)
        x = torch.randn(B, dim, dim, dim, cin_pad, device=dev).to(dt)
        if args.bf16 and k == 7:
            x = x.view(B, cin_pad // 8, dim, dim, dim, 8)        # octet-planar input of the bf16 front layer
        res = torch.randn(B, dim, dim, dim, cout, device=dev).to(dt)
        out = torch.empty(B, dim, dim, dim, cout, device=dev, dtype=dt)
        flop = 2.0 * B * dim ** 3 * k ** 3 * cin * cout
        octet = 0
        if args.octet and not args.bf16 and k == 3 and _lib.conv3d_algo(dim, cin_pad, cout, 3) == 2:
            octet = (_lib.IN_OCTET if args.octet & 1 else 0) | (_lib.OUT_OCTET if args.octet & 2 else 0)
        if args.quad and not args.bf16 and k == 3 and _lib.conv3d_variant(B, dim, cin_pad, cout, 3, _lib.IN_QUAD) == 3:
            octet = (_lib.IN_QUAD if args.quad & 1 else 0) | (_lib.OUT_QUAD if args.quad & 2 else 0) | (_lib.RES_QUAD if args.quad & 4 and not args.no_res else 0)
        times = {v: [] for v in variants}
        outs = {}
        for r in range(args.rounds + 2):
            for v in variants:
                if hasattr(lib, "se_debug_set_variant"):
                    lib.se_debug_set_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                no_res = args.no_res or k == 7     # the front-layer Winograd / bf16 kernels have no skip input
                _lib.conv3d(x, pc.w, pc.b, None if no_res else res, out, B, dim, cin, cin_pad, cout, k,
                            _lib.EPI_RELU | (0 if no_res else _lib.EPI_RES_PRE_RELU) | octet, None if args.bf16 else ws)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times[v].append(e0.elapsed_time(e1))
                if r == 0:
                    outs[v] = out.clone()
        if hasattr(lib, "se_debug_set_variant"):
            lib.se_debug_set_variant(0)
        base = outs[variants[0]]
        msg = f"k{k} {cin:3d}->{cout:3d} @{dim}^3 B={B}:"
        for v in variants:
            t = sorted(times[v])
            med = t[len(t) // 2]
            diff = float((outs[v].float() - base.float()).abs().max())
            gbs = B * dim ** 3 * (cin_pad + cout * (1 if args.no_res else 2)) * x.element_size() / med / 1e6
            msg += f"  v{v}: med {med:.3f} ms min {t[0]:.3f} ({flop / med / 1e9:.1f} TF/s, {gbs:.0f} GB/s) maxdiff {diff:.2e}"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
