"""A/B timing of se_conv3d_f32 between several BUILDS of the library in one process (interleaved rounds, guide rule 24).

usage: python tools/ab_libs.py libA.so libB.so [...] [--shapes 0,3] [--rounds 12] [--batch 8] [--octet 3]

Each library packs its own weights (se_conv3d_pack_f32) and runs the same input; per shape the median / min launch time per
library and the max |difference| of every library's output to the first one's are printed.
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sceneego_amd import _lib  # noqa: E402

SHAPES = [  # (dim, cin, cin_pad, cout, k)
    (64, 32, 32, 32, 3), (64, 16, 16, 32, 3), (64, 33, 48, 16, 7), (32, 64, 64, 64, 3), (32, 32, 32, 64, 3), (16, 128, 128, 128, 3),
    (128, 32, 32, 32, 3), (8, 128, 128, 128, 3), (4, 128, 128, 128, 3), (2, 128, 128, 128, 3),
]


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--shapes", default="0")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--octet", type=int, default=3, help="2-D Winograd shapes: 1 IN_OCTET, 2 OUT_OCTET, 3 both, 0 channels-last")
    ap.add_argument("--no-res", action="store_true")
    ap.add_argument("--res-octet", action="store_true", help="2-D Winograd shapes: the skip tensor is read in the planar layout too (SE_RES_OCTET / SE_RES_QUAD)")
    ap.add_argument("--layout", default="quad", choices=["quad", "oct"],
                    help="planar layout --octet's bits select: quad-planar (F(4,3) x F(4,3) kernel, the default since round 5) or octet-planar (F(4,3) x F(2,3))")
    ap.add_argument("--planar3", action="store_true", help="7^3 shapes: triplet-planar input (SE_IN_PLANAR3)")
    args = ap.parse_args()
    dev = "cuda:0"
    libs = [load(p) for p in args.libs]
    B = args.batch
    vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(32 << 20, device=dev)
    for si in [int(x) for x in args.shapes.split(",")]:
        dim, cin, cin_pad, cout, k = SHAPES[si]
        torch.manual_seed(si)
        conv = torch.nn.Conv3d(cin, cout, k, padding=(k - 1) // 2).to(dev)
        w = conv.weight.detach().float().contiguous()
        bias = conv.bias.detach().float().contiguous()
        packs = []
        for lib in libs:
            n = int(lib.se_conv3d_packed_elems(cout, cin_pad, k, 0))
            wp = torch.empty(n, device=dev)
            bp = torch.empty((cout + 15) // 16 * 16, device=dev)
            rc = lib.se_conv3d_pack_f32(vp(w), vp(bias), None, None, None, None, 0.0, vp(wp), vp(bp), cout, cin, cin_pad, k, 0, st())
            assert rc == 0, rc
            packs.append((wp, bp))
        x = torch.randn(B, dim, dim, dim, cin_pad, device=dev)
        if cin_pad > cin:
            x[..., cin:] = 0
        if k == 7 and args.planar3:          # triplet-planar input [B][ceil(cin/3)][D][D][D][3] of the production 7^3 layer
            nt = (cin + 2) // 3
            xp = torch.zeros(B, nt * 3, dim, dim, dim, device=dev)
            xp[:, :cin] = x[..., :cin].permute(0, 4, 1, 2, 3)
            x = xp.view(B, nt, 3, dim, dim, dim).permute(0, 1, 3, 4, 5, 2).contiguous()
        no_res = args.no_res or k == 7
        res = None if no_res else torch.randn(B, dim, dim, dim, cout, device=dev)
        flags = _lib.EPI_RELU | (0 if no_res else _lib.EPI_RES_PRE_RELU)
        if k == 7 and args.planar3:
            flags |= _lib.IN_PLANAR3
        if k == 3 and libs[0].se_conv3d_f32_algo(dim, cin_pad, cout, 3) == 2:
            IN, OUT, RES = (_lib.IN_QUAD, _lib.OUT_QUAD, _lib.RES_QUAD) if args.layout == "quad" else (_lib.IN_OCTET, _lib.OUT_OCTET, _lib.RES_OCTET)
            if args.layout == "quad" and libs[0].se_conv3d_f32_variant(B, dim, cin_pad, cout, 3, IN) != 3:
                IN, OUT, RES = _lib.IN_OCTET, _lib.OUT_OCTET, _lib.RES_OCTET          # (levels the F(4,3) x F(4,3) kernel does not take)
            flags |= (IN if args.octet & 1 else 0) | (OUT if args.octet & 2 else 0)
            if args.res_octet and not no_res:
                flags |= RES
        outs = [torch.empty(B, dim, dim, dim, cout, device=dev) for _ in libs]
        times = [[] for _ in libs]
        for r in range(args.rounds + 2):
            for i, lib in enumerate(libs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = lib.se_conv3d_f32(vp(x), vp(packs[i][0]), vp(packs[i][1]), vp(res), vp(outs[i]), B, dim, cin, cin_pad, cout, k, flags,
                                       vp(ws), ws.numel(), st())
                e1.record()
                torch.cuda.synchronize()
                assert rc == 0, rc
                if r >= 2:
                    times[i].append(e0.elapsed_time(e1))
        flop = 2.0 * B * dim ** 3 * k ** 3 * cin * cout
        print(f"k{k} {cin}->{cout} @{dim}^3 B={B} flags {flags}:")
        for i, p in enumerate(args.libs):
            t = sorted(times[i])
            med = t[len(t) // 2]
            diff = float((outs[i] - outs[0]).abs().max())
            print(f"   {os.path.basename(p):32s} med {med:.4f} ms  min {t[0]:.4f}  ({flop / med / 1e9:.1f} TF/s direct-equivalent)  max|out - out[0]| {diff:.2e}",
                  flush=True)


if __name__ == "__main__":
    main()
