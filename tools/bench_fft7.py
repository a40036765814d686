#!/usr/bin/env python3
"""Go / no-go driver of the frequency-domain 7x7x7 front layer (csrc/conv3d_fft7.hip).

  python tools/bench_fft7.py --check          correctness vs torch conv3d + BN + ReLU on the device (32^3 B=1, 64^3 B=2, both layouts)
  python tools/bench_fft7.py --time [--batch 8] [--chunk N]
        HIP-event time of the whole call at 64^3 (chunk = samples per workspace pass; default = batch) next to the F(6,7) Winograd
        kernel on the same input; run it under `rocprofv3 --kernel-trace --stats` for the per-pass split.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sceneego_amd import _lib, synth          # noqa: E402
from sceneego_amd.v2v import _PackedConv      # noqa: E402

DEV = "cuda:0"


def make_layer(seed=67):
    conv = torch.nn.Conv3d(33, 16, 7, padding=3)
    bn = torch.nn.BatchNorm3d(16)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(synth.normal(seed, "w", tuple(conv.weight.shape))) * 0.02)
        conv.bias.copy_(torch.from_numpy(synth.normal(seed, "b", (16,))) * 0.1)
        bn.weight.copy_(torch.from_numpy(synth.uniform(seed, "g", (16,), 0.5, 1.5)))
        bn.bias.copy_(torch.from_numpy(synth.uniform(seed, "be", (16,), -0.2, 0.2)))
        bn.running_mean.copy_(torch.from_numpy(synth.uniform(seed, "m", (16,), -0.2, 0.2)))
        bn.running_var.copy_(torch.from_numpy(synth.uniform(seed, "v", (16,), 0.5, 1.5)))
    return conv.eval(), bn.eval()


def pack(conv, bn):
    pc = _PackedConv(conv, bn, cin_pad=48)
    hf = _lib.conv3d_k7_fft_pack(conv.weight.detach().float().contiguous(), bn.weight.detach().float().contiguous(),
                                 bn.running_var.detach().float().contiguous(), bn.eps, 16, 33)
    return pc, hf


def check():
    conv, bn = make_layer()
    conv, bn = conv.to(DEV), bn.to(DEV)
    pc, hf = pack(conv, bn)
    ok = True
    for B, dim in ((1, 32), (2, 64), (3, 48)):
        x = torch.from_numpy(synth.normal(5, "x%d" % dim, (B, 33, dim, dim, dim))).to(DEV)
        with torch.no_grad():
            want = F.relu(bn(conv(x)))                                   # MIOpen float32 on the device
            want64 = F.relu(F.batch_norm(F.conv3d(x.double().cpu(), conv.weight.double().cpu(), conv.bias.double().cpu(), padding=3),
                                         bn.running_mean.double().cpu(), bn.running_var.double().cpu(), bn.weight.double().cpu(),
                                         bn.bias.double().cpu(), False, 0.0, bn.eps)) if dim <= 32 else None
        for chunk in (B, 1):
            ws = torch.full((_lib.conv3d_k7_fft_workspace_elems(chunk, dim, 33),), float("nan"), device=DEV)
            for quad in (False, True):
                out = torch.full((B, 16 * dim ** 3), -77.0, device=DEV)
                _lib.conv3d_k7_fft(x.contiguous(), hf, pc.b, out, B, dim, 33, 16, _lib.EPI_RELU | (_lib.OUT_QUAD if quad else 0), ws)
                torch.cuda.synchronize()
                got = (out.view(B, 4, dim, dim, dim, 4).permute(0, 1, 5, 2, 3, 4).reshape(B, 16, dim, dim, dim) if quad
                       else out.view(B, dim, dim, dim, 16).permute(0, 4, 1, 2, 3))
                scale = float(want.abs().max())
                err = float((got - want).abs().max())
                msg = f"B={B} dim={dim} chunk={chunk} quad={quad}: max|fft - torch f32| = {err:.3e} ({err / scale:.2e} of max|y| = {scale:.2f})"
                if want64 is not None:
                    e64 = float((got.double().cpu() - want64).abs().max())
                    t64 = float((want.double().cpu() - want64).abs().max())
                    msg += f"; vs float64: fft {e64:.2e}, torch f32 {t64:.2e}"
                print(msg, flush=True)
                ok &= err < 2e-5 * scale and bool(torch.isfinite(got).all())
    print("CHECK", "OK" if ok else "FAILED")
    return ok


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def time_it(B, chunk, dim=64, quick=False):
    conv, bn = make_layer()
    conv, bn = conv.to(DEV), bn.to(DEV)
    pc, hf = pack(conv, bn)
    x = torch.from_numpy(synth.normal(5, "xt", (B, 33, dim, dim, dim))).to(DEV).contiguous()
    ws = torch.empty((_lib.conv3d_k7_fft_workspace_elems(chunk, dim, 33),), device=DEV)
    out = torch.empty((B, 16 * dim ** 3), device=DEV)
    for quad in ((True,) if quick else (True, False)):
        fl = _lib.EPI_RELU | (_lib.OUT_QUAD if quad else 0)
        ms = timeit(lambda: _lib.conv3d_k7_fft(x, hf, pc.b, out, B, dim, 33, 16, fl, ws))
        print(f"fft7 B={B} dim={dim} chunk={chunk} quad={quad}: {ms:.3f} ms per call", flush=True)
    if quick:
        return
    # the F(6,7) Winograd kernel on the same data (triplet-planar input)
    xin = torch.zeros(B, dim, dim, dim, 33, device=DEV)
    xin.copy_(x.permute(0, 2, 3, 4, 1))
    x3 = xin.view(B, dim, dim, dim, 11, 3).permute(0, 4, 1, 2, 3, 5).contiguous()
    o2 = torch.empty((B, dim, dim, dim, 16), device=DEV)
    wsw = torch.empty(32 << 20, device=DEV)
    ms = timeit(lambda: _lib.conv3d(x3, pc.w, pc.b, None, o2, B, dim, 33, 48, 16, 7, _lib.EPI_RELU | _lib.IN_PLANAR3, wsw))
    print(f"wino67 B={B} dim={dim}: {ms:.3f} ms per call", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--quick", action="store_true", help="--time: the quad-planar form only, no Winograd comparison (attribution builds)")
    a = ap.parse_args()
    rc = 0
    if a.check:
        rc = 0 if check() else 1
    if a.time:
        time_it(a.batch, a.chunk or a.batch, a.dim, a.quick)
    sys.exit(rc)
