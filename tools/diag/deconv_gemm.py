"""Diagnostic: ConvTranspose2d(k=4, s=2, p=1) of the pose head as 16 batched GEMMs (4 output parities x 2x2 taps) vs MIOpen."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

KD = {0: ((1, 0), (3, -1)), 1: ((0, 1), (2, 0))}      # output parity -> ((kernel index, input offset), ...)


def prepare(w, B, H, W, dev):
    """w [ci, co, 4, 4] -> W16 [16, co, ci], idx [16 * B*H*W] into the (B, H+2, W+2)-flattened padded input"""
    mats, idx = [], []
    b = torch.arange(B).view(B, 1, 1)
    j = torch.arange(H).view(1, H, 1)
    i = torch.arange(W).view(1, 1, W)
    for a in (0, 1):
        for bb in (0, 1):
            for ky, dy in KD[a]:
                for kx, dx in KD[bb]:
                    mats.append(w[:, :, ky, kx].t().contiguous())
                    idx.append((b * (H + 2) * (W + 2) + (j + 1 + dy) * (W + 2) + (i + 1 + dx)).reshape(-1))
    return torch.stack(mats).to(dev), torch.cat(idx).to(dev)


def deconv_gemm(x, W16, idx):
    B, C, H, W = x.shape
    co = W16.shape[1]
    xp = F.pad(x, (1, 1, 1, 1)).permute(1, 0, 2, 3).reshape(C, -1)          # [C, B*(H+2)*(W+2)]
    xg = xp.index_select(1, idx).view(C, 16, B * H * W)                        # [C, 16, N]
    y = torch.bmm(W16, xg.permute(1, 0, 2))                                    # [16, co, N]
    y = y.view(4, 4, co, B * H * W).sum(1)                                     # [4 parities, co, N]
    y = y.view(2, 2, co, B, H, W).permute(3, 2, 4, 0, 5, 1).reshape(B, co, 2 * H, 2 * W)
    return y


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


dev = "cuda:0"
for (B, C, H, co) in ((8, 2048, 8, 256), (8, 256, 16, 256), (8, 256, 32, 256)):
    x = torch.randn(B, C, H, H, device=dev)
    w = torch.randn(C, co, 4, 4, device=dev) * (1.0 / C) ** 0.5
    W16, idx = prepare(w.cpu(), B, H, H, dev)
    ref = F.conv_transpose2d(x, w, None, stride=2, padding=1)
    got = deconv_gemm(x, W16, idx)
    err = float((ref - got).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = []
    for fn in (lambda: F.conv_transpose2d(x, w, None, stride=2, padding=1), lambda: deconv_gemm(x, W16, idx)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"[{B},{C},{H},{H}] -> {co}: MIOpen {res[0]:.1f} us, 16 batched GEMMs {res[1]:.1f} us, max diff {err:.2e}")
