"""Diagnostic: ConvTranspose2d(k4, s2, p1) through MIOpen directly vs as 4 sub-pixel 2x2 stride-1 convolutions."""
import time
import torch
import torch.nn.functional as F

dev = "cuda:0"


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def phase_weights(w):
    """w [Cin, Cout, 4, 4] -> 4 conv2d weights [Cout, Cin, 2, 2] for output phases (py, px)."""
    out = {}
    for py in (0, 1):
        ky = [3, 1] if py == 0 else [2, 0]
        for px in (0, 1):
            kx = [3, 1] if px == 0 else [2, 0]
            sub = w[:, :, ky][:, :, :, kx]                    # [Cin, Cout, 2, 2]
            out[(py, px)] = sub.permute(1, 0, 2, 3).contiguous()
    return out


def deconv_phases(x, pw, out):
    for (py, px), w in pw.items():
        pad = (1 if px == 0 else 0, 0 if px == 0 else 1, 1 if py == 0 else 0, 0 if py == 0 else 1)
        y = F.conv2d(F.pad(x, pad), w)
        out[:, :, py::2, px::2] = y
    return out


for B in (8, 32):
    for cin, cout, hw in ((2048, 256, 8), (256, 256, 16), (256, 256, 32)):
        x = torch.randn(B, cin, hw, hw, device=dev)
        w = torch.randn(cin, cout, 4, 4, device=dev) * 0.01
        ref = F.conv_transpose2d(x, w, stride=2, padding=1)
        pw = phase_weights(w)
        out = torch.empty_like(ref)
        got = deconv_phases(x, pw, out)
        err = float((got - ref).abs().max() / ref.abs().max())
        t0 = timeit(lambda: F.conv_transpose2d(x, w, stride=2, padding=1))
        t1 = timeit(lambda: deconv_phases(x, pw, out))
        print(f"B={B} {cin}->{cout} @{hw}: conv_transpose2d {t0:.3f} ms, 4-phase conv2d {t1:.3f} ms, rel diff {err:.1e}")
