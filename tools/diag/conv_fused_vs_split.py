"""Diagnostic: per backbone shape, MIOpen conv + fused HIP bias/ReLU pass vs torch.miopen_convolution_relu (MIOpen's fused op)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sceneego_amd import _lib  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [  # (cin, H, cout, k, stride)   conv1 / conv2 of the bottlenecks + stem
    (3, 256, 64, 7, 2), (64, 64, 64, 1, 1), (64, 64, 64, 3, 1), (256, 64, 64, 1, 1), (256, 64, 128, 1, 1), (128, 64, 128, 3, 2),
    (512, 32, 128, 1, 1), (128, 32, 128, 3, 1), (512, 32, 256, 1, 1), (256, 32, 256, 3, 2), (1024, 16, 256, 1, 1),
    (256, 16, 256, 3, 1), (1024, 16, 512, 1, 1), (512, 16, 512, 3, 2), (2048, 8, 512, 1, 1), (512, 8, 512, 3, 1)]


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, H, cout, k, st in shapes:
    x = torch.randn(B, cin, H, H, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    p = (k - 1) // 2
    split = lambda: _lib.bias_act_nchw(F.conv2d(x, w, None, stride=st, padding=p), b, None, True)
    fused = lambda: torch.miopen_convolution_relu(x, w, b, (st, st), (p, p), (1, 1), 1)
    d = float((split() - fused()).abs().max())
    print(f"{cin:5d}->{cout:4d} k{k} s{st} @{H:3d}: conv + bias_act {t(split):7.1f} us   miopen_convolution_relu {t(fused):7.1f} us   diff {d:.1e}")
