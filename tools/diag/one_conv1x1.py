#!/usr/bin/env python3
"""400 launches of se_conv2d_1x1_f32 on one backbone shape (for rocprofv3 --kernel-trace --stats: the kernel's own duration, which the
HIP-event microbenchmark cannot resolve below the host's ~10 us per call).  usage: one_conv1x1.py cin cout H residual(0/1) [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sceneego_amd import _lib      # noqa: E402

cin, cout, H, has_res = (int(a) for a in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 8
dev = "cuda:0"
x = torch.randn(B, cin, H, H, device=dev)
w = torch.randn(cout, cin, device=dev) * (2.0 / cin) ** 0.5
b = torch.randn(cout, device=dev)
res = torch.randn(B, cout, H, H, device=dev) if has_res else None
ib = torch.randn(cin, device=dev)
wp = _lib.conv2d_1x1_pack(w, _lib.conv2d_1x1_tile(B, cin, cout, H * H))
for _ in range(400):
    _lib.conv2d_1x1(x, wp, b, res, True, ib)
torch.cuda.synchronize()
