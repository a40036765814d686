"""Times N single launches of the 4096-voxel level kernel (conv3d_k3_halo64_kernel) with HIP events and prints the distribution:
a launch far outside it would point at the kernel; usage: python tools/diag/halo64_outliers.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = "cuda:0"
torch.manual_seed(0)
conv = torch.nn.Conv3d(128, 128, 3, padding=1).to(dev)
pc = _PackedConv(conv, None, None, torch.float32)
for B, dim in ((8, 8), (1, 16)):
    x = torch.randn(B, dim, dim, dim, 128, device=dev)
    r = torch.randn(B, dim, dim, dim, 128, device=dev)
    out = torch.empty_like(x)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
    for e0, e1 in ev:
        e0.record()
        _lib.conv3d(x, pc.w, pc.b, r, out, B, dim, 128, 128, 128, 3, _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU)
        e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    print(f"B={B} {dim}^3: {N} launches, us: min {t[0]:.1f} median {t[N // 2]:.1f} p99 {t[int(N * 0.99)]:.1f} p99.9 {t[int(N * 0.999)]:.1f} max {t[-1]:.1f}")
