"""Localise errors of the 2-D Winograd 3x3x3 kernel with structured weights (single tap / single channel pair)."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
dev = "cuda:0"
torch.manual_seed(0)

def run(B, dim, cin, cout, w, x, res=None):
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w); conv.bias.zero_()
    pc = _PackedConv(conv, None)
    out = torch.full((B, dim, dim, dim, cout), -77.0, device=dev)
    xin = x.permute(0, 2, 3, 4, 1).contiguous()
    _lib.conv3d(xin, pc.w, pc.b, None, out, B, dim, cin, cin, cout, 3, 0, None)
    torch.cuda.synchronize()
    want = F.conv3d(x.cpu(), w.cpu(), padding=1)
    got = out.permute(0, 4, 1, 2, 3).cpu()
    return got, want

def report(name, got, want):
    d = (got - want).abs()
    print(f"{name}: max err {float(d.max()):.3e} (max |want| {float(want.abs().max()):.3f})", end="")
    if float(d.max()) > 1e-4:
        bad = (d > 1e-4)
        # which coordinates are bad
        idx = bad.nonzero()
        print(f"  bad {int(bad.sum())}/{bad.numel()}")
        for ax, nm in ((1, "cout"), (2, "z"), (3, "y"), (4, "x")):
            vals = sorted(set(idx[:, ax].tolist()))
            print(f"     bad {nm}: {vals[:40]}{'...' if len(vals) > 40 else ''}")
    else:
        print("  OK")

B, dim, cin, cout = 1, 16, 16, 32
x = torch.randn(B, cin, dim, dim, dim, device=dev)
# 1. centre tap identity on channel 0 -> cout 0
for (kz, ky, kx) in [(1, 1, 1), (0, 1, 1), (2, 1, 1), (1, 0, 1), (1, 2, 1), (1, 1, 0), (1, 1, 2)]:
    w = torch.zeros(cout, cin, 3, 3, 3, device=dev)
    w[0, 0, kz, ky, kx] = 1.0
    got, want = run(B, dim, cin, cout, w, x)
    report(f"tap({kz},{ky},{kx}) c0->o0", got, want)
for (ci, co) in [(1, 0), (2, 0), (7, 0), (8, 0), (13, 0), (0, 1), (0, 5), (0, 17), (3, 31)]:
    w = torch.zeros(cout, cin, 3, 3, 3, device=dev)
    w[co, ci, 1, 1, 1] = 1.0
    got, want = run(B, dim, cin, cout, w, x)
    report(f"centre c{ci}->o{co}", got, want)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.1
got, want = run(B, dim, cin, cout, w, x)
report("random 16->32 @16", got, want)
for (B, dim, cin, cout) in [(1, 16, 16, 32), (1, 16, 32, 64), (2, 32, 32, 32), (1, 64, 32, 32)]:
    x = torch.randn(B, cin, dim, dim, dim, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.1
    got, want = run(B, dim, cin, cout, w, x)
    report(f"random {cin}->{cout} @{dim} B{B}", got, want)
