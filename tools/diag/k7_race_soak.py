"""Soak test of the F(6,7) 7^3 kernel (LDS-DMA regions, counted vmcnt waits, two barriers per item): the kernel is deterministic, so
N launches on the same input must be bit-identical; any difference is a race.  usage: python tools/diag/k7_race_soak.py [N]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
bad = 0
for B, dim, cin, planar in ((8, 64, 33, True), (2, 32, 33, True), (3, 32, 32, False), (1, 128, 33, True)):
    torch.manual_seed(dim + cin)
    conv = torch.nn.Conv3d(cin, 16, 7, padding=3).to(dev)
    cin_pad = (cin + 15) // 16 * 16
    pc = _PackedConv(conv, None, cin_pad, torch.float32)
    x = torch.randn(B, cin, dim, dim, dim, device=dev)
    if planar:
        nt = (cin + 2) // 3
        xp = torch.zeros(B, nt * 3, dim, dim, dim, device=dev); xp[:, :cin] = x
        xin = xp.view(B, nt, 3, dim, dim, dim).permute(0, 1, 3, 4, 5, 2).contiguous()
    else:
        xin = torch.zeros(B, dim, dim, dim, cin_pad, device=dev); xin[..., :cin] = x.permute(0, 2, 3, 4, 1)
    flags = _lib.EPI_RELU | (_lib.IN_PLANAR3 if planar else 0)
    ref = torch.empty(B, dim, dim, dim, 16, device=dev)
    _lib.conv3d(xin, pc.w, pc.b, None, ref, B, dim, cin, cin_pad, 16, 7, flags, None)
    out = torch.empty_like(ref)
    n_bad = 0
    s2 = torch.cuda.Stream()
    junk = torch.randn(64 << 20, device=dev)
    for i in range(N if dim < 128 else max(8, N // 10)):
        out.fill_(-1.0)
        if i % 3 == 1:      # memory traffic on a second stream beside the kernel
            with torch.cuda.stream(s2):
                junk.mul_(1.0001)
        _lib.conv3d(xin, pc.w, pc.b, None, out, B, dim, cin, cin_pad, 16, 7, flags, None)
        if not torch.equal(out, ref):
            n_bad += 1
    torch.cuda.synchronize()
    print(f"7^3 {cin}->16 @{dim}^3 B={B} {'planar3' if planar else 'channels-last'}: {n_bad} launches differed from the first")
    bad += n_bad
assert bad == 0
