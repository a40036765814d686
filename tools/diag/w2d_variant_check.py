import os, sys, torch
sys.path.insert(0, os.getcwd())
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
lib = _lib.load(); dev = "cuda:0"
for (B, dim, cin, cout) in ((8, 64, 32, 32), (2, 32, 64, 64), (3, 16, 128, 128), (1, 64, 16, 32)):
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
    pc = _PackedConv(conv, None)
    x = torch.randn(B, dim, dim, dim, cin, device=dev); res = torch.randn(B, dim, dim, dim, cout, device=dev)
    outs = []
    for v in (0, 60):
        lib.se_debug_set_variant(v)
        out = torch.empty_like(res)
        _lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, 3, None)
        torch.cuda.synchronize()
        outs.append(out)
    lib.se_debug_set_variant(0)
    print((B, dim, cin, cout), "max diff variant 60 vs 0:", float((outs[0] - outs[1]).abs().max()))
