"""Diagnostic: phase timing of the ping-pong F(4,3) kernel (needs sceneego_amd/csrc/build.sh -DSE_STAMPPP)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
lib = _lib.load(); dev = "cuda:0"
B, dim, cin, cout = 8, 64, 32, 32
conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
pc = _PackedConv(conv, None)
x = torch.randn(B, dim, dim, dim, cin, device=dev); res = torch.randn(B, dim, dim, dim, cout, device=dev)
out = torch.empty_like(res)
lib.se_debug_set_variant(19)
dbg = torch.zeros(256 * 8 * 6, dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, 3, None)
lib.se_debug_set_stamp_buffer(ctypes.c_void_p(dbg.data_ptr()))
_lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, 3, None)
torch.cuda.synchronize()
lib.se_debug_set_stamp_buffer(None)
d = dbg.view(256, 8, 6).double()
nph = d[:, :, 4].mean()
items = 64.0          # half tiles per group per workgroup (32 units x 2 chunks)
print(f"phases per wave {nph:.0f}")
for g, name in ((0, "group A (waves 0-3)"), (1, "group B (waves 4-7)")):
    w = d[:, 4 * g:4 * g + 4]
    print(f"{name}: MFMA phase {w[:, :, 0].mean() / items:8.0f} cycles/item  staging work {w[:, :, 1].mean() / items:8.0f}  "
          f"barrier wait {w[:, :, 2].mean() / nph:8.0f} per phase  idle phases {w[:, :, 3].mean():8.0f} total")
tot = (d[:, :, 0] + d[:, :, 1] + d[:, :, 2] + d[:, :, 3]).mean()
print(f"sum per wave {tot:.0f} cycles = {tot / items:.0f} per item (ideal MFMA: 6912 per phase, 13824 per item)")
