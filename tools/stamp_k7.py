"""Diagnostic: phase timing of the F(4,7) 7^3 kernel (needs sceneego_amd/csrc/build.sh -DSE_STAMP47)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
lib = _lib.load(); dev = "cuda:0"
B, dim, cin, cout = 8, 64, 33, 16
conv = torch.nn.Conv3d(cin, cout, 7, padding=3).to(dev)
pc = _PackedConv(conv, None, 48, torch.float32)
planar = len(sys.argv) > 1 and sys.argv[1] == "p3"
x = torch.randn(B, 11, dim, dim, dim, 3, device=dev) if planar else torch.randn(B, dim, dim, dim, 48, device=dev)
FL = 1 | (_lib.IN_PLANAR3 if planar else 0)
out = torch.empty(B, dim, dim, dim, cout, device=dev)
dbg = torch.zeros(256 * 8 * 6, dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.conv3d(x, pc.w, pc.b, None, out, B, dim, cin, 48, cout, 7, FL, None)
lib.se_debug_set_stamp_buffer(ctypes.c_void_p(dbg.data_ptr()))
_lib.conv3d(x, pc.w, pc.b, None, out, B, dim, cin, 48, cout, 7, FL, None)
torch.cuda.synchronize()
lib.se_debug_set_stamp_buffer(None)
d = dbg.view(256, 8, 6).double()
items = d[:, :, 4].mean()
print(f"items per workgroup {items:.0f} (s_memtime ticks)")
for w in range(8):
    v = d[:, w]
    print(f"wave {w}: MFMA block {v[:, 0].mean() / items:8.1f}  barrier1 {v[:, 1].mean() / items:8.1f}  commit+stores {v[:, 2].mean() / items:8.1f}"
          f"  barrier2 {v[:, 3].mean() / items:8.1f}  sum {(v[:, :4].sum(1)).mean() / items:8.1f}")
