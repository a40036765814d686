"""Diagnostic: phase timing of the tile-outer F(6,7) 7^3 kernel (needs a development build with the stamps:
   sceneego_amd/csrc/build.sh --devtools -DSE_STAMP67; run with SCENEEGO_HIP_LIB=sceneego_amd/libsceneego_hip_dev.so)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
lib = _lib.load(); dev = "cuda:0"
B, dim, cin, cout = 8, 64, 33, 16
conv = torch.nn.Conv3d(cin, cout, 7, padding=3).to(dev)
pc = _PackedConv(conv, None, 48, torch.float32)
x = torch.randn(B, 11, dim, dim, dim, 3, device=dev)
FL = 1 | _lib.IN_PLANAR3
out = torch.empty(B, dim, dim, dim, cout, device=dev)
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.conv3d(x, pc.w, pc.b, None, out, B, dim, cin, 48, cout, 7, FL, None)
lib.se_debug_set_stamp_buffer(ctypes.c_void_p(dbg.data_ptr()))
_lib.conv3d(x, pc.w, pc.b, None, out, B, dim, cin, 48, cout, 7, FL, None)
torch.cuda.synchronize()
lib.se_debug_set_stamp_buffer(None)
d = dbg.view(256, 8, 8).double()
items = d[:, :, 7].mean()
names = ["MFMA groups 0-18", "mid barrier", "MFMA groups 19-36", "A^T", "barrier 1", "commit (+store)", "barrier 2"]
print(f"items per workgroup {items:.0f}; s_memtime ticks (100 MHz) per item, mean over workgroups")
for w in range(8):
    v = d[:, w]
    print(f"wave {w}: " + "  ".join(f"{n} {v[:, i].mean() / items:7.1f}" for i, n in enumerate(names)) + f"  sum {(v[:, :7].sum(1)).mean() / items:8.1f}")
