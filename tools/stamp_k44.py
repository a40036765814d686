"""Diagnostic: phase timing of the F(4,3) x F(4,3) 3^3 kernel (needs a development build with the stamps:
   sceneego_amd/csrc/build.sh --devtools -DSE_STAMP44; run with SCENEEGO_HIP_LIB=sceneego_amd/libsceneego_hip_dev.so)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
lib = _lib.load(); dev = "cuda:0"
lib.se_debug_set_variant(63)      # the F(4,3) x F(4,3) experiment
B, dim, cin, cout = 8, 64, 32, 32
conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
pc = _PackedConv(conv, None, None, torch.float32)
x = torch.randn(B, dim, dim, dim, cin, device=dev); res = torch.randn(B, dim, dim, dim, cout, device=dev)
out = torch.empty(B, dim, dim, dim, cout, device=dev)
FL = _lib.EPI_RELU | (0 if os.environ.get("NO_RES") else _lib.EPI_RES_PRE_RELU) | _lib.IN_OCTET | _lib.OUT_OCTET
dbg = torch.zeros(256 * 8 * 14, dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.conv3d(x, pc.w, pc.b, None if os.environ.get("NO_RES") else res, out, B, dim, cin, cin, cout, 3, FL, None)
lib.se_debug_set_stamp_buffer_44.argtypes = [ctypes.c_void_p]
lib.se_debug_set_stamp_buffer_44(ctypes.c_void_p(dbg.data_ptr()))
_lib.conv3d(x, pc.w, pc.b, None if os.environ.get("NO_RES") else res, out, B, dim, cin, cin, cout, 3, FL, None)
torch.cuda.synchronize()
lib.se_debug_set_stamp_buffer_44(None)
d = dbg.view(256, 8, 14).double()
steps = d[:, :, 13].mean()
names = ["MFMA q0-4", "mid barrier", "MFMA q5-8", "tile end: row fetch", "barrier 1", "pass 1", "barrier 2", "E2 + vmcnt | pass 2 + vmcnt", "barrier 3", "(-)", "epi: setup + first loads", "epi: y transform", "epi: z transform + stores"]
print(f"steps per workgroup {steps:.0f}; cycles per step, mean over workgroups")
for w in range(8):
    v = d[:, w]
    print(f"wave {w}: " + "  ".join(f"{n} {v[:, i].mean() / steps:7.1f}" for i, n in enumerate(names)) + f"  sum {(v[:, :13].sum(1)).mean() / steps:8.1f}")
