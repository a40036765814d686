"""Diagnostic: phase timing of the F(4,3) x F(4,3) ping-pong 3^3 kernel (conv3d_wino44pp.hip; needs a development build with the
stamps: sceneego_amd/csrc/build.sh --devtools -DSE_STAMP44P; run with SCENEEGO_HIP_LIB=sceneego_amd/libsceneego_hip_dev.so).
env: NO_RES=1 no skip tensor, CL=1 channels-last tensors, SHAPE=dim,cin,cout (default 64,32,32)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
lib = _lib.load(); dev = "cuda:0"
B = int(os.environ.get("BATCH", "8"))
dim, cin, cout = (int(v) for v in os.environ.get("SHAPE", "64,32,32").split(","))
conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
pc = _PackedConv(conv, None, None, torch.float32)
x = torch.randn(B, dim, dim, dim, cin, device=dev); res = torch.randn(B, dim, dim, dim, cout, device=dev)
out = torch.empty(B, dim, dim, dim, cout, device=dev)
no_res = bool(os.environ.get("NO_RES"))
FL = _lib.EPI_RELU | (0 if no_res else _lib.EPI_RES_PRE_RELU) | (0 if os.environ.get("CL") else _lib.IN_OCTET | _lib.OUT_OCTET)
dbg = torch.zeros(256 * 8 * 12, dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.conv3d(x, pc.w, pc.b, None if no_res else res, out, B, dim, cin, cin, cout, 3, FL, None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib.se_debug_set_stamp_buffer_44p.argtypes = [ctypes.c_void_p]
lib.se_debug_set_stamp_buffer_44p(ctypes.c_void_p(dbg.data_ptr()))
e0.record()
_lib.conv3d(x, pc.w, pc.b, None if no_res else res, out, B, dim, cin, cin, cout, 3, FL, None)
e1.record()
torch.cuda.synchronize()
lib.se_debug_set_stamp_buffer_44p(None)
d = dbg.view(256, 8, 12).double()
steps = d[:, :, 11].mean()
names = ["MFMA half A", "mid barrier", "MFMA half B", "end barrier", "rows + pass 1 (+ epi y)", "mid barrier", "pass 2", "epi z", "walk", "end barrier"]
print(f"{cin}->{cout} @{dim}^3 B={B} {'no skip' if no_res else 'skip'} {'channels-last' if os.environ.get('CL') else 'octet-planar'}: "
      f"launch {e0.elapsed_time(e1):.3f} ms (stamp build), steps per workgroup {steps:.0f}; cycles per step, mean over workgroups "
      f"(s_memtime counts at 100 MHz x ... : constant-rate ticks, compare ratios)")
for w in range(8):
    v = d[:, w]
    print(f"wave {w} (group {w >> 2}): " + "  ".join(f"{n} {v[:, i].mean() / steps:7.1f}" for i, n in enumerate(names)) + f"  sum {(v[:, :10].sum(1)).mean() / steps:8.1f}")
