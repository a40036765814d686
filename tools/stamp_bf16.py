"""Diagnostic: phase timing of conv_bf16_k3_kernel from s_memtime stamps (needs a -DSE_STAMPB build of the library:
sceneego_amd/csrc/build.sh -DSE_STAMPB)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sceneego_amd import _lib  # noqa: E402
from sceneego_amd.v2v import _PackedConv  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
B, dim, cin, cout = 8, 64, 32, 32
BF = torch.bfloat16
conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
pc = _PackedConv(conv, None, None, BF)
x = torch.randn(B, dim, dim, dim, cin, device=dev).to(BF)
res = torch.randn(B, dim, dim, dim, cout, device=dev).to(BF)
out = torch.empty_like(res)
nwg = B * (dim // 4) * (dim // 8) * (dim // 16)
dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, 3)
lib.se_debug_set_stamp_buffer_b.argtypes = [ctypes.c_void_p]
lib.se_debug_set_stamp_buffer_b(ctypes.c_void_p(dbg.data_ptr()))
_lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, 3)
torch.cuda.synchronize()
lib.se_debug_set_stamp_buffer_b(None)
d = dbg.view(nwg, 4, 8).double()
names = ["issue stage-0 loads", "wait + commit + barriers (chunk 0)", "compute chunk 0 (+ issue stage 1)", "commit + barriers (chunk 1)",
         "compute chunk 1 (+ epilogue loads)", "epilogue stores issued"]
tot = d[:, :, 6] - d[:, :, 0]
print(f"workgroups {nwg}; per-wave lifetime cycles (100 MHz s_memtime ticks x?): mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}")
for i, n in enumerate(names):
    seg = d[:, :, i + 1] - d[:, :, i]
    print(f"{n:42s} mean {seg.mean():9.0f}  p10 {seg.flatten().kthvalue(int(0.1 * seg.numel()))[0]:9.0f}  p90 {seg.flatten().kthvalue(int(0.9 * seg.numel()))[0]:9.0f}")
span = d[:, :, 6].max() - d[:, :, 0].min()
print(f"kernel span {span:.0f} ticks; sum of lifetimes / (span * 256 CUs * 2 slots * 4 waves) = {tot.sum() / (span * 256 * 8):.2f}")
