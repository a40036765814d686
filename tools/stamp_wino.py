"""Diagnostic: where the Winograd conv kernel spends its wave cycles (s_memtime stamps, separate STAMP build)."""
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
NP = int(os.environ.get("SE_STAMP_PHASES", "4"))
dbg = torch.zeros(256 * 8 * NP, dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, 3, None)
lib.se_debug_set_stamp_buffer(ctypes.c_void_p(dbg.data_ptr()))
_lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, 3, None)
torch.cuda.synchronize()
lib.se_debug_set_stamp_buffer(None)
d = dbg.view(256, 8, NP).double()
tot = d.sum(dim=2)
print("per-wave total cycles: mean %.0f min %.0f max %.0f" % (tot.mean(), tot.min(), tot.max()))
names = ["setup(item start->first ds_read)", "mfma block (36 substeps)", "tail(commit+epilogue)", "barrier(+weights)"]
if NP == 6:
    names = ["setup", "mfma block (54 substeps)", "A^T + barrier 1", "commit (+weights)", "epilogue", "barrier 2"]
for i, n in enumerate(names):
    print(f"{n:36s} mean {d[:, :, i].mean():10.0f} cycles = {100 * d[:, :, i].mean() / tot.mean():5.1f} %   per item {d[:, :, i].mean() / 64:8.0f}")
if NP == 6:
    print("per wave index (mean over workgroups), cycles per item:")
    for w in range(8):
        print(f"  wave {w}: " + "  ".join(f"{names[i][:10]}={d[:, w, i].mean() / 64:7.0f}" for i in range(5)))
