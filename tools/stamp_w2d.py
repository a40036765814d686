"""Diagnostic: phase timing of conv3d_k3_wino2d_kernel (needs `csrc/build.sh --devtools -DSE_STAMP2D`, run with
SCENEEGO_HIP_LIB=sceneego_amd/libsceneego_hip_dev.so)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
lib = _lib.load(); dev = "cuda:0"
B, dim, cin, cout = 8, 64, int(os.environ.get("CIN", 32)), int(os.environ.get("COUT", 32))
conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
pc = _PackedConv(conv, None)
x = torch.randn(B, dim, dim, dim, cin, device=dev); res = torch.randn(B, dim, dim, dim, cout, device=dev)
out = torch.empty_like(res)
lib.se_debug_set_variant(int(os.environ.get("VARIANT", 0)))
FL = (_lib.IN_OCTET if int(os.environ.get("OCTET", 0)) & 1 else 0) | (_lib.OUT_OCTET if int(os.environ.get("OCTET", 0)) & 2 else 0)
dbg = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, 3 | FL, None)
lib.se_debug_set_stamp_buffer_2d.argtypes = [ctypes.c_void_p]
lib.se_debug_set_stamp_buffer_2d(ctypes.c_void_p(dbg.data_ptr()))
_lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, 3 | FL, None)
torch.cuda.synchronize()
lib.se_debug_set_stamp_buffer_2d(None)
d = dbg.view(256, 8, 16).double()
for g, name in ((0, "group A (waves 0-3)"), (1, "group B (waves 4-7)")):
    w = d[:, 4 * g:4 * g + 4]
    nm, ns = w[:, :, 8].mean(), w[:, :, 9].mean()
    print(f"{name}: {nm:.0f} MFMA phases, {ns:.0f} staging phases per wave")
    print(f"   MFMA phase   : first half {w[:, :, 0].mean() / nm:7.0f}  mid-barrier wait {w[:, :, 1].mean() / nm:7.0f}  second half {w[:, :, 2].mean() / nm:7.0f}  end-barrier wait {w[:, :, 3].mean() / nm:7.0f}   (ideal 2304 + 2304 cycles of MFMA issue)")
    print(f"   staging phase: first half {w[:, :, 4].mean() / ns:7.0f}  mid-barrier wait {w[:, :, 5].mean() / ns:7.0f}  second half {w[:, :, 6].mean() / ns:7.0f}  end-barrier wait {w[:, :, 7].mean() / ns:7.0f}")
    print(f"   staging: first half = commit, second half = epilogue (1 phase in {cin // 8}) {w[:, :, 13].mean() / ns:7.0f} + unit walk")
if int(os.environ.get("VARIANT", 0)) == 61:
    w = d.mean(dim=(0, 1))
    n = float(w[8])
    print("lockstep form, cycles per step (mean over waves): MFMA phase %.0f  barrier %.0f  V-tile transform %.0f  weight commit %.0f  epilogue+walk %.0f  barrier %.0f  (ideal MFMA 9216 per SIMD = 2 waves x 144 x 32)" % tuple(float(w[k]) / n for k in range(6)))
if os.environ.get("PERWAVE"):
    names = ["mfma1", "midwait", "mfma2", "endwait", "stg1", "midwait", "stg2", "endwait"]
    print("per wave (mean over workgroups), cycles per phase:  " + "  ".join(f"{n:>8s}" for n in names))
    for wv in range(8):
        row = d[:, wv, :8].mean(0) / d[:, wv, 8].mean()
        print(f"   wave {wv} (G{wv >> 2} ct{(wv >> 1) & 1} jt{wv & 1}):                          " + "  ".join(f"{float(v):8.0f}" for v in row))
