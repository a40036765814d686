"""Error of the 7x7x7 front-layer kernels of several library builds against a float64 convolution of the same float32 data (GPU).
usage: python tools/diag/k7_accuracy.py libA.so [libB.so ...] [--dim 64] [--batch 2] [--cin 33]"""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sceneego_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+"); ap.add_argument("--dim", type=int, default=64); ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--cin", type=int, default=33); ap.add_argument("--no-planar", action="store_true")
args = ap.parse_args()
dev = "cuda:0"
B, dim, cin, cout = args.batch, args.dim, args.cin, 16
cin_pad = (cin + 15) // 16 * 16
torch.manual_seed(0)
conv = torch.nn.Conv3d(cin, cout, 7, padding=3).to(dev)
w = conv.weight.detach().float().contiguous(); bias = conv.bias.detach().float().contiguous()
x = torch.randn(B, cin, dim, dim, dim, device=dev)
with torch.no_grad():
    ref = torch.relu(torch.nn.functional.conv3d(x.double(), w.double(), bias.double(), padding=3)).permute(0, 2, 3, 4, 1).contiguous()
planar = not args.no_planar
if planar:
    nt = (cin + 2) // 3
    xp = torch.zeros(B, nt * 3, dim, dim, dim, device=dev); xp[:, :cin] = x
    xin = xp.view(B, nt, 3, dim, dim, dim).permute(0, 1, 3, 4, 5, 2).contiguous()
else:
    xin = torch.zeros(B, dim, dim, dim, cin_pad, device=dev); xin[..., :cin] = x.permute(0, 2, 3, 4, 1)
vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(32 << 20, device=dev)
for p in args.libs:
    lib = ctypes.CDLL(os.path.abspath(p))
    for name, (res, a) in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None: fn.restype, fn.argtypes = res, a
    n = int(lib.se_conv3d_packed_elems(cout, cin_pad, 7, 0))
    wp = torch.empty(n, device=dev); bp = torch.empty(16, device=dev)
    assert lib.se_conv3d_pack_f32(vp(w), vp(bias), None, None, None, None, 0.0, vp(wp), vp(bp), cout, cin, cin_pad, 7, 0, st()) == 0
    out = torch.empty(B, dim, dim, dim, cout, device=dev)
    flags = _lib.EPI_RELU | (_lib.IN_PLANAR3 if planar else 0)
    rc = lib.se_conv3d_f32(vp(xin), vp(wp), vp(bp), None, vp(out), B, dim, cin, cin_pad, cout, 7, flags, vp(ws), ws.numel(), st())
    torch.cuda.synchronize(); assert rc == 0, rc
    e = (out.double() - ref).abs()
    print(f"{os.path.basename(p):30s} max err {float(e.max()):.2e} = {float(e.max() / ref.abs().max()):.2e} of max|y| ({float(ref.abs().max()):.2f}); "
          f"mean err {float(e.mean()):.2e} = {float(e.mean() / ref.std()):.2e} of std", flush=True)
