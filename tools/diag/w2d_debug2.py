import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
dev = "cuda:0"
torch.manual_seed(0)
B, dim, cin, cout = 1, 16, 16, 32
# x = coordinate code: value = 100*z + 10*y... use distinct per position: z*256 + y*16 + x  (channel 0 only)
x = torch.zeros(B, cin, dim, dim, dim, device=dev)
zz, yy, xx = torch.meshgrid(torch.arange(dim), torch.arange(dim), torch.arange(dim), indexing="ij")
x[0, 0] = (zz * 256 + yy * 16 + xx).float().to(dev) + 1
w = torch.zeros(cout, cin, 3, 3, 3, device=dev)
w[1, 0, 1, 1, 1] = 1.0
conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
with torch.no_grad():
    conv.weight.copy_(w); conv.bias.zero_()
pc = _PackedConv(conv, None)
out = torch.full((B, dim, dim, dim, cout), -77.0, device=dev)
_lib.conv3d(x.permute(0, 2, 3, 4, 1).contiguous(), pc.w, pc.b, None, out, B, dim, cin, cin, cout, 3, 0, None)
torch.cuda.synchronize()
got = out[0, :, :, :, 1].cpu()
want = x[0, 0].cpu()
bad = ((got - want).abs() > 1e-3).nonzero()
print("bad count", len(bad))
for i in range(0, min(len(bad), 400), 13):
    z, y, xq = bad[i].tolist()
    g = float(got[z, y, xq]); wv = float(want[z, y, xq])
    print(f"z{z:2d} y{y:2d} x{xq:2d}: want {wv:7.0f} got {g:10.3f}  diff {g - wv:9.3f}")
# other couts at a bad position
z, y, xq = bad[0].tolist()
print("all couts at first bad pos:", [round(float(v), 2) for v in out[0, z, y, xq].cpu()])
z, y, xq = 5, 5, 5
print("all couts at a good pos:", [round(float(v), 2) for v in out[0, z, y, xq].cpu()])
