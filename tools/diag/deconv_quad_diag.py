import sys, torch
sys.path.insert(0, '/root/repo')
import torch.nn as nn
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
DEV = "cuda:0"
torch.manual_seed(0)
B, dim, cin, cout = 1, 16, 64, 32
up = nn.ConvTranspose3d(cin, cout, 2, stride=2).to(DEV)
pc = _PackedConv(up, None)
x = torch.randn(B, dim, dim, dim, cin, device=DEV)
out = torch.empty(B, 2*dim, 2*dim, 2*dim, cout, device=DEV)
_lib.deconv3d_k2s2(x, pc.w, pc.b, None, out, B, dim, cin, cout, _lib.EPI_RELU)
outq = torch.full((B, cout // 4, 2*dim, 2*dim, 2*dim, 4), -7.0, device=DEV)
_lib.deconv3d_k2s2(x, pc.w, pc.b, None, outq, B, dim, cin, cout, _lib.EPI_RELU | _lib.OUT_QUAD)
got = outq.permute(0, 2, 3, 4, 1, 5).reshape(B, 2*dim, 2*dim, 2*dim, cout)
bad = (got != out)
print("mismatch fraction", float(bad.float().mean()), "untouched (-7)", float((got == -7).float().mean()))
idx = bad.nonzero()
print(idx[:20].tolist())
# per x position / channel pattern
print("bad by x:", bad.float().mean(dim=(0,1,2,4)).tolist())
print("bad by channel:", bad.float().mean(dim=(0,1,2,3)).tolist())
print("bad by z:", bad.float().mean(dim=(0,2,3,4)).tolist()[:8])
# is got[x] equal to out at another x?
g0 = got[0, 0, 0, :, :4]; o0 = out[0, 0, 0, :, :4]
print("got row0 ch0-3:\n", g0[:8]); print("want:\n", o0[:8])
