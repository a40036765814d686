import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/sceneego_amd") else os.getcwd())
from sceneego_amd import _lib
from sceneego_amd.v2v import _PackedConv
libs = sys.argv[1:]
dev = "cuda:0"
_lib.load()
def load(p):
    l = ctypes.CDLL(os.path.abspath(p)); res, args = _lib.SIGNATURES["se_conv3d_k3_split3_f32"]; l.se_conv3d_k3_split3_f32.restype = res; l.se_conv3d_k3_split3_f32.argtypes = args; return l
L = [load(p) for p in libs]
vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
for dim, cin, cout in ((64, 32, 32), (32, 64, 64), (16, 128, 128)):
    B = 8
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).to(dev)
    pc = _PackedConv(conv, None, None, torch.float32, split3=True)
    x = torch.randn(B, dim, dim, dim, cin, device=dev); res = torch.randn(B, dim, dim, dim, cout, device=dev); out = torch.empty_like(res)
    FL = int(os.environ.get("SPLIT_FLAGS", "3"))      # + 32 IN_OCTET, 64 OUT_OCTET, 128 RES_OCTET: layouts are only addressing here
    times = [[] for _ in L]
    for r in range(12):
        for i, l in enumerate(L):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = l.se_conv3d_k3_split3_f32(vp(x), vp(pc.w_split), vp(pc.b), vp(res), vp(out), B, dim, pc.cin_pad, cout, FL, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            e1.record(); torch.cuda.synchronize(); assert rc == 0
            if r >= 2: times[i].append(e0.elapsed_time(e1))
    print(f"{cin}->{cout}@{dim}^3 flags {FL}: " + "  ".join(f"{os.path.basename(p)[15:-3] or 'base'} {sorted(t)[len(t)//2]:.4f}" for p, t in zip(libs, times)), flush=True)
