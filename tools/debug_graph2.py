import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sceneego_amd import _lib, load_config, synth, op
from sceneego_amd.voxel_net_depth import VoxelNetwork_depth
dev = "cuda:0"
cfg = load_config()
net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
net.load_state_dict(synth.make_state_dict(net.state_dict(), seed=0), strict=True)
net = net.to(dev).eval()
net.compile()
img, depth = synth.make_inputs(91, 2, "floor")
img, depth = img.to(dev), depth.to(dev)
net._device_tables(net.grid_coord_proj_batch, net.coord_volumes, torch.device(dev))
prog = net.volume_net.program
B, G, N = 2, 64, 64 ** 3

def graphed(fn, name):
    ref = fn(); torch.cuda.synchronize(); ref = [r.clone() for r in ref]
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    for k in range(3):
        g.replay(); torch.cuda.synchronize()
        print(name, "replay", k, [f"{float((o - r).abs().max()):.2e}" for o, r in zip(out, ref)])

x_in = torch.randn(B, G, G, G, prog.cin_pad, device=dev); x_in[..., 33:] = 0
graphed(lambda: [net.depth_map_to_voxel(depth)], "voxelize")
feat = torch.randn(B, 64, 64, 32, device=dev)
def gath():
    x = torch.empty((B, G, G, G, prog.cin_pad), device=dev); x[..., 32:].zero_()
    _lib.unproject_gather(feat, net._gather_idx, net._gather_w, x, B, 4096, 32, N, prog.cin_pad, 0)
    return [x]
graphed(gath, "gather")
graphed(lambda: [prog.run(x_in, B, G)], "v2v")
t16 = torch.randn(B, G, G, G, 16, device=dev)
graphed(lambda: [prog._conv(x_in, prog.front0, B, G, _lib.EPI_RELU)], "conv7")
graphed(lambda: [prog._res(t16, prog.front_res[0], B, G)], "res16->32")
t128 = torch.randn(B, 4, 4, 4, 128, device=dev)
graphed(lambda: [prog._res(t128, prog.mid, B, 4)], "res128@4 (splitk)")
t64 = torch.randn(B, 32, 32, 32, 64, device=dev)
graphed(lambda: [prog._res(t64, prog.skip[1], B, 32)], "res64@32 (tiled)")
lg = torch.randn(B, 15, N, device=dev) * 6
def sa():
    j = torch.empty((B, 15, 3), device=dev); v = torch.empty_like(lg)
    _lib.softargmax3d(lg, net._coord_flat, v, j, B * 15, N, 1)
    return [j, v]
graphed(sa, "softargmax")
