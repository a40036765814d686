import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sceneego_amd import _lib, load_config, synth
from sceneego_amd.voxel_net_depth import VoxelNetwork_depth
dev = "cuda:0"
cfg = load_config()
net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
net.load_state_dict(synth.make_state_dict(net.state_dict(), seed=0), strict=True)
net = net.to(dev).eval()
lib = _lib.load()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib.se_debug_set_variant(variant)
img, depth = synth.make_inputs(91, 2, "floor")
img2, depth2 = synth.make_inputs(92, 2, "uniform")
img, depth, img2, depth2 = [t.to(dev) for t in (img, depth, img2, depth2)]
def run(i, d):
    with torch.no_grad():
        out = net(i, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=d)
    torch.cuda.synchronize()
    return [o.clone() for o in out[:3]]
e1 = run(img, depth); e2 = run(img2, depth2)
net.enable_graphs(True)
for k in range(4):
    g1 = run(img, depth); g2 = run(img2, depth2)
    print(f"variant {variant} replay {k}: joints diff A {float((g1[0]-e1[0]).abs().max()):.2e} B {float((g2[0]-e2[0]).abs().max()):.2e}"
          f" | feat diff A {float((g1[1]-e1[1]).abs().max()):.2e} B {float((g2[1]-e2[1]).abs().max()):.2e}"
          f" | vol diff A {float((g1[2]-e1[2]).abs().max()):.2e} B {float((g2[2]-e2[2]).abs().max()):.2e}")
