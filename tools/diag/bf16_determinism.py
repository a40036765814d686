"""Diagnostic: is the bf16 V2V program bitwise reproducible on identical input (it must be: no atomics anywhere)?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synthetic_state_dict
from sceneego_amd import load_config, synth
from sceneego_amd.voxel_net_depth import VoxelNetwork_depth
cfg = load_config(); cfg.model.v2v_dtype = "bf16"
net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
net.load_state_dict(synthetic_state_dict(False, 0), strict=True)
net = net.to("cuda:0").eval()
img, depth = synth.make_inputs(77, 2, "floor")
img, depth = img.to("cuda:0"), depth.to("cuda:0")
with torch.no_grad():
    kp0 = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)[0].clone()
    x = next(iter(net._xbuf.values())).clone()               # the bf16 octet-planar V2V input of that forward
    prog = net.volume_net.program
    l0 = prog.run(x, 2, 64).clone()
    for i in range(3):
        l1 = prog.run(x, 2, 64)
        print("V2V logits identical on identical input:", bool(torch.equal(l0, l1)), float((l0 - l1).abs().max()))
    for i in range(3):
        kp = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)[0]
        x2 = next(iter(net._xbuf.values()))
        print("forward", i, "joints diff", float((kp - kp0).abs().max()), "V2V input elements that differ:", int((x2 != x).sum()), "of", x.numel())
