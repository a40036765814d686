"""Diagnostic: float32 HIP forward vs the oracle (float64 soft-argmax evaluation) on extra seeded inputs, both depth kinds."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synthetic_state_dict
from oracle import sceneego_oracle as O
from sceneego_amd import load_config, synth
from sceneego_amd.voxel_net_depth import VoxelNetwork_depth
cfg = load_config()
net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
sd = synthetic_state_dict(False, 0)
net.load_state_dict(sd, strict=True)
net = net.to("cuda:0").eval()
if len(sys.argv) > 1 and sys.argv[1] == "split_bf16":
    net.set_v2v_dtype("split_bf16")
const = O.Constants(os.path.join(ROOT, "sceneego_amd", "calibration", "fisheye.calibration_05_08.json"))
worst = 0.0
# batch sizes 2, 3, 5: odd batches put a persistent workgroup's unit range across a sample boundary (conv3d_wino44pp.hip `advance`)
for seed, nb in ((3, 2), (19, 3), (101, 5), (2027, 2), (5150, 3), (90210, 5)):
    for kind in ("uniform", "floor"):
        img, depth = synth.make_inputs(seed, nb, kind)
        with torch.no_grad():
            kp = net(img.to("cuda:0"), net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth.to("cuda:0"))[0].cpu()
        taps = {}
        oj, _, _ = O.forward(sd, const, img, depth, taps=taps, accumulate64=True)
        err = float((kp - oj).abs().max())
        worst = max(worst, err)
        print(f"seed {seed:6d} {kind:8s} max|hip - oracle| = {err:.2e}")
print("worst", worst)
assert worst <= 1e-3
