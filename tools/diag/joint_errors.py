"""Diagnostic: joint error of the float32 HIP forward vs every golden case (and vs the oracle run on this host)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import case_inputs, GOLD
from sceneego_amd import load_config, synth
from sceneego_amd.voxel_net_depth import VoxelNetwork_depth
meta = json.load(open(os.path.join(GOLD, "META.json")))
for m in meta["cases"]:
    cfg = load_config(); cfg.model.with_intersection = m["with_intersection"]; cfg.model.volume_size = m["volume_size"]
    cfg.model.with_scene = m.get("with_scene", True); cfg.model.volume_softmax = m.get("volume_softmax", True)
    cfg.model.volume_multiplier = m.get("volume_multiplier", 1.0)
    if len(sys.argv) > 1 and sys.argv[1] == "bf16":
        cfg.model.v2v_dtype = "bf16"
        if len(sys.argv) > 2:
            cfg.model.backbone_dtype = "bf16"
    net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
    net.load_state_dict(synth.make_state_dict(net.state_dict(), seed=m["weight_seed"]), strict=True)
    net = net.to("cuda:0").eval()
    if len(sys.argv) > 1 and sys.argv[1] == "split_bf16":      # float32 tensors, 3x3x3 layers with split-bf16 arithmetic
        net.set_v2v_dtype("split_bf16")
    img, depth = case_inputs(m)
    g = dict(np.load(os.path.join(GOLD, m["name"] + ".npz")))
    errs = []
    for rep in range(3):
        with torch.no_grad():
            kp, _, vols, _ = net(img.to("cuda:0"), net.grid_coord_proj_batch, net.coord_volumes,
                                 depth_map_batch=depth.to("cuda:0") if cfg.model.with_scene else None)
        errs.append(float(np.abs(kp.cpu().numpy() - g["joints"]).max()))
    vmax = vols.reshape(vols.shape[0], vols.shape[1], -1).max(dim=2)[0].cpu().numpy()
    print(f"{m['name']:18s} joint err {errs}  peak softmax prob: min {vmax.min():.3f} median {np.median(vmax):.3f}  rel peak err {np.abs(vmax - g['volumes_max']).max() / g['volumes_max'].max():.2e}")
    del net
