"""Diagnostic: how accurate must the logits be for a given joint accuracy (synthetic weights), and where the bf16 program stands."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import case_inputs, synthetic_state_dict, GOLD
from sceneego_amd import load_config, _lib
from sceneego_amd.voxel_net_depth import VoxelNetwork_depth
meta = json.load(open(os.path.join(GOLD, "META.json")))
m = next(c for c in meta["cases"] if c["name"] == "b2_uniform")
cfg = load_config()
net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
net.load_state_dict(synthetic_state_dict(False, 0), strict=True)
net = net.to("cuda:0").eval()
img, depth = case_inputs(m)
img, depth = img.cuda(), depth.cuda()
g = dict(np.load(os.path.join(GOLD, "b2_uniform.npz")))
# capture logits by hooking softargmax3d
cap = {}
orig = _lib.softargmax3d
def hook(vol, coord, out_vol, joints, rows, voxels, mode, scratch=None):
    cap["logits"] = vol.clone(); cap["coord"] = coord
    return orig(vol, coord, out_vol, joints, rows, voxels, mode, scratch)
_lib.softargmax3d = hook
import sceneego_amd.voxel_net_depth as V
kp32 = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)[0].clone()
lg32 = cap["logits"].double()
print("fp32 joints vs golden:", float(np.abs(kp32.cpu().numpy() - g["joints"]).max()), " logits std", float(lg32.std()), "max", float(lg32.abs().max()))
coord = cap["coord"].double()
def joints_of(lg):
    p = torch.softmax(lg, dim=-1)
    return p @ coord
j0 = joints_of(lg32)
peak = torch.softmax(lg32, dim=-1).max(dim=-1)[0]
print("softmax peak prob: min %.3g median %.3g max %.3g" % (float(peak.min()), float(peak.median()), float(peak.max())))
for eps in (1e-4, 3e-4, 1e-3, 3e-3, 1e-2):
    errs = []
    for rep in range(3):
        noise = torch.randn_like(lg32) * eps * lg32.abs().max()
        errs.append(float((joints_of(lg32 + noise) - j0).abs().max()))
    print(f"iid logit noise {eps:.0e} x max|logit| -> joint change {max(errs):.2e} m")
for rel in (1e-3, 3e-3, 1e-2):
    errs = []
    for rep in range(3):
        noise = torch.randn_like(lg32) * rel * lg32.abs()
        errs.append(float((joints_of(lg32 + noise) - j0).abs().max()))
    print(f"relative logit noise {rel:.0e} -> joint change {max(errs):.2e} m")
net.set_v2v_dtype("bf16")
kpb = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)[0].clone()
lgb = cap["logits"].double()
d = lgb - lg32
print("bf16 program: joints vs fp32 program %.2e m; logits abs err max %.3e rms %.3e (max|logit| %.2f); correlation of error with logit: %.3f"
      % (float((kpb - kp32).abs().max()), float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(lg32.abs().max()),
         float(((d - d.mean()) * (lg32 - lg32.mean())).mean() / (d.std() * lg32.std()))))
# smooth (spatially correlated) error matters more than iid noise: error after removing a global affine fit per row
a = (d * lg32).sum(-1, keepdim=True) / (lg32 * lg32).sum(-1, keepdim=True)
print("after removing the per-row gain error: residual rms %.3e ; gain error (a) range %.4f .. %.4f" % (float((d - a * lg32).pow(2).mean().sqrt()), float(a.min()), float(a.max())))
print("joint change if only the gain error is applied: %.2e m" % float((joints_of(lg32 * (1 + a)) - j0).abs().max()))
