"""Measure the per-BatchNorm-layer scalars of sceneego_amd/synth_calibration.json (build container or any CPU box).

One forward of the CPU oracle on the seeded golden inputs; at every BatchNorm the pre-BN activation's
global mean m and std s are recorded and that layer's running statistics are set to
(m + s*U(-0.2,0.2), s^2*U(0.5,1.5)) before continuing, so later layers are measured on the calibrated
network.  Finally the output-layer gain is chosen so that the 3D logits have std ~ 6.
Usage: python tools/calibrate_synth.py   (writes sceneego_amd/synth_calibration.json)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sceneego_oracle as O  # noqa: E402
from sceneego_amd import load_config, synth  # noqa: E402
from sceneego_amd.voxel_net_depth import VoxelNetwork_depth  # noqa: E402

TARGET_LOGIT_STD = 6.0


def main():
    cfg = load_config()
    net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
    ref_sd = net.state_dict()
    seed = 0
    sd = synth.make_state_dict(ref_sd, seed=seed, calibration={"bn": {}, "output_gain": 1.0})
    const = O.Constants(os.path.join(ROOT, "sceneego_amd", "calibration", "fisheye.calibration_05_08.json"))
    img, depth = synth.make_inputs(1234, 2, "uniform")
    cal = {}

    def hooked(sd_, p, x):
        m = float(x.mean())
        s = float(x.std())
        cal[p] = [m, s]
        shape = tuple(sd_[p + ".running_mean"].shape)
        sd_[p + ".running_mean"] = torch.from_numpy(
            (synth.uniform(seed, p + ".running_mean", shape, -0.2, 0.2) * np.float32(s) + np.float32(m)).astype(np.float32))
        sd_[p + ".running_var"] = torch.from_numpy(
            (synth.uniform(seed, p + ".running_var", shape, 0.5, 1.5) * np.float32(s) * np.float32(s)).astype(np.float32))
        return torch.nn.functional.batch_norm(x, sd_[p + ".running_mean"], sd_[p + ".running_var"], sd_[p + ".weight"],
                                              sd_[p + ".bias"], False, 0.1, O.BN_EPS)

    O._bn = hooked
    O._bn3 = hooked
    taps = {}
    O.forward(sd, const, img, depth, taps=taps)
    gain = TARGET_LOGIT_STD / float(taps["logits"].std())
    out = {"seed": seed, "input_seed": 1234, "target_logit_std": TARGET_LOGIT_STD, "output_gain": gain, "bn": cal}
    path = os.path.join(ROOT, "sceneego_amd", "synth_calibration.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, "layers", len(cal), "gain", gain)
    for k in ("features64", "front0", "front3", "enc3", "mid", "dec1", "back2", "logits"):
        v = taps[k]
        print(f"{k:12s} mean|x|={float(v.abs().mean()):.4g} std={float(v.std()):.4g}")


if __name__ == "__main__":
    main()
