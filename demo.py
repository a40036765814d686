#!/usr/bin/env python3
"""Drop-in for the reference's ``demo.py`` (``demo.py:19-101``) on MI355X: image dir + depth dir -> one pickle of
15x3 joints per image.

    python demo.py --config experiments/sceneego/test/sceneego.yaml --img_dir data/demo/imgs \\
                   --depth_dir data/demo/depths --output_dir data/demo/out [--weights synthetic]

Differences: ``--vis`` (open3d GUI) is out of scope; depth maps are read from ``<img_name>.exr`` (the reference's format:
scanline OpenEXR, NONE/ZIP/PIZ, decoded by ``sceneego_amd/exr.py``) or ``.npy`` / ``.npz``; ``--weights synthetic`` uses the portable seeded weights when no checkpoint exists
(``config.test.model_path`` is loaded strictly otherwise, exactly like ``demo.py:29-31``).
"""
import argparse
import os
import pickle

import torch

from sceneego_amd import load_config, synth
from sceneego_amd.preprocess import (DEPTH_CLAMP, load_depth, load_image_bgr, prepare_depth, preprocess_image,
                                     preprocess_image_device)
from sceneego_amd.voxel_net_depth import VoxelNetwork_depth

JOINT_NAMES = ["Neck", "Right_shoulder", "Right_elbow", "Right_wrist", "Left_shoulder", "Left_elbow", "Left_wrist",
               "Right_hip", "Right_knee", "Right_ankle", "Right_foot", "Left_hip", "Left_knee", "Left_ankle",
               "Left_foot"]   # heatmap_sequence, reference utils/skeleton.py:17-19


class Demo:
    def __init__(self, config, img_dir, depth_dir, weights=None):
        if not torch.cuda.is_available():
            raise RuntimeError("demo.py needs an MI355X (HIP device); the hot path has no CPU fallback")
        self.device = torch.device("cuda")
        self.config = config
        self.items = []
        for img_name in sorted(os.listdir(img_dir)):
            img_path = os.path.join(img_dir, img_name)
            cands = [os.path.join(depth_dir, img_name + ext) for ext in (".npy", ".npz", ".exr")]
            depth_path = next((c for c in cands if os.path.exists(c)), None)
            if depth_path is None:
                raise Exception(f"The depth map {cands[-1]} does not exist!")
            self.items.append((img_path, depth_path))
        self.network = VoxelNetwork_depth(config, device="cpu")
        if weights == "synthetic":
            self.network.load_state_dict(synth.make_state_dict(self.network.state_dict(), seed=0), strict=True)
        else:
            loads = torch.load(weights or config.test.model_path, map_location="cpu")
            self.network.load_state_dict(loads["state_dict"])
        self.network = self.network.to(self.device).eval()
        self.network.enable_graphs(True)       # batch 1: replay the captured forward

    def run(self):
        results = []
        with torch.no_grad():
            for img_path, depth_path in self.items:
                frame = load_image_bgr(img_path)
                W, H = self.config.dataset.image_width, self.config.dataset.image_height
                if frame.shape[:2] == (4 * self.config.image_shape[0], 4 * self.config.image_shape[1] + 256):
                    # raw uint8 frame to the device; crop / quarter-resize / normalise there (se_preprocess_image_u8)
                    img = preprocess_image_device(torch.from_numpy(frame).to(self.device), self.config.image_shape)
                else:
                    img = preprocess_image(frame, self.config.image_shape)[None].to(self.device)
                d = load_depth(depth_path)
                if d.shape == (H // 2, W // 2):
                    # half-size depth map (the demo EXRs): upload as is, clamp on the device; the voxeliser's nearest
                    # lookup floor(x * 640 / 1024) equals the reference's two nearest resizes floor(floor(1.25 x) / 2)
                    depth = torch.from_numpy(d).to(self.device).clamp_(max=DEPTH_CLAMP)[None]
                else:
                    depth = prepare_depth(d, W, H)[None].to(self.device)
                kp, _, _, _ = self.network(img, self.network.grid_coord_proj_batch, self.network.coord_volumes,
                                           depth_map_batch=depth)
                assert len(kp) == 1
                results.append({"img_path": img_path, "predicted_keypoints": kp.cpu().numpy()[0]})
        return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="experiments/sceneego/test/sceneego.yaml")
    ap.add_argument("--img_dir", type=str, default="data/demo/imgs")
    ap.add_argument("--depth_dir", type=str, default="data/demo/depths")
    ap.add_argument("--output_dir", type=str, default="data/demo/out")
    ap.add_argument("--vis", type=str, default="false")
    ap.add_argument("--weights", type=str, default=None, help="checkpoint path, or 'synthetic'")
    args = ap.parse_args()
    if args.vis.lower() == "true":
        raise SystemExit("--vis true (open3d visualisation) is out of scope of this build")
    config = load_config(args.config)
    demo = Demo(config, args.img_dir, args.depth_dir, weights=args.weights)
    os.makedirs(args.output_dir, exist_ok=True)
    for r in demo.run():
        out_path = os.path.join(args.output_dir, os.path.split(r["img_path"])[1] + ".pkl")
        with open(out_path, "wb") as f:
            pickle.dump(r["predicted_keypoints"], f)      # np.float32 [15,3], reference demo.py:88-97
        print(out_path, r["predicted_keypoints"][0])


if __name__ == "__main__":
    main()
