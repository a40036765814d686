#!/usr/bin/env python3
"""Accuracy evaluation of saved predictions: MPJPE / PA-MPJPE of a directory of ``demo.py`` outputs.

The reference evaluates a sequence by collecting ``vol_keypoints_3d`` of every batch (``test.py:42-57``) and passing the list with
the ground truth through ``utils/calculate_errors.py``: ``align_skeleton`` (``:60-91``, per-pose similarity alignment) followed by
``calculate_error`` (``:22-28``).  Here the predictions are the ``<image name>.pkl`` files ``demo.py`` writes (one float32 [15,3]
array each, ``demo.py:88-97`` of the reference) and the ground truth is ONE pickle: either a dict ``{image name or stem: [15,3]}``
or a sequence / array [T,15,3] in the sorted order of the prediction files.  Host-side numpy (``sceneego_amd/metrics.py``); the
dataset classes of the reference (``dataset/test_dataset.py``) are not rebuilt - neither data nor weights ship with it.
"""
import argparse
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_predictions(pred_dir):
    names = sorted(n for n in os.listdir(pred_dir) if n.endswith(".pkl"))
    if not names:
        raise SystemExit(f"no .pkl predictions in {pred_dir}")
    poses = []
    for n in names:
        with open(os.path.join(pred_dir, n), "rb") as f:
            p = np.asarray(pickle.load(f), dtype=np.float64)
        if p.shape != (15, 3):
            raise SystemExit(f"{n}: expected a [15,3] pose, got {p.shape}")
        poses.append(p)
    return names, np.stack(poses)


def match_ground_truth(gt, names):
    if isinstance(gt, dict):
        def find(n):
            stem = n[:-4]                                    # "<image file name>.pkl" -> "<image file name>"
            for k in (stem, os.path.splitext(stem)[0], n):
                if k in gt:
                    return gt[k]
            raise SystemExit(f"ground truth has no entry for {stem}")
        return np.stack([np.asarray(find(n), dtype=np.float64) for n in names])
    arr = np.asarray(gt, dtype=np.float64)
    if arr.shape != (len(names), 15, 3):
        raise SystemExit(f"ground truth shape {arr.shape} does not match {len(names)} predictions of [15,3]")
    return arr


def evaluate(pred, gt, scale=True):
    from sceneego_amd import metrics as M
    return {"frames": int(pred.shape[0]), "mpjpe": M.mpjpe(pred, gt), "pa_mpjpe": M.pa_mpjpe(pred, gt, scale=scale),
            "per_joint": M.per_joint_error(pred, gt).tolist(), "root_trajectory": M.root_trajectory_error(pred, gt)}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--pred_dir", required=True, help="directory of <image name>.pkl files written by demo.py")
    ap.add_argument("--gt", required=True, help="pickle: dict name -> [15,3], or [T,15,3] in sorted file order")
    ap.add_argument("--no-scale", action="store_true", help="rigid instead of similarity alignment (align_skeleton(scale=False))")
    ap.add_argument("--unit", default="m", help="label only; the numbers are in the unit of the inputs")
    args = ap.parse_args(argv)
    names, pred = load_predictions(args.pred_dir)
    with open(args.gt, "rb") as f:
        gt = match_ground_truth(pickle.load(f), names)
    r = evaluate(pred, gt, scale=not args.no_scale)
    print(f"{r['frames']} frames  MPJPE {r['mpjpe']:.6f} {args.unit}  PA-MPJPE {r['pa_mpjpe']:.6f} {args.unit}  "
          f"root trajectory {r['root_trajectory']:.6f} {args.unit}")
    print("per joint: " + " ".join(f"{v:.4f}" for v in r["per_joint"]))
    return r


if __name__ == "__main__":
    main()
