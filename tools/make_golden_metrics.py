"""Golden vectors for sceneego_amd/metrics.py: the reference's own Umeyama alignment
(/root/reference/utils/rigid_transform_with_scale.py:18-43, imported here; runs only in the build container) on seeded poses.
Writes tests/golden/metrics.npz; tests/test_host_logic.py rebuilds the same inputs from the seeds."""
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sceneego_amd import synth  # noqa: E402


def inputs():
    est = synth.normal(11, "metrics/est", (6, 15, 3)).astype(np.float64)
    rot = np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]])
    gt = est @ rot * 1.3 + np.array([0.1, -0.2, 0.5]) + 0.05 * synth.normal(12, "metrics/noise", (6, 15, 3))
    return est, gt


def main():
    spec = importlib.util.spec_from_file_location("ref_rt", "/root/reference/utils/rigid_transform_with_scale.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    est, gt = inputs()
    per = [ref.umeyama(est[s], gt[s]) for s in range(len(est))]
    aligned = np.stack([est[s].dot(R) * c + t for s, (c, R, t) in enumerate(per)])     # calculate_errors.py:86-88
    cg, Rg, tg = ref.umeyama(est.reshape(-1, 3), gt.reshape(-1, 3))                     # calculate_errors.py:8-19
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics.npz"),
                        c=np.array([p[0] for p in per]), R=np.stack([p[1] for p in per]), t=np.stack([p[2] for p in per]),
                        pa_mpjpe=np.array(np.linalg.norm(aligned - gt, axis=2).mean()),
                        mpjpe=np.array(np.linalg.norm(est - gt, axis=2).mean()),
                        global_aligned=(est.reshape(-1, 3).dot(Rg) * cg + tg).reshape(-1, 15, 3))


if __name__ == "__main__":
    main()
