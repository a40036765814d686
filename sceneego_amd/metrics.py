"""Pose error metrics of the reference's evaluation path, vectorised over the sequence.

Restates what ``utils/calculate_errors.py`` computes (``calculate_error`` ``:22-28`` = MPJPE, ``align_skeleton`` ``:60-91`` +
``calculate_error`` = PA-MPJPE, ``calculate_joint_error`` ``:94-100``, ``calculate_slam_error`` ``:31-46``) on top of the
similarity alignment of ``utils/rigid_transform_with_scale.py:18-43`` (Umeyama: ``Q ~ c P R + t`` for row-vector points).
Host-side numpy in float64; only meaningful with real weights and ground truth, neither of which ships with the reference.
"""
from __future__ import annotations

import numpy as np

LEFT_HIP, RIGHT_HIP = 11, 7      # indices in the 15-joint order of utils/skeleton.py:17-19


def umeyama(P, Q):
    """Least-squares similarity transform between corresponding point sets.

    P, Q: [..., n, d].  Returns (c [...], R [..., d, d], t [..., d]) with ``Q ~= c * (P @ R) + t``; reflections are excluded
    (the last singular direction is flipped when det < 0), exactly as the reference does."""
    P = np.asarray(P, dtype=np.float64)
    Q = np.asarray(Q, dtype=np.float64)
    if P.shape != Q.shape:
        raise ValueError(f"shape mismatch {P.shape} vs {Q.shape}")
    n = P.shape[-2]
    mp, mq = P.mean(axis=-2, keepdims=True), Q.mean(axis=-2, keepdims=True)
    cov = np.swapaxes(P - mp, -1, -2) @ (Q - mq) / n
    U, S, Vt = np.linalg.svd(cov)
    flip = (np.linalg.det(U) * np.linalg.det(Vt)) < 0.0
    S = S.copy()
    U = U.copy()
    S[..., -1] = np.where(flip, -S[..., -1], S[..., -1])
    U[..., :, -1] = np.where(flip[..., None], -U[..., :, -1], U[..., :, -1])
    R = U @ Vt
    c = S.sum(axis=-1) / P.var(axis=-2).sum(axis=-1)
    t = mq[..., 0, :] - (mp @ (c[..., None, None] * R))[..., 0, :]
    return c, R, t


def mpjpe(estimated, gt) -> float:
    """Mean per-joint position error over a sequence [T, J, 3] (same unit as the input)."""
    e = np.asarray(estimated, dtype=np.float64) - np.asarray(gt, dtype=np.float64)
    return float(np.linalg.norm(e, axis=-1).mean())


def procrustes_align(estimated, gt, scale: bool = True):
    """Per-pose similarity (or rigid, ``scale=False``: both poses are centred first) alignment of estimated [T,J,3] onto gt.
    Returns (aligned estimate, gt as used)."""
    est = np.array(estimated, dtype=np.float64)
    ref = np.array(gt, dtype=np.float64)
    if not scale:
        est -= est.mean(axis=1, keepdims=True)
        ref -= ref.mean(axis=1, keepdims=True)
    c, R, t = umeyama(est, ref)
    if scale:
        out = c[:, None, None] * (est @ R) + t[:, None, :]
    else:
        out = est @ R + t[:, None, :]
    return out, ref


def pa_mpjpe(estimated, gt, scale: bool = True) -> float:
    aligned, ref = procrustes_align(estimated, gt, scale)
    return mpjpe(aligned, ref)


def per_joint_error(estimated, gt):
    """[J] mean Euclidean error of every joint over the sequence."""
    e = np.asarray(estimated, dtype=np.float64) - np.asarray(gt, dtype=np.float64)
    return np.linalg.norm(e, axis=-1).mean(axis=0)


def root_trajectory_error(estimated, gt, align: bool = False) -> float:
    """Mean error of the hip-centre trajectory; ``align`` fits one similarity transform to the whole trajectory first."""
    est = np.asarray(estimated, dtype=np.float64)
    ref = np.asarray(gt, dtype=np.float64)
    re = 0.5 * (est[:, RIGHT_HIP] + est[:, LEFT_HIP])
    rg = 0.5 * (ref[:, RIGHT_HIP] + ref[:, LEFT_HIP])
    if align:
        c, R, t = umeyama(re, rg)
        re = c * (re @ R) + t
    return float(np.linalg.norm(re - rg, axis=1).mean())


def global_align_sequence(estimated, gt):
    """One similarity transform for the whole sequence (all joints of all frames pooled), reshaped back to [T,J,3]."""
    est = np.asarray(estimated, dtype=np.float64)
    c, R, t = umeyama(est.reshape(-1, 3), np.asarray(gt, dtype=np.float64).reshape(-1, 3))
    return (c * (est.reshape(-1, 3) @ R) + t).reshape(est.shape)
