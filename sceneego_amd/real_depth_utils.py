"""Dataset-side scene voxeliser (``voxel_output=True`` path): stands in for the reference's
``dataset/real_depth_utils.py`` (``depth_map_to_voxel`` ``:29-45``, ``point_cloud_to_voxel_pytorch`` ``:47-60``,
``calcualate_depth_scale`` ``:6-27`` — the reference's spelling is kept).

Unlike the network's own voxeliser (``voxel_net_depth.py:194-222``) there is no 1024x1024 resize and no padding: the rays of
the full ``W x H`` image are multiplied by the depth map directly.  Runs ``se_voxelize_full_f64`` on the GPU (float64,
bit-identical voxel set); the result is what ``VoxelNetwork_depth.forward(..., scene_volumes=...)`` takes.
"""
from __future__ import annotations

import hashlib
import json

import numpy as np
import torch

from . import _lib

_TABLE = None       # (ray array - held, so its identity cannot be reused -, content digest, height, width, device, table)


def calcualate_depth_scale(depth_scale_json_file, log_err=False):
    """Mean of real / measured distance over the annotated point pairs (reference ``:6-27``)."""
    with open(depth_scale_json_file, "r") as f:
        pairs = json.load(f)
    scales = [p["real"] / float(np.linalg.norm(np.asarray(p["x2"]) - np.asarray(p["x1"]))) for p in pairs]
    if log_err:
        print(scales)
        print(np.std(scales) / np.average(scales))
    return np.average(scales)


def _ray_table(ray, height, width, device):
    """x-major reference rays [W*H,3] (index x*H + y) -> device table [H][W][3] float64."""
    global _TABLE
    shape = (height, width, str(device))
    # keyed on CONTENT, hashed on every call (an `is`-shortcut for the same array object returned a stale table after an
    # in-place edit of the calibration rays; an `id()` key can be reused by another array).  blake2b over 31 MB: a few ms per item.
    r64 = np.ascontiguousarray(np.asarray(ray, dtype=np.float64))
    digest = hashlib.blake2b(r64.view(np.uint8).reshape(-1), digest_size=16).digest()
    if _TABLE is not None and _TABLE[0] == digest and _TABLE[1:4] == shape:
        return _TABLE[4]
    tab = torch.from_numpy(np.ascontiguousarray(r64.reshape(width, height, 3).transpose(1, 0, 2))).to(device)
    _TABLE = (digest,) + shape + (tab,)
    return tab


def depth_map_to_voxel(ray, depth, cuboid_side, volume_size, device="cuda"):
    """ray: [W*H,3] float64 unit rays (x-major, as ``calculated_ray_direction_numpy`` returns them); depth: [H,W] or
    [B,H,W] metres (numpy or tensor).  Returns a float32 {0,1} tensor [G,G,G] (or [B,G,G,G]) on ``device``."""
    d = torch.as_tensor(np.asarray(depth) if not isinstance(depth, torch.Tensor) else depth)
    single = d.dim() == 2
    if single:
        d = d[None]
    d = d.to(device=device, dtype=torch.float32).contiguous()
    B, H, W = d.shape
    tab = _ray_table(ray, H, W, d.device)
    occ = torch.empty((B, volume_size, volume_size, volume_size), device=d.device, dtype=torch.float32)
    _lib.voxelize_full(d, tab, occ, B, H, W, volume_size, float(cuboid_side))
    return occ[0] if single else occ
