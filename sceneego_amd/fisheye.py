"""Calibrated polynomial fisheye model — the two directions the hot path uses at init time.

Restates ``FishEyeCameraCalibrated`` of the reference (``utils/fisheye/FishEyeCalibrated.py``):
  * ``__init__``            -> ``:8-16``   (reads intrinsic[0][2], intrinsic[1][2], size, both polynomials)
  * ``camera2world_ray``    -> ``:36-51``  (pixel -> unit ray, numpy float64)
  * ``camera2world``        -> ``:19-34``  (pixel + depth -> point; used by utils/depth2pointcloud.py:32)
  * ``world2camera_pytorch``-> ``:137-187``(3D point -> pixel, torch float32, raises on |xy| == 0; ``normalize`` -> [-1, 1])
  * ``world2camera``        -> ``:93-123`` (numpy twin; used by utils/multiview.py:126)
Everything here runs once, on the host, when the network is constructed; the results are uploaded
as constant tables for the HIP kernels (voxeliser ray table, gather tap table).  The operation
ORDER is kept (explicit running power for the W2C polynomial, ``np.polyval`` for C2W) because the
tables feed bit-exact (voxeliser) and <=1e-6 (gather) parity checks.
"""
from __future__ import annotations

import json

import numpy as np
import torch


class FishEyeCameraCalibrated:
    def __init__(self, calibration_file_path, use_gpu=False):
        with open(calibration_file_path) as f:
            calib = json.load(f)
        self.intrinsic = np.array(calib["intrinsic"])
        self.img_size = np.array(calib["size"])  # (w, h)
        self.fisheye_polynomial = np.array(calib["polynomialC2W"])          # ascending powers of r [px]
        self.fisheye_inverse_polynomial = np.array(calib["polynomialW2C"])  # ascending powers of theta
        self.img_center = np.array([self.intrinsic[0][2], self.intrinsic[1][2]])
        self.use_gpu = use_gpu

    # -- pixel -> ray -------------------------------------------------------------------------
    def camera2world_ray(self, point: np.ndarray) -> np.ndarray:
        """(n,2) pixel coordinates -> (n,3) unit rays, float64 (reference ``:36-51``)."""
        centred = point.astype(np.float64) - self.img_center
        x = centred[:, 0]
        y = centred[:, 1]
        r = np.sqrt(np.square(x) + np.square(y))
        z = np.polyval(self.fisheye_polynomial[::-1], r)
        p = np.array([x, y, -z])  # (3, n)
        p = p / np.linalg.norm(p, axis=0)
        return p.transpose()

    def camera2world(self, point: np.ndarray, depth: np.ndarray) -> np.ndarray:
        """(n,2) pixels + (n,) depths -> (n,3) points along the pixel's ray (reference ``:19-34``; float32 inputs)."""
        depth = depth.astype(np.float32)
        centred = point.astype(np.float32) - self.img_center
        x = centred[:, 0]
        y = centred[:, 1]
        r = np.sqrt(np.square(x) + np.square(y))
        z = np.polyval(self.fisheye_polynomial[::-1], r)
        p = np.array([x, y, -z])
        p = p / np.linalg.norm(p, axis=0) * depth
        return p.transpose()

    # -- 3D point -> pixel --------------------------------------------------------------------
    def world2camera_pytorch(self, point3d_original: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """(n,3) float32 points (camera frame, z forward-down) -> (n,2) pixels (reference ``:137-187``); ``normalize``: to
        [-1, 1] of the centred square crop (``:178-185``)."""
        poly = self.fisheye_inverse_polynomial
        p = point3d_original.clone()
        p[:, 2] = point3d_original[:, 2] * -1
        p = p.transpose(0, 1)
        dev = p.device
        xc = torch.Tensor([self.img_center[0]]).float().to(dev)
        yc = torch.Tensor([self.img_center[1]]).float().to(dev)
        out = torch.empty((2, p.shape[-1])).to(dev)

        norm = torch.norm(p[:2], dim=0)
        if not bool((norm != 0).all()):
            raise Exception("norm is zero!")
        theta = torch.atan(p[2] / norm)
        invnorm = 1.0 / norm
        rho = poly[0]
        t_i = 1.0
        for i in range(1, len(poly)):
            t_i = t_i * theta          # running power, NOT Horner (keeps the reference's rounding)
            rho = rho + t_i * poly[i]
        out[0] = p[0] * invnorm * rho + xc
        out[1] = p[1] * invnorm * rho + yc

        if normalize:
            w, h = self.img_size[0], self.img_size[1]
            assert w > h
            out[0] = out[0] - (w - h) // 2
            out = out / (h - 1) * 2
            out -= 1
        return out.transpose(0, 1)

    def world2camera(self, point3D: np.ndarray) -> np.ndarray:
        """numpy float64 twin of ``world2camera_pytorch`` (reference ``:93-123``)."""
        p = np.array(point3D, dtype=np.float64, copy=True)
        p[:, 2] = p[:, 2] * -1
        p = p.T
        norm = np.linalg.norm(p[:2], axis=0)
        if not (norm != 0).all():
            raise Exception("norm is zero!")
        theta = np.arctan(p[2] / norm)
        invnorm = 1.0 / norm
        rho = self.fisheye_inverse_polynomial[0]
        t_i = 1.0
        for i in range(1, len(self.fisheye_inverse_polynomial)):
            t_i = t_i * theta
            rho = rho + t_i * self.fisheye_inverse_polynomial[i]
        return np.asarray([p[0] * invnorm * rho + self.img_center[0],
                           p[1] * invnorm * rho + self.img_center[1]]).T
