"""ORACLE — CPU restatement of SceneEgo's depth-aware voxel pose forward.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file,
and only as the checker / the timed CPU baseline.  Nothing under ``sceneego_amd/`` imports it; the product
path has no CPU fallback.

What it is: a functional (state-dict in, tensors out) re-statement, in plain PyTorch-CPU + numpy, of the
arithmetic ``VoxelNetwork_depth.forward`` executes in the reference — the SAME ATen CPU operators in the
same order, float32 (float64 in the voxeliser), including the literal 1024x1280 upsample+pad +
``grid_sample`` formulation the product replaces by a table.  It shares no code with ``sceneego_amd``.

Pinning: the reference has no tests/golden vectors of its own.  This oracle is pinned
against outputs of the reference itself, imported in the build container by ``tools/make_golden.py``
(shimmed cv2/easydict, numpy aliases, and the tuple-index patch of ``voxel_net_depth.py:221``); those
outputs are committed under ``tests/golden/`` and ``tests/test_oracle_golden.py`` checks this file against
them.  Third-party arithmetic: PyTorch ATen CPU kernels (reference pins torch 1.13.1 in prose, README.md:60;
this image has 2.10) and OpenCV's INTER_NEAREST resize (unpinned in requirements.txt:4; absent here —
restated as floor(dst * src/dst_size); that one line is "parity unpinned" against real OpenCV).

Each function cites the reference file:line it follows (paths relative to the reference repo).
"""
from __future__ import annotations

import json

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# ------------------------------------------------------------------------------------------------
# init-time constants
# ------------------------------------------------------------------------------------------------
class Calibration:
    """utils/fisheye/FishEyeCalibrated.py:8-16."""

    def __init__(self, path):
        with open(path) as f:
            d = json.load(f)
        intr = np.array(d["intrinsic"])
        self.center = np.array([intr[0][2], intr[1][2]])
        self.c2w = np.array(d["polynomialC2W"])
        self.w2c = np.array(d["polynomialW2C"])
        self.size = np.array(d["size"])


def coord_volume(G, side):
    """network/voxel_net_depth.py:110-134 — [G,G,G,3] float32, x,y in [-side/2, side/2], z in [0, side]."""
    i = torch.arange(G)
    gx, gy, gz = torch.meshgrid(i, i, i, indexing="ij")
    grid = torch.stack([gx, gy, gz], dim=-1).type(torch.float).reshape(-1, 3)
    pos = np.array([-side / 2, -side / 2, 0])
    sides = np.array([side, side, side])
    out = torch.zeros_like(grid)
    out[:, 0] = pos[0] + (sides[0] / (G - 1)) * grid[:, 0]
    out[:, 1] = pos[1] + (sides[1] / (G - 1)) * grid[:, 1]
    out[:, 2] = pos[2] + (sides[2] / (G - 1)) * grid[:, 2]
    return out.reshape(G, G, G, 3)


def project_fisheye(calib, pts):
    """utils/fisheye/FishEyeCalibrated.py:137-177 (float32 torch; running power of theta, not Horner)."""
    p = pts.clone()
    p[:, 2] = pts[:, 2] * -1
    p = p.transpose(0, 1)
    xc = torch.Tensor([calib.center[0]]).float()
    yc = torch.Tensor([calib.center[1]]).float()
    norm = torch.norm(p[:2], dim=0)
    if not (norm != 0).all():
        raise Exception("norm is zero!")
    theta = torch.atan(p[2] / norm)
    inv = 1.0 / norm
    rho = calib.w2c[0]
    t_i = 1.0
    for k in range(1, len(calib.w2c)):
        t_i = t_i * theta
        rho = rho + t_i * calib.w2c[k]
    out = torch.empty((2, p.shape[-1]))
    out[0] = p[0] * inv * rho + xc
    out[1] = p[1] * inv * rho + yc
    return out.transpose(0, 1)


def normalised_grid(proj, heatmap_shape):
    """utils/op.py:177-184 (without the batch expand): [N,2] in grid_sample's [-1,1]."""
    g = torch.zeros_like(proj)
    g[:, 0] = 2 * (proj[:, 0] / heatmap_shape[1] - 0.5)
    g[:, 1] = 2 * (proj[:, 1] / heatmap_shape[0] - 0.5)
    return g


def rays(calib, width, height):
    """network/voxel_net_depth.py:147-155 + FishEyeCalibrated.py:36-51 — [W*H,3] float64, flat = x*H + y."""
    pts = np.zeros((width, height, 2))
    pts[:, :, 0] = np.arange(width)[:, None]
    pts[:, :, 1] = np.arange(height)[None, :]
    pts = pts.reshape(-1, 2)
    c = pts.astype(np.float64) - calib.center
    x, y = c[:, 0], c[:, 1]
    r = np.sqrt(np.square(x) + np.square(y))
    z = np.polyval(calib.c2w[::-1], r)
    p = np.array([x, y, -z])
    p = p / np.linalg.norm(p, axis=0)
    return p.transpose()


class Constants:
    def __init__(self, calibration_path, G=64, side=2, heatmap_shape=(1024, 1280), width=1280, height=1024):
        self.G, self.side, self.heatmap_shape = G, side, tuple(heatmap_shape)
        self.width, self.height = width, height
        self.calib = Calibration(calibration_path)
        self.coord = coord_volume(G, side)
        self.proj = project_fisheye(self.calib, self.coord.reshape(-1, 3))
        self.grid = normalised_grid(self.proj, heatmap_shape)
        self.ray = rays(self.calib, width, height)


# ------------------------------------------------------------------------------------------------
# 2D backbone  (network/pose_resnet.py)
# ------------------------------------------------------------------------------------------------
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.1, BN_EPS)


def _bottleneck(sd, p, x, stride):
    """network/pose_resnet.py:52-90."""
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    y = F.relu(_bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], stride=stride, padding=1)))
    y = _bn(sd, p + ".bn3", F.conv2d(y, sd[p + ".conv3.weight"]))
    if (p + ".downsample.0.weight") in sd:
        x = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride))
    return F.relu(y + x)


def backbone(sd, images, prefix="backbone"):
    """network/pose_resnet.py:225-246 -> features [B,256,64,64] (heatmaps are dead on this path, :235 of voxel_net_depth)."""
    p = prefix
    x = F.relu(_bn(sd, p + ".bn1", F.conv2d(images, sd[p + ".conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2)), start=1):
        for bi in range(blocks):
            x = _bottleneck(sd, f"{p}.layer{li}.{bi}", x, stride if bi == 0 else 1)
    for i in (0, 3, 6):  # ConvTranspose2d k4 s2 p1, no bias (:198-223)
        x = F.conv_transpose2d(x, sd[f"{p}.deconv_layers.{i}.weight"], stride=2, padding=1)
        x = F.relu(_bn(sd, f"{p}.deconv_layers.{i + 1}", x))
    return x


def process_features(sd, feats):
    """network/voxel_net_depth.py:58-62,238 — 1x1 conv, nearest upsample to 1024^2, zero-pad 128 cols."""
    x = F.conv2d(feats, sd["process_features.0.weight"], sd["process_features.0.bias"])
    x = F.interpolate(x, size=(1024, 1024), mode="nearest")
    return F.pad(x, (128, 128, 0, 0), value=0.0)


def unproject(features_big, grid, G):
    """utils/op.py:194-214 — grid_sample(bilinear, zeros, align_corners=True) -> [B,C,G,G,G]."""
    B, C = features_big.shape[:2]
    g = grid.unsqueeze(1).unsqueeze(0).expand(B, -1, -1, -1)
    v = F.grid_sample(features_big, g, align_corners=True)
    return v.view(B, C, G, G, G)


# ------------------------------------------------------------------------------------------------
# voxeliser  (network/voxel_net_depth.py:194-222), per-point index semantics
# ------------------------------------------------------------------------------------------------
def resize_nearest(src, w, h):
    """cv2.resize(src, (w,h), INTER_NEAREST): src index = min(floor(dst * (src/dst)), src-1)."""
    ys = np.minimum(np.floor(np.arange(h) * (src.shape[0] / h)).astype(np.int64), src.shape[0] - 1)
    xs = np.minimum(np.floor(np.arange(w) * (src.shape[1] / w)).astype(np.int64), src.shape[1] - 1)
    return src[ys][:, xs]


def point_cloud_to_voxel(pc, G, side):
    """network/voxel_net_depth.py:207-222 (float64; ``voxel[idx.T] = 1`` with torch<=2.8 tuple semantics)."""
    p = pc.copy()
    p[:, 0] = (p[:, 0] + side / 2) * G / side
    p[:, 1] = (p[:, 1] + side / 2) * G / side
    p[:, 2] = (p[:, 2]) * G / side
    p = np.round(p)
    good = np.all(np.logical_and(G - 1 >= p, p >= 0), axis=1)
    p = p[good].astype(np.int64)
    vox = torch.zeros((G, G, G))
    vox[p[:, 0], p[:, 1], p[:, 2]] = 1
    return vox


def depth_to_voxel(depth, ray, G, side):
    """network/voxel_net_depth.py:194-205 — one [H,W] float32 depth map -> [G,G,G] occupancy."""
    d = np.asarray(depth, dtype=np.float32)
    d = resize_nearest(d, 1024, 1024)
    d = np.pad(d, ((0, 0), (128, 128)), "constant", constant_values=0)
    flat = d.T.reshape(-1)
    pc = (ray.T * flat).T
    return point_cloud_to_voxel(pc, G, side)


def depth_to_voxel_full(depth, ray, G, side):
    """dataset/real_depth_utils.py:29-60 — no resize/pad (``voxel_output=True`` path)."""
    d = np.asarray(depth, dtype=np.float32)
    pc = (ray.T * d.T.reshape(-1)).T
    return point_cloud_to_voxel(pc, G, side)


# ------------------------------------------------------------------------------------------------
# V2V  (network/v2v.py)
# ------------------------------------------------------------------------------------------------
def _bn3(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.1, BN_EPS)


def _basic3d(sd, p, x):
    """v2v.py:8-18."""
    w = sd[p + ".block.0.weight"]
    return F.relu(_bn3(sd, p + ".block.1", F.conv3d(x, w, sd[p + ".block.0.bias"], padding=(w.shape[2] - 1) // 2)))


def _res3d(sd, p, x):
    """v2v.py:21-43."""
    r = F.relu(_bn3(sd, p + ".res_branch.1", F.conv3d(x, sd[p + ".res_branch.0.weight"], sd[p + ".res_branch.0.bias"], padding=1)))
    r = _bn3(sd, p + ".res_branch.4", F.conv3d(r, sd[p + ".res_branch.3.weight"], sd[p + ".res_branch.3.bias"], padding=1))
    if (p + ".skip_con.0.weight") in sd:
        x = _bn3(sd, p + ".skip_con.1", F.conv3d(x, sd[p + ".skip_con.0.weight"], sd[p + ".skip_con.0.bias"]))
    return F.relu(r + x)


def _up3d(sd, p, x):
    """v2v.py:55-67."""
    y = F.conv_transpose3d(x, sd[p + ".block.0.weight"], sd[p + ".block.0.bias"], stride=2)
    return F.relu(_bn3(sd, p + ".block.1", y))


def v2v(sd, x, prefix="volume_net", taps=None):
    """v2v.py:165-170 with the EncoderDecorder of :104-139.  ``taps``: optional dict collecting named intermediates."""
    p = prefix
    rec = (lambda n, t: taps.__setitem__(n, t)) if taps is not None else (lambda n, t: None)
    x = _basic3d(sd, p + ".front_layers.0", x); rec("front0", x)
    for i in (1, 2, 3):
        x = _res3d(sd, f"{p}.front_layers.{i}", x); rec(f"front{i}", x)
    e = p + ".encoder_decoder"
    skips = []
    for k in range(1, 6):
        skips.append(_res3d(sd, f"{e}.skip_res{k}", x)); rec(f"skip{k}", skips[-1])
        x = F.max_pool3d(x, 2, 2)
        x = _res3d(sd, f"{e}.encoder_res{k}", x); rec(f"enc{k}", x)
    x = _res3d(sd, e + ".mid_res", x); rec("mid", x)
    for k in range(5, 0, -1):
        x = _res3d(sd, f"{e}.decoder_res{k}", x)
        x = _up3d(sd, f"{e}.decoder_upsample{k}", x)
        x = x + skips[k - 1]; rec(f"dec{k}", x)
    x = _res3d(sd, p + ".back_layers.0", x); rec("back0", x)
    x = _basic3d(sd, p + ".back_layers.1", x)
    x = _basic3d(sd, p + ".back_layers.2", x); rec("back2", x)
    x = F.conv3d(x, sd[p + ".output_layer.weight"], sd[p + ".output_layer.bias"]); rec("logits", x)
    return x


# ------------------------------------------------------------------------------------------------
# soft-argmax  (utils/op.py:83-96)
# ------------------------------------------------------------------------------------------------
def integrate(volumes, coord, softmax=True, accumulate64=False):
    """utils/op.py:83-96.  ``accumulate64``: evaluate the same formula with float64 softmax / sums from the float32 logits.
    The float32 einsum over 64^3 = 262 144 terms is reduction-order dependent: on the same logits (equal to 5e-6) two x86 hosts
    gave joints 6e-4 m apart (tools/diag/oracle_host_noise.py), so checks that run the oracle on an arbitrary host use the
    float64 evaluation as the platform-stable value of the reference's formula; the goldens keep the reference's own float32."""
    B, J = volumes.shape[:2]
    shp = volumes.shape
    v = volumes.reshape(B, J, -1)
    if accumulate64:
        v = v.double()
    v = F.softmax(v, dim=2) if softmax else F.relu(v)
    v = v.reshape(shp)
    cv = coord.unsqueeze(0).expand(B, -1, -1, -1, -1).to(v.dtype)
    kp = torch.einsum("bnxyz, bxyzc -> bnc", v, cv)
    return (kp.float(), v.float()) if accumulate64 else (kp, v)


# ------------------------------------------------------------------------------------------------
# whole forward  (network/voxel_net_depth.py:224-275)
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def forward(sd, const, images, depth=None, scene_volumes=None, with_scene=True, with_intersection=False,
            volume_multiplier=1.0, volume_softmax=True, taps=None, times=None, accumulate64=False):
    import time
    G = const.G
    tm = (lambda k, t0: times.__setitem__(k, times.get(k, 0.0) + time.perf_counter() - t0)) if times is not None else (lambda k, t0: None)
    t0 = time.perf_counter()
    feats = backbone(sd, images); tm("backbone", t0)
    t0 = time.perf_counter()
    big = process_features(sd, feats); tm("process_features", t0)
    t0 = time.perf_counter()
    vol = unproject(big, const.grid, G); tm("grid_sample", t0)
    if taps is not None:
        taps["features64"] = F.conv2d(feats, sd["process_features.0.weight"], sd["process_features.0.bias"])
        taps["feature_volume"] = vol
    if with_scene:
        t0 = time.perf_counter()
        if scene_volumes is not None:
            occ = scene_volumes.unsqueeze(1)
            vol = torch.cat([vol, occ], dim=1)
        elif depth is not None:
            occ = torch.stack([depth_to_voxel(d.numpy(), const.ray, G, const.side) for d in depth], dim=0).unsqueeze(1)
            if with_intersection:
                vol = torch.cat([vol, vol * occ, occ], dim=1)
            else:
                vol = torch.cat([vol, occ], dim=1)
        else:
            return None
        tm("voxelise", t0)
        if taps is not None:
            taps["occupancy"] = occ[:, 0]
    t0 = time.perf_counter()
    logits = v2v(sd, vol, taps=taps); tm("v2v", t0)
    t0 = time.perf_counter()
    joints, volumes = integrate(logits * volume_multiplier, const.coord, softmax=volume_softmax, accumulate64=accumulate64)
    tm("softargmax", t0)
    return joints, big, volumes
