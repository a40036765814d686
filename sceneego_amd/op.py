"""Host-side geometry of the hot path + the operator surface of the reference's ``utils/op.py``.

Init-time constants (run once on the host, exactly as the reference does):
  * ``build_coord_volume``                      <- ``network/voxel_net_depth.py:110-134``
  * ``get_projected_2d_points_with_coord_volumes`` <- ``utils/op.py:98-116`` (-> ``utils/multiview.py:114-132``)
  * ``get_grid_coord_proj_batch``               <- ``utils/op.py:177-184``
  * ``calculated_ray_direction_numpy``          <- ``network/voxel_net_depth.py:147-155``
plus two tables that exist only in this build (they are what the HIP kernels read):
  * ``build_gather_table``  — per voxel 4 texel indices + 4 bilinear weights into the 64x64 map
    (fuses Upsample(1024^2, nearest) + ConstantPad2d(128) + grid_sample);
  * ``build_voxelizer_ray_table`` — the float64 rays of the centre 1024 columns, in the order the
    voxeliser kernel walks the depth map.

Per-forward operators (device; thin wrappers over the C-ABI in ``include/sceneego_hip.h``):
  * ``unproject_heatmaps_one_view_batch``       <- ``utils/op.py:194-214``
  * ``integrate_tensor_3d_with_coordinates``    <- ``utils/op.py:83-96``
These two keep the reference's names and argument meaning; they run on HIP tensors only and raise
if the extension is missing (there is no CPU fallback in the product path).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

# Geometry the reference hard-codes (voxel_net_depth.py:60-61,197-198): the 64x64 feature map is
# nearest-upsampled to 1024x1024 and zero-padded by 128 columns left/right to the 1280-wide image.
UPSAMPLED = 1024
PAD_X = 128


# ----------------------------------------------------------------------------------------------
# init-time constants
# ----------------------------------------------------------------------------------------------
def build_coord_volume(volume_size: int, cuboid_side: float) -> torch.Tensor:
    """[G,G,G,3] float32 voxel-centre coordinates in metres (reference ``voxel_net_depth.py:110-134``)."""
    sides = np.array([cuboid_side, cuboid_side, cuboid_side])
    position = np.array([-cuboid_side / 2, -cuboid_side / 2, 0])
    r = torch.arange(volume_size)
    xxx, yyy, zzz = torch.meshgrid(r, r, r, indexing="ij")
    grid = torch.stack([xxx, yyy, zzz], dim=-1).type(torch.float).reshape((-1, 3))
    coord = torch.zeros_like(grid)
    for a in range(3):
        coord[:, a] = position[a] + (sides[a] / (volume_size - 1)) * grid[:, a]
    return coord.reshape(volume_size, volume_size, volume_size, 3)


def get_projected_2d_points_with_coord_volumes(fisheye_model, coord_volume: torch.Tensor) -> torch.Tensor:
    """[N,2] pixel position of every voxel centre (reference ``utils/op.py:98-116``)."""
    return fisheye_model.world2camera_pytorch(coord_volume.reshape((-1, 3)))


def get_grid_coord_proj_batch(grid_coord_proj: torch.Tensor, batch_size: int, heatmap_shape) -> torch.Tensor:
    """Normalise to grid_sample's [-1,1] and expand (stride 0) to the batch (reference ``utils/op.py:177-184``)."""
    g = torch.zeros_like(grid_coord_proj)
    g[:, 0] = 2 * (grid_coord_proj[:, 0] / heatmap_shape[1] - 0.5)
    g[:, 1] = 2 * (grid_coord_proj[:, 1] / heatmap_shape[0] - 0.5)
    g = g.unsqueeze(1).unsqueeze(0)
    return g.expand(batch_size, -1, -1, -1)


def calculated_ray_direction_numpy(fisheye_model, image_width: int, image_height: int) -> np.ndarray:
    """[W*H,3] float64 unit rays, pixel order x-major (flat = x*H + y) (reference ``voxel_net_depth.py:147-155``)."""
    xs = np.arange(image_width, dtype=np.float64)
    ys = np.arange(image_height, dtype=np.float64)
    points = np.zeros((image_width, image_height, 2))
    points[:, :, 0] = xs[:, None]
    points[:, :, 1] = ys[None, :]
    return fisheye_model.camera2world_ray(points.reshape((-1, 2)))


def _nearest_src(dst: torch.Tensor, in_size: int, out_size: int) -> torch.Tensor:
    """Source index of ``nn.Upsample(mode='nearest')`` (ATen ``nearest_neighbor_compute_source_index``): min(floor(dst * scale), in - 1) with
    scale = in / out evaluated in float32 (exact for the power-of-two ratios of the 256x256 crop, and the reference's rounding elsewhere)."""
    scale = torch.tensor(float(in_size), dtype=torch.float32) / torch.tensor(float(out_size), dtype=torch.float32)
    src = torch.floor(dst.to(torch.float32) * scale).to(torch.int64)
    return torch.clamp(src, max=in_size - 1)


def build_gather_table(grid_coord_proj_norm: torch.Tensor, heatmap_shape, feat_hw=64):
    """Per-voxel 4-tap lookup into the compact feature map.

    ``grid_coord_proj_norm`` is [N,2] in grid_sample's normalised coordinates for an image of
    ``heatmap_shape`` = (H=1024, W=1280).  grid_sample(align_corners=True, bilinear, zeros) un-normalises
    ix = (gx+1)/2*(W-1), iy = (gy+1)/2*(H-1) and blends the 4 neighbouring texels; a texel (x,y) of the
    virtual 1024x1280 image is ``F[src(y), src(x-128)]`` inside the 1024 centre columns and 0 elsewhere, src = the nearest-neighbour
    source index of ``nn.Upsample(size=(1024, 1024))`` for a feature map of ``feat_hw`` = (h, w) (an int means square; the reference
    upsamples ANY feature-map size, ``network/voxel_net_depth.py:59-60,238``; 64 x 64 for the 256 x 256 crop: src = dst >> 4).
    Returns (idx int32 [N,4], w float32 [N,4]); idx = fy * w + fx, -1 marks a zero tap.
    """
    H, W = int(heatmap_shape[0]), int(heatmap_shape[1])
    fh, fw = (int(feat_hw), int(feat_hw)) if isinstance(feat_hw, int) else (int(feat_hw[0]), int(feat_hw[1]))
    if fh <= 0 or fw <= 0 or H != UPSAMPLED or W != UPSAMPLED + 2 * PAD_X:
        raise ValueError("gather table: heatmap %dx%d / feature map %dx%d not representable" % (H, W, fh, fw))
    g = grid_coord_proj_norm.detach().to(torch.float32).cpu()
    # same float32 arithmetic as ATen's grid_sampler_unnormalize(align_corners=True): ((g + 1) / 2) * (size - 1)
    ix = ((g[:, 0] + 1) / 2) * (W - 1)
    iy = ((g[:, 1] + 1) / 2) * (H - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    # ATen weights: nw=(x1-ix)(y1-iy)  ne=(ix-x0)(y1-iy)  sw=(x1-ix)(iy-y0)  se=(ix-x0)(iy-y0)
    w = torch.stack([(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)], dim=1)
    taps_x = torch.stack([x0, x1, x0, x1], dim=1).to(torch.int64)
    taps_y = torch.stack([y0, y0, y1, y1], dim=1).to(torch.int64)
    inside = (taps_x >= PAD_X) & (taps_x < PAD_X + UPSAMPLED) & (taps_y >= 0) & (taps_y < H)
    fx = _nearest_src(torch.clamp(taps_x - PAD_X, 0, UPSAMPLED - 1), fw, UPSAMPLED)
    fy = _nearest_src(torch.clamp(taps_y, 0, UPSAMPLED - 1), fh, UPSAMPLED)
    idx = torch.where(inside, fy * fw + fx, torch.full_like(fx, -1)).to(torch.int32)
    w = torch.where(inside, w, torch.zeros_like(w)).to(torch.float32)
    return idx.contiguous(), w.contiguous()


def build_voxelizer_ray_table(ray: np.ndarray, image_width: int, image_height: int) -> np.ndarray:
    """Rays of the centre ``UPSAMPLED`` columns, laid out [y, x', 3] float64 (row-major like the depth map).

    The reference multiplies ``ray`` (x-major, [W*H,3]) with the padded depth transposed
    (``voxel_net_depth.py:198-200``); the 2x128 pad columns carry depth 0 and need no ray.
    """
    assert image_width == UPSAMPLED + 2 * PAD_X
    r = ray.reshape(image_width, image_height, 3)[PAD_X:PAD_X + UPSAMPLED]  # [x', y, 3]
    return np.ascontiguousarray(r.transpose(1, 0, 2))                        # [y, x', 3]


# ----------------------------------------------------------------------------------------------
# per-forward operators (HIP)
# ----------------------------------------------------------------------------------------------
def unproject_heatmaps_one_view_batch(heatmaps, grid_coord_proj_transformed_batch, volume_size):
    """Reference ``utils/op.py:194-214``: sample ``heatmaps`` [B,C,H,W] at the projected voxel centres.

    Generic form (any H,W, bilinear, zeros padding, align_corners=True) on the HIP gather kernel;
    returns [B,C,G,G,G] like the reference.  The network's forward uses the fused table-driven form
    instead (no 1024x1280 intermediate), see ``VoxelNetwork_depth.forward``.
    """
    _lib.require_hip(heatmaps)
    B, C, H, W = heatmaps.shape
    G = int(volume_size)
    grid = grid_coord_proj_transformed_batch[0].reshape(-1, 2)   # identical for every sample (stride-0 expand)
    idx, w = build_gather_table_generic(grid, H, W)
    idx = idx.to(heatmaps.device)
    w = w.to(heatmaps.device)
    feat = heatmaps.permute(0, 2, 3, 1).contiguous().float()     # NHWC
    out = torch.empty((B, G * G * G, C), device=heatmaps.device, dtype=torch.float32)
    _lib.unproject_gather(feat, idx, w, out, B, H * W, C, G * G * G, C, 0)
    return out.view(B, G, G, G, C).permute(0, 4, 1, 2, 3)


def build_gather_table_generic(grid_norm: torch.Tensor, H: int, W: int):
    """4-tap table for a plain [H,W] image (no upsample/pad folding)."""
    g = grid_norm.detach().to(torch.float32).cpu()
    ix = ((g[:, 0] + 1) / 2) * (W - 1)
    iy = ((g[:, 1] + 1) / 2) * (H - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    w = torch.stack([(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)], dim=1)
    tx = torch.stack([x0, x1, x0, x1], dim=1).to(torch.int64)
    ty = torch.stack([y0, y0, y1, y1], dim=1).to(torch.int64)
    inside = (tx >= 0) & (tx < W) & (ty >= 0) & (ty < H)
    idx = torch.where(inside, ty * W + tx, torch.full_like(tx, -1)).to(torch.int32)
    w = torch.where(inside, w, torch.zeros_like(w)).to(torch.float32)
    return idx.contiguous(), w.contiguous()


def integrate_tensor_3d_with_coordinates(volumes, coord_volumes, softmax=True):
    """Reference ``utils/op.py:83-96``: softmax over all voxels per (b,joint), then expectation of the coordinates.

    ``volumes`` [B,J,X,Y,Z] float32 (NCDHW, contiguous), ``coord_volumes`` [B,X,Y,Z,3] (any batch stride;
    sample 0 is used for all — the reference's grid is a stride-0 expand).  Returns
    (coordinates [B,J,3], volumes [B,J,X,Y,Z]) with the softmax (or ReLU, ``softmax=False``) applied.
    """
    _lib.require_hip(volumes)
    B, J, X, Y, Z = volumes.shape
    vol = volumes.contiguous().float()
    coord = coord_volumes[0].contiguous().float()
    out_vol = torch.empty_like(vol)
    joints = torch.empty((B, J, 3), device=vol.device, dtype=torch.float32)
    _lib.softargmax3d(vol, coord, out_vol, joints, B * J, X * Y * Z, 1 if softmax else 0)
    return joints, out_vol
