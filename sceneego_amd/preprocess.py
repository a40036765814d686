"""Input side of the demo path: image / depth loading and pre-processing.

Restates ``DemoDataset.__getitem__`` (reference ``dataset/demo_dataset.py:67-98``) + ``Normalize`` / ``ToTensor``
(``utils/data_transforms.py:38-72``):

  image : ``cv2.imread`` (BGR uint8) -> crop columns 128:-128 -> ``cv2.resize(256, 256)`` -> /255 -> subtract mean,
          divide by std (ImageNet RGB statistics applied to the BGR channels, as the reference does) -> CHW float32
  depth : EXR read -> nearest resize to 1280x1024 if needed -> first channel -> clamp to 10 m -> float32

cv2 is not available in this image, so: JPEG decoding uses PIL (libjpeg; may differ from OpenCV's decoder by +-1
level); the 1024 -> 256 bilinear resize is restated from OpenCV's pixel mapping (src = 4*dst + 1.5: the rounded mean of
the central 2x2 pixels of every 4x4 block); the nearest resize is ``floor(dst * src / dst_size)``.  These three
restatements are "parity unpinned" against real OpenCV.  EXR depth maps (the demo files are
PIZ-compressed HALF) are decoded by ``sceneego_amd/exr.py``; ``.npy`` / ``.npz`` arrays are accepted too.
"""
from __future__ import annotations

import os

import numpy as np
import torch

IMG_MEAN = (0.485, 0.456, 0.406)
IMG_STD = (0.229, 0.224, 0.225)
DEPTH_CLAMP = 10.0


def load_image_bgr(path: str) -> np.ndarray:
    """[H,W,3] uint8, channel order B,G,R (what ``cv2.imread`` returns)."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def resize_quarter_linear_u8(img: np.ndarray) -> np.ndarray:
    """``cv2.resize`` INTER_LINEAR of a uint8 image at an exact 1/4 scale: rounded mean of the central 2x2 of each 4x4 block."""
    h, w = img.shape[:2]
    assert h % 4 == 0 and w % 4 == 0
    a = img.astype(np.uint16)
    s = a[1::4, 1::4] + a[1::4, 2::4] + a[2::4, 1::4] + a[2::4, 2::4]
    return ((s + 2) >> 2).astype(np.uint8)


def preprocess_image(img_bgr: np.ndarray, image_shape=(256, 256)) -> torch.Tensor:
    """Reference ``demo_dataset.py:72-82`` -> [3,256,256] float32 (BGR order, normalised)."""
    raw = img_bgr[:, 128:-128, :]
    if raw.shape[0] == 4 * image_shape[0] and raw.shape[1] == 4 * image_shape[1]:
        small = resize_quarter_linear_u8(raw)
    else:   # generic fallback: area-free bilinear through torch (not the reference's exact arithmetic)
        t = torch.from_numpy(raw.astype(np.float32)).permute(2, 0, 1)[None]
        small = torch.nn.functional.interpolate(t, size=tuple(image_shape), mode="bilinear", align_corners=False)[0]
        small = small.permute(1, 2, 0).round().clamp(0, 255).numpy().astype(np.uint8)
    return normalize_u8(small)


def preprocess_image_device(img_bgr_u8: torch.Tensor, image_shape=(256, 256)) -> torch.Tensor:
    """Device form of :func:`preprocess_image` for full frames: uint8 BGR [H,W,3] or [B,H,W,3] on a HIP device ->
    [B,3,256,256] float32, same arithmetic (``se_preprocess_image_u8``).  Only the exact 1/4 scale is supported."""
    from . import _lib
    x = img_bgr_u8 if img_bgr_u8.dim() == 4 else img_bgr_u8[None]
    B, H, W, _ = x.shape
    if H != 4 * image_shape[0] or W - 256 != 4 * image_shape[1]:
        raise ValueError(f"device pre-processing needs a {4 * image_shape[0]}x{4 * image_shape[1] + 256} frame, got {H}x{W}")
    out = torch.empty((B, 3, image_shape[0], image_shape[1]), device=x.device, dtype=torch.float32)
    return _lib.preprocess_image_u8(x.contiguous(), out, 128, IMG_MEAN, IMG_STD)


def normalize_u8(small_bgr_u8: np.ndarray) -> torch.Tensor:
    """[256,256,3] uint8 (BGR) -> normalised CHW float32 (reference ``demo_dataset.py:76-82``: /255, -mean, /std, ToTensor)."""
    img = small_bgr_u8.astype(np.float64) / 255.0
    img -= IMG_MEAN
    img /= IMG_STD
    return torch.from_numpy(np.transpose(img, (2, 0, 1))).float()


def load_depth(path: str) -> np.ndarray:
    """Depth map in metres as float32 [H,W]: ``.exr`` (first channel, like cv2's ``[:, :, 0]``), ``.npy`` / ``.npz``."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        d = np.load(path)
    elif ext == ".npz":
        with np.load(path) as z:
            d = z[list(z.keys())[0]]
    elif ext == ".exr":
        from .exr import read_depth_exr
        d = read_depth_exr(path)
    else:
        raise ValueError(f"unsupported depth file {path}")
    if d.ndim == 3:
        d = d[:, :, 0]
    return np.asarray(d, dtype=np.float32)


def prepare_depth(depth: np.ndarray, width: int = 1280, height: int = 1024) -> torch.Tensor:
    """Reference ``demo_dataset.py:86-91``: nearest resize to (height, width) if needed, clamp to 10 m."""
    d = depth
    if d.shape[0] != height or d.shape[1] != width:
        ys = np.minimum(np.floor(np.arange(height) * (d.shape[0] / height)).astype(np.int64), d.shape[0] - 1)
        xs = np.minimum(np.floor(np.arange(width) * (d.shape[1] / width)).astype(np.int64), d.shape[1] - 1)
        d = d[ys][:, xs]
    d = np.array(d, dtype=np.float32, copy=True)
    d[d > DEPTH_CLAMP] = DEPTH_CLAMP
    return torch.from_numpy(d)
