"""Portable synthetic weights and inputs (no checkpoint or dataset is reachable: no network).

A counter-based generator (splitmix64 over ``hash(name) + index``) written in plain numpy integer
arithmetic, so the SAME tensors are produced in the build container (where the golden vectors are
captured from the imported reference) and on the GPU box (where only seeds and golden outputs exist).
``torch.randn`` is deliberately not used: its stream is not guaranteed across builds/devices.

Weights are "sharpened" (SURVEY.md §8c): BatchNorm statistics are non-trivial and the output layer is
scaled so the 3D logits have a standard deviation of several units — with PyTorch's default init the
logits have std 0.18, the softmax is flat and every input yields joints ~ (0,0,1), which would let a
broken V2V pass a joint-level parity test.
"""
from __future__ import annotations

import json
import os
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

# Per-BatchNorm-layer scalars (pre-BN activation mean m and std s, measured once by
# tools/calibrate_synth.py with the CPU oracle) and the output-layer gain that gives logits std ~ 6.
# They make the synthetic network behave like a trained one: every BN sees roughly standardised input,
# so activations stay O(1) through all ~100 layers and biases / BN shifts are NOT numerically negligible
# (with un-calibrated statistics activations grow to 1e4 and a kernel that dropped its bias would still
# pass a 1e-3 parity test).  The file is data (about 100 pairs of floats); absent keys mean (0, 1).
_CALIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_calibration.json")


def load_calibration(path: str = _CALIB_PATH):
    if not os.path.isfile(path):
        return {"bn": {}, "output_gain": 1.0}
    with open(path) as f:
        return json.load(f)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream_base(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    return _splitmix64(np.array([(seed << 32) ^ h], dtype=np.uint64))[0]


def uniform01(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """n float64 in [0,1), element i depends only on (seed, name, offset+i)."""
    with np.errstate(over="ignore"):
        ctr = _stream_base(seed, name) + np.arange(offset, offset + n, dtype=np.uint64)
    return (_splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, name, shape, lo, hi) -> np.ndarray:
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(seed, name, n)).reshape(shape).astype(np.float32)


def normal(seed, name, shape, std=1.0) -> np.ndarray:
    """Box-Muller on two counter streams; float64 math, cast to float32."""
    n = int(np.prod(shape))
    u1 = uniform01(seed, name + "/u1", n)
    u2 = uniform01(seed, name + "/u2", n)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return (z * std).reshape(shape).astype(np.float32)


def make_state_dict(reference_state_dict, seed: int = 0, calibration=None):
    """Fill every tensor of a ``VoxelNetwork_depth`` state dict (names/shapes taken from ``reference_state_dict``).

    conv / deconv weights : N(0, sqrt(2 / fan_in)) ; conv biases U(-0.1, 0.1)
    BatchNorm             : gamma U(0.5,1.5) (x0.5 on the last BN of a residual branch), beta U(-0.2,0.2),
                            running_mean m + s*U(-0.2,0.2), running_var s^2*U(0.5,1.5) with the per-layer
                            scalars (m, s) of ``calibration["bn"]``
    volume_net.output_layer.weight additionally x ``calibration["output_gain"]``.
    """
    if calibration is None:
        calibration = load_calibration()
    output_gain = float(calibration.get("output_gain", 1.0))
    bn_cal = calibration.get("bn", {})
    out = {}
    keys = list(reference_state_dict.keys())
    keyset = set(keys)
    for k in keys:
        ref = reference_state_dict[k]
        shape = tuple(ref.shape)
        stem, leaf = k.rsplit(".", 1)
        is_bn = (stem + ".running_mean") in keyset
        if leaf == "num_batches_tracked":
            out[k] = torch.zeros(shape, dtype=ref.dtype)
            continue
        if is_bn:
            if leaf == "weight":
                v = uniform(seed, k, shape, 0.5, 1.5)
                last_of_branch = stem.endswith("bn3") or stem.endswith("res_branch.4")
                if last_of_branch:
                    v = v * np.float32(0.5)
            elif leaf == "bias":
                v = uniform(seed, k, shape, -0.2, 0.2)
            elif leaf == "running_mean":
                m, sdev = bn_cal.get(stem, (0.0, 1.0))
                v = (uniform(seed, k, shape, -0.2, 0.2) * np.float32(sdev) + np.float32(m)).astype(np.float32)
            else:  # running_var
                m, sdev = bn_cal.get(stem, (0.0, 1.0))
                v = (uniform(seed, k, shape, 0.5, 1.5) * np.float32(sdev) * np.float32(sdev)).astype(np.float32)
        elif leaf == "weight":
            if "deconv_layers" in k or "decoder_upsample" in k:   # transposed conv: [Cin, Cout, k...]
                fan_in = shape[0] * int(np.prod(shape[2:])) / (2 ** (len(shape) - 2))  # each output sees k/stride taps
            else:
                fan_in = shape[1] * int(np.prod(shape[2:]))
            v = normal(seed, k, shape, std=float(np.sqrt(2.0 / fan_in)))
            if k == "volume_net.output_layer.weight":
                v = v * np.float32(output_gain)
        else:  # conv bias
            v = uniform(seed, k, shape, -0.1, 0.1)
        out[k] = torch.from_numpy(np.ascontiguousarray(v))
    return out


def make_inputs(seed: int, batch: int, depth_kind: str = "uniform", image_hw=(256, 256), depth_hw=(1024, 1280)):
    """Seeded (image, depth) pair(s): image ~ N(0,1) float32 [B,3,256,256]; depth float32 [B,1024,1280] metres.

    depth_kind 'uniform': iid U(0.3, 3.0) per pixel (worst-case scatter locality, BASELINE config 2);
               'floor'  : a smooth scene: floor plane 1.4 m below the camera seen through a pinhole-ish
                          mapping, clamped to 10 m, plus seeded per-sample tilt (realistic sparsity).
    """
    imgs = normal(seed, "input/image", (batch, 3) + tuple(image_hw), 1.0)
    H, W = depth_hw
    if depth_kind == "uniform":
        depth = uniform(seed, "input/depth", (batch, H, W), 0.3, 3.0)
    elif depth_kind == "floor":
        ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
        r = np.sqrt((xs - W / 2) ** 2 + (ys - H / 2) ** 2) / (H / 2)
        tilt = uniform(seed, "input/tilt", (batch, 2), -0.15, 0.15).astype(np.float64)
        depth = np.empty((batch, H, W), dtype=np.float32)
        for b in range(batch):
            cosang = np.cos(np.minimum(r, 1.45) * (np.pi / 3)) + tilt[b, 0] * (xs - W / 2) / W + tilt[b, 1] * (ys - H / 2) / H
            d = 1.4 / np.maximum(cosang, 0.14)
            depth[b] = np.minimum(d, 10.0).astype(np.float32)
    else:
        raise ValueError(depth_kind)
    return torch.from_numpy(imgs), torch.from_numpy(depth)
