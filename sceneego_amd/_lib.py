"""ctypes binding of ``libsceneego_hip.so`` (C ABI declared in ``include/sceneego_hip.h``).

PyTorch is only the allocator and stream provider here: every call passes ``tensor.data_ptr()`` and the
current HIP stream.  There is deliberately NO fallback: if the shared library is missing, or a tensor
is not on a HIP device, the call raises.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# tools/ may point at the development build (csrc/build.sh --devtools -> libsceneego_hip_dev.so)
LIB_PATH = os.environ.get("SCENEEGO_HIP_LIB") or os.path.join(_HERE, "libsceneego_hip.so")
ABI_VERSION = 22

EPI_RELU = 1
EPI_RES_PRE_RELU = 2
EPI_RES_POST_RELU = 4
EPI_OUT_PLANAR = 8
IN_OCTET = 32       # se_conv3d_f32, 2-D Winograd 3x3x3 shapes: octet-planar input [B][C/8][D][D][D][8]
OUT_OCTET = 64      # ... octet-planar output
RES_OCTET = 128     # ... octet-planar skip tensor
IN_QUAD = 1024      # se_conv3d_f32, launches whose variant (with these flags) is 3: quad-planar input [B][C/4][D][D][D][4]
OUT_QUAD = 2048     # ... quad-planar output
RES_QUAD = 4096     # ... quad-planar skip tensor
IN_PLANAR3 = 16     # se_conv3d_f32, k = 7: triplet-planar input [B][ceil(cin/3)][D][D][D][3]

_vp, _i, _f, _d, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_longlong

# name -> (restype, argtypes); must list every symbol of include/sceneego_hip.h (tests/test_abi.py checks)
SIGNATURES = {
    "se_abi_version": (_i, []),
    "se_voxelize_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _d, _vp]),
    "se_voxelize_strided_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _d, _i, _i, _vp]),
    "se_voxelize_full_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _d, _vp]),
    "se_unproject_gather_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "se_unproject_gather_planar3_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_voxelize_planar3_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _d, _i, _i, _vp]),
    "se_unproject_gather_planar1_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_voxelize_planar1_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _d, _i, _i, _vp]),
    "se_intersection_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "se_bias_act_nchw_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "se_deconv2d_k4s2_assemble_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_bias_relu_maxpool3x3s2_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "se_conv2d_1x1_tile_f32": (_i, [_i, _i, _i, _i]),
    "se_conv2d_1x1_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_conv2d_1x1_small_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_conv2d_1x1_small_s2_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "se_conv2d_1x1_s2_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "se_conv2d_3x3_tile_f32": (_i, [_i, _i, _i, _i, _i]),
    "se_conv2d_3x3_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "se_conv2d_3x3_s2_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "se_conv3d_pack_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_conv3d_packed_elems": (_ll, [_i, _i, _i, _i]),
    "se_conv3d_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp]),
    "se_conv3d_pool_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp]),
    "se_conv3d_skip16_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_pointwise_chain3_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "se_deconv3d_k2s2_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_maxpool3d_2_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "se_maxpool3d_2_octin_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "se_softargmax3d_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "se_softargmax3d_scratch_elems": (_ll, [_i]),
    "se_softargmax3d_finish_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "se_pointwise_chain3_softargmax_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "se_conv3d_pack_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_conv3d_packed_elems_bf16": (_ll, [_i, _i, _i, _i]),
    "se_conv3d_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "se_pointwise_chain3_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "se_pointwise_chain3_softargmax_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "se_deconv3d_k2s2_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "se_maxpool3d_2_bf16": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "se_unproject_gather_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "se_voxelize_strided_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _d, _i, _i, _vp]),
    "se_preprocess_image_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "se_bias_act_nchw_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "se_conv3d_f32_algo": (_i, [_i, _i, _i, _i]),
    "se_conv3d_f32_variant": (_i, [_i, _i, _i, _i, _i, _i]),
    "se_conv3d_k7_fft_packed_elems": (_ll, [_i, _i]),
    "se_conv3d_k7_fft_pack_f32": (_i, [_vp, _vp, _vp, _f, _vp, _i, _i, _vp]),
    "se_conv3d_k7_fft_workspace_elems": (_ll, [_i, _i, _i]),
    "se_conv3d_k7_fft_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _ll, _vp]),
    "se_conv3d_split3_packed_elems": (_ll, [_i, _i]),
    "se_conv3d_split3_pack": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "se_conv3d_k3_split3_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
}
# present only in development builds (csrc/build.sh --devtools): A/B kernel selection and cycle-stamp diagnostics (tools/)
DEVTOOLS_SIGNATURES = {
    "se_debug_set_variant": (None, [_i]),
    "se_debug_set_stamp_buffer": (None, [_vp]),
}

_lib = None


class HipExtensionError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the shared library once; raise loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or sceneego_amd/csrc/build.sh). The SceneEgo hot path has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DEVTOOLS_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    got = lib.se_abi_version()
    if got != ABI_VERSION:
        raise HipExtensionError(f"libsceneego_hip.so ABI {got} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def source_fingerprint() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, build.sh and the C-ABI header), names included, in sorted order.
    rocprofv3 counter records under profiles/ carry it, so a record taken on other kernels is recognised as stale (bench.py)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".sh")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "sceneego_hip.h"))
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_fingerprint() -> str | None:
    """source_fingerprint() of the sources the loaded shared library was built from (written by csrc/build.sh next to the .so),
    or None when the stamp is missing."""
    try:
        with open(os.path.splitext(LIB_PATH)[0] + ".srchash") as f:
            return f.read().strip() or None
    except OSError:
        return None


def require_hip(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipExtensionError("SceneEgo HIP operators need tensors on a HIP device (got %s)" % t.device)
    load()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(code: int, what: str) -> None:
    if code != 0:
        raise HipExtensionError(f"{what} failed with code {code}" + (" (bad argument)" if code == -1 else " (hipError_t)"))


def _chk_f32(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())


# ---------------------------------------------------------------------------------------------
def voxelize(depth, ray_tab, occ, batch, depth_h, depth_w, up, pad_x, volume_size, cuboid_side):
    require_hip(depth, ray_tab, occ)
    _chk_f32(depth, occ)
    assert ray_tab.dtype == torch.float64 and ray_tab.is_contiguous()
    _check(load().se_voxelize_f64(_ptr(depth), _ptr(ray_tab), _ptr(occ), batch, depth_h, depth_w, up, pad_x,
                                  volume_size, float(cuboid_side), _stream()), "se_voxelize_f64")


def voxelize_strided(depth, ray_tab, buf, batch, depth_h, depth_w, up, pad_x, volume_size, cuboid_side, stride_c, c_offset):
    require_hip(depth, ray_tab, buf)
    _chk_f32(depth)
    assert ray_tab.dtype == torch.float64 and ray_tab.is_contiguous() and buf.is_contiguous()
    assert buf.dtype in (torch.bfloat16, torch.float32)
    if buf.dtype == torch.bfloat16:     # octet-planar [B][stride_c / 8][N][8]
        _check(load().se_voxelize_strided_bf16(_ptr(depth), _ptr(ray_tab), _ptr(buf), batch, depth_h, depth_w, up, pad_x,
                                               volume_size, float(cuboid_side), stride_c // 8, c_offset, _stream()),
               "se_voxelize_strided_bf16")
        return
    _check(load().se_voxelize_strided_f64(_ptr(depth), _ptr(ray_tab), _ptr(buf), batch, depth_h, depth_w, up, pad_x,
                                          volume_size, float(cuboid_side), stride_c, c_offset, _stream()),
           "se_voxelize_strided_f64")


def voxelize_planar3(depth, ray_tab, buf, batch, depth_h, depth_w, up, pad_x, volume_size, cuboid_side, triplets_total, channel):
    """Scatter the occupancy into slot `channel` of a triplet-planar float32 buffer (cleared by unproject_gather_planar3)."""
    require_hip(depth, ray_tab, buf)
    _chk_f32(depth, buf)
    assert ray_tab.dtype == torch.float64 and ray_tab.is_contiguous()
    assert buf.numel() == batch * triplets_total * volume_size ** 3 * 3
    _check(load().se_voxelize_planar3_f64(_ptr(depth), _ptr(ray_tab), _ptr(buf), batch, depth_h, depth_w, up, pad_x,
                                          volume_size, float(cuboid_side), triplets_total, channel, _stream()),
           "se_voxelize_planar3_f64")


def voxelize_planar1(depth, ray_tab, buf, batch, depth_h, depth_w, up, pad_x, volume_size, cuboid_side, planes_total, channel):
    """Scatter the occupancy into plane `channel` of a planar float32 buffer [B, planes_total, G^3] (cleared by unproject_gather_planar1)."""
    require_hip(depth, ray_tab, buf)
    _chk_f32(depth, buf)
    assert ray_tab.dtype == torch.float64 and ray_tab.is_contiguous()
    assert buf.numel() == batch * planes_total * volume_size ** 3
    _check(load().se_voxelize_planar1_f64(_ptr(depth), _ptr(ray_tab), _ptr(buf), batch, depth_h, depth_w, up, pad_x,
                                          volume_size, float(cuboid_side), planes_total, channel, _stream()),
           "se_voxelize_planar1_f64")


def voxelize_full(depth, ray_tab, occ, batch, depth_h, depth_w, volume_size, cuboid_side):
    require_hip(depth, ray_tab, occ)
    _chk_f32(depth, occ)
    assert ray_tab.dtype == torch.float64 and ray_tab.is_contiguous()
    _check(load().se_voxelize_full_f64(_ptr(depth), _ptr(ray_tab), _ptr(occ), batch, depth_h, depth_w,
                                       volume_size, float(cuboid_side), _stream()), "se_voxelize_full_f64")


def preprocess_image_u8(img_u8, out, crop_x, mean3, std3):
    """img_u8 [B,H,W,3] uint8 BGR on the device -> out [B,3,H/4,(W-2*crop_x)/4] float32 (normalised, reference arithmetic)."""
    require_hip(img_u8, out)
    assert img_u8.dtype == torch.uint8 and img_u8.is_contiguous() and img_u8.shape[-1] == 3
    _chk_f32(out)
    B, H, W, _ = img_u8.shape
    m = (ctypes.c_double * 3)(*[float(x) for x in mean3])
    sd = (ctypes.c_double * 3)(*[float(x) for x in std3])
    _check(load().se_preprocess_image_u8(_ptr(img_u8), _ptr(out), B, H, W, crop_x, m, sd, _stream()), "se_preprocess_image_u8")
    return out


def unproject_gather(feat, idx, w, out, batch, texels, channels, voxels, out_stride_c, out_c_offset):
    require_hip(feat, idx, w, out)
    _chk_f32(feat, w)
    assert idx.dtype == torch.int32 and idx.is_contiguous() and out.is_contiguous()
    assert out.dtype in (torch.bfloat16, torch.float32)
    if out.dtype == torch.bfloat16:     # octet-planar [B][out_stride_c / 8][voxels][8]
        _check(load().se_unproject_gather_bf16(_ptr(feat), _ptr(idx), _ptr(w), _ptr(out), batch, texels, channels,
                                               voxels, out_stride_c // 8, out_c_offset, _stream()), "se_unproject_gather_bf16")
        return
    _check(load().se_unproject_gather_f32(_ptr(feat), _ptr(idx), _ptr(w), _ptr(out), batch, texels, channels,
                                          voxels, out_stride_c, out_c_offset, _stream()), "se_unproject_gather_f32")


def unproject_gather_planar3(feat, idx, w, out, batch, texels, channels, voxels, triplets_total):
    require_hip(feat, idx, w, out)
    _chk_f32(feat, w, out)
    assert idx.dtype == torch.int32 and idx.is_contiguous()
    assert out.numel() == batch * triplets_total * voxels * 3
    _check(load().se_unproject_gather_planar3_f32(_ptr(feat), _ptr(idx), _ptr(w), _ptr(out), batch, texels, channels,
                                                  voxels, triplets_total, _stream()), "se_unproject_gather_planar3_f32")


def unproject_gather_planar1(feat, idx, w, out, batch, texels, channels, voxels, planes_total):
    require_hip(feat, idx, w, out)
    _chk_f32(feat, w, out)
    assert idx.dtype == torch.int32 and idx.is_contiguous()
    assert out.numel() == batch * planes_total * voxels
    _check(load().se_unproject_gather_planar1_f32(_ptr(feat), _ptr(idx), _ptr(w), _ptr(out), batch, texels, channels,
                                                  voxels, planes_total, _stream()), "se_unproject_gather_planar1_f32")


def intersection(buf, occ, batch, voxels, channels, stride_c):
    require_hip(buf, occ)
    _chk_f32(buf, occ)
    _check(load().se_intersection_f32(_ptr(buf), _ptr(occ), batch, voxels, channels, stride_c, _stream()),
           "se_intersection_f32")


def bias_act_nchw(x, bias, residual, relu):
    """In place on ``x`` [N,C,H,W] (contiguous): x = relu?(x + bias[c] (+ residual))."""
    require_hip(x, bias)
    n, c, hh, ww = x.shape
    if x.dtype == torch.bfloat16:
        assert bias.dtype == torch.bfloat16 and x.is_contiguous() and (residual is None or (residual.dtype == x.dtype and residual.is_contiguous()))
        _check(load().se_bias_act_nchw_bf16(_ptr(x), _ptr(bias), _ptr(residual), _ptr(x), n, c, hh * ww, 1 if relu else 0,
                                            _stream()), "se_bias_act_nchw_bf16")
        return x
    _chk_f32(x, bias, residual)
    _check(load().se_bias_act_nchw_f32(_ptr(x), _ptr(bias), _ptr(residual), _ptr(x), n, c, hh * ww, 1 if relu else 0,
                                       _stream()), "se_bias_act_nchw_f32")
    return x


def bias_relu_maxpool(x, bias):
    """``x`` [B, C, 2 ho, 2 wo] float32 (raw stem convolution) -> max_pool2d(relu(x + bias), 3, stride 2, padding 1) [B, C, ho, wo] in one
    pass (se_bias_relu_maxpool3x3s2_f32); wo % 4 == 0."""
    require_hip(x, bias)
    _chk_f32(x, bias)
    B, C, H, W = x.shape
    assert H % 2 == 0 and W % 8 == 0 and bias.numel() == C
    out = torch.empty((B, C, H // 2, W // 2), device=x.device, dtype=torch.float32)
    _check(load().se_bias_relu_maxpool3x3s2_f32(_ptr(x), _ptr(bias), _ptr(out), B, C, H // 2, W // 2, _stream()), "se_bias_relu_maxpool3x3s2_f32")
    return out


_TILE_MEMO = {}


def _tile_memo(fn_name, *args) -> int:
    """The library's tile answers depend on the arguments and the current device's CU count only: one ctypes call per distinct question."""
    key = (fn_name, torch.cuda.current_device() if torch.cuda.is_available() else -1) + args
    t = _TILE_MEMO.get(key)
    if t is None:
        t = _TILE_MEMO[key] = int(getattr(load(), fn_name)(*args))
    return t


def conv2d_1x1_tile(batch, cin, cout, hw) -> int:
    """Channel-tile width the packed weights of a 1x1 convolution need (128 / 64), 0 when se_conv2d_1x1_f32 does not cover the shape."""
    return _tile_memo("se_conv2d_1x1_tile_f32", int(batch), int(cin), int(cout), int(hw))


def conv2d_1x1_pack(w2d, tile):
    """Folded [cout, cin] matrix -> [cout / tile, cin / 16, tile, 16] (the layout se_conv2d_1x1_f32 streams per (channel tile, k step))."""
    cout, cin = w2d.shape
    return w2d.reshape(cout // tile, tile, cin // 16, 16).permute(0, 2, 1, 3).contiguous()


def conv2d_1x1(x, wpack, bias, residual, relu, in_bias=None):
    """``x`` [B, cin, H, W] NCHW float32 -> relu?(W x' + bias (+ residual)) [B, cout, H, W] in one launch (se_conv2d_1x1_f32);
    ``in_bias`` [cin]: x' = relu(x + in_bias) (the producing convolution's bias + ReLU applied on the way in), else x' = x."""
    require_hip(x, wpack, bias)
    _chk_f32(x, wpack, bias, residual, in_bias)
    B, cin, H, W = x.shape
    cout = wpack.shape[0] * wpack.shape[2]
    assert wpack.dim() == 4 and wpack.shape[1] * 16 == cin and wpack.shape[2] == conv2d_1x1_tile(B, cin, cout, H * W)
    assert in_bias is None or in_bias.numel() == cin
    out = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    _check(load().se_conv2d_1x1_f32(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(residual), _ptr(in_bias), _ptr(out), B, cin, cout, H * W,
                                    1 if relu else 0, _stream()), "se_conv2d_1x1_f32")
    return out


def conv2d_1x1_small_ok(batch, cin, cout, hw) -> bool:
    """Shapes se_conv2d_1x1_small_f32 covers."""
    return cin % 64 == 0 and cout % 16 == 0 and hw % 16 == 0 and (batch * hw) % 64 == 0 and batch > 0


def conv2d_1x1_small(x, wpack16, bias, residual, relu, in_bias=None):
    """conv2d_1x1's result from the small-M kernel (se_conv2d_1x1_small_f32: 64 pixels x 16 channels per workgroup, k split over four / two
    wave groups); ``wpack16`` = ``conv2d_1x1_pack(w2d, 16)``."""
    require_hip(x, wpack16, bias)
    _chk_f32(x, wpack16, bias, residual, in_bias)
    B, cin, H, W = x.shape
    cout = wpack16.shape[0] * 16
    assert wpack16.dim() == 4 and wpack16.shape[1] * 16 == cin and wpack16.shape[2] == 16 and conv2d_1x1_small_ok(B, cin, cout, H * W)
    assert in_bias is None or in_bias.numel() == cin
    out = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    _check(load().se_conv2d_1x1_small_f32(_ptr(x), _ptr(wpack16), _ptr(bias), _ptr(residual), _ptr(in_bias), _ptr(out), B, cin, cout, H * W,
                                          1 if relu else 0, _stream()), "se_conv2d_1x1_small_f32")
    return out


def conv2d_1x1_s2(x, wpack, bias, relu=False):
    """``x`` [B, cin, 2 ho, 2 wo] -> relu?(W x[:, :, ::2, ::2] + bias) [B, cout, ho, wo]: the stride-2 1x1 convolution (se_conv2d_1x1_s2_f32;
    ``wpack`` packed for ``conv2d_1x1_tile(B, cin, cout, ho * wo)``) - or, with a 16-channel packing, its small-M form
    (se_conv2d_1x1_small_s2_f32)."""
    require_hip(x, wpack, bias)
    _chk_f32(x, wpack, bias)
    B, cin, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    ho, wo = H // 2, W // 2
    cout = wpack.shape[0] * wpack.shape[2]
    out = torch.empty((B, cout, ho, wo), device=x.device, dtype=torch.float32)
    assert wpack.dim() == 4 and wpack.shape[1] * 16 == cin
    if wpack.shape[2] == 16:
        assert conv2d_1x1_small_ok(B, cin, cout, ho * wo)
        _check(load().se_conv2d_1x1_small_s2_f32(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(out), B, cin, cout, ho, wo, 1 if relu else 0, _stream()),
               "se_conv2d_1x1_small_s2_f32")
        return out
    assert wpack.shape[2] == conv2d_1x1_tile(B, cin, cout, ho * wo)
    _check(load().se_conv2d_1x1_s2_f32(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(out), B, cin, cout, ho, wo, 1 if relu else 0, _stream()),
           "se_conv2d_1x1_s2_f32")
    return out


def conv2d_3x3_tile(batch, cin, cout, h, w) -> int:
    """Channel-tile width the packed weights of a 3x3 stride-1 convolution need (32 / 16), 0 when se_conv2d_3x3_f32 does not cover the shape."""
    return _tile_memo("se_conv2d_3x3_tile_f32", int(batch), int(cin), int(cout), int(h), int(w))


def conv2d_3x3_pack(w4d, tile):
    """Folded [cout, cin, 3, 3] tensor -> [cout / tile, cin / 16, 9, tile, 16] (what se_conv2d_3x3_f32 streams per (channel tile, k step))."""
    cout, cin = w4d.shape[:2]
    return w4d.reshape(cout // tile, tile, cin // 16, 16, 9).permute(0, 2, 4, 1, 3).contiguous()


def conv2d_3x3(x, wpack, bias, relu):
    """``x`` [B, cin, H, W] NCHW float32 -> relu?(conv3x3(x) (+ bias)) [B, cout, H, W], stride 1, padding 1 (se_conv2d_3x3_f32);
    ``bias`` None: the raw sums (the consumer applies the bias - conv2d_1x1's ``in_bias``)."""
    require_hip(x, wpack)
    _chk_f32(x, wpack, bias)
    B, cin, H, W = x.shape
    cout = wpack.shape[0] * wpack.shape[3]
    assert wpack.dim() == 5 and wpack.shape[1] * 16 == cin and wpack.shape[2] == 9 and wpack.shape[3] == conv2d_3x3_tile(B, cin, cout, H, W)
    out = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    _check(load().se_conv2d_3x3_f32(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(out), B, cin, cout, H, W, 1 if relu else 0, _stream()),
           "se_conv2d_3x3_f32")
    return out


def conv2d_3x3_s2_ok(cin, cout, ho, wo) -> bool:
    """Shapes se_conv2d_3x3_s2_f32 covers (output map ho x wo)."""
    tile_ok = (ho % 4 == 0) if wo % 16 == 0 else (wo % 8 == 0 and ho % 8 == 0)
    return cin % 32 == 0 and cout % 16 == 0 and ho > 0 and wo > 0 and tile_ok and cin * ho * wo * 4 < (1 << 31) // 4


def conv2d_3x3_s2(x, wpack, bias, relu):
    """``x`` [B, cin, 2 ho, 2 wo] -> relu?(conv3x3 stride 2 padding 1 (x) (+ bias)) [B, cout, ho, wo] (se_conv2d_3x3_s2_f32); ``wpack`` =
    ``conv2d_3x3_pack(w, 16)``."""
    require_hip(x, wpack)
    _chk_f32(x, wpack, bias)
    B, cin, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0
    cout = wpack.shape[0] * wpack.shape[3]
    assert wpack.dim() == 5 and wpack.shape[1] * 16 == cin and wpack.shape[2] == 9 and wpack.shape[3] == 16
    out = torch.empty((B, cout, H // 2, W // 2), device=x.device, dtype=torch.float32)
    _check(load().se_conv2d_3x3_s2_f32(_ptr(x), _ptr(wpack), _ptr(bias), _ptr(out), B, cin, cout, H // 2, W // 2, 1 if relu else 0, _stream()),
           "se_conv2d_3x3_s2_f32")
    return out


def deconv2d_k4s2_assemble(z, bias, batch, cout, h, w, relu=True):
    """``z`` [B,16,cout,h,w] (per-tap GEMM results of a ConvTranspose2d(4, 2, 1), tap = 4 ky + kx) -> relu?(y + bias) [B,cout,2h,2w]."""
    require_hip(z, bias)
    _chk_f32(z, bias)
    assert z.is_contiguous() and z.numel() == batch * 16 * cout * h * w
    out = torch.empty((batch, cout, 2 * h, 2 * w), device=z.device, dtype=torch.float32)
    _check(load().se_deconv2d_k4s2_assemble_f32(_ptr(z), _ptr(bias), _ptr(out), batch, cout, h, w, 1 if relu else 0, _stream()),
           "se_deconv2d_k4s2_assemble_f32")
    return out


def conv3d_packed_elems(cout, cin_pad, ksize, transposed, bf16=False) -> int:
    fn = load().se_conv3d_packed_elems_bf16 if bf16 else load().se_conv3d_packed_elems
    n = int(fn(cout, cin_pad, ksize, 1 if transposed else 0))
    if n <= 0:
        raise HipExtensionError(f"unsupported conv shape cout={cout} cin_pad={cin_pad} k={ksize}")
    return n


def conv3d_variant(batch, dim, cin, cout, ksize, flags=0) -> int:
    """The kernel a launch of ``batch`` samples with these layout flags (IN_OCTET / IN_QUAD ...) runs on: conv3d_algo()'s value, except
    3 = the F(4,3) x F(4,3) member of the 2-D Winograd family (same fused forms as 2; its planar layout is QUAD-planar: it takes a
    quad-planar input or a channels-last one with < 32 channels; any octet-planar flag keeps the launch on 2) and 0 for a 2-D Winograd
    shape with <= 4096 voxels in the batch (16^3 at batch 1) called without planar forms, which the in-workgroup split-K kernel serves
    better; the V2V program keeps such a level channels-last."""
    return int(load().se_conv3d_f32_variant(batch, dim, cin, cout, ksize, flags))


def conv3d_algo(dim, cin, cout, ksize) -> int:
    """Kernel family se_conv3d_f32 selects for this float32 shape (include/sceneego_hip.h: 0 direct, 1 / 2 Winograd 1-D / 2-D)."""
    return int(load().se_conv3d_f32_algo(dim, cin, cout, ksize))


def conv3d_pack(w, b, gamma, beta, mean, var, eps, wpack, bpack, cout, cin, cin_pad, ksize, transposed):
    require_hip(w, wpack, bpack)
    _chk_f32(w, b, gamma, beta, mean, var, bpack)
    if wpack.dtype == torch.bfloat16:
        _check(load().se_conv3d_pack_bf16(_ptr(w), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), float(eps),
                                          _ptr(wpack), _ptr(bpack), cout, cin, cin_pad, ksize, 1 if transposed else 0,
                                          _stream()), "se_conv3d_pack_bf16")
        return
    _chk_f32(wpack)
    _check(load().se_conv3d_pack_f32(_ptr(w), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), float(eps),
                                     _ptr(wpack), _ptr(bpack), cout, cin, cin_pad, ksize, 1 if transposed else 0,
                                     _stream()), "se_conv3d_pack_f32")


# Optional per-launch timing (bench.py's roofline leg): HIP events recorded on the launch stream around every
# conv launch, keyed by shape.  None = off (the default; nothing is recorded in normal operation).
_prof = None
last_launch_detail = []     # (key, flags or None, ms) of every launch of the last profiled pass, in issue order (bench.py: per-launch kernel variant)
last_launch_order = []      # launch keys of the last profiled pass, in issue order (tools/pmc_r03.py labels rocprofv3 dispatches with it)


def start_profile():
    global _prof
    _prof = []


def stop_profile():
    """-> {key: [ms, ...]} ; caller must have synchronised the device."""
    global _prof, last_launch_order, last_launch_detail
    rec, _prof = _prof, None
    out = {}
    detail = []
    for r in rec or []:
        key, e0, e1 = r[0], r[1], r[2]
        ms = e0.elapsed_time(e1)
        out.setdefault(key, []).append(ms)
        if key[0] != "stage":
            detail.append((key, r[3] if len(r) > 3 else None, ms))
    last_launch_order = [d[0] for d in detail]
    last_launch_detail = detail
    return out


class stage:
    """HIP events around one STAGE of the forward (backbone / gather / voxelise / v2v / softargmax) when profiling is on
    (bench.py's separate untimed pass); a no-op otherwise.  Keys are ("stage", name)."""

    def __init__(self, name):
        self.key = ("stage", name)

    def __enter__(self):
        if _prof is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _prof is not None and exc[0] is None:
            self.e1.record()
            _prof.append((self.key, self.e0, self.e1))
        return False


class _timed:
    """HIP events around one launch when profiling is on (bench.py); a no-op otherwise."""

    def __init__(self, key, flags=None):
        self.key = key
        self.flags = flags

    def __enter__(self):
        if _prof is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _prof is not None and exc[0] is None:
            self.e1.record()
            _prof.append((self.key, self.e0, self.e1, self.flags))
        return False


def _tag(t):
    return "" if t.dtype == torch.float32 else "_bf16"


def conv3d(inp, wpack, bpack, residual, out, batch, dim, cin, cin_pad, cout, ksize, flags, workspace=None, pool_out=None):
    """``pool_out`` (float32, 2-D Winograd 3x3x3 shapes only): channels-last [B, D/2, D/2, D/2, cout] tensor that also receives
    max_pool3d(out, 2, 2) from the kernel's epilogue (se_conv3d_pool_f32).
    ``workspace`` serves ONE stream at a time: the split-K levels (4^3, 2^3) and - when given - the chunk-half split of the 7^3 Winograd
    layer at batch 1 pass partial sums through it (a second stream needs its own: V2VProgram.workspace_side).  With a workspace the 7^3
    Winograd layer sums a frame in a different float32 order at batch 1 than at batch > 1; the frequency-domain front
    layer (conv3d_k7_fft, the production path since round 6) is batch-invariant bit for bit."""
    require_hip(inp, out)
    if _prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if inp.dtype == torch.bfloat16:
        assert wpack.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
        assert inp.shape[-1] == (8 if ksize == 7 else cin_pad)       # the 7^3 layer reads octet-planar input
        assert residual is None or residual.dtype == torch.bfloat16
        _check(load().se_conv3d_bf16(_ptr(inp), _ptr(wpack), _ptr(bpack), _ptr(residual), _ptr(out), batch, dim,
                                     cin_pad, cout, ksize, flags, _stream()), "se_conv3d_bf16")
    elif pool_out is not None:
        require_hip(pool_out)
        assert pool_out.dtype == torch.float32 and pool_out.numel() == batch * (dim // 2) ** 3 * cout and pool_out.is_contiguous()
        _check(load().se_conv3d_pool_f32(_ptr(inp), _ptr(wpack), _ptr(bpack), _ptr(residual), _ptr(out), _ptr(pool_out), batch,
                                         dim, cin, cin_pad, cout, ksize, flags, _ptr(workspace),
                                         0 if workspace is None else workspace.numel(), _stream()), "se_conv3d_pool_f32")
    else:
        _check(load().se_conv3d_f32(_ptr(inp), _ptr(wpack), _ptr(bpack), _ptr(residual), _ptr(out), batch, dim, cin,
                                    cin_pad, cout, ksize, flags, _ptr(workspace),
                                    0 if workspace is None else workspace.numel(), _stream()), "se_conv3d_f32")
    if _prof is not None:
        e1.record()
        _prof.append((("conv3d" if inp.dtype == torch.float32 else "conv3d_bf16", ksize, cin_pad, cout, dim), e0, e1, flags))


def conv3d_k7_fft_supported(dim, cin, cout) -> bool:
    """Shapes the frequency-domain form of the 7x7x7 layer covers (se_conv3d_k7_fft_f32)."""
    return dim >= 16 and dim % 16 == 0 and int(load().se_conv3d_k7_fft_packed_elems(cin, cout)) > 0


def conv3d_k7_fft_pack(w, gamma, var, eps, cout, cin):
    """Conv3d weight [cout][cin][7][7][7] (+ the BatchNorm3d scale) -> its spectra in MFMA fragment order (se_conv3d_k7_fft_pack_f32)."""
    require_hip(w)
    _chk_f32(w, gamma, var)
    n = int(load().se_conv3d_k7_fft_packed_elems(cin, cout))
    if n <= 0:
        raise HipExtensionError(f"frequency-domain 7x7x7 convolution does not cover cin={cin} cout={cout}")
    out = torch.empty(n, device=w.device, dtype=torch.float32)
    _check(load().se_conv3d_k7_fft_pack_f32(_ptr(w), _ptr(gamma), _ptr(var), float(eps), _ptr(out), cout, cin, _stream()),
           "se_conv3d_k7_fft_pack_f32")
    return out


def conv3d_k7_fft_workspace_elems(batch, dim, cin) -> int:
    return int(load().se_conv3d_k7_fft_workspace_elems(batch, dim, cin))


def conv3d_k7_fft(inp, hfrag, bpack, out, batch, dim, cin, cout, flags, workspace):
    """``inp`` planar [B, cin, D, D, D]; ``out`` channels-last [B, D, D, D, 16] or (OUT_QUAD) quad-planar [B, 4, D, D, D, 4];
    ``workspace``: float32 scratch for the spectra of >= 1 sample (conv3d_k7_fft_workspace_elems); one stream at a time."""
    require_hip(inp, hfrag, bpack, out, workspace)
    _chk_f32(inp, hfrag, bpack, out, workspace)
    assert inp.numel() == batch * cin * dim ** 3 and out.numel() == batch * cout * dim ** 3
    with _timed(("conv3d_fft", 7, cin, cout, dim), flags):
        _check(load().se_conv3d_k7_fft_f32(_ptr(inp), _ptr(hfrag), _ptr(bpack), _ptr(out), batch, dim, cin, cout, flags,
                                           _ptr(workspace), workspace.numel(), _stream()), "se_conv3d_k7_fft_f32")


def conv3d_split3_pack(w_folded, cout, cin, cin_pad):
    """EXPERIMENTAL: float32 [cout][cin][3][3][3] weights (BatchNorm scale folded in) -> the split-bf16 kernel's packed (hi, lo) halves."""
    require_hip(w_folded)
    _chk_f32(w_folded)
    n = int(load().se_conv3d_split3_packed_elems(cout, cin_pad))
    if n <= 0:
        raise HipExtensionError(f"split-bf16 convolution does not cover cout={cout} cin_pad={cin_pad}")
    out = torch.empty(n, device=w_folded.device, dtype=torch.bfloat16)
    _check(load().se_conv3d_split3_pack(_ptr(w_folded), _ptr(out), cout, cin, cin_pad, _stream()), "se_conv3d_split3_pack")
    return out


def conv3d_k3_split3(inp, wsplit, bpack, residual, out, batch, dim, cin_pad, cout, flags):
    """EXPERIMENTAL: 3x3x3 convolution on float32 tensors (channels-last, or octet-planar per IN_/OUT_/RES_OCTET) with split-bf16
    arithmetic (se_conv3d_k3_split3_f32)."""
    require_hip(inp, out, wsplit, bpack)
    _chk_f32(inp, out, bpack, residual)
    assert wsplit.dtype == torch.bfloat16
    with _timed(("conv3d_split3", 3, cin_pad, cout, dim)):
        _check(load().se_conv3d_k3_split3_f32(_ptr(inp), _ptr(wsplit), _ptr(bpack), _ptr(residual), _ptr(out), batch, dim, cin_pad, cout,
                                              flags, _stream()), "se_conv3d_k3_split3_f32")


def conv3d_skip16(inp, wpack, bpack_sum, skip_in, skip_w, out, batch, dim, cin, cout, flags):
    """3x3x3 convolution (2-D Winograd kernels, planar in / out) + the block's 1x1x1 skip convolution over the 16-channel ``skip_in``
    (channels-last, or quad-planar with RES_QUAD on the quad family) in one launch (se_conv3d_skip16_f32); ``bpack_sum`` = sum of both
    folded biases."""
    require_hip(inp, out, skip_in, skip_w, bpack_sum)
    _chk_f32(inp, out, skip_in, skip_w, bpack_sum)
    # skip_in: channels-last [B, D, D, D, 16], or with RES_QUAD quad-planar [B, 4, D, D, D, 4]
    assert skip_in.numel() == batch * dim ** 3 * 16
    assert tuple(skip_w.shape) == (cout, 16) and skip_w.is_contiguous() and skip_in.is_contiguous()
    with _timed(("conv3d", 3, cin, cout, dim), flags | 256):        # 256 = SE_EPI_SKIPCONV16 (the entry point sets it)
        _check(load().se_conv3d_skip16_f32(_ptr(inp), _ptr(wpack), _ptr(bpack_sum), _ptr(skip_in), _ptr(skip_w), _ptr(out), batch,
                                           dim, cin, cout, flags, _stream()), "se_conv3d_skip16_f32")


def pointwise_chain3(inp, pc1, pc2, pc3, out, batch, dim, softargmax=None, in_quad=False):
    """back_layers.1 -> back_layers.2 -> output_layer in one launch; pc* are packed 1x1x1 convs (32->32, 32->32, 32->J).
    ``softargmax`` = (coord [dim^3, 3], scratch): float32 only - the launch also writes pass 1 of the soft-argmax (softmax mode) into
    ``scratch``; finish with softargmax3d_finish.  ``in_quad`` (that form only): ``inp`` is quad-planar [B][8][dim^3][4]."""
    require_hip(inp, out)
    assert out.dtype == torch.float32
    assert not in_quad or softargmax is not None
    if softargmax is not None:
        coord, scratch = softargmax
        require_hip(coord, scratch)
        _chk_f32(coord, scratch)
        assert scratch.numel() >= softargmax3d_scratch_elems(batch * pc3.cout) and coord.numel() == 3 * dim ** 3
        if inp.dtype == torch.bfloat16:
            assert not in_quad
            with _timed(("tail_bf16", 1, 32, pc3.cout, dim)):
                _check(load().se_pointwise_chain3_softargmax_bf16(_ptr(inp), _ptr(pc1.w), _ptr(pc1.b), _ptr(pc2.w), _ptr(pc2.b), _ptr(pc3.w),
                                                                  _ptr(pc3.b), _ptr(out), _ptr(coord), _ptr(scratch), batch, dim, pc3.cout,
                                                                  _stream()), "se_pointwise_chain3_softargmax_bf16")
            return
        _chk_f32(inp)
        with _timed(("tail", 1, 32, pc3.cout, dim)):
            _check(load().se_pointwise_chain3_softargmax_f32(_ptr(inp), _ptr(pc1.w), _ptr(pc1.b), _ptr(pc2.w), _ptr(pc2.b), _ptr(pc3.w),
                                                             _ptr(pc3.b), _ptr(out), _ptr(coord), _ptr(scratch), batch, dim, pc3.cout,
                                                             IN_QUAD if in_quad else 0, _stream()), "se_pointwise_chain3_softargmax_f32")
        return
    fn = load().se_pointwise_chain3_bf16 if inp.dtype == torch.bfloat16 else load().se_pointwise_chain3_f32
    with _timed(("tail" + _tag(inp), 1, 32, pc3.cout, dim)):
        _check(fn(_ptr(inp), _ptr(pc1.w), _ptr(pc1.b), _ptr(pc2.w), _ptr(pc2.b), _ptr(pc3.w),
                  _ptr(pc3.b), _ptr(out), batch, dim, pc3.cout, _stream()), "se_pointwise_chain3")


def deconv3d_k2s2(inp, wpack, bpack, residual, out, batch, dim, cin, cout, flags):
    """``flags`` may carry OUT_QUAD (float32, 64 -> 32 and 128 -> 64, dim % 16 == 0): ``out`` is then quad-planar [B][cout/4][2D][2D][2D][4]."""
    require_hip(inp, out)
    fn = load().se_deconv3d_k2s2_bf16 if inp.dtype == torch.bfloat16 else load().se_deconv3d_k2s2_f32
    assert out.dtype == inp.dtype and wpack.dtype == inp.dtype and (residual is None or residual.dtype == inp.dtype)
    with _timed(("deconv" + _tag(inp), 2, cin, cout, dim)):
        _check(fn(_ptr(inp), _ptr(wpack), _ptr(bpack), _ptr(residual), _ptr(out), batch, dim,
                  cin, cout, flags, _stream()), "se_deconv3d_k2s2")


def maxpool3d_2(inp, out, batch, dim, channels, in_octet=False):
    require_hip(inp, out)
    fn = load().se_maxpool3d_2_bf16 if inp.dtype == torch.bfloat16 else load().se_maxpool3d_2_f32
    if in_octet:
        assert inp.dtype == torch.float32
        fn = load().se_maxpool3d_2_octin_f32
    assert out.dtype == inp.dtype
    with _timed(("maxpool" + _tag(inp), 2, channels, channels, dim)):
        _check(fn(_ptr(inp), _ptr(out), batch, dim, channels, _stream()), "se_maxpool3d_2")


def softargmax3d_scratch_elems(rows) -> int:
    return int(load().se_softargmax3d_scratch_elems(rows))


def softargmax3d_finish(vol, scratch, out_vol, joints, rows, voxels, mode):
    """Pass 2 of the soft-argmax from partial records written by pointwise_chain3(..., softargmax=...)."""
    require_hip(vol, scratch, out_vol, joints)
    _chk_f32(vol, scratch, out_vol, joints)
    _check(load().se_softargmax3d_finish_f32(_ptr(vol), _ptr(scratch), _ptr(out_vol), _ptr(joints), rows, voxels, mode, _stream()),
           "se_softargmax3d_finish_f32")


def softargmax3d(vol, coord, out_vol, joints, rows, voxels, mode, scratch=None):
    require_hip(vol, coord, out_vol, joints)
    _chk_f32(vol, coord, out_vol, joints)
    if scratch is None:
        scratch = torch.empty(softargmax3d_scratch_elems(rows), device=vol.device, dtype=torch.float32)
    _check(load().se_softargmax3d_f32(_ptr(vol), _ptr(coord), _ptr(out_vol), _ptr(joints), _ptr(scratch), rows,
                                      voxels, mode, _stream()), "se_softargmax3d_f32")
