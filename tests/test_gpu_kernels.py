"""GPU: every HIP kernel, called through the C ABI, against the CPU oracle / plain PyTorch-CPU float32 ops.

Integer / index work (voxeliser) must be bit-exact; float32 kernels use the tolerances written next to
each assertion (the f32 MFMA is an exact fmaf chain, so differences are summation-order only).
"""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import sceneego_oracle as O
from sceneego_amd import _lib, load_config, op, synth
from sceneego_amd.v2v import V2VModel, _PackedConv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ndhwc(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


# ------------------------------------------------------------------------------------------------
# voxeliser: bit-exact
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def voxel_setup(oracle_constants):
    c = oracle_constants(64)
    tab = torch.from_numpy(op.build_voxelizer_ray_table(c.ray, 1280, 1024)).to(DEV)
    return c, tab


def _hip_voxelize(depth, tab, G=64, side=2):
    B = depth.shape[0]
    occ = torch.full((B, G, G, G), 7.0, device=DEV)      # poison: the kernel must clear it
    _lib.voxelize(depth.to(DEV).contiguous(), tab, occ, B, depth.shape[1], depth.shape[2], 1024, 128, G, side)
    torch.cuda.synchronize()
    return occ.cpu()


@pytest.mark.parametrize("kind", ["uniform", "floor"])
def test_voxelize_bit_exact(voxel_setup, kind):
    c, tab = voxel_setup
    _, depth = synth.make_inputs(11, 2, kind)
    got = _hip_voxelize(depth, tab)
    for b in range(2):
        want = O.depth_to_voxel(depth[b].numpy(), c.ray, 64, 2)
        assert torch.equal(got[b], want), f"{int((got[b] != want).sum())} voxels differ"


def test_voxelize_edge_cases(voxel_setup):
    c, tab = voxel_setup
    # all-zero depth -> only the (G/2, G/2, 0) voxel; huge depth -> nothing but that voxel; half-way values (ties)
    d0 = torch.zeros(1, 1024, 1280)
    got = _hip_voxelize(d0, tab)
    assert int(got.sum()) == 1 and float(got[0, 32, 32, 0]) == 1.0
    assert torch.equal(got[0], O.depth_to_voxel(d0[0].numpy(), c.ray, 64, 2))
    d1 = torch.full((1, 1024, 1280), 10.0)
    assert torch.equal(_hip_voxelize(d1, tab)[0], O.depth_to_voxel(d1[0].numpy(), c.ray, 64, 2))
    # depths chosen so that p.z*32 lands exactly on k + 0.5 for the central pixel: exercises half-to-even
    d2 = torch.full((1, 1024, 1280), 0.25)
    d2[0, 512, 640] = float(np.float32(4.5 / 32.0 / c.ray[(640 + 0) * 1024 + 512][2]))
    assert torch.equal(_hip_voxelize(d2, tab)[0], O.depth_to_voxel(d2[0].numpy(), c.ray, 64, 2))
    # non-native depth size (640x512, the demo EXRs' native size): resize indices floor(dst*src/dst)
    d3 = synth.uniform(5, "d3", (1, 512, 640), 0.3, 3.0)
    got = _hip_voxelize(torch.from_numpy(d3), tab)
    assert torch.equal(got[0], O.depth_to_voxel(d3[0], c.ray, 64, 2))


def test_voxelize_golden_bits(voxel_setup, golden, golden_meta):
    c, tab = voxel_setup
    for case in ("b2_uniform", "b1_floor"):
        m = next(x for x in golden_meta["cases"] if x["name"] == case)
        g = golden(case)
        _, depth = synth.make_inputs(m["input_seed"], m["batch"], m["depth_kind"])
        got = _hip_voxelize(depth, tab).reshape(m["batch"], -1).numpy().astype(np.uint8)
        for b in range(m["batch"]):
            assert np.array_equal(np.packbits(got[b]), g["occupancy_bits"][b])


def test_voxelize_128_and_full(oracle_constants):
    c = oracle_constants(64)
    tab = torch.from_numpy(op.build_voxelizer_ray_table(c.ray, 1280, 1024)).to(DEV)
    _, depth = synth.make_inputs(3, 1, "floor")
    occ = torch.empty((1, 128, 128, 128), device=DEV)
    _lib.voxelize(depth.to(DEV), tab, occ, 1, 1024, 1280, 1024, 128, 128, 2)
    assert torch.equal(occ.cpu()[0], O.depth_to_voxel(depth[0].numpy(), c.ray, 128, 2))
    # full-width variant (dataset/real_depth_utils.py:29-60): rays of all 1280 columns, [y][x] order
    full_tab = torch.from_numpy(np.ascontiguousarray(c.ray.reshape(1280, 1024, 3).transpose(1, 0, 2))).to(DEV)
    occ2 = torch.empty((1, 64, 64, 64), device=DEV)
    _lib.voxelize_full(depth.to(DEV), full_tab, occ2, 1, 1024, 1280, 64, 2)
    assert torch.equal(occ2.cpu()[0], O.depth_to_voxel_full(depth[0].numpy(), c.ray, 64, 2))


# ------------------------------------------------------------------------------------------------
# gather
# ------------------------------------------------------------------------------------------------
def test_gather_vs_literal_grid_sample(oracle_constants):
    c = oracle_constants(64)
    feat = torch.from_numpy(synth.normal(1, "feat", (2, 32, 64, 64)))
    big = F.pad(F.interpolate(feat, size=(1024, 1024), mode="nearest"), (128, 128, 0, 0))
    want = O.unproject(big, c.grid, 64)                                   # [2,32,64,64,64]
    idx, w = op.build_gather_table(c.grid, (1024, 1280), 64)
    out = torch.full((2, 64 ** 3, 48), -5.0, device=DEV)
    _lib.unproject_gather(feat.permute(0, 2, 3, 1).contiguous().to(DEV), idx.to(DEV), w.to(DEV), out, 2, 4096, 32,
                          64 ** 3, 48, 0)
    got = out.cpu()
    assert float((got[..., :32].reshape(2, 64, 64, 64, 32).permute(0, 4, 1, 2, 3) - want).abs().max()) < 2e-6
    assert float(got[..., 32:].min()) == -5.0 and float(got[..., 32:].max()) == -5.0   # untouched channels
    # generic operator form (utils/op.py:194-214 signature) on an arbitrary image
    img = torch.from_numpy(synth.normal(2, "img", (1, 4, 96, 120)))
    g = torch.from_numpy(synth.uniform(3, "g", (1, 8 ** 3, 1, 2), -1.1, 1.1))
    want2 = F.grid_sample(img, g, align_corners=True).view(1, 4, 8, 8, 8)
    got2 = op.unproject_heatmaps_one_view_batch(img.to(DEV), g.to(DEV), 8).cpu()
    assert float((got2 - want2).abs().max()) < 2e-6


def test_intersection():
    buf = torch.zeros(1, 1000, 80, device=DEV)
    vol = torch.randn(1, 1000, 32, device=DEV)
    occ = (torch.rand(1, 1000, device=DEV) > 0.5).float()
    buf[..., :32] = vol
    _lib.intersection(buf, occ, 1, 1000, 32, 80)
    assert torch.equal(buf[..., 32:64], vol * occ[..., None]) and float(buf[..., 64:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------
# conv3d / deconv / pool against torch CPU float32
# ------------------------------------------------------------------------------------------------
def _rand_bn(c, seed):
    bn = nn.BatchNorm3d(c).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(synth.uniform(seed, "g", (c,), 0.5, 1.5)))
        bn.bias.copy_(torch.from_numpy(synth.uniform(seed, "b", (c,), -0.3, 0.3)))
        bn.running_mean.copy_(torch.from_numpy(synth.uniform(seed, "m", (c,), -0.3, 0.3)))
        bn.running_var.copy_(torch.from_numpy(synth.uniform(seed, "v", (c,), 0.5, 1.5)))
    return bn


CONV_CASES = [
    # (B, dim, cin, cout, k, relu, residual, bn)
    (2, 8, 16, 32, 3, True, False, True),
    (1, 16, 32, 32, 3, True, True, True),
    (3, 4, 64, 64, 3, False, False, True),
    (2, 2, 128, 128, 3, True, True, True),
    (8, 2, 128, 128, 3, True, True, True),      # split-K shapes of the real pyramid at B = 8
    (8, 4, 128, 128, 3, True, False, True),
    (2, 8, 128, 128, 3, True, True, True),
    (4, 8, 128, 128, 3, True, True, True),      # 2048..8192 voxels at 8^3 / 16^3, cin % 32 == 0: 64-voxel LDS-halo tiles (conv3d_k3_halo64_kernel)
    (8, 8, 128, 128, 3, True, True, True),      # ... the 8^3 level of the pyramid at B = 8
    (1, 16, 128, 128, 3, True, True, True),     # ... the 16^3 level at B = 1 (four-row tiles)
    (8, 8, 64, 128, 3, True, False, True),      # ... two 32-channel stages
    (16, 8, 32, 64, 3, False, True, False),     # ... one stage, 8192 voxels, skip tensor without a ReLU
    (4, 8, 48, 64, 3, True, True, True),        # cin % 32 != 0: in-workgroup split-K from the vector cache (conv3d_k3_wavesplit_kernel)
    (8, 8, 64, 64, 3, False, False, True),
    (1, 8, 128, 64, 3, True, False, False),
    (2, 8, 16, 32, 1, False, False, True),
    (1, 16, 32, 32, 1, True, False, True),
    (1, 12, 32, 16, 7, True, False, True),      # non power-of-two volume
    (2, 8, 48, 16, 7, True, False, True),
    (1, 6, 16, 48, 3, True, True, True),        # odd number of cout tiles -> N_T = 1 path
    # LDS-tiled kernels (dim % 8 == 0, dim >= 16)
    (2, 16, 16, 32, 3, True, False, True),
    (1, 32, 32, 32, 3, True, True, True),       # persistent LDS-weights kernel, 2 chunks
    (2, 32, 16, 32, 3, True, False, True),      # persistent, 1 chunk
    (1, 16, 64, 64, 3, True, True, True),
    (2, 16, 128, 128, 3, False, False, True),
    (1, 24, 32, 64, 3, True, False, True),      # non power-of-two volume, tiled
    (1, 16, 16, 48, 3, True, True, True),       # N_T = 1 tiled
    (1, 16, 32, 16, 7, True, False, True),      # tiled 7^3, 4-channel chunks / tap lanes
    (2, 24, 48, 16, 7, False, False, True),
    (34, 16, 32, 32, 3, True, True, True),      # batch > 32: sliced launches of the persistent kernels
]


@pytest.mark.parametrize("B,dim,cin,cout,k,relu,residual,bn", CONV_CASES)
def test_conv3d_vs_torch(B, dim, cin, cout, k, relu, residual, bn):
    seed = hash((B, dim, cin, cout, k)) % 1000
    conv = nn.Conv3d(cin, cout, k, padding=(k - 1) // 2)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(synth.normal(seed, "w", tuple(conv.weight.shape), (2.0 / (cin * k ** 3)) ** 0.5)))
        conv.bias.copy_(torch.from_numpy(synth.uniform(seed, "cb", (cout,), -0.2, 0.2)))
    bnm = _rand_bn(cout, seed) if bn else None
    x = torch.from_numpy(synth.normal(seed, "x", (B, cin, dim, dim, dim)))
    res = torch.from_numpy(synth.normal(seed, "r", (B, cout, dim, dim, dim))) if residual else None
    with torch.no_grad():
        want = conv(x)
        if bnm is not None:
            want = bnm(want)
        if residual:
            want = want + res
        if relu:
            want = F.relu(want)
    pc = _PackedConv(conv.to(DEV), bnm.to(DEV) if bnm is not None else None)
    out = torch.empty((B, dim, dim, dim, cout), device=DEV)
    flags = (_lib.EPI_RELU if relu else 0) | (_lib.EPI_RES_PRE_RELU if residual else 0)
    ws = torch.empty(4 << 20, device=DEV)
    for workspace in (None, ws):      # without / with the split-K workspace (only small volumes take that path)
        out.fill_(-77.0)
        _lib.conv3d(_ndhwc(x).to(DEV), pc.w, pc.b, _ndhwc(res).to(DEV) if residual else None, out, B, dim, cin, cin,
                    cout, k, flags, workspace)
        got = _ncdhw(out.cpu())
        err = float((got - want).abs().max())
        assert err < 2e-5 * max(1.0, float(want.abs().max())), (err, workspace is not None)


SPLIT3_CASES = [(1, 16, 32, 32, True, True), (2, 32, 16, 32, True, False), (1, 16, 64, 128, False, False), (1, 32, 32, 64, True, True),
                (2, 16, 128, 128, True, True), (3, 16, 48, 32, True, True)]


@pytest.mark.parametrize("B,dim,cin,cout,relu,residual", SPLIT3_CASES)
def test_conv3d_split_bf16_vs_torch(B, dim, cin, cout, relu, residual):
    """EXPERIMENTAL split-bf16 3x3x3 convolution (se_conv3d_k3_split3_f32: float32 tensors, every product as hi*hi + hi*lo + lo*hi on
    the bf16 MFMA, float32 accumulation) against torch-CPU float32 Conv3d + BatchNorm3d (+ skip) (+ ReLU), reference
    network/v2v.py:21-43.  Bound: 1e-4 of max|y| (16 mantissa bits per operand; the float32 kernels' bound is 2e-5); measured ~1e-5."""
    seed = hash((B, dim, cin, cout, 3)) % 1000
    conv = nn.Conv3d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(synth.normal(seed, "w", tuple(conv.weight.shape), (2.0 / (cin * 27)) ** 0.5)))
        conv.bias.copy_(torch.from_numpy(synth.uniform(seed, "cb", (cout,), -0.2, 0.2)))
    bnm = _rand_bn(cout, seed)
    x = torch.from_numpy(synth.normal(seed, "x", (B, cin, dim, dim, dim)))
    res = torch.from_numpy(synth.normal(seed, "r", (B, cout, dim, dim, dim))) if residual else None
    with torch.no_grad():
        want = bnm(conv(x))
        if residual:
            want = want + res
        if relu:
            want = F.relu(want)
    pc = _PackedConv(conv.to(DEV), bnm.to(DEV), split3=True)
    assert pc.w_split is not None and pc.w_split.dtype == torch.bfloat16
    out = torch.full((B, dim, dim, dim, cout), -77.0, device=DEV)
    flags = (_lib.EPI_RELU if relu else 0) | (_lib.EPI_RES_PRE_RELU if residual else 0)
    _lib.conv3d_k3_split3(_ndhwc(x).to(DEV), pc.w_split, pc.b, _ndhwc(res).to(DEV) if residual else None, out, B, dim, pc.cin_pad,
                          cout, flags)
    got = _ncdhw(out.cpu())
    err = float((got - want).abs().max())
    scale = max(1.0, float(want.abs().max()))
    print(f"split-bf16 conv {cin}->{cout} @{dim}^3: max error {err:.2e} = {err / scale:.2e} of max|y|")
    assert err < 1e-4 * scale, (err, scale)
    # the float32 kernel on the same data, for scale
    out32 = torch.empty_like(out)
    _lib.conv3d(_ndhwc(x).to(DEV), pc.w, pc.b, _ndhwc(res).to(DEV) if residual else None, out32, B, dim, cin, pc.cin_pad, cout, 3, flags, None)
    assert float((out - out32).abs().max()) < 1e-4 * scale
    # octet-planar forms (SE_IN_OCTET / SE_OUT_OCTET / SE_RES_OCTET): same arithmetic in the same order -> bit-identical
    to_oct = lambda t, c: t.view(B, dim, dim, dim, c // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous()
    from_oct = lambda t, c: t.permute(0, 2, 3, 4, 1, 5).reshape(B, dim, dim, dim, c)
    xin = torch.zeros(B, dim, dim, dim, pc.cin_pad, device=DEV)
    xin[..., :cin] = _ndhwc(x).to(DEV)
    res_cl = _ndhwc(res).to(DEV) if residual else None
    for fl in (_lib.IN_OCTET, _lib.OUT_OCTET, _lib.IN_OCTET | _lib.OUT_OCTET | (_lib.RES_OCTET if residual else 0)):
        o = torch.full((B, cout // 8, dim, dim, dim, 8) if fl & _lib.OUT_OCTET else (B, dim, dim, dim, cout), -77.0, device=DEV)
        _lib.conv3d_k3_split3(to_oct(xin, pc.cin_pad) if fl & _lib.IN_OCTET else xin, pc.w_split, pc.b,
                              (to_oct(res_cl, cout) if fl & _lib.RES_OCTET else res_cl) if residual else None, o, B, dim, pc.cin_pad, cout,
                              flags | fl)
        assert torch.equal(from_oct(o, cout) if fl & _lib.OUT_OCTET else o, out), fl


def test_conv3d_split_bf16_refuses_unsupported_shapes():
    t = torch.zeros(64, device=DEV)
    import ctypes
    p = ctypes.c_void_p(t.data_ptr())
    lib = _lib.load()
    assert lib.se_conv3d_k3_split3_f32(p, p, p, None, p, 1, 8, 32, 32, 0, None) == -1      # dim % 16
    assert lib.se_conv3d_k3_split3_f32(p, p, p, None, p, 1, 16, 20, 32, 0, None) == -1     # cin_pad % 8
    assert lib.se_conv3d_k3_split3_f32(p, p, p, None, p, 1, 16, 32, 16, 0, None) == -1     # cout % 32
    assert lib.se_conv3d_k3_split3_f32(p, p, p, None, p, 1, 16, 32, 32, 256, None) == -1        # unknown flag bit
    assert lib.se_conv3d_k3_split3_f32(p, None, p, None, p, 1, 16, 32, 32, 0, None) == -1
    assert lib.se_conv3d_split3_packed_elems(48, 32) == -1 and lib.se_conv3d_split3_packed_elems(32, 32) == 2 * 1 * 4 * 896 * 8


def test_conv3d_padded_input_channels_and_planar_output():
    """33 real channels inside a 48-channel buffer (front conv) and the 15-channel planar output layer."""
    for dim in (8, 16):     # direct kernel, then the LDS-tiled 7^3 kernel (occupancy channel in a 9th 4-channel chunk)
        conv = nn.Conv3d(33, 16, 7, padding=3)
        x = torch.from_numpy(synth.normal(5, "x", (1, 33, dim, dim, dim)))
        with torch.no_grad():
            want = conv(x)
        pc = _PackedConv(conv.to(DEV), None, cin_pad=48)
        xin = torch.zeros(1, dim, dim, dim, 48, device=DEV)
        xin[..., :33] = _ndhwc(x).to(DEV)
        xin[..., 33:] = 3.0   # finite garbage in the pad channels is multiplied by zero weights
        out = torch.empty((1, dim, dim, dim, 16), device=DEV)
        _lib.conv3d(xin, pc.w, pc.b, None, out, 1, dim, 33, 48, 16, 7, 0)
        assert float((_ncdhw(out.cpu()) - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    conv2 = nn.Conv3d(32, 15, 1)
    x2 = torch.from_numpy(synth.normal(6, "x", (2, 32, 8, 8, 8)))
    with torch.no_grad():
        want2 = conv2(x2)
    pc2 = _PackedConv(conv2.to(DEV), None)
    out2 = torch.full((2, 15, 512), 9.0, device=DEV)
    _lib.conv3d(_ndhwc(x2).to(DEV), pc2.w, pc2.b, None, out2, 2, 8, 32, 32, 15, 1, _lib.EPI_OUT_PLANAR)
    assert float((out2.cpu().view(2, 15, 8, 8, 8) - want2).abs().max()) < 1e-5


def test_conv3d_linearity_full_size():
    """Size-independent property at the real 64^3 level: conv(a*x + y) == a*conv(x) + conv(y) (no bias/ReLU)."""
    conv = nn.Conv3d(32, 32, 3, padding=1, bias=False).to(DEV)
    conv.bias = None
    pc = _PackedConv(conv, None)
    assert float(pc.b.abs().max()) == 0.0
    x = torch.randn(1, 64, 64, 64, 32, device=DEV)
    y = torch.randn(1, 64, 64, 64, 32, device=DEV)
    f = lambda t: (lambda o: (_lib.conv3d(t, pc.w, pc.b, None, o, 1, 64, 32, 32, 32, 3, 0), o)[1])(torch.empty_like(t))
    lhs = f(2.0 * x + y)
    rhs = 2.0 * f(x) + f(y)
    assert float((lhs - rhs).abs().max()) < 1e-4
    # and a sampled check against the direct definition at 2 000 random output positions is done in test_v2v_*


# the last three run on deconv3d_k2s2_kernel (all eight sub-positions per workgroup): 2 tiles per wave at >= 4 workgroups per CU
# (the 64^3-level shape at B=8, and a ragged voxel count), 1 tile per wave at >= 2 per CU; the others on the grid.z form
@pytest.mark.parametrize("B,dim,cin,cout,skip", [(2, 4, 128, 128, True), (1, 8, 128, 64, True), (1, 16, 64, 32, False),
                                                  (3, 2, 32, 16, True), (8, 32, 64, 32, True), (4, 34, 64, 32, False),
                                                  (8, 16, 128, 64, True), (1, 32, 64, 32, True), (32, 16, 128, 64, False)])
def test_deconv_vs_torch(B, dim, cin, cout, skip):
    seed = cin + cout + dim
    up = nn.ConvTranspose3d(cin, cout, 2, stride=2)
    with torch.no_grad():
        up.weight.copy_(torch.from_numpy(synth.normal(seed, "w", tuple(up.weight.shape), (2.0 / cin) ** 0.5)))
        up.bias.copy_(torch.from_numpy(synth.uniform(seed, "cb", (cout,), -0.2, 0.2)))
    bn = _rand_bn(cout, seed)
    x = torch.from_numpy(synth.normal(seed, "x", (B, cin, dim, dim, dim)))
    sk = torch.from_numpy(synth.normal(seed, "s", (B, cout, 2 * dim, 2 * dim, 2 * dim)))
    with torch.no_grad():
        want = F.relu(bn(up(x)))
        if skip:
            want = want + sk
    pc = _PackedConv(up.to(DEV), bn.to(DEV))
    out = torch.empty((B, 2 * dim, 2 * dim, 2 * dim, cout), device=DEV)
    _lib.deconv3d_k2s2(_ndhwc(x).to(DEV), pc.w, pc.b, _ndhwc(sk).to(DEV) if skip else None, out, B, dim, cin, cout,
                       _lib.EPI_RELU | (_lib.EPI_RES_POST_RELU if skip else 0))
    assert float((_ncdhw(out.cpu()) - want).abs().max()) < 2e-5
    # quad-planar output (SE_OUT_QUAD, round 5: the decoder hands the block behind it whole 16-byte records): the same arithmetic,
    # the records exchanged across lanes before the store - bit-identical to the channels-last launch
    if (cin, cout) in ((64, 32), (128, 64)) and dim % 16 == 0:
        outq = torch.full((B, cout // 4, 2 * dim, 2 * dim, 2 * dim, 4), -7.0, device=DEV)
        _lib.deconv3d_k2s2(_ndhwc(x).to(DEV), pc.w, pc.b, _ndhwc(sk).to(DEV) if skip else None, outq, B, dim, cin, cout,
                           _lib.EPI_RELU | (_lib.EPI_RES_POST_RELU if skip else 0) | _lib.OUT_QUAD)
        # (bit-identical where the channels-last launch runs the same kernel; the small volumes' channels-last form is the grid.z kernel,
        # whose channel groups are summed in another order)
        same_kernel = (B * dim ** 3 + 63) // 64 >= 512
        assert torch.equal(_unquad(outq), out) if same_kernel else float((_unquad(outq) - out).abs().max()) < 1e-5
        assert float((_ncdhw(_unquad(outq).cpu()) - want).abs().max()) < 2e-5
        if skip:    # ... and with the skip tensor quad-planar too (SE_RES_QUAD: added behind the exchange - the same single addition)
            outq2 = torch.full_like(outq, -7.0)
            _lib.deconv3d_k2s2(_ndhwc(x).to(DEV), pc.w, pc.b, _quad(_ndhwc(sk).to(DEV)), outq2, B, dim, cin, cout,
                               _lib.EPI_RELU | _lib.EPI_RES_POST_RELU | _lib.OUT_QUAD | _lib.RES_QUAD)
            assert torch.equal(outq2, outq)
            with pytest.raises(_lib.HipExtensionError):          # a quad-planar skip tensor needs the quad-planar output
                _lib.deconv3d_k2s2(_ndhwc(x).to(DEV), pc.w, pc.b, _quad(_ndhwc(sk).to(DEV)), out, B, dim, cin, cout,
                                   _lib.EPI_RELU | _lib.EPI_RES_POST_RELU | _lib.RES_QUAD)
    else:
        with pytest.raises(_lib.HipExtensionError):
            _lib.deconv3d_k2s2(_ndhwc(x).to(DEV), pc.w, pc.b, None, out, B, dim, cin, cout, _lib.EPI_RELU | _lib.OUT_QUAD)


def test_maxpool_exact():
    x = torch.from_numpy(synth.normal(9, "x", (2, 32, 16, 16, 16)))
    want = F.max_pool3d(x, 2, 2)
    out = torch.empty((2, 8, 8, 8, 32), device=DEV)
    _lib.maxpool3d_2(_ndhwc(x).to(DEV), out, 2, 16, 32)
    assert torch.equal(_ncdhw(out.cpu()), want)


def test_bad_arguments_are_refused():
    t = torch.zeros(16, device=DEV)
    lib = _lib.load()
    import ctypes
    p = ctypes.c_void_p(t.data_ptr())
    assert lib.se_conv3d_f32(p, p, p, None, p, 1, 8, 20, 20, 16, 3, 0, None, 0, None) == -1  # cin_pad % 16
    assert lib.se_conv3d_f32(p, p, p, None, p, 1, 8, 16, 16, 16, 5, 0, None, 0, None) == -1  # ksize
    assert lib.se_conv3d_f32(p, p, p, None, p, 1, 8, 16, 16, 20, 3, 0, None, 0, None) == -1  # cout % 16 (non planar)
    assert lib.se_conv3d_f32(p, p, p, None, p, 1, 8, 40, 32, 16, 3, 0, None, 0, None) == -1  # cin > cin_pad
    assert lib.se_maxpool3d_2_f32(p, p, 1, 7, 16, None) == -1                       # odd volume
    assert lib.se_softargmax3d_f32(p, p, p, p, p, 1, 6, 1, None) == -1              # voxels % 4
    # the octet-planar / pooled / fused-skip forms exist in the 2-D Winograd kernel only.  Shapes and flag sets that kernel
    # declines are refused (nothing launched) instead of falling through to a kernel that ignores those layouts:
    OCT = _lib.IN_OCTET | _lib.OUT_OCTET
    assert lib.se_conv3d_f32_algo(256, 32, 32, 3) != 2                               # 256^3 x 32 ch x 4 B >= 2^31: not a 2-D Winograd shape
    assert lib.se_conv3d_f32(p, p, p, None, p, 1, 256, 32, 32, 32, 3, OCT, None, 0, None) == -1
    assert lib.se_conv3d_pool_f32(p, p, p, None, p, p, 1, 256, 32, 32, 32, 3, OCT, None, 0, None) == -1
    assert lib.se_conv3d_skip16_f32(p, p, p, p, p, p, 1, 256, 32, 32, OCT | _lib.EPI_RELU, None) == -1
    assert lib.se_conv3d_f32_algo(64, 32, 32, 3) == 2
    assert lib.se_conv3d_f32(p, p, p, p, p, 1, 64, 32, 32, 32, 3, OCT | _lib.EPI_RES_POST_RELU, None, 0, None) == -1   # flag the kernel declines
    assert lib.se_conv3d_f32(p, p, p, None, p, 1, 64, 24, 32, 32, 3, OCT, None, 0, None) == -1                        # cin_pad != cin
    assert lib.se_conv3d_pool_f32(p, p, p, None, p, p, 1, 64, 24, 32, 32, 3, OCT, None, 0, None) == -1


# ------------------------------------------------------------------------------------------------
# soft-argmax
# ------------------------------------------------------------------------------------------------
def test_softargmax_kat_and_random(oracle_constants, golden):
    c = oracle_constants(64)
    g = golden("constants")
    vol = torch.zeros((4, 15, 64, 64, 64))
    vol[:, :, 32, 32, 32] = 1
    vol[:, :, 31, 31, 31] = 1
    cv = c.coord.unsqueeze(0).expand(4, -1, -1, -1, -1).to(DEV)
    kp, v = op.integrate_tensor_3d_with_coordinates(vol.to(DEV), cv, softmax=True)
    # reference's own known-answer case (voxel_net_depth.py:302-320): ~(0,0,1); fp32 noise floor 1.3e-4
    assert float((kp.cpu() - torch.from_numpy(g["kat_softargmax_joints"])).abs().max()) < 2e-4
    assert abs(float(v[0, 0, 32, 32, 32]) / float(g["kat_softargmax_peak"][0]) - 1.0) < 1e-3
    kp2, v2 = op.integrate_tensor_3d_with_coordinates(vol.to(DEV), cv, softmax=False)
    assert float((kp2.cpu() - torch.from_numpy(g["kat_relu_joints"])).abs().max()) < 1e-5
    assert torch.equal(v2.cpu(), F.relu(vol))
    # random peaked logits
    lg = torch.from_numpy(synth.normal(4, "lg", (2, 15, 64, 64, 64), 6.0))
    want_kp, want_v = O.integrate(lg, c.coord, softmax=True, accumulate64=True)   # float64 value of the reference formula
    kp3, v3 = op.integrate_tensor_3d_with_coordinates(lg.to(DEV), cv[:2], softmax=True)
    assert float((kp3.cpu() - want_kp).abs().max()) < 3e-4   # fp32 noise floor of a 262 144-term expectation (KAT: 1.3e-4)
    want_kp32, _ = O.integrate(lg, c.coord, softmax=True)     # the reference formula in float32 on this host (its own einsum noise included)
    assert float((kp3.cpu() - want_kp32).abs().max()) <= 1e-3
    assert float((v3.cpu() - want_v).abs().max()) < 1e-6 + 1e-4 * float(want_v.max())
    s = v3.reshape(2, 15, -1).sum(dim=2)
    assert float((s - 1).abs().max()) < 1e-4
    # spike: forces the max-rescale across chunk partials
    lg2 = lg.clone()
    lg2[0, 3, 63, 63, 63] = 80.0
    kp4, _ = op.integrate_tensor_3d_with_coordinates(lg2.to(DEV), cv[:2], softmax=True)
    assert float((kp4[0, 3].cpu() - c.coord[63, 63, 63]).abs().max()) < 1e-5


def test_bias_act_and_fused_backbone():
    x = torch.randn(2, 8, 6, 10, device=DEV)
    b = torch.randn(8, device=DEV)
    r = torch.randn(2, 8, 6, 10, device=DEV)
    want = F.relu(x + b.view(1, -1, 1, 1) + r)
    got = _lib.bias_act_nchw(x.clone(), b, r, True)
    assert float((got - want).abs().max()) < 1e-6
    got2 = _lib.bias_act_nchw(x.clone(), b, None, False)
    assert float((got2 - (x + b.view(1, -1, 1, 1))).abs().max()) < 1e-6
    from sceneego_amd import pose_resnet
    net = pose_resnet.get_pose_net(None).to(DEV).eval()
    img = torch.randn(2, 3, 256, 256, device=DEV)
    with torch.no_grad():
        ref = net(img, compute_heatmaps=False)[1]
        fused = pose_resnet.FoldedBackbone(net)(img)
    assert float((fused - ref).abs().max()) < 1e-4 * float(ref.abs().max()) + 1e-5


@pytest.mark.parametrize("B,cin,cout,H,residual,relu", [
    (8, 64, 256, 64, True, True), (8, 64, 64, 64, False, True), (8, 256, 128, 64, False, True), (8, 128, 512, 32, True, True),
    (8, 512, 128, 32, False, True), (8, 256, 1024, 16, True, True), (8, 512, 2048, 8, True, True), (8, 2048, 512, 8, False, True), (8, 1024, 256, 16, False, True), (8, 1024, 512, 16, False, True),
    (1, 64, 256, 64, True, False), (3, 128, 64, 16, False, False), (4, 48, 192, 12, True, True)])
def test_conv2d_1x1_fused_gemm_vs_float64(B, cin, cout, H, residual, relu):
    """se_conv2d_1x1_f32 (the backbone's stride-1 1x1 convolutions, network/pose_resnet.py:72-90 with folded BatchNorm) against a float64
    product on the host: the backbone's shapes at B = 8, cout = 192 (64-channel tiles), a map whose side is not a multiple of 16 but whose
    area is (12^2 = 144 = 9 * 16: pixel tiles straddle samples), both epilogue forms.  float32 products summed in float32:
    1e-5 of the largest |y| (measured 1-3e-6)."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g)
    r = torch.randn(B, cout, H, H, generator=g) if residual else None
    want = torch.einsum("oc,bchw->bohw", w.double(), x.double()) + b.double().view(1, -1, 1, 1)
    if residual:
        want = want + r.double()
    if relu:
        want = want.clamp_min(0)
    tile = _lib.conv2d_1x1_tile(B, cin, cout, H * H)
    assert tile in (64, 128)
    wp = _lib.conv2d_1x1_pack(w, tile).to(DEV)
    got = _lib.conv2d_1x1(x.to(DEV), wp, b.to(DEV), r.to(DEV) if residual else None, relu)
    assert got.shape == (B, cout, H, H) and bool(torch.isfinite(got).all())
    err = float((got.double().cpu() - want).abs().max())
    assert err < 1e-5 * float(want.abs().max()), err
    # the producing convolution's bias + ReLU applied on the way in: x' = relu(x + in_bias)
    ib = torch.randn(cin, generator=g)
    want2 = torch.einsum("oc,bchw->bohw", w.double(), (x.double() + ib.double().view(1, -1, 1, 1)).clamp_min(0)) + b.double().view(1, -1, 1, 1)
    if residual:
        want2 = want2 + r.double()
    if relu:
        want2 = want2.clamp_min(0)
    got2 = _lib.conv2d_1x1(x.to(DEV), wp, b.to(DEV), r.to(DEV) if residual else None, relu, ib.to(DEV))
    err2 = float((got2.double().cpu() - want2).abs().max())
    assert err2 < 1e-5 * float(want2.abs().max()), err2
    # uncovered shapes are refused, not mis-computed
    assert _lib.conv2d_1x1_tile(B, cin + 8, cout, H * H) == 0 and _lib.conv2d_1x1_tile(B, cin, cout + 32, H * H) == 0
    assert _lib.conv2d_1x1_tile(1, cin, cout, 40) == 0


@pytest.mark.parametrize("B,cin,cout,H,residual,relu", [
    (1, 512, 128, 32, False, True), (1, 1024, 256, 16, False, True), (1, 2048, 512, 8, False, True), (1, 512, 2048, 8, True, True),
    (2, 1024, 512, 16, False, True), (1, 256, 128, 16, True, False), (4, 64, 48, 4, False, False), (1, 192, 16, 8, True, True)])
def test_conv2d_1x1_small_m_form_vs_float64(B, cin, cout, H, residual, relu):
    """se_conv2d_1x1_small_f32 (the long-K 1x1 layers at batch 1-2: 64 x 16 tiles, k over four / two wave groups) against a float64 product,
    with and without the input-side bias + ReLU; 4 x 4 maps (a 64-pixel tile spans four samples), cin = 192 (two groups, run-time loop)."""
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g)
    r = torch.randn(B, cout, H, H, generator=g) if residual else None
    ib = torch.randn(cin, generator=g)
    assert _lib.conv2d_1x1_small_ok(B, cin, cout, H * H) and not _lib.conv2d_1x1_small_ok(B, cin + 32, cout, H * H)
    wp = _lib.conv2d_1x1_pack(w, 16).to(DEV)
    for inb in (None, ib):
        xin = x.double() if inb is None else (x.double() + inb.double().view(1, -1, 1, 1)).clamp_min(0)
        want = torch.einsum("oc,bchw->bohw", w.double(), xin) + b.double().view(1, -1, 1, 1)
        if residual:
            want = want + r.double()
        if relu:
            want = want.clamp_min(0)
        got = _lib.conv2d_1x1_small(x.to(DEV), wp, b.to(DEV), r.to(DEV) if residual else None, relu, None if inb is None else inb.to(DEV))
        assert got.shape == (B, cout, H, H) and bool(torch.isfinite(got).all())
        err = float((got.double().cpu() - want).abs().max())
        assert err < 1e-5 * float(want.abs().max()), err


def test_bias_relu_maxpool_stem_tail_exact():
    """se_bias_relu_maxpool3x3s2_f32 against max_pool2d(relu(x + bias), 3, 2, 1): the same float32 operations in another order (max is
    exact, relu(. + b) monotone) - bit-identical, borders included."""
    g = torch.Generator().manual_seed(11)
    for B, C, H, W in ((2, 64, 128, 128), (1, 3, 6, 8), (3, 5, 10, 24)):
        x = torch.randn(B, C, H, W, generator=g).to(DEV)
        b = torch.randn(C, generator=g).to(DEV)
        want = F.max_pool2d(F.relu(x + b.view(1, -1, 1, 1)), 3, stride=2, padding=1)
        got = _lib.bias_relu_maxpool(x, b)
        assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("B,cin,cout,ho,wo", [(8, 256, 512, 32, 32), (8, 512, 1024, 16, 16), (8, 1024, 2048, 8, 8), (1, 256, 512, 32, 32), (2, 64, 128, 4, 24)])
def test_conv2d_1x1_stride2_vs_float64(B, cin, cout, ho, wo):
    """se_conv2d_1x1_s2_f32 (the `downsample` convolution of a stage's first Bottleneck, network/pose_resnet.py:140-146, BatchNorm folded)
    against a float64 product of x[:, :, ::2, ::2] on the host; the odd rows / columns of x hold NaN (never read)."""
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.full((B, cin, 2 * ho, 2 * wo), float("nan"))
    x[:, :, ::2, ::2] = torch.randn(B, cin, ho, wo, generator=g)
    w = torch.randn(cout, cin, generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g)
    want = torch.einsum("oc,bchw->bohw", w.double(), x[:, :, ::2, ::2].double()) + b.double().view(1, -1, 1, 1)
    tile = _lib.conv2d_1x1_tile(B, cin, cout, ho * wo)
    got = _lib.conv2d_1x1_s2(x.to(DEV), _lib.conv2d_1x1_pack(w, tile).to(DEV), b.to(DEV), False)
    assert got.shape == (B, cout, ho, wo) and bool(torch.isfinite(got).all())
    assert float((got.double().cpu() - want).abs().max()) < 1e-5 * float(want.abs().max())
    # the small-M form of the same operator (16-channel packing)
    got = _lib.conv2d_1x1_s2(x.to(DEV), _lib.conv2d_1x1_pack(w, 16).to(DEV), b.to(DEV), False)
    assert got.shape == (B, cout, ho, wo) and bool(torch.isfinite(got).all())
    assert float((got.double().cpu() - want).abs().max()) < 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("B,cin,cout,H,W,bias,relu", [
    (8, 256, 256, 16, 16, False, False), (8, 512, 512, 8, 8, False, False), (8, 128, 128, 32, 32, True, True), (1, 64, 64, 64, 64, True, False),
    (2, 32, 96, 8, 24, True, True), (3, 64, 32, 12, 16, False, True), (1, 512, 512, 8, 8, True, True)])
def test_conv2d_3x3_direct_mfma_vs_float64(B, cin, cout, H, W, bias, relu):
    """se_conv2d_3x3_f32 (`conv3x3` of the Bottlenecks, network/pose_resnet.py:22-25,78 with folded BatchNorm) against a float64 convolution
    on the host: the layer3 / layer4 shapes at B = 8 (raw sums, as the backbone uses them), 16- and 8-pixel-wide tiles, non-square maps,
    both channel tiles, bias + ReLU in the epilogue.  Zero padding at every map border.  1e-5 of the largest |y|."""
    g = torch.Generator().manual_seed(cin * 3 + cout + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    want = F.conv2d(x.double(), w.double(), b.double() if bias else None, padding=1)
    if relu:
        want = want.clamp_min(0)
    tile = _lib.conv2d_3x3_tile(B, cin, cout, H, W)
    assert tile in (16, 32)
    wp = _lib.conv2d_3x3_pack(w, tile).to(DEV)
    got = _lib.conv2d_3x3(x.to(DEV), wp, b.to(DEV) if bias else None, relu)
    assert got.shape == (B, cout, H, W) and bool(torch.isfinite(got).all())
    err = float((got.double().cpu() - want).abs().max())
    assert err < 1e-5 * float(want.abs().max()), err
    assert _lib.conv2d_3x3_tile(B, cin + 16, cout, H, W) == 0 and _lib.conv2d_3x3_tile(B, cin, cout + 16, H, W) == 0
    assert _lib.conv2d_3x3_tile(B, cin, cout, 6, 16) == 0 and _lib.conv2d_3x3_tile(B, cin, cout, 8, 12) == 0


@pytest.mark.parametrize("B,cin,cout,ho,wo,bias", [(8, 128, 128, 32, 32, False), (8, 256, 256, 16, 16, False), (8, 512, 512, 8, 8, False),
                                                    (1, 256, 256, 16, 16, True), (2, 32, 48, 8, 24, True), (3, 64, 16, 4, 16, False)])
def test_conv2d_3x3_stride2_vs_float64(B, cin, cout, ho, wo, bias):
    """se_conv2d_3x3_s2_f32 (conv2 of a stage's first Bottleneck: stride 2, padding 1; network/pose_resnet.py:22-25,78) against a float64
    convolution on the host: the three backbone shapes at B = 8, 16- and 8-pixel-wide tiles, non-square maps.  1e-5 of the largest |y|."""
    g = torch.Generator().manual_seed(cin + 3 * cout + ho)
    x = torch.randn(B, cin, 2 * ho, 2 * wo, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    want = F.conv2d(x.double(), w.double(), b.double() if bias else None, stride=2, padding=1)
    assert _lib.conv2d_3x3_s2_ok(cin, cout, ho, wo) and not _lib.conv2d_3x3_s2_ok(cin + 16, cout, ho, wo) and not _lib.conv2d_3x3_s2_ok(cin, cout, 6, 16)
    got = _lib.conv2d_3x3_s2(x.to(DEV), _lib.conv2d_3x3_pack(w, 16).to(DEV), b.to(DEV) if bias else None, False)
    assert got.shape == (B, cout, ho, wo) and bool(torch.isfinite(got).all())
    err = float((got.double().cpu() - want).abs().max())
    assert err < 1e-5 * float(want.abs().max()), err


def test_backbone_fused_1x1_matches_miopen_route(monkeypatch):
    """FoldedBackbone with the 1x1 convolutions on se_conv2d_1x1_f32 (default) against the same folded network with every one of them on
    MIOpen + se_bias_act_nchw_f32 (SCENEEGO_CONV1X1=0), B = 8 and B = 1 (the routing rule keeps the small launches on MIOpen)."""
    from sceneego_amd import pose_resnet
    net = pose_resnet.get_pose_net(None).to(DEV).eval()
    for B in (8, 1):
        img = torch.randn(B, 3, 256, 256, device=DEV)
        monkeypatch.setenv("SCENEEGO_CONV1X1", "1")
        fb = pose_resnet.FoldedBackbone(net)
        calls = []
        real = _lib.conv2d_1x1
        monkeypatch.setattr(_lib, "conv2d_1x1", lambda *a: (calls.append(1), real(*a))[1])
        a = fb(img)
        monkeypatch.setattr(_lib, "conv2d_1x1", real)
        monkeypatch.setenv("SCENEEGO_CONV1X1", "0")
        bref = pose_resnet.FoldedBackbone(net)(img)
        assert len(calls) >= (20 if B == 8 else 1)
        assert float((a - bref).abs().max()) < 2e-5 * float(bref.abs().max()) + 1e-6


@pytest.mark.parametrize("B,ci,co,H,W", [(2, 24, 8, 5, 3), (8, 2048, 256, 8, 8), (1, 256, 256, 16, 16)])
def test_deconv2d_gemm_plus_assemble_vs_torch(B, ci, co, H, W):
    """se_deconv2d_k4s2_assemble_f32: the pose head's ConvTranspose2d(4, 2, 1) + folded BN + ReLU (reference
    network/pose_resnet.py:205-224,238) as one GEMM over the un-shifted input + the assembly pass, against torch-CPU
    F.conv_transpose2d (odd sizes: every edge case of the 2 x 2 tap table; then the two production shapes)."""
    g = torch.Generator().manual_seed(ci + H)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(ci, co, 4, 4, generator=g) * (1.0 / (4 * ci)) ** 0.5
    bias = torch.randn(co, generator=g)
    want = F.relu(F.conv_transpose2d(x, w, bias, stride=2, padding=1))
    w_all = w.permute(2, 3, 1, 0).reshape(16 * co, ci).contiguous().to(DEV)
    z = torch.matmul(w_all, x.to(DEV).reshape(B, ci, H * W))
    got = _lib.deconv2d_k4s2_assemble(z, bias.to(DEV), B, co, H, W, relu=True).cpu()
    assert tuple(got.shape) == (B, co, 2 * H, 2 * W)
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    lin = _lib.deconv2d_k4s2_assemble(z, bias.to(DEV), B, co, H, W, relu=False).cpu()
    assert float((lin - F.conv_transpose2d(x, w, bias, stride=2, padding=1)).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def test_pointwise_chain_matches_separate_layers():
    """Fused back_layers.1/.2 + output_layer (one launch) == the three separate 1x1x1 launches == torch CPU."""
    c1, c2, c3 = nn.Conv3d(32, 32, 1), nn.Conv3d(32, 32, 1), nn.Conv3d(32, 15, 1)
    bn1, bn2 = _rand_bn(32, 1), _rand_bn(32, 2)
    x = torch.from_numpy(synth.normal(3, "x", (2, 32, 8, 8, 8)))
    with torch.no_grad():
        want = c3(F.relu(bn2(c2(F.relu(bn1(c1(x)))))))
    p1, p2, p3 = _PackedConv(c1.to(DEV), bn1.to(DEV)), _PackedConv(c2.to(DEV), bn2.to(DEV)), _PackedConv(c3.to(DEV), None)
    out = torch.full((2, 15, 512), 3.0, device=DEV)
    _lib.pointwise_chain3(_ndhwc(x).to(DEV), p1, p2, p3, out, 2, 8)
    assert float((out.cpu().view(2, 15, 8, 8, 8) - want).abs().max()) < 2e-5


@pytest.mark.parametrize("B,G", [(2, 32), (1, 64)])
def test_pointwise_chain_with_softargmax_pass1(B, G):
    """se_pointwise_chain3_softargmax_f32: same logits as the plain chain, bit for bit, and its pass-1 records give the same joints
    and softmaxed volumes through se_softargmax3d_finish_f32 as the two-pass kernel on those logits (online softmax: a different
    summation order, hence 1e-5 relative on the volumes and 2e-6 m on the joints)."""
    torch.manual_seed(G)
    c1, c2, c3 = nn.Conv3d(32, 32, 1), nn.Conv3d(32, 32, 1), nn.Conv3d(32, 15, 1)
    with torch.no_grad():
        c3.weight.mul_(6.0)                       # peaky logits: the running maximum changes often
    p1 = _PackedConv(c1.to(DEV), _rand_bn(32, 1).to(DEV))
    p2 = _PackedConv(c2.to(DEV), _rand_bn(32, 2).to(DEV))
    p3 = _PackedConv(c3.to(DEV), None)
    N = G ** 3
    x = torch.randn(B, G, G, G, 32, device=DEV)
    coord = (torch.rand(N, 3, device=DEV) - 0.5) * 2.0
    ref = torch.empty(B, 15, N, device=DEV)
    _lib.pointwise_chain3(x, p1, p2, p3, ref, B, G)
    vol_ref, j_ref = torch.empty_like(ref), torch.empty(B, 15, 3, device=DEV)
    _lib.softargmax3d(ref, coord, vol_ref, j_ref, B * 15, N, 1)
    out = torch.full_like(ref, float("nan"))
    scratch = torch.full((_lib.softargmax3d_scratch_elems(B * 15),), float("nan"), device=DEV)
    _lib.pointwise_chain3(x, p1, p2, p3, out, B, G, softargmax=(coord, scratch))
    assert torch.equal(out, ref)
    vol, j = torch.empty_like(ref), torch.empty(B, 15, 3, device=DEV)
    _lib.softargmax3d_finish(out, scratch, vol, j, B * 15, N, 1)
    assert float((j - j_ref).abs().max()) < 2e-6
    assert float((vol - vol_ref).abs().max()) <= 1e-5 * float(vol_ref.max())
    assert abs(float(vol.sum()) - B * 15) < 1e-3
    # quad-planar input (SE_IN_QUAD, round 5: back_layers.0 hands the tail whole 16-byte records): the same fragments, the same bits
    out_q = torch.full_like(ref, float("nan"))
    scratch_q = torch.full_like(scratch, float("nan"))
    _lib.pointwise_chain3(_quad(x), p1, p2, p3, out_q, B, G, softargmax=(coord, scratch_q), in_quad=True)
    assert torch.equal(out_q, ref)
    assert torch.equal(torch.nan_to_num(scratch_q, nan=-1.0), torch.nan_to_num(scratch, nan=-1.0))      # (the pad slots of a record keep their NaN fill)


def test_voxelize_strided_into_v2v_buffer(voxel_setup):
    """Occupancy written straight into channel 32 of a [B,G^3,48] buffer: bit-exact, channels 33..35 cleared, others untouched."""
    c, tab = voxel_setup
    _, depth = synth.make_inputs(21, 2, "floor")
    buf = torch.full((2, 64 ** 3, 48), 5.0, device=DEV)
    _lib.voxelize_strided(depth.to(DEV), tab, buf, 2, 1024, 1280, 1024, 128, 64, 2, 48, 32)
    got = buf.cpu()
    for b in range(2):
        want = O.depth_to_voxel(depth[b].numpy(), c.ray, 64, 2).reshape(-1)
        assert torch.equal(got[b, :, 32], want)
    assert float(got[..., 33:36].abs().max()) == 0.0
    assert float(got[..., :32].min()) == 5.0 and float(got[..., 36:].min()) == 5.0


def test_preprocess_image_device_matches_reference_fixture(golden):
    """f1: crop 128 / exact quarter resize / BGR normalisation on the device (se_preprocess_image_u8) against the output of the
    REFERENCE's DemoDataset.__getitem__ + Normalize + ToTensor on the same seeded frames (tests/golden/preprocess.npz, captured by
    tools/make_golden.py --only-preprocess), bit for bit; the half-size depth upload + device clamp + the voxeliser's own nearest
    lookup against the oracle's voxeliser fed with the reference's depth tensor (whose hash tests/test_host_logic.py pins)."""
    from sceneego_amd import preprocess as pp
    g = golden("preprocess")
    frame = lambda seed: (synth.uniform01(seed, "f1/frame", 1024 * 1280 * 3) * 256.0).astype(np.uint8).reshape(1024, 1280, 3)
    frames = np.stack([frame(31), frame(33)])
    got = pp.preprocess_image_device(torch.from_numpy(frames).to(DEV)).cpu().numpy()
    assert got.shape == (2, 3, 256, 256)
    assert np.array_equal(got[0], g["synth_a_image"]) and np.array_equal(got[1], g["synth_b_image"])
    with pytest.raises(ValueError):
        pp.preprocess_image_device(torch.zeros((1, 512, 640, 3), dtype=torch.uint8, device=DEV))
    # half-size depth map (seed 32, [512,640], values up to 12 m): device clamp + the voxeliser's nearest lookup == the oracle's
    # voxeliser on the reference's [1024,1280] depth tensor (nearest resize + clamp done by the reference's code)
    c = O.Constants(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sceneego_amd", "calibration",
                                 "fisheye.calibration_05_08.json"), G=64)
    tab = torch.from_numpy(op.build_voxelizer_ray_table(c.ray, 1280, 1024)).to(DEV)
    d = (synth.uniform01(32, "f1/depth", 512 * 640) * 12.0).astype(np.float32).reshape(1, 512, 640)
    a = _hip_voxelize(torch.from_numpy(d).to(DEV).clamp_(max=10.0), tab)
    ref_depth = pp.prepare_depth(d[0])                                   # == the reference's tensor (hash-pinned on the CPU side)
    assert np.array_equal(ref_depth.numpy()[::16, ::16], g["synth_a_depth_sub"])
    want = O.depth_to_voxel(ref_depth.numpy(), c.ray, 64, 2)
    assert torch.equal(a[0].cpu(), want)


# ------------------------------------------------------------------------------------------------
# triplet-planar float32 V2V input (SE_IN_PLANAR3): producers and the 7^3 layer, bit-exact against the channels-last forms
# ------------------------------------------------------------------------------------------------
def _to_planar3(x_cl, channels):
    """[B,G,G,G,C>=channels] channels-last -> [B,ceil(channels/3),G,G,G,3] (slots beyond `channels` zero)."""
    B, G = x_cl.shape[0], x_cl.shape[1]
    T = (channels + 2) // 3
    p = torch.zeros(B, G, G, G, T * 3, device=x_cl.device)
    p[..., :channels] = x_cl[..., :channels]
    return p.view(B, G, G, G, T, 3).permute(0, 4, 1, 2, 3, 5).contiguous()


def test_planar3_producers_bit_exact(voxel_setup, oracle_constants):
    c, tab = voxel_setup
    feat = torch.from_numpy(synth.normal(31, "feat", (2, 64, 64, 32))).to(DEV)
    idx, w = op.build_gather_table(c.grid, (1024, 1280), 64)
    idx, w = idx.to(DEV), w.to(DEV)
    _, depth = synth.make_inputs(22, 2, "floor")
    N = 64 ** 3
    cl = torch.zeros((2, N, 48), device=DEV)
    _lib.unproject_gather(feat, idx, w, cl, 2, 4096, 32, N, 48, 0)
    _lib.voxelize_strided(depth.to(DEV), tab, cl, 2, 1024, 1280, 1024, 128, 64, 2, 48, 32)
    p3 = torch.full((2, 11, N, 3), 7.0, device=DEV)      # poison: every slot of the 11 triplets must be written
    _lib.unproject_gather_planar3(feat, idx, w, p3, 2, 4096, 32, N, 11)
    assert float(p3[:, 10, :, 2].abs().max()) == 0.0     # the occupancy slot is cleared by the gather
    _lib.voxelize_planar3(depth.to(DEV), tab, p3, 2, 1024, 1280, 1024, 128, 64, 2, 11, 32)
    want = _to_planar3(cl.view(2, 64, 64, 64, 48), 33).view(2, 11, N, 3)
    assert torch.equal(p3, want)
    assert float(p3[:, 10, :, 2].sum()) > 0
    # extra triplets (triplets_total > 11) are left alone
    p4 = torch.full((1, 12, N, 3), 7.0, device=DEV)
    _lib.unproject_gather_planar3(feat[:1], idx, w, p4, 1, 4096, 32, N, 12)
    assert float(p4[:, 11].min()) == 7.0 and torch.equal(p4[:, :10], want[:1, :10])


@pytest.mark.parametrize("B,dim", [(1, 16), (2, 32), (8, 64), (35, 32)])      # B > 32: sliced into launches of 32 samples
def test_conv7_planar3_input_bit_exact(B, dim):
    conv = nn.Conv3d(33, 16, 7, padding=3).to(DEV)
    bn = nn.BatchNorm3d(16).to(DEV).eval()
    with torch.no_grad():
        bn.running_var.uniform_(0.5, 2.0); bn.running_mean.normal_(); bn.weight.normal_(); bn.bias.normal_()
    pc = _PackedConv(conv, bn, cin_pad=48)
    x = torch.zeros(B, dim, dim, dim, 48, device=DEV)
    x[..., :33] = torch.randn(B, dim, dim, dim, 33, device=DEV)
    out_cl = torch.empty((B, dim, dim, dim, 16), device=DEV)
    _lib.conv3d(x, pc.w, pc.b, None, out_cl, B, dim, 33, 48, 16, 7, _lib.EPI_RELU)
    out_p3 = torch.full_like(out_cl, -3.0)
    _lib.conv3d(_to_planar3(x, 33), pc.w, pc.b, None, out_p3, B, dim, 33, 48, 16, 7, _lib.EPI_RELU | _lib.IN_PLANAR3)
    assert torch.equal(out_cl, out_p3)
    if dim <= 32:
        with torch.no_grad():
            want = torch.relu(bn(conv(_ncdhw(x[..., :33]))))
        assert float((_ncdhw(out_p3) - want).abs().max()) < 1e-4


def test_planar3_bad_arguments():
    lib = _lib.load()
    d = torch.zeros(64, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    # the planar input flag is only defined for the 7^3 layer
    assert lib.se_conv3d_f32(p(d), p(d), p(d), None, p(d), 1, 16, 32, 32, 32, 3, _lib.IN_PLANAR3, None, 0, None) == -1
    assert lib.se_unproject_gather_planar3_f32(p(d), p(d), p(d), p(d), 1, 16, 20, 8, 11, None) == -1     # channels not 16/32/64
    assert lib.se_unproject_gather_planar3_f32(p(d), p(d), p(d), p(d), 1, 16, 32, 8, 10, None) == -1     # too few triplets
    assert lib.se_voxelize_planar3_f64(p(d), p(d), p(d), 1, 8, 8, 8, 0, 8, 2.0, 11, 33, None) == -1      # channel out of range


@pytest.mark.parametrize("channels", [16, 64])
def test_gather_planar3_other_channel_counts(channels):
    """The 16- and 64-channel instantiations of the planar gather against the channels-last gather (random tap table, some
    taps disabled): identical bits, trailing slots of the last triplet zeroed."""
    g = torch.Generator().manual_seed(channels)
    V, TEX, B = 1000, 96, 3
    feat = torch.randn(B, TEX, channels, generator=g).to(DEV)
    idx = torch.randint(-1, TEX, (V, 4), generator=g, dtype=torch.int32).to(DEV)
    w = torch.rand(V, 4, generator=g).to(DEV)
    cl = torch.zeros(B, V, channels, device=DEV)
    _lib.unproject_gather(feat, idx, w, cl, B, TEX, channels, V, channels, 0)
    T = (channels + 2) // 3
    p3 = torch.full((B, T, V, 3), 9.0, device=DEV)
    _lib.unproject_gather_planar3(feat, idx, w, p3, B, TEX, channels, V, T)
    flat = p3.permute(0, 2, 1, 3).reshape(B, V, T * 3)
    assert torch.equal(flat[..., :channels], cl)
    assert float(flat[..., channels:].abs().max()) == 0.0


@pytest.mark.parametrize("B,dim", [(1, 16), (2, 32), (33, 16)])
def test_conv3d_fused_skip_convolution(B, dim):
    """se_conv3d_skip16_f32: the 3x3x3 convolution with the block's 1x1x1 skip convolution (16 -> 32 channels) computed in its
    epilogue equals conv3(in) + conv1(x) + biases, ReLU (reference network/v2v.py:40-43); the skip path runs on the MFMA in a
    different summation order than the separate launch, hence a tolerance (2e-5 of the output range)."""
    torch.manual_seed(dim)
    cin, cout = 32, 32
    conv = nn.Conv3d(cin, cout, 3, padding=1).to(DEV)
    skip = nn.Conv3d(16, cout, 1).to(DEV)
    bn, bns = _rand_bn(cout, 3).to(DEV), _rand_bn(cout, 4).to(DEV)
    pc, ps = _PackedConv(conv, bn), _PackedConv(skip, bns)
    a = torch.randn(B, dim, dim, dim, cin, device=DEV)
    x = torch.randn(B, dim, dim, dim, 16, device=DEV)
    s_out = torch.empty(B, dim, dim, dim, cout, device=DEV)
    _lib.conv3d(x, ps.w, ps.b, None, s_out, B, dim, 16, 16, cout, 1, 0)
    want = torch.empty_like(s_out)
    _lib.conv3d(a, pc.w, pc.b, s_out, want, B, dim, cin, cin, cout, 3, _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU)
    scale = (bns.weight / torch.sqrt(bns.running_var + bns.eps)).detach()
    w_skip = (skip.weight.detach().reshape(cout, 16) * scale[:, None]).contiguous()
    a_oct = a.view(B, dim, dim, dim, cin // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous()
    got_oct = torch.empty(B, cout // 8, dim, dim, dim, 8, device=DEV)
    _lib.conv3d_skip16(a_oct, pc.w, (pc.b + ps.b).contiguous(), x, w_skip, got_oct, B, dim, cin, cout,
                       _lib.EPI_RELU | _lib.IN_OCTET | _lib.OUT_OCTET)
    got = got_oct.permute(0, 2, 3, 4, 1, 5).reshape(B, dim, dim, dim, cout)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    with pytest.raises(_lib.HipExtensionError):      # channels-last output is not instantiated for the fused form
        _lib.conv3d_skip16(a_oct, pc.w, pc.b, x, w_skip, want, B, dim, cin, cout, _lib.EPI_RELU | _lib.IN_OCTET)


# (34, 16, 32, 32): more samples than one launch takes (slices of 32), so every per-slice pointer is exercised
@pytest.mark.parametrize("B,dim,cin,cout", [(1, 16, 32, 32), (2, 32, 16, 32), (1, 16, 64, 128), (34, 16, 32, 32)])
def test_conv3d_octet_planar_forms_match_channels_last(B, dim, cin, cout):
    """The 2-D Winograd 3x3x3 kernel reads / writes the octet-planar layout [B][C/8][D][D][D][8] (SE_IN_OCTET / SE_OUT_OCTET, used
    between the two convolutions of a Res3DBlock): same arithmetic in the same order, so results are bit-identical to the
    channels-last call."""
    torch.manual_seed(dim + cin)
    conv = nn.Conv3d(cin, cout, 3, padding=1).to(DEV)
    pc = _PackedConv(conv, _rand_bn(cout, 5).to(DEV))
    assert _lib.conv3d_algo(dim, cin, cout, 3) == 2
    x = torch.randn(B, dim, dim, dim, cin, device=DEV)
    res = torch.randn(B, dim, dim, dim, cout, device=DEV)
    flags = _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU
    # reference of the bit-for-bit comparisons: the octet-planar-output form (always the 2-D Winograd family); the plain channels-last
    # call runs the same kernel - identical bits - unless the batch has <= 4096 voxels (16^3 at batch 1), where it is served by the
    # in-workgroup split-K kernel (se_conv3d_f32_variant == 0): then equal to rounding
    out_oct = torch.empty(B, cout // 8, dim, dim, dim, 8, device=DEV)
    _lib.conv3d(x, pc.w, pc.b, res, out_oct, B, dim, cin, cin, cout, 3, flags | _lib.OUT_OCTET)
    ref = out_oct.permute(0, 2, 3, 4, 1, 5).reshape(B, dim, dim, dim, cout).contiguous()
    plain = torch.empty(B, dim, dim, dim, cout, device=DEV)
    _lib.conv3d(x, pc.w, pc.b, res, plain, B, dim, cin, cin, cout, 3, flags)
    if _lib.conv3d_variant(B, dim, cin, cout, 3) in (2, 3):
        assert torch.equal(plain, ref)
    else:
        assert _lib.conv3d_variant(B, dim, cin, cout, 3) == 0 and B * dim ** 3 <= 4096
        assert float((plain - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    x_oct = x.view(B, dim, dim, dim, cin // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous()
    out = torch.empty_like(ref)
    _lib.conv3d(x_oct, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, flags | _lib.IN_OCTET)
    assert torch.equal(out, ref)
    _lib.conv3d(x_oct, pc.w, pc.b, res, out_oct, B, dim, cin, cin, cout, 3, flags | _lib.IN_OCTET | _lib.OUT_OCTET)
    assert torch.equal(out_oct.permute(0, 2, 3, 4, 1, 5).reshape(B, dim, dim, dim, cout), ref)
    if cin == cout:     # octet-planar skip tensor (SE_RES_OCTET): the block input doubles as skip tensor in Res3DBlock
        res_oct = res.view(B, dim, dim, dim, cout // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous()
        _lib.conv3d(x_oct, pc.w, pc.b, res_oct, out_oct, B, dim, cin, cin, cout, 3, flags | _lib.IN_OCTET | _lib.OUT_OCTET | _lib.RES_OCTET)
        assert torch.equal(out_oct.permute(0, 2, 3, 4, 1, 5).reshape(B, dim, dim, dim, cout), ref)
        _lib.conv3d(x, pc.w, pc.b, res_oct, out, B, dim, cin, cin, cout, 3, flags | _lib.RES_OCTET)
        assert torch.equal(out, ref)
    # max-pool with octet-planar input == max-pool of the channels-last tensor
    p_ref = torch.empty(B, dim // 2, dim // 2, dim // 2, cin, device=DEV)
    p_oct = torch.empty_like(p_ref)
    _lib.maxpool3d_2(x, p_ref, B, dim, cin)
    _lib.maxpool3d_2(x_oct, p_oct, B, dim, cin, in_octet=True)
    assert torch.equal(p_ref, p_oct)
    # fused 2x max-pool of the output (se_conv3d_pool_f32: octet-planar in / out, skip tensor channels-last or octet-planar):
    # the full-resolution output is unchanged and the pooled tensor equals the max-pool of it, bit for bit
    p_fused = torch.full((B, dim // 2, dim // 2, dim // 2, cout), float("nan"), device=DEV)
    _lib.conv3d(x_oct, pc.w, pc.b, res, out_oct, B, dim, cin, cin, cout, 3, flags | _lib.IN_OCTET | _lib.OUT_OCTET, pool_out=p_fused)
    assert torch.equal(out_oct.permute(0, 2, 3, 4, 1, 5).reshape(B, dim, dim, dim, cout), ref)
    p_want = torch.empty_like(p_fused)
    _lib.maxpool3d_2(ref, p_want, B, dim, cout)
    assert torch.equal(p_fused, p_want)
    if cin == cout:
        p_fused.fill_(float("nan"))
        _lib.conv3d(x_oct, pc.w, pc.b, res_oct, out_oct, B, dim, cin, cin, cout, 3,
                    flags | _lib.IN_OCTET | _lib.OUT_OCTET | _lib.RES_OCTET, pool_out=p_fused)
        assert torch.equal(p_fused, p_want)
    with pytest.raises(_lib.HipExtensionError):      # other layouts are not instantiated with the pooled epilogue
        _lib.conv3d(x, pc.w, pc.b, res, out, B, dim, cin, cin, cout, 3, flags, pool_out=p_fused)
    # shapes the 2-D kernel does not take refuse the flags
    small = torch.randn(1, 8, 8, 8, cin, device=DEV)
    with pytest.raises(_lib.HipExtensionError):
        _lib.conv3d(small, pc.w, pc.b, None, torch.empty(1, 8, 8, 8, cout, device=DEV), 1, 8, cin, cin, cout, 3, _lib.IN_OCTET)


@pytest.mark.parametrize("B,dim,cin,cout,residual", [(1, 32, 32, 32, True), (2, 32, 16, 32, False), (1, 32, 64, 64, True)])
def test_conv3d_wino44_experiment_vs_torch(B, dim, cin, cout, residual):
    """Round-3 experiment, development builds only (csrc/build.sh --devtools; se_debug_set_variant(63)): 2-D Winograd F(4,3) x F(4,3)
    for the 3x3x3 layers (csrc/conv3d_wino44.hip; reference network/v2v.py:21-43) against torch-CPU float32, channels-last and
    octet-planar forms.  Skipped when the development library has not been built; the production library does not contain it."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libsceneego_hip_dev.so")
    if not os.path.exists(path):
        pytest.skip("development library not built")
    lib = ctypes.CDLL(path)
    if lib.se_abi_version() != _lib.ABI_VERSION:
        pytest.skip("development library is stale")
    for name, (res_t, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res_t, args
    seed = hash((B, dim, cin, cout, 44)) % 1000
    conv = nn.Conv3d(cin, cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(synth.normal(seed, "w", tuple(conv.weight.shape), (2.0 / (cin * 27)) ** 0.5)))
        conv.bias.copy_(torch.from_numpy(synth.uniform(seed, "cb", (cout,), -0.2, 0.2)))
    x = torch.from_numpy(synth.normal(seed, "x", (B, cin, dim, dim, dim)))
    res = torch.from_numpy(synth.normal(seed, "r", (B, cout, dim, dim, dim))) if residual else None
    with torch.no_grad():
        want = conv(x) + (res if residual else 0)
        want = F.relu(want)
    vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    w = conv.weight.detach().float().contiguous().to(DEV)
    bias = conv.bias.detach().float().contiguous().to(DEV)
    wp = torch.empty(int(lib.se_conv3d_packed_elems(cout, cin, 3, 0)), device=DEV)
    bp = torch.empty((cout + 15) // 16 * 16, device=DEV)
    assert lib.se_conv3d_pack_f32(vp(w), vp(bias), None, None, None, None, 0.0, vp(wp), vp(bp), cout, cin, cin, 3, 0, None) == 0
    xin = _ndhwc(x).to(DEV)
    rin = _ndhwc(res).to(DEV) if residual else None
    flags = _lib.EPI_RELU | (_lib.EPI_RES_PRE_RELU if residual else 0)
    to_oct = lambda t, c: t.view(B, dim, dim, dim, c // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous()
    lib.se_debug_set_variant(63)
    try:
        out = torch.full((B, dim, dim, dim, cout), -7.0, device=DEV)
        assert lib.se_conv3d_f32(vp(xin), vp(wp), vp(bp), vp(rin), vp(out), B, dim, cin, cin, cout, 3, flags, None, 0, None) == 0
        out_o = torch.full((B, cout // 8, dim, dim, dim, 8), -7.0, device=DEV)
        fl = flags | _lib.IN_OCTET | _lib.OUT_OCTET | (_lib.RES_OCTET if residual else 0)
        assert lib.se_conv3d_f32(vp(to_oct(xin, cin)), vp(wp), vp(bp), vp(to_oct(rin, cout)) if residual else None, vp(out_o), B, dim, cin, cin,
                                 cout, 3, fl, None, 0, None) == 0
        torch.cuda.synchronize()
    finally:
        lib.se_debug_set_variant(0)
    got = _ncdhw(out.cpu())
    err = float((got - want).abs().max())
    scale = max(1.0, float(want.abs().max()))
    print(f"F(4,3)xF(4,3) experiment {cin}->{cout} @{dim}^3: max error {err:.2e} = {err / scale:.2e} of max|y|")
    assert err < 2e-5 * scale
    assert torch.equal(out_o.permute(0, 2, 3, 4, 1, 5).reshape(B, dim, dim, dim, cout), out)      # same arithmetic in every layout


# ------------------------------------------------------------------------------------------------
# PRODUCTION shapes against torch CPU: the kernels that own two thirds of a step, at the sizes the step runs them
# ------------------------------------------------------------------------------------------------
def _conv_bn(cin, cout, k, seed):
    conv = nn.Conv3d(cin, cout, k, padding=(k - 1) // 2)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(synth.normal(seed, "w", tuple(conv.weight.shape), (2.0 / (cin * k ** 3)) ** 0.5)))
        conv.bias.copy_(torch.from_numpy(synth.uniform(seed, "cb", (cout,), -0.2, 0.2)))
    return conv, _rand_bn(cout, seed)


def _oct(t):        # [B,D,D,D,C] -> [B,C/8,D,D,D,8]
    B, D, C = t.shape[0], t.shape[1], t.shape[-1]
    return t.view(B, D, D, D, C // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous()


def _unoct(t):      # [B,C/8,D,D,D,8] -> [B,D,D,D,C]
    B, O, D = t.shape[0], t.shape[1], t.shape[2]
    return t.permute(0, 2, 3, 4, 1, 5).reshape(B, D, D, D, O * 8)


def _quad(t):       # [B,D,D,D,C] -> [B,C/4,D,D,D,4]
    B, D, C = t.shape[0], t.shape[1], t.shape[-1]
    return t.view(B, D, D, D, C // 4, 4).permute(0, 4, 1, 2, 3, 5).contiguous()


def _unquad(t):     # [B,C/4,D,D,D,4] -> [B,D,D,D,C]
    B, Q, D = t.shape[0], t.shape[1], t.shape[2]
    return t.permute(0, 2, 3, 4, 1, 5).reshape(B, D, D, D, Q * 4)


# planar hand-over layouts of the two 2-D Winograd kernels: (to planar, from planar, record width, IN / OUT / RES flags)
def _lay(kind):
    if kind == "quad":      # conv3d_k3_wino44pp_kernel (64^3 / 32^3 levels)
        return _quad, _unquad, 4, _lib.IN_QUAD, _lib.OUT_QUAD, _lib.RES_QUAD
    return _oct, _unoct, 8, _lib.IN_OCTET, _lib.OUT_OCTET, _lib.RES_OCTET      # conv3d_k3_wino2d_kernel


def test_conv7_front_layer_64_planar3_vs_torch():
    """front_layers.0 at its production size: Conv3d(33, 16, 7) + BN + ReLU (reference network/v2v.py:8-19,147) on 64^3, B=2, the
    triplet-planar input (SE_IN_PLANAR3) -> conv3d_k7_wino67_kernel<true>.  64 = 10 x 6 + 4: the 11th z tile of F(6,7) is ragged,
    and this is the only place it meets an independent reference.  Bound 2e-5 of max|y| against torch-CPU float32; the measured
    F(6,7) error against a float64 evaluation at 2048 sampled outputs is printed and bounded too (kernel header: mean 7e-6)."""
    B, dim = 2, 64
    conv, bn = _conv_bn(33, 16, 7, 67)
    x = torch.from_numpy(synth.normal(67, "x", (B, 33, dim, dim, dim)))
    with torch.no_grad():
        want = F.relu(bn(conv(x)))
    pc = _PackedConv(conv.to(DEV), bn.to(DEV), cin_pad=48)
    xin = torch.zeros(B, dim, dim, dim, 48, device=DEV)
    xin[..., :33] = _ndhwc(x).to(DEV)
    out = torch.full((B, dim, dim, dim, 16), -77.0, device=DEV)
    _lib.conv3d(_to_planar3(xin, 33), pc.w, pc.b, None, out, B, dim, 33, 48, 16, 7, _lib.EPI_RELU | _lib.IN_PLANAR3)
    got = _ncdhw(out.cpu())
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    err_last = float((got[:, :, 60:] - want[:, :, 60:]).abs().max())           # the ragged tile's four z slabs
    # float64 evaluation at sampled positions (biased towards the ragged tile and the volume faces)
    g = torch.Generator().manual_seed(5)
    n = 2048
    pos = torch.randint(0, dim, (n, 3), generator=g)
    pos[: n // 4, 0] = torch.randint(60, 64, (n // 4,), generator=g)
    bi = torch.randint(0, B, (n,), generator=g)
    xp = F.pad(x.double(), (3, 3, 3, 3, 3, 3))
    ar = torch.arange(7)
    zz = (pos[:, 0, None] + ar)[:, :, None, None]
    yy = (pos[:, 1, None] + ar)[:, None, :, None]
    xx = (pos[:, 2, None] + ar)[:, None, None, :]
    patches = xp[bi[:, None, None, None], :, zz, yy, xx]                         # [n,7,7,7,33]
    cw, cb = conv.weight.detach().cpu().double(), conv.bias.detach().cpu().double()       # (the modules moved to the device above)
    bw, bb, bm, bv = (t.detach().cpu().double() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
    y64 = torch.einsum("nzyxc,zyxco->no", patches, cw.permute(2, 3, 4, 1, 0)) + cb
    y64 = F.relu((y64 - bm) * (bw / torch.sqrt(bv + bn.eps)) + bb)
    got_s = got[bi, :, pos[:, 0], pos[:, 1], pos[:, 2]].double()
    want_s = want[bi, :, pos[:, 0], pos[:, 1], pos[:, 2]].double()
    e_hip, e_cpu = float((got_s - y64).abs().max()), float((want_s - y64).abs().max())
    print(f"F(6,7) 33->16 @64^3 B=2 planar3: max|hip - torch f32| = {err:.2e} ({err / scale:.2e} of max|y| = {scale:.2f}), ragged z tile "
          f"{err_last:.2e}; against float64 at {n} samples: hip {e_hip:.2e} ({e_hip / scale:.2e}), torch-CPU f32 {e_cpu:.2e}")
    assert err < 2e-5 * scale, (err, scale)
    assert e_hip < 1.5e-5 * scale, (e_hip, scale)          # accuracy regression guard of the F(6,7) transform (measured ~4e-6)
    # batch 1 with a workspace (what V2VProgram passes): 352 tiles < 2 x 256 CUs -> every tile's 11 chunks are split 6 + 5 between two
    # workgroups, the second halves' sums go through the workspace and k7_combine_kernel adds them (round 5).  The workspace is
    # NaN-filled: a voxel the second halves did not write would surface as NaN.
    ws = torch.full((dim ** 3 * 16 + 4096,), float("nan"), device=DEV)
    out1 = torch.full((1, dim, dim, dim, 16), -77.0, device=DEV)
    _lib.conv3d(_to_planar3(xin[:1].contiguous(), 33), pc.w, pc.b, None, out1, 1, dim, 33, 48, 16, 7, _lib.EPI_RELU | _lib.IN_PLANAR3, ws)
    got1 = _ncdhw(out1.cpu())
    assert bool(torch.isfinite(got1).all())
    err1 = float((got1 - want[:1]).abs().max())
    print(f"   B=1 with a workspace (chunk halves split over two workgroups): max|hip - torch f32| = {err1:.2e} ({err1 / scale:.2e} of max|y|), "
          f"max|split - unsplit| = {float((got1 - got[:1]).abs().max()):.2e}")
    assert err1 < 2e-5 * scale, (err1, scale)
    assert bool(torch.isnan(ws[dim ** 3 * 16:]).all())      # nothing written behind the B * D^3 * 16 floats the header names


def test_conv7_split_small_volume_channels_last_vs_unsplit():
    """The chunk-half split of conv3d_k7_wino67_kernel on a small channels-last volume (16^3, 32 channels = 11 chunks, the last one with
    two real channels; 6 tiles): with a workspace the launch splits, without one it does not; both against torch."""
    B, dim, cin = 1, 16, 32
    conv, bn = _conv_bn(cin, 16, 7, 77)
    x = torch.from_numpy(synth.normal(77, "x", (B, cin, dim, dim, dim)))
    with torch.no_grad():
        want = F.relu(bn(conv(x)))
    pc = _PackedConv(conv.to(DEV), bn.to(DEV))
    xin = _ndhwc(x).to(DEV).contiguous()
    outs = []
    for ws in (None, torch.full((B * dim ** 3 * 16,), float("nan"), device=DEV)):
        out = torch.full((B, dim, dim, dim, 16), -77.0, device=DEV)
        _lib.conv3d(xin, pc.w, pc.b, None, out, B, dim, cin, cin, 16, 7, _lib.EPI_RELU, ws)
        outs.append(_ncdhw(out.cpu()))
        assert float((outs[-1] - want).abs().max()) < 2e-5 * float(want.abs().max())
    assert float((outs[0] - outs[1]).abs().max()) < 1e-5 * float(want.abs().max())      # the two forms differ by summation order only


def test_conv7_channels_last_16_channels_with_nan_behind_the_tensor():
    """k=7 on a 16-channel channels-last input (cin_pad % 3 == 1, cin == cin_pad): the last 3-channel chunk holds ONE real
    channel; the two slots behind it must read as zeros and never touch the next voxel's channels or the memory behind the tensor
    (NaN there would come out as 0 * NaN)."""
    B, dim, cin = 1, 16, 16
    conv, bn = _conv_bn(cin, 16, 7, 16)
    x = torch.from_numpy(synth.normal(16, "x", (B, cin, dim, dim, dim)))
    with torch.no_grad():
        want = bn(conv(x))           # no ReLU: fmaxf(NaN, 0) = 0 would hide exactly what this test looks for
    pc = _PackedConv(conv.to(DEV), bn.to(DEV))
    n = B * dim ** 3 * cin
    buf = torch.full((n + 64,), float("nan"), device=DEV)             # NaN directly behind the last voxel's record
    buf[:n] = _ndhwc(x).to(DEV).reshape(-1)
    out = torch.full((B, dim, dim, dim, 16), -77.0, device=DEV)
    _lib.conv3d(buf[:n].view(B, dim, dim, dim, cin), pc.w, pc.b, None, out, B, dim, cin, cin, 16, 7, 0)
    got = _ncdhw(out.cpu())
    assert bool(torch.isfinite(got).all())
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    # a NaN in channel 0 of ONE voxel reaches its 7^3 neighbourhood (along z the Winograd transform spreads it over the z tiles it
    # touches) and no other (y, x) column: the chunk {15, pad, pad} of the voxel in front of it reads channel 15 only
    x2 = _ndhwc(x).to(DEV).clone()
    x2[0, 8, 8, 8, 0] = float("nan")
    _lib.conv3d(x2, pc.w, pc.b, None, out, B, dim, cin, cin, 16, 7, 0)
    bad = ~torch.isfinite(out).all(dim=-1)[0]
    assert bool(bad[5:12, 5:12, 5:12].all())
    outside = bad.clone()
    outside[:, 5:12, 5:12] = False
    assert not bool(outside.any())


@pytest.mark.parametrize("kind", ["quad", "oct"])
@pytest.mark.parametrize("cin", [32, 16])
def test_conv3d_k3_64_planar_pool_skip_forms_vs_torch(cin, kind):
    """The dominant kernels at their production size and in their production forms (reference network/v2v.py:21-43, 46-52): Res3DBlock
    convolutions cin -> 32 on 64^3, B=1, planar in / out / skip tensor, the pooled epilogue (se_conv3d_pool_f32) and - for the 16 -> 32
    block - the fused 1x1x1 skip convolution (se_conv3d_skip16_f32), each against torch-CPU float32 Conv3d + BatchNorm3d (+ skip) +
    ReLU (+ max_pool3d), 2e-5 of max|y|.  kind "quad": quad-planar tensors, conv3d_k3_wino44pp_kernel (what the program runs at this
    level since round 5); "oct": octet-planar tensors, conv3d_k3_wino2d_kernel (the 16^3 level's kernel and the A/B reference)."""
    B, dim, cout = 1, 64, 32
    to_pl, from_pl, _, IN, OUT, RES = _lay(kind)
    conv, bn = _conv_bn(cin, cout, 3, 640 + cin)
    x = torch.from_numpy(synth.normal(640 + cin, "x", (B, cin, dim, dim, dim)))
    res = torch.from_numpy(synth.normal(640 + cin, "r", (B, cout, dim, dim, dim)))
    with torch.no_grad():
        lin = bn(conv(x))
        want_plain = F.relu(lin)
        want_res = F.relu(lin + res)
    assert _lib.conv3d_algo(dim, cin, cout, 3) == 2
    assert _lib.conv3d_variant(B, dim, cin, cout, 3, IN | OUT) == (3 if kind == "quad" else 2)
    pc = _PackedConv(conv.to(DEV), bn.to(DEV))
    x_pl, res_cl = to_pl(_ndhwc(x).to(DEV)), _ndhwc(res).to(DEV)
    res_pl = to_pl(res_cl)
    out_pl = torch.full_like(res_pl, -77.0)
    tol = lambda w: 2e-5 * max(1.0, float(w.abs().max()))
    # conv1 of a block: planar in / out, ReLU, no skip tensor
    _lib.conv3d(x_pl, pc.w, pc.b, None, out_pl, B, dim, cin, cin, cout, 3, _lib.EPI_RELU | IN | OUT)
    e1 = float((_ncdhw(from_pl(out_pl).cpu()) - want_plain).abs().max())
    assert e1 < tol(want_plain), e1
    # conv2 of a block: + planar skip tensor, ReLU; then the same with the pooled epilogue
    fl = _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU | IN | OUT | RES
    out_pl.fill_(-77.0)
    _lib.conv3d(x_pl, pc.w, pc.b, res_pl, out_pl, B, dim, cin, cin, cout, 3, fl)
    e2 = float((_ncdhw(from_pl(out_pl).cpu()) - want_res).abs().max())
    assert e2 < tol(want_res), e2
    keep = out_pl.clone()
    pooled = torch.full((B, dim // 2, dim // 2, dim // 2, cout), float("nan"), device=DEV)
    out_pl.fill_(-77.0)
    _lib.conv3d(x_pl, pc.w, pc.b, res_pl, out_pl, B, dim, cin, cin, cout, 3, fl, pool_out=pooled)
    e3 = float((_ncdhw(from_pl(out_pl).cpu()) - want_res).abs().max())
    e4 = float((_ncdhw(pooled.cpu()) - F.max_pool3d(want_res, 2)).abs().max())
    assert e3 < tol(want_res) and e4 < tol(want_res), (e3, e4)
    assert torch.equal(out_pl, keep)                                     # the pooled form leaves the full-resolution output as it was
    # channels-last skip tensor and output with a planar input (a block's second convolution in front of a deconvolution / the tail)
    out_cl = torch.full((B, dim, dim, dim, cout), -77.0, device=DEV)
    _lib.conv3d(x_pl, pc.w, pc.b, res_cl, out_cl, B, dim, cin, cin, cout, 3, _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU | IN)
    assert torch.equal(out_cl, from_pl(keep))                            # same arithmetic in every layout
    # channels-last input (a block's first convolution behind a max-pool or the 7^3 layer), planar output
    out_pl.fill_(-77.0)
    _lib.conv3d(_ndhwc(x).to(DEV), pc.w, pc.b, None, out_pl, B, dim, cin, cin, cout, 3, _lib.EPI_RELU | OUT)
    e5 = float((_ncdhw(from_pl(out_pl).cpu()) - want_plain).abs().max())
    assert e5 < tol(want_plain), e5
    # plain channels-last call
    _lib.conv3d(_ndhwc(x).to(DEV), pc.w, pc.b, res_cl, out_cl, B, dim, cin, cin, cout, 3, _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU)
    e6 = float((_ncdhw(out_cl.cpu()) - want_res).abs().max())
    assert e6 < tol(want_res), e6
    msg = f"k3 {cin}->32 @64^3 {kind}: plain {e1:.2e}, +skip {e2:.2e}, pooled {e3:.2e}/{e4:.2e}, cl-in planar-out {e5:.2e}, channels-last {e6:.2e}"
    if cin == 32:
        # front_layers.1 (Res3DBlock(16, 32)): second convolution 32 -> 32 with the block's 1x1x1 skip convolution 16 -> 32 fused
        skip, bns = _conv_bn(16, cout, 1, 77)
        xs = torch.from_numpy(synth.normal(78, "xs", (B, 16, dim, dim, dim)))
        with torch.no_grad():
            want_s = F.relu(lin + bns(skip(xs)))
        ps = _PackedConv(skip.to(DEV), bns.to(DEV))
        scale = (bns.weight / torch.sqrt(bns.running_var + bns.eps)).detach()
        w_skip = (skip.weight.detach().reshape(cout, 16) * scale[:, None]).contiguous().to(DEV)
        out_pl.fill_(-77.0)
        _lib.conv3d_skip16(x_pl, pc.w, (pc.b + ps.b).contiguous(), _ndhwc(xs).to(DEV), w_skip, out_pl, B, dim, cin, cout,
                           _lib.EPI_RELU | IN | OUT)
        e7 = float((_ncdhw(from_pl(out_pl).cpu()) - want_s).abs().max())
        assert e7 < tol(want_s), e7
        msg += f", fused skip16 {e7:.2e}"
    print(msg)
    # the two planar layouts do not mix in one launch
    with pytest.raises(_lib.HipExtensionError):
        _lib.conv3d(x_pl, pc.w, pc.b, None, out_pl, B, dim, cin, cin, cout, 3, _lib.EPI_RELU | _lib.IN_QUAD | _lib.OUT_OCTET)


def test_conv3d_k3_128_vs_torch():
    """BASELINE configs[4] (128^3 grid): the 3x3x3 32 -> 32 layer at 128^3, B=1, quad-planar forms (conv3d_k3_wino44pp_kernel), against
    torch-CPU float32."""
    B, dim, cin, cout = 1, 128, 32, 32
    conv, bn = _conv_bn(cin, cout, 3, 128)
    x = torch.from_numpy(synth.normal(128, "x", (B, cin, dim, dim, dim)))
    with torch.no_grad():
        want = F.relu(bn(conv(x)) + x)              # Res3DBlock with an identity skip: the block input is the skip tensor
    pc = _PackedConv(conv.to(DEV), bn.to(DEV))
    x_q = _quad(_ndhwc(x).to(DEV))
    out_q = torch.full((B, cout // 4, dim, dim, dim, 4), -77.0, device=DEV)
    assert _lib.conv3d_variant(B, dim, cin, cout, 3, _lib.IN_QUAD) == 3
    _lib.conv3d(x_q, pc.w, pc.b, x_q, out_q, B, dim, cin, cin, cout, 3,
                _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU | _lib.IN_QUAD | _lib.OUT_QUAD | _lib.RES_QUAD)
    err = float((_ncdhw(_unquad(out_q).cpu()) - want).abs().max())
    print(f"k3 32->32 @128^3 quad-planar + skip: max error {err:.2e} of max|y| {float(want.abs().max()):.2f}")
    assert err < 2e-5 * max(1.0, float(want.abs().max())), err


@pytest.mark.parametrize("B,dim,cin,cout", [(3, 64, 32, 32), (10, 32, 64, 64)])
def test_conv3d_k3_wino44pp_unit_walk_vs_torch(B, dim, cin, cout):
    """conv3d_k3_wino44pp_kernel against torch-CPU float32 where a persistent workgroup's unit range CROSSES a sample
    boundary and a cout-block boundary (conv3d_wino44pp.hip `advance`: u.b += 1 / u.cb += 1).  Units are walked x, y, z, sample, cout
    block; 256 workgroups take ceil(units / 256) consecutive units each:
      32 -> 32 @64^3, B=3:  768 units, 3 per workgroup, 256 per sample -> workgroup 85 walks units 255, 256, 257 (two samples);
      64 -> 64 @32^3, B=10: 640 units (2 cout blocks x 320 tiles), 3 per workgroup, 32 tiles per sample -> workgroup 10 crosses a sample
                            boundary (units 30, 31, 32) and workgroup 106 the cout-block boundary (318, 319, 320).
    Quad-planar in / out, with and without a quad-planar skip tensor (reference network/v2v.py:21-43); 2e-5 of max|y|."""
    conv, bn = _conv_bn(cin, cout, 3, 4400 + dim)
    x = torch.from_numpy(synth.normal(4400 + dim, "x", (B, cin, dim, dim, dim)))
    with torch.no_grad():
        lin = bn(conv(x))
        want_plain, want_res = F.relu(lin), F.relu(lin + x)          # identity skip: the block input is the skip tensor
    assert _lib.conv3d_variant(B, dim, cin, cout, 3, _lib.IN_QUAD) == 3
    units = B * (dim // 16) * (dim // 8) ** 2 * (cout // 32)
    per = -(-units // 256)
    tiles_per_sample = (dim // 16) * (dim // 8) ** 2
    assert per > 1 and tiles_per_sample % per != 0, "no workgroup would cross a sample boundary"
    pc = _PackedConv(conv.to(DEV), bn.to(DEV))
    x_q = _quad(_ndhwc(x).to(DEV))
    out_q = torch.full((B, cout // 4, dim, dim, dim, 4), -77.0, device=DEV)
    tol = lambda w: 2e-5 * max(1.0, float(w.abs().max()))
    _lib.conv3d(x_q, pc.w, pc.b, None, out_q, B, dim, cin, cin, cout, 3, _lib.EPI_RELU | _lib.IN_QUAD | _lib.OUT_QUAD)
    e1 = float((_ncdhw(_unquad(out_q).cpu()) - want_plain).abs().max())
    out_q.fill_(-77.0)
    _lib.conv3d(x_q, pc.w, pc.b, x_q, out_q, B, dim, cin, cin, cout, 3,
                _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU | _lib.IN_QUAD | _lib.OUT_QUAD | _lib.RES_QUAD)
    got = _ncdhw(_unquad(out_q).cpu())
    e2 = float((got - want_res).abs().max())
    per_sample = (got - want_res).abs().amax(dim=(1, 2, 3, 4))
    print(f"wino44pp {cin}->{cout} @{dim}^3 B={B} ({units} units, {per} per workgroup): plain {e1:.2e}, +skip {e2:.2e}; per sample "
          + " ".join(f"{float(v):.1e}" for v in per_sample))
    assert e1 < tol(want_plain) and e2 < tol(want_res), (e1, e2)


def test_wino44pp_and_wino67_repeat_launches_bit_identical():
    """the two kernels whose LDS-DMAs are inline assembly with hand-counted `s_waitcnt vmcnt` waits (conv3d_wino44pp.hip,
    conv3d_wino67.hip) - a wait that is one too loose is a data race that shows as run-to-run differences.  40 launches per form on
    the same input must give the same bits (short form of tools/diag/k44p_race_soak.py; the build itself refuses VGPR spills and
    foreign M0 writes in these kernels: csrc/check_codeobj.py)."""
    B, dim = 8, 64
    conv, bn = _conv_bn(32, 32, 3, 91)
    pc = _PackedConv(conv.to(DEV), bn.to(DEV))
    x = torch.randn(B, 8, dim, dim, dim, 4, device=DEV)
    res = torch.randn(B, 8, dim, dim, dim, 4, device=DEV)
    out = torch.empty_like(x)
    assert _lib.conv3d_variant(B, dim, 32, 32, 3, _lib.IN_QUAD) == 3
    for flags, r in ((_lib.EPI_RELU | _lib.IN_QUAD | _lib.OUT_QUAD, None),
                     (_lib.EPI_RELU | _lib.EPI_RES_PRE_RELU | _lib.IN_QUAD | _lib.OUT_QUAD | _lib.RES_QUAD, res)):
        first = None
        for i in range(40):
            out.fill_(float(i))
            _lib.conv3d(x, pc.w, pc.b, r, out, B, dim, 32, 32, 32, 3, flags)
            if first is None:
                first = out.clone()
            else:
                assert torch.equal(out, first), f"wino44pp launch {i} differs (flags {flags})"
    conv7, bn7 = _conv_bn(33, 16, 7, 92)
    pc7 = _PackedConv(conv7.to(DEV), bn7.to(DEV), cin_pad=48)
    x7 = torch.zeros(2, dim, dim, dim, 48, device=DEV)
    x7[..., :33] = torch.randn(2, dim, dim, dim, 33, device=DEV)
    x7p = _to_planar3(x7, 33)
    out7 = torch.empty(2, dim, dim, dim, 16, device=DEV)
    first = None
    for i in range(40):
        out7.fill_(float(i))
        _lib.conv3d(x7p, pc7.w, pc7.b, None, out7, 2, dim, 33, 48, 16, 7, _lib.EPI_RELU | _lib.IN_PLANAR3)
        if first is None:
            first = out7.clone()
        else:
            assert torch.equal(out7, first), f"wino67 launch {i} differs"
    # round 5: the chunk-half split of the same kernel at batch 1 (two workgroups per tile + k7_combine_kernel through the workspace) and the
    # LDS-halo kernel of the 4096-voxel levels (8 waves' partial sums meet in LDS in wave order): both deterministic by construction
    ws = torch.empty(dim ** 3 * 16, device=DEV)
    x7s = x7p[:1].contiguous()
    out7s = torch.empty(1, dim, dim, dim, 16, device=DEV)
    first = None
    for i in range(20):
        out7s.fill_(float(i))
        ws.fill_(float(-i))
        _lib.conv3d(x7s, pc7.w, pc7.b, None, out7s, 1, dim, 33, 48, 16, 7, _lib.EPI_RELU | _lib.IN_PLANAR3, ws)
        if first is None:
            first = out7s.clone()
        else:
            assert torch.equal(out7s, first), f"split wino67 launch {i} differs"
    convh, bnh = _conv_bn(128, 128, 3, 93)
    pch = _PackedConv(convh.to(DEV), bnh.to(DEV))
    xh = torch.randn(8, 8, 8, 8, 128, device=DEV)
    rh = torch.randn(8, 8, 8, 8, 128, device=DEV)
    outh = torch.empty_like(xh)
    first = None
    for i in range(20):
        outh.fill_(float(i))
        _lib.conv3d(xh, pch.w, pch.b, rh, outh, 8, 8, 128, 128, 128, 3, _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU)
        if first is None:
            first = outh.clone()
        else:
            assert torch.equal(outh, first), f"halo64 launch {i} differs"


# ------------------------------------------------------------------------------------------------
# round 6: the 7x7x7 front layer in the frequency domain (csrc/conv3d_fft7.hip) and what feeds / reads it
# ------------------------------------------------------------------------------------------------
def _fft7_pack(conv, bn):
    pc = _PackedConv(conv, bn, cin_pad=48)
    hf = _lib.conv3d_k7_fft_pack(conv.weight.detach().float().contiguous(), bn.weight.detach().float().contiguous(),
                                 bn.running_var.detach().float().contiguous(), bn.eps, 16, 33)
    return pc, hf


def _fft7_unquad(out, B, dim):
    return out.view(B, 4, dim, dim, dim, 4).permute(0, 1, 5, 2, 3, 4).reshape(B, 16, dim, dim, dim)


def test_conv7_fft_front_layer_vs_torch():
    """front_layers.0 (Conv3d(33, 16, 7) + BN + ReLU, reference network/v2v.py:8-18,147) in the frequency domain - se_conv3d_k7_fft_f32:
    24^3 tile DFT, per-frequency MFMA GEMM, inverse - against torch-CPU float32 at the production size (64^3, B = 2: interior and face
    tiles), at 32^3 / 48^3 (every tile touches a face; 48 = 3 tiles) and B = 3 walked in chunks of 1 and 2 samples through a NaN-filled
    workspace; both output layouts.  Bound: the 2e-5 of max|y| of the Winograd kernel it replaces; the error against a float64
    evaluation at sampled outputs is printed and bounded (the orthogonal transform has no conditioning penalty: measured ~2e-7 of max|y|,
    below torch's own float32 summation error)."""
    conv, bn = _conv_bn(33, 16, 7, 67)
    cw, cb = conv.weight.detach().double(), conv.bias.detach().double()
    bw, bb, bm, bv = (t.detach().double() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
    cases = ((2, 64, (2,)), (1, 32, (1,)), (3, 48, (1, 2)))
    pc = hf = None
    for B, dim, chunks in cases:
        x = torch.from_numpy(synth.normal(67, "x%d" % dim, (B, 33, dim, dim, dim)))
        with torch.no_grad():
            want = F.relu(bn(conv(x)))
        if pc is None:
            pc, hf = _fft7_pack(conv.to(DEV), bn.to(DEV))
            conv, bn = conv.cpu(), bn.cpu()
        scale = float(want.abs().max())
        xd = x.to(DEV).contiguous()
        for chunk in chunks:
            ws = torch.full((_lib.conv3d_k7_fft_workspace_elems(chunk, dim, 33),), float("nan"), device=DEV)
            for quad in (False, True):
                out = torch.full((B, 16 * dim ** 3), -77.0, device=DEV)
                _lib.conv3d_k7_fft(xd, hf, pc.b, out, B, dim, 33, 16, _lib.EPI_RELU | (_lib.OUT_QUAD if quad else 0), ws)
                got = (_fft7_unquad(out, B, dim) if quad else out.view(B, dim, dim, dim, 16).permute(0, 4, 1, 2, 3)).cpu()
                assert bool(torch.isfinite(got).all())
                err = float((got - want).abs().max())
                assert err < 2e-5 * scale, (B, dim, chunk, quad, err, scale)
        # float64 evaluation at sampled positions (a quarter of them on the volume faces)
        g = torch.Generator().manual_seed(6)
        n = 1024
        pos = torch.randint(0, dim, (n, 3), generator=g)
        pos[: n // 4, 0] = torch.randint(0, 2, (n // 4,), generator=g) * (dim - 1)
        bi = torch.randint(0, B, (n,), generator=g)
        xp = F.pad(x.double(), (3, 3, 3, 3, 3, 3))
        ar = torch.arange(7)
        patches = xp[bi[:, None, None, None], :, (pos[:, 0, None] + ar)[:, :, None, None], (pos[:, 1, None] + ar)[:, None, :, None],
                     (pos[:, 2, None] + ar)[:, None, None, :]]
        y64 = torch.einsum("nzyxc,zyxco->no", patches, cw.permute(2, 3, 4, 1, 0)) + cb
        y64 = F.relu((y64 - bm) * (bw / torch.sqrt(bv + bn.eps)) + bb)
        e_hip = float((got[bi, :, pos[:, 0], pos[:, 1], pos[:, 2]].double() - y64).abs().max())
        e_cpu = float((want[bi, :, pos[:, 0], pos[:, 1], pos[:, 2]].double() - y64).abs().max())
        print(f"fft7 33->16 @{dim}^3 B={B}: max|hip - torch f32| = {err:.2e} ({err / scale:.2e} of max|y| = {scale:.2f}); against float64 at "
              f"{n} samples: hip {e_hip:.2e} ({e_hip / scale:.2e}), torch-CPU f32 {e_cpu:.2e}")
        assert e_hip < 3e-6 * scale, (e_hip, scale)
    # without ReLU (negative outputs survive), and the shapes the entry point refuses
    out = torch.empty((1, 16 * 32 ** 3), device=DEV)
    x = torch.from_numpy(synth.normal(67, "x32", (1, 33, 32, 32, 32)))
    with torch.no_grad():
        want = bn(conv(x))
    ws = torch.empty((_lib.conv3d_k7_fft_workspace_elems(1, 32, 33),), device=DEV)
    _lib.conv3d_k7_fft(x.to(DEV), hf, pc.b, out, 1, 32, 33, 16, 0, ws)
    got = out.view(1, 32, 32, 32, 16).permute(0, 4, 1, 2, 3).cpu()
    assert float(got.min()) < 0 and float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
    lib = _lib.load()
    assert lib.se_conv3d_k7_fft_packed_elems(65, 16) == -1 and lib.se_conv3d_k7_fft_workspace_elems(1, 24, 33) == -1
    assert lib.se_conv3d_k7_fft_packed_elems(33, 32) == -1
    p = ctypes.c_void_p(out.data_ptr())
    assert lib.se_conv3d_k7_fft_f32(p, p, p, p, 1, 24, 33, 16, 0, p, 1 << 40, None) == -1               # dim % 16
    assert lib.se_conv3d_k7_fft_f32(p, p, p, p, 1, 32, 33, 16, 0, p, 1000, None) == -1                   # workspace below one sample
    assert lib.se_conv3d_k7_fft_f32(p, p, p, p, 1, 32, 33, 16, _lib.EPI_RES_PRE_RELU, p, 1 << 40, None) == -1


def test_conv7_fft_32_input_channels_vs_torch():
    """`with_scene: False` (reference network/voxel_net_depth.py:65-77): V2VModel(32, 15), the front layer has 32 input channels and no
    occupancy plane - fft7_gemm_kernel<32> (K = 64: the 17th k-step meets zero weights and the zeroed LDS padding)."""
    B, dim = 2, 32
    conv, bn = _conv_bn(32, 16, 7, 68)
    x = torch.from_numpy(synth.normal(68, "x", (B, 32, dim, dim, dim)))
    with torch.no_grad():
        want = F.relu(bn(conv(x)))
    conv, bn = conv.to(DEV), bn.to(DEV)
    pc = _PackedConv(conv, bn, cin_pad=32)
    hf = _lib.conv3d_k7_fft_pack(conv.weight.detach().float().contiguous(), bn.weight.detach().float().contiguous(),
                                 bn.running_var.detach().float().contiguous(), bn.eps, 16, 32)
    ws = torch.full((_lib.conv3d_k7_fft_workspace_elems(B, dim, 32),), float("nan"), device=DEV)
    out = torch.full((B, 16 * dim ** 3), -77.0, device=DEV)
    _lib.conv3d_k7_fft(x.to(DEV), hf, pc.b, out, B, dim, 32, 16, _lib.EPI_RELU | _lib.OUT_QUAD, ws)
    got = _fft7_unquad(out, B, dim).cpu()
    err, scale = float((got - want).abs().max()), float(want.abs().max())
    print(f"fft7 32->16 @{dim}^3 B={B}: max|hip - torch f32| = {err:.2e} ({err / scale:.2e} of max|y|)")
    assert bool(torch.isfinite(got).all()) and err < 2e-5 * scale, (err, scale)


def test_conv7_fft_repeat_launches_bit_identical_and_batch_invariant():
    """The three passes have no atomics and no split sums: 20 launches give the same bits, and a sample's result does not depend on the
    batch it rides in or on the workspace chunking (unlike the chunk-half split of the Winograd kernel)."""
    B, dim = 3, 64
    conv, bn = _conv_bn(33, 16, 7, 92)
    pc, hf = _fft7_pack(conv.to(DEV), bn.to(DEV))
    x = torch.randn(B, 33, dim, dim, dim, device=DEV)
    ws = torch.empty((_lib.conv3d_k7_fft_workspace_elems(B, dim, 33),), device=DEV)
    out = torch.empty((B, 16 * dim ** 3), device=DEV)
    first = None
    for i in range(20):
        out.fill_(float(i))
        ws.fill_(float(-i))
        _lib.conv3d_k7_fft(x, hf, pc.b, out, B, dim, 33, 16, _lib.EPI_RELU | _lib.OUT_QUAD, ws)
        if first is None:
            first = out.clone()
        else:
            assert torch.equal(out, first), f"fft7 launch {i} differs"
    one = torch.empty((1, 16 * dim ** 3), device=DEV)
    for b in range(B):
        _lib.conv3d_k7_fft(x[b:b + 1].contiguous(), hf, pc.b, one, 1, dim, 33, 16, _lib.EPI_RELU | _lib.OUT_QUAD, ws)
        assert torch.equal(one[0], first[b]), f"sample {b}: batch-1 launch differs from its batch-{B} result"


def test_planar1_producers_bit_exact(voxel_setup, oracle_constants):
    """se_unproject_gather_planar1_f32 + se_voxelize_planar1_f64 write exactly the bits of the channels-last pair, as planes."""
    c, tab = voxel_setup
    feat = torch.from_numpy(synth.normal(31, "feat", (2, 64, 64, 32))).to(DEV)
    idx, w = op.build_gather_table(c.grid, (1024, 1280), 64)
    idx, w = idx.to(DEV), w.to(DEV)
    _, depth = synth.make_inputs(22, 2, "floor")
    N = 64 ** 3
    cl = torch.zeros((2, N, 48), device=DEV)
    _lib.unproject_gather(feat, idx, w, cl, 2, 4096, 32, N, 48, 0)
    _lib.voxelize_strided(depth.to(DEV), tab, cl, 2, 1024, 1280, 1024, 128, 64, 2, 48, 32)
    p1 = torch.full((2, 33, N), 7.0, device=DEV)          # poison: every plane must be written
    _lib.unproject_gather_planar1(feat, idx, w, p1, 2, 4096, 32, N, 33)
    assert float(p1[:, 32].abs().max()) == 0.0            # the occupancy plane is cleared by the gather
    _lib.voxelize_planar1(depth.to(DEV), tab, p1, 2, 1024, 1280, 1024, 128, 64, 2, 33, 32)
    assert torch.equal(p1, cl[..., :33].permute(0, 2, 1).contiguous())
    assert float(p1[:, 32].sum()) > 0
    lib = _lib.load()
    p = ctypes.c_void_p(p1.data_ptr())
    assert lib.se_unproject_gather_planar1_f32(p, p, p, p, 1, 4096, 32, N, 31, None) == -1             # fewer planes than channels
    assert lib.se_unproject_gather_planar1_f32(p, p, p, p, 1, 4096, 24, N, 33, None) == -1             # channel count not instantiated
    assert lib.se_voxelize_planar1_f64(p, p, p, 1, 1024, 1280, 1024, 128, 64, 2.0, 33, 33, None) == -1


@pytest.mark.parametrize("B,dim", [(1, 64), (8, 32)])      # 32^3 is on the F(4,3) x F(4,3) kernel from 256 units on: batch 8
def test_conv3d_fused_skip_convolution_quad_planar_skip_input(B, dim):
    """se_conv3d_skip16_f32 with SE_RES_QUAD: the 16-channel input of the fused 1x1x1 skip convolution is quad-planar [B][4][D^3][4]
    (what the frequency-domain front layer writes) - same bits as the channels-last form of the same launch."""
    cin = cout = 32
    if _lib.conv3d_variant(B, dim, cin, cout, 3, _lib.IN_QUAD) != 3:
        pytest.skip("shape not on the F(4,3) x F(4,3) kernel at this batch")
    conv, bn = _conv_bn(cin, cout, 3, 75)
    skip, bns = _conv_bn(16, cout, 1, 77)
    pc, ps = _PackedConv(conv.to(DEV), bn.to(DEV)), _PackedConv(skip.to(DEV), bns.to(DEV))
    scale = (bns.weight / torch.sqrt(bns.running_var + bns.eps)).detach()
    w_skip = (skip.weight.detach().reshape(cout, 16) * scale[:, None]).contiguous().to(DEV)
    a = _quad(torch.randn(B, dim, dim, dim, cin, device=DEV))
    xs = torch.randn(B, dim, dim, dim, 16, device=DEV)
    bsum = (pc.b + ps.b).contiguous()
    want = torch.full((B, 8, dim, dim, dim, 4), -77.0, device=DEV)
    got = torch.full_like(want, -78.0)
    _lib.conv3d_skip16(a, pc.w, bsum, xs, w_skip, want, B, dim, cin, cout, _lib.EPI_RELU | _lib.IN_QUAD | _lib.OUT_QUAD)
    _lib.conv3d_skip16(a, pc.w, bsum, _quad(xs), w_skip, got, B, dim, cin, cout, _lib.EPI_RELU | _lib.IN_QUAD | _lib.OUT_QUAD | _lib.RES_QUAD)
    assert torch.equal(got, want)
    # and against torch for the whole fused launch
    with torch.no_grad():
        ref = F.relu(bn.cpu()(conv.cpu()(_ncdhw(_unquad(a).cpu()))) + bns.cpu()(skip.cpu()(_ncdhw(xs.cpu()))))
    err = float((_ncdhw(_unquad(got).cpu()) - ref).abs().max())
    assert err < 2e-5 * float(ref.abs().max()), err


def test_integration_md_bottleneck_stub_runs_as_written():
    """The reference-side binding INTEGRATION.md shows for Bottleneck.forward (raw ctypes on the shared library, no sceneego_amd import in
    the stub itself) is executed as written and compared with the reference module's own forward."""
    import re
    from sceneego_amd import pose_resnet
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    block = next(b for b in re.findall(r"```python\n(.*?)```", text, re.S) if "def bottleneck_forward(" in b)
    head = block[:block.index("def ")]
    stub = block[block.index("def bottleneck_forward("):]
    stub = stub[:stub.index("# (sceneego_amd/pose_resnet.py:FoldedBackbone")]
    ns = {}
    exec(head.replace('ctypes.CDLL("libsceneego_hip.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})'), ns)
    exec(stub, ns)
    torch.manual_seed(2)
    blk = pose_resnet.Bottleneck(256, 64).to(DEV).eval()
    for bn in (blk.bn1, blk.bn2, blk.bn3):
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.7, 1.3); bn.weight.data.uniform_(0.7, 1.3); bn.bias.data.uniform_(-0.2, 0.2)
    x = torch.randn(8, 256, 16, 16, device=DEV)
    with torch.no_grad():
        want = blk(x)
        got = ns["bottleneck_forward"](blk, x)
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
