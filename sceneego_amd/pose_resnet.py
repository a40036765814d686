"""2D backbone: ResNet-50 trunk + three stride-2 transposed-conv stages ("simple baseline" pose net).

Stands in for the reference's ``network/pose_resnet.py`` (``PoseResNet`` ``:135-246``, ``Bottleneck``
``:52-90``, ``get_pose_net`` ``:313-332``).  The parameter tree reproduces the reference's state-dict
key names and shapes exactly so the published checkpoint loads with ``strict=True``;
the code that builds and runs it is this build's own:

  * the trunk is described by a stage table and built in a loop;
  * ``forward`` returns ``(heatmaps, features)`` like the reference (``:225-246``), but the dead
    ``final_layer`` (1x1, 256->16) is only evaluated when ``compute_heatmaps=True`` — the voxel path
    discards the heatmaps (``network/voxel_net_depth.py:235``);
  * ``FoldedBackbone`` is the inference executor used on the GPU: every BatchNorm is folded into the
    preceding convolution once and the residual add + ReLU are the only element-wise launches left.
    Layout is NCHW: on MI355X / MIOpen (ROCm 7.2) the fp32 NCHW kernels measured 2.7 ms per B=8 forward
    against 3.9 ms channels-last (profiles/r01_backbone_variants.txt).  The dense 2D convolutions are
    the one place the north-star assigns to MIOpen rather than to hand-written HIP.
"""
from __future__ import annotations

from collections import OrderedDict

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_MOMENTUM = 0.1

# (planes, blocks, stride of the first block) — ResNet-50
_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))
_EXPANSION = 4
_DECONV_PLANES = (256, 256, 256)


def _bn(c):
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (carries the stride) -> 1x1 (x4), each followed by BN; ReLU(out + shortcut)."""

    expansion = _EXPANSION

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = nn.Conv2d(planes, planes * _EXPANSION, 1, bias=False)
        self.bn3 = _bn(planes * _EXPANSION)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        sc = x if self.downsample is None else self.downsample(x)
        return self.relu(y + sc)


class PoseResNet(nn.Module):
    def __init__(self, num_heatmaps: int = 16, joints_out: int = 15):
        super().__init__()
        self.joints_out = joints_out
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = _bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = 64
        for si, (planes, blocks, stride) in enumerate(_STAGES, start=1):
            mods = []
            for bi in range(blocks):
                s = stride if bi == 0 else 1
                ds = None
                if bi == 0 and (s != 1 or inplanes != planes * _EXPANSION):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * _EXPANSION, 1, stride=s, bias=False),
                                       _bn(planes * _EXPANSION))
                mods.append(Bottleneck(inplanes, planes, s, ds))
                inplanes = planes * _EXPANSION
            setattr(self, f"layer{si}", nn.Sequential(*mods))
        up = []
        for planes in _DECONV_PLANES:  # k4 s2 p1, no bias (reference :198-223)
            up += [nn.ConvTranspose2d(inplanes, planes, 4, stride=2, padding=1, output_padding=0, bias=False),
                   _bn(planes), nn.ReLU(inplace=True)]
            inplanes = planes
        self.deconv_layers = nn.Sequential(*up)
        self.final_layer = nn.Conv2d(inplanes, num_heatmaps, 1)

    def trunk(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return x

    def forward(self, x, return_mid_layer=False, compute_heatmaps=True):
        mid = self.trunk(x)
        features = self.deconv_layers(mid)
        heatmaps = self.final_layer(features)[:, :self.joints_out] if compute_heatmaps else None
        if return_mid_layer:
            return heatmaps, features, mid
        return heatmaps, features


def strip_module_prefix(state_dict):
    """Checkpoints saved from DataParallel carry a ``module.`` prefix (reference ``:282-289,326-330``)."""
    keys = list(state_dict.keys())
    if keys and keys[0].startswith("module"):
        return OrderedDict((k[7:], v) for k, v in state_dict.items())
    return state_dict


def load_state_dict(model, new_state_dict):
    """Merge-then-load, as the reference's helper (``:306-310``): missing keys keep their init values."""
    state = model.state_dict()
    state.update(new_state_dict)
    model.load_state_dict(state)
    return model


def get_pose_net(model_path=None, state_dict=None):
    """Reference ``get_pose_net`` (``:313-332``).  ``get_pose_net(None)`` loads nothing."""
    model = PoseResNet()
    if state_dict is None:
        if model_path is not None:
            model = load_state_dict(model, torch.load(model_path, map_location="cpu"))
    else:
        model = load_state_dict(model, strip_module_prefix(state_dict))
    return model


# ------------------------------------------------------------------------------------------------
# inference executor: BN folded, channels-last, MIOpen convolutions
# ------------------------------------------------------------------------------------------------
def _fold(conv_w, bn, transposed=False):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    if transposed:   # ConvTranspose2d weight is [Cin, Cout, kh, kw]
        w = conv_w * scale.view(1, -1, 1, 1)
    else:
        w = conv_w * scale.view(-1, 1, 1, 1)
    return w.detach().contiguous(), shift.detach().contiguous()


class FoldedBackbone:
    """Folded copy of a ``PoseResNet`` in eval mode; ``__call__(images) -> features [B,256,64,64]``.

    ``dtype`` torch.float32 (parity path) or torch.bfloat16 (BASELINE config 3).
    """

    def __init__(self, net: PoseResNet, dtype=torch.float32, channels_last: bool = False):
        self.dtype = dtype
        self.conv1x1 = os.environ.get("SCENEEGO_CONV1X1", "1") != "0"
        self.conv3x3 = os.environ.get("SCENEEGO_CONV3X3", "1") != "0"
        self.conv3x3_min_wg = int(os.environ.get("SCENEEGO_CONV3X3_MIN_WG", self.CONV3X3_MIN_WORKGROUPS))
        self.conv1x1_min_wg = int(os.environ.get("SCENEEGO_CONV1X1_MIN_WG", self.CONV1X1_MIN_WORKGROUPS))    # A/B knobs of the routing rule
        self.conv1x1_max_cin = int(os.environ.get("SCENEEGO_CONV1X1_MAX_CIN", self.CONV1X1_MAX_CIN))
        self.conv1x1_small_max_wg = int(os.environ.get("SCENEEGO_CONV1X1_SMALL_MAX_WG", self.CONV1X1_SMALL_MAX_WORKGROUPS))
        self.memory_format = torch.channels_last if channels_last else torch.contiguous_format
        cvt = lambda wb: (wb[0].to(dtype).contiguous(memory_format=self.memory_format), wb[1].to(dtype))
        self.stem = cvt(_fold(net.conv1.weight, net.bn1))
        self.blocks = []
        for si in range(1, 5):
            for blk in getattr(net, f"layer{si}"):
                ds = None
                if blk.downsample is not None:
                    ds = cvt(_fold(blk.downsample[0].weight, blk.downsample[1])) + (blk.downsample[0].stride,)
                self.blocks.append((cvt(_fold(blk.conv1.weight, blk.bn1)), cvt(_fold(blk.conv2.weight, blk.bn2)),
                                    cvt(_fold(blk.conv3.weight, blk.bn3)), blk.conv2.stride, ds))
        self.ups = []
        mods = list(net.deconv_layers)
        for i in range(0, len(mods), 3):
            self.ups.append(cvt(_fold(mods[i].weight, mods[i + 1], transposed=True)))

    @torch.no_grad()
    def __call__(self, images):
        if self.memory_format == torch.contiguous_format and images.is_cuda:
            return self._call_fused(images.to(self.dtype))
        x = images.to(self.dtype).contiguous(memory_format=self.memory_format)
        x = F.relu_(F.conv2d(x, self.stem[0], self.stem[1], stride=2, padding=3))
        x = F.max_pool2d(x, 3, stride=2, padding=1)
        for c1, c2, c3, stride, ds in self.blocks:
            y = F.relu_(F.conv2d(x, c1[0], c1[1]))
            y = F.relu_(F.conv2d(y, c2[0], c2[1], stride=stride, padding=1))
            y = F.conv2d(y, c3[0], c3[1])
            sc = x if ds is None else F.conv2d(x, ds[0], ds[1], stride=ds[2])
            x = F.relu_(y.add_(sc))
        for w, b in self.ups:
            x = F.relu_(F.conv_transpose2d(x, w, b, stride=2, padding=1))
        return x

    # float32 1x1 convolutions with stride 1 (conv1, conv3 and layer1's downsample of every Bottleneck: 33 of the 53 convolutions) run
    # on se_conv2d_1x1_f32 - one MFMA GEMM with bias, residual add and ReLU in its epilogue (round 6) - where the shape is covered and
    # large enough to fill the chip; SCENEEGO_CONV1X1=0 keeps MIOpen + se_bias_act_nchw_f32 for all of them (A/B).
    CONV1X1_MIN_WORKGROUPS = 64        # from a quarter of a workgroup per CU on (batch 1 as a graph: 414.4 frames/s with 256, 417.3 / 420.8 / 418.3 with 128 / 64 / 0; tools/ab_conv1x1_b1g.sh)
    CONV1X1_SMALL_MAX_WORKGROUPS = 512  # the small-M form (se_conv2d_1x1_small_f32) for what is left, up to this many of ITS workgroups (batch 1-2)
    CONV1X1_MAX_CIN = 512              # ... and so they do on the long-K layers (1024 / 2048 input channels: 64 serial k steps per workgroup)

    def _plan(self, kind, slot, x, make):
        """Routing decisions are taken once per (layer, input shape, device): ``make()`` -> the packed operands of the HIP kernel, or
        False for MIOpen.  (The decision asks the library for its tile width - a ctypes call per layer and forward would cost batch 1,
        which is bound by the host when launched eagerly, ~3 % .)"""
        plans = self.__dict__.setdefault("_plans", {})
        key = (kind, slot, tuple(x.shape), x.device)
        p = plans.get(key)
        if p is None:
            p = plans[key] = make()
        return p

    def _packed(self, kind, slot, tile, device, make):
        """Packed weights are shared by every plan of a layer that asks for the same tile (plans are per input shape: a caller that walks
        through many batch sizes must not get a copy of the weights for each)."""
        packs = self.__dict__.setdefault("_packs", {})
        key = (kind, slot, tile, device)
        w = packs.get(key)
        if w is None:
            w = packs[key] = make()
        return w

    def _pw_s2(self, x, wb, slot):
        """The stride-2 1x1 downsample convolution + bias: se_conv2d_1x1_s2_f32 when covered (MIOpen's route: a transpose in, a GEMM, a
        transpose out, then the bias pass)."""
        from . import _lib
        stride = wb[2]

        def make():
            B, cin, H, W = x.shape
            cout = wb[0].shape[0]
            ok = (stride in (2, (2, 2)) and self.dtype == torch.float32 and self.conv1x1 and H % 2 == 0 and W % 8 == 0
                  and cin <= 2 * self.conv1x1_max_cin)
            tile = _lib.conv2d_1x1_tile(B, cin, cout, (H // 2) * (W // 2)) if ok else 0
            if tile and ((B * H * W // 4) // 64) * (cout // tile) >= self.conv1x1_min_wg:
                return (self._packed("1x1", slot, tile, x.device, lambda: _lib.conv2d_1x1_pack(wb[0].reshape(cout, cin).float(), tile)),
                        self._packed("bias", slot, 0, x.device, lambda: wb[1].float().contiguous()))
            if (ok and _lib.conv2d_1x1_small_ok(B, cin, cout, (H // 2) * (W // 2))
                    and ((B * H * W // 4) // 64) * (cout // 16) <= self.conv1x1_small_max_wg):
                return (self._packed("1x1", slot, 16, x.device, lambda: _lib.conv2d_1x1_pack(wb[0].reshape(cout, cin).float(), 16)),
                        self._packed("bias", slot, 0, x.device, lambda: wb[1].float().contiguous()))      # the small-M form
            return False

        p = self._plan("pw_s2", slot, x, make)
        if p:
            return _lib.conv2d_1x1_s2(x, p[0], p[1], False)
        return _lib.bias_act_nchw(F.conv2d(x, wb[0], None, stride=stride), wb[1], None, False)

    def _pw(self, x, wb, residual, relu, slot, in_bias=None):
        """1x1 stride-1 convolution + bias (+ residual) (+ ReLU): the fused GEMM when it covers the shape, else MIOpen + epilogue pass.
        ``in_bias``: ``x`` is the raw result of the convolution in front and relu(x + in_bias) is the real input - applied inside
        the fused GEMM, or by an epilogue pass of its own on the MIOpen route."""
        from . import _lib

        def make():
            B, cin, H, W = x.shape
            cout = wb[0].shape[0]
            if self.dtype != torch.float32 or not self.conv1x1:
                return False
            tile = _lib.conv2d_1x1_tile(B, cin, cout, H * W)
            if tile and cin <= self.conv1x1_max_cin and ((B * H * W) // 64) * (cout // tile) >= self.conv1x1_min_wg:
                return (1, self._packed("1x1", slot, tile, x.device, lambda: _lib.conv2d_1x1_pack(wb[0].reshape(cout, cin).float(), tile)),
                        self._packed("bias", slot, 0, x.device, lambda: wb[1].float().contiguous()))
            # batch 1-2: few pixels against megabytes of weights - 64 x 16 tiles, k over four wave groups
            if _lib.conv2d_1x1_small_ok(B, cin, cout, H * W) and ((B * H * W) // 64) * (cout // 16) <= self.conv1x1_small_max_wg:
                return (2, self._packed("1x1", slot, 16, x.device, lambda: _lib.conv2d_1x1_pack(wb[0].reshape(cout, cin).float(), 16)),
                        self._packed("bias", slot, 0, x.device, lambda: wb[1].float().contiguous()))
            return False

        p = self._plan("pw", slot, x, make)
        if p:
            fn = _lib.conv2d_1x1 if p[0] == 1 else _lib.conv2d_1x1_small
            return fn(x, p[1], p[2], residual, relu, in_bias)
        if in_bias is not None:
            x = _lib.bias_act_nchw(x, in_bias, None, True)
        return _lib.bias_act_nchw(F.conv2d(x, wb[0]), wb[1], residual, relu)

    # float32 3x3 convolutions of the Bottlenecks run on se_conv2d_3x3_f32 / _s2_f32, a direct MFMA product: 26 / 28 us at B = 8 for layer3 /
    # layer4 where MIOpen's Winograd assembly kernel takes 53 and its NHWC implicit GEMM with the transposes around it 62; on the wide maps of
    # layer1 / layer2 (one wave group) 28 / 26 us against MIOpen's 30 - level at B = 8 (one stream 1003 -> 1009 frames/s), ahead at batch 1
    # (404.6 -> 414.5 as a graph; tools/ab_conv3x3_wide.sh).  SCENEEGO_CONV3X3=0: MIOpen for all (A/B).
    CONV3X3_MAX_PIXELS = int(os.environ.get("SCENEEGO_CONV3X3_MAX_PIXELS", 4096))   # layer1 .. layer4 for 256 x 256 images
    CONV3X3_S2_MAX_PIXELS = int(os.environ.get("SCENEEGO_CONV3X3_S2_MAX_PIXELS", 1024))   # output pixels of the stride-2 form
    CONV3X3_MIN_WORKGROUPS = 0          # also at batch 1 (64 workgroups): 398.3 -> 401 frames/s as a graph, batch 2: 560 -> 570 (tools/ab_conv3x3_b1.sh)

    def _c3(self, x, wb, stride, slot):
        """conv2 of a Bottleneck WITHOUT its bias (raw sums)."""
        from . import _lib

        def make():
            B, cin, H, W = x.shape
            cout = wb[0].shape[0]
            if self.dtype != torch.float32 or not self.conv3x3:
                return False
            if (stride in (1, (1, 1)) and H * W <= self.CONV3X3_MAX_PIXELS and ((B * H * W) // 64) * (cout // 16) >= self.conv3x3_min_wg):
                tile = _lib.conv2d_3x3_tile(B, cin, cout, H, W)
                if tile:
                    return (1, self._packed("3x3", slot, tile, x.device, lambda: _lib.conv2d_3x3_pack(wb[0].float(), tile)))
            if (stride in (2, (2, 2)) and H % 2 == 0 and W % 2 == 0 and (H * W) // 4 <= self.CONV3X3_S2_MAX_PIXELS
                    and _lib.conv2d_3x3_s2_ok(cin, cout, H // 2, W // 2)):
                return (2, self._packed("3x3", slot, 16, x.device, lambda: _lib.conv2d_3x3_pack(wb[0].float(), 16)))
            return False

        p = self._plan("c3", slot, x, make)
        if p:
            return _lib.conv2d_3x3(x, p[1], None, False) if p[0] == 1 else _lib.conv2d_3x3_s2(x, p[1], None, False)
        return F.conv2d(x, wb[0], None, stride=stride, padding=1)

    def _call_fused(self, images):
        """NCHW on a HIP device (float32, or bfloat16 for config 3): the 1x1 convolutions as fused GEMMs (_pw), the others as MIOpen
        convolutions WITHOUT bias + one fused HIP epilogue (``se_bias_act_nchw_f32`` / ``_bf16``: bias, residual add, ReLU in a
        single pass)."""
        from . import _lib
        ba = _lib.bias_act_nchw
        x = F.conv2d(images.contiguous(), self.stem[0], None, stride=2, padding=3)
        if self.dtype == torch.float32 and x.shape[2] % 2 == 0 and x.shape[3] % 8 == 0:
            x = _lib.bias_relu_maxpool(x, self.stem[1])                    # bn1 + relu + maxpool in one pass over the stem's largest tensor
        else:
            x = F.max_pool2d(ba(x, self.stem[1], None, True), 3, stride=2, padding=1)
        for bi, (c1, c2, c3, stride, ds) in enumerate(self.blocks):
            y = self._pw(x, c1, None, True, (bi, 1))
            y = self._c3(y, c2, stride, (bi, 2))                            # raw: bn2's bias + ReLU ride into conv3's launch (_pw in_bias)
            if ds is None:
                sc = x
            elif ds[2] in (1, (1, 1)):
                sc = self._pw(x, ds, None, False, (bi, 0))
            else:
                sc = self._pw_s2(x, ds, (bi, 0))
            x = self._pw(y, c3, sc, True, (bi, 3), in_bias=c2[1])
        for li, (w, b) in enumerate(self.ups):
            B, _, H, W = x.shape
            if B * H * W <= self.DECONV_GEMM_MAX_POSITIONS and x.dtype == torch.float32:
                x = self._deconv_gemm(li, x, b)
            else:
                x = ba(F.conv_transpose2d(x, w, None, stride=2, padding=1), b, None, True)
        return x

    # ConvTranspose2d(k=4, s=2, p=1) of the pose head (reference network/pose_resnet.py:205-224) as ONE GEMM + an assembly pass:
    # Z[b] = W_all [16 co x ci] @ x[b] [ci x HW] holds, for each of the 16 taps, that tap's contribution at the UN-shifted input
    # position; output row 2j+a / column 2i+b only sees 2 x 2 of the taps (a = 0: kernel rows 1, 3 on input rows j, j-1; a = 1: rows
    # 0, 2 on j+1, j), which ``se_deconv2d_k4s2_assemble_f32`` sums together with the bias and the ReLU.  Round 2 gathered the 16
    # shifted copies of x first (index_select: 67 MB for the 2048-channel layer) and summed four batched GEMMs per parity - 8 launches,
    # ~150 us for the 2048 -> 256 layer at 8 x 8; MIOpen's backward-data igemm takes 300 us there (profiles/r04_backbone_solvers.txt).
    # At 32 x 32 MIOpen's Winograd form (204 us, 84 TFLOP/s) is faster than any GEMM of Z's size and stays.
    DECONV_GEMM_MAX_POSITIONS = 2048

    def _deconv_gemm(self, li, x, bias):
        from . import _lib
        B, C, H, W = x.shape
        cache = self.__dict__.setdefault("_deconv_gemm_cache", {})
        key = (li, x.device)
        if key not in cache:
            w = self.ups[li][0]                                  # [ci, co, 4, 4]
            cache[key] = w.permute(2, 3, 1, 0).reshape(16 * w.shape[1], C).contiguous()      # row = (4 ky + kx) * co + co_idx
        w_all = cache[key]
        co = w_all.shape[0] // 16
        z = torch.matmul(w_all, x.reshape(B, C, H * W))         # [B, 16 co, HW]: one strided-batched rocBLAS GEMM
        return _lib.deconv2d_k4s2_assemble(z, bias, B, co, H, W, relu=True)
