"""V2V-PoseNet 3D encoder-decoder on hand-written HIP kernels.

Stands in for the reference's ``network/v2v.py`` (``V2VModel`` ``:142-181``, ``EncoderDecorder`` ``:70-139``,
``Res3DBlock`` ``:21-43``, ``Basic3DBlock`` ``:8-18``, ``Pool3DBlock`` ``:46-52``, ``Upsample3DBlock`` ``:55-67``).

Two layers:

* The ``nn.Module`` tree below is a *parameter container*: same sub-module names, parameter shapes and
  initialisation as the reference, so ``state_dict()`` has the reference's keys and the
  published checkpoint loads strictly.  The modules own no arithmetic.
* ``V2VProgram`` is what runs: built once from the tree (``V2VModel.compile``), it folds every eval-mode
  BatchNorm3d into its convolution, re-orders the weights into MFMA fragment order on the device
  (``se_conv3d_pack_f32``) and replays a fixed launch list over channels-last ``[B,Z,Y,X,C]`` buffers:
  one kernel per Conv3d+BN(+ReLU)(+residual) / ConvTranspose3d+BN+ReLU(+skip) / max-pool.  There is no
  ATen fallback: without ``libsceneego_hip.so`` or off a HIP device it raises.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib


# A level's skip block is forked onto the side stream when batch x voxels of the level is at most this.  0 = never: measured in
# round 5 (tools/diag/fork_sweep.py, tools/diag/splitk_ab.py; profiles/r05_fork_and_splitk_ab.txt) every fork set is SLOWER than the
# single-stream order on this platform - batch 1: 2.83 -> 3.0-3.2 ms eager, 2.82 -> 3.02 ms as a replayed hipGraph; batch 8: +0.1-0.2 ms -
# a cross-queue dependency costs more than the overlap of the under-filled launches returns.  The capability stays (fork_levels).
FORK_MAX_VOXELS = 0


def _round16(c):
    return (c + 15) // 16 * 16


def _round8(c):
    return (c + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------
# parameter containers (reference key names)
# ------------------------------------------------------------------------------------------------
class Basic3DBlock(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size):
        super().__init__()
        self.block = nn.Sequential(
            nn.Conv3d(in_planes, out_planes, kernel_size=kernel_size, stride=1, padding=(kernel_size - 1) // 2),
            nn.BatchNorm3d(out_planes), nn.ReLU(True))


class Res3DBlock(nn.Module):
    def __init__(self, in_planes, out_planes):
        super().__init__()
        self.res_branch = nn.Sequential(
            nn.Conv3d(in_planes, out_planes, kernel_size=3, stride=1, padding=1), nn.BatchNorm3d(out_planes),
            nn.ReLU(True),
            nn.Conv3d(out_planes, out_planes, kernel_size=3, stride=1, padding=1), nn.BatchNorm3d(out_planes))
        if in_planes == out_planes:
            self.skip_con = nn.Sequential()
        else:
            self.skip_con = nn.Sequential(nn.Conv3d(in_planes, out_planes, kernel_size=1, stride=1, padding=0),
                                          nn.BatchNorm3d(out_planes))


class Pool3DBlock(nn.Module):
    def __init__(self, pool_size):
        super().__init__()
        assert pool_size == 2
        self.pool_size = pool_size


class Upsample3DBlock(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size, stride):
        super().__init__()
        assert kernel_size == 2 and stride == 2
        self.block = nn.Sequential(
            nn.ConvTranspose3d(in_planes, out_planes, kernel_size=2, stride=2, padding=0, output_padding=0),
            nn.BatchNorm3d(out_planes), nn.ReLU(True))


# (name, in, out) of the encoder/decoder pyramid: level k works at G / 2^k
_ENC = ((32, 64), (64, 128), (128, 128), (128, 128), (128, 128))
_SKIP = (32, 64, 128, 128, 128)
_DEC_UP = ((64, 32), (128, 64), (128, 128), (128, 128), (128, 128))  # decoder_upsample1..5 (in, out)
_DEC_RES = (64, 128, 128, 128, 128)                                  # decoder_res1..5


class EncoderDecorder(nn.Module):  # (sic) the reference's spelling is part of the checkpoint key names
    def __init__(self):
        super().__init__()
        for k in range(5):
            setattr(self, f"encoder_pool{k + 1}", Pool3DBlock(2))
            setattr(self, f"encoder_res{k + 1}", Res3DBlock(*_ENC[k]))
            setattr(self, f"skip_res{k + 1}", Res3DBlock(_SKIP[k], _SKIP[k]))
            setattr(self, f"decoder_res{k + 1}", Res3DBlock(_DEC_RES[k], _DEC_RES[k]))
            setattr(self, f"decoder_upsample{k + 1}", Upsample3DBlock(_DEC_UP[k][0], _DEC_UP[k][1], 2, 2))
        self.mid_res = Res3DBlock(128, 128)


class V2VModel(nn.Module):
    def __init__(self, input_channels, output_channels):
        super().__init__()
        self.input_channels = input_channels
        self.output_channels = output_channels
        self.front_layers = nn.Sequential(Basic3DBlock(input_channels, 16, 7), Res3DBlock(16, 32),
                                          Res3DBlock(32, 32), Res3DBlock(32, 32))
        self.encoder_decoder = EncoderDecorder()
        self.back_layers = nn.Sequential(Res3DBlock(32, 32), Basic3DBlock(32, 32, 1), Basic3DBlock(32, 32, 1))
        self.output_layer = nn.Conv3d(32, output_channels, kernel_size=1, stride=1, padding=0)
        self._program = None
        self._initialize_weights()

    def _initialize_weights(self):
        # reference v2v.py:172-181
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)

    # -- execution ---------------------------------------------------------------------------
    def compile(self, dtype=None, output_scale=1.0, split3=False) -> "V2VProgram":
        """(Re)build the HIP launch program from the current parameters (call after loading weights).
        ``dtype``: torch.float32 (default; parity path) or torch.bfloat16 (bf16 storage, float32 accumulation).
        ``output_scale``: ``run(..., scaled=True)`` returns output_scale * logits (the caller's ``volume_multiplier``,
        reference network/voxel_net_depth.py:271, folded into the output layer); ``forward`` / ``run()`` return the plain logits."""
        self._program = V2VProgram(self, dtype or getattr(self, "program_dtype", torch.float32), output_scale, split3)
        return self._program

    @property
    def program(self) -> "V2VProgram":
        if self._program is None:
            self.compile()
        return self._program

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._program = None

    def _apply(self, fn, *a, **k):
        self._program = None  # .to(device) / .float() invalidate packed weights
        return super()._apply(fn, *a, **k)

    def forward(self, x):
        """Reference signature: ``[B,Cin,G,G,G] -> [B,Cout,G,G,G]`` logits (NCDHW in and out)."""
        _lib.require_hip(x)
        B, C, G = x.shape[0], x.shape[1], x.shape[2]
        prog = self.program
        if prog.fft7_ready(G) and C == prog.cin and G % 32 == 0:
            # NCDHW is the planar input of the frequency-domain front layer as it stands
            return prog.run(x.float().contiguous(), B, G, planar1=True).view(B, self.output_channels, G, G, G)
        buf = torch.zeros((B, G, G, G, prog.cin_pad), device=x.device, dtype=prog.dtype)
        buf[..., :C] = x.permute(0, 2, 3, 4, 1)
        if prog.dtype == torch.bfloat16:
            buf = channels_last_to_octet_planar(buf)
        logits = prog.run(buf, B, G)
        return logits.view(B, self.output_channels, G, G, G)


def channels_last_to_octet_planar(x):
    """[B,G,G,G,C] -> [B,C/8,G,G,G,8]: the input layout of the bf16 7^3 front layer (include/sceneego_hip.h)."""
    B, G, C = x.shape[0], x.shape[1], x.shape[-1]
    return x.view(B, G, G, G, C // 8, 8).permute(0, 4, 1, 2, 3, 5).contiguous()


# ------------------------------------------------------------------------------------------------
# the launch program
# ------------------------------------------------------------------------------------------------
class _PackedConv:
    __slots__ = ("w", "b", "cin", "cin_pad", "cout", "k", "transposed", "fused", "w_split")

    def __init__(self, conv, bn, cin_pad=None, dtype=torch.float32, scale=1.0, split3=False):
        """``scale``: the packed layer computes scale * conv(x) (weights and bias multiplied before packing).
        ``split3`` (float32 3x3x3 layers, experimental): also pack the two bfloat16 halves of the BatchNorm-folded weights for
        se_conv3d_k3_split3_f32."""
        transposed = isinstance(conv, nn.ConvTranspose3d)
        w = conv.weight.detach().float().contiguous()
        if scale != 1.0:
            assert bn is None
            w = w * float(scale)
        dev = w.device
        if transposed:
            cin, cout = w.shape[0], w.shape[1]
            k = 2
        else:
            cout, cin = w.shape[0], w.shape[1]
            k = w.shape[2]
        bf16 = dtype == torch.bfloat16
        self.cin_pad = cin_pad if cin_pad is not None else (_round8(cin) if bf16 else _round16(cin))
        self.cin, self.cout, self.k, self.transposed = cin, cout, k, transposed
        self.fused = None     # 16-channel skip convolution only: (folded weights [cout][16], summed bias) for se_conv3d_skip16_f32
        n = _lib.conv3d_packed_elems(cout, self.cin_pad, k, transposed, bf16=bf16)
        self.w = torch.empty(n, device=dev, dtype=dtype)
        self.b = torch.empty(_round16(cout), device=dev, dtype=torch.float32)
        f = lambda t: None if t is None else t.detach().float().contiguous()
        if bn is not None:
            assert not bn.training, "V2VProgram folds BatchNorm3d running statistics: call .eval() first"
            g, be, mu, var, eps = f(bn.weight), f(bn.bias), f(bn.running_mean), f(bn.running_var), bn.eps
        else:
            g = be = mu = var = None
            eps = 0.0
        bias = f(conv.bias)
        if scale != 1.0 and bias is not None:
            bias = bias * float(scale)
        _lib.conv3d_pack(w, bias, g, be, mu, var, eps, self.w, self.b, cout, cin, self.cin_pad, k, transposed)
        self.w_split = None
        if split3 and not bf16 and not transposed and k == 3 and cout % 32 == 0 and self.cin_pad % 8 == 0:
            wf = w if g is None else w * (g / torch.sqrt(var + eps)).view(-1, 1, 1, 1, 1)      # the folding se_conv3d_pack_f32 applies
            self.w_split = _lib.conv3d_split3_pack(wf.contiguous(), cout, cin, self.cin_pad)

class V2VProgram:
    def __init__(self, model: V2VModel, dtype=torch.float32, output_scale=1.0, split3=False):
        assert dtype in (torch.float32, torch.bfloat16)
        self.dtype = dtype
        self.output_scale = float(output_scale)
        # EXPERIMENTAL: float32 tensors, 3x3x3 layers of the 64^3 / 32^3 / 16^3 levels on split-bf16 arithmetic (csrc/conv3d_split.hip)
        self.split3 = bool(split3) and dtype == torch.float32
        p = next(model.parameters())
        if not p.is_cuda:
            raise _lib.HipExtensionError("V2VModel must live on a HIP device to be compiled (got %s)" % p.device)
        _lib.load()
        self.device = p.device
        self.cout = model.output_channels
        self.cin = model.input_channels
        self.cin_pad = _round8(self.cin) if dtype == torch.bfloat16 else _round16(self.cin)
        fl, ed, bl = model.front_layers, model.encoder_decoder, model.back_layers
        basic = lambda m, cin_pad=None: _PackedConv(m.block[0], m.block[1], cin_pad, dtype)
        self.front0 = basic(fl[0], self.cin_pad)
        # frequency-domain form of the 7^3 front layer (csrc/conv3d_fft7.hip, round 6): the weight spectra in MFMA fragment order.
        # Used when run() gets the planar input [B, cin, G, G, G]; SCENEEGO_FFT7=0 keeps the F(6,7) Winograd kernel (A/B).
        self.front0_fft = None
        self._fft_ws = None
        conv0, bn0 = fl[0].block[0], fl[0].block[1]
        if (dtype == torch.float32 and os.environ.get("SCENEEGO_FFT7", "1") != "0"
                and int(_lib.load().se_conv3d_k7_fft_packed_elems(self.cin, conv0.out_channels)) > 0):
            self.front0_fft = _lib.conv3d_k7_fft_pack(conv0.weight.detach().float().contiguous(), bn0.weight.detach().float().contiguous(),
                                                      bn0.running_var.detach().float().contiguous(), bn0.eps, conv0.out_channels, self.cin)
        self.front_res = [self._pack_res(fl[i]) for i in (1, 2, 3)]
        self.enc = [self._pack_res(getattr(ed, f"encoder_res{k}")) for k in range(1, 6)]
        self.skip = [self._pack_res(getattr(ed, f"skip_res{k}")) for k in range(1, 6)]
        self.dec = [self._pack_res(getattr(ed, f"decoder_res{k}")) for k in range(1, 6)]
        self.up = [basic(getattr(ed, f"decoder_upsample{k}")) for k in range(1, 6)]
        self.mid = self._pack_res(ed.mid_res)
        self.back_res = self._pack_res(bl[0])
        self.back1 = basic(bl[1])
        self.back2 = basic(bl[2])
        self.out = _PackedConv(model.output_layer, None, None, dtype)
        self.out_scaled = self.out if self.output_scale == 1.0 else _PackedConv(model.output_layer, None, None, dtype, scale=self.output_scale)
        # scratch for the split-K path of the small pyramid levels (se_conv3d_f32 workspace): 32 Mi floats.  A workspace serves ONE
        # stream at a time: the forked skip branches (run(), off by default) have their own.
        self.workspace = torch.empty(32 << 20, device=self.device, dtype=torch.float32) if dtype == torch.float32 else None
        self.workspace_side = None
        self._side_stream = None
        self._ws = self.workspace              # the workspace of the stream run() is issuing on
        # skip_res{k+1} of the levels in fork_levels run on a side stream beside the encoder chain (reference network/v2v.py:104-119:
        # skip_x_k = skip_res_k(x) is not read before decoder_upsample_k).  None = decide per batch in run().
        self.fork_levels = None

    def _pack_res(self, m):
        c1 = _PackedConv(m.res_branch[0], m.res_branch[1], None, self.dtype, split3=self.split3)
        c2 = _PackedConv(m.res_branch[3], m.res_branch[4], None, self.dtype, split3=self.split3)
        sk = _PackedConv(m.skip_con[0], m.skip_con[1], None, self.dtype) if len(m.skip_con) else None
        if sk is not None and sk.cin == 16 and self.dtype == torch.float32:
            # 16-channel skip convolution (front_layers.1): folded weights [cout][16] + summed bias for se_conv3d_skip16_f32, which
            # computes the skip path inside the second 3x3x3 convolution's launch
            conv, bn = m.skip_con[0], m.skip_con[1]
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
            sk.fused = ((conv.weight.detach().float().reshape(sk.cout, 16) * scale[:, None]).contiguous(), (c2.b + sk.b).contiguous())
        return (c1, c2, sk)

    # -- primitive launches ------------------------------------------------------------------
    def _new(self, B, dim, c):
        return torch.empty((B, dim, dim, dim, c), device=self.device, dtype=self.dtype)

    def _conv(self, x, pc, B, dim, flags, residual=None, out=None, pool_out=None):
        if out is None:
            out = self._new(B, dim, pc.cout)
        if self.split3 and pc.w_split is not None and dim % 16 == 0 and pool_out is None:
            _lib.conv3d_k3_split3(x, pc.w_split, pc.b, residual, out, B, dim, pc.cin_pad, pc.cout, flags)
            return out
        _lib.conv3d(x, pc.w, pc.b, residual, out, B, dim, pc.cin, pc.cin_pad, pc.cout, pc.k, flags, self._ws, pool_out=pool_out)
        return out

    _LAY = {"quad": (_lib.IN_QUAD, _lib.OUT_QUAD, _lib.RES_QUAD), "oct": (_lib.IN_OCTET, _lib.OUT_OCTET, _lib.RES_OCTET)}

    def _res(self, x, blk, B, dim, x_lay=None, out_planar=False, pool_out=None):
        """Res3DBlock (v2v.py:40-43): relu(bn(conv(relu(bn(conv(x))))) + skip(x)).

        Layouts (float32 program, blocks whose two 3x3x3 convolutions run on a 2-D Winograd kernel): the tensor between the two
        convolutions is always in the block's planar layout (``_planar``: quad-planar [B][C/4][D][D][D][4] on the F(4,3) x F(4,3) kernel,
        octet-planar [B][C/8][D][D][D][8] on the F(4,3) x F(2,3) one); ``x_lay`` names the layout of the block input (None =
        channels-last), ``out_planar`` asks for the block output in the block's planar layout (see run() for who reads what)."""
        c1, c2, sk = blk
        kind = self._planar(blk, dim, B)    # both convolutions take the planar / pooled / fused-skip forms of this kernel family
        assert kind or not (x_lay or out_planar)
        assert x_lay in (None, kind)
        fused = sk.fused if sk is not None else None
        fuse = fused is not None and kind and not self.split3 and out_planar and pool_out is None and (not x_lay or x_lay == "quad")
        # a 1x1x1 skip convolution reads channels-last - except the fused 16-channel one of the quad family, which also takes the
        # quad-planar tensor the frequency-domain front layer writes (SE_RES_QUAD of se_conv3d_skip16_f32)
        assert sk is None or not x_lay or fuse
        IN, OUT, RES = self._LAY[kind] if kind else (0, 0, 0)
        a = self._conv(x, c1, B, dim, _lib.EPI_RELU | OUT | (IN if x_lay else 0))
        if fuse:
            out = torch.empty((B, dim, dim, dim, c2.cout), device=self.device, dtype=self.dtype)
            _lib.conv3d_skip16(a, c2.w, fused[1], x, fused[0], out, B, dim, c2.cin, c2.cout,
                               _lib.EPI_RELU | IN | OUT | (RES if x_lay == "quad" else 0))
            return out
        s = x if sk is None else self._conv(x, sk, B, dim, 0)
        f2 = _lib.EPI_RELU | _lib.EPI_RES_PRE_RELU | IN
        if x_lay and sk is None:
            f2 |= RES
        if out_planar:
            f2 |= OUT
        return self._conv(a, c2, B, dim, f2, residual=s, pool_out=pool_out)      # pool_out: the block's 2x max-pool, written by the same launch

    def _planar(self, blk, dim, B):
        """Planar hand-over layout of a block at this (batch, level): "quad" when both 3x3x3 convolutions run on the F(4,3) x F(4,3)
        kernel (variant 3 with quad flags: the 64^3 / 32^3 levels), "oct" when they run on the F(4,3) x F(2,3) kernel (the 16^3
        level) or on the split-bf16 kernel, None for a level with so few voxels that the plain split-K kernel is faster (16^3 at batch
        1: it stays channels-last) and for the levels below."""
        c1, c2, _ = blk
        if self.dtype != torch.float32:
            return None
        if self.split3:
            return "oct" if c1.w_split is not None and c2.w_split is not None and dim % 16 == 0 else None
        var = lambda c, fl: _lib.conv3d_variant(B, dim, c.cin_pad, c.cout, 3, fl)
        if var(c1, _lib.IN_QUAD) == 3 and var(c2, _lib.IN_QUAD) == 3:
            return "quad"
        if var(c1, 0) in (2, 3) and var(c2, 0) in (2, 3):
            return "oct"
        return None

    def _fork_set(self, B, G):
        """Levels (0 = G^3 ... 4 = (G/16)^3) whose skip block runs on the side stream: ``fork_levels`` if set, else the levels
        with at most FORK_MAX_VOXELS voxels in the batch (default 0 = none: measured slower, see the constant)."""
        if self.dtype != torch.float32 or self.split3:
            return frozenset()
        if self.fork_levels is not None:
            return frozenset(self.fork_levels)
        env = os.environ.get("SCENEEGO_FORK_LEVELS")          # experiments (tools/diag/fork_sweep.py): "" = none, "0,1,2,3,4" = all
        if env is not None:
            return frozenset(int(t) for t in env.split(",") if t.strip())
        return frozenset(k for k in range(5) if B * (G >> k) ** 3 <= FORK_MAX_VOXELS)

    def _side(self, main):
        if self._side_stream is None:      # first use of the (off by default) fork: outside any graph capture thanks to the warm-up forwards
            self._side_stream = torch.cuda.Stream(device=self.device)
            self.workspace_side = torch.empty_like(self.workspace)
        return self._side_stream

    def _pool(self, x, B, dim, c, x_oct=False):
        out = self._new(B, dim // 2, c)
        _lib.maxpool3d_2(x, out, B, dim, c, in_octet=x_oct)
        return out

    def _up(self, x, pc, skip, B, dim, out_quad=False, res_quad=False):
        """Upsample3DBlock + decoder add (v2v.py:64-67,124-137): relu(bn(convT(x))) + skip.  ``out_quad``: the output is written
        quad-planar [B][C/4][2D][2D][2D][4] for a block behind it whose convolutions run on the F(4,3) x F(4,3) kernel;
        ``res_quad`` (with it): ``skip`` is quad-planar too (the skip block wrote whole records)."""
        out = self._new(B, dim * 2, pc.cout)
        _lib.deconv3d_k2s2(x, pc.w, pc.b, skip, out, B, dim, pc.cin_pad, pc.cout,
                           _lib.EPI_RELU | _lib.EPI_RES_POST_RELU | (_lib.OUT_QUAD if out_quad else 0) | (_lib.RES_QUAD if res_quad else 0))
        return out

    @staticmethod
    def _up_quad_ok(pc, dim):
        """Shapes whose transposed convolution has the quad-planar output form (se_deconv3d_k2s2_f32 with SE_OUT_QUAD)."""
        return dim % 16 == 0 and (pc.cin_pad, pc.cout) in ((64, 32), (128, 64))

    # -- the network -------------------------------------------------------------------------
    def fft7_ready(self, G):
        """True when run(..., planar1=True) can take the planar input [B, cin, G, G, G] (the frequency-domain front layer covers it)."""
        return self.front0_fft is not None and G >= 16 and G % 16 == 0

    def _front0_fft(self, x, B, G, out_quad):
        """front_layers.0 in the frequency domain (se_conv3d_k7_fft_f32): planar x [B,cin,G,G,G] -> 16 channels, channels-last or quad-planar.
        The spectra of up to 8 samples live in a workspace owned by the program (1.5 GB at 64^3; larger batches walk it in chunks)."""
        need = _lib.conv3d_k7_fft_workspace_elems(min(B, 8), G, self.cin)
        if self._fft_ws is None or self._fft_ws.numel() < need:
            self._fft_ws = torch.empty(need, device=self.device, dtype=torch.float32)
        out = self._new(B, G, self.front0.cout)
        _lib.conv3d_k7_fft(x, self.front0_fft, self.front0.b, out, B, G, self.cin, self.front0.cout,
                           _lib.EPI_RELU | (_lib.OUT_QUAD if out_quad else 0), self._fft_ws)
        return out

    def run(self, x, B, G, out=None, softargmax=None, scaled=False, planar1=False):
        """x: [B,G,G,G,cin_pad] channels-last (channels >= cin zero; bf16: octet-planar [B,cin_pad/8,G,G,G,8]; float32 may
        also be triplet-planar [B,ceil(cin/3),G,G,G,3], which the 7^3 Winograd front layer reads with ~5x fewer cache-line requests,
        or - ``planar1`` - fully planar [B,cin,G,G,G] for the frequency-domain front layer, see fft7_ready())
        -> planar logits [B,cout,G^3] (``scaled``: times ``output_scale``)."""
        assert x.is_contiguous() and x.dtype == self.dtype
        outc = self.out_scaled if scaled else self.out
        planar3 = self.dtype == torch.float32 and x.dim() == 6       # float32 triplet-planar [B,ceil(cin/3),G,G,G,3]
        if planar1:
            assert self.fft7_ready(G) and tuple(x.shape) == (B, self.cin, G, G, G)
        elif planar3:
            assert tuple(x.shape) == (B, (self.cin + 2) // 3, G, G, G, 3)
        else:
            assert tuple(x.shape) == ((B, self.cin_pad // 8, G, G, G, 8) if self.dtype == torch.bfloat16 else (B, G, G, G, self.cin_pad))
        if G % 32:
            raise ValueError("volume_size must be a multiple of 32 (five 2x max-pools), got %d" % G)
        x_lay = None
        if planar1:
            # quad-planar hand-over when front_layers.1 takes it: both its 3^3 convolutions on the F(4,3) x F(4,3) kernel and the fused
            # 16-channel skip convolution (which then reads the quad-planar tensor too)
            blk0 = self.front_res[0]
            q = (not self.split3 and self._planar(blk0, G, B) == "quad" and blk0[2] is not None and blk0[2].fused is not None
                 and len(self.front_res) > 1)
            x = self._front0_fft(x, B, G, q)
            x_lay = "quad" if q else None
        else:
            x = self._conv(x, self.front0, B, G, _lib.EPI_RELU | (_lib.IN_PLANAR3 if planar3 else 0))
        # Tensor layouts of the float32 program: a Res3DBlock output that is read only by 2-D Winograd convolutions of the same kernel
        # family (as input or as skip tensor) and by a max-pool is kept in that family's planar layout (_planar: quad-planar at 64^3 /
        # 32^3, octet-planar at 16^3); what the deconvolutions, the 1x1x1 convolutions and the fused tail read stays channels-last.
        # x_lay tracks the layout of the running tensor.
        # A block whose output goes to an encoder max-pool writes the pooled tensor from its last convolution's epilogue when
        # that convolution runs on a 2-D Winograd kernel (`pooled`); the pool kernel is then not launched.
        pooled = None
        for i, blk in enumerate(self.front_res):
            kind = self._planar(blk, G, B)
            if kind and not self.split3 and i == len(self.front_res) - 1:      # (the split-bf16 kernel has no pooled form)
                pooled = self._new(B, G // 2, blk[1].cout)
            x = self._res(x, blk, B, G, x_lay=x_lay, out_planar=bool(kind), pool_out=pooled)
            x_lay = kind
        # encoder (v2v.py:104-119).  skip_res_k(x) is not read before decoder_upsample_k: the skip blocks of the levels in `fork` are
        # issued on a side stream (event fork behind the producer of x, event join in front of the deconvolution that reads the
        # result) and run beside the encoder / middle / decoder chain, which at small batches leaves most of the chip idle
        # (16^3 at batch 1: 32 work units on 256 CUs).  Inside a hipGraph capture the fork and the joins become graph edges.
        # Which transposed convolutions write (and read their skip tensor) quad-planar: those in front of a block whose convolutions
        # run on the F(4,3) x F(4,3) kernel (decoder_res1 at 32^3, back_layers.0 at 64^3); the skip block of that level then writes its
        # output quad-planar as well - whole 16-byte records instead of 64 of every 128 bytes of a channels-last record.
        up_quad = [False] * 5
        for k in range(5):
            nxt = self.dec[k - 1] if k > 0 else self.back_res
            up_quad[k] = (not self.split3 and self._planar(nxt, G >> k, B) == "quad" and self._up_quad_ok(self.up[k], G >> (k + 1))
                          and self._planar(self.skip[k], G >> k, B) == "quad")
        skips = []
        joins = [None] * 5
        fork = self._fork_set(B, G)
        main = torch.cuda.current_stream(self.device) if fork else None
        dim = G
        for k in range(5):
            if k in fork:
                side = self._side(main)
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                x.record_stream(side)
                with torch.cuda.stream(side):
                    self._ws = self.workspace_side
                    try:
                        sk = self._res(x, self.skip[k], B, dim, x_lay=x_lay, out_planar=up_quad[k])
                    finally:
                        self._ws = self.workspace
                    joins[k] = torch.cuda.Event()
                    joins[k].record(side)
                sk.record_stream(main)
                skips.append(sk)
            else:
                skips.append(self._res(x, self.skip[k], B, dim, x_lay=x_lay, out_planar=up_quad[k]))    # read by the decoder's deconvolution
            if pooled is None and x_lay == "quad":
                raise RuntimeError("a quad-planar block output is always pooled by its producer")
            x = pooled if pooled is not None else self._pool(x, B, dim, x.numel() // (B * dim ** 3), x_oct=x_lay == "oct")
            pooled = None
            dim //= 2
            kind = self._planar(self.enc[k], dim, B)
            if kind and not self.split3 and k < 4:
                pooled = self._new(B, dim // 2, self.enc[k][1].cout)
            x = self._res(x, self.enc[k], B, dim, x_lay=None, out_planar=bool(kind), pool_out=pooled)
            x_lay = kind
        if x_lay:    # cannot happen: the deepest levels are too small for the 2-D kernels
            raise RuntimeError("planar tensor reached the middle block")
        x = self._res(x, self.mid, B, dim)
        # decoder (v2v.py:121-137)
        # A decoder / back block whose convolutions run on the F(4,3) x F(4,3) kernel gets its input quad-planar straight from the
        # transposed convolution in front of it (round 5; rounds 2-4: channels-last, which kept the block's first convolution on the
        # F(4,3) x F(2,3) kernel - 0.41 ms against 0.32 for back_layers.0's).
        x_lay = None
        for k in range(4, -1, -1):
            x = self._res(x, self.dec[k], B, dim, x_lay=x_lay)
            if joins[k] is not None:
                main.wait_event(joins[k])
            quad = up_quad[k]
            x = self._up(x, self.up[k], skips[k], B, dim, out_quad=quad, res_quad=quad)
            x_lay = "quad" if quad else None
            skips[k] = None
            dim *= 2
        # back layers + output (v2v.py:155-161)
        # the fused tail with the soft-argmax pass reads a quad-planar tensor as well: back_layers.0 then writes whole records
        sa = softargmax            # (coord, scratch): pass 1 of the soft-argmax rides in the tail launch (float32 since round 4, bf16 since round 6)
        tail_quad = (self.dtype == torch.float32 and self.cout <= 16 and sa is not None and not self.split3
                     and self._planar(self.back_res, G, B) == "quad")
        x = self._res(x, self.back_res, B, G, x_lay=x_lay, out_planar=tail_quad)
        if out is None:
            out = torch.empty((B, self.cout, G * G * G), device=self.device, dtype=torch.float32)
        if self.cout <= 16:
            # back_layers.1 / .2 / output_layer fused: one read of x, one planar write of the logits
            # ``softargmax`` = (coord, scratch): float32 program only - pass 1 of the soft-argmax rides in the same launch
            _lib.pointwise_chain3(x, self.back1, self.back2, outc, out, B, G, softargmax=sa, in_quad=tail_quad)
            return out
        x = self._conv(x, self.back1, B, G, _lib.EPI_RELU)
        x = self._conv(x, self.back2, B, G, _lib.EPI_RELU)
        _lib.conv3d(x, outc.w, outc.b, None, out, B, G, outc.cin, outc.cin_pad, outc.cout, 1, _lib.EPI_OUT_PLANAR)
        return out
