"""``VoxelNetwork_depth`` — drop-in for the reference's top module on MI355X.

Boundary: same constructor ``VoxelNetwork_depth(config, device='cuda')``, same public
attributes (``grid_coord_proj_batch``, ``coord_volumes``, ``coord_volume``, ``grid_coord_proj``, ``ray``,
``backbone``, ``process_features``, ``volume_net``, ``fisheye_camera_model``), same
``forward(images, grid_coord_proj_batch, coord_volumes, scene_volumes=None, depth_map_batch=None)`` and
same 4-tuple return as ``network/voxel_net_depth.py:19-275``; ``state_dict()`` has the reference's 699 keys.

What runs underneath is this build's own pipeline (one HIP stream, no host round trips):

    images --MIOpen--> features[B,64,64,256] --1x1 conv--> F[B,64,64,32]          (pose_resnet.FoldedBackbone)
    F --se_unproject_gather_planar3_f32--> X channels 0..31   (table-driven 4-tap, no 1024x1280 intermediate)
    depth --se_voxelize_planar3_f64--> X channel 32             (float64, bit-exact; occupancy straight into X)
         X is triplet-planar [B,11,G,G,G,3] on the float32 production path; channels-last [B,G,G,G,48|80] for
         with_intersection / scene_volumes inputs (se_unproject_gather_f32, se_voxelize_*_f64, se_intersection_f32)
    X --V2VProgram (se_conv3d_f32 / se_deconv3d_k2s2_f32 / se_maxpool3d_2_f32)--> logits[B,15,G^3]
    logits --se_softargmax3d_f32--> joints[B,15,3], volumes[B,15,G,G,G]

Deviations from the reference, all documented in DESIGN.md:
  * B may exceed ``opt.batch_size`` (the reference silently requires B <= 40, ``:241-242,269-270``);
  * the second return value is the compact ``[B,32,64,64]`` feature map unless the module was built with
    ``materialize_features=True`` (then the literal ``[B,32,1024,1280]`` tensor of ``:238`` is produced);
  * the voxeliser uses per-point index semantics (torch<=2.8 meaning of ``voxel[idx.T] = 1``).
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import _lib, op, pose_resnet
from .config import resolve_calibration_path
from .fisheye import FishEyeCameraCalibrated
from .v2v import V2VModel

FEATURE_CHANNELS = 32


class VoxelNetwork_depth(nn.Module):
    def __init__(self, config, device="cuda", materialize_features: bool = False, verbose: bool = True):
        super().__init__()
        say = print if verbose else (lambda *a, **k: None)
        self.device = device
        self.num_joints = config.model.backbone.num_joints

        # volume
        self.volume_softmax = config.model.volume_softmax
        self.volume_multiplier = config.model.volume_multiplier
        self.volume_size = config.model.volume_size
        self.cuboid_side = config.model.cuboid_side
        self.kind = config.model.kind

        # heatmap (stored, unused on this path — as in the reference)
        self.heatmap_softmax = config.model.heatmap_softmax
        self.heatmap_multiplier = config.model.heatmap_multiplier
        self.heatmap_shape = tuple(config.heatmap_shape)
        self.materialize_features = materialize_features

        if config.model.backbone.local_checkpoint:
            loads = torch.load(config.model.backbone.checkpoint, map_location="cpu")
            self.backbone = pose_resnet.get_pose_net(state_dict=loads["state_dict"])
        else:
            say("Do not load checkpoint")
            self.backbone = pose_resnet.get_pose_net(None)
        self.backbone = self.backbone.to(device)
        if config.opt.train_2d is False:
            for p in self.backbone.parameters():
                p.requires_grad = False

        # 1x1 conv + nearest upsample + pad: kept as modules for the state-dict keys
        # (``process_features.0.{weight,bias}``); the upsample/pad are folded into the gather table.
        self.process_features = nn.Sequential(
            nn.Conv2d(256, FEATURE_CHANNELS, 1),
            nn.Upsample(size=(op.UPSAMPLED, op.UPSAMPLED)),
            nn.ConstantPad2d(padding=(op.PAD_X, op.PAD_X, 0, 0), value=0.0),
        ).to(device)

        self.with_scene = config.model.with_scene
        self.with_intersection = False
        if config.model.with_scene is True:
            if config.model.with_intersection is True:
                volume_input_channel_num = FEATURE_CHANNELS + 1 + FEATURE_CHANNELS
                self.with_intersection = True
            else:
                volume_input_channel_num = FEATURE_CHANNELS + 1
        else:
            volume_input_channel_num = FEATURE_CHANNELS
        self.volume_net = V2VModel(volume_input_channel_num, self.num_joints).to(device)

        say("build coord volume")
        self.coord_volume = op.build_coord_volume(self.volume_size, self.cuboid_side)
        self.coord_volumes = self.coord_volume.unsqueeze(0).expand(config.opt.batch_size, -1, -1, -1, -1).to(device)

        self.fisheye_camera_model = FishEyeCameraCalibrated(
            calibration_file_path=resolve_calibration_path(config.dataset.camera_calibration_path))
        say("build reprojected grid coord")
        self.grid_coord_proj = op.get_projected_2d_points_with_coord_volumes(
            fisheye_model=self.fisheye_camera_model, coord_volume=self.coord_volume)
        self.grid_coord_proj.requires_grad = False
        self.grid_coord_proj_batch = op.get_grid_coord_proj_batch(
            self.grid_coord_proj, batch_size=config.opt.batch_size, heatmap_shape=config.heatmap_shape)
        self.grid_coord_proj_batch.requires_grad = False
        self.grid_coord_proj_batch = self.grid_coord_proj_batch.to(device)

        self.image_width = config.dataset.image_width
        self.image_height = config.dataset.image_height
        self.ray = op.calculated_ray_direction_numpy(self.fisheye_camera_model, self.image_width, self.image_height)

        # device-side constant tables of this build (lazily uploaded, rebuilt if a caller passes other grids)
        self._tables_for = None
        self._gather_idx = self._gather_w = self._ray_tab = self._coord_flat = None
        self._folded = None
        self.use_graphs = False
        self.planar3_input = True      # float32 V2V input in the triplet-planar layout (False: channels-last; A/B switch)
        self._graphs = {}
        self._xbuf = {}
        # V2V storage type: "fp32" (parity path, default) or "bf16" (BASELINE config 3: bf16 activations/weights,
        # float32 accumulation; joints differ from the float32 reference by more than 1e-3)
        self.v2v_dtype = torch.bfloat16 if str(config.model.get("v2v_dtype", "fp32")).lower() in ("bf16", "bfloat16") \
            else torch.float32
        self.backbone_dtype = torch.bfloat16 if str(config.model.get("backbone_dtype", "fp32")).lower() in ("bf16", "bfloat16") \
            else torch.float32

    def set_backbone_dtype(self, dtype):
        """'fp32' (default) / 'bf16': MIOpen 2-D backbone precision (bf16 only pays off at batch >= 16 or under hipGraph
        replay: it issues the same number of launches)."""
        if isinstance(dtype, str):
            dtype = torch.bfloat16 if dtype.lower() in ("bf16", "bfloat16") else torch.float32
        assert dtype in (torch.float32, torch.bfloat16)
        self.backbone_dtype = dtype
        self._invalidate()
        return self

    def set_v2v_dtype(self, dtype):
        """'fp32' / 'bf16' (or the torch dtypes); takes effect at the next forward.  'split_bf16' (EXPERIMENTAL): float32 tensors
        everywhere, the 3x3x3 layers of the 64^3 / 32^3 / 16^3 levels computed with split-bf16 operands (three bf16 MFMA products per
        float32 product, float32 accumulation; csrc/conv3d_split.hip) - measured beside the headline, never part of it."""
        self.v2v_split3 = isinstance(dtype, str) and dtype.lower() in ("split_bf16", "split-bf16", "bf16x3")
        if isinstance(dtype, str):
            dtype = torch.bfloat16 if dtype.lower() in ("bf16", "bfloat16") else torch.float32
        assert dtype in (torch.float32, torch.bfloat16)
        self.v2v_dtype = dtype
        self._invalidate()
        return self

    # ------------------------------------------------------------------------------------------
    def _invalidate(self):
        self._folded = None
        self._graphs = {}
        self._xbuf = {}

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._invalidate()

    def _apply(self, fn, *a, **k):
        self._invalidate()
        self._tables_for = None
        out = super()._apply(fn, *a, **k)
        # plain-attribute constants follow the module like the reference's ``.to(device)`` in __init__
        for name in ("coord_volumes", "grid_coord_proj_batch"):
            t = getattr(self, name, None)
            if isinstance(t, torch.Tensor):
                setattr(self, name, fn(t))
        return out

    def compile(self, dtype=None):
        """Fold/pack all weights for inference (done lazily by forward; call after changing weights in place)."""
        dtype = dtype or self.backbone_dtype
        if self.training:
            raise RuntimeError("VoxelNetwork_depth runs inference only: call .eval() before forward()")
        dev = next(self.volume_net.parameters()).device
        if dev.type != "cuda":
            raise _lib.HipExtensionError("VoxelNetwork_depth.forward needs the module on a HIP device (got %s); "
                                         "there is no CPU fallback" % dev)
        fb = pose_resnet.FoldedBackbone(self.backbone, dtype=dtype)
        pf = self.process_features[0]
        w = pf.weight.detach().to(dtype).contiguous()
        b = pf.bias.detach().to(dtype)
        self._folded = (fb, w, b)
        # volume_multiplier (reference :271 ``volumes * self.volume_multiplier``) is folded into the output layer's packed weights
        self.volume_net.compile(self.v2v_dtype, output_scale=float(self.volume_multiplier), split3=getattr(self, "v2v_split3", False))
        return self

    def _device_tables(self, grid_coord_proj_batch, coord_volumes, device, feat_hw=(64, 64)):
        """Device constants of the forward.  ``feat_hw``: size of the backbone's feature map - the reference upsamples ANY size to
        1024 x 1024 (``network/voxel_net_depth.py:59-60,238``), so the 4-tap table is built for the map this call really has (64 x 64
        for the 256 x 256 crop) and the largest texel index it holds is kept beside it (checked against the map in _forward_impl)."""
        key = (grid_coord_proj_batch.data_ptr(), coord_volumes.data_ptr(), str(device), tuple(feat_hw))
        if self._tables_for == key:
            return
        G = self.volume_size
        grid = grid_coord_proj_batch[0].reshape(-1, 2)
        idx, w = op.build_gather_table(grid, self.heatmap_shape, feat_hw=tuple(feat_hw))
        self._gather_max = int(idx.max())
        self._gather_idx, self._gather_w = idx.to(device), w.to(device)
        self._coord_flat = coord_volumes[0].reshape(G * G * G, 3).to(device=device, dtype=torch.float32).contiguous()
        ray_tab = op.build_voxelizer_ray_table(self.ray, self.image_width, self.image_height)
        self._ray_tab = torch.from_numpy(ray_tab).to(device)
        self._tables_for = key

    # ------------------------------------------------------------------------------------------
    def depth_map_to_voxel(self, depth_map_batch):
        """[B,H,W] depth (metres) -> [B,G,G,G] occupancy, on device (reference ``:194-222`` for the whole batch)."""
        dev = depth_map_batch.device
        self._device_tables(self.grid_coord_proj_batch, self.coord_volumes, dev, feat_hw=getattr(self, "_feat_hw", (64, 64)))
        B = depth_map_batch.shape[0]
        depth = depth_map_batch.reshape(B, depth_map_batch.shape[-2], depth_map_batch.shape[-1]).float().contiguous()
        G = self.volume_size
        occ = torch.empty((B, G, G, G), device=dev, dtype=torch.float32)
        _lib.voxelize(depth, self._ray_tab, occ, B, depth.shape[1], depth.shape[2], op.UPSAMPLED, op.PAD_X, G,
                      self.cuboid_side)
        return occ

    # ------------------------------------------------------------------------------------------
    # hipGraph replay: at small batch the forward is ~250 short launches and the host (eager dispatch of the MIOpen
    # backbone + ctypes calls) is slower than the GPU; capturing the whole forward once per (batch, depth shape) and
    # replaying it removes that.  Inputs are copied into static buffers, outputs are static tensors that the next
    # call overwrites (callers such as demo.py consume them immediately).
    # ------------------------------------------------------------------------------------------
    def enable_graphs(self, flag: bool = True):
        self.use_graphs = flag
        if not flag:
            self._graphs = {}
        return self

    def _graph_for(self, images, depth, grid_coord_proj_batch, coord_volumes):
        key = (tuple(images.shape), tuple(depth.shape), grid_coord_proj_batch.data_ptr(), coord_volumes.data_ptr())
        ent = self._graphs.get(key)
        if ent is not None:
            return ent
        s_img = torch.empty_like(images).copy_(images)
        s_depth = torch.empty_like(depth).copy_(depth)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the capture: lazy compile(), MIOpen find, hipFuncSetAttribute
            for _ in range(2):
                self._forward_impl(s_img, grid_coord_proj_batch, coord_volumes, None, s_depth)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self._forward_impl(s_img, grid_coord_proj_batch, coord_volumes, None, s_depth)
        ent = (graph, s_img, s_depth, out)
        self._graphs[key] = ent
        return ent

    @torch.no_grad()
    def forward(self, images, grid_coord_proj_batch, coord_volumes, scene_volumes=None, depth_map_batch=None):
        """See the module docstring; reference ``network/voxel_net_depth.py:224-275``."""
        _lib.require_hip(images)
        if self.with_scene is True and scene_volumes is None and depth_map_batch is None:
            print("no scene volume or depth input!")
            return None
        if getattr(self, "use_graphs", False) and depth_map_batch is not None and scene_volumes is None \
                and _lib._prof is None and images.dtype == torch.float32 and depth_map_batch.dtype == torch.float32:
            graph, s_img, s_depth, out = self._graph_for(images.contiguous(), depth_map_batch.contiguous(),
                                                         grid_coord_proj_batch, coord_volumes)
            s_img.copy_(images)
            s_depth.copy_(depth_map_batch)
            graph.replay()
            return out
        return self._forward_impl(images, grid_coord_proj_batch, coord_volumes, scene_volumes, depth_map_batch)

    @torch.no_grad()
    def _forward_impl(self, images, grid_coord_proj_batch, coord_volumes, scene_volumes=None, depth_map_batch=None):
        if self._folded is None:
            self.compile()
        dev = images.device
        B = images.shape[0]
        G = self.volume_size
        N = G * G * G
        fb, pw, pb = self._folded

        # 2D: backbone (MIOpen) + 1x1 channel reduction, channels-last
        with _lib.stage("backbone"):
            feat2d = torch.nn.functional.conv2d(fb(images), pw, pb)              # [B,32,64,64] for the 256 x 256 crop
            feat_nhwc = feat2d.permute(0, 2, 3, 1)
            if feat_nhwc.dtype != torch.float32 or not feat_nhwc.is_contiguous():
                feat_nhwc = feat_nhwc.float().contiguous()
        self._feat_hw = (int(feat2d.shape[2]), int(feat2d.shape[3]))
        self._device_tables(grid_coord_proj_batch, coord_volumes, dev, feat_hw=self._feat_hw)

        # lift to the volume: V2V input buffer [B,G,G,G,cin_pad], zero beyond the real channels
        prog = self.volume_net.program
        if prog.output_scale != float(self.volume_multiplier):      # the attribute was changed after compile()
            prog = self.volume_net.compile(self.v2v_dtype, output_scale=float(self.volume_multiplier), split3=getattr(self, "v2v_split3", False))
        C = FEATURE_CHANNELS
        # V2V input buffer: persistent per batch size, zero-filled once (pad channels stay zero; every call rewrites
        # the real channels), so no per-call clearing pass is needed
        bf16 = prog.dtype == torch.bfloat16
        fast_occ = (self.with_scene is True and scene_volumes is None and not self.with_intersection
                    and prog.cin_pad >= C + (8 if bf16 else 4))
        # float32 production case (features + depth occupancy): triplet-planar input [B,11,G,G,G,3] for the 7^3 front layer
        planar3 = (fast_occ and not bf16 and prog.cin == C + 1 and G % 8 == 0 and G >= 16 and self.planar3_input)
        # ... fully planar [B,33,G,G,G] when the frequency-domain front layer takes it (round 6; csrc/conv3d_fft7.hip)
        planar1 = planar3 and prog.fft7_ready(G)
        if planar1:
            planar3 = False
        # `with_scene: False` (V2VModel(32, 15) on the feature volume alone, reference :65-77): the same planar form, 32 planes, no occupancy
        if self.with_scene is not True and not bf16 and prog.cin == C and prog.fft7_ready(G) and self.planar3_input:
            planar1 = True
        xkey = (B, G, prog.cin_pad, str(dev), prog.dtype, planar3, planar1)
        x = self._xbuf.get(xkey)
        if x is None:
            self._xbuf.clear()
            shape = (B, prog.cin_pad // 8, G, G, G, 8) if bf16 else (B, G, G, G, prog.cin_pad)   # bf16: octet-planar
            if planar3:
                shape = (B, (C + 3) // 3, G, G, G, 3)
            if planar1:
                shape = (B, prog.cin, G, G, G)
            x = torch.zeros(shape, device=dev, dtype=prog.dtype)
            self._xbuf[xkey] = x
        xb = None
        if bf16 and not fast_occ and self.with_scene is True:
            # scene_volumes / with_intersection inputs: assembled in float32 by the _f32 operators, rounded once
            xb, x = x, torch.zeros((B, G, G, G, prog.cin_pad), device=dev, dtype=torch.float32)
        texels = feat_nhwc.shape[1] * feat_nhwc.shape[2]
        if self._gather_max >= texels:      # cannot happen with the table built above for this very map; the kernels do not check
            raise ValueError("gather table addresses texel %d of a %d-texel feature map" % (self._gather_max, texels))
        with _lib.stage("gather"):
            if planar1:
                _lib.unproject_gather_planar1(feat_nhwc, self._gather_idx, self._gather_w, x, B, texels, C, N, prog.cin)
            elif planar3:
                _lib.unproject_gather_planar3(feat_nhwc, self._gather_idx, self._gather_w, x, B, texels, C, N, x.shape[1])
            else:
                _lib.unproject_gather(feat_nhwc, self._gather_idx, self._gather_w, x, B, texels, C, N, prog.cin_pad, 0)

        with _lib.stage("voxelise"):
            self._voxelise(x, planar3, fast_occ, prog, scene_volumes, depth_map_batch, B, G, N, C, dev, planar1)
        if xb is not None:
            xb.copy_(x.view(B, G, G, G, prog.cin_pad // 8, 8).permute(0, 4, 1, 2, 3, 5))
            x = xb
        # float32 V2V with softmax volumes: pass 1 of the soft-argmax is computed by the V2V tail launch while the (scaled) logits
        # are in registers (se_pointwise_chain3_softargmax_f32); otherwise the two-pass kernel reads them back
        fused_sa = (self.volume_softmax and prog.cout <= 16
                    and ((N + 31) // 32 + 3) // 4 * 4 % 16 == 0)
        sa_scratch = torch.empty(_lib.softargmax3d_scratch_elems(B * self.num_joints), device=dev, dtype=torch.float32) if fused_sa else None
        with _lib.stage("v2v"):
            logits = prog.run(x, B, G, softargmax=(self._coord_flat, sa_scratch) if fused_sa else None, scaled=True, planar1=planar1)   # [B,J,N] planar, x volume_multiplier
        joints = torch.empty((B, self.num_joints, 3), device=dev, dtype=torch.float32)
        volumes = torch.empty_like(logits)
        with _lib.stage("softargmax"):
            if fused_sa:
                _lib.softargmax3d_finish(logits, sa_scratch, volumes, joints, B * self.num_joints, N, 1)
            else:
                _lib.softargmax3d(logits, self._coord_flat, volumes, joints, B * self.num_joints, N,
                                  1 if self.volume_softmax else 0)
        volumes = volumes.view(B, self.num_joints, G, G, G)

        features = feat2d
        if self.materialize_features:
            features = self.process_features[2](self.process_features[1](feat2d.float()))
        return joints, features, volumes, self.coord_volumes

    def _voxelise(self, x, planar3, fast_occ, prog, scene_volumes, depth_map_batch, B, G, N, C, dev, planar1=False):
        """Occupancy into the V2V input buffer ``x`` (reference ``:246-262``)."""
        if planar1 and self.with_scene is not True:
            return                    # no occupancy channel
        if planar1:
            # the gather zeroed plane 32; the voxeliser scatters the occupancy into it
            depth = depth_map_batch.reshape(B, depth_map_batch.shape[-2], depth_map_batch.shape[-1]).float().contiguous()
            _lib.voxelize_planar1(depth, self._ray_tab, x, B, depth.shape[1], depth.shape[2], op.UPSAMPLED, op.PAD_X, G,
                                  self.cuboid_side, C + 1, C)
        elif planar3:
            # the gather zeroed slot (10, 2) = channel 32; the voxeliser scatters the occupancy into it
            depth = depth_map_batch.reshape(B, depth_map_batch.shape[-2], depth_map_batch.shape[-1]).float().contiguous()
            _lib.voxelize_planar3(depth, self._ray_tab, x, B, depth.shape[1], depth.shape[2], op.UPSAMPLED, op.PAD_X, G,
                                  self.cuboid_side, x.shape[1], C)
        elif fast_occ:
            # occupancy straight into channel 32 of the V2V input (channels 33..35 cleared; 36.. are never read: the
            # 7^3 kernels walk ceil(33/4) = 9 four-channel chunks and the packed weights beyond channel 32 are zero)
            depth = depth_map_batch.reshape(B, depth_map_batch.shape[-2], depth_map_batch.shape[-1]).float().contiguous()
            _lib.voxelize_strided(depth, self._ray_tab, x, B, depth.shape[1], depth.shape[2], op.UPSAMPLED, op.PAD_X, G,
                                  self.cuboid_side, prog.cin_pad, C)
        elif self.with_scene is True:
            # the reference's scene_volumes branch ignores with_intersection (:246-249); kept
            use_inter = self.with_intersection and scene_volumes is None
            if scene_volumes is not None:
                occ = scene_volumes.reshape(B, G, G, G).to(device=dev, dtype=torch.float32).contiguous()
            else:
                occ = self.depth_map_to_voxel(depth_map_batch)
            if use_inter:
                _lib.intersection(x, occ, B, N, C, prog.cin_pad)                 # channels [32,64) = vol * occ
                x[..., 2 * C] = occ
            else:
                x[..., C] = occ



VoxelNetDepth = VoxelNetwork_depth  # the name BASELINE.json uses
