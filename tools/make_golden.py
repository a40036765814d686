"""Capture golden vectors from the REAL reference (build container only; /root/reference never travels).

Imports jianwang-mpi/SceneEgo from /root/reference with the shims of SURVEY.md Appendix B
(no bytecode written there; ``cv2``/``easydict`` stub modules; removed numpy aliases; the tuple-index patch
of ``network/voxel_net_depth.py:221``), loads the portable synthetic weights (sceneego_amd/synth.py), runs
``VoxelNetwork_depth.forward`` on seeded inputs and writes SMALL fixtures (inputs are seeds, outputs are
joints + sampled intermediates) under tests/golden/.  It also runs the CPU oracle (oracle/sceneego_oracle.py)
on the same inputs and prints / records the oracle-vs-reference differences — that is the oracle's pin.

Usage:  python tools/make_golden.py            (about 2 minutes, ~20 GB peak RSS)
"""
import sys

sys.dont_write_bytecode = True  # do not litter /root/reference with __pycache__

import hashlib
import json
import os
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------------------------------------
# shims
# ------------------------------------------------------------------------------------------------
def install_shims():
    np.float = float          # utils/fisheye/FishEyeCalibrated.py:42
    np.round_ = np.round      # network/voxel_net_depth.py:215
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 1
    cv2.IMREAD_ANYCOLOR, cv2.IMREAD_ANYDEPTH = 4, 2

    def _resize(src, dsize, interpolation=1):   # only INTER_NEAREST is hit on the path (voxel_net_depth.py:197)
        assert interpolation == 0
        w, h = dsize
        ys = np.minimum((np.arange(h) * (src.shape[0] / h)).astype(np.int64), src.shape[0] - 1)
        xs = np.minimum((np.arange(w) * (src.shape[1] / w)).astype(np.int64), src.shape[1] - 1)
        return src[ys][:, xs]

    cv2.resize = _resize
    sys.modules["cv2"] = cv2

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            d = dict(d or {}, **kw)
            for k, v in d.items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            elif isinstance(v, (list, tuple)):
                v = type(v)(EasyDict(x) if isinstance(x, dict) else x for x in v)
            super().__setattr__(k, v)
            super().__setitem__(k, v)

        __setitem__ = __setattr__

    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed


def import_reference():
    install_shims()
    sys.path.insert(0, REF)
    os.chdir(REF)  # calibration path in the YAML is relative (sceneego.yaml:72)
    from network.voxel_net_depth import VoxelNetwork_depth
    from utils import cfg, op

    # restore torch<=2.8 tuple semantics of  voxel[(3,N) ndarray] = 1  (voxel_net_depth.py:221)
    def _pc2vox(self, pc):
        p = pc.copy()
        p[:, 0] = (p[:, 0] + self.cuboid_side / 2) * self.volume_size / self.cuboid_side
        p[:, 1] = (p[:, 1] + self.cuboid_side / 2) * self.volume_size / self.cuboid_side
        p[:, 2] = (p[:, 2]) * self.volume_size / self.cuboid_side
        p = np.round(p)
        p = p[np.all((p >= 0) & (p <= self.volume_size - 1), axis=1)]
        v = torch.zeros((self.volume_size,) * 3)
        v[tuple(torch.from_numpy(p.T.astype(np.int64)))] = 1
        return v

    VoxelNetwork_depth.point_cloud_to_voxel_numpy = _pc2vox
    return VoxelNetwork_depth, cfg, op


# ------------------------------------------------------------------------------------------------
def sample_positions(n_total, count, seed):
    sys.path.insert(0, ROOT)
    from sceneego_amd import synth
    return np.unique((synth.uniform01(seed, "golden/positions", count) * n_total).astype(np.int64))


def run_case(name, RefNet, cfg_mod, synth, O, *, batch, in_seed, depth_kind, with_intersection=False, volume_size=64,
             weight_seed=0, extra=None, with_scene=True, volume_softmax=True, volume_multiplier=1.0):
    print(f"== case {name}")
    config = cfg_mod.load_config("experiments/sceneego/test/sceneego.yaml")
    config.model.with_intersection = with_intersection
    config.model.volume_size = volume_size
    config.model.with_scene = with_scene                  # network/voxel_net_depth.py:65-77 (False: V2VModel(32, 15), no depth input)
    config.model.volume_softmax = volume_softmax          # utils/op.py:86-91 (False: ReLU, no normalisation)
    config.model.volume_multiplier = volume_multiplier    # network/voxel_net_depth.py:271
    t0 = time.time()
    net = RefNet(config, device="cpu").eval()
    sd = synth.make_state_dict(net.state_dict(), seed=weight_seed)
    net.load_state_dict(sd, strict=True)
    img, depth = synth.make_inputs(in_seed, batch, "floor" if depth_kind == "demo_exr" else depth_kind)
    if name == "demo_b1":   # BASELINE config 1: the reference's demo frame (derived fixture) + seeded synthetic depth
        from sceneego_amd.preprocess import normalize_u8
        small = np.load(os.path.join(GOLD, "demo", "img_001000_256_bgr_u8.npz"))["img"]
        img = normalize_u8(small)[None]
    if name == "demo_exr_b1":   # config 1 with the reference's own depth map, decoded by sceneego_amd/exr.py
        from sceneego_amd.preprocess import normalize_u8, load_depth, prepare_depth
        small = np.load(os.path.join(GOLD, "demo", "img_001000_256_bgr_u8.npz"))["img"]
        img = normalize_u8(small)[None]
        depth = prepare_depth(load_depth(os.path.join(GOLD, "demo", "img_001000.jpg.exr")))[None]

    taps = {}
    hooks = []
    def _h_feat(m, i, o):
        taps["features64"] = o.detach()

    def _h_v2v(m, i, o):
        taps["v2v_in"] = i[0].detach()
        taps["logits"] = o.detach()

    hooks.append(net.process_features[0].register_forward_hook(_h_feat))
    hooks.append(net.volume_net.register_forward_hook(_h_v2v))
    layer_absmean = {}
    for mod_name, mod in net.volume_net.named_modules():
        if isinstance(mod, (torch.nn.Conv3d, torch.nn.ConvTranspose3d)):
            def _h_layer(m, i, o, n=mod_name):
                layer_absmean[n] = float(o.detach().abs().mean())

            hooks.append(mod.register_forward_hook(_h_layer))
    with torch.no_grad():
        t1 = time.time()
        kp, feats, vols, cv = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth if with_scene else None)
        t_fwd = time.time() - t1
    for h in hooks:
        h.remove()
    G = volume_size
    N = G ** 3
    occ_ch = 64 if with_intersection else 32
    occ = taps["v2v_in"][:, occ_ch] if with_scene else torch.zeros((batch, G, G, G))
    assert taps["v2v_in"].shape[1] == (65 if with_intersection else 33 if with_scene else 32)
    print(f"   init+load {t1 - t0:.1f}s forward {t_fwd:.1f}s joints[0,0]={kp[0, 0].tolist()} occupied={occ.sum(dim=(1, 2, 3)).tolist()}")

    pos = sample_positions(N, 1024, 99)
    gold = {
        "joints": kp.numpy().astype(np.float32),
        "features64_sub": taps["features64"][:, :, ::8, ::8].numpy().astype(np.float32),
        "occupancy_bits": np.stack([np.packbits(o.numpy().reshape(-1).astype(np.uint8)) for o in occ]),
        "occupancy_count": occ.sum(dim=(1, 2, 3)).numpy().astype(np.int64),
        "sample_pos": pos,
        "feature_volume_samples": taps["v2v_in"][:, :32].reshape(batch, 32, N)[:, :, pos].numpy().astype(np.float32),
        "logits_samples": taps["logits"].reshape(batch, -1, N)[:, :, pos].numpy().astype(np.float32),
        "volumes_samples": vols.reshape(batch, -1, N)[:, :, pos].numpy().astype(np.float32),
        "volumes_max": vols.reshape(batch, -1, N).max(dim=2)[0].numpy().astype(np.float32),
        "layer_names": np.array(sorted(layer_absmean.keys())),
        "layer_absmean": np.array([layer_absmean[k] for k in sorted(layer_absmean.keys())], dtype=np.float64),
    }
    if with_intersection:
        gold["intersection_samples"] = taps["v2v_in"][:, 32:64].reshape(batch, 32, N)[:, :, pos].numpy().astype(np.float32)

    # ---- oracle on the same inputs: this is the pin -------------------------------------------
    const = O.Constants(os.path.join(ROOT, "sceneego_amd", "calibration", "fisheye.calibration_05_08.json"), G=G)
    otaps = {}
    oj, obig, ovols = O.forward(sd, const, img, depth if with_scene else None, with_scene=with_scene, with_intersection=with_intersection,
                                volume_softmax=volume_softmax, volume_multiplier=volume_multiplier, taps=otaps)
    if not with_scene:
        otaps["occupancy"] = occ
    # the same formula with float64 sums, from the REFERENCE's logits: what a platform-independent evaluation gives (the float32
    # einsum over G^3 terms is reduction-order dependent; for the un-normalised ReLU mode the sums are ~1e5 and the float32 noise
    # is correspondingly large) - recorded so the GPU tests can state their tolerance against both
    kp64 = O.integrate(taps["logits"] * volume_multiplier, const.coord, softmax=volume_softmax, accumulate64=True)[0]
    gold["joints_f64_evaluation"] = kp64.numpy().astype(np.float32)
    diffs = {
        "joints": float((oj - kp).abs().max()),
        "features64": float((otaps["features64"] - taps["features64"]).abs().max()),
        "feature_volume": float((otaps["feature_volume"] - taps["v2v_in"][:, :32]).abs().max()),
        "occupancy_mismatch_voxels": int((otaps["occupancy"] != occ).sum()),
        "logits": float((otaps["logits"] - taps["logits"]).abs().max()),
        "volumes": float((ovols - vols).abs().max()),
        "features_big": float((obig - feats).abs().max()),
        "joints_f32_vs_f64_evaluation": float((kp64 - kp).abs().max()),
        "joints_absmax": float(kp.abs().max()),
    }
    print("   oracle vs reference:", diffs)
    if extra is not None:
        extra(net, gold)
    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), **gold)
    meta = dict(name=name, batch=batch, input_seed=in_seed, depth_kind=depth_kind, with_intersection=with_intersection,
                volume_size=volume_size, weight_seed=weight_seed, with_scene=with_scene, volume_softmax=volume_softmax,
                volume_multiplier=volume_multiplier, oracle_vs_reference=diffs,
                reference_forward_s=round(t_fwd, 2))
    return meta, net


def preprocess_case(synth):
    """f1 pin: the REAL ``DemoDataset.__getitem__`` (dataset/demo_dataset.py:67-98) with the real ``Normalize`` / ``ToTensor``
    (utils/data_transforms.py:38-72) run on known arrays.  Only the three OpenCV calls are stubs (no cv2 in this image):
    ``cv2.imread`` hands over a prepared array (PIL's libjpeg-turbo decode for the demo JPEG, oracle/exr_oracle.py for the demo
    EXR, seeded arrays otherwise), ``cv2.resize`` INTER_LINEAR is the restated exact-quarter mapping, INTER_NEAREST the restated
    floor mapping - those stay "parity unpinned".  Everything between them is the reference's own code: the 128-column crop, /255,
    mean / std in BGR order, HWC->CHW, float(); ``[:, :, 0]``, the 10 m clamp, float()."""
    import tempfile
    from dataset.demo_dataset import DemoDataset
    from utils import cfg as cfg_mod
    import cv2 as cv2_stub
    from oracle import exr_oracle
    from PIL import Image

    def frame_u8(seed):        # portable seeded full frame [1024,1280,3] uint8 (what tests regenerate on any box)
        return (synth.uniform01(seed, "f1/frame", 1024 * 1280 * 3) * 256.0).astype(np.uint8).reshape(1024, 1280, 3)

    def depth_f32(seed, shape):  # seeded depth in [0, 12) m: about 1/6 of the pixels exceed the 10 m clamp
        return (synth.uniform01(seed, "f1/depth", int(np.prod(shape))) * 12.0).astype(np.float32).reshape(shape)

    with Image.open(os.path.join(REF, "data", "demo", "imgs", "img_001000.jpg")) as im:
        demo_bgr = np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
    demo_depth = exr_oracle.read(os.path.join(REF, "data", "demo", "depths", "img_001000.jpg.exr"))["Y"].astype(np.float32)
    arrays = {
        "synth_a.png": frame_u8(31), "synth_a.png.exr": depth_f32(32, (512, 640)),            # native half-size depth, 1 channel
        "synth_b.png": frame_u8(33), "synth_b.png.exr": depth_f32(34, (1024, 1280, 3)),       # full size, 3 channels
        "img_001000.jpg": demo_bgr, "img_001000.jpg.exr": demo_depth,
    }
    calls = []

    def _imread(path, flags=None):
        calls.append(("imread", os.path.basename(path), flags))
        return arrays[os.path.basename(path)].copy()

    def _resize(src, dsize, interpolation=1):
        calls.append(("resize", tuple(src.shape), tuple(dsize), interpolation))
        w, h = dsize
        if interpolation == 0:
            ys = np.minimum((np.arange(h) * (src.shape[0] / h)).astype(np.int64), src.shape[0] - 1)
            xs = np.minimum((np.arange(w) * (src.shape[1] / w)).astype(np.int64), src.shape[1] - 1)
            return src[ys][:, xs]
        assert src.dtype == np.uint8 and src.shape[0] == 4 * h and src.shape[1] == 4 * w      # INTER_LINEAR at exactly 1/4
        a = src.astype(np.uint16)
        return ((a[1::4, 1::4] + a[1::4, 2::4] + a[2::4, 1::4] + a[2::4, 2::4] + 2) >> 2).astype(np.uint8)

    cv2_stub.imread, cv2_stub.resize = _imread, _resize
    import dataset.demo_dataset as dd
    dd.cv2.imread, dd.cv2.resize = _imread, _resize
    config = cfg_mod.load_config("experiments/sceneego/test/sceneego.yaml")
    gold = {}
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    info = {}
    with tempfile.TemporaryDirectory() as tmp:
        idir, ddir = os.path.join(tmp, "imgs"), os.path.join(tmp, "depths")
        os.makedirs(idir); os.makedirs(ddir)
        for name in ("synth_a.png", "synth_b.png", "img_001000.jpg"):
            open(os.path.join(idir, name), "wb").close()
            open(os.path.join(ddir, name + ".exr"), "wb").close()
        ds = DemoDataset(config, idir, ddir, voxel_output=False)
        for i in range(len(ds)):
            img_t, img_rgb_t, depth_t, path = ds[i]
            name = os.path.splitext(os.path.basename(path))[0]
            assert img_t.dtype == torch.float32 and tuple(img_t.shape) == (3, 256, 256)
            assert depth_t.dtype == torch.float32 and tuple(depth_t.shape) == (1024, 1280)
            d = depth_t.numpy()
            info[name] = {"image_sha256": sha(img_t.numpy()), "image_rgb_sha256": sha(img_rgb_t.numpy()), "depth_sha256": sha(d),
                          "depth_max": float(d.max()), "depth_clamped_pixels": int((d == 10.0).sum())}
            gold[name + "_image"] = img_t.numpy()
            gold[name + "_depth_sub"] = d[::16, ::16].copy()
            print("  ", name, info[name])
    gold["seeds"] = np.array([31, 32, 33, 34])
    np.savez_compressed(os.path.join(GOLD, "preprocess.npz"), **gold)
    with open(os.path.join(GOLD, "preprocess.json"), "w") as f:
        json.dump({"generator": "tools/make_golden.py --only-preprocess: reference DemoDataset.__getitem__ + Normalize + ToTensor on known "
                                "arrays; cv2.imread / cv2.resize are stubs (see preprocess_case)",
                   "inputs": {"synth_a": "frame synth.uniform01(31, 'f1/frame') * 256 -> uint8 [1024,1280,3]; depth uniform01(32, 'f1/depth') * 12 -> float32 [512,640]",
                              "synth_b": "frame seed 33; depth seed 34, float32 [1024,1280,3] (channel 0 is the map)",
                              "img_001000": "reference data/demo/imgs/img_001000.jpg decoded by PIL (libjpeg-turbo) + data/demo/depths/img_001000.jpg.exr decoded by oracle/exr_oracle.py"},
                   "cv2_calls_made_by_the_reference": [list(map(str, c)) for c in calls], "outputs": info}, f, indent=1)
    print("wrote preprocess.npz / preprocess.json")


def constants_case(net, op_mod):
    """Init-time constants + the reference's own soft-argmax known-answer case (voxel_net_depth.py:302-320)."""
    gold = {}
    gp = net.grid_coord_proj.numpy()
    gold["grid_coord_proj_every997"] = gp[::997].astype(np.float32)
    gold["grid_norm_every997"] = net.grid_coord_proj_batch[0, ::997, 0].numpy().astype(np.float32)
    gold["coord_volume_corners"] = net.coord_volume[[0, 0, 63, 63, 32], [0, 63, 0, 63, 32], [0, 63, 63, 0, 32]].numpy().astype(np.float32)
    ray = net.ray
    ridx = np.array([0, 1, 1023, 1024, 640 * 1024 + 512, 128 * 1024, 1151 * 1024 + 1023, 1279 * 1024 + 1023] +
                    list(range(5000, 1310720, 23411)))
    gold["ray_idx"] = ridx
    gold["ray_values"] = ray[ridx].astype(np.float64)
    gold["ray_sha256_f64"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(ray).tobytes()).digest(), dtype=np.uint8)
    gold["img_center"] = net.fisheye_camera_model.img_center.astype(np.float64)
    # KAT: volumes all zero except ones at [32,32,32] and [31,31,31]
    volumes = torch.zeros((4, 15, 64, 64, 64))
    volumes[:, :, 32, 32, 32] = 1
    volumes[:, :, 31, 31, 31] = 1
    kp, v = op_mod.integrate_tensor_3d_with_coordinates(volumes, net.coord_volumes[:4], softmax=True)
    gold["kat_softargmax_joints"] = kp.numpy().astype(np.float32)
    gold["kat_softargmax_peak"] = np.array([float(v[0, 0, 32, 32, 32]), float(v[0, 0, 0, 0, 0])], dtype=np.float32)
    kp2, v2 = op_mod.integrate_tensor_3d_with_coordinates(volumes, net.coord_volumes[:4], softmax=False)
    gold["kat_relu_joints"] = kp2.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "constants.npz"), **gold)
    print("   KAT softargmax joints[0,0] =", kp[0, 0].tolist())


def main():
    os.makedirs(GOLD, exist_ok=True)
    sys.path.insert(0, ROOT)
    from oracle import sceneego_oracle as O
    from sceneego_amd import synth

    RefNet, cfg_mod, op_mod = import_reference()
    metas = []
    if len(sys.argv) > 1 and sys.argv[1] == "--exr-hashes":
        # f1: every demo depth map of the reference decoded by the INDEPENDENT brute-force decoder (oracle/exr_oracle.py);
        # tests compare the product reader (sceneego_amd/exr.py) with these digests and, on the committed file, array by array
        from oracle import exr_oracle
        from sceneego_amd import exr
        d = os.path.join(REF, "data", "demo", "depths")
        out = {}
        for f in sorted(os.listdir(d)):
            y = exr_oracle.read(os.path.join(d, f))["Y"]
            prod = exr.read_depth_exr(os.path.join(d, f))
            out[f] = {"shape": list(y.shape), "dtype": str(y.dtype), "sha256_float32": hashlib.sha256(np.ascontiguousarray(y.astype(np.float32)).tobytes()).hexdigest(),
                      "min": float(y.min()), "max": float(y.max()), "product_reader_equal": bool(np.array_equal(y.astype(np.float32), prod))}
            print(f, out[f])
        with open(os.path.join(GOLD, "exr_hashes.json"), "w") as fo:
            json.dump({"generator": "tools/make_golden.py --exr-hashes (oracle/exr_oracle.py on /root/reference/data/demo/depths)", "files": out}, fo, indent=1)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-features-big":
        # 2nd return value of the reference forward, the literal [B,32,1024,1280] tensor (voxel_net_depth.py:238,275),
        # sub-sampled on a fixed lattice that hits the zero pad columns, block interiors and block edges
        config = cfg_mod.load_config("experiments/sceneego/test/sceneego.yaml")
        net = RefNet(config, device="cpu").eval()
        net.load_state_dict(synth.make_state_dict(net.state_dict(), seed=0), strict=True)
        img, depth = synth.make_inputs(77, 1, "floor")
        with torch.no_grad():
            _, feats, _, _ = net(img, net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth)
        assert tuple(feats.shape) == (1, 32, 1024, 1280)
        rows = np.arange(0, 1024, 37)
        cols = np.concatenate([np.arange(0, 1280, 41), np.array([127, 128, 143, 144, 1151, 1152, 1279])])
        np.savez_compressed(os.path.join(GOLD, "b1_floor_features_big.npz"), rows=rows, cols=cols,
                            values=feats[0][:, rows][:, :, cols].numpy().astype(np.float32),
                            shape=np.array(feats.shape), absmax=np.float32(feats.abs().max()))
        print("wrote b1_floor_features_big.npz", feats.shape)
        return
    if len(sys.argv) > 1 and sys.argv[1] in ("--only-demo", "--only-demo-exr"):
        if sys.argv[1] == "--only-demo":
            m, _ = run_case("demo_b1", RefNet, cfg_mod, synth, O, batch=1, in_seed=1000, depth_kind="floor")
        else:
            m, _ = run_case("demo_exr_b1", RefNet, cfg_mod, synth, O, batch=1, in_seed=1000, depth_kind="demo_exr")
        with open(os.path.join(GOLD, "META.json")) as f:
            meta = json.load(f)
        meta["cases"] = [c for c in meta["cases"] if c["name"] != m["name"]] + [m]
        with open(os.path.join(GOLD, "META.json"), "w") as f:
            json.dump(meta, f, indent=1)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-preprocess":
        preprocess_case(synth)
        assert not any(d == "__pycache__" for _, ds, _ in os.walk(REF) for d in ds), "reference tree was modified!"
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-branches":
        # round 3: the configuration branches of the reference forward no golden covered yet
        new = [
            run_case("b1_noscene", RefNet, cfg_mod, synth, O, batch=1, in_seed=606, depth_kind="floor", with_scene=False)[0],
            run_case("b1_relu_volumes", RefNet, cfg_mod, synth, O, batch=1, in_seed=707, depth_kind="floor", volume_softmax=False)[0],
            run_case("b1_multiplier2", RefNet, cfg_mod, synth, O, batch=1, in_seed=808, depth_kind="uniform", volume_multiplier=2.0)[0],
        ]
        with open(os.path.join(GOLD, "META.json")) as f:
            meta = json.load(f)
        names = {m["name"] for m in new}
        meta["cases"] = [c for c in meta["cases"] if c["name"] not in names] + new
        with open(os.path.join(GOLD, "META.json"), "w") as f:
            json.dump(meta, f, indent=1)
        assert not any(d == "__pycache__" for _, ds, _ in os.walk(REF) for d in ds), "reference tree was modified!"
        return
    m, net = run_case("b2_uniform", RefNet, cfg_mod, synth, O, batch=2, in_seed=1234, depth_kind="uniform")
    metas.append(m)
    constants_case(net, op_mod)
    del net
    m, _ = run_case("b1_floor", RefNet, cfg_mod, synth, O, batch=1, in_seed=77, depth_kind="floor")
    metas.append(m)
    m, _ = run_case("b1_intersection", RefNet, cfg_mod, synth, O, batch=1, in_seed=4321, depth_kind="uniform",
                    with_intersection=True)
    metas.append(m)
    m, _ = run_case("b1_g128_floor", RefNet, cfg_mod, synth, O, batch=1, in_seed=555, depth_kind="floor", volume_size=128)
    metas.append(m)
    m, _ = run_case("demo_b1", RefNet, cfg_mod, synth, O, batch=1, in_seed=1000, depth_kind="floor")
    metas.append(m)
    m, _ = run_case("demo_exr_b1", RefNet, cfg_mod, synth, O, batch=1, in_seed=1000, depth_kind="demo_exr")
    metas.append(m)

    meta = {
        "generator": "tools/make_golden.py",
        "reference": "jianwang-mpi/SceneEgo @ /root/reference (read-only mount)",
        "torch": torch.__version__, "numpy": np.__version__,
        "shims": ["sys.dont_write_bytecode", "np.float=float", "np.round_=np.round", "cv2 stub (resize INTER_NEAREST = floor(dst*src/dst))",
                  "easydict stub"],
        "patch": "VoxelNetwork_depth.point_cloud_to_voxel_numpy: voxel[tuple(idx.T)] = 1 (torch<=2.8 semantics of voxel_net_depth.py:221); "
                 "un-patched torch 2.10 treats the (3,N) ndarray as a dim-0 tensor index and fills whole slabs",
        "weights": "sceneego_amd.synth.make_state_dict(seed) + sceneego_amd/synth_calibration.json",
        "cases": metas,
    }
    with open(os.path.join(GOLD, "META.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", GOLD)
    assert not any(d == "__pycache__" for _, ds, _ in os.walk(REF) for d in ds), "reference tree was modified!"


if __name__ == "__main__":
    main()
