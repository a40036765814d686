"""GPU: V2V program and the whole VoxelNetwork_depth.forward against the oracle and the reference goldens.

Tolerance from BASELINE.json north_star: <= 1e-3 on the 15x3 joint coordinates (metres).
"""
import numpy as np
import pytest
import torch

from oracle import sceneego_oracle as O
from sceneego_amd import load_config, synth
from sceneego_amd.voxel_net_depth import VoxelNetwork_depth

from conftest import synthetic_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
JOINT_TOL = 1e-3


def _build(with_intersection=False, volume_size=64):
    cfg = load_config()
    cfg.model.with_intersection = with_intersection
    cfg.model.volume_size = volume_size
    net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
    net.load_state_dict(synthetic_state_dict(with_intersection), strict=True)
    return net.to(DEV).eval()


@pytest.fixture(scope="module")
def net64():
    return _build()


def _forward(net, img, depth):
    with torch.no_grad():
        out = net(img.to(DEV), net.grid_coord_proj_batch, net.coord_volumes, depth_map_batch=depth.to(DEV))
    torch.cuda.synchronize()
    return out


def _check_against_golden(net, g, m):
    img, depth = synth.make_inputs(m["input_seed"], m["batch"], m["depth_kind"])
    kp, feats, vols, cv = _forward(net, img, depth)
    B, N = m["batch"], m["volume_size"] ** 3
    assert tuple(kp.shape) == (B, 15, 3) and tuple(vols.shape) == (B, 15) + (m["volume_size"],) * 3
    err = float(np.abs(kp.cpu().numpy() - g["joints"]).max())
    assert err <= JOINT_TOL, f"joints differ from the reference golden by {err}"
    pos = torch.from_numpy(g["sample_pos"]).to(DEV)
    vs = vols.reshape(B, 15, N)[:, :, pos].cpu().numpy()
    assert np.abs(vs - g["volumes_samples"]).max() <= 2e-3 * g["volumes_max"].max() + 1e-7
    np.testing.assert_allclose(feats[:, :, ::8, ::8].float().cpu().numpy(), g["features64_sub"], rtol=2e-3, atol=2e-3)
    return err


def test_forward_b1_floor_vs_golden(net64, golden, golden_meta):
    m = next(c for c in golden_meta["cases"] if c["name"] == "b1_floor")
    _check_against_golden(net64, golden("b1_floor"), m)


def test_forward_b2_uniform_vs_golden(net64, golden, golden_meta):
    m = next(c for c in golden_meta["cases"] if c["name"] == "b2_uniform")
    _check_against_golden(net64, golden("b2_uniform"), m)


def test_forward_intersection_vs_golden(golden, golden_meta):
    m = next(c for c in golden_meta["cases"] if c["name"] == "b1_intersection")
    _check_against_golden(_build(with_intersection=True), golden("b1_intersection"), m)


def test_forward_g128_vs_golden(golden, golden_meta):
    m = next(c for c in golden_meta["cases"] if c["name"] == "b1_g128_floor")
    _check_against_golden(_build(volume_size=128), golden("b1_g128_floor"), m)


# ---- configuration branches of the reference forward (goldens: tools/make_golden.py --only-branches) --------------------------
def _build_cfg(**model_overrides):
    cfg = load_config()
    for k, v in model_overrides.items():
        setattr(cfg.model, k, v)
    net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
    net.load_state_dict(synth.make_state_dict(net.state_dict(), seed=0), strict=True)
    return net.to(DEV).eval()


def test_forward_no_scene_vs_golden(golden, golden_meta):
    """`with_scene: False` (network/voxel_net_depth.py:65-77): V2VModel(32, 15) on the gathered feature volume alone - the shape of the
    reference author's own benchmark (network/v2v.py:259-270).  The 7^3 front layer then has 32 input channels (channels-last F(4,7)
    form, ragged last 3-channel chunk)."""
    m = next(c for c in golden_meta["cases"] if c["name"] == "b1_noscene")
    g = golden("b1_noscene")
    net = _build_cfg(with_scene=False)
    assert net.volume_net.input_channels == 32
    img, _ = synth.make_inputs(m["input_seed"], m["batch"], m["depth_kind"])
    with torch.no_grad():
        kp, feats, vols, _ = net(img.to(DEV), net.grid_coord_proj_batch, net.coord_volumes)      # no scene input needed (:263 not taken)
    torch.cuda.synchronize()
    err = float(np.abs(kp.cpu().numpy() - g["joints"]).max())
    print(f"with_scene False: joints vs reference golden {err:.2e} m")
    assert err <= JOINT_TOL, err
    pos = torch.from_numpy(g["sample_pos"]).to(DEV)
    vs = vols.reshape(1, 15, -1)[:, :, pos].cpu().numpy()
    assert np.abs(vs - g["volumes_samples"]).max() <= 2e-3 * g["volumes_max"].max() + 1e-7
    # B=8 (the author's benchmark batch): every frame equals its own B=1 run
    img8, _ = synth.make_inputs(61, 8, "uniform")
    with torch.no_grad():
        kp8 = net(img8.to(DEV), net.grid_coord_proj_batch, net.coord_volumes)[0]
        one = net(img8[5:6].to(DEV), net.grid_coord_proj_batch, net.coord_volumes)[0]
    assert float((kp8[5:6] - one).abs().max()) < 5e-5


def test_forward_volume_multiplier_vs_golden(golden, golden_meta):
    """`volume_multiplier: 2.0` (network/voxel_net_depth.py:271): folded into the output layer's packed weights, so the fused
    soft-argmax pass stays on; the plain V2VModel.forward still returns the unscaled logits."""
    m = next(c for c in golden_meta["cases"] if c["name"] == "b1_multiplier2")
    g = golden("b1_multiplier2")
    net = _build_cfg(volume_multiplier=2.0)
    img, depth = synth.make_inputs(m["input_seed"], m["batch"], m["depth_kind"])
    from sceneego_amd import _lib
    calls = []
    orig = _lib.softargmax3d_finish
    _lib.softargmax3d_finish = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        kp, _, vols, _ = _forward(net, img, depth)
    finally:
        _lib.softargmax3d_finish = orig
    assert calls, "the fused soft-argmax path must stay on with volume_multiplier != 1"
    err = float(np.abs(kp.cpu().numpy() - g["joints"]).max())
    print(f"volume_multiplier 2: joints vs reference golden {err:.2e} m")
    assert err <= JOINT_TOL, err
    pos = torch.from_numpy(g["sample_pos"]).to(DEV)
    vs = vols.reshape(1, 15, -1)[:, :, pos].cpu().numpy()
    assert np.abs(vs - g["volumes_samples"]).max() <= 2e-3 * g["volumes_max"].max() + 1e-7
    prog = net.volume_net.program
    assert prog.output_scale == 2.0 and prog.out_scaled is not prog.out
    # changing the attribute after compile() takes effect at the next forward
    net.volume_multiplier = 1.0
    kp1 = _forward(net, img, depth)[0]
    assert net.volume_net.program.output_scale == 1.0 and float((kp1 - kp).abs().max()) > 1e-3


def test_forward_relu_volumes_vs_golden(golden, golden_meta):
    """`volume_softmax: False` (utils/op.py:89-91): ReLU without normalisation, joints = sum relu(v) * coord ~ 1e4..1e6 (not
    metres).  The reference's float32 einsum of that sum differs from the float64 evaluation of its own logits by 1e-7 relative
    (META.json: joints_f32_vs_f64_evaluation 0.25 at |joint| 2.3e6), so the bound is relative: 1e-4 of the largest coordinate."""
    m = next(c for c in golden_meta["cases"] if c["name"] == "b1_relu_volumes")
    g = golden("b1_relu_volumes")
    net = _build_cfg(volume_softmax=False)
    img, depth = synth.make_inputs(m["input_seed"], m["batch"], m["depth_kind"])
    kp, _, vols, _ = _forward(net, img, depth)
    scale = float(np.abs(g["joints"]).max())
    err = float(np.abs(kp.cpu().numpy() - g["joints"]).max())
    err64 = float(np.abs(kp.cpu().numpy() - g["joints_f64_evaluation"]).max())
    print(f"volume_softmax False: |joint| up to {scale:.3e}; vs reference float32 {err:.3e}, vs float64 evaluation {err64:.3e}")
    assert err <= 1e-4 * scale and err64 <= 1e-4 * scale, (err, err64, scale)
    pos = torch.from_numpy(g["sample_pos"]).to(DEV)
    vs = vols.reshape(1, 15, -1)[:, :, pos].cpu().numpy()
    assert float(vs.min()) >= 0.0
    assert np.abs(vs - g["volumes_samples"]).max() <= 1e-4 * float(np.abs(g["logits_samples"]).max())


def test_forward_split_bf16_mode_vs_goldens(golden, golden_meta):
    """EXPERIMENTAL `set_v2v_dtype("split_bf16")`: float32 tensors, the 3x3x3 layers of the 64^3 / 32^3 / 16^3 levels on split-bf16
    operands (three bf16 MFMA products per float32 product).  Unlike the bf16-STORAGE mode this one is inside the north-star
    tolerance: joints <= 1e-3 m of the reference goldens (CPU emulation of the scheme: 2e-5 m, tests/test_oracle_golden.py)."""
    net = _build()
    net.set_v2v_dtype("split_bf16")
    for case in ("b1_floor", "b2_uniform"):
        m = next(c for c in golden_meta["cases"] if c["name"] == case)
        g = golden(case)
        img, depth = synth.make_inputs(m["input_seed"], m["batch"], m["depth_kind"])
        kp, _, vols, _ = _forward(net, img, depth)
        prog = net.volume_net.program
        assert prog.split3 and prog.dtype == torch.float32 and prog.front_res[1][0].w_split is not None
        err = float(np.abs(kp.cpu().numpy() - g["joints"]).max())
        print(f"split_bf16 mode, {case}: joints vs reference golden {err:.2e} m")
        assert err <= JOINT_TOL, err
        pos = torch.from_numpy(g["sample_pos"]).to(DEV)
        vs = vols.reshape(m["batch"], 15, -1)[:, :, pos].cpu().numpy()
        assert np.abs(vs - g["volumes_samples"]).max() <= 2e-3 * g["volumes_max"].max() + 1e-7
    net.set_v2v_dtype("fp32")
    kp32 = _forward(net, img, depth)[0]
    assert not net.volume_net.program.split3 and float((kp32 - kp).abs().max()) < 1e-3


def test_v2v_stagewise_vs_oracle(net64, oracle_constants):
    """Every stage boundary of the pipeline against the oracle on the same seeded input (B=1)."""
    sd = synthetic_state_dict(False)
    const = oracle_constants(64)
    img, depth = synth.make_inputs(2024, 1, "uniform")
    taps = {}
    oj, obig, ovols = O.forward(sd, const, img, depth, taps=taps, accumulate64=True)   # platform-stable soft-argmax value
    kp, feats, vols, _ = _forward(net64, img, depth)
    # backbone + 1x1 (MIOpen): float32, different conv algorithms than oneDNN
    f_err = float((feats.float().cpu() - taps["features64"]).abs().max())
    assert f_err < 2e-3, f_err
    # V2V alone: feed the ORACLE's V2V input to the HIP program, compare logits (isolates the hand-written kernels)
    x = torch.cat([taps["feature_volume"], taps["occupancy"].unsqueeze(1)], dim=1)           # [1,33,64,64,64]
    with torch.no_grad():
        lg = net64.volume_net(x.to(DEV)).cpu()
    scale = float(taps["logits"].abs().max())
    l_err = float((lg - taps["logits"]).abs().max())
    assert l_err < 1e-4 * scale, (l_err, scale)
    assert float((kp.cpu() - oj).abs().max()) <= JOINT_TOL
    # ... and against the reference formula's own float32 evaluation on this host (einsum noise of the host included): same bar
    oj32, _ = O.integrate(taps["logits"], const.coord, softmax=True)
    assert float((kp.cpu() - oj32).abs().max()) <= JOINT_TOL
    assert float((vols.cpu() - ovols).abs().max()) <= 2e-3 * float(ovols.max())


def test_forward_17_joints_unfused_tail_vs_oracle(oracle_constants):
    """`num_joints` > 16 leaves the fused tail (one MFMA column block holds 16 output channels) and runs
    back_layers.1 / .2 / output_layer as three launches with the planar store of the direct kernel (sceneego_amd/v2v.py, the branch
    behind `if self.cout <= 16`) plus the stand-alone soft-argmax.  Whole forward with 17 joints against the oracle
    (reference network/v2v.py:155-169, utils/op.py:83-96), B=2 so that the planar store's sample stride is exercised."""
    cfg = load_config()
    cfg.model.backbone.num_joints = 17
    net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
    sd = synth.make_state_dict(net.state_dict(), seed=0)
    net.load_state_dict(sd, strict=True)
    assert tuple(sd["volume_net.output_layer.weight"].shape) == (17, 32, 1, 1, 1)
    net = net.to(DEV).eval()
    img, depth = synth.make_inputs(1717, 2, "uniform")
    taps = {}
    oj, _, ovols = O.forward(sd, oracle_constants(64), img, depth, taps=taps, accumulate64=True)
    kp, _, vols, _ = _forward(net, img, depth)
    assert tuple(kp.shape) == (2, 17, 3) and tuple(vols.shape) == (2, 17, 64, 64, 64)
    x = torch.cat([taps["feature_volume"], taps["occupancy"].unsqueeze(1)], dim=1)
    with torch.no_grad():
        lg = net.volume_net(x.to(DEV)).cpu()
    scale = float(taps["logits"].abs().max())
    l_err = float((lg - taps["logits"]).abs().max())
    j_err = float((kp.cpu() - oj).abs().max())
    print(f"17 joints (un-fused tail): logits {l_err:.2e} of max {scale:.2f}, joints {j_err:.2e} m")
    assert l_err < 1e-4 * scale, (l_err, scale)
    assert j_err <= JOINT_TOL
    assert float((vols.cpu() - ovols).abs().max()) <= 2e-3 * float(ovols.max())


def test_v2v_fork_levels_bit_identical(net64):
    """The optional fork of the skip blocks onto a side stream (V2VProgram.fork_levels; off by default: measured slower, DESIGN.md
    section 4 round 5) changes the ISSUE order only: reference network/v2v.py:104-137 reads skip_x_k first in decoder_upsample_k.  Same
    V2V input -> bit-identical logits with no fork, with every level forked and with a mixed set, eagerly and replayed 3 times."""
    x = torch.randn(2, 33, 64, 64, 64, generator=torch.Generator().manual_seed(5)).to(DEV)
    prog = net64.volume_net.program
    try:
        prog.fork_levels = ()
        with torch.no_grad():
            ref = net64.volume_net(x).clone()
        for levels in ((0, 1, 2, 3, 4), (1, 3)):
            prog.fork_levels = levels
            for _ in range(3):
                with torch.no_grad():
                    got = net64.volume_net(x)
                torch.cuda.synchronize()
                assert torch.equal(got, ref), levels
    finally:
        prog.fork_levels = None


def test_scene_volumes_branch_and_none(net64, oracle_constants):
    """Pre-voxelised input (voxel_net_depth.py:246-249) gives the same joints as the depth branch; no scene -> None."""
    const = oracle_constants(64)
    img, depth = synth.make_inputs(31, 1, "floor")
    kp_depth = _forward(net64, img, depth)[0]
    occ = O.depth_to_voxel(depth[0].numpy(), const.ray, 64, 2).unsqueeze(0)
    with torch.no_grad():
        kp_vox = net64(img.to(DEV), net64.grid_coord_proj_batch, net64.coord_volumes, scene_volumes=occ.to(DEV))[0]
        assert net64(img.to(DEV), net64.grid_coord_proj_batch, net64.coord_volumes) is None
    assert float((kp_depth - kp_vox).abs().max()) < 5e-5   # same occupancy grid -> same joints (MIOpen may pick another algo)


def test_full_size_properties_b8(net64):
    """BASELINE config 2 size (B=8): frames are independent, so a batch permutation permutes the joints exactly,
    the result is bitwise reproducible, and each frame equals its B=1 run."""
    img, depth = synth.make_inputs(8, 8, "uniform")
    kp = _forward(net64, img, depth)[0]
    kp_again = _forward(net64, img, depth)[0]
    assert float((kp - kp_again).abs().max()) < 5e-5   # MIOpen's split-K (atomic) igemm kernels are not bitwise reproducible
    perm = torch.tensor([3, 1, 7, 0, 2, 6, 5, 4])
    kp_perm = _forward(net64, img[perm], depth[perm])[0]
    assert float((kp_perm - kp[perm.to(DEV)]).abs().max()) < 5e-5
    kp_one = _forward(net64, img[5:6], depth[5:6])[0]
    assert float((kp_one - kp[5:6]).abs().max()) < 5e-5
    assert bool(torch.isfinite(kp).all())
    assert float(kp[..., 2].min()) >= 0.0 and float(kp[..., 2].max()) <= 2.0 and float(kp[..., :2].abs().max()) <= 1.0


def test_batch_larger_than_opt_batch_size(net64):
    """The reference silently requires B <= opt.batch_size (40); this build does not."""
    img, depth = synth.make_inputs(5, 2, "floor")
    img = img.repeat(21, 1, 1, 1)
    depth = depth.repeat(21, 1, 1)
    kp = _forward(net64, img, depth)[0]
    assert tuple(kp.shape) == (42, 15, 3)
    assert float((kp[0] - kp[40]).abs().max()) < 5e-5


def test_hipgraph_replay_matches_eager(net64):
    """Captured-graph replay (small-batch latency path) returns the same joints as the eager launch sequence."""
    img, depth = synth.make_inputs(91, 2, "floor")
    kp_eager = _forward(net64, img, depth)[0].clone()
    net64.enable_graphs(True)
    try:
        kp_g1 = _forward(net64, img, depth)[0].clone()
        img2, depth2 = synth.make_inputs(92, 2, "uniform")
        kp_other = _forward(net64, img2, depth2)[0].clone()
        kp_g2 = _forward(net64, img, depth)[0].clone()
    finally:
        net64.enable_graphs(False)
    assert float((kp_g1 - kp_eager).abs().max()) < 5e-5
    assert float((kp_g2 - kp_g1).abs().max()) < 5e-5
    assert float((kp_other - kp_g1).abs().max()) > 1e-3     # the replay really consumed the new inputs


def test_demo_cli_single_frame(tmp_path, golden, golden_meta):
    """BASELINE config 1 end to end through demo.py's classes: demo frame fixture + .npy depth -> pickle of [15,3] joints,
    equal to the reference golden (<= 1e-3)."""
    import os
    import pickle
    import sys
    from PIL import Image
    from conftest import GOLD, ROOT
    sys.path.insert(0, ROOT)
    import demo as demo_mod
    m = next(c for c in golden_meta["cases"] if c["name"] == "demo_b1")
    g = golden("demo_b1")
    small = np.load(os.path.join(GOLD, "demo", "img_001000_256_bgr_u8.npz"))["img"]
    # rebuild a 1280x1024 frame whose preprocessing gives exactly `small`: replicate each pixel 4x4, pad 128 columns
    big = np.repeat(np.repeat(small, 4, axis=0), 4, axis=1)
    frame = np.zeros((1024, 1280, 3), dtype=np.uint8)
    frame[:, 128:-128] = big
    img_dir, depth_dir, out_dir = tmp_path / "imgs", tmp_path / "depths", tmp_path / "out"
    img_dir.mkdir(); depth_dir.mkdir()
    Image.fromarray(frame[:, :, ::-1]).save(img_dir / "img_001000.png")        # lossless, RGB order on disk
    _, depth = synth.make_inputs(m["input_seed"], 1, m["depth_kind"])
    np.save(depth_dir / "img_001000.png.npy", depth[0].numpy().astype(np.float32))
    cfg = load_config()
    d = demo_mod.Demo(cfg, str(img_dir), str(depth_dir), weights="synthetic")
    res = d.run()
    assert len(res) == 1 and res[0]["predicted_keypoints"].shape == (15, 3)
    err = float(np.abs(res[0]["predicted_keypoints"] - g["joints"][0]).max())
    assert err <= JOINT_TOL, err
    os.makedirs(out_dir, exist_ok=True)
    with open(out_dir / "img_001000.png.pkl", "wb") as f:
        pickle.dump(res[0]["predicted_keypoints"], f)
    with open(out_dir / "img_001000.png.pkl", "rb") as f:
        assert pickle.load(f).dtype == np.float32


def test_demo_cli_exr_depth(tmp_path, golden):
    """Config 1 with the reference's own depth map: demo.py reads the PIZ EXR (sceneego_amd/exr.py) next to the frame."""
    import os
    import shutil
    import sys
    from PIL import Image
    from conftest import GOLD, ROOT
    sys.path.insert(0, ROOT)
    import demo as demo_mod
    g = golden("demo_exr_b1")
    small = np.load(os.path.join(GOLD, "demo", "img_001000_256_bgr_u8.npz"))["img"]
    frame = np.zeros((1024, 1280, 3), dtype=np.uint8)
    frame[:, 128:-128] = np.repeat(np.repeat(small, 4, axis=0), 4, axis=1)
    img_dir, depth_dir = tmp_path / "imgs", tmp_path / "depths"
    img_dir.mkdir(); depth_dir.mkdir()
    Image.fromarray(frame[:, :, ::-1]).save(img_dir / "img_001000.png")
    shutil.copy(os.path.join(GOLD, "demo", "img_001000.jpg.exr"), depth_dir / "img_001000.png.exr")
    d = demo_mod.Demo(load_config(), str(img_dir), str(depth_dir), weights="synthetic")
    res = d.run()
    err = float(np.abs(res[0]["predicted_keypoints"] - g["joints"][0]).max())
    assert err <= JOINT_TOL, err


def test_demo_to_evaluate_mpjpe_on_gpu(tmp_path, golden, golden_meta):
    """f4 on the GPU: the reference scores a run by collecting the forward's joints (test.py:42-57) and passing
    them through utils/calculate_errors.py (align_skeleton :60-91, calculate_error :22-28).  Here: demo.Demo forward on two frames (the
    demo frame with the reference's own EXR depth map, and with the seeded .npy depth map of golden `demo_b1`) -> the pickles demo.py
    writes -> evaluate.main with the REFERENCE's joints (goldens) as ground truth: MPJPE and PA-MPJPE <= 1e-3 m."""
    import os
    import pickle
    import shutil
    import sys
    from PIL import Image
    from conftest import GOLD, ROOT
    sys.path.insert(0, ROOT)
    import demo as demo_mod
    import evaluate as ev
    m = next(c for c in golden_meta["cases"] if c["name"] == "demo_b1")
    g_npy, g_exr = golden("demo_b1"), golden("demo_exr_b1")
    small = np.load(os.path.join(GOLD, "demo", "img_001000_256_bgr_u8.npz"))["img"]
    frame = np.zeros((1024, 1280, 3), dtype=np.uint8)
    frame[:, 128:-128] = np.repeat(np.repeat(small, 4, axis=0), 4, axis=1)
    img_dir, depth_dir, out_dir = tmp_path / "imgs", tmp_path / "depths", tmp_path / "out"
    img_dir.mkdir(); depth_dir.mkdir(); out_dir.mkdir()
    for name in ("img_001000.png", "img_001001.png"):
        Image.fromarray(frame[:, :, ::-1]).save(img_dir / name)
    shutil.copy(os.path.join(GOLD, "demo", "img_001000.jpg.exr"), depth_dir / "img_001000.png.exr")
    _, depth = synth.make_inputs(m["input_seed"], 1, m["depth_kind"])
    np.save(depth_dir / "img_001001.png.npy", depth[0].numpy().astype(np.float32))
    res = demo_mod.Demo(load_config(), str(img_dir), str(depth_dir), weights="synthetic").run()
    assert [os.path.basename(r["img_path"]) for r in res] == ["img_001000.png", "img_001001.png"]
    for r in res:                                                   # what demo.main writes (reference demo.py:88-97)
        with open(out_dir / (os.path.basename(r["img_path"]) + ".pkl"), "wb") as f:
            pickle.dump(r["predicted_keypoints"], f)
    with open(tmp_path / "gt.pkl", "wb") as f:
        pickle.dump({"img_001000.png": g_exr["joints"][0], "img_001001.png": g_npy["joints"][0]}, f)
    r = ev.main(["--pred_dir", str(out_dir), "--gt", str(tmp_path / "gt.pkl")])
    print(f"demo -> evaluate on the GPU: MPJPE {r['mpjpe']:.3e} m, PA-MPJPE {r['pa_mpjpe']:.3e} m over {r['frames']} frames")
    assert r["frames"] == 2 and r["mpjpe"] <= JOINT_TOL and r["pa_mpjpe"] <= JOINT_TOL, r
    assert max(r["per_joint"]) <= JOINT_TOL


def test_scene_volumes_from_dataset_side_voxeliser(net64, oracle_constants):
    """f3 (voxel_output=True): dataset/real_depth_utils.py:29-60 restated on the GPU feeds forward(scene_volumes=...);
    the voxel set is bit-identical to the oracle's and the joints equal the oracle's for the same scene volume."""
    from sceneego_amd import real_depth_utils as rdu
    c = oracle_constants(64)
    img, depth = synth.make_inputs(77, 2, "floor")
    occ = rdu.depth_map_to_voxel(c.ray, depth, 2, 64, device=DEV)
    want = torch.stack([O.depth_to_voxel_full(depth[b].numpy(), c.ray, 64, 2) for b in range(2)])
    assert torch.equal(occ.cpu(), want)
    assert torch.equal(rdu.depth_map_to_voxel(c.ray, depth[0].numpy(), 2, 64, device=DEV).cpu(), want[0])
    kp, _, _, _ = net64(img.to(DEV), net64.grid_coord_proj_batch, net64.coord_volumes, scene_volumes=occ)
    sd = synthetic_state_dict()
    oj, _, _ = O.forward(sd, c, img, None, scene_volumes=want, accumulate64=True)
    assert float((kp.cpu() - oj).abs().max()) <= JOINT_TOL
    oj32, _, _ = O.forward(sd, c, img, None, scene_volumes=want)          # the reference's float32 soft-argmax on this host: same bar
    assert float((kp.cpu() - oj32).abs().max()) <= JOINT_TOL


def test_planar3_input_layout_matches_channels_last(net64):
    """The production float32 path feeds the 7^3 layer a triplet-planar V2V input; the channels-last layout (A/B switch) runs the
    same arithmetic in the same order, so logits-derived outputs agree up to MIOpen's run-to-run noise in the backbone."""
    img, depth = synth.make_inputs(77, 2, "floor")
    assert net64.planar3_input
    kp_p, _, vol_p, _ = _forward(net64, img, depth)
    kp_p, vol_p = kp_p.clone(), vol_p.clone()
    net64.planar3_input = False
    try:
        kp_c, _, vol_c, _ = _forward(net64, img, depth)
    finally:
        net64.planar3_input = True
    assert float((kp_p - kp_c).abs().max()) < 5e-5
    assert float((vol_p - vol_c).abs().max()) <= 1e-4 * float(vol_c.max())


def test_pipelined_forward_matches_plain_forward(net64):
    """sceneego_amd.pipeline.PipelinedForward: consecutive forwards issued round-robin on two streams (replicas aliasing the
    parameters) return what the plain forward returns (to the backbone's run-to-run reproducibility), whichever replica / stream
    served the call."""
    from sceneego_amd.pipeline import PipelinedForward
    img, depth = synth.make_inputs(11, 2, "floor")
    img2, depth2 = synth.make_inputs(12, 2, "floor")
    ref = [_forward(net64, i, d) for i, d in ((img, depth), (img2, depth2))]
    torch.cuda.synchronize()
    pf = PipelinedForward(net64, 2)
    assert len(pf) == 2 and pf.nets[1] is not net64
    assert all(a.data_ptr() == b.data_ptr() for a, b in zip(pf.nets[1].parameters(), net64.parameters()))
    dev_in = [(i.to(DEV), d.to(DEV)) for i, d in ((img, depth), (img2, depth2))]
    ready = torch.cuda.Event()
    ready.record()                                  # the uploads run on the current stream: hand the pipeline their event
    outs = []
    for rep in range(2):
        for i, d in dev_in:
            outs.append(pf(i, net64.grid_coord_proj_batch, net64.coord_volumes, depth_map_batch=d, inputs_ready=ready))
    pf.synchronize()
    diffs = [(float((out[0] - ref[n % 2][0]).abs().max()), float((out[2] - ref[n % 2][2]).abs().max())) for n, (out, _) in enumerate(outs)]
    print("pipelined vs plain: max |joint diff|, max |volume diff| per call:", diffs)
    # not bitwise: the MIOpen backbone itself is not reproducible run to run (its igemm "gkgs" kernels split K over workgroups with
    # atomic adds: two plain forwards on one stream differ by ~6e-6 m in the joints, tools/diag/stream_determinism.py)
    for n, (out, done) in enumerate(outs):
        assert done.query()
        assert diffs[n][0] < 1e-4 and diffs[n][1] < 1e-4, diffs
    # a replica of an ALREADY COMPILED module packs its own program and still takes the fused 16-channel skip path
    # (the fused weights live on the packed convolution, not in a table keyed by object identity)
    for rep in pf.nets:
        prog = rep.volume_net.program
        assert prog.front_res[0][2] is not None and prog.front_res[0][2].fused is not None
    assert pf.nets[1].volume_net.program is not net64.volume_net.program
    # default ordering - the pipeline stream waits for what the caller's stream has queued (here: non-blocking uploads
    # from pinned memory and an in-place scaling kernel), and the inputs may be dropped right after the call
    pin = [(i.pin_memory(), d.pin_memory()) for i, d in ((img, depth), (img2, depth2))]
    outs = []
    for n in range(4):
        i, d = pin[n % 2]
        di = i.to(DEV, non_blocking=True).mul_(1.0)
        dd = d.to(DEV, non_blocking=True).mul_(1.0)
        outs.append(pf(di, net64.grid_coord_proj_batch, net64.coord_volumes, depth_map_batch=dd))
        del di, dd
    for n, (out, done) in enumerate(outs):
        PipelinedForward.hand_over(out, done)
        err = float((out[0] - ref[n % 2][0]).abs().max())
        assert err < 1e-4, (n, err)
    # round 4: every replica replayed as a captured hipGraph (the throughput mode of batch 1, where the eager pipeline is bound by the
    # host's launch rate).  A replayed forward returns the replica's static output tensors: read them before that replica runs again
    pf.enable_graphs(True)
    try:
        for rep in range(3):
            for n, (i, d) in enumerate(dev_in):
                out, done = pf(i, net64.grid_coord_proj_batch, net64.coord_volumes, depth_map_batch=d)
                done.synchronize()
                err = float((out[0] - ref[n][0]).abs().max())
                assert err < 1e-4, (rep, n, err)
        assert all(len(r._graphs) == 1 for r in pf.nets)
    finally:
        pf.enable_graphs(False)


@pytest.mark.parametrize("side", [128, 512])
def test_forward_other_image_sizes_vs_oracle(side, oracle_constants):
    """the reference upsamples ANY feature map to 1024 x 1024 (`nn.Upsample(size=(1024, 1024))`,
    network/voxel_net_depth.py:59-60,238) before it samples it, so an image that is not 256 x 256 is legal input: 128^2 gives a 32 x 32
    feature map (1024 texels), 512^2 a 128 x 128 one.  Until round 6 the 4-tap table was hard-wired to 64 x 64: a 512^2 image sampled the
    first quarter of its map, a 128^2 image read past its buffer.  The table is now built for the map the call has; whole forward
    against the oracle (literal Upsample + pad + grid_sample)."""
    net = _build()
    sd = synthetic_state_dict(False)
    const = oracle_constants(64)
    img = torch.from_numpy(synth.normal(900 + side, "img", (1, 3, side, side)))
    _, depth = synth.make_inputs(900 + side, 1, "floor")
    taps = {}
    oj, _, ovols = O.forward(sd, const, img, depth, taps=taps, accumulate64=True)
    kp, feats, vols, _ = _forward(net, img, depth)
    assert tuple(feats.shape[-2:]) == (side // 4, side // 4) and net._gather_max < (side // 4) ** 2
    f_err = float((feats.float().cpu() - taps["features64"]).abs().max())
    j_err = float((kp.cpu() - oj).abs().max())
    print(f"{side}x{side} image: feature map {side // 4}^2, features {f_err:.2e}, joints vs oracle {j_err:.2e} m")
    assert f_err < 2e-3 and j_err <= JOINT_TOL
    assert float((vols.cpu() - ovols).abs().max()) <= 2e-3 * float(ovols.max())
    # back to the 256 x 256 crop on the same module: the table follows the map
    img2, depth2 = synth.make_inputs(7, 1, "floor")
    oj2, _, _ = O.forward(sd, const, img2, depth2, accumulate64=True)
    assert float((_forward(net, img2, depth2)[0].cpu() - oj2).abs().max()) <= JOINT_TOL


def test_forward_fft_front_layer_matches_winograd_front_layer(monkeypatch):
    """The frequency-domain front layer (round 6, csrc/conv3d_fft7.hip) against the F(6,7) Winograd kernel it replaces in the same
    forward: SCENEEGO_FFT7=0 compiles the program without the spectra (triplet-planar input, conv3d_wino67.hip).  Joints of both
    programs agree to 1e-4 m and both meet the reference golden."""
    from sceneego_amd import _lib
    img, depth = synth.make_inputs(311, 2, "uniform")
    net = _build()
    kp_fft = _forward(net, img, depth)[0]
    assert net.volume_net.program.front0_fft is not None
    assert any(k[-1] for k in net._xbuf), "the forward did not take the planar (frequency-domain) input path"
    monkeypatch.setenv("SCENEEGO_FFT7", "0")
    net2 = _build()
    kp_w = _forward(net2, img, depth)[0]
    assert net2.volume_net.program.front0_fft is None
    d = float((kp_fft - kp_w).abs().max())
    print(f"fft7 vs wino67 front layer: joints differ by {d:.2e} m")
    assert d < 1e-4
