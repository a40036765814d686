import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
CALIB = os.path.join(ROOT, "sceneego_amd", "calibration", "fisheye.calibration_05_08.json")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU: skip instead of erroring out of every test
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))
    return load


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLD, "META.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def config():
    from sceneego_amd import load_config
    return load_config()


_SD_CACHE = {}


def case_inputs(m):
    """(image, depth) of a golden case (META.json entry), rebuilt exactly as tools/make_golden.py built them."""
    from sceneego_amd import synth
    kind = m["depth_kind"]
    img, depth = synth.make_inputs(m["input_seed"], m["batch"], "floor" if kind == "demo_exr" else kind)
    if m["name"].startswith("demo"):        # BASELINE config 1: derived fixture of the reference's demo frame
        from sceneego_amd.preprocess import normalize_u8
        img = normalize_u8(np.load(os.path.join(GOLD, "demo", "img_001000_256_bgr_u8.npz"))["img"])[None]
    if kind == "demo_exr":                  # ... and the reference's own depth map, through sceneego_amd/exr.py
        from sceneego_amd.preprocess import load_depth, prepare_depth
        depth = prepare_depth(load_depth(os.path.join(GOLD, "demo", "img_001000.jpg.exr")))[None]
    return img, depth


def synthetic_state_dict(with_intersection=False, seed=0):
    """Synthetic weights for the (33|65)-channel network; cached per session (46 M parameters)."""
    from sceneego_amd import load_config, synth
    from sceneego_amd.voxel_net_depth import VoxelNetwork_depth
    key = (with_intersection, seed)
    if key not in _SD_CACHE:
        cfg = load_config()
        cfg.model.with_intersection = with_intersection
        net = VoxelNetwork_depth(cfg, device="cpu", verbose=False)
        _SD_CACHE[key] = synth.make_state_dict(net.state_dict(), seed=seed)
    return _SD_CACHE[key]


@pytest.fixture(scope="session")
def state_dict():
    return synthetic_state_dict(False)


@pytest.fixture(scope="session")
def oracle_constants():
    from oracle import sceneego_oracle as O
    cache = {}

    def get(G=64):
        if G not in cache:
            cache[G] = O.Constants(CALIB, G=G)
        return cache[G]
    return get
