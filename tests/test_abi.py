"""CPU: the C-ABI shared library loads and exports every symbol include/sceneego_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from sceneego_amd import _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sceneego_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef SE_DEVTOOLS.*?#endif", "", text, flags=re.S)      # development-build-only entry points
    return sorted(set(re.findall(r"\b(se_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = _declared_symbols()
    assert "se_conv3d_f32" in syms and "se_voxelize_f64" in syms and len(syms) >= 10


def test_library_exports_every_declared_symbol():
    assert os.path.isfile(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/sceneego_hip.h but not exported"


def test_python_binding_covers_the_header():
    assert sorted(_lib.SIGNATURES.keys()) == _declared_symbols()
    lib = _lib.load()
    assert lib.se_abi_version() == _lib.ABI_VERSION


def test_production_library_has_no_debug_entry_points():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in _lib.DEVTOOLS_SIGNATURES:
        assert not hasattr(lib, s), f"{s} must only exist in --devtools builds"


def test_pure_host_entry_points():
    lib = _lib.load()
    assert lib.se_conv3d_f32_algo(64, 32, 32, 3) in (1, 2) and lib.se_conv3d_f32_algo(64, 33, 16, 7) == 7
    assert lib.se_conv3d_f32_algo(8, 128, 128, 3) == 0 and lib.se_conv3d_f32_algo(64, 32, 15, 1) == 0
    # the 2-D Winograd family (2) is reported exactly for the shapes its launcher takes: 32-bit offsets inside one sample
    assert lib.se_conv3d_f32_algo(128, 32, 32, 3) == 2 and lib.se_conv3d_f32_algo(256, 32, 32, 3) != 2
    assert lib.se_conv3d_f32_algo(256, 8, 32, 3) != 2 and lib.se_conv3d_f32_algo(64, 36, 32, 3) != 2
    # which kernel a launch runs on: 3 = F(4,3) x F(4,3) ping-pong at the 64^3 / 32^3 levels, 2 = F(4,3) x F(2,3) at 16^3; the last
    # argument carries the launch's layout flags: the ping-pong kernel takes a QUAD-planar input or < 32 channels-last channels, every
    # octet-planar launch belongs to the F(4,3) x F(2,3) kernel
    IN_Q, IN_OCT = _lib.IN_QUAD, _lib.IN_OCTET
    assert lib.se_conv3d_f32_variant(8, 64, 32, 32, 3, IN_Q) == 3 and lib.se_conv3d_f32_variant(8, 16, 128, 128, 3, IN_OCT) == 2
    assert lib.se_conv3d_f32_variant(8, 64, 32, 32, 3, IN_OCT) == 2 and lib.se_conv3d_f32_variant(8, 16, 128, 128, 3, IN_Q) == 2
    assert lib.se_conv3d_f32_variant(8, 64, 32, 32, 3, 0) == 2 and lib.se_conv3d_f32_variant(8, 64, 16, 32, 3, 0) == 3
    assert lib.se_conv3d_f32_variant(8, 64, 33, 16, 7, 0) == 7 and lib.se_conv3d_f32_variant(1, 64, 32, 32, 3, IN_Q | _lib.OUT_QUAD) == 3
    assert lib.se_conv3d_f32_variant(4, 32, 64, 64, 3, IN_Q) == 3 and lib.se_conv3d_f32_variant(2, 32, 64, 64, 3, IN_Q) == 2
    # <= 4096 voxels in the batch (16^3 at batch 1): the plain call of a 2-D Winograd shape is served by the in-workgroup split-K kernel
    assert lib.se_conv3d_f32_variant(1, 16, 128, 128, 3, 0) == 0 and lib.se_conv3d_f32_variant(2, 16, 128, 128, 3, 0) == 2
    assert lib.se_conv3d_f32_variant(1, 16, 128, 128, 3, IN_OCT) == 2        # ... a call with octet-planar forms keeps the 2-D kernel
    # soft-argmax partial records: 8 floats per (row, chunk); 256 chunks per row at batch 1 (15 rows), 32 from batch 8 (120 rows) on
    lib.se_softargmax3d_scratch_elems.restype = ctypes.c_longlong
    assert lib.se_softargmax3d_scratch_elems(15) == 15 * 256 * 8 and lib.se_softargmax3d_scratch_elems(30) == 30 * 128 * 8
    assert lib.se_softargmax3d_scratch_elems(60) == 60 * 64 * 8 and lib.se_softargmax3d_scratch_elems(120) == 120 * 32 * 8
    # packed weight sizes: taps * cin_pad/16 * ceil(cout/16) * 256 floats
    assert lib.se_conv3d_packed_elems(32, 32, 3, 0) == 27 * 2 * 2 * 256 + 2 * 9 * 4 * 2 * 256 + 2 * 9 * 6 * 2 * 256 + 4 * 24 * 3 * 2 * 128 \
        + 8 * 9 * 3 * 2 * 256   # + F(2,3), F(4,3), F(4,3)xF(2,3), F(4,3)xF(4,3) (section I: 8 four-channel chunks of 55,296 B)
    assert lib.se_conv3d_packed_elems(15, 32, 1, 0) == 1 * 2 * 1 * 256
    assert lib.se_conv3d_packed_elems(16, 48, 7, 0) == 343 * 3 * 256 + 12 * 86 * 256 + 12 * 13 * 8 * 256 + 16 * 13 * 10 * 64 * 3 + 16 * 37 * 64 * 12   # sections A, B, D, F (F(4,7): 3-channel chunks), H (F(6,7))
    assert lib.se_conv3d_packed_elems(64, 128, 2, 1) == 8 * 8 * 4 * 256
    assert lib.se_softargmax3d_scratch_elems(30) > 0


def test_ops_fail_loudly_without_gpu_tensor():
    import torch
    from sceneego_amd import op
    with pytest.raises(_lib.HipExtensionError):
        op.integrate_tensor_3d_with_coordinates(torch.zeros(1, 2, 4, 4, 4), torch.zeros(1, 4, 4, 4, 3))
