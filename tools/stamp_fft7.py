#!/usr/bin/env python3
"""Per-phase cycle sums of the frequency-domain front layer's passes 1 and 2 (csrc/conv3d_fft7.hip built with -DSE_FFT7_STAMP:
tools/build_variant.sh stamp conv3d_fft7 -DSE_FFT7_STAMP -> sceneego_amd/libse_stamp.so).  Every wave sums the s_memtime cycles it
spends in each phase of its loop; this prints the median / max over waves per phase and the share of the wave's loop time."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SCENEEGO_HIP_LIB", os.path.join(ROOT, "sceneego_amd", "libse_stamp.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sceneego_amd import _lib      # noqa: E402
import bench_fft7                  # noqa: E402

PHASES = {0: ["loop top", "issue next tile's loads", "stage 2 (x, in LDS)", "barrier", "stage 3 (LDS reads, y transform, stores)", "barrier",
              "wait next tile + stage 1", "barrier"],
          1: ["loop top", "issue loads g+1", "matrix phase", "D -> LDS", "barrier", "O -> global", "wait group g+1 + LDS transpose", "barrier"]}


def main():
    B, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 64
    conv, bn = bench_fft7.make_layer()
    conv, bn = conv.to("cuda:0"), bn.to("cuda:0")
    pc, hf = bench_fft7.pack(conv, bn)
    x = torch.randn(B, 33, dim, dim, dim, device="cuda:0")
    ws = torch.empty((_lib.conv3d_k7_fft_workspace_elems(B, dim, 33),), device="cuda:0")
    out = torch.empty((B, 16 * dim ** 3), device="cuda:0")
    for _ in range(5):
        _lib.conv3d_k7_fft(x, hf, pc.b, out, B, dim, 33, 16, _lib.EPI_RELU | _lib.OUT_QUAD, ws)
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.se_debug_fft7_stamps.restype = ctypes.c_int
    lib.se_debug_fft7_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    occ = (ctypes.c_int * 3)()
    if hasattr(lib, "se_debug_fft7_occupancy"):
        lib.se_debug_fft7_occupancy.argtypes = [ctypes.c_void_p]
        lib.se_debug_fft7_occupancy(occ)
    print("resident workgroups per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor): pass 1 %d, pass 2 %d, pass 3 %d" % tuple(occ))
    buf = np.zeros((3, 4096 * 8, 16), dtype=np.uint64)
    rc = lib.se_debug_fft7_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    assert rc == 0, rc
    census_only = os.environ.get("SE_FFT7_CENSUS_ONLY") == "1"
    for k, name in ((0, "pass 1 (fft7_fwd_kernel)"), (1, "pass 2 (fft7_gemm_kernel)"), (2, "pass 3 (fft7_inv_kernel)")):
        raw = buf[k][buf[k][:, 15] > 0]
        # residency census: per CU (XCC id, SE / SH / CU id of HW_ID) the largest number of workgroups alive at one time (100 MHz clock)
        wpw = 4
        first = raw[::wpw] if len(raw) % wpw == 0 else raw
        cu = (first[:, 13] >> np.uint64(32)) << np.uint64(16) | (first[:, 13] & np.uint64(0xFF00))
        worst = {}
        for c in np.unique(cu):
            ev = sorted([(int(t0), 1) for t0 in first[cu == c, 14]] + [(int(t1), -1) for t1 in first[cu == c, 15]])
            n = m = 0
            for _, d in ev:
                n += d
                m = max(m, n)
            worst[m] = worst.get(m, 0) + 1
        print(f"{name}: CUs seen {len(np.unique(cu))}; CUs by max workgroups alive together: {worst}")
        if census_only or k == 2:
            continue
        a = buf[k].astype(np.float64)[:, :13]
        a = a[a.sum(1) > 0]
        tot = a.sum(1)
        print(f"{name}: {len(a)} waves, loop cycles per wave: median {np.median(tot):.0f}, max {tot.max():.0f}")
        for i, ph in enumerate(PHASES[k]):
            print(f"   {ph:38s} median {np.median(a[:, i]):9.0f}  max {a[:, i].max():9.0f}  share {a[:, i].sum() / tot.sum() * 100:5.1f} %")


if __name__ == "__main__":
    main()
