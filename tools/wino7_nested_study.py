"""Float32 error of a SECOND Winograd axis for the 7x7x7 front layer: the layer's arithmetic emulated in numpy
float32 for the current 1-D form F(6,7) along z and for nested 2-D forms along (z, y), with the output transform applied per channel
chunk ("item") or once per tile, against a float64 evaluation.  Layer statistics of the production shape: 33 input channels, 7 dx taps
on the MFMA k lanes, He-scaled weights, N(0,1) inputs; errors are quoted relative to max|y| like the kernel tests (bound 2e-5).

    python tools/wino7_nested_study.py [--tiles 40]
"""
import argparse
import contextlib
import importlib.util
import io
import os
from fractions import Fraction as Fr

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("w27", os.path.join(ROOT, "tools", "wino27_matrices.py"))
w27 = importlib.util.module_from_spec(spec)
with contextlib.redirect_stdout(io.StringIO()):
    spec.loader.exec_module(w27)


def pts(pairs):
    p = [Fr(0)]
    for q in pairs:
        p += [q, -q]
    return p


def mats(m, pairs):
    AT, G, BT = w27.cook_toom(m, 7, pts(pairs))
    f = lambda M: np.array(M.tolist(), dtype=np.float64)
    return f(AT), f(G), f(BT)


FORMS = {
    "F(6,7)": mats(6, [Fr(1), Fr(3, 4), Fr(3, 2), Fr(1, 3), Fr(5, 2)]),        # tools/wino67_matrices.py
    "F(4,7)": mats(4, [Fr(1), Fr(2), Fr(1, 2), Fr(3, 4)]),                     # tools/wino47_matrices.py
    "F(2,7)": mats(2, [Fr(1), Fr(2), Fr(1, 2)]),                               # tools/wino27_matrices.py
}
f32 = np.float32


def run(form_z, form_y, chunk, per_item, n_tiles, seed=0, cin=33):
    """One (z, y) output tile per trial at a fixed x; the 7 dx taps and the channels are the contraction the MFMA k lanes carry.
    form_y None = direct along y (the 7 dy taps join the contraction).  Returns max / mean |error| over max|y|."""
    ATz, Gz, BTz = FORMS[form_z]
    mz, nz = ATz.shape
    if form_y:
        ATy, Gy, BTy = FORMS[form_y]
        my, ny = ATy.shape
    else:
        my, ny = 1, 7
    rng = np.random.default_rng(seed)
    std = (2.0 / (cin * 343)) ** 0.5
    errs, scale = [], 0.0
    for _ in range(n_tiles):
        g = rng.standard_normal((cin, 7, 7, 7)) * std                  # [c][dx][kz][ky]
        d = rng.standard_normal((cin, 7, nz, ny))                      # [c][dx][z][y] (the dx-shifted columns, independent samples)
        # float64 reference
        ref = np.zeros((mz, my))
        for oz in range(mz):
            for oy in range(my):
                ref[oz, oy] = np.einsum("cxzy,cxzy->", g, d[:, :, oz:oz + 7, oy:oy + 7])
        # float32 Winograd
        U = np.einsum("pk,cxkl->cxpl", Gz, g)                          # weights are transformed in float64 at pack time, stored float32
        U = (np.einsum("ql,cxpl->cxpq", Gy, U) if form_y else U).astype(f32)
        V = np.einsum("pz,cxzy->cxpy", BTz.astype(f32), d.astype(f32)).astype(f32)
        if form_y:
            V = np.einsum("qy,cxpy->cxpq", BTy.astype(f32), V).astype(f32)
        out = np.zeros((mz, my), dtype=f32)
        acc = None
        hyb = None
        for c0 in range(0, cin, chunk):
            for c in range(c0, min(c0 + chunk, cin)):
                for x in range(7):
                    if form_y:
                        term = U[c, x] * V[c, x]                       # [pz][py]
                        acc = term if acc is None else (acc + term).astype(f32)
                    else:
                        for ky in range(7):                            # direct along y: dy joins the accumulation chain
                            term = U[c, x, :, ky] * V[c, x, :, ky]
                            acc = term if acc is None else (acc + term).astype(f32)
            if per_item == "z":           # hybrid: A^T along z per item (in-wave), the xi_y domain sums persist, A^T along y once per tile
                o = (ATz.astype(f32) @ acc.reshape(nz, -1)).astype(f32)
                hyb = o if hyb is None else (hyb + o).astype(f32)
                acc = None
            elif per_item:
                o = (ATz.astype(f32) @ acc.reshape(nz, -1)).astype(f32)
                if form_y:
                    o = (o @ ATy.astype(f32).T).astype(f32)
                out = (out + o.reshape(mz, my)).astype(f32)
                acc = None
        if per_item == "z":
            out = (hyb @ ATy.astype(f32).T).astype(f32)
        elif not per_item:
            o = (ATz.astype(f32) @ acc.reshape(nz, -1)).astype(f32)
            if form_y:
                o = (o @ ATy.astype(f32).T).astype(f32)
            out = o.reshape(mz, my)
        errs.append(np.abs(out.astype(np.float64) - ref).max())
        scale = max(scale, np.abs(ref).max())
    e = np.array(errs)
    return e.max() / scale, e.mean() / scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=40)
    a = ap.parse_args()
    rows = [("F(6,7) z, direct y   (production)", "F(6,7)", None, 3, True),
            ("F(6,7) z, direct y, A^T per tile", "F(6,7)", None, 3, False),
            ("F(6,7) z x F(2,7) y, A^T per item", "F(6,7)", "F(2,7)", 2, True),
            ("F(6,7) z x F(2,7) y, A^T per tile", "F(6,7)", "F(2,7)", 2, False),
            ("F(6,7) z x F(2,7) y, A^T_z per item, A^T_y per tile", "F(6,7)", "F(2,7)", 2, "z"),
            ("F(4,7) z x F(4,7) y, A^T per item", "F(4,7)", "F(4,7)", 2, True),
            ("F(4,7) z x F(4,7) y, A^T per tile", "F(4,7)", "F(4,7)", 2, False),
            ("F(6,7) z x F(4,7) y, A^T per item", "F(6,7)", "F(4,7)", 1, True),
            ("F(6,7) z x F(4,7) y, A^T per tile", "F(6,7)", "F(4,7)", 1, False),
            ("F(4,7) z x F(2,7) y, A^T per tile", "F(4,7)", "F(2,7)", 2, False)]
    print(f"float32 error / max|y| over {a.tiles} random tiles (33 channels x 7 dx x 7 x 7 taps; kernel-test bound: 2e-5)")
    for name, fz, fy, ch, per in rows:
        mx, mn = run(fz, fy, ch, per, a.tiles)
        print(f"  {name:52s} chunk {ch}: max {mx:.2e}  mean {mn:.2e}")


if __name__ == "__main__":
    main()
