import numpy as np
from fractions import Fraction as Fr
# Winograd F(m=2, r=7): n = m + r - 1 = 8 points (7 finite + infinity), Cook-Toom construction.
def cook_toom(m, r, pts):
    n = m + r - 1
    assert len(pts) == n - 1
    # polynomial evaluation matrices (with point at infinity as last row)
    def V(k, pts):   # n x k Vandermonde incl. infinity row
        M = [[p ** j for j in range(k)] for p in pts]
        M.append([Fr(0)] * (k - 1) + [Fr(1)])
        return M
    # A^T: m x n, G: n x r, B^T: n x n  such that y = A^T[(G g) * (B^T d)]
    # Use the transposed-Toom-Cook construction: y = A^T ((G g) .* (B^T d)) with
    # A = V(m) , G = V(r) scaled by 1/prod, B^T = inverse-Vandermonde^T of size n
    import sympy as sp
    P = [sp.Rational(p.numerator, p.denominator) for p in pts]
    x = sp.symbols('x')
    Vn = sp.Matrix([[p ** j for j in range(n)] for p in P] + [[0] * (n - 1) + [1]])
    AT = sp.Matrix([[p ** i for p in P] + [1 if i == m - 1 else 0] for i in range(m)])
    G = sp.Matrix([[p ** j for j in range(r)] for p in P] + [[0] * (r - 1) + [1]])
    # scale rows of G by 1/N_i where N_i = prod_{j != i}(p_i - p_j)
    for i, p in enumerate(P):
        Ni = sp.prod([p - q for j, q in enumerate(P) if j != i])
        G[i, :] = G[i, :] / Ni
    # B^T rows: coefficients of M_i(x) = prod_{j != i} (x - p_j), last row M(x) = prod (x - p_j)
    BT = sp.zeros(n, n)
    for i, p in enumerate(P):
        poly = sp.Poly(sp.prod([x - q for j, q in enumerate(P) if j != i]), x)
        co = poly.all_coeffs()[::-1]
        for j, c in enumerate(co):
            BT[i, j] = c
    poly = sp.Poly(sp.prod([x - q for q in P]), x)
    co = poly.all_coeffs()[::-1]
    for j, c in enumerate(co):
        BT[n - 1, j] = c
    return AT, G, BT

pts = [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2), Fr(1, 2), Fr(-1, 2)]
AT, G, BT = cook_toom(2, 7, pts)
import sympy as sp
AT_f = np.array(AT.tolist(), dtype=np.float64); G_f = np.array(G.tolist(), dtype=np.float64); BT_f = np.array(BT.tolist(), dtype=np.float64)
print("A^T", AT_f); print("G", np.round(G_f, 5)); print("B^T", BT_f)
# verify exactness in float64 and error in float32
rng = np.random.default_rng(0)
errs = []; 
for trial in range(2000):
    g = rng.standard_normal(7); d = rng.standard_normal(8)
    y_ref = np.array([np.dot(g, d[0:7]), np.dot(g, d[1:8])])
    y64 = AT_f @ ((G_f @ g) * (BT_f @ d))
    U32 = (G_f @ g).astype(np.float32); V32 = (BT_f.astype(np.float32) @ d.astype(np.float32)).astype(np.float32)
    y32 = AT_f.astype(np.float32) @ (U32 * V32)
    d32 = np.array([np.dot(g.astype(np.float32), d[0:7].astype(np.float32)), np.dot(g.astype(np.float32), d[1:8].astype(np.float32))])
    errs.append((np.abs(y64 - y_ref).max(), np.abs(y32 - y_ref).max(), np.abs(d32 - y_ref).max()))
e = np.array(errs)
print("max err f64 wino", e[:, 0].max(), " f32 wino: max", e[:, 1].max(), "mean", e[:, 1].mean(), " f32 direct: max", e[:, 2].max(), "mean", e[:, 2].mean())
