import numpy as np, sympy as sp
from fractions import Fraction as Fr
def cook_toom(m, r, pts):
    n = m + r - 1
    P = [sp.Rational(p.numerator, p.denominator) for p in pts]
    x = sp.symbols('x')
    AT = sp.Matrix([[p ** i for p in P] + [1 if i == m - 1 else 0] for i in range(m)])
    G = sp.Matrix([[p ** j for j in range(r)] for p in P] + [[0] * (r - 1) + [1]])
    for i, p in enumerate(P):
        Ni = sp.prod([p - q for j, q in enumerate(P) if j != i])
        G[i, :] = G[i, :] / Ni
    BT = sp.zeros(n, n)
    for i, p in enumerate(P):
        co = sp.Poly(sp.prod([x - q for j, q in enumerate(P) if j != i]), x).all_coeffs()[::-1]
        for j, c in enumerate(co): BT[i, j] = c
    co = sp.Poly(sp.prod([x - q for q in P]), x).all_coeffs()[::-1]
    for j, c in enumerate(co): BT[n - 1, j] = c
    return AT, G, BT
AT, G, BT = cook_toom(4, 3, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2)])
print("AT", AT.tolist()); print("G", G.tolist()); print("BT", BT.tolist())
ATf, Gf, BTf = [np.array(M.tolist(), dtype=np.float64) for M in (AT, G, BT)]
rng = np.random.default_rng(0); e = []
for _ in range(2000):
    g = rng.standard_normal(3); d = rng.standard_normal(6)
    ref = np.array([np.dot(g, d[i:i+3]) for i in range(4)])
    y64 = ATf @ ((Gf @ g) * (BTf @ d))
    U = (Gf @ g).astype(np.float32); V = (BTf.astype(np.float32) @ d.astype(np.float32))
    y32 = ATf.astype(np.float32) @ (U * V)
    e.append((np.abs(y64-ref).max(), np.abs(y32-ref).max()))
e = np.array(e); print("f64 err", e[:,0].max(), "f32 err max", e[:,1].max(), "mean", e[:,1].mean())
