#!/usr/bin/env python3
"""numpy model of the frequency-domain form of the 7x7x7 front layer (csrc/conv3d_fft7.hip), pass by pass.

Design study + index reference for the three HIP passes (reference semantics: Conv3d(k=7, pad 3) + BatchNorm3d + ReLU,
/root/reference network/v2v.py:8-18,147).  Not imported by the product; tests/ do not use it either (they compare the HIP path with
torch's conv3d) - it exists so that every index map of the kernels has a few lines of numpy that say the same thing:

  tile          24^3 input voxels at origin 16 t - 4 per axis (zero outside the volume) -> 16^3 valid outputs at 16 t
  circular form y[j] = sum_d w[d] x[j + d + 1]  (j in [0,16), no wrap)  =  (x (*) h)[j]  with  h[23 - d] = w[d]
  spectrum      half along kz (0..12); frequency index f = (ky * 13 + kz) * 24 + kx  (7488 per tile and channel)
  pass 1        z real FFT of two x-adjacent columns as one complex FFT + split  ->  x FFT  ->  y FFT  ->  X[f]
  pass 2        Y[f][tile][co] = sum_c X[f][tile][c] H[f][co][c]   (complex; 1/24^3 and the BatchNorm scale folded into H)
  pass 3        y inverse FFT -> x inverse FFT -> z complex-to-real from the half spectrum -> + bias, ReLU

Run: python tools/fft7_model.py   (checks the model against a direct convolution and prints the error)
"""
import numpy as np

P = 24          # tile points per axis
V = 16          # valid outputs per axis
KZ = P // 2 + 1  # 13
NF = P * KZ * P  # 7488


def weight_spectrum(w):
    """w [co][ci][7][7][7] (BN scale folded) -> H [NF][co][ci] complex128, scaled by 1/P^3."""
    co, ci = w.shape[:2]
    h = np.zeros((co, ci, P, P, P))
    for dz in range(7):
        for dy in range(7):
            for dx in range(7):
                h[:, :, 23 - dz, 23 - dy, 23 - dx] = w[:, :, dz, dy, dx]
    Hf = np.fft.fftn(h, axes=(2, 3, 4)) / P ** 3          # [co][ci][kz][ky][kx]
    Hf = Hf[:, :, :KZ]                                     # half along z
    # f = (ky * 13 + kz) * 24 + kx
    return np.ascontiguousarray(Hf.transpose(3, 2, 4, 0, 1)).reshape(NF, co, ci)


def fft24(v, inverse=False):
    """the in-register 24-point transform (the kernels use the prime-factor 3 x 8 form: same values)."""
    return np.fft.ifft(v, axis=-1) * P if inverse else np.fft.fft(v, axis=-1)


def pass1_tile(xt):
    """xt [24 z][24 y][24 x] real -> X [NF] complex, through the same stages as the kernel."""
    # stage 1: z transform of column pairs (x even + i * x odd), split into the two Hermitian half spectra
    u = xt[:, :, 0::2] + 1j * xt[:, :, 1::2]               # [z][y][xp]
    U = fft24(np.moveaxis(u, 0, -1))                        # [y][xp][kz 24]
    Ur = np.conj(U[..., (-np.arange(P)) % P])               # conj(U[24 - k])
    A = 0.5 * (U + Ur)[..., :KZ]                            # spectrum of the even column
    Bc = -0.5j * (U - Ur)[..., :KZ]                         # spectrum of the odd column
    W = np.empty((KZ, P, P), complex)                       # LDS image [kz][y][x]
    W[:, :, 0::2] = np.moveaxis(A, -1, 0)
    W[:, :, 1::2] = np.moveaxis(Bc, -1, 0)
    W = fft24(W)                                            # stage 2: x transform, rows (kz, y)        -> [kz][y][kx]
    W = np.moveaxis(fft24(np.moveaxis(W, 1, -1)), -1, 1)    # stage 3: y transform, columns (kz, kx)    -> [kz][ky][kx]
    return np.ascontiguousarray(W.transpose(1, 0, 2)).reshape(NF)   # f = (ky * 13 + kz) * 24 + kx


def pass3_tile(Y):
    """Y [NF] complex (half spectrum of a real 24^3 array times P^3 ... the 1/P^3 sits in H) -> the 16^3 valid outputs."""
    W = Y.reshape(P, KZ, P).transpose(1, 0, 2).copy()       # [kz][ky][kx]
    W = np.moveaxis(fft24(np.moveaxis(W, 1, -1), inverse=True), -1, 1)[:, :V]    # stage 1: y inverse, keep 16   -> [kz][y][kx]
    W = fft24(W, inverse=True)[:, :, :V]                                          # stage 2: x inverse, keep 16   -> [kz][y][x]
    full = np.empty((P, V, V), complex)                     # stage 3: Hermitian extension along z, inverse, real part
    full[:KZ] = W
    full[KZ:] = np.conj(W[P - np.arange(KZ, P)])
    out = fft24(np.moveaxis(full, 0, -1), inverse=True).real[..., :V]             # [y][x][z]
    return np.moveaxis(out, -1, 0)                          # [z][y][x]


def conv7_fft(x, w, bias):
    """x [ci][D][D][D], w [co][ci][7][7][7], bias [co] -> relu(conv + bias) [co][D][D][D] through the three passes."""
    ci, D = x.shape[0], x.shape[1]
    co = w.shape[0]
    T = D // V
    H = weight_spectrum(w)
    xp = np.zeros((ci, D + 8, D + 8, D + 8))
    xp[:, 4:4 + D, 4:4 + D, 4:4 + D] = x                   # tile origin 16 t - 4  ->  index 16 t in the padded array
    out = np.empty((co, D, D, D))
    for tz in range(T):
        for ty in range(T):
            for tx in range(T):
                X = np.stack([pass1_tile(xp[c, 16 * tz:16 * tz + P, 16 * ty:16 * ty + P, 16 * tx:16 * tx + P]) for c in range(ci)], 1)
                Y = np.einsum("fc,foc->fo", X, H)            # pass 2 for this tile (M = 1)
                for o in range(co):
                    out[o, 16 * tz:16 * tz + V, 16 * ty:16 * ty + V, 16 * tx:16 * tx + V] = pass3_tile(Y[:, o]) + bias[o]
    return np.maximum(out, 0.0)


def conv7_direct(x, w, bias):
    ci, D = x.shape[0], x.shape[1]
    xp = np.zeros((ci, D + 6, D + 6, D + 6))
    xp[:, 3:3 + D, 3:3 + D, 3:3 + D] = x
    out = np.zeros((w.shape[0], D, D, D))
    for dz in range(7):
        for dy in range(7):
            for dx in range(7):
                out += np.einsum("oc,czyx->ozyx", w[:, :, dz, dy, dx], xp[:, dz:dz + D, dy:dy + D, dx:dx + D])
    return np.maximum(out + bias[:, None, None, None], 0.0)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    D, ci, co = 32, 3, 2
    x = rng.standard_normal((ci, D, D, D))
    w = rng.standard_normal((co, ci, 7, 7, 7)) * 0.05
    b = rng.standard_normal(co)
    ref = conv7_direct(x, w, b)
    got = conv7_fft(x, w, b)
    print("max|fft - direct| =", np.abs(ref - got).max(), " max|ref| =", np.abs(ref).max())
    # traffic / work model at B = 8, 64^3
    B, Dd, Ci, Co = 8, 64, 33, 16
    M = B * (Dd // V) ** 3
    print("tiles M = %d, X = %.3f GB, Y = %.3f GB, H (expanded real A-fragments, K padded to 68) = %.1f MB" %
          (M, M * Ci * NF * 8 / 1e9, M * Co * NF * 8 / 1e9, NF * 68 * 32 * 4 / 1e6))
    print("pass-2 MFMA FLOP = %.1f G (direct conv %.1f G)" % (2.0 * M * 68 * 32 * NF / 1e9, 2.0 * B * Dd ** 3 * 343 * Ci * Co / 1e9))
