// 24-point complex DFT in registers: prime-factor (Good-Thomas) 3 x 8 form - 24 = 3 * 8 with gcd 1, so the two stages need NO
// twiddle multiplies between them: only the constants 1/sqrt 2 (radix 8) and 1/2, sqrt 3 / 2 (radix 3) appear.
//   input  n = (3 n1 + 8 n2) mod 24,  output k = (9 k1 + 16 k2) mod 24   (n1, k1 < 8; n2, k2 < 3):  n k = 3 n1 k1 + 8 n2 k2 (mod 24)
// ~290 float operations per transform.  Every index is a compile-time constant after unrolling: arrays live in registers.
// Plain C++ (host and device): tests/cpp/fft24_check.cpp compiles this header with g++ and checks it against a naive DFT.
// Used by conv3d_fft7.hip (the frequency-domain form of the 7x7x7 front layer).
#pragma once

#if defined(__HIPCC__)
#define SE_FFT_FN __host__ __device__ __forceinline__
#else
#define SE_FFT_FN inline
#endif

// INV = false: X[k] = sum_n x[n] exp(-2 pi i n k / N);  INV = true: the conjugate kernel, NOT divided by N.
template <bool INV>
SE_FFT_FN void se_fft8(float (&r)[8], float (&i)[8]) {
    constexpr float H = 0.70710678118654752440f;
    // x * (-i) forward, x * (+i) inverse
#define SE_MULJ(xr, xi, yr, yi) do { if (INV) { yr = -(xi); yi = (xr); } else { yr = (xi); yi = -(xr); } } while (0)
    const float b0r = r[0] + r[4], b0i = i[0] + i[4], b4r = r[0] - r[4], b4i = i[0] - i[4];
    const float b1r = r[1] + r[5], b1i = i[1] + i[5], b5r = r[1] - r[5], b5i = i[1] - i[5];
    const float b2r = r[2] + r[6], b2i = i[2] + i[6], b6r = r[2] - r[6], b6i = i[2] - i[6];
    const float b3r = r[3] + r[7], b3i = i[3] + i[7], b7r = r[3] - r[7], b7i = i[3] - i[7];
    // even outputs: 4-point transform of b0..b3
    {
        const float c0r = b0r + b2r, c0i = b0i + b2i, c2r = b0r - b2r, c2i = b0i - b2i;
        const float c1r = b1r + b3r, c1i = b1i + b3i, tr = b1r - b3r, ti = b1i - b3i;
        float c3r, c3i;
        SE_MULJ(tr, ti, c3r, c3i);
        r[0] = c0r + c1r; i[0] = c0i + c1i; r[4] = c0r - c1r; i[4] = c0i - c1i;
        r[2] = c2r + c3r; i[2] = c2i + c3i; r[6] = c2r - c3r; i[6] = c2i - c3i;
    }
    // odd outputs: 4-point transform of (b4, b5 w, b6 w^2, b7 w^3), w = exp(-+ 2 pi i / 8)
    {
        float d5r, d5i, d6r, d6i, d7r, d7i;
        if (INV) {
            d5r = (b5r - b5i) * H; d5i = (b5r + b5i) * H;
            d7r = (-b7r - b7i) * H; d7i = (b7r - b7i) * H;
        } else {
            d5r = (b5r + b5i) * H; d5i = (b5i - b5r) * H;
            d7r = (b7i - b7r) * H; d7i = (-b7r - b7i) * H;
        }
        SE_MULJ(b6r, b6i, d6r, d6i);
        const float e0r = b4r + d6r, e0i = b4i + d6i, e2r = b4r - d6r, e2i = b4i - d6i;
        const float e1r = d5r + d7r, e1i = d5i + d7i, tr = d5r - d7r, ti = d5i - d7i;
        float e3r, e3i;
        SE_MULJ(tr, ti, e3r, e3i);
        r[1] = e0r + e1r; i[1] = e0i + e1i; r[5] = e0r - e1r; i[5] = e0i - e1i;
        r[3] = e2r + e3r; i[3] = e2i + e3i; r[7] = e2r - e3r; i[7] = e2i - e3i;
    }
#undef SE_MULJ
}

template <bool INV>
SE_FFT_FN void se_fft24(float (&re)[24], float (&im)[24]) {
    constexpr float S3 = 0.86602540378443864676f;
    float tr[3][8], ti[3][8];
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            tr[n2][n1] = re[(3 * n1 + 8 * n2) % 24];
            ti[n2][n1] = im[(3 * n1 + 8 * n2) % 24];
        }
        se_fft8<INV>(tr[n2], ti[n2]);
    }
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
        const float ar = tr[0][k1], ai = ti[0][k1];
        const float sr = tr[1][k1] + tr[2][k1], si = ti[1][k1] + ti[2][k1];
        const float dr = (tr[1][k1] - tr[2][k1]) * S3, di = (ti[1][k1] - ti[2][k1]) * S3;
        const float mr = ar - 0.5f * sr, mi = ai - 0.5f * si;
        re[(9 * k1) % 24] = ar + sr;
        im[(9 * k1) % 24] = ai + si;
        // forward: X1 = m - i d, X2 = m + i d; inverse: swapped
        const float pr = mr + di, pi = mi - dr, qr = mr - di, qi = mi + dr;
        re[(9 * k1 + 16) % 24] = INV ? qr : pr;
        im[(9 * k1 + 16) % 24] = INV ? qi : pi;
        re[(9 * k1 + 32) % 24] = INV ? pr : qr;
        im[(9 * k1 + 32) % 24] = INV ? pi : qi;
    }
}
