// The 7x7x7 front layer of V2V in the FREQUENCY domain (round 6).
//
// Stands in for Basic3DBlock(33 -> 16, k = 7): Conv3d(pad 3) + BatchNorm3d + ReLU (reference network/v2v.py:8-18, built at :147) for
// volumes with dim % 16 == 0.  The Winograd form (conv3d_wino67.hip) executes 12/42 of the direct products and sat at 0.76 of the f32 MFMA
// peak for three rounds; a 7-tap filter is where the convolution theorem wins: per 16^3 output tile a 24^3 transform turns the 343 taps
// into ONE complex product per frequency, input and output channel:
//
//   pass 1  fft7_fwd_kernel   per (tile, input channel): 24^3 real tile (origin 16 t - 4, zeros outside the volume) -> 3-D DFT, half
//                             spectrum along kz: 7488 complex, written in blocks of 16 frequencies  X[fb][tile][c][16]
//   pass 2  fft7_gemm_kernel  per frequency: Y[tile][co] = sum_c X[tile][c] H[co][c] (complex) as a REAL GEMM on v_mfma_f32_16x16x4_f32:
//                             [Yr Yi](M x 32) = [Xr Xi](M x 66) . [[Hr Hi]; [-Hi Hr]](66 x 32), M = all tiles of the batch (512 at B = 8)
//                             -> 16.7 GFLOP per launch against 217 G executed by the F(6,7) kernel (759.6 G direct)
//   pass 3  fft7_inv_kernel   per (tile, 4 output channels): inverse transform, the 16^3 valid outputs, + bias, ReLU, quad-planar or
//                             channels-last store
//
// All three passes are HBM-bound: at B = 8 the spectra are X = 1.01 GB, Y = 0.49 GB (float32 complex), so a launch moves
// 0.28 (+ halo) + 1.01 | 1.01 + 0.06 + 0.49 | 0.49 + 0.13 = 3.5 GB.  Index maps and the traffic model: tools/fft7_model.py (numpy, pass by
// pass; checked against a direct convolution).  The in-register 24-point transform: fft24.h (prime-factor 3 x 8, no twiddles).
//
// Tile algebra (per axis): inputs x[i] = in[16 t - 4 + i], i < 24; valid outputs j < 16: out[16 t + j] = sum_d w[d] x[j + d + 1]
// = (x (*) h)[j] circular with h[23 - d] = w[d] - j + d + 1 <= 22: no wrap-around reaches a valid output.
// Frequency order: f = (ky * 13 + kz) * 24 + kx, kz in [0, 13) (half spectrum along z).
#include "common.h"

#include "fft24.h"

namespace {

constexpr int FP = 24;                   // tile points per axis
constexpr int FV = 16;                   // valid outputs per axis
constexpr int FKZ = 13;                  // kept kz
constexpr int FROW = FKZ * FP;           // 312 = transforms per pass = frequencies per ky
constexpr int FNF = FP * FROW;           // 7488 frequencies
constexpr int FNFB = FNF / 16;           // 468 frequency blocks
constexpr int F_THREADS = 320;           // 5 waves: 312 transforms per stage
constexpr int F_RS = 52;                 // LDS row stride in floats: 24 complex + 4 (16-byte aligned rows, conflict-free b128 row reads)
constexpr int F1_LDS_FLOATS = FKZ * FP * F_RS;      // pass 1: [kz 13][y 24] rows  = 64,896 B
constexpr int F3_LDS_FLOATS = FKZ * FV * F_RS;      // pass 3: [kz 13][y 16] rows  = 43,264 B
constexpr int G_KSTEPS = 17;             // 66 real k (33 channels x re / im) in MFMA steps of 4 (2 pad)
constexpr long long G_HF_PER_FREQ = G_KSTEPS * 2 * 64;      // A-fragment floats per frequency: [step][cout tile 2][lane]

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// weight spectra: H[f][co][c] = DFT of h (h[23 - d] = w[d] per axis) x BatchNorm scale / (2 * 24^3), float64 arithmetic, written as the
// MFMA A fragments of pass 2:  hf[f][step][nt][lane] = A[n = 16 nt + lane % 16][k = kmap(step, lane / 16)] with
//   n = 2 co + (0: real row, 1: imaginary row),  k = 2 c + (0: real column, 1: imaginary column), k >= 2 cin: 0
//   kmap(step < 16, kg) = 16 (step / 4) + 4 kg + step % 4   (a lane's 16-byte LDS read covers its four steps),  kmap(16, kg) = 64 + kg
//   A[(co, re)][(c, re)] = Hr   A[(co, re)][(c, im)] = -Hi   A[(co, im)][(c, re)] = Hi   A[(co, im)][(c, im)] = Hr
// The 1/2 pays for pass 1's un-normalised split of the paired real transform, the 1/24^3 for the un-normalised inverse.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fft7_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                        const float* __restrict__ var, float eps, float* __restrict__ hf, int cin,
                                                        long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    long long r = t >> 6;
    const int nt = (int)(r & 1); r >>= 1;
    const int step = (int)(r % G_KSTEPS);
    const int f = (int)(r / G_KSTEPS);
    const int kg = lane >> 4;
    const int n = 16 * nt + (lane & 15);
    const int k = step < 16 ? 16 * (step >> 2) + 4 * kg + (step & 3) : 64 + kg;
    const int co = n >> 1, c = k >> 1;
    float v = 0.f;
    if (c < cin) {
        const int ky = f / FROW, rem = f - ky * FROW, kz = rem / FP, kx = rem - kz * FP;
        const double two_pi_24 = 6.283185307179586476925286766559 / 24.0;
        const float* wp = w + ((long long)co * cin + c) * 343;
        double hr = 0.0, hi = 0.0;
        for (int dz = 0; dz < 7; ++dz)
            for (int dy = 0; dy < 7; ++dy)
                for (int dx = 0; dx < 7; ++dx) {
                    const int ph = (kz * (23 - dz) + ky * (23 - dy) + kx * (23 - dx)) % 24;
                    const double wv = (double)wp[(dz * 7 + dy) * 7 + dx];
                    hr += wv * cos(two_pi_24 * ph);
                    hi -= wv * sin(two_pi_24 * ph);
                }
        const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;       // the fold of se_conv3d_pack_f32
        const double s = (double)sc / (2.0 * 13824.0);
        hr *= s; hi *= s;
        const bool n_im = n & 1, k_im = k & 1;
        v = (float)(n_im == k_im ? hr : (n_im ? hi : -hi));
    }
    hf[t] = v;
}

// ------------------------------------------------------------------------------------------------
// pass 1: forward transform of one (tile, channel) per loop trip.  `in` is PLANAR [B][C][D^3].
//   stage 1 (288 threads = (y, x pair)): the 24 z values of two x-adjacent columns are ONE complex column (re = even x, im = odd x - an
//            8-byte load per z is the complex input as it lies in memory); transform, split into the two Hermitian half spectra
//            A[kz] = U[kz] + conj U[24 - kz], B[kz] = -i (U[kz] - conj U[24 - kz])  (x 1/2 folded into H), kz <= 12  -> LDS [kz][y][x]
//   stage 2 (312 threads = rows (kz, y)): transform along x in place
//   stage 3 (312 threads = columns (kz, kx)): transform along y, output ky goes straight to global memory: frequency ky * 312 + thread,
//            i.e. a wave stores 64 consecutive frequencies (512 contiguous bytes of 128-byte blocks X[fb][tile][c][16])
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(F_THREADS) void fft7_fwd_kernel(const float* __restrict__ in, float* __restrict__ X, int C, int D, int T,
                                                             int M, int n_units) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const size_t plane = (size_t)D * D * D;
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int c = u % C, m = u / C;
        int r = m;
        const int tx = r % T; r /= T;
        const int ty = r % T; r /= T;
        const int tz = r % T;
        const int b = r / T;
        if (t < 288) {
            const int y = t / 12, xp = t - y * 12;
            const int gy = 16 * ty - 4 + y, gx = 16 * tx - 4 + 2 * xp;
            const bool okyx = (unsigned)gy < (unsigned)D && (unsigned)gx < (unsigned)D;      // D even: the pair is inside or outside together
            // raw buffer loads: the plane of (sample, channel) is the buffer, the lane part of the address is one 32-bit offset whose
            // bit 31 marks a column outside the volume (reads zero), the z slab is the scalar offset, a slab outside the volume is read
            // through a zero-record descriptor - no branch, no per-load address arithmetic in vector registers
            const float* pl = in + ((size_t)b * C + c) * plane;
            const unsigned voff = okyx ? (unsigned)((gy * D + gx) * 4) : 0x80000000u;
            float re[24], im[24];
#pragma unroll
            for (int z = 0; z < 24; ++z) {
                const int gz = 16 * tz - 4 + z;
                const bool okz = (unsigned)gz < (unsigned)D;
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pl), 0, okz ? (int)(plane * 4) : 0, 0x00020000);
                const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, okz ? gz * D * D * 4 : 0, 0));
                re[z] = v.x; im[z] = v.y;
            }
            se_fft24<false>(re, im);
            float* dst = lds + y * F_RS + 4 * xp;
#pragma unroll
            for (int kz = 0; kz < FKZ; ++kz) {
                const int km = (24 - kz) % 24;
                *reinterpret_cast<f32x4*>(dst + kz * (FP * F_RS)) = (f32x4){re[kz] + re[km], im[kz] - im[km], im[kz] + im[km], re[km] - re[kz]};
            }
        }
        __syncthreads();
        if (t < FROW) {
            float* row = lds + t * F_RS;
            float re[24], im[24];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * q);
                re[2 * q] = v.x; im[2 * q] = v.y; re[2 * q + 1] = v.z; im[2 * q + 1] = v.w;
            }
            se_fft24<false>(re, im);
#pragma unroll
            for (int q = 0; q < 12; ++q) *reinterpret_cast<f32x4*>(row + 4 * q) = (f32x4){re[2 * q], im[2 * q], re[2 * q + 1], im[2 * q + 1]};
        }
        __syncthreads();
        if (t < FROW) {
            const int kz = t / FP, kx = t - kz * FP;
            const float* col = lds + kz * (FP * F_RS) + 2 * kx;
            float re[24], im[24];
#pragma unroll
            for (int y = 0; y < 24; ++y) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(col + y * F_RS);
                re[y] = v.x; im[y] = v.y;
            }
            se_fft24<false>(re, im);
            // frequency f = ky * 312 + t in blocks of 16: 312 = 16 * 19.5, so even ky = 2 e starts at block 39 e with lane part t, odd
            // ky at block 39 e + 19 with lane part t + 8: two 32-bit lane offsets, the rest of the address is uniform
            const size_t blk = (size_t)M * C * 32;                          // floats per frequency block
            float* base = X + ((size_t)m * C + c) * 32;
            const unsigned off_e = (unsigned)((t >> 4) * blk + (t & 15) * 2), off_o = (unsigned)(((t + 8) >> 4) * blk + ((t + 8) & 15) * 2);
#pragma unroll
            for (int ky = 0; ky < 24; ++ky) {
                float* dst = base + (size_t)(39 * (ky >> 1) + ((ky & 1) ? 19 : 0)) * blk + ((ky & 1) ? off_o : off_e);
                *reinterpret_cast<f32x2*>(dst) = (f32x2){re[ky], im[ky]};
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: per-frequency complex GEMM over the channels as a real GEMM on the matrix cores.
// Workgroup (512 threads) = one block of 16 frequencies x a range of 16-tile groups; wave w owns frequencies 2 w, 2 w + 1 of the block and
// keeps their A fragments (2 x 34 registers) for the whole range.  Per 16-tile group:
//   the X block [16 tiles][C][16 f] complex is ONE contiguous run (67.6 KB at C = 33): 16-byte pieces, coalesced, prefetched into
//   registers one group ahead and transposed into LDS  I[f][tile][k = 2 c + re/im]  (rows of 72 floats: conflict-free b128 reads of
//   the B operand; k = 66..71 stay zero); each wave: 2 x 17 k-steps x 2 cout tiles = 68 MFMAs; D fragments (lane = tile, 2 couts
//   complex) -> LDS  O[tile][co][f]  -> the Y block [16 tiles][16 co][16 f] complex, again one contiguous 32 KB run, 16-byte pieces.
// ------------------------------------------------------------------------------------------------
constexpr int G_THREADS = 512;
constexpr int G_MS = 72;                          // floats per (f, tile) row of I
constexpr int G_FS = 16 * G_MS + 4;               // floats per frequency of I (the 4: ds_write_b64 of the transpose spread over banks)
constexpr int G_I_FLOATS = 16 * G_FS;             // 18,496
constexpr int G_OC = 36;                          // floats per (tile, co) row of O: 16 f complex + 4
constexpr int G_OM = 16 * G_OC + 4;               // floats per tile of O
constexpr int G_O_FLOATS = 16 * G_OM;             // 9,280
constexpr int G_LDS_BYTES = (G_I_FLOATS + G_O_FLOATS) * 4;      // 111,104

template <int C>
__global__ __launch_bounds__(G_THREADS) void fft7_gemm_kernel(const float* __restrict__ X, const float* __restrict__ hf,
                                                              float* __restrict__ Y, int M, int groups_per_wg) {
    static_assert(2 * C <= 66, "k layout: 16 b128-fed steps + one scalar step");
    constexpr int PIECES = 16 * C * 8;            // 16-byte pieces of an X block
    constexpr int NLOAD = (PIECES + G_THREADS - 1) / G_THREADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* I = lds;
    float* O = lds + G_I_FLOATS;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fb = blockIdx.x;
    const int n_groups = (M + 15) >> 4;
    const int g0 = blockIdx.y * groups_per_wg, g1 = min(g0 + groups_per_wg, n_groups);
    if (g0 >= g1) return;

    // A fragments of this wave's two frequencies
    float a[2][G_KSTEPS][2];
#pragma unroll
    for (int ff = 0; ff < 2; ++ff)
#pragma unroll
        for (int s = 0; s < G_KSTEPS; ++s)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                a[ff][s][nt] = hf[((size_t)(fb * 16 + 2 * wave + ff) * G_KSTEPS + s) * 128 + nt * 64 + lane];
    // zero the k padding of every (f, tile) row once (never overwritten): k = 2 C .. 71
    if (t < 256) {
        float* row = I + (t >> 4) * G_FS + (t & 15) * G_MS;
#pragma unroll
        for (int k = 2 * C; k < G_MS; k += 2) *reinterpret_cast<f32x2*>(row + k) = (f32x2){0.f, 0.f};
    }
    // where this thread's pieces go in I: piece p = (tile m, channel c, frequency pair fp) -> rows (2 fp, m) and (2 fp + 1, m), column 2 c
    int ioff[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
        const int p = min(t + j * G_THREADS, PIECES - 1);
        const int m = p / (C * 8), c = (p >> 3) % C, fp = p & 7;
        ioff[j] = (2 * fp) * G_FS + m * G_MS + 2 * c;
    }
    f32x4 pre[NLOAD];
    auto prefetch = [&](int g) {
        const float* src = X + ((size_t)fb * M + (size_t)g * 16) * C * 32;
        const int valid = (min(M - g * 16, 16)) * C * 8;        // pieces of tiles that exist
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            const int p = t + j * G_THREADS;
            pre[j] = p < valid ? *reinterpret_cast<const f32x4*>(src + (size_t)p * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    prefetch(g0);
    const int mcol = lane & 15, kg = lane >> 4;
    for (int g = g0; g < g1; ++g) {
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            if (t + j * G_THREADS < PIECES) {
                *reinterpret_cast<f32x2*>(I + ioff[j]) = (f32x2){pre[j].x, pre[j].y};
                *reinterpret_cast<f32x2*>(I + ioff[j] + G_FS) = (f32x2){pre[j].z, pre[j].w};
            }
        }
        __syncthreads();
        if (g + 1 < g1) prefetch(g + 1);
        f32x4 acc[2][2];
#pragma unroll
        for (int ff = 0; ff < 2; ++ff) {
            acc[ff][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[ff][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* brow = I + (2 * wave + ff) * G_FS + mcol * G_MS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(brow + 16 * i + 4 * kg);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[ff][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ff][4 * i + j][0], bv[j], acc[ff][0], 0, 0, 0);
                    acc[ff][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ff][4 * i + j][1], bv[j], acc[ff][1], 0, 0, 0);
                }
            }
            const float bl = brow[64 + kg];
            acc[ff][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ff][16][0], bl, acc[ff][0], 0, 0, 0);
            acc[ff][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ff][16][1], bl, acc[ff][1], 0, 0, 0);
        }
        // D fragment: lane (tile mcol, kg) holds n = 16 nt + 4 kg + j -> couts 8 nt + 2 kg, + 1, complex
#pragma unroll
        for (int ff = 0; ff < 2; ++ff)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float* o = O + mcol * G_OM + (8 * nt + 2 * kg) * G_OC + 2 * (2 * wave + ff);
                *reinterpret_cast<f32x2*>(o) = (f32x2){acc[ff][nt].x, acc[ff][nt].y};
                *reinterpret_cast<f32x2*>(o + G_OC) = (f32x2){acc[ff][nt].z, acc[ff][nt].w};
            }
        __syncthreads();
        {
            float* dst = Y + ((size_t)fb * M + (size_t)g * 16) * 16 * 32;
            const int valid = min(M - g * 16, 16) * 128;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = t + j * G_THREADS;
                const int m = p >> 7, co = (p >> 3) & 15, fp = p & 7;
                const f32x4 v = *reinterpret_cast<const f32x4*>(O + m * G_OM + co * G_OC + 4 * fp);
                if (p < valid) *reinterpret_cast<f32x4*>(dst + (size_t)p * 4) = v;
            }
        }
        // the next trip's I writes need this trip's MFMA reads done (the barrier above), its O writes need these O reads done: they sit
        // behind the next trip's first barrier
    }
}

// ------------------------------------------------------------------------------------------------
// pass 3: inverse transform of (tile, 4 output channels) per loop trip.  Per channel:
//   stage 1 (312 threads = columns (kz, kx)): 24 ky straight from global memory (a wave reads 64 consecutive frequencies), inverse
//            transform along y, the 16 valid y -> LDS [kz][y][kx]
//   stage 2 (208 threads = rows (kz, y < 16)): inverse along x in place, 16 valid x written back
//   stage 3 (256 threads = (y, x)): the 13 kz of the half spectrum, Hermitian extension, inverse along z: the real parts of z < 16 stay
//            in 16 registers
// then (y, x) owns 16 z x 4 channels: + bias, ReLU, one 16-byte store per z - 16 lanes write 256 contiguous bytes of a quad-planar
// output row [B][4][D^3][4] (OUTQ), or 16 of every 64 bytes of a channels-last record [B][D^3][16].
// ------------------------------------------------------------------------------------------------
template <bool OUTQ>
__global__ __launch_bounds__(F_THREADS, 3) void fft7_inv_kernel(const float* __restrict__ Y, const float* __restrict__ bias,
                                                             float* __restrict__ out, int D, int T, int M, int n_units, int relu) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int q = u & 3, m = u >> 2;
        int r = m;
        const int tx = r % T; r /= T;
        const int ty = r % T; r /= T;
        const int tz = r % T;
        const int b = r / T;
        float stash[4][16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = 4 * q + j;
            if (t < FROW) {
                const int kz = t / FP, kx = t - kz * FP;
                float re[24], im[24];
                const size_t blk = (size_t)M * 16 * 32;                      // floats per frequency block (as in pass 1)
                const float* base = Y + ((size_t)m * 16 + co) * 32;
                const unsigned off_e = (unsigned)((t >> 4) * blk + (t & 15) * 2), off_o = (unsigned)(((t + 8) >> 4) * blk + ((t + 8) & 15) * 2);
#pragma unroll
                for (int ky = 0; ky < 24; ++ky) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(base + (size_t)(39 * (ky >> 1) + ((ky & 1) ? 19 : 0)) * blk + ((ky & 1) ? off_o : off_e));
                    re[ky] = v.x; im[ky] = v.y;
                }
                se_fft24<true>(re, im);
                float* col = lds + kz * (FV * F_RS) + 2 * kx;
#pragma unroll
                for (int y = 0; y < FV; ++y) *reinterpret_cast<f32x2*>(col + y * F_RS) = (f32x2){re[y], im[y]};
            }
            __syncthreads();
            if (t < FKZ * FV) {
                float* row = lds + t * F_RS;
                float re[24], im[24];
#pragma unroll
                for (int p = 0; p < 12; ++p) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * p);
                    re[2 * p] = v.x; im[2 * p] = v.y; re[2 * p + 1] = v.z; im[2 * p + 1] = v.w;
                }
                se_fft24<true>(re, im);
#pragma unroll
                for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4*>(row + 4 * p) = (f32x4){re[2 * p], im[2 * p], re[2 * p + 1], im[2 * p + 1]};
            }
            __syncthreads();
            if (t < 256) {
                const int y = t >> 4, x = t & 15;
                const float* col = lds + y * F_RS + 2 * x;
                float re[24], im[24];
#pragma unroll
                for (int kz = 0; kz < FKZ; ++kz) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(col + kz * (FV * F_RS));
                    re[kz] = v.x; im[kz] = v.y;
                }
#pragma unroll
                for (int kz = FKZ; kz < 24; ++kz) { re[kz] = re[24 - kz]; im[kz] = -im[24 - kz]; }
                se_fft24<true>(re, im);
#pragma unroll
                for (int z = 0; z < FV; ++z) stash[j][z] = re[z];
            }
            __syncthreads();
        }
        if (t < 256) {
            const int y = t >> 4, x = t & 15;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * q);
            const int gy = 16 * ty + y, gx = 16 * tx + x;
#pragma unroll
            for (int z = 0; z < FV; ++z) {
                const int gz = 16 * tz + z;
                f32x4 v = (f32x4){stash[0][z], stash[1][z], stash[2][z], stash[3][z]} + bv;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                const size_t vox = ((size_t)gz * D + gy) * D + gx;
                float* o = OUTQ ? out + (((size_t)b * 4 + q) * D * D * D + vox) * 4 : out + ((size_t)b * D * D * D + vox) * 16 + 4 * q;
                *reinterpret_cast<f32x4*>(o) = v;
            }
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" long long se_conv3d_k7_fft_packed_elems(int cin, int cout) {
    if (cin != 33 || cout != 16) return -1;
    return (long long)FNF * G_HF_PER_FREQ;
}

extern "C" int se_conv3d_k7_fft_pack_f32(const float* w, const float* gamma, const float* var, float eps, float* hfrag, int cout,
                                         int cin, void* stream) {
    const long long total = se_conv3d_k7_fft_packed_elems(cin, cout);
    if (total <= 0 || !w || !hfrag || (gamma != nullptr) != (var != nullptr)) return SE_ERR_BAD_ARG;
    hipLaunchKernelGGL(fft7_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, se_stream(stream), w, gamma, var, eps, hfrag,
                       cin, total);
    SE_CHECK_LAUNCH();
    return 0;
}

// floats of workspace for `batch` samples in ONE chunk (spectra X and Y of every tile)
extern "C" long long se_conv3d_k7_fft_workspace_elems(int batch, int dim, int cin) {
    if (batch <= 0 || dim < 16 || (dim & 15) || cin != 33) return -1;
    const long long T = dim / 16, M = (long long)batch * T * T * T;
    return M * (cin + 16) * FNF * 2;
}

extern "C" int se_conv3d_k7_fft_f32(const float* in, const float* hfrag, const float* bpack, float* out, int batch, int dim, int cin,
                                    int cout, int flags, float* workspace, long long workspace_elems, void* stream) {
    if (batch <= 0 || dim < 16 || (dim & 15) || cin != 33 || cout != 16 || !in || !hfrag || !bpack || !out || !workspace) return SE_ERR_BAD_ARG;
    if (flags & ~(SE_EPI_RELU | SE_OUT_QUAD)) return SE_ERR_BAD_ARG;
    const long long per_sample = se_conv3d_k7_fft_workspace_elems(1, dim, cin);
    int chunk = (int)(workspace_elems / per_sample < batch ? workspace_elems / per_sample : batch);
    if (chunk <= 0) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    const int T = dim / 16, cus = se_num_cus();
    const size_t vox = (size_t)dim * dim * dim;
    SE_ENSURE_LDS(fft7_fwd_kernel, F1_LDS_FLOATS * 4);
    SE_ENSURE_LDS(fft7_gemm_kernel<33>, G_LDS_BYTES);
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = batch - b0 < chunk ? batch - b0 : chunk;
        const int M = nb * T * T * T;
        float* X = workspace;
        float* Yb = workspace + (size_t)M * cin * FNF * 2;
        {
            const int units = M * cin;
            const int grid = units < 2 * cus * 4 ? units : 2 * cus * 4;
            hipLaunchKernelGGL(fft7_fwd_kernel, dim3(grid), dim3(F_THREADS), F1_LDS_FLOATS * 4, s, in + (size_t)b0 * cin * vox, X, cin, dim, T,
                               M, units);
            SE_CHECK_LAUNCH();
        }
        {
            const int n_groups = (M + 15) / 16;
            int split = (4 * cus + FNFB - 1) / FNFB;                    // ~4 workgroups per CU over the launch
            if (split > n_groups) split = n_groups;
            const int gpw = (n_groups + split - 1) / split;
            hipLaunchKernelGGL(fft7_gemm_kernel<33>, dim3(FNFB, (n_groups + gpw - 1) / gpw), dim3(G_THREADS), G_LDS_BYTES, s, X, hfrag, Yb, M, gpw);
            SE_CHECK_LAUNCH();
        }
        {
            const int units = M * 4;
            const int grid = units < 3 * cus * 4 ? units : 3 * cus * 4;
            const int relu = (flags & SE_EPI_RELU) ? 1 : 0;
            float* o = out + (size_t)b0 * 16 * vox;
            if (flags & SE_OUT_QUAD)
                hipLaunchKernelGGL(fft7_inv_kernel<true>, dim3(grid), dim3(F_THREADS), F3_LDS_FLOATS * 4, s, Yb, bpack, o, dim, T, M, units, relu);
            else
                hipLaunchKernelGGL(fft7_inv_kernel<false>, dim3(grid), dim3(F_THREADS), F3_LDS_FLOATS * 4, s, Yb, bpack, o, dim, T, M, units, relu);
            SE_CHECK_LAUNCH();
        }
    }
    return 0;
}
