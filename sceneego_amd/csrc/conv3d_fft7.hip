// The 7x7x7 front layer of V2V in the FREQUENCY domain (round 6).
//
// Stands in for Basic3DBlock(33 -> 16, k = 7): Conv3d(pad 3) + BatchNorm3d + ReLU (reference network/v2v.py:8-18, built at :147) for
// volumes with dim % 16 == 0.  The Winograd form (conv3d_wino67.hip) executes 12/42 of the direct products and sat at 0.76 of the f32 MFMA
// peak for three rounds; a 7-tap filter is where the convolution theorem wins: per 16^3 output tile a 24^3 transform turns the 343 taps
// into ONE complex product per frequency, input and output channel:
//
//   pass 1  fft7_fwd_kernel   per (tile, input channel): 24^3 real tile (origin 16 t - 4, zeros outside the volume) -> 3-D DFT, half
//                             spectrum along kz: 7488 complex, written in blocks of 16 frequencies  X[tile / 16][fb][tile % 16][c][16]
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
// What bounds the passes (round-6 measurements, profiles/r06_fft7_experiments.txt; tools: tools/stamp_fft7.py = per-phase cycle stamps and a
// residency census, tools/fft7_attr.sh = knock-out builds): the vector-memory operations a CU can have in flight.  A wave spends a third of
// its loop in ISSUING its loads and another third in issuing its stores (the queue is full: ~10 B/clk/CU), so the levers were the bytes per
// operation (16-byte accesses through lane-pair exchanges), whole 128-byte segments (64-byte ones, two units per cache line written from two
// XCDs: pass 1 0.42 -> 0.47 ms), non-temporal moves of the once-read spectra in pass 2, loops ROTATED so that no loaded register is
// carried around them (hipcc otherwise waits with vmcnt(0): it cannot order loop-carried loads against younger stores) and
// unconditional loads / stores in pass 2 (same reason).  Five-wave workgroups (320 threads = one transform per thread and stage) share a
// CU only below 128 registers (census); four-wave forms with a prefetched tile and two workgroups per CU were built and measured slower
// (the 312 transforms of a stage become 256 + 56: a second round per wave costs more than the overlap returns) - kept as template forms.
//
// Tile algebra (per axis): inputs x[i] = in[16 t - 4 + i], i < 24; valid outputs j < 16: out[16 t + j] = sum_d w[d] x[j + d + 1]
// = (x (*) h)[j] circular with h[23 - d] = w[d] - j + d + 1 <= 22: no wrap-around reaches a valid output.
// Frequency order: f = (ky * 13 + kz) * 24 + kx, kz in [0, 13) (half spectrum along z); 312 = 8 * 39 frequencies per ky.
#include "common.h"

#include <stdlib.h>

#include "fft24.h"

namespace {

constexpr int FP = 24;                   // tile points per axis
constexpr int FV = 16;                   // valid outputs per axis
constexpr int FKZ = 13;                  // kept kz
constexpr int FROW = FKZ * FP;           // 312 = transforms per stage = frequencies per ky
constexpr int FNF = FP * FROW;           // 7488 frequencies
constexpr int FB = 16;                   // frequencies per block: 128-byte segments (64-byte ones - two units' segments per cache line, written
                                         // from two XCDs - cost pass 1 0.42 -> 0.47 ms and pass 3 0.15 -> 0.175: profiles/r06_fft7_experiments.txt)
constexpr int FNFB = FNF / FB;           // 468 frequency blocks
constexpr int FB2KY = 2 * FROW / FB;     // 39 blocks per PAIR of ky (312 = 19.5 blocks: an odd ky starts in the middle of block 19)
constexpr int F_RS = 52;                 // LDS row stride in floats: 24 complex + 4 (16-byte aligned rows, conflict-free b128 row reads)
constexpr int F1_LDS_FLOATS = FKZ * FP * F_RS;      // pass 1: [kz 13][y 24] rows  = 64,896 B
constexpr int F3_LDS_FLOATS = FKZ * FV * F_RS;      // pass 3: [kz 13][y 16] rows  = 43,264 B
constexpr int G_KSTEPS = 17;             // 66 real k (33 channels x re / im) in MFMA steps of 4 (2 pad)
constexpr long long G_HF_PER_FREQ = G_KSTEPS * 2 * 64;      // A-fragment floats per frequency: [step][cout tile 2][lane]

typedef float f32x2 __attribute__((ext_vector_type(2)));

// attribution builds (tools/build_variant.sh <name> conv3d_fft7 -DSE_FFT7_EXP=<bits>; results are wrong by construction):
// 1 pass 1 without its global stores, 4 without transform arithmetic (passes 1 and 3), 16 pass 3 without its stores, 32 pass 2 without
// MFMAs, 64 pass 2 without stores
#ifndef SE_FFT7_EXP
#define SE_FFT7_EXP 0
#endif

// census builds (-DSE_FFT7_CENSUS, registers as in production): a wave only records where (XCC, HW_ID) and when (100 MHz clock) it ran
#if defined(SE_FFT7_CENSUS) && !defined(SE_FFT7_STAMP)
#define FFT7_STAMP_WAVES (4096 * 8)
__device__ unsigned long long g_fft7_stamp[3][FFT7_STAMP_WAVES][16];
#define FFT7_STAMP_DECL const unsigned long long census_t0_ = __builtin_amdgcn_s_memrealtime()
#define FFT7_MARK(i)
#define FFT7_FLUSH(k)                                                                                                                 \
    do {                                                                                                                              \
        const int w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) + blockIdx.y * gridDim.x * (blockDim.x >> 6);              \
        if ((threadIdx.x & 63) == 0 && w_ < FFT7_STAMP_WAVES) {                                                                       \
            g_fft7_stamp[k][w_][13] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |                        \
                                      (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);                                            \
            g_fft7_stamp[k][w_][14] = census_t0_;                                                                                     \
            g_fft7_stamp[k][w_][15] = __builtin_amdgcn_s_memrealtime();                                                               \
        }                                                                                                                             \
    } while (0)
#elif defined(SE_FFT7_STAMP)
// cycle-stamp builds (tools/build_variant.sh stamp conv3d_fft7 -DSE_FFT7_STAMP; tools/stamp_fft7.py): every wave sums the s_memtime
// cycles it spends in each phase of its loop; 13 sums + place + start / end time per wave in a device array (se_debug_fft7_stamps)
#define FFT7_STAMP_WAVES (4096 * 8)
__device__ unsigned long long g_fft7_stamp[3][FFT7_STAMP_WAVES][16];
struct Fft7Stamp {
    unsigned long long last, sum[16];
    __device__ void start() { for (int i = 0; i < 16; ++i) sum[i] = 0; sum[14] = __builtin_amdgcn_s_memrealtime(); last = __builtin_amdgcn_s_memtime(); }
    __device__ void mark(int i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); sum[i] += now - last; last = now; }
    __device__ void flush(int kernel) {
        sum[13] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        sum[15] = __builtin_amdgcn_s_memrealtime();
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) + blockIdx.y * gridDim.x * (blockDim.x >> 6);
        if ((threadIdx.x & 63) == 0 && w < FFT7_STAMP_WAVES) for (int i = 0; i < 16; ++i) g_fft7_stamp[kernel][w][i] = sum[i];
    }
};
#define FFT7_STAMP_DECL Fft7Stamp st_; st_.start()
#define FFT7_MARK(i) st_.mark(i)
#define FFT7_FLUSH(k) st_.flush(k)
#else
#define FFT7_STAMP_DECL
#define FFT7_MARK(i)
#define FFT7_FLUSH(k)
#endif

// the spectra are written once and read once, 1.5 GB per call at B = 8: which of those streams move with non-temporal loads / stores
// (bits: 1 pass-1 stores of X, 2 pass-2 loads of X, 4 pass-2 stores of Y, 8 pass-3 loads of Y)
#ifndef SE_FFT7_NT
#define SE_FFT7_NT 6      // measured (profiles/r06_fft7_experiments.txt): pass 2 0.337 -> 0.301 ms with 6; bits 1 and 8 buy nothing
#endif
template <int BIT>
__device__ __forceinline__ void st_stream(float* p, f32x4 v) {
    if (SE_FFT7_NT & BIT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}
template <int BIT>
__device__ __forceinline__ f32x4 ld_stream(const float* p) {
    if (SE_FFT7_NT & BIT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return *reinterpret_cast<const f32x4*>(p);
}

// value of the neighbouring lane (lane ^ 1): one DPP move (quad_perm [1,0,3,2]), no LDS
__device__ __forceinline__ float lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// The N = 312 (or 288) items of a transform stage on NT threads.  NT >= N (320): item = thread, one round.  NT = 256: round 0 = item t;
// round 1 = the remaining N - 256 items, dealt evenly to the four waves (PER = (N - 256) / 4 lanes each, an even number: lane pairs stay
// item pairs for the DPP exchanges).  -1 = no item.
template <int N, int NT>
__device__ __forceinline__ int stage_item(int t, int round) {
    if constexpr (NT >= N) {
        return round == 0 && t < N ? t : -1;
    } else {
        static_assert(NT == 256 && (N - 256) % 8 == 0, "second round: whole lane pairs per wave");
        constexpr int PER = (N - 256) / 4;
        if (round == 0) return t;
        return (t & 63) < PER ? 256 + (t >> 6) * PER + (t & 63) : -1;
    }
}

// two complex values per 16-byte vector-memory access: the lane pair (even, odd) owns adjacent items; the even lane moves both items'
// values of step 2 e, the odd lane both of step 2 e + 1, and one complex value crosses the pair each way (DPP).  unpack: v = what this
// lane loaded -> (re, im) of its own item for steps 2 e and 2 e + 1.
__device__ __forceinline__ void pair_unpack(const f32x4 v, bool odd, float& r0, float& i0, float& r1, float& i1) {
    // even lane: v = step 2e of (own .xy, partner .zw); odd lane: v = step 2e + 1 of (partner .xy, own .zw)
    const float sx = odd ? v.x : v.z, sy = odd ? v.y : v.w;
    const float rx = lane_xor1(sx), ry = lane_xor1(sy);
    r0 = odd ? rx : v.x; i0 = odd ? ry : v.y;
    r1 = odd ? v.z : rx; i1 = odd ? v.w : ry;
}
__device__ __forceinline__ f32x4 pair_pack(bool odd, float r0, float i0, float r1, float i1) {
    const float sx = odd ? r0 : r1, sy = odd ? i0 : i1;
    const float rx = lane_xor1(sx), ry = lane_xor1(sy);
    return odd ? (f32x4){rx, ry, r1, i1} : (f32x4){r0, i0, rx, ry};
}

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
//   stage 1 (288 items = (y, x pair)): the 24 z values of two x-adjacent columns are ONE complex column (re = even x, im = odd x);
//            transform, split into the two Hermitian half spectra A[kz] = U[kz] + conj U[24 - kz], B[kz] = -i (U[kz] - conj U[24 - kz])
//            (x 1/2 folded into H), kz <= 12  -> LDS [kz][y][x]
//   stage 2 (312 items = rows (kz, y)): transform along x in place
//   stage 3 (312 items = columns (kz, kx)): transform along y, output ky goes straight to global memory: frequency ky * 312 + item,
//            i.e. a wave stores 64 consecutive frequencies (16 bytes per lane: pair_pack)
// The tile: raw buffer loads - the plane of (sample, channel) is the buffer, the lane part of the address is one 32-bit offset whose bit 31
// marks a column outside the volume (reads zero), the z slab is the scalar offset, a slab outside the volume is read through a zero-record
// descriptor: no branch, no per-load address arithmetic in vector registers.  16-byte loads: the lane pair (xp even, xp odd) shares an
// aligned x quad; the even lane loads it for z, the odd lane for z + 1 (pair_unpack).  The loads of unit i + 1 are issued right behind
// the unpacking of unit i and fly under its three stages.
// ------------------------------------------------------------------------------------------------
// NT threads: 320 = one item per thread and stage (5 waves: two workgroups share a CU only below 128 registers - no room for a
// prefetched tile), 256 = 4 waves, a second 14-lane round per stage, a prefetched tile (PF) in registers and two workgroups per CU.
// Measured at B = 8 (profiles/r06_fft7_experiments.txt): <320, no prefetch> 0.42 ms, <256, prefetch> 0.52 ms, <320, prefetch> (one
// workgroup per CU) 0.45 ms: the pass is bound by the vector-memory operations a CU can have in flight (cycle stamps: a third of a
// wave's time goes into ISSUING its 24 loads, another third into issuing its 24 stores), and the second round of the 4-wave form costs
// more than its overlap returns.
template <int NT, bool PF>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : (PF ? 3 : 4)) void fft7_fwd_kernel(const float* __restrict__ in, float* __restrict__ X, int C, int D, int T,
                                                                                    int M, int n_units) {
    constexpr int ROUNDS = NT >= FROW ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const size_t plane = (size_t)D * D * D;
    const bool odd = t & 1;
    const int it1[2] = {stage_item<288, NT>(t, 0), stage_item<288, NT>(t, 1)};        // stage-1 items of this thread: column pairs (y, xp)
    f32x4 raw[ROUNDS][12];
    auto issue_loads = [&](int u, int round) {
        const int c = u % C, m = u / C;
        int r = m;
        const int tx = r % T; r /= T;
        const int ty = r % T; r /= T;
        const int tz = r % T;
        const int b = r / T;
        const int item = it1[round];
        const int y = max(item, 0) / 12, xp = max(item, 0) - y * 12;
        const int gy = 16 * ty - 4 + y, gx = 16 * tx - 4 + 2 * xp;
        const bool okyx = item >= 0 && (unsigned)gy < (unsigned)D && (unsigned)gx < (unsigned)D;      // D even: the pair is inside or outside together
        const float* pl = in + ((size_t)b * C + c) * plane;
        const unsigned voff = okyx ? (unsigned)((gy * D + (gx & ~3)) * 4) + (odd ? (unsigned)(D * D * 4) : 0u) : 0x80000000u;   // gx & ~3: 16 tx - 4 + 4 (xp / 2)
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int gz0 = 16 * tz - 4 + 2 * e;                                                    // even: gz0, gz0 + 1 inside or outside together
            const bool okz = (unsigned)gz0 < (unsigned)D;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pl), 0, okz ? (int)(plane * 4) : 0, 0x00020000);
            raw[round][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, okz ? gz0 * D * D * 4 : 0, 0));
        }
    };
    // stage 1 of a unit: unpack the tile that was loaded into `raw`, transform along z, split, write the LDS image
    auto stage1 = [&]() {
#pragma unroll
        for (int round = 0; round < ROUNDS; ++round) {
            float re[24], im[24];
#pragma unroll
            for (int e = 0; e < 12; ++e) pair_unpack(raw[round][e], odd, re[2 * e], im[2 * e], re[2 * e + 1], im[2 * e + 1]);
            const int item = it1[round];
            if (item >= 0) {
                const int y = item / 12, xp = item - y * 12;
                if (!(SE_FFT7_EXP & 4)) se_fft24<false>(re, im);
                float* dst = lds + y * F_RS + 4 * xp;
#pragma unroll
                for (int kz = 0; kz < FKZ; ++kz) {
                    const int km = (24 - kz) % 24;
                    *reinterpret_cast<f32x4*>(dst + kz * (FP * F_RS)) = (f32x4){re[kz] + re[km], im[kz] - im[km], im[kz] + im[km], re[km] - re[kz]};
                }
            }
        }
    };
    // The loop is ROTATED: a trip runs stages 2 and 3 of unit u and stage 1 of the NEXT unit, whose tile is requested at the top of the
    // trip and unpacked at its end - the loads fly under two transform stages and no loaded register is carried around the loop (with
    // loop-carried prefetch registers hipcc's wait-count pass lost the order of the outstanding operations and put an
    // `s_waitcnt vmcnt(0)` in front of every unpack: the prefetch bought nothing - disassembly, round 6).
    // XCD-aware walk (SE_FFT7_XCD_WALK, speed only - workgroup b runs on XCD b % 8 by round-robin dispatch): a (sample, channel) PLANE
    // belongs to one XCD, whose resident workgroups walk its 64 tiles together - the 24^3 / 16^3 halo a tile shares with its neighbours
    // (2.4 of the 3.4 x the tile loads beyond its core) is then a hit in that XCD's L2 instead of the Infinity Cache.  Unit = m * C + c.
#ifndef SE_FFT7_XCD_WALK
#define SE_FFT7_XCD_WALK 1
#endif
    const int TT = T * T * T;
    const bool walk = SE_FFT7_XCD_WALK && (gridDim.x & 7) == 0;
    auto unit_of = [&](int i) {          // the i-th unit of this workgroup, or -1
        if (!walk) {
            const int u = (int)blockIdx.x + i * (int)gridDim.x;
            return u < n_units ? u : -1;
        }
        // granule dealt to an XCD: a whole plane when the planes divide evenly over the 8 XCDs (batch % 8 == 0), else a z slab of T x T
        // tiles (33 planes at batch 1 would leave one XCD with 5 planes against 4)
        const int G = (n_units / TT) % 8 == 0 ? TT : T * T;
        const int j = ((int)blockIdx.x >> 3) + i * ((int)gridDim.x >> 3);
        const int lt = (((int)blockIdx.x & 7) + 8 * (j / G)) * G + j % G;       // (plane = sample * C + channel) * TT + tile in plane
        if (lt >= n_units) return -1;
        const int pl = lt / TT;
        return ((pl / C) * TT + lt % TT) * C + pl % C;
    };
    if (unit_of(0) < 0) return;
    auto load_tile = [&](int u) {
#pragma unroll
        for (int round = 0; round < ROUNDS; ++round) issue_loads(u, round);
    };
    if (PF) {
        load_tile(unit_of(0));
        stage1();
        __syncthreads();
    }
    FFT7_STAMP_DECL;
    for (int i = 0, u = unit_of(0); u >= 0; ++i) {
        const int c = u % C, m = u / C;
        const int un = unit_of(i + 1);
        FFT7_MARK(0);
        if (PF) {
            if (un >= 0) load_tile(un);
        } else {
            load_tile(u);
            stage1();
            __syncthreads();
        }
        FFT7_MARK(1);          // PF: next tile's loads issued; else: this tile loaded + stage 1
#pragma unroll
        for (int round = 0; round < ROUNDS; ++round) {
            const int item = stage_item<FROW, NT>(t, round);
            if (item < 0) continue;
            float* row = lds + item * F_RS;
            float re[24], im[24];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * q);
                re[2 * q] = v.x; im[2 * q] = v.y; re[2 * q + 1] = v.z; im[2 * q + 1] = v.w;
            }
            if (!(SE_FFT7_EXP & 4)) se_fft24<false>(re, im);
#pragma unroll
            for (int q = 0; q < 12; ++q) *reinterpret_cast<f32x4*>(row + 4 * q) = (f32x4){re[2 * q], im[2 * q], re[2 * q + 1], im[2 * q + 1]};
        }
        FFT7_MARK(2);          // stage 2
        __syncthreads();
        FFT7_MARK(3);          // barrier
        // X[group of 16 tiles][frequency block][tile in group][c][16 f]: what a unit writes stays inside its group's 468 x 67.6 KB, what
        // pass 2 reads per (frequency block, group) is one contiguous run.
        const unsigned blk = 16u * C * (FB * 2);                       // floats per (group, frequency block)
        float* base = X + (size_t)(m >> 4) * FNFB * blk + ((m & 15) * C + c) * (FB * 2);
#pragma unroll
        for (int round = 0; round < ROUNDS; ++round) {
            const int item = stage_item<FROW, NT>(t, round);
            if (item < 0) continue;
            const int kz = item / FP, kx = item - kz * FP;
            const float* col = lds + kz * (FP * F_RS) + 2 * kx;
            float re[24], im[24];
#pragma unroll
            for (int y = 0; y < 24; ++y) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(col + y * F_RS);
                re[y] = v.x; im[y] = v.y;
            }
            if (!(SE_FFT7_EXP & 4)) se_fft24<false>(re, im);
            // 16-byte stores (pair_pack): the even lane stores its pair's two frequencies of ky = 2 e, the odd lane those of ky = 2 e + 1
            // frequency ky * 312 + item in blocks of 16: ky = 2 e starts at block 39 e with lane part item, ky = 2 e + 1 at block
            // 39 e + 19 with lane part item + 8
            const int ie = item & ~1;
            const unsigned loff = odd ? (unsigned)((((ie + 8) >> 4) + 19) * blk + ((ie + 8) & 15) * 2) : (unsigned)((ie >> 4) * blk + (ie & 15) * 2);
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const f32x4 v = pair_pack(odd, re[2 * e], im[2 * e], re[2 * e + 1], im[2 * e + 1]);
                if (!(SE_FFT7_EXP & 1) || v.x == 123.456f) st_stream<1>(base + (size_t)(FB2KY * e) * blk + loff, v);
            }
        }
        FFT7_MARK(4);          // stage 3: LDS reads, transform, stores issued
        __syncthreads();       // every stage-3 read of the LDS image is done
        FFT7_MARK(5);          // barrier
        if (PF) {
            if (un >= 0) stage1();
            FFT7_MARK(6);      // wait for the next tile + its stage 1
            __syncthreads();
            FFT7_MARK(7);      // barrier
        }
        u = un;
    }
    FFT7_FLUSH(0);
}

// ------------------------------------------------------------------------------------------------
// pass 2: per-frequency complex GEMM over the channels as a real GEMM on the matrix cores.
// Workgroup (512 threads) = one block of 16 frequencies x a range of 16-tile groups; wave w owns frequencies 2 w, 2 w + 1 of the block and
// keeps their A fragments (2 x 34 registers) for the whole range.  Per 16-tile group:
//   the X block [16 tiles][C][16 f] complex is ONE contiguous run (67.6 KB at C = 33): 16-byte pieces, coalesced, requested one group
//   ahead (rotated loop) and transposed into LDS  I[f][tile][k = 2 c + re/im]  (rows of 72 floats: conflict-free b128 reads of the B
//   operand; k = 66..71 stay zero); each wave: 2 x 17 k-steps x 2 cout tiles = 68 MFMAs; D fragments (lane = tile, 2 couts complex)
//   -> LDS  O[tile][co][f]  -> the Y block [16 tiles][16 co][16 f] complex, again one contiguous 32 KB run, 16-byte pieces.
// (A four-wave form over 8-frequency blocks, two workgroups per CU, ran this pass in 0.27 ms - but 64-byte segments cost passes 1 and 3
// more than that: see the file header.)
// ------------------------------------------------------------------------------------------------
constexpr int G_THREADS = 32 * FB;             // one wave per two frequencies: 8 waves
constexpr int G_MS = 72;                          // floats per (f, tile) row of I
constexpr int G_FS = 16 * G_MS + 4;               // floats per frequency of I (the 4: the transpose's ds_write_b64 spread over all banks)
constexpr int G_I_FLOATS = FB * G_FS;             // 18,496
constexpr int G_OC = 2 * FB + 4;                  // floats per (tile, co) row of O: 16 f complex + 4
constexpr int G_OM = 16 * G_OC + 4;               // floats per tile of O
constexpr int G_O_FLOATS = 16 * G_OM;             // 9,280
constexpr int G_LDS_BYTES = (G_I_FLOATS + G_O_FLOATS) * 4;      // 111,104

template <int C>
__global__ __launch_bounds__(G_THREADS) void fft7_gemm_kernel(const float* __restrict__ X, const float* __restrict__ hf,
                                                              float* __restrict__ Y, int M, int groups_per_wg) {
    static_assert(2 * C <= 66, "k layout: 16 b128-fed steps + one scalar step");
    constexpr int PIECES = 16 * C * (FB / 2);     // 16-byte pieces of an X block
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
                a[ff][s][nt] = hf[((size_t)(fb * FB + 2 * wave + ff) * G_KSTEPS + s) * 128 + nt * 64 + lane];
    // zero the k padding of every (f, tile) row once (never overwritten): k = 2 C .. 71
    if (t < FB * 16) {
        float* row = I + (t >> 4) * G_FS + (t & 15) * G_MS;
#pragma unroll
        for (int k = 2 * C; k < G_MS; k += 2) *reinterpret_cast<f32x2*>(row + k) = (f32x2){0.f, 0.f};
    }
    // where this thread's pieces go in I: piece p = (tile m, channel c, frequency pair fp) -> rows (2 fp, m) and (2 fp + 1, m), column 2 c
    int ioff[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
        const int p = min(t + j * G_THREADS, PIECES - 1);
        const int m = p / (C * (FB / 2)), c = (p / (FB / 2)) % C, fp = p % (FB / 2);
        ioff[j] = (2 * fp) * G_FS + m * G_MS + 2 * c;
    }
    // One group ahead, in a ROTATED loop: a trip requests group g + 1 at its top, runs the matrix phase and the output stage of group g
    // and transposes group g + 1 into LDS at its end - the loads fly under the matrix phase and no loaded register is carried around
    // the loop (see pass 1).
    f32x4 pre[NLOAD];
    // Every load and every store of the loop is UNCONDITIONAL: with a lane- or tile-dependent condition around them hipcc cannot count
    // the operations in flight and waits with vmcnt(0) - for the stores of the group before as well.  Both buffers hold whole groups of
    // 16 tiles (se_conv3d_k7_fft_workspace_elems), a tile beyond M is a column of its own in every product (garbage in, garbage out,
    // never read by pass 3), and the pieces beyond the block's last (NLOAD * 256 > PIECES) re-read the last piece.
    auto prefetch = [&](int g) {
        const float* src = X + ((size_t)g * FNFB + fb) * (16 * C * FB * 2);
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) pre[j] = ld_stream<2>(src + (size_t)min(t + j * G_THREADS, PIECES - 1) * 4);
    };
    auto transpose = [&]() {
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            if (t + j * G_THREADS < PIECES) {
                *reinterpret_cast<f32x2*>(I + ioff[j]) = (f32x2){pre[j].x, pre[j].y};
                *reinterpret_cast<f32x2*>(I + ioff[j] + G_FS) = (f32x2){pre[j].z, pre[j].w};
            }
        }
    };
    const int mcol = lane & 15, kg = lane >> 4;
    prefetch(g0);
    transpose();
    __syncthreads();
    FFT7_STAMP_DECL;
    for (int g = g0; g < g1; ++g) {
        FFT7_MARK(0);
        if (g + 1 < g1) prefetch(g + 1);
        FFT7_MARK(1);          // loads of group g + 1 issued
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
                    if (SE_FFT7_EXP & 32) { acc[ff][0].x += a[ff][4 * i + j][0] + bv[j]; acc[ff][1].x += a[ff][4 * i + j][1]; continue; }
                    acc[ff][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ff][4 * i + j][0], bv[j], acc[ff][0], 0, 0, 0);
                    acc[ff][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ff][4 * i + j][1], bv[j], acc[ff][1], 0, 0, 0);
                }
            }
            const float bl = brow[64 + kg];
            acc[ff][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ff][16][0], bl, acc[ff][0], 0, 0, 0);
            acc[ff][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ff][16][1], bl, acc[ff][1], 0, 0, 0);
        }
        FFT7_MARK(2);          // matrix phase
        // D fragment: lane (tile mcol, kg) holds n = 16 nt + 4 kg + j -> couts 8 nt + 2 kg, + 1, complex
#pragma unroll
        for (int ff = 0; ff < 2; ++ff)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float* o = O + mcol * G_OM + (8 * nt + 2 * kg) * G_OC + 2 * (2 * wave + ff);
                *reinterpret_cast<f32x2*>(o) = (f32x2){acc[ff][nt].x, acc[ff][nt].y};
                *reinterpret_cast<f32x2*>(o + G_OC) = (f32x2){acc[ff][nt].z, acc[ff][nt].w};
            }
        FFT7_MARK(3);          // D -> LDS
        __syncthreads();       // O complete; every matrix-phase read of I done
        FFT7_MARK(4);          // barrier
        {
            float* dst = Y + ((size_t)g * FNFB + fb) * (16 * 16 * FB * 2);
#pragma unroll
            for (int j = 0; j < (16 * 16 * FB / 2) / G_THREADS; ++j) {
                const int p = t + j * G_THREADS;
                const int m = p / (16 * FB / 2), co = (p / (FB / 2)) & 15, fp = p % (FB / 2);
                const f32x4 v = *reinterpret_cast<const f32x4*>(O + m * G_OM + co * G_OC + 4 * fp);
                if (!(SE_FFT7_EXP & 64) || v.x == 123.456f) st_stream<4>(dst + (size_t)p * 4, v);
            }
        }
        FFT7_MARK(5);          // Y block stored
        if (g + 1 < g1) transpose();
        FFT7_MARK(6);          // group g + 1 arrived + transposed into LDS
        __syncthreads();       // I complete; every read of O done
        FFT7_MARK(7);          // barrier
    }
    FFT7_FLUSH(1);
}

// ------------------------------------------------------------------------------------------------
// pass 3: inverse transform of (tile, 4 output channels) per loop trip.  Per channel:
//   stage 1 (312 items = columns (kz, kx)): 24 ky straight from global memory (a wave reads 64 consecutive frequencies, 16 bytes per lane:
//            pair_unpack; the spectrum of the NEXT channel is requested right behind the unpacking), inverse along y, the 16 valid y
//            -> LDS [kz][y][kx]
//   stage 2 (208 items = rows (kz, y < 16)): inverse along x in place, 16 valid x written back
//   stage 3 (256 items = (y, x)): the 13 kz of the half spectrum, Hermitian extension, inverse along z: the real parts of z < 16 stay
//            in 16 registers
// then (y, x) owns 16 z x 4 channels: + bias, ReLU, one 16-byte store per z - 16 lanes write 256 contiguous bytes of a quad-planar
// output row [B][4][D^3][4] (OUTQ), or 16 of every 64 bytes of a channels-last record [B][D^3][16].
// ------------------------------------------------------------------------------------------------
template <bool OUTQ, int NT>
__global__ __launch_bounds__(NT) void fft7_inv_kernel(const float* __restrict__ Y, const float* __restrict__ bias,
                                                      float* __restrict__ out, int D, int T, int M, int n_units, int relu) {
    constexpr int ROUNDS = NT >= FROW ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    constexpr unsigned blk = 16u * 16 * (FB * 2);                    // floats per (group of 16 tiles, frequency block)
    const bool odd = t & 1;
    const int it1[2] = {stage_item<FROW, NT>(t, 0), stage_item<FROW, NT>(t, 1)};
    unsigned loff[2];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int ie = max(it1[round], 0) & ~1;
        loff[round] = odd ? (unsigned)((((ie + 8) >> 4) + 19) * blk + ((ie + 8) & 15) * 2) : (unsigned)((ie >> 4) * blk + (ie & 15) * 2);
    }
    // round 0 (256 columns) of the next spectrum is prefetched across stages 2 and 3 (48 registers); round 1 (14 lanes per wave) is
    // requested at the start of stage 1 and lands under round 0's transform: with both rounds held across the stages the kernel needed
    // more than 256 registers (AGPR copies of loaded data, one wave per SIMD)
    f32x4 raw0[12];
    auto issue_loads = [&](int m, int co, int round, f32x4 (&dst)[12]) {
        const float* base = Y + (size_t)(m >> 4) * FNFB * blk + ((m & 15) * 16 + co) * (FB * 2);
#pragma unroll
        for (int e = 0; e < 12; ++e)       // unconditional (a lane without an item re-reads item 0: countable operations, see pass 2)
            dst[e] = ld_stream<8>(base + (size_t)(FB2KY * e) * blk + loff[round]);
    };
    // stage 1 of a (tile, channel): unpack the spectrum, inverse along y, the 16 valid y -> LDS
    auto stage1 = [&](int m, int co) {
        f32x4 raw1[12];
#pragma unroll
        for (int round = 0; round < ROUNDS; ++round) {
            float re[24], im[24];
#pragma unroll
            for (int e = 0; e < 12; ++e) pair_unpack(round ? raw1[e] : raw0[e], odd, re[2 * e], im[2 * e], re[2 * e + 1], im[2 * e + 1]);
            if (round == 0 && ROUNDS == 2) issue_loads(m, co, 1, raw1);       // behind the unpacking: raw0's registers are free again
            const int item = it1[round];
            if (item >= 0) {
                const int kz = item / FP, kx = item - kz * FP;
                if (!(SE_FFT7_EXP & 4)) se_fft24<true>(re, im);
                float* col = lds + kz * (FV * F_RS) + 2 * kx;
#pragma unroll
                for (int y = 0; y < FV; ++y) *reinterpret_cast<f32x2*>(col + y * F_RS) = (f32x2){re[y], im[y]};
            }
        }
    };
    // ROTATED as pass 1: a step runs stages 2 and 3 of (unit, channel j) and stage 1 of the next (unit, channel), whose spectrum is
    // requested at the top of the step - no loaded register is carried around the loop
    if ((int)blockIdx.x >= n_units) return;
    issue_loads((int)blockIdx.x >> 2, 4 * ((int)blockIdx.x & 3), 0, raw0);
    stage1((int)blockIdx.x >> 2, 4 * ((int)blockIdx.x & 3));
    __syncthreads();
    FFT7_STAMP_DECL;
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int q = u & 3, m = u >> 2;
        int r = m;
        const int tx = r % T; r /= T;
        const int ty = r % T; r /= T;
        const int tz = r % T;
        const int b = r / T;
        const int un = u + (int)gridDim.x;
        float stash[4][16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool more = j < 3 || un < n_units;
            const int mn = j < 3 ? m : un >> 2, con = j < 3 ? 4 * q + j + 1 : 4 * (un & 3);       // the next (tile, channel)
            if (more) issue_loads(mn, con, 0, raw0);
            if (t < FKZ * FV) {
                float* row = lds + t * F_RS;
                float re[24], im[24];
#pragma unroll
                for (int p = 0; p < 12; ++p) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * p);
                    re[2 * p] = v.x; im[2 * p] = v.y; re[2 * p + 1] = v.z; im[2 * p + 1] = v.w;
                }
                if (!(SE_FFT7_EXP & 4)) se_fft24<true>(re, im);
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
                if (!(SE_FFT7_EXP & 4)) se_fft24<true>(re, im);
#pragma unroll
                for (int z = 0; z < FV; ++z) stash[j][z] = re[z];
            }
            if (j == 3 && t < 256) {
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
                    if (!(SE_FFT7_EXP & 16) || v.x == 123.456f) *reinterpret_cast<f32x4*>(o) = v;
                }
            }
            __syncthreads();       // every stage-3 read of the LDS image is done
            if (more) stage1(mn, con);
            __syncthreads();
        }
    }
    FFT7_FLUSH(2);
}

}  // namespace

#if defined(SE_FFT7_STAMP) || defined(SE_FFT7_CENSUS)
extern "C" int se_debug_fft7_stamps(unsigned long long* host, long long bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fft7_stamp), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#endif


// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" long long se_conv3d_k7_fft_packed_elems(int cin, int cout) {
    if ((cin != 33 && cin != 32) || cout != 16) return -1;       // 33: features + occupancy; 32: `with_scene: False` (network/voxel_net_depth.py:65-77)
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
    if (batch <= 0 || dim < 16 || (dim & 15) || (cin != 33 && cin != 32)) return -1;
    const long long T = dim / 16, M = ((long long)batch * T * T * T + 15) / 16 * 16;       // whole groups of 16 tiles
    return M * (cin + 16) * FNF * 2;
}

extern "C" int se_conv3d_k7_fft_f32(const float* in, const float* hfrag, const float* bpack, float* out, int batch, int dim, int cin,
                                    int cout, int flags, float* workspace, long long workspace_elems, void* stream) {
    if (batch <= 0 || dim < 16 || (dim & 15) || (cin != 33 && cin != 32) || cout != 16 || !in || !hfrag || !bpack || !out || !workspace) return SE_ERR_BAD_ARG;
    if (flags & ~(SE_EPI_RELU | SE_OUT_QUAD)) return SE_ERR_BAD_ARG;
    const long long per_sample = se_conv3d_k7_fft_workspace_elems(1, dim, cin);
    int chunk = (int)(workspace_elems / per_sample < batch ? workspace_elems / per_sample : batch);
    if (chunk <= 0) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    const int T = dim / 16, cus = se_num_cus();
    const size_t vox = (size_t)dim * dim * dim;
    const int f1_lds = F1_LDS_FLOATS * 4, f3_lds = F3_LDS_FLOATS * 4;
    // kernel forms (measured, see fft7_fwd_kernel): pass 1 = 320 threads without prefetch, pass 3 = 320 threads with the next spectrum
    // prefetched; SE_FFT7_FWD_NT / SE_FFT7_INV_NT = 256 build the four-wave forms (A/B)
#ifndef SE_FFT7_FWD_NT
#define SE_FFT7_FWD_NT 320
#endif
#ifndef SE_FFT7_FWD_PF
#define SE_FFT7_FWD_PF 0
#endif
#ifndef SE_FFT7_INV_NT
#define SE_FFT7_INV_NT 320
#endif
    auto fwd = fft7_fwd_kernel<SE_FFT7_FWD_NT, SE_FFT7_FWD_PF != 0>;
    auto invq = fft7_inv_kernel<true, SE_FFT7_INV_NT>;
    auto invc = fft7_inv_kernel<false, SE_FFT7_INV_NT>;
    SE_ENSURE_LDS(fwd, f1_lds);
    SE_ENSURE_LDS(fft7_gemm_kernel<33>, G_LDS_BYTES);
    SE_ENSURE_LDS(fft7_gemm_kernel<32>, G_LDS_BYTES);
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = batch - b0 < chunk ? batch - b0 : chunk;
        const int M = nb * T * T * T;
        float* X = workspace;
        float* Yb = workspace + (size_t)((M + 15) / 16 * 16) * cin * FNF * 2;
        {
            const int units = M * cin;
            const int grid = units < 2 * cus * 4 ? units : 2 * cus * 4;
            hipLaunchKernelGGL(fwd, dim3(grid), dim3(SE_FFT7_FWD_NT), f1_lds, s, in + (size_t)b0 * cin * vox, X, cin, dim, T,
                               M, units);
            SE_CHECK_LAUNCH();
        }
        {
            const int n_groups = (M + 15) / 16;
            int split = (4 * cus + FNFB - 1) / FNFB;                    // ~4 workgroups per CU over the launch
            if (split > n_groups) split = n_groups;
            const int gpw = (n_groups + split - 1) / split;
            const dim3 ggrid(FNFB, (n_groups + gpw - 1) / gpw);
            if (cin == 33) hipLaunchKernelGGL(fft7_gemm_kernel<33>, ggrid, dim3(G_THREADS), G_LDS_BYTES, s, X, hfrag, Yb, M, gpw);
            else hipLaunchKernelGGL(fft7_gemm_kernel<32>, ggrid, dim3(G_THREADS), G_LDS_BYTES, s, X, hfrag, Yb, M, gpw);
            SE_CHECK_LAUNCH();
        }
        {
            const int units = M * 4;
            const int grid = units < 4 * cus ? units : 4 * cus;       // two resident workgroups per CU, two rounds: each walks >= 2 units
            const int relu = (flags & SE_EPI_RELU) ? 1 : 0;
            float* o = out + (size_t)b0 * 16 * vox;
            if (flags & SE_OUT_QUAD)
                hipLaunchKernelGGL(invq, dim3(grid), dim3(SE_FFT7_INV_NT), f3_lds, s, Yb, bpack, o, dim, T, M, units, relu);
            else
                hipLaunchKernelGGL(invc, dim3(grid), dim3(SE_FFT7_INV_NT), f3_lds, s, Yb, bpack, o, dim, T, M, units, relu);
            SE_CHECK_LAUNCH();
        }
    }
    return 0;
}
