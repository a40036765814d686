// 1x1 convolutions of the 2-D backbone as a float32 MFMA GEMM with the whole epilogue fused (round 6).
//
// Stands in for the 1x1 convolutions of every ResNet-50 Bottleneck - conv1, conv3 and the downsample convolution (stride 1 and 2) - with
// their folded BatchNorm, the residual add and the ReLU (reference network/pose_resnet.py:52-90, Bottleneck.forward :72-90, downsample
// :140-146): 36 of the backbone's 53 convolutions.  MIOpen runs them as strided-batched Tensile GEMMs at 32-68 TFLOP/s
// (profiles/r04_backbone_solvers.txt) and the bias / residual / ReLU cost one more pass each (se_bias_act_nchw_f32): 1.04 + 0.22 ms of the
// 2.03 ms backbone at B = 8.  Most of these layers are short-K products (K = 64 ... 512 channels) over tens of thousands of pixels: the
// expanding convolutions of layer1 move 75 MB for 2 GFLOP - they are bound by bytes, and a fused epilogue removes two of their four tensor
// passes (three with in_bias: the producing 3x3 convolution's bias + ReLU applied on the way into LDS).
// Three forms: conv1x1_kernel (64 pixels x 64 / 128 channels per workgroup: batch 8), its stride-2 mode, and conv1x1_small_kernel (64 x 16,
// the k steps over four wave groups: the long-K layers at batch 1-4).
//
// NCHW in, NCHW out (what the 3x3 convolutions of MIOpen around them read and write).  GEMM roles: MFMA rows = PIXELS, MFMA columns =
// output channels, so a lane's D fragment is 4 consecutive pixels of one channel: one 16-byte store (and one 16-byte residual load) in NCHW.
//   workgroup (256 threads = 2 x 2 waves) = BP pixels x BC output channels; per 16 input channels: the X tile [16][BP] (rows of the NCHW
//   tensor as they lie in memory, 16-byte loads) and the W tile [BC][16] (pre-packed per (channel tile, k step): one contiguous run) go
//   through registers into the other half of a double-buffered LDS image while the matrix cores work on this one; operands: ds_read_b32
//   of X[4 kg + j][pixel] and ds_read_b128 of W[cout][4 kg .. 4 kg + 3] (k = 4 kg + j on both sides), rows padded conflict-free.
// float32 in, float32 accumulate (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain) - results differ from MIOpen's by summation order only.
#include "common.h"

namespace {

constexpr int C1_SPLIT_MIN_CIN = 128;   // k split over two wave groups from here on (8+ k steps): 17.4 -> 14.8 us at 512 -> 128 @32^2, 18.5 -> 16.1 at 256 -> 1024 @16^2; layer1's 64-channel inputs lose 1 us
#ifndef C1_PREFETCH
#define C1_PREFETCH 2
#endif
constexpr int C1_LDW = 24;          // floats per cout row of the W image (16 k + 8: conflict-free ds_read_b128 over the 4 x 16 lane groups)

template <int BP>
struct C1Geom {
    static constexpr int LDX = BP + 4;                                  // floats per k row of the X image
    static constexpr int X_FLOATS = 16 * LDX;
};

// one K step of 16 input channels on the LDS images xs [16][LDX], ws [BC][24]
template <int PT, int CT, int LDX>
__device__ __forceinline__ void c1_step(const float* __restrict__ xs, const float* __restrict__ ws, f32x4 (&acc)[PT][CT], int i, int kg) {
    f32x4 wv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wv[ct] = *reinterpret_cast<const f32x4*>(ws + (ct * 16 + i) * C1_LDW + 4 * kg);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float xv[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xv[pt] = xs[(4 * kg + j) * LDX + pt * 16 + i];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[pt], wv[ct][j], acc[pt][ct], 0, 0, 0);
    }
}

// x [B][cin][P] (P = H * W), wpack [cout / BC][cin / 16][BC][16], bias [cout], res / out [B][cout][P]; total pixels B * P % BP == 0,
// P % 16 == 0, cin % 16 == 0, cout % BC == 0.
// KS = 2: two groups of four waves share the tile and take alternate k steps (each group has its own double-buffered LDS images and runs
// the same rotated loop; the barriers are common), so a workgroup has half the serial k steps and a CU twice the waves to hide the
// step's latencies behind; the groups swap half of their accumulators through LDS at the end and each finishes half of the channels.
// INB: the input is the raw result of the producing convolution and its bias + ReLU happen HERE, on the way into LDS: x' = max(x +
// in_bias[channel], 0) (the 3x3 convolution of a Bottleneck in front of conv3: its own epilogue pass over the tensor disappears).
// TRIPS > 0: the group's k loop has exactly TRIPS steps and is unrolled completely, with the loads TWO steps ahead of their use (two
// register sets; in straight-line code hipcc counts its vmcnt waits exactly - loads carried around a loop's back edge get vmcnt(0)): a
// workgroup's steps are a chain of load latencies (16-32 MFMAs per wave and step against ~2 us to L2 / HBM), and most launches have one
// or two workgroups per CU.  TRIPS = 0: the run-time loop, loads one step ahead.
// MODE 2 (stride 2, the downsample convolution of a stage's first Bottleneck): P counts OUTPUT pixels per sample, `wo` is the output
// width; output pixel (y, x) reads input pixel (2 y, 2 x) of a [2 ho][2 wo] map - a thread's four pixels are the even elements of
// eight consecutive input floats (two 16-byte loads).
template <int BP, int BC, int KS, int MODE, int TRIPS>
__global__ __launch_bounds__(256 * KS) void conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
                                                      const float* __restrict__ res, const float* __restrict__ in_bias, float* __restrict__ out,
                                                      int cin, int cout, int P, int relu, int wo) {
    constexpr bool INB = MODE == 1, S2 = MODE == 2;
    constexpr int PT = BP / 32, CT = BC / 32;                            // 16 x 16 tiles per wave: (BP / 2) pixels x (BC / 2) channels
    constexpr int LDX = C1Geom<BP>::LDX;
    constexpr int XF = C1Geom<BP>::X_FLOATS, WF = BC * C1_LDW;
    constexpr int XV = BP * 16 / 4 / 256, WV = BC * 16 / 4 / 256;         // 16-byte pieces per thread and step
    static_assert(XV >= 1 && WV >= 1 && PT >= 1 && CT >= 1, "tile too small for 256 threads");
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [2][XF] X images, [2][WF] W images
    const int grp = KS > 1 ? (int)(threadIdx.x >> 8) : 0;                  // k-split group (wave-uniform)
    float* xs = lds + grp * (2 * XF + 2 * WF);
    float* ws = xs + 2 * XF;
    const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
    const int i = lane & 15, kg = lane >> 4;
    const int wp = wave & 1, wc = wave >> 1;
    const long long n0 = (long long)blockIdx.x * BP;                      // first pixel (over the batch) of the tile
    const int c0 = blockIdx.y * BC;
    const int steps = cin >> 4, trips = steps / KS;                        // this group's k steps: grp, grp + KS, ...

    // this thread's pieces of an X tile: piece q = t + v * 256 -> row k = q / (BP / 4), pixels 4 (q % (BP / 4)) ..
    const float* xsrc[XV];
    int xdst[XV];
#pragma unroll
    for (int v = 0; v < XV; ++v) {
        const int q = t + v * 256, k = q / (BP / 4), c4 = q % (BP / 4);
        const long long n = n0 + 4 * c4;
        const long long b = n / P;
        const int p = (int)(n - b * P);
        if (S2) {
            const int y = p / wo, xo = p - y * wo;
            xsrc[v] = x + (b * cin + k) * (4LL * P) + (2LL * y) * (2 * wo) + 2 * xo;
        } else {
            xsrc[v] = x + (b * cin + k) * P + p;                          // + step * 16 * P
        }
        xdst[v] = k * LDX + 4 * c4;
    }
    const float* wsrc = wpack + (long long)blockIdx.y * steps * (BC * 16) + t * 4;      // + step * BC * 16 + v * 1024
    int wdst[WV];
#pragma unroll
    for (int v = 0; v < WV; ++v) {
        const int q = t + v * 256;
        wdst[v] = (q >> 2) * C1_LDW + 4 * (q & 3);
    }
    constexpr int PD = C1_PREFETCH;                                        // steps the loads run ahead in the unrolled forms
    f32x4 xr2[PD][XV], wr2[PD][WV];
    float xb2[PD][XV];
    auto fetch = [&](int s, int set = 0) {
        f32x4 (&xr)[XV] = xr2[set];
        f32x4 (&wr)[WV] = wr2[set];
        float (&xb)[XV] = xb2[set];
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            if (S2) {
                const float* q = xsrc[v] + (long long)s * 64 * P;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(q), hi = *reinterpret_cast<const f32x4*>(q + 4);
                xr[v] = (f32x4){lo.x, lo.z, hi.x, hi.z};
            } else {
                xr[v] = *reinterpret_cast<const f32x4*>(xsrc[v] + (long long)s * 16 * P);
            }
            if (INB) xb[v] = in_bias[s * 16 + (t + v * 256) / (BP / 4)];
        }
#pragma unroll
        for (int v = 0; v < WV; ++v) wr[v] = *reinterpret_cast<const f32x4*>(wsrc + (long long)s * (BC * 16) + v * 1024);
    };
    auto commit = [&](int buf, int set = 0) {
        f32x4 (&xr)[XV] = xr2[set];
        f32x4 (&wr)[WV] = wr2[set];
        float (&xb)[XV] = xb2[set];
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            if (INB) {
                xr[v].x = fmaxf(xr[v].x + xb[v], 0.f); xr[v].y = fmaxf(xr[v].y + xb[v], 0.f);
                xr[v].z = fmaxf(xr[v].z + xb[v], 0.f); xr[v].w = fmaxf(xr[v].w + xb[v], 0.f);
            }
            *reinterpret_cast<f32x4*>(xs + buf * XF + xdst[v]) = xr[v];
        }
#pragma unroll
        for (int v = 0; v < WV; ++v) *reinterpret_cast<f32x4*>(ws + buf * WF + wdst[v]) = wr[v];
    };
    f32x4 acc[PT][CT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (TRIPS > 0) {
        // step n travels in register set n % PD: requested at the top of trip n - PD, written to LDS half n & 1 at the end of trip n - 1
        fetch(grp, 0);
#pragma unroll
        for (int n = 1; n < PD; ++n)
            if (n < TRIPS) fetch(n * KS + grp, n);
        commit(0, 0);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < TRIPS; ++s) {
            if (s + PD < TRIPS) fetch((s + PD) * KS + grp, s % PD);
            c1_step<PT, CT, LDX>(xs + (s & 1) * XF + wp * (BP / 2), ws + (s & 1) * WF + wc * (BC / 2) * C1_LDW, acc, i, kg);
            if (s + 1 < TRIPS) commit((s + 1) & 1, (s + 1) % PD);
            __syncthreads();
        }
    } else {
        fetch(grp);
        commit(0);
        __syncthreads();
        // rotated: a trip requests the group's next step at its top and writes it into the other LDS half at its end (no loop-carried load registers)
        for (int s = 0; s < trips; ++s) {
            const int buf = s & 1;
            if (s + 1 < trips) fetch((s + 1) * KS + grp);
            c1_step<PT, CT, LDX>(xs + buf * XF + wp * (BP / 2), ws + buf * WF + wc * (BC / 2) * C1_LDW, acc, i, kg);
            if (s + 1 < trips) commit(buf ^ 1);
            __syncthreads();
        }
    }
    if (KS > 1) {
        // accumulator tile q = pt * CT + ct belongs to group q % KS: every group writes the tiles it does not own into its OWN LDS region
        // (free after the last barrier; slot = the tile's rank among those), then adds the other groups' copies of its own tiles
        constexpr int NQ = PT * CT, REGION = 2 * XF + 2 * WF;
        static_assert(NQ % KS == 0 && (NQ - NQ / KS) * 1024 <= REGION, "exchange image must fit the group's LDS region");
        f32x4* mine = reinterpret_cast<f32x4*>(xs);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (q % KS != grp) mine[(q - q / KS - (q % KS > grp ? 1 : 0)) * 256 + t] = acc[q / CT][q % CT];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (q % KS == grp) {
#pragma unroll
                for (int o = 0; o < KS; ++o)
                    if (o != grp) acc[q / CT][q % CT] += reinterpret_cast<const f32x4*>(lds + o * REGION)[(q - q / KS - (q % KS > o ? 1 : 0)) * 256 + t];
            }
    }
    // epilogue: lane (channel i of the tile, pixel quad kg): + bias (+ residual) (ReLU), 16-byte NCHW accesses
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int co = c0 + wc * (BC / 2) + ct * 16 + i;
        const float bv = bias[co];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            if (KS > 1 && (pt * CT + ct) % KS != grp) continue;
            const long long n = n0 + wp * (BP / 2) + pt * 16 + 4 * kg;
            const long long b = n / P;
            const int p = (int)(n - b * P);
            const long long off = (b * cout + co) * P + p;
            f32x4 v = acc[pt][ct] + bv;
            if (res) v += *reinterpret_cast<const f32x4*>(res + off);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<f32x4*>(out + off) = v;
        }
    }
}

// Small-M form (batch 1-2: the long-K layers of layer2 .. layer4, 64 - 1024 pixels against 0.3 - 4 MB of weights): a workgroup owns 64
// pixels x 16 output channels, so that even a 64-pixel map gives cout / 16 workgroups, and its KS wave groups take alternate 32-channel k
// steps (wave = one 16-pixel row tile, all 16 channels); the groups' sums are added through LDS at the end and group 0 finishes.
//   wpack16 = [cout / 16][cin / 16][16][16] (conv2d_1x1's packing with a 16-channel tile: a 32-channel step is one contiguous 2 KB run).
// The X tile is re-read by every channel tile (cout / 16 times): right where the weights dominate the bytes, wrong for large maps - the caller
// routes by shape.  MODE 1: in_bias as above.  TRIPS as above.
template <int KS, int MODE, int TRIPS>
__global__ __launch_bounds__(256 * KS) void conv1x1_small_kernel(const float* __restrict__ x, const float* __restrict__ wpack16, const float* __restrict__ bias,
                                                                 const float* __restrict__ res, const float* __restrict__ in_bias, float* __restrict__ out,
                                                                 int cin, int cout, int P, int relu, int wo) {
    constexpr bool INB = MODE == 1, S2 = MODE == 2;
    constexpr int KC = 32, LDX = 68, XF = KC * LDX, WF = (KC / 16) * 16 * C1_LDW, REGION = 2 * XF + 2 * WF;
    constexpr int PD = C1_PREFETCH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int grp = KS > 1 ? (int)(threadIdx.x >> 8) : 0;
    float* xs = lds + grp * REGION;
    float* ws = xs + 2 * XF;
    const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
    const int i = lane & 15, kg = lane >> 4;
    const long long n0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 16;
    const int trips = cin / (KC * KS);
    // X pieces: q = t + 256 v -> channel q / 16 of the step, pixels 4 (q % 16) ..; W piece: q = t % 128 -> 16 bytes of the step's 2 KB run
    const float* xsrc[2];
    int xdst[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int q = t + 256 * v, k = q >> 4, c4 = q & 15;
        const long long n = n0 + 4 * c4;
        const long long b = n / P;
        const int p = (int)(n - b * P);
        if (S2) {
            const int y = p / wo, xo = p - y * wo;
            xsrc[v] = x + (b * cin + k) * (4LL * P) + (2LL * y) * (2 * wo) + 2 * xo;
        } else {
            xsrc[v] = x + (b * cin + k) * P + p;
        }
        xdst[v] = k * LDX + 4 * c4;
    }
    const int wq = t & 127;
    const float* wsrc = wpack16 + (long long)blockIdx.y * (cin >> 4) * 256 + wq * 4;
    const int wdst = (wq >> 2) * C1_LDW + 4 * (wq & 3);
    f32x4 xr2[PD][2], wr2[PD];
    float xb2[PD][2];
    auto fetch = [&](int s, int set = 0) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            if (S2) {
                const float* q = xsrc[v] + (long long)s * KC * 4 * P;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(q), hi = *reinterpret_cast<const f32x4*>(q + 4);
                xr2[set][v] = (f32x4){lo.x, lo.z, hi.x, hi.z};
            } else {
                xr2[set][v] = *reinterpret_cast<const f32x4*>(xsrc[v] + (long long)s * KC * P);
            }
            if (INB) xb2[set][v] = in_bias[s * KC + ((t + 256 * v) >> 4)];
        }
        wr2[set] = *reinterpret_cast<const f32x4*>(wsrc + (long long)s * (KC * 16));
    };
    auto commit = [&](int buf, int set = 0) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            f32x4 r = xr2[set][v];
            if (INB) {
                const float bb = xb2[set][v];
                r.x = fmaxf(r.x + bb, 0.f); r.y = fmaxf(r.y + bb, 0.f); r.z = fmaxf(r.z + bb, 0.f); r.w = fmaxf(r.w + bb, 0.f);
            }
            *reinterpret_cast<f32x4*>(xs + buf * XF + xdst[v]) = r;
        }
        *reinterpret_cast<f32x4*>(ws + buf * WF + wdst) = wr2[set];
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const float* xa = xs + buf * XF + 4 * kg * LDX + wave * 16 + i;
        const float* wa = ws + buf * WF + i * C1_LDW + 4 * kg;
#pragma unroll
        for (int sub = 0; sub < KC / 16; ++sub) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wa + sub * 16 * C1_LDW);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[(sub * 16 + j) * LDX], wv[j], acc, 0, 0, 0);
        }
    };
    if (TRIPS > 0) {
        fetch(grp, 0);
#pragma unroll
        for (int n = 1; n < PD; ++n)
            if (n < TRIPS) fetch(n * KS + grp, n);
        commit(0, 0);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < TRIPS; ++s) {
            if (s + PD < TRIPS) fetch((s + PD) * KS + grp, s % PD);
            compute(s & 1);
            if (s + 1 < TRIPS) commit((s + 1) & 1, (s + 1) % PD);
            __syncthreads();
        }
    } else {
        fetch(grp);
        commit(0);
        __syncthreads();
        for (int s = 0; s < trips; ++s) {
            const int buf = s & 1;
            if (s + 1 < trips) fetch((s + 1) * KS + grp);
            compute(buf);
            if (s + 1 < trips) commit(buf ^ 1);
            __syncthreads();
        }
    }
    if (KS > 1) {
        if (grp != 0) reinterpret_cast<f32x4*>(xs)[t] = acc;
        __syncthreads();
        if (grp != 0) return;
#pragma unroll
        for (int o = 1; o < KS; ++o) acc += reinterpret_cast<const f32x4*>(lds + o * REGION)[t];
    }
    const int co = c0 + i;
    const long long n = n0 + wave * 16 + 4 * kg;
    const long long b = n / P;
    const long long off = (b * cout + co) * P + (n - b * P);
    f32x4 v = acc + bias[co];
    if (res) v += *reinterpret_cast<const f32x4*>(res + off);
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4*>(out + off) = v;
}

template <int KS, int TRIPS>
int launch_c1s(const float* x, const float* wpack16, const float* bias, const float* res, const float* in_bias, float* out, long long pixels,
               int cin, int cout, int P, int relu, hipStream_t s, int wo = 0) {
    constexpr int LDS = KS * (2 * 32 * 68 + 2 * 2 * 16 * C1_LDW) * 4;
    const dim3 grid((unsigned)(pixels / 64), cout / 16), block(256 * KS);
    if (wo) {
        auto kern = conv1x1_small_kernel<KS, 2, TRIPS>;
        SE_ENSURE_LDS(kern, LDS);
        hipLaunchKernelGGL(kern, grid, block, LDS, s, x, wpack16, bias, res, in_bias, out, cin, cout, P, relu, wo);
    } else if (in_bias) {
        auto kern = conv1x1_small_kernel<KS, 1, TRIPS>;
        SE_ENSURE_LDS(kern, LDS);
        hipLaunchKernelGGL(kern, grid, block, LDS, s, x, wpack16, bias, res, in_bias, out, cin, cout, P, relu, 0);
    } else {
        auto kern = conv1x1_small_kernel<KS, 0, TRIPS>;
        SE_ENSURE_LDS(kern, LDS);
        hipLaunchKernelGGL(kern, grid, block, LDS, s, x, wpack16, bias, res, in_bias, out, cin, cout, P, relu, 0);
    }
    SE_CHECK_LAUNCH();
    return 0;
}

template <int BP, int BC, int KS, int TRIPS>
int launch_c1t(const float* x, const float* wpack, const float* bias, const float* res, const float* in_bias, float* out, long long pixels,
               int cin, int cout, int P, int relu, hipStream_t s, int wo) {
    constexpr int LDS = KS * (2 * C1Geom<BP>::X_FLOATS + 2 * BC * C1_LDW) * 4;
    const dim3 grid((unsigned)(pixels / BP), cout / BC), block(256 * KS);
    if (wo) {
        auto kern = conv1x1_kernel<BP, BC, KS, 2, TRIPS>;
        SE_ENSURE_LDS(kern, LDS);
        hipLaunchKernelGGL(kern, grid, block, LDS, s, x, wpack, bias, res, in_bias, out, cin, cout, P, relu, wo);
    } else if (in_bias) {
        auto kern = conv1x1_kernel<BP, BC, KS, 1, TRIPS>;
        SE_ENSURE_LDS(kern, LDS);
        hipLaunchKernelGGL(kern, grid, block, LDS, s, x, wpack, bias, res, in_bias, out, cin, cout, P, relu, 0);
    } else {
        auto kern = conv1x1_kernel<BP, BC, KS, 0, TRIPS>;
        SE_ENSURE_LDS(kern, LDS);
        hipLaunchKernelGGL(kern, grid, block, LDS, s, x, wpack, bias, res, in_bias, out, cin, cout, P, relu, 0);
    }
    SE_CHECK_LAUNCH();
    return 0;
}

// the unrolled forms for the trip counts of the backbone (4: 64 channels, or 128 over two groups; 8: 256; 16: 512 - all over two groups),
// the run-time loop for everything else; se_debug_set_variant(78): the run-time loop always (A/B in development builds)
template <int BP, int BC, int KS>
int launch_c1(const float* x, const float* wpack, const float* bias, const float* res, const float* in_bias, float* out, long long pixels,
              int cin, int cout, int P, int relu, hipStream_t s, int wo = 0) {
    const int trips = (cin >> 4) / KS;
    if (g_variant != 78 && KS <= 2) {
        if (trips == 4) return launch_c1t<BP, BC, KS, 4>(x, wpack, bias, res, in_bias, out, pixels, cin, cout, P, relu, s, wo);
        if (trips == 8) return launch_c1t<BP, BC, KS, 8>(x, wpack, bias, res, in_bias, out, pixels, cin, cout, P, relu, s, wo);
        if (trips == 16) return launch_c1t<BP, BC, KS, 16>(x, wpack, bias, res, in_bias, out, pixels, cin, cout, P, relu, s, wo);
    }
    return launch_c1t<BP, BC, KS, 0>(x, wpack, bias, res, in_bias, out, pixels, cin, cout, P, relu, s, wo);
}

}  // namespace

// Channel-tile width of the packed weights for a layer (the caller packs [cout / BC][cin / 16][BC][16] from the folded [cout][cin]
// matrix); 0 = shape not covered (the caller keeps its MIOpen convolution).  Depends on the arguments and the device's CU count only.
extern "C" int se_conv2d_1x1_tile_f32(int batch, int cin, int cout, int hw) {
    if (batch <= 0 || cin <= 0 || (cin & 15) || cout <= 0 || (cout & 63) || hw <= 0 || (hw & 15)) return 0;
    const long long pixels = (long long)batch * hw;
    if (pixels % 64) return 0;
    // 128 channels per workgroup where that still gives every CU a workgroup (64-pixel tiles), else 64 (measured, tools/bench_conv1x1.py:
    // the kernel wins against MIOpen + epilogue pass from ~256 workgroups on and loses below ~128: those shapes would need a split K)
    return (cout % 128 == 0 && (pixels / 64) * (cout / 128) >= se_num_cus()) ? 128 : 64;
}

extern "C" int se_conv2d_1x1_f32(const float* x, const float* wpack, const float* bias, const float* residual, const float* in_bias, float* out,
                                 int batch, int cin, int cout, int hw, int relu, void* stream) {
    const int bc = se_conv2d_1x1_tile_f32(batch, cin, cout, hw);
    if (!bc || !x || !wpack || !bias || !out) return SE_ERR_BAD_ARG;
    const long long pixels = (long long)batch * hw;
    hipStream_t s = se_stream(stream);
    // 64-pixel tiles (128 measured equal or slower on every backbone shape: fewer workgroups in flight; tools/bench_conv1x1.py).
    // k split over two wave groups from 128 input channels on; se_debug_set_variant(73 / 74 / 75) = 1 / 2 / 4 groups (A/B in development
    // builds: four groups tie with MIOpen + epilogue on the 1024-channel layers, 23.3 against 23.1 us, and gain < 1 us elsewhere - not routed)
    const int ks = (g_variant == 73) ? 1 : (g_variant == 74) ? 2 : (g_variant == 75) ? 4 : cin >= C1_SPLIT_MIN_CIN ? 2 : 1;
    const int kse = (cin % (16 * ks)) ? 1 : ks;
    if (bc == 128) {
        if (kse == 4) return launch_c1<64, 128, 4>(x, wpack, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s);
        if (kse == 2) return launch_c1<64, 128, 2>(x, wpack, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s);
        return launch_c1<64, 128, 1>(x, wpack, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s);
    }
    if (kse == 4) return launch_c1<64, 64, 4>(x, wpack, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s);
    if (kse == 2) return launch_c1<64, 64, 2>(x, wpack, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s);
    return launch_c1<64, 64, 1>(x, wpack, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s);
}

// The stride-2 form (the `downsample` convolution of the first Bottleneck of layer2 / layer3 / layer4, network/pose_resnet.py:140-146):
// x [batch][cin][2 ho][2 wo] -> out [batch][cout][ho][wo] = W x[:, :, ::2, ::2] + bias (ReLU optional).  wpack and the covered shapes as
// se_conv2d_1x1_tile_f32(batch, cin, cout, ho * wo) says, and wo % 4 == 0.
extern "C" int se_conv2d_1x1_s2_f32(const float* x, const float* wpack, const float* bias, float* out, int batch, int cin, int cout, int ho,
                                    int wo, int relu, void* stream) {
    if (ho <= 0 || wo <= 0 || (wo & 3)) return SE_ERR_BAD_ARG;
    const int hw = ho * wo;
    const int bc = se_conv2d_1x1_tile_f32(batch, cin, cout, hw);
    if (!bc || !x || !wpack || !bias || !out) return SE_ERR_BAD_ARG;
    const long long pixels = (long long)batch * hw;
    hipStream_t s = se_stream(stream);
    const bool split = cin >= C1_SPLIT_MIN_CIN && cin % 32 == 0;
    if (bc == 128) return split ? launch_c1<64, 128, 2>(x, wpack, bias, nullptr, nullptr, out, pixels, cin, cout, hw, relu, s, wo)
                                : launch_c1<64, 128, 1>(x, wpack, bias, nullptr, nullptr, out, pixels, cin, cout, hw, relu, s, wo);
    return split ? launch_c1<64, 64, 2>(x, wpack, bias, nullptr, nullptr, out, pixels, cin, cout, hw, relu, s, wo)
                 : launch_c1<64, 64, 1>(x, wpack, bias, nullptr, nullptr, out, pixels, cin, cout, hw, relu, s, wo);
}

// The small-M form: 64 pixels x 16 channels per workgroup, k steps over four (cin % 128 == 0) or two (cin % 64 == 0) wave groups; for the
// launches of batch 1-2 that se_conv2d_1x1_f32 would run with a handful of workgroups (a 64-pixel map gives it cout / 64, this one cout / 16
// with four times the waves each).  wpack16 = the folded [cout][cin] matrix as [cout / 16][cin / 16][16][16].  cin % 64 == 0, cout % 16 == 0,
// hw % 16 == 0, batch * hw % 64 == 0; SE_ERR_BAD_ARG otherwise.  Same arithmetic, another summation order (k steps interleaved over the groups).
static int c1_small(const float* x, const float* wpack16, const float* bias, const float* residual, const float* in_bias, float* out, int batch,
                    int cin, int cout, int hw, int relu, void* stream, int wo) {
    if (batch <= 0 || cin <= 0 || (cin & 63) || cout <= 0 || (cout & 15) || hw <= 0 || (hw & 15) || !x || !wpack16 || !bias || !out) return SE_ERR_BAD_ARG;
    const long long pixels = (long long)batch * hw;
    if (pixels % 64) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    if (cin % 128 == 0) {
        const int trips = cin / 128;
        if (g_variant != 78) {
            if (trips == 2) return launch_c1s<4, 2>(x, wpack16, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s, wo);
            if (trips == 4) return launch_c1s<4, 4>(x, wpack16, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s, wo);
            if (trips == 8) return launch_c1s<4, 8>(x, wpack16, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s, wo);
            if (trips == 16) return launch_c1s<4, 16>(x, wpack16, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s, wo);
        }
        return launch_c1s<4, 0>(x, wpack16, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s, wo);
    }
    return launch_c1s<2, 0>(x, wpack16, bias, residual, in_bias, out, pixels, cin, cout, hw, relu, s, wo);
}

extern "C" int se_conv2d_1x1_small_f32(const float* x, const float* wpack16, const float* bias, const float* residual, const float* in_bias,
                                       float* out, int batch, int cin, int cout, int hw, int relu, void* stream) {
    return c1_small(x, wpack16, bias, residual, in_bias, out, batch, cin, cout, hw, relu, stream, 0);
}

// ... and its stride-2 form: x [batch][cin][2 ho][2 wo] -> out [batch][cout][ho][wo] = W x[:, :, ::2, ::2] + bias (+ ReLU); wo % 4 == 0.
extern "C" int se_conv2d_1x1_small_s2_f32(const float* x, const float* wpack16, const float* bias, float* out, int batch, int cin, int cout, int ho,
                                          int wo, int relu, void* stream) {
    if (ho <= 0 || wo <= 0 || (wo & 3)) return SE_ERR_BAD_ARG;
    return c1_small(x, wpack16, bias, nullptr, nullptr, out, batch, cin, cout, ho * wo, relu, stream, wo);
}
