// 3x3 convolutions (padding 1, stride 1 or 2) of the 2-D backbone's Bottlenecks as a direct float32 MFMA product (round 6).
//
// Stands in for conv2 (`conv3x3`, reference network/pose_resnet.py:22-25, run at :78) of all sixteen Bottlenecks (:52-90).  Built for the
// deep stages - 256 -> 256 at 16 x 16 and 512 -> 512 at 8 x 8 for a 256 x 256 image: few pixels, many channels, 2.4 GFLOP against 2.4 - 9.4 MB
// of weights per launch - and for the three stride-2 layers; on the wide maps of layer1 / layer2 it is level with MIOpen at batch 8 and ahead
// at batch 1.
// MIOpen's picks there: its float32 Winograd assembly kernel (vector ALUs, 38 us) and, for the 8 x 8 maps, an NHWC implicit GEMM between
// three layout transposes and a workspace fill (46 + 20 us).  Here: K = 9 * cin walked as (16-channel step, tap), MFMA rows = the 64
// pixels of a TH x TW tile, MFMA columns = BC output channels; a workgroup is KS groups of four waves which take alternate k steps
// (as in conv2d_1x1.hip) and add their sums through LDS at the end.
//   per group and k step: the X patch [16 channels][(TH + 2) x (TW + 2)] (zero outside the map) and the W block [9 taps][BC][16]
//   (pre-packed: one contiguous run) go through registers into the other half of a double-buffered LDS image; a wave owns one
//   16-pixel row tile and all BC channels: per tap one ds_read_b128 per 16 channels (W[co][4 kg .. 4 kg + 3]) and four
//   ds_read_b32 (X[4 kg + j][pixel + tap offset]) feed 4 * BC / 16 MFMAs.
// Where the 28 us of a 256 -> 256 launch at 16 x 16, B = 8 go (knock-out builds, profiles/r06_conv3x3.txt): the MFMA phases with their LDS
// operand reads alone 22 us (the float32 MFMA peak prices the 2.4 GFLOP at 15.4), the loads + LDS writes + barriers alone 11.6; every
// workgroup streams its 9 * cin * BC * 4 bytes of weights once (295 KB), 256 workgroups at once.
// float32 in, float32 accumulate (v_mfma_f32_16x16x4_f32); the result differs from MIOpen's by summation order only.
#include "common.h"

namespace {

constexpr int C3_LDW = 24;          // floats per (tap, cout) row of the W image (16 k + 8: conflict-free ds_read_b128)

template <int TW, int S>
struct C3Geom {
    static constexpr int TH = 64 / TW, PH = S * TH + 3 - S, PW = S * TW + 3 - S;     // stride 1: (TH + 2) x (TW + 2); stride 2: (2 TH + 1) x (2 TW + 1)
    // floats per channel of the X image: = 4 (mod 16), so the four k groups of a wave read four disjoint 16-bank windows
    static constexpr int XS = (S == 1) ? ((TW == 16) ? 116 : 100) : ((TW == 16) ? 308 : 292);
    static constexpr int X_FLOATS = 16 * XS;
    static_assert(XS >= PH * PW && XS % 16 == 4, "X image stride");
};

// x [B][cin][S H][S W], wpack [cout / BC][cin / 16][9][BC][16], bias (or null) [cout], out [B][cout][H][W] (H, W: the OUTPUT map; S = stride,
// padding 1: output (y, x) reads input rows S y - 1 .. S y + 1); H % TH == 0, W % TW == 0, cin % (16 KS) == 0, cout % BC == 0.
// Grid: (pixel tiles, cout / BC), walked XCD by XCD (see below).
// TRIPS > 0: the group's k loop has exactly TRIPS steps and is unrolled completely with the loads TWO steps ahead (two register sets; see
// conv2d_1x1.hip); TRIPS = 0: the run-time loop, loads one step ahead.
template <int TW, int BC, int KS, int S, int TRIPS>
__global__ __launch_bounds__(256 * KS) void conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
                                                           float* __restrict__ out, int cin, int cout, int H, int W, int relu) {
    using G = C3Geom<TW, S>;
    constexpr int TH = G::TH, PH = G::PH, PW = G::PW, XS = G::XS, XF = G::X_FLOATS;
    constexpr int CT = BC / 16;
    constexpr int WF = 9 * BC * C3_LDW;
    constexpr int REGION = 2 * XF + 2 * WF;
    constexpr int NXE = 16 * PH * PW, NX = (NXE + 255) / 256;            // X patch elements per step, per thread
    constexpr int NWP = 9 * BC * 4, NW = (NWP + 255) / 256;              // 16-byte W pieces per step, per thread
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int grp = KS > 1 ? (int)(threadIdx.x >> 8) : 0;
    float* xs = lds + grp * REGION;
    float* ws = xs + 2 * XF;
    const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
    const int i = lane & 15, kg = lane >> 4;

    // XCD-aware walk: the hardware deals consecutive workgroup ids to the 8 XCDs in turn; id -> (id % 8) * (n / 8) + id / 8 gives every
    // XCD one contiguous run of (pixel tile, cout tile) pairs, pixel tiles fastest: its L2 holds the weights of a few cout tiles only
    const int nx = gridDim.x, n_wg = nx * gridDim.y;
    int id = blockIdx.y * nx + blockIdx.x;
    if ((n_wg & 7) == 0) id = (id & 7) * (n_wg >> 3) + (id >> 3);
    const int tile = id % nx, ctile = id / nx;
    const int tiles_x = W / TW, tiles_y = H / TH;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    const int c0 = ctile * BC;
    const int steps = cin >> 4, trips = steps / KS;
    const long long HW = (long long)H * W, HWI = HW * (S * S);
    const int HI = S * H, WI = S * W;
    const float* xb = x + (long long)b * cin * HWI;

    // this thread's elements of an X patch: e = t + 256 v -> (channel, patch row, patch column); the same every step but for the channel base
    int xoff[NX], xdst[NX];
    unsigned xok = 0;
#pragma unroll
    for (int v = 0; v < NX; ++v) {
        int e = t + 256 * v;
        if (e >= NXE) e = NXE - 1;                                       // the tail repeats the last element (same value, same slot)
        const int ch = e / (PH * PW), rem = e - ch * (PH * PW), r = rem / PW, c = rem - r * PW;
        const int y = S * y0 - 1 + r, xx = S * x0 - 1 + c;
        const bool ok = y >= 0 && y < HI && xx >= 0 && xx < WI;
        xok |= (ok ? 1u : 0u) << v;
        xoff[v] = ch * (int)HWI + (ok ? y * WI + xx : 0);
        xdst[v] = ch * XS + r * PW + c;
    }
    const float* wsrc = wpack + (long long)ctile * steps * (9 * BC * 16);
    int woff[NW], wdst[NW];
#pragma unroll
    for (int v = 0; v < NW; ++v) {
        int q = t + 256 * v;
        if (q >= NWP) q = NWP - 1;
        woff[v] = 4 * q;
        wdst[v] = (q >> 2) * C3_LDW + 4 * (q & 3);
    }
    float xr2[TRIPS > 0 ? 2 : 1][NX];
    f32x4 wr2[TRIPS > 0 ? 2 : 1][NW];
    auto fetch = [&](int s, int set = 0) {
        float (&xr)[NX] = xr2[set];
        f32x4 (&wr)[NW] = wr2[set];
        const float* xc = xb + (long long)s * 16 * HWI;
#pragma unroll
        for (int v = 0; v < NX; ++v) xr[v] = xc[xoff[v]];
        const float* wc = wsrc + (long long)s * (9 * BC * 16);
#pragma unroll
        for (int v = 0; v < NW; ++v) wr[v] = *reinterpret_cast<const f32x4*>(wc + woff[v]);
    };
    auto commit = [&](int buf, int set = 0) {
        float (&xr)[NX] = xr2[set];
        f32x4 (&wr)[NW] = wr2[set];
#pragma unroll
        for (int v = 0; v < NX; ++v) xs[buf * XF + xdst[v]] = ((xok >> v) & 1u) ? xr[v] : 0.f;
#pragma unroll
        for (int v = 0; v < NW; ++v) *reinterpret_cast<f32x4*>(ws + buf * WF + wdst[v]) = wr[v];
    };
    // this lane's pixel of the wave's row tile, as an offset into a channel of the patch (tap (0, 0) = the pixel's upper left neighbour)
    const int ry = (TW == 16) ? wave : 2 * wave + (i >> 3), rx = (TW == 16) ? i : (i & 7);
    const int apix = S * (ry * PW + rx);
    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
            // operands one tap ahead of the MFMAs that use them (two register sets; the scheduling barriers keep hipcc from sinking the
            // reads back down to their first use, where the LDS latency is exposed with two waves per SIMD)
            const float* xa = xs + buf * XF + 4 * kg * XS + apix;
            const float* wa = ws + buf * WF + i * C3_LDW + 4 * kg;
            f32x4 wv[2][CT];
            float av[2][4];
            auto load_tap = [&](int tap, int slot) {
                const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) wv[slot][ct] = *reinterpret_cast<const f32x4*>(wa + (tap * BC + ct * 16) * C3_LDW);
#pragma unroll
                for (int j = 0; j < 4; ++j) av[slot][j] = xa[j * XS + dy * PW + dx];
            };
            load_tap(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) load_tap(tap + 1, (tap + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap & 1][j], wv[tap & 1][ct][j], acc[ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    if (TRIPS > 0) {
        // step n travels in register set n & 1: requested at the top of trip n - 2, written to LDS half n & 1 at the end of trip n - 1
        fetch(grp, 0);
        if (TRIPS > 1) fetch(KS + grp, 1);
        commit(0, 0);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < TRIPS; ++s) {
            if (s + 2 < TRIPS) fetch((s + 2) * KS + grp, s & 1);
            compute(s & 1);
            if (s + 1 < TRIPS) commit((s + 1) & 1, (s + 1) & 1);
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
        // the other groups' sums travel through their own LDS regions (free after the last barrier); group 0 finishes
        static_assert(CT * 1024 <= REGION, "exchange image");
        if (grp != 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) reinterpret_cast<f32x4*>(xs)[ct * 256 + t] = acc[ct];
        }
        __syncthreads();
        if (grp != 0) return;
#pragma unroll
        for (int o = 1; o < KS; ++o)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] += reinterpret_cast<const f32x4*>(lds + o * REGION)[ct * 256 + t];
    }
    // lane (channel i of a 16-channel tile, pixels 4 kg .. 4 kg + 3 of the row tile = 4 consecutive x of one map row): 16-byte NCHW stores
    const int py = (TW == 16) ? wave : 2 * wave + (kg >> 1), px = (TW == 16) ? 4 * kg : 4 * (kg & 1);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int co = c0 + ct * 16 + i;
        f32x4 v = acc[ct];
        if (bias) v += bias[co];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<f32x4*>(out + ((long long)b * cout + co) * HW + (long long)(y0 + py) * W + x0 + px) = v;
    }
}

template <int TW, int BC, int KS, int S, int TRIPS>
int launch_c3t(const float* x, const float* wpack, const float* bias, float* out, int batch, int cin, int cout, int H, int W, int relu, hipStream_t s) {
    using G = C3Geom<TW, S>;
    constexpr int LDS = KS * (2 * G::X_FLOATS + 2 * 9 * BC * C3_LDW) * 4;
    static_assert(LDS <= 160 * 1024, "LDS image");
    auto kern = conv3x3_kernel<TW, BC, KS, S, TRIPS>;
    SE_ENSURE_LDS(kern, LDS);
    const dim3 grid((unsigned)(batch * (H / G::TH) * (W / TW)), cout / BC);
    hipLaunchKernelGGL(kern, grid, dim3(256 * KS), LDS, s, x, wpack, bias, out, cin, cout, H, W, relu);
    SE_CHECK_LAUNCH();
    return 0;
}

// unrolled forms for the stride-1 trip counts of the backbone (4 / 8: 64 / 128 channels in one group, 8 / 16: 256 / 512 over two,
// 2 / 4 / 8: 128 / 256 / 512 over four at batch 1-2);
// se_debug_set_variant(78): never
template <int TW, int BC, int KS, int S = 1>
int launch_c3(const float* x, const float* wpack, const float* bias, float* out, int batch, int cin, int cout, int H, int W, int relu, hipStream_t s) {
    const int trips = (cin >> 4) / KS;
    if (S == 1 && g_variant != 78) {
        constexpr bool U = (S == 1);
        if (trips == 2) return launch_c3t<TW, BC, KS, S, U ? 2 : 0>(x, wpack, bias, out, batch, cin, cout, H, W, relu, s);
        if (trips == 4) return launch_c3t<TW, BC, KS, S, U ? 4 : 0>(x, wpack, bias, out, batch, cin, cout, H, W, relu, s);
        if (trips == 8) return launch_c3t<TW, BC, KS, S, U ? 8 : 0>(x, wpack, bias, out, batch, cin, cout, H, W, relu, s);
        if (trips == 16) return launch_c3t<TW, BC, KS, S, U ? 16 : 0>(x, wpack, bias, out, batch, cin, cout, H, W, relu, s);
    }
    return launch_c3t<TW, BC, KS, S, 0>(x, wpack, bias, out, batch, cin, cout, H, W, relu, s);
}

}  // namespace

// Channel-tile width of the packed weights ([cout / BC][cin / 16][9][BC][16] from the folded [cout][cin][3][3] tensor); 0 = shape not
// covered.  Covered: cin % 32 == 0, cout % 32 == 0 and a map of 8 x 8 or with h % 4 == 0, w % 16 == 0; the tile is 16 channels when 32
// would leave CUs without a workgroup.  Depends on the arguments and the device's CU count only.
extern "C" int se_conv2d_3x3_tile_f32(int batch, int cin, int cout, int h, int w) {
    if (batch <= 0 || cin <= 0 || (cin & 31) || cout <= 0 || (cout & 31) || h <= 0 || w <= 0) return 0;
    const bool t8 = (w % 16 != 0);
    if (t8 ? (w % 8 != 0 || h % 8 != 0) : (h % 4 != 0)) return 0;
    if ((long long)cin * h * w >= (1LL << 31) / 4) return 0;            // 32-bit element offsets within a sample
    const long long tiles = (long long)batch * h * w / 64;
    if (tiles * (cout / 16) > (1LL << 30)) return 0;
    return tiles * (cout / 32) >= se_num_cus() ? 32 : 16;
}

// out = relu?(conv3x3(x) (+ bias)): stride 1, zero padding 1.  bias may be NULL (the consumer applies it: se_conv2d_1x1_f32's in_bias).
extern "C" int se_conv2d_3x3_f32(const float* x, const float* wpack, const float* bias, float* out, int batch, int cin, int cout, int h, int w,
                                 int relu, void* stream) {
    const int bc = se_conv2d_3x3_tile_f32(batch, cin, cout, h, w);
    if (!bc || !x || !wpack || !out) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    const bool t8 = (w % 16 != 0);
    // one wave group where the grid alone fills the device (two or more workgroups per CU: the wide maps of layer1 / layer2 - 27.9 / 25.8 us
    // against 32.0 / 27.8 with two groups); se_debug_set_variant(76): always one group, (79): never (A/B in development builds)
    const long long wgs0 = ((long long)batch * h * w / 64) * (cout / bc);
    if (g_variant == 76 || (wgs0 >= 2LL * se_num_cus() && g_variant != 79)) {
        if (t8) return bc == 32 ? launch_c3<8, 32, 1>(x, wpack, bias, out, batch, cin, cout, h, w, relu, s) : launch_c3<8, 16, 1>(x, wpack, bias, out, batch, cin, cout, h, w, relu, s);
        return bc == 32 ? launch_c3<16, 32, 1>(x, wpack, bias, out, batch, cin, cout, h, w, relu, s) : launch_c3<16, 16, 1>(x, wpack, bias, out, batch, cin, cout, h, w, relu, s);
    }
    // few workgroups (batch 1-2): four wave groups on 8 x 8 tiles - a workgroup's k steps are a chain of load latencies there, not of MFMAs
    const long long wgs = ((long long)batch * h * w / 64) * (cout / bc);
    if (bc == 16 && h % 8 == 0 && w % 8 == 0 && cin % 64 == 0 && 2 * wgs <= se_num_cus() && g_variant != 77)
        return launch_c3<8, 16, 4>(x, wpack, bias, out, batch, cin, cout, h, w, relu, s);
    if (t8) return bc == 32 ? launch_c3<8, 32, 2>(x, wpack, bias, out, batch, cin, cout, h, w, relu, s) : launch_c3<8, 16, 2>(x, wpack, bias, out, batch, cin, cout, h, w, relu, s);
    return bc == 32 ? launch_c3<16, 32, 2>(x, wpack, bias, out, batch, cin, cout, h, w, relu, s) : launch_c3<16, 16, 2>(x, wpack, bias, out, batch, cin, cout, h, w, relu, s);
}

// The stride-2 form (conv2 of the first Bottleneck of layer2 / layer3 / layer4, network/pose_resnet.py:78 with stride 2):
// x [batch][cin][2 ho][2 wo] -> out [batch][cout][ho][wo], padding 1.  wpack = [cout / 16][cin / 16][9][16][16] (channel tile 16: the input
// patch of a 64-pixel output tile is 2.7 times the stride-1 one and leaves the LDS room for 16 channels of weights per wave group).
// Covered: cin % 32 == 0, cout % 16 == 0 and an output map of 8k x 8m or 4k x 16m; SE_ERR_BAD_ARG otherwise.
extern "C" int se_conv2d_3x3_s2_f32(const float* x, const float* wpack, const float* bias, float* out, int batch, int cin, int cout, int ho, int wo,
                                    int relu, void* stream) {
    if (batch <= 0 || cin <= 0 || (cin & 31) || cout <= 0 || (cout & 15) || ho <= 0 || wo <= 0 || !x || !wpack || !out) return SE_ERR_BAD_ARG;
    const bool t8 = (wo % 16 != 0);
    if (t8 ? (wo % 8 != 0 || ho % 8 != 0) : (ho % 4 != 0)) return SE_ERR_BAD_ARG;
    if ((long long)cin * ho * wo * 4 >= (1LL << 31) / 4) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    if (t8) return launch_c3<8, 16, 2, 2>(x, wpack, bias, out, batch, cin, cout, ho, wo, relu, s);
    return launch_c3<16, 16, 2, 2>(x, wpack, bias, out, batch, cin, cout, ho, wo, relu, s);
}
