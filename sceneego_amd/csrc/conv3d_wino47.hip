// 7x7x7 convolution (V2V front layer, cout = 16) with the 1-D Winograd transform F(4,7) along z on the f32 MFMA (production):
// 10 multiplies per 4 z-neighbouring outputs instead of 28 — 1.6x fewer MFMAs than the F(2,7) kernel in conv3d_wino.hip, 2.8x
// fewer than the direct form.  Points {0, +-1, +-2, +-1/2, +-3/4, inf} (tools/wino47_matrices.py -> wino47_matrices.h); float32
// error of the transform on N(0,1) data: 3.6e-6 mean, 3.2e-5 max per 7-tap dot product (F(2,7): 8.6e-7 / 8.2e-6; direct float32:
// 1.4e-7 / 7.8e-7).  Replaces Basic3DBlock(33, 16, 7) of network/v2v.py:75-77 (conv + folded BN + ReLU).
//
// Structure: one persistent 512-thread workgroup per CU, chunk-outer (the 97.5 KB of transformed weights of a 3-channel chunk are
// loaded into LDS once per chunk; 33 input channels = 11 chunks exactly), y-domain partial sums through the output tensor, B^T
// applied once per element when the halo is committed, G folded into the packed weights (section F of se_conv3d_pack_f32), A^T in
// the epilogue.  A work unit is an 8(z) x 8 x 8 output tile = two z quads; its transformed halo is 2 x 14 x 14 columns.
//
// LDS layouts are built for 16-byte reads (12-byte operands compile to ds_read2_b32 + ds_read_b32):
//   weights  [g(13)] { Q0 [lane][xi 0..3][3] , Q1 [lane][xi 4..7][3] , T [lane][xi 8..9][3] }     lane stride 48 B / 24 B
//   inputs   [z quad][column][12 xi slots (10 used)][3]                                             column stride 144 B
// so a step (tap group g, xi quad) is 3 + 3 ds_read_b128 and 12 MFMAs (tail: 6).  Wave w = row w of the tile; its 16 positions are
// the 8 x of that row in both z quads, and the z-quad stride is 200 columns (= 8 mod 16): a 16-lane ds_read_b128 group then hits 16
// distinct (column mod 16) classes = 16 distinct 4-bank groups, i.e. no bank conflicts (SQ_LDS_BANK_CONFLICT 8 % of LDS cycles).
//
// Input layout (compile-time, one translation unit each: -DSE_K7F_PLANAR=0 / 1):
//   0  channels-last [B][D][D][D][cin_pad]: a wave's 64 halo columns are 64 different cache lines per load instruction; the VMEM
//      issue of the 10 loads per thread costs the MFMA waves ~7k of ~33k cycles per item (measured by removing the fetch);
//   1  triplet-planar [B][chunks][D][D][D][3] (SE_IN_PLANAR3, written by se_unproject_gather_planar3_f32 / se_voxelize_planar3_f64):
//      14 consecutive columns of a halo row are 168 contiguous bytes, ~12 lines per instruction.
#include "conv_common.h"
#include "wino47_matrices.h"

#include <type_traits>
#include <utility>

#ifndef SE_K7F_PLANAR
#error "compile with -DSE_K7F_PLANAR=0 (channels-last input) or 1 (triplet-planar input)"
#endif

namespace {

template <typename F, int... S>
__device__ __forceinline__ void for_each_index(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}

// global -> LDS copy of n4 16-byte pieces by NT threads, 8 loads in flight per thread
template <int NT>
__device__ __forceinline__ void fill_lds(float* dst, const f32x4* __restrict__ src, int n4, int tid) {
    for (int base = 0; base < n4; base += NT * 8) {
        f32x4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + k * NT + tid;
            t[k] = src[i < n4 ? i : 0];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + k * NT + tid;
            if (i < n4) reinterpret_cast<f32x4*>(dst)[i] = t[k];
        }
    }
}

constexpr bool PLANAR = SE_K7F_PLANAR != 0;
#if SE_K7F_PLANAR
#define conv3d_k7_wino47_kernel conv3d_k7_wino47p3_kernel     // distinct kernel names in profiles
#endif
constexpr int K7_HY = 14, K7_HX = 14, K7_COLS = K7_HY * K7_HX;                // 196 halo columns of an 8 x 8 tile
struct f32x3 { float x, y, z; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int K7F_SLOT = 36;                                                  // floats per halo column: 12 xi slots x 3 channels (10 used)
constexpr int K7F_ZQ_COLS = 200;                                              // column stride between the two z quads: = 8 (mod 16)
constexpr int K7F_VT_FLOATS = (K7F_ZQ_COLS + K7_COLS) * K7F_SLOT;             // 14256 floats = 57024 B
constexpr int K7F_W_FLOATS = SE_K7F_CHUNK_FLOATS;                             // 24960 floats = 99840 B
constexpr int K7F_STEPS = SE_K7W_GROUPS * 3;                                  // (tap group, xi quad 0 / quad 1 / tail)

struct K7FOps {            // operands of one step: 4 xi x 3 channels of weights (A) and of transformed inputs (B); the tail uses half
    f32x4 a0, a1, a2, b0, b1, b2;
};

#ifdef SE_STAMP47
#define T47(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define T47(var)
#endif

__global__ __launch_bounds__(512) void conv3d_k7_wino47_kernel(ConvArgs a, int tiles_per_dim, int ztiles, int total_tiles,
                                                               int units_per_wg, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* vt = lds + K7F_W_FLOATS;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4* utab = reinterpret_cast<i32x4*>(lds + K7F_W_FLOATS + K7F_VT_FLOATS);
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, s_mfma = 0, s_b1 = 0, s_stage = 0, s_b2 = 0;
    (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)s_mfma; (void)s_b1; (void)s_stage; (void)s_b2; (void)dbg;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int vl = lane & 15;
    const int h = lane >> 4;
    const int dim = a.dim;
    const int chunks = (a.cin + 2) / 3;
    const int u_begin = (int)blockIdx.x * units_per_wg;
    const int u_end = min(u_begin + units_per_wg, total_tiles);
    if (u_begin >= u_end) return;
    const int n = u_end - u_begin;

    for (int i = tid; i < n; i += 512) {
        int t = u_begin + i;
        i32x4 e;
        e.w = t % tiles_per_dim; t /= tiles_per_dim;
        e.z = t % tiles_per_dim; t /= tiles_per_dim;
        e.y = t % ztiles; t /= ztiles;
        e.x = t;
        utab[i] = e;
    }

    // compute role: wave = row of the 8x8 tile, 16 positions = the 8 x of that row in both z quads
    const int zq = vl >> 3;
    const int ry = wave;
    const int rx = vl & 7;
    int toff[SE_K7W_GROUPS];       // per-lane LDS offsets (floats) of the 13 tap groups: tap 4g+h -> (dy,dx)
#pragma unroll
    for (int g = 0; g < SE_K7W_GROUPS; ++g) {
        int tap = 4 * g + h;
        tap = tap < 49 ? tap : 0;   // zero-weight padding
        toff[g] = (zq * K7F_ZQ_COLS + (ry + tap / 7) * K7_HX + rx + tap % 7) * K7F_SLOT;
    }

    // staging role: thread t < 392 owns halo column (t % 196) of z quad (t / 196): 10 raw slabs -> 10 transformed slabs
    const bool s_on = tid < 2 * K7_COLS;
    const int s_col = tid % K7_COLS, s_zq = s_on ? tid / K7_COLS : 0;
    const int s_cy = s_col / K7_HX, s_cx = s_col % K7_HX;
    f32x3 raw[10];
    // halo fetch of one item, split so that the 10 loads can be issued one per MFMA step (-DSE_K47_SPREAD) instead of back to
    // back at step 1: measured no difference (2.375 vs 2.365 ms per launch inside bench.py), the back-to-back form is production
    long long f_base = 0;
    int f_gz0 = 0;
    bool f_okc = false;
    const long long f_zs = (long long)dim * dim * (PLANAR ? 3 : a.cin_pad);
    auto fetch_setup = [&](int k, int c) {
        const i32x4 e = utab[k];
        const int gy = e.z * 8 - 3 + s_cy, gx = e.w * 8 - 3 + s_cx;
        f_gz0 = e.y * 8 + 4 * s_zq - 3;
        f_okc = s_on && (unsigned)gy < (unsigned)dim && (unsigned)gx < (unsigned)dim;
        f_base = PLANAR ? ((((long long)e.x * chunks + c) * dim * dim + gy) * dim + gx) * 3
                        : ((((long long)e.x * dim) * dim + gy) * dim + gx) * a.cin_pad + c * 3;
    };
    auto fetch_one = [&](auto q_tag) {   // out-of-volume taps load the buffer's first record (one cache line for all of them) and are zeroed
        constexpr int q = decltype(q_tag)::value;
        const bool ok = f_okc && (unsigned)(f_gz0 + q) < (unsigned)dim;
        const f32x3 t = *reinterpret_cast<const f32x3*>(a.in + (ok ? f_base + (f_gz0 + q) * f_zs : 0));
        raw[q].x = ok ? t.x : 0.f; raw[q].y = ok ? t.y : 0.f; raw[q].z = ok ? t.z : 0.f;
    };
    auto fetch = [&](int k, int c) {
        fetch_setup(k, c);
        for_each_index(fetch_one, std::make_integer_sequence<int, 10>{});
    };
    auto commit = [&]() {   // V = B^T d (rows 1..8 come in +- pairs: even-q part + / - odd-q part); 30 floats per column, 16-byte stores
        if (!s_on) return;
        float o[32];
#pragma unroll
        for (int i = 30; i < 32; ++i) o[i] = 0.f;
        {   // xi = 0: even q only; xi = 9: odd q only
            float ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                if (SE_W47_BT[0][q] != 0.f) { ax += SE_W47_BT[0][q] * raw[q].x; ay += SE_W47_BT[0][q] * raw[q].y; az += SE_W47_BT[0][q] * raw[q].z; }
                if (SE_W47_BT[9][q] != 0.f) { bx += SE_W47_BT[9][q] * raw[q].x; by += SE_W47_BT[9][q] * raw[q].y; bz += SE_W47_BT[9][q] * raw[q].z; }
            }
            o[0] = ax; o[1] = ay; o[2] = az;
            o[27] = bx; o[28] = by; o[29] = bz;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int xi = 2 * p + 1;
            float ex = 0.f, ey = 0.f, ez = 0.f, ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
            for (int q = 1; q < 9; ++q) {
                const float cf = SE_W47_BT[xi][q];
                if (q & 1) { ox += cf * raw[q].x; oy += cf * raw[q].y; oz += cf * raw[q].z; }
                else { ex += cf * raw[q].x; ey += cf * raw[q].y; ez += cf * raw[q].z; }
            }
            o[3 * xi] = ex + ox; o[3 * xi + 1] = ey + oy; o[3 * xi + 2] = ez + oz;
            o[3 * xi + 3] = ex - ox; o[3 * xi + 4] = ey - oy; o[3 * xi + 5] = ez - oz;
        }
        f32x4* dst = reinterpret_cast<f32x4*>(vt + (s_zq * K7F_ZQ_COLS + s_col) * K7F_SLOT);
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = (f32x4){o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
    };
    auto load_weights = [&](int c) {
        fill_lds<512>(wl, reinterpret_cast<const f32x4*>(a.wpack_f + (size_t)c * K7F_W_FLOATS), K7F_W_FLOATS / 4, tid);
    };
    auto out_offset = [&](int k) -> long long {
        const i32x4 e = utab[k];
        const int oz = e.y * 8 + 4 * zq, oy = e.z * 8 + ry, ox = e.w * 8 + rx;
        return ((((long long)e.x * dim + oz) * dim + oy) * dim + ox) * 16 + 4 * h;
    };
    const long long zstride = (long long)dim * dim * 16;

    auto lds_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    const bool relu = a.flags & SE_EPI_RELU;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bpack + 4 * h);
    const bool lone = n == 1;

    __syncthreads();   // utab
    fetch(0, 0);
    commit();
    load_weights(0);
    __syncthreads();

    // operands of step (g, part): part 0 / 1 = xi quads (48 B = 3 x ds_read_b128 each for A and B), part 2 = tail xi 8,9 (24 B)
    auto read_ops = [&](K7FOps& r, auto g_tag, auto p_tag) {
        constexpr int g = decltype(g_tag)::value, part = decltype(p_tag)::value;
        if constexpr (part < 2) {
            const f32x4* ap = reinterpret_cast<const f32x4*>(wl + g * 1920 + part * 768 + lane * 12);
            const f32x4* bp = reinterpret_cast<const f32x4*>(vt + toff[g] + part * 12);
            r.a0 = ap[0]; r.a1 = ap[1]; r.a2 = ap[2];
            r.b0 = bp[0]; r.b1 = bp[1]; r.b2 = bp[2];
        } else {
            const float* ap = wl + g * 1920 + 1536 + lane * 6;
            const float* bp = vt + toff[g] + 24;
            const f32x2 a01 = *reinterpret_cast<const f32x2*>(ap), a23 = *reinterpret_cast<const f32x2*>(ap + 2), a45 = *reinterpret_cast<const f32x2*>(ap + 4);
            r.a0 = (f32x4){a01.x, a01.y, a23.x, a23.y};
            r.a1 = (f32x4){a45.x, a45.y, 0.f, 0.f};
            r.b0 = *reinterpret_cast<const f32x4*>(bp);
            const f32x2 b45 = *reinterpret_cast<const f32x2*>(bp + 4);
            r.b1 = (f32x4){b45.x, b45.y, 0.f, 0.f};
        }
    };

    f32x4 part[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) part[z] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n_items = chunks * n;
    for (int item = 0; item < n_items; ++item) {
        const int c = item / n, k = item - c * n;
        const bool has_next = item + 1 < n_items;
        const int c_next = has_next ? (item + 1) / n : c;
        const int k_next = has_next ? (item + 1) - c_next * n : k;
        const bool last_chunk = c == chunks - 1;
        const long long o0 = out_offset(k);
        T47(t0);
        if (c > 0) {
            if (lone) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // same tile as the previous item: stores first
#pragma unroll
            for (int z = 0; z < 4; ++z) part[z] = *reinterpret_cast<const f32x4*>(a.out + o0 + z * zstride);
        }

        f32x4 acc[SE_K7F_XI];
#pragma unroll
        for (int x = 0; x < SE_K7F_XI; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
        K7FOps cur, nxt;
        read_ops(cur, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        nxt = cur;
        auto step = [&](auto s_tag) {
            constexpr int S = decltype(s_tag)::value;
            constexpr int p = S % 3;
#ifdef SE_K47_SPREAD
            if constexpr (S == 1) fetch_setup(k_next, c_next);   // next item's raw columns: one global load per step under the MFMAs
            if constexpr (S >= 2 && S < 12) fetch_one(std::integral_constant<int, S - 2>{});
#else
            if constexpr (S == 1) fetch(k_next, c_next);   // next item's raw columns: global loads under the MFMAs
#endif

            if constexpr (S + 1 < K7F_STEPS) read_ops(nxt, std::integral_constant<int, (S + 1) / 3>{}, std::integral_constant<int, (S + 1) % 3>{});
            const float av[12] = {cur.a0.x, cur.a0.y, cur.a0.z, cur.a0.w, cur.a1.x, cur.a1.y, cur.a1.z, cur.a1.w, cur.a2.x, cur.a2.y, cur.a2.z, cur.a2.w};
            const float bv[12] = {cur.b0.x, cur.b0.y, cur.b0.z, cur.b0.w, cur.b1.x, cur.b1.y, cur.b1.z, cur.b1.w, cur.b2.x, cur.b2.y, cur.b2.z, cur.b2.w};
            constexpr int NX = p < 2 ? 4 : 2;
            // channel-outer: consecutive MFMAs go to different accumulators (a dependent MFMA issues 40 cycles after its producer)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                constexpr int x0 = 4 * p;
#pragma unroll
                for (int i = 0; i < NX; ++i)
                    acc[x0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3 * i + j], bv[3 * i + j], acc[x0 + i], 0, 0, 0);
            }
            if constexpr (S + 1 < K7F_STEPS) {
                constexpr int pn = (S + 1) % 3;
                // spread the next step's reads between this step's MFMAs
                if constexpr (pn < 2) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NX, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NX, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NX, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NX, 0);
            }
            cur = nxt;
        };
        for_each_index(step, std::make_integer_sequence<int, K7F_STEPS>{});
        T47(t1);

        // A^T (4 x 10): xi 1..8 in +- pairs
        f32x4 y[4];
        {
            const f32x4 s1 = acc[1] + acc[2], d1 = acc[1] - acc[2], s2 = acc[3] + acc[4], d2 = acc[3] - acc[4];
            const f32x4 s3 = acc[5] + acc[6], d3 = acc[5] - acc[6], s4 = acc[7] + acc[8], d4 = acc[7] - acc[8];
            y[0] = acc[0] + (s1 + s2) + (s3 + s4);
            y[1] = d1 + 2.f * d2 + 0.5f * d3 + 0.75f * d4;
            y[2] = s1 + 4.f * s2 + 0.25f * s3 + 0.5625f * s4;
            y[3] = d1 + 8.f * d2 + 0.125f * d3 + 0.421875f * d4 + acc[9];
        }
        if (c > 0) {
#pragma unroll
            for (int z = 0; z < 4; ++z) y[z] += part[z];
        }
        // single transformed tile: every wave must be done reading it before the next item's columns are committed.
        // Barriers inside the loop wait for this wave's LDS traffic only (a __syncthreads() also waits for vmcnt(0): the output
        // stores of the item would have to be acknowledged before the next MFMA phase may start); register-only arithmetic must
        // not drift across them either.
        lds_barrier();
        T47(t2);
        if (has_next) {
            commit();
            if (c_next != c) load_weights(c_next);
        }
        if (last_chunk) {
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                y[z] += bias;
                if (relu) {
                    y[z].x = fmaxf(y[z].x, 0.f); y[z].y = fmaxf(y[z].y, 0.f); y[z].z = fmaxf(y[z].z, 0.f); y[z].w = fmaxf(y[z].w, 0.f);
                }
            }
        }
#pragma unroll
        for (int z = 0; z < 4; ++z) *reinterpret_cast<f32x4*>(a.out + o0 + z * zstride) = y[z];
        T47(t3);
        if (!has_next) break;
        lds_barrier();
        T47(t4);
#ifdef SE_STAMP47
        s_mfma += t1 - t0; s_b1 += t2 - t1; s_stage += t3 - t2; s_b2 += t4 - t3;
#endif
    }
#ifdef SE_STAMP47
    if (lane == 0 && dbg) {
        unsigned long long* o = dbg + ((size_t)blockIdx.x * 8 + wave) * 6;
        o[0] = s_mfma; o[1] = s_b1; o[2] = s_stage; o[3] = s_b2; o[4] = n_items; o[5] = 0;
    }
#endif
}

}  // namespace

// Returns 0 on launch, SE_TILED_NOT_TAKEN if the unit table does not fit, else a hipError_t.  Preconditions (checked by the caller,
// se_conv3d_k7_wino_try): ksize 7, cout 16, dim % 8 == 0, dim >= 16, no residual, channels-last output, a.wpack_f set.
#if SE_K7F_PLANAR
int se_conv3d_k7_wino47_launch_p3(const ConvArgs& a, int batch, int num_cus, hipStream_t s, unsigned long long* dbg) {
#else
int se_conv3d_k7_wino47_launch_cl(const ConvArgs& a, int batch, int num_cus, hipStream_t s, unsigned long long* dbg) {
#endif
    constexpr int LDS_BYTES = 160 * 1024;
    constexpr int LDS_FIXED = (K7F_W_FLOATS + K7F_VT_FLOATS) * 4;
    constexpr int MAX_UNITS = (LDS_BYTES - LDS_FIXED) / 16;
    SE_ENSURE_LDS(conv3d_k7_wino47_kernel, LDS_BYTES);
    const int tl = a.dim / 8;
    const int total = batch * tl * tl * tl;
    const int grid = total < num_cus ? total : num_cus;
    const int per = (total + grid - 1) / grid;
    if (per > MAX_UNITS) return SE_TILED_NOT_TAKEN;
    hipLaunchKernelGGL(conv3d_k7_wino47_kernel, dim3((total + per - 1) / per), dim3(512), LDS_BYTES, s, a, tl, tl, total, per, dbg);
    SE_CHECK_LAUNCH();
    return 0;
}
