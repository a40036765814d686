// 3x3x3 convolution with a 1-D Winograd F(2,3) transform along z, on the f32 MFMA — the V2V workhorse.
//
// For two output voxels that are neighbours in z, (y0, y1), and the four inputs d0..d3 on that z line,
//     m0 = (d0 - d2) g0      m1 = (d1 + d2)(g0+g1+g2)/2      m2 = (d2 - d1)(g0-g1+g2)/2      m3 = (d1 - d3) g2
//     y0 = m0 + m1 + m2      y1 = m1 - m2 - m3
// i.e. 4 multiplies instead of 6 per (dy, dx, cin, cout): the MFMA work of a 3x3x3 layer drops by 1.5x.  All of
// it is linear, so the four M_xi = sum over (dy, dx, cin) of U_xi * V_xi are accumulated by the matrix cores
// (U_xi: transformed weights, packed by se_conv3d_pack_f32 section C; V_xi: 3 packed adds per lane on the four
// 16-byte activation fragments it has just read from LDS) and only the epilogue forms y0 / y1.
// The LDS traffic per MFMA is the same as for the direct form (one ds_read_b128 per 4 MFMAs per operand), the
// halo tile is the same 6 x 10 x 10, and no transformed tensor is ever materialised.
//
// Structure (as conv3d_k7_persistent_kernel): one persistent 512-thread workgroup per CU; a work unit is
// (32-cout block, 4x8x8 output tile); loop order cout block -> 16-channel chunk -> unit, so the 72 KB of
// transformed weights of a (block, chunk) are loaded once and the halo tiles are double buffered (one barrier per
// item).  For cin > 16 the per-chunk results are summed through the output tensor (read y-partial, add, write),
// prefetched one item ahead; residual add + ReLU happen on the last chunk.
// Wave w of 8: z pair w>>2 (outputs z = 2*zp, 2*zp+1 of the tile), rows 2*(w&3), 2*(w&3)+1 -> one 16-position tile,
// 4 xi x 2 cout tiles = 8 accumulators.
#include "conv_common.h"

#include <type_traits>
#include <utility>

// F(4,7) 7^3 kernel, one translation unit per input layout (conv3d_wino47.hip compiled with -DSE_K7F_PLANAR=0 / 1)
int se_conv3d_k7_wino47_launch_cl(const ConvArgs& a, int batch, int num_cus, hipStream_t s, unsigned long long* dbg);
int se_conv3d_k7_wino47_launch_p3(const ConvArgs& a, int batch, int num_cus, hipStream_t s, unsigned long long* dbg);
int se_conv3d_k7_wino67_launch(const ConvArgs& a, int batch, int num_cus, hipStream_t s, unsigned long long* dbg);   // conv3d_wino67.hip

namespace {

constexpr int HZ = 6, HY = 10, HX = 10, HV = HZ * HY * HX;   // halo voxels of a 4x8x8 tile
[[maybe_unused]] constexpr int TILE_FLOATS = HV * 16;                          // one 16-channel chunk
[[maybe_unused]] constexpr int PF = (HV * 4 + 511) / 512;                      // 16-byte pieces per thread (5)

template <typename F, int... S>
__device__ __forceinline__ void for_each_index(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}

// global -> LDS copy of N4 16-byte pieces by NT threads, 8 loads in flight per thread (a plain loop is compiled to
// load / s_waitcnt vmcnt(0) / ds_write per iteration: one exposed L2 round trip per 16 bytes per thread).
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

struct WinoIter {   // uniform walk over this workgroup's items: runs of units with equal cout block, chunk-outer
    int u_lo, n, cb, c, k;
    bool valid;
};

#ifdef SE_DEVTOOLS   // retired A/B variants: F(2,3) 3^3 kernel, F(2,7) 7^3 kernels (single-phase and ping-pong)
#include "devtools/wino_f23_f27_kernels.inc"
#endif  // SE_DEVTOOLS (retired F(2,3) / F(2,7) kernels)
// ------------------------------------------------------------------------------------------------
// 3x3x3 convolution with 1-D Winograd F(4,3) along z: 6 multiplies per 4 z-neighbouring outputs instead of 12
// -> HALF the MFMAs of the direct form (F(2,3) above: 2/3).  Lavin-Gray matrices, points {0, +-1, +-2, inf}:
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]          (G is folded into the packed weights)
// A 4x8x8 output tile is exactly one z quad; its 6 raw input slabs become 6 transformed slabs V_xi when the halo is
// committed to LDS (B^T applied once per element, as in the 7^3 kernel), so the inner loop is LDS reads + MFMAs only.
// LDS: 108 KB transformed weights of one (32-cout block, 16-channel chunk) + ONE 6 x 10 x 10 x 16-channel V tile
// (38.4 KB).  Wave w of 8: 16 positions (rows 2(w&3), +1), cout tile w>>2 -> 6 accumulators (one per xi).
// fp32 error of the transform ~4e-7 mean / 5e-6 max per 3-tap dot product (tools/wino_matrices.py).
// ------------------------------------------------------------------------------------------------
constexpr int W43_FLOATS = SE_WINO43_CHUNK_FLOATS;
[[maybe_unused]] constexpr int V43_FLOATS = 6 * HY * HX * 16;

#ifdef SE_DEVTOOLS   // retired A/B variant: single-phase form of the F(4,3) kernel
#include "devtools/wino_f43_single_phase_kernel.inc"
#endif  // SE_DEVTOOLS (single-phase F(4,3) kernel)

// ------------------------------------------------------------------------------------------------
// F(4,3) kernel, "ping-pong" form (production; 4-7 % faster than the single-phase form in one-process A/B runs): the same arithmetic, weights and work units as conv3d_k3_wino43_kernel,
// but the 8 waves form two groups of 4 (one wave per SIMD each) that work on the two y-halves (4 rows) of every 4x8x8 tile
// HALF A PHASE APART: while group A issues the 216 MFMAs of its half tile, group B runs everything else for its own
// half — A^T, partial-sum add, epilogue, output stores and the B^T transform + LDS commit of its next half tile — and in the
// next phase the roles swap.  One workgroup barrier per phase.  In the single-phase kernel all 8 waves do these things at
// the same time, so the matrix pipe idles through every staging step (MFMA busy 62 % of the launch); here a SIMD always has
// one wave in its MFMA block.  A lone MFMA wave per SIMD must not issue dependent MFMAs back to back (40-cycle latency vs
// 32-cycle issue), so two (tap, xi) sub-steps are interleaved on two different accumulators.
// LDS: 108 KB weights + 2 x 23 KB half V tiles (6 slabs x 6 rows x 10 columns x 16 channels) + unit table.
// ------------------------------------------------------------------------------------------------
constexpr int PP_ROWS = 6;                                    // 4 rows + halo
constexpr int PP_COLS = PP_ROWS * HX;                         // 60 halo columns per half tile
constexpr int PP_VH_FLOATS = 6 * PP_COLS * 16;                // 5760 floats = 23040 B

#ifdef SE_STAMPPP
#define PP_T(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define PP_T(var)
#endif

__global__ __launch_bounds__(512) void conv3d_k3_wino43pp_kernel(ConvArgs a, int tiles_per_dim, int ztiles, int total_tiles,
                                                                 int n_cb, int units_per_wg, int mode, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    (void)mode;
    unsigned long long t_a = 0, t_b = 0, t_c = 0, acc_mfma = 0, acc_stage = 0, acc_bar = 0, acc_idle = 0, n_ph = 0;
    (void)t_a; (void)t_b; (void)t_c; (void)acc_mfma; (void)acc_stage; (void)acc_bar; (void)acc_idle; (void)n_ph; (void)dbg;
    float* wl = lds;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4* utab = reinterpret_cast<i32x4*>(lds + W43_FLOATS + 2 * PP_VH_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int G = wave >> 2;                      // group: y half of the tile
    const int wg = wave & 3;
    const int vl = lane & 15;
    const int h = lane >> 4;
    const int dim = a.dim;
    const int chunks = a.cin >> 4;
    const int n_units = n_cb * total_tiles;
    const int u_begin = (int)blockIdx.x * units_per_wg;
    const int u_end = min(u_begin + units_per_wg, n_units);
    if (u_begin >= u_end) return;
    float* vt = lds + W43_FLOATS + G * PP_VH_FLOATS;

    for (int i = tid; i < u_end - u_begin; i += 512) {
        int t = (u_begin + i) % total_tiles;
        i32x4 e;
        e.w = t % tiles_per_dim; t /= tiles_per_dim;
        e.z = t % tiles_per_dim; t /= tiles_per_dim;
        e.y = t % ztiles; t /= ztiles;
        e.x = t;
        utab[i] = e;
    }

    // compute role inside the group: cout tile nt, rows 2*(wg&1), +1 of the half tile
    const int nt = wg >> 1;
    const int ry = (wg & 1) * 2 + (vl >> 3);
    const int rx = vl & 7;
    const float* vb = vt + (ry * HX + rx) * 16 + 4 * h;

    // staging role inside the group: thread tg < 240 owns halo column (tg >> 2) and channel quad (tg & 3)
    const int tg = tid & 255;
    const bool s_on = tg < PP_COLS * 4;
    const int s_col = tg >> 2, s_q = tg & 3;
    const int s_cy = s_col / HX, s_cx = s_col % HX;
    f32x4 raw[6];
    auto fetch = [&](int u, int c) {
        const i32x4 e = utab[u - u_begin];
        const int gy = e.z * 8 + G * 4 - 1 + s_cy, gx = e.w * 8 - 1 + s_cx, gz0 = e.y * 4 - 1;
        const bool okc = s_on && (unsigned)gy < (unsigned)dim && (unsigned)gx < (unsigned)dim;
        const long long base = ((((long long)e.x * dim) * dim + gy) * dim + gx) * a.cin_pad + c * 16 + s_q * 4;
        const long long zs = (long long)dim * dim * a.cin_pad;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const bool ok = okc && (unsigned)(gz0 + q) < (unsigned)dim;
            const f32x4 t = *reinterpret_cast<const f32x4*>(a.in + (ok ? base + (gz0 + q) * zs : 0));
            raw[q] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&]() {
        if (!s_on) return;
        const f32x4 d0 = raw[0], d1 = raw[1], d2 = raw[2], d3 = raw[3], d4 = raw[4], d5 = raw[5];
        f32x4 v[6];
        v[0] = 4.f * d0 - 5.f * d2 + d4;
        v[5] = 4.f * d1 - 5.f * d3 + d5;
        const f32x4 e1 = d4 - 4.f * d2, o1 = d3 - 4.f * d1;
        v[1] = e1 + o1;
        v[2] = e1 - o1;
        const f32x4 e2 = d4 - d2, o2 = 2.f * (d3 - d1);
        v[3] = e2 + o2;
        v[4] = e2 - o2;
#pragma unroll
        for (int x = 0; x < 6; ++x) *reinterpret_cast<f32x4*>(vt + (x * PP_COLS + s_col) * 16 + s_q * 4) = v[x];
    };
    auto load_weights = [&](int cb, int c) {
        fill_lds<512>(wl, reinterpret_cast<const f32x4*>(a.wpack_e) + ((size_t)c * n_cb + cb) * (W43_FLOATS / 4), W43_FLOATS / 4, tid);
    };
    auto out_offset = [&](int u, int cb) -> long long {
        const i32x4 e = utab[u - u_begin];
        return ((((long long)e.x * dim + e.y * 4) * dim + e.z * 8 + G * 4 + ry) * dim + e.w * 8 + rx) * a.cout + cb * 32 + nt * 16 + 4 * h;
    };
    const long long zstride = (long long)dim * dim * a.cout;
    const bool relu = a.flags & SE_EPI_RELU;
    const bool use_res = (a.flags & SE_EPI_RES_PRE_RELU) && a.res;

    auto first_item = [&]() {
        WinoIter it;
        it.u_lo = u_begin;
        it.cb = u_begin / total_tiles;
        it.n = min(u_end, (it.cb + 1) * total_tiles) - u_begin;
        it.c = 0; it.k = 0; it.valid = true;
        return it;
    };
    auto next_item = [&](WinoIter it) {
        if (++it.k == it.n) {
            it.k = 0;
            if (++it.c == chunks) {
                it.c = 0;
                it.u_lo += it.n;
                if (it.u_lo >= u_end) { it.valid = false; return it; }
                it.cb = it.u_lo / total_tiles;
                it.n = min(u_end, (it.cb + 1) * total_tiles) - it.u_lo;
            }
        }
        return it;
    };

    f32x4 acc[6], part[4], resv[4], bias = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int z = 0; z < 4; ++z) part[z] = resv[z] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 6; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long long o0 = 0;
    const f32x4* wrow = reinterpret_cast<const f32x4*>(wl) + nt * 64 + lane;

    // epilogue operands of an item (running partial sums of earlier chunks, skip tensor, bias): issued early in its own MFMA
    // block, consumed one phase later (the previous chunk's stores of this half tile are at least two barriers old)
    auto load_operands = [&](const WinoIter& it) {
        o0 = out_offset(it.u_lo + it.k, it.cb);
        if (it.c > 0) {
#pragma unroll
            for (int z = 0; z < 4; ++z) part[z] = *reinterpret_cast<const f32x4*>(a.out + o0 + z * zstride);
        }
        if (it.c == chunks - 1) {
            bias = *reinterpret_cast<const f32x4*>(a.bpack + it.cb * 32 + nt * 16 + 4 * h);
            if (use_res) {
#pragma unroll
                for (int z = 0; z < 4; ++z) resv[z] = *reinterpret_cast<const f32x4*>(a.res + o0 + z * zstride);
            }
        }
    };

    __syncthreads();   // utab
    {
        const WinoIter it0 = first_item();
        fetch(it0.u_lo, 0);
        commit();
        load_weights(it0.cb, 0);
    }
    __syncthreads();

    int u_lo = u_begin;
    while (u_lo < u_end) {
        const int cb = u_lo / total_tiles;
        const int n = min(u_end, (cb + 1) * total_tiles) - u_lo;
        for (int c = 0; c < chunks; ++c) {
            const bool last_chunk = c == chunks - 1;
            const bool seg_next = c + 1 < chunks || u_lo + n < u_end;
            const int nc = c + 1 < chunks ? c + 1 : 0;
            // phases: group G issues the MFMAs of item j in phase 2j + G and finishes it (and stages what follows) in 2j + G + 1
            for (int t = 0; t <= 2 * n; ++t) {
                const int r = t - G;
                int kind = 0;
                PP_T(t_a)
                if (r >= 0 && !(r & 1) && (r >> 1) < n) {
                    kind = 1;
                    // ------------------------------ MFMA phase of item j: LDS reads and MFMAs only ------------------------------
                    __builtin_amdgcn_s_setprio(3);      // the partner wave on this SIMD is staging: its VALU must not delay MFMA issue
                    WinoIter mcur;
                    mcur.u_lo = u_lo; mcur.n = n; mcur.cb = cb; mcur.c = c; mcur.k = r >> 1; mcur.valid = true;
                    const WinoIter mnxt = next_item(mcur);
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    f32x4 w0 = wrow[0], v0 = *reinterpret_cast<const f32x4*>(vb);
                    f32x4 w1 = wrow[128], v1 = *reinterpret_cast<const f32x4*>(vb + PP_COLS * 16);
                    f32x4 w2 = w0, v2 = v0, w3 = w1, v3 = v1;
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    auto pairstep = [&](auto p_tag) {
                        constexpr int P = decltype(p_tag)::value;
                        constexpr int S0 = 2 * P, S1 = 2 * P + 1;
                        constexpr int x0 = S0 % 6, x1 = S1 % 6;
                        // global loads ride inside the MFMA block (the MFMA wave has issue priority; in the staging phase, at
                        // low priority and behind the output stores, they measured 3 % slower overall)
                        if constexpr (P == 1) { if (mnxt.valid) fetch(mnxt.u_lo + mnxt.k, mnxt.c); }   // next half tile's columns
                        if constexpr (P == 4) load_operands(mcur);                                     // this item's epilogue operands
                        if constexpr (P + 1 < 27) {
                            constexpr int T2 = (S0 + 2) / 6, X2 = (S0 + 2) % 6, T3 = (S1 + 2) / 6, X3 = (S1 + 2) % 6;
                            w2 = wrow[(S0 + 2) * 128];
                            v2 = *reinterpret_cast<const f32x4*>(vb + (X2 * PP_COLS + (T2 / 3) * HX + (T2 % 3)) * 16);
                            w3 = wrow[(S1 + 2) * 128];
                            v3 = *reinterpret_cast<const f32x4*>(vb + (X3 * PP_COLS + (T3 / 3) * HX + (T3 % 3)) * 16);
                        }
                        acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, v0.x, acc[x0], 0, 0, 0);
                        acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, v1.x, acc[x1], 0, 0, 0);
                        acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, v0.y, acc[x0], 0, 0, 0);
                        acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, v1.y, acc[x1], 0, 0, 0);
                        acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, v0.z, acc[x0], 0, 0, 0);
                        acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, v1.z, acc[x1], 0, 0, 0);
                        acc[x0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, v0.w, acc[x0], 0, 0, 0);
                        acc[x1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, v1.w, acc[x1], 0, 0, 0);
                        if constexpr (P + 1 < 27) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        } else {
                            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                        }
                        w0 = w2; v0 = v2; w1 = w3; v1 = v3;
                    };
                    for_each_index(pairstep, std::make_integer_sequence<int, 27>{});
                    __builtin_amdgcn_s_setprio(0);
                } else if (r >= 1 && (r & 1) && ((r - 1) >> 1) < n) {
                    kind = 2;
                    // ------------------------------ finish item j, stage what follows ------------------------------
                    WinoIter cur;
                    cur.u_lo = u_lo; cur.n = n; cur.cb = cb; cur.c = c; cur.k = (r - 1) >> 1; cur.valid = true;
                    const WinoIter nxt = next_item(cur);
                    f32x4 y[4];
                    {
                        const f32x4 s12 = acc[1] + acc[2], d12 = acc[1] - acc[2], s34 = acc[3] + acc[4], d34 = acc[3] - acc[4];
                        y[0] = acc[0] + s12 + s34;
                        y[1] = d12 + 2.f * d34;
                        y[2] = s12 + 4.f * s34;
                        y[3] = d12 + 8.f * d34 + acc[5];
                    }
                    if (c > 0) {
#pragma unroll
                        for (int z = 0; z < 4; ++z) y[z] += part[z];
                    }
                    if (nxt.valid) commit();         // this group's V half tile is free: its MFMA block ended before the barrier
                    if (last_chunk) {
#pragma unroll
                        for (int z = 0; z < 4; ++z) {
                            y[z] += bias;
                            if (use_res) y[z] += resv[z];
                            if (relu) {
                                y[z].x = fmaxf(y[z].x, 0.f); y[z].y = fmaxf(y[z].y, 0.f);
                                y[z].z = fmaxf(y[z].z, 0.f); y[z].w = fmaxf(y[z].w, 0.f);
                            }
                        }
                    }
#pragma unroll
                    for (int z = 0; z < 4; ++z) *reinterpret_cast<f32x4*>(a.out + o0 + z * zstride) = y[z];
                }
                PP_T(t_b)
                __syncthreads();
                PP_T(t_c)
#ifdef SE_STAMPPP
                if (kind == 1) acc_mfma += t_b - t_a; else if (kind == 2) acc_stage += t_b - t_a; else acc_idle += t_b - t_a;
                acc_bar += t_c - t_b;
                ++n_ph;
#endif
                (void)kind;
            }
            if (seg_next) {
                load_weights(c + 1 < chunks ? cb : (u_lo + n) / total_tiles, nc);
                __syncthreads();
            }
        }
        u_lo += n;
    }
#ifdef SE_STAMPPP
    if (lane == 0 && dbg) {
        unsigned long long* o = dbg + ((size_t)blockIdx.x * 8 + wave) * 6;
        o[0] = acc_mfma; o[1] = acc_stage; o[2] = acc_bar; o[3] = acc_idle; o[4] = n_ph; o[5] = 0;
    }
#endif
}

[[maybe_unused]] unsigned long long* g_wino_dbg = nullptr;
unsigned long long* g_wino_dbg43 = nullptr;

}  // namespace

// Returns 0 on launch, SE_TILED_NOT_TAKEN if the shape/flags are not covered, else a hipError_t.
// Production: the ping-pong F(4,3) kernel (shapes the 2-D kernel of conv3d_wino2d.hip does not take, e.g. dim % 16 != 0).
// Development builds add the retired forms: se_debug_set_variant(4) F(2,3), (19) single-phase F(4,3), stamp builds.
int se_conv3d_wino_try(const ConvArgs& a, int batch, hipStream_t s) {
    const int dim = a.dim;
    if (!a.wpack_b || !a.wpack_e || dim < 16 || (dim & 7) || (a.cout & 31) || (a.cin & 15) || a.cin_pad != a.cin) return SE_TILED_NOT_TAKEN;
    if (a.flags & (SE_EPI_RES_POST_RELU | SE_EPI_OUT_PLANAR | SE_LAYOUT_OCTET_BITS | SE_LAYOUT_QUAD_BITS)) return SE_TILED_NOT_TAKEN;
    constexpr int LDS43 = 160 * 1024;
    constexpr int MAXPP = (LDS43 - (W43_FLOATS + 2 * PP_VH_FLOATS) * 4) / 16;
    const int num_cus = se_num_cus();
    const int tiles = dim / 8, ztiles = dim / 4;
    const int total_tiles = batch * ztiles * tiles * tiles;
    const int n_cb = a.cout / 32;
    const int n_units = n_cb * total_tiles;
    const int grid = n_units < num_cus ? n_units : num_cus;
    const int per = (n_units + grid - 1) / grid;
#ifdef SE_DEVTOOLS
    {
        constexpr int LDS_FIXED = (SE_WINO_CHUNK_FLOATS + 2 * TILE_FLOATS) * 4;
        constexpr int MAX_UNITS_PER_WG = (LDS43 - LDS_FIXED) / 16;
        constexpr int MAX43 = (LDS43 - (W43_FLOATS + V43_FLOATS) * 4) / 16;
        if ((g_variant == 4 || g_wino_dbg) && per <= MAX_UNITS_PER_WG) {
            SE_ENSURE_LDS(conv3d_k3_wino_kernel<false>, LDS43);
            SE_ENSURE_LDS(conv3d_k3_wino_kernel<true>, LDS43);
            if (g_wino_dbg)
                hipLaunchKernelGGL(conv3d_k3_wino_kernel<true>, dim3((n_units + per - 1) / per), dim3(512), LDS43, s, a, tiles, ztiles,
                                   total_tiles, n_cb, per, 0, g_wino_dbg);
            else
                hipLaunchKernelGGL(conv3d_k3_wino_kernel<false>, dim3((n_units + per - 1) / per), dim3(512), LDS43, s, a, tiles, ztiles,
                                   total_tiles, n_cb, per, 0, nullptr);
            SE_CHECK_LAUNCH();
            return 0;
        }
        if (g_variant == 19 && per <= MAX43) {
            SE_ENSURE_LDS(conv3d_k3_wino43_kernel, LDS43);
            hipLaunchKernelGGL(conv3d_k3_wino43_kernel, dim3((n_units + per - 1) / per), dim3(512), LDS43, s, a, tiles, ztiles,
                               total_tiles, n_cb, per, 0, g_wino_dbg43);
            SE_CHECK_LAUNCH();
            return 0;
        }
    }
#endif
    if (per > MAXPP) return SE_TILED_NOT_TAKEN;      // cannot happen behind se_conv3d_tiled_try's unit-budget batch slices
    SE_ENSURE_LDS(conv3d_k3_wino43pp_kernel, LDS43);
    hipLaunchKernelGGL(conv3d_k3_wino43pp_kernel, dim3((n_units + per - 1) / per), dim3(512), LDS43, s, a, tiles, ztiles, total_tiles,
                       n_cb, per, 0, g_wino_dbg43);
    SE_CHECK_LAUNCH();
    return 0;
}

#ifdef SE_DEVTOOLS
// Debug only: device buffer (grid * 8 waves * 4 u64) that makes the Winograd kernel run its STAMP build.
extern "C" void se_debug_set_stamp_buffer(void* p) {
#if defined(SE_STAMP43) || defined(SE_STAMPPP) || defined(SE_STAMP47) || defined(SE_STAMP67)
    g_wino_dbg43 = reinterpret_cast<unsigned long long*>(p);
#else
    g_wino_dbg = reinterpret_cast<unsigned long long*>(p);
#endif
}
#endif

// Returns 0 on launch, SE_TILED_NOT_TAKEN if not covered, else a hipError_t.
// Production: the F(4,7) kernel of conv3d_wino47.hip.  Development builds add the retired F(2,7) kernels
// (se_debug_set_variant(17) single-phase, (19) ping-pong).
int se_conv3d_k7_wino_try(const ConvArgs& a, int batch, hipStream_t s) {
    const int dim = a.dim;
    if (!a.wpack_d || !a.wpack_f || dim < 16 || (dim & 7) || a.cout != 16 || a.res || (a.flags & (SE_EPI_OUT_PLANAR))) return SE_TILED_NOT_TAKEN;
    const int num_cus = se_num_cus();
#ifdef SE_DEVTOOLS
    if (g_variant == 17 || g_variant == 19) {
        if (a.flags & SE_IN_PLANAR3) return SE_ERR_BAD_ARG;   // only the F(4,7) kernel reads the triplet-planar layout
        constexpr int LDS_BYTES = 160 * 1024;
        constexpr int MAX_UNITS = (LDS_BYTES - (K7_W_FLOATS + K7_VT_FLOATS) * 4) / 16;
        const int tiles = dim / 8, ztiles = dim / 4;
        const int total_tiles = batch * ztiles * tiles * tiles;
        const int grid = total_tiles < num_cus ? total_tiles : num_cus;
        const int per = (total_tiles + grid - 1) / grid;
        if (per > MAX_UNITS) return SE_TILED_NOT_TAKEN;
        SE_ENSURE_LDS(conv3d_k7_wino_kernel, LDS_BYTES);
        SE_ENSURE_LDS(conv3d_k7_winopp_kernel, LDS_BYTES);
        if (g_variant == 17)
            hipLaunchKernelGGL(conv3d_k7_wino_kernel, dim3((total_tiles + per - 1) / per), dim3(512), LDS_BYTES, s, a, tiles, ztiles,
                               total_tiles, per);
        else
            hipLaunchKernelGGL(conv3d_k7_winopp_kernel, dim3((total_tiles + per - 1) / per), dim3(512), LDS_BYTES, s, a, tiles, ztiles,
                               total_tiles, per);
        SE_CHECK_LAUNCH();
        return 0;
    }
#endif
    // dim % 16 == 0: the tile-outer F(6,7) kernel (conv3d_wino67.hip); development builds keep the F(4,7) kernel selectable (variant 47)
    if (a.wpack_h && (dim & 15) == 0
#ifdef SE_DEVTOOLS
        && g_variant != 47
#endif
    ) {
        const int rc67 = se_conv3d_k7_wino67_launch(a, batch, num_cus, s, g_wino_dbg43);
        if (rc67 != SE_TILED_NOT_TAKEN) return rc67;
    }
    const int rc = (a.flags & SE_IN_PLANAR3) ? se_conv3d_k7_wino47_launch_p3(a, batch, num_cus, s, g_wino_dbg43)
                                             : se_conv3d_k7_wino47_launch_cl(a, batch, num_cus, s, g_wino_dbg43);
    if (rc == SE_TILED_NOT_TAKEN && (a.flags & SE_IN_PLANAR3)) return SE_ERR_BAD_ARG;   // cannot happen behind the unit-budget batch slices
    return rc;
}
