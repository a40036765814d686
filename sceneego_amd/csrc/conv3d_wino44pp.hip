// 3x3x3 convolution, float32, 2-D Winograd F(4,3) along z AND y (direct along x), on v_mfma_f32_16x16x4_f32 - ping-pong form.
// Stands in for Conv3d(k=3) + BatchNorm3d (+ReLU) (+skip add) of Res3DBlock (reference network/v2v.py:21-43) at the 64^3 / 32^3
// levels (round 4; the F(4,3) x F(2,3) kernel of conv3d_wino2d.hip keeps the 16^3 level and is the A/B reference).
//
// Why: the 3^3 layers are bound by the float32 matrix pipe, which on gfx950 shares the vector ALUs with the transforms (DESIGN.md
// section 4, round 3), so the lever is the NUMBER of products.  Per 4(z) x 4(y) outputs and x tap the transform domain has
// 6 x 6 = 36 points instead of 4*4*3*3 = 144 products: 1/4 of the direct MFMAs (F(4,3) x F(2,3): 1/3).  The round-3 experiment
// (conv3d_wino44.hip, lockstep waves) showed the MFMA phases at 0.63 x the production step but its transforms and its tile end
// (304 KB of loads and stores per CU with no MFMA phase beside them) exposed; this kernel puts the same arithmetic into the
// ping-pong structure of the production kernel, where the other wave group's MFMA phase covers exactly that.
//
// Work unit = (32-cout block, tile of 8(z) x 8(y) x 16(x) outputs); a persistent 512-thread workgroup per CU walks a contiguous
// range of units.  The 8 waves form two groups of 4 (one wave per SIMD each).  Group G owns y-tile G (rows 4G..4G+3) of the tile;
// inside a group wave (ct, zt) computes cout tile ct of z-tile zt: 16 x positions on the MFMA columns, 36 (xi_y, xi_z)
// accumulators of 16 couts x 16 positions = 144 registers, over ALL input channels (the output is written once).
// Channels are walked in chunks of 4 = the four k lanes of one MFMA.  Per chunk and group a "step" is
//     MFMA phase     108 MFMAs per wave: 9 xi quads x 3 dx x 4, operands by ds_read_b128 from
//                      W [q 9][dx 3][ct 2][lane][4 xi]                  55.3 KB per chunk, G-transformed weights; xi = 6 xi_y + xi_z
//                      V [G 2][zt 2][18 x records][channel 4][36 xi]     42.6 KB, B^T-transformed input (record stride 148 floats)
//     staging phase  (the other group meanwhile) half 1: B^T along y of the next step's 10(z) x 6(y) x 18(x) halo rows (loaded as
//                    riders of the group's own MFMA phase) into the scratch tile T; half 2: B^T along z from T into the group's V
//                    tiles, and after the last chunk the output transform + epilogue of the finished tile.
// The two groups run half a step apart, so every SIMD always has one wave in its MFMA block; four workgroup barriers per step
// (start / middle of either group's MFMA phase = middle / end of the other's staging phase).
// Weight stream: LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write) of HALF chunks (quads 0..4 / 5..8) into THREE
// rotating slots.  Half n = 2 step + (0 | 1) lives in slot n mod 3: with three slots the slot of half n is free two (odd n) or four
// (even n) quarter steps before its first reader, so a DMA has >= half a step (~2 us) to land - with two slots (one chunk buffer,
// the production kernel's scheme) the window is a quarter step, shorter than an LDS-DMA's landing time (round-2 finding).
//   half A of step i+1  is issued by group 0's MFMA waves at the start of their MFMA phase of step i (slot (2i+2) mod 3: last read
//                       by group 1 in the quarter before), waited by them in front of their staging phase's mid barrier;
//   half B of step i+1  is issued by group 1's MFMA waves behind the mid barrier of their MFMA phase of step i (slot 2i mod 3 = half A
//                       of step i: group 1 has just finished with it), waited in front of their staging phase's mid barrier, which
//                       is the barrier in front of group 0's first read of it.
// LDS: 3 x 30,720 (slots) + 42,624 (V) + 17,280 (T) = 152,064 B.
#include "conv_common.h"

#include <type_traits>
#include <utility>

namespace {

template <typename F, int... S>
__device__ __forceinline__ void for_each_c(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}

#ifndef SE_K44P_QA
#define SE_K44P_QA 4
#endif
constexpr int P_QA = SE_K44P_QA;                         // half A = xi quads 0..P_QA-1, half B = the rest (4 | 5: the slot holds 5 quads)
constexpr int P_HALF_A = P_QA * 3 * 2 * 256;             // 7680 floats
constexpr int P_HALF_B = (9 - P_QA) * 3 * 2 * 256;       // 6144 floats
static_assert(P_HALF_A + P_HALF_B == SE_WINO44_CHUNK_FLOATS, "chunk = two halves");
constexpr int P_SLOT = P_HALF_A > P_HALF_B ? P_HALF_A : P_HALF_B;      // floats per weight slot
constexpr int P_NA = P_HALF_A / 256;                     // wave-instructions (64 lanes x 16 B) of half A
constexpr int P_NB = P_HALF_B / 256;
#ifndef SE_K44P_RS
#define SE_K44P_RS 148
#endif
constexpr int P_RS = SE_K44P_RS;                         // floats per x record of V: 4 channels x 36 xi + 4 pad (37 x 16 B: odd -> bank spread)
constexpr int P_VT = 18 * P_RS;                          // one (G, zt) tile of V
constexpr int P_V_FLOATS = 4 * P_VT;                     // 10,656 floats
constexpr int P_T_FLOATS = 6 * 10 * 18 * 4;              // scratch of pass 1: [xi_y 6][z 10][x 18][4 channels] = 4320 floats
constexpr int P_LDS_BYTES = (3 * P_SLOT + P_V_FLOATS + P_T_FLOATS) * 4;   // 152,064 B
constexpr int P_GROUPS = 27;                             // (quad, dx) groups of 4 MFMAs
constexpr int P_GA = P_QA * 3;                           // groups that read half A: 15
constexpr int P_DMA_A = (P_NA + 3) / 4;                  // LDS-DMA instructions per wave of a group for half A: 8
constexpr int P_DMA_B = (P_NB + 3) / 4;                  // 6

#ifndef SE_K44P_EXP      // attribution builds only (results wrong): 1 no weight DMAs, 2 no input-row loads, 4 no transform passes,
#define SE_K44P_EXP 0    // 16 no skip-tensor loads, 32 no output stores
#endif

#ifndef SE_K44P_EPRIO
#define SE_K44P_EPRIO 0      // s_setprio of a staging wave while it runs a tile end (its SIMD partner, the other group's MFMA wave, runs at 3)
#endif
#ifndef SE_K44P_PK
#define SE_K44P_PK 1         // 1: the output transform of the epilogue on whole f32x4 vectors (v_pk_add_f32 / v_pk_fma_f32)
#endif

#ifndef SE_K44P_PKT
#define SE_K44P_PKT 0        // 1: the per-step B^T transforms on float pairs as well
#endif

#ifdef SE_STAMP44P   // cycle stamps (development builds with -DSE_STAMP44P; tools/stamp_k44p.py)
#define TP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
                st_sum[i] += (unsigned)(t_ - st_last); st_last = t_; __builtin_amdgcn_sched_barrier(0); }
unsigned long long* g_w44p_dbg = nullptr;
#else
#define TP(i)
#endif

struct UnitP {
    int cb, b, z0, y0, x0;
    int i;          // linear unit index (x fastest, then y, z, sample, cout block)
};

// F(4,3) B^T (points 0, +-1, +-2, inf) on six values, per component (scalar float arithmetic: packed VALU beside the partner
// wave's MFMA stream is an anti-lever, see conv3d_wino2d.hip)
__device__ __forceinline__ void bt43p(const f32x4 (&d)[6], f32x4 (&o)[6]) {
    if (SE_K44P_PKT) {
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f2 v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = hf ? (f2){d[i].z, d[i].w} : (f2){d[i].x, d[i].y};
            const f2 c4 = {4.f, 4.f}, cm5 = {-5.f, -5.f}, cm4 = {-4.f, -4.f}, c2 = {2.f, 2.f}, cm2 = {-2.f, -2.f};
            f2 r[6];
            r[0] = __builtin_elementwise_fma(c4, v[0], __builtin_elementwise_fma(cm5, v[2], v[4]));
            r[5] = __builtin_elementwise_fma(c4, v[1], __builtin_elementwise_fma(cm5, v[3], v[5]));
            const f2 e1 = __builtin_elementwise_fma(cm4, v[2], v[4]), o1 = __builtin_elementwise_fma(cm4, v[1], v[3]);
            r[1] = e1 + o1;
            r[2] = e1 - o1;
            const f2 e2 = v[4] - v[2], o2 = v[3] - v[1];
            r[3] = __builtin_elementwise_fma(c2, o2, e2);
            r[4] = __builtin_elementwise_fma(cm2, o2, e2);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (hf) { o[i].z = r[i].x; o[i].w = r[i].y; }
                else { o[i].x = r[i].x; o[i].y = r[i].y; }
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float d0 = d[0][c], d1 = d[1][c], d2 = d[2][c], d3 = d[3][c], d4 = d[4][c], d5 = d[5][c];
        o[0][c] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        o[5][c] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
        const float e1 = fmaf(-4.f, d2, d4), o1 = fmaf(-4.f, d1, d3);
        o[1][c] = e1 + o1;
        o[2][c] = e1 - o1;
        const float e2 = d4 - d2, o2 = d3 - d1;
        o[3][c] = fmaf(2.f, o2, e2);
        o[4][c] = fmaf(-2.f, o2, e2);
    }
}
// F(4,3) A^T on six values -> four outputs
__device__ __forceinline__ void at43p(const f32x4& m0, const f32x4& m1, const f32x4& m2, const f32x4& m3, const f32x4& m4, const f32x4& m5,
                                      f32x4 (&y)[4]) {
    if (SE_K44P_PK) {
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f2 a0 = hf ? (f2){m0.z, m0.w} : (f2){m0.x, m0.y}, a1 = hf ? (f2){m1.z, m1.w} : (f2){m1.x, m1.y},
                     a2 = hf ? (f2){m2.z, m2.w} : (f2){m2.x, m2.y}, a3 = hf ? (f2){m3.z, m3.w} : (f2){m3.x, m3.y},
                     a4 = hf ? (f2){m4.z, m4.w} : (f2){m4.x, m4.y}, a5 = hf ? (f2){m5.z, m5.w} : (f2){m5.x, m5.y};
            const f2 s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
            const f2 y0 = (a0 + s12) + s34;
            const f2 y1 = __builtin_elementwise_fma((f2){2.f, 2.f}, d34, d12);
            const f2 y2 = __builtin_elementwise_fma((f2){4.f, 4.f}, s34, s12);
            const f2 y3 = __builtin_elementwise_fma((f2){8.f, 8.f}, d34, d12) + a5;
            if (hf) { y[0].z = y0.x; y[0].w = y0.y; y[1].z = y1.x; y[1].w = y1.y; y[2].z = y2.x; y[2].w = y2.y; y[3].z = y3.x; y[3].w = y3.y; }
            else { y[0].x = y0.x; y[0].y = y0.y; y[1].x = y1.x; y[1].y = y1.y; y[2].x = y2.x; y[2].y = y2.y; y[3].x = y3.x; y[3].y = y3.y; }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float s12 = m1[c] + m2[c], d12 = m1[c] - m2[c], s34 = m3[c] + m4[c], d34 = m3[c] - m4[c];
        y[0][c] = (m0[c] + s12) + s34;
        y[1][c] = fmaf(2.f, d34, d12);
        y[2][c] = fmaf(4.f, s34, s12);
        y[3][c] = fmaf(8.f, d34, d12) + m5[c];
    }
}

// LAYOUT: bit 0 = input QUAD-planar [B][cin/4][D][D][D][4] (SE_IN_QUAD), bit 1 = output quad-planar (SE_OUT_QUAD), bit 2 = skip
// tensor quad-planar (SE_RES_QUAD), bit 3 = also write the 2x2x2 max-pool of the output (se_conv3d_pool_f32), bit 4 = the skip path
// is a 1x1x1 convolution over a 16-channel tensor computed in the epilogue (se_conv3d_skip16_f32): channels-last, or with bit 2 quad-planar.
// Quad-planar (round 5; rounds 2-4 handed octet-planar tensors [B][C/8][D^3][8] between these launches): a 4-channel step reads WHOLE
// 16-byte records, 18 of them contiguous per halo row (288 B), instead of half of every 32-byte octet record - that half was the
// 1.57 x of the counter traffic over the algorithmic bytes; an MFMA D fragment (4 couts of a voxel per lane) is
// exactly one record, 16 lanes write 256 contiguous bytes.  The octet-planar flags are served by the F(4,3) x F(2,3) kernel only.
template <int LAYOUT>
__global__ __launch_bounds__(512) void conv3d_k3_wino44pp_kernel(ConvArgs a, const float* __restrict__ wg, int tiles_x, int tiles_y, int tiles_z,
                                                                 int total_tiles, int n_units, int units_per_wg, unsigned long long* dbg) {
    (void)dbg;
#ifdef SE_STAMP44P
    unsigned st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = 0;
#endif
    constexpr bool in_pl = LAYOUT & 1, out_pl = LAYOUT & 2, res_pl = LAYOUT & 4, pool = LAYOUT & 8, skc = LAYOUT & 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* vt = lds + 3 * P_SLOT;
    float* tt = vt + P_V_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifndef SE_K44P_SWAP
#define SE_K44P_SWAP 0     // experiment: 1 = waves 4..7 are group 0 (does the asymmetry between the groups follow the role or the wave slot?)
#endif
    const int G = SE_K44P_SWAP ? 1 - (wave >> 2) : wave >> 2;      // group: y-tile of the tile
    const int wq = wave & 3;
    const int ct = wq >> 1;                     // cout tile of the 32-cout block
    const int zt = wq & 1;                      // z-tile
    const int px = lane & 15, h = lane >> 4;
    const int tg = tid & 255;                   // thread inside the group
    const int dim = a.dim, cin = a.cin;
    const int chunks = cin >> 2;
    // This workgroup's units: u_first, u_first + u_stride, ... (n_mine of them).  SE_XCD_WALK (conv_common.h) 1: a contiguous range of the
    // XCD's share; 2 (round 5): the workgroups of an XCD INTERLEAVE - workgroup w of the XCD's S takes units w, w + S, w + 2 S ... of the
    // XCD's range - so that the S tiles in flight on an XCD at any time are S consecutive units (at 64^3: a whole z slab of a sample)
    // and the halo rows they share are hits in the XCD's L2 while both readers are at work.
    int u_first, u_stride, n_mine;
#ifndef SE_K44P_XCD_WALK
#define SE_K44P_XCD_WALK 2       // measured (profiles/r05_wino44pp_experiments.txt section 5): reads of a launch 602 -> 358 MB, time -0.7 % inside the forward
#endif
    if (SE_K44P_XCD_WALK == 2 && (gridDim.x & 7) == 0) {
        const int S = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, w = (int)blockIdx.x >> 3;
        const int r0 = xcd * S * units_per_wg, r1 = min(r0 + S * units_per_wg, n_units);
        u_first = r0 + w; u_stride = S;
        n_mine = u_first < r1 ? (r1 - u_first + S - 1) / S : 0;
    } else {
        u_first = se_xcd_walk_index((int)blockIdx.x, (int)gridDim.x) * units_per_wg;
        u_stride = 1;
        n_mine = min(u_first + units_per_wg, n_units) - u_first;
    }
    if (n_mine <= 0) return;
    const int n_steps = n_mine * chunks;

    auto decode = [&](int u) {
        UnitP r;
        r.cb = u / total_tiles;
        int t = u - r.cb * total_tiles;
        const int xt = t % tiles_x; t /= tiles_x;
        const int yy = t % tiles_y; t /= tiles_y;
        const int zz = t % tiles_z;
        r.b = t / tiles_z;
        r.z0 = zz * 8; r.y0 = yy * 8; r.x0 = xt * 16;
        r.i = u;
        return r;
    };
    // the unit after u in the walk (x fastest, then y, z, sample, cout block): carries instead of divisions
    // interleaved walk: the stride in mixed radix (tiles_x, tiles_y, tiles_z, samples) - a jump is four adds with carries, no division
    const int batch_n = total_tiles / (tiles_x * tiles_y * tiles_z);
    int sj_x = 0, sj_y = 0, sj_z = 0, sj_b = 0, sj_cb = 0;
    if (u_stride != 1) {
        int t = u_stride;
        sj_x = t % tiles_x; t /= tiles_x;
        sj_y = t % tiles_y; t /= tiles_y;
        sj_z = t % tiles_z; t /= tiles_z;
        sj_b = t % batch_n; sj_cb = t / batch_n;
    }
    auto advance = [&](UnitP u) {
        if (u_stride != 1) {
            u.i += u_stride;
            int x = (u.x0 >> 4) + sj_x, y = (u.y0 >> 3) + sj_y, z = (u.z0 >> 3) + sj_z, b = u.b + sj_b, cb = u.cb + sj_cb;
            if (x >= tiles_x) { x -= tiles_x; y += 1; }
            if (y >= tiles_y) { y -= tiles_y; z += 1; }
            if (z >= tiles_z) { z -= tiles_z; b += 1; }
            if (b >= batch_n) { b -= batch_n; cb += 1; }
            u.x0 = x << 4; u.y0 = y << 3; u.z0 = z << 3; u.b = b; u.cb = cb;
            return u;
        }
        u.i += 1;
        u.x0 += 16;
        if (u.x0 == dim) {
            u.x0 = 0; u.y0 += 8;
            if (u.y0 == dim) {
                u.y0 = 0; u.z0 += 8;
                if (u.z0 == dim) {
                    u.z0 = 0; u.b += 1;
                    if (u.b * tiles_z * tiles_y * tiles_x == total_tiles) { u.b = 0; u.cb += 1; }
                }
            }
        }
        return u;
    };
    // (unit, chunk) one step after (u, c); behind the last step of this workgroup the walk stays where it is, so neither the staging
    // code nor the riders need "is there a next step" branches (they then reload data nobody reads)
    auto step_after = [&](UnitP& u, int& c, int& idx) {
        if (idx + 1 >= n_steps) return;
        ++idx;
        if (++c == chunks) { c = 0; u = advance(u); }
    };

    // ---- MFMA operand addresses ----
    const float* a_lane = wl + ct * 256 + lane * 4;                                   // + slot * P_SLOT + (q_local * 3 + dx) * 512
    const float* b_base = vt + (G * 2 + zt) * P_VT + px * P_RS + h * 36;              // + dx * P_RS + q * 4

    // ---- staging roles inside the group (256 threads) ----
    // pass 1: task (z1, x1) = thread tg < 180: the six y rows of the group's halo at (z1, x1), 4 channels -> six xi_y into T
    // pass 2: task (zt2, xi_y, x2) = thread tg < 216: six z slabs of T -> six xi_z into the group's V tile zt2
    // The task coordinates are recomputed from the thread index where they are used (a few VALU outside the MFMA phase): kept in
    // registers across the loop they are spilled, and a scratch reload waits for every older vector-memory operation.
    const bool p1_on = tg < 180, p2_on = tg < 216;
#ifndef SE_K44P_KEEP
#define SE_K44P_KEEP 1       // 1: the pass-1 / pass-2 LDS offsets live in three registers; 0: recomputed from the thread index per step
#endif
    auto opaque_tg = [&]() { int t = tg; if (!SE_K44P_KEEP) asm volatile("" : "+v"(t)); return t; };
    const int k_p1 = ((min(tg, 179) / 18) * 18 + min(tg, 179) % 18) * 4;                                    // pass 1: T offset of (z1, x1)
    const int k_p2s = ((((min(tg, 215) / 18) % 6) * 10 + 4 * (min(tg, 215) / 108)) * 18 + min(tg, 215) % 18) * 4;   // pass 2: T offset
    const int k_p2d = (G * 2 + min(tg, 215) / 108) * P_VT + (min(tg, 215) % 18) * P_RS + ((min(tg, 215) / 18) % 6) * 6;   // pass 2: V offset

    // ---- input rows of the next step: raw buffer loads.  Everything uniform - sample, chunk, the ROW (the six rows of a group
    // are the same for all of its lanes) - sits in the descriptor base, built with scalar instructions; a row outside the volume is
    // read through a zero-record descriptor; the lane part (z slab, x) is one 32-bit offset whose bit 31 marks a voxel outside the
    // volume (reads zero).  No vector instruction but the load itself rides in the MFMA stream.
    constexpr unsigned OOB = 0x80000000u;
    const unsigned in_bytes = (unsigned)dim * dim * dim * cin * 4u;
    const int vfl = in_pl ? 4 : cin;                                                  // floats between x neighbours
    f32x4 raw[6];
    unsigned f_voff = OOB;
    auto fetch_setup = [&](const UnitP& u) {
        const int t = min(opaque_tg(), 179);
        const int x1 = t % 18, z1 = t / 18;
        const int gz = u.z0 - 1 + z1, gx = u.x0 - 1 + x1;
        const bool ok = p1_on & ((unsigned)gz < (unsigned)dim) & ((unsigned)gx < (unsigned)dim);
        f_voff = ok ? (unsigned)((gz * dim * dim + gx) * vfl * 4) : OOB;
    };
    auto row_base = [&](const UnitP& u, int c4) {        // element (z 0, y 0, x 0) of the step's sample / channel quad
        return a.in + (long long)u.b * dim * dim * dim * cin + (in_pl ? (long long)c4 * dim * dim * dim * 4 : c4 * 4);
    };
    auto fetch_one = [&](const UnitP& u, int c4, auto r_tag) {
        constexpr int r = decltype(r_tag)::value;
        const int gy = u.y0 + 4 * G - 1 + r;                       // uniform
        const bool ok = (unsigned)gy < (unsigned)dim;
        const float* p = row_base(u, c4) + (long long)gy * dim * vfl;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, ok ? (int)in_bytes : 0, 0x00020000);
        if (SE_K44P_EXP & 2) { raw[r] = (f32x4){(float)ok, 0.f, 0.f, 0.f}; return; }
        raw[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)f_voff, 0, 0));
    };
    auto pass1 = [&](float* tdst) {
        if (!p1_on || (SE_K44P_EXP & 4)) return;
        const int t = opaque_tg();
        const int x1 = t % 18, z1 = t / 18;
        float* dst = tdst + (SE_K44P_KEEP ? k_p1 : (z1 * 18 + x1) * 4);               // + xi_y * 720
        f32x4 o[6];
        bt43p(raw, o);
#pragma unroll
        for (int e = 0; e < 6; ++e) *reinterpret_cast<f32x4*>(dst + e * 720) = o[e];
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto pass2 = [&](const float* tsrc) {
        if (!p2_on || (SE_K44P_EXP & 4)) return;
        const int t = opaque_tg();
        const int x2 = t % 18, xy2 = (t / 18) % 6, zt2 = t / 108;
        const float* src = tsrc + (SE_K44P_KEEP ? k_p2s : ((xy2 * 10 + 4 * zt2) * 18 + x2) * 4);      // + s * 72 (z slab)
        float* vdst = vt + (SE_K44P_KEEP ? k_p2d : (G * 2 + zt2) * P_VT + x2 * P_RS + xy2 * 6);     // + channel * 36
        f32x4 d[6], o[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) d[s] = *reinterpret_cast<const f32x4*>(src + s * 72);
        bt43p(d, o);
#pragma unroll
        for (int c = 0; c < 4; ++c) {       // channel c: its six xi_z are 24 consecutive bytes (8-byte aligned)
            f32x2* dst = reinterpret_cast<f32x2*>(vdst + c * 36);
            dst[0] = (f32x2){o[0][c], o[1][c]};
            dst[1] = (f32x2){o[2][c], o[3][c]};
            dst[2] = (f32x2){o[4][c], o[5][c]};
        }
    };

    // ---- weight stream: LDS-DMA piece J of a half for this wave = wave-instruction 4 J + wq of the group's share (surplus ones
    // repeat the last) ----
    // Issued through inline assembly on purpose: hipcc models a global_load_lds as an LDS access of unknown address and then (a)
    // waits with lgkmcnt(0) - i.e. for the operand read it has just issued - at every group of MFMAs that follows one in the phase,
    // and (b) puts an s_waitcnt vmcnt(0) in front of the first operand read of the next phase, which also waits for the acknowledgement
    // of a finished tile's 16 output stores (disassembly, round 4).  The counted waits this kernel needs are written out below; the
    // compiler's own vmcnt waits stay correct (vmcnt retires in order, an operation it does not know of only makes them stricter).
    const int lane16 = lane * 16;
    auto wdma = [&](const float* src, float* region, auto n_tag, auto j_tag) {
        constexpr int NWI = decltype(n_tag)::value, J = decltype(j_tag)::value;
        int piece = J * 4 + wq;
        piece = piece < NWI ? piece : NWI - 1;
        const float* sp = src + piece * 256;        // uniform: the per-lane part of every LDS-DMA address is the same lane * 16 bytes
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            (unsigned)(__UINTPTR_TYPE__)((float __attribute__((address_space(3)))*)(region + piece * 256)));      // LDS byte address
        if ((SE_K44P_EXP & 1) || ((SE_K44P_EXP & 64) && G == 1) || ((SE_K44P_EXP & 128) && G == 0)) return;      // 64 / 128: no DMAs of group 1 / 0
        const int l16 = lane16;                    // (asm operands do not capture: name a local of the lambda)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(l16), "s"(sp) : "m0");
    };
    using NA = std::integral_constant<int, P_NA>;
    using NB = std::integral_constant<int, P_NB>;

    // Workgroup barrier that waits for this wave's LDS traffic only (lgkmcnt): global loads and LDS-DMAs stay in flight across it.
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    const float relu_lo = (a.flags & SE_EPI_RELU) ? 0.f : -__builtin_inff();
    const bool use_res = (a.flags & SE_EPI_RES_PRE_RELU) && a.res;

    f32x4 acc[36];

    // ---- epilogue of a finished tile (staging half 2 of its last step; the other group is in its MFMA phase) ----
    // addresses: uniform 64-bit base of the wave's first output row + a 32-bit per-lane offset + uniform (y, z) strides (global_*
    // saddr form: one VGPR of address for all 16 accesses); GLOBAL pointers (rebuilt from an integer as generic ones every access
    // became a flat_load, which counts on lgkmcnt as well)
    typedef float __attribute__((address_space(1))) gfloat;
    typedef f32x4 __attribute__((address_space(1))) gf32x4;
    auto uniform_ptr = [&](const float* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<gfloat*>(((unsigned long long)hi << 32) | lo);
    };
    auto lane_off = [&](bool pl) {           // per-lane element offset of this lane's 4 couts of x position px (recomputed per tile)
        int l = lane;
        asm volatile("" : "+v"(l));
        const int pxx = l & 15, hh = l >> 4;
        return pl ? hh * dim * dim * dim * 4 + pxx * 4 : pxx * a.cout + 4 * hh;      // quad-planar: the lane's 4 couts are ONE record of plane hh
    };
    const int o_ys = out_pl ? dim * 4 : dim * a.cout, o_zs = o_ys * dim;
    const int r_ys = res_pl ? dim * 4 : dim * a.cout, r_zs = r_ys * dim;
    auto tile_base = [&](const float* t, const UnitP& u, bool pl) {
        const int gz0 = u.z0 + 4 * zt, gy0 = u.y0 + 4 * G;
        const long long cl = ((((long long)u.b * dim + gz0) * dim + gy0) * dim + u.x0) * a.cout + u.cb * 32 + ct * 16;
        const long long qp = (((((long long)u.b * (a.cout >> 2) + u.cb * 8 + ct * 4) * dim + gz0) * dim + gy0) * dim + u.x0) * 4;
        return uniform_ptr(t + (pl ? qp : cl));
    };
    // skip tensor rv[y][z]; the fused 1x1x1 skip convolution reads its 16-channel channels-last input the same way (4 channels per k lane)
    f32x4 wsk = {0.f, 0.f, 0.f, 0.f};
    auto res_ptr = [&](const UnitP& u) {
        if constexpr (skc) {
            const int gz0 = u.z0 + 4 * zt, gy0 = u.y0 + 4 * G;
            // the skip convolution's 16-channel input: channels-last [B][D^3][16], or (round 6, bit 2) quad-planar [B][4][D^3][4] - what
            // the frequency-domain front layer writes: the lane's 4 channels are one record of plane h
            if constexpr (res_pl) return uniform_ptr(a.res + ((((long long)u.b * 4 * dim + gz0) * dim + gy0) * dim + u.x0) * 4);
            return uniform_ptr(a.res + ((((long long)u.b * dim + gz0) * dim + gy0) * dim + u.x0) * 16);
        } else {
            return tile_base(a.res, u, res_pl);
        }
    };
    // Raw buffer loads, issued for EVERY tile: without a skip tensor the descriptor has zero records and the loads return zeros
    // without touching memory - no run-time condition around the 16 loads or the 16 adds (with one, hipcc kept the skip registers
    // live on the path that never loads them and spilled 44 registers per wave).
    // the skip tensor is read once, by this launch: non-temporal loads (aux 2).  Round 5, tools/ab_libs.py: 32->32 @64^3 0.3792 -> 0.3682 ms,
    // 64->64 @32^3 0.1754 -> 0.1714 in isolation; inside the forward 0.3292 -> 0.3269 ms per launch; non-temporal output STORES: nothing (+0.2 %)
#ifndef SE_K44P_RES_AUX
#define SE_K44P_RES_AUX 2
#endif
    const int rk_ys = (skc && res_pl) ? dim * 4 : dim * 16, rk_zs = rk_ys * dim;
    auto load_rv = [&](f32x4 (&rv)[4][4], const gfloat* rb) {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int r_lane = skc ? (res_pl ? (l >> 4) * dim * dim * dim * 4 + (l & 15) * 4 : (l & 15) * 16 + 4 * (l >> 4)) : lane_off(res_pl);
        const bool on = (skc || use_res) && !(SE_K44P_EXP & 16);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((float*)rb, 0, on ? 0x7fffffff : 0, 0x00020000);
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int z = 0; z < 4; ++z)
                rv[y][z] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, r_lane * 4, (skc ? z * rk_zs + y * rk_ys : z * r_zs + y * r_ys) * 4, SE_K44P_RES_AUX));
    };
    // Part 1 (staging half 1, behind pass 1; the other group is in the first half of its MFMA phase): the first half of the skip
    // tensor goes out, the output transform along y runs IN PLACE in the accumulator registers - xi_z by xi_z,
    // acc[6 y + xi_z] <- t[y][xi_z], xi_y = 4, 5 of that column die - and the second half of the skip tensor is requested into the
    // registers that died.  Part 2 (half 2, behind pass 2): transform along z, y by y, bias, skip, ReLU, 16 x 16-byte stores.
    auto epilogue_y = [&](const UnitP& u, f32x4 (&rv)[4][4]) {
#pragma unroll
        for (int xz = 0; xz < 6; ++xz) {
            f32x4 y4[4];
            at43p(acc[0 * 6 + xz], acc[1 * 6 + xz], acc[2 * 6 + xz], acc[3 * 6 + xz], acc[4 * 6 + xz], acc[5 * 6 + xz], y4);
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[y * 6 + xz] = y4[y];
        }
        // the skip tensor goes out behind the y transform, when 48 accumulator registers have died; it lands under the mid
        // barrier and pass 2.  (Pinned: hipcc otherwise hoists the loads above the transform and spills them on arrival,
        // and sinks the register-only transform below any fence that does not name its results)
#pragma unroll
        for (int e = 0; e < 24; ++e) asm volatile("" : "+v"(acc[e]) : : "memory");
        load_rv(rv, res_ptr(u));
    };
    // Part 2 does NOT store: the 16 output vectors of the tile are PARKED in acc[20 + 4 y + z] - the registers of xi quads 5..8, which
    // the next tile's first MFMA phase does not touch before its group 15 - and leave as riders of that phase's first 15 groups
    // (park_store below): a vector-memory instruction costs a staging wave ~200 cycles next to the partner's MFMA stream, an MFMA
    // wave next to nothing (attribution build without the stores: -5.6 % per launch).  The next tile needs no zeroed accumulators
    // either: the first MFMA of every accumulator in a tile's first step takes an inline-constant zero as its C operand.
    // y runs 3..0: y = 3 reads acc[18..23] before y = 0 parks into acc[20..23].
    gfloat* p_ob = nullptr;             // output rows of the tile whose outputs are parked
    const int o_lane = lane_off(out_pl);
    auto epilogue_z = [&](const UnitP& u, f32x4 (&rv)[4][4]) {
        p_ob = tile_base(a.out, u, out_pl);
        const f32x4 bias = *reinterpret_cast<const gf32x4*>(uniform_ptr(a.bpack + u.cb * 32 + ct * 16) + 4 * h);
        if constexpr (skc) wsk = *reinterpret_cast<const gf32x4*>(uniform_ptr(a.skip_w + (u.cb * 32 + ct * 16) * 16) + px * 16 + 4 * h);
#pragma unroll
        for (int yy = 0; yy < 4; ++yy) {
            const int y = 3 - yy;
            f32x4 o[4];
            at43p(acc[y * 6 + 0], acc[y * 6 + 1], acc[y * 6 + 2], acc[y * 6 + 3], acc[y * 6 + 4], acc[y * 6 + 5], o);
            if constexpr (skc) {
#pragma unroll
                for (int z = 0; z < 4; ++z) o[z] += bias;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)          // k step outer: the four accumulators alternate (no dependent back-to-back MFMAs)
#pragma unroll
                    for (int z = 0; z < 4; ++z) o[z] = __builtin_amdgcn_mfma_f32_16x16x4f32(wsk[ks], rv[y][z][ks], o[z], 0, 0, 0);
            }
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                f32x4 v = o[z];
                if constexpr (!skc) {
                    v += bias;
                    v += rv[y][z];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], relu_lo);
                acc[20 + 4 * y + z] = v;
            }
        }
        // fused 2x2x2 max-pool (se_conv3d_pool_f32): the wave's tile is 4 (z) x 4 (y) x 16 (x) outputs = 2 x 2 x 8 pooled voxels; z and y
        // pairs sit in this lane's registers, the x neighbour in the adjacent lane (quad_perm swap); even-x lanes store 16 bytes of
        // the channels-last pooled tensor [B][D/2][D/2][D/2][cout]
        if constexpr (pool) {
            const int hd = dim >> 1;
            const int gz0 = u.z0 + 4 * zt, gy0 = u.y0 + 4 * G;
            float* pb = a.pool_out + ((((long long)u.b * hd + (gz0 >> 1)) * hd + (gy0 >> 1)) * hd + (u.x0 >> 1) + (px >> 1)) * a.cout
                        + u.cb * 32 + ct * 16 + 4 * h;
#pragma unroll
            for (int pz = 0; pz < 2; ++pz)
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    f32x4 m;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float m4 = fmaxf(fmaxf(acc[20 + 4 * (2 * py) + 2 * pz][c], acc[20 + 4 * (2 * py) + 2 * pz + 1][c]),
                                               fmaxf(acc[20 + 4 * (2 * py + 1) + 2 * pz][c], acc[20 + 4 * (2 * py + 1) + 2 * pz + 1][c]));
                        const float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m4), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                        m[c] = fmaxf(m4, nb);
                    }
                    if (!(px & 1)) *reinterpret_cast<f32x4*>(pb + ((long long)pz * hd + py) * hd * a.cout) = m;
                }
        }
        // accumulators 0..19 are dead from here (the next tile's first step defines them): say so, or they stay live around the loop
#pragma unroll
        for (int e = 0; e < 20; ++e) asm volatile("" : "=v"(acc[e]));
    };
    // parked output k = 4 y + z of the tile behind p_ob
    auto park_store = [&](auto k_tag) {
        constexpr int k = decltype(k_tag)::value;
        constexpr int y = k >> 2, z = k & 3;
        if (!(SE_K44P_EXP & 32) || acc[20 + k].x == 12345.f) *reinterpret_cast<gf32x4*>(p_ob + z * o_zs + y * o_ys + o_lane) = acc[20 + k];
    };

    // State of the walk: (ucur, ccur) = the step this group computes next, (unx, cnx) = the step after it.
    UnitP ucur = decode(u_first);
    UnitP unx = ucur;
    int cnx = 0, inx = 0;
    int slot = 0;                   // slot of half A of step (ucur, ccur) = (2 step) mod 3; half B: slot + 1 (mod 3)

    auto half_src = [&](const UnitP& u, int c4, int half) {
        return wg + ((size_t)u.cb * chunks + c4) * SE_WINO44_CHUNK_FLOATS + half * P_HALF_A;
    };

    // ---- MFMA phase of group GG: 27 groups (quad, dx) of 4 MFMAs; the workgroup's mid-phase barrier sits in front of the first
    // access to weight half B.  Riders (a few per group of MFMAs, where they are nearly free for the issuing wave):
    //   group 0: the LDS-DMAs of half A of the NEXT step (6 per wave) from group 1 on, then the six input rows of its next step from group 8;
    //   group 1: its six input rows in groups 1..6, the LDS-DMAs of half B of the next step (8 per wave) right behind the mid barrier
    //   (A/B in one process, 32->32 @64^3: rows from group 8 instead of 10 -2 %, DMAs one group earlier -1 %, no wave priorities +4 %);
    //   the first step of a tile also carries the 16 parked output stores of the tile before it (groups 0..14).
    // KIND: 0 = first step of the workgroup's first tile, 1 = first step of a later tile (carries the 16 parked output stores of the
    // tile before it), 2 = middle step, 3 = last step of a tile.  In the first step every accumulator's first MFMA (dx = 0) takes C = 0.
    auto mfma_phase = [&](auto gg_tag, auto kind_tag) {
        constexpr int GG = decltype(gg_tag)::value;
        constexpr int KIND = decltype(kind_tag)::value;
        const int s_a = slot, s_b = slot == 2 ? 0 : slot + 1, s_n = slot == 0 ? 2 : slot - 1;      // s_n = (slot + 2) mod 3: half A of the next step
        const float* a_h0 = a_lane + s_a * P_SLOT;
        const float* a_h1 = a_lane + s_b * P_SLOT;
        const float* w_next = half_src(unx, cnx, GG == 0 ? 0 : 1);
        float* dma_dst = wl + (GG == 0 ? s_n : s_a) * P_SLOT;
        f32x4 oa[3], ov[3];
        auto read_ops = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            constexpr int q = g / 3, dx = g % 3, b = g % 3;
            if constexpr (q < P_QA) oa[b] = *reinterpret_cast<const f32x4*>(a_h0 + (q * 3 + dx) * 512);
            else oa[b] = *reinterpret_cast<const f32x4*>(a_h1 + ((q - P_QA) * 3 + dx) * 512);
            ov[b] = *reinterpret_cast<const f32x4*>(b_base + dx * P_RS + q * 4);
        };
        read_ops(std::integral_constant<int, 0>{});
        read_ops(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // pin the first four reads in front of the pipeline (conv3d_wino2d.hip)
        auto group = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            constexpr int q = g / 3, b = g % 3;
#ifndef SE_K44P_DMA1
#define SE_K44P_DMA1 P_GA
#endif
#ifndef SE_K44P_ROW0
#define SE_K44P_ROW0 8
#endif
#ifndef SE_K44P_DMA0
#define SE_K44P_DMA0 1       // first group of group 0's phase that carries an LDS-DMA (behind its row loads when > SE_K44P_ROW0)
#endif
#ifndef SE_K44P_PRIO
#define SE_K44P_PRIO 3
#endif
            constexpr int dma0 = GG == 0 ? SE_K44P_DMA0 : SE_K44P_DMA1;    // first group that carries an LDS-DMA
            constexpr int row0 = GG == 0 ? SE_K44P_ROW0 : 1;               // ... an input row
            constexpr bool dma = g >= dma0 && g < dma0 + (GG == 0 ? P_DMA_A : P_DMA_B);
            constexpr bool row = g >= row0 && g < row0 + 6;
            if constexpr (g == P_GA) {
                // everybody is past half A; the first reads of half B sit behind this barrier
                TP(0)
                barrier();
                TP(1)
                read_ops(std::integral_constant<int, P_GA>{});
                read_ops(std::integral_constant<int, P_GA + 1>{});
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
            if constexpr (dma) {
                if constexpr (GG == 0) wdma(w_next, dma_dst, NA{}, std::integral_constant<int, dma ? g - dma0 : 0>{});
                else wdma(w_next, dma_dst, NB{}, std::integral_constant<int, dma ? g - dma0 : 0>{});
            }
            if constexpr (row) fetch_one(unx, cnx, std::integral_constant<int, row ? g - row0 : 0>{});
            constexpr bool pre = g + 2 < P_GROUPS && g + 2 != P_GA && g + 2 != P_GA + 1;
            if constexpr (pre) read_ops(std::integral_constant<int, g + 2>{});
            // parked outputs of the previous tile: one store per group in groups 0..13, two in group 14 (quad 5 = acc[20..23] starts at 15)
            constexpr int nst = KIND == 1 ? (g < 14 ? 1 : g == 14 ? 2 : 0) : 0;
            if constexpr (nst >= 1) park_store(std::integral_constant<int, nst ? g : 0>{});
            if constexpr (nst == 2) park_store(std::integral_constant<int, 15>{});
            constexpr bool zero_c = KIND <= 1 && g % 3 == 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[4 * q + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][j], ov[b][j], zero_c ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[4 * q + j], 0, 0, 0);
            // issue order inside the group: operand reads and the rider between the MFMAs
            if constexpr (pre) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            if constexpr (pre) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if constexpr (row) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if constexpr (nst > 0) __builtin_amdgcn_sched_group_barrier(0x040, nst, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        };
        for_each_c(group, std::make_integer_sequence<int, P_GROUPS>{});
    };

    // ---- prologue: V of step 0 for both groups (group 1's pass-1 scratch is weight slot 2, which nobody uses before group 0's
    // first MFMA phase), weight halves A, B of step 0 into slots 0, 1 by all eight waves ----
    fetch_setup(ucur);
    for_each_c([&](auto r_tag) { fetch_one(ucur, 0, r_tag); }, std::make_integer_sequence<int, 6>{});
    {
        const float* src = wg + ((size_t)ucur.cb * chunks) * SE_WINO44_CHUNK_FLOATS;
        for (int i = tid; i < SE_WINO44_CHUNK_FLOATS / 4; i += 512) {
            const int fl = i * 4;
            const int dst = fl < P_HALF_A ? fl : P_SLOT + (fl - P_HALF_A);
            *reinterpret_cast<f32x4*>(wl + dst) = *reinterpret_cast<const f32x4*>(src + fl);
        }
    }
    float* t_pro = G == 0 ? tt : wl + 2 * P_SLOT;
    pass1(t_pro);
    __syncthreads();
    pass2(t_pro);
    step_after(unx, cnx, inx);
    if (cnx == 0) fetch_setup(unx);
    __syncthreads();

    // The main loop exists once per group (its MFMA phase differs).  A workgroup walks whole tiles, so the loop is a tile loop with
    // the first and the last step of a tile peeled: which step carries the parked stores, which one ends in an epilogue and which
    // accumulators are dead where is then control flow the compiler sees, not a run-time flag.
    auto run = [&](auto gg_tag) {
        constexpr int GG = decltype(gg_tag)::value;
        // staging phase behind an MFMA phase; `raw` holds the group's input rows of the next step.  Half 1: B^T along y into T; half 2:
        // B^T along z into V and the walk.  LAST: the epilogue of the finished tile around them.
        auto staging = [&](auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            // rows: group 0's are its youngest vector-memory operations; behind group 1's fly the LDS-DMAs of half B
            constexpr bool dma_young = GG == 1 || SE_K44P_DMA0 > SE_K44P_ROW0;       // this group's DMAs were issued behind its rows
            if constexpr (!dma_young) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(GG == 0 ? P_DMA_A : P_DMA_B) : "memory");
            pass1(tt);
            f32x4 rv[4][4];
            if constexpr (LAST) {
                if (SE_K44P_EPRIO) __builtin_amdgcn_s_setprio(SE_K44P_EPRIO);
                epilogue_y(ucur, rv);
                // the weight half this group issued has landed before anybody reads it (sixteen skip-tensor loads are younger)
                if constexpr (dma_young) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else {
                if constexpr (dma_young) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the weight half this group issued has landed
            }
            TP(4)
            barrier();                                                // mid-phase barrier
            TP(5)
            pass2(tt);
            TP(6)
            if constexpr (LAST) {
                epilogue_z(ucur, rv);
                if (SE_K44P_EPRIO) __builtin_amdgcn_s_setprio(0);
            }
            TP(7)
            ucur = unx;
            slot = slot == 0 ? 2 : slot - 1;                          // (slot + 2) mod 3
            step_after(unx, cnx, inx);
            if (cnx == 0) fetch_setup(unx);                           // new unit (or, behind the last step, the same one again)
            TP(8)
            barrier();                                                // end of the staging phase
            TP(9)
        };
        auto step = [&](auto kind_tag) {
            constexpr int KIND = decltype(kind_tag)::value;
#ifdef SE_STAMP44P
            { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
            if (SE_K44P_PRIO > 0) __builtin_amdgcn_s_setprio(SE_K44P_PRIO);
            if (SE_K44P_PRIO < 0) __builtin_amdgcn_s_setprio(0);                 // experiment: the STAGING wave above the MFMA wave
            mfma_phase(gg_tag, kind_tag);
            if (SE_K44P_PRIO > 0) __builtin_amdgcn_s_setprio(0);
            if (SE_K44P_PRIO < 0) __builtin_amdgcn_s_setprio(-(SE_K44P_PRIO));
            TP(2)
            barrier();                                                // end of the MFMA phase
            TP(3)
            staging(std::integral_constant<bool, KIND == 3>{});
        };
        if constexpr (GG == 1) {          // group 1 runs one phase behind group 0
            barrier();
            barrier();
        }
        const int n_tiles = n_mine;
        for (int t = 0; t < n_tiles; ++t) {
            if (t == 0) step(std::integral_constant<int, 0>{});
            else step(std::integral_constant<int, 1>{});
            for (int c = 1; c + 1 < chunks; ++c) step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
        }
        // the last tile's outputs are still parked
        for_each_c(park_store, std::make_integer_sequence<int, 16>{});
        if constexpr (GG == 0) {          // group 0 idles through group 1's last MFMA phase
            barrier();
            barrier();
        }
    };
    if (G == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
#ifdef SE_STAMP44P
    if (lane == 0 && dbg) {
        unsigned long long* o = dbg + ((size_t)blockIdx.x * 8 + wave) * 12;
        for (int k = 0; k < 11; ++k) o[k] = st_sum[k];
        o[11] = n_steps;
    }
#endif
}

}  // namespace

// Section I of the packed 3x3x3 weights (appended by se_conv3d_pack_f32): per (32-cout block cb, 4-channel chunk)
//   [q 9][dx 3][ct 2][lane 64][j 4] = U[xi = 4 q + j][dx] of cout cb*32 + ct*16 + (lane & 15), cin chunk*4 + (lane >> 4);
//   xi = 6 xi_y + xi_z;  U = (G43 (x) G43) g over (dz, dy), times the folded BatchNorm scale.
__global__ void pack_k3_wino44_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                      float eps, float* __restrict__ out, int cout, int cin, int cin_pad, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t & 3);
    const int lane = (int)((t >> 2) & 63);
    long long r = t >> 8;
    const int ct = (int)(r % 2); r /= 2;
    const int dx = (int)(r % 3); r /= 3;
    const int q = (int)(r % 9); r /= 9;
    const int chunks = cin_pad / 4;
    const int chunk = (int)(r % chunks);
    const int cb = (int)(r / chunks);
    const int xi = 4 * q + j, xy = xi / 6, xz = xi % 6;
    const int co = cb * 32 + ct * 16 + (lane & 15);
    const int ci = chunk * 4 + (lane >> 4);
    float v = 0.f;
    if (co < cout && ci < cin) {
        const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
        const float* wp = w + ((size_t)co * cin + ci) * 27 + dx;
        // G of F(4,3), points {0, 1, -1, 2, -2, inf}
        const float g43[6][3] = {{0.25f, 0.f, 0.f},          {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                                 {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
        double u = 0.0;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) u += (double)g43[xz][kz] * (double)g43[xy][ky] * (double)wp[kz * 9 + ky * 3];
        v = (float)(u * (double)sc);
    }
    out[t] = v;
}

int se_conv3d_pack_wino44(const float* w, const float* gamma, const float* var, float eps, float* out, int cout, int cin,
                          int cin_pad, hipStream_t s) {
    const long long total = (long long)(cout / 32) * (cin_pad / 4) * SE_WINO44_CHUNK_FLOATS;
    hipLaunchKernelGGL(pack_k3_wino44_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, gamma, var, eps, out, cout,
                       cin, cin_pad, total);
    SE_CHECK_LAUNCH();
    return 0;
}

// Shapes this kernel takes (the caller, se_conv3d_wino2d_try, has checked se_wino2d_shape_ok and cin_pad == cin): dim >= 32 - at 16^3
// a batch of 8 has only 128 tiles of 8 x 8 x 16 per cout block - and enough units to give every CU one.
bool se_conv3d_wino44pp_shape(int batch, int dim, int cout) {
    if (dim < 32) return false;
    const long long units = (long long)batch * (dim / 16) * (dim / 8) * (dim / 8) * (cout / 32);
    return units >= se_num_cus() || dim >= 64;
}
// A channels-last input with >= 32 channels stays on the F(4,3) x F(2,3) kernel: a 4-channel chunk is 16 bytes of every 128-byte
// record there (measured 0.578 against 0.506 ms at 32->32 @64^3).
// input layouts: quad-planar, or channels-last with fewer than 32 channels (a 4-channel chunk is 16 bytes of every cin * 4-byte
// record); any octet-planar tensor belongs to the F(4,3) x F(2,3) kernel
bool se_conv3d_wino44pp_layout_ok(int cin, int flags) {
    if (flags & (SE_IN_OCTET | SE_OUT_OCTET | SE_RES_OCTET)) return false;
    return (flags & SE_IN_QUAD) || cin < 32;
}
bool se_conv3d_wino44pp_takes(const ConvArgs& a, int batch) {
    if (!se_conv3d_wino44pp_layout_ok(a.cin, a.flags)) return false;
    return a.wpack_i && se_conv3d_wino44pp_shape(batch, a.dim, a.cout);
}

#if defined(SE_STAMP44P)
extern "C" void se_debug_set_stamp_buffer_44p(void* p) { g_w44p_dbg = reinterpret_cast<unsigned long long*>(p); }
#endif

// Returns 0 on launch, SE_ERR_BAD_ARG for a flag combination that is not instantiated, else a hipError_t.
int se_conv3d_wino44pp_launch(const ConvArgs& a, int batch, hipStream_t s) {
    const int dim = a.dim;
    const int tx = dim / 16, ty = dim / 8, tz = dim / 8;
    const long long total_tiles = (long long)batch * tx * ty * tz;
    const long long n_units = total_tiles * (a.cout / 32);
    if (n_units >= (1LL << 30)) return SE_TILED_NOT_TAKEN;
    const int cus = se_num_cus();
    const int grid = (int)(n_units < cus ? n_units : cus);
    const int per = (int)((n_units + grid - 1) / grid);
    unsigned long long* dbg = nullptr;
#ifdef SE_STAMP44P
    dbg = g_w44p_dbg;
#endif
#define P_LAUNCH(L)                                                                                                             \
    do {                                                                                                                        \
        auto kern = conv3d_k3_wino44pp_kernel<L>;                                                                               \
        SE_ENSURE_LDS(kern, P_LDS_BYTES);                                                                                       \
        hipLaunchKernelGGL(kern, dim3((unsigned)((n_units + per - 1) / per)), dim3(512), P_LDS_BYTES, s, a, a.wpack_i, tx, ty,  \
                           tz, (int)total_tiles, (int)n_units, per, dbg);                                                       \
    } while (0)
    const int layout = ((a.flags & SE_IN_QUAD) ? 1 : 0) | ((a.flags & SE_OUT_QUAD) ? 2 : 0) | ((a.flags & SE_RES_QUAD) && a.res ? 4 : 0);
    if (a.flags & SE_EPI_SKIPCONV16) {
        if ((layout != 3 && layout != 7) || a.pool_out || !a.skip_w || !a.res) return SE_ERR_BAD_ARG;
        if (layout == 7) P_LAUNCH(23);      // the 16-channel skip input is quad-planar
        else P_LAUNCH(19);
        SE_CHECK_LAUNCH();
        return 0;
    }
    if (a.pool_out) {
        if (layout == 3) P_LAUNCH(11);
        else if (layout == 7) P_LAUNCH(15);
        else return SE_ERR_BAD_ARG;       // pooled output: quad-planar in / out only (what the V2V program uses)
        SE_CHECK_LAUNCH();
        return 0;
    }
    switch (layout) {
        case 1: P_LAUNCH(1); break;
        case 2: P_LAUNCH(2); break;
        case 3: P_LAUNCH(3); break;
        case 4: P_LAUNCH(4); break;
        case 5: P_LAUNCH(5); break;
        case 6: P_LAUNCH(6); break;
        case 7: P_LAUNCH(7); break;
        default: P_LAUNCH(0); break;
    }
#undef P_LAUNCH
    SE_CHECK_LAUNCH();
    return 0;
}
