// 3x3x3 convolution, float32, 2-D Winograd F(4,3) (z) x F(2,3) (y), direct along x, on v_mfma_f32_16x16x4_f32.
// Stands in for Conv3d(k=3) + BatchNorm3d (+ReLU) (+skip add) of Res3DBlock (reference network/v2v.py:21-43) at the
// 64^3 / 32^3 / 16^3 levels.
//
// Why: the 3^3 layers are matrix-pipe bound (218 FLOP/B), so the lever is the NUMBER of products.  Per 4(z) x 2(y) outputs and
// x tap the transform domain has 6 x 4 = 24 points instead of 4*2*3*3 = 72 products: 1/3 of the direct MFMAs (the 1-D F(4,3)
// kernel of conv3d_wino.hip executes 1/2).  The second lever is memory: that kernel walks the input channels in the OUTER loop
// and carries its partial sums through the output tensor (1.7x the algorithmic HBM bytes); here the accumulators of a tile stay
// in registers over ALL input channels and the output is written once.
//
// Work unit = (32-cout block, tile of 4(z) x 8(y) x 16(x) outputs); a persistent 512-thread workgroup per CU walks a contiguous
// range of units.  The 8 waves form two groups of 4 (one wave per SIMD each).  Group G owns the y half [4G, 4G+4) of the tile
// = two y-tiles of 2 rows; inside a group wave (ct, j) computes cout tile ct for y-tile j: 16 x positions on the MFMA columns,
// 24 (xi_z, xi_y) accumulators of 16 couts x 16 positions = 96 registers.
// Channels are walked in chunks of 8 (two MFMA k steps; k lane h carries channels 2h, 2h+1).  Per chunk and group a "step" is
//     MFMA phase     144 MFMAs per wave: 24 xi x 3 dx x 2, operands by ds_read_b64 from
//                      W [xi_z][xi_y][dx][ct][lane][2]             73.7 KB, G-transformed weights of the (cout block, chunk)
//                      V [xi][y-tile][18 x records of 8 channels]   27 KB per group, B^T-transformed input
//     staging phase  the other group meanwhile: output transform + epilogue + stores of the tile it just finished (only after the
//                    last chunk), global loads + 2-D B^T transform + LDS commit of its next chunk, and its share of the weight stream
// and the two groups run half a step apart (as in conv3d_k3_wino43pp_kernel), so each SIMD always has one wave in its MFMA block.
// The weights of a chunk are re-streamed from L2 for every (unit, chunk) - 8 B/clk/CU - into ONE chunk buffer: its halves (xi_z < 3
// / >= 3) are consumed a quarter step apart, so the half a group has finished with is refilled while the other half is in use;
// two workgroup barriers per phase (start / middle) order this.
// LDS reads are conflict-free: W is lane-linear; in V the two 16-byte halves of an 8-channel record are swapped for x records
// 8..15, which puts the 32 lanes of a ds_read_b64 pass (16 positions x 2 k lanes) on 64 different banks for each of the three
// x-shifted windows.
#include "common.h"

#include "conv_common.h"

#include <type_traits>
#include <utility>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int W2_HALF_FLOATS = 3 * 4 * 3 * 2 * 128;      // xi_z 0..2 (or 3..5): 9216 floats = 36,864 B
constexpr int W2_CHUNK_FLOATS = SE_WINO2D_CHUNK_FLOATS;  // 18,432 floats = 73,728 B
constexpr int W2_VROW = 144;                             // floats per (xi, y-tile) row: 18 x records of 8 channels
constexpr int W2_VG_FLOATS = 24 * 2 * W2_VROW;           // 6912 floats = 27,648 B per group
constexpr int W2_LDS_BYTES = (W2_CHUNK_FLOATS + 2 * W2_VG_FLOATS) * 4;   // 129,024 B
static_assert(W2_CHUNK_FLOATS == 2 * W2_HALF_FLOATS, "chunk = two halves");

template <typename F, int... S>
__device__ __forceinline__ void for_each_idx(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}

// float offset of channel pair p (k lane) of x record xx inside a V row
__device__ __forceinline__ int v_rec_offset(int xx, int p) { return xx * 8 + ((((p >> 1) ^ (xx >> 3)) & 1) << 2) + (p & 1) * 2; }

struct Unit {
    int cb, b, z0, y0, x0;
};

__global__ __launch_bounds__(512) void conv3d_k3_wino2d_kernel(ConvArgs a, const float* __restrict__ wg, int tiles_x, int tiles_y,
                                                               int tiles_z, int total_tiles, int n_units, int units_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wave-uniform by construction; readfirstlane makes that provable, so the unit walk below stays in scalar registers and
    // the buffer descriptors need no waterfall loops
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = wave >> 2;                    // group: y half of the tile
    const int wq = wave & 3;
    const int ct = wq >> 1;                     // cout tile of the 32-cout block
    const int jt = wq & 1;                      // y-tile of the group
    const int px = lane & 15;                   // x position (MFMA column)
    const int h = lane >> 4;                    // MFMA k lane
    const int dim = a.dim;
    const int cin = a.cin;
    const int chunks = cin >> 3;
    const int u_begin = (int)blockIdx.x * units_per_wg;
    const int u_end = min(u_begin + units_per_wg, n_units);
    if (u_begin >= u_end) return;
    const int n_steps = (u_end - u_begin) * chunks;
    float* vt = lds + W2_CHUNK_FLOATS + G * W2_VG_FLOATS;

    // ---- MFMA operand addresses ----
    const float* a_h0 = wl + ct * 128 + lane * 2;
    const float* a_h1 = a_h0 + W2_HALF_FLOATS;
    const float* b_dx[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) b_dx[dx] = vt + jt * W2_VROW + v_rec_offset(px + dx, h);

    // ---- staging role inside the group: thread tg < 144 owns (y-tile sj, x record sxx, channel pair sp) ----
    const int tg = tid & 255;
    const bool s_on = tg < 144;
    const int sp = tg & 3;
    const int sq = s_on ? (tg >> 2) : 0;
    const int sxx = sq % 18, sj = sq / 18;
    float* v_w = vt + sj * W2_VROW + v_rec_offset(sxx, sp);

    auto decode = [&](int u) {
        Unit r;
        r.cb = u / total_tiles;
        int t = u - r.cb * total_tiles;
        const int xt = t % tiles_x; t /= tiles_x;
        const int yt = t % tiles_y; t /= tiles_y;
        const int zt = t % tiles_z;
        r.b = t / tiles_z;
        r.z0 = zt * 4; r.y0 = yt * 8; r.x0 = xt * 16;
        return r;
    };
    // the unit after u in the walk (x fastest, then y, z, sample, cout block): carries instead of divisions
    auto advance = [&](Unit u) {
        u.x0 += 16;
        if (u.x0 == dim) {
            u.x0 = 0; u.y0 += 8;
            if (u.y0 == dim) {
                u.y0 = 0; u.z0 += 4;
                if (u.z0 == dim) {
                    u.z0 = 0; u.b += 1;
                    if (u.b * tiles_z * tiles_y * tiles_x == total_tiles) { u.b = 0; u.cb += 1; }
                }
            }
        }
        return u;
    };

    // Global accesses go through raw buffer descriptors of ONE sample (base = sample b, num_records = bytes of a sample):
    // per-lane part of the address in one 32-bit voffset, the uniform (z, y) part in the scalar offset, out-of-volume lanes
    // get voffset = num_records and read zero.
    const unsigned in_bytes = (unsigned)dim * dim * dim * cin * 4u;
    auto rsrc_of = [&](const float* base, long long sample_floats, int b, unsigned bytes) {
        const float* p0 = base + (long long)b * sample_floats;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p0), 0, (int)bytes, 0x00020000);
    };

    f32x2 raw[6][4];
    // global loads of one chunk of this thread's halo column block: 6 z slabs x 4 rows, 8 bytes each
    auto fetch = [&](const Unit& u, int chunk) {
        const auto rs = rsrc_of(a.in, (long long)dim * dim * dim * cin, u.b, in_bytes);
        const int gx = u.x0 - 1 + sxx;
        const bool okx = s_on && (unsigned)gx < (unsigned)dim;
        const int gy0 = u.y0 + G * 4 + sj * 2 - 1;
        const int gz0 = u.z0 - 1;
        // per-lane byte offset of (row gy0 + r, column gx, channel pair) inside a z slab; the scalar offset carries the slab
        // (scalar offsets are unsigned: nothing negative may go there)
        unsigned voff[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gy = gy0 + r;
            voff[r] = (okx && (unsigned)gy < (unsigned)dim) ? (unsigned)(((gy * dim + gx) * cin + chunk * 8 + sp * 2) * 4) : in_bytes;
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int gz = gz0 + s;
            const bool okz = (unsigned)gz < (unsigned)dim;       // uniform
            const int soff = gz * dim * dim * cin * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2 t = {0.f, 0.f};
                if (okz) t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff[r], soff, 0));
                raw[s][r] = t;
            }
        }
    };
    // B^T along z (F(4,3), points 0, +-1, +-2, inf) then along y (F(2,3)), commit to the group's V buffer
    auto commit = [&]() {
        if (!s_on) return;
        f32x2 t[6][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 d0 = raw[0][r], d1 = raw[1][r], d2 = raw[2][r], d3 = raw[3][r], d4 = raw[4][r], d5 = raw[5][r];
            t[0][r] = 4.f * d0 - 5.f * d2 + d4;
            t[5][r] = 4.f * d1 - 5.f * d3 + d5;
            const f32x2 e1 = d4 - 4.f * d2, o1 = d3 - 4.f * d1;
            t[1][r] = e1 + o1;
            t[2][r] = e1 - o1;
            const f32x2 e2 = d4 - d2, o2 = 2.f * (d3 - d1);
            t[3][r] = e2 + o2;
            t[4][r] = e2 - o2;
        }
#pragma unroll
        for (int z = 0; z < 6; ++z) {
            const f32x2 v0 = t[z][0] - t[z][2], v1 = t[z][1] + t[z][2], v2 = t[z][2] - t[z][1], v3 = t[z][1] - t[z][3];
            *reinterpret_cast<f32x2*>(v_w + ((z * 4 + 0) * 2) * W2_VROW) = v0;
            *reinterpret_cast<f32x2*>(v_w + ((z * 4 + 1) * 2) * W2_VROW) = v1;
            *reinterpret_cast<f32x2*>(v_w + ((z * 4 + 2) * 2) * W2_VROW) = v2;
            *reinterpret_cast<f32x2*>(v_w + ((z * 4 + 3) * 2) * W2_VROW) = v3;
        }
    };

    // weight stream: one half chunk (36,864 B = 36 pieces of 1 KiB) straight from L2 into the LDS by LDS-DMA
    // (global_load_lds_dwordx4: lane l of a wave moves 16 bytes to M0 base + 16 l), 9 pieces per wave of the staging group.
    // The pieces land asynchronously: the issuing wave waits with vmcnt before the workgroup barrier that publishes the half.
    auto w_stream = [&](const Unit& u, int chunk, int half) {
        const float* src = wg + ((size_t)u.cb * chunks + chunk) * W2_CHUNK_FLOATS + half * W2_HALF_FLOATS + lane * 4;
        float* dst = wl + half * W2_HALF_FLOATS;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int piece = k * 4 + wq;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + piece * 256),
                                             (void __attribute__((address_space(3)))*)(dst + piece * 256), 16, 0, 0);
        }
    };

    f32x4 acc[24];
    const bool relu = a.flags & SE_EPI_RELU;
    const bool use_res = (a.flags & SE_EPI_RES_PRE_RELU) && a.res;

    // output transform (A^T along y, then along z), bias, residual, ReLU, 8 x 16-byte channels-last stores
    auto epilogue = [&](const Unit& u) {
        const int co = u.cb * 32 + ct * 16 + 4 * h;
        // uniform 64-bit base of the wave's first output row + a 32-bit per-lane offset (global_* saddr form); raw buffer
        // STORES with a scalar offset dropped data here, so stores and skip loads use plain global accesses
        const long long s00 = (((((long long)u.b * dim + u.z0) * dim + u.y0 + G * 4 + jt * 2) * dim + u.x0) * a.cout + u.cb * 32 + ct * 16);
        const int voff = px * a.cout + 4 * h;
        const int ystride = dim * a.cout, zstride = dim * dim * a.cout;
        float* ob = a.out + s00;
        const float* rb = a.res + s00;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bpack + co);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x4 resv[4];
            if (use_res) {
#pragma unroll
                for (int z = 0; z < 4; ++z) resv[z] = *reinterpret_cast<const f32x4*>(rb + z * zstride + r * ystride + voff);
            }
            f32x4 m[6];
#pragma unroll
            for (int z = 0; z < 6; ++z)
                m[z] = r == 0 ? acc[z * 4 + 0] + acc[z * 4 + 1] + acc[z * 4 + 2] : acc[z * 4 + 1] - acc[z * 4 + 2] - acc[z * 4 + 3];
            const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
            f32x4 y[4];
            y[0] = m[0] + s12 + s34;
            y[1] = d12 + 2.f * d34;
            y[2] = s12 + 4.f * s34;
            y[3] = d12 + 8.f * d34 + m[5];
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                f32x4 v = y[z] + bias;
                if (use_res) v += resv[z];
                if (relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *reinterpret_cast<f32x4*>(ob + z * zstride + r * ystride + voff) = v;
            }
        }
    };

    // Workgroup barrier that waits for this wave's LDS traffic only (lgkmcnt): global loads stay in flight across it — a
    // __syncthreads() (and, in a kernel that uses LDS-DMA, every fence-based barrier) also waits for vmcnt(0).  The "memory"
    // clobber keeps the compiler from moving LDS / global accesses across it.  LDS-DMA pieces are counted by vmcnt: the wave
    // that issued them waits explicitly (wait_vm) before the barrier that publishes them.
    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto wait_vm0 = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto wait_vm8 = [&]() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); };   // all but the 8 youngest (the epilogue's stores)

    // ---- MFMA phase: 18 groups (xi_z, dx) of 4 xi_y x 2 k steps; the workgroup's mid-phase barrier sits in front of the first
    // access to the second weight half ----
    auto mfma_phase = [&]() {
        f32x2 ca[4], cv[4], na[4], nv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ca[e] = *reinterpret_cast<const f32x2*>(a_h0 + (e * 3 + 0) * 256);
            cv[e] = *reinterpret_cast<const f32x2*>(b_dx[0] + (e * 2) * W2_VROW);
        }
        auto group = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            constexpr int xz = g / 3;
            if constexpr (g == 8) {
                // the prefetch below is the first read of weight half 1; all reads of half 0 have been issued: drain them so the
                // staging group may refill half 0 right after the barrier
                __builtin_amdgcn_sched_barrier(0);
                barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (g + 1 < 18) {
                constexpr int g2 = g + 1, xz2 = g2 / 3, dx2 = g2 % 3;
                const float* ab = (xz2 < 3) ? a_h0 : a_h1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    na[e] = *reinterpret_cast<const f32x2*>(ab + (((xz2 % 3) * 4 + e) * 3 + dx2) * 256);
                    nv[e] = *reinterpret_cast<const f32x2*>(b_dx[dx2] + ((xz2 * 4 + e) * 2) * W2_VROW);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xz * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[e].x, cv[e].x, acc[xz * 4 + e], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xz * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[e].y, cv[e].y, acc[xz * 4 + e], 0, 0, 0);
            if constexpr (g + 1 < 18) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { ca[e] = na[e]; cv[e] = nv[e]; }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        };
        for_each_idx(group, std::make_integer_sequence<int, 18>{});
    };

    // ---- prologue: weights of step 0 (group G streams half G), each group's V tile of step 0 ----
    Unit ucur = decode(u_begin);      // unit of the step this group computes next / has just computed
    int ccur = 0;                     // its chunk
    w_stream(ucur, 0, G);
    fetch(ucur, 0);
    commit();
    wait_vm0();
    barrier();

    for (int p = 0; p <= 2 * n_steps; ++p) {
        const int r = p - G;
        if (r >= 0 && !(r & 1) && (r >> 1) < n_steps) {
            // ------------------------------ MFMA phase of step r/2 = (ucur, ccur) ------------------------------
            if (ccur == 0) {
#pragma unroll
                for (int e = 0; e < 24; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_s_setprio(3);
            mfma_phase();
            __builtin_amdgcn_s_setprio(0);
        } else {
            // ------------------------------ staging phase ------------------------------
            const bool stage = r >= 1 && (r & 1);                 // this group has just computed step (r-1)/2 = (ucur, ccur)
            const bool has_next = stage && ((r + 1) >> 1) < n_steps;
            const bool epi = stage && ccur == chunks - 1;
            Unit unext = ucur;
            int cnext = ccur + 1;
            if (cnext == chunks) { cnext = 0; unext = advance(ucur); }
            // first half.  Group 1 (even p >= 2) streams weight half 1 of its next step: it must have landed at the mid-phase barrier.
            const bool w1 = G == 1 && has_next;
            if (has_next) fetch(unext, cnext);                    // stays in flight across the mid-phase barrier
            if (w1) w_stream(unext, cnext, 1);
            if (epi) epilogue(ucur);
            if (w1) { if (epi) wait_vm8(); else wait_vm0(); }
            barrier();                                            // mid-phase barrier
            // second half.  Group 0 (odd p) streams weight half 0 of its next step.
            const bool w0 = G == 0 && has_next;
            if (w0) w_stream(unext, cnext, 0);
            if (has_next) commit();
            if (w0) wait_vm0();
            if (stage) { ucur = unext; ccur = cnext; }
        }
        barrier();                                                // end-of-phase barrier
    }
}

}  // namespace

// Section G of the packed 3x3x3 weights (appended by se_conv3d_pack_f32): per (32-cout block cb, 8-channel chunk)
//   [xi_z 6][xi_y 4][dx 3][ct 2][lane 64][e 2] = U[xi_z][xi_y][dx] of cout cb*32 + ct*16 + (lane & 15), cin chunk*8 + 2*(lane >> 4) + e,
//   U = (G43 (x) G23) g over (dz, dy), times the folded BatchNorm scale.
__global__ void pack_k3_wino2d_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                      float eps, float* __restrict__ out, int cout, int cin, int cin_pad, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int e = (int)(t & 1);
    const int lane = (int)((t >> 1) & 63);
    long long r = t >> 7;
    const int ct = (int)(r % 2); r /= 2;
    const int dx = (int)(r % 3); r /= 3;
    const int xy = (int)(r % 4); r /= 4;
    const int xz = (int)(r % 6); r /= 6;
    const int chunks = cin_pad / 8;
    const int chunk = (int)(r % chunks);
    const int cb = (int)(r / chunks);
    const int co = cb * 32 + ct * 16 + (lane & 15);
    const int ci = chunk * 8 + 2 * (lane >> 4) + e;
    float v = 0.f;
    if (co < cout && ci < cin) {
        const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
        const float* wp = w + ((size_t)co * cin + ci) * 27 + dx;
        // G of F(4,3), points {0, 1, -1, 2, -2, inf}; G of F(2,3)
        const float gz[6][3] = {{0.25f, 0.f, 0.f},          {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                                {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
        const float gy[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
        double u = 0.0;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) u += (double)gz[xz][kz] * (double)gy[xy][ky] * (double)wp[kz * 9 + ky * 3];
        v = (float)(u * (double)sc);
    }
    out[t] = v;
}

int se_conv3d_pack_wino2d(const float* w, const float* gamma, const float* var, float eps, float* out, int cout, int cin,
                          int cin_pad, hipStream_t s) {
    const long long total = (long long)(cout / 32) * (cin_pad / 8) * SE_WINO2D_CHUNK_FLOATS;
    hipLaunchKernelGGL(pack_k3_wino2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, gamma, var, eps, out, cout,
                       cin, cin_pad, total);
    SE_CHECK_LAUNCH();
    return 0;
}

// Returns 0 on launch, SE_TILED_NOT_TAKEN if the shape/flags are not covered, else a hipError_t.
int se_conv3d_wino2d_try(const ConvArgs& a, int batch, hipStream_t s) {
    const int dim = a.dim;
    if (!a.wpack_g || dim < 16 || (dim & 15) || (a.cout & 31) || (a.cin & 7) || a.cin_pad != a.cin) return SE_TILED_NOT_TAKEN;
    if (a.flags & (SE_EPI_RES_POST_RELU | SE_EPI_OUT_PLANAR)) return SE_TILED_NOT_TAKEN;
    SE_ENSURE_LDS(conv3d_k3_wino2d_kernel, W2_LDS_BYTES);
    const int tx = dim / 16, ty = dim / 8, tz = dim / 4;
    const long long total_tiles = (long long)batch * tx * ty * tz;
    const long long n_units = total_tiles * (a.cout / 32);
    if (n_units >= (1LL << 30)) return SE_TILED_NOT_TAKEN;
    const int cus = se_num_cus();
    const int grid = (int)(n_units < cus ? n_units : cus);
    const int per = (int)((n_units + grid - 1) / grid);
    hipLaunchKernelGGL(conv3d_k3_wino2d_kernel, dim3((unsigned)((n_units + per - 1) / per)), dim3(512), W2_LDS_BYTES, s, a, a.wpack_g,
                       tx, ty, tz, (int)total_tiles, (int)n_units, per);
    SE_CHECK_LAUNCH();
    return 0;
}
