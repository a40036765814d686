// 7x7x7 convolution (V2V front layer, cout = 16) with the 1-D Winograd transform F(6,7) along z on the f32 MFMA, tile-outer:
// 12 multiplies per 6 z-neighbouring outputs (F(4,7): 10 per 4, direct: 42), ALL input channels of a tile accumulated in registers,
// the output written once.  Replaces Basic3DBlock(33, 16, 7) of network/v2v.py:75-77 (conv + folded BN + ReLU) for volumes with
// dim % 16 == 0; conv3d_wino47.hip (F(4,7), chunk-outer, partial sums through the output tensor: 10x the algorithmic HBM bytes)
// keeps the other shapes.  Points {0, +-1, +-3/4, +-3/2, +-1/3, +-5/2, inf} (tools/wino67_matrices.py -> wino67_matrices.h); float32
// error of the transform on N(0,1) data: 7.3e-6 mean per 7-tap dot product (F(4,7): 3.6e-6).
//
// Work unit = 6(z) x 8(y) x 16(x) output tile, one persistent 512-thread workgroup per CU, wave w = row y of the tile, the 16 MFMA
// columns = the 16 x of that row.  An item = (tile, 3-channel chunk): the 147 (channel, dy, dx) taps of the chunk sit 4 at a time on
// the MFMA k lanes (37 groups, one pad slot), each group is 12 MFMAs (one per xi) from 3 + 3 ds_read_b128:
//   weights  [g(37)][lane][12 xi]            (section H of the packed weights; lane stride 48 B)                        113,664 B
//   inputs   [channel(3)][column(14 x 22)][12 xi]   (B^T applied when the halo is committed; column stride 48 B)         44,352 B
// 48-byte strides: a 16-lane ds_read_b128 group hits 16 distinct 4-bank groups.  A^T is applied after every item and the six
// output-domain sums of a wave live in registers over all chunks of a tile; bias, ReLU and the only store happen once per tile.
//
// Weight stream: the LDS holds ONE chunk of weights, refilled in two regions while the other one is being read - region B (groups
// 16..36) of the CURRENT chunk during the first 16 groups of an item, region A (groups 0..15) of the NEXT chunk during the last 21;
// a workgroup barrier between the two halves and one at the end of the item order this.  The refill is LDS-DMA
// (global_load_lds_dwordx4: 8 + 6 instructions per wave and item, no registers, no ds_write) and, like the 12 halo loads of the next
// item (raw buffer loads: scalar slab offset, one 32-bit column offset per lane, out-of-volume -> zeros by the bounds check), rides
// inside the MFMA stream; L2 serves the 113 KB per item.  Measured steps (33->16 @64^3, B = 8; F(4,7) kernel 2.34 ms): first form
// with register-staged refill and global loads 1.97, + A^T per item 2.06 (accuracy), operand reads 3 + 3 ahead of 10 MFMAs 2.02,
// buffer loads 1.92, LDS-DMA 1.80 ms; without any rider 1.63 ms = the MFMA issue time (profiles/r03_k67_*).
#include "conv_common.h"
#include "wino67_matrices.h"

#include <type_traits>
#include <utility>

namespace {

template <typename F, int... S>
__device__ __forceinline__ void for_each_index(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}

constexpr int S_TZ = 6, S_TY = 8, S_TX = 16;
constexpr int S_HY = S_TY + 6, S_HX = S_TX + 6, S_COLS = S_HY * S_HX;          // 308 halo columns, 12 raw slabs each
constexpr int S_XI = 12;
constexpr int S_G = SE_K7H_GROUPS;                                             // 37 k groups per chunk
constexpr int S_GA = 16;                                                       // weight region A = groups 0..15, region B = 16..36
constexpr int S_W_FLOATS = SE_K7H_CHUNK_FLOATS;                                // 28416
constexpr int S_WA_F4 = S_GA * 64 * 3;                                         // 16-byte pieces of region A
constexpr int S_CS = S_COLS * S_XI;                                            // channel stride of the transformed tile (floats)
constexpr int S_VT_FLOATS = 3 * S_CS;                                          // 11088
constexpr int S_LDS_FIXED = (S_W_FLOATS + S_VT_FLOATS) * 4;                    // 158,016 B

struct f32x3 { float x, y, z; };
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct K7SOps { f32x4 a0, a1, a2, b0, b1, b2; };    // 12 xi of weights (A) and of transformed inputs (B) for one k group

#ifndef SE_K67_EXP      // attribution builds only (results wrong): 1 no weight riders, 2 no halo fetch riders, 4 no mid barrier, 8 no A^T, 16 no commit
#define SE_K67_EXP 0
#endif
#ifdef SE_STAMP67   // cycle stamps (tools/stamp_k67.py; development builds with -DSE_STAMP67)
#define T67(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
                 st_sum[i] += t_ - st_last; st_last = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define T67(i)
#endif

template <bool PLANAR>
__global__ __launch_bounds__(512) void conv3d_k7_wino67_kernel(ConvArgs a, int tiles_x, int tiles_y, int tiles_z, int total_tiles,
                                                               int units_per_wg, int halves, float* part, unsigned long long* dbg) {
    (void)dbg;
#ifdef SE_STAMP67
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* vt = lds + S_W_FLOATS;
    i32x4* utab = reinterpret_cast<i32x4*>(vt + S_VT_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int vl = lane & 15;
    const int h = lane >> 4;
    const int dim = a.dim;
    const int chunks = (a.cin + 2) / 3;
    // this workgroup's tiles: u_first, u_first + u_stride, ... (n of them): a contiguous range of its XCD's share (SE_XCD_WALK 1) or
    // interleaved with the other workgroups of the XCD (2): conv_common.h, conv3d_wino44pp.hip
    int u_first, u_stride, n;
#ifndef SE_K67_XCD_WALK
#define SE_K67_XCD_WALK 1        // the interleaved form measured +0.7 % here (1.797 / 1.782 against 1.775 / 1.777 ms inside the forward): contiguous ranges kept
#endif
    if (SE_K67_XCD_WALK == 2 && (gridDim.x & 7) == 0) {
        const int S = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, w = (int)blockIdx.x >> 3;
        const int r0 = xcd * S * units_per_wg, r1 = min(r0 + S * units_per_wg, total_tiles);
        u_first = r0 + w; u_stride = S;
        n = u_first < r1 ? (r1 - u_first + S - 1) / S : 0;
    } else {
        u_first = se_xcd_walk_index((int)blockIdx.x, (int)gridDim.x) * units_per_wg;
        u_stride = 1;
        n = min(u_first + units_per_wg, total_tiles) - u_first;
    }
    if (n <= 0) return;

    for (int i = tid; i < n; i += 512) {
        int t = u_first + i * u_stride;
        const int half = halves == 2 ? (t & 1) : 0;       // halves == 2: unit = (tile, half of its chunks); total_tiles counts units
        if (halves == 2) t >>= 1;
        i32x4 e;
        e.w = t % tiles_x; t /= tiles_x;
        e.z = t % tiles_y; t /= tiles_y;
        e.y = t % tiles_z; t /= tiles_z;
        e.x = t | (half << 30);
        utab[i] = e;
    }

    // compute role: per-lane LDS offsets (floats) of the 37 k groups: slot 4g+h -> (channel, dy, dx)
    int toff[S_G];
#pragma unroll
    for (int g = 0; g < S_G; ++g) {
        int slot = 4 * g + h;
        slot = slot < 147 ? slot : 0;   // zero-weight padding
        const int cl = slot / 49, tap = slot - cl * 49;
        toff[g] = cl * S_CS + ((wave + tap / 7) * S_HX + vl + tap % 7) * S_XI;
    }

    // staging role: thread t < 308 owns halo column t: 12 raw slabs x 3 channels -> 12 transformed slabs x 3 channels
    const bool s_on = tid < S_COLS;
    const int s_col = s_on ? tid : 0;
    const int s_cy = s_col / S_HX, s_cx = s_col - s_cy * S_HX;
    f32x3 raw[S_XI];
    const long long f_zs = (long long)dim * dim * (PLANAR ? 3 : a.cin_pad);
    // buffer loads: the item's uniform part (sample, chunk) is the descriptor base, the z slab a scalar offset, the column a 32-bit
    // per-lane offset whose bit 31 marks a column outside the volume (-> zeros); a slab outside the volume reads a zero-size descriptor
    const float* f_xb = a.in;
    unsigned f_voff = 0x80000000u;
    int f_gz0 = 0;
    int f_shift = 0;       // channels-last: channels of the last chunk beyond cin_pad (0, 1 or 2): the load starts that many channels early
    const int f_rec_bytes = PLANAR ? dim * dim * dim * 12 : dim * dim * dim * a.cin_pad * 4;
    auto fetch_setup = [&](int k, int c) {
        const i32x4 e = utab[k];
        const int gy = e.z * S_TY - 3 + s_cy, gx = e.w * S_TX - 3 + s_cx;
        f_gz0 = e.y * S_TZ - 3;
        const bool okc = s_on && (unsigned)gy < (unsigned)dim && (unsigned)gx < (unsigned)dim;
        f_shift = PLANAR ? 0 : max(0, c * 3 + 3 - a.cin_pad);
        const int b = __builtin_amdgcn_readfirstlane(e.x) & 0x3fffffff;
        f_xb = PLANAR ? a.in + ((long long)b * chunks + c) * dim * dim * dim * 3
                      : a.in + (long long)b * dim * dim * dim * a.cin_pad + c * 3 - f_shift;
        f_voff = okc ? (unsigned)((gy * dim + gx) * (PLANAR ? 12 : a.cin_pad * 4)) : 0x80000000u;
    };
    auto fetch_one = [&](auto q_tag) {
        constexpr int q = decltype(q_tag)::value;
        const int z = __builtin_amdgcn_readfirstlane(f_gz0) + q;
        const bool zok = (unsigned)z < (unsigned)dim;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f_xb), 0, zok ? f_rec_bytes : 0, 0x00020000);
        typedef float f32x3v __attribute__((ext_vector_type(3)));
        const f32x3v t = __builtin_bit_cast(f32x3v, __builtin_amdgcn_raw_buffer_load_b96(rs, (int)f_voff, z * (int)(f_zs * 4), 0));
        const int shift = f_shift;
        if (PLANAR) {
            raw[q].x = t.x; raw[q].y = t.y; raw[q].z = t.z;
        } else {    // the record never reads past channel cin_pad - 1 (the next voxel's data, or the end of the tensor): slots beyond it are zero
            raw[q].x = shift == 0 ? t.x : shift == 1 ? t.y : t.z;
            raw[q].y = shift == 0 ? t.y : shift == 1 ? t.z : 0.f;
            raw[q].z = shift == 0 ? t.z : 0.f;
        }
    };
    auto fetch = [&](int k, int c) {
        fetch_setup(k, c);
        for_each_index(fetch_one, std::make_integer_sequence<int, S_XI>{});
    };
    auto commit = [&]() {   // V = B^T d per channel: row 0 even q, row 11 odd q, rows 2k+1 / 2k+2 = even part +- odd part; 16-byte stores
        if (!s_on) return;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float d[S_XI];
#pragma unroll
            for (int q = 0; q < S_XI; ++q) d[q] = j == 0 ? raw[q].x : j == 1 ? raw[q].y : raw[q].z;
            // explicit fmaf chains in a fixed order: both input-layout instantiations round identically (bit-equal outputs)
            float o[S_XI];
            float r0 = 0.f, r11 = 0.f;
#pragma unroll
            for (int q = 0; q < S_XI; ++q) {
                if (SE_W67_BT[0][q] != 0.f) r0 = fmaf(SE_W67_BT[0][q], d[q], r0);
                if (SE_W67_BT[11][q] != 0.f) r11 = fmaf(SE_W67_BT[11][q], d[q], r11);
            }
            o[0] = r0; o[11] = r11;
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                const int xi = 2 * p + 1;
                float ev = 0.f, od = 0.f;
#pragma unroll
                for (int q = 1; q < 11; ++q) {
                    const float cf = SE_W67_BT[xi][q];
                    if (q & 1) od = fmaf(cf, d[q], od); else ev = fmaf(cf, d[q], ev);
                }
                o[xi] = ev + od; o[xi + 1] = ev - od;
            }
            f32x4* dst = reinterpret_cast<f32x4*>(vt + j * S_CS + s_col * S_XI);
            dst[0] = (f32x4){o[0], o[1], o[2], o[3]};
            dst[1] = (f32x4){o[4], o[5], o[6], o[7]};
            dst[2] = (f32x4){o[8], o[9], o[10], o[11]};
        }
    };

    // LDS-DMA piece J of a weight region for this wave: wave-instruction 8 J + wave (clamped: the surplus ones repeat the last piece)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto wglds = [&](const float* src, float* region, auto n_tag, auto j_tag) {
        constexpr int NWI = decltype(n_tag)::value, J = decltype(j_tag)::value;
        int piece = J * 8 + wave_u;
        piece = piece < NWI ? piece : NWI - 1;
#ifndef SE_K67_DMA_BUILTIN
        // Inline assembly, not __builtin_amdgcn_global_load_lds (round 4, found on the F(4,3) x F(4,3) kernel): hipcc models the builtin
        // as an LDS access of unknown address, after which every operand wait of the item is an s_waitcnt lgkmcnt(0) - for the reads just
        // issued three groups ahead as well (disassembly: 50 of them per 444 MFMAs).  The landing of the DMAs is covered by the counted
        // vmcnt waits in front of the two barriers below, as before.
        const float* sp = src + piece * 256;        // uniform; the per-lane part of every LDS-DMA address is the same lane * 16 bytes
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)((float __attribute__((address_space(3)))*)(region + piece * 256)));
        const int l16 = lane * 16;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(l16), "s"(sp) : "m0");
#else
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (piece * 64 + lane) * 4),
                                         (void __attribute__((address_space(3)))*)(region + piece * 256), 16, 0, 0);
#endif
    };
    using NA = std::integral_constant<int, S_GA * 3>;
    using NB = std::integral_constant<int, (S_G - S_GA) * 3>;

    auto lds_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    const bool relu = a.flags & SE_EPI_RELU;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bpack + 4 * h);

    __syncthreads();   // utab
    // chunk range of a unit: all of them, or (halves == 2: a launch with fewer than two tiles per CU, batch 1 at 64^3) the first / second
    // half - the first half's sums go to the output tensor with the bias, the second half's to `part`; k7_combine_kernel adds them
    const int c_mid = (chunks + 1) >> 1;
    auto unit_half = [&](int kk) { return (__builtin_amdgcn_readfirstlane(utab[kk].x) >> 30) & 1; };
    auto c_lo = [&](int hf) { return hf ? c_mid : 0; };
    auto c_hi = [&](int hf) { return (halves == 2 && !hf) ? c_mid : chunks; };
    int k = 0, half_k = unit_half(0), c = c_lo(half_k);
    fetch(0, c);
    commit();
    for (int i = tid; i < S_WA_F4; i += 512)     // region A of the first chunk
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(a.wpack_h + (size_t)c * S_W_FLOATS)[i];
    __syncthreads();

    auto read_ops = [&](K7SOps& r, auto g_tag) {
        constexpr int g = decltype(g_tag)::value;
        const f32x4* ap = reinterpret_cast<const f32x4*>(wl + g * 768 + lane * 12);
        const f32x4* bp = reinterpret_cast<const f32x4*>(vt + toff[g]);
        r.a0 = ap[0]; r.a1 = ap[1]; r.a2 = ap[2];
        r.b0 = bp[0]; r.b1 = bp[1]; r.b2 = bp[2];
    };

    f32x4 acc[S_XI];
    f32x4 y[S_TZ];
#pragma unroll
    for (int i = 0; i < S_TZ; ++i) y[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] int n_items = 0;
    for (;;) {
        ++n_items;
        const bool last_chunk = c == c_hi(half_k) - 1;
        const bool has_next = !(last_chunk && k == n - 1);
        const int k_next = (has_next && last_chunk) ? k + 1 : k;
        const int half_next = (has_next && last_chunk) ? unit_half(k_next) : half_k;
        const int c_next = has_next ? (last_chunk ? c_lo(half_next) : c + 1) : c;
        const float* w_cur_b = a.wpack_h + (size_t)c * S_W_FLOATS + S_GA * 768;     // region B of this chunk
        const float* w_next_a = a.wpack_h + (size_t)c_next * S_W_FLOATS;            // region A of the next item's chunk

        K7SOps cur, nxt;
#ifdef SE_STAMP67
        { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
        read_ops(cur, std::integral_constant<int, 0>{});
        nxt = cur;
        auto step = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            // first half: region B of this chunk (7 pieces per thread) and the next item's raw halo columns
            constexpr bool WR = !(SE_K67_EXP & 1), FR = !(SE_K67_EXP & 2);
            // first half: region B of this chunk by LDS-DMA (63 wave-instructions = 8 per wave, the 64th repeats the 63rd), then the
            // next item's raw halo columns; second half: region A of the next item's chunk (48 = 6 per wave).  All of them early in
            // their half: a wave stalled on a vector-memory issue is covered by its SIMD partner only while that one still has MFMAs
            // (measured: two LDS-DMAs per group = one per group; skipping the fetch in the waves without columns by a branch: +6 %;
            // hipcc sinks the halo loads towards the mid barrier - pinning them to groups 5-10 by sched_group_barrier: +0.5 %, all
            // twelve in group 5: +2.6 %)
            if constexpr (WR && g >= 1 && g <= 4) {
                wglds(w_cur_b, wl + S_GA * 768, NB{}, std::integral_constant<int, 2 * (g - 1)>{});
                wglds(w_cur_b, wl + S_GA * 768, NB{}, std::integral_constant<int, 2 * (g - 1) + 1>{});
            }
            if constexpr (FR && g == 5) {
                asm volatile("" ::: "memory");   // the halo loads stay behind the LDS-DMAs: the counted wait in front of the mid barrier relies on it
                fetch(k_next, c_next);
            }
            if constexpr (WR && g >= S_GA + 1 && g <= S_GA + 3) {
                wglds(w_next_a, wl, NA{}, std::integral_constant<int, 2 * (g - S_GA - 1)>{});
                wglds(w_next_a, wl, NA{}, std::integral_constant<int, 2 * (g - S_GA - 1) + 1>{});
            }
            constexpr bool pipelined = g + 1 < S_G && g + 1 != S_GA;     // the first read of region B waits for the mid barrier
            if constexpr (pipelined) read_ops(nxt, std::integral_constant<int, g + 1>{});
            const float av[12] = {cur.a0.x, cur.a0.y, cur.a0.z, cur.a0.w, cur.a1.x, cur.a1.y, cur.a1.z, cur.a1.w, cur.a2.x, cur.a2.y, cur.a2.z, cur.a2.w};
            const float bv[12] = {cur.b0.x, cur.b0.y, cur.b0.z, cur.b0.w, cur.b1.x, cur.b1.y, cur.b1.z, cur.b1.w, cur.b2.x, cur.b2.y, cur.b2.z, cur.b2.w};
            // group 0 starts the item's accumulators from the MFMA's zero C operand (no v_mov per item)
#pragma unroll
            for (int x = 0; x < S_XI; ++x)
                acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x], bv[x], g == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[x], 0, 0, 0);
            if constexpr (pipelined) {   // the next group's reads between this group's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            }
            if constexpr (g + 1 == S_GA) {   // everybody is past region A; region B of this chunk is complete
                T67(0);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // region B has landed; the 12 halo loads behind it may still fly
                if (!(SE_K67_EXP & 4)) lds_barrier();
                T67(1);
                read_ops(nxt, std::integral_constant<int, S_GA>{});
            }
            cur = nxt;
        };
        for_each_index(step, std::make_integer_sequence<int, S_G>{});
        T67(2);

        // A^T (6 x 12; xi 1..10 in +- pairs) after EVERY item, summed in the output domain: the rounding error of a Winograd-domain
        // accumulator is amplified by A^T (|A^T| <= 98), so the sums that live across chunks are the six outputs, not the twelve xi
        // (float32 model, error / std of the output: 1.2e-5 mean with one A^T per tile, 4.3e-6 per chunk; F(4,7) per chunk: 2.5e-6)
        if (!(SE_K67_EXP & 8) || last_chunk) {
            f32x4 s[5], d[5];
#pragma unroll
            for (int p = 0; p < 5; ++p) { s[p] = acc[2 * p + 1] + acc[2 * p + 2]; d[p] = acc[2 * p + 1] - acc[2 * p + 2]; }
#pragma unroll
            for (int i = 0; i < S_TZ; ++i) {
                f32x4 v = y[i];
                if (i == 0) v += acc[0];
                if (i == S_TZ - 1) v += acc[11];
#pragma unroll
                for (int p = 0; p < 5; ++p) {
                    const float cf = SE_W67_AT[i][2 * p + 1];
                    const f32x4 t = (i & 1) ? d[p] : s[p];
                    v.x = fmaf(cf, t.x, v.x); v.y = fmaf(cf, t.y, v.y); v.z = fmaf(cf, t.z, v.z); v.w = fmaf(cf, t.w, v.w);
                }
                y[i] = v;
            }
        }
        T67(3);
        // single transformed tile: every wave must be done reading it before the next item's columns are committed.  Barriers inside
        // the loop wait for this wave's LDS traffic only (a __syncthreads() would also wait for vmcnt(0))
        lds_barrier();
        T67(4);
        if (has_next && !(SE_K67_EXP & 16)) commit();
        if (last_chunk) {   // bias, ReLU, the only store of this tile
            const i32x4 e = utab[k];
            const int oz0 = e.y * S_TZ, oy = e.z * S_TY + wave, ox = e.w * S_TX + vl;
            float* ob = (halves == 2 && half_k) ? part : a.out;           // second half of the chunks: raw sums beside the output tensor
            float* op = ob + ((((long long)(e.x & 0x3fffffff) * dim + oz0) * dim + oy) * dim + ox) * 16 + 4 * h;
            const long long zstride = (long long)dim * dim * 16;
            const bool fin = halves != 2;
#pragma unroll
            for (int i = 0; i < S_TZ; ++i) {
                f32x4 v = y[i];
                if (fin || !half_k) v += bias;
                if (fin && relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (oz0 + i < dim) *reinterpret_cast<f32x4*>(op + i * zstride) = v;
                y[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        T67(5);
        if (!has_next) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // region A of the next chunk has landed
        lds_barrier();
        T67(6);
        k = k_next; c = c_next; half_k = half_next;
    }
#ifdef SE_STAMP67
    if (lane == 0 && dbg) {
        unsigned long long* o = dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 7; ++i) o[i] = st_sum[i];
        o[7] = n_items;
    }
#endif
}

// halves == 2: out = [relu](out + part), thread = 4 floats
__global__ __launch_bounds__(256) void k7_combine_kernel(float* __restrict__ out, const float* __restrict__ part, long long n4, int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<f32x4*>(out)[i] + reinterpret_cast<const f32x4*>(part)[i];
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<f32x4*>(out)[i] = v;
}

}  // namespace

// Returns 0 on launch, SE_TILED_NOT_TAKEN if the shape / unit table is not covered, else a hipError_t.  Preconditions (checked by the
// caller, se_conv3d_k7_wino_try): ksize 7, cout 16, no residual, channels-last output, a.wpack_h set.
int se_conv3d_k7_wino67_launch(const ConvArgs& a, int batch, int num_cus, hipStream_t s, unsigned long long* dbg) {
    constexpr int LDS_BYTES = 160 * 1024;
    constexpr int MAX_UNITS = (LDS_BYTES - S_LDS_FIXED) / 16;
    const int dim = a.dim;
    if (dim < 16 || (dim & 15)) return SE_TILED_NOT_TAKEN;
    // 32-bit buffer offsets inside one sample (channels-last) / one chunk volume (triplet-planar); bit 31 marks out-of-volume lanes
    if ((long long)dim * dim * dim * ((a.flags & SE_IN_PLANAR3) ? 12 : a.cin_pad * 4) >= (1LL << 31)) return SE_TILED_NOT_TAKEN;
    const int tx = dim / S_TX, ty = dim / S_TY, tz = (dim + S_TZ - 1) / S_TZ;
    const long long total_ll = (long long)batch * tx * ty * tz;
    if (total_ll > (1 << 30)) return SE_TILED_NOT_TAKEN;
#ifndef SE_K67_SPLIT
#define SE_K67_SPLIT 1
#endif
    // fewer than two tiles per CU (batch 1 at 64^3: 352 tiles = two rounds for 1.4 tiles' worth of work): units of half a tile's chunks
    // (704 units, at most 17 instead of 22 items per workgroup); needs the caller's workspace for the second halves' sums
    const long long out_elems = a.total_vox * 16;
    const bool split = SE_K67_SPLIT && total_ll < 2LL * num_cus && (a.cin + 2) / 3 >= 2 && a.ws && a.ws_elems >= out_elems;
    const int halves = split ? 2 : 1;
    const int total = (int)total_ll * halves;
    const int grid = total < num_cus ? total : num_cus;
    const int per = (total + grid - 1) / grid;
    if (per > MAX_UNITS) return SE_TILED_NOT_TAKEN;
    if (a.flags & SE_IN_PLANAR3) {
        SE_ENSURE_LDS(conv3d_k7_wino67_kernel<true>, LDS_BYTES);
        hipLaunchKernelGGL(conv3d_k7_wino67_kernel<true>, dim3((total + per - 1) / per), dim3(512), LDS_BYTES, s, a, tx, ty, tz, total, per, halves, a.ws, dbg);
    } else {
        SE_ENSURE_LDS(conv3d_k7_wino67_kernel<false>, LDS_BYTES);
        hipLaunchKernelGGL(conv3d_k7_wino67_kernel<false>, dim3((total + per - 1) / per), dim3(512), LDS_BYTES, s, a, tx, ty, tz, total, per, halves, a.ws, dbg);
    }
    SE_CHECK_LAUNCH();
    if (split) {
        const long long n4 = out_elems / 4;
        hipLaunchKernelGGL(k7_combine_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a.out, a.ws, n4, (a.flags & SE_EPI_RELU) ? 1 : 0);
        SE_CHECK_LAUNCH();
    }
    return 0;
}
