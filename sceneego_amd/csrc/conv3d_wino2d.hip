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
//                      V [y-tile][18 x records][xi][8 channels]     28 KB per group, B^T-transformed input
//     staging phase  the other group meanwhile: output transform + epilogue + stores of the tile it just finished (only after the
//                    last chunk), global loads + 2-D B^T transform + LDS commit of its next chunk, and its share of the weight stream
// and the two groups run half a step apart (as in conv3d_k3_wino43pp_kernel), so each SIMD always has one wave in its MFMA block.
// The weights of a chunk are re-streamed from L2 for every (unit, chunk) - 8 B/clk/CU - into ONE chunk buffer: its halves (xi_z < 3
// / >= 3) are consumed a quarter step apart, so the half a group has finished with is refilled while the other half is in use;
// two workgroup barriers per phase (start / middle) order this.
// LDS operand reads are conflict-free: W is lane-linear; V records are 216 floats apart (see v_rec_offset).
#include "common.h"

#include "conv_common.h"

#include <type_traits>
#include <utility>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int W2_HALF_FLOATS = 3 * 4 * 3 * 2 * 128;      // xi_z 0..2 (or 3..5): 9216 floats = 36,864 B
constexpr int W2_CHUNK_FLOATS = SE_WINO2D_CHUNK_FLOATS;  // 18,432 floats = 73,728 B
constexpr int W2_VREC = 216;                             // floats per x record: 24 xi x 8 channels + 24 pad (bank spread, see below)
constexpr int W2_VTILE = 18 * W2_VREC + 8;               // one y-tile: 18 x records (+ 8: see v_rec_offset)
constexpr int W2_VG_FLOATS = 2 * W2_VTILE;               // 7792 floats = 31,168 B per group
constexpr int W2_DUMMY_FLOATS = 512;                     // landing zone of the lanes without a y-transform output: 8 slots x 4 floats + 6 x 32 floats of xi_z offsets
constexpr int W2_LDS_BYTES = (W2_CHUNK_FLOATS + 2 * W2_VG_FLOATS + W2_DUMMY_FLOATS) * 4;   // 138,112 B
static_assert(W2_CHUNK_FLOATS == 2 * W2_HALF_FLOATS, "chunk = two halves");
static_assert(W2_VREC % 64 == 24 && (W2_CHUNK_FLOATS + 2 * W2_VG_FLOATS) % 32 == 0, "bank layout of the V tiles, see v_rec_offset");

template <typename F, int... S>
__device__ __forceinline__ void for_each_idx(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}

// LDS operand layouts are built for ds_read_b128 (4 LDS cycles per KiB; the two-address ds_read2_b64 hipcc forms from adjacent
// 8-byte reads takes 8): a lane's 16 bytes carry the channel pair of its k lane for TWO transform points xi_y = 2q, 2q+1.
//   W [xi_z][q][dx][ct][lane][e][c]        lane-linear: conflict-free
//   V [y-tile][x record][xi_z][q][k lane h][e][c], 216 floats per x record (= 24 mod 64): the 16 lanes of a ds_read_b128 pass
//     (k lane h, positions per MI355X_MICROARCH.md's b128 lane groups) fall on 64 different banks for all three x-shifted windows;
//     all 24 xi of a position sit within 768 B of one base address (immediate offsets, no address arithmetic in the MFMA stream).
//   The STORES of the V-tile transform are ds_write_b128.  They are the ONLY source of LDS bank conflicts in this kernel: the
//   development variant without them counts SQ_LDS_BANK_CONFLICT = 0, the ones without weight writes / without operand reads
//   count the full value (tools/pmc_lds_attr.sh, profiles/r03_lds_conflict_attribution.txt).  With the guide's bank model of
//   ds_write_b128 (8 consecutive lanes per pass, bank = dword mod 32: a lane's 16-byte slot is 6 sxx + sp + 6 sk (+ 4 for the
//   xi_y 2, 3 pair) mod 8 here) this layout - 216 floats per record, 3896 per y-tile, lanes without an output parked on a free
//   slot of their pass - should leave 96 extra cycles per step where the round-2 layout (200 / 3600, parked at lane * 16 B) had
//   792 (tools/lds_conflicts_w2d.py; measured then: 804).  Measured now: 590-710, i.e. -12..27 %, not -88 %: the model is not the
//   hardware's for 16-byte stores.  It does not matter for time (bit-identical results, same launch time with either layout): a
//   ds_write_b128 occupies its wave's LDS data path for 13 cycles, longer than the 8 + ~7 array cycles it needs.
__device__ __forceinline__ int v_rec_offset(int xx, int p) { return xx * W2_VREC + p * 4; }

// cycle stamps of the phase structure (diagnostic builds: build.sh --devtools -DSE_STAMP2D, tools/stamp_w2d.py)
#ifdef SE_STAMP2D
// 32-bit stamps (low word of s_memtime) and 32-bit sums: the phase structure lives in scalar registers and 64-bit stamps spill them
#define W2_T(var) { unsigned long long w2_t64; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w2_t64)::"memory"); var = (unsigned)w2_t64; __builtin_amdgcn_sched_barrier(0); }
unsigned long long* g_w2d_dbg = nullptr;
#else
#define W2_T(var)
#endif

struct Unit {
    int cb, b, z0, y0, x0;
};

// Rider schedule of the MFMA phase (see mfma_phase): which of the 18 groups of 8 MFMAs of group GG's phase carries
//   the loads of weight set k (3 x 16 B per thread)  /  its LDS writes, at least three groups (768 cycles) later, at most two sets
//   in registers at a time, all writes of group 0 in front of the mid-phase barrier (group 7), all of group 1 behind it;
//   the input-row load j of the next step (12 per phase), thinned out where weight sets are in flight.
constexpr int W2_WLOAD[2][3] = {{0, 2, 3}, {6, 8, 10}};
constexpr int W2_WWRITE[2][3] = {{3, 5, 6}, {9, 11, 13}};
// Loads return in order, so a weight load (L2 hit) issued behind an input-row load (HBM) of the same wave is delivered only after
// that row: group 0 issues its rows behind its last weight load; group 1, whose weight window lies in the second half of the
// phase, keeps its row loads at the end of its staging phase (-1: not carried).
// Group 0's rows go out two per group right behind its weight loads: the V-tile transform needs them as soon as the phase ends,
// and a row takes ~2 k cycles from HBM.
constexpr int W2_RAWG[2][12] = {{4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9}, {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}};
constexpr int rider_w_load(int gg, int g) {
    for (int k = 0; k < 3; ++k)
        if (W2_WLOAD[gg][k] == g) return k;
    return -1;
}
constexpr int rider_w_write(int gg, int g) {
    for (int k = 0; k < 3; ++k)
        if (W2_WWRITE[gg][k] == g) return k;
    return -1;
}
constexpr int rider_raw_count(int gg, int g) {
    int n = 0;
    for (int j = 0; j < 12; ++j) n += W2_RAWG[gg][j] == g;
    return n;
}

// EXP: timing experiments of development builds (0 = the real kernel; bit 0: compact input addresses, bit 1: no weight stream,
// bit 2: no epilogue memory traffic - all three give wrong results and exist only to attribute time)
// LAYOUT: bit 0 = input is octet-planar [B][cin/8][D][D][D][8] (SE_IN_OCTET), bit 1 = output is octet-planar (SE_OUT_OCTET),
// bit 2 = the skip tensor is octet-planar (SE_RES_OCTET), bit 3 = also write the 2x2x2 max-pool of the output (se_conv3d_pool_f32;
// instantiated for the layouts the V2V program pools: 3 and 7), bit 4 = the skip path is a 1x1x1 convolution over a 16-channel
// channels-last tensor computed in the epilogue (se_conv3d_skip16_f32; instantiated for layout 3), bit 5 = the output is QUAD-planar
// [B][cout/4][D][D][D][4] (SE_OUT_QUAD alone, instantiated for layout 32: channels-last input, no skip tensor - the first convolution
// of a block whose second one runs on the F(4,3) x F(4,3) kernel and whose input comes channels-last from a max-pool; round 5).
// In the octet-planar form an 8-channel chunk of a halo row is ONE contiguous run (18 positions x 32 B) instead of 18 pieces of
// 32 B at a 4*cin-byte stride: 4x fewer cache lines per load instruction.
template <int EXP, int LAYOUT>
__global__ __launch_bounds__(512) void conv3d_k3_wino2d_kernel(ConvArgs a, const float* __restrict__ wg, int tiles_x, int tiles_y,
                                                               int tiles_z, int total_tiles, int n_units, int units_per_wg, unsigned long long* dbg) {
    constexpr int exp = EXP;
    constexpr bool in_oct = LAYOUT & 1, out_oct = LAYOUT & 2, res_oct = LAYOUT & 4, pool = LAYOUT & 8, skc = LAYOUT & 16, out_quad = LAYOUT & 32;
    unsigned t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, t8 = 0, st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)t5; (void)t6; (void)t7; (void)t8; (void)st; (void)dbg;
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
    const int u_begin = se_xcd_walk_index((int)blockIdx.x, (int)gridDim.x) * units_per_wg;      // XCD-aware: conv_common.h
    const int u_end = min(u_begin + units_per_wg, n_units);
    if (u_begin >= u_end) return;
    const int n_steps = (u_end - u_begin) * chunks;
    float* vt = lds + W2_CHUNK_FLOATS + G * W2_VG_FLOATS;

    // ---- MFMA operand addresses ----
    const float* a_h0 = wl + ct * 256 + lane * 4;
    const float* a_h1 = a_h0 + W2_HALF_FLOATS;
    const float* b_dx[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) b_dx[dx] = vt + jt * W2_VTILE + v_rec_offset(px + dx, h);

    // ---- staging role inside the group ----
    // A staging TASK is (x record, channel pair): 18 x 4 = 72 per group; it covers the group's 6 halo rows (y-tile 0 uses rows 0..3,
    // y-tile 1 rows 2..5 - the two shared rows are loaded once).  Three adjacent lanes of a 16-lane row split a task by ROW PAIR
    // (k = 0, 1, 2 -> rows 2k, 2k+1): 12 loads per lane and all four waves of the group take part, which matters because every
    // vector-memory instruction costs the issuing wave ~140 cycles next to a streaming MFMA partner.  The z transform is
    // lane-local; the y transform needs one value from each neighbour lane (DPP row shifts).
    const int tg = tid & 255;
    const int i16 = lane & 15;
    const int sk = i16 % 3;                                    // row pair of this lane
    const int stask = (wq * 4 + (lane >> 4)) * 5 + i16 / 3;    // 5 tasks per 16-lane row, lane 15 idle
    const bool s_on = i16 < 15 && stask < 72;
    const int sxx = s_on ? stask >> 2 : 0, sp = stask & 3;
    // this lane's y-transform outputs: P, Q -> (y-tile sk, xi_y 0, 1) when sk < 2;  R, S -> (y-tile sk - 1, xi_y 2, 3) when sk > 0
    // (lanes without that output write into a per-lane slot of a dummy area instead of being masked off: no EXEC toggling in the
    // transform code, and distinct banks inside every store instruction)
    const int pq_off = W2_CHUNK_FLOATS + G * W2_VG_FLOATS + sk * W2_VTILE + v_rec_offset(sxx, sp);               // floats from the LDS base
    const int rs_off = W2_CHUNK_FLOATS + G * W2_VG_FLOATS + (sk - 1) * W2_VTILE + v_rec_offset(sxx, sp) + 16;
    const bool has_pq = s_on && sk < 2, has_rs = s_on && sk > 0;
    // lanes without an output park their store in a dummy area - on a 16-byte slot no other lane of their 8-lane LDS pass uses
    // (a ds_write_b128 pass covers lanes 8k..8k+7; slot = bits 2..4 of the dword address)
    auto dummy_slot = [&](bool has, int off) {
        const int mine = has ? (off >> 2) & 7 : -1;
        unsigned used = 0;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = __shfl(mine, (lane & ~7) + j);
            if (o >= 0) used |= 1u << o;
            else if (j < (lane & 7)) ++rank;
        }
        unsigned free_slots = ~used & 0xffu;
        for (int r = 0; r < rank && (free_slots & (free_slots - 1)); ++r) free_slots &= free_slots - 1;
        return free_slots ? __builtin_ctz(free_slots) : (lane & 7);
    };
    float* v_dummy = lds + W2_CHUNK_FLOATS + 2 * W2_VG_FLOATS;              // = 0 mod 32 floats: slot k of the area is slot k of the LDS
    float* v_pq = has_pq ? lds + pq_off : v_dummy + dummy_slot(has_pq, pq_off) * 4;
    float* v_rs = has_rs ? lds + rs_off : v_dummy + dummy_slot(has_rs, rs_off) * 4;

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
    // (unit, chunk) one step after (u, c); behind the last step of this workgroup the walk stays where it is, so the staging
    // code needs no "is there a next step" branches (it then reloads data nobody reads)
    auto step_after = [&](Unit& u, int& c, int& idx) {
        if (idx + 1 >= n_steps) return;
        ++idx;
        if (++c == chunks) { c = 0; u = advance(u); }
    };

    // Input loads go through a raw buffer descriptor of ONE sample (base = sample b, num_records = bytes of a sample): the
    // per-lane part of the address in a 32-bit voffset, the uniform z-slab part in the (unsigned) scalar offset; halo
    // positions outside the volume get voffset bit 31 set, which is out of range and reads zero - no branches.
    const unsigned in_bytes = (unsigned)dim * dim * dim * cin * 4u;
    constexpr unsigned OOB = 0x80000000u;
    // Every vector instruction of a staging wave costs ~25 cycles next to the partner wave's float32 MFMA stream (stamps,
    // DESIGN.md section 4), so the per-chunk part of the address lives in SCALAR registers: the per-lane offsets of the two
    // rows are computed once per unit (fetch_setup), the chunk and the z slab go into the scalar offset, and a slab outside
    // the volume is read through a descriptor of zero records (all lanes out of range) instead of masking lanes.
    unsigned fvoff[2] = {OOB, OOB};
    auto fetch_setup = [&](const Unit& u) {
        const int gx = u.x0 - 1 + sxx;
        const bool okx = s_on && (unsigned)gx < (unsigned)dim;
        const int gy0 = u.y0 + G * 4 - 1 + 2 * sk;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int gy = gy0 + r;
            fvoff[r] = (okx && (unsigned)gy < (unsigned)dim) ? (unsigned)(((gy * dim + gx) * ((in_oct || (exp & 1)) ? 8 : cin) + sp * 2) * 4) : OOB;
        }
    };
    auto fetch_one = [&](f32x2 (&raw)[6][2], const Unit& u, int chunk, auto s_tag, auto r_tag) {
        constexpr int s = decltype(s_tag)::value, r = decltype(r_tag)::value;
        const float* p0 = a.in + (long long)u.b * dim * dim * dim * cin;
        const int gz = u.z0 - 1 + s;
        const bool okz = (unsigned)gz < (unsigned)dim;       // uniform
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p0), 0, okz ? (int)in_bytes : 0, 0x00020000);
        // (a slab outside the volume is read through the zero-record descriptor: its scalar offset is never used for an access)
        const int soff = in_oct ? (chunk * dim + gz) * dim * dim * 32 : gz * dim * dim * ((exp & 1) ? 8 : cin) * 4 + ((exp & 1) ? 0 : chunk * 32);
        if constexpr ((exp & 0x10000) != 0) {
            raw[s][r] = (f32x2){__builtin_bit_cast(float, fvoff[r]), __builtin_bit_cast(float, soff)};
        } else {
            raw[s][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)fvoff[r], soff, 0));
        }
    };
    auto fetch = [&](f32x2 (&raw)[6][2], const Unit& u, int chunk) {       // all twelve at once (prologue)
        for_each_idx([&](auto j_tag) {
            constexpr int jj = decltype(j_tag)::value;
            fetch_one(raw, u, chunk, std::integral_constant<int, jj / 2>{}, std::integral_constant<int, jj % 2>{});
        }, std::make_integer_sequence<int, 12>{});
    };
    // B^T along z (F(4,3), points 0, +-1, +-2, inf) on the lane's two rows, then along y (F(2,3)) with the neighbour lanes' rows;
    // commit to the group's V buffer.
    // Scalar fmaf / adds on purpose (this file is built with -fno-slp-vectorize): packed float32 VALU beside the partner wave's MFMA
    // stream is an anti-lever (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
    auto commit = [&](const f32x2 (&raw)[6][2]) {
        float t[2][6][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float d0 = raw[0][r][c], d1 = raw[1][r][c], d2 = raw[2][r][c], d3 = raw[3][r][c], d4 = raw[4][r][c], d5 = raw[5][r][c];
                t[c][0][r] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                t[c][5][r] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
                const float e1 = fmaf(-4.f, d2, d4), o1 = fmaf(-4.f, d1, d3);
                t[c][1][r] = e1 + o1;
                t[c][2][r] = e1 - o1;
                const float e2 = d4 - d2, o2 = d3 - d1;
                t[c][3][r] = fmaf(2.f, o2, e2);
                t[c][4][r] = fmaf(-2.f, o2, e2);
            }
        }
#pragma unroll
        for (int z = 0; z < 6; ++z) {
            f32x4 pq, rs;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float ta = t[c][z][0], tb = t[c][z][1];
                // ra: first row of the lane to the right (row_shl:1), lb: second row of the lane to the left (row_shr:1)
                const float ra = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ta), 0x101, 0xf, 0xf, true));
                const float lb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tb), 0x111, 0xf, 0xf, true));
                pq[c] = ta - ra;          // xi_y = 0
                pq[2 + c] = tb + ra;      // xi_y = 1
                rs[c] = ta - lb;          // xi_y = 2
                rs[2 + c] = lb - tb;      // xi_y = 3
            }
            if constexpr ((exp & 0x8000) != 0) {
                asm volatile("" ::"v"(pq), "v"(rs));
            } else {
                *reinterpret_cast<f32x4*>(v_pq + z * 32) = pq;
                *reinterpret_cast<f32x4*>(v_rs + z * 32) = rs;
            }
        }
    };

    // weight stream: one half chunk (36,864 B = 9 x 16 B per thread of a group), global (L2) -> registers -> LDS, in pieces of
    // 3 x 16 B per thread.  Plain loads, not LDS-DMA: an L2-hit load returns in a few hundred cycles, a DMA piece lands ~1 us after
    // issue (MI355X_MICROARCH.md, ldsdma-fill) and the refill window of a half is half a phase.
    auto w_fetch3 = [&](f32x4 (&wreg)[9], const Unit& u, int chunk, int half, auto k0_tag) {
        constexpr int k0 = decltype(k0_tag)::value;
        if (exp & 2) return;
        // uniform base in the descriptor, lane offset tg * 16 B, piece offset in the scalar/immediate offset: no address VALU
        const float* src = wg + ((size_t)u.cb * chunks + chunk) * W2_CHUNK_FLOATS + half * W2_HALF_FLOATS;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, W2_HALF_FLOATS * 4, 0x00020000);
#pragma unroll
        for (int k = k0; k < k0 + 3; ++k) wreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tg * 16, k * 4096, 0));
    };
    auto w_commit3 = [&](const f32x4 (&wreg)[9], int half, auto k0_tag) {
        constexpr int k0 = decltype(k0_tag)::value;
        f32x4* dst = reinterpret_cast<f32x4*>(wl + half * W2_HALF_FLOATS);
        if (exp & 2) return;
        if constexpr ((exp & 0x4000) != 0) {
#pragma unroll
            for (int k = k0; k < k0 + 3; ++k) asm volatile("" ::"v"(wreg[k]));
            return;
        }
#pragma unroll
        for (int k = k0; k < k0 + 3; ++k) dst[tg + k * 256] = wreg[k];
    };

    f32x4 acc[24];
    f32x2 raw[6][2];      // input rows of the next step (12 loads per lane)
    f32x4 wreg[9];        // weight stream pieces in flight
    // ReLU as max(v, 0) / no ReLU as max(v, -inf): one v_max_f32 per element, no per-element select on a run-time flag
    const float relu_lo = (a.flags & SE_EPI_RELU) ? 0.f : -__builtin_inff();
    const bool use_res = (a.flags & SE_EPI_RES_PRE_RELU) && a.res;

    // output transform (A^T along y, then along z), bias, residual, ReLU, 8 x 16-byte channels-last stores
    // skip tensor + bias of a finished tile, issued in front of the staging phase's mid barrier
    // fused 1x1x1 skip convolution (skc): `a.res` is its 16-channel channels-last input x, `a.skip_w` its folded weights [cout][16].
    // The epilogue adds W_skip x to the outputs with 4 MFMAs per output vector: A = weights (row = cout, k lane h carries input
    // channels 4h..4h+3, one per k step), B = x (column = voxel, same channel split): both operands are one 16-byte load per lane.
    f32x4 wsk = {0.f, 0.f, 0.f, 0.f};
    // The bias of a finished tile is requested here, with its skip tensor, and added in the epilogue (32 adds per tile).  Rounds 2-3
    // carried it through the accumulators (the point (xi_z, xi_y) = (1, 1) has the coefficient 1 in every row of both output
    // transforms) in four registers held across the loop; hipcc spilled exactly those in five of the eleven layouts, and a scratch
    // reload is a vmcnt(0) behind everything in flight: -5 % at 16->32 @64^3, -2.7 % at 32->32 @64^3 for the <0,2> layout
    // (profiles/r03_wino2d_and_small_level_experiments.txt section 9); no instantiation uses scratch now.
    auto epi_prefetch = [&](const Unit& u, f32x4 (&resv)[2][4], f32x4& bias_e) {
        bias_e = *reinterpret_cast<const f32x4*>(a.bpack + u.cb * 32 + ct * 16 + 4 * h);
        if constexpr (skc) {
            wsk = *reinterpret_cast<const f32x4*>(a.skip_w + (u.cb * 32 + ct * 16 + px) * 16 + 4 * h);
            const float* xb = a.res + (((((long long)u.b * dim + u.z0) * dim + u.y0 + G * 4 + jt * 2) * dim + u.x0) * 16);
            const int xvoff = px * 16 + 4 * h;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int z = 0; z < 4; ++z) resv[r][z] = *reinterpret_cast<const f32x4*>(xb + z * dim * dim * 16 + r * dim * 16 + xvoff);
            return;
        }
        if (!use_res || (exp & 4) || (exp & 0x20000)) {          // no skip tensor: add zeros (the epilogue has no per-element selects)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int z = 0; z < 4; ++z) resv[r][z] = (f32x4){0.f, 0.f, 0.f, 0.f};
            return;
        }
        const long long s00 = (((((long long)u.b * dim + u.z0) * dim + u.y0 + G * 4 + jt * 2) * dim + u.x0) * a.cout + u.cb * 32 + ct * 16);
        const long long s00o = (((((long long)u.b * (a.cout >> 3) + u.cb * 4 + ct * 2) * dim + u.z0) * dim + u.y0 + G * 4 + jt * 2) * dim + u.x0) * 8;
        const float* rb = a.res + (res_oct ? s00o : s00);
        const int rvoff = res_oct ? (h >> 1) * dim * dim * dim * 8 + px * 8 + (h & 1) * 4 : px * a.cout + 4 * h;
        const int rystride = res_oct ? dim * 8 : dim * a.cout, rzstride = res_oct ? dim * dim * 8 : dim * dim * a.cout;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int z = 0; z < 4; ++z) resv[r][z] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rb + z * rzstride + r * rystride + rvoff));   // read once
    };
    auto epilogue = [&](const Unit& u, const f32x4 (&resv_all)[2][4], const f32x4& bias_e) {
        // uniform 64-bit base of the wave's first output row + a 32-bit per-lane offset (global_* saddr form); raw buffer
        // STORES with a scalar offset dropped data here, so stores and skip loads use plain global accesses
        const long long s00 = (((((long long)u.b * dim + u.z0) * dim + u.y0 + G * 4 + jt * 2) * dim + u.x0) * a.cout + u.cb * 32 + ct * 16);
        const int voff_cl = px * a.cout + 4 * h;
        const int ystride_cl = dim * a.cout, zstride_cl = dim * dim * a.cout;
        // octet-planar output: octet (cb*4 + ct*2 + h/2), 16 bytes at (h & 1) * 4 inside the 8-channel record of voxel (z, y, x)
        const long long s00o = (((((long long)u.b * (a.cout >> 3) + u.cb * 4 + ct * 2) * dim + u.z0) * dim + u.y0 + G * 4 + jt * 2) * dim + u.x0) * 8;
        // quad-planar output: plane (cb*8 + ct*4 + h), the lane's 4 couts are the 16-byte record of voxel (z, y, x)
        const long long s00q = (((((long long)u.b * (a.cout >> 2) + u.cb * 8 + ct * 4) * dim + u.z0) * dim + u.y0 + G * 4 + jt * 2) * dim + u.x0) * 4;
        const int voff = out_quad ? h * dim * dim * dim * 4 + px * 4 : out_oct ? (h >> 1) * dim * dim * dim * 8 + px * 8 + (h & 1) * 4 : voff_cl;   // (a.cout == channels of out and of the skip tensor)
        const int ystride = out_quad ? dim * 4 : out_oct ? dim * 8 : ystride_cl, zstride = out_quad ? dim * dim * 4 : out_oct ? dim * dim * 8 : zstride_cl;
        float* ob = a.out + (out_quad ? s00q : out_oct ? s00o : s00);
        // all eight output vectors first, then the skip tensor (prefetched behind the V-tile transform; its latency also
        // overlaps this arithmetic), ReLU and the stores
        f32x4 out[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if constexpr ((exp & 0x2000) != 0) {       // experiment: packed float32 arithmetic (v_pk_add_f32 / v_pk_fma_f32)
                f32x4 m[6];
#pragma unroll
                for (int z = 0; z < 6; ++z)
                    m[z] = r == 0 ? (acc[z * 4 + 0] + acc[z * 4 + 1]) + acc[z * 4 + 2] : (acc[z * 4 + 1] - acc[z * 4 + 2]) - acc[z * 4 + 3];
                const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
                out[r][0] = (m[0] + s12) + s34;
                out[r][1] = 2.f * d34 + d12;
                out[r][2] = 4.f * s34 + s12;
                out[r][3] = (8.f * d34 + d12) + m[5];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {      // scalar on purpose, see commit()
                    float m[6];
#pragma unroll
                    for (int z = 0; z < 6; ++z)
                        m[z] = r == 0 ? (acc[z * 4 + 0][c] + acc[z * 4 + 1][c]) + acc[z * 4 + 2][c] : (acc[z * 4 + 1][c] - acc[z * 4 + 2][c]) - acc[z * 4 + 3][c];
                    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
                    out[r][0][c] = (m[0] + s12) + s34;
                    out[r][1][c] = fmaf(2.f, d34, d12);
                    out[r][2][c] = fmaf(4.f, s34, s12);
                    out[r][3][c] = fmaf(8.f, d34, d12) + m[5];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int z = 0; z < 4; ++z)
#pragma unroll
                for (int c = 0; c < 4; ++c) out[r][z][c] += bias_e[c];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (skc) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)          // k step outer: the eight accumulators alternate (no dependent back-to-back MFMAs)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int z = 0; z < 4; ++z)
                        out[r][z] = __builtin_amdgcn_mfma_f32_16x16x4f32(wsk[ks], resv_all[r][z][ks], out[r][z], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                f32x4 v = out[r][z];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(skc ? v[c] : v[c] + resv_all[r][z][c], relu_lo);
                if (!(exp & (4 | 0x40000)) || v.x == 12345.f) *reinterpret_cast<f32x4*>(ob + z * zstride + r * ystride + voff) = v;
                out[r][z] = v;
            }
        }
        // fused 2x2x2 max-pool (se_conv3d_pool_f32): the wave's tile is 4 (z) x 2 (y) x 16 (x) outputs = 2 x 1 x 8 pooled voxels;
        // z and y pairs sit in this lane's registers, the x neighbour in the adjacent lane (quad_perm swap); even-x lanes store
        // 16 bytes of the channels-last pooled tensor [B][D/2][D/2][D/2][cout]
        if constexpr (pool) {
            const int hd = dim >> 1;
            float* pb = a.pool_out + ((((long long)u.b * hd + (u.z0 >> 1)) * hd + ((u.y0 + G * 4 + jt * 2) >> 1)) * hd + (u.x0 >> 1) + (px >> 1)) * a.cout
                        + u.cb * 32 + ct * 16 + 4 * h;
#pragma unroll
            for (int pz = 0; pz < 2; ++pz) {
                f32x4 m;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float m4 = fmaxf(fmaxf(out[0][2 * pz][c], out[0][2 * pz + 1][c]), fmaxf(out[1][2 * pz][c], out[1][2 * pz + 1][c]));
                    const float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m4), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                    m[c] = fmaxf(m4, nb);
                }
                if (!(px & 1)) *reinterpret_cast<f32x4*>(pb + (long long)pz * hd * hd * a.cout) = m;
            }
        }
    };

    // Workgroup barrier that waits for this wave's LDS traffic only (lgkmcnt): global loads stay in flight across it (a
    // __syncthreads() also waits for vmcnt(0)).  The "memory" clobber keeps the compiler from moving LDS / global accesses across it.
    // sched_barrier on both sides: register-only arithmetic must not drift across a phase boundary either (hipcc moved half of the
    // V-tile transform up into the MFMA stream, where it costs the MFMA wave more than it saves the staging wave: 9 % per launch).
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // Experiment (exp & 0x100000): no workgroup barriers inside the loop.  Each group synchronises its own four waves with a
    // monotonic arrival counter in LDS, and the only two cross-group dependencies of the weight halves are waits on the OTHER group's
    // counter: group 0 may start the first MFMA half of step s+1 when group 1 has finished the MFMA phase of step s (its barrier
    // number 4s+2: half 1 of step s read, half 0 of step s+1 written), group 1 may enter the second MFMA half of step s when group 0
    // has passed the mid barrier of its MFMA phase of step s (number 4s+1: half 1 written, half 0 read).  A group that carries an
    // epilogue then no longer holds the other one at the next barrier.
    constexpr bool flagsync = (exp & 0x100000) != 0;
    unsigned* sync_cnt = reinterpret_cast<unsigned*>(lds + W2_CHUNK_FLOATS + 2 * W2_VG_FLOATS + 480);    // [0]: group 0, [1]: group 1
    int my_barriers = 0;
    auto wait_ge = [&](int which, int target) {
        while (true) {
            const unsigned v = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(sync_cnt + which));
            if ((int)v >= target) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto group_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(sync_cnt + G, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ++my_barriers;
        wait_ge(G, 4 * my_barriers);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_barrier = [&]() {
        if constexpr (flagsync) group_barrier();
        else barrier();
    };

    // State of the walk: (ucur, ccur) = the step this group computes next, (unx, cnx) = the step after it.
    Unit ucur = decode(u_begin);
    int ccur = 0, icur = 0;
    Unit unx = ucur;
    int cnx = 0, inx = 0;
    int step_index = 0;

    // ---- MFMA phase of group GG: 18 groups (xi_z, dx) of 4 xi_y x 2 k steps; the workgroup's mid-phase barrier sits in front of
    // the first access to the second weight half.
    // ALL global loads and the weight stream's LDS writes ride inside this instruction stream, a few per group of 8 MFMAs: beside
    // an MFMA they are nearly free for the issuing wave, while every instruction of a staging wave costs it 10-25 cycles next to
    // the partner wave's MFMA stream (stamps) - the staging phase keeps only the transform work.
    //   * the 12 input-row loads of this group's NEXT step (unx, cnx), one per group of MFMAs;
    //   * the weight stream.  Half 0 (xi_z < 3) is read in the first half of a phase, half 1 in the second, and group 1 runs one
    //     phase behind group 0, so (half-phases numbered from group 0's step s: 4s, 4s+1 MFMA, 4s+2, 4s+3 staging)
    //       half 1 of step s   is read at 4s+1 (group 0) and 4s+3 (group 1) -> written at 4s   = group 0's first  MFMA half,
    //       half 0 of step s+1 is read at 4s+4 (group 0) and 4s+6 (group 1) -> written at 4s+3 = group 1's second MFMA half,
    //     each by the MFMA waves of that group themselves, in three sets of 3 x 16 B per thread (schedule: W2_WLOAD / W2_WWRITE).
    auto mfma_phase = [&](auto gg_tag) {
        constexpr int GG = decltype(gg_tag)::value;
        constexpr int WH = GG == 0 ? 1 : 0;      // weight half this group's MFMA waves refill
        // operands of group g live in buffer g % 3 and are fetched two groups (16 MFMAs = 512 cycles) ahead of their use
        f32x4 oa[3][2], ov[3][2];
        auto load_group = [&](auto g_tag) {
            constexpr int g2 = decltype(g_tag)::value;
            constexpr int xz2 = g2 / 3, dx2 = g2 % 3, b = g2 % 3;
            const float* ab = (xz2 < 3) ? a_h0 : a_h1;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                oa[b][q] = *reinterpret_cast<const f32x4*>(ab + (((xz2 % 3) * 2 + q) * 3 + dx2) * 512);
                ov[b][q] = *reinterpret_cast<const f32x4*>(b_dx[dx2] + (xz2 * 2 + q) * 16);
            }
        };
        if constexpr (!(exp & 0x400)) {
            load_group(std::integral_constant<int, 0>{});
            load_group(std::integral_constant<int, 1>{});
            // pin these eight reads in FRONT of the pipeline below.  Without this group the scheduler fills the "one DS read per
            // pair of MFMAs" slots of groups 0, 1, ... with exactly these reads first, every later read slides two groups down
            // with them, and each operand read ends up right in front of its MFMAs behind an s_waitcnt lgkmcnt(0) (round-3
            // disassembly: the first 56 MFMAs of every phase waited for a just-issued ds_read_b128 every 4 MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        }
        auto group = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            constexpr int xz = g / 3, b = g % 3;
            if constexpr (g == 7) {
                // the prefetch below (group 9) is the first read of weight half 1; all reads of half 0 have been issued and are
                // drained by the barrier's lgkmcnt(0)
                W2_T(t1)
                phase_barrier();
                if constexpr (flagsync && GG == 1) wait_ge(0, 4 * (4 * step_index + 1));     // group 0 past the mid barrier of this step
                W2_T(t2)
            }
            // riders of this group: LDS writes of a weight set first (frees its registers), then loads
            constexpr int kw = rider_w_write(GG, g), kl = rider_w_load(GG, g), nraw = rider_raw_count(GG, g);
            if constexpr (kw >= 0) w_commit3(wreg, WH, std::integral_constant<int, 3 * (kw < 0 ? 0 : kw)>{});
            if constexpr (kl >= 0) w_fetch3(wreg, GG == 0 ? ucur : unx, GG == 0 ? ccur : cnx, WH, std::integral_constant<int, 3 * (kl < 0 ? 0 : kl)>{});
            if constexpr (nraw > 0)
                for_each_idx([&](auto j_tag) {
                    constexpr int jj = decltype(j_tag)::value;
                    if constexpr (W2_RAWG[GG][jj] == g) fetch_one(raw, unx, cnx, std::integral_constant<int, jj / 2>{}, std::integral_constant<int, jj % 2>{});
                }, std::make_integer_sequence<int, 12>{});
            if constexpr ((exp & 0x400) != 0) return;      // attribution: no MFMA stream at all
            if constexpr (g + 2 < 18) load_group(std::integral_constant<int, g + 2>{});
            // xi_y = 2q + e: operand components (x, y) = channels of e = 0, (z, w) = channels of e = 1; first channel of all four
            // xi_y, then the second (dependent MFMAs on one accumulator stay 4 apart)
            acc[xz * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][0].x, ov[b][0].x, acc[xz * 4 + 0], 0, 0, 0);
            acc[xz * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][0].z, ov[b][0].z, acc[xz * 4 + 1], 0, 0, 0);
            acc[xz * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][1].x, ov[b][1].x, acc[xz * 4 + 2], 0, 0, 0);
            acc[xz * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][1].z, ov[b][1].z, acc[xz * 4 + 3], 0, 0, 0);
            acc[xz * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][0].y, ov[b][0].y, acc[xz * 4 + 0], 0, 0, 0);
            acc[xz * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][0].w, ov[b][0].w, acc[xz * 4 + 1], 0, 0, 0);
            acc[xz * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][1].y, ov[b][1].y, acc[xz * 4 + 2], 0, 0, 0);
            acc[xz * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][1].w, ov[b][1].w, acc[xz * 4 + 3], 0, 0, 0);
            // issue order inside the group: per pair of MFMAs one operand read, then one of the riders (global load / LDS write)
            constexpr int n_vm = ((exp & 2) ? 0 : (kl >= 0 ? 3 : 0)) + ((exp & 0x10000) ? 0 : nraw);
            constexpr int n_dw = (kw >= 0 && !(exp & 2) && !(exp & 0x4000)) ? 3 : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (g + 2 < 18) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (e < n_vm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (e < n_dw) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        };
        for_each_idx(group, std::make_integer_sequence<int, 18>{});
    };

#ifdef SE_DEVTOOLS
#include "devtools/wino2d_lockstep.inc"      // experiment (exp & 0x200000): both waves of a SIMD in the same phase, no ping-pong
#endif
    // ---- prologue: each group's V tile of step 0; weight half 0 of step 0 (by group 1's threads - half 1 of step 0 is written by
    // group 0 inside its first MFMA phase) ----
    fetch_setup(ucur);
    fetch(raw, ucur, 0);
    if (G == 1) {
        w_fetch3(wreg, ucur, 0, 0, std::integral_constant<int, 0>{});
        w_fetch3(wreg, ucur, 0, 0, std::integral_constant<int, 3>{});
        w_fetch3(wreg, ucur, 0, 0, std::integral_constant<int, 6>{});
    }
    commit(raw);
    if (G == 1) {
        w_commit3(wreg, 0, std::integral_constant<int, 0>{});
        w_commit3(wreg, 0, std::integral_constant<int, 3>{});
        w_commit3(wreg, 0, std::integral_constant<int, 6>{});
    }
    step_after(unx, cnx, inx);
    if (cnx == 0) fetch_setup(unx);
    if (G == 1) fetch(raw, unx, cnx);
    if constexpr ((exp & 0x300) == 0x200) __builtin_amdgcn_s_setprio(2);
    if (flagsync && tid < 2) sync_cnt[tid] = 0u;
    barrier();

    // The main loop exists once per group (its MFMA phase differs): one uniform branch in front of the loops instead of one inside
    // every iteration (a join inside the loop made the register allocator spill the accumulators).
    auto run = [&](auto gg_tag) {
    constexpr int GG = decltype(gg_tag)::value;
    if constexpr (GG == 1 && !flagsync) {          // group 1 runs one phase behind group 0
        barrier();
        barrier();
    }
    for (int i = 0; i < n_steps; ++i) {
        step_index = i;
        if constexpr (flagsync && GG == 0) {
            if (i > 0) wait_ge(1, 4 * (4 * (i - 1) + 2));         // group 1 finished the MFMA phase of step i-1
        }
        // ------------------------------ MFMA phase of step i = (ucur, ccur) ------------------------------
        W2_T(t0)
        if (ccur == 0) {
            // (the bias is added in the epilogue, see epi_prefetch)
#pragma unroll
            for (int e = 0; e < 24; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr ((exp & 0x300) == 0) __builtin_amdgcn_s_setprio(3);          // production
        if constexpr ((exp & 0x300) == 0x200) __builtin_amdgcn_s_setprio(0);      // experiment: staging wave above the MFMA wave
        mfma_phase(gg_tag);
        if constexpr ((exp & 0x300) == 0) __builtin_amdgcn_s_setprio(0);
        if constexpr ((exp & 0x300) == 0x200) __builtin_amdgcn_s_setprio(2);
        W2_T(t3)
        phase_barrier();                                          // end of the MFMA phase
        W2_T(t4)
        // ------------------------------ staging phase ------------------------------
        // `raw` holds the input rows of step i+1, loaded during the MFMA phase: first half = their transform into the group's V
        // tile, second half = the epilogue of a finished tile and the unit walk.
        const bool epi = ccur == chunks - 1;
        f32x4 resv[2][4];
        f32x4 bias_e = {0.f, 0.f, 0.f, 0.f};
        commit(raw);
        // skip tensor of a finished tile: in flight across the barrier and the output transform (issued in front of the V-tile
        // transform it measured slower, 0.428 vs 0.415 ms: that transform then waits behind these loads for its input rows)
        if (epi) epi_prefetch(ucur, resv, bias_e);
        W2_T(t5)
        phase_barrier();                                          // mid-phase barrier
        W2_T(t6)
        if (epi) epilogue(ucur, resv, bias_e);
#ifdef SE_STAMP2D
        unsigned tb = 0;
        W2_T(tb)
        st[13] += tb - t6;
#endif
        ucur = unx; ccur = cnx; icur = inx;
        step_after(unx, cnx, inx);
        if (cnx == 0) fetch_setup(unx);                           // new unit (or, behind the last step, the same one again)
        if constexpr (GG == 1) fetch(raw, unx, cnx);              // group 1: input rows of step i+2 (group 0: inside its MFMA phase)
        W2_T(t7)
        phase_barrier();                                          // end of the staging phase
        W2_T(t8)
#ifdef SE_STAMP2D
        st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2; st[3] += t4 - t3;
        st[4] += t5 - t4; st[5] += t6 - t5; st[6] += t7 - t6; st[7] += t8 - t7;
        st[8] += 1; st[9] += 1;
#endif
    }
    if constexpr (GG == 0 && !flagsync) {          // group 0 idles through group 1's last two phases
        barrier();
        barrier();
    }
    };
    if (G == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
    (void)icur;
#ifdef SE_STAMP2D
    if (lane == 0 && dbg) {
        unsigned long long* o = dbg + ((size_t)blockIdx.x * 8 + wave) * 16;
        for (int k = 0; k < 16; ++k) o[k] = st[k];
    }
#endif
}

}  // namespace

// Section G of the packed 3x3x3 weights (appended by se_conv3d_pack_f32): per (32-cout block cb, 8-channel chunk)
//   [xi_z 6][q 2][dx 3][ct 2][lane 64][e 2][c 2] = U[xi_z][xi_y = 2q + e][dx] of cout cb*32 + ct*16 + (lane & 15),
//   cin chunk*8 + 2*(lane >> 4) + c;  U = (G43 (x) G23) g over (dz, dy), times the folded BatchNorm scale.
__global__ void pack_k3_wino2d_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                      float eps, float* __restrict__ out, int cout, int cin, int cin_pad, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int e = (int)(t & 1);                 // channel of the k lane's pair
    const int exi = (int)((t >> 1) & 1);        // xi_y inside the pair
    const int lane = (int)((t >> 2) & 63);
    long long r = t >> 8;
    const int ct = (int)(r % 2); r /= 2;
    const int dx = (int)(r % 3); r /= 3;
    const int xy = (int)(r % 2) * 2 + exi; r /= 2;
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

bool se_conv3d_wino44pp_takes(const ConvArgs& a, int batch);                  // conv3d_wino44pp.hip: F(4,3) x F(4,3), ping-pong form
int se_conv3d_wino44pp_launch(const ConvArgs& a, int batch, hipStream_t s);
#ifdef SE_DEVTOOLS
bool se_conv3d_wino44_takes(const ConvArgs& a);                               // conv3d_wino44.hip (development builds)
int se_conv3d_wino44_launch(const ConvArgs& a, int batch, hipStream_t s);
#endif

// Returns 0 on launch, SE_TILED_NOT_TAKEN if the shape/flags are not covered, else a hipError_t.
int se_conv3d_wino2d_try(const ConvArgs& a, int batch, int launch_batch, hipStream_t s) {
    const int dim = a.dim;
    if (!a.wpack_g || a.cin_pad != a.cin || !se_wino2d_shape_ok(dim, a.cin, a.cout)) return SE_TILED_NOT_TAKEN;
    if (a.flags & (SE_EPI_RES_POST_RELU | SE_EPI_OUT_PLANAR)) return SE_TILED_NOT_TAKEN;
#ifdef SE_DEVTOOLS
    // experiment (round 3, se_debug_set_variant(63)): F(4,3) x F(4,3), lockstep form with an LDS-DMA weight stream (conv3d_wino44.hip)
    if (g_variant == 63 && se_conv3d_wino44_takes(a)) return se_conv3d_wino44_launch(a, batch, s);
#endif
    // round 4: the 64^3 / 32^3 levels run on the F(4,3) x F(4,3) ping-pong kernel (1/4 of the direct MFMAs; this kernel: 1/3);
    // development builds: se_debug_set_variant(64) keeps them here (A/B)
    if (g_variant != 64 && g_variant < 41 && se_conv3d_wino44pp_takes(a, launch_batch)) return se_conv3d_wino44pp_launch(a, batch, s);
    // this kernel's planar layout is octet-planar; the one quad-planar form it has: channels-last in, quad-planar out, no skip tensor
    const bool quad_out_only = (a.flags & SE_LAYOUT_QUAD_BITS) == SE_OUT_QUAD && !(a.flags & SE_LAYOUT_OCTET_BITS) && !a.res && !a.pool_out &&
                               !(a.flags & SE_EPI_SKIPCONV16);
    if ((a.flags & SE_LAYOUT_QUAD_BITS) && !quad_out_only) return SE_ERR_BAD_ARG;
    const int tx = dim / 16, ty = dim / 8, tz = dim / 4;
    const long long total_tiles = (long long)batch * tx * ty * tz;
    const long long n_units = total_tiles * (a.cout / 32);
    if (n_units >= (1LL << 30)) return SE_TILED_NOT_TAKEN;
    const int cus = se_num_cus();
    const int grid = (int)(n_units < cus ? n_units : cus);
    const int per = (int)((n_units + grid - 1) / grid);
    unsigned long long* dbg = nullptr;
#ifdef SE_STAMP2D
    dbg = g_w2d_dbg;
#endif
#define W2_LAUNCH(E, L)                                                                                                         \
    do {                                                                                                                        \
        auto kern = conv3d_k3_wino2d_kernel<E, L>;                                                                              \
        SE_ENSURE_LDS(kern, W2_LDS_BYTES);                                                                                      \
        hipLaunchKernelGGL(kern, dim3((unsigned)((n_units + per - 1) / per)), dim3(512), W2_LDS_BYTES, s, a, a.wpack_g, tx, ty, \
                           tz, (int)total_tiles, (int)n_units, per, dbg);                                                       \
    } while (0)
    const int layout = ((a.flags & SE_IN_OCTET) ? 1 : 0) | ((a.flags & SE_OUT_OCTET) ? 2 : 0) | ((a.flags & SE_RES_OCTET) && a.res ? 4 : 0);
#ifdef SE_DEVTOOLS
    if ((layout == 0 || layout == 3) && g_variant >= 41) {
#define W2_VAR(E) do { if (layout == 3) W2_LAUNCH(E, 3); else W2_LAUNCH(E, 0); } while (0)
        switch (g_variant) {
            case 41: W2_VAR(0x2000); break;   // packed epilogue arithmetic
            case 42: W2_VAR(0x1000); break;   // input loads in the staging phase
            case 43: W2_VAR(0x1800); break;   // both
            case 44: W2_VAR(4); break;
            case 47: W2_VAR(7); break;
            case 48: W2_VAR(0x400); break;   // staging alone
            case 49: W2_VAR(0x200); break;   // staging wave prioritised
            case 50: W2_VAR(0x100); break;   // no priorities
            case 51: W2_VAR(0x4000); break;  // attribution: weight half not written to LDS
            case 52: W2_VAR(0x8000); break;  // attribution: V tile not written to LDS
            case 53: W2_VAR(0x10000); break; // attribution: no input loads
            case 54: W2_VAR(0x1C000); break; // attribution: all three
            case 55: W2_VAR(0x1C400); break; // ... and no MFMA stream
            case 56: W2_VAR(0x1C004); break; // all three + no epilogue memory traffic
            case 57: W2_VAR(0x20000); break; // attribution: no skip-tensor loads
            case 58: W2_VAR(0x40000); break; // attribution: no output stores
            case 60: W2_VAR(0x100000); break; // experiment: LDS-counter group barriers + cross-group waits instead of workgroup barriers
            case 61: W2_VAR(0x200000); break; // experiment (round 3): lockstep form, no ping-pong
            default: W2_VAR(0); break;
        }
#undef W2_VAR
        SE_CHECK_LAUNCH();
        return 0;
    }
#endif
    if (a.flags & SE_EPI_SKIPCONV16) {
        if (layout != 3 || a.pool_out || !a.skip_w || !a.res) return SE_ERR_BAD_ARG;
        W2_LAUNCH(0, 19);
        SE_CHECK_LAUNCH();
        return 0;
    }
    if (a.pool_out) {
        if (layout == 3) W2_LAUNCH(0, 11);
        else if (layout == 7) W2_LAUNCH(0, 15);
        else return SE_ERR_BAD_ARG;       // pooled output: octet-planar in / out only (what the V2V program uses)
        SE_CHECK_LAUNCH();
        return 0;
    }
    if (quad_out_only) {
        W2_LAUNCH(0, 32);
        SE_CHECK_LAUNCH();
        return 0;
    }
    switch (layout) {
        case 1: W2_LAUNCH(0, 1); break;
        case 2: W2_LAUNCH(0, 2); break;
        case 3: W2_LAUNCH(0, 3); break;
        case 4: W2_LAUNCH(0, 4); break;
        case 5: W2_LAUNCH(0, 5); break;
        case 6: W2_LAUNCH(0, 6); break;
        case 7: W2_LAUNCH(0, 7); break;
        default: W2_LAUNCH(0, 0); break;
    }
#undef W2_LAUNCH
    SE_CHECK_LAUNCH();
    return 0;
}

#if defined(SE_DEVTOOLS) && defined(SE_STAMP2D)
// Diagnostic builds only: u64 device buffer [workgroups][8 waves][16] filled by the stamp build of conv3d_k3_wino2d_kernel.
extern "C" void se_debug_set_stamp_buffer_2d(void* p) { g_w2d_dbg = reinterpret_cast<unsigned long long*>(p); }
#endif
