// EXPERIMENT of round 3, development builds only (csrc/build.sh --devtools, se_debug_set_variant(63)); production 3x3x3 layers run on
// conv3d_wino2d.hip.  Status and measurements: DESIGN.md section 4, round-3 finding 9.
//
// 3x3x3 convolution, float32, 2-D Winograd F(4,3) along z AND y (direct along x), on v_mfma_f32_16x16x4_f32.
// Stands in for Conv3d(k=3) + BatchNorm3d (+ReLU) (+skip add) of Res3DBlock (reference network/v2v.py:21-43) at the 64^3 / 32^3
// levels (plain forms: no pooled output, no fused skip convolution).
//
// Per 4(z) x 4(y) outputs and x tap the transform domain has 6 x 6 = 36 points instead of 4*4*3*3 = 144 products: 1/4 of the direct
// MFMAs (F(4,3) x F(2,3): 1/3).  The float32 MFMA shares the vector ALUs with the transforms, so the
// structure is the one that worked for the 7^3 layer (conv3d_wino67.hip): all eight waves run the MFMA phase together (a wave
// stalled on a rider is covered by its SIMD partner), the transforms run between the phases, the weights arrive by LDS-DMA.
//
// Work unit = (32-cout block, tile of 8(z) x 8(y) x 16(x) outputs); wave (zt, yt, ct) owns z-tile zt, y-tile yt, cout tile ct:
// 36 accumulators of 16 couts x 16 x positions (144 registers) over ALL input channels, the output is written once.
// Channels are walked in chunks of 4 = the four k lanes of one MFMA.  Per chunk a step is
//     MFMA phase   108 MFMAs per wave: 9 xi quads x 3 dx x 4, operands by ds_read_b128 from
//                    W [q 9][dx 3][ct 2][lane][4 xi]                 55.3 KB, G-transformed weights of the (cout block, chunk); xi = 6 xi_y + xi_z
//                    V [zt 2][yt 2][18 x records][channel 4][36 xi]  42.6 KB, B^T-transformed input (record stride 148 floats)
//     transform    pass 1: B^T along y of the next chunk's 10 x 10 x 18 halo (raw rows loaded inside the MFMA phase) into a scratch
//                  tile T, barrier, pass 2: B^T along z from T into V; after the last chunk the output transform + epilogue.
// Weight stream: ONE chunk buffer, refilled by global_load_lds_dwordx4 in two regions while the other one is read (quads 5..8 of
// the current chunk during quads 0..4, quads 0..4 of the next chunk during quads 5..8), a barrier between the halves.
#include "conv_common.h"

#include <type_traits>
#include <utility>

namespace {

template <typename F, int... S>
__device__ __forceinline__ void for_each_i(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}

constexpr int Q_W_FLOATS = SE_WINO44_CHUNK_FLOATS;        // 13,824 floats = 55,296 B
constexpr int Q_QA = 5;                                   // weight region A = quads 0..4, region B = quads 5..8
constexpr int Q_WA_FLOATS = Q_QA * 3 * 2 * 256;           // 7680
constexpr int Q_NA = Q_QA * 3 * 2 * 64 / 64;              // wave-instructions (64 lanes x 16 B) of region A: 30
constexpr int Q_NB = (9 - Q_QA) * 3 * 2;                  // 24
constexpr int Q_RS = 148;                                 // floats per x record of V: 4 channels x 36 xi + 4 pad (37 x 16 B: odd -> bank spread)
constexpr int Q_VT = 18 * Q_RS;                           // one (zt, yt) tile of V
constexpr int Q_V_FLOATS = 4 * Q_VT;                      // 10,656 floats = 42,624 B
constexpr int Q_T_FLOATS = 2 * 6 * 10 * 18 * 4;           // scratch of pass 1: [yt][xi_y][z 10][x 18][4 channels] = 8640 floats = 34,560 B
constexpr int Q_LDS_BYTES = (Q_W_FLOATS + Q_V_FLOATS + Q_T_FLOATS) * 4;   // 132,480 B
constexpr int Q_GROUPS = 27;                              // (quad, dx) groups of 4 MFMAs
constexpr int Q_GA = Q_QA * 3;                            // groups that read region A: 15

#ifndef SE_K44_EXP      // attribution builds only (results wrong): 1 no weight riders, 2 no input riders, 4 no transforms, 8 no epilogue
#define SE_K44_EXP 0
#endif

#ifdef SE_STAMP44   // cycle stamps (tools/stamp_k44.py; development builds with -DSE_STAMP44)
#define T44(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
                 st_sum[i] += (unsigned)(t_ - st_last); st_last = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define T44(i)
#endif

// F(4,3) B^T (points 0, +-1, +-2, inf) on six values, per component
__device__ __forceinline__ void bt43(const f32x4 (&d)[6], f32x4 (&o)[6]) {
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
__device__ __forceinline__ void at43(const f32x4& m0, const f32x4& m1, const f32x4& m2, const f32x4& m3, const f32x4& m4, const f32x4& m5,
                                     f32x4 (&y)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float s12 = m1[c] + m2[c], d12 = m1[c] - m2[c], s34 = m3[c] + m4[c], d34 = m3[c] - m4[c];
        y[0][c] = (m0[c] + s12) + s34;
        y[1][c] = fmaf(2.f, d34, d12);
        y[2][c] = fmaf(4.f, s34, s12);
        y[3][c] = fmaf(8.f, d34, d12) + m5[c];
    }
}

// the same on one half (two components) of the vectors: the output transform runs per half to keep its temporaries at 48 registers
typedef float f32x2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2h half_of(const f32x4& v, int hf) { return hf ? (f32x2h){v.z, v.w} : (f32x2h){v.x, v.y}; }
__device__ __forceinline__ void at43h(const f32x2h& m0, const f32x2h& m1, const f32x2h& m2, const f32x2h& m3, const f32x2h& m4, const f32x2h& m5,
                                      f32x2h (&y)[4]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float s12 = m1[c] + m2[c], d12 = m1[c] - m2[c], s34 = m3[c] + m4[c], d34 = m3[c] - m4[c];
        y[0][c] = (m0[c] + s12) + s34;
        y[1][c] = fmaf(2.f, d34, d12);
        y[2][c] = fmaf(4.f, s34, s12);
        y[3][c] = fmaf(8.f, d34, d12) + m5[c];
    }
}

// LAYOUT: bit 0 = input octet-planar [B][cin/8][D][D][D][8] (SE_IN_OCTET), bit 1 = output octet-planar, bit 2 = skip tensor octet-planar
template <int LAYOUT>
__global__ __launch_bounds__(512) void conv3d_k3_wino44_kernel(ConvArgs a, const float* __restrict__ wg, int tiles_x, int tiles_y, int tiles_z,
                                                               int total_tiles, int n_units, int units_per_wg, unsigned long long* dbg) {
    (void)dbg;
#ifdef SE_STAMP44
    unsigned st_sum[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = 0;
#endif
    constexpr bool in_oct = LAYOUT & 1, out_oct = LAYOUT & 2, res_oct = LAYOUT & 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* vt = lds + Q_W_FLOATS;
    float* tt = vt + Q_V_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int zt = wave >> 2, yt = (wave >> 1) & 1, ct = wave & 1;
    const int px = lane & 15, h = lane >> 4;
    const int dim = a.dim, cin = a.cin;
    const int chunks = cin >> 2;
    const int u_begin = (int)blockIdx.x * units_per_wg;
    const int u_end = min(u_begin + units_per_wg, n_units);
    if (u_begin >= u_end) return;
    const int n_steps = (u_end - u_begin) * chunks;

    struct Unit { int cb, b, z0, y0, x0; };
    auto decode = [&](int u) {
        Unit r;
        r.cb = u / total_tiles;
        int t = u - r.cb * total_tiles;
        const int xt = t % tiles_x; t /= tiles_x;
        const int yy = t % tiles_y; t /= tiles_y;
        const int zz = t % tiles_z;
        r.b = t / tiles_z;
        r.z0 = zz * 8; r.y0 = yy * 8; r.x0 = xt * 16;
        return r;
    };

    // ---- MFMA operand addresses ----
    const float* a_base = wl + ct * 256 + lane * 4;                                   // + (q * 3 + dx) * 512
    const float* b_base = vt + (zt * 2 + yt) * Q_VT + px * Q_RS + h * 36;             // + dx * Q_RS + q * 4

    // ---- pass 1 role: task (yt1, z1, x1) = thread, 360 of them: six y rows of 4 channels -> six xi_y.
    // ---- pass 2 role: task (zt2, yt2, xi_y, x2) = thread, 432 of them: six z slabs of 4 channels -> six xi_z.
    // The task coordinates are recomputed from the thread index where they are used (a few VALU outside the MFMA phase): kept in
    // registers across the loop they were spilled, and a scratch reload waits for every older vector-memory operation.
    const bool p1_on = tid < 360, p2_on = tid < 432;
    auto opaque_tid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };

    // ---- input rows of the next step: raw buffer loads, descriptor base = sample (+ chunk), row r in the scalar offset, the
    // rest per lane; bit 31 of the lane offset marks a voxel outside the volume (reads zero) ----
    constexpr unsigned OOB = 0x80000000u;
    const unsigned in_bytes = (unsigned)dim * dim * dim * cin * 4u;
    const int vstride = in_oct ? 32 : cin * 4;                                        // bytes between x neighbours
    f32x4 raw[6];
    unsigned f_voff = OOB, f_rowok = 0;
    const float* f_base = a.in;
    auto fetch_setup = [&](const Unit& u, int c4) {      // branch-free (bitwise conditions): nothing here may split a scheduling region
        const int p1 = min(opaque_tid(), 359);
        const int x1 = p1 % 18, z1 = (p1 / 18) % 10, yt1 = p1 / 180;
        const int gz = u.z0 - 1 + z1, gx = u.x0 - 1 + x1, gy0 = u.y0 - 1 + 4 * yt1;
        const bool ok = p1_on & ((unsigned)gz < (unsigned)dim) & ((unsigned)gx < (unsigned)dim);
        // (gy0 may be -1: the row term is added modulo 2^32; rows outside the volume are masked per row below)
        f_voff = ok ? (unsigned)(((gz * dim + gy0) * dim + gx) * vstride) : OOB;
        unsigned m = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) m |= ((unsigned)(gy0 + r) < (unsigned)dim ? 1u : 0u) << r;
        f_rowok = ok ? m : 0u;
        f_base = a.in + (long long)u.b * dim * dim * dim * cin + (in_oct ? (long long)(c4 >> 1) * dim * dim * dim * 8 + (c4 & 1) * 4 : c4 * 4);
    };
    auto fetch_one = [&](auto r_tag) {
        constexpr int r = decltype(r_tag)::value;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f_base), 0, (int)in_bytes, 0x00020000);
        const unsigned v = ((f_rowok >> r) & 1u) ? f_voff + (unsigned)(r * dim * vstride) : OOB;      // (f_rowok = 0 for lanes without a voxel)
        raw[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)v, 0, 0));
    };
    auto fetch_rows = [&]() { for_each_i(fetch_one, std::make_integer_sequence<int, 6>{}); };
    auto pass1 = [&]() {
        if (!p1_on || (SE_K44_EXP & 4)) return;
        const int p1 = opaque_tid();
        const int x1 = p1 % 18, z1 = (p1 / 18) % 10, yt1 = p1 / 180;
        float* t_dst = tt + ((yt1 * 6 * 10 + z1) * 18 + x1) * 4;                      // + xi_y * 720
        f32x4 o[6];
        bt43(raw, o);
#pragma unroll
        for (int e = 0; e < 6; ++e) *reinterpret_cast<f32x4*>(t_dst + e * 720) = o[e];
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto pass2 = [&]() {
        if (!p2_on || (SE_K44_EXP & 4)) return;
        const int q2 = opaque_tid();
        const int x2 = q2 % 18, xy2 = (q2 / 18) % 6, yt2 = (q2 / 108) & 1, zt2 = q2 / 216;
        const float* t_src = tt + (((yt2 * 6 + xy2) * 10 + 4 * zt2) * 18 + x2) * 4;   // + s * 72 (z slab)
        float* v_dst = vt + (zt2 * 2 + yt2) * Q_VT + x2 * Q_RS + xy2 * 6;             // + channel * 36
        f32x4 d[6], o[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) d[s] = *reinterpret_cast<const f32x4*>(t_src + s * 72);
        bt43(d, o);
#pragma unroll
        for (int c = 0; c < 4; ++c) {       // channel c: its six xi_z are 24 consecutive bytes (8-byte aligned)
            f32x2* dst = reinterpret_cast<f32x2*>(v_dst + c * 36);
            dst[0] = (f32x2){o[0][c], o[1][c]};
            dst[1] = (f32x2){o[2][c], o[3][c]};
            dst[2] = (f32x2){o[4][c], o[5][c]};
        }
    };

    // ---- weight stream: LDS-DMA piece J of a region for this wave = wave-instruction 8 J + wave (surplus ones repeat the last) ----
    auto wglds = [&](const float* src, float* region, auto n_tag, auto j_tag) {
        constexpr int NWI = decltype(n_tag)::value, J = decltype(j_tag)::value;
        if (SE_K44_EXP & 1) return;
        int piece = J * 8 + wave;
        piece = piece < NWI ? piece : NWI - 1;
        const float* sp = src + piece * 256;        // uniform: the per-lane part of every LDS-DMA address is the same lane * 16 bytes
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(sp + lane * 4),
                                         (void __attribute__((address_space(3)))*)(region + piece * 256), 16, 0, 0);
    };
    using NA = std::integral_constant<int, Q_NA>;
    using NB = std::integral_constant<int, Q_NB>;

    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    const float relu_lo = (a.flags & SE_EPI_RELU) ? 0.f : -__builtin_inff();
    const bool use_res = (a.flags & SE_EPI_RES_PRE_RELU) && a.res;

    f32x4 acc[36];

    // Epilogue of a finished tile, in two parts around the transforms of the next step: part 1 right behind the MFMA phase = output
    // transform (A^T along y - one xi_z at a time, its six accumulators die - then along z) into 16 output vectors, with the 16
    // skip-tensor loads issued between the two halves; part 2 behind pass 2 = bias, skip tensor, ReLU, 16 x 16-byte stores.  The
    // loads have the two transform passes to land (waited one by one they cost 23 k cycles per tile).
    // addresses: uniform 64-bit base of the wave's first output row + a 32-bit per-lane offset + uniform (y, z) strides (global_*
    // saddr form: one VGPR of address for all 16 accesses)
    const int lo_cl = px * a.cout + 4 * h, lo_oc = (h >> 1) * dim * dim * dim * 8 + px * 8 + (h & 1) * 4;
    const int o_lane = out_oct ? lo_oc : lo_cl, r_lane = res_oct ? lo_oc : lo_cl;
    const int o_ys = out_oct ? dim * 8 : dim * a.cout, o_zs = o_ys * dim;
    const int r_ys = res_oct ? dim * 8 : dim * a.cout, r_zs = r_ys * dim;
    // a pointer the compiler can prove wave-uniform (scalar base of the global_* saddr form instead of a 64-bit address per lane)
    // (returned as a GLOBAL pointer: rebuilt from an integer as a generic one it turned every access into a flat_load, which counts on
    // lgkmcnt as well and stalls the LDS traffic of the transform passes behind HBM latency)
    typedef float __attribute__((address_space(1))) gfloat;
    typedef f32x4 __attribute__((address_space(1))) gf32x4;
    auto uniform_ptr = [&](const float* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<gfloat*>(((unsigned long long)hi << 32) | lo);
    };
    // Epilogue of a finished tile (tile-end steps do NOT prefetch the next step's rows inside their MFMA phase, so nothing but the
    // accumulators is live here): the 16 skip-tensor loads and the bias go out first, the output transform runs IN PLACE in the
    // accumulator registers under their latency - along y, xi_z by xi_z, acc[6 y + xi_z] <- t[xi_z][y] (xi_y = 4, 5 of that column
    // die); then along z, y by y, acc[6 y + z] <- output (y, z) - then bias, skip tensor, ReLU and 16 x 16-byte stores.
    // skip-tensor addressing of a tile (uniform base, see above); rv[y][z]
    auto res_base = [&](const Unit& u) {
        const int gz0 = u.z0 + 4 * zt, gy0 = u.y0 + 4 * yt;
        const long long cl = ((((long long)u.b * dim + gz0) * dim + gy0) * dim + u.x0) * a.cout + u.cb * 32 + ct * 16;
        const long long oc = (((((long long)u.b * (a.cout >> 3) + u.cb * 4 + ct * 2) * dim + gz0) * dim + gy0) * dim + u.x0) * 8;
        return uniform_ptr(a.res + (res_oct ? oc : cl));
    };
    auto load_rv = [&](f32x4 (&rv)[4][4], const gfloat* rb, int y) {
#pragma unroll
        for (int z = 0; z < 4; ++z) rv[y][z] = *reinterpret_cast<const gf32x4*>(rb + z * r_zs + y * r_ys + r_lane);
    };
    auto epilogue = [&](const Unit& u) {
        if (SE_K44_EXP & 8) return;
        f32x4 rv[4][4];
        const gfloat* rb = res_base(u);
        if (use_res) { load_rv(rv, rb, 0); load_rv(rv, rb, 1); }
        const int gz0 = u.z0 + 4 * zt, gy0 = u.y0 + 4 * yt;
        const long long cl = ((((long long)u.b * dim + gz0) * dim + gy0) * dim + u.x0) * a.cout + u.cb * 32 + ct * 16;
        const long long oc = (((((long long)u.b * (a.cout >> 3) + u.cb * 4 + ct * 2) * dim + gz0) * dim + gy0) * dim + u.x0) * 8;
        gfloat* ob = uniform_ptr(a.out + (out_oct ? oc : cl));
        const f32x4 bias = *reinterpret_cast<const gf32x4*>(uniform_ptr(a.bpack + u.cb * 32 + ct * 16) + 4 * h);
        // y = 2, 3 of the skip tensor (y = 0, 1 came in under the MFMA phase): behind the y transform, when 48 accumulator registers
        // have died - a spill here costs a memory round trip per reload
#pragma unroll
        for (int xz = 0; xz < 6; ++xz) {
            f32x4 y4[4];
            at43(acc[0 * 6 + xz], acc[1 * 6 + xz], acc[2 * 6 + xz], acc[3 * 6 + xz], acc[4 * 6 + xz], acc[5 * 6 + xz], y4);
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[y * 6 + xz] = y4[y];
        }
        T44(11);
        if (use_res) { load_rv(rv, rb, 2); load_rv(rv, rb, 3); }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            f32x4 o[4];
            at43(acc[y * 6 + 0], acc[y * 6 + 1], acc[y * 6 + 2], acc[y * 6 + 3], acc[y * 6 + 4], acc[y * 6 + 5], o);
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                f32x4 v = o[z] + bias;
                if (use_res) v += rv[y][z];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], relu_lo);
                *reinterpret_cast<gf32x4*>(ob + z * o_zs + y * o_ys + o_lane) = v;
            }
        }
    };

    auto read_ops = [&](f32x4& oa, f32x4& ov, auto g_tag) {
        constexpr int g = decltype(g_tag)::value;
        constexpr int q = g / 3, dx = g % 3;
        oa = *reinterpret_cast<const f32x4*>(a_base + (q * 3 + dx) * 512);
        ov = *reinterpret_cast<const f32x4*>(b_base + dx * Q_RS + q * 4);
    };

#ifndef SE_K44_STAGGER
#define SE_K44_STAGGER 0
#endif
    // Stagger the workgroups of an XCD (blockIdx % 8 = XCD) over the tile period: all workgroups have identical work, so without
    // this every CU reaches its tile end - skip-tensor loads, 16 stores per lane, the next tile's rows - in the same microsecond
    // and a memory round trip inside the epilogue takes ~10 k cycles instead of ~2 k (stamps, round 3)
    if (SE_K44_STAGGER > 0) {
        const int ph = ((int)blockIdx.x >> 3) & 7;
        for (int k = 0; k < ph * SE_K44_STAGGER; ++k) __builtin_amdgcn_s_sleep(100);       // 6400 cycles each
    }
    // ---- prologue: V of step 0, weight region A of step 0 ----
    Unit ucur = decode(u_begin);
    int ccur = 0;
    int ui = u_begin;
    fetch_setup(ucur, 0);
    fetch_rows();
    {
        const float* src = wg + ((size_t)ucur.cb * chunks) * Q_W_FLOATS;
        for (int i = tid; i < Q_WA_FLOATS / 4; i += 512) reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(src)[i];
    }
    pass1();
    __syncthreads();
    pass2();
    __syncthreads();

    for (int i = 0; i < n_steps; ++i) {
        const bool has_next = i + 1 < n_steps;
        const bool last_chunk = ccur == chunks - 1;
        Unit unx = ucur;
        int cnx = ccur;
        if (has_next) {
            if (last_chunk) { cnx = 0; unx = decode(ui + 1); }
            else cnx = ccur + 1;
        }
        const float* w_cur = wg + ((size_t)ucur.cb * chunks + ccur) * Q_W_FLOATS;
        const float* w_nxt = wg + ((size_t)unx.cb * chunks + cnx) * Q_W_FLOATS;
        fetch_setup(unx, cnx);        // lane offsets / row mask / base of the next step's input rows (loaded inside the MFMA phase)

        // ------------------------------ MFMA phase ------------------------------
        if (ccur == 0) {
#pragma unroll
            for (int e = 0; e < 36; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        f32x4 oa[3], ov[3];
#ifdef SE_STAMP44
        { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
        read_ops(oa[0], ov[0], std::integral_constant<int, 0>{});
        read_ops(oa[1], ov[1], std::integral_constant<int, 1>{});
        auto group = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            constexpr int q = g / 3, b = g % 3;
            // riders: region B of this chunk (3 LDS-DMAs per wave) and the next step's six input rows in the first half, region A of
            // the next chunk (4 per wave) behind the mid barrier
            if constexpr (g < 3) wglds(w_cur + Q_WA_FLOATS, wl + Q_WA_FLOATS, NB{}, std::integral_constant<int, g>{});
            if constexpr (g == 3 && !(SE_K44_EXP & 2)) {
                asm volatile("" ::: "memory");       // the input loads stay behind the LDS-DMAs: the counted wait at the mid barrier relies on it
                // a tile-end step loads half of the finishing tile's skip tensor here and its next rows behind the MFMA phase
                if (!last_chunk) fetch_rows();
            }
            if constexpr (g >= Q_GA && g < Q_GA + 4) wglds(w_nxt, wl, NA{}, std::integral_constant<int, g - Q_GA>{});
            constexpr bool pre = g + 2 < Q_GROUPS && g + 2 != Q_GA && g + 2 != Q_GA + 1;      // the first reads of region B wait for the mid barrier
            if constexpr (pre) read_ops(oa[(g + 2) % 3], ov[(g + 2) % 3], std::integral_constant<int, g + 2>{});
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * q + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][j], ov[b][j], acc[4 * q + j], 0, 0, 0);
            if constexpr (pre) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if constexpr (g + 1 == Q_GA) {      // everybody is past region A; region B of this chunk has landed (6 input loads may still fly)
                T44(0);
                if (last_chunk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                barrier();
                T44(1);
                read_ops(oa[Q_GA % 3], ov[Q_GA % 3], std::integral_constant<int, Q_GA>{});
                read_ops(oa[(Q_GA + 1) % 3], ov[(Q_GA + 1) % 3], std::integral_constant<int, Q_GA + 1>{});
            }
        };
        for_each_i(group, std::make_integer_sequence<int, Q_GROUPS>{});

        // ------------------------------ transforms / epilogue ------------------------------
        T44(2);
        // (the tile-end step is its own code path: the 32 output / skip vectors of the split epilogue exist only there)
        if (last_chunk) {
            T44(10);
            epilogue(ucur);
            T44(12);
            fetch_rows();       // first chunk of the next tile, latency exposed once per tile (unconditional: behind the last step it
                                // reloads rows nobody reads, but the old rows are provably dead across the epilogue)
        }
        T44(3);
        barrier();                    // V and W have been read by everybody
        T44(4);
        if (has_next) pass1();
        T44(5);
        barrier();
        T44(6);
        if (has_next) pass2();
        // region A of the next chunk has landed; the 16 stores of a finished tile (younger) may still be in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        T44(7);
        barrier();
        T44(8);
        if (last_chunk) ++ui;
        ucur = unx; ccur = cnx;
    }
#ifdef SE_STAMP44
    if (lane == 0 && dbg) {
        unsigned long long* o = dbg + ((size_t)blockIdx.x * 8 + wave) * 14;
        for (int k = 0; k < 13; ++k) o[k] = st_sum[k];
        o[13] = n_steps;
    }
#endif
}

}  // namespace

// (section I of the packed weights is written by conv3d_wino44pp.hip, the production form of this experiment)

// Shapes / flag sets this kernel takes (the caller, se_conv3d_wino2d_try, has checked se_wino2d_shape_ok and cin_pad == cin):
// no pooled output, no fused skip convolution, dim >= 32 (at 16^3 a batch of 8 has only 128 tiles of 8 x 8 x 16).
bool se_conv3d_wino44_takes(const ConvArgs& a) {
    return a.wpack_i && a.dim >= 32 && !a.pool_out && !a.skip_w && !(a.flags & SE_EPI_SKIPCONV16);
}

// Returns 0 on launch, else a hipError_t.
unsigned long long* g_w44_dbg = nullptr;
#if defined(SE_STAMP44)
extern "C" void se_debug_set_stamp_buffer_44(void* p) { g_w44_dbg = reinterpret_cast<unsigned long long*>(p); }
#endif
int se_conv3d_wino44_launch(const ConvArgs& a, int batch, hipStream_t s) {
    const int dim = a.dim;
    const int tx = dim / 16, ty = dim / 8, tz = dim / 8;
    const long long total_tiles = (long long)batch * tx * ty * tz;
    const long long n_units = total_tiles * (a.cout / 32);
    const int cus = se_num_cus();
    const int grid = (int)(n_units < cus ? n_units : cus);
    const int per = (int)((n_units + grid - 1) / grid);
#define Q_LAUNCH(L)                                                                                                             \
    do {                                                                                                                        \
        auto kern = conv3d_k3_wino44_kernel<L>;                                                                                 \
        SE_ENSURE_LDS(kern, Q_LDS_BYTES);                                                                                       \
        hipLaunchKernelGGL(kern, dim3((unsigned)((n_units + per - 1) / per)), dim3(512), Q_LDS_BYTES, s, a, a.wpack_i, tx, ty,  \
                           tz, (int)total_tiles, (int)n_units, per, g_w44_dbg);                                                 \
    } while (0)
    const int layout = ((a.flags & SE_IN_OCTET) ? 1 : 0) | ((a.flags & SE_OUT_OCTET) ? 2 : 0) | ((a.flags & SE_RES_OCTET) && a.res ? 4 : 0);
    switch (layout) {
        case 1: Q_LAUNCH(1); break;
        case 2: Q_LAUNCH(2); break;
        case 3: Q_LAUNCH(3); break;
        case 4: Q_LAUNCH(4); break;
        case 5: Q_LAUNCH(5); break;
        case 6: Q_LAUNCH(6); break;
        case 7: Q_LAUNCH(7); break;
        default: Q_LAUNCH(0); break;
    }
#undef Q_LAUNCH
    SE_CHECK_LAUNCH();
    return 0;
}
