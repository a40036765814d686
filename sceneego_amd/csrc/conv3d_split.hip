// 3x3x3 convolution with float32 tensors and SPLIT-bf16 arithmetic on v_mfma_f32_16x16x32_bf16 (round 3, EXPERIMENTAL: reported as
// an `extra`, never part of the float32 headline).  Stands in for Conv3d(k=3) + BatchNorm3d (+ReLU) (+skip add) of Res3DBlock
// (reference network/v2v.py:21-43) at the 64^3 / 32^3 / 16^3 levels, same tensors and epilogue flags as se_conv3d_f32.
//
// Why: the float32 MFMA executes on the vector ALUs and shares them with the Winograd transforms (DESIGN.md section 4, round-3
// finding 3: the 2-D Winograd kernel stops at 0.60 of the f32 peak); the bf16 MFMA has its own pipe at 16x the rate.  Every float32
// operand is split into two bfloat16 values, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits together), and a
// product is taken as  hi*hi + hi*lo + lo*hi  with float32 accumulation (the dropped lo*lo term is 2^-16 of a product that is itself
// known to 2^-16: relative error of a product ~ 2^-16..2^-17).  tests/test_oracle_golden.py emulates this scheme through the whole
// V2V with exact float32 arithmetic on bf16-valued tensors: joints 2.2e-5 m from the reference golden, the same distance as the
// float32 Winograd kernels.
//
// Structure = conv_bf16_k3_kernel of conv3d_bf16_tiled.hip (LDS-tiled direct convolution, 16-channel chunks, one k step = 2 taps x
// 2 octets, all loads of a stage in flight under the MFMAs of the previous one) with: 512 threads (wave w = x plane w of an
// 8(x) x 4(y) x 16(z) tile, two waves per SIMD), float32 activations converted to the (hi, lo) pair while they are staged (hi and lo
// halo tiles side by side in LDS), two packed weight arrays (hi, lo: se_conv3d_pack_bf16 of the folded weights' halves), three MFMAs
// per (k step, voxel tile, cout tile), float32 bias / skip tensor / ReLU / stores.
#include "bf16_common.h"

namespace {

__device__ __forceinline__ u16x8 lds16(const unsigned char* p) { return *reinterpret_cast<const u16x8*>(p); }

constexpr int S3_TX = 8, S3_TY = 4, S3_TZ = 16;
constexpr int S3_HX = S3_TX + 2, S3_HY = S3_TY + 2, S3_HZ = S3_TZ + 2;
constexpr int S3_HALO_VOX = S3_HX * S3_HY * S3_HZ;           // 1080
constexpr int S3_HALO_PIECES = S3_HALO_VOX * 2;              // (voxel, octet) pairs: 2160
constexpr int S3_HALO_BYTES = S3_HALO_PIECES * 16;           // 34,560 per half (hi | lo)
constexpr int S3_KPC = 14;                                   // k steps per 16-channel chunk: ceil(27 taps x 2 octets / 4)
constexpr int S3_W_PIECES = S3_KPC * 2 * 64;                 // 1792 x 16 B per half (two cout tiles)
constexpr int S3_W_BYTES = S3_W_PIECES * 16;                 // 28,672 per half
constexpr int S3_LDS_BYTES = 2 * S3_HALO_BYTES + 2 * S3_W_BYTES;   // 126,464
constexpr int S3_NT = 512;
constexpr int S3_HP = (S3_HALO_PIECES + S3_NT - 1) / S3_NT;  // 5 halo pieces per thread
constexpr int S3_WP = 2 * S3_W_PIECES / S3_NT;               // 7 weight pieces per thread (hi pieces first, then lo)
static_assert(2 * S3_W_PIECES % S3_NT == 0, "weight pieces per thread");

struct SplitArgs {
    const float* in;
    const unsigned short* wpack_hi;
    const unsigned short* wpack_lo;
    const float* bpack;
    const float* res;
    float* out;
    int dim, cin_pad, cout, flags, nchunk, ksteps;
};

struct S3Stage {
    f32x4 h[S3_HP][2];      // 8 float32 channels of a (voxel, octet) piece
    u16x8 w[S3_WP];
};

__device__ __forceinline__ void s3_load_stage(S3Stage& st, const SplitArgs& a, const int* hvox, unsigned hmask, int mb, int c, int tid) {
    const int coff = c * 16 + (tid & 1) * 8;      // piece i = tid + 512 j: octet i & 1 = tid & 1
#pragma unroll
    for (int j = 0; j < S3_HP; ++j) {             // branch-free: out-of-volume pieces read voxel 0 and are zeroed by a select
        const float* p = a.in + (long long)hvox[j] * a.cin_pad + coff;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
        const bool ok = (hmask >> j) & 1;
        st.h[j][0] = ok ? v0 : (f32x4){0.f, 0.f, 0.f, 0.f};
        st.h[j][1] = ok ? v1 : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < S3_WP; ++j) {
        int i = tid + S3_NT * j;
        const unsigned short* src = i < S3_W_PIECES ? a.wpack_hi : a.wpack_lo;
        i = i < S3_W_PIECES ? i : i - S3_W_PIECES;
        const int m = i / (S3_KPC * 64), r = i - m * (S3_KPC * 64);
        st.w[j] = *reinterpret_cast<const u16x8*>(src + ((size_t)(mb * 2 + m) * a.ksteps + c * S3_KPC) * 512 + r * 8);
    }
}

__device__ __forceinline__ void s3_commit_stage(const S3Stage& st, unsigned char* halo, unsigned char* wts, int tid) {
#pragma unroll
    for (int j = 0; j < S3_HP; ++j) {
        const int i = tid + S3_NT * j;
        const float x[8] = {st.h[j][0].x, st.h[j][0].y, st.h[j][0].z, st.h[j][0].w, st.h[j][1].x, st.h[j][1].y, st.h[j][1].z, st.h[j][1].w};
        u16x8 hi, lo;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            hi[k] = f2bf(x[k]);
            lo[k] = f2bf(x[k] - bf2f(hi[k]));
        }
        if (i < S3_HALO_PIECES) {
            *reinterpret_cast<u16x8*>(halo + i * 16) = hi;
            *reinterpret_cast<u16x8*>(halo + S3_HALO_BYTES + i * 16) = lo;
        }
    }
#pragma unroll
    for (int j = 0; j < S3_WP; ++j) *reinterpret_cast<u16x8*>(wts + (tid + S3_NT * j) * 16) = st.w[j];     // hi half, then lo half
}

// the 14 k steps of one chunk: per (voxel tile n, cout tile m) three MFMAs  Ah*Bh + Ah*Bl + Al*Bh ; operand fragments of the next
// k step are read between the MFMAs of this one
__device__ __forceinline__ void s3_compute(f32x4 (&acc)[2][S3_TY], const unsigned char* brow, const unsigned char* arow, int g) {
    u16x8 Ah[2], Al[2], Bh[S3_TY], Bl[S3_TY];
    auto read_step = [&](int sl, u16x8 (&ah)[2], u16x8 (&al)[2], u16x8 (&bh)[S3_TY], u16x8 (&bl)[S3_TY]) {
        int tap = 2 * sl + (g >> 1);
        tap = tap > 26 ? 26 : tap;             // padding group of the last k step: zero weights, any valid address
        const unsigned char* bp = brow + (((tap / 9) * S3_HY + (tap / 3) % 3) * S3_HZ + tap % 3) * 32;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ah[m] = lds16(arow + (m * S3_KPC + sl) * 1024);
            al[m] = lds16(arow + S3_W_BYTES + (m * S3_KPC + sl) * 1024);
        }
#pragma unroll
        for (int n = 0; n < S3_TY; ++n) {
            bh[n] = lds16(bp + n * (S3_HZ * 32));
            bl[n] = lds16(bp + S3_HALO_BYTES + n * (S3_HZ * 32));
        }
    };
    read_step(0, Ah, Al, Bh, Bl);
    __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * S3_TY, 0);
#pragma unroll
    for (int sl = 0; sl < S3_KPC; ++sl) {
        u16x8 nAh[2], nAl[2], nBh[S3_TY], nBl[S3_TY];
        if (sl + 1 < S3_KPC) read_step(sl + 1, nAh, nAl, nBh, nBl);
        // product outer: the eight accumulators rotate, a dependent MFMA is eight issues away
#pragma unroll
        for (int n = 0; n < S3_TY; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m][n] = mfma_bf16(Ah[m], Bh[n], acc[m][n]);
#pragma unroll
        for (int n = 0; n < S3_TY; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m][n] = mfma_bf16(Ah[m], Bl[n], acc[m][n]);
#pragma unroll
        for (int n = 0; n < S3_TY; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m][n] = mfma_bf16(Al[m], Bh[n], acc[m][n]);
        if (sl + 1 < S3_KPC) {
#pragma unroll
            for (int m = 0; m < 2; ++m) { Ah[m] = nAh[m]; Al[m] = nAl[m]; }
#pragma unroll
            for (int n = 0; n < S3_TY; ++n) { Bh[n] = nBh[n]; Bl[n] = nBl[n]; }
            // 12 reads of the next step spread over the 24 MFMAs of this one
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * S3_TY, 0);
        }
    }
}

__global__ __launch_bounds__(S3_NT) void conv_split3_k3_kernel(SplitArgs a, int tiles_x, int tiles_y, int tiles_z) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* halo = lds;                              // [hi | lo] x [hx][hy][hz][2 octets][16 B]
    unsigned char* wts = lds + 2 * S3_HALO_BYTES;           // [hi | lo] x [cout tile][k step][lane][16 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int v = lane & 15, g = lane >> 4;
    const int D = a.dim;
    const int mb = blockIdx.y;
    int t = blockIdx.x;
    const int tz = t % tiles_z; t /= tiles_z;
    const int ty = t % tiles_y; t /= tiles_y;
    const int tx = t % tiles_x;
    const int b = t / tiles_x;
    const int x0 = tx * S3_TX, y0 = ty * S3_TY, z0 = tz * S3_TZ;

    // this thread's halo pieces: global voxel index and in-volume mask, fixed for the whole tile
    int hoff[S3_HP];
    unsigned hmask = 0;
#pragma unroll
    for (int j = 0; j < S3_HP; ++j) {
        const int i = tid + S3_NT * j;
        const int hv = i >> 1;
        const int hz = hv % S3_HZ, hy = (hv / S3_HZ) % S3_HY, hx = hv / (S3_HZ * S3_HY);
        const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
        const bool ok = i < S3_HALO_PIECES && (unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && (unsigned)gz < (unsigned)D;
        hoff[j] = ok ? ((b * D + gx) * D + gy) * D + gz : 0;
        hmask |= (ok ? 1u : 0u) << j;
    }

    f32x4 acc[2][S3_TY];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < S3_TY; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned char* brow = halo + ((w * S3_HY) * S3_HZ + v) * 32 + (g & 1) * 16;
    const unsigned char* arow = wts + lane * 16;
    const long long obase = (((long long)b * D + (x0 + w)) * D + y0) * D + (z0 + v);   // output voxel of tile n: + n * D
    const bool has_res = a.res && (a.flags & (SE_EPI_RES_PRE_RELU | SE_EPI_RES_POST_RELU));

    S3Stage st;
    s3_load_stage(st, a, hoff, hmask, mb, 0, tid);
    for (int c = 0; c + 1 < a.nchunk; ++c) {
        __syncthreads();                       // every wave is done reading the previous chunk
        s3_commit_stage(st, halo, wts, tid);
        __syncthreads();
        s3_load_stage(st, a, hoff, hmask, mb, c + 1, tid);   // in flight under the MFMAs below
        s3_compute(acc, brow, arow, g);
    }
    __syncthreads();
    s3_commit_stage(st, halo, wts, tid);
    __syncthreads();
    // last chunk: bias and the skip tensor travel under the MFMAs
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bpack + mb * 32 + 8 * g);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bpack + mb * 32 + 8 * g + 4);
    f32x4 r0[S3_TY], r1[S3_TY];
#pragma unroll
    for (int n = 0; n < S3_TY; ++n) {
        r0[n] = r1[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has_res) {
            const float* rp = a.res + (obase + (long long)n * D) * a.cout + mb * 32 + 8 * g;
            r0[n] = *reinterpret_cast<const f32x4*>(rp);
            r1[n] = *reinterpret_cast<const f32x4*>(rp + 4);
        }
    }
    s3_compute(acc, brow, arow, g);
    // lane (v, g) holds channels mb*32 + 8g .. +7 of voxel tile n (se_bf16_cout_of: the two D fragments of a tile pair)
    const float lo_clamp = (a.flags & SE_EPI_RELU) ? 0.f : -__builtin_inff();
#pragma unroll
    for (int n = 0; n < S3_TY; ++n) {
        f32x4 o0 = acc[0][n] + b0, o1 = acc[1][n] + b1;
        if (a.flags & SE_EPI_RES_PRE_RELU) { o0 += r0[n]; o1 += r1[n]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { o0[k] = fmaxf(o0[k], lo_clamp); o1[k] = fmaxf(o1[k], lo_clamp); }
        if (a.flags & SE_EPI_RES_POST_RELU) { o0 += r0[n]; o1 += r1[n]; }
        float* op = a.out + (obase + (long long)n * D) * a.cout + mb * 32 + 8 * g;
        *reinterpret_cast<f32x4*>(op) = o0;
        *reinterpret_cast<f32x4*>(op + 4) = o1;
    }
}

}  // namespace

// float32 tensors (channels-last [B][D][D][D][C]), split-bf16 arithmetic.  `wpack_hi` / `wpack_lo`: se_conv3d_pack_bf16 (ksize 3, no
// BatchNorm arguments) of hi = bf16(w') and lo = w' - hi, w' = the BatchNorm-folded float32 weights; `bpack`: the folded float32 bias
// (round_up16(cout) floats, as se_conv3d_pack_f32 writes it).  Shapes: dim % 16 == 0, cin_pad % 16 == 0, cout % 32 == 0;
// flags: SE_EPI_RELU, SE_EPI_RES_PRE_RELU / _POST_RELU.  SE_ERR_BAD_ARG otherwise (nothing launched).
extern "C" int se_conv3d_k3_split3_f32(const float* in, const se_bf16* wpack_hi, const se_bf16* wpack_lo, const float* bpack,
                                       const float* residual, float* out, int batch, int dim, int cin_pad, int cout, int flags,
                                       void* stream) {
    if (!in || !wpack_hi || !wpack_lo || !bpack || !out) return SE_ERR_BAD_ARG;
    if (batch <= 0 || dim < 16 || (dim & 15) || cin_pad <= 0 || (cin_pad & 15) || cout <= 0 || (cout & 31)) return SE_ERR_BAD_ARG;
    if (flags & ~(SE_EPI_RELU | SE_EPI_RES_PRE_RELU | SE_EPI_RES_POST_RELU)) return SE_ERR_BAD_ARG;
    if ((flags & SE_EPI_RES_PRE_RELU) && (flags & SE_EPI_RES_POST_RELU)) return SE_ERR_BAD_ARG;
    if ((long long)batch * dim * dim * dim >= (1LL << 31)) return SE_ERR_BAD_ARG;          // 32-bit voxel indices in the halo table
    const PackGeomB p = pack_geom_b(cout, cin_pad, 3, 0);
    if (p.oc != 2 || p.kpc != S3_KPC) return SE_ERR_BAD_ARG;
    SplitArgs a;
    a.in = in; a.wpack_hi = wpack_hi; a.wpack_lo = wpack_lo; a.bpack = bpack; a.res = residual; a.out = out;
    a.dim = dim; a.cin_pad = cin_pad; a.cout = cout; a.flags = flags; a.nchunk = p.nchunk; a.ksteps = p.ksteps;
    const int tx = dim / S3_TX, ty = dim / S3_TY, tz = dim / S3_TZ;
    // one workgroup per tile (a persistent form with the staging pipeline running across tiles measured SLOWER, 0.525 vs 0.421 ms
    // at 32->32 @64^3: its extra live state spilled 184 B of registers)
    SE_ENSURE_LDS(conv_split3_k3_kernel, S3_LDS_BYTES);
    hipLaunchKernelGGL(conv_split3_k3_kernel, dim3((unsigned)(batch * tx * ty * tz), cout / 32), dim3(S3_NT), S3_LDS_BYTES,
                       se_stream(stream), a, tx, ty, tz);
    SE_CHECK_LAUNCH();
    return 0;
}
