// 3x3x3 convolution with float32 tensors and SPLIT-bf16 arithmetic on v_mfma_f32_16x16x32_bf16 (round 3, EXPERIMENTAL: reported as
// an `extra`, never part of the float32 headline).  Stands in for Conv3d(k=3) + BatchNorm3d (+ReLU) (+skip add) of Res3DBlock
// (reference network/v2v.py:21-43) at the 64^3 / 32^3 / 16^3 levels, same tensors and epilogue flags as se_conv3d_f32.
//
// Why: the float32 MFMA executes on the vector ALUs and shares them with the Winograd transforms (DESIGN.md section 4, round-3
// finding 3: the 2-D Winograd kernel stops at 0.60 of the f32 peak); the bf16 MFMA has its own pipe at 16x the rate.  Every float32
// operand is split into two bfloat16 values, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits together), and a
// product is taken as  hi*hi + hi*lo + lo*hi  with float32 accumulation (the dropped lo*lo term is 2^-16 of a product that is itself
// known to 2^-16).  tests/test_oracle_golden.py emulates this scheme through the whole V2V with exact float32 arithmetic on
// bf16-valued tensors: joints 2.2e-5 m from the reference golden, the same distance as the float32 Winograd kernels; the kernel
// itself: 5e-6 of max|y| against torch-CPU float32 (tests/test_gpu_kernels.py), 4e-5 m on the joints (tests/test_gpu_forward.py).
//
// Structure (second form): LDS-tiled direct convolution, MFMA operand maps of bf16_common.h (A = weights, row = cout; B = activations,
// column = voxel; a lane's 16 bytes = 8 consecutive input channels).  Workgroup = 4 waves, output tile 4(x) x 4(y) x 16(z) voxels x 32
// couts; wave w owns x = w, its 4 voxel tiles are the y rows (16 z each).  The input channels are walked in STAGES of 8: the LDS holds
// the 6x6x18 halo of one octet as (hi, lo) pairs (20.7 KB) and the 7 k steps x 2 cout tiles of (hi, lo) weights (28.7 KB) = 49.4 KB,
// so two workgroups share a CU (2 waves per SIMD, 256 registers each) and one's loads / conversion / barriers / epilogue overlap the
// other's MFMAs.  (The first form - 512 threads, 16-channel chunks, 126 KB: ONE workgroup per CU - spent 57 % of a launch outside its
// MFMA phases: profiles/r03_wino2d_and_small_level_experiments.txt, item 8.)  A k step = 4 taps x 1 octet: lane group g reads tap
// 4s + g.  All loads of a stage are issued before the MFMAs of the previous one and committed behind them (float32 -> (hi, lo) on the
// way into the LDS); per (k step, voxel tile, cout tile) three MFMAs; float32 bias / skip tensor / ReLU / stores.
#include "bf16_common.h"

namespace {

__device__ __forceinline__ u16x8 lds16(const unsigned char* p) { return *reinterpret_cast<const u16x8*>(p); }

constexpr int S3_TX = 4, S3_TY = 4, S3_TZ = 16;
constexpr int S3_HX = S3_TX + 2, S3_HY = S3_TY + 2, S3_HZ = S3_TZ + 2;
constexpr int S3_HALO_VOX = S3_HX * S3_HY * S3_HZ;           // 648 voxels = 648 16-byte pieces per half (one octet)
constexpr int S3_HALO_BYTES = S3_HALO_VOX * 16;              // 10,368 per half (hi | lo)
constexpr int S3_KPS = 7;                                    // k steps per 8-channel stage: ceil(27 taps / 4)
constexpr int S3_W_PIECES = S3_KPS * 2 * 64;                 // 896 x 16 B per half (two cout tiles)
constexpr int S3_W_BYTES = S3_W_PIECES * 16;                 // 14,336 per half
constexpr int S3_LDS_BYTES = 2 * S3_HALO_BYTES + 2 * S3_W_BYTES;   // 49,408
constexpr int S3_NT = 256;
constexpr int S3_HP = (S3_HALO_VOX + S3_NT - 1) / S3_NT;     // 3 halo pieces per thread
constexpr int S3_WP = 2 * S3_W_PIECES / S3_NT;               // 7 weight pieces per thread (hi pieces first, then lo)
static_assert(2 * S3_W_PIECES % S3_NT == 0, "weight pieces per thread");

struct SplitArgs {
    const float* in;
    const unsigned short* wsplit;   // [half][cout block][stage][896 pieces][8]
    const float* bpack;
    const float* res;
    float* out;
    int dim, cin_pad, cout, flags, nstage;
};

struct S3Stage {
    f32x4 h[S3_HP][2];      // 8 float32 channels of a halo voxel
    u16x8 w[S3_WP];
};

// `ibase`: the sample's first record (channels-last) or first octet plane (octet-planar); hvox: voxel index inside the sample
__device__ __forceinline__ void s3_load_stage(S3Stage& st, const SplitArgs& a, const float* ibase, const int* hvox, unsigned hmask, int mb,
                                              int c, int tid) {
    const long long d3 = (long long)a.dim * a.dim * a.dim;
    const float* sbase = (a.flags & SE_IN_OCTET) ? ibase + c * d3 * 8 : ibase + c * 8;
    const int vstride = (a.flags & SE_IN_OCTET) ? 8 : a.cin_pad;
#pragma unroll
    for (int j = 0; j < S3_HP; ++j) {             // branch-free: out-of-volume pieces read voxel 0 and are zeroed by a select
        const float* p = sbase + (long long)hvox[j] * vstride;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
        const bool ok = (hmask >> j) & 1;
        st.h[j][0] = ok ? v0 : (f32x4){0.f, 0.f, 0.f, 0.f};
        st.h[j][1] = ok ? v1 : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int mbs = a.cout >> 5;
#pragma unroll
    for (int j = 0; j < S3_WP; ++j) {
        int i = tid + S3_NT * j;
        const int half = i >= S3_W_PIECES ? 1 : 0;
        i -= half * S3_W_PIECES;
        st.w[j] = *reinterpret_cast<const u16x8*>(a.wsplit + ((((size_t)half * mbs + mb) * a.nstage + c) * S3_W_PIECES + i) * 8);
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
        if (i < S3_HALO_VOX) {
            *reinterpret_cast<u16x8*>(halo + i * 16) = hi;
            *reinterpret_cast<u16x8*>(halo + S3_HALO_BYTES + i * 16) = lo;
        }
    }
#pragma unroll
    for (int j = 0; j < S3_WP; ++j) *reinterpret_cast<u16x8*>(wts + (tid + S3_NT * j) * 16) = st.w[j];     // hi half, then lo half
}

// the 7 k steps of one stage: per (voxel tile n, cout tile m) three MFMAs  Ah*Bh + Ah*Bl + Al*Bh ; operand fragments of the next
// k step are read between the MFMAs of this one
__device__ __forceinline__ void s3_compute(f32x4 (&acc)[2][S3_TY], const unsigned char* brow, const unsigned char* arow, int g) {
    u16x8 Ah[2], Al[2], Bh[S3_TY], Bl[S3_TY];
    auto read_step = [&](int sl, u16x8 (&ah)[2], u16x8 (&al)[2], u16x8 (&bh)[S3_TY], u16x8 (&bl)[S3_TY]) {
        int tap = 4 * sl + g;
        tap = tap > 26 ? 26 : tap;             // padding slot of the last k step: zero weights, any valid address
        const unsigned char* bp = brow + (((tap / 9) * S3_HY + (tap / 3) % 3) * S3_HZ + tap % 3) * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ah[m] = lds16(arow + (m * S3_KPS + sl) * 1024);
            al[m] = lds16(arow + S3_W_BYTES + (m * S3_KPS + sl) * 1024);
        }
#pragma unroll
        for (int n = 0; n < S3_TY; ++n) {
            bh[n] = lds16(bp + n * (S3_HZ * 16));
            bl[n] = lds16(bp + S3_HALO_BYTES + n * (S3_HZ * 16));
        }
    };
    read_step(0, Ah, Al, Bh, Bl);
    __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * S3_TY, 0);
#pragma unroll
    for (int sl = 0; sl < S3_KPS; ++sl) {
        u16x8 nAh[2], nAl[2], nBh[S3_TY], nBl[S3_TY];
        if (sl + 1 < S3_KPS) read_step(sl + 1, nAh, nAl, nBh, nBl);
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
        if (sl + 1 < S3_KPS) {
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

__global__ __launch_bounds__(S3_NT, 2) void conv_split3_k3_kernel(SplitArgs a, int tiles_x, int tiles_y, int tiles_z) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* halo = lds;                              // [hi | lo] x [hx][hy][hz][16 B]
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

    // this thread's halo voxels: global voxel index and in-volume mask, fixed for the whole tile
    int hoff[S3_HP];
    unsigned hmask = 0;
#pragma unroll
    for (int j = 0; j < S3_HP; ++j) {
        const int hv = tid + S3_NT * j;
        const int hz = hv % S3_HZ, hy = (hv / S3_HZ) % S3_HY, hx = hv / (S3_HZ * S3_HY);
        const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
        const bool ok = hv < S3_HALO_VOX && (unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && (unsigned)gz < (unsigned)D;
        hoff[j] = ok ? (gx * D + gy) * D + gz : 0;
        hmask |= (ok ? 1u : 0u) << j;
    }

    f32x4 acc[2][S3_TY];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < S3_TY; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned char* brow = halo + ((w * S3_HY) * S3_HZ + v) * 16;
    const unsigned char* arow = wts + lane * 16;
    const long long d3 = (long long)D * D * D;
    const int ovox = ((x0 + w) * D + y0) * D + (z0 + v);      // output voxel (inside the sample) of tile n: + n * D
    const bool has_res = a.res && (a.flags & (SE_EPI_RES_PRE_RELU | SE_EPI_RES_POST_RELU));
    // tensor layouts (as se_conv3d_f32's 2-D Winograd shapes): channels-last [B][D^3][C] or octet-planar [B][C/8][D^3][8]; lane (v, g)
    // owns the 8 consecutive channels mb*32 + 8g .. +7 = octet mb*4 + g of its voxels
    const float* ibase = (a.flags & SE_IN_OCTET) ? a.in + (long long)b * a.nstage * d3 * 8 : a.in + (long long)b * d3 * a.cin_pad;
    auto rec = [&](const float* base, bool oct, int n) -> long long {
        const long long vs = ovox + (long long)n * D;
        return oct ? (((long long)b * (a.cout >> 3) + mb * 4 + g) * d3 + vs) * 8 : ((long long)b * d3 + vs) * a.cout + mb * 32 + 8 * g;
    };

    S3Stage st;
    s3_load_stage(st, a, ibase, hoff, hmask, mb, 0, tid);
    for (int c = 0; c + 1 < a.nstage; ++c) {
        __syncthreads();                       // every wave is done reading the previous stage
        s3_commit_stage(st, halo, wts, tid);
        __syncthreads();
        s3_load_stage(st, a, ibase, hoff, hmask, mb, c + 1, tid);   // in flight under the MFMAs below
        s3_compute(acc, brow, arow, g);
    }
    __syncthreads();
    s3_commit_stage(st, halo, wts, tid);
    __syncthreads();
    // last stage: bias and the skip tensor travel under the MFMAs
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bpack + mb * 32 + 8 * g);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bpack + mb * 32 + 8 * g + 4);
    f32x4 r0[S3_TY], r1[S3_TY];
#pragma unroll
    for (int n = 0; n < S3_TY; ++n) {
        r0[n] = r1[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has_res) {
            const float* rp = a.res + rec(a.res, a.flags & SE_RES_OCTET, n);
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
        float* op = a.out + rec(a.out, a.flags & SE_OUT_OCTET, n);
        *reinterpret_cast<f32x4*>(op) = o0;
        *reinterpret_cast<f32x4*>(op + 4) = o1;
    }
}

// w: float32 [cout][cin][27] (BatchNorm already folded in) -> wsplit [half][cout block][stage][m][k step][lane][8]:
// lane (g, r): cout = se_bf16_cout_of(block*2 + m, r), tap = 4 * (k step) + g (27: zero), channel = stage*8 + j;
// half 0 = bf16(w), half 1 = bf16(w - half 0).
__global__ void pack_split3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout, int cin, int nstage, long long per_half) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= 2 * per_half) return;
    const int half = e >= per_half ? 1 : 0;
    long long r = e - (long long)half * per_half;
    const int j = (int)(r & 7); r >>= 3;
    const int lane = (int)(r & 63); r >>= 6;
    const int sl = (int)(r % S3_KPS); r /= S3_KPS;
    const int m = (int)(r & 1); r >>= 1;
    const int c = (int)(r % nstage);
    const int mb = (int)(r / nstage);
    const int g = lane >> 4, row = lane & 15;
    const int co = se_bf16_cout_of(cout, mb * 2 + m, row);
    const int tap = 4 * sl + g, ci = c * 8 + j;
    float v = 0.f;
    if (tap < 27 && co < cout && ci < cin) v = w[((size_t)co * cin + ci) * 27 + tap];
    const unsigned short hi = f2bf(v);
    out[e] = half ? f2bf(v - bf2f(hi)) : hi;
}

}  // namespace

extern "C" long long se_conv3d_split3_packed_elems(int cout, int cin_pad) {
    if (cout <= 0 || (cout & 31) || cin_pad <= 0 || (cin_pad & 7)) return -1;
    return 2LL * (cout / 32) * (cin_pad / 8) * S3_W_PIECES * 8;
}

// w: float32 [cout][cin][3][3][3] with the BatchNorm scale already folded in (host); wsplit: se_conv3d_split3_packed_elems bf16.
extern "C" int se_conv3d_split3_pack(const float* w, se_bf16* wsplit, int cout, int cin, int cin_pad, void* stream) {
    if (!w || !wsplit || cout <= 0 || (cout & 31) || cin <= 0 || cin_pad < cin || (cin_pad & 7)) return SE_ERR_BAD_ARG;
    const int nstage = cin_pad / 8;
    const long long per_half = (long long)(cout / 32) * nstage * S3_W_PIECES * 8;
    hipLaunchKernelGGL(pack_split3_kernel, dim3((unsigned)((2 * per_half + 255) / 256)), dim3(256), 0, se_stream(stream), w, wsplit, cout, cin,
                       nstage, per_half);
    SE_CHECK_LAUNCH();
    return 0;
}

// float32 tensors (channels-last [B][D][D][D][C]), split-bf16 arithmetic.  `wsplit`: se_conv3d_split3_pack; `bpack`: the folded float32
// bias (round_up16(cout) floats, as se_conv3d_pack_f32 writes it).  Shapes: dim % 16 == 0, cin_pad % 8 == 0, cout % 32 == 0;
// flags: SE_EPI_RELU, SE_EPI_RES_PRE_RELU / _POST_RELU, SE_IN_OCTET / SE_OUT_OCTET / SE_RES_OCTET (octet-planar tensors
// [B][C/8][D][D][D][8]: an 8-channel stage of a halo row is one contiguous run, and a lane's 8 output channels are one octet record).
// SE_ERR_BAD_ARG otherwise (nothing launched).
extern "C" int se_conv3d_k3_split3_f32(const float* in, const se_bf16* wsplit, const float* bpack, const float* residual, float* out,
                                       int batch, int dim, int cin_pad, int cout, int flags, void* stream) {
    if (!in || !wsplit || !bpack || !out) return SE_ERR_BAD_ARG;
    if (batch <= 0 || dim < 16 || (dim & 15) || cin_pad <= 0 || (cin_pad & 7) || cout <= 0 || (cout & 31)) return SE_ERR_BAD_ARG;
    if (flags & ~(SE_EPI_RELU | SE_EPI_RES_PRE_RELU | SE_EPI_RES_POST_RELU | SE_IN_OCTET | SE_OUT_OCTET | SE_RES_OCTET)) return SE_ERR_BAD_ARG;
    if ((flags & SE_EPI_RES_PRE_RELU) && (flags & SE_EPI_RES_POST_RELU)) return SE_ERR_BAD_ARG;
    if ((long long)batch * dim * dim * dim >= (1LL << 31)) return SE_ERR_BAD_ARG;          // 32-bit voxel indices in the halo table
    SplitArgs a;
    a.in = in; a.wsplit = wsplit; a.bpack = bpack; a.res = residual; a.out = out;
    a.dim = dim; a.cin_pad = cin_pad; a.cout = cout; a.flags = flags; a.nstage = cin_pad / 8;
    const int tx = dim / S3_TX, ty = dim / S3_TY, tz = dim / S3_TZ;
    SE_ENSURE_LDS(conv_split3_k3_kernel, S3_LDS_BYTES);
    hipLaunchKernelGGL(conv_split3_k3_kernel, dim3((unsigned)(batch * tx * ty * tz), cout / 32), dim3(S3_NT), S3_LDS_BYTES,
                       se_stream(stream), a, tx, ty, tz);
    SE_CHECK_LAUNCH();
    return 0;
}
