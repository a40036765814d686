// V2V 3D convolutions on the gfx950 matrix cores, exact float32 (v_mfma_f32_16x16x4_f32).
//
// Layout: activations are channels-last [B][Z][Y][X][C]; a convolution is the implicit GEMM
//     out[cout][voxel] = sum_{tap, cin} W[cout][cin][tap] * in[voxel + tap][cin]
// One MFMA computes a 16(cout) x 16(voxel) tile over 4 values of k.  Operand maps (guide §3):
//     A: lane l holds A[i = l&15][k = l>>4]   -> weights,     i = cout
//     B: lane l holds B[k = l>>4][j = l&15]   -> activations, j = voxel
//     D: lane l holds D[i = 4*(l>>4) + r][j = l&15], r = 0..3 -> 4 consecutive couts of ONE voxel,
//        i.e. a 16-byte channels-last store per lane.
// A lane reads its activations as ONE 16-byte load: 4 consecutive channels c0..c0+3 with
// c0 = 16*cg + 4*(l>>4).  The four MFMAs j = 0..3 that consume it contract the channel set
// {16*cg + 4*h + j : h = 0..3}; the weight blocks are packed to match (se_conv3d_pack_f32), so a
// 16-channel group of one tap costs one 16-B load per operand and 4 MFMAs per (cout tile, voxel tile).
// The f32 MFMA is bit-for-bit an fmaf chain: no precision is traded for the matrix pipe.
//
// This file: weight packer (+ BatchNorm fold), the generic "direct" kernel (activations straight from
// global/L2, any k in {1,3,7}, any volume size, batch flattened into the voxel axis — used for the small
// pyramid levels and as the always-available path), ConvTranspose3d k2s2, and max-pool.
// The LDS-tiled kernels for the 64^3/32^3 levels live in conv3d_tiled.hip.
#include "common.h"

#include "conv_common.h"
#include "wino47_matrices.h"
#include "wino67_matrices.h"

namespace {

// ------------------------------------------------------------------------------------------------
// weight packing (see conv_common.h for the two block orders)
//   section A: wpack[cg][tap][nt][lane][j] = W[cout = 16*nt + (lane&15)][cin = 16*cg + 4*(lane>>4) + j][tap] * scale[cout]
//   section B (k = 7 only, read by the LDS-tiled 7x7x7 kernel: 4-channel chunks, 4 taps on the MFMA k lanes):
//              wpack[A + [ch][g][nt][lane][j]] = W[cout][cin = 4*ch + j][tap = 4*g + (lane>>4)] * scale[cout]
//   bpack[cout] = (b - mean) * scale + beta,  scale = gamma / sqrt(var + eps)    (identity without BN)
// ------------------------------------------------------------------------------------------------
__global__ void pack_kernel(const float* __restrict__ w, const float* __restrict__ b,
                            const float* __restrict__ gamma, const float* __restrict__ beta,
                            const float* __restrict__ mean, const float* __restrict__ var, float eps,
                            float* __restrict__ wpack, float* __restrict__ bpack, int cout, int cin, int cin_pad,
                            int taps, int transposed, long long total_a, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nts = round_up16(cout) / 16;
    if (t < round_up16(cout)) {
        float v = 0.f;
        if (t < cout) {
            const float sc = gamma ? gamma[t] / sqrtf(var[t] + eps) : 1.f;
            const float b0 = b ? b[t] : 0.f;
            v = gamma ? (b0 - mean[t]) * sc + beta[t] : b0;
        }
        bpack[t] = v;
    }
    if (t >= total) return;
    const int j = (int)(t & 3);
    const int lane = (int)((t >> 2) & 63);
    int nt, tap, ci;
    if (t < total_a) {
        long long r = t >> 8;
        nt = (int)(r % nts); r /= nts;
        tap = (int)(r % taps); r /= taps;
        ci = (int)r * 16 + 4 * (lane >> 4) + j;
    } else if (taps == 27 && t - total_a >= (long long)(cin_pad / 16) * (nts / 2) * SE_WINO_CHUNK_FLOATS) {
        // section E: 1-D Winograd F(4,3) along z, U = G g with G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],
        // [1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]; blocks [cg][cb][tap2d(9)][xi(6)][nt2][lane][j]
        long long r = (t - total_a - (long long)(cin_pad / 16) * (nts / 2) * SE_WINO_CHUNK_FLOATS) >> 8;
        const int nt2 = (int)(r % 2); r /= 2;
        const int xi = (int)(r % 6); r /= 6;
        const int tap2d = (int)(r % 9); r /= 9;
        const int n_cb = nts / 2;
        const int cb = (int)(r % n_cb); r /= n_cb;
        const int co = cb * 32 + nt2 * 16 + (lane & 15);
        const int cc = (int)r * 16 + 4 * (lane >> 4) + j;
        float v = 0.f;
        if (co < cout && cc < cin) {
            const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
            const float* wp = w + ((size_t)co * cin + cc) * 27 + tap2d;
            const float g0 = wp[0], g1 = wp[9], g2 = wp[18];
            float u;
            switch (xi) {
                case 0: u = g0 * 0.25f; break;
                case 1: u = -(g0 + g1 + g2) * (1.f / 6.f); break;
                case 2: u = -(g0 - g1 + g2) * (1.f / 6.f); break;
                case 3: u = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f); break;
                case 4: u = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f); break;
                default: u = g2; break;
            }
            v = u * sc;
        }
        wpack[t] = v;
        return;
    } else if (taps == 27) {
        // section C: 1-D Winograd F(2,3) along z.  U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2 of the
        // three z taps (g0: dz=-1) for every (dy,dx); blocks [cg][cb][tap2d][xi][nt2][lane][j], cb = 32-cout block
        long long r = (t - total_a) >> 8;
        const int nt2 = (int)(r % 2); r /= 2;
        const int xi = (int)(r % 4); r /= 4;
        const int tap2d = (int)(r % 9); r /= 9;
        const int n_cb = nts / 2;
        const int cb = (int)(r % n_cb); r /= n_cb;
        const int co = cb * 32 + nt2 * 16 + (lane & 15);
        const int cc = (int)r * 16 + 4 * (lane >> 4) + j;
        float v = 0.f;
        if (co < cout && cc < cin) {
            const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
            const float* wp = w + ((size_t)co * cin + cc) * 27 + tap2d;
            const float g0 = wp[0], g1 = wp[9], g2 = wp[18];
            const float u = xi == 0 ? g0 : xi == 1 ? (g0 + g1 + g2) * 0.5f : xi == 2 ? (g0 - g1 + g2) * 0.5f : g2;
            v = u * sc;
        }
        wpack[t] = v;
        return;
    } else if (t - total_a >= (long long)(cin_pad / 4) * SE_K7_GROUPS * nts * 256) {
        // section D (k = 7, cout <= 16): 1-D Winograd F(2,7) along z.  U_xi = sum_kz G[xi][kz] * W[..][kz][dy][dx];
        // blocks [chunk4][g(13)][xi(8)][lane][j]: k lane h carries the (dy,dx) tap 4g+h (taps >= 49 are zero padding)
        long long r = (t - total_a - (long long)(cin_pad / 4) * SE_K7_GROUPS * nts * 256) >> 8;
        const int xi = (int)(r % 8); r /= 8;
        const int g = (int)(r % SE_K7W_GROUPS); r /= SE_K7W_GROUPS;
        const int tap2d = 4 * g + (lane >> 4);
        const int cc = (int)r * 4 + j;
        const int co = lane & 15;
        float v = 0.f;
        if (co < cout && cc < cin && tap2d < 49) {
            const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
            const float* wp = w + ((size_t)co * cin + cc) * 343 + tap2d;
            float u = 0.f;
#pragma unroll
            for (int kz = 0; kz < 7; ++kz) u += se_wino27_G(xi, kz) * wp[kz * 49];
            v = u * sc;
        }
        wpack[t] = v;
        return;
    } else {
        long long r = (t - total_a) >> 8;
        nt = (int)(r % nts); r /= nts;
        const int g = (int)(r % SE_K7_GROUPS); r /= SE_K7_GROUPS;
        tap = 4 * g + (lane >> 4);
        ci = (int)r * 4 + j;
    }
    const int co = nt * 16 + (lane & 15);
    float v = 0.f;
    if (co < cout && ci < cin && tap < taps) {
        const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
        const float wv = transposed ? w[((size_t)ci * cout + co) * taps + tap] : w[((size_t)co * cin + ci) * taps + tap];
        v = wv * sc;
    }
    wpack[t] = v;
}

// Section F (k = 7, cout <= 16): 1-D Winograd F(4,7) along z.  U_xi = sum_kz G[xi][kz] * W[..][kz][dy][dx] (G: wino47_matrices.h).
// Per 3-channel chunk and (dy,dx) tap group g (k lane h carries tap 4g+h; taps >= 49 are zero padding) the 10 xi are stored as
// two quads and a tail so that a lane reads 4 xi x 3 channels = 48 B with three ds_read_b128:
//   [chunk3][g(13)] { Q0 [lane][xi 0..3][j] , Q1 [lane][xi 4..7][j] , T [lane][xi 8..9][j] }   (768 + 768 + 384 floats)
__global__ void pack_k7f_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                float eps, float* __restrict__ out, int cout, int cin, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int chunk = (int)(t / SE_K7F_CHUNK_FLOATS);
    int r = (int)(t - (long long)chunk * SE_K7F_CHUNK_FLOATS);
    const int g = r / 1920;
    r -= g * 1920;
    int lane, xi, j;
    if (r < 1536) {
        const int q = r / 768, rr = r - q * 768;
        lane = rr / 12;
        const int e = rr - lane * 12;
        xi = 4 * q + e / 3;
        j = e % 3;
    } else {
        const int rr = r - 1536;
        lane = rr / 6;
        const int e = rr - lane * 6;
        xi = 8 + e / 3;
        j = e % 3;
    }
    const int tap2d = 4 * g + (lane >> 4);
    const int cc = chunk * 3 + j;
    const int co = lane & 15;
    float v = 0.f;
    if (co < cout && cc < cin && tap2d < 49) {
        const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
        const float* wp = w + ((size_t)co * cin + cc) * 343 + tap2d;
        float u = 0.f;
#pragma unroll
        for (int kz = 0; kz < 7; ++kz) u += SE_W47_G[xi][kz] * wp[kz * 49];
        v = u * sc;
    }
    out[t] = v;
}

// Section H (k = 7, cout <= 16): 1-D Winograd F(6,7) along z for conv3d_wino67.hip.  U_xi = sum_kz G[xi][kz] * W[..][kz][dy][dx]
// (G: wino67_matrices.h).  Per 3-channel chunk the 147 (channel, dy, dx) taps are taken 4 at a time on the MFMA k lanes: k lane h of
// group g carries slot 4g+h = (channel * 49 + dy * 7 + dx), slot 147 is zero padding; a lane's 12 xi are 48 contiguous bytes:
//   [chunk3][g(37)][lane][xi 0..11]
__global__ void pack_k7h_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ var,
                                float eps, float* __restrict__ out, int cout, int cin, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int chunk = (int)(t / SE_K7H_CHUNK_FLOATS);
    int r = (int)(t - (long long)chunk * SE_K7H_CHUNK_FLOATS);
    const int g = r / 768;
    r -= g * 768;
    const int lane = r / 12, xi = r - lane * 12;
    const int slot = 4 * g + (lane >> 4);
    const int cl = slot / 49, tap2d = slot - cl * 49;
    const int cc = chunk * 3 + cl;
    const int co = lane & 15;
    float v = 0.f;
    if (co < cout && cc < cin && slot < 147) {
        const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
        const float* wp = w + ((size_t)co * cin + cc) * 343 + tap2d;
        float u = 0.f;
#pragma unroll
        for (int kz = 0; kz < 7; ++kz) u += SE_W67_G[xi][kz] * wp[kz * 49];
        v = u * sc;
    }
    out[t] = v;
}

// ------------------------------------------------------------------------------------------------
// direct implicit-GEMM kernel
// ------------------------------------------------------------------------------------------------
template <int KS, int M_T, int N_T, bool DECONV>
__global__ __launch_bounds__(256) void conv3d_direct_kernel(ConvArgs a) {
    constexpr int P = (KS - 1) / 2;
    constexpr int TAPS = KS * KS * KS;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int vl = lane & 15;   // voxel within the 16-voxel tile / cout within the weight tile
    const int h = lane >> 4;    // k group
    const int dim = a.dim;
    const int cgs = (a.cin + 15) >> 4;   // channel groups that hold real channels
    const int nt0 = blockIdx.y * N_T;
    const int sub = DECONV ? blockIdx.z : 0;  // (a,b,c) sub-position of the 2x2x2 transposed kernel

    // voxel coordinates of this lane's column in each of the wave's M_T tiles
    const long long v_base = ((long long)blockIdx.x * 4 + wave) * (M_T * 16);
    int vb[M_T], vz[M_T], vy[M_T], vx[M_T];
    bool vok[M_T];
#pragma unroll
    for (int m = 0; m < M_T; ++m) {
        const long long vid = v_base + m * 16 + vl;
        vok[m] = vid < a.total_vox;
        long long t = vok[m] ? vid : 0;
        vx[m] = (int)(t % dim); t /= dim;
        vy[m] = (int)(t % dim); t /= dim;
        vz[m] = (int)(t % dim); t /= dim;
        vb[m] = (int)t;
    }

    f32x4 acc[M_T][N_T];
#pragma unroll
    for (int m = 0; m < M_T; ++m)
#pragma unroll
        for (int n = 0; n < N_T; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack);
    for (int tap = 0; tap < (DECONV ? 1 : TAPS); ++tap) {
        const int dz = tap / (KS * KS) - P;
        const int dy = (tap / KS) % KS - P;
        const int dx = tap % KS - P;
        const float* src[M_T];
        bool ok[M_T];
#pragma unroll
        for (int m = 0; m < M_T; ++m) {
            const int zz = vz[m] + dz, yy = vy[m] + dy, xx = vx[m] + dx;
            ok[m] = vok[m] && (unsigned)zz < (unsigned)dim && (unsigned)yy < (unsigned)dim && (unsigned)xx < (unsigned)dim;
            const long long off = ((((long long)vb[m] * dim + zz) * dim + yy) * dim + xx) * a.cin_pad + 4 * h;
            src[m] = a.in + (ok[m] ? off : 0);
        }
        const int wtap = DECONV ? sub : tap;
        for (int cg = 0; cg < cgs; ++cg) {
            f32x4 xf[M_T];
#pragma unroll
            for (int m = 0; m < M_T; ++m) {
                xf[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok[m]) xf[m] = *reinterpret_cast<const f32x4*>(src[m] + cg * 16);
            }
            const f32x4* wrow = wp + ((size_t)(cg * (DECONV ? 8 : TAPS) + wtap) * a.nts + nt0) * 64 + lane;
#pragma unroll
            for (int n = 0; n < N_T; ++n) {
                const f32x4 wf = wrow[n * 64];
#pragma unroll
                for (int m = 0; m < M_T; ++m) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.x, xf[m].x, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.y, xf[m].y, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.z, xf[m].z, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.w, xf[m].w, acc[m][n], 0, 0, 0);
                }
            }
        }
    }

    // epilogue: lane owns couts co0..co0+3 of voxel (m, vl)
    const int odim = DECONV ? dim * 2 : dim;
    const long long ovox_per_b = (long long)odim * odim * odim;
#pragma unroll
    for (int m = 0; m < M_T; ++m) {
        if (!vok[m]) continue;
        int oz = vz[m], oy = vy[m], ox = vx[m];
        if (DECONV) {
            oz = 2 * oz + (sub >> 2);
            oy = 2 * oy + ((sub >> 1) & 1);
            ox = 2 * ox + (sub & 1);
        }
        const long long on = ((long long)oz * odim + oy) * odim + ox;  // voxel index inside the sample
#pragma unroll
        for (int n = 0; n < N_T; ++n) conv_epilogue(a, acc[m][n], vb[m], on, ovox_per_b, (nt0 + n) * 16 + 4 * h);
    }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose3d(k=2, s=2) + folded BN (+skip add) + ReLU (Upsample3DBlock, reference network/v2v.py:55-67), all eight
// sub-positions of a voxel tile in ONE workgroup.  HBM-bound (2 B of input, 16 B of output + 16 B of skip tensor per
// input-voxel-channel pair at 64->32): the direct kernel's grid.z form re-reads the input eight times, leaves the two x
// neighbours of an output line pair to different workgroups and waits for the skip tensor behind its MFMAs; here the input
// fragments of a tile stay in registers over the eight sub-positions, the (2x, 2x+1) records of all cout tiles are written back to
// back by the same wave, and the skip records of a sub-position are requested in front of its MFMAs.
// A wave owns M_T tiles of 16 consecutive voxels; CGS = cin / 16, N_T = cout / 16 (all couts in one workgroup).
// ------------------------------------------------------------------------------------------------
// OUTQ (round 5, SE_OUT_QUAD; dim % 16 == 0): the output is QUAD-planar [B][cout/4][2D][2D][2D][4] - what the 3x3x3 kernel of the
// block behind it (back_layers.0, reference network/v2v.py:155) reads whole 16-byte records of.  A lane's D fragment is one record,
// but the records of ONE x parity are 32 bytes apart (round 2 measured such half-line stores at 0.147 -> 0.229 ms for the octet-planar
// form); here both parities of a (z, y) sub-position are computed first and exchanged across lanes (ds_bpermute: lane l takes voxel
// l >> 1, parity l & 1) so that a store instruction writes 16 consecutive records = 256 contiguous bytes per cout quad.
// RESQ (with OUTQ, SE_RES_QUAD; SE_EPI_RES_POST_RELU only): the skip tensor is quad-planar as well and is added BEHIND the exchange, where
// a lane holds the record it stores - the skip read is then the same 256 contiguous bytes per 16 lanes as the store (the channels-last
// skip read of a sub-position is 64-byte pieces at a 256-byte stride), and the block that produced the skip tensor writes whole records.
template <int CGS, int N_T, int M_T, bool OUTQ, bool RESQ = false>
__global__ __launch_bounds__(256) void deconv3d_k2s2_kernel(ConvArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int vl = lane & 15;
    const int h = lane >> 4;
    const int dim = a.dim;
    const long long v_base = ((long long)blockIdx.x * 4 + wave) * (M_T * 16);
    int vb[M_T], vz[M_T], vy[M_T], vx[M_T];
    bool vok[M_T];
    f32x4 xf[M_T][CGS];
#pragma unroll
    for (int m = 0; m < M_T; ++m) {
        const long long vid = v_base + m * 16 + vl;
        vok[m] = vid < a.total_vox;
        long long t = vok[m] ? vid : 0;
        const float* src = a.in + t * a.cin_pad + 4 * h;
        vx[m] = (int)(t % dim); t /= dim;
        vy[m] = (int)(t % dim); t /= dim;
        vz[m] = (int)(t % dim); t /= dim;
        vb[m] = (int)t;
#pragma unroll
        for (int cg = 0; cg < CGS; ++cg) {
            xf[m][cg] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (vok[m]) xf[m][cg] = *reinterpret_cast<const f32x4*>(src + cg * 16);
        }
    }
    const int odim = dim * 2;
    const long long ovox_per_b = (long long)odim * odim * odim;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack) + lane;
    const bool relu = a.flags & SE_EPI_RELU;
    const bool res_pre = !RESQ && (a.flags & SE_EPI_RES_PRE_RELU) && a.res, res_post = !RESQ && (a.flags & SE_EPI_RES_POST_RELU) && a.res;
    f32x4 bias[N_T];
#pragma unroll
    for (int n = 0; n < N_T; ++n) bias[n] = *reinterpret_cast<const f32x4*>(a.bpack + n * 16 + 4 * h);
    f32x4 val[OUTQ ? 2 : 1][M_T][N_T];      // OUTQ: finished records of the two x parities of a (z, y) sub-position
#pragma unroll 1
    for (int sub = 0; sub < 8; ++sub) {
        // output records of this sub-position; the skip tensor is requested first so that its latency lies under the MFMAs
        long long ooff[M_T];
        f32x4 rv[M_T][N_T];
#pragma unroll
        for (int m = 0; m < M_T; ++m) {
            const int oz = 2 * vz[m] + (sub >> 2), oy = 2 * vy[m] + ((sub >> 1) & 1), ox = 2 * vx[m] + (sub & 1);
            ooff[m] = ((long long)vb[m] * ovox_per_b + ((long long)oz * odim + oy) * odim + ox) * a.cout + 4 * h;
#pragma unroll
            for (int n = 0; n < N_T; ++n) {
                rv[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if ((res_pre || res_post) && vok[m]) rv[m][n] = *reinterpret_cast<const f32x4*>(a.res + ooff[m] + n * 16);
            }
        }
        f32x4 acc[M_T][N_T];
#pragma unroll
        for (int m = 0; m < M_T; ++m)
#pragma unroll
            for (int n = 0; n < N_T; ++n) acc[m][n] = bias[n];
#pragma unroll
        for (int cg = 0; cg < CGS; ++cg) {
            f32x4 wf[N_T];
#pragma unroll
            for (int n = 0; n < N_T; ++n) wf[n] = wp[((size_t)(cg * 8 + sub) * N_T + n) * 64];
            // k step outer, accumulators inner: consecutive MFMAs never share an accumulator
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int m = 0; m < M_T; ++m)
#pragma unroll
                    for (int n = 0; n < N_T; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][k], xf[m][cg][k], acc[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < M_T; ++m) {
            if (!OUTQ && !vok[m]) continue;
#pragma unroll
            for (int n = 0; n < N_T; ++n) {
                f32x4 v = acc[m][n];
                if (res_pre) v += rv[m][n];
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (res_post) v += rv[m][n];
                if constexpr (OUTQ) {
                    if (sub & 1) val[1][m][n] = v; else val[0][m][n] = v;
                } else {
                    *reinterpret_cast<f32x4*>(a.out + ooff[m] + n * 16) = v;
                }
            }
        }
        if constexpr (OUTQ) {
            if (!(sub & 1)) continue;
            // both x parities of (oz, oy) are finished: lane l of a 16-lane row takes (voxel l >> 1, parity l & 1) for the records
            // x0 .. x0 + 15 and (voxel 8 + (l >> 1), parity l & 1) for x0 + 16 .. x0 + 31; the tile's 16 voxels are one x run
            const int src_a = ((lane & 48) | (vl >> 1)) * 4, src_b = src_a + 32;
            const bool odd = vl & 1;
#pragma unroll
            for (int m = 0; m < M_T; ++m) {
                if (!vok[m]) continue;                 // uniform over the wave: the tile is 16 aligned voxels
                const int oz = 2 * vz[m] + (sub >> 2), oy = 2 * vy[m] + ((sub >> 1) & 1), ox0 = 2 * (vx[m] - vl) + vl;
#pragma unroll
                for (int n = 0; n < N_T; ++n) {
                    // (component by component, written out: with a `for (c)` loop over the vector elements hipcc 7.2 permuted only
                    // element 0 and splatted it over the record - disassembly, round 5)
                    auto pull = [&](int src, float e) {
                        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, e)));
                    };
                    const f32x4 v0 = val[0][m][n], v1 = val[1][m][n];
                    const f32x4 a0 = {pull(src_a, v0.x), pull(src_a, v0.y), pull(src_a, v0.z), pull(src_a, v0.w)};
                    const f32x4 a1 = {pull(src_a, v1.x), pull(src_a, v1.y), pull(src_a, v1.z), pull(src_a, v1.w)};
                    const f32x4 b0 = {pull(src_b, v0.x), pull(src_b, v0.y), pull(src_b, v0.z), pull(src_b, v0.w)};
                    const f32x4 b1 = {pull(src_b, v1.x), pull(src_b, v1.y), pull(src_b, v1.z), pull(src_b, v1.w)};
                    f32x4 ra = odd ? a1 : a0, rb = odd ? b1 : b0;
                    const long long qo = (((((long long)vb[m] * (a.cout >> 2) + n * 4 + h) * odim + oz) * odim + oy) * odim + ox0) * 4;
                    if constexpr (RESQ) {       // quad-planar skip tensor: the records this lane stores
                        ra += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.res + qo));          // the skip tensor: read once
                        rb += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.res + qo + 64));
                    }
                    float* o = a.out + qo;
                    *reinterpret_cast<f32x4*>(o) = ra;
                    *reinterpret_cast<f32x4*>(o + 64) = rb;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// split-K variant for the small pyramid levels (8^3, 4^3, 2^3: 128 -> 128 channels, a few thousand voxels).
// There the direct kernel has only a handful of workgroups, each walking all 27 x 8 steps serially
// (0.24 ms per conv, MFMA-latency-bound on <10 % of the chip).  Here grid.z splits the 27 taps, every
// (voxel tile, cout tile, tap slice) is its own workgroup writing a partial sum to the workspace, and a second
// tiny kernel adds the slices in a FIXED order (bitwise deterministic, no atomics) and applies the epilogue.
// ------------------------------------------------------------------------------------------------
// Round 5 tried ONE launch (arrival counter per tile in the workspace, the last-arriving block reduces in the same fixed order):
// with device-scope release / acquire fences it cost +50 us per launch (a cache-wide write-back / invalidate per wave, 864 blocks),
// with device-scope (sc1) stores / loads of the partials it measured EQUAL to the two launches (the reduce launches are ~4.5 us each)
// and was not bit-stable from launch to launch - not shipped (profiles/r05_fork_and_splitk_ab.txt).
template <int N_T>
__global__ __launch_bounds__(256) void conv3d_k3_splitk_kernel(ConvArgs a, float* __restrict__ ws, int taps_per_split) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int vl = lane & 15;
    const int h = lane >> 4;
    const int dim = a.dim;
    const int cgs = (a.cin + 15) >> 4;
    const int nt0 = blockIdx.y * N_T;
    const int split = blockIdx.z;

    const long long vid = ((long long)blockIdx.x * 4 + wave) * 16 + vl;
    const bool vok = vid < a.total_vox;
    long long t = vok ? vid : 0;
    const int vx = (int)(t % dim); t /= dim;
    const int vy = (int)(t % dim); t /= dim;
    const int vz = (int)(t % dim); t /= dim;
    const int vb = (int)t;

    f32x4 acc[N_T];
#pragma unroll
    for (int n = 0; n < N_T; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack);
    const int tap0 = split * taps_per_split;
    for (int tap = tap0; tap < tap0 + taps_per_split; ++tap) {
        const int zz = vz + tap / 9 - 1, yy = vy + (tap / 3) % 3 - 1, xx = vx + tap % 3 - 1;
        const bool ok = vok && (unsigned)zz < (unsigned)dim && (unsigned)yy < (unsigned)dim && (unsigned)xx < (unsigned)dim;
        const float* src = a.in + (ok ? ((((long long)vb * dim + zz) * dim + yy) * dim + xx) * a.cin_pad + 4 * h : 0);
        // the launch is latency-bound: issue the loads of 4 channel groups (4 x (1 + N_T) x 16 B per lane) before their MFMAs
        for (int cg0 = 0; cg0 < cgs; cg0 += 4) {
            f32x4 xf[4], wf[4][N_T];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cg = cg0 + u < cgs ? cg0 + u : cgs - 1;
                xf[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok && cg0 + u < cgs) xf[u] = *reinterpret_cast<const f32x4*>(src + cg * 16);
                const f32x4* wrow = wp + ((size_t)(cg * 27 + tap) * a.nts + nt0) * 64 + lane;
#pragma unroll
                for (int n = 0; n < N_T; ++n) wf[u][n] = wrow[n * 64];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int n = 0; n < N_T; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][n].x, xf[u].x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][n].y, xf[u].y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][n].z, xf[u].z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u][n].w, xf[u].w, acc[n], 0, 0, 0);
                }
            }
        }
    }
    if (!vok) return;
    float* o = ws + ((size_t)split * a.total_vox + vid) * a.cout;
#pragma unroll
    for (int n = 0; n < N_T; ++n) *reinterpret_cast<f32x4*>(o + (nt0 + n) * 16 + 4 * h) = acc[n];
}

// In-workgroup split-K for the small pyramid levels (8^3, 4^3, 2^3; 128 -> 128 channels): the WAVES waves of a workgroup share ONE
// set of N_T voxel tiles x 2 cout tiles and split the 27 x cin/16 (tap, channel group) steps between them; the partial sums meet
// in LDS in a fixed order (deterministic) and the epilogue runs in the same launch - no workspace round trip, no second kernel.
//
// Round 3: the float32 MFMA runs on the vector ALUs (64 FLOP/clk/SIMD = the VALU rate; tools/diag/mfma_selfmix.hip: one VALU
// instruction per MFMA costs 17 % of the matrix rate), so the per-step address arithmetic of the first form - tap decomposition,
// three bounds compares, 64-bit address adds and a branch per load: ~6 VALU instructions per MFMA - took as long as the MFMAs.
// Now the whole k-step walk lives in scalar registers (the wave index goes through readfirstlane), loads are raw buffer loads
// whose uniform part (tap shift, channel group, weight block) sits in the descriptor base / scalar offset, a lane's voxel offset
// is computed once, and a neighbour outside the volume is one bit of a 27-bit per-lane mask that turns the voxel offset into an
// out-of-range one (the load returns zeros; no branch): 2 VALU instructions per activation load, none per weight load.
// Loads of block i+1 (U steps) are in flight under the MFMAs of block i (two named register sets).
template <int N_T, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void conv3d_k3_wavesplit_kernel(ConvArgs a) {
    __shared__ f32x4 red[WAVES][2 * N_T][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int vl = lane & 15;
    const int h = lane >> 4;
    const int dim = a.dim;
    const int cgs = (a.cin + 15) >> 4;
    const int nt0 = blockIdx.y * 2;
    constexpr unsigned OOB = 0x80000000u;
    // per lane, once: byte offset of the voxel's record (+ this k lane's channel quad) and the mask of taps that leave the volume
    unsigned voff[N_T], bad[N_T];
#pragma unroll
    for (int n = 0; n < N_T; ++n) {
        const long long vid = ((long long)blockIdx.x * N_T + n) * 16 + vl;
        const bool vok = vid < a.total_vox;
        long long t = vok ? vid : 0;
        const int vx = (int)(t % dim); t /= dim;
        const int vy = (int)(t % dim); t /= dim;
        const int vz = (int)(t % dim);
        voff[n] = (unsigned)((vok ? vid : 0) * a.cin_pad + 4 * h) * 4u;
        unsigned m = 0;
        for (int tap = 0; tap < 27; ++tap) {
            const int zz = vz + tap / 9 - 1, yy = vy + (tap / 3) % 3 - 1, xx = vx + tap % 3 - 1;
            const bool ok = vok && (unsigned)zz < (unsigned)dim && (unsigned)yy < (unsigned)dim && (unsigned)xx < (unsigned)dim;
            m |= (ok ? 0u : 1u) << tap;
        }
        bad[n] = m;
    }
    const unsigned in_bytes = (unsigned)(a.total_vox * a.cin_pad * 4);       // launcher: total_vox <= 8192, fits 32 bits
    const unsigned w_bytes = (unsigned)(27 * cgs * a.nts * 1024);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, (int)w_bytes, 0x00020000);
    const int woff = lane * 16;

    f32x4 acc[2][N_T];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < N_T; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int steps = 27 * cgs;
    constexpr int U = 4;
    struct Ops { f32x4 wf[U][2], xf[U][N_T]; };
    // the wave's k steps: s = wave, wave + WAVES, ...; (tap, cg) of the next step to LOAD, advanced incrementally (scalar)
    int ld_s = wave, ld_tap = wave / cgs, ld_cg = wave - (wave / cgs) * cgs;
    auto load_block = [&](Ops& o) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = ld_s < steps;                                   // uniform
            const int tap = live ? ld_tap : 0, cg = live ? ld_cg : 0;
            const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
            // activations: descriptor base = in + (tap shift, channel group); a dead step reads through a zero-record descriptor
            const float* xb = a.in + ((long long)(dz * dim + dy) * dim + dx) * a.cin_pad + cg * 16;
            const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, live ? (int)in_bytes : 0, 0x00020000);
#pragma unroll
            for (int n = 0; n < N_T; ++n) {
                const unsigned sel = (unsigned)__builtin_amdgcn_sbfe(bad[n], tap, 1);      // all ones when the tap leaves the volume
                o.xf[u][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(voff[n] | (sel & OOB)), 0, 0));
            }
            const int wso = live ? ((cg * 27 + tap) * a.nts + nt0) * 1024 : 0;
            o.wf[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, wso, 0));
            o.wf[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, wso + 1024, 0));
            ld_s += WAVES;
            ld_cg += WAVES;
            while (ld_cg >= cgs) { ld_cg -= cgs; ++ld_tap; }
        }
    };
    auto mfma_block = [&](const Ops& o) {      // a dead k step carries zero activations: it adds exact zeros
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float wv[2][4] = {{o.wf[u][0].x, o.wf[u][0].y, o.wf[u][0].z, o.wf[u][0].w}, {o.wf[u][1].x, o.wf[u][1].y, o.wf[u][1].z, o.wf[u][1].w}};
#pragma unroll
            for (int c = 0; c < 4; ++c)        // component outer: consecutive MFMAs go to different accumulators
#pragma unroll
                for (int n = 0; n < N_T; ++n) {
                    const float xv[4] = {o.xf[u][n].x, o.xf[u][n].y, o.xf[u][n].z, o.xf[u][n].w};
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[m][c], xv[c], acc[m][n], 0, 0, 0);
                }
        }
    };
    Ops oa, ob;
    load_block(oa);
    for (int s0 = wave; s0 < steps; s0 += 2 * U * WAVES) {
        load_block(ob);                         // beyond the last block: dead steps (zero-record descriptor, zero operands)
        mfma_block(oa);
        if (s0 + U * WAVES >= steps) break;
        load_block(oa);
        mfma_block(ob);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < N_T; ++n) red[wave][m * N_T + n][lane] = acc[m][n];
    __syncthreads();
    // 2 * N_T fragments, WAVES waves: wave w finishes fragments w, w + WAVES, ... (partials added in wave order: deterministic)
    const long long per_b = (long long)dim * dim * dim;
    for (int f = wave; f < 2 * N_T; f += WAVES) {
        const int m = f / N_T, n = f - m * N_T;
        f32x4 v = red[0][f][lane];
#pragma unroll
        for (int w2 = 1; w2 < WAVES; ++w2) v += red[w2][f][lane];
        const long long ov = ((long long)blockIdx.x * N_T + n) * 16 + vl;
        if (ov < a.total_vox) {
            const int b = (int)(ov / per_b);
            conv_epilogue(a, v, b, ov - (long long)b * per_b, per_b, (nt0 + m) * 16 + 4 * h);
        }
    }
}

// 4096-voxel levels with wide channels (8^3 at batch 8, 16^3 at batch 1; 128 -> 128): the in-workgroup split-K form above reads
// every operand of every k step straight from the vector cache - each voxel record 27 times, the weights once per 32 voxels:
// 442 MB of cache traffic per launch, 30 of its 45 us with the MFMAs knocked out (round 5, profiles/r05_small_levels.txt).
// This form stages the activations through LDS and shares a weight fragment between four voxel tiles:
//   workgroup = 64 consecutive voxels (whole rows: one z slab at dim 8, four rows at dim 16) x 32 couts, 8 waves;
//   per 32-channel stage the 3 x (R+2) x (DIM+2) halo of the tile is staged once (raw buffer loads, a voxel outside the volume
//   reads zeros through an out-of-range offset; global -> registers under the previous stage's MFMAs -> LDS, two buffers, one
//   barrier per stage); a halo voxel takes 144 bytes of LDS (128 + 16 pad: every 8 lanes of a ds_read_b128 cover the 32 banks);
//   the 54 (tap, 16-channel group) k steps of a stage are dealt to the waves (step = wave + 8 j: 7, 7, 7, 7, 7, 7, 6, 6); a k step
//   = 2 weight fragments (global; the whole next stage's seven steps in flight in a register ring) + 4 ds_read_b128 + 32 MFMAs;
//   the waves' partial sums meet in LDS in wave order (deterministic) and the epilogue runs in the same launch, its bias and
//   skip-tensor loads issued ahead of the last stage's MFMAs.
template <int DIM>
struct Halo64 {
    static constexpr int R = 64 / DIM;                // rows of a tile
    static constexpr int HX = DIM + 2, HY = R + 2;
    static constexpr int HV = 3 * HY * HX;            // halo voxels: 300 (dim 8) / 324 (dim 16)
    static constexpr int PITCH = 144;
    static constexpr int HB = HV * PITCH;
    static constexpr int PIECES = HV * 8;             // 16-byte pieces of a stage
    static constexpr int HP = (PIECES + 511) / 512;
    static constexpr int MT_OFF = (16 / DIM) * HX * PITCH;   // LDS bytes between consecutive voxel tiles (16 voxels = 16 / DIM rows)
    static constexpr int LDS_BYTES = 2 * HB > 65536 ? 2 * HB : 65536;
};

template <int DIM>
__global__ __launch_bounds__(512) void conv3d_k3_halo64_kernel(ConvArgs a) {
    using G = Halo64<DIM>;
    constexpr int HX = G::HX, HY = G::HY, PITCH = G::PITCH, HP = G::HP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vl = lane & 15, h = lane >> 4;
    const int stages = a.cin >> 5;
    const int nt0 = blockIdx.y * 2;
    constexpr unsigned OOB = 0x80000000u;
    int t = blockIdx.x;
    const int ty = t % (DIM / G::R); t /= (DIM / G::R);
    const int z = t % DIM;
    const int b = t / DIM;
    const int y0 = ty * G::R;
    const long long vid0 = (((long long)b * DIM + z) * DIM + y0) * DIM;       // the tile's 64 voxels are consecutive in memory

    // this thread's halo pieces: byte offset in the input tensor (out of range for a voxel outside the volume) - fixed over the stages
    unsigned goff[HP];
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        const int p = tid + 512 * j;
        const int hv = p >> 3, q = p & 7;
        const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
        const int gz = z + hz - 1, gy = y0 + hy - 1, gx = hx - 1;
        const bool ok = p < G::PIECES && (unsigned)gz < (unsigned)DIM && (unsigned)gy < (unsigned)DIM && (unsigned)gx < (unsigned)DIM;
        goff[j] = ok ? (unsigned)(((((long long)b * DIM + gz) * DIM + gy) * DIM + gx) * a.cin_pad * 4 + q * 16) : OOB;
    }
    const int lpiece = (tid >> 3) * PITCH + (tid & 7) * 16;                   // piece j: + j * 64 * PITCH
    const unsigned in_bytes = (unsigned)(a.total_vox * a.cin_pad * 4);        // launcher: fits 31 bits
    const unsigned w_bytes = (unsigned)(27 * (a.cin >> 4) * a.nts * 1024);
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)in_bytes, 0x00020000);
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpack), 0, (int)w_bytes, 0x00020000);
    const int woff = lane * 16;
    // operand reads: lane (vl, h) = voxel vl of the tile's first 16, channels 4 h .. 4 h + 3, at the tap (-1, -1, -1) corner
    const int lrow = DIM == 8 ? (vl >> 3) : 0, lx = DIM == 8 ? (vl & 7) : vl;
    const int lbase = (lrow * HX + lx) * PITCH + h * 16;

    f32x4 st[HP];
    auto load_stage = [&](int sg) {
#pragma unroll
        for (int j = 0; j < HP; ++j) st[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)goff[j], sg * 128, 0));
    };
    auto commit_stage = [&](unsigned char* buf) {
#pragma unroll
        for (int j = 0; j < HP; ++j)
            if (tid + 512 * j < G::PIECES) *reinterpret_cast<f32x4*>(buf + lpiece + j * 64 * PITCH) = st[j];
    };
    f32x4 wr[7][2];
    auto load_w = [&](int j, int sg) {                 // slot j of stage sg: step wave + 8 j = (tap, group); slot 6 of waves 6, 7: dead, never used
        const int step = wave + 8 * j;
        const int tap = step >> 1, cg = sg * 2 + (step & 1);
        const int wso = step < 54 ? ((cg * 27 + tap) * a.nts + nt0) * 1024 : 0;
        wr[j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, wso, 0));
        wr[j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, wso + 1024, 0));
    };
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the epilogue's operands (wave w finishes fragment w = (cout tile w >> 2, voxel tile w & 3))
    const long long per_b = (long long)DIM * DIM * DIM;
    const int co0 = (nt0 + (wave >> 2)) * 16 + 4 * h;
    const long long ooff = (vid0 + (wave & 3) * 16 + vl) * a.cout + co0;
    f32x4 ebias = (f32x4){0.f, 0.f, 0.f, 0.f}, eres = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_stage(0);
#pragma unroll
    for (int j = 0; j < 7; ++j) load_w(j, 0);
    for (int sg = 0; sg < stages; ++sg) {
        unsigned char* buf = lds + (sg & 1) * G::HB;
        commit_stage(buf);
        __syncthreads();          // the stage is in LDS; every wave has left the stage before the previous one (the buffer written next)
        const bool more = sg + 1 < stages;
        if (more) {
            load_stage(sg + 1);
        } else {
            ebias = *reinterpret_cast<const f32x4*>(a.bpack + co0);
            if (a.res) eres = *reinterpret_cast<const f32x4*>(a.res + ooff);
        }
        f32x4 xf[2][4];
        auto read_x = [&](f32x4 (&x)[4], int j) {
            int step = wave + 8 * j;
            step = step < 54 ? step : 53;
            const int tap = step >> 1;
            const int toff = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * PITCH + (step & 1) * 64;     // wave-uniform
            const unsigned char* p = buf + lbase + toff;
#pragma unroll
            for (int n = 0; n < 4; ++n) x[n] = *reinterpret_cast<const f32x4*>(p + n * G::MT_OFF);
        };
        read_x(xf[0], 0);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (j + 1 < 7) read_x(xf[(j + 1) & 1], j + 1);
            if (wave + 8 * j < 54) {                   // uniform
                const f32x4(&x)[4] = xf[j & 1];
                const float wv[2][4] = {{wr[j][0].x, wr[j][0].y, wr[j][0].z, wr[j][0].w}, {wr[j][1].x, wr[j][1].y, wr[j][1].z, wr[j][1].w}};
#pragma unroll
                for (int c = 0; c < 4; ++c)            // component outer: consecutive MFMAs go to different accumulators
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const float xv[4] = {x[n].x, x[n].y, x[n].z, x[n].w};
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[m][c], xv[c], acc[m][n], 0, 0, 0);
                    }
            }
            if (more) load_w(j, sg + 1);               // the slot's registers are free: next stage's weights, seven steps ahead
        }
    }
    __syncthreads();              // every wave is done with the halo buffers: the partial sums go on top of them
    f32x4(*red)[8][64] = reinterpret_cast<f32x4(*)[8][64]>(lds);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) red[wave][m * 4 + n][lane] = acc[m][n];
    __syncthreads();
    f32x4 v = red[0][wave][lane];                      // partials added in wave order
#pragma unroll
    for (int w2 = 1; w2 < 8; ++w2) v += red[w2][wave][lane];
    v += ebias;
    if ((a.flags & SE_EPI_RES_PRE_RELU) && a.res) v += eres;
    if (a.flags & SE_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if ((a.flags & SE_EPI_RES_POST_RELU) && a.res) v += eres;
    *reinterpret_cast<f32x4*>(a.out + ooff) = v;
    (void)per_b;
}

// thread = (voxel, cout quad): sum the split partials in order, then the usual epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs a, const float* __restrict__ ws, int splits) {
    const int cq = a.cout >> 2;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.total_vox * cq) return;
    const long long vid = t / cq;
    const int q = (int)(t - vid * cq);
    // all partials requested before the first add (splits is 3, 9 or 27; the sum keeps its fixed order): the kernel is pure load
    // latency, one dependent round trip per split otherwise
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* p0 = ws + vid * a.cout + q * 4;
    const size_t sstride = (size_t)a.total_vox * a.cout;
    for (int s0 = 0; s0 < splits; s0 += 9) {
        f32x4 part[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) part[u] = (s0 + u < splits) ? *reinterpret_cast<const f32x4*>(p0 + (size_t)(s0 + u) * sstride) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 9; ++u) v += part[u];
    }
    const long long per_b = (long long)a.dim * a.dim * a.dim;
    const int b = (int)(vid / per_b);
    conv_epilogue(a, v, b, vid - (long long)b * per_b, per_b, q * 4);
}

// ------------------------------------------------------------------------------------------------
// Fused tail of V2V: back_layers.1, back_layers.2 (1x1x1 32->32 + BN + ReLU each) and output_layer (1x1x1 32->J, planar
// store) in ONE pass over the 64^3 activations (network/v2v.py:155-161,167-169): 128 B read + 4*J B written per voxel
// instead of three read+write round trips.  No data movement is needed between the layers: with weights as the MFMA
// A operand, the D fragment of cout tile nt (lane = voxel, 4 consecutive couts per lane) IS the B-operand fragment of
// channel group cg = nt of the next layer.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pointwise_chain3_kernel(const float* __restrict__ in, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, const float* __restrict__ w2,
                                                               const float* __restrict__ b2, const float* __restrict__ w3,
                                                               const float* __restrict__ b3, float* __restrict__ out,
                                                               long long total_vox, long long vox_per_b, int cout3) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int vl = lane & 15, h = lane >> 4;
    const f32x4* W1 = reinterpret_cast<const f32x4*>(w1) + lane;   // blocks [cg][tap=0][nt][lane]: index (cg*2 + nt)*64
    const f32x4* W2 = reinterpret_cast<const f32x4*>(w2) + lane;
    const f32x4* W3 = reinterpret_cast<const f32x4*>(w3) + lane;   // nts = 1: index cg*64
    // the next tile's activations are requested before this tile's MFMA chain (three dependent layers = ~40 MFMA latencies)
    const long long tstride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + wave;
    f32x4 xn[2];
    {
        const long long v0 = tile * 16 + vl;
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
            xn[cg] = (tile * 16 < total_vox && v0 < total_vox) ? *reinterpret_cast<const f32x4*>(in + v0 * 32 + cg * 16 + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (; tile * 16 < total_vox; tile += tstride) {
        const long long vid = tile * 16 + vl;
        const bool ok = vid < total_vox;
        f32x4 x[2], y[2];
        x[0] = xn[0]; x[1] = xn[1];
        {
            const long long vnext = (tile + tstride) * 16 + vl;
            const bool okn = vnext < total_vox;
#pragma unroll
            for (int cg = 0; cg < 2; ++cg)
                xn[cg] = okn ? *reinterpret_cast<const f32x4*>(in + vnext * 32 + cg * 16 + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#define SE_PW_LAYER(W, B, X, Y)                                                                 \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                          \
        f32x4 acc = *reinterpret_cast<const f32x4*>(B + nt * 16 + 4 * h);                       \
        _Pragma("unroll") for (int cg = 0; cg < 2; ++cg) {                                      \
            const f32x4 wf = W[(cg * 2 + nt) * 64];                                             \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.x, X[cg].x, acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.y, X[cg].y, acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.z, X[cg].z, acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.w, X[cg].w, acc, 0, 0, 0);            \
        }                                                                                       \
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); \
        Y[nt] = acc;                                                                            \
    }
        SE_PW_LAYER(W1, b1, x, y)
        SE_PW_LAYER(W2, b2, y, x)
#undef SE_PW_LAYER
        f32x4 acc = *reinterpret_cast<const f32x4*>(b3 + 4 * h);
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const f32x4 wf = W3[cg * 64];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.x, x[cg].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.y, x[cg].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.z, x[cg].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.w, x[cg].w, acc, 0, 0, 0);
        }
        if (ok) {
            const long long b = vid / vox_per_b, n = vid - b * vox_per_b;
            float* o = out + (b * cout3 + 4 * h) * vox_per_b + n;
            const float vv[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * h + r < cout3) o[(long long)r * vox_per_b] = vv[r];
        }
    }
}

// The same chain with pass 1 of the soft-argmax (reference utils/op.py:83-96) folded in: the workgroup (chunk s, sample b) owns the
// voxels [s * chunk, (s+1) * chunk) of its sample - the chunking of softargmax_partial_kernel - keeps a running
// (max, sum exp, sum exp * coord) per lane and joint while the logits are still in registers (online softmax: a new maximum
// rescales the sums), folds the lanes in a fixed order and writes the SE_SA_PART record softargmax_finish_kernel reads (se_sa_splits(batch * cout3) chunks per row).  The
// logits are stored as before (forward() returns the softmaxed volumes, pass 2 reads them) but are not read back for pass 1.
constexpr int PW_SA_WAVES = 16;      // waves per workgroup of pointwise_chain3_sa_kernel (one workgroup per CU: 4 waves per SIMD)
__global__ __launch_bounds__(PW_SA_WAVES * 64) void pointwise_chain3_sa_kernel(const float* __restrict__ in, const float* __restrict__ w1,
                                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                                  const float* __restrict__ b2, const float* __restrict__ w3,
                                                                  const float* __restrict__ b3, float* __restrict__ out,
                                                                  const float* __restrict__ coord, float* __restrict__ scratch,
                                                                  int vox_per_b, int chunk, int cout3, int splits, int in_quad) {
    extern __shared__ __attribute__((aligned(16))) float pw_sa_lds[];
    float (*red)[64][20] = reinterpret_cast<float (*)[64][20]>(pw_sa_lds);        // [PW_SA_WAVES][64][20]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int vl = lane & 15, h = lane >> 4;
    const int s = blockIdx.x, b = blockIdx.y;
    const int v0 = s * chunk, v1 = min(v0 + chunk, vox_per_b);
    const f32x4* W1 = reinterpret_cast<const f32x4*>(w1) + lane;
    const f32x4* W2 = reinterpret_cast<const f32x4*>(w2) + lane;
    const f32x4* W3 = reinterpret_cast<const f32x4*>(w3) + lane;
    const float* inb = in + (long long)b * vox_per_b * 32;
    // B fragment of channel group cg: this lane's 4 channels cg * 16 + 4 h ..: channels-last record n, or (round 5, SE_IN_QUAD) record n
    // of quad plane cg * 4 + h - 16 lanes read 256 contiguous bytes either way
    const long long x_str = in_quad ? 4 : 32, x_cg = in_quad ? 4LL * vox_per_b * 4 : 16;
    const float* inl = inb + (in_quad ? (long long)h * vox_per_b * 4 : 4 * h);
    float m[4], l[4], sx[4], sy[4], sz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = sx[r] = sy[r] = sz[r] = 0.f; }
    f32x4 xn[2];
    {
        const int n0 = v0 + wave * 16 + vl;
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
            xn[cg] = n0 < v1 ? *reinterpret_cast<const f32x4*>(inl + n0 * x_str + cg * x_cg) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int t0 = v0 + wave * 16; t0 < v1; t0 += PW_SA_WAVES * 16) {
        const int n = t0 + vl;
        const bool ok = n < v1;
        f32x4 x[2], y[2];
        x[0] = xn[0]; x[1] = xn[1];
        {
            const int nn = n + PW_SA_WAVES * 16;
#pragma unroll
            for (int cg = 0; cg < 2; ++cg)
                xn[cg] = nn < v1 ? *reinterpret_cast<const f32x4*>(inl + nn * x_str + cg * x_cg) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (ok) { cx = coord[(size_t)n * 3]; cy = coord[(size_t)n * 3 + 1]; cz = coord[(size_t)n * 3 + 2]; }
#define SE_PW_LAYER(W, B, X, Y)                                                                 \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                          \
        f32x4 acc = *reinterpret_cast<const f32x4*>(B + nt * 16 + 4 * h);                       \
        _Pragma("unroll") for (int cg = 0; cg < 2; ++cg) {                                      \
            const f32x4 wf = W[(cg * 2 + nt) * 64];                                             \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.x, X[cg].x, acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.y, X[cg].y, acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.z, X[cg].z, acc, 0, 0, 0);            \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.w, X[cg].w, acc, 0, 0, 0);            \
        }                                                                                       \
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); \
        Y[nt] = acc;                                                                            \
    }
        SE_PW_LAYER(W1, b1, x, y)
        SE_PW_LAYER(W2, b2, y, x)
#undef SE_PW_LAYER
        f32x4 acc = *reinterpret_cast<const f32x4*>(b3 + 4 * h);
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const f32x4 wf = W3[cg * 64];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.x, x[cg].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.y, x[cg].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.z, x[cg].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf.w, x[cg].w, acc, 0, 0, 0);
        }
        if (ok) {
            float* o = out + ((long long)b * cout3 + 4 * h) * vox_per_b + n;
            const float vv[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * h + r < cout3) {
                    o[(long long)r * vox_per_b] = vv[r];
                    if (vv[r] > m[r]) {      // new running maximum (rare after the first tiles): rescale the sums; the first element
                        const float sc = __expf(m[r] - vv[r]);                            // meets m = -inf: exp(-inf) = 0 clears them
                        l[r] *= sc; sx[r] *= sc; sy[r] *= sc; sz[r] *= sc;
                        m[r] = vv[r];
                    }
                    const float e = __expf(vv[r] - m[r]);
                    l[r] += e;
                    sx[r] = fmaf(e, cx, sx[r]);
                    sy[r] = fmaf(e, cy, sy[r]);
                    sz[r] = fmaf(e, cz, sz[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float* e = &red[wave][lane][r * 5];
        e[0] = m[r]; e[1] = l[r]; e[2] = sx[r]; e[3] = sy[r]; e[4] = sz[r];
    }
    __syncthreads();
    const int j = threadIdx.x;
    if (j < cout3) {          // joint j: the 16 x PW_SA_WAVES (wave, voxel lane) partials of k lane j / 4, slot j % 4, folded in a fixed order
        const int hh = j >> 2, rr = (j & 3) * 5;
        float M = -INFINITY;
        for (int w = 0; w < PW_SA_WAVES; ++w)
            for (int q = 0; q < 16; ++q) M = fmaxf(M, red[w][hh * 16 + q][rr]);
        float L = 0.f, SX = 0.f, SY = 0.f, SZ = 0.f;
        for (int w = 0; w < PW_SA_WAVES; ++w)
            for (int q = 0; q < 16; ++q) {
                const float* e = &red[w][hh * 16 + q][rr];
                const float f = (e[0] == -INFINITY) ? 0.f : __expf(e[0] - M);     // a lane that saw no voxel contributes nothing
                L += e[1] * f; SX += e[2] * f; SY += e[3] * f; SZ += e[4] * f;
            }
        float* p = scratch + (((size_t)b * cout3 + j) * splits + s) * SE_SA_PART;
        p[0] = M; p[1] = L; p[2] = SX; p[3] = SY; p[4] = SZ;
    }
}

// ------------------------------------------------------------------------------------------------
// max-pool 2x2x2 stride 2, channels-last, 16 B per lane
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       long long total /* B*od^3*cq */, int od, int cq) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int q = (int)(t % cq);
    long long r = t / cq;
    const int x = (int)(r % od); r /= od;
    const int y = (int)(r % od); r /= od;
    const int z = (int)(r % od); r /= od;
    const long long b = r;
    const int id = od * 2;
    const int C = cq * 4;
    const float* p = in + ((((b * id + 2 * z) * id + 2 * y) * id + 2 * x) * (long long)C) + q * 4;
    f32x4 m = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const long long off = (((long long)(k >> 2) * id + ((k >> 1) & 1)) * id + (k & 1)) * C;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + off);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    *reinterpret_cast<f32x4*>(out + t * 4) = m;
}

// same, octet-planar input [B][C/8][id^3][8]: thread = (voxel, channel quad); the two quads of an octet are adjacent threads, so a
// wave reads 32 contiguous bytes per input voxel and octet and still writes 16-byte channels-last pieces
__global__ __launch_bounds__(256) void maxpool2_octin_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             long long total /* B*od^3*cq */, int od, int cq) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int q = (int)(t % cq);
    long long r = t / cq;
    const int x = (int)(r % od); r /= od;
    const int y = (int)(r % od); r /= od;
    const int z = (int)(r % od); r /= od;
    const long long b = r;
    const int id = od * 2;
    const long long vox = (long long)id * id * id;
    const float* p = in + ((b * (cq >> 1) + (q >> 1)) * vox + ((long long)(2 * z) * id + 2 * y) * id + 2 * x) * 8 + (q & 1) * 4;
    f32x4 m = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const long long off = (((long long)(k >> 2) * id + ((k >> 1) & 1)) * id + (k & 1)) * 8;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + off);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    *reinterpret_cast<f32x4*>(out + t * 4) = m;
}

template <int KS>
int launch_direct(const ConvArgs& a, hipStream_t s) {
    // cout tiles per workgroup: 2 when the layer has an even number of 16-wide cout tiles
    const long long vox_per_wg = 4 * 4 * 16;  // 4 waves x M_T=4 x 16
    const unsigned gx = (unsigned)((a.total_vox + vox_per_wg - 1) / vox_per_wg);
    if (a.nts % 2 == 0) {
        hipLaunchKernelGGL((conv3d_direct_kernel<KS, 4, 2, false>), dim3(gx, a.nts / 2), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((conv3d_direct_kernel<KS, 4, 1, false>), dim3(gx, a.nts), dim3(256), 0, s, a);
    }
    SE_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int se_abi_version(void) { return 22; }

// Mirrors the dispatch of se_conv3d_f32 -> se_conv3d_tiled_try -> se_conv3d_wino_try / se_conv3d_k7_wino_try for the
// default channels-last call (no SE_EPI_OUT_PLANAR / SE_EPI_RES_POST_RELU flags).
extern "C" int se_conv3d_f32_algo(int dim, int cin, int cout, int ksize) {
    if (ksize == 3 && se_wino2d_shape_ok(dim, cin, cout)) return 2;
    if (ksize == 3 && dim >= 16 && (dim & 7) == 0 && (cout & 31) == 0 && (cin & 15) == 0) return 1;
    if (ksize == 7 && dim >= 16 && (dim & 7) == 0 && cout == 16) return 7;
    return 0;
}

// Which kernel a launch of `batch` samples with these layout flags (SE_IN_OCTET ...) runs on: se_conv3d_f32_algo's value, except
//   3 = the F(4,3) x F(4,3) ping-pong kernel (conv3d_wino44pp.hip; a member of the 2-D Winograd family: same layouts, flags and fused
//       forms as algo 2) - it declines a channels-last input with >= 32 channels, which stays on algo 2 (se_conv3d_wino44pp_takes);
//   0 for a 2-D Winograd shape with <= 4096 voxels in the batch when the call asks for no octet-planar / pooled / fused form.
// Mirrors se_conv3d_tiled_try / se_conv3d_wino2d_try for the WHOLE batch (a batch above 32 is cut into slices that keep this decision).
bool se_conv3d_wino44pp_shape(int batch, int dim, int cout);      // conv3d_wino44pp.hip
bool se_conv3d_wino44pp_layout_ok(int cin, int flags);
extern "C" int se_conv3d_f32_variant(int batch, int dim, int cin, int cout, int ksize, int flags) {
    const int algo = se_conv3d_f32_algo(dim, cin, cout, ksize);
    const bool forms = flags & (SE_LAYOUT_OCTET_BITS | SE_LAYOUT_QUAD_BITS | SE_EPI_SKIPCONV16);
    if (algo == 2 && !forms && se_conv3d_small_volume(batch, dim)) return 0;       // a plain channels-last call runs the in-workgroup split-K form
    if (algo == 2 && g_variant != 64 && se_conv3d_wino44pp_shape(batch, dim, cout) && se_conv3d_wino44pp_layout_ok(cin, flags)) return 3;
    return algo;
}

static long long packed_elems_a(int cout, int cin_pad, int ksize, int transposed) {
    const long long taps = transposed ? 8 : (long long)ksize * ksize * ksize;
    return taps * (cin_pad / 16) * (round_up16(cout) / 16) * 256;
}

extern "C" long long se_conv3d_packed_elems(int cout, int cin_pad, int ksize, int transposed) {
    long long n = packed_elems_a(cout, cin_pad, ksize, transposed);
    if (!transposed && ksize == 7) n += (long long)(cin_pad / 4) * SE_K7_GROUPS * (round_up16(cout) / 16) * 256;
    if (!transposed && ksize == 7 && cout <= 16) n += (long long)(cin_pad / 4) * SE_K7W_CHUNK_FLOATS;
    if (!transposed && ksize == 7 && cout <= 16) n += (long long)((cin_pad + 2) / 3) * (SE_K7F_CHUNK_FLOATS + SE_K7H_CHUNK_FLOATS);     // sections F, H (last)
    if (!transposed && ksize == 3 && cout % 32 == 0)
        n += (long long)(cin_pad / 16) * (cout / 32) * (SE_WINO_CHUNK_FLOATS + SE_WINO43_CHUNK_FLOATS);
    if (!transposed && ksize == 3 && cout % 32 == 0) n += (long long)(cin_pad / 8) * (cout / 32) * SE_WINO2D_CHUNK_FLOATS;   // section G
    if (!transposed && ksize == 3 && cout % 32 == 0) n += (long long)(cin_pad / 4) * (cout / 32) * SE_WINO44_CHUNK_FLOATS;   // section I (last)
    return n;
}

// conv3d_wino2d.hip
int se_conv3d_pack_wino2d(const float* w, const float* gamma, const float* var, float eps, float* out, int cout, int cin,
                          int cin_pad, hipStream_t s);
// conv3d_wino44.hip
int se_conv3d_pack_wino44(const float* w, const float* gamma, const float* var, float eps, float* out, int cout, int cin,
                          int cin_pad, hipStream_t s);

extern "C" int se_conv3d_pack_f32(const float* w, const float* b, const float* gamma, const float* beta,
                                  const float* mean, const float* var, float eps, float* wpack, float* bpack,
                                  int cout, int cin, int cin_pad, int ksize, int transposed, void* stream) {
    if (cout <= 0 || cin <= 0 || cin_pad < cin || (cin_pad & 15)) return SE_ERR_BAD_ARG;
    if (transposed ? (ksize != 2) : (ksize != 1 && ksize != 3 && ksize != 7)) return SE_ERR_BAD_ARG;
    if ((gamma != nullptr) != (beta != nullptr) || (gamma != nullptr) != (mean != nullptr) ||
        (gamma != nullptr) != (var != nullptr))
        return SE_ERR_BAD_ARG;
    const int taps = ksize * ksize * ksize;
    const long long total = se_conv3d_packed_elems(cout, cin_pad, ksize, transposed);
    const long long total_a = packed_elems_a(cout, cin_pad, ksize, transposed);
    const long long threads = total > round_up16(cout) ? total : round_up16(cout);
    long long total_main = total;
    if (!transposed && ksize == 7 && cout <= 16) total_main -= (long long)((cin_pad + 2) / 3) * (SE_K7F_CHUNK_FLOATS + SE_K7H_CHUNK_FLOATS);
    const long long n_g = (!transposed && ksize == 3 && cout % 32 == 0) ? (long long)(cin_pad / 8) * (cout / 32) * SE_WINO2D_CHUNK_FLOATS : 0;
    const long long n_i = n_g ? (long long)(cin_pad / 4) * (cout / 32) * SE_WINO44_CHUNK_FLOATS : 0;
    total_main -= n_g + n_i;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, se_stream(stream), w, b,
                       gamma, beta, mean, var, eps, wpack, bpack, cout, cin, cin_pad, taps, transposed, total_a, total_main);
    SE_CHECK_LAUNCH();
    if (n_g) {
        const int rc = se_conv3d_pack_wino2d(w, gamma, var, eps, wpack + total_main, cout, cin, cin_pad, se_stream(stream));
        if (rc) return rc;
        return se_conv3d_pack_wino44(w, gamma, var, eps, wpack + total_main + n_g, cout, cin, cin_pad, se_stream(stream));
    }
    if (total_main != total) {
        const long long nf = (long long)((cin_pad + 2) / 3) * SE_K7F_CHUNK_FLOATS, nh = total - total_main - nf;
        hipLaunchKernelGGL(pack_k7f_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, se_stream(stream), w, gamma, var, eps,
                           wpack + total_main, cout, cin, nf);
        SE_CHECK_LAUNCH();
        hipLaunchKernelGGL(pack_k7h_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, se_stream(stream), w, gamma, var, eps,
                           wpack + total_main + nf, cout, cin, nh);
        SE_CHECK_LAUNCH();
    }
    return 0;
}

// implemented in conv3d_tiled.hip; returns 1 if it took the launch, 0 if the shape is not covered, <0 / hipError on failure
int se_conv3d_tiled_try(const ConvArgs& a, int batch, int ksize, hipStream_t s);

#define g_variant_direct (g_variant == 18 ? 1 : g_variant == 19 ? 2 : 0)   // se_debug_set_variant(18): A/B, grid-level split-K for every small level; (19): in-workgroup split-K for every small level

static int conv3d_f32_impl(const float* in, const float* wpack, const float* bpack, const float* residual, float* out,
                           float* pool_out, const float* skip_w, int batch, int dim, int cin, int cin_pad, int cout, int ksize, int flags,
                           float* workspace, long long workspace_elems, void* stream) {
    if (batch <= 0 || dim <= 0 || cin <= 0 || cin_pad < cin || (cin_pad & 15) || cout <= 0) return SE_ERR_BAD_ARG;
    if (ksize != 1 && ksize != 3 && ksize != 7) return SE_ERR_BAD_ARG;
    const bool planar = flags & SE_EPI_OUT_PLANAR;
    if (!planar && (cout & 15)) return SE_ERR_BAD_ARG;
    if (planar && (flags & (SE_EPI_RES_PRE_RELU | SE_EPI_RES_POST_RELU))) return SE_ERR_BAD_ARG;
    if ((flags & SE_EPI_RES_PRE_RELU) && (flags & SE_EPI_RES_POST_RELU)) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    ConvArgs a;
    a.in = in; a.wpack = wpack; a.bpack = bpack; a.res = residual; a.out = out;
    a.total_vox = (long long)batch * dim * dim * dim;
    a.dim = dim; a.cin = cin; a.cin_pad = cin_pad; a.cout = cout; a.nts = round_up16(cout) / 16; a.flags = flags;
    a.wpack_b = wpack + packed_elems_a(cout, cin_pad, ksize, 0);
    a.wpack_d = nullptr;
    if (ksize == 7 && cout <= 16)
        a.wpack_d = a.wpack_b + (long long)(cin_pad / 4) * SE_K7_GROUPS * (round_up16(cout) / 16) * 256;
    a.wpack_f = nullptr;
    if (ksize == 7 && cout <= 16) a.wpack_f = a.wpack_d + (long long)(cin_pad / 4) * SE_K7W_CHUNK_FLOATS;
    a.wpack_h = nullptr;
    if (ksize == 7 && cout <= 16) a.wpack_h = a.wpack_f + (long long)((cin_pad + 2) / 3) * SE_K7F_CHUNK_FLOATS;
    a.wpack_e = nullptr;
    if (ksize == 3 && cout % 32 == 0) a.wpack_e = a.wpack_b + (long long)(cin_pad / 16) * (cout / 32) * SE_WINO_CHUNK_FLOATS;
    a.wpack_g = nullptr;
    if (ksize == 3 && cout % 32 == 0) a.wpack_g = a.wpack_e + (long long)(cin_pad / 16) * (cout / 32) * SE_WINO43_CHUNK_FLOATS;
    a.wpack_i = nullptr;
    if (ksize == 3 && cout % 32 == 0) a.wpack_i = a.wpack_g + (long long)(cin_pad / 8) * (cout / 32) * SE_WINO2D_CHUNK_FLOATS;
    a.pool_out = pool_out;
    a.skip_w = skip_w;
    a.ws = workspace;
    a.ws_elems = workspace ? workspace_elems : 0;
    if (!skip_w && (flags & SE_EPI_SKIPCONV16)) return SE_ERR_BAD_ARG;
    if (skip_w && (!residual || se_conv3d_f32_algo(dim, cin, cout, ksize) != 2 || cin_pad != cin)) return SE_ERR_BAD_ARG;
    if (pool_out && ((dim & 1) || se_conv3d_f32_algo(dim, cin, cout, ksize) != 2 || cin_pad != cin)) return SE_ERR_BAD_ARG;   // only the 2-D Winograd kernel pools
    if (ksize == 3 && (cout % 32)) a.wpack_b = nullptr;
    if (ksize == 1) a.wpack_b = nullptr;
    if ((flags & SE_IN_PLANAR3) && ksize != 7) return SE_ERR_BAD_ARG;
    if ((flags & (SE_LAYOUT_OCTET_BITS | SE_LAYOUT_QUAD_BITS)) && se_conv3d_f32_algo(dim, cin, cout, ksize) != 2) return SE_ERR_BAD_ARG;
    if ((flags & SE_LAYOUT_OCTET_BITS) && (flags & SE_LAYOUT_QUAD_BITS)) return SE_ERR_BAD_ARG;
    // quad-planar tensors exist in the F(4,3) x F(4,3) kernel only - and as the OUTPUT of a channels-last launch of the F(4,3) x F(2,3) one
    if ((flags & SE_LAYOUT_QUAD_BITS) && se_conv3d_f32_variant(batch, dim, cin, cout, ksize, flags) != 3 &&
        (flags & SE_LAYOUT_QUAD_BITS) != SE_OUT_QUAD)
        return SE_ERR_BAD_ARG;
    const int took = se_conv3d_tiled_try(a, batch, ksize, s);
    if (took != SE_TILED_NOT_TAKEN) return took;
    if (flags & (SE_LAYOUT_OCTET_BITS | SE_LAYOUT_QUAD_BITS)) return SE_ERR_BAD_ARG;   // only the 2-D Winograd kernels know the planar forms
    if (flags & SE_IN_PLANAR3) return SE_ERR_BAD_ARG;
    // small volumes with wide channels: split the taps over grid.z when the plain launch would not fill the chip
    if (ksize == 3 && !planar && a.nts % 2 == 0 && a.total_vox <= 8192 && (a.total_vox * cin_pad * 4LL) < (1LL << 31) && g_variant_direct != 1 &&
        (a.total_vox >= 2048 || g_variant_direct == 2)) {
        // 8^3-sized levels: in-workgroup split-K, single launch (round 2: 0.063 vs 0.077 ms for grid split-K + reduce at B = 8)
        const long long tiles = (a.total_vox + 15) / 16;
#ifndef SE_HALO64
#define SE_HALO64 1
#endif
        if (SE_HALO64 && a.total_vox >= 2048 && (dim == 8 || dim == 16) && (a.cin & 31) == 0 && !(flags & SE_EPI_OUT_PLANAR)) {   // 64-voxel tiles, activations through LDS
            const dim3 grid((unsigned)(a.total_vox / 64), a.nts / 2);
            if (dim == 8) {
                SE_ENSURE_LDS(conv3d_k3_halo64_kernel<8>, Halo64<8>::LDS_BYTES);
                hipLaunchKernelGGL((conv3d_k3_halo64_kernel<8>), grid, dim3(512), Halo64<8>::LDS_BYTES, s, a);
            } else {
                SE_ENSURE_LDS(conv3d_k3_halo64_kernel<16>, Halo64<16>::LDS_BYTES);
                hipLaunchKernelGGL((conv3d_k3_halo64_kernel<16>), grid, dim3(512), Halo64<16>::LDS_BYTES, s, a);
            }
        } else if (a.total_vox >= 2048)
            hipLaunchKernelGGL((conv3d_k3_wavesplit_kernel<2, 4>), dim3((unsigned)((tiles + 1) / 2), a.nts / 2), dim3(256), 0, s, a);
        else if (a.total_vox >= 256)
            hipLaunchKernelGGL((conv3d_k3_wavesplit_kernel<1, 8>), dim3((unsigned)tiles, a.nts / 2), dim3(512), 0, s, a);
        else
            hipLaunchKernelGGL((conv3d_k3_wavesplit_kernel<1, 16>), dim3((unsigned)tiles, a.nts / 2), dim3(1024), 0, s, a);
        SE_CHECK_LAUNCH();
        return 0;
    }
    if (ksize == 3 && !planar && workspace && a.nts % 2 == 0) {
        const long long m_blocks = (a.total_vox + 63) / 64;
        const long long wgs = m_blocks * (a.nts / 2);
        if (wgs < 1024) {
            int splits = 27;                                  // taps per split: 1, 3, 9 (or no split)
            while (splits > 1 && (wgs * (splits / 3) >= 2048 || (long long)splits * a.total_vox * cout > workspace_elems))
                splits /= 3;
            if (splits > 1) {
                hipLaunchKernelGGL((conv3d_k3_splitk_kernel<2>), dim3((unsigned)m_blocks, a.nts / 2, splits), dim3(256), 0, s,
                                   a, workspace, 27 / splits);
                SE_CHECK_LAUNCH();
                const long long threads = a.total_vox * (cout / 4);
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a,
                                   workspace, splits);
                SE_CHECK_LAUNCH();
                return 0;
            }
        }
    }
    switch (ksize) {
        case 1: return launch_direct<1>(a, s);
        case 3: return launch_direct<3>(a, s);
        default: return launch_direct<7>(a, s);
    }
}

extern "C" int se_conv3d_f32(const float* in, const float* wpack, const float* bpack, const float* residual,
                             float* out, int batch, int dim, int cin, int cin_pad, int cout, int ksize, int flags,
                             float* workspace, long long workspace_elems, void* stream) {
    return conv3d_f32_impl(in, wpack, bpack, residual, out, nullptr, nullptr, batch, dim, cin, cin_pad, cout, ksize, flags, workspace,
                           workspace_elems, stream);
}

// Same convolution; the kernel also writes max_pool3d(out, kernel 2, stride 2) (channels-last) from its epilogue, so the 2x pool
// of a Res3DBlock output (reference network/v2v.py:104-119: encoder_pool after the front / encoder blocks) does not re-read it.
extern "C" int se_conv3d_pool_f32(const float* in, const float* wpack, const float* bpack, const float* residual,
                                  float* out, float* pool_out, int batch, int dim, int cin, int cin_pad, int cout, int ksize,
                                  int flags, float* workspace, long long workspace_elems, void* stream) {
    if (!pool_out) return SE_ERR_BAD_ARG;
    return conv3d_f32_impl(in, wpack, bpack, residual, out, pool_out, nullptr, batch, dim, cin, cin_pad, cout, ksize, flags, workspace,
                           workspace_elems, stream);
}

extern "C" int se_conv3d_skip16_f32(const float* in, const float* wpack, const float* bpack, const float* skip_in,
                                    const float* skip_w, float* out, int batch, int dim, int cin, int cout, int flags,
                                    void* stream) {
    if (!skip_in || !skip_w) return SE_ERR_BAD_ARG;
    const int lay = flags & ~SE_EPI_RELU;
    if (lay != (SE_IN_OCTET | SE_OUT_OCTET) && lay != (SE_IN_QUAD | SE_OUT_QUAD) && lay != (SE_IN_QUAD | SE_OUT_QUAD | SE_RES_QUAD)) return SE_ERR_BAD_ARG;
    return conv3d_f32_impl(in, wpack, bpack, skip_in, out, nullptr, skip_w, batch, dim, cin, cin, cout, 3,
                           flags | SE_EPI_SKIPCONV16, nullptr, 0, stream);
}

extern "C" int se_deconv3d_k2s2_f32(const float* in, const float* wpack, const float* bpack, const float* residual,
                                    float* out, int batch, int dim, int cin, int cout, int flags, void* stream) {
    if (batch <= 0 || dim <= 0 || cin <= 0 || (cin & 15) || cout <= 0 || (cout & 15)) return SE_ERR_BAD_ARG;
    if (flags & SE_EPI_OUT_PLANAR) return SE_ERR_BAD_ARG;
    if ((flags & SE_EPI_RES_PRE_RELU) && (flags & SE_EPI_RES_POST_RELU)) return SE_ERR_BAD_ARG;
    if (flags & (SE_LAYOUT_OCTET_BITS | SE_IN_QUAD)) return SE_ERR_BAD_ARG;
    const bool outq = flags & SE_OUT_QUAD, resq = flags & SE_RES_QUAD;
    // a quad-planar skip tensor: only with the quad-planar output, added behind the ReLU (what the decoder does)
    if (resq && (!outq || !residual || !(flags & SE_EPI_RES_POST_RELU))) return SE_ERR_BAD_ARG;
    ConvArgs a;
    a.in = in; a.wpack = wpack; a.bpack = bpack; a.res = residual; a.out = out;
    a.total_vox = (long long)batch * dim * dim * dim;
    a.dim = dim; a.cin = cin; a.cin_pad = cin; a.cout = cout; a.nts = cout / 16; a.flags = flags & ~(SE_OUT_QUAD | SE_RES_QUAD);
    a.wpack_b = nullptr;
    a.wpack_d = nullptr;
    a.wpack_e = nullptr;
    a.wpack_f = nullptr;
    a.wpack_g = nullptr;
    a.wpack_h = nullptr;
    a.wpack_i = nullptr;
    a.pool_out = nullptr;
    a.skip_w = nullptr;
    // all eight sub-positions per workgroup where that still fills the chip (measured at B=8: 64->32 from 32^3, 2048 workgroups,
    // 0.241 -> 0.149 ms = 4.05 TB/s; the 128->128 levels, 1..64 workgroups, 0.02 -> 0.14 ms: those keep the grid.z form, whose 8x
    // more workgroups matter more than the input re-reads there)
    const unsigned g2 = (unsigned)((a.total_vox + 127) / 128), g1 = (unsigned)((a.total_vox + 63) / 64);
#define SE_DECONV_CASE(CI, CO, MT, G)                                                                                          \
    if (cin == CI && cout == CO) {                                                                                             \
        hipLaunchKernelGGL((deconv3d_k2s2_kernel<CI / 16, CO / 16, MT, false>), dim3(G), dim3(256), 0, se_stream(stream), a); \
        SE_CHECK_LAUNCH();                                                                                                     \
        return 0;                                                                                                              \
    }
    if (outq) {
        // quad-planar output (SE_OUT_QUAD): the 64 -> 32 layer in front of back_layers.0, volumes whose x rows hold whole 16-voxel tiles
        if ((dim & 15) || !((cin == 64 && cout == 32) || (cin == 128 && cout == 64))) return SE_ERR_BAD_ARG;
        const bool two = g2 >= 4u * (unsigned)se_num_cus();
#define SE_DECONVQ(CG, NT, MT, G)                                                                                                      \
        do {                                                                                                                           \
            if (resq) hipLaunchKernelGGL((deconv3d_k2s2_kernel<CG, NT, MT, true, true>), dim3(G), dim3(256), 0, se_stream(stream), a); \
            else hipLaunchKernelGGL((deconv3d_k2s2_kernel<CG, NT, MT, true, false>), dim3(G), dim3(256), 0, se_stream(stream), a);     \
        } while (0)
        if (cin == 64 && two) SE_DECONVQ(4, 2, 2, g2);
        else if (cin == 64) SE_DECONVQ(4, 2, 1, g1);
        else if (two) SE_DECONVQ(8, 4, 2, g2);
        else SE_DECONVQ(8, 4, 1, g1);
#undef SE_DECONVQ
        SE_CHECK_LAUNCH();
        return 0;
    }
    if (g2 >= 4u * (unsigned)se_num_cus()) {
        SE_DECONV_CASE(64, 32, 2, g2)
        SE_DECONV_CASE(128, 64, 2, g2)
        SE_DECONV_CASE(128, 128, 2, g2)
    } else if (g1 >= 2u * (unsigned)se_num_cus()) {
        SE_DECONV_CASE(64, 32, 1, g1)
        SE_DECONV_CASE(128, 64, 1, g1)
    }
#undef SE_DECONV_CASE
    const long long vox_per_wg = 4 * 4 * 16;
    const unsigned gx = (unsigned)((a.total_vox + vox_per_wg - 1) / vox_per_wg);
    if (a.nts % 2 == 0) {
        hipLaunchKernelGGL((conv3d_direct_kernel<1, 4, 2, true>), dim3(gx, a.nts / 2, 8), dim3(256), 0, se_stream(stream), a);
    } else {
        hipLaunchKernelGGL((conv3d_direct_kernel<1, 4, 1, true>), dim3(gx, a.nts, 8), dim3(256), 0, se_stream(stream), a);
    }
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_pointwise_chain3_f32(const float* in, const float* wpack1, const float* bpack1, const float* wpack2,
                                       const float* bpack2, const float* wpack3, const float* bpack3, float* out,
                                       int batch, int dim, int cout3, void* stream) {
    if (batch <= 0 || dim <= 0 || cout3 <= 0 || cout3 > 16) return SE_ERR_BAD_ARG;
    const long long vox_per_b = (long long)dim * dim * dim;
    const long long total = vox_per_b * batch;
    const long long tiles = (total + 15) / 16;
    const unsigned grid = (unsigned)((tiles + 3) / 4 < 8192 ? (tiles + 3) / 4 : 8192);
    hipLaunchKernelGGL(pointwise_chain3_kernel, dim3(grid), dim3(256), 0, se_stream(stream), in, wpack1, bpack1, wpack2, bpack2,
                       wpack3, bpack3, out, total, vox_per_b, cout3);
    SE_CHECK_LAUNCH();
    return 0;
}

// se_pointwise_chain3_f32 + pass 1 of se_softargmax3d_f32 (mode 1) in one launch: `scratch` receives the partial records
// (se_softargmax3d_scratch_elems(batch * cout3) floats); finish with se_softargmax3d_finish_f32.  `coord` [dim^3][3].
extern "C" int se_pointwise_chain3_softargmax_f32(const float* in, const float* wpack1, const float* bpack1, const float* wpack2,
                                                  const float* bpack2, const float* wpack3, const float* bpack3, float* out,
                                                  const float* coord, float* scratch, int batch, int dim, int cout3,
                                                  int flags, void* stream) {
    if (batch <= 0 || dim <= 0 || cout3 <= 0 || cout3 > 16 || !coord || !scratch || (flags & ~SE_IN_QUAD)) return SE_ERR_BAD_ARG;
    const long long vox_per_b = (long long)dim * dim * dim;
    if (vox_per_b >= (1LL << 31) || (vox_per_b & 3)) return SE_ERR_BAD_ARG;
    const int splits = se_sa_splits(batch * cout3);                                               // as softargmax.hip
    const int chunk = (int)((((vox_per_b + splits - 1) / splits) + 3) & ~3LL);
    if (chunk & 15) return SE_ERR_BAD_ARG;                                                       // whole 16-voxel tiles per chunk
    constexpr int LDS = PW_SA_WAVES * 64 * 20 * 4;
    SE_ENSURE_LDS(pointwise_chain3_sa_kernel, LDS);
    hipLaunchKernelGGL(pointwise_chain3_sa_kernel, dim3(splits, batch), dim3(PW_SA_WAVES * 64), LDS, se_stream(stream), in, wpack1, bpack1,
                       wpack2, bpack2, wpack3, bpack3, out, coord, scratch, (int)vox_per_b, chunk, cout3, splits, (flags & SE_IN_QUAD) ? 1 : 0);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_maxpool3d_2_octin_f32(const float* in, float* out, int batch, int dim, int channels, void* stream) {
    if (batch <= 0 || dim <= 0 || (dim & 1) || channels <= 0 || (channels & 7)) return SE_ERR_BAD_ARG;
    const int od = dim / 2, cq = channels / 4;
    const long long total = (long long)batch * od * od * od * cq;
    hipLaunchKernelGGL(maxpool2_octin_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, se_stream(stream), in, out,
                       total, od, cq);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_maxpool3d_2_f32(const float* in, float* out, int batch, int dim, int channels, void* stream) {
    if (batch <= 0 || dim <= 0 || (dim & 1) || channels <= 0 || (channels & 3)) return SE_ERR_BAD_ARG;
    const int od = dim / 2, cq = channels / 4;
    const long long total = (long long)batch * od * od * od * cq;
    hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, se_stream(stream), in, out,
                       total, od, cq);
    SE_CHECK_LAUNCH();
    return 0;
}
