// bf16-storage V2V kernels (BASELINE config 3): weight packer, the global-memory ("direct") convolution used for the
// 1x1x1 skip convs and the small pyramid levels, k2s2 transposed conv, max-pool, the fused tail and the bf16 gather.
// The LDS-tiled 3^3 / 7^3 kernels for the large levels are in conv3d_bf16_tiled.hip.
// Reference call sites: network/v2v.py:8-43 (Basic3DBlock / Res3DBlock), :46-67 (pool / upsample), :155-161 (tail).
#include "bf16_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// weight packer
// ------------------------------------------------------------------------------------------------
__global__ void pack_bf16_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                 unsigned short* __restrict__ wpack, float* __restrict__ bpack, int cout, int cin,
                                 int cin_pad, int ksize, int transposed, long long total) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const PackGeomB p = pack_geom_b(cout, cin_pad, ksize, transposed);
    if (e < p.mtiles * 16) {
        float v = 0.f;
        if (e < cout) {
            const float sc = gamma ? gamma[e] / sqrtf(var[e] + eps) : 1.f;
            const float b0 = b ? b[e] : 0.f;
            v = gamma ? (b0 - mean[e]) * sc + beta[e] : b0;
        }
        bpack[e] = v;
    }
    if (e >= total) return;
    const long long total_a = (long long)p.mtiles * p.ksteps * 512;
    if (e >= total_a) {       // section R (7^3, cout <= 16): [octet][q][dy][lane][8], bf16_common.h se_k7r_slot
        const long long eb = e - total_a;
        const int j = (int)(eb & 7);
        const int lane = (int)((eb >> 3) & 63);
        const long long blk = eb >> 9;
        const int dy = (int)(blk % 7);
        const int q = (int)((blk / 7) % SE_K7R_GROUPS);
        const int c = (int)(blk / (7 * SE_K7R_GROUPS));
        const int g = lane >> 4, co = lane & 15;
        int dx, dz;
        const bool valid = se_k7r_slot(4 * q + g, dx, dz);
        const int ci = c * 8 + j;
        float v = 0.f;
        if (valid && co < cout && ci < cin) {
            const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
            v = w[((size_t)co * cin + ci) * 343 + (dx * 7 + dy) * 7 + dz] * sc;
        }
        wpack[e] = f2bf(v);
        return;
    }
    const int j = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    const long long blk = e >> 9;
    const int s = (int)(blk % p.ksteps);
    const int m = (int)(blk / p.ksteps);
    const int g = lane >> 4, r = lane & 15;
    const int co = se_bf16_cout_of(cout, m, r);
    const int c = s / p.kpc, sl = s - c * p.kpc;
    const int ql = 4 * sl + g;
    int tap = -1, o = 0;
    if (!transposed && ksize == 7) {
        int dx, dy, dz;
        if (se_k7b_slot(ql, dx, dy, dz)) tap = (dx * 7 + dy) * 7 + dz;
    } else if (ql < p.taps * p.oc) {
        tap = ql / p.oc;
        o = ql - tap * p.oc;
    }
    const int ci = (c * p.oc + o) * 8 + j;
    float v = 0.f;
    if (tap >= 0 && co < cout && ci < cin) {
        const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
        const size_t idx = transposed ? ((size_t)ci * cout + co) * 8 + tap : ((size_t)co * cin + ci) * p.taps + tap;
        v = w[idx] * sc;
    }
    wpack[e] = f2bf(v);
}

// ------------------------------------------------------------------------------------------------
// direct convolution: activations straight from global memory (L2), no LDS.  One wave = M_T cout tiles x N_T voxel tiles.
// ------------------------------------------------------------------------------------------------
template <int KS, int OC, int M_T, int N_T>
__global__ __launch_bounds__(256) void conv_bf16_direct_kernel(ConvBArgs a) {
    constexpr int H = KS / 2;
    constexpr int TAPS = KS * KS * KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = lane & 15, g = lane >> 4;
    const int mb = blockIdx.y;
    const int D = a.dim;
    const long long tile0 = ((long long)blockIdx.x * 4 + wave) * N_T;
    long long vox[N_T];
    int vx[N_T], vy[N_T], vz[N_T];
#pragma unroll
    for (int n = 0; n < N_T; ++n) {
        vox[n] = (tile0 + n) * 16 + v;
        const long long q = vox[n] < a.total_vox ? vox[n] : 0;
        vz[n] = (int)(q % D);
        vy[n] = (int)((q / D) % D);
        vx[n] = (int)((q / ((long long)D * D)) % D);
    }
    f32x4 acc[M_T][N_T];
#pragma unroll
    for (int m = 0; m < M_T; ++m)
#pragma unroll
        for (int n = 0; n < N_T; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned short* wp = a.wpack + ((size_t)mb * M_T * a.ksteps) * 512 + lane * 8;
    for (int c = 0; c < a.nchunk; ++c) {
        for (int sl = 0; sl < a.kpc; ++sl) {
            const int s = c * a.kpc + sl;
            const int ql = 4 * sl + g;
            int dx, dy, dz, o = 0;
            bool valid;
            if (KS == 7) {
                valid = se_k7b_slot(ql, dx, dy, dz);
            } else {
                valid = ql < TAPS * OC;
                const int t = ql / OC;
                o = ql - t * OC;
                dx = t / (KS * KS);
                dy = (t / KS) % KS;
                dz = t % KS;
            }
            u16x8 A[M_T];
#pragma unroll
            for (int m = 0; m < M_T; ++m) A[m] = *reinterpret_cast<const u16x8*>(wp + ((size_t)m * a.ksteps + s) * 512);
            const int coff = (c * OC + o) * 8;
#pragma unroll
            for (int n = 0; n < N_T; ++n) {
                const int xx = vx[n] + dx - H, yy = vy[n] + dy - H, zz = vz[n] + dz - H;
                const bool ok = valid && vox[n] < a.total_vox && (unsigned)xx < (unsigned)D && (unsigned)yy < (unsigned)D &&
                                (unsigned)zz < (unsigned)D;
                u16x8 Bf = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) {
                    const long long nb = vox[n] + ((long long)(dx - H) * D + (dy - H)) * D + (dz - H);
                    if (KS == 7) {   // octet-planar input [B][octs][N][8]; OC == 1, so the chunk index is the octet
                        const long long N = (long long)D * D * D;
                        const long long bq = nb / N;
                        Bf = *reinterpret_cast<const u16x8*>(a.in + ((bq * a.nchunk + c) * N + (nb - bq * N)) * 8);
                    } else {
                        Bf = *reinterpret_cast<const u16x8*>(a.in + nb * a.cin_pad + coff);
                    }
                }
#pragma unroll
                for (int m = 0; m < M_T; ++m) acc[m][n] = mfma_bf16(A[m], Bf, acc[m][n]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < N_T; ++n) {
        if (vox[n] >= a.total_vox) continue;
        if (M_T == 2) epilogue_pair_bf16(a, acc[0][n], acc[M_T - 1][n], vox[n], mb, g);
        else epilogue_single_bf16(a, acc[0][n], vox[n], g);
    }
}

template <int KS, int OC, int M_T, int N_T>
int launch_direct_b(const ConvBArgs& a, hipStream_t s) {
    const long long tiles = (a.total_vox + 15) / 16;
    const long long wgs = (tiles + 4 * N_T - 1) / (4 * N_T);
    const int mblocks = M_T == 2 ? a.cout / 32 : 1;
    hipLaunchKernelGGL((conv_bf16_direct_kernel<KS, OC, M_T, N_T>), dim3((unsigned)wgs, mblocks), dim3(256), 0, s, a);
    SE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 3x3x3 on the deep pyramid levels (8^3, 4^3, 2^3; 128 -> 128): a handful of voxels against 885 KB of weights, so the launch is
// bound by the serial chain of dependent global loads.  The four waves of a workgroup split the k steps (wave w takes
// s = w, w+4, ...), every wave issues the loads of U k steps before their MFMAs, and the partial sums meet in LDS in a fixed
// order (deterministic).  Workgroup = N_T voxel tiles x 32 couts.
// ------------------------------------------------------------------------------------------------
template <int N_T, int U>
__global__ __launch_bounds__(256) void conv_bf16_k3_splitk_kernel(ConvBArgs a) {
    __shared__ f32x4 red[4][2 * N_T][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = lane & 15, g = lane >> 4;
    const int mb = blockIdx.y;
    const int D = a.dim;
    long long vox[N_T];
    int vx[N_T], vy[N_T], vz[N_T];
#pragma unroll
    for (int n = 0; n < N_T; ++n) {
        vox[n] = ((long long)blockIdx.x * N_T + n) * 16 + v;
        const long long q = vox[n] < a.total_vox ? vox[n] : 0;
        vz[n] = (int)(q % D);
        vy[n] = (int)((q / D) % D);
        vx[n] = (int)((q / ((long long)D * D)) % D);
    }
    f32x4 acc[2][N_T];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < N_T; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned short* wp = a.wpack + ((size_t)mb * 2 * a.ksteps) * 512 + lane * 8;
    const int o = g & 1;
    for (int s0 = wave; s0 < a.ksteps; s0 += 4 * U) {
        u16x8 A[U][2], Bf[U][N_T];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + 4 * u;
            const bool live = s < a.ksteps;
            const int ss = live ? s : 0;
            const int c = ss / 14, sl = ss - c * 14;         // kpc == 14 (3^3, 16-channel chunks)
            int tap = 2 * sl + (g >> 1);
            const bool valid = live && tap < 27;
            tap = tap > 26 ? 26 : tap;
            const int dx = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dz = tap % 3 - 1;
            A[u][0] = *reinterpret_cast<const u16x8*>(wp + (size_t)ss * 512);
            A[u][1] = *reinterpret_cast<const u16x8*>(wp + ((size_t)a.ksteps + ss) * 512);
            if (!live) { A[u][0] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0}; A[u][1] = A[u][0]; }
#pragma unroll
            for (int n = 0; n < N_T; ++n) {
                const int xx = vx[n] + dx, yy = vy[n] + dy, zz = vz[n] + dz;
                const bool ok = valid && vox[n] < a.total_vox && (unsigned)xx < (unsigned)D && (unsigned)yy < (unsigned)D &&
                                (unsigned)zz < (unsigned)D;
                Bf[u][n] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) Bf[u][n] = *reinterpret_cast<const u16x8*>(a.in + (vox[n] + ((long long)dx * D + dy) * D + dz) * a.cin_pad + c * 16 + o * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int n = 0; n < N_T; ++n) {
                acc[0][n] = mfma_bf16(A[u][0], Bf[u][n], acc[0][n]);
                acc[1][n] = mfma_bf16(A[u][1], Bf[u][n], acc[1][n]);
            }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < N_T; ++n) red[wave][m * N_T + n][lane] = acc[m][n];
    __syncthreads();
    if (wave < N_T) {
        const int n = wave;
        f32x4 lo = red[0][n][lane], hi = red[0][N_T + n][lane];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) { lo += red[w2][n][lane]; hi += red[w2][N_T + n][lane]; }
        const long long ov = ((long long)blockIdx.x * N_T + n) * 16 + v;
        if (ov < a.total_vox) epilogue_pair_bf16(a, lo, hi, ov, mb, g);
    }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose3d k2s2: per output parity p a [cout x cin] GEMM on the input voxels; the 8 parities reuse the B fragments.
// ------------------------------------------------------------------------------------------------
template <int NCHUNK>
__global__ __launch_bounds__(256) void deconv_bf16_kernel(ConvBArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = lane & 15, g = lane >> 4;
    const int mb = blockIdx.y;
    const int D = a.dim;
    const long long vox = ((long long)blockIdx.x * 4 + wave) * 16 + v;
    const bool ok = vox < a.total_vox;
    const long long q = ok ? vox : 0;
    const int z = (int)(q % D), y = (int)((q / D) % D), x = (int)((q / ((long long)D * D)) % D);
    const long long bidx = q / ((long long)D * D * D);
    u16x8 Bf[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        Bf[c] = (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) Bf[c] = *reinterpret_cast<const u16x8*>(a.in + vox * a.cin_pad + c * 32 + g * 8);
    }
    const unsigned short* wp = a.wpack + ((size_t)mb * 2 * a.ksteps) * 512 + lane * 8;
    const int D2 = 2 * D;
#pragma unroll 2
    for (int p = 0; p < 8; ++p) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            const u16x8 A0 = *reinterpret_cast<const u16x8*>(wp + ((size_t)(c * 8 + p)) * 512);
            const u16x8 A1 = *reinterpret_cast<const u16x8*>(wp + ((size_t)a.ksteps + c * 8 + p) * 512);
            acc0 = mfma_bf16(A0, Bf[c], acc0);
            acc1 = mfma_bf16(A1, Bf[c], acc1);
        }
        if (ok) {
            const int px = p >> 2, py = (p >> 1) & 1, pz = p & 1;
            const long long ovox = ((bidx * D2 + (2 * x + px)) * D2 + (2 * y + py)) * D2 + (2 * z + pz);
            epilogue_pair_bf16(a, acc0, acc1, ovox, mb, g);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// max-pool 2x2x2: thread = (output voxel, octet)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_bf16_kernel(const unsigned short* __restrict__ in,
                                                            unsigned short* __restrict__ out, long long total_out_vox,
                                                            int dim_in, int octs) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long ov = t / octs;
    const int o = (int)(t - ov * octs);
    if (ov >= total_out_vox) return;
    const int Do = dim_in / 2;
    const int z = (int)(ov % Do), y = (int)((ov / Do) % Do), x = (int)((ov / ((long long)Do * Do)) % Do);
    const long long b = ov / ((long long)Do * Do * Do);
    const int C = octs * 8;
    float m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const long long iv = ((b * dim_in + (2 * x + (p >> 2))) * dim_in + (2 * y + ((p >> 1) & 1))) * dim_in + (2 * z + (p & 1));
        const u16x8 r = *reinterpret_cast<const u16x8*>(in + iv * C + o * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], bf2f(r[i]));
    }
    u16x8 w;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = f2bf(m[i]);   // exact: the maximum is one of the bf16 inputs
    *reinterpret_cast<u16x8*>(out + ov * C + o * 8) = w;
}

// ------------------------------------------------------------------------------------------------
// fused tail: 1x1x1 32->32 (+ReLU), 32->32 (+ReLU), 32->cout3 (<=16) -> float32 planar logits.  One k step per layer; the D
// fragments of a tile pair ARE the next layer's B fragment (channels 8g..8g+7), so the chain stays in registers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u16x8 relu_pack(f32x4 lo, f32x4 hi, const float* __restrict__ bias, int g) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 8 * g);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + 8 * g + 4);
    u16x8 o;
    o[0] = f2bf(fmaxf(lo.x + b0.x, 0.f)); o[1] = f2bf(fmaxf(lo.y + b0.y, 0.f));
    o[2] = f2bf(fmaxf(lo.z + b0.z, 0.f)); o[3] = f2bf(fmaxf(lo.w + b0.w, 0.f));
    o[4] = f2bf(fmaxf(hi.x + b1.x, 0.f)); o[5] = f2bf(fmaxf(hi.y + b1.y, 0.f));
    o[6] = f2bf(fmaxf(hi.z + b1.z, 0.f)); o[7] = f2bf(fmaxf(hi.w + b1.w, 0.f));
    return o;
}

__global__ __launch_bounds__(256) void pointwise_chain3_bf16_kernel(
    const unsigned short* __restrict__ in, const unsigned short* __restrict__ w1, const float* __restrict__ b1,
    const unsigned short* __restrict__ w2, const float* __restrict__ b2, const unsigned short* __restrict__ w3,
    const float* __restrict__ b3, float* __restrict__ out, long long total_vox, long long vox_per_b, int cout3) {
    const int lane = threadIdx.x & 63;
    const int v = lane & 15, g = lane >> 4;
    const u16x8 A10 = *reinterpret_cast<const u16x8*>(w1 + lane * 8);
    const u16x8 A11 = *reinterpret_cast<const u16x8*>(w1 + 512 + lane * 8);
    const u16x8 A20 = *reinterpret_cast<const u16x8*>(w2 + lane * 8);
    const u16x8 A21 = *reinterpret_cast<const u16x8*>(w2 + 512 + lane * 8);
    const u16x8 A30 = *reinterpret_cast<const u16x8*>(w3 + lane * 8);
    const f32x4 bias3 = *reinterpret_cast<const f32x4*>(b3 + 4 * g);
    const long long tiles = (total_vox + 15) / 16;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += (long long)gridDim.x * 4) {
        const long long vox = t * 16 + v;
        const bool ok = vox < total_vox;
        u16x8 B0 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) B0 = *reinterpret_cast<const u16x8*>(in + vox * 32 + g * 8);
        const u16x8 B1 = relu_pack(mfma_bf16(A10, B0, zero), mfma_bf16(A11, B0, zero), b1, g);
        const u16x8 B2 = relu_pack(mfma_bf16(A20, B1, zero), mfma_bf16(A21, B1, zero), b2, g);
        f32x4 r = mfma_bf16(A30, B2, zero);
        r += bias3;
        if (ok) {
            const long long b = vox / vox_per_b, n = vox - b * vox_per_b;
            float* o = out + (b * cout3 + 4 * g) * vox_per_b + n;
            const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * g + i < cout3) o[(long long)i * vox_per_b] = rv[i];
        }
    }
}

// The same chain with pass 1 of the soft-argmax folded in (round 6; the float32 program has had this since round 4:
// pointwise_chain3_sa_kernel in conv3d.hip, same chunking, same SE_SA_PART records, same fixed-order fold): the workgroup (chunk s,
// sample b) owns the voxels [s * chunk, (s + 1) * chunk) of its sample, every lane keeps a running (max, sum exp, sum exp * coord) for its
// 4 joints while the float32 logits are in registers, and softargmax_finish_kernel does pass 2.  Saves the two stand-alone launches of
// se_softargmax3d_f32's pass 1 and their re-read of the logits (0.35 ms at B = 32).  Reference: utils/op.py:83-96.
constexpr int PWB_SA_WAVES = 16;
__global__ __launch_bounds__(PWB_SA_WAVES * 64) void pointwise_chain3_sa_bf16_kernel(
    const unsigned short* __restrict__ in, const unsigned short* __restrict__ w1, const float* __restrict__ b1,
    const unsigned short* __restrict__ w2, const float* __restrict__ b2, const unsigned short* __restrict__ w3,
    const float* __restrict__ b3, float* __restrict__ out, const float* __restrict__ coord, float* __restrict__ scratch, int vox_per_b,
    int chunk, int cout3, int splits) {
    extern __shared__ __attribute__((aligned(16))) float pwb_sa_lds[];
    float (*red)[64][20] = reinterpret_cast<float (*)[64][20]>(pwb_sa_lds);        // [PWB_SA_WAVES][64][20]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int v = lane & 15, g = lane >> 4;
    const int s = blockIdx.x, b = blockIdx.y;
    const int v0 = s * chunk, v1 = min(v0 + chunk, vox_per_b);
    const u16x8 A10 = *reinterpret_cast<const u16x8*>(w1 + lane * 8);
    const u16x8 A11 = *reinterpret_cast<const u16x8*>(w1 + 512 + lane * 8);
    const u16x8 A20 = *reinterpret_cast<const u16x8*>(w2 + lane * 8);
    const u16x8 A21 = *reinterpret_cast<const u16x8*>(w2 + 512 + lane * 8);
    const u16x8 A30 = *reinterpret_cast<const u16x8*>(w3 + lane * 8);
    const f32x4 bias3 = *reinterpret_cast<const f32x4*>(b3 + 4 * g);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* inb = in + (long long)b * vox_per_b * 32 + g * 8;
    float m[4], l[4], sx[4], sy[4], sz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = sx[r] = sy[r] = sz[r] = 0.f; }
    for (int t0 = v0 + wave * 16; t0 < v1; t0 += PWB_SA_WAVES * 16) {
        const int n = t0 + v;
        const bool ok = n < v1;
        u16x8 B0 = {0, 0, 0, 0, 0, 0, 0, 0};
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (ok) {
            B0 = *reinterpret_cast<const u16x8*>(inb + (long long)n * 32);
            cx = coord[(size_t)n * 3]; cy = coord[(size_t)n * 3 + 1]; cz = coord[(size_t)n * 3 + 2];
        }
        const u16x8 B1 = relu_pack(mfma_bf16(A10, B0, zero), mfma_bf16(A11, B0, zero), b1, g);
        const u16x8 B2 = relu_pack(mfma_bf16(A20, B1, zero), mfma_bf16(A21, B1, zero), b2, g);
        f32x4 r4 = mfma_bf16(A30, B2, zero);
        r4 += bias3;
        if (ok) {
            float* o = out + ((long long)b * cout3 + 4 * g) * vox_per_b + n;
            const float vv[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * g + r < cout3) {
                    o[(long long)r * vox_per_b] = vv[r];
                    if (vv[r] > m[r]) {      // new running maximum: rescale the sums (exp(-inf) = 0 clears them the first time)
                        const float sc = __expf(m[r] - vv[r]);
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
    if (j < cout3) {          // joint j: the 16 x PWB_SA_WAVES (wave, voxel lane) partials of k lane j / 4, slot j % 4, folded in a fixed order
        const int hh = j >> 2, rr = (j & 3) * 5;
        float M = -INFINITY;
        for (int w = 0; w < PWB_SA_WAVES; ++w)
            for (int q = 0; q < 16; ++q) M = fmaxf(M, red[w][hh * 16 + q][rr]);
        float L = 0.f, SX = 0.f, SY = 0.f, SZ = 0.f;
        for (int w = 0; w < PWB_SA_WAVES; ++w)
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
// bilinear voxel gather with bf16 output: thread = (voxel, octet); see gather.hip for the float32 form
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_bf16_kernel(const float* __restrict__ feat, const int4* __restrict__ idx,
                                                          const f32x4* __restrict__ w, unsigned short* __restrict__ out,
                                                          int batch, int texels, int octs, int voxels, int octs_total,
                                                          int oct_offset) {
    // octet-major thread order: the 16-byte records of one octet plane are written contiguously.  The table entry of a voxel
    // (32 B) is read once and reused for the whole batch (grid.y covers batch slices of 8).
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int o = (int)(t / voxels);
    const int v = (int)(t - (long long)o * voxels);
    if (o >= octs) return;
    const int4 id = idx[v];
    const f32x4 wt = w[v];
    const int C = octs * 8;
    const int b1 = min(batch, (int)(blockIdx.y + 1) * 8);
    for (int b = blockIdx.y * 8; b < b1; ++b) {
        const float* fb = feat + (size_t)b * texels * C + o * 8;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        // same tap order (nw, ne, sw, se) and float32 arithmetic as gather_kernel; only the store rounds to bf16
        if (id.x >= 0) { const float* p = fb + (size_t)id.x * C; a0 += *reinterpret_cast<const f32x4*>(p) * wt.x; a1 += *reinterpret_cast<const f32x4*>(p + 4) * wt.x; }
        if (id.y >= 0) { const float* p = fb + (size_t)id.y * C; a0 += *reinterpret_cast<const f32x4*>(p) * wt.y; a1 += *reinterpret_cast<const f32x4*>(p + 4) * wt.y; }
        if (id.z >= 0) { const float* p = fb + (size_t)id.z * C; a0 += *reinterpret_cast<const f32x4*>(p) * wt.z; a1 += *reinterpret_cast<const f32x4*>(p + 4) * wt.z; }
        if (id.w >= 0) { const float* p = fb + (size_t)id.w * C; a0 += *reinterpret_cast<const f32x4*>(p) * wt.w; a1 += *reinterpret_cast<const f32x4*>(p + 4) * wt.w; }
        u16x8 r;
        r[0] = f2bf(a0.x); r[1] = f2bf(a0.y); r[2] = f2bf(a0.z); r[3] = f2bf(a0.w);
        r[4] = f2bf(a1.x); r[5] = f2bf(a1.y); r[6] = f2bf(a1.z); r[7] = f2bf(a1.w);
        *reinterpret_cast<u16x8*>(out + (((size_t)b * octs_total + oct_offset + o) * voxels + v) * 8) = r;
    }
}

}  // namespace

// implemented in conv3d_bf16_tiled.hip: returns SE_TILED_NOT_TAKEN_B if the shape is not covered
#define SE_TILED_NOT_TAKEN_B (-1000)
int se_conv3d_bf16_tiled_try(const ConvBArgs& a, int batch, int ksize, hipStream_t s);

extern "C" long long se_conv3d_packed_elems_bf16(int cout, int cin_pad, int ksize, int transposed) {
    if (cout <= 0 || cin_pad <= 0 || (cin_pad & 7)) return SE_ERR_BAD_ARG;
    if (cout % 32 != 0 && cout > 16) return SE_ERR_BAD_ARG;
    const PackGeomB p = pack_geom_b(cout, cin_pad, ksize, transposed);
    return (long long)p.mtiles * p.ksteps * 512 + se_k7r_elems(cout, cin_pad, ksize, transposed);
}

extern "C" int se_conv3d_pack_bf16(const float* w, const float* b, const float* gamma, const float* beta,
                                   const float* mean, const float* var, float eps, se_bf16* wpack, float* bpack,
                                   int cout, int cin, int cin_pad, int ksize, int transposed, void* stream) {
    if (cout <= 0 || cin <= 0 || cin_pad < cin || (cin_pad & 7)) return SE_ERR_BAD_ARG;
    if (cout % 32 != 0 && cout > 16) return SE_ERR_BAD_ARG;
    if (transposed ? ksize != 2 : (ksize != 1 && ksize != 3 && ksize != 7)) return SE_ERR_BAD_ARG;
    if (transposed && ((cin_pad & 31) || (cout & 31))) return SE_ERR_BAD_ARG;
    const long long total = se_conv3d_packed_elems_bf16(cout, cin_pad, ksize, transposed);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, se_stream(stream), w, b, gamma,
                       beta, mean, var, eps, wpack, bpack, cout, cin, cin_pad, ksize, transposed, total);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_conv3d_bf16(const se_bf16* in, const se_bf16* wpack, const float* bpack, const se_bf16* residual,
                              se_bf16* out, int batch, int dim, int cin_pad, int cout, int ksize, int flags, void* stream) {
    if (batch <= 0 || dim <= 0 || cin_pad <= 0 || (cin_pad & 7) || cout <= 0) return SE_ERR_BAD_ARG;
    if (ksize != 1 && ksize != 3 && ksize != 7) return SE_ERR_BAD_ARG;
    if (cout % 32 != 0 && cout != 16) return SE_ERR_BAD_ARG;
    if (flags & SE_EPI_OUT_PLANAR) return SE_ERR_BAD_ARG;
    if ((flags & SE_EPI_RES_PRE_RELU) && (flags & SE_EPI_RES_POST_RELU)) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    const PackGeomB p = pack_geom_b(cout, cin_pad, ksize, 0);
    ConvBArgs a;
    a.in = in; a.wpack = wpack; a.bpack = bpack; a.res = residual; a.out = out;
    a.total_vox = (long long)batch * dim * dim * dim;
    a.dim = dim; a.cin_pad = cin_pad; a.cout = cout; a.flags = flags;
    a.kpc = p.kpc; a.nchunk = p.nchunk; a.ksteps = p.ksteps;
    const int took = se_conv3d_bf16_tiled_try(a, batch, ksize, s);
    if (took != SE_TILED_NOT_TAKEN_B) return took;
    const bool pair = cout % 32 == 0;
    if (ksize == 3 && pair && p.oc == 2 && a.total_vox <= 8 * 8 * 8 * 64) {
        const long long tiles = (a.total_vox + 15) / 16;
        if (tiles >= 128) {
            hipLaunchKernelGGL((conv_bf16_k3_splitk_kernel<2, 4>), dim3((unsigned)((tiles + 1) / 2), cout / 32), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL((conv_bf16_k3_splitk_kernel<1, 4>), dim3((unsigned)tiles, cout / 32), dim3(256), 0, s, a);
        }
        SE_CHECK_LAUNCH();
        return 0;
    }
    // few voxels (deep pyramid levels): one voxel tile per wave so that the launch still spreads over the chip
    const bool small = a.total_vox * (cout / 16) < 256 * 4 * 32;
    if (ksize == 7) return pair ? launch_direct_b<7, 1, 2, 2>(a, s) : launch_direct_b<7, 1, 1, 2>(a, s);
    if (ksize == 3) {
        if (p.oc == 2) return !pair ? launch_direct_b<3, 2, 1, 2>(a, s) : small ? launch_direct_b<3, 2, 2, 1>(a, s) : launch_direct_b<3, 2, 2, 2>(a, s);
        return pair ? launch_direct_b<3, 1, 2, 2>(a, s) : launch_direct_b<3, 1, 1, 2>(a, s);
    }
    if (p.oc == 4) return pair ? launch_direct_b<1, 4, 2, 4>(a, s) : launch_direct_b<1, 4, 1, 4>(a, s);
    if (p.oc == 2) return pair ? launch_direct_b<1, 2, 2, 4>(a, s) : launch_direct_b<1, 2, 1, 4>(a, s);
    return pair ? launch_direct_b<1, 1, 2, 4>(a, s) : launch_direct_b<1, 1, 1, 4>(a, s);
}

extern "C" int se_deconv3d_k2s2_bf16(const se_bf16* in, const se_bf16* wpack, const float* bpack, const se_bf16* residual,
                                     se_bf16* out, int batch, int dim, int cin, int cout, int flags, void* stream) {
    if (batch <= 0 || dim <= 0 || cin <= 0 || (cin & 31) || cin > 128 || cout <= 0 || (cout & 31)) return SE_ERR_BAD_ARG;
    if ((flags & SE_EPI_RES_PRE_RELU) && (flags & SE_EPI_RES_POST_RELU)) return SE_ERR_BAD_ARG;
    const PackGeomB p = pack_geom_b(cout, cin, 2, 1);
    ConvBArgs a;
    a.in = in; a.wpack = wpack; a.bpack = bpack; a.res = residual; a.out = out;
    a.total_vox = (long long)batch * dim * dim * dim;
    a.dim = dim; a.cin_pad = cin; a.cout = cout; a.flags = flags;
    a.kpc = p.kpc; a.nchunk = p.nchunk; a.ksteps = p.ksteps;
    const dim3 grid((unsigned)((a.total_vox + 63) / 64), cout / 32);
    hipStream_t s = se_stream(stream);
    switch (p.nchunk) {
        case 1: hipLaunchKernelGGL(deconv_bf16_kernel<1>, grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(deconv_bf16_kernel<2>, grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(deconv_bf16_kernel<3>, grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(deconv_bf16_kernel<4>, grid, dim3(256), 0, s, a); break;
    }
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_maxpool3d_2_bf16(const se_bf16* in, se_bf16* out, int batch, int dim, int channels, void* stream) {
    if (batch <= 0 || dim <= 0 || (dim & 1) || channels <= 0 || (channels & 7)) return SE_ERR_BAD_ARG;
    const int Do = dim / 2;
    const long long ovox = (long long)batch * Do * Do * Do;
    const long long threads = ovox * (channels / 8);
    hipLaunchKernelGGL(maxpool2_bf16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, se_stream(stream), in, out,
                       ovox, dim, channels / 8);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_pointwise_chain3_bf16(const se_bf16* in, const se_bf16* wpack1, const float* bpack1,
                                        const se_bf16* wpack2, const float* bpack2, const se_bf16* wpack3,
                                        const float* bpack3, float* out, int batch, int dim, int cout3, void* stream) {
    if (batch <= 0 || dim <= 0 || cout3 <= 0 || cout3 > 16) return SE_ERR_BAD_ARG;
    const long long vpb = (long long)dim * dim * dim;
    const long long total = vpb * batch;
    const long long tiles = (total + 15) / 16;
    const unsigned grid = (unsigned)((tiles + 3) / 4 < 256 * 16 ? (tiles + 3) / 4 : 256 * 16);
    hipLaunchKernelGGL(pointwise_chain3_bf16_kernel, dim3(grid), dim3(256), 0, se_stream(stream), in, wpack1, bpack1, wpack2,
                       bpack2, wpack3, bpack3, out, total, vpb, cout3);
    SE_CHECK_LAUNCH();
    return 0;
}

// se_pointwise_chain3_bf16 + pass 1 of se_softargmax3d_f32 (mode 1) in one launch; finish with se_softargmax3d_finish_f32.
extern "C" int se_pointwise_chain3_softargmax_bf16(const se_bf16* in, const se_bf16* wpack1, const float* bpack1, const se_bf16* wpack2,
                                                   const float* bpack2, const se_bf16* wpack3, const float* bpack3, float* out,
                                                   const float* coord, float* scratch, int batch, int dim, int cout3, void* stream) {
    if (batch <= 0 || dim <= 0 || cout3 <= 0 || cout3 > 16 || !coord || !scratch) return SE_ERR_BAD_ARG;
    const long long vox_per_b = (long long)dim * dim * dim;
    if (vox_per_b >= (1LL << 31) || (vox_per_b & 3)) return SE_ERR_BAD_ARG;
    const int splits = se_sa_splits(batch * cout3);                                               // as softargmax.hip
    const int chunk = (int)((((vox_per_b + splits - 1) / splits) + 3) & ~3LL);
    if (chunk & 15) return SE_ERR_BAD_ARG;                                                       // whole 16-voxel tiles per chunk
    constexpr int LDS = PWB_SA_WAVES * 64 * 20 * 4;
    SE_ENSURE_LDS(pointwise_chain3_sa_bf16_kernel, LDS);
    hipLaunchKernelGGL(pointwise_chain3_sa_bf16_kernel, dim3(splits, batch), dim3(PWB_SA_WAVES * 64), LDS, se_stream(stream), in, wpack1,
                       bpack1, wpack2, bpack2, wpack3, bpack3, out, coord, scratch, (int)vox_per_b, chunk, cout3, splits);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_unproject_gather_bf16(const float* feat, const int* idx, const float* w, se_bf16* out, int batch,
                                        int texels, int channels, int voxels, int octs_total, int out_c_offset,
                                        void* stream) {
    if (batch <= 0 || texels <= 0 || voxels <= 0 || channels <= 0) return SE_ERR_BAD_ARG;
    if ((channels & 7) || octs_total <= 0 || (out_c_offset & 7) || out_c_offset < 0 || out_c_offset + channels > octs_total * 8)
        return SE_ERR_BAD_ARG;
    const int octs = channels / 8;
    const long long threads = (long long)voxels * octs;
    dim3 grid((unsigned)((threads + 255) / 256), (batch + 7) / 8);
    hipLaunchKernelGGL(gather_bf16_kernel, grid, dim3(256), 0, se_stream(stream), feat, reinterpret_cast<const int4*>(idx),
                       reinterpret_cast<const f32x4*>(w), out, batch, texels, octs, voxels, octs_total, out_c_offset / 8);
    SE_CHECK_LAUNCH();
    return 0;
}
