// LDS-tiled 3D convolution kernels for the large pyramid levels (64^3 / 32^3 / 16^3), exact float32 MFMA.
//
// One workgroup (4 waves) owns a TZ x 8 x 8 block of output voxels of one sample and N_T*16 output channels.
// The input halo (TZ+k-1) x (8+k-1) x (8+k-1) is staged in LDS one channel CHUNK at a time (CK channels,
// 16 B per lane, zero-filled outside the volume), then every wave walks the taps reading its activation
// fragments from LDS (one ds_read_b128 per 16-voxel tile and step) and its weight fragments straight from
// global memory (packed in fragment order, 1 KiB contiguous per wave-instruction, L1/L2 resident, prefetched one
// step ahead).  MFMA: v_mfma_f32_16x16x4_f32, operand maps as in conv3d.hip.
//
//   k = 3, k = 1: CK = 16.  A step = (tap, 16-channel group): lane (voxel v, k-lane h) reads channels 4h..4h+3;
//                 the 4 MFMAs of a step contract channel j of every k-lane's quad.   LDS: 600 x 64 B = 38.4 KB
//                 (TZ = 4) -> 4 workgroups / CU, so staging of one overlaps MFMA work of the others.
//   k = 7:        CK = 4, and the k lanes carry 4 consecutive TAPS instead of 4 channel quads: a step = 4 taps x
//                 4 channels, lane (v, h) reads channels 0..3 of tap 4g+h.  The halo of a 7^3 filter is large
//                 (10 x 14 x 14 voxels for TZ = 4) — 4-channel chunks keep it at 31 KB.  The 33rd input channel
//                 (scene occupancy) rides in a 9th chunk whose other 3 channels have zero weights.
//
// Why this is MFMA-bound, not LDS/HBM-bound: per step a wave issues M_T ds_read_b128 + N_T global 16-B loads for
// 4*M_T*N_T MFMAs of 32 cycles each (k3 32->32: 6 loads for 32 MFMAs = 1024 SIMD cycles).
#include "conv_common.h"

#include <type_traits>

#define SE_K7_TSTRIDE 92   // per-k-lane row of the 7^3 tap-offset table (86 groups + pad, multiple of 4)

#ifdef SE_DEVTOOLS
thread_local int g_variant = 0;   // A/B switch of development builds (se_debug_set_variant): 1 = disable the persistent 64^3 kernel
#endif

namespace {


template <int KS, int CK, int TZ>
struct TileGeom {
    static constexpr int P = (KS - 1) / 2;
    static constexpr int TAPS = KS * KS * KS;
    static constexpr int TY = 8, TX = 8;
    static constexpr int HZ = TZ + 2 * P, HY = TY + 2 * P, HX = TX + 2 * P;
    static constexpr int HV = HZ * HY * HX;        // halo voxels
    static constexpr int QPC = CK / 4;             // 16-byte quads per voxel in the LDS tile
    static constexpr int M_T = TZ;                 // 16-voxel tiles per wave: TZ/4 z-slabs x 4 row pairs
    static constexpr bool TAP_LANES = CK < 16;     // k lanes carry taps (CK = 4) instead of channel quads
    static constexpr int GROUPS = TAP_LANES ? (TAPS + 3) / 4 : TAPS * (CK / 16);
    static constexpr int LDS_BYTES = HV * CK * 4;
};

template <int KS, int CK, int TZ, int N_T>
__global__ __launch_bounds__(256) void conv3d_tiled_kernel(ConvArgs a, int tiles_per_dim, int ztiles) {
    using G = TileGeom<KS, CK, TZ>;
    constexpr int P = G::P, HY = G::HY, HX = G::HX, HV = G::HV, QPC = G::QPC, M_T = G::M_T;
    static_assert(CK == 4 || CK == 16, "chunk is 4 (tap lanes) or 16 (channel lanes)");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int vl = lane & 15;
    const int h = lane >> 4;
    const int dim = a.dim;
    const int nt0 = blockIdx.y * N_T;

    // tile origin
    int t = blockIdx.x;
    const int tx = t % tiles_per_dim; t /= tiles_per_dim;
    const int ty = t % tiles_per_dim; t /= tiles_per_dim;
    const int tz = t % ztiles; t /= ztiles;
    const int b = t;
    const int x0 = tx * 8, y0 = ty * 8, z0 = tz * TZ;
    const float* in_b = a.in + (size_t)b * dim * dim * dim * a.cin_pad;

    // LDS voxel index of this lane's output voxel (tap 0,0,0 corner) in each of the wave's tiles
    int vbase[M_T];
#pragma unroll
    for (int m = 0; m < M_T; ++m) {
        const int zl = (m >> 2) * 4 + wave;
        const int yl = (m & 3) * 2 + (vl >> 3);
        const int xl = vl & 7;
        vbase[m] = (zl * HY + yl) * HX + xl;
    }

    f32x4 acc[M_T][N_T];
#pragma unroll
    for (int m = 0; m < M_T; ++m)
#pragma unroll
        for (int n = 0; n < N_T; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int chunks = (a.cin + CK - 1) / CK;
    const f32x4* wp = reinterpret_cast<const f32x4*>(G::TAP_LANES ? a.wpack_b : a.wpack);

    for (int ch = 0; ch < chunks; ++ch) {
        if (ch > 0) __syncthreads();  // every wave is done reading the previous chunk
        // ---- stage the halo of this chunk: global (16 B per lane) -> LDS, zero outside the volume ----
#pragma unroll 4
        for (int it = tid; it < HV * QPC; it += 256) {
            const int hv = it / QPC, q = it - hv * QPC;
            const int hx = hv % HX;
            const int t2 = hv / HX;
            const int hy = t2 % HY;
            const int hz = t2 / HY;
            const int gz = z0 - P + hz, gy = y0 - P + hy, gx = x0 - P + hx;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gz < (unsigned)dim && (unsigned)gy < (unsigned)dim && (unsigned)gx < (unsigned)dim)
                val = *reinterpret_cast<const f32x4*>(in_b + ((size_t)(gz * dim + gy) * dim + gx) * a.cin_pad + ch * CK + q * 4);
            *reinterpret_cast<f32x4*>(lds + (size_t)it * 4) = val;
        }
        __syncthreads();

        // ---- MFMA over the taps of this chunk ----
        const f32x4* wrow = wp + ((size_t)ch * G::GROUPS * a.nts + nt0) * 64 + lane;
        f32x4 wcur[N_T];
#pragma unroll
        for (int n = 0; n < N_T; ++n) wcur[n] = wrow[n * 64];
        for (int g = 0; g < G::GROUPS; ++g) {
            // prefetch the next step's weights (clamped on the last step: harmless re-read)
            const int gn = (g + 1 < G::GROUPS) ? g + 1 : g;
            f32x4 wnext[N_T];
#pragma unroll
            for (int n = 0; n < N_T; ++n) wnext[n] = wrow[((size_t)gn * a.nts + n) * 64];
            // LDS offset (in floats) of this lane's fragment for step g
            int off;
            if (G::TAP_LANES) {
                int tap = 4 * g + h;
                tap = tap < G::TAPS ? tap : 0;            // pad tap of the last group: weights are zero
                const int kz = tap / (KS * KS);
                const int r = tap - kz * (KS * KS);
                const int ky = r / KS;
                const int kx = r - ky * KS;
                off = ((kz * HY + ky) * HX + kx) * CK;
            } else {
                const int kz = g / (KS * KS);
                const int r = g - kz * (KS * KS);
                const int ky = r / KS;
                const int kx = r - ky * KS;
                off = ((kz * HY + ky) * HX + kx) * CK + 4 * h;
            }
            f32x4 xf[M_T];
#pragma unroll
            for (int m = 0; m < M_T; ++m) xf[m] = *reinterpret_cast<const f32x4*>(lds + vbase[m] * CK + off);
#pragma unroll
            for (int n = 0; n < N_T; ++n) {
#pragma unroll
                for (int m = 0; m < M_T; ++m) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[n].x, xf[m].x, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[n].y, xf[m].y, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[n].z, xf[m].z, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[n].w, xf[m].w, acc[m][n], 0, 0, 0);
                }
            }
#pragma unroll
            for (int n = 0; n < N_T; ++n) wcur[n] = wnext[n];
        }
    }

    // ---- epilogue ----
    const long long ovox_per_b = (long long)dim * dim * dim;
#pragma unroll
    for (int m = 0; m < M_T; ++m) {
        const int oz = z0 + (m >> 2) * 4 + wave;
        const int oy = y0 + (m & 3) * 2 + (vl >> 3);
        const int ox = x0 + (vl & 7);
        const long long on = ((long long)oz * dim + oy) * dim + ox;
#pragma unroll
        for (int n = 0; n < N_T; ++n) conv_epilogue(a, acc[m][n], b, on, ovox_per_b, (nt0 + n) * 16 + 4 * h);
    }
}

#ifdef SE_DEVTOOLS   // retired A/B variants: persistent direct (non-Winograd) kernels of the 64^3 level
#include "devtools/tiled_persistent_direct_kernels.inc"
#endif  // SE_DEVTOOLS (persistent direct kernels)

template <int KS, int CK, int TZ, int N_T>
int launch_tiled(const ConvArgs& a, int batch, hipStream_t s) {
    using G = TileGeom<KS, CK, TZ>;
    const int tiles = a.dim / 8, ztiles = a.dim / TZ;
    dim3 grid((unsigned)(batch * ztiles * tiles * tiles), (unsigned)(a.nts / N_T));
    auto kern = conv3d_tiled_kernel<KS, CK, TZ, N_T>;
    if (G::LDS_BYTES > 48 * 1024) {
        SE_ENSURE_LDS(kern, G::LDS_BYTES);
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, s, a, tiles, ztiles);
    SE_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Returns 0 if a tiled kernel took the launch, SE_TILED_NOT_TAKEN if the shape is left to the direct kernel,
// otherwise the hipError_t of the failed launch.
int se_conv3d_wino_try(const ConvArgs& a, int batch, hipStream_t s);   // conv3d_wino.hip
int se_conv3d_wino2d_try(const ConvArgs& a, int batch, int launch_batch, hipStream_t s); // conv3d_wino2d.hip
int se_conv3d_k7_wino_try(const ConvArgs& a, int batch, hipStream_t s);

static int tiled_try_one(const ConvArgs& a, int batch, int launch_batch, int ksize, hipStream_t s);

// The persistent kernels keep a per-workgroup table of their work units in the LDS left over beside weights and tiles
// (a few hundred entries): large batches are cut into slices of 32 samples, one launch each (weak-scaling config 4 runs
// 32 samples per GPU, i.e. exactly one slice).
int se_conv3d_tiled_try(const ConvArgs& a, int batch, int ksize, hipStream_t s) {
    // unit-table budget: the smallest table among the persistent kernels holds ~430 entries per workgroup
    const long long t8 = a.dim / 8;
    const long long units_per_sample = ksize == 7 ? t8 * t8 * t8 : (long long)(a.cout >= 32 ? a.cout / 32 : 1) * (a.dim / 4) * t8 * t8;
    long long budget = units_per_sample > 0 ? 400LL * se_num_cus() / units_per_sample : 32;
    const int SLICE = (int)(budget < 1 ? 1 : budget > 32 ? 32 : budget);
    if (batch <= SLICE) return tiled_try_one(a, batch, batch, ksize, s);
    const long long vox = (long long)a.dim * a.dim * a.dim;
    for (int b0 = 0; b0 < batch; b0 += SLICE) {
        const int nb = batch - b0 < SLICE ? batch - b0 : SLICE;
        ConvArgs sl = a;
        // floats per voxel of the input: channels-last record, or 3 x ceil(cin/3) planes for the triplet-planar 7^3 input
        sl.in = a.in + (long long)b0 * vox * ((a.flags & SE_IN_PLANAR3) ? (a.cin + 2) / 3 * 3 : a.cin_pad);
        sl.out = a.out + (long long)b0 * vox * a.cout;
        if (a.res) sl.res = a.res + (long long)b0 * vox * ((a.flags & SE_EPI_SKIPCONV16) ? 16 : a.cout);
        if (a.pool_out) sl.pool_out = a.pool_out + (long long)b0 * (vox / 8) * a.cout;
        sl.total_vox = (long long)nb * vox;
        // batch-dependent decisions (small-volume early-out, which 2-D Winograd kernel) are taken on the WHOLE batch, so every slice
        // decides alike (a one-sample tail slice used to say "not taken" after the first slices had been launched)
        const int rc = tiled_try_one(sl, nb, batch, ksize, s);
        if (rc != 0) return rc;   // not taken (same decision for every slice: nothing launched yet) or an error
    }
    return 0;
}

// `batch` samples are launched; `launch_batch` = samples of the whole call this slice belongs to (what batch-dependent choices look at)
static int tiled_try_one(const ConvArgs& a, int batch, int launch_batch, int ksize, hipStream_t s) {
    const int dim = a.dim;
    // tiny volumes of 2-D Winograd shapes (16^3 at batch 1: 32 work units for 256 CUs): a plain channels-last call is left to the
    // in-workgroup split-K kernel of the 8^3 level (conv_common.h: se_conv3d_small_volume; 47 against 92 us per 128 -> 128 launch);
    // a caller that asks for an octet-planar / pooled / fused-skip form gets the 2-D kernel as before
    if (ksize == 3 && g_variant == 0 && se_conv3d_small_volume(launch_batch, dim) && a.cin_pad == a.cin && se_wino2d_shape_ok(dim, a.cin, a.cout) &&
        !(a.flags & (SE_LAYOUT_OCTET_BITS | SE_LAYOUT_QUAD_BITS | SE_EPI_SKIPCONV16 | SE_EPI_RES_POST_RELU | SE_EPI_OUT_PLANAR)) && !a.pool_out &&
        !a.skip_w && (a.nts % 2) == 0)
        return SE_TILED_NOT_TAKEN;
    if (ksize == 3 && (g_variant == 0 || (g_variant >= 40 && g_variant < 70))) {   // production: 2-D Winograd F(4,3) x F(2,3), register accumulators
        const int rc = se_conv3d_wino2d_try(a, batch, launch_batch, s);
        if (rc != SE_TILED_NOT_TAKEN) return rc;
    }
    // octet-planar tensors, the pooled second output and the fused 16-channel skip convolution exist in the 2-D Winograd kernel only:
    // a launch that asks for one of them and was declined (cin_pad != cin, SE_EPI_RES_POST_RELU / SE_EPI_OUT_PLANAR, ...) is an error -
    // none of the kernels below would read or write those tensors the way the caller laid them out
    if ((a.flags & (SE_LAYOUT_OCTET_BITS | SE_LAYOUT_QUAD_BITS | SE_EPI_SKIPCONV16)) || a.pool_out || a.skip_w) return SE_ERR_BAD_ARG;
    if (ksize == 3 && (g_variant == 0 || g_variant == 4 || g_variant == 30 || (g_variant >= 10 && g_variant < 20))) {   // 1-D Winograd F(4,3) (se_debug_set_variant(30): instead of the 2-D kernel)
        const int rc = se_conv3d_wino_try(a, batch, s);
        if (rc != SE_TILED_NOT_TAKEN) return rc;
    }
    if (dim < 16 || (dim & 7)) return SE_TILED_NOT_TAKEN;   // small / odd volumes: direct kernel
    if (a.cout & 15) return SE_TILED_NOT_TAKEN;               // planar 15-channel output layer: direct kernel
    const int nts = a.nts;
    if (ksize == 3) {
#ifdef SE_DEVTOOLS
        // BASELINE config 5 (LDS tile-size sweep, tools/bench_conv.py --variants 21,22,23): non-persistent LDS-tiled direct
        // kernel with 8x8x{4,8,16} output tiles = 38 / 64 / 115 KB of halo per 16-channel chunk
        if (g_variant >= 21 && g_variant <= 23 && nts % 2 == 0 && dim % 16 == 0) {
            if (g_variant == 21) return launch_tiled<3, 16, 4, 2>(a, batch, s);
            if (g_variant == 22) return launch_tiled<3, 16, 8, 2>(a, batch, s);
            return launch_tiled<3, 16, 16, 2>(a, batch, s);
        }
        if ((g_variant == 2 || g_variant == 3) && a.cout == 32 && dim >= 32 && (a.cin == 16 || a.cin == 32) && a.cin_pad == a.cin &&
            !(a.flags & (SE_EPI_RES_POST_RELU | SE_EPI_OUT_PLANAR))) {
            if (g_variant == 2)
                return a.cin == 32 ? launch_k3_c32_persistent<2, false>(a, batch, s) : launch_k3_c32_persistent<1, false>(a, batch, s);
            return a.cin == 32 ? launch_k3_c32_persistent<2, true>(a, batch, s) : launch_k3_c32_persistent<1, true>(a, batch, s);
        }
#endif
        // shapes no Winograd kernel covers (cout % 32 != 0, cin % 16 != 0): LDS-tiled direct kernel
        if (nts % 4 == 0) return launch_tiled<3, 16, 4, 4>(a, batch, s);
        if (nts % 2 == 0) return launch_tiled<3, 16, 4, 2>(a, batch, s);
        return launch_tiled<3, 16, 4, 1>(a, batch, s);
    }
    if (ksize == 7 && nts == 1) {
        if (g_variant == 0 || g_variant >= 10) {   // production: F(4,7) Winograd persistent kernel
            const int rc = se_conv3d_k7_wino_try(a, batch, s);
            if (rc != SE_TILED_NOT_TAKEN) return rc;
        }
        if (a.flags & SE_IN_PLANAR3) return SE_ERR_BAD_ARG;
#ifdef SE_DEVTOOLS
        if (g_variant == 2 && dim >= 32 && a.cout == 16 && !a.res && !(a.flags & SE_EPI_OUT_PLANAR)) return launch_k7_persistent(a, batch, s);
#endif
        return launch_tiled<7, 4, 4, 1>(a, batch, s);
    }
    return SE_TILED_NOT_TAKEN;
}

#ifdef SE_DEVTOOLS
extern "C" void se_debug_set_variant(int v) { g_variant = v; }
#endif
