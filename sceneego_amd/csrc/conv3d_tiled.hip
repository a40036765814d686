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
// ------------------------------------------------------------------------------------------------
// Persistent 3x3x3 kernel for the 64^3 level (cout = 32, cin = 16 or 32): the workhorse of V2V (10 launches,
// 46 % of all MACs).  One 512-thread workgroup per CU keeps ALL packed weights of the layer in LDS (54 KB per
// 16-channel chunk) next to one 6x10x10x16-channel halo tile (38.4 KB) and walks its share of the 4x8x8 output
// tiles.  The inner loop therefore touches LDS only (ds_read_b128 for both MFMA operands); the halo of the NEXT
// (tile, chunk) item is fetched global -> registers while the current item's 27 x 16 MFMAs per wave run, and is
// written to LDS between two barriers at the item boundary (issue-early / write-late staging).
// Wave w of 8: z-slab w>>1, rows 4*(w&1)..+3 -> two 16-voxel tiles x two 16-cout tiles = 4 accumulators.
// ------------------------------------------------------------------------------------------------
template <int CHUNKS, bool PIPE>
__global__ __launch_bounds__(512) void conv3d_k3_c32_persistent_kernel(ConvArgs a, int tiles_per_dim, int ztiles,
                                                                       int total_tiles, int diag) {
    constexpr int HY = 10, HX = 10, HV = 6 * 10 * 10;
    constexpr int W_FLOATS = CHUNKS * 27 * 2 * 256;
    constexpr int ITEMS4 = HV * 4;                       // 16-byte pieces of one halo chunk
    constexpr int PF = (ITEMS4 + 511) / 512;             // pieces per thread (5)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* tile = lds + W_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int vl = lane & 15;
    const int h = lane >> 4;
    const int dim = a.dim;

    // all weights of the layer -> LDS (packed order [chunk][tap][nt][lane][4] is already contiguous)
    for (int i = tid; i < W_FLOATS / 4; i += 512)
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(a.wpack)[i];

    int vbase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) vbase[m] = (((wave >> 1) * HY) + (wave & 1) * 4 + m * 2 + (vl >> 3)) * HX + (vl & 7);

    // per-thread halo piece coordinates (constant over items): piece it = tid + k*512 -> (halo voxel, quad)
    int p_off[PF];      // offset inside the sample of the piece's voxel relative to the tile origin voxel, or <0 if unused
    int p_hz[PF], p_hy[PF], p_hx[PF], p_q[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int it = tid + k * 512;
        const int hv = it >> 2;
        p_q[k] = it & 3;
        p_hx[k] = hv % HX;
        const int t2 = hv / HX;
        p_hy[k] = t2 % HY;
        p_hz[k] = t2 / HY;
        p_off[k] = it < ITEMS4 ? 0 : -1;
    }

    f32x4 pf[PF];
    auto fetch = [&](int item) {   // item = tile_index * CHUNKS + chunk
        const int ch = item % CHUNKS;
        int t = item / CHUNKS;
        const int tx = t % tiles_per_dim; t /= tiles_per_dim;
        const int ty = t % tiles_per_dim; t /= tiles_per_dim;
        const int tz = t % ztiles; t /= ztiles;
        const float* in_b = a.in + (size_t)t * dim * dim * dim * a.cin_pad + ch * 16;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int gz = tz * 4 - 1 + p_hz[k], gy = ty * 8 - 1 + p_hy[k], gx = tx * 8 - 1 + p_hx[k];
            pf[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p_off[k] == 0 && (unsigned)gz < (unsigned)dim && (unsigned)gy < (unsigned)dim && (unsigned)gx < (unsigned)dim)
                pf[k] = *reinterpret_cast<const f32x4*>(in_b + ((size_t)(gz * dim + gy) * dim + gx) * a.cin_pad + p_q[k] * 4);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < PF; ++k)
            if (p_off[k] == 0) *reinterpret_cast<f32x4*>(tile + (size_t)(tid + k * 512) * 4) = pf[k];
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // epilogue state (cout is 32: lane owns channels 4h..4h+3 of cout tile n)
    const bool relu = a.flags & SE_EPI_RELU;
    const bool use_res = (a.flags & SE_EPI_RES_PRE_RELU) && a.res;
    f32x4 bias[2];
    bias[0] = *reinterpret_cast<const f32x4*>(a.bpack + 4 * h);
    bias[1] = *reinterpret_cast<const f32x4*>(a.bpack + 16 + 4 * h);
    f32x4 resv[2][2];
    long long out_off[2] = {0, 0};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) resv[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int first_tile = blockIdx.x;
    if (first_tile >= total_tiles) return;
    int item = first_tile * CHUNKS;
    fetch(item);
    commit();
    __syncthreads();
    while (true) {
        // next item: next chunk of this tile, or chunk 0 of this workgroup's next tile
        int next = item + 1;
        if (next % CHUNKS == 0) next = (item / CHUNKS + (int)gridDim.x) * CHUNKS;
        const bool has_next = next < total_tiles * CHUNKS;
        if (has_next && !(diag & 1)) fetch(next);
        if (item % CHUNKS == CHUNKS - 1) {   // last chunk of the tile: output addresses + residual prefetch
            int t = item / CHUNKS;
            const int tx = t % tiles_per_dim; t /= tiles_per_dim;
            const int ty = t % tiles_per_dim; t /= tiles_per_dim;
            const int tz = t % ztiles; t /= ztiles;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int oz = tz * 4 + (wave >> 1);
                const int oy = ty * 8 + (wave & 1) * 4 + m * 2 + (vl >> 3);
                const int ox = tx * 8 + (vl & 7);
                out_off[m] = (((long long)t * dim + oz) * dim + oy) * dim * 32 + ox * 32 + 4 * h;
                if (use_res) {
                    resv[m][0] = *reinterpret_cast<const f32x4*>(a.res + out_off[m]);
                    resv[m][1] = *reinterpret_cast<const f32x4*>(a.res + out_off[m] + 16);
                }
            }
        }

        const int ch = item % CHUNKS;
        const f32x4* wrow = reinterpret_cast<const f32x4*>(wl) + (size_t)ch * 27 * 2 * 64 + lane;
#define SE_MFMA4(A, W, X)                                                      \
    A = __builtin_amdgcn_mfma_f32_16x16x4f32(W.x, X.x, A, 0, 0, 0);            \
    A = __builtin_amdgcn_mfma_f32_16x16x4f32(W.y, X.y, A, 0, 0, 0);            \
    A = __builtin_amdgcn_mfma_f32_16x16x4f32(W.z, X.z, A, 0, 0, 0);            \
    A = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w, X.w, A, 0, 0, 0);
#define SE_TAP_OFF(tap) ((((tap) / 9) * HY + ((tap) / 3) % 3) * HX + (tap) % 3) * 16 + 4 * h
        if (PIPE) {
            // software pipeline: the 4 operand reads of tap+1 are issued BEFORE the 16 MFMAs of tap, so the
            // LDS latency (~100-200 cycles with 8 waves reading) hides under 512 cycles of matrix work.
            f32x4 x0 = *reinterpret_cast<const f32x4*>(tile + vbase[0] * 16 + (SE_TAP_OFF(0)));
            f32x4 x1 = *reinterpret_cast<const f32x4*>(tile + vbase[1] * 16 + (SE_TAP_OFF(0)));
            f32x4 w0 = wrow[0];
            f32x4 w1 = wrow[64];
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the prologue's reads are their own group
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                f32x4 nx0 = x0, nx1 = x1, nw0 = w0, nw1 = w1;
                if (tap + 1 < 27) {
                    const int off = SE_TAP_OFF(tap + 1);
                    nx0 = *reinterpret_cast<const f32x4*>(tile + vbase[0] * 16 + off);
                    nx1 = *reinterpret_cast<const f32x4*>(tile + vbase[1] * 16 + off);
                    nw0 = wrow[((tap + 1) * 2 + 0) * 64];
                    nw1 = wrow[((tap + 1) * 2 + 1) * 64];
                }
                SE_MFMA4(acc[0][0], w0, x0)
                SE_MFMA4(acc[1][0], w0, x1)
                SE_MFMA4(acc[0][1], w1, x0)
                SE_MFMA4(acc[1][1], w1, x1)
                if (tap + 1 < 27) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 DS reads (tap+1) ...
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                    // ... then 16 MFMAs (tap)
                x0 = nx0; x1 = nx1; w0 = nw0; w1 = nw1;
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int off = SE_TAP_OFF(tap);
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(tile + vbase[0] * 16 + off);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(tile + vbase[1] * 16 + off);
                const f32x4 w0 = wrow[(tap * 2 + 0) * 64];
                const f32x4 w1 = wrow[(tap * 2 + 1) * 64];
                SE_MFMA4(acc[0][0], w0, x0)
                SE_MFMA4(acc[1][0], w0, x1)
                SE_MFMA4(acc[0][1], w1, x0)
                SE_MFMA4(acc[1][1], w1, x1)
            }
        }
#undef SE_TAP_OFF
#undef SE_MFMA4

        // ---- item boundary.  Order matters: vmcnt counts loads AND stores in issue order, so the wait that
        // guards commit() must come BEFORE this tile's output stores are issued; the stores then drain under the
        // next item's MFMAs instead of being waited for (measured: 6 % of the kernel when they were). ----
        const bool last_chunk = ch == CHUNKS - 1;
        if (has_next && !(diag & 1)) {
            __syncthreads();   // every wave is done reading the tile
            commit();          // waits for the prefetched halo (and residual) loads only
        }
        if (last_chunk && !(diag & 2)) {   // tile finished: epilogue from registers, reset accumulators
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4 v = acc[m][n] + bias[n];
                    if (use_res) v += resv[m][n];
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<f32x4*>(a.out + out_off[m] + n * 16) = v;
                    acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
        }
        if (!has_next) break;
        if (!(diag & 1)) __syncthreads();
        item = next;
    }
}

template <int CHUNKS, bool PIPE>
int launch_k3_c32_persistent(const ConvArgs& a, int batch, hipStream_t s) {
    constexpr int LDS_BYTES = (CHUNKS * 27 * 2 * 256 + 600 * 16) * 4;
    const int tiles = a.dim / 8, ztiles = a.dim / 4;
    const int total_tiles = batch * ztiles * tiles * tiles;
    auto kern = conv3d_k3_c32_persistent_kernel<CHUNKS, PIPE>;
    SE_ENSURE_LDS(kern, LDS_BYTES);
    const int num_cus = se_num_cus();
    const int grid = total_tiles < num_cus ? total_tiles : num_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, s, a, tiles, ztiles, total_tiles, g_variant >= 10 ? g_variant - 10 : 0);
    SE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Persistent 7x7x7 kernel (front conv of V2V: 33|65 -> 16 channels at 64^3, 32 % of all MACs).
// One 512-thread workgroup per CU.  LDS: the packed weights of ONE 4-channel chunk (86 tap groups x 1 KiB),
// TWO 10x14x14x4-channel halo tiles (double buffer, 31 KB each) and the 343-entry tap-offset table.
// Loop order is chunk-outer / tile-inner, so a chunk's weights are loaded once per workgroup and stay put; the
// price is that a tile's accumulators cannot live in registers across chunks: after every (tile, chunk) item
// the 16-byte accumulator fragments are stored to the output tensor (used as scratch) and re-loaded as the
// initial MFMA C operand when the same lane meets that tile again one chunk later — the summation order, and
// therefore the result, is bit-identical to one long register-resident chain.  32 KB of read-modify-write per
// 44k MFMA cycles is noise.  The next item's halo and partial sums are fetched into registers while the current
// item computes; commit-to-LDS goes to the other buffer, so there is ONE barrier per item.
// The last chunk of a 33/65-channel input holds 1 real channel: only MFMA j = 0 of each tap group is issued.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv3d_k7_persistent_kernel(ConvArgs a, int tiles_per_dim, int ztiles,
                                                                   int total_tiles, int diag) {
    constexpr int HZ = 10, HY = 14, HX = 14, HV = HZ * HY * HX;   // 1960 halo voxels, 16 B each
    constexpr int W_FLOATS = SE_K7_GROUPS * 256;
    constexpr int TILE_FLOATS = HV * 4;
    constexpr int PF = (HV + 511) / 512;                           // 4 pieces per thread
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* wl = lds;
    float* tiles = lds + W_FLOATS;                                 // 2 buffers
    int* toff = reinterpret_cast<int*>(lds + W_FLOATS + 2 * TILE_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int vl = lane & 15;
    const int h = lane >> 4;
    const int dim = a.dim;
    const int chunks = (a.cin + 3) >> 2;
    const int rem = a.cin & 3;                                     // real channels in the last chunk (0 = all 4)

    for (int t = tid; t < 4 * SE_K7_TSTRIDE; t += 512) {          // table [h][g]: offset (floats) of tap 4g+h
        const int hh = t / SE_K7_TSTRIDE, g = t - hh * SE_K7_TSTRIDE;
        int tap = 4 * g + hh;
        tap = tap < SE_K7_TAPS ? tap : 0;                          // pad taps: weights are zero
        const int kz = tap / 49, r = tap - kz * 49, ky = r / 7, kx = r - ky * 7;
        toff[t] = ((kz * HY + ky) * HX + kx) * 4;
    }

    int vbase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) vbase[m] = ((((wave >> 1) * HY) + (wave & 1) * 4 + m * 2 + (vl >> 3)) * HX + (vl & 7)) * 4;

    // this workgroup's tiles: blockIdx.x, blockIdx.x + gridDim.x, ...
    const int ntl = (total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (ntl <= 0) return;
    const int n_items = chunks * ntl;
    const bool keep_in_regs = ntl == 1;                            // same tile every item: no partial round trip

    int p_hz[PF], p_hy[PF], p_hx[PF];
    bool p_ok[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int hv = tid + k * 512;
        p_ok[k] = hv < HV;
        p_hx[k] = hv % HX;
        const int t2 = hv / HX;
        p_hy[k] = t2 % HY;
        p_hz[k] = t2 / HY;
    }

    auto tile_coords = [&](int item, int& b, int& tz, int& ty, int& tx) {
        int t = (int)blockIdx.x + (item % ntl) * (int)gridDim.x;
        tx = t % tiles_per_dim; t /= tiles_per_dim;
        ty = t % tiles_per_dim; t /= tiles_per_dim;
        tz = t % ztiles; t /= ztiles;
        b = t;
    };
    auto out_offset = [&](int item, int m) -> long long {
        int b, tz, ty, tx;
        tile_coords(item, b, tz, ty, tx);
        const int oz = tz * 4 + (wave >> 1);
        const int oy = ty * 8 + (wave & 1) * 4 + m * 2 + (vl >> 3);
        const int ox = tx * 8 + (vl & 7);
        return ((((long long)b * dim + oz) * dim + oy) * dim + ox) * 16 + 4 * h;
    };

    f32x4 pf[PF];
    auto fetch = [&](int item) {
        int b, tz, ty, tx;
        tile_coords(item, b, tz, ty, tx);
        const int ch = item / ntl;
        const float* in_b = a.in + (size_t)b * dim * dim * dim * a.cin_pad + ch * 4;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int gz = tz * 4 - 3 + p_hz[k], gy = ty * 8 - 3 + p_hy[k], gx = tx * 8 - 3 + p_hx[k];
            pf[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p_ok[k] && (unsigned)gz < (unsigned)dim && (unsigned)gy < (unsigned)dim && (unsigned)gx < (unsigned)dim)
                pf[k] = *reinterpret_cast<const f32x4*>(in_b + ((size_t)(gz * dim + gy) * dim + gx) * a.cin_pad);
        }
    };
    auto commit = [&](int buf) {
        float* tb = tiles + buf * TILE_FLOATS;
#pragma unroll
        for (int k = 0; k < PF; ++k)
            if (p_ok[k]) *reinterpret_cast<f32x4*>(tb + (size_t)(tid + k * 512) * 4) = pf[k];
    };
    auto load_weights = [&](int ch) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.wpack_b) + (size_t)ch * SE_K7_GROUPS * 64;
        for (int i = tid; i < SE_K7_GROUPS * 64; i += 512) reinterpret_cast<f32x4*>(wl)[i] = src[i];
    };

    const bool relu = a.flags & SE_EPI_RELU;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.bpack + 4 * h);
    f32x4 acc[2], pn[2];
    acc[0] = acc[1] = pn[0] = pn[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    fetch(0);
    commit(0);
    load_weights(0);
    __syncthreads();

    for (int item = 0; item < n_items; ++item) {
        const int ch = item / ntl;
        const bool has_next = item + 1 < n_items;
        const int ch_next = (item + 1) / ntl;
        if (has_next) {
            if (!(diag & 1)) fetch(item + 1);
            if (ch_next > 0 && !keep_in_regs && !(diag & 2)) {
                pn[0] = *reinterpret_cast<const f32x4*>(a.out + out_offset(item + 1, 0));
                pn[1] = *reinterpret_cast<const f32x4*>(a.out + out_offset(item + 1, 1));
            }
        }

        const float* tb = tiles + (item & 1) * TILE_FLOATS;
        const int nj = (ch == chunks - 1 && rem != 0) ? rem : 4;   // uniform
        const f32x4* wrow = reinterpret_cast<const f32x4*>(wl) + lane;
        // Per-lane LDS offsets of tap 4g+h come from a table stored [h][g], so ONE ds_read_b128 yields the offsets of 4
        // consecutive groups; it is fetched one 4-group block ahead, and the operands of group g+1 are read BEFORE
        // the MFMAs of group g (ping-pong register sets A/B), so no LDS latency sits in front of an MFMA.
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const i32x4* otab = reinterpret_cast<const i32x4*>(toff + h * SE_K7_TSTRIDE);
#define SE_K7_LOAD(X0, X1, W, OFF, G)                                                   \
    X0 = *reinterpret_cast<const f32x4*>(tb + vbase[0] + (OFF));                        \
    X1 = *reinterpret_cast<const f32x4*>(tb + vbase[1] + (OFF));                        \
    W = wrow[(G) * 64];
#define SE_K7_MFMA(X0, X1, W, NJ)                                                       \
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.x, X0.x, acc[0], 0, 0, 0);          \
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.x, X1.x, acc[1], 0, 0, 0);          \
    if (NJ > 1) {                                                                       \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.y, X0.y, acc[0], 0, 0, 0);      \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.y, X1.y, acc[1], 0, 0, 0);      \
    }                                                                                   \
    if (NJ > 2) {                                                                       \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.z, X0.z, acc[0], 0, 0, 0);      \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.z, X1.z, acc[1], 0, 0, 0);      \
    }                                                                                   \
    if (NJ > 3) {                                                                       \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w, X0.w, acc[0], 0, 0, 0);      \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W.w, X1.w, acc[1], 0, 0, 0);      \
    }
#define SE_K7_SCHED(NJ)                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);          \
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * (NJ), 0);
        f32x4 xa0, xa1, wa, xb0, xb1, wb;
        i32x4 o = otab[0];
        SE_K7_LOAD(xa0, xa1, wa, o.x, 0)
        auto body = [&](auto nj_tag) {
            constexpr int NJ = decltype(nj_tag)::value;
            for (int g4 = 0; g4 < SE_K7_GROUPS / 4; ++g4) {     // 21 blocks of 4 groups; groups 84, 85 below
                const int g = 4 * g4;
                const i32x4 on = otab[g4 + 1];
                SE_K7_LOAD(xb0, xb1, wb, o.y, g + 1)
                SE_K7_MFMA(xa0, xa1, wa, NJ)
                SE_K7_SCHED(NJ)
                SE_K7_LOAD(xa0, xa1, wa, o.z, g + 2)
                SE_K7_MFMA(xb0, xb1, wb, NJ)
                SE_K7_SCHED(NJ)
                SE_K7_LOAD(xb0, xb1, wb, o.w, g + 3)
                SE_K7_MFMA(xa0, xa1, wa, NJ)
                SE_K7_SCHED(NJ)
                SE_K7_LOAD(xa0, xa1, wa, on.x, g + 4)
                SE_K7_MFMA(xb0, xb1, wb, NJ)
                SE_K7_SCHED(NJ)
                o = on;
            }
            SE_K7_LOAD(xb0, xb1, wb, o.y, SE_K7_GROUPS - 1)
            SE_K7_MFMA(xa0, xa1, wa, NJ)
            SE_K7_MFMA(xb0, xb1, wb, NJ)
        };
        if (nj == 4) body(std::integral_constant<int, 4>{});
        else if (nj == 1) body(std::integral_constant<int, 1>{});      // 33 / 65 input channels: the occupancy chunk
        else if (nj == 2) body(std::integral_constant<int, 2>{});
        else body(std::integral_constant<int, 3>{});
#undef SE_K7_LOAD
#undef SE_K7_MFMA
#undef SE_K7_SCHED

        // item boundary: commit the prefetched halo first (its vmcnt wait must not cover this item's stores)
        if (has_next && !(diag & 1)) commit((item + 1) & 1);
        if (diag & 2) {
        } else if (ch == chunks - 1) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x4 v = acc[m] + bias;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<f32x4*>(a.out + out_offset(item, m)) = v;
            }
        } else if (!keep_in_regs) {
            *reinterpret_cast<f32x4*>(a.out + out_offset(item, 0)) = acc[0];
            *reinterpret_cast<f32x4*>(a.out + out_offset(item, 1)) = acc[1];
        }
        if (!has_next) break;
        if (!keep_in_regs) {
            if (ch_next > 0) { acc[0] = pn[0]; acc[1] = pn[1]; }
            else acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (!(diag & 4)) __syncthreads();   // all waves: done reading this tile buffer and the weights; next buffer is complete
        if (ch_next != ch) {
            load_weights(ch_next);
            __syncthreads();
        }
    }
}

int launch_k7_persistent(const ConvArgs& a, int batch, hipStream_t s) {
    constexpr int LDS_BYTES = (SE_K7_GROUPS * 256 + 2 * 1960 * 4) * 4 + 4 * SE_K7_TSTRIDE * 4;
    const int tiles = a.dim / 8, ztiles = a.dim / 4;
    const int total_tiles = batch * ztiles * tiles * tiles;
    SE_ENSURE_LDS(conv3d_k7_persistent_kernel, LDS_BYTES);
    const int num_cus = se_num_cus();
    const int grid = total_tiles < num_cus ? total_tiles : num_cus;
    hipLaunchKernelGGL(conv3d_k7_persistent_kernel, dim3(grid), dim3(512), LDS_BYTES, s, a, tiles, ztiles, total_tiles,
                       g_variant >= 10 ? g_variant - 10 : 0);
    SE_CHECK_LAUNCH();
    return 0;
}

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
int se_conv3d_wino2d_try(const ConvArgs& a, int batch, hipStream_t s); // conv3d_wino2d.hip
int se_conv3d_k7_wino_try(const ConvArgs& a, int batch, hipStream_t s);

static int tiled_try_one(const ConvArgs& a, int batch, int ksize, hipStream_t s);

// The persistent kernels keep a per-workgroup table of their work units in the LDS left over beside weights and tiles
// (a few hundred entries): large batches are cut into slices of 32 samples, one launch each (weak-scaling config 4 runs
// 32 samples per GPU, i.e. exactly one slice).
int se_conv3d_tiled_try(const ConvArgs& a, int batch, int ksize, hipStream_t s) {
    // unit-table budget: the smallest table among the persistent kernels holds ~430 entries per workgroup
    const long long t8 = a.dim / 8;
    const long long units_per_sample = ksize == 7 ? t8 * t8 * t8 : (long long)(a.cout >= 32 ? a.cout / 32 : 1) * (a.dim / 4) * t8 * t8;
    long long budget = units_per_sample > 0 ? 400LL * se_num_cus() / units_per_sample : 32;
    const int SLICE = (int)(budget < 1 ? 1 : budget > 32 ? 32 : budget);
    if (batch <= SLICE) return tiled_try_one(a, batch, ksize, s);
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
        const int rc = tiled_try_one(sl, nb, ksize, s);
        if (rc != 0) return rc;   // not taken (same decision for every slice: nothing launched yet) or an error
    }
    return 0;
}

static int tiled_try_one(const ConvArgs& a, int batch, int ksize, hipStream_t s) {
    const int dim = a.dim;
    if (ksize == 3 && (g_variant == 0 || (g_variant >= 40 && g_variant < 70))) {   // production: 2-D Winograd F(4,3) x F(2,3), register accumulators
        const int rc = se_conv3d_wino2d_try(a, batch, s);
        if (rc != SE_TILED_NOT_TAKEN) return rc;
    }
    // octet-planar tensors, the pooled second output and the fused 16-channel skip convolution exist in the 2-D Winograd kernel only:
    // a launch that asks for one of them and was declined (cin_pad != cin, SE_EPI_RES_POST_RELU / SE_EPI_OUT_PLANAR, ...) is an error -
    // none of the kernels below would read or write those tensors the way the caller laid them out
    if ((a.flags & (SE_IN_OCTET | SE_OUT_OCTET | SE_RES_OCTET | SE_EPI_SKIPCONV16)) || a.pool_out || a.skip_w) return SE_ERR_BAD_ARG;
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
