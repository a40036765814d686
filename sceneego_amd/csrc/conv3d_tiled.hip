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

int g_variant = 0;   // debug/bench switch (se_debug_set_variant): 1 = disable the persistent 64^3 kernel

namespace {

int g_num_cus = 256;
void ensure_device_info() {
    static bool done = false;
    if (done) return;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
        g_num_cus = n;
    done = true;
}

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
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int grid = total_tiles < g_num_cus ? total_tiles : g_num_cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, s, a, tiles, ztiles, total_tiles, g_variant >= 10 ? g_variant - 10 : 0);
    SE_CHECK_LAUNCH();
    return 0;
}

template <int KS, int CK, int TZ, int N_T>
int launch_tiled(const ConvArgs& a, int batch, hipStream_t s) {
    using G = TileGeom<KS, CK, TZ>;
    const int tiles = a.dim / 8, ztiles = a.dim / TZ;
    dim3 grid((unsigned)(batch * ztiles * tiles * tiles), (unsigned)(a.nts / N_T));
    auto kern = conv3d_tiled_kernel<KS, CK, TZ, N_T>;
    if (G::LDS_BYTES > 48 * 1024) {
        static bool attr_set = false;  // per instantiation
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, s, a, tiles, ztiles);
    SE_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Returns 0 if a tiled kernel took the launch, SE_TILED_NOT_TAKEN if the shape is left to the direct kernel,
// otherwise the hipError_t of the failed launch.
int se_conv3d_tiled_try(const ConvArgs& a, int batch, int ksize, hipStream_t s) {
    const int dim = a.dim;
    if (dim < 16 || (dim & 7)) return SE_TILED_NOT_TAKEN;   // small / odd volumes: direct kernel
    if (a.cout & 15) return SE_TILED_NOT_TAKEN;               // planar 15-channel output layer: direct kernel
    const int nts = a.nts;
    if (ksize == 3) {
        if (g_variant != 1 && a.cout == 32 && dim >= 32 && (a.cin == 16 || a.cin == 32) && a.cin_pad == a.cin &&
            !(a.flags & (SE_EPI_RES_POST_RELU | SE_EPI_OUT_PLANAR))) {
            ensure_device_info();
            if (g_variant == 2)
                return a.cin == 32 ? launch_k3_c32_persistent<2, false>(a, batch, s) : launch_k3_c32_persistent<1, false>(a, batch, s);
            return a.cin == 32 ? launch_k3_c32_persistent<2, true>(a, batch, s) : launch_k3_c32_persistent<1, true>(a, batch, s);
        }
        if (nts % 4 == 0) return launch_tiled<3, 16, 4, 4>(a, batch, s);
        if (nts % 2 == 0) return launch_tiled<3, 16, 4, 2>(a, batch, s);
        return launch_tiled<3, 16, 4, 1>(a, batch, s);
    }
    if (ksize == 7 && nts == 1) return launch_tiled<7, 4, 4, 1>(a, batch, s);
    return SE_TILED_NOT_TAKEN;
}

extern "C" void se_debug_set_variant(int v) { g_variant = v; }
