// LDS-tiled bf16 convolutions for the large V2V levels (see conv3d_bf16.hip for the packer and the direct form).
//
// Both kernels stage (halo tile + weight chunk) global -> registers -> LDS with ALL loads of a stage issued before the first
// LDS store (a load/store-per-iteration loop exposes one global latency per piece), prefetch the NEXT stage into registers
// while the MFMAs of the current one run, and double-buffer the MFMA operand fragments across k steps.
#include "bf16_common.h"

#define SE_TILED_NOT_TAKEN_B (-1000)

namespace {

__device__ __forceinline__ u16x8 lds_read16(const unsigned char* p) { return *reinterpret_cast<const u16x8*>(p); }
__device__ __forceinline__ u16x8 zero8() { return (u16x8){0, 0, 0, 0, 0, 0, 0, 0}; }

// ------------------------------------------------------------------------------------------------
// 3x3x3, dim % 16 == 0, cin % 16 == 0, cout % 32 == 0.
// Workgroup = 4 waves, output tile 4(x) x 8(y) x 16(z) voxels x 32 couts; wave w owns x = w, its 8 voxel tiles are the
// y rows (16 z each), so a lane's B address is   halo[(w+dx)][(n+dy)][(v+dz)]   and advances by one row per tile.
// Per 16-channel chunk the LDS holds the 6x10x18 halo (32 B / voxel) and the 14 k steps x 2 cout tiles of weights.
// A k step = 2 taps x 2 octets: lane group g reads tap 2s + (g >> 1), octet g & 1.  The two lane groups that share a
// ds_read_b128 pass (g = 0,1 and g = 2,3) read the same tap, so the 16 lanes cover 16 distinct 16-byte slots
// (tools/lds_conflicts_bf16.py).
// ------------------------------------------------------------------------------------------------
constexpr int K3_TX = 4, K3_TY = 8, K3_TZ = 16;
constexpr int K3_HX = K3_TX + 2, K3_HY = K3_TY + 2, K3_HZ = K3_TZ + 2;
constexpr int K3_HALO_VOX = K3_HX * K3_HY * K3_HZ;            // 1080
constexpr int K3_HALO_PIECES = K3_HALO_VOX * 2;               // 2160 x 16 B
[[maybe_unused]] constexpr int K3_HALO_BYTES = K3_HALO_PIECES * 16;            // 34560
constexpr int K3_KPC = 14;
constexpr int K3_W_PIECES = K3_KPC * 2 * 64;                  // 1792 x 16 B
constexpr int K3_W_BYTES = K3_W_PIECES * 16;                  // 28672
constexpr int K3_WP = K3_W_PIECES / 256;                      // 7 weight pieces per thread

template <int TY>
struct K3G {
    static constexpr int HY = TY + 2;
    static constexpr int HALO_VOX = K3_HX * HY * K3_HZ;
    static constexpr int HALO_PIECES = HALO_VOX * 2;
    static constexpr int HALO_BYTES = HALO_PIECES * 16;
    static constexpr int LDS_BYTES = HALO_BYTES + K3_W_BYTES;      // TY = 8: 63232 (2 WG/CU); TY = 4: 49408 (3 WG/CU)
    static constexpr int HP = (HALO_PIECES + 255) / 256;
};

template <int TY>
struct K3Stage {
    u16x8 h[K3G<TY>::HP];
    u16x8 w[K3_WP];
};

template <int TY>
__device__ __forceinline__ void k3_load_stage(K3Stage<TY>& st, const ConvBArgs& a, const int* hvox, unsigned hmask,
                                              const unsigned short* wsrc, int c, int tid) {
    const int coff = c * 16 + (tid & 1) * 8;      // piece i = tid + 256 j: octet i & 1 = tid & 1
#pragma unroll
    for (int j = 0; j < K3G<TY>::HP; ++j) {      // branch-free: out-of-volume pieces read voxel 0 and are zeroed by a select
        const u16x8 val = *reinterpret_cast<const u16x8*>(a.in + (long long)hvox[j] * a.cin_pad + coff);
        st.h[j] = ((hmask >> j) & 1) ? val : zero8();
    }
#pragma unroll
    for (int j = 0; j < K3_WP; ++j) {
        const int i = tid + 256 * j;
        const int m = i / (K3_KPC * 64), r = i - m * (K3_KPC * 64);
        st.w[j] = *reinterpret_cast<const u16x8*>(wsrc + ((size_t)m * a.ksteps + c * K3_KPC) * 512 + r * 8);
    }
}

template <int TY>
__device__ __forceinline__ void k3_commit_stage(const K3Stage<TY>& st, unsigned char* halo, unsigned char* wts, int tid) {
#pragma unroll
    for (int j = 0; j < K3G<TY>::HP; ++j) {
        const int i = tid + 256 * j;
        if (i < K3G<TY>::HALO_PIECES) *reinterpret_cast<u16x8*>(halo + i * 16) = st.h[j];
    }
#pragma unroll
    for (int j = 0; j < K3_WP; ++j) *reinterpret_cast<u16x8*>(wts + (tid + 256 * j) * 16) = st.w[j];
}

// the 14 k steps of one chunk; operand fragments double-buffered, next k step's reads spread between this one's MFMAs
template <int TY>
__device__ __forceinline__ void k3_compute(f32x4 (&acc)[2][TY], const unsigned char* brow, const unsigned char* arow, int g) {
    u16x8 A0, A1, Bf[TY];
    {
        const int tap = g >> 1;
        const unsigned char* bp = brow + (((tap / 9) * K3G<TY>::HY + (tap / 3) % 3) * K3_HZ + tap % 3) * 32;
        A0 = lds_read16(arow);
        A1 = lds_read16(arow + K3_KPC * 1024);
#pragma unroll
        for (int n = 0; n < TY; ++n) Bf[n] = lds_read16(bp + n * (K3_HZ * 32));
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + TY, 0);
    }
#pragma unroll
    for (int sl = 0; sl < K3_KPC; ++sl) {
        u16x8 nA0 = A0, nA1 = A1, nB[TY];
#pragma unroll
        for (int n = 0; n < TY; ++n) nB[n] = Bf[n];
        if (sl + 1 < K3_KPC) {
            int tap = 2 * (sl + 1) + (g >> 1);
            tap = tap > 26 ? 26 : tap;             // padding group: zero weights, any valid address
            const unsigned char* bp = brow + (((tap / 9) * K3G<TY>::HY + (tap / 3) % 3) * K3_HZ + tap % 3) * 32;
            nA0 = lds_read16(arow + (sl + 1) * 1024);
            nA1 = lds_read16(arow + (K3_KPC + sl + 1) * 1024);
#pragma unroll
            for (int n = 0; n < TY; ++n) nB[n] = lds_read16(bp + n * (K3_HZ * 32));
        }
#pragma unroll
        for (int n = 0; n < TY; ++n) {
            acc[0][n] = mfma_bf16(A0, Bf[n], acc[0][n]);
            acc[1][n] = mfma_bf16(A1, Bf[n], acc[1][n]);
        }
        A0 = nA0; A1 = nA1;
#pragma unroll
        for (int n = 0; n < TY; ++n) Bf[n] = nB[n];
        if (sl + 1 < K3_KPC) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int n = 0; n < TY; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TY, 0);
        }
    }
}

#ifdef SE_STAMPB
#define K3_STAMP(i) do { if (stamps && lane == 0) stamps[((size_t)blockIdx.x * 4 + w) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define K3_STAMP_ARG , unsigned long long* stamps
#else
#define K3_STAMP(i) do {} while (0)
#define K3_STAMP_ARG
#endif

template <int TY>
__global__ __launch_bounds__(256, TY == 8 ? 2 : 3) void conv_bf16_k3_kernel(ConvBArgs a, int tiles_x, int tiles_y, int tiles_z K3_STAMP_ARG) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* halo = lds;
    unsigned char* wts = lds + K3G<TY>::HALO_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int v = lane & 15, g = lane >> 4;
    const int D = a.dim;
    const int mb = blockIdx.y;
    int t = se_xcd_tile_bf16((int)blockIdx.x, (int)gridDim.x);
    const int tz = t % tiles_z; t /= tiles_z;
    const int ty = t % tiles_y; t /= tiles_y;
    const int tx = t % tiles_x;
    const int b = t / tiles_x;
    const int x0 = tx * K3_TX, y0 = ty * TY, z0 = tz * K3_TZ;

    // this thread's halo pieces: global voxel index and in-volume mask, fixed for the whole tile
    int hoff[K3G<TY>::HP];
    unsigned hmask = 0;
#pragma unroll
    for (int j = 0; j < K3G<TY>::HP; ++j) {
        const int i = tid + 256 * j;
        const int hv = i >> 1;
        const int hz = hv % K3_HZ, hy = (hv / K3_HZ) % K3G<TY>::HY, hx = hv / (K3_HZ * K3G<TY>::HY);
        const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
        const bool ok = i < K3G<TY>::HALO_PIECES && (unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && (unsigned)gz < (unsigned)D;
        hoff[j] = ok ? ((b * D + gx) * D + gy) * D + gz : 0;
        hmask |= (ok ? 1u : 0u) << j;
    }
    const unsigned short* wsrc = a.wpack + ((size_t)mb * 2 * a.ksteps) * 512;

    f32x4 acc[2][TY];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < TY; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned char* brow = halo + ((w * K3G<TY>::HY) * K3_HZ + v) * 32 + (g & 1) * 16;
    const unsigned char* arow = wts + lane * 16;

    const long long obase = (((long long)b * D + (x0 + w)) * D + y0) * D + (z0 + v);   // output voxel of tile n: + n * D
    const bool has_res = epi_has_res(a);

    K3Stage<TY> st;
    K3_STAMP(0);
    k3_load_stage<TY>(st, a, hoff, hmask, wsrc, 0, tid);
    K3_STAMP(1);
    for (int c = 0; c + 1 < a.nchunk; ++c) {
        __syncthreads();                       // every wave is done reading the previous chunk
        k3_commit_stage<TY>(st, halo, wts, tid);
        __syncthreads();
        K3_STAMP(2);
        k3_load_stage<TY>(st, a, hoff, hmask, wsrc, c + 1, tid);   // in flight under the MFMAs below
        k3_compute<TY>(acc, brow, arow, g);
        K3_STAMP(3);
    }
    // last chunk: the staging registers are free; fetch bias and the skip tensor under the MFMAs instead
    __syncthreads();
    k3_commit_stage<TY>(st, halo, wts, tid);
    __syncthreads();
    K3_STAMP(4);
    const EpiBias8 ebias = epi_load_bias8(a, mb, g);
    u16x8 rv[TY];
#pragma unroll
    for (int n = 0; n < TY; ++n) {
        rv[n] = zero8();
        // skip tensor: read once, by this launch only -> non-temporal (round 5: -2.5 % / -5 % per launch with the non-temporal stores of the epilogue)
        if (has_res) rv[n] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(a.res + (obase + (long long)n * D) * a.cout + mb * 32 + 8 * g));
    }
    k3_compute<TY>(acc, brow, arow, g);
    K3_STAMP(5);
#pragma unroll
    for (int n = 0; n < TY; ++n)
        epi_store_pair(a, acc[0][n], acc[1][n], ebias, has_res, rv[n], (obase + (long long)n * D) * a.cout + mb * 32 + 8 * g);
    K3_STAMP(6);
}

#ifdef SE_DEVTOOLS   // retired A/B variant: persistent double-buffered form of the bf16 3x3x3 kernel
#include "devtools/bf16_persistent_k3_kernel.inc"
#endif  // SE_DEVTOOLS (persistent bf16 3x3x3 kernel)

// ------------------------------------------------------------------------------------------------
// 7x7x7 front layer, cout = 16, dim % 8 == 0, octet-planar input [B][octs][D^3][8].
// Workgroup = 4 waves, output tile 8x8x8; wave w owns x in {2w, 2w+1}; its 8 voxel tiles are (x, y pair): a tile's 16
// columns are 2 y rows x 8 z.  Octet-outer: the LDS holds the 14x14x14 halo of ONE 8-channel octet (16 B / voxel, z pitch
// 24 voxels) and, double-buffered, the 13 k steps (52 tap slots) of one dz plane of weights; the accumulators stay in
// registers over all octets.  Tap slots are paired so that lane groups sharing a ds_read_b128 pass read addresses that
// differ by a multiple of 256 B (bf16_common.h, tools/lds_conflicts_bf16.py: conflict-free).
// ------------------------------------------------------------------------------------------------
// Two tile geometries: ROW16 = true: tile 8(x) x 4(y) x 16(z), a voxel tile = one row of 16 z (halo 14x10x22, 80 KB of LDS with
// the weight buffers -> 2 workgroups per CU; needs dim % 16 == 0); ROW16 = false: tile 8x8x8, a voxel tile = 2 y rows x 8 z
// (halo 14^3, 100 KB -> 1 workgroup per CU; dim % 8 == 0).  z pitch 24 voxels in both.
constexpr int K7_P = 24;
constexpr int K7_KPD = SE_K7B_SLOTS_PER_DZ / 4;               // 13 k steps per dz plane
constexpr int K7_WBUF_BYTES = K7_KPD * 1024;                  // 13312
constexpr int K7_WPIECES = K7_WBUF_BYTES / 16;                // 832
constexpr int K7_WP = (K7_WPIECES + 255) / 256;               // 4

template <bool ROW16>
struct K7Geo {
    static constexpr int TX = 8, TY = ROW16 ? 4 : 8, TZ = ROW16 ? 16 : 8;
    static constexpr int HX = TX + 6, HY = TY + 6, HZ = TZ + 6;
    static constexpr int PIECES = HX * HY * HZ;                       // 16-byte records
    static constexpr int HALO_BYTES = HX * HY * K7_P * 16;
    static constexpr int LDS_BYTES = HALO_BYTES + 2 * K7_WBUF_BYTES;  // 80384 / 101888
    static constexpr int HP = (PIECES + 255) / 256;                   // 13 / 11 pieces per thread
    // voxel tile n of wave w, lane column v -> (x, y, z) inside the output tile
    __device__ static __forceinline__ int tx(int w, int n) { return 2 * w + (n >> 2); }
    __device__ static __forceinline__ int ty(int n, int v) { return ROW16 ? (n & 3) : 2 * (n & 3) + (v >> 3); }
    __device__ static __forceinline__ int tz(int v) { return ROW16 ? v : (v & 7); }
    // LDS byte offset of tile n relative to tile 0 of the same wave (compile-time per n)
    __device__ static __forceinline__ int tile_off(int n) { return (((n >> 2) * HY + (ROW16 ? 1 : 2) * (n & 3)) * K7_P) * 16; }
};

template <bool ROW16>
__device__ __forceinline__ int k7_tap_offset(int sl, int g) {
    int r = 4 * sl + g;
    r = r > 48 ? 48 : r;                       // padding slots alias the last tap (zero weights)
    const int r2 = r < 28 ? r : r - 28;
    const int q7 = r2 / 7;
    const int dx = r2 - 7 * q7, dy = 2 * q7 + (r < 28 ? 0 : 1);
    return ((dx * K7Geo<ROW16>::HY + dy) * K7_P) * 16;
}

template <bool ROW16>
__global__ __launch_bounds__(256, ROW16 ? 2 : 1) void conv_bf16_k7_kernel(ConvBArgs a, int tiles_x, int tiles_y, int tiles_z) {
    using G = K7Geo<ROW16>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* halo = lds;
    unsigned char* wbuf = lds + G::HALO_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int v = lane & 15, g = lane >> 4;
    const int D = a.dim;
    const int octs = a.nchunk;
    int t = se_xcd_tile_bf16((int)blockIdx.x, (int)gridDim.x);
    const int tz = t % tiles_z; t /= tiles_z;
    const int ty = t % tiles_y; t /= tiles_y;
    const int tx = t % tiles_x;
    const int b = t / tiles_x;
    const int x0 = tx * G::TX, y0 = ty * G::TY, z0 = tz * G::TZ;
    const long long N = (long long)D * D * D;
    const unsigned short* inb = a.in + (long long)b * octs * N * 8;

    f32x4 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned char* lbase = halo + (((2 * w) * G::HY + G::ty(0, v)) * K7_P + G::tz(v)) * 16;

    // halo staging: piece i = tid + 256 j -> (hx, hy, hz); addresses are recomputed per use (cheap) to save registers
    auto halo_load = [&](u16x8 (&hreg)[G::HP], const unsigned short* plane) {
#pragma unroll
        for (int j = 0; j < G::HP; ++j) {
            const int i = tid + 256 * j;
            const int hz = i % G::HZ, hy = (i / G::HZ) % G::HY, hx = i / (G::HZ * G::HY);
            const int gx = x0 + hx - 3, gy = y0 + hy - 3, gz = z0 + hz - 3;
            const bool ok = i < G::PIECES && (unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && (unsigned)gz < (unsigned)D;
            const u16x8 val = *reinterpret_cast<const u16x8*>(plane + (ok ? ((gx * D + gy) * D + gz) * 8 : 0));
            hreg[j] = ok ? val : zero8();
        }
    };
    auto halo_commit = [&](const u16x8 (&hreg)[G::HP]) {
#pragma unroll
        for (int j = 0; j < G::HP; ++j) {
            const int i = tid + 256 * j;
            const int hz = i % G::HZ, hy = (i / G::HZ) % G::HY, hx = i / (G::HZ * G::HY);
            if (i < G::PIECES) *reinterpret_cast<u16x8*>(halo + ((hx * G::HY + hy) * K7_P + hz) * 16) = hreg[j];
        }
    };

    u16x8 hreg[G::HP], wreg[K7_WP];
    halo_load(hreg, inb);
#pragma unroll
    for (int j = 0; j < K7_WP; ++j) {
        const int i = tid + 256 * j;
        wreg[j] = zero8();
        if (i < K7_WPIECES) wreg[j] = *reinterpret_cast<const u16x8*>(a.wpack + (size_t)i * 8);
    }
    int cur = 0;
    const int phases = octs * 7;
    for (int c = 0; c < octs; ++c) {
        __syncthreads();                      // previous octet fully consumed
        halo_commit(hreg);
        if (c == 0) {
#pragma unroll
            for (int j = 0; j < K7_WP; ++j) {
                const int i = tid + 256 * j;
                if (i < K7_WPIECES) *reinterpret_cast<u16x8*>(wbuf + i * 16) = wreg[j];
            }
        }
        __syncthreads();
        for (int dz = 0; dz < 7; ++dz) {
            const int ph = c * 7 + dz + 1;
            const bool has_next = ph < phases;
            if (has_next) {
                const unsigned short* src = a.wpack + (size_t)ph * K7_KPD * 512;
#pragma unroll
                for (int j = 0; j < K7_WP; ++j) {
                    const int i = tid + 256 * j;
                    if (i < K7_WPIECES) wreg[j] = *reinterpret_cast<const u16x8*>(src + (size_t)i * 8);
                }
            }
            if (dz == 6 && c + 1 < octs) halo_load(hreg, inb + (long long)(c + 1) * N * 8);   // in flight under the last dz plane
            const unsigned char* wb = wbuf + cur * K7_WBUF_BYTES + lane * 16;
            const unsigned char* lz = lbase + dz * 16;
            u16x8 A, Bf[8];
            {
                const unsigned char* bp = lz + k7_tap_offset<ROW16>(0, g);
                A = lds_read16(wb);
#pragma unroll
                for (int n = 0; n < 8; ++n) Bf[n] = lds_read16(bp + G::tile_off(n));
                __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
            }
#pragma unroll
            for (int sl = 0; sl < K7_KPD; ++sl) {
                u16x8 nA = A, nB[8];
#pragma unroll
                for (int n = 0; n < 8; ++n) nB[n] = Bf[n];
                if (sl + 1 < K7_KPD) {
                    const unsigned char* bp = lz + k7_tap_offset<ROW16>(sl + 1, g);
                    nA = lds_read16(wb + (sl + 1) * 1024);
#pragma unroll
                    for (int n = 0; n < 8; ++n) nB[n] = lds_read16(bp + G::tile_off(n));
                }
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[n] = mfma_bf16(A, Bf[n], acc[n]);
                A = nA;
#pragma unroll
                for (int n = 0; n < 8; ++n) Bf[n] = nB[n];
                if (sl + 1 < K7_KPD) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int n = 0; n < 7; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                }
            }
            if (has_next) {
#pragma unroll
                for (int j = 0; j < K7_WP; ++j) {
                    const int i = tid + 256 * j;
                    if (i < K7_WPIECES) *reinterpret_cast<u16x8*>(wbuf + (cur ^ 1) * K7_WBUF_BYTES + i * 16) = wreg[j];
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.bpack + 4 * g);
    const bool relu = a.flags & SE_EPI_RELU;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int x = x0 + G::tx(w, n), y = y0 + G::ty(n, v), z = z0 + G::tz(v);
        const long long ovox = (((long long)b * D + x) * D + y) * D + z;
        f32x4 r = acc[n] + bias4;
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        u16x4 o;
        o[0] = f2bf(r.x); o[1] = f2bf(r.y); o[2] = f2bf(r.z); o[3] = f2bf(r.w);
        *reinterpret_cast<u16x4*>(a.out + ovox * 16 + 4 * g) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// 7x7x7 front layer, row-reuse form (round 5; tile 8(x) x 4(y) x 16(z), dim % 16 == 0, cout = 16, octet-planar input).
// conv_bf16_k7_kernel<true> issues 8 B-fragment reads + 1 A read per 8 MFMAs and a weight phase (stage + barrier) per dz plane.  The B
// fragment of voxel row (x, y) for tap (dx, dy, dz) IS the fragment of row (x, y + 1) for tap (dx, dy - 1, dz): with the k groups made of
// (dx, dz) slots only (bf16_common.h: se_k7r_slot; 14 groups of 4 slots for the 49 combinations) a wave reads the 10 halo rows of an
// x once per group and uses row r for every (y, dy) with y + dy = r: 20 B reads per 56 MFMAs (0.36 per MFMA instead of 1.0) for 8 %
// more MFMAs (56 slots per dy instead of 52 per dz).  The seven weight fragments of a k group come straight from global memory one
// group ahead (the same 7 KB for every wave of the launch: cache hits) - no weight image in LDS, no barrier inside an octet.
// 33 -> 16 @64^3, B = 32 (tools/diag/bf16_ab.py): dz-phase kernel 3.06-3.13 ms; this walk with the weights staged through LDS (a
// barrier and an exposed load per group) 3.02; weights from global 2.67-2.82.  Knock-outs of this form: no halo staging after the
// first octet 2.36, no weight loads 2.49, no B reads 2.39, none of the three 2.21 = the MFMA stream at the clock it gets.
// ------------------------------------------------------------------------------------------------
constexpr int K7R_LDS_BYTES = K7Geo<true>::HALO_BYTES;   // 53760: the halo of one octet; two workgroups per CU

__global__ __launch_bounds__(256, 2) void conv_bf16_k7r_kernel(ConvBArgs a, int tiles_x, int tiles_y, int tiles_z) {
    using G = K7Geo<true>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* halo = lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int v = lane & 15, g = lane >> 4;
    const int D = a.dim;
    const int octs = a.nchunk;
    int t = se_xcd_tile_bf16((int)blockIdx.x, (int)gridDim.x);
    const int tz = t % tiles_z; t /= tiles_z;
    const int ty = t % tiles_y; t /= tiles_y;
    const int tx = t % tiles_x;
    const int b = t / tiles_x;
    const int x0 = tx * G::TX, y0 = ty * G::TY, z0 = tz * G::TZ;
    const long long N = (long long)D * D * D;
    const unsigned short* inb = a.in + (long long)b * octs * N * 8;
    // section R behind section A (cout <= 16: one cout tile): [octet][q][dy][lane][8]; a k group's seven fragments come straight from
    // global memory (every wave of the launch reads the same 7 KB per group: L2 / vector-cache hits), one group ahead of their use
    const unsigned short* wr = a.wpack + (size_t)a.ksteps * 512 + lane * 8;

    f32x4 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto halo_load = [&](u16x8 (&hreg)[G::HP], const unsigned short* plane) {
#pragma unroll
        for (int j = 0; j < G::HP; ++j) {
            const int i = tid + 256 * j;
            const int hz = i % G::HZ, hy = (i / G::HZ) % G::HY, hx = i / (G::HZ * G::HY);
            const int gx = x0 + hx - 3, gy = y0 + hy - 3, gz = z0 + hz - 3;
            const bool ok = i < G::PIECES && (unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && (unsigned)gz < (unsigned)D;
            const u16x8 val = *reinterpret_cast<const u16x8*>(plane + (ok ? ((gx * D + gy) * D + gz) * 8 : 0));
            hreg[j] = ok ? val : zero8();
        }
    };
    auto halo_commit = [&](const u16x8 (&hreg)[G::HP]) {
#pragma unroll
        for (int j = 0; j < G::HP; ++j) {
            const int i = tid + 256 * j;
            const int hz = i % G::HZ, hy = (i / G::HZ) % G::HY, hx = i / (G::HZ * G::HY);
            if (i < G::PIECES) *reinterpret_cast<u16x8*>(halo + ((hx * G::HY + hy) * K7_P + hz) * 16) = hreg[j];
        }
    };
    auto a_load = [&](u16x8 (&A)[7], int ph) {
        const unsigned short* src = wr + (size_t)ph * 7 * 512;
#pragma unroll
        for (int dy = 0; dy < 7; ++dy) A[dy] = *reinterpret_cast<const u16x8*>(src + dy * 512);
    };
    // rows (xi, r), r = 0..9: fragment of halo row r of x = 2 w + xi; used by the tiles y = r - dy, 0 <= y < 4, 0 <= dy < 7
    auto group = [&](const u16x8 (&A)[7], int q) {
        int dx, dz;
        se_k7r_slot(4 * q + g, dx, dz);
        // lane base: halo row (x = 2 w, y = 0) of the tile, column z = v, + the slot's (dx, dz)
        const unsigned char* bp = halo + (((2 * w + dx) * G::HY) * K7_P + v + dz) * 16;
        constexpr int AH = 1;                         // rows of B fragments in flight ahead of the MFMAs (3 and 6 measured the same)
        u16x8 Br[AH + 1];
#pragma unroll
        for (int i = 0; i < AH; ++i) Br[i] = lds_read16(bp + (((i / 10) * G::HY + i % 10) * K7_P) * 16);
#pragma unroll
        for (int rr = 0; rr < 20; ++rr) {
            const int xi = rr / 10, r = rr % 10;
            if (rr + AH < 20) Br[(rr + AH) % (AH + 1)] = lds_read16(bp + ((((rr + AH) / 10) * G::HY + (rr + AH) % 10) * K7_P) * 16);
#pragma unroll
            for (int dy = 0; dy < 7; ++dy) {
                const int y = r - dy;
                if (y >= 0 && y < 4) acc[xi * 4 + y] = mfma_bf16(A[dy], Br[rr % (AH + 1)], acc[xi * 4 + y]);
            }
        }
    };

    u16x8 hreg[G::HP], A0[7], A1[7];
    halo_load(hreg, inb);
    a_load(A0, 0);
    const int phases = octs * SE_K7R_GROUPS;
    for (int c = 0; c < octs; ++c) {
        __syncthreads();                      // previous octet fully consumed
        halo_commit(hreg);
        __syncthreads();
        for (int q = 0; q < SE_K7R_GROUPS; q += 2) {      // two groups per iteration: the fragment sets A0 / A1 alternate
            const int ph = c * SE_K7R_GROUPS + q;
            a_load(A1, ph + 1);
            group(A0, q);
            if (q == SE_K7R_GROUPS - 2 && c + 1 < octs) halo_load(hreg, inb + (long long)(c + 1) * N * 8);   // in flight under the last group
            a_load(A0, ph + 2 < phases ? ph + 2 : ph);
            group(A1, q + 1);
        }
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(a.bpack + 4 * g);
    const bool relu = a.flags & SE_EPI_RELU;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int x = x0 + G::tx(w, n), y = y0 + G::ty(n, v), z = z0 + G::tz(v);
        const long long ovox = (((long long)b * D + x) * D + y) * D + z;
        f32x4 r = acc[n] + bias4;
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        u16x4 o;
        o[0] = f2bf(r.x); o[1] = f2bf(r.y); o[2] = f2bf(r.z); o[3] = f2bf(r.w);
        *reinterpret_cast<u16x4*>(a.out + ovox * 16 + 4 * g) = o;
    }
}

static int launch_k7r(const ConvBArgs& a, int batch, hipStream_t s) {
    using G = K7Geo<true>;
    SE_ENSURE_LDS(conv_bf16_k7r_kernel, K7R_LDS_BYTES);
    const int tx = a.dim / G::TX, ty = a.dim / G::TY, tz = a.dim / G::TZ;
    hipLaunchKernelGGL(conv_bf16_k7r_kernel, dim3((unsigned)(batch * tx * ty * tz)), dim3(256), K7R_LDS_BYTES, s, a, tx, ty, tz);
    SE_CHECK_LAUNCH();
    return 0;
}

template <bool ROW16>
int launch_k7(const ConvBArgs& a, int batch, hipStream_t s) {
    using G = K7Geo<ROW16>;
    SE_ENSURE_LDS(conv_bf16_k7_kernel<ROW16>, G::LDS_BYTES);
    const int tx = a.dim / G::TX, ty = a.dim / G::TY, tz = a.dim / G::TZ;
    hipLaunchKernelGGL(conv_bf16_k7_kernel<ROW16>, dim3((unsigned)(batch * tx * ty * tz)), dim3(256), G::LDS_BYTES, s, a, tx, ty, tz);
    SE_CHECK_LAUNCH();
    return 0;
}

}  // namespace

#ifdef SE_STAMPB
static unsigned long long* g_stampb = nullptr;
#ifdef SE_DEVTOOLS
extern "C" void se_debug_set_stamp_buffer_b(void* p) { g_stampb = reinterpret_cast<unsigned long long*>(p); }
#endif
#endif

template <int TY>
static int launch_k3(const ConvBArgs& a, int batch, hipStream_t s) {
    SE_ENSURE_LDS(conv_bf16_k3_kernel<TY>, K3G<TY>::LDS_BYTES);
    const int tx = a.dim / K3_TX, ty = a.dim / TY, tz = a.dim / K3_TZ;
#ifdef SE_STAMPB
    hipLaunchKernelGGL(conv_bf16_k3_kernel<TY>, dim3((unsigned)(batch * tx * ty * tz), a.cout / 32), dim3(256), K3G<TY>::LDS_BYTES, s,
                       a, tx, ty, tz, g_stampb);
#else
    hipLaunchKernelGGL(conv_bf16_k3_kernel<TY>, dim3((unsigned)(batch * tx * ty * tz), a.cout / 32), dim3(256), K3G<TY>::LDS_BYTES, s,
                       a, tx, ty, tz);
#endif
    SE_CHECK_LAUNCH();
    return 0;
}


static bool epi_has_res_host(const ConvBArgs& a) { return a.res && (a.flags & (SE_EPI_RES_PRE_RELU | SE_EPI_RES_POST_RELU)); }

int se_conv3d_bf16_tiled_try(const ConvBArgs& a, int batch, int ksize, hipStream_t s) {
#ifdef SE_DEVTOOLS
    if (ksize == 3 && a.dim % 16 == 0 && a.cin_pad % 16 == 0 && a.cout % 32 == 0 && a.kpc == K3_KPC && a.total_vox < (1LL << 31) &&
        g_variant == 3) {      // A/B only: the persistent form measured 13-18 % SLOWER than two independent workgroups per CU
        SE_ENSURE_LDS(conv_bf16_k3p_kernel, K3P_LDS_BYTES);
        const int num_cus = se_num_cus();
        const int tx = a.dim / K3_TX, ty = a.dim / K3_TY, tz = a.dim / K3_TZ;
        const long long nitems = (long long)batch * tx * ty * tz * (a.cout / 32);
        if (nitems < (1LL << 31)) {
            const int grid = (int)(nitems < num_cus ? nitems : num_cus);
            hipLaunchKernelGGL(conv_bf16_k3p_kernel, dim3(grid), dim3(512), K3P_LDS_BYTES, s, a, tx, ty, tz, (int)nitems);
            SE_CHECK_LAUNCH();
            return 0;
        }
    }
#endif
    if (ksize == 3 && a.dim % 16 == 0 && a.cin_pad % 16 == 0 && a.cout % 32 == 0 && a.kpc == K3_KPC && a.total_vox < (1LL << 31))
        return g_variant == 4 ? launch_k3<4>(a, batch, s) : launch_k3<8>(a, batch, s);   // 4: tile 4x4x16, 3 workgroups per CU (A/B)
    if (ksize == 7 && a.dim % 8 == 0 && a.cout == 16 && a.kpc == SE_K7B_KPC && !epi_has_res_host(a) &&
        (long long)a.dim * a.dim * a.dim * 8 < (1LL << 31)) {
        if (a.dim % 16 == 0 && g_variant != 1) return g_variant == 5 ? launch_k7<true>(a, batch, s) : launch_k7r(a, batch, s);   // 5: the dz-phase form (A/B)
        return launch_k7<false>(a, batch, s);
    }
    return SE_TILED_NOT_TAKEN_B;
}
