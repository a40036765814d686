// LDS-tiled bf16 convolutions for the large V2V levels (see conv3d_bf16.hip for the packer and the direct form).
#include "bf16_common.h"

#define SE_TILED_NOT_TAKEN_B (-1000)

namespace {

__device__ __forceinline__ u16x8 lds_read16(const unsigned char* p) { return *reinterpret_cast<const u16x8*>(p); }

// ------------------------------------------------------------------------------------------------
// 3x3x3, dim % 16 == 0, cin % 16 == 0, cout % 32 == 0.
// Workgroup = 4 waves, output tile 4(x) x 8(y) x 16(z) voxels x 32 couts; wave w owns x = w, its 8 voxel tiles are the
// y rows (16 z each), so a lane's B address is   halo[(w+dx)][(n+dy)][(v+dz)]   and advances by one row per tile.
// Per 16-channel chunk the LDS holds the 6x10x18 halo (32 B / voxel) and the 14 k steps x 2 cout tiles of weights.
// A k step = 2 taps x 2 octets: lane group g reads tap 2s + (g >> 1), octet g & 1.  The two lane groups that share a
// ds_read_b128 cycle (g = 0,1 and g = 2,3) read the same tap, so the 16 lanes cover 16 distinct 16-byte slots.
// ------------------------------------------------------------------------------------------------
constexpr int K3_TX = 4, K3_TY = 8, K3_TZ = 16;
constexpr int K3_HX = K3_TX + 2, K3_HY = K3_TY + 2, K3_HZ = K3_TZ + 2;
constexpr int K3_HALO_VOX = K3_HX * K3_HY * K3_HZ;            // 1080
constexpr int K3_HALO_BYTES = K3_HALO_VOX * 32;               // 34560
constexpr int K3_KPC = 14;
constexpr int K3_W_BYTES = K3_KPC * 2 * 1024;                 // 28672
constexpr int K3_LDS_BYTES = K3_HALO_BYTES + K3_W_BYTES;      // 63232 -> two workgroups per CU

__global__ __launch_bounds__(256, 2) void conv_bf16_k3_kernel(ConvBArgs a, int tiles_x, int tiles_y, int tiles_z) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* halo = lds;
    unsigned char* wts = lds + K3_HALO_BYTES;
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
    const int x0 = tx * K3_TX, y0 = ty * K3_TY, z0 = tz * K3_TZ;

    f32x4 acc[2][K3_TY];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < K3_TY; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned char* brow = halo + ((w * K3_HY) * K3_HZ + v) * 32 + (g & 1) * 16;
    const unsigned short* wsrc = a.wpack + ((size_t)mb * 2 * a.ksteps) * 512;

    for (int c = 0; c < a.nchunk; ++c) {
        __syncthreads();
        // halo: 2160 16-byte pieces (voxel, octet); zero outside the volume
        for (int i = tid; i < K3_HALO_VOX * 2; i += 256) {
            const int hv = i >> 1, o = i & 1;
            const int hz = hv % K3_HZ, hy = (hv / K3_HZ) % K3_HY, hx = hv / (K3_HZ * K3_HY);
            const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
            u16x8 val = {0, 0, 0, 0, 0, 0, 0, 0};
            if ((unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && (unsigned)gz < (unsigned)D)
                val = *reinterpret_cast<const u16x8*>(a.in + ((((long long)b * D + gx) * D + gy) * D + gz) * a.cin_pad + c * 16 + o * 8);
            *reinterpret_cast<u16x8*>(halo + i * 16) = val;
        }
        // weights of this chunk: [m][14][lane][16 B]
        for (int i = tid; i < K3_W_BYTES / 16; i += 256) {
            const int m = i / (K3_KPC * 64), r = i - m * (K3_KPC * 64);
            *reinterpret_cast<u16x8*>(wts + i * 16) =
                *reinterpret_cast<const u16x8*>(wsrc + ((size_t)m * a.ksteps + c * K3_KPC) * 512 + r * 8);
        }
        __syncthreads();
#pragma unroll 2
        for (int sl = 0; sl < K3_KPC; ++sl) {
            int tap = 2 * sl + (g >> 1);
            tap = tap > 26 ? 26 : tap;                 // padding group: zero weights, any valid address
            const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
            const unsigned char* bp = brow + ((dx * K3_HY + dy) * K3_HZ + dz) * 32;
            const u16x8 A0 = lds_read16(wts + sl * 1024 + lane * 16);
            const u16x8 A1 = lds_read16(wts + (K3_KPC + sl) * 1024 + lane * 16);
#pragma unroll
            for (int n = 0; n < K3_TY; ++n) {
                const u16x8 Bf = lds_read16(bp + n * (K3_HZ * 32));
                acc[0][n] = mfma_bf16(A0, Bf, acc[0][n]);
                acc[1][n] = mfma_bf16(A1, Bf, acc[1][n]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < K3_TY; ++n) {
        const long long ovox = (((long long)b * D + (x0 + w)) * D + (y0 + n)) * D + (z0 + v);
        epilogue_pair_bf16(a, acc[0][n], acc[1][n], ovox, mb, g);
    }
}

// ------------------------------------------------------------------------------------------------
// 7x7x7 front layer, cout = 16, dim % 8 == 0, octet-planar input [B][octs][D^3][8].
// Workgroup = 4 waves, output tile 8x8x8; wave w owns x in {2w, 2w+1}; its 8 voxel tiles are (x, y pair): a tile's 16
// columns are 2 y rows x 8 z.  Octet-outer: the LDS holds the 14x14x14 halo of ONE 8-channel octet (16 B / voxel, z pitch
// 24 voxels) and, double-buffered, the 13 k steps (52 tap slots) of one dz plane of weights; the accumulators stay in
// registers over all octets.  Tap slots are paired so that lane groups sharing a ds_read_b128 pass read addresses that
// differ by a multiple of 256 B (bf16_common.h, tools/lds_conflicts_bf16.py: conflict-free).
// ------------------------------------------------------------------------------------------------
constexpr int K7_T = 8, K7_H = 14, K7_P = 24;
constexpr int K7_HALO_BYTES = K7_H * K7_H * K7_P * 16;        // 75264
constexpr int K7_KPD = SE_K7B_SLOTS_PER_DZ / 4;               // 13 k steps per dz plane
constexpr int K7_WBUF_BYTES = K7_KPD * 1024;                  // 13312
constexpr int K7_LDS_BYTES = K7_HALO_BYTES + 2 * K7_WBUF_BYTES;   // 101888
constexpr int K7_WPIECES = K7_WBUF_BYTES / 16;                // 832

__global__ __launch_bounds__(256) void conv_bf16_k7_kernel(ConvBArgs a, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* halo = lds;
    unsigned char* wbuf = lds + K7_HALO_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int v = lane & 15, g = lane >> 4;
    const int D = a.dim;
    const int octs = a.nchunk;
    int t = blockIdx.x;
    const int tz = t % tiles; t /= tiles;
    const int ty = t % tiles; t /= tiles;
    const int tx = t % tiles;
    const int b = t / tiles;
    const int x0 = tx * K7_T, y0 = ty * K7_T, z0 = tz * K7_T;
    const long long N = (long long)D * D * D;
    const unsigned short* inb = a.in + (long long)b * octs * N * 8;

    f32x4 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned char* lbase = halo + (((2 * w) * K7_H + (v >> 3)) * K7_P + (v & 7)) * 16;

    // weights of phase 0
    for (int i = tid; i < K7_WPIECES; i += 256)
        *reinterpret_cast<u16x8*>(wbuf + i * 16) = *reinterpret_cast<const u16x8*>(a.wpack + (size_t)i * 8);
    int cur = 0;
    const int phases = octs * 7;
    for (int c = 0; c < octs; ++c) {
        __syncthreads();
        const unsigned short* inc = inb + (long long)c * N * 8;
        for (int i = tid; i < K7_H * K7_H * K7_H; i += 256) {
            const int hz = i % K7_H, hy = (i / K7_H) % K7_H, hx = i / (K7_H * K7_H);
            const int gx = x0 + hx - 3, gy = y0 + hy - 3, gz = z0 + hz - 3;
            u16x8 val = {0, 0, 0, 0, 0, 0, 0, 0};
            if ((unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && (unsigned)gz < (unsigned)D)
                val = *reinterpret_cast<const u16x8*>(inc + (((long long)gx * D + gy) * D + gz) * 8);
            *reinterpret_cast<u16x8*>(halo + ((hx * K7_H + hy) * K7_P + hz) * 16) = val;
        }
        __syncthreads();
        for (int dz = 0; dz < 7; ++dz) {
            const int ph = c * 7 + dz + 1;
            const bool has_next = ph < phases;
            u16x8 pre[4];
            if (has_next) {
                const unsigned short* src = a.wpack + (size_t)ph * K7_KPD * 512;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + 256 * j;
                    if (i < K7_WPIECES) pre[j] = *reinterpret_cast<const u16x8*>(src + (size_t)i * 8);
                }
            }
            const unsigned char* wb = wbuf + cur * K7_WBUF_BYTES + lane * 16;
            const unsigned char* lz = lbase + dz * 16;
#pragma unroll 1
            for (int sl = 0; sl < K7_KPD; ++sl) {
                int r = 4 * sl + g;
                r = r > 48 ? 48 : r;                       // padding slots alias the last tap (zero weights)
                const int r2 = r < 28 ? r : r - 28;
                const int q7 = r2 / 7;
                const int dx = r2 - 7 * q7, dy = 2 * q7 + (r < 28 ? 0 : 1);
                const unsigned char* bp = lz + ((dx * K7_H + dy) * K7_P) * 16;
                const u16x8 A = lds_read16(wb + sl * 1024);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const u16x8 Bf = lds_read16(bp + (((n >> 2) * K7_H + 2 * (n & 3)) * K7_P) * 16);
                    acc[n] = mfma_bf16(A, Bf, acc[n]);
                }
            }
            if (has_next) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + 256 * j;
                    if (i < K7_WPIECES) *reinterpret_cast<u16x8*>(wbuf + (cur ^ 1) * K7_WBUF_BYTES + i * 16) = pre[j];
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int x = x0 + 2 * w + (n >> 2), y = y0 + 2 * (n & 3) + (v >> 3), z = z0 + (v & 7);
        const long long ovox = (((long long)b * D + x) * D + y) * D + z;
        epilogue_single_bf16(a, acc[n], ovox, g);
    }
}

}  // namespace

int se_conv3d_bf16_tiled_try(const ConvBArgs& a, int batch, int ksize, hipStream_t s) {
    if (ksize == 3 && a.dim % 16 == 0 && a.cin_pad % 16 == 0 && a.cout % 32 == 0 && a.kpc == K3_KPC) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_k3_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, K3_LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        const int tx = a.dim / K3_TX, ty = a.dim / K3_TY, tz = a.dim / K3_TZ;
        hipLaunchKernelGGL(conv_bf16_k3_kernel, dim3((unsigned)(batch * tx * ty * tz), a.cout / 32), dim3(256), K3_LDS_BYTES, s,
                           a, tx, ty, tz);
        SE_CHECK_LAUNCH();
        return 0;
    }
    if (ksize == 7 && a.dim % 8 == 0 && a.cout == 16 && a.kpc == SE_K7B_KPC) {
        static bool attr_set7 = false;
        if (!attr_set7) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_k7_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, K7_LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            attr_set7 = true;
        }
        const int tiles = a.dim / K7_T;
        hipLaunchKernelGGL(conv_bf16_k7_kernel, dim3((unsigned)(batch * tiles * tiles * tiles)), dim3(256), K7_LDS_BYTES, s, a, tiles);
        SE_CHECK_LAUNCH();
        return 0;
    }
    return SE_TILED_NOT_TAKEN_B;
}
