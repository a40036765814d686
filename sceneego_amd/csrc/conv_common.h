// Shared between conv3d.hip (direct kernels, packer, C ABI) and conv3d_tiled.hip (LDS-tiled kernels).
#pragma once
#include "common.h"

// 7x7x7 tiled kernel: the 343 taps are taken 4 at a time on the MFMA k lanes -> 86 groups (last one has 1 pad tap)
#define SE_K7_TAPS 343
#define SE_K7_GROUPS 86
#define SE_TILED_NOT_TAKEN (-1000)
// 1-D Winograd section of the packed 3x3x3 weights: per (16-cin group, 32-cout block): 9 (dy,dx) x 4 xi x 2 cout tiles x 1 KiB
#define SE_WINO_CHUNK_FLOATS (9 * 4 * 2 * 256)
// F(4,3) variant (section E): 9 (dy,dx) x 6 xi x 2 cout tiles x 1 KiB = 108 KB per (16-cin group, 32-cout block)
#define SE_WINO43_CHUNK_FLOATS (9 * 6 * 2 * 256)
// 2-D Winograd F(4,3) x F(2,3) section (G) of the packed 3x3x3 weights (conv3d_wino2d.hip): per (32-cout block, 8-cin chunk)
// 24 xi x 3 dx x 2 cout tiles x 64 lanes x 2 = 73,728 B
#define SE_WINO2D_CHUNK_FLOATS (24 * 3 * 2 * 128)
// 2-D Winograd F(4,3) x F(4,3) section (I) of the packed 3x3x3 weights (conv3d_wino44.hip): per (32-cout block, 4-cin chunk)
// 9 xi quads x 3 dx x 2 cout tiles x 64 lanes x 4 = 55,296 B
#define SE_WINO44_CHUNK_FLOATS (9 * 3 * 2 * 256)
// 1-D Winograd F(2,7) section of the packed 7x7x7 weights: per 4-channel chunk 13 (dy,dx) tap groups x 8 xi x 1 KiB
#define SE_K7W_GROUPS 13
#define SE_K7W_CHUNK_FLOATS (SE_K7W_GROUPS * 8 * 256)
// 1-D Winograd F(4,7) section (F) of the packed 7x7x7 weights: per 3-channel chunk 13 (dy,dx) tap groups x 10 xi x 64 lanes x 3
#define SE_K7F_XI 10
#define SE_K7F_CHUNK_FLOATS (SE_K7W_GROUPS * SE_K7F_XI * 64 * 3)
// 1-D Winograd F(6,7) section (H) of the packed 7x7x7 weights (conv3d_wino67.hip): per 3-channel chunk the 147 (channel, dy, dx)
// taps 4 at a time on the k lanes = 37 groups x 64 lanes x 12 xi
#define SE_K7H_GROUPS 37
#define SE_K7H_CHUNK_FLOATS (SE_K7H_GROUPS * 64 * 12)

// G matrix of F(2,7) with interpolation points {0, 1, -1, 2, -2, 1/2, -1/2, inf} (Cook-Toom; tools/wino27_matrices.py):
// row xi, column kz.  y = A^T [(G g) .* (B^T d)].
__host__ __device__ inline float se_wino27_G(int xi, int kz) {
    switch (xi) {
        case 0: return kz == 0 ? -1.f : 0.f;
        case 1: return -2.f / 9.f;
        case 2: return (kz & 1) ? 2.f / 9.f : -2.f / 9.f;
        case 3: return (float)(1 << kz) / 90.f;
        case 4: return ((kz & 1) ? -1.f : 1.f) * (float)(1 << kz) / 90.f;
        case 5: return (float)(64 >> kz) / 90.f;
        case 6: return ((kz & 1) ? -1.f : 1.f) * (float)(64 >> kz) / 90.f;
        default: return kz == 6 ? 1.f : 0.f;
    }
}

// XCD-aware walk of the persistent kernels (round 5).  Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8,
// each XCD has its own L2), so with unit ranges in blockIdx order the NEIGHBOURING ranges - which share their halo rows - sit in eight
// different L2s and every shared row is fetched from the fabric once per L2.  Virtual index: the workgroups of one XCD walk
// consecutive ranges, the halo between them is an L2 hit.  (Round 4 measured the same map at -2 ... +1 % of time - the kernels are not
// bound by where their rows come from; what it buys is fabric traffic: SE_XCD_WALK=0 in an A/B build restores blockIdx order.)
#ifndef SE_XCD_WALK
#define SE_XCD_WALK 1
#endif
__device__ __forceinline__ int se_xcd_walk_index(int wg, int n_wg) {
    return (SE_XCD_WALK && (n_wg & 7) == 0) ? (wg & 7) * (n_wg >> 3) + (wg >> 3) : wg;
}
// The interleaved form of the same idea (workgroup w of an XCD's S takes units w, w + S, ...: the S tiles in flight on an XCD are S
// consecutive units) lives in the kernels that have it: SE_K44P_XCD_WALK (on), SE_K67_XCD_WALK (off).

__host__ __device__ inline int round_up16(int v) { return (v + 15) & ~15; }

// Shapes the 2-D Winograd 3x3x3 kernel (conv3d_wino2d.hip) covers - ONE predicate for se_conv3d_f32_algo() (what callers use to
// choose the octet-planar / pooled / fused-skip forms) and for the launcher.  `cin` is the channel stride of the input tensor (the
// kernel needs cin_pad == cin).  The last term: 32-bit byte offsets inside one sample (bit 31 marks out-of-volume lanes).
inline bool se_wino2d_shape_ok(int dim, int cin, int cout) {
    if (dim < 16 || (dim & 15) || cout <= 0 || (cout & 31) || cin <= 0 || (cin & 7)) return false;
    return (long long)dim * dim * dim * (cin > cout ? cin : cout) * 4 < (1LL << 31);
}

// A 2-D Winograd shape with so few voxels that its 4 x 8 x 16 tiles leave most of the chip idle (16^3 at batch 1: 8 tiles x 4 cout
// blocks on 256 CUs, 92 us per 128 -> 128 launch): a call WITHOUT octet-planar / pooled / fused-skip forms then runs on the
// in-workgroup split-K kernel of the 8^3 level (32-voxel tiles, 512 workgroups, 47 us).  se_conv3d_f32_variant() reports it, the
// V2V program asks per (batch, level).
inline bool se_conv3d_small_volume(int batch, int dim) { return (long long)batch * dim * dim * dim <= 4096; }

struct ConvArgs {
    const float* in;
    const float* wpack;    // section A: [cg][tap][nt][lane][4]
    const float* wpack_b;  // k = 7: section B [chunk4][group][nt][lane][4];  k = 3: Winograd section C (NULL if cout % 32)
    const float* wpack_e;  // k = 3, cout % 32 == 0: Winograd F(4,3) section E (else NULL)
    const float* wpack_d;  // k = 7, cout <= 16: Winograd F(2,7) section D [chunk4][g13][xi8][lane][4] (else NULL)
    const float* wpack_f;  // k = 7, cout <= 16: Winograd F(4,7) section F [chunk3][g13][xi10][lane][3] (else NULL)
    const float* wpack_h;  // k = 7, cout <= 16: Winograd F(6,7) section H [chunk3][g37][lane][xi12] (else NULL)
    const float* wpack_i;  // k = 3, cout % 32 == 0: 2-D Winograd F(4,3) x F(4,3) section I (else NULL)
    const float* wpack_g;  // k = 3, cout % 32 == 0: 2-D Winograd F(4,3) x F(2,3) section G (else NULL)
    const float* skip_w;   // SE_EPI_SKIPCONV16: folded 1x1x1 skip weights [cout][16]; `res` then is the skip convolution's 16-channel input
    float* pool_out;       // 2-D Winograd kernel only: also write max_pool3d(out, 2, 2), channels-last [B][D/2][D/2][D/2][cout] (else NULL)
    const float* bpack;
    const float* res;
    float* out;
    long long total_vox;  // B * dim^3 (input voxels)
    int dim;
    int cin;       // real input channels (chunks beyond it are skipped: their weights are zero)
    int cin_pad;   // channel stride of the input tensor
    int cout;      // real output channels
    int nts;       // cout tiles of 16 in the packed weights
    int flags;
    float* ws = nullptr;          // the caller's workspace (se_conv3d_f32): split-K partial sums of the small levels, second-half sums of a split 7^3 launch
    long long ws_elems = 0;
};

// Epilogue for one accumulator fragment: this lane owns output channels co0..co0+3 of one voxel
// (`on` = voxel index inside sample `b`, `ovox_per_b` voxels per sample).  bias (+BN shift) -> [+res] -> [ReLU]
// -> [+res] -> 16-byte channels-last store, or 4 planar stores for SE_EPI_OUT_PLANAR.
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 v, int b, long long on, long long ovox_per_b,
                                              int co0) {
    if (co0 >= a.cout) return;
    v += *reinterpret_cast<const f32x4*>(a.bpack + co0);
    const bool relu = a.flags & SE_EPI_RELU;
    if (a.flags & SE_EPI_OUT_PLANAR) {
        float* o = a.out + ((long long)b * a.cout + co0) * ovox_per_b + on;
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (co0 + r < a.cout) o[(long long)r * ovox_per_b] = relu ? fmaxf(vv[r], 0.f) : vv[r];
        }
        return;
    }
    const long long ooff = ((long long)b * ovox_per_b + on) * a.cout + co0;
    if ((a.flags & SE_EPI_RES_PRE_RELU) && a.res) v += *reinterpret_cast<const f32x4*>(a.res + ooff);
    if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if ((a.flags & SE_EPI_RES_POST_RELU) && a.res) v += *reinterpret_cast<const f32x4*>(a.res + ooff);
    *reinterpret_cast<f32x4*>(a.out + ooff) = v;
}
