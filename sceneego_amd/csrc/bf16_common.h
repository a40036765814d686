// bf16-storage V2V path (BASELINE config 3): shared types, weight-pack geometry and the fused epilogue.
//
// MFMA: v_mfma_f32_16x16x32_bf16.  A = weights (row r = l & 15 -> cout, K octet g = l >> 4 -> 8 bf16 = 16 B per lane),
// B = activations (column v = l & 15 -> voxel, K octet g), D = 4 float32 per lane: rows 4g..4g+3 of column v.
// A "group" is (tap, 8 consecutive input channels); one MFMA contracts 4 groups.
#pragma once
#include "common.h"

// Workgroup b runs on XCD b % 8 (conv_common.h: se_xcd_walk_index): tile index such that the workgroups of one XCD take consecutive
// tiles - neighbouring halos meet in one L2 (round 5, tools/diag/bf16_ab.py: 64 -> 64 @32^3 -4 %, 32 -> 32 @64^3 +1 %, forward +0.7 %)
__device__ __forceinline__ int se_xcd_tile_bf16(int wg, int n_wg) { return (n_wg & 7) == 0 ? (wg & 7) * (n_wg >> 3) + (wg >> 3) : wg; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    const __bf16 b = (__bf16)f;   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}

// 7^3 front layer: octet-outer chunks; inside a chunk the 343 taps are stored dz-major in 52 slots per dz so that the two
// taps of an MFMA lane-group pair (slots 2i, 2i+1) differ by an even dy (and any dx): with the LDS tile pitches of
// conv_bf16_k7_kernel their addresses then differ by a multiple of 16 voxels = 256 B and a ds_read_b128 is conflict-free.
// slots 0..27: dy in {0,2,4,6} x dx 0..6; 28..48: dy in {1,3,5} x dx; 49..51: padding.
#define SE_K7B_SLOTS_PER_DZ 52
#define SE_K7B_KPC (7 * SE_K7B_SLOTS_PER_DZ / 4)   // 91 k steps per octet chunk
__host__ __device__ inline bool se_k7b_slot(int slot, int& dx, int& dy, int& dz) {
    dz = slot / SE_K7B_SLOTS_PER_DZ;
    const int r = slot - dz * SE_K7B_SLOTS_PER_DZ;
    if (r < 28) { dy = 2 * (r / 7); dx = r % 7; return true; }
    if (r < 49) { dy = 1 + 2 * ((r - 28) / 7); dx = (r - 28) % 7; return true; }
    dx = 6; dy = 5;   // padding slots alias the last real tap's address (zero weights): keeps the read conflict-free
    return false;
}

// 7^3 front layer, row-reuse form (conv_bf16_k7r_kernel, round 5): second section of the packed weights, [octet][q 14][dy 7][lane][8].
// Slot 4 q + g of a k group q names a (dx, dz) pair; the two slots of a lane-group pair (g = 0,1 and g = 2,3) share dz and differ in
// dx, so their LDS addresses differ by a multiple of 256 B (conflict-free, as above).  dz-major: per dz the dx pairs (0,1) (2,3) (4,5)
// (6, pad): 28 pairs = 14 groups = 56 slots for the 49 (dx, dz) combinations.  dy is the axis along which B fragments are reused.
#define SE_K7R_GROUPS 14
__host__ __device__ inline bool se_k7r_slot(int slot, int& dx, int& dz) {
    const int p = slot >> 1;
    dz = p >> 2;
    dx = 2 * (p & 3) + (slot & 1);
    if (dx < 7) return true;
    dx = 6;       // padding slot: zero weights, the partner's address
    return false;
}
__host__ __device__ inline long long se_k7r_elems(int cout, int cin_pad, int ksize, int transposed) {
    return (!transposed && ksize == 7 && cout <= 16) ? (long long)(cin_pad / 8) * SE_K7R_GROUPS * 7 * 512 : 0;
}

struct PackGeomB {
    int taps;     // k^3, or 8 output parities for the k2s2 transposed conv
    int oc;       // octets (8-channel groups) per channel chunk
    int nchunk;   // cin_pad / (8 * oc)
    int kpc;      // k steps per chunk
    int ksteps;   // nchunk * kpc
    int mtiles;   // 16-row cout tiles
};
__host__ __device__ inline PackGeomB pack_geom_b(int cout, int cin_pad, int ksize, int transposed) {
    PackGeomB p;
    const int octs = cin_pad / 8;
    p.taps = transposed ? 8 : ksize * ksize * ksize;
    // 3^3: 16-channel chunks (one k step = 2 taps x 2 octets) keep the LDS tile of conv_bf16_k3_kernel at 62 KB
    p.oc = (!transposed && ksize == 7) ? 1
         : (!transposed && ksize == 3) ? (octs % 2 == 0 ? 2 : 1)
         : (octs % 4 == 0 ? 4 : (octs % 2 == 0 ? 2 : 1));
    p.nchunk = octs / p.oc;
    p.kpc = (!transposed && ksize == 7) ? SE_K7B_KPC : (p.taps * p.oc + 3) / 4;
    p.ksteps = p.nchunk * p.kpc;
    p.mtiles = (cout + 15) / 16;
    return p;
}

struct ConvBArgs {
    const unsigned short* in;
    const unsigned short* wpack;
    const float* bpack;
    const unsigned short* res;
    unsigned short* out;
    long long total_vox;   // B * dim^3 (input voxels)
    int dim;
    int cin_pad;
    int cout;
    int flags;
    int kpc, nchunk, ksteps;
};

// Output channel owned by row r of cout tile m: couts that are multiples of 32 are permuted inside each 32-block so that
// the two D fragments of a tile pair give one lane 8 CONSECUTIVE channels (8g .. 8g+7): a 16-byte store, and exactly the
// B fragment (octet g) of a following 1x1x1 layer.
__host__ __device__ inline int se_bf16_cout_of(int cout, int m, int r) {
    if (cout % 32 == 0) return 32 * (m >> 1) + 8 * (r >> 2) + 4 * (m & 1) + (r & 3);
    return 16 * m + r;
}

// Epilogue of a tile pair: lane (v, g) holds channels cb*32 + 8g .. +7 of one voxel (record index `ovox`).
// Split into load and apply/store halves so that a kernel can issue the bias and ALL residual loads of its tiles before the
// first use (one exposed latency per workgroup instead of two per tile).
struct EpiBias8 { f32x4 b0, b1; };
__device__ __forceinline__ EpiBias8 epi_load_bias8(const ConvBArgs& a, int cb, int g) {
    EpiBias8 e;
    e.b0 = *reinterpret_cast<const f32x4*>(a.bpack + cb * 32 + 8 * g);
    e.b1 = *reinterpret_cast<const f32x4*>(a.bpack + cb * 32 + 8 * g + 4);
    return e;
}
__device__ __forceinline__ bool epi_has_res(const ConvBArgs& a) {
    return a.res && (a.flags & (SE_EPI_RES_PRE_RELU | SE_EPI_RES_POST_RELU));
}
__device__ __forceinline__ void epi_store_pair(const ConvBArgs& a, f32x4 lo, f32x4 hi, const EpiBias8& e, bool has_res, u16x8 rv,
                                               long long off) {
    float v[8] = {lo.x + e.b0.x, lo.y + e.b0.y, lo.z + e.b0.z, lo.w + e.b0.w,
                  hi.x + e.b1.x, hi.y + e.b1.y, hi.z + e.b1.z, hi.w + e.b1.w};
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = has_res ? bf2f(rv[i]) : 0.f;
    if (a.flags & SE_EPI_RES_PRE_RELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += r[i];
    }
    if (a.flags & SE_EPI_RELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (a.flags & SE_EPI_RES_POST_RELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += r[i];
    }
    u16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f2bf(v[i]);
    __builtin_nontemporal_store(o, reinterpret_cast<u16x8*>(a.out + off));      // written once, read by the NEXT launch after 0.5 GB of other traffic
}
__device__ __forceinline__ void epilogue_pair_bf16(const ConvBArgs& a, f32x4 lo, f32x4 hi, long long ovox, int cb, int g) {
    const long long off = ovox * a.cout + cb * 32 + 8 * g;
    const bool has_res = epi_has_res(a);
    u16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (has_res) rv = *reinterpret_cast<const u16x8*>(a.res + off);
    epi_store_pair(a, lo, hi, epi_load_bias8(a, cb, g), has_res, rv, off);
}

// Single 16-cout tile (cout <= 16, e.g. the front layer): lane (v, g) holds channels 4g .. 4g+3.
__device__ __forceinline__ void epilogue_single_bf16(const ConvBArgs& a, f32x4 acc, long long ovox, int g) {
    const int co0 = 4 * g;
    if (co0 >= a.cout) return;
    float v[4] = {acc.x, acc.y, acc.z, acc.w};
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bpack + co0);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
    const long long off = ovox * a.cout + co0;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    const bool has_res = a.res && (a.flags & (SE_EPI_RES_PRE_RELU | SE_EPI_RES_POST_RELU));
    if (has_res) {
        const u16x4 rv = *reinterpret_cast<const u16x4*>(a.res + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = bf2f(rv[i]);
    }
    if (has_res && (a.flags & SE_EPI_RES_PRE_RELU)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += r[i];
    }
    if (a.flags & SE_EPI_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (has_res && (a.flags & SE_EPI_RES_POST_RELU)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += r[i];
    }
    u16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f2bf(v[i]);
    *reinterpret_cast<u16x4*>(a.out + off) = o;
}

__device__ __forceinline__ f32x4 mfma_bf16(u16x8 a, u16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
