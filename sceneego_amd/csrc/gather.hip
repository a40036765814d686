// Table-driven bilinear voxel gather (HBM-bound on the 128 B/voxel output stream).
//
// The reference materialises a [B,32,1024,1280] tensor (nearest upsample + zero pad) only to point-sample
// it at the G^3 projected voxel centres with grid_sample (voxel_net_depth.py:60-61,238,243; op.py:209).
// Here the 4 bilinear taps of every voxel are resolved at init time to texel indices of the compact
// 64x64 map (sceneego_amd/op.py:build_gather_table), so the per-frame work is 4 x 128 B reads (L2
// resident: the whole map is 512 KB) and one 128 B write per voxel.
#include "common.h"

namespace {

// thread = (voxel, channel quad); channel quads of one voxel are adjacent lanes => 128 B contiguous
// stores for C = 32, and the 4 tap rows are read as 16 B per lane, 128 B per voxel.
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ feat, const int4* __restrict__ idx,
                                                     const f32x4* __restrict__ w, float* __restrict__ out,
                                                     int texels, int cq /* channels/4 */, int voxels,
                                                     int out_stride_c, int out_c_offset) {
    const int b = blockIdx.y;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int v = (int)(t / cq);
    const int q = (int)(t - (long long)v * cq);
    if (v >= voxels) return;
    const int4 id = idx[v];
    const f32x4 wt = w[v];
    const float* fb = feat + (size_t)b * texels * (cq * 4) + q * 4;
    const int C = cq * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // accumulation order nw, ne, sw, se as ATen's grid_sampler_2d (zeros padding => skipped taps add 0)
    if (id.x >= 0) acc += *reinterpret_cast<const f32x4*>(fb + (size_t)id.x * C) * wt.x;
    if (id.y >= 0) acc += *reinterpret_cast<const f32x4*>(fb + (size_t)id.y * C) * wt.y;
    if (id.z >= 0) acc += *reinterpret_cast<const f32x4*>(fb + (size_t)id.z * C) * wt.z;
    if (id.w >= 0) acc += *reinterpret_cast<const f32x4*>(fb + (size_t)id.w * C) * wt.w;
    float* o = out + ((size_t)b * voxels + v) * out_stride_c + out_c_offset + q * 4;
    *reinterpret_cast<f32x4*>(o) = acc;
}

// Triplet-planar form (SE_IN_PLANAR3 input of the 7^3 layer): one thread per voxel gathers all CQ*4 channels (same tap order and
// arithmetic as gather_kernel) and stores them as [triplet][voxel][3]: a wave writes 64 x 12 contiguous bytes per triplet.
// Slots beyond the last channel in the last triplet are written as zero (the occupancy slot: se_voxelize_planar3_f64).
struct f32x3s { float x, y, z; };
template <int CQ>
__global__ __launch_bounds__(256) void gather_planar3_kernel(const float* __restrict__ feat, const int4* __restrict__ idx,
                                                             const f32x4* __restrict__ w, float* __restrict__ out,
                                                             int texels, int voxels, int triplets_total) {
    constexpr int C = CQ * 4;
    constexpr int T = (C + 2) / 3;
    const int b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= voxels) return;
    const int4 id = idx[v];
    const f32x4 wt = w[v];
    const float* fb = feat + (size_t)b * texels * C;
    f32x4 acc[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (id.x >= 0) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) acc[q] += *reinterpret_cast<const f32x4*>(fb + (size_t)id.x * C + q * 4) * wt.x;
    }
    if (id.y >= 0) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) acc[q] += *reinterpret_cast<const f32x4*>(fb + (size_t)id.y * C + q * 4) * wt.y;
    }
    if (id.z >= 0) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) acc[q] += *reinterpret_cast<const f32x4*>(fb + (size_t)id.z * C + q * 4) * wt.z;
    }
    if (id.w >= 0) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) acc[q] += *reinterpret_cast<const f32x4*>(fb + (size_t)id.w * C + q * 4) * wt.w;
    }
    float f[T * 3];
#pragma unroll
    for (int q = 0; q < CQ; ++q) { f[4 * q] = acc[q].x; f[4 * q + 1] = acc[q].y; f[4 * q + 2] = acc[q].z; f[4 * q + 3] = acc[q].w; }
#pragma unroll
    for (int c = C; c < T * 3; ++c) f[c] = 0.f;
    float* o = out + ((size_t)b * triplets_total * voxels + v) * 3;
#pragma unroll
    for (int t = 0; t < T; ++t)
        *reinterpret_cast<f32x3s*>(o + (size_t)t * voxels * 3) = (f32x3s){f[3 * t], f[3 * t + 1], f[3 * t + 2]};
}

// Fully planar form [B][planes_total][voxels] (round 6: the input of the frequency-domain 7^3 layer, conv3d_fft7.hip, whose tile
// transform walks one channel at a time): same taps, order and arithmetic as gather_kernel; a wave writes 256 contiguous bytes per
// channel plane.  Planes [C, planes_total) are written as zero (the occupancy plane: se_voxelize_planar1_f64 only scatters).
template <int CQ>
__global__ __launch_bounds__(256) void gather_planar1_kernel(const float* __restrict__ feat, const int4* __restrict__ idx,
                                                             const f32x4* __restrict__ w, float* __restrict__ out,
                                                             int texels, int voxels, int planes_total) {
    constexpr int C = CQ * 4;
    const int b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= voxels) return;
    const int4 id = idx[v];
    const f32x4 wt = w[v];
    const float* fb = feat + (size_t)b * texels * C;
    f32x4 acc[CQ];
#pragma unroll
    for (int q = 0; q < CQ; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (id.x >= 0) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) acc[q] += *reinterpret_cast<const f32x4*>(fb + (size_t)id.x * C + q * 4) * wt.x;
    }
    if (id.y >= 0) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) acc[q] += *reinterpret_cast<const f32x4*>(fb + (size_t)id.y * C + q * 4) * wt.y;
    }
    if (id.z >= 0) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) acc[q] += *reinterpret_cast<const f32x4*>(fb + (size_t)id.z * C + q * 4) * wt.z;
    }
    if (id.w >= 0) {
#pragma unroll
        for (int q = 0; q < CQ; ++q) acc[q] += *reinterpret_cast<const f32x4*>(fb + (size_t)id.w * C + q * 4) * wt.w;
    }
    float* o = out + (size_t)b * planes_total * voxels + v;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
        o[(size_t)(4 * q) * voxels] = acc[q].x;
        o[(size_t)(4 * q + 1) * voxels] = acc[q].y;
        o[(size_t)(4 * q + 2) * voxels] = acc[q].z;
        o[(size_t)(4 * q + 3) * voxels] = acc[q].w;
    }
    for (int c = C; c < planes_total; ++c) o[(size_t)c * voxels] = 0.f;
}

__global__ __launch_bounds__(256) void intersection_kernel(float* __restrict__ buf, const float* __restrict__ occ,
                                                           long long total_vox, int cq, int stride_c) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long v = t / cq;
    const int q = (int)(t - v * cq);
    if (v >= total_vox) return;
    const float o = occ[v];
    float* p = buf + v * stride_c + q * 4;
    const f32x4 x = *reinterpret_cast<const f32x4*>(p);
    *reinterpret_cast<f32x4*>(p + cq * 4) = x * o;
}

}  // namespace

extern "C" int se_unproject_gather_f32(const float* feat, const int* idx, const float* w, float* out, int batch,
                                       int texels, int channels, int voxels, int out_stride_c, int out_c_offset,
                                       void* stream) {
    if (batch <= 0 || texels <= 0 || voxels <= 0 || channels <= 0) return SE_ERR_BAD_ARG;
    if ((channels & 3) || (out_stride_c & 3) || (out_c_offset & 3) || out_c_offset + channels > out_stride_c)
        return SE_ERR_BAD_ARG;
    const int cq = channels / 4;
    const long long threads = (long long)voxels * cq;
    dim3 grid((unsigned)((threads + 255) / 256), batch);
    hipLaunchKernelGGL(gather_kernel, grid, dim3(256), 0, se_stream(stream), feat,
                       reinterpret_cast<const int4*>(idx), reinterpret_cast<const f32x4*>(w), out, texels, cq,
                       voxels, out_stride_c, out_c_offset);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_unproject_gather_planar3_f32(const float* feat, const int* idx, const float* w, float* out, int batch,
                                               int texels, int channels, int voxels, int triplets_total, void* stream) {
    if (batch <= 0 || texels <= 0 || voxels <= 0 || channels <= 0 || triplets_total * 3 < channels) return SE_ERR_BAD_ARG;
    dim3 grid((unsigned)((voxels + 255) / 256), batch);
    hipStream_t s = se_stream(stream);
    const int4* ip = reinterpret_cast<const int4*>(idx);
    const f32x4* wp = reinterpret_cast<const f32x4*>(w);
    switch (channels) {
        case 16: hipLaunchKernelGGL(gather_planar3_kernel<4>, grid, dim3(256), 0, s, feat, ip, wp, out, texels, voxels, triplets_total); break;
        case 32: hipLaunchKernelGGL(gather_planar3_kernel<8>, grid, dim3(256), 0, s, feat, ip, wp, out, texels, voxels, triplets_total); break;
        case 64: hipLaunchKernelGGL(gather_planar3_kernel<16>, grid, dim3(256), 0, s, feat, ip, wp, out, texels, voxels, triplets_total); break;
        default: return SE_ERR_BAD_ARG;
    }
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_unproject_gather_planar1_f32(const float* feat, const int* idx, const float* w, float* out, int batch,
                                               int texels, int channels, int voxels, int planes_total, void* stream) {
    if (batch <= 0 || texels <= 0 || voxels <= 0 || channels <= 0 || planes_total < channels) return SE_ERR_BAD_ARG;
    dim3 grid((unsigned)((voxels + 255) / 256), batch);
    hipStream_t s = se_stream(stream);
    const int4* ip = reinterpret_cast<const int4*>(idx);
    const f32x4* wp = reinterpret_cast<const f32x4*>(w);
    switch (channels) {
        case 16: hipLaunchKernelGGL(gather_planar1_kernel<4>, grid, dim3(256), 0, s, feat, ip, wp, out, texels, voxels, planes_total); break;
        case 32: hipLaunchKernelGGL(gather_planar1_kernel<8>, grid, dim3(256), 0, s, feat, ip, wp, out, texels, voxels, planes_total); break;
        case 64: hipLaunchKernelGGL(gather_planar1_kernel<16>, grid, dim3(256), 0, s, feat, ip, wp, out, texels, voxels, planes_total); break;
        default: return SE_ERR_BAD_ARG;
    }
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_intersection_f32(float* buf, const float* occ, int batch, int voxels, int channels, int stride_c,
                                   void* stream) {
    if (batch <= 0 || voxels <= 0 || channels <= 0 || (channels & 3) || (stride_c & 3) || 2 * channels > stride_c)
        return SE_ERR_BAD_ARG;
    const int cq = channels / 4;
    const long long total_vox = (long long)batch * voxels;
    const long long threads = total_vox * cq;
    hipLaunchKernelGGL(intersection_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, se_stream(stream),
                       buf, occ, total_vox, cq, stride_c);
    SE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// NCHW bias (+ residual) (+ ReLU) epilogue for the MIOpen 2D convolutions of the backbone: one pass instead of the
// separate bias-add / add / clamp kernels PyTorch issues after F.conv2d (network/pose_resnet.py:52-90 per block).
// x, res, out: [N][C][HW] float32 (out may alias x); bias [C].  HW % 4 == 0.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void bias_act_kernel(const f32x4* __restrict__ x, const float* __restrict__ bias,
                                                       const f32x4* __restrict__ res, f32x4* __restrict__ out,
                                                       long long total4, int hw4, int channels, int relu) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c = (int)((i / hw4) % channels);
        f32x4 v = x[i] + bias[c];
        if (res) v += res[i];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        out[i] = v;
    }
}
}  // namespace

extern "C" int se_bias_act_nchw_f32(const float* x, const float* bias, const float* residual, float* out, int batch,
                                    int channels, int hw, int relu, void* stream) {
    if (batch <= 0 || channels <= 0 || hw <= 0 || (hw & 3)) return SE_ERR_BAD_ARG;
    const long long total4 = (long long)batch * channels * (hw / 4);
    const unsigned grid = (unsigned)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(bias_act_kernel, dim3(grid), dim3(256), 0, se_stream(stream), reinterpret_cast<const f32x4*>(x), bias,
                       reinterpret_cast<const f32x4*>(residual), reinterpret_cast<f32x4*>(out), total4, hw / 4, channels, relu);
    SE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Stem tail of the backbone in one pass: `bn1` (folded: + bias) + `relu` + `maxpool` (3 x 3, stride 2, padding 1) of
// network/pose_resnet.py:229-232 on the raw result of conv1.  relu(. + b) is monotone, so max first: out = relu(max_window(x) + bias[c]).
// x [N][C][2 ho][2 wo], out [N][C][ho][wo]; wo % 4 == 0 (a thread makes 4 consecutive outputs from 3 rows x 9 columns).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void bias_relu_maxpool_kernel(const float* __restrict__ x, const float* __restrict__ bias, f32x4* __restrict__ out,
                                                                long long total4, int channels, int ho, int wo) {
    const int wo4 = wo >> 2, wi = 2 * wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int xo4 = (int)(i % wo4), yo = (int)((i / wo4) % ho);
        const long long nc = i / ((long long)wo4 * ho);
        const float* base = x + nc * (4LL * ho * wo) + 8 * xo4;
        const float ninf = -__builtin_inff();
        f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int r = -1; r <= 1; ++r) {
            const int y = 2 * yo + r;
            if (y < 0) continue;
            const float* row = base + (long long)y * wi;
            const f32x4 a = *reinterpret_cast<const f32x4*>(row), b = *reinterpret_cast<const f32x4*>(row + 4);
            const float l = xo4 > 0 ? row[-1] : ninf;
            m.x = fmaxf(m.x, fmaxf(l, fmaxf(a.x, a.y)));
            m.y = fmaxf(m.y, fmaxf(a.y, fmaxf(a.z, a.w)));
            m.z = fmaxf(m.z, fmaxf(a.w, fmaxf(b.x, b.y)));
            m.w = fmaxf(m.w, fmaxf(b.y, fmaxf(b.z, b.w)));
        }
        const float bv = bias[(int)(nc % channels)];
        m += bv;
        m.x = fmaxf(m.x, 0.f); m.y = fmaxf(m.y, 0.f); m.z = fmaxf(m.z, 0.f); m.w = fmaxf(m.w, 0.f);
        out[i] = m;
    }
}
}  // namespace

extern "C" int se_bias_relu_maxpool3x3s2_f32(const float* x, const float* bias, float* out, int batch, int channels, int ho, int wo, void* stream) {
    if (batch <= 0 || channels <= 0 || ho <= 0 || wo <= 0 || (wo & 3) || !x || !bias || !out) return SE_ERR_BAD_ARG;
    const long long total4 = (long long)batch * channels * ho * (wo / 4);
    const unsigned grid = (unsigned)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(bias_relu_maxpool_kernel, dim3(grid), dim3(256), 0, se_stream(stream), x, bias, reinterpret_cast<f32x4*>(out), total4, channels,
                       ho, wo);
    SE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose2d(k = 4, stride 2, padding 1) + folded BatchNorm + ReLU of the 2-D pose head (network/pose_resnet.py:205-224,238)
// as ONE GEMM over the un-shifted input plus this assembly pass.  With Z[b][ky][kx][co][j][i] = sum_ci w[ci][co][ky][kx] x[b][ci][j][i]
// (a plain [16 co x ci] x [ci x HW] GEMM per sample: no gathered / shifted copies of x, which for the 2048-channel layer were
// 16 x the input), output row 2j+a / column 2i+c only sees 2 x 2 of the 16 taps:
//     a = 0: (ky, dy) in {(1, 0), (3, -1)}     a = 1: {(0, +1), (2, 0)}          (same table for c, kx, dx)
//     y[b][co][2j+a][2i+c] = bias[co] + sum Z[b][ky][kx][co][j+dy][i+dx]           (terms outside the map are zero)
// One thread per output pixel pair (c = 0, 1): 8 coalesced loads of Z, one 8-byte store.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void deconv2d_k4s2_assemble_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                                     float* __restrict__ out, long long total_pairs, int cout, int h,
                                                                     int w, int relu) {
    typedef float f32x2a __attribute__((ext_vector_type(2)));
    const long long plane = (long long)h * w;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total_pairs; t += (long long)gridDim.x * 256) {
        const int i = (int)(t % w);
        long long r = t / w;
        const int oy = (int)(r % (2 * h)); r /= 2 * h;
        const int co = (int)(r % cout);
        const long long b = r / cout;
        const int a = oy & 1, j = oy >> 1;
        // rows: (ky0, j0) always inside the map; (ky1, j1) may fall outside
        const int ky0 = a ? 2 : 1, ky1 = a ? 0 : 3, j1 = a ? j + 1 : j - 1;
        const bool r1 = (unsigned)j1 < (unsigned)h;
        const float* zb = z + (b * 16 * cout + co) * plane;                     // + (ky * 4 + kx) * cout * plane + jj * w + ii
        auto at = [&](int ky, int kx, int jj, int ii) { return zb[(long long)(ky * 4 + kx) * cout * plane + (long long)jj * w + ii]; };
        const float bv = bias[co];
        // column c = 0: kx 1 at i, kx 3 at i - 1;  c = 1: kx 2 at i, kx 0 at i + 1
        float y0 = bv + at(ky0, 1, j, i), y1 = bv + at(ky0, 2, j, i);
        if (i > 0) y0 += at(ky0, 3, j, i - 1);
        if (i + 1 < w) y1 += at(ky0, 0, j, i + 1);
        if (r1) {
            y0 += at(ky1, 1, j1, i);
            y1 += at(ky1, 2, j1, i);
            if (i > 0) y0 += at(ky1, 3, j1, i - 1);
            if (i + 1 < w) y1 += at(ky1, 0, j1, i + 1);
        }
        if (relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
        *reinterpret_cast<f32x2a*>(out + ((b * cout + co) * 2 * h + oy) * 2 * w + 2 * i) = (f32x2a){y0, y1};
    }
}
}  // namespace

extern "C" int se_deconv2d_k4s2_assemble_f32(const float* z, const float* bias, float* out, int batch, int cout, int h, int w,
                                             int relu, void* stream) {
    if (batch <= 0 || cout <= 0 || h <= 0 || w <= 0 || !z || !bias || !out) return SE_ERR_BAD_ARG;
    const long long pairs = (long long)batch * cout * 2 * h * w;
    const unsigned grid = (unsigned)((pairs + 255) / 256 < 8192 ? (pairs + 255) / 256 : 8192);
    hipLaunchKernelGGL(deconv2d_k4s2_assemble_kernel, dim3(grid), dim3(256), 0, se_stream(stream), z, bias, out, pairs, cout, h, w, relu);
    SE_CHECK_LAUNCH();
    return 0;
}

// bfloat16 form (BASELINE config 3 backbone): 8 elements per lane, float32 arithmetic, one rounding
namespace {
typedef unsigned short u16x8g __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float bf2f_g(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf_g(float f) { const __bf16 b = (__bf16)f; return __builtin_bit_cast(unsigned short, b); }
__global__ __launch_bounds__(256) void bias_act_bf16_kernel(const u16x8g* __restrict__ x, const unsigned short* __restrict__ bias,
                                                            const u16x8g* __restrict__ res, u16x8g* __restrict__ out,
                                                            long long total8, int hw8, int channels, int relu) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const int c = (int)((i / hw8) % channels);
        const float b = bf2f_g(bias[c]);
        const u16x8g xv = x[i];
        u16x8g rv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (res) rv = res[i];
        u16x8g o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = bf2f_g(xv[k]) + b;
            if (res) v += bf2f_g(rv[k]);
            if (relu) v = fmaxf(v, 0.f);
            o[k] = f2bf_g(v);
        }
        out[i] = o;
    }
}
}  // namespace

extern "C" int se_bias_act_nchw_bf16(const se_bf16* x, const se_bf16* bias, const se_bf16* residual, se_bf16* out, int batch,
                                     int channels, int hw, int relu, void* stream) {
    if (batch <= 0 || channels <= 0 || hw <= 0 || (hw & 7)) return SE_ERR_BAD_ARG;
    const long long total8 = (long long)batch * channels * (hw / 8);
    const unsigned grid = (unsigned)((total8 + 255) / 256 < 4096 ? (total8 + 255) / 256 : 4096);
    hipLaunchKernelGGL(bias_act_bf16_kernel, dim3(grid), dim3(256), 0, se_stream(stream), reinterpret_cast<const u16x8g*>(x), bias,
                       reinterpret_cast<const u16x8g*>(residual), reinterpret_cast<u16x8g*>(out), total8, hw / 8, channels, relu);
    SE_CHECK_LAUNCH();
    return 0;
}
