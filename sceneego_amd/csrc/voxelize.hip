// Depth map -> occupancy grid (scatter).  HBM-bound: 4 B depth + 24 B ray per pixel in, 4 B scattered out.
//
// Follows the reference arithmetic operation by operation so the set of occupied voxels is
// bit-identical (network/voxel_net_depth.py:194-222):
//   depth -> cv2.resize(1024x1024, INTER_NEAREST) -> zero pad 128 cols -> transpose/flatten (x-major)
//   point = ray(float64) * depth ; q = round_half_even(((p + side/2) * G) / side) ; 0 <= q <= G-1 ; occ[q] = 1
// All float64, and NO fused multiply-add: numpy rounds the product and the sum separately.
#include "common.h"

#pragma clang fp contract(off)

namespace {

template <typename T> __device__ __forceinline__ T se_one();
template <> __device__ __forceinline__ float se_one<float>() { return 1.0f; }
template <> __device__ __forceinline__ unsigned short se_one<unsigned short>() { return 0x3F80; }   // bfloat16 1.0

template <typename T>
__device__ __forceinline__ void se_splat(double px, double py, double pz, T* __restrict__ occ_b,
                                          int G, double half_side, double dG, double side, int stride = 1) {
    // (p + side/2) * G / side, evaluated left to right like numpy (voxel_net_depth.py:209-213)
    double qx = ((px + half_side) * dG) / side;
    double qy = ((py + half_side) * dG) / side;
    double qz = (pz * dG) / side;
    qx = rint(qx);  // np.round_: half to even (:215)
    qy = rint(qy);
    qz = rint(qz);
    const double hi = (double)(G - 1);
    if (qx >= 0.0 && qx <= hi && qy >= 0.0 && qy <= hi && qz >= 0.0 && qz <= hi) {  // :216-218
        const int ix = (int)qx, iy = (int)qy, iz = (int)qz;
        occ_b[(((size_t)ix * G + iy) * G + iz) * stride] = se_one<T>();  // benign race: every writer stores 1.0
    }
}

// grid: (ceil(up*up/256), B).  Thread = one pixel (y, x') of the resized depth; x' fastest => the
// ray table (24 B/pixel) and the depth row are read coalesced.
template <typename T>
__global__ __launch_bounds__(256) void voxelize_kernel(const float* __restrict__ depth,
                                                       const double* __restrict__ ray_tab,
                                                       T* __restrict__ occ, int depth_h, int depth_w,
                                                       int up_h, int up_w, int has_pad, int G, double side,
                                                       int stride, long long offset, long long batch_stride) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    T* occ_b = occ + (size_t)b * batch_stride + offset;
    const double half_side = side / 2.0;
    const double dG = (double)G;
    if (has_pad && pix == 0) {
        // the zero-padded columns (:198): depth 0 -> point (0,0,0)
        se_splat(0.0, 0.0, 0.0, occ_b, G, half_side, dG, side, stride);
    }
    if (pix >= up_h * up_w) return;
    const int y = pix / up_w;
    const int xp = pix - y * up_w;
    // cv2 INTER_NEAREST: src = min(floor(dst * (src_size / dst_size)), src_size - 1), scale in double
    int sy = (int)floor((double)y * ((double)depth_h / (double)up_h));
    int sx = (int)floor((double)xp * ((double)depth_w / (double)up_w));
    sy = min(sy, depth_h - 1);
    sx = min(sx, depth_w - 1);
    const double d = (double)depth[((size_t)b * depth_h + sy) * depth_w + sx];
    const double* r = ray_tab + (size_t)pix * 3;
    se_splat(r[0] * d, r[1] * d, r[2] * d, occ_b, G, half_side, dG, side, stride);
}

// Clears the occupancy grid.  A kernel rather than hipMemsetAsync: on ROCm 7.2 a memset node captured into a
// hipGraph from this call replayed with garbage (1e10) in the grid from the second replay on (tools/debug_graph2.py).
__global__ __launch_bounds__(256) void zero_kernel(f32x4* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// strided form: zero 16 bytes (4 float32 / 8 bfloat16 channels) at [offset, ...) of every voxel record
template <typename T>
__global__ __launch_bounds__(256) void zero_strided_kernel(T* __restrict__ p, size_t voxels, int stride, long long offset,
                                                           long long batch_stride) {
    p += (size_t)blockIdx.y * batch_stride + offset;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < voxels; i += (size_t)gridDim.x * 256)
        *reinterpret_cast<f32x4*>(p + i * stride) = (f32x4){0.f, 0.f, 0.f, 0.f};
}

int clear_occupancy(float* occ, size_t elems, hipStream_t s) {
    if (elems & 3) return SE_ERR_BAD_ARG;   // G^3 with even G is a multiple of 8
    const size_t n4 = elems / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<f32x4*>(occ), n4);
    SE_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int se_voxelize_f64(const float* depth, const double* ray_tab, float* occ, int batch, int depth_h,
                               int depth_w, int up, int pad_x, int volume_size, double cuboid_side,
                               void* stream) {
    if (batch <= 0 || depth_h <= 0 || depth_w <= 0 || up <= 0 || volume_size <= 0 || pad_x < 0) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    const int rc = clear_occupancy(occ, (size_t)batch * volume_size * volume_size * volume_size, s);
    if (rc != 0) return rc;
    dim3 grid((up * up + 255) / 256, batch);
    hipLaunchKernelGGL(voxelize_kernel<float>, grid, dim3(256), 0, s, depth, ray_tab, occ, depth_h, depth_w, up, up,
                       pad_x > 0 ? 1 : 0, volume_size, cuboid_side, 1, 0LL, (long long)volume_size * volume_size * volume_size);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_voxelize_strided_f64(const float* depth, const double* ray_tab, float* buf, int batch, int depth_h,
                                       int depth_w, int up, int pad_x, int volume_size, double cuboid_side,
                                       int stride_c, int c_offset, void* stream) {
    if (batch <= 0 || depth_h <= 0 || depth_w <= 0 || up <= 0 || volume_size <= 0 || pad_x < 0) return SE_ERR_BAD_ARG;
    if ((stride_c & 3) || (c_offset & 3) || c_offset + 4 > stride_c) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    const size_t voxels = (size_t)batch * volume_size * volume_size * volume_size;
    const unsigned zgrid = (unsigned)((voxels + 255) / 256 < 4096 ? (voxels + 255) / 256 : 4096);
    hipLaunchKernelGGL(zero_strided_kernel<float>, dim3(zgrid), dim3(256), 0, s, buf, voxels, stride_c, (long long)c_offset, 0LL);
    SE_CHECK_LAUNCH();
    dim3 grid((up * up + 255) / 256, batch);
    hipLaunchKernelGGL(voxelize_kernel<float>, grid, dim3(256), 0, s, depth, ray_tab, buf, depth_h, depth_w, up, up,
                       pad_x > 0 ? 1 : 0, volume_size, cuboid_side, stride_c, (long long)c_offset,
                       (long long)volume_size * volume_size * volume_size * stride_c);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_voxelize_planar3_f64(const float* depth, const double* ray_tab, float* buf, int batch, int depth_h,
                                       int depth_w, int up, int pad_x, int volume_size, double cuboid_side,
                                       int triplets_total, int channel, void* stream) {
    if (batch <= 0 || depth_h <= 0 || depth_w <= 0 || up <= 0 || volume_size <= 0 || pad_x < 0) return SE_ERR_BAD_ARG;
    if (triplets_total <= 0 || channel < 0 || channel >= 3 * triplets_total) return SE_ERR_BAD_ARG;
    // triplet-planar [B][triplets_total][N][3]: scatter only - the slot was zeroed by se_unproject_gather_planar3_f32
    const long long N = (long long)volume_size * volume_size * volume_size;
    dim3 grid((up * up + 255) / 256, batch);
    hipLaunchKernelGGL(voxelize_kernel<float>, grid, dim3(256), 0, se_stream(stream), depth, ray_tab, buf, depth_h, depth_w, up, up,
                       pad_x > 0 ? 1 : 0, volume_size, cuboid_side, 3, (long long)(channel / 3) * N * 3 + channel % 3,
                       (long long)triplets_total * N * 3);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_voxelize_planar1_f64(const float* depth, const double* ray_tab, float* buf, int batch, int depth_h,
                                       int depth_w, int up, int pad_x, int volume_size, double cuboid_side,
                                       int planes_total, int channel, void* stream) {
    if (batch <= 0 || depth_h <= 0 || depth_w <= 0 || up <= 0 || volume_size <= 0 || pad_x < 0) return SE_ERR_BAD_ARG;
    if (planes_total <= 0 || channel < 0 || channel >= planes_total) return SE_ERR_BAD_ARG;
    // planar [B][planes_total][N]: scatter only - the plane was zeroed by se_unproject_gather_planar1_f32
    const long long N = (long long)volume_size * volume_size * volume_size;
    dim3 grid((up * up + 255) / 256, batch);
    hipLaunchKernelGGL(voxelize_kernel<float>, grid, dim3(256), 0, se_stream(stream), depth, ray_tab, buf, depth_h, depth_w, up, up,
                       pad_x > 0 ? 1 : 0, volume_size, cuboid_side, 1, (long long)channel * N, (long long)planes_total * N);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_voxelize_strided_bf16(const float* depth, const double* ray_tab, se_bf16* buf, int batch, int depth_h,
                                        int depth_w, int up, int pad_x, int volume_size, double cuboid_side,
                                        int octs_total, int c_offset, void* stream) {
    if (batch <= 0 || depth_h <= 0 || depth_w <= 0 || up <= 0 || volume_size <= 0 || pad_x < 0) return SE_ERR_BAD_ARG;
    if (octs_total <= 0 || (c_offset & 7) || c_offset < 0 || c_offset / 8 >= octs_total) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    // octet-planar [B][octs_total][N][8]: the occupancy plane is octet c_offset / 8, lane 0 of every 16-byte record
    const long long N = (long long)volume_size * volume_size * volume_size;
    const long long plane = (long long)(c_offset / 8) * N * 8, bstride = (long long)octs_total * N * 8;
    const unsigned zgrid = (unsigned)((N + 255) / 256 < 4096 ? (N + 255) / 256 : 4096);
    hipLaunchKernelGGL(zero_strided_kernel<unsigned short>, dim3(zgrid, batch), dim3(256), 0, s, buf, (size_t)N, 8, plane, bstride);
    SE_CHECK_LAUNCH();
    dim3 grid((up * up + 255) / 256, batch);
    hipLaunchKernelGGL(voxelize_kernel<unsigned short>, grid, dim3(256), 0, s, depth, ray_tab, buf, depth_h, depth_w, up, up,
                       pad_x > 0 ? 1 : 0, volume_size, cuboid_side, 8, plane, bstride);
    SE_CHECK_LAUNCH();
    return 0;
}

extern "C" int se_voxelize_full_f64(const float* depth, const double* ray_tab, float* occ, int batch,
                                    int depth_h, int depth_w, int volume_size, double cuboid_side,
                                    void* stream) {
    if (batch <= 0 || depth_h <= 0 || depth_w <= 0 || volume_size <= 0) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    const int rc = clear_occupancy(occ, (size_t)batch * volume_size * volume_size * volume_size, s);
    if (rc != 0) return rc;
    dim3 grid((depth_h * depth_w + 255) / 256, batch);
    // no resize (up == depth size => sy = y, sx = x) and no padding
    hipLaunchKernelGGL(voxelize_kernel<float>, grid, dim3(256), 0, s, depth, ray_tab, occ, depth_h, depth_w, depth_h,
                       depth_w, 0, volume_size, cuboid_side, 1, 0LL, (long long)volume_size * volume_size * volume_size);
    SE_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Image pre-processing of the demo path (dataset/demo_dataset.py:72-82 + utils/data_transforms.py:38-72) on the device:
// BGR uint8 [B][H][W][3] -> crop `crop_x` columns left and right -> exact 1/4 bilinear resize (cv2.resize INTER_LINEAR maps
// dst d to src 4d + 1.5: the rounded mean of the central 2x2 of each 4x4 block) -> /255 -> (x - mean[c]) / std[c] in float64
// like numpy (the reference applies the RGB statistics to the BGR channels as they come) -> float32 CHW.
// mean3 / std3 are HOST pointers (3 doubles each).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ img, float* __restrict__ out,
                                                            int H, int W, int crop_x, int oh, int ow, double m0, double m1,
                                                            double m2, double s0, double s1, double s2) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= oh * ow) return;
    const int oy = t / ow, ox = t - oy * ow;
    const unsigned char* p = img + ((size_t)b * H + (4 * oy + 1)) * W * 3 + (size_t)(crop_x + 4 * ox + 1) * 3;
    const double mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sum = p[c] + p[3 + c] + p[(size_t)W * 3 + c] + p[(size_t)W * 3 + 3 + c];
        const int q = (sum + 2) >> 2;
        double v = (double)q / 255.0;
        v -= mean[c];
        v /= sd[c];
        out[((size_t)b * 3 + c) * oh * ow + t] = (float)v;
    }
}
}  // namespace

extern "C" int se_preprocess_image_u8(const unsigned char* img, float* out, int batch, int height, int width, int crop_x,
                                      const double* mean3, const double* std3, void* stream) {
    if (batch <= 0 || height <= 0 || width <= 0 || crop_x < 0 || !mean3 || !std3) return SE_ERR_BAD_ARG;
    const int cw = width - 2 * crop_x;
    if (cw <= 0 || (cw & 3) || (height & 3)) return SE_ERR_BAD_ARG;
    const int oh = height / 4, ow = cw / 4;
    dim3 grid((oh * ow + 255) / 256, batch);
    hipLaunchKernelGGL(preprocess_u8_kernel, grid, dim3(256), 0, se_stream(stream), img, out, height, width, crop_x, oh, ow,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    SE_CHECK_LAUNCH();
    return 0;
}
