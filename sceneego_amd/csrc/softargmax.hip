// 3D soft-argmax (utils/op.py:83-96): softmax over all G^3 voxels of every (sample, joint) row, then the
// expectation of the voxel-centre coordinates.  HBM-bound: 4 B/voxel/row read (twice, second pass is
// L2/MALL resident at 64^3) + 4 B written (the softmaxed volumes are part of forward()'s return value).
//
// Two launches, split-row so that B*15 rows x se_sa_splits(rows) chunks fill the 256 CUs:
//   pass 1: per chunk  m = max v, l = sum exp(v-m), s = sum exp(v-m) * coord      -> scratch
//   pass 2: every chunk re-derives the row's (M, L) from the row's partials in a fixed order
//           (bitwise deterministic), writes exp(v-M)/L; chunk 0 also writes the joint.
#include "common.h"

// se_sa_splits() / SE_SA_PART (chunks per row, floats per partial record: m, l, sx, sy, sz, pad): common.h - the fused V2V tail
// (pointwise_chain3_sa_kernel in conv3d.hip) writes the same records

namespace {

__device__ __forceinline__ float block_reduce_max(float v, float* sm) {
    v = wave_reduce_max(v);
    const int wid = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[wid] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__device__ __forceinline__ float block_reduce_sum(float v, float* sm) {
    v = wave_reduce_sum(v);
    const int wid = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[wid] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// grid (splits, rows), block 256
__global__ __launch_bounds__(256) void softargmax_partial_kernel(const float* __restrict__ vol,
                                                                 const float* __restrict__ coord,
                                                                 float* __restrict__ scratch, int voxels,
                                                                 int mode, int splits) {
    __shared__ float sm[4];
    const int row = blockIdx.y, s = blockIdx.x;
    const int chunk = (((voxels + splits - 1) / splits) + 3) & ~3;
    const int c0 = s * chunk;
    const int c1 = min(c0 + chunk, voxels);
    const float* v = vol + (size_t)row * voxels;

    float m = -INFINITY;
    if (mode == 1) {
        for (int i = c0 + threadIdx.x * 4; i < c1; i += 1024) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(v + i);
            m = fmaxf(m, fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w)));
        }
        m = block_reduce_max(m, sm);
    }
    float l = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = c0 + threadIdx.x * 4; i < c1; i += 1024) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(v + i);
        const f32x4 c_a = *reinterpret_cast<const f32x4*>(coord + (size_t)i * 3);
        const f32x4 c_b = *reinterpret_cast<const f32x4*>(coord + (size_t)i * 3 + 4);
        const f32x4 c_c = *reinterpret_cast<const f32x4*>(coord + (size_t)i * 3 + 8);
        float e0, e1, e2, e3;
        if (mode == 1) {
            e0 = expf(x.x - m); e1 = expf(x.y - m); e2 = expf(x.z - m); e3 = expf(x.w - m);
        } else {
            e0 = fmaxf(x.x, 0.f); e1 = fmaxf(x.y, 0.f); e2 = fmaxf(x.z, 0.f); e3 = fmaxf(x.w, 0.f);
        }
        l += (e0 + e1) + (e2 + e3);
        sx += e0 * c_a.x + e1 * c_a.w + e2 * c_b.z + e3 * c_c.y;
        sy += e0 * c_a.y + e1 * c_b.x + e2 * c_b.w + e3 * c_c.z;
        sz += e0 * c_a.z + e1 * c_b.y + e2 * c_c.x + e3 * c_c.w;
    }
    l = block_reduce_sum(l, sm);
    sx = block_reduce_sum(sx, sm);
    sy = block_reduce_sum(sy, sm);
    sz = block_reduce_sum(sz, sm);
    if (threadIdx.x == 0) {
        float* p = scratch + ((size_t)row * splits + s) * SE_SA_PART;
        p[0] = m; p[1] = l; p[2] = sx; p[3] = sy; p[4] = sz;
    }
}

// grid (splits, rows), block 256
__global__ __launch_bounds__(256) void softargmax_finish_kernel(const float* __restrict__ vol,
                                                                const float* __restrict__ scratch,
                                                                float* __restrict__ out_vol,
                                                                float* __restrict__ joints, int voxels, int mode, int splits) {
    const int row = blockIdx.y, s = blockIdx.x;
    const float* part = scratch + (size_t)row * splits * SE_SA_PART;
    // The first wave folds the row's partials ONCE per workgroup, in a fixed order (lane k takes chunks k, k + 64, ... in sequence, then
    // a butterfly over the lanes): bitwise deterministic, every workgroup of the row gets the identical (M, L).  (Until round 4 every
    // thread folded all partials itself: with 256 chunks per row - batch 1 - that was 256 expf per thread, 214 us per launch.)
    // A chunk is skipped only when it is EMPTY (k * chunk >= voxels), never for the value of its partial sums: a NaN logit makes its
    // chunk's l NaN, and that must reach L, the joints and the whole row of volumes as it does through torch.softmax + einsum in
    // the reference (utils/op.py:83-96)
    __shared__ float fold[5];
    const int chunk = (((voxels + splits - 1) / splits) + 3) & ~3;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        float M = -INFINITY;
        if (mode == 1) {
            for (int k = lane; k < splits; k += 64)
                if (k * chunk < voxels) M = fmaxf(M, part[k * SE_SA_PART + 0]);
            M = wave_reduce_max(M);
        }
        float L = 0.f, SX = 0.f, SY = 0.f, SZ = 0.f;
        for (int k = lane; k < splits; k += 64) {
            if (k * chunk >= voxels) continue;
            const float* p = part + k * SE_SA_PART;
            const float f = (mode == 1) ? expf(p[0] - M) : 1.f;
            L += p[1] * f; SX += p[2] * f; SY += p[3] * f; SZ += p[4] * f;
        }
        L = wave_reduce_sum(L); SX = wave_reduce_sum(SX); SY = wave_reduce_sum(SY); SZ = wave_reduce_sum(SZ);
        if (lane == 0) { fold[0] = M; fold[1] = L; fold[2] = SX; fold[3] = SY; fold[4] = SZ; }
    }
    __syncthreads();
    const float M = fold[0], L = fold[1], SX = fold[2], SY = fold[3], SZ = fold[4];
    const float invL = (mode == 1) ? 1.f / L : 1.f;
    if (s == 0 && threadIdx.x == 0) {
        joints[row * 3 + 0] = SX * invL;
        joints[row * 3 + 1] = SY * invL;
        joints[row * 3 + 2] = SZ * invL;
    }
    const int c0 = s * chunk;
    const int c1 = min(c0 + chunk, voxels);
    const float* v = vol + (size_t)row * voxels;
    float* o = out_vol + (size_t)row * voxels;
    for (int i = c0 + threadIdx.x * 4; i < c1; i += 1024) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(v + i);
        f32x4 r;
        if (mode == 1) {
            r.x = expf(x.x - M) * invL; r.y = expf(x.y - M) * invL;
            r.z = expf(x.z - M) * invL; r.w = expf(x.w - M) * invL;
        } else {
            r.x = fmaxf(x.x, 0.f); r.y = fmaxf(x.y, 0.f); r.z = fmaxf(x.z, 0.f); r.w = fmaxf(x.w, 0.f);
        }
        *reinterpret_cast<f32x4*>(o + i) = r;
    }
}

}  // namespace

extern "C" long long se_softargmax3d_scratch_elems(int rows) {
    return rows > 0 ? (long long)rows * se_sa_splits(rows) * SE_SA_PART : 0;
}

extern "C" int se_softargmax3d_f32(const float* vol, const float* coord, float* out_vol, float* joints,
                                   float* scratch, int rows, int voxels, int mode, void* stream) {
    if (rows <= 0 || voxels <= 0 || (voxels & 3) || (mode != 0 && mode != 1)) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    const int splits = se_sa_splits(rows);
    dim3 grid(splits, rows);
    hipLaunchKernelGGL(softargmax_partial_kernel, grid, dim3(256), 0, s, vol, coord, scratch, voxels, mode, splits);
    SE_CHECK_LAUNCH();
    hipLaunchKernelGGL(softargmax_finish_kernel, grid, dim3(256), 0, s, vol, scratch, out_vol, joints, voxels, mode, splits);
    SE_CHECK_LAUNCH();
    return 0;
}

// Pass 2 alone: for callers whose pass-1 partials were written by another kernel (se_pointwise_chain3_softargmax_f32).
extern "C" int se_softargmax3d_finish_f32(const float* vol, const float* scratch, float* out_vol, float* joints, int rows,
                                          int voxels, int mode, void* stream) {
    if (rows <= 0 || voxels <= 0 || (voxels & 3) || (mode != 0 && mode != 1)) return SE_ERR_BAD_ARG;
    hipStream_t s = se_stream(stream);
    const int splits = se_sa_splits(rows);
    hipLaunchKernelGGL(softargmax_finish_kernel, dim3(splits, rows), dim3(256), 0, s, vol, scratch, out_vol, joints, voxels, mode, splits);
    SE_CHECK_LAUNCH();
    return 0;
}
