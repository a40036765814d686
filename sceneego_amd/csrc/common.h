// Shared helpers for the gfx950 kernels of libsceneego_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sceneego_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SE_WAVE 64

#define SE_CHECK_LAUNCH()                       \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

static inline hipStream_t se_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
