// Shared helpers for the gfx950 kernels of libsceneego_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/sceneego_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SE_WAVE 64

// A/B kernel selector of development builds (se_debug_set_variant, thread-local); the production library has no selector:
// g_variant is the constant 0 there and every `g_variant == n` branch folds away.
#ifdef SE_DEVTOOLS
extern thread_local int g_variant;
#else
static constexpr int g_variant = 0;
#endif

#define SE_CHECK_LAUNCH()                       \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

static inline hipStream_t se_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Per-device launch state.  A process may drive several devices from several threads, so nothing device-dependent is
// cached per process: the CU count and the "dynamic LDS limit raised" flags are kept per device (ids 0..63).
inline int se_current_device() {
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) ? d : 0;
}
// soft-argmax pass-1 partials (softargmax.hip, conv3d.hip): chunks per (sample, joint) row and floats per record (m, l, sx, sy, sz, pad)
#define SE_SA_PART 8
// chunks per row: 32 from batch 8 on (15 rows per sample: 120 rows x 32 chunks = 3840 workgroups for the two-pass form, 256 for the
// fused tail, which runs one workgroup per (chunk, sample)); fewer rows get more chunks so that the fused tail still has ~256
// workgroups (batch 1: 256 chunks of 1024 voxels at 64^3 - it ran 32 workgroups on 256 CUs before: 116 us of a 3.3 ms frame)
inline int se_sa_splits(int rows) { return rows >= 120 ? 32 : rows >= 60 ? 64 : rows >= 30 ? 128 : 256; }

inline int se_num_cus() {
    static std::atomic<int> cus[64];
    const int d = se_current_device();
    int n = cus[d].load(std::memory_order_relaxed);
    if (n <= 0) {
        n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
        cus[d].store(n, std::memory_order_relaxed);
    }
    return n;
}
inline int se_ensure_lds_attr(const void* fn, int bytes, std::atomic<unsigned long long>& mask) {
    const unsigned long long bit = 1ull << se_current_device();
    if (mask.load(std::memory_order_acquire) & bit) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    mask.fetch_or(bit, std::memory_order_release);
    return 0;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (call site, device); returns the hipError_t from the enclosing function
#define SE_ENSURE_LDS(kernel, bytes)                                                                  \
    do {                                                                                              \
        static std::atomic<unsigned long long> m__{0};                                                \
        const int rc__ = se_ensure_lds_attr(reinterpret_cast<const void*>(kernel), (bytes), m__);     \
        if (rc__) return rc__;                                                                        \
    } while (0)

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
