// Diagnostic (round 3): can ONE wave carry its own staging work inside its MFMA stream?  Shape of the 2-D Winograd inner loop:
// per step 144 (or 72) v_mfma_f32_16x16x4_f32 on 24 accumulators, one ds_read_b128 per pair of MFMAs prefetched two groups ahead,
// plus per step F independent scalar-float VALU instructions (the V-tile transform), D ds_write_b128 (V-tile / weight commit) and
// one workgroup barrier; 512-thread workgroups (two such waves per SIMD), one per CU.
// Compare with the phase-alternating form of conv3d_k3_wino2d_kernel, whose matrix pipe is busy 64 % of the cycles.
// Build: hipcc -O3 --offload-arch=gfx950 tools/diag/mfma_selfmix.hip -o tools/diag/mfma_selfmix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F, int D, int BAR, int GROUPS, int THREADS = 512>
__global__ __launch_bounds__(THREADS) void kself(float* out, const float* in, int steps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 24576; i += THREADS) lds[i] = (float)(i & 15) * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    f32x4 acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    const f32x4* abase = reinterpret_cast<const f32x4*>(lds) + lane;                 // "weights": lane-linear
    const f32x4* bbase = reinterpret_cast<const f32x4*>(lds + 16384) + (lane & 15) * 54 + (lane >> 4);   // "V": 216-float records
    f32x4* wdst = reinterpret_cast<f32x4*>(lds + 8192) + wave * 64 + lane;
    // transform state: F independent fma chains of length ~steps
    float t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = in[lane + i * 64];
    f32x4 oa[3][2], ov[3][2];
    for (int s = 0; s < steps; ++s) {
        auto load_group = [&](int g, int b) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                oa[b][q] = abase[((g * 2 + q) & 31) * 128];
                ov[b][q] = bbase[(g % 6) * 8 + q * 4];
            }
        };
        load_group(0, 0);
        load_group(1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int b = g % 3, xz = g % 6;
            if (g + 2 < GROUPS) load_group(g + 2, (g + 2) % 3);
            // fillers of this group: F * 8 / 8 ... F VALU per MFMA -> 8 F per group
#pragma unroll
            for (int k = 0; k < F * 8; ++k) t[(g * 8 * F + k) & 15] = fmaf(t[(g * 8 * F + k + 5) & 15], 1.0001f, t[(g * 8 * F + k + 9) & 15]);
            if (D > 0 && g < D) {
                f32x4 v = {t[0], t[1], t[2], t[3]};
                wdst[(g & 7) * 512] = v;
            }
            acc[xz * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][0].x, ov[b][0].x, acc[xz * 4 + 0], 0, 0, 0);
            acc[xz * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][0].z, ov[b][0].z, acc[xz * 4 + 1], 0, 0, 0);
            acc[xz * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][1].x, ov[b][1].x, acc[xz * 4 + 2], 0, 0, 0);
            acc[xz * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][1].z, ov[b][1].z, acc[xz * 4 + 3], 0, 0, 0);
            acc[xz * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][0].y, ov[b][0].y, acc[xz * 4 + 0], 0, 0, 0);
            acc[xz * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][0].w, ov[b][0].w, acc[xz * 4 + 1], 0, 0, 0);
            acc[xz * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][1].y, ov[b][1].y, acc[xz * 4 + 2], 0, 0, 0);
            acc[xz * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[b][1].w, ov[b][1].w, acc[xz * 4 + 3], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (g + 2 < GROUPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (F > 0) __builtin_amdgcn_sched_group_barrier(0x002, 2 * F, 0);
                if (D > 0 && g < D && e == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        if (BAR) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 sres = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 24; ++i) sres += acc[i];
    float ts = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) ts += t[i];
    out[blockIdx.x * THREADS + threadIdx.x] = sres.x + sres.y + sres.z + sres.w + ts;
}

template <typename Fn>
double time_ms(Fn f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

#define RUN(F, D, BAR, GROUPS)                                                                                                     \
    {                                                                                                                              \
        auto k = kself<F, D, BAR, GROUPS>;                                                                                         \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);                              \
        double ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(256), dim3(512), 98304, 0, out, in, steps); });                       \
        double flop = 256.0 * 8 * steps * (GROUPS * 8) * (16.0 * 16 * 4 * 2);                                                      \
        printf("VALU/MFMA=%d ds_write_b128/step=%2d barrier=%d MFMAs/step=%3d : %.3f ms  %.1f TF/s  (%.2f of 157.3)\n", F, D, BAR, \
               GROUPS * 8, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);                                                          \
    }

#define RUN1(F, D, BAR, GROUPS)   /* ONE wave per SIMD (256-thread workgroups, one per CU) */                                           \
    {                                                                                                                              \
        auto k = kself<F, D, BAR, GROUPS, 256>;                                                                                    \
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);                              \
        double ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(256), dim3(256), 98304, 0, out, in, steps); });                       \
        double flop = 256.0 * 4 * steps * (GROUPS * 8) * (16.0 * 16 * 4 * 2);                                                      \
        printf("ONE wave per SIMD: VALU/MFMA=%d ds_write_b128/step=%2d barrier=%d MFMAs/step=%3d : %.3f ms  %.1f TF/s  (%.2f of 157.3)\n", F, D, \
               BAR, GROUPS * 8, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);                                                     \
    }

int main() {
    float *out, *in;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&in, 64 * 16 * sizeof(float));
    (void)hipMemset(in, 0, 64 * 16 * sizeof(float));
    const int steps = 4000;
    RUN(0, 0, 0, 18) RUN(1, 0, 0, 18) RUN(2, 0, 0, 18) RUN(3, 0, 0, 18) RUN(4, 0, 0, 18)
    RUN(1, 12, 1, 18) RUN(2, 12, 1, 18) RUN(2, 18, 1, 18) RUN(3, 18, 1, 18)
    RUN(2, 9, 1, 9) RUN(3, 9, 1, 9)
    RUN1(0, 0, 0, 18) RUN1(1, 0, 0, 18) RUN1(0, 0, 1, 18)
    return 0;
}
