// Diagnostic: practical f32 MFMA ceiling on this MI355X (v_mfma_f32_16x16x4_f32 and 32x32x2), with and without
// LDS operand reads, at 1/2/4 waves per SIMD.  Not part of the product.  hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDSREAD>
__global__ __launch_bounds__(256) void k16(float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 x = {a, a, a, a}, w = {b, b, b, b};
        if (LDSREAD) {
            x = *reinterpret_cast<f32x4*>(&lds[((threadIdx.x * 4 + it * 64) & 8188)]);
            w = *reinterpret_cast<f32x4*>(&lds[((threadIdx.x * 4 + it * 128 + 2048) & 8188)]);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc[i], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 0.001f, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
double time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
    unsigned long long* clk; hipMalloc(&clk, 16);
    const int iters = 20000;
    for (int wg_per_cu : {1, 2, 4}) {
        int grid = 256 * wg_per_cu;
        {
            double ms = time_ms([&] { hipLaunchKernelGGL((k16<8, false>), dim3(grid), dim3(256), 0, 0, out, iters, clk); });
            double flop = (double)grid * 4 * iters * 4 * 8 * (16.0 * 16 * 4 * 2);
            unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("16x16x4 regs   %d wave/SIMD: %.3f ms  %.1f TF/s  in-kernel clock %.0f MHz\n", wg_per_cu, ms, flop / ms / 1e9, (double)h[0] / (double)h[1] * 100.0);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL((k16<8, true>), dim3(grid), dim3(256), 0, 0, out, iters, clk); });
            double flop = (double)grid * 4 * iters * 4 * 8 * (16.0 * 16 * 4 * 2);
            unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            printf("16x16x4 +LDS   %d wave/SIMD: %.3f ms  %.1f TF/s  in-kernel clock %.0f MHz\n", wg_per_cu, ms, flop / ms / 1e9, (double)h[0] / (double)h[1] * 100.0);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL((k32<4>), dim3(grid), dim3(256), 0, 0, out, iters / 2); });
            double flop = (double)grid * 4 * (iters / 2) * 4 * 4 * (32.0 * 32 * 2 * 2);
            printf("32x32x2 regs   %d wave/SIMD: %.3f ms  %.1f TF/s\n", wg_per_cu, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
