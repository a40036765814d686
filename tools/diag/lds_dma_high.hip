// Does global_load_lds_dwordx4 reach LDS addresses above 64 KiB on gfx950 (M0 as the destination base)?
// Build: hipcc -O3 --offload-arch=gfx950 tools/diag/lds_dma_high.hip -o tools/diag/lds_dma_high ; prints the read-back per base.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* __restrict__ src, float* out, int base_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 40960; i += 64) lds[i] = -1.f;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + tid * 4),
                                     (void __attribute__((address_space(3)))*)(lds + base_floats), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    out[tid] = lds[base_floats + tid * 4 + 1];
    if (tid == 0) out[64] = lds[(base_floats & 16383) + 1];   // where a 16-bit wrap would land
}
int main() {
    float *src, *out;
    hipMalloc(&src, 1024);
    hipMalloc(&out, 1024);
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int base : {0, 8192, 16128, 16384, 20480, 32768, 36864}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, src, out, base);
        float r[65];
        hipMemcpy(r, out, 65 * 4, hipMemcpyDeviceToHost);
        printf("base %6d B: lane0 %.0f lane1 %.0f lane63 %.0f (expect 1 5 253); wrapped slot %.0f\n", base * 4, r[0], r[1], r[63], r[64]);
    }
    return 0;
}
