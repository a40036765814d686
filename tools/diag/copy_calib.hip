// FETCH_SIZE / WRITE_SIZE calibration: streaming copies of a KNOWN byte count with the access widths the conv kernels use
// (16 B/lane contiguous, 8 B/lane contiguous, 8 B/lane in 32-byte pieces at a 128-byte stride = the channels-last halo fetch).
// Build: hipcc -O3 --offload-arch=gfx950 tools/diag/copy_calib.hip -o tools/diag/copy_calib ; run under rocprofv3 --pmc.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void copy16(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy8(const f32x2* __restrict__ in, f32x2* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// reads 32 of every 128 bytes (one 8-channel chunk of 32-channel records), 8 B per lane; writes them densely
__global__ void gather32of128(const f32x2* __restrict__ in, f32x2* __restrict__ out, size_t nrec, int chunk) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec * 4; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[(i >> 2) * 16 + chunk * 4 + (i & 3)];
}
int main() {
    const size_t bytes = 1ull << 30;    // 1 GiB source: far beyond the 256 MiB Infinity Cache
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(copy16, dim3(4096), dim3(256), 0, 0, (const f32x4*)a, (f32x4*)b, bytes / 16);
        hipLaunchKernelGGL(copy8, dim3(4096), dim3(256), 0, 0, (const f32x2*)a, (f32x2*)b, bytes / 8);
        hipLaunchKernelGGL(gather32of128, dim3(4096), dim3(256), 0, 0, (const f32x2*)a, (f32x2*)b, bytes / 128, rep);
    }
    (void)hipDeviceSynchronize();
    printf("copy16 / copy8: %zu bytes read + %zu bytes written per launch; gather32of128: %zu bytes read (useful; %zu bytes of lines touched) + %zu written\n",
           bytes, bytes, bytes / 4, bytes, bytes / 4);
    return 0;
}
