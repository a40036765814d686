// Diagnostic: MFMA rate of a loop shaped like the Winograd inner loop: per 8 MFMAs, R ds_read_b128 + V valu ops.
// 512-thread WG (2 waves/SIMD), 1 WG per CU.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R, int V, bool PIPE, int NT>
__global__ __launch_bounds__(NT) void kmix(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += NT) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    const f32x4* base = reinterpret_cast<const f32x4*>(lds) + lane;
    f32x4 w0 = base[0], w1 = base[64], x = base[128];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 nw0 = w0, nw1 = w1, nx = x;
            const int o = ((it * 4 + s) & 15) * 192;
            if (R >= 1) nw0 = base[o];
            if (R >= 2) nw1 = base[o + 64];
            if (R >= 3) nx = base[o + 128];
            f32x4 xv = x;
            if (V == 4) xv = x - nx * 0.f + w0 * 0.f;   // packed-ish vector math
            if (V == 8) { xv.x = x.x - nx.y; xv.y = x.y + nx.z; xv.z = x.z - nx.w; xv.w = x.w + nx.x; }   // 4 scalar adds
            acc[s * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, xv.x, acc[s * 2 + 0], 0, 0, 0);
            acc[s * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, xv.x, acc[s * 2 + 1], 0, 0, 0);
            acc[s * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, xv.y, acc[s * 2 + 0], 0, 0, 0);
            acc[s * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, xv.y, acc[s * 2 + 1], 0, 0, 0);
            acc[s * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, xv.z, acc[s * 2 + 0], 0, 0, 0);
            acc[s * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, xv.z, acc[s * 2 + 1], 0, 0, 0);
            acc[s * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, xv.w, acc[s * 2 + 0], 0, 0, 0);
            acc[s * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, xv.w, acc[s * 2 + 1], 0, 0, 0);
            if (PIPE) {
                __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            w0 = nw0; w1 = nw1; x = nx;
        }
    }
    f32x4 sres = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) sres += acc[i];
    out[blockIdx.x * NT + threadIdx.x] = sres.x + sres.y + sres.z + sres.w;
}

template <typename F>
double time_ms(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

#define RUN(R, V, P, NT)                                                                                     \
    {                                                                                                        \
        double ms = time_ms([&] { hipLaunchKernelGGL((kmix<R, V, P, NT>), dim3(256), dim3(NT), 0, 0, out, iters); }); \
        double flop = 256.0 * (NT / 64) * iters * 32 * (16.0 * 16 * 4 * 2);                                  \
        printf("waves/SIMD=%d R=%d V=%d pipe=%d : %.3f ms  %.1f TF/s\n", NT / 256, R, V, (int)P, ms, flop / ms / 1e9); \
    }

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 20000;
    RUN(3, 0, true, 512) RUN(3, 4, true, 512) RUN(3, 8, true, 512) RUN(3, 8, false, 512)
    RUN(3, 0, true, 256) RUN(3, 4, true, 256) RUN(3, 8, true, 256) RUN(3, 8, false, 256)
    RUN(3, 0, true, 1024) RUN(3, 8, true, 1024)
    return 0;
}
