// Same-wave filler budget: one wave per SIMD issues [f32 MFMA, F independent VALU ops] x 1024.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int F, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int nm, float x) {
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    float u = x + threadIdx.x, v = x - threadIdx.x;
    float f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    unsigned g[8] = {threadIdx.x, 2, 3, 4, 5, 6, 7, 8};
    for (int i = 0; i < nm; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u, v, a0, 0, 0, 0);
            else a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, u, a1, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < F; ++q) {
                if (KIND == 0) f[q % 8] = __builtin_fmaf(f[q % 8], 0.999f, 0.5f);
                else if (KIND == 1) g[q % 8] = __builtin_amdgcn_alignbit(g[q % 8], g[q % 8], 19) + g[(q + 1) % 8];
                else f[q % 8] = __builtin_amdgcn_sinf(f[q % 8]);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, F, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    for (int q = 0; q < 8; ++q) s += f[q] + g[q];
    if (s == 123.456f) out[0] = s;
}
template <typename F> float time_us(F&& f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 50.f;
}
#define RUN(F, K) printf("  F=%2d %-6s: %7.2f us\n", F, K == 0 ? "fma" : (K == 1 ? "int" : "sin"), time_us([&] { hipLaunchKernelGGL((k<F, K>), dim3(256), dim3(256), 0, 0, out, 512, 1.0f); }))
int main() {
    float* out; hipMalloc(&out, 64);
    printf("1024 f32 MFMA per wave, F same-wave VALU fillers after each\n");
    RUN(0, 0); RUN(2, 0); RUN(4, 0); RUN(6, 0); RUN(8, 0); RUN(10, 0); RUN(12, 0); RUN(16, 0); RUN(24, 0);
    RUN(4, 1); RUN(8, 1); RUN(12, 1); RUN(16, 1);
    RUN(1, 2); RUN(2, 2); RUN(4, 2);
    return 0;
}
