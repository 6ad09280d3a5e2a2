// Price of non-VALU fillers between f32 MFMAs (one wave per SIMD, 1024 MFMAs per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_cvoid;
// KIND 0 none, 1 one ds_read_b128 per MFMA, 2 one ds_read_b32 per MFMA, 3 one LDS-DMA per 4 MFMAs,
// 4 four SALU ops per MFMA, 5 one global_load_dwordx4 per 4 MFMAs, 6 ds_read_b128 + ds_read2 per MFMA
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* src, int nm, float x) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    float u = x + threadIdx.x, v = x - threadIdx.x;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i;
    __syncthreads();
    f32x4 acc4 = {0, 0, 0, 0};
    float accs = 0.f;
    int sacc = nm;
    const float* g = src + threadIdx.x * 4;
    for (int i = 0; i < nm; ++i) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            __builtin_amdgcn_sched_barrier(0);
            if (h & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, u, a1, 0, 0, 0);
            else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u, v, a0, 0, 0, 0);
            if (KIND == 1 || KIND == 6) acc4 += *reinterpret_cast<const f32x4*>(&lds[((threadIdx.x * 4 + i * 16 + h * 1024) & 16383)]);
            if (KIND == 2 || KIND == 6) accs += lds[(threadIdx.x + i * 64 + h * 4096) & 16383];
            if (KIND == 3 && h == 0)
                __builtin_amdgcn_global_load_lds((global_cvoid*)(g + (i & 63) * 1024), (lds_void*)(lds + (threadIdx.x >> 6) * 256 + ((i & 3) * 1024)), 16, 0, 0);
            if (KIND == 4) { sacc = sacc * 3 + i; sacc ^= (sacc >> 3); }
            if (KIND == 5 && h == 0) acc4 += *reinterpret_cast<const f32x4*>(g + (i & 63) * 1024);
        }
    }
    float s = accs + acc4[0] + acc4[1] + acc4[2] + acc4[3] + sacc;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
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
#define RUN(K, name) printf("  %-44s: %7.2f us\n", name, time_us([&] { hipLaunchKernelGGL((k<K>), dim3(256), dim3(256), 0, 0, out, src, 256, 1.0f); }))
int main() {
    float *out, *src; hipMalloc(&out, 64); hipMalloc(&src, 1 << 22); hipMemset(src, 0, 1 << 22);
    printf("1024 f32 MFMA per wave + fillers (consumed with an accumulate, i.e. +1 VALU each where a value returns)\n");
    RUN(0, "none");
    RUN(1, "ds_read_b128 (+4 VALU add) per MFMA");
    RUN(2, "ds_read_b32 (+1 VALU add) per MFMA");
    RUN(6, "ds_read_b128 + ds_read_b32 per MFMA");
    RUN(3, "global_load_lds_dwordx4 per 4 MFMA");
    RUN(5, "global_load_dwordx4 (+4 VALU) per 4 MFMA");
    RUN(4, "~4 SALU per MFMA");
    return 0;
}
