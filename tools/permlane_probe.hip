#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float swap_add16(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("" : "+v"(b));
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add32(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("" : "+v"(b));
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__global__ void k(float* out) {
    float v = out[threadIdx.x];
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v = swap_add16(v);
    v = swap_add32(v);
    out[threadIdx.x] = v;
}
int main() {
    float h[64], *d; for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i / 4)) * (1 + (i % 4) * 65536.0f * 0 ) + (i % 4) * 0.0f;
    // lane l = 4 b + j: value 2^b -> every lane should end with 65535 ; add j-dependent term to check lanes keep their j
    for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i / 4)) + 100000.0f * (i % 4);
    hipMalloc(&d, 256); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%.0f%c", h[i], i % 16 == 15 ? '\n' : ' ');
    return 0;
}
