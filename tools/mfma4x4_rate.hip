// Issue rate of v_mfma_f32_4x4x1_16B_f32 vs the number of independent accumulators (developer tool).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_rate.hip -o tools/mfma4x4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(64) void k4(float* out, int n, float x) {
    f4 acc[NACC];
    for (int p = 0; p < NACC; ++p) acc[p] = f4{0, 0, 0, 0};
    float a = x + threadIdx.x, b = x - threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m % NACC], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int p = 0; p < NACC; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = (float)(t1 - t0); out[blockIdx.x * 2 + 1] = s; }
}
template <int... K>
__device__ __forceinline__ void chain(float a, const float* b, f4* acc, std::integer_sequence<int, K...>) {
    ((acc[K & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b[K], acc[K & 3], 4, K % 16, 0)), ...);
}
// as the persistent kernel issues them: A broadcast from block ABID, 16 distinct B registers
template <int NACC>
__global__ __launch_bounds__(64) void kb(float* out, int n, float x) {
    f4 acc[4];
    for (int p = 0; p < 4; ++p) acc[p] = f4{0, 0, 0, 0};
    float a = x + threadIdx.x, b[16];
    for (int i = 0; i < 16; ++i) b[i] = x - threadIdx.x * (i + 1);
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) chain(a, b, acc, std::make_integer_sequence<int, 16>{});
    long long t1 = clock64();
    float s = 0;
    for (int p = 0; p < 4; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = (float)(t1 - t0); out[blockIdx.x * 2 + 1] = s; }
}
// NV independent VALU instructions (integer add / rotate / xor, as the generator's) after every MFMA:
// does a wave that is alone on its SIMD overlap them with the matrix pipe?
template <int NV>
__global__ __launch_bounds__(64) void kmix(float* out, int n, float x) {
    f4 acc[4];
    for (int p = 0; p < 4; ++p) acc[p] = f4{0, 0, 0, 0};
    float a = x + threadIdx.x, b = x - threadIdx.x;
    unsigned u0 = threadIdx.x * 2654435761u, u1 = threadIdx.x + 12345u, u2 = u0 ^ 77u, u3 = u1 + 99u;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {  // four independent chains: add, rotate, xor
                unsigned& w = (v & 3) == 0 ? u0 : (v & 3) == 1 ? u1 : (v & 3) == 2 ? u2 : u3;
                w = (v % 3 == 0) ? w + 0x9E3779B9u : (v % 3 == 1) ? __builtin_amdgcn_alignbit(w, w, 19) : (w ^ (unsigned)m);
                asm volatile("" : "+v"(w));
            }
        }
    }
    long long t1 = clock64();
    float s = (float)(u0 ^ u1 ^ u2 ^ u3);
    for (int p = 0; p < 4; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = (float)(t1 - t0); out[blockIdx.x * 2 + 1] = s; }
}
template <int NACC>
__global__ __launch_bounds__(64) void k16(float* out, int n, float x) {
    f4 acc[NACC];
    for (int p = 0; p < NACC; ++p) acc[p] = f4{0, 0, 0, 0};
    float a = x + threadIdx.x, b = x - threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % NACC], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int p = 0; p < NACC; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = (float)(t1 - t0); out[blockIdx.x * 2 + 1] = s; }
}
template <typename K>
void run(const char* name, K kern, float* d) {
    const int n = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1, 1024, 2048}) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, n, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d, n, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%-22s grid %4d: %7.1f us, %6.2f ns per MFMA per wave (clock64 delta %.0f)\n", name, grid, ms * 1e3, ms * 1e6 / (n * 16.0), h[0]);
    }
}
int main() {
    float* d; hipMalloc(&d, 2048 * 8);
    run("4x4x1 NACC=1", k4<1>, d); run("4x4x1 NACC=2", k4<2>, d); run("4x4x1 NACC=4", k4<4>, d); run("4x4x1 NACC=8", k4<8>, d);
    run("4x4x1 cbsz=4 abid=k", kb<4>, d);
    run("4x4x1 + 0 VALU per MFMA", kmix<0>, d); run("4x4x1 + 1 VALU per MFMA", kmix<1>, d); run("4x4x1 + 2 VALU per MFMA", kmix<2>, d); run("4x4x1 + 4 VALU per MFMA", kmix<4>, d);
    run("16x16x4 NACC=1", k16<1>, d); run("16x16x4 NACC=2", k16<2>, d); run("16x16x4 NACC=4", k16<4>, d);
    return 0;
}
