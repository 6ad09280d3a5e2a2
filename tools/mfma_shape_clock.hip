// Sustained fp32-MFMA rate by instruction shape on random operands (developer tool): does the chip hold a higher
// clock under v_mfma_f32_16x16x4_f32 than under v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md, DVFS give-back item 7,
// measured there for bf16)?  One wave per SIMD (256 threads per CU-filling block), operands in registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_shape_clock.hip -o tools/mfma_shape_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void loop(const float* in, float* out, int iters) {
    const float a0 = in[threadIdx.x], b0 = in[256 + threadIdx.x];
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = a0 * (1.0f + 0.01f * i); b[i] = b0 * (1.0f - 0.01f * i); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if constexpr (SHAPE == 32) {
        f16v acc[2] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i & 1], 0, 0, 0);
        for (int i = 0; i < 16; ++i) sum += acc[0][i] + acc[1][i];
    } else if constexpr (SHAPE == 16) {
        f4v acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 7], b[i & 7], acc[i & 3], 0, 0, 0);
        for (int i = 0; i < 4; ++i) sum += acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
    } else {
        f4v acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 64; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i & 7], b[i & 7], acc[i & 3], 0, 0, 0);
        for (int i = 0; i < 4; ++i) sum += acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { out[1 << 20 | blockIdx.x * 2] = (float)(t1 - t0); out[1 << 20 | (blockIdx.x * 2 + 1)] = (float)(r1 - r0); }
}
template <int SHAPE>
void run(const float* in, float* out, const char* name, double flop_per_iter_per_wave) {
    const int iters = 200000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 0, clock = 0;
    for (int rep = 0; rep < 12; ++rep) {  // ~2 s of load in all
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((loop<SHAPE>), dim3(grid), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double tf = flop_per_iter_per_wave * iters * grid * 4 / (ms * 1e-3) / 1e12;
        if (rep >= 6) {
            best = best > tf ? best : tf;
            std::vector<float> h(512);
            hipMemcpy(h.data(), out + (1 << 20), 512 * 4, hipMemcpyDeviceToHost);
            clock = h[0] / h[1] * 100.0;  // MHz: shader cycles per 100 MHz tick
        }
    }
    printf("%-28s %7.1f TFLOP/s sustained, in-kernel clock %.0f MHz\n", name, best, clock);
}
int main() {
    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, (2 << 20) * 4);
    std::vector<float> h(1024);
    unsigned rng = 7;
    for (auto& x : h) { rng = rng * 1664525u + 1013904223u; x = ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1e-3f; }
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) {
        run<32>(in, out, "v_mfma_f32_32x32x2_f32", 8 * 2.0 * 32 * 32 * 2);
        run<16>(in, out, "v_mfma_f32_16x16x4_f32", 16 * 2.0 * 16 * 16 * 4);
        run<4>(in, out, "v_mfma_f32_4x4x1_16B_f32", 64 * 2.0 * 4 * 4 * 16);
    }
    return 0;
}
