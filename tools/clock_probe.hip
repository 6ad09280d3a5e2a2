// Core clock under light and heavy load (developer tool): the shader clock (s_memtime, clock64) against the constant
// 100 MHz reference (s_memrealtime, wall_clock64) around a fixed dependent-VALU loop, sampled launch by launch.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe && tools/clock_probe
// Phases: LIGHT (250 workgroups x 4 waves: the occupancy of the row-owner kernel at N = 20, B = 1000) from idle for
// 400 ms, HEAVY (4096 x 256 threads, MFMA loop) for 300 ms, LIGHT again for 400 ms -- does the clock follow the load?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned long long* out, int iters, int mfma) {
    const unsigned long long r0 = wall_clock64(), t0 = clock64();
    float x = threadIdx.x * 1e-3f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
        if (mfma) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
        }
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (x + acc[0] == 12345.678f) out[0] = 0;
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 4096 * 16);
    std::vector<unsigned long long> h(2 * 4096);
    auto phase = [&](const char* name, int grid, int threads, int mfma, double ms) {
        const auto start = std::chrono::steady_clock::now();
        int launches = 0;
        double next_print = 0.0;
        while (true) {
            hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), 0, 0, d, mfma ? 2000 : 4000, mfma);
            hipDeviceSynchronize();
            ++launches;
            const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count();
            if (t >= next_print) {
                hipMemcpy(h.data(), d, 16, hipMemcpyDeviceToHost);
                printf("%-6s t = %6.1f ms  launch %5d  shader clock %.0f MHz  (%.1f us per launch)\n", name, t, launches,
                       100.0 * (double)h[0] / (double)h[1], (double)h[1] / 100.0);
                next_print += ms / 12.0;
            }
            if (t > ms) break;
        }
    };
    phase("light", 250, 256, 0, 400.0);
    phase("HEAVY", 4096, 256, 1, 300.0);
    phase("light", 250, 256, 0, 400.0);
    phase("1wave", 16, 64, 0, 200.0);
    return 0;
}
