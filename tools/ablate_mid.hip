// Ablation harness for the mid-size regime (N = 257 .. ~640: Langevin / MF steps at 8-10 us where the
// MFMA work is 3-4 us).  Developer tool, not product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ablate_mid.hip -o tools/ablate_mid && tools/ablate_mid 500 1000
// Times the Langevin step kernel (one state array, one accumulator) for both tile shapes with parts
// removed (ABL bits of ccvm_kernels.h), ping-pong state like the ABI, at steady-state clocks.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../ccvm_amd/csrc/ccvm_kernels.h"

using namespace ccvm;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

__global__ void empty_kernel(int) {}

template <typename F>
float time_us(F&& launch, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 300; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / iters;
}

static void grid_for(StepArgs& a, int ks) {
    a.ks = ks;
    a.nrb = (a.B + BM - 1) / BM;
    a.ncb = (a.N + BN / ks - 1) / (BN / ks);
    a.xr = a.xc = 0;
    const int total = a.nrb * a.ncb;
    if (total % 8 == 0) {
        const int per = total / 8;
        long best = -1;
        for (int xc = 1; xc <= a.ncb; ++xc) {
            if (per % xc || a.ncb % xc) continue;
            const int xr = per / xc;
            if (xr > a.nrb || a.nrb % xr || (a.nrb / xr) * (a.ncb / xc) != 8) continue;
            const long cost = 2L * xr + (4L / ks) * xc;
            if (best < 0 || cost < best) { best = cost; a.xr = xr; a.xc = xc; }
        }
    }
}

template <int MODE, int ABL, int KS, int RING = 0>
float run(const StepArgs& a, const StepArgs& b, int iters) {
    int flip = 0;
    return time_us([&] {
        const StepArgs& x = (flip ^= 1) ? a : b;
        hipLaunchKernelGGL((step_kernel<MODE, false, ABL, KS, false, RING>), dim3(x.nrb * x.ncb), dim3(WG_THREADS), 0, 0, x);
    }, iters);
}

template <int KS>
void sweep(StepArgs a, StepArgs b, int it) {
    grid_for(a, KS);
    grid_for(b, KS);
    printf("---- KS=%d: grid %d (%d x %d tiles of 32 x %d), XCD rect %d x %d\n", KS, a.nrb * a.ncb, a.nrb, a.ncb, BN / KS,
           a.xr, a.xc);
    printf("full, ring 4 / 6 / 8 / 10         : %7.2f / %7.2f / %7.2f / %7.2f us (KS = 1: 8 and 10 do not fit the LDS, ring 6 shown)\n",
           run<MODE_LANGEVIN, 0, KS, 4>(a, b, it), run<MODE_LANGEVIN, 0, KS, 6>(a, b, it),
           run<MODE_LANGEVIN, 0, KS, (KS == 2 ? 8 : 6)>(a, b, it), run<MODE_LANGEVIN, 0, KS, (KS == 2 ? 10 : 6)>(a, b, it));
    printf("no MFMA, ring 4 / 8               : %7.2f / %7.2f us\n", run<MODE_LANGEVIN, 8, KS, 4>(a, b, it),
           run<MODE_LANGEVIN, 8, KS, (KS == 2 ? 8 : 6)>(a, b, it));
    printf("full (default ring)               : %7.2f us\n", run<MODE_LANGEVIN, 0, KS>(a, b, it));
    printf("no noise                     (64) : %7.2f us\n", run<MODE_LANGEVIN, 64, KS>(a, b, it));
    printf("no epilogue                  (16) : %7.2f us\n", run<MODE_LANGEVIN, 16, KS>(a, b, it));
    printf("no epilogue, no noise        (80) : %7.2f us\n", run<MODE_LANGEVIN, 80, KS>(a, b, it));
    printf("no DMA loads                  (1) : %7.2f us\n", run<MODE_LANGEVIN, 1, KS>(a, b, it));
    printf("no MFMA                       (8) : %7.2f us\n", run<MODE_LANGEVIN, 8, KS>(a, b, it));
    printf("no MFMA/noise/epilogue       (88) : %7.2f us\n", run<MODE_LANGEVIN, 88, KS>(a, b, it));
    printf("no MFMA/noise/epi/DMA/reads  (93) : %7.2f us\n", run<MODE_LANGEVIN, 93, KS>(a, b, it));
    printf("MFMA + barriers only         (85) : %7.2f us\n", run<MODE_LANGEVIN, 85, KS>(a, b, it));
    printf("no loop barrier (timing only)(32) : %7.2f us\n", run<MODE_LANGEVIN, 32, KS>(a, b, it));
    printf("MFMA only, no loop barrier  (117) : %7.2f us\n", run<MODE_LANGEVIN, 117, KS>(a, b, it));
    printf("affine epilogue (no noise, x'=f(qx)) [MODE_GD] : %7.2f us\n", run<MODE_GD, 0, KS>(a, b, it));
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 500, B = argc > 2 ? atoi(argv[2]) : 1000;
    const int ld = (N + 127) / 128 * 128, rows = (B + 63) / 64 * 64;
    const size_t state = (size_t)rows * ld;
    float *Q, *V, *c, *c2;
    CK(hipMalloc(&Q, (size_t)ld * ld * 4));
    CK(hipMalloc(&V, ld * 4));
    CK(hipMalloc(&c, state * 4));
    CK(hipMalloc(&c2, state * 4));
    std::vector<float> h(std::max((size_t)ld * ld, state));
    unsigned rng = 12345;
    auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f); };
    for (auto& x : h) x = rnd() * 0.01f;
    CK(hipMemcpy(Q, h.data(), (size_t)ld * ld * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(V, h.data(), ld * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(c, h.data(), state * 4, hipMemcpyHostToDevice));
    CK(hipMemset(c2, 0, state * 4));

    StepArgs a;
    memset(&a, 0, sizeof(a));
    a.Q = Q; a.V = V; a.qsum = V; a.a0 = c; a.o0 = c2;
    a.B = B; a.N = N; a.ld = ld;
    a.in_scale = 0.37f; a.in_shift = 1.0f; a.seed = 42; a.step = 3;
    a.s.lv = LvScalars{-1.0f, -1.0f, 1.0f, 2e-3f, 2e-3f, 0.02f, 0.5f, 1};
    StepArgs b = a;
    b.a0 = c2; b.o0 = c;
    grid_for(a, 1); grid_for(b, 1);
    for (int r = 0; r < 20000; ++r)  // leave the idle clocks (~15 ms of load)
        hipLaunchKernelGGL((step_kernel<MODE_LANGEVIN, false, 0, 1>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, 0, a);
    CK(hipDeviceSynchronize());
    const int it = 3000;
    printf("N=%d B=%d ld=%d\n", N, B, ld);
    printf("empty kernel, 256 x 512 threads   : %7.2f us\n",
           time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, 0, 0); }, it));
    printf("empty kernel, 512 x 512 threads   : %7.2f us\n",
           time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(512), 0, 0, 0); }, it));
    sweep<1>(a, b, it);
    sweep<2>(a, b, it);
    sweep<4>(a, b, it);
    return 0;
}
