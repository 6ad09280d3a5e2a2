// Ablation / calibration harness for the step kernel (developer tool, not product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ablate.hip -o tools/ablate && tools/ablate
// Times (a) a bare v_mfma_f32_32x32x2_f32 loop = the MFMA roofline as THIS chip delivers it,
// (b) the DL step kernel at N=1000, B=1000 with parts removed (ABL bits, see ccvm_kernels.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

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

// 1024 MFMAs per wave on two alternating accumulators, operands in registers.
__global__ __launch_bounds__(256) void mfma_only(float* out, int n, float x) {
    f32x16 a0, a1;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    float u = x + threadIdx.x, v = x - threadIdx.x;
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u, v, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, u, a1, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    if (s == 123.456f) out[0] = s;
}

template <typename F>
float time_us(F&& launch, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / iters;
}

template <int ABL>
float run_variant(const StepArgs& a, int iters) {
    return time_us([&] { hipLaunchKernelGGL((step_kernel<MODE_DL, false, ABL>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, 0, a); }, iters);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1000, B = argc > 2 ? atoi(argv[2]) : 1000;
    const int ld = argc > 3 ? atoi(argv[3]) : (N + 127) / 128 * 128, rows = (B + 63) / 64 * 64;
    const size_t state = (size_t)rows * ld;
    float *Q, *V, *c, *s, *c2, *s2, *out;
    CK(hipMalloc(&Q, (size_t)ld * ld * 4));
    CK(hipMalloc(&V, ld * 4));
    CK(hipMalloc(&c, state * 4)); CK(hipMalloc(&s, state * 4));
    CK(hipMalloc(&c2, state * 4)); CK(hipMalloc(&s2, state * 4));
    CK(hipMalloc(&out, 1024));
    std::vector<float> h((size_t)ld * ld);
    unsigned rng = 12345;
    auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f); };
    for (auto& x : h) x = rnd() * 0.01f;
    CK(hipMemcpy(Q, h.data(), (size_t)ld * ld * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(V, h.data(), ld * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < state; ++i) h[i] = rnd();
    CK(hipMemcpy(c, h.data(), state * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(s, h.data(), state * 4, hipMemcpyHostToDevice));
    CK(hipMemset(c2, 0, state * 4)); CK(hipMemset(s2, 0, state * 4));

    const int nm = 512;  // 2 MFMAs per iteration -> 1024 per wave, as the N=1000 DL step
    float t = time_us([&] { hipLaunchKernelGGL(mfma_only, dim3(256), dim3(256), 0, 0, out, nm, 1.0f); }, 50);
    printf("mfma_only 256 WG x 4 waves x 1024 MFMA: %8.2f us  (%.1f TF, %.1f cycles/MFMA @2.4GHz)\n", t,
           256.0 * 4 * 1024 * 4096 / t / 1e6, t * 2400.0 / 1024);
    t = time_us([&] { hipLaunchKernelGGL(mfma_only, dim3(512), dim3(256), 0, 0, out, nm, 1.0f); }, 50);
    printf("mfma_only 512 WG (2 waves/SIMD)        : %8.2f us  (%.1f TF)\n", t, 512.0 * 4 * 1024 * 4096 / t / 1e6);

    StepArgs a;
    memset(&a, 0, sizeof(a));
    a.Q = Q; a.V = V; a.a0 = c; a.a1 = s; a.o0 = c2; a.o1 = s2;
    a.B = B; a.N = N; a.ld = ld; a.nrb = (B + BM - 1) / BM; a.ncb = (N + BN - 1) / BN;
    a.in_scale = 0.37f; a.in_shift = 1.0f; a.seed = 42; a.step = 3;
    a.s.dl = DlScalars{-1e-4f, -1e-4f, 1.f, -3.f, 1e-3f, 0.1f, 0.03f, 0.03f};
    a.qsum = V;
    unsigned long long* dbg;
    CK(hipMalloc(&dbg, 256 * 8 * 8));
    CK(hipMemset(dbg, 0, 256 * 8 * 8));
    a.dbg = dbg;
    // steady-state clocks: the chip needs ~15 ms of load to leave its idle clocks, and a variant that
    // runs for 2 ms is timed in whatever state the previous one left (run-to-run +-1 us): spin first,
    // then 1000 launches per variant behind 200 warm-up launches
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 6000; ++r)
            hipLaunchKernelGGL((step_kernel<MODE_DL, false, 0>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, 0, a);
        hipDeviceSynchronize();
    }
    const int it = 1000;
    printf("grid %d x %d threads, N=%d B=%d ld=%d\n", a.nrb * a.ncb, WG_THREADS, N, B, ld);
    printf("full                         : %8.2f us\n", run_variant<0>(a, it));
    {   // as the ABI runs it: ping-pong state buffers (each step reads what the previous one wrote)
        StepArgs b = a;
        b.a0 = c2; b.a1 = s2; b.o0 = c; b.o1 = s;
        int flip = 0;
        auto pp = [&] {
            const StepArgs& x = (flip ^= 1) ? a : b;
            hipLaunchKernelGGL((step_kernel<MODE_DL, false, 0>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, 0, x);
        };
        printf("full, ping-pong state        : %8.2f us\n", time_us(pp, it));
        if ((a.nrb * a.ncb) % 64 == 0) {
            a.xr = a.nrb / 8; a.xc = a.ncb; b.xr = a.xr; b.xc = a.xc;
            printf("full, ping-pong, XCD %2dx%-2d   : %8.2f us\n", a.xr, a.xc, time_us(pp, it));
            printf("full, XCD rect, no ping-pong : %8.2f us\n", run_variant<0>(a, it));
            a.xr = a.xc = b.xr = b.xc = 0;
        }
    }
    {   // the same chain as a hipGraph (does graph submission shorten the kernel-to-kernel hand-over?)
        hipStream_t st; hipStreamCreate(&st);
        StepArgs b = a; b.a0 = c2; b.a1 = s2; b.o0 = c; b.o1 = s;
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 50; ++i)
            hipLaunchKernelGGL((step_kernel<MODE_DL, false, 0>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, st, (i & 1) ? b : a);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int r = 0; r < 4; ++r) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("full, ping-pong, hipGraph x50: %8.2f us\n", ms * 1000.f / 1000);
        hipEventRecord(e0, st);
        for (int r = 0; r < 1000; ++r)
            hipLaunchKernelGGL((step_kernel<MODE_DL, false, 0>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, st, (r & 1) ? b : a);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("full, ping-pong, stream x1000: %8.2f us\n", ms * 1000.f / 1000);
    }
    printf("no epilogue            (16)  : %8.2f us\n", run_variant<16>(a, it));
    printf("no DMA loads            (1)  : %8.2f us\n", run_variant<1>(a, it));
    printf("no frag reads           (4)  : %8.2f us\n", run_variant<4>(a, it));
    printf("no noise               (64)  : %8.2f us\n", run_variant<64>(a, it));
    printf("no loads, no noise     (65)  : %8.2f us\n", run_variant<65>(a, it));
    printf("no loads/reads/noise   (69)  : %8.2f us\n", run_variant<69>(a, it));
    printf("... and no epilogue    (85)  : %8.2f us\n", run_variant<85>(a, it));
    printf("no loop barrier        (32)  : %8.2f us\n", run_variant<32>(a, it));
    printf("no barrier, no DMA     (33)  : %8.2f us\n", run_variant<33>(a, it));
    printf("no barrier, no noise   (96)  : %8.2f us\n", run_variant<96>(a, it));
    printf("no barrier/noise/epi  (112)  : %8.2f us\n", run_variant<112>(a, it));
    printf("no MFMA                 (8)  : %8.2f us\n", run_variant<8>(a, it));
    printf("no MFMA, no epilogue   (24)  : %8.2f us\n", run_variant<24>(a, it));
    printf("no MFMA, no noise      (72)  : %8.2f us\n", run_variant<72>(a, it));
    printf("no MFMA, no DMA         (9)  : %8.2f us\n", run_variant<9>(a, it));
    printf("no MFMA, no reads      (12)  : %8.2f us\n", run_variant<12>(a, it));
    printf("no MFMA/noise/epilogue (88)  : %8.2f us\n", run_variant<88>(a, it));
    printf("no MFMA/noise/epi/DMA  (89)  : %8.2f us\n", run_variant<89>(a, it));
    printf("no MFMA/noise/epi/reads(92)  : %8.2f us\n", run_variant<92>(a, it));
    printf("no MFMA/noise/epi/DMA/reads(93): %6.2f us\n", run_variant<93>(a, it));
    printf("... and no barrier    (125)  : %8.2f us\n", run_variant<125>(a, it));
    {   // phase shares from the stamped build (cycles per tile, median over workgroups)
        hipLaunchKernelGGL((step_kernel<MODE_DL, false, 128>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hd(256 * 8);
        CK(hipMemcpy(hd.data(), dbg, 256 * 8 * 8, hipMemcpyDeviceToHost));
        const char* names[6] = {"P dma issue", "P noise", "P vmcnt wait", "P barrier wait", "C work (reads+MFMA)", "C barrier wait"};
        const int nkt = (N + 31) / 32, nwg = a.nrb * a.ncb;
        for (int k = 0; k < 6; ++k) {
            std::vector<double> v;
            for (int w = 0; w < nwg && w < 256; ++w) v.push_back((double)hd[w * 8 + k] / nkt);
            std::sort(v.begin(), v.end());
            printf("%-22s: min %7.0f  median %7.0f  max %7.0f cycles/tile\n", names[k], v.front(), v[v.size() / 2], v.back());
        }
    }
    return 0;
}
