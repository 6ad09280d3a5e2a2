// Ablation / stamp harness for the persistent streamed-Q tile kernel (developer tool, not product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ptile_ablate.hip -o /tmp/ptile_ablate && /tmp/ptile_ablate [N B steps]
// Times one launch of `steps` DL steps with parts removed (PtileArgs::abl bits, timing only: results are garbage),
// next to the per-step kernel run `steps` times, and prints where consumer wave 0 spends a step (s_memtime stamps).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../ccvm_amd/csrc/ccvm_ptile.h"
#include "../ccvm_amd/csrc/ccvm_schedule.h"

using namespace ccvm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void set_words(unsigned* p, int n, unsigned v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1000, B = argc > 2 ? atoi(argv[2]) : 1000;
    const int steps = argc > 3 ? atoi(argv[3]) : 1000;
    const int xc_force = argc > 4 ? atoi(argv[4]) : 0;
    const int ld = (N + 127) / 128 * 128, rows = (B + 63) / 64 * 64;
    const size_t state = (size_t)rows * ld;
    float *Q, *V, *c[2], *s[2], *table;
    unsigned *flags, *status;
    unsigned long long* dbg;
    CK(hipMalloc(&Q, (size_t)ld * ld * 4));
    CK(hipMalloc(&V, ld * 4));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&c[i], state * 4)); CK(hipMalloc(&s[i], state * 4)); CK(hipMemset(c[i], 0, state * 4)); CK(hipMemset(s[i], 0, state * 4)); }
    CK(hipMalloc(&table, 4096 * TABLE_WORDS * 4));
    CK(hipMalloc(&flags, 4096 * 4));
    CK(hipMalloc(&status, 128));
    CK(hipMemset(status, 0, 128));
    CK(hipMalloc(&dbg, 256 * 16 * 8));
    std::vector<float> h((size_t)ld * ld, 0.0f);
    unsigned rng = 12345;
    auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f); };
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) h[(size_t)i * ld + j] = rnd() * 0.01f;
    CK(hipMemcpy(Q, h.data(), (size_t)ld * ld * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(V, h.data(), ld * 4, hipMemcpyHostToDevice));
    DlSched sc{8.0, 0.001, 10.0, 100.0, 0.05, 1.0, 2.6457513, 1, 100000, 0, std::min(steps, 4096), nullptr, 0};
    hipLaunchKernelGGL(dl_schedule_kernel, dim3((sc.nsteps + 255) / 256), dim3(256), 0, 0, sc, table);

    PtileArgs a;
    memset(&a, 0, sizeof(a));
    a.Q = Q; a.V = V; a.qsum = V; a.table = table;
    a.x0[0] = c[0]; a.x0[1] = c[1]; a.x1[0] = s[0]; a.x1[1] = s[1];
    a.flags = flags; a.status = status;
    a.seed = 42; a.B = B; a.N = N; a.ld = ld;
    a.nrb = (B + BM - 1) / BM; a.ncb = ld / BN;
    a.in_scale = 0.37f; a.in_shift = 1.0f; a.spin_limit = 50000000u;
    a.step0 = 0; a.nsteps = std::min(steps, 4096);
    const int grid = a.nrb * a.ncb;
    if (grid % 8 == 0) {  // XCD rectangle as the ABI's set_grid picks it (or forced width)
        const int per = grid / 8;
        long best = -1;
        for (int xc = 1; xc <= a.ncb; ++xc) {
            if (per % xc || a.ncb % xc) continue;
            const int xr = per / xc;
            if (xr > a.nrb || a.nrb % xr || (a.nrb / xr) * (a.ncb / xc) != 8) continue;
            long cost = 2L * xr + 4L * xc;
            if (xc_force == xc) cost = 0;
            if (best < 0 || cost < best) { best = cost; a.xr = xr; a.xc = xc; }
        }
    }
    printf("N=%d B=%d ld=%d grid %d (%d x %d), XCD rectangle %d x %d, %d steps per launch\n", N, B, ld, grid, a.nrb, a.ncb, a.xr, a.xc, a.nsteps);

    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int da = 4;
    auto run = [&](int abl, bool stamps) {
        PtileArgs b = a;
        b.abl = abl;
        b.dbg = stamps ? dbg : nullptr;
        hipLaunchKernelGGL(set_words, dim3(16), dim3(256), 0, 0, flags, 4096, 0u);
        hipLaunchKernelGGL((ptile_kernel<MODE_DL, false, true>), dim3(grid), dim3(WG_THREADS), 0, 0, b);
    };
    auto timed = [&](int abl) {
        for (int i = 0; i < 3; ++i) run(abl, false);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 5; ++i) run(abl, false);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        return ms * 1000.f / 5 / a.nsteps;
    };
    {   // reference: the per-step kernel, one launch per step, ping-pong
        StepArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.Q = Q; sa.V = V; sa.qsum = V; sa.B = B; sa.N = N; sa.ld = ld; sa.nrb = a.nrb; sa.ncb = a.ncb; sa.xr = a.xr; sa.xc = a.xc;
        sa.in_scale = a.in_scale; sa.in_shift = a.in_shift; sa.seed = 42;
        sa.s.dl = DlScalars{-1e-4f, -1e-4f, 1.f, -3.f, 1e-3f, 0.1f, 0.03f, 0.03f};
        auto steps_k = [&](int n) {
            for (int i = 0; i < n; ++i) {
                sa.a0 = c[i & 1]; sa.a1 = s[i & 1]; sa.o0 = c[(i & 1) ^ 1]; sa.o1 = s[(i & 1) ^ 1]; sa.step = i;
                hipLaunchKernelGGL((step_kernel<MODE_DL, false, 0>), dim3(grid), dim3(WG_THREADS), 0, 0, sa);
            }
        };
        steps_k(3000);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        steps_k(2000);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("per-step kernel, 2000 launches     : %8.2f us per step\n", ms * 1000.f / 2000);
    }
    struct { int abl; const char* name; } v[] = {
        {0, "persistent, full"}, {1, "A tiles without sc1 (1)"}, {2, "never wait for a flag (2)"}, {3, "1 + 2"},
        {4, "plain stores (4)"}, {8, "no stores (8)"}, {16, "no own-A LDS writes (16)"}, {32, "no noise (32)"},
        {64, "no A DMA (64)"}, {66, "no A DMA, no waits (66)"}, {128, "no landing waits (128)"},
    };
    for (da = 3; da >= 3; --da) {
    for (auto& x : v) printf("%-36s: %8.2f us per step\n", x.name, timed(x.abl));
    for (int abl : {0, 66}) {
        CK(hipMemset(dbg, 0, 256 * 16 * 8));
        run(abl, true);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hd(256 * 16);
        CK(hipMemcpy(hd.data(), dbg, 256 * 16 * 8, hipMemcpyDeviceToHost));
        const char* names[8] = {"M -> B2", "main loop", "loop end -> N", "epilogue", "epilogue end -> M", "  (barrier waits in loop)", "polls (producer 0)", "failed must-polls"};
        printf("stamps, abl %d, per step (consumer wave 0; s_memtime ticks):\n", abl);
        for (int k = 0; k < 8; ++k) {
            std::vector<double> w;
            for (int g = 0; g < grid && g < 256; ++g) w.push_back((double)hd[g * 16 + k] / a.nsteps);
            std::sort(w.begin(), w.end());
            printf("  %-26s: min %9.1f  median %9.1f  max %9.1f\n", names[k], w.front(), w[w.size() / 2], w.back());
        }
    }
    }
    unsigned st = 0;
    CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    printf("status word %u\n", st);
    return 0;
}
