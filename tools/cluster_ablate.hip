// Ablation harness for the column-cluster persistent kernel (developer tool): Langevin (or DL: -DCCVM_ABL_DL, MF: -DCCVM_ABL_MF), N = 500, B = 1000.
//   for b in 0 1 2 4 ...; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DCCVM_CLUSTER_ABL=$b tools/cluster_ablate.hip -o tools/cluster_ablate_$b; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../ccvm_amd/csrc/ccvm_cluster.h"
#include "../ccvm_amd/csrc/ccvm_schedule.h"
using namespace ccvm;
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 500, B = argc > 2 ? atoi(argv[2]) : 1000;
    const int ld = (N + 127) / 128 * 128, rows = (B + 63) / 64 * 64, steps = 4096;
    const size_t state = (size_t)rows * ld;
    float *Q, *V, *c, *c2, *xb0, *xb1, *table; unsigned* sync;
    hipMalloc(&Q, (size_t)ld * ld * 4); hipMalloc(&V, ld * 4); hipMalloc(&c, state * 4); hipMalloc(&c2, state * 4); hipMemset(c2, 0, state * 4); hipMalloc(&xb0, state * 16);
    hipMalloc(&xb1, state * 16); hipMalloc(&table, steps * TABLE_WORDS * 4); hipMalloc(&sync, 1 << 20);
    std::vector<float> h((size_t)ld * ld, 0.f);
    unsigned rng = 1;
    auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f); };
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) h[(size_t)i * ld + j] = rnd() * 0.02f;
    hipMemcpy(Q, h.data(), (size_t)ld * ld * 4, hipMemcpyHostToDevice);
    hipMemcpy(V, h.data(), ld * 4, hipMemcpyHostToDevice);
    hipMemset(c, 0, state * 4); 
#ifdef CCVM_ABL_DL
    constexpr int MODE = MODE_DL;
    DlSched sc{2.5, 0.002, 10.0, 1.0, 0.05, 1.0, 1.2247, 1, 15000, 0, steps};
    hipLaunchKernelGGL(dl_schedule_kernel, dim3((steps + 255) / 256), dim3(256), 0, 0, sc, table);
#elif defined(CCVM_ABL_MF)
    constexpr int MODE = MODE_MF;  // the example's parameters (examples/ccvm_boxqp_mf.py: pump 0, feedback_scale 4000, j 5, S 20, dt 0.0025)
    MfSched sc{0.0, 0.0025, 5.0, 4000.0, 0.01, 20.0, 1.0, 1, 15000, 0, steps, AdamSched{}};
    hipLaunchKernelGGL(mf_schedule_kernel, dim3((steps + 255) / 256), dim3(256), 0, 0, sc, table);
#else
    constexpr int MODE = MODE_LANGEVIN;
    LvSched sc{0.002, 0.5, 1.0, 0.5, 2.0, 1.0, 1, 1, 15000, 0, steps, AdamSched{}};
    hipLaunchKernelGGL(lv_schedule_kernel, dim3((steps + 255) / 256), dim3(256), 0, 0, sc, table);
#endif
    ClusterArgs a; memset(&a, 0, sizeof(a));
    a.Q = Q; a.V = V; a.qsum = V; a.x0 = c; a.x1 = c2; a.xb0 = xb0; a.xb1 = xb1; a.table = table; a.seed = 7; a.nsteps = steps;
    a.status = sync; a.spin_limit = getenv("CL_SPIN") ? (unsigned)atoi(getenv("CL_SPIN")) : 200000000u;  // ticks of 100 MHz
    a.B = B; a.N = N; a.ld = ld; a.in_scale = 1.0f; a.in_shift = 0.5f;
    a.k_first = 4.47f; a.S = 20.0f;  // MF only: sqrt(1 / (4 j)) / sqrt(dt), the measured amplitude's clamp
    const int crows = ld > 512 ? 48 : 32;  // three row sets above K = 512
    a.nclusters = (B + crows - 1) / crows; a.G = (N + 63) / 64;
    a.spread = getenv("CL_SPREAD") ? atoi(getenv("CL_SPREAD")) : 0;
    // CL_DROP=1: launch without the last 8 workgroups (the last member of the last 8 clusters never runs): exercises the
    // bounded waits' give-up path -- expect "(SPIN LIMIT HIT)" after ~1-2 s per launch, not a hang
    const int grid = (a.spread ? a.nclusters * a.G : (a.nclusters + 7) / 8 * 8 * a.G) - (getenv("CL_DROP") ? 8 : 0);
    unsigned long long* dbg; hipMalloc(&dbg, (size_t)grid * 16 * 8); hipMemset(dbg, 0, (size_t)grid * 16 * 8);
    a.dbg = dbg;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(sync, 0, 1 << 20);
        hipMemset(xb0, 0, state * 16); hipMemset(xb1, 0, state * 16);  // LL exchange: no stale tags
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        switch (ld / 128) {
            case 3: hipLaunchKernelGGL((cluster_kernel<MODE, false, 3, false>), dim3(grid), dim3(CL_THREADS), 0, 0, a); break;
            case 4: hipLaunchKernelGGL((cluster_kernel<MODE, false, 4, false>), dim3(grid), dim3(CL_THREADS), 0, 0, a); break;
            case 5: hipLaunchKernelGGL((cluster_kernel<MODE, false, 5, false>), dim3(grid), dim3(CL_THREADS), 0, 0, a); break;
            default: hipLaunchKernelGGL((cluster_kernel<MODE, false, 6, false>), dim3(grid), dim3(CL_THREADS), 0, 0, a); break;
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned st; hipMemcpy(&st, sync, 4, hipMemcpyDeviceToHost);
        if (rep == 2 && getenv("CL_DROP")) printf("   (launch with 8 workgroups missing: %.1f ms end to end)\n", ms);
        if (rep == 2) printf("%s ABL=%2d N=%d B=%d grid %d: %.3f us/step%s\n", MODE == MODE_DL ? "DL" : MODE == MODE_MF ? "MF" : "LV", CCVM_CLUSTER_ABL, N, B, grid, ms * 1e3 / steps, st ? "  (SPIN LIMIT HIT)" : "");
    }
    if (CCVM_CLUSTER_ABL & 64) {
        std::vector<unsigned long long> hd((size_t)grid * 16);
        hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost);
        const char* names[16] = {"MFMA wave: wait at B_0", "MFMA wave: first read + chunks", "MFMA wave: noise + update", "MFMA wave: wait at inner barriers",
                                 "MFMA wave:   at B_1", "MFMA wave:   at B_2", "MFMA wave:   at B_3..", "MFMA wave: publish stores",
                                 "fetch wave: wait at B_0", "fetch wave: staging, loads, inner barriers", "MFMA wave 1: wait at B_0",
                                 "MFMA wave 2: wait at B_0", "fetch wave: tag check + stage next", "fetch wave: RETRY ROUNDS per phase",
                                 "MFMA wave 3: wait at B_0", "MFMA wave 3: noise + update"};
        for (int k = 0; k < 16; ++k) {
            if (!names[k][0]) continue;
            std::vector<double> v;
            for (int w = 0; w < grid; ++w) v.push_back((double)hd[(size_t)w * 16 + k] / ((ld > 512 ? 3.0 : 2.0) * steps));
            std::sort(v.begin(), v.end());
            printf("%-38s: min %8.1f  median %8.1f  max %8.1f ticks/phase\n", names[k], v.front(), v[v.size() / 2], v.back());
        }
    }
    return 0;
}
