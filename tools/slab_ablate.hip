// Ablation / stamp harness for the column-slab small-batch kernel (developer tool): Langevin (or DL: -DCCVM_ABL_DL).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DCCVM_SLAB_ABL=64 tools/slab_ablate.hip -o tools/slab_ablate_64
//   tools/slab_ablate_64 N B CGRP [RG]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../ccvm_amd/csrc/ccvm_slab.h"
#include "../ccvm_amd/csrc/ccvm_schedule.h"
using namespace ccvm;
__global__ void init_xb(uint2* xb, size_t packets, int K, int Kx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < packets; i += (size_t)gridDim.x * blockDim.x)
        xb[i] = make_uint2(0u, (int)((i >> 2) % (size_t)K) >= Kx ? 0xFFFFFFFFu : 0u);
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1000, B = argc > 2 ? atoi(argv[2]) : 4;
    const int cgrp = argc > 3 ? atoi(argv[3]) : 0, rgf = argc > 4 ? atoi(argv[4]) : 0;
    const int ld = (N + 127) / 128 * 128, rows = (B + 63) / 64 * 64, steps = 4096;
    const size_t state = (size_t)rows * ld;
#ifdef CCVM_ABL_DL
    constexpr int MODE = MODE_DL;
    const int planes = 2;
#else
    constexpr int MODE = MODE_LANGEVIN;
    const int planes = 1;
#endif
    const SlabPlan p = slab_plan(B, N, planes, ChipGeometry{256, 8}, cgrp, rgf);
    if (!p.ok) { printf("no plan\n"); return 1; }
    const size_t half = (size_t)p.nclusters * planes * p.rg * p.K * 32;
    float *Q, *V, *c, *c2, *xb, *table; unsigned* sync;
    hipMalloc(&Q, (size_t)ld * ld * 4); hipMalloc(&V, ld * 4); hipMalloc(&c, state * 4); hipMalloc(&c2, state * 4);
    hipMemset(c2, 0, state * 4); hipMalloc(&xb, 2 * half); hipMalloc(&table, steps * TABLE_WORDS * 4); hipMalloc(&sync, 1 << 20);
    std::vector<float> h((size_t)ld * ld, 0.f);
    unsigned rng = 1;
    auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f); };
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) h[(size_t)i * ld + j] = rnd() * 0.02f;
    hipMemcpy(Q, h.data(), (size_t)ld * ld * 4, hipMemcpyHostToDevice);
    hipMemcpy(V, h.data(), ld * 4, hipMemcpyHostToDevice);
    hipMemset(c, 0, state * 4);
#ifdef CCVM_ABL_DL
    DlSched sc{2.5, 0.002, 10.0, 1.0, 0.05, 1.0, 1.2247, 1, 15000, 0, steps};
    hipLaunchKernelGGL(dl_schedule_kernel, dim3((steps + 255) / 256), dim3(256), 0, 0, sc, table);
#else
    LvSched sc{0.002, 0.5, 1.0, 0.5, 2.0, 1.0, 1, 1, 15000, 0, steps, AdamSched{}};
    hipLaunchKernelGGL(lv_schedule_kernel, dim3((steps + 255) / 256), dim3(256), 0, 0, sc, table);
#endif
    SlabArgs a; memset(&a, 0, sizeof(a));
    a.Q = Q; a.V = V; a.qsum = V; a.x0 = c; a.x1 = c2; a.xb0 = xb; a.xb1 = (float*)((char*)xb + half); a.table = table;
    a.seed = 7; a.nsteps = steps; a.status = sync; a.spin_limit = 200000000u;
    a.B = B; a.N = N; a.ld = ld; a.in_scale = 1.0f; a.in_shift = 0.5f;
    a.nclusters = p.nclusters; a.G = p.G; a.RG = p.rg; a.span = p.span; a.nxcd = 8; a.delay_fabric = getenv("SL_DELAY") ? atoi(getenv("SL_DELAY")) : slab_fabric_delay(planes, p.rg, p.K);
    SlabPlan q = p;
    const int grid = q.grid;
    unsigned long long* dbg; hipMalloc(&dbg, (size_t)grid * 16 * 8); hipMemset(dbg, 0, (size_t)grid * 16 * 8);
    a.dbg = dbg;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(sync, 0, 1 << 20);
        hipLaunchKernelGGL(init_xb, dim3(1024), dim3(256), 0, 0, (uint2*)xb, 2 * half / 8, p.K, p.G * 4 * p.cgrp);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        if (slab_calibrates(a)) launch_slab_cal<MODE, true>(a, q, 0); else launch_slab_cal<MODE, false>(a, q, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned st; hipMemcpy(&st, sync, 4, hipMemcpyDeviceToHost);
        if (rep == 2) printf("%s ABL=%2d SLEEP=%d N=%d B=%d: %d clusters x %d members x %d columns, %d rows, K=%d, %d XCD(s) per cluster, grid %d: %.3f us/step%s\n",
                             MODE == MODE_DL ? "DL" : "LV", CCVM_SLAB_ABL, CCVM_SL_SLEEP, N, B, p.nclusters, p.G, 4 * p.cgrp, 4 * p.rg, p.K,
                             q.span, grid, ms * 1e3 / steps, st ? "  (SPIN LIMIT HIT)" : "");
    }
    if (CCVM_SLAB_ABL & 8) {
        std::vector<unsigned long long> hd((size_t)grid * 16);
        hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost);
        for (int w = 0; w < 4; ++w) {
            std::vector<double> v;
            for (int g = 0; g < grid; ++g) v.push_back((double)hd[(size_t)g * 16 + 8 + w]);
            std::sort(v.begin(), v.end());
            printf("final delay of wave %d (x 64 cycles): min %.0f median %.0f max %.0f\n", w, v.front(), v[v.size() / 2], v.back());
        }
    }
    if (CCVM_SLAB_ABL & 64) {
        std::vector<unsigned long long> hd((size_t)grid * 16);
        hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost);
        const char* names[8] = {"issue + noise + schedule row", "wait for the first unit", "rest of the fetch + staging", "wait at B1",
                                "contraction + reductions", "wait at B2", "update + publish", "RETRY ROUNDS per step"};
        for (int k = 0; k < 8; ++k) {
            std::vector<double> v;
            for (int w = 0; w < grid; ++w) if (hd[(size_t)w * 16 + 4]) v.push_back((double)hd[(size_t)w * 16 + k] / steps);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("%-32s: min %8.1f  median %8.1f  max %8.1f ticks/step\n", names[k], v.front(), v[v.size() / 2], v.back());
        }
    }
    return 0;
}
