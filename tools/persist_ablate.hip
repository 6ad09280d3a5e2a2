// Ablation harness for the persistent row-owner kernel (developer tool): DL at N = 100 and N = 20, every variant the
// launch policy can pick (rows in use, K split, noise producer waves), B from the command line.
//   for b in 0 1 2 4 8 16; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DCCVM_PERSIST_ABL=$b \
//       tools/persist_ablate.hip -o tools/persist_ablate_$b; done
//   tools/persist_ablate_0 [B]        (bit 16: s_memtime stamps of the step's segments, cycles per step)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../ccvm_amd/csrc/ccvm_schedule.h"
using namespace ccvm;

static const char* SEG[16] = {"consumer: A (+ noise slot) reads land", "consumer: noise made in front of the MFMAs", "consumer: MFMAs + partial sums",
                              "consumer: noise made behind the MFMAs", "consumer: K-split barrier", "consumer: twin's sums + update",
                              "consumer: publish (LDS writes land)", "consumer: the step's last barrier",
                              "producer: generator + slot write", "producer: waiting at the barriers", "", "", "", "", "", ""};

template <int CW, int NCG, int NCH, int RU, int KH, int PW>
void run(const char* what, PersistArgs a, int N, int B, int steps) {
    constexpr int wps = NCG * KH * (1 + PW), sets = wps > 4 ? 1 : 4 / wps, threads = wps > 4 ? 64 * wps : 256;
    constexpr int per = 2 * (64 / CW) * RU / 4 * sets;  // DL batch rows per workgroup
    const int grid = (B + per - 1) / per;
    a.N = N; a.B = B; a.nsteps = steps;
    unsigned long long* dbg = nullptr;
    hipMalloc(&dbg, (size_t)grid * 16 * 8);
    hipMemset(dbg, 0, (size_t)grid * 16 * 8);
    a.dbg = dbg;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((persist_kernel<MODE_DL, false, CW, NCG, NCH, RU, KH, PW>), dim3(grid), dim3(threads), 0, 0, a);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = std::min(best, ms);
    }
    printf("ABL=%2d DL N=%3d B=%4d %-34s <%d,%d,%d,RU=%d,KH=%d,PW=%d> grid %4d x %3d: %.3f us/step\n", CCVM_PERSIST_ABL, N, B, what,
           CW, NCG, NCH, RU, KH, PW, grid, threads, best * 1e3 / steps);
    if (CCVM_PERSIST_ABL & 16) {
        std::vector<unsigned long long> h((size_t)grid * 16);
        hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        for (int k = 0; k < 16; ++k) {
            if (!SEG[k][0]) continue;
            std::vector<double> v;
            for (int w = 0; w < grid; ++w) v.push_back((double)h[(size_t)w * 16 + k] / steps);
            std::sort(v.begin(), v.end());
            if (v.back() == 0.0) continue;
            printf("      %-46s min %7.1f  median %7.1f  max %7.1f  (s_memtime ticks per step)\n", SEG[k], v.front(), v[v.size() / 2], v.back());
        }
    }
    {   // the state the variant left behind (a diverged state changes what the step costs?)
        std::vector<float> hc(64 * 128);
        hipMemcpy(hc.data(), a.x0, hc.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0; double mx = 0.0;
        for (int r = 0; r < 64; ++r) for (int j = 0; j < N; ++j) { const float x = hc[r * 128 + j]; if (!(x == x) || x > 1e30f || x < -1e30f) ++bad; else mx = std::max(mx, (double)(x < 0 ? -x : x)); }
        printf("      state after: %d non-finite of %d, max |c| %.3g\n", bad, 64 * N, mx);
    }
    hipFree(dbg);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1000;
    const bool small_first = argc > 2 && atoi(argv[2]) == 1;   // N = 20 before N = 100 (state arrays still zero)
    const bool fresh = argc > 2 && atoi(argv[2]) == 2;         // zero the state arrays in front of every variant
    const int ld = 128, rows = (B + 63) / 64 * 64 + 64, steps = 4096;
    float *Q, *V, *c, *s, *table;
    hipMalloc(&Q, ld * ld * 4); hipMalloc(&V, ld * 4); hipMalloc(&c, (size_t)rows * ld * 4); hipMalloc(&s, (size_t)rows * ld * 4);
    hipMalloc(&table, steps * TABLE_WORDS * 4);
    DlSched sc{8.0, 0.001, 10.0, 100.0, 0.05, 1.0, 2.6457513, 1, 15000, 0, steps};
    hipLaunchKernelGGL(dl_schedule_kernel, dim3((steps + 255) / 256), dim3(256), 0, 0, sc, table);
    for (int pass = 0; pass < 2; ++pass) {
        const int N = (pass == 0) != small_first ? 100 : 20;
        std::vector<float> h(ld * ld, 0.f);
        unsigned rng = 1;
        auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f); };
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) h[i * ld + j] = rnd() * 0.02f;
        hipMemcpy(Q, h.data(), ld * ld * 4, hipMemcpyHostToDevice);
        hipMemcpy(V, h.data(), ld * 4, hipMemcpyHostToDevice);
        hipMemset(c, 0, (size_t)rows * ld * 4); hipMemset(s, 0, (size_t)rows * ld * 4);
        PersistArgs a; memset(&a, 0, sizeof(a));
        a.Q = Q; a.V = V; a.qsum = V; a.x0 = c; a.x1 = s; a.table = table; a.seed = 7; a.ld = ld; a.in_scale = 0.378f; a.in_shift = 1.0f;
        auto reset = [&] { if (fresh) { hipMemset(c, 0, (size_t)rows * ld * 4); hipMemset(s, 0, (size_t)rows * ld * 4); } };
        if (N == 100) {
            run<64, 2, 7, 4, 2, 0>("K split (shipped at B = 1000)", a, N, B, steps);
            run<64, 2, 7, 4, 2, 1>("K split + producers", a, N, B, steps);
            run<64, 2, 7, 4, 1, 0>("whole chains, 4 rows", a, N, B, steps);
            run<64, 2, 7, 4, 1, 1>("whole chains, 4 rows + producers", a, N, B, steps);
            run<64, 2, 7, 2, 1, 0>("whole chains, 2 rows", a, N, B, steps);
            run<64, 2, 7, 2, 1, 1>("whole chains, 2 rows + producers", a, N, B, steps);
        } else {
            reset(); run<32, 1, 2, 2, 1, 0>("one wave per row set, 2 rows", a, N, B, steps);
            reset(); run<32, 1, 2, 2, 1, 1>("2 rows + producers", a, N, B, steps);
            reset(); run<32, 1, 2, 4, 1, 0>("one wave per row set, 4 rows", a, N, B, steps);
            reset(); run<32, 1, 2, 4, 1, 1>("4 rows + producers", a, N, B, steps);
        }
    }
    return 0;
}
