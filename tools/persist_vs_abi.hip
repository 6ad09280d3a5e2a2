// The row-owner kernel launched directly (as tools/persist_ablate.hip does) against the SAME shape through the C ABI
// (ccvm_dl_run of libccvm_hip.so), on the same data, in one process (developer tool, round 6: the harness measured
// 0.27-0.30 us per step at DL N = 20 where bench.py measured 0.43 -- which of the two is the kernel's time?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -mllvm -amdgpu-mfma-vgpr-form -Iinclude tools/persist_vs_abi.hip \
//       -Lccvm_amd -lccvm_hip -Wl,-rpath,$PWD/ccvm_amd -o tools/persist_vs_abi
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../ccvm_amd/csrc/ccvm_schedule.h"
#include "ccvm_hip.h"
using namespace ccvm;

int main(int argc, char** argv) {
    const int N = 20, B = argc > 1 ? atoi(argv[1]) : 1000, ld = 128, rows = (B + 63) / 64 * 64, steps = 4096, T = 15000;
    float *Q, *V, *c, *s, *table, *qsum;
    hipMalloc(&Q, ld * ld * 4); hipMalloc(&V, ld * 4); hipMalloc(&qsum, ld * 4);
    hipMalloc(&c, (size_t)rows * ld * 4); hipMalloc(&s, (size_t)rows * ld * 4);
    hipMalloc(&table, steps * TABLE_WORDS * 4);
    const size_t wsb = ccvm_workspace_bytes(0, B, N);
    void* ws; hipMalloc(&ws, wsb); hipMemset(ws, 0, wsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int data = 0; data < 3; ++data) {
        // 0: the ablation harness's matrix (uniform +-0.01); 1: the scale the solvers' scaling produces; 2: Q = 0
        std::vector<float> h(ld * ld, 0.f), hv(ld, 0.f), hs(ld, 0.f);
        unsigned rng = 1;
        auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f); };
        const float scale = data == 0 ? 0.02f : data == 1 ? 0.2f / std::sqrt((float)N) * 5.0f : 0.0f;
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) h[i * ld + j] = rnd() * scale;
        for (int j = 0; j < N; ++j) hv[j] = rnd() * (data == 2 ? 0.0f : 0.3f);
        for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) hs[j] += h[i * ld + j];
        hipMemcpy(Q, h.data(), ld * ld * 4, hipMemcpyHostToDevice);
        hipMemcpy(V, hv.data(), ld * 4, hipMemcpyHostToDevice);
        hipMemcpy(qsum, hs.data(), ld * 4, hipMemcpyHostToDevice);
        for (int path = 0; path < 2; ++path) {
            hipMemset(c, 0, (size_t)rows * ld * 4); hipMemset(s, 0, (size_t)rows * ld * 4);
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                if (path == 0) {
                    DlSched sc{8.0, 0.001, 10.0, 100.0, 0.05, 1.0, 2.6457513, 1, T, 0, steps};
                    hipLaunchKernelGGL(dl_schedule_kernel, dim3((steps + 255) / 256), dim3(256), 0, 0, sc, table);
                    PersistArgs a; memset(&a, 0, sizeof(a));
                    a.Q = Q; a.V = V; a.qsum = qsum; a.x0 = c; a.x1 = s; a.table = table; a.seed = 7; a.ld = ld;
                    a.in_scale = (float)(1.0 / 2.6457513); a.in_shift = 1.0f; a.N = N; a.B = B; a.nsteps = steps; a.simds = 1024;
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL((persist_kernel<MODE_DL, false, 32, 1, 2, 2, 1, 1>), dim3((B + 3) / 4), dim3(256), 0, 0, a);
                    hipEventRecord(e1, 0);
                } else {
                    ccvm_dl_params p; memset(&p, 0, sizeof(p));
                    p.pump = 8.0; p.dt = 0.001; p.noise_ratio = 10.0; p.feedback_scale = 100.0; p.g = 0.05; p.lower = 0.0; p.upper = 1.0;
                    p.pump_rate_flag = 1; p.qsum = qsum;
                    ccvm_noise nz; memset(&nz, 0, sizeof(nz)); nz.mode = CCVM_NOISE_PHILOX; nz.seed = 7;
                    hipEventRecord(e0, 0);
                    if (ccvm_dl_run(Q, V, c, s, B, N, ld, 0, steps, T, &p, &nz, ws, wsb, nullptr)) { printf("ABI: %s\n", ccvm_last_error()); return 1; }
                    hipEventRecord(e1, 0);
                }
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) best = std::min(best, ms);
            }
            std::vector<float> hc((size_t)rows * ld);
            hipMemcpy(hc.data(), c, hc.size() * 4, hipMemcpyDeviceToHost);
            double sum = 0.0, mx = 0.0; int bad = 0;
            for (int r = 0; r < B; ++r) for (int j = 0; j < N; ++j) { const float x = hc[(size_t)r * ld + j]; if (!std::isfinite(x)) ++bad; else { sum += x; mx = std::max(mx, (double)std::fabs(x)); } }
            char what[256] = "";
            if (path) ccvm_describe_launch(0, B, N, 0, 0, what, sizeof(what));
            printf("data %d %-14s %.3f us/step   state: mean %.4f max|c| %.3f non-finite %d  %s\n", data, path ? "ccvm_dl_run" : "direct launch", best * 1e3 / steps,
                   sum / ((double)B * N), mx, bad, what);
        }
    }
    return 0;
}
