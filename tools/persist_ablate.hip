// Ablation harness for the persistent small-N kernel (developer tool): DL, N=100, B=1000.
//   for b in 0 1 2 4 8 ...; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCCVM_PERSIST_ABL=$b tools/persist_ablate.hip -o tools/persist_ablate_$b; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../ccvm_amd/csrc/ccvm_schedule.h"
using namespace ccvm;
int main(int argc, char** argv) {
    const int N = 100, B = 1000, ld = 128, rows = 1024, steps = 4096;
    float *Q, *V, *c, *s, *table;
    hipMalloc(&Q, ld * ld * 4); hipMalloc(&V, ld * 4); hipMalloc(&c, rows * ld * 4); hipMalloc(&s, rows * ld * 4);
    hipMalloc(&table, steps * TABLE_WORDS * 4);
    std::vector<float> h(ld * ld, 0.f);
    unsigned rng = 1;
    auto rnd = [&] { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) * (1.0f / 16777216.0f) - 0.5f); };
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) h[i * ld + j] = rnd() * 0.02f;
    hipMemcpy(Q, h.data(), ld * ld * 4, hipMemcpyHostToDevice);
    hipMemcpy(V, h.data(), ld * 4, hipMemcpyHostToDevice);
    hipMemset(c, 0, rows * ld * 4); hipMemset(s, 0, rows * ld * 4);
    DlSched sc{8.0, 0.001, 10.0, 100.0, 0.05, 1.0, 2.6457513, 1, 15000, 0, steps};
    hipLaunchKernelGGL(dl_schedule_kernel, dim3((steps + 255) / 256), dim3(256), 0, 0, sc, table);
    PersistArgs a; memset(&a, 0, sizeof(a));
    a.Q = Q; a.V = V; a.qsum = V; a.x0 = c; a.x1 = s; a.table = table; a.seed = 7; a.nsteps = steps;
    a.B = B; a.N = N; a.ld = ld; a.in_scale = 0.378f; a.in_shift = 1.0f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ru : {4, 2}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (ru == 4) hipLaunchKernelGGL((persist_kernel<MODE_DL, false, 64, 2, 7, 4>), dim3(B / 4), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL((persist_kernel<MODE_DL, false, 64, 2, 7, 2>), dim3(B / 2), dim3(256), 0, 0, a);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("ABL=%2d RU=%d: %.3f us/step\n", CCVM_PERSIST_ABL, ru, ms * 1e3 / steps);
        }
    }
    return 0;
}
