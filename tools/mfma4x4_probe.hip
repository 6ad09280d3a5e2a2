// Probe of the v_mfma_f32_4x4x1_16B_f32 operand / result lane mapping and of the A-broadcast
// controls CBSZ / ABID (developer tool).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void probe(float* out, int which) {
    const int lane = threadIdx.x;
    const float a = which == 0 ? (float)(lane + 1) : 1.0f;
    const float b = which == 1 ? (float)(lane + 1) : 1.0f;
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, CBSZ, ABID, 0);
    for (int r = 0; r < 4; ++r) out[r * 64 + lane] = acc[r];
}
template <int CBSZ, int ABID>
void show(float* d, int which) {
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d, which);
    float h[4 * 64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("cbsz=%d abid=%d: %s supplier lane (+1) of D[reg r][lane]:\n", CBSZ, ABID, which == 0 ? "A" : "B");
    for (int r = 0; r < 4; ++r) {
        printf(" r=%d:", r);
        for (int l = 0; l < 64; ++l) printf(" %2.0f", h[r * 64 + l]);
        printf("\n");
    }
}
int main() {
    float* d;
    hipMalloc(&d, 4 * 64 * 4);
    show<0, 0>(d, 0);
    show<0, 0>(d, 1);
    show<4, 5>(d, 0);
    show<3, 2>(d, 0);
    show<2, 1>(d, 0);
    show<4, 5>(d, 1);
    return 0;
}
