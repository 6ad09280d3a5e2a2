// Does an f32 MFMA stream share its datapath with ordinary VALU work on gfx950?
// Workgroup = 512 threads: waves 0-3 run 1024 v_mfma_f32_32x32x2_f32 each, waves 4-7 run NV
// dependent-free VALU ops of a chosen kind.  time ~ max(...) => separate units; ~ sum => shared.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int nm, int nv, float x) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (KIND >= 10) {  // bf16 MFMA instead of f32
            f32x16 a0, a1;
            for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
            bf16x8 u, v;
            for (int i = 0; i < 8; ++i) { u[i] = (short)(threadIdx.x + i); v[i] = (short)(threadIdx.x * 3 + i); }
            for (int i = 0; i < nm; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u, v, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, u, a1, 0, 0, 0);
            }
            float s = 0.f;
            for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
            if (s == 123.456f) out[0] = s;
        } else {
            f32x16 a0, a1;
            for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
            float u = x + threadIdx.x, v = x - threadIdx.x;
            for (int i = 0; i < nm; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(u, v, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, u, a1, 0, 0, 0);
            }
            float s = 0.f;
            for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
            if (s == 123.456f) out[0] = s;
        }
    } else {
        constexpr int KK = KIND % 10;
        float a = x + threadIdx.x, b = x * 0.5f, c = 1.0f, d = 2.0f, e = 3.f, f = 4.f, g = 5.f, h = 6.f;
        unsigned ia = threadIdx.x, ib = 77, ic = 3, id = 9;
        for (int i = 0; i < nv; i += 8) {
            if (KK == 1) {  // 8 independent f32 FMAs
                a = __builtin_fmaf(a, b, c); c = __builtin_fmaf(c, b, d); d = __builtin_fmaf(d, b, e); e = __builtin_fmaf(e, b, f);
                f = __builtin_fmaf(f, b, g); g = __builtin_fmaf(g, b, h); h = __builtin_fmaf(h, b, a); b = __builtin_fmaf(b, 0.999f, 1e-3f);
            } else if (KK == 2) {  // 8 integer add/xor/rotate (Threefry-like)
                ia += ib; ib = __builtin_amdgcn_alignbit(ib, ib, 19) ^ ia; ic += id; id = __builtin_amdgcn_alignbit(id, id, 17) ^ ic;
                ia += ic; ic = __builtin_amdgcn_alignbit(ic, ic, 6) ^ ia; ib += id; id = __builtin_amdgcn_alignbit(id, id, 3) ^ ib;
            }
        }
        if (a + c + d + e + f + g + h + b == 123.456f || (ia ^ ib ^ ic ^ id) == 0x12345u) out[1] = a;
    }
}

template <typename F> float time_us(F&& f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 50.f;
}
int main() {
    float* out; hipMalloc(&out, 64);
    const int nm = 512;
    for (int kind : {1, 2}) {
        printf("== f32 MFMA (1024/wave) + %s VALU in a sibling wave\n", kind == 1 ? "f32 FMA" : "int add/rot/xor");
        for (int nv : {0, 2048, 4096, 8192, 16384}) {
            float t = kind == 1 ? time_us([&] { hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, nm, nv, 1.0f); })
                                : time_us([&] { hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, nm, nv, 1.0f); });
            float tv = kind == 1 ? time_us([&] { hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, 0, nv, 1.0f); })
                                 : time_us([&] { hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, 0, nv, 1.0f); });
            printf("  nv %6d: both %7.2f us   valu alone %7.2f us\n", nv, t, tv);
        }
    }
    printf("== bf16 MFMA 32x32x16 (1024/wave) + f32 FMA VALU in a sibling wave\n");
    for (int nv : {0, 4096, 16384}) {
        float t = time_us([&] { hipLaunchKernelGGL(k<11>, dim3(256), dim3(512), 0, 0, out, nm, nv, 1.0f); });
        printf("  nv %6d: both %7.2f us\n", nv, t);
    }
    return 0;
}
