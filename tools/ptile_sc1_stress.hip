// Stress of the persistent tile kernel's hand-off form (ccvm_ptile.h), VERDICT r5 item 3.3 -- the guide's recipe for a
// form its table does not list ("test every hand-off under UNEVEN load, consumer L1-warm, checking every word"):
//
//   producer workgroup: 4 KB payload with `global_store_dwordx4 ... sc1`, every storing wave `s_waitcnt vmcnt(0)`, the
//     workgroup's barrier, ONE lane's `global_store_dword ... sc1` of the epoch into its flag line;
//   consumer workgroup: ONE lane polls the flag with `global_load_dword ... sc1`, a barrier, then EVERY load of the payload
//     is `global_load_lds_dwordx4 ... sc1` (LDS-DMA: MI355X_MICROARCH.md's table of measured forms lists register loads
//     only), `s_waitcnt vmcnt(0)`, barrier, every word compared.
//
// Conditions: 256 workgroups (one per CU), each producer AND consumer (workgroup b reads b + shift: shift 8 keeps a pair in
// one XCD, shift 1 puts it across the fabric -- the XCD of every workgroup is read from HW_REG_XCC_ID and reported, not
// assumed); between hand-offs every consumer thread RE-READS both 4 KB slots of its producer with PLAIN loads, so the
// lines about to be overwritten remotely sit in this CU's L1 (the "L1-warm consumer"); a second stream runs a bandwidth
// hog on a quarter of the chip (uneven load).  Control: the same with PLAIN LDS-DMA -- it must show stale words, or the
// test would prove nothing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ptile_sc1_stress.hip -o tools/ptile_sc1_stress && tools/ptile_sc1_stress [epochs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int WORDS = 1024;        // 4 KB payload
constexpr int THREADS = 256;       // 4 waves: one LDS-DMA instruction of 1 KB each
constexpr unsigned SPIN = 1u << 22;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ unsigned word_of(unsigned b, unsigned e, unsigned i) {
    unsigned x = b * 0x9E3779B1u ^ e * 0x85EBCA77u ^ i * 0xC2B2AE3Du;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x | 1u;  // never 0 (the buffers start zeroed)
}

struct Args {
    unsigned* payload;   // [grid][2][WORDS]
    unsigned* flag;      // [grid][32]  (a 128-byte line each)
    unsigned* ack;       // [grid][32]
    unsigned long long* bad;   // [grid][4]: mismatching words, of which stale (the slot's previous epoch), give-ups, hand-offs
    unsigned* xcc;       // [grid]
    int shift, epochs;
};

template <bool SC1, bool WARM>
__global__ __launch_bounds__(THREADS) void handoff(const Args a) {
    __shared__ __attribute__((aligned(16))) unsigned lds[WORDS + 4];
    const unsigned b = blockIdx.x, g = gridDim.x, tid = threadIdx.x, lane = tid & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned src = (b + a.shift) % g;          // the workgroup whose payload this one consumes
    const unsigned reader = (b + g - a.shift) % g;   // ... and the one that consumes this workgroup's
    if (tid == 0) a.xcc[b] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;  // HW_REG_XCC_ID[3:0]
    unsigned long long n_bad = 0, n_stale = 0, n_gave_up = 0, n_done = 0;
    unsigned sink = 0;
    bool dead = false;
    if (tid == 0) lds[WORDS] = 0u;  // set by a lane that gave up a bounded wait; read by everyone behind the next barrier
    __syncthreads();
    for (int e = 1; e <= a.epochs && !dead; ++e) {
        // ---- producer role: slot e & 1 is free once the reader has checked epoch e - 2 ----------------------------------
        if (tid == 0 && e > 2) {
            unsigned spins = 0, v = 0;
            const unsigned* p = a.ack + (size_t)b * 32;
            do {
                asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            } while ((int)(v - (unsigned)(e - 2)) < 0 && ++spins < SPIN);
            if (spins >= SPIN) lds[WORDS] = 1u;
        }
        __syncthreads();
        if (lds[WORDS]) { dead = true; ++n_gave_up; break; }
        {
            unsigned* dst = a.payload + ((size_t)b * 2 + (e & 1)) * WORDS + 4 * tid;
            const unsigned w0 = word_of(b, e, 4 * tid), w1 = word_of(b, e, 4 * tid + 1), w2 = word_of(b, e, 4 * tid + 2),
                           w3 = word_of(b, e, 4 * tid + 3);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 val = {w0, w1, w2, w3};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(val) : "memory");
        }
        __syncthreads();  // every storing wave has drained its stores
        if (tid == 0) {
            unsigned* fp = a.flag + (size_t)b * 32;
            const unsigned ev = (unsigned)e;
            asm volatile("global_store_dword %0, %1, off sc1" ::"v"(fp), "v"(ev) : "memory");
        }
        // ---- consumer role --------------------------------------------------------------------------------------------
        if (tid == 0) {
            unsigned spins = 0, v = 0;
            const unsigned* p = a.flag + (size_t)src * 32;
            do {
                asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
            } while ((int)(v - (unsigned)e) < 0 && ++spins < SPIN);
            if (spins >= SPIN) lds[WORDS] = 1u;
        }
        __syncthreads();
        if (lds[WORDS]) { dead = true; ++n_gave_up; break; }
        {
            // one LDS-DMA instruction per wave: lanes x 16 B = 1 KB of the producer's slot into LDS (ccvm_ptile.h: dma)
            const char* base = reinterpret_cast<const char*>(a.payload + ((size_t)src * 2 + (e & 1)) * WORDS + 256 * wave);
            const unsigned voff = 16 * lane;
            const unsigned ldst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void*)(lds + 256 * wave));
            unsigned keep;
            if constexpr (SC1) {
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                             : "=&s"(keep) : "v"(voff), "s"(base), "s"(ldst) : "memory");
            } else {
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                             : "=&s"(keep) : "v"(voff), "s"(base), "s"(ldst) : "memory");
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned i = 4 * tid + k, got = lds[i];
            if (got != word_of(src, e, i)) {
                ++n_bad;
                if (got == (e > 2 ? word_of(src, e - 2, i) : 0u)) ++n_stale;
            }
        }
        ++n_done;
        __syncthreads();  // everyone has read its words: the LDS copy may be overwritten, the slot re-used
        if (tid == 0) {
            unsigned* ap = a.ack + (size_t)src * 32;
            const unsigned ev = (unsigned)e;
            asm volatile("global_store_dword %0, %1, off sc1" ::"v"(ap), "v"(ev) : "memory");
        }
        if constexpr (WARM) {
            // the L1-warm consumer: plain re-reads of BOTH slots of the producer (the next epoch overwrites one of them)
            const unsigned* p = a.payload + (size_t)src * 2 * WORDS;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 2 * WORDS / THREADS; ++k) {
                    unsigned v;  // PLAIN vector loads (no sc bits: they allocate in this CU's L1)
                    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p + k * THREADS + tid) : "memory");
                    sink += v;
                }
        }
    }
    if (sink == 0x12345u) a.flag[0] = sink;  // (keeps the re-reads)
    atomicAdd(&a.bad[(size_t)b * 4 + 0], n_bad);
    atomicAdd(&a.bad[(size_t)b * 4 + 1], n_stale);
    if (tid == 0) {
        atomicAdd(&a.bad[(size_t)b * 4 + 2], n_gave_up);
        atomicAdd(&a.bad[(size_t)b * 4 + 3], n_done);
    }
}

__global__ void hog(const float4* src, float* out, size_t n, int iters) {
    float acc = 0.f;
    for (int it = 0; it < iters; ++it)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const float4 v = src[i];
            acc += v.x + v.y + v.z + v.w;
        }
    if (acc == 1.2345f) out[0] = acc;
}

template <bool SC1, bool WARM>
int run(const char* what, int shift, int epochs, bool loaded) {
    const int grid = 256;
    Args a;
    hipMalloc(&a.payload, (size_t)grid * 2 * WORDS * 4); hipMemset(a.payload, 0, (size_t)grid * 2 * WORDS * 4);
    hipMalloc(&a.flag, (size_t)grid * 128); hipMemset(a.flag, 0, (size_t)grid * 128);
    hipMalloc(&a.ack, (size_t)grid * 128); hipMemset(a.ack, 0, (size_t)grid * 128);
    hipMalloc(&a.bad, (size_t)grid * 32); hipMemset(a.bad, 0, (size_t)grid * 32);
    hipMalloc(&a.xcc, grid * 4); hipMemset(a.xcc, 0xFF, grid * 4);
    a.shift = shift; a.epochs = epochs;
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    float4* big = nullptr; float* out = nullptr;
    const size_t n = (size_t)64 << 20;  // 1 GiB of float4 = 64 Mi elements
    if (loaded) {
        hipMalloc(&big, n * 16); hipMalloc(&out, 4); hipMemsetAsync(big, 0, n * 16, s2); hipStreamSynchronize(s2);
        hipLaunchKernelGGL(hog, dim3(64), dim3(256), 0, s2, big, out, n, 8);  // a quarter of the CUs stream 8 GiB
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s1);
    hipLaunchKernelGGL((handoff<SC1, WARM>), dim3(grid), dim3(THREADS), 0, s1, a);
    hipEventRecord(e1, s1);
    hipStreamSynchronize(s1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipStreamSynchronize(s2);
    std::vector<unsigned long long> bad(grid * 4);
    std::vector<unsigned> xcc(grid);
    hipMemcpy(bad.data(), a.bad, grid * 32, hipMemcpyDeviceToHost);
    hipMemcpy(xcc.data(), a.xcc, grid * 4, hipMemcpyDeviceToHost);
    unsigned long long nb = 0, ns = 0, ng = 0, nd = 0;
    int same = 0, cross = 0;
    for (int b = 0; b < grid; ++b) {
        nb += bad[4 * b]; ns += bad[4 * b + 1]; ng += bad[4 * b + 2]; nd += bad[4 * b + 3];
        (xcc[b] == xcc[(b + shift) % grid] ? same : cross)++;
    }
    printf("%-44s shift %d (%3d pairs inside an XCD, %3d across)%s: %llu hand-offs of 4 KB, %llu words checked, %llu wrong (%llu = the slot's "
           "previous epoch), %llu give-ups, %.1f ms (%.2f us per epoch)\n", what, shift, same, cross, loaded ? ", hog on a 2nd stream" : "",
           nd, nd * WORDS, nb, ns, ng, ms, ms * 1e3 / epochs);
    hipFree(a.payload); hipFree(a.flag); hipFree(a.ack); hipFree(a.bad); hipFree(a.xcc);
    if (big) { hipFree(big); hipFree(out); }
    hipStreamDestroy(s1); hipStreamDestroy(s2);
    return (int)(SC1 ? (nb + ng) != 0 : ng != 0);  // the product's form must be clean; the control only must not hang
}

int main(int argc, char** argv) {
    const int epochs = argc > 1 ? atoi(argv[1]) : 4096;
    int rc = 0;
    for (int shift : {8, 1}) {
        rc |= run<true, true>("sc1 LDS-DMA, L1-warm consumers", shift, epochs, false);
        rc |= run<true, true>("sc1 LDS-DMA, L1-warm consumers", shift, epochs, true);
        rc |= run<true, false>("sc1 LDS-DMA, cold consumers", shift, epochs, true);
        rc |= run<false, true>("CONTROL: plain LDS-DMA, L1-warm consumers", shift, epochs / 4, true);
    }
    printf(rc ? "FAILED\n" : "PASSED: no stale word behind a drained sc1 store + sc1 flag + sc1 LDS-DMA\n");
    return rc;
}
