// Cost (and visibility check) of a per-step barrier among the 8 workgroups of a row block
// (developer tool): would a multi-step tile kernel with group barriers beat one launch per step
// (hand-over ~1.6 us + pipeline fill)?
//   hipcc --offload-arch=gfx950 -O3 tools/groupsync_probe.hip -o tools/groupsync_probe
// Variants: 0 agent-scope release/acquire fences (what the memory model prescribes),
//           1 atomics only, no fences (lower bound; visibility NOT guaranteed),
//           2 no fences, but the exchanged data is written and read with system-scope (sc0 sc1)
//             accesses that bypass the non-coherent cache levels.
// Each step every thread publishes a value derived from the step, and after the barrier reads its
// group neighbour's value of THIS step; stale reads are counted.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void store_sys(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float load_sys(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// 256 workgroups; blocks b and b+8 share an XCD; group = 8 consecutive local indices on one XCD
// (SAME_XCD) or 8 consecutive block ids = one workgroup on each XCD (the worst case).
template <int VARIANT, bool SAME_XCD>
__global__ __launch_bounds__(512) void k(unsigned* counters, float* data, int steps, int* failed, unsigned* stale) {
    int group, member;
    if (SAME_XCD) { const int x = blockIdx.x & 7, i = blockIdx.x >> 3; group = x * 4 + i / 8; member = i & 7; }
    else { group = blockIdx.x >> 3; member = blockIdx.x & 7; }
    unsigned* ctr = counters + group * 32;  // own cache line
    auto block_of = [&](int m) { return SAME_XCD ? ((group >> 2) + 8 * ((group & 3) * 8 + m)) : (group * 8 + m); };
    float* mine = data + (size_t)block_of(member) * 512 * 2;
    const float* theirs = data + (size_t)block_of((member + 1) & 7) * 512 * 2;
    unsigned bad = 0;
    for (int s = 0; s < steps; ++s) {
        const float value = (float)(s * 8 + 1);
        float* slot = mine + (s & 1) * 512 + threadIdx.x;  // ping-pong like the state buffers
        if (VARIANT == 2) store_sys(slot, value); else *slot = value;
        if (VARIANT == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (VARIANT == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the store has left the CU
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = 8u * (s + 1);
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > (1 << 22)) { *failed = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (VARIANT == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float* src = theirs + (s & 1) * 512 + threadIdx.x;
        const float got = (VARIANT == 2) ? load_sys(src) : *(const volatile float*)src;
        bad += (got != value);
    }
    if (bad) atomicAdd(stale, bad);
}

template <int VARIANT, bool SAME_XCD>
void run(const char* name, unsigned* c, float* d, int* f, unsigned* st) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int steps = 2000;
    float best = 1e9;
    unsigned hs = 0; int hf = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(c, 0, 32 * 32 * 4); hipMemset(f, 0, 4); hipMemset(st, 0, 4); hipMemset(d, 0, 256 * 512 * 2 * 4);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k<VARIANT, SAME_XCD>), dim3(256), dim3(512), 0, 0, c, d, steps, f, st);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        unsigned s1; int f1;
        hipMemcpy(&s1, st, 4, hipMemcpyDeviceToHost); hipMemcpy(&f1, f, 4, hipMemcpyDeviceToHost);
        hs += s1; hf |= f1;
    }
    printf("%-44s %s: %6.2f us per step, stale reads %u of %u%s\n", name, SAME_XCD ? "group on one XCD " : "group across XCDs",
           best * 1000 / steps, hs, 3u * 2000u * 256u * 512u, hf ? "  (SPIN LIMIT HIT)" : "");
}

int main() {
    unsigned* c; float* d; int* f; unsigned* st;
    hipMalloc(&c, 32 * 32 * 4); hipMalloc(&d, 256 * 512 * 2 * 4); hipMalloc(&f, 4); hipMalloc(&st, 4);
    run<0, true>("agent-scope release/acquire fences", c, d, f, st);
    run<1, true>("atomics only, plain accesses (unsafe)", c, d, f, st);
    run<2, true>("no fences, sc0 sc1 data accesses", c, d, f, st);
    run<0, false>("agent-scope release/acquire fences", c, d, f, st);
    run<1, false>("atomics only, plain accesses (unsafe)", c, d, f, st);
    run<2, false>("no fences, sc0 sc1 data accesses", c, d, f, st);
    return 0;
}
