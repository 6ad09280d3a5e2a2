// Counter-based Wiener noise for the fused step kernels (gfx950 device code).
//
// Philox4x32-10 (Salmon et al., SC'11) keyed by the run seed; the counter is
// (column, global_row_lo, step, global_row_hi) so a trajectory's noise depends only
// on its GLOBAL row index, the column and the step -- never on the tiling, the grid
// or how the batch is sharded over GPUs.  One call yields four 32-bit words; words
// 0,1 give one Box-Muller pair (n0, n1).  The DL solver uses n0 for the in-phase and
// n1 for the quadrature increment; single-state solvers use n0.
//
// The generator is written as a resumable state machine (init / rounds / finish) so the
// step kernel can slice it into the issue gaps between MFMAs.
//
// oracle/philox_ref.py restates exactly this mapping on the host (integer part
// bit-exact, float part to ~1e-6) so that PHILOX-mode runs are checkable too.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccvm {

struct NormalPair {
    float n0, n1;
};

struct PhiloxState {
    uint32_t c0, c1, c2, c3;
};

__device__ __forceinline__ PhiloxState philox_init(int64_t grow, int step, int col) {
    PhiloxState s;
    s.c0 = static_cast<uint32_t>(col);
    s.c1 = static_cast<uint32_t>(grow);
    s.c2 = static_cast<uint32_t>(step);
    s.c3 = static_cast<uint32_t>(static_cast<uint64_t>(grow) >> 32);
    return s;
}

// rounds [r0, r1) of the ten; the key of round r is seed + r * (W0, W1)
__device__ __forceinline__ void philox_rounds(PhiloxState& s, uint64_t seed, int r0, int r1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = r0; r < r1; ++r) {
        const uint32_t k0 = static_cast<uint32_t>(seed) + static_cast<uint32_t>(r) * W0;
        const uint32_t k1 = static_cast<uint32_t>(seed >> 32) + static_cast<uint32_t>(r) * W1;
        const uint32_t hi0 = __umulhi(M0, s.c0), lo0 = M0 * s.c0;
        const uint32_t hi1 = __umulhi(M1, s.c2), lo1 = M1 * s.c2;
        s.c0 = hi1 ^ s.c1 ^ k0;
        s.c1 = lo1;
        s.c2 = hi0 ^ s.c3 ^ k1;
        s.c3 = lo0;
    }
}

// 24-bit uniform strictly inside (0, 1): ((x >> 8) + 0.5) * 2^-24.
__device__ __forceinline__ float u01(uint32_t x) {
    return (static_cast<float>(x >> 8) + 0.5f) * 5.9604644775390625e-8f;
}

// Box-Muller radius from word 0:  r = sqrt(-2 ln u1)
__device__ __forceinline__ float philox_radius(const PhiloxState& s) {
    return __builtin_sqrtf(-2.0f * __logf(u01(s.c0)));
}

// the pair from the radius and word 1 (v_sin/v_cos take their argument in revolutions)
__device__ __forceinline__ NormalPair philox_pair(const PhiloxState& s, float radius) {
    const float u2 = u01(s.c1);
    NormalPair p;
    p.n0 = radius * __builtin_amdgcn_cosf(u2);
    p.n1 = radius * __builtin_amdgcn_sinf(u2);
    return p;
}

__device__ __forceinline__ NormalPair normal_pair(uint64_t seed, int64_t grow, int step, int col) {
    PhiloxState s = philox_init(grow, step, col);
    philox_rounds(s, seed, 0, 10);
    return philox_pair(s, philox_radius(s));
}

}  // namespace ccvm
