// Counter-based Wiener noise for the fused step kernels (gfx950 device code).
//
// Philox4x32-10 (Salmon et al., SC'11) keyed by the run seed; the counter is
// (column, global_row_lo, step, global_row_hi) so a trajectory's noise depends only
// on its GLOBAL row index, the column and the step -- never on the tiling, the grid
// or how the batch is sharded over GPUs.  One call yields four 32-bit words; words
// 0,1 give one Box-Muller pair (n0, n1).  The DL solver uses n0 for the in-phase and
// n1 for the quadrature increment; single-state solvers use n0.
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

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1,
                                              uint32_t& o0, uint32_t& o1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    o0 = c0;
    o1 = c1;
}

// 24-bit uniform strictly inside (0, 1): ((x >> 8) + 0.5) * 2^-24.
__device__ __forceinline__ float u01(uint32_t x) {
    return (static_cast<float>(x >> 8) + 0.5f) * 5.9604644775390625e-8f;
}

__device__ __forceinline__ NormalPair normal_pair(uint64_t seed, int64_t grow, int step, int col) {
    uint32_t x0, x1;
    philox4x32_10(static_cast<uint32_t>(col), static_cast<uint32_t>(grow),
                  static_cast<uint32_t>(step), static_cast<uint32_t>(static_cast<uint64_t>(grow) >> 32),
                  static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), x0, x1);
    const float u1 = u01(x0), u2 = u01(x1);
    // r = sqrt(-2 ln u1); v_sin/v_cos take their argument in revolutions.
    const float r = __builtin_sqrtf(-2.0f * __logf(u1));
    NormalPair p;
    p.n0 = r * __builtin_amdgcn_cosf(u2);
    p.n1 = r * __builtin_amdgcn_sinf(u2);
    return p;
}

}  // namespace ccvm
