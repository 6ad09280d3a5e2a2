// Counter-based Wiener noise for the fused step kernels (gfx950 device code).
//
// Threefry2x32-13 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11 -- the Random123 family JAX also uses).  13 rounds is the paper's Crush-resistant
// configuration (passes SmallCrush, Crush and BigCrush); 20 is its safety-margin default.
// Chosen over Philox4x32 and over more rounds for this chip: on gfx950 every VALU instruction
// costs matrix-pipe time (the f32 MFMA shares the FP32 datapath; measured with tools/coissue.hip
// and tools/filler.hip), Philox needs two quarter-rate 32x32->64 multiplies per round, and
// Threefry is add / rotate / xor only (docs/noise.md).
//
//   counter = (column, global_row_lo),  key = (K_lo, K_hi ^ global_row_hi),
//   K = step_key(seed, step) = splitmix64 finaliser of  seed + 0x9E3779B97F4A7C15 * (step + 1)
//
// (the step is mixed into the 64-bit key by a bijective hash, NOT xor-ed into the seed: with
// seed ^ step the runs (seed s, step i) and (seed s ^ d, step i ^ d) consumed the same normals, so
// repetitions with consecutive small seeds reused each other's noise blocks).  The hash runs on the
// scalar ALU (seed and step are wave-uniform): no VALU cost.
// A trajectory's noise depends only on its GLOBAL row index, the column and the step --
// never on the tiling, the grid or how the batch is sharded over GPUs.  The two output words
// give one Box-Muller pair (n0, n1):
//   DL (two Wiener streams per element):  normal_pair(row)  -> (W_c, W_s) of element (row, col);
//   MF / Langevin / pumped Langevin (one stream): global rows 2p and 2p+1 SHARE the call of "row" p,
//     normal_single(row) = (row even ? n0 : n1) of normal_pair(row >> 1)   -- half the calls.
//
// oracle/noise_ref.py restates exactly this mapping on the host (integer part bit-exact,
// checked against the Random123 known-answer vectors; float part to ~1e-6).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccvm {

struct NormalPair {
    float n0, n1;
};

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) {
    return __builtin_amdgcn_alignbit(x, x, 32 - r);  // v_alignbit_b32: one full-rate op
}

__device__ __forceinline__ void threefry2x32_13(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1,
                                                uint32_t& o0, uint32_t& o1) {
    const uint32_t ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
#define CCVM_TF_ROUND(R) x0 += x1; x1 = rotl32(x1, R) ^ x0
#define CCVM_TF_KEY(S) x0 += ks[(S) % 3]; x1 += ks[((S) + 1) % 3] + (S)
    CCVM_TF_ROUND(13); CCVM_TF_ROUND(15); CCVM_TF_ROUND(26); CCVM_TF_ROUND(6);  CCVM_TF_KEY(1u);
    CCVM_TF_ROUND(17); CCVM_TF_ROUND(29); CCVM_TF_ROUND(16); CCVM_TF_ROUND(24); CCVM_TF_KEY(2u);
    CCVM_TF_ROUND(13); CCVM_TF_ROUND(15); CCVM_TF_ROUND(26); CCVM_TF_ROUND(6);  CCVM_TF_KEY(3u);
    CCVM_TF_ROUND(17);  // round 13; a key injection follows every FOURTH round only
#undef CCVM_TF_ROUND
#undef CCVM_TF_KEY
    o0 = x0;
    o1 = x1;
}

// Per-step 64-bit Threefry key: the SplitMix64 output function (Steele, Lea, Flood 2014; Vigna's
// splitmix64.c constants) applied to seed + golden * (step + 1).  step_key(0, 0) = 0xE220A8397B1DCDAF,
// the first output of splitmix64 seeded with 0 (known answer in tests/test_noise.py).
__host__ __device__ __forceinline__ uint64_t step_key(uint64_t seed, int step) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (static_cast<uint64_t>(static_cast<uint32_t>(step)) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// 24-bit uniform strictly inside (0, 1): ((x >> 8) + 0.5) * 2^-24.
__device__ __forceinline__ float u01(uint32_t x) {
    return (static_cast<float>(x >> 8) + 0.5f) * 5.9604644775390625e-8f;
}

__device__ __forceinline__ NormalPair normal_pair(uint64_t seed, int64_t grow, int step, int col) {
    uint32_t x0, x1;
    const uint64_t key = step_key(seed, step);  // wave-uniform: scalar ALU
    threefry2x32_13(static_cast<uint32_t>(col), static_cast<uint32_t>(grow),
                    static_cast<uint32_t>(key),
                    static_cast<uint32_t>(key >> 32) ^ static_cast<uint32_t>(static_cast<uint64_t>(grow) >> 32),
                    x0, x1);
    // Box-Muller: r = sqrt(-2 ln u1) with the raw v_log_f32 (log2) / v_sqrt_f32 (u1 >= 2^-25: no
    // denormal fix-ups needed; every VALU op here costs matrix-pipe time); v_sin/v_cos take
    // their argument in revolutions
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(x0)));
    const float u2 = u01(x1);
    NormalPair p;
    p.n0 = r * __builtin_amdgcn_cosf(u2);
    p.n1 = r * __builtin_amdgcn_sinf(u2);
    return p;
}

// Two independent calls in lockstep.  A wave that is alone on its SIMD issues a DEPENDENT VALU
// instruction only every ~8 cycles (an independent one every 4) and hipcc does not interleave two
// calls by itself; the persistent kernel, one wave per SIMD, gets its ILP from here.  Bit-identical
// to two normal_pair calls.
__device__ __forceinline__ void normal_pair_x2(uint64_t seed, int64_t grow_a, int64_t grow_b, int step, int col,
                                               NormalPair& pa, NormalPair& pb) {
    const uint64_t key = step_key(seed, step);  // wave-uniform: scalar ALU
    const uint32_t k0 = static_cast<uint32_t>(key);
    const uint32_t hi = static_cast<uint32_t>(key >> 32);
    const uint32_t k1a = hi ^ static_cast<uint32_t>(static_cast<uint64_t>(grow_a) >> 32);
    const uint32_t k1b = hi ^ static_cast<uint32_t>(static_cast<uint64_t>(grow_b) >> 32);
    const uint32_t ksa[3] = {k0, k1a, 0x1BD11BDAu ^ k0 ^ k1a};
    const uint32_t ksb[3] = {k0, k1b, 0x1BD11BDAu ^ k0 ^ k1b};
    uint32_t a0 = static_cast<uint32_t>(col) + ksa[0], a1 = static_cast<uint32_t>(grow_a) + ksa[1];
    uint32_t b0 = static_cast<uint32_t>(col) + ksb[0], b1 = static_cast<uint32_t>(grow_b) + ksb[1];
#define CCVM_TF_ROUND2(R) a0 += a1; b0 += b1; a1 = rotl32(a1, R); b1 = rotl32(b1, R); a1 ^= a0; b1 ^= b0
#define CCVM_TF_KEY2(S) a0 += ksa[(S) % 3]; b0 += ksb[(S) % 3]; a1 += ksa[((S) + 1) % 3] + (S); b1 += ksb[((S) + 1) % 3] + (S)
    CCVM_TF_ROUND2(13); CCVM_TF_ROUND2(15); CCVM_TF_ROUND2(26); CCVM_TF_ROUND2(6);  CCVM_TF_KEY2(1u);
    CCVM_TF_ROUND2(17); CCVM_TF_ROUND2(29); CCVM_TF_ROUND2(16); CCVM_TF_ROUND2(24); CCVM_TF_KEY2(2u);
    CCVM_TF_ROUND2(13); CCVM_TF_ROUND2(15); CCVM_TF_ROUND2(26); CCVM_TF_ROUND2(6);  CCVM_TF_KEY2(3u);
    CCVM_TF_ROUND2(17);
#undef CCVM_TF_ROUND2
#undef CCVM_TF_KEY2
    const float la = __builtin_amdgcn_logf(u01(a0)), lb = __builtin_amdgcn_logf(u01(b0));
    const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * la);
    const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * lb);
    const float ua = u01(a1), ub = u01(b1);
    pa.n0 = ra * __builtin_amdgcn_cosf(ua);
    pb.n0 = rb * __builtin_amdgcn_cosf(ub);
    pa.n1 = ra * __builtin_amdgcn_sinf(ua);
    pb.n1 = rb * __builtin_amdgcn_sinf(ub);
}

// One-stream solvers: the normal of element (global row, col) at `step`.
__device__ __forceinline__ float normal_single(uint64_t seed, int64_t grow, int step, int col) {
    const NormalPair p = normal_pair(seed, grow >> 1, step, col);
    return (grow & 1) ? p.n1 : p.n0;
}

// The normals of two ADJACENT local rows b (even) and b+1 whose global rows are off+b, off+b+1:
// one call when `off` is even (the common case), two when a shard starts at an odd row.
__device__ __forceinline__ NormalPair normal_two_rows(uint64_t seed, int64_t grow_even_local, int step, int col) {
    const NormalPair p = normal_pair(seed, grow_even_local >> 1, step, col);
    if (!(grow_even_local & 1)) return p;
    NormalPair r;
    r.n0 = p.n1;
    r.n1 = normal_pair(seed, (grow_even_local >> 1) + 1, step, col).n0;
    return r;
}

// normal_two_rows for two row pairs at once (local rows e and e + 2 of a lane), in lockstep when the
// shard starts at an even global row.
__device__ __forceinline__ void normal_two_rows_x2(uint64_t seed, int64_t grow_a, int64_t grow_b, int step, int col,
                                                   NormalPair& pa, NormalPair& pb) {
    if (!((grow_a | grow_b) & 1)) {
        normal_pair_x2(seed, grow_a >> 1, grow_b >> 1, step, col, pa, pb);
    } else {
        pa = normal_two_rows(seed, grow_a, step, col);
        pb = normal_two_rows(seed, grow_b, step, col);
    }
}

}  // namespace ccvm
