// Shared definitions of the CCVM dynamics engine's device code (gfx950 / MI355X only): solver modes,
// per-step scalar blocks, and the pinned per-element update arithmetic used by BOTH the per-step tile
// kernel (ccvm_kernels.h) and the persistent small-N kernel (ccvm_persist.h).  Header-only, inline
// device functions only, so every translation unit of the library can include it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "ccvm_noise.h"

namespace ccvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// f(integral_constant<int, 0>{}), ..., f(integral_constant<int, N - 1>{})
template <typename F, int... I>
__device__ __forceinline__ void unroll_indices(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

enum Mode : int {
    MODE_DL = 0,        // two-state DL-CCVM step
    MODE_MF = 1,        // mean-field step (mu, sigma) + next measured amplitude
    MODE_LANGEVIN = 2,  // Langevin / pumped Langevin step
    MODE_ENERGY = 3,    // row partials of 1/2 xQx + Vx
    MODE_GD = 4,        // projected gradient step (post-processor)
    MODE_ADAMPP = 5,    // one Adam step from zero moments (post-processor)
    MODE_AFFINE = 6,    // y = f_q * (A(x) @ Q) + f_v * V   (the bare feedback term)
    MODE_ASGDPP = 7     // one torch.optim.ASGD step (post-processor)
};

// Per-step scalars, computed on the host in fp64 exactly where the reference uses
// Python/numpy doubles, then rounded once to fp32.
struct DlScalars {
    float a_q;      // -dt * fs*(1/2+rate) * (u-l)/(4 Sd)      coefficient of (x@Q)
    float a_v;      // -dt * fs*(1/2+rate) * (u-l)/(2 Sd)      coefficient of V
    float pm_c;     // -1 + pump*rate
    float pm_s;     // -1 - pump*rate
    float dt;
    float g2;       // 2 g
    float w_c;      // sqrt(dt) * noise_ratio_i
    float w_s;      // sqrt(dt) / noise_ratio_i
};
struct MfScalars {
    float a0;       // -(1 + j_i) + p_i
    float g2;       // g^2
    float f_q;      // -fs * (u-l)/(4 S)
    float f_v;      // -fs * (u-l)/(2 S)
    float j_i;
    float one_j;    // 1 + j_i
    float sqrt_j;   // sqrt(j_i)
    float inv_sdt;  // 1/sqrt(dt)
    float dt;
    float k_next;   // sqrt(1/(4 j_{i+1})) / sqrt(dt)   (measured amplitude of the NEXT step)
    float S;
    int has_next;
};
struct LvScalars {
    float g_q;      // -(u-l)/(2S)
    float g_v;      // -(u-l)/(2S)
    float pm;       // -1 + p_i (pumped only)
    float dt;
    float dt_fs;    // dt * feedback_scale
    float w;        // sigma * sqrt(dt)
    float S;
    int use_pump;
};
struct PpScalars {
    float step;     // GD step size / Adam lr
    float eps;
    float lo, hi;
};
struct AdamScalars {
    float beta1, one_m_beta1, inv_bc1;  // inv_bc1 = 1/(1-beta1^(i+1))
    float beta2, one_m_beta2, inv_bc2;
    float alpha;
    float eps;
    int use_v;       // beta2 != 1
    int add_assign;
};

// torch.clamp semantics: NaN propagates (fminf / fmaxf would turn a diverged amplitude into a bound
// and hide the divergence the reference reports as NaN objective values).
__device__ __forceinline__ float clampf(float x, float lo, float hi) {
    const float c = fminf(fmaxf(x, lo), hi);
    return x != x ? x : c;
}

// ---- per-element updates ---------------------------------------------------------------
// The epilogue is unrolled over the 16 accumulator registers; with free FMA contraction the
// compiler fuses differently for different registers and a row's rounding would depend on its
// position in the tile (breaking "shards are bit-identical to the unsharded run").  These
// helpers pin the operation sequence: contraction off, fmaf where fusion is intended.
#pragma clang fp contract(off)
__device__ __forceinline__ float adam_precondition(const AdamScalars& ad, float g, float m_old, float v_old,
                                                   float& m_new, float& v_new) {
    m_new = __builtin_fmaf(ad.beta1, m_old, ad.one_m_beta1 * g);
    const float mhat = m_new * ad.inv_bc1;
    float upd;
    if (ad.use_v) {
        v_new = __builtin_fmaf(ad.beta2, v_old, ad.one_m_beta2 * (g * g));
        const float vhat = v_new * ad.inv_bc2;
        upd = ad.alpha * (mhat / (__builtin_sqrtf(vhat) + ad.eps));
    } else {
        v_new = 0.0f;
        upd = ad.alpha * mhat;
    }
    return ad.add_assign ? g + upd : upd;
}

// (Round 4 tried the two quadratures of ONE element as the halves of packed fp32 instructions -- 13 instructions
// instead of 24, same operations: neutral to 2 % slower on every kernel family in a same-box A/B, and twice as slow
// on the 32 x 32 split-K tiles (tools/ab_build.sh, gpurun_out/r04_ab_dlupdate.txt): the pair has to be assembled in
// adjacent registers first, and the updates of these kernels are dependent-issue chains, not instruction-count
// bound.  Pairs of ELEMENTS whose operands already sit in adjacent registers are another matter: dl_update2.)
__device__ __forceinline__ void dl_update(const DlScalars& k, float c, float s, float qc, float qs, float vj,
                                          float n0, float n1, float& cn, float& sn) {
    const float c2 = c * c, s2 = s * s;
    const float r2 = c2 + s2;
    const float diff = k.g2 * __builtin_amdgcn_sqrtf(r2 + 0.5f);  // raw v_sqrt_f32 (1 ulp)
    const float fbk = k.a_v * vj;
    const float dc = __builtin_fmaf(k.a_q, qc, fbk) + k.dt * ((k.pm_c - r2) * c);
    const float ds = __builtin_fmaf(k.a_q, qs, fbk) + k.dt * ((k.pm_s - r2) * s);
    cn = c + __builtin_fmaf(diff, n0 * k.w_c, dc);
    sn = s + __builtin_fmaf(diff, n1 * k.w_s, ds);
}

// dl_update for TWO elements at once on gfx950's packed fp32 pipe (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two
// IEEE operations per lane and instruction at the scalar forms' issue rate): the same operations in the same order
// per element, so the results equal dl_update's bit for bit -- 24 instructions for two elements instead of 2 x 24.
// Only for phases where no MFMA is in flight (packed VALU next to MFMAs is an anti-lever: cdna_hip_programming.md).
__device__ __forceinline__ void dl_update2(const DlScalars& k, f32x2 c, f32x2 s, f32x2 qc, f32x2 qs, float vj,
                                           f32x2 n0, f32x2 n1, f32x2& cn, f32x2& sn) {
    const f32x2 c2 = c * c, s2 = s * s;
    const f32x2 r2 = c2 + s2;
    const f32x2 h = r2 + 0.5f;
    f32x2 root;
    root.x = __builtin_amdgcn_sqrtf(h.x);
    root.y = __builtin_amdgcn_sqrtf(h.y);
    const f32x2 diff = k.g2 * root;
    const float fbk1 = k.a_v * vj;
    const f32x2 fbk = {fbk1, fbk1}, aq = {k.a_q, k.a_q};
    const f32x2 dc = __builtin_elementwise_fma(aq, qc, fbk) + k.dt * ((k.pm_c - r2) * c);
    const f32x2 ds = __builtin_elementwise_fma(aq, qs, fbk) + k.dt * ((k.pm_s - r2) * s);
    cn = c + __builtin_elementwise_fma(diff, n0 * k.w_c, dc);
    sn = s + __builtin_elementwise_fma(diff, n1 * k.w_s, ds);
}

__device__ __forceinline__ void mf_update(const MfScalars& k, float mu, float sg, float fb, float n0,
                                          float& mun, float& sgn) {
    const float wdot = n0 * k.inv_sdt;
    const float mu2 = mu * mu;
    const float term1 = (k.a0 - k.g2 * mu2) * mu;
    const float sh = sg - 0.5f;
    const float dsig = 2.0f * (k.a0 - 3.0f * k.g2 * mu2) * sg - 2.0f * k.j_i * (sh * sh) + (k.one_j + 2.0f * k.g2 * mu2);
    const float diffusion = k.sqrt_j * sh * wdot;
    mun = __builtin_fmaf(k.dt, term1 + fb + diffusion, mu);
    sgn = __builtin_fmaf(k.dt, dsig, sg);
}

// change_variables (dl_solver.py:219-235): 0.5 * y / S * (u - l) + 0.5 * (u + l), one rounding per
// operation in the reference's order (torch evaluates it as four elementwise fp32 ops); half_up is
// 0.5 * (u + l) rounded once on the host.
__device__ __forceinline__ float change_var(float y, float S, float ul, float half_up) {
    return 0.5f * y / S * ul + half_up;
}

// `S`: the clamp bound of this column (k.S, or the per-variable saturation of the column)
__device__ __forceinline__ float lv_update(const LvScalars& k, float c, float g, float n0, float S) {
    float x = __builtin_fmaf(k.dt_fs, g, c) + k.w * n0;
    if (k.use_pump) x = __builtin_fmaf(k.dt, (k.pm - c * c) * c, x);
    return clampf(x, -S, S);
}
#pragma clang fp contract(fast)

}  // namespace ccvm
