// Per-step schedule tables of the persistent kernel, built on the device in fp64 (included by the
// ABI translation unit only: these are ordinary, non-template kernels).
#pragma once
#include "ccvm_persist.h"

namespace ccvm {

// ---- per-step schedule tables, built on the device in fp64 (same formulas as the host side of
// ccvm_dl_run / ccvm_mf_run / ccvm_langevin_run; reference lines cited there) ------------------------
struct AdamSched {
    double beta1, beta2;
    int enabled, use_v;
};
__device__ __forceinline__ void adam_bias(const AdamSched& ad, int i, float* row) {
    row[12] = ad.enabled ? (float)(1.0 / (1.0 - pow(ad.beta1, (double)(i + 1)))) : 1.0f;
    row[13] = (ad.enabled && ad.use_v) ? (float)(1.0 / (1.0 - pow(ad.beta2, (double)(i + 1)))) : 1.0f;
}

// `flags` / `flag_words`: the persistent tile kernel's flag lines (ccvm_ptile.h), set to step0 by the same launch
// ("every workgroup has completed the steps before this chunk"); NULL / 0 otherwise
__device__ __forceinline__ void init_flags(unsigned* flags, int flag_words, int step0, int it) {
    if (flags && it < flag_words) flags[it] = (unsigned)step0;
}
struct DlSched {
    double pump, dt, noise_ratio, feedback_scale, g, ul, Sd;
    int pump_rate_flag, T, step0, nsteps;
    unsigned* flags;
    int flag_words;
};
__global__ void dl_schedule_kernel(const DlSched p, float* table) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    init_flags(p.flags, p.flag_words, p.step0, it);
    if (it >= p.nsteps) return;
    const int i = p.step0 + it;
    const double frac = (double)(i + 1) / (double)p.T;
    const double rate = p.pump_rate_flag ? frac : 1.0;
    const double ratio = (p.noise_ratio - 1.0) * exp(-frac * 3.0) + 1.0;
    const double fsd = p.feedback_scale * (0.5 + rate);
    DlScalars k;
    k.a_q = (float)(-p.dt * fsd * 0.25 * p.ul / p.Sd);
    k.a_v = (float)(-p.dt * fsd * p.ul / (2.0 * p.Sd));
    k.pm_c = (float)(-1.0 + p.pump * rate);
    k.pm_s = (float)(-1.0 - p.pump * rate);
    k.dt = (float)p.dt;
    k.g2 = (float)(2.0 * p.g);
    k.w_c = (float)(sqrt(p.dt) * ratio);
    k.w_s = (float)(sqrt(p.dt) / ratio);
    *reinterpret_cast<DlScalars*>(table + (size_t)it * TABLE_WORDS) = k;
}

struct MfSched {
    double pump, dt, j, feedback_scale, g, S, ul;
    int pump_rate_flag, T, step0, nsteps;
    AdamSched ad;
    unsigned* flags;
    int flag_words;
};
__global__ void mf_schedule_kernel(const MfSched p, float* table) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    init_flags(p.flags, p.flag_words, p.step0, it);
    if (it >= p.nsteps) return;
    const int i = p.step0 + it;
    const double sdt = sqrt(p.dt);
    const double j_i = p.j * exp(-(double)(i + 1) / (double)p.T * 3.0);
    const double j_n = p.j * exp(-(double)(i + 2) / (double)p.T * 3.0);
    const double rate = p.pump_rate_flag ? (double)(i + 1) / (double)p.T : 1.0;
    const double p_i = p.pump * rate + 1.0 + j_i;
    MfScalars k;
    k.a0 = (float)(-(1.0 + j_i) + p_i);
    k.g2 = (float)(p.g * p.g);
    k.f_q = (float)(-p.feedback_scale * 0.25 * p.ul / p.S);
    k.f_v = (float)(-p.feedback_scale * p.ul / (2.0 * p.S));
    k.j_i = (float)j_i;
    k.one_j = (float)(1.0 + j_i);
    k.sqrt_j = (float)sqrt(j_i);
    k.inv_sdt = (float)(1.0 / sdt);
    k.dt = (float)p.dt;
    // (k_next and has_next of a row are the WHOLE-RUN values: the persistent kernels decide "the launch's last step"
    // themselves -- it + 1 == nsteps -- so that one table serves every way of chunking the run)
    k.k_next = (float)(sqrt(1.0 / (4.0 * j_n)) / sdt);
    k.S = (float)p.S;
    k.has_next = p.step0 + it + 1 < p.T;
    float* row = table + (size_t)it * TABLE_WORDS;
    *reinterpret_cast<MfScalars*>(row) = k;
    adam_bias(p.ad, i, row);
}

struct LvSched {
    double dt, sigma, feedback_scale, S, pump, ul;
    int use_pump, pump_rate_flag, T, step0, nsteps;
    AdamSched ad;
    unsigned* flags;
    int flag_words;
};
__global__ void lv_schedule_kernel(const LvSched p, float* table) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    init_flags(p.flags, p.flag_words, p.step0, it);
    if (it >= p.nsteps) return;
    const int i = p.step0 + it;
    LvScalars k;
    k.g_q = (float)(-p.ul / (2.0 * p.S));
    k.g_v = k.g_q;
    const double p_i = p.pump_rate_flag ? p.pump * (double)(i + 1) / (double)p.T : p.pump;
    k.pm = (float)(-1.0 + p_i);
    k.dt = (float)p.dt;
    k.dt_fs = (float)(p.dt * p.feedback_scale);
    k.w = (float)(p.sigma * sqrt(p.dt));
    k.S = (float)p.S;
    k.use_pump = p.use_pump;
    float* row = table + (size_t)it * TABLE_WORDS;
    *reinterpret_cast<LvScalars*>(row) = k;
    adam_bias(p.ad, i, row);
}

}  // namespace ccvm
