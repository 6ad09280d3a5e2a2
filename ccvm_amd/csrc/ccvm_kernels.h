// Device kernels of the CCVM dynamics engine (gfx950 / MI355X only).
//
// One Euler-Maruyama step of every solver is  X' = f(X, A(X) @ Q, noise)  with B
// independent rows and one dense N x N coupling matrix.  `step_kernel` is that whole
// step in one launch: an fp32-MFMA GEMM (v_mfma_f32_32x32x2_f32, exact f32) over
// LDS-staged tiles, and an epilogue that applies the solver's drift/diffusion/clamp
// with in-kernel Philox noise, in the MFMA accumulator layout (no LDS round trip).
//
// Tiling (64-wide waves, one wave per SIMD):
//   workgroup = 256 threads = 4 waves -> 32 batch rows x 128 columns;
//   wave w owns columns [32w, 32w+32) and NA accumulators of 32x32 (DL: c and s share
//   the Q fragments, so each Q element read from LDS feeds two MFMAs);
//   K is walked in tiles of 32, double-buffered in LDS with register-staged
//   prefetch (global loads for tile t+1 are issued before the MFMAs of tile t).
//   Inside a K tile lane-half h owns k in [16h, 16h+16): A fragments are four
//   ds_read_b128 per accumulator (row stride 36 floats: conflict-free), Q fragments
//   are conflict-free ds_read_b32.  The k order differs from the reference's BLAS,
//   which is inside the stated fp32 tolerance.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccvm_philox.h"

namespace ccvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 32;        // batch rows per workgroup
constexpr int BN = 128;       // output columns per workgroup
constexpr int KT = 32;        // K tile
constexpr int LDA = KT + 4;   // LDS row stride of an A tile (floats)
constexpr int NTHREADS = 256;

enum Mode : int {
    MODE_DL = 0,        // two-state DL-CCVM step
    MODE_MF = 1,        // mean-field step (mu, sigma) + next measured amplitude
    MODE_LANGEVIN = 2,  // Langevin / pumped Langevin step
    MODE_ENERGY = 3,    // row partials of 1/2 xQx + Vx
    MODE_GD = 4,        // projected gradient step (post-processor)
    MODE_ADAMPP = 5,    // one Adam step from zero moments (post-processor)
    MODE_AFFINE = 6     // y = f_q * (A(x) @ Q) + f_v * V   (the bare feedback term)
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

struct StepArgs {
    const float* Q;
    const float* V;
    const float* a0;    // GEMM input 0 (pitched B x N)
    const float* a1;    // GEMM input 1 (DL: s)
    float* o0;          // DL: c'; MF: next measured amplitude; LV/GD/ADAMPP: x'; ENERGY: partials
    float* o1;          // DL: s'
    float* st0;         // MF: mu (in place)
    float* st1;         // MF: sigma (in place)
    float* am;          // Adam first moment (in place)
    float* av;          // Adam second moment (in place)
    const float* w0;    // REPLAY: this step's [N][B] block
    const float* w1;    // REPLAY: DL second stream
    const float* w0n;   // REPLAY (MF): next step's block
    uint64_t seed;
    int64_t row_offset;
    int step;
    int replay;
    int B, N, ld;
    int nrb, ncb;       // row blocks, column blocks
    float in_scale, in_shift;  // GEMM input = x * in_scale + in_shift
    union {
        DlScalars dl;
        MfScalars mf;
        LvScalars lv;
        PpScalars pp;
    } s;
    AdamScalars ad;
};

__device__ __forceinline__ float clampf(float x, float lo, float hi) {
    return fminf(fmaxf(x, lo), hi);
}

// Blocks b and b+8 share an XCD (round-robin dispatch; speed only, never
// correctness).  Give each XCD a contiguous run of logical tiles, column block
// fastest, so the tiles resident on one XCD share A row blocks and Q panels in L2.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int MODE, bool ADAM>
__global__ __launch_bounds__(NTHREADS) void step_kernel(const StepArgs a) {
    constexpr int NA = (MODE == MODE_DL) ? 2 : 1;
    constexpr int A_TILE = BM * LDA;
    constexpr int STAGE = NA * A_TILE + KT * BN;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int tile = xcd_remap(blockIdx.x, a.nrb * a.ncb);
    const int rb = tile / a.ncb, cb = tile - rb * a.ncb;
    const int row0 = rb * BM, col0 = cb * BN;
    const int ld = a.ld;

    // ---- global -> register staging addresses ------------------------------------
    const int a_r = tid >> 3, a_k = (tid & 7) << 2;   // A tile: 32 rows x 8 float4
    const int b_r = tid >> 5, b_c = (tid & 31) << 2;  // Q tile: rows b_r + 8q, 32 float4 per row
    const float* gA0 = a.a0 + (size_t)(row0 + a_r) * ld + a_k;
    const float* gA1 = (NA == 2) ? a.a1 + (size_t)(row0 + a_r) * ld + a_k : nullptr;
    const float* gQ = a.Q + (size_t)b_r * ld + col0 + b_c;
    const size_t q_step = (size_t)8 * ld;

    f32x4 ra[NA], rq[4];
    auto load_tile = [&](int kt) {
        const int k0 = kt * KT;
        ra[0] = *reinterpret_cast<const f32x4*>(gA0 + k0);
        if constexpr (NA == 2) ra[1] = *reinterpret_cast<const f32x4*>(gA1 + k0);
        const float* q = gQ + (size_t)k0 * ld;
#pragma unroll
        for (int i = 0; i < 4; ++i) rq[i] = *reinterpret_cast<const f32x4*>(q + i * q_step);
    };
    auto store_tile = [&](int buf) {
        float* base = lds + buf * STAGE;
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            f32x4 x = ra[n] * a.in_scale + a.in_shift;
            *reinterpret_cast<f32x4*>(base + n * A_TILE + a_r * LDA + a_k) = x;
        }
        float* bs = base + NA * A_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(bs + (b_r + 8 * i) * BN + b_c) = rq[i];
    };

    f32x16 acc[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    const int nkt = (a.N + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    // fragment read offsets inside a stage
    const int fa = l31 * LDA + 16 * half;
    const int fb = NA * A_TILE + (16 * half) * BN + 32 * wave + l31;

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nkt);
        if (more) load_tile(kt + 1);

        const float* st = lds + cur * STAGE;
        f32x4 af[NA][4];
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                af[n][q] = *reinterpret_cast<const f32x4*>(st + n * A_TILE + fa + 4 * q);
        float bf[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) bf[m] = st[fb + m * BN];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
#pragma unroll
            for (int n = 0; n < NA; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[n][m >> 2][m & 3], bf[m], acc[n], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue in the accumulator layout ---------------------------------------
    // reg r of lane (half, l31): row = (r&3) + 8*(r>>2) + 4*half, col = l31.
    const int j = col0 + 32 * wave + l31;
    const bool col_ok = j < a.N;
    const float vj = col_ok ? a.V[j] : 0.0f;

    if constexpr (MODE == MODE_ENERGY) {
        // partial over this wave's 32 columns of (1/2 (x@Q)[b,j] + V[j]) * x[b,j]
        float part[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float x = col_ok ? a.a0[(size_t)b * ld + j] : 0.0f;
            float p = (0.5f * acc[0][r] + vj) * x;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) p += __shfl_xor(p, off, 64);
            part[r] = p;
        }
        if (l31 == 0) {
            // o0: [ncb*4 column strips][rows_pad] partial sums
            const int strip = cb * 4 + wave;
            const int rows_pad = a.nrb * BM;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                a.o0[(size_t)strip * rows_pad + b] = part[r];
            }
        }
        return;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool ok = col_ok && (b < a.B);
            const size_t idx = (size_t)b * ld + j;
            // noise
            float n0 = 0.0f, n1 = 0.0f, n0n = 0.0f;
            if constexpr (MODE == MODE_DL || MODE == MODE_MF || MODE == MODE_LANGEVIN) {
                if (a.replay) {
                    if (ok) {
                        const size_t widx = (size_t)j * a.B + b;
                        n0 = a.w0[widx];
                        if constexpr (MODE == MODE_DL) n1 = a.w1[widx];
                        if constexpr (MODE == MODE_MF)
                            if (a.s.mf.has_next) n0n = a.w0n[widx];
                    }
                } else {
                    const NormalPair p = normal_pair(a.seed, a.row_offset + b, a.step, j);
                    n0 = p.n0;
                    n1 = p.n1;
                    if constexpr (MODE == MODE_MF)
                        if (a.s.mf.has_next) n0n = normal_pair(a.seed, a.row_offset + b, a.step + 1, j).n0;
                }
            }

            if constexpr (MODE == MODE_DL) {
                const DlScalars& k = a.s.dl;
                const float c = a.a0[idx], s = a.a1[idx];
                const float c2 = c * c, s2 = s * s;
                const float diff = k.g2 * __builtin_sqrtf(c2 + s2 + 0.5f);
                const float fbk = k.a_v * vj;
                const float cn = c + (k.a_q * acc[0][r] + fbk + k.dt * ((k.pm_c - c2 - s2) * c)) + diff * (n0 * k.w_c);
                const float sn = s + (k.a_q * acc[1][r] + fbk + k.dt * ((k.pm_s - c2 - s2) * s)) + diff * (n1 * k.w_s);
                if (ok) {
                    a.o0[idx] = cn;
                    a.o1[idx] = sn;
                }
            } else if constexpr (MODE == MODE_MF) {
                const MfScalars& k = a.s.mf;
                const float mu = a.st0[idx], sg = a.st1[idx];
                const float wdot = n0 * k.inv_sdt;
                const float mu2 = mu * mu;
                const float term1 = (k.a0 - k.g2 * mu2) * mu;
                float fb = k.f_q * acc[0][r] + k.f_v * vj;
                if constexpr (ADAM) {
                    const AdamScalars& ad = a.ad;
                    const float m = ad.beta1 * a.am[idx] + ad.one_m_beta1 * fb;
                    const float mhat = m * ad.inv_bc1;
                    float upd;
                    if (ad.use_v) {
                        const float v = ad.beta2 * a.av[idx] + ad.one_m_beta2 * (fb * fb);
                        const float vhat = v * ad.inv_bc2;
                        upd = ad.alpha * (mhat / (__builtin_sqrtf(vhat) + ad.eps));
                        if (ok) a.av[idx] = v;
                    } else {
                        upd = ad.alpha * mhat;
                    }
                    if (ok) a.am[idx] = m;
                    fb = ad.add_assign ? fb + upd : upd;
                }
                const float sh = sg - 0.5f;
                const float dsig = 2.0f * (k.a0 - 3.0f * k.g2 * mu2) * sg - 2.0f * k.j_i * (sh * sh) + (k.one_j + 2.0f * k.g2 * mu2);
                const float diffusion = k.sqrt_j * sh * wdot;
                const float mun = mu + k.dt * (term1 + fb + diffusion);
                const float sgn = sg + k.dt * dsig;
                if (ok) {
                    a.st0[idx] = mun;
                    a.st1[idx] = sgn;
                    if (k.has_next) a.o0[idx] = clampf(mun + k.k_next * n0n, -k.S, k.S);
                }
            } else if constexpr (MODE == MODE_LANGEVIN) {
                const LvScalars& k = a.s.lv;
                const float c = a.a0[idx];
                float g = k.g_q * acc[0][r] + k.g_v * vj;
                if constexpr (ADAM) {
                    const AdamScalars& ad = a.ad;
                    const float m = ad.beta1 * a.am[idx] + ad.one_m_beta1 * g;
                    const float mhat = m * ad.inv_bc1;
                    float upd;
                    if (ad.use_v) {
                        const float v = ad.beta2 * a.av[idx] + ad.one_m_beta2 * (g * g);
                        const float vhat = v * ad.inv_bc2;
                        upd = ad.alpha * (mhat / (__builtin_sqrtf(vhat) + ad.eps));
                        if (ok) a.av[idx] = v;
                    } else {
                        upd = ad.alpha * mhat;
                    }
                    if (ok) a.am[idx] = m;
                    g = ad.add_assign ? g + upd : upd;
                }
                float x = c + k.dt_fs * g + k.w * n0;
                if (k.use_pump) x += k.dt * ((k.pm - c * c) * c);
                if (ok) a.o0[idx] = clampf(x, -k.S, k.S);
            } else if constexpr (MODE == MODE_GD) {
                const PpScalars& k = a.s.pp;
                const float x = a.a0[idx];
                if (ok) a.o0[idx] = clampf(x - k.step * (acc[0][r] + vj), k.lo, k.hi);
            } else if constexpr (MODE == MODE_ADAMPP) {
                const PpScalars& k = a.s.pp;
                const float x = a.a0[idx];
                const float g = acc[0][r] + vj;
                if (ok) a.o0[idx] = clampf(x - k.step * (g / (fabsf(g) + k.eps)), k.lo, k.hi);
            } else if constexpr (MODE == MODE_AFFINE) {
                const PpScalars& k = a.s.pp;  // step = f_q, eps = f_v
                if (ok) a.o0[idx] = k.step * acc[0][r] + k.eps * vj;
            }
        }
    }
}

// ---- small elementwise kernels ---------------------------------------------------

__global__ void pack_kernel(const float* __restrict__ src, int rows, int cols, int src_ld,
                            float* __restrict__ dst, int dst_rows, int dst_ld) {
    const size_t total = (size_t)dst_rows * dst_ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / dst_ld), c = (int)(i - (size_t)r * dst_ld);
        dst[i] = (r < rows && c < cols) ? src[(size_t)r * src_ld + c] : 0.0f;
    }
}

__global__ void unpack_kernel(const float* __restrict__ src, int src_ld,
                              float* __restrict__ dst, int rows, int cols, int dst_ld) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
        dst[(size_t)r * dst_ld + c] = src[(size_t)r * src_ld + c];
    }
}

__global__ void clamp_kernel(float* x, int B, int N, int ld, float lo, float hi) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        float* p = x + (size_t)r * ld + c;
        *p = clampf(*p, lo, hi);
    }
}

// y = 0.5 * x / S * (u - l) + 0.5 * (u + l), in the reference's operation order.
__global__ void change_variables_kernel(const float* x, float* y, int B, int N, int ld,
                                        float S, float ul, float half_up) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        const size_t idx = (size_t)r * ld + c;
        y[idx] = 0.5f * x[idx] / S * ul + half_up;
    }
}

// Measured amplitude of step `step` from the current mu (start of an MF chunk):
//   mu_tilde_c = clamp(mu + k * W, -S, S)    (reference mf_solver.py:551-554)
__global__ void mf_prepare_kernel(const float* mu, float* out, int B, int N, int ld,
                                  float k, float S, uint64_t seed, int64_t row_offset, int step,
                                  const float* w0) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / N), j = (int)(i - (size_t)b * N);
        const float n0 = w0 ? w0[(size_t)j * B + b] : normal_pair(seed, row_offset + b, step, j).n0;
        const size_t idx = (size_t)b * ld + j;
        out[idx] = clampf(mu[idx] + k * n0, -S, S);
    }
}

__global__ void philox_fill_kernel(uint64_t seed, int64_t row_offset, int step, int B, int N,
                                   float* w0, float* w1) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / B), b = (int)(i - (size_t)j * B);
        const NormalPair p = normal_pair(seed, row_offset + b, step, j);
        w0[i] = p.n0;
        if (w1) w1[i] = p.n1;
    }
}

// obj[b] = scaled_by * sum over column strips (fixed order -> deterministic)
__global__ void energy_reduce_kernel(const float* partial, int nstrips, int rows_pad, int B,
                                     float scaled_by, float* obj) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float e = 0.0f;
    for (int s = 0; s < nstrips; ++s) e += partial[(size_t)s * rows_pad + b];
    obj[b] = e * scaled_by;
}

// Qs = 1/2 (Q + Q^T) on the padded [ld][ld] matrix
__global__ void symmetrize_kernel(const float* Q, float* Qs, int ld) {
    const size_t total = (size_t)ld * ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / ld), c = (int)(i - (size_t)r * ld);
        Qs[i] = 0.5f * (Q[i] + Q[(size_t)c * ld + r]);
    }
}

}  // namespace ccvm
