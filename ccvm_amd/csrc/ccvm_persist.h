// Row-owner persistent kernel for small problems (N <= 128): the whole chunk of time steps in ONE
// launch.  At these sizes a per-step launch is latency-bound (N=100, B=1000: ~0.3 us of arithmetic
// per step against a ~10 us launch+prologue), and batch rows never interact, so a workgroup can own
// its rows for the entire trajectory with no inter-workgroup traffic at all:
//
//   workgroup = 256 threads = 4 waves -> the 16 rows of a v_mfma_f32_16x16x4_f32 tile:
//     DL: 8 batch rows, c stacked on s (rows 0-7 = c, rows 8-15 = s);  Langevin/PL: 16 batch rows
//   Q fragments are loaded ONCE and stay in registers for every step (wave w owns column tiles
//     w, w+4: KQ x NCT VGPRs); the state makes a round trip through a double-buffered 16 x Kpad LDS
//     tile (it is the MFMA A operand); per step: fragment reads, KQ x NCT MFMAs, update, barrier.
//   lane-quarter q of a wave owns k in [q*KQ, (q+1)*KQ) (one b128 read per four k-steps).
//   DL epilogue: the C/D layout puts c[b,j] in lane L and s[b,j] in lane L^32; the partner value
//     comes by one cross-half shuffle, and the pair splits the noise calls (each Threefry call yields
//     the (W_c, W_s) pair of one element) and swaps the halves the same way.
//   Per-step schedule scalars come from a table built on the device in fp64 (schedule kernels).
//
// Same noise definition, same folded affine map and same pinned update arithmetic as step_kernel.
#pragma once
#include "ccvm_kernels.h"

namespace ccvm {

struct PersistArgs {
    const float* Q;
    const float* V;
    const float* qsum;
    float* x0;          // DL: c; Langevin: c   (pitched, in/out)
    float* x1;          // DL: s
    const float* table; // [nsteps] x DlScalars / LvScalars (fp32 words)
    const float* w0;    // REPLAY noise for the chunk: [nsteps][N][B]
    const float* w1;
    uint64_t seed;
    int64_t row_offset;
    int step0, nsteps;
    int replay;
    int B, N, ld;
    float in_scale, in_shift;
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int MODE, int KQ4, int NCT>
__global__ __launch_bounds__(256) void persist_kernel(const PersistArgs a) {
    static_assert(MODE == MODE_DL || MODE == MODE_LANGEVIN, "persistent kernel: DL and Langevin family");
    constexpr int KQ = 4 * KQ4;          // k values per lane quarter
    constexpr int KPAD = 4 * KQ;         // padded K
    constexpr int LDX = KPAD + 4;        // LDS row stride (floats)
    constexpr int ROWS = (MODE == MODE_DL) ? 8 : 16;  // batch rows per workgroup
    __shared__ __attribute__((aligned(16))) float xs[2 * 16 * LDX];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;      // C/D: rows 4g .. 4g+3;  A/B operands: k quarter
    const int ci = lane & 15;     // C/D: column inside the tile;  A operand: row
    const int row0 = blockIdx.x * ROWS;
    const int ld = a.ld, N = a.N;

    // ---- Q fragments, resident for the whole launch ---------------------------------------
    float qf[NCT][KQ];
    int col[NCT];
    float vj[NCT], shift_j[NCT];
    bool col_ok[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        col[ct] = 16 * (wave + 4 * ct) + ci;
        col_ok[ct] = col[ct] < N;
#pragma unroll
        for (int m = 0; m < KQ; ++m) qf[ct][m] = a.Q[(size_t)(g * KQ + m) * ld + col[ct]];  // zero padded
        vj[ct] = col_ok[ct] ? a.V[col[ct]] : 0.0f;
        shift_j[ct] = col_ok[ct] ? a.in_shift * a.qsum[col[ct]] : 0.0f;
    }

    // ---- this lane's state elements: C/D register reg of tile ct is (row 4g+reg, col[ct]) ----
    // DL: rows 0-7 are c of batch rows row0 .. row0+7, rows 8-15 are s of the same batch rows.
    float own[NCT][4];
    int brow[4];          // batch row (local to this call) of register reg
    bool row_ok[4];
    const bool second = (MODE == MODE_DL) && (g >= 2);  // this lane holds s (quadrature) elements
    float* const arr = second ? a.x1 : a.x0;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int r16 = 4 * g + reg;
        brow[reg] = row0 + ((MODE == MODE_DL) ? (r16 & 7) : r16);
        row_ok[reg] = brow[reg] < a.B;
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) own[ct][reg] = arr[(size_t)brow[reg] * ld + col[ct]];  // padded arrays

    // ---- state tile into LDS buffer 0 (zero beyond N: the arrays are zero padded) ---------------
    for (int i = tid; i < 16 * KPAD; i += 256) {
        const int r = i / KPAD, k = i - r * KPAD;
        const float* src = (MODE == MODE_DL && r >= 8) ? a.x1 : a.x0;
        const int b = row0 + ((MODE == MODE_DL) ? (r & 7) : r);
        xs[r * LDX + k] = (k < ld) ? src[(size_t)b * ld + k] : 0.0f;
    }
    __syncthreads();

    int cur = 0;
    for (int it = 0; it < a.nsteps; ++it) {
        const int step = a.step0 + it;
        // ---- GEMM: acc[ct] = X(16 x KPAD) @ Q(KPAD x 16 cols of tile ct) ---------------------
        const float* xb = xs + cur * 16 * LDX + ci * LDX + g * KQ;
        f32x4v af[KQ4];
#pragma unroll
        for (int t = 0; t < KQ4; ++t) af[t] = *reinterpret_cast<const f32x4v*>(xb + 4 * t);
        f32x4v acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int m = 0; m < KQ; ++m)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m >> 2][m & 3], qf[ct][m], acc[ct], 0, 0, 0);

        // ---- update -------------------------------------------------------------------------
        float* xn = xs + (cur ^ 1) * 16 * LDX;
        if constexpr (MODE == MODE_DL) {
            const DlScalars k = *reinterpret_cast<const DlScalars*>(a.table + (size_t)it * 8);
            const float pm_own = second ? k.pm_s : k.pm_c;
            const float w_own = second ? k.w_s : k.w_c;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (16 * (wave + 4 * ct) >= N) continue;  // wave-uniform: a column tile entirely in the padding
                // noise: the pair (lane, lane^32) needs the four (W_c, W_s) pairs of registers 0..3 of
                // batch rows 4*(g&1)+reg.  The c lane computes registers 0,1, the s lane 2,3; halves swap.
                float n_own[4];
                if (a.replay) {
                    const float* w = second ? a.w1 : a.w0;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        n_own[reg] = (col_ok[ct] && row_ok[reg])
                                         ? w[((size_t)it * N + col[ct]) * a.B + brow[reg]] : 0.0f;
                } else {
                    const int rb = second ? 2 : 0;
                    const NormalPair p0 = normal_pair(a.seed, a.row_offset + brow[0] + rb, step, col[ct]);
                    const NormalPair p1 = normal_pair(a.seed, a.row_offset + brow[1] + rb, step, col[ct]);
                    // keep my quadrature's normals, give the partner its own
                    const float keep0 = second ? p0.n1 : p0.n0, give0 = second ? p0.n0 : p0.n1;
                    const float keep1 = second ? p1.n1 : p1.n0, give1 = second ? p1.n0 : p1.n1;
                    const float recv0 = __shfl_xor(give0, 32, 64), recv1 = __shfl_xor(give1, 32, 64);
                    n_own[0] = second ? recv0 : keep0;
                    n_own[1] = second ? recv1 : keep1;
                    n_own[2] = second ? keep0 : recv0;
                    n_own[3] = second ? keep1 : recv1;
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float mine = own[ct][reg];
                    const float other = __shfl_xor(mine, 32, 64);
                    const float qx = __builtin_fmaf(a.in_scale, acc[ct][reg], shift_j[ct]);
                    // dl_update, symmetric in (c, s): r2 = c^2 + s^2 either way
                    float cn, sn;
                    DlScalars kk = k;
                    kk.pm_c = pm_own;
                    kk.w_c = w_own;
                    dl_update(kk, mine, other, qx, qx, vj[ct], n_own[reg], 0.0f, cn, sn);
                    const float nv = (col_ok[ct] && row_ok[reg]) ? cn : mine;
                    own[ct][reg] = nv;
                    xn[(4 * g + reg) * LDX + col[ct]] = nv;
                }
            }
        } else {
            const LvScalars k = *reinterpret_cast<const LvScalars*>(a.table + (size_t)it * 8);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (16 * (wave + 4 * ct) >= N) continue;
                // registers (0,1) and (2,3) are adjacent rows: one call per pair (ccvm_noise.h)
                float ns[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (!a.replay) {
                    const NormalPair pa = normal_two_rows(a.seed, a.row_offset + brow[0], step, col[ct]);
                    const NormalPair pb = normal_two_rows(a.seed, a.row_offset + brow[2], step, col[ct]);
                    ns[0] = pa.n0; ns[1] = pa.n1; ns[2] = pb.n0; ns[3] = pb.n1;
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const bool ok = col_ok[ct] && row_ok[reg];
                    float n0 = 0.0f;
                    if (a.replay) {
                        if (ok) n0 = a.w0[((size_t)it * N + col[ct]) * a.B + brow[reg]];
                    } else {
                        n0 = ns[reg];
                    }
                    const float mine = own[ct][reg];
                    const float qx = __builtin_fmaf(a.in_scale, acc[ct][reg], shift_j[ct]);
                    const float gq = __builtin_fmaf(k.g_q, qx, k.g_v * vj[ct]);
                    const float nv = ok ? lv_update(k, mine, gq, n0) : mine;
                    own[ct][reg] = nv;
                    xn[(4 * g + reg) * LDX + col[ct]] = nv;
                }
            }
        }
        cur ^= 1;
        __syncthreads();
    }

    // ---- write the state back -----------------------------------------------------------------
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            if (col_ok[ct] && row_ok[reg]) arr[(size_t)brow[reg] * ld + col[ct]] = own[ct][reg];
}

// ---- per-step schedule tables, built on the device in fp64 (same formulas as the host side of
// ccvm_dl_run / ccvm_langevin_run; reference lines cited there) ------------------------------------
struct DlSched {
    double pump, dt, noise_ratio, feedback_scale, g, ul, Sd;
    int pump_rate_flag, T, step0, nsteps;
};
__global__ void dl_schedule_kernel(const DlSched p, float* table) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
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
    *reinterpret_cast<DlScalars*>(table + (size_t)it * 8) = k;
}

struct LvSched {
    double dt, sigma, feedback_scale, S, pump, ul;
    int use_pump, pump_rate_flag, T, step0, nsteps;
};
__global__ void lv_schedule_kernel(const LvSched p, float* table) {
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
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
    *reinterpret_cast<LvScalars*>(table + (size_t)it * 8) = k;
}

static_assert(sizeof(DlScalars) == 32 && sizeof(LvScalars) == 32, "schedule table rows are 8 words");

}  // namespace ccvm
