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

#include <type_traits>

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

template <int NA>
struct Frags {
    f32x4 a[NA][4];  // lane (row l31, half h): k = 16h .. 16h+15 of the tile, 4 x b128
    float b[16];     // Q[k = 16h + m][col 32w + l31]
};

constexpr int NSTAGE = 3;          // LDS ring: tile t computing (in registers), t+1 readable, t+2 being written
constexpr int NOISE_SLOTS = 2;     // normals kept per element (DL: c,s; MF: this step, next step)

// ABL: ablation bits for tools/ablate.hip (0 in the product): 1 no global loads in the loop,
// 2 no ring writes, 4 no fragment reads, 8 no MFMA, 16 no epilogue, 32 no loop barrier.
template <int MODE, bool ADAM, int ABL = 0>
__global__ __launch_bounds__(NTHREADS) void step_kernel(const StepArgs a) {
    constexpr int NA = (MODE == MODE_DL) ? 2 : 1;
    constexpr bool NOISY = (MODE == MODE_DL || MODE == MODE_MF || MODE == MODE_LANGEVIN);
    constexpr int A_TILE = BM * LDA;
    constexpr int STAGE = NA * A_TILE + KT * BN;
    constexpr int NOISE_LDS = NOISY ? NOISE_SLOTS * 16 * NTHREADS : 0;
    // one array (guide: a second __shared__ object can de-pipeline the loop)
    __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE + NOISE_LDS];
    float* const lds_noise = lds + NSTAGE * STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int tile = xcd_remap(blockIdx.x, a.nrb * a.ncb);
    const int rb = tile / a.ncb, cb = tile - rb * a.ncb;
    const int row0 = rb * BM, col0 = cb * BN;
    const int ld = a.ld;
    const int j = col0 + 32 * wave + l31;  // this lane's output column
    const bool col_ok = j < a.N;

    // ---- epilogue operands, fetched now so their latency hides under the whole GEMM -----
    // accumulator register r of lane (half, l31) is element (row0 + erow(r), j):
    //   erow(r) = (r & 3) + 8 * (r >> 2) + 4 * half
    // e0/e1: old state at that element (DL: c, s; MF: mu, sigma; others: x);
    // e2/e3: Adam moments.  Rows >= B and columns >= N are inside the padded arrays.
    constexpr bool HAS_E0 = (MODE != MODE_AFFINE);
    constexpr bool HAS_E1 = (MODE == MODE_DL || MODE == MODE_MF);
    float e0[16], e1[16], e2[16], e3[16];
    const size_t ebase = (size_t)(row0 + 4 * half) * ld + j;
    {
        const float* p0 = (MODE == MODE_MF) ? a.st0 : a.a0;
        const float* p1 = (MODE == MODE_MF) ? a.st1 : a.a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t idx = ebase + (size_t)((r & 3) + 8 * (r >> 2)) * ld;
            if constexpr (HAS_E0) e0[r] = p0[idx];
            if constexpr (HAS_E1) e1[r] = p1[idx];
            if constexpr (ADAM) {
                e2[r] = a.am[idx];
                e3[r] = a.ad.use_v ? a.av[idx] : 0.0f;
            }
        }
    }
    const float vj = col_ok ? a.V[j] : 0.0f;

    // ---- global -> register staging addresses ------------------------------------
    const int a_r = tid >> 3, a_k = (tid & 7) << 2;   // A tile: 32 rows x 8 float4
    const int b_r = tid >> 5, b_c = (tid & 31) << 2;  // Q tile: rows b_r + 8q, 32 float4 per row
    const float* gA0 = a.a0 + (size_t)(row0 + a_r) * ld + a_k;
    const float* gA1 = (NA == 2) ? a.a1 + (size_t)(row0 + a_r) * ld + a_k : nullptr;
    const float* gQ = a.Q + (size_t)b_r * ld + col0 + b_c;
    const size_t q_step = (size_t)8 * ld;
    const int sa_off = a_r * LDA + a_k;
    const int sb_off = NA * A_TILE + b_r * BN + b_c;

    struct Staged {
        f32x4 ra[NA], rq[4];
    };
    auto load_tile = [&](Staged& g, int kt) {
        const int k0 = kt * KT;
        g.ra[0] = *reinterpret_cast<const f32x4*>(gA0 + k0);
        if constexpr (NA == 2) g.ra[1] = *reinterpret_cast<const f32x4*>(gA1 + k0);
        const float* q = gQ + (size_t)k0 * ld;
#pragma unroll
        for (int i = 0; i < 4; ++i) g.rq[i] = *reinterpret_cast<const f32x4*>(q + i * q_step);
    };
    auto store_tile = [&](const Staged& g, int stage) {
        float* base = lds + stage * STAGE;
#pragma unroll
        for (int n = 0; n < NA; ++n)
            *reinterpret_cast<f32x4*>(base + n * A_TILE + sa_off) = g.ra[n] * a.in_scale + a.in_shift;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(base + sb_off + 8 * i * BN) = g.rq[i];
    };

    // fragment read offsets inside a stage
    const int fa = l31 * LDA + 16 * half;
    const int fb = NA * A_TILE + (16 * half) * BN + 32 * wave + l31;
    auto read_frags = [&](Frags<NA>& f, int stage) {
        const float* st = lds + stage * STAGE;
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                f.a[n][q] = *reinterpret_cast<const f32x4*>(st + n * A_TILE + fa + 4 * q);
#pragma unroll
        for (int m = 0; m < 16; ++m) f.b[m] = st[fb + m * BN];
    };

    // one eighth of a tile's fragments (slot sl of 8): keeps the LDS queue shallow so MFMA
    // issue never waits behind a burst of reads
    auto read_frags_part = [&](Frags<NA>& f, int stage, int sl) {
        const float* st = lds + stage * STAGE;
        if (sl < 4 * NA) f.a[sl >> 2][sl & 3] = *reinterpret_cast<const f32x4*>(st + (sl >> 2) * A_TILE + fa + 4 * (sl & 3));
        f.b[2 * sl] = st[fb + (2 * sl) * BN];
        f.b[2 * sl + 1] = st[fb + (2 * sl + 1) * BN];
    };

    f32x16 acc[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    auto mfma_range = [&](const Frags<NA>& f, int m0, int m1) {
#pragma unroll
        for (int m = m0; m < m1; ++m)
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                if constexpr (ABL & 8) {
                    acc[n][m] += f.a[n][m >> 2][m & 3] * f.b[m];  // keeps the operands live
                } else {
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[n][m >> 2][m & 3], f.b[m], acc[n], 0, 0, 0);
                }
            }
    };

    // Philox noise for accumulator register r of this lane, generated under the MFMAs of
    // K tile r (sliced into the MFMA issue gaps) and parked in LDS until the epilogue.
    // MF carries a second stream: the NEXT step's normals (for the next measured amplitude).
    constexpr int NPH = (MODE == MODE_MF) ? 2 : 1;
    PhiloxState ph[NPH];
    float ph_radius[NPH];
    auto noise_begin = [&](int r) {
        const int b = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int i = 0; i < NPH; ++i) ph[i] = philox_init(a.row_offset + b, a.step + i, j);
    };
    auto noise_rounds = [&](int r0, int r1) {
#pragma unroll
        for (int i = 0; i < NPH; ++i) philox_rounds(ph[i], a.seed, r0, r1);
    };
    auto noise_radius = [&]() {
#pragma unroll
        for (int i = 0; i < NPH; ++i) ph_radius[i] = philox_radius(ph[i]);
    };
    auto noise_end = [&](int r) {
        const NormalPair p = philox_pair(ph[0], ph_radius[0]);
        lds_noise[(0 * 16 + r) * NTHREADS + tid] = p.n0;
        if constexpr (MODE == MODE_DL) lds_noise[(1 * 16 + r) * NTHREADS + tid] = p.n1;
        if constexpr (MODE == MODE_MF) lds_noise[(1 * 16 + r) * NTHREADS + tid] = philox_pair(ph[1], ph_radius[1]).n0;
    };
    auto make_noise = [&](int r) {
        noise_begin(r);
        noise_rounds(0, 10);
        noise_radius();
        noise_end(r);
    };
    const bool gen_noise = NOISY && !a.replay;

    const int nkt = (a.N + KT - 1) / KT;
    const int last = nkt - 1;
    // Two staging register sets: the loads of tile t+4 are issued in iteration t and written
    // to the ring in iteration t+2, ~1.6 tiles (~1.5 us) of MFMA time later.
    Staged gs0, gs1;
    Frags<NA> f0, f1;
    {   // prologue: tiles 0,1 into the ring, tiles 2,3 in flight (indices clamped: a
        // duplicate of the last tile in a ring slot nobody consumes is harmless)
        Staged t0, t1;
        load_tile(t0, 0);
        load_tile(t1, min(1, last));
        load_tile(gs0, min(2, last));
        load_tile(gs1, min(3, last));
        store_tile(t0, 0);
        store_tile(t1, 1);
    }
    __syncthreads();
    read_frags(f0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the loop body starts with settled LDS counters

    // Iteration t, branch-free.  Eight order-pinned slots (sched_barrier(0): nothing moves
    // across); slot s = the MFMAs of k-steps 2s, 2s+1 of tile t, plus ONE staging action
    // (fragment reads of tile t+1 | ring write of tile t+2 | global loads of tile t+4) and ONE
    // slice of the Philox state machine for accumulator register t, alternated with the
    // MFMAs by sched_group_barrier so the matrix pipe never waits on VALU/LDS/VMEM issue.
    // One barrier per tile.  Reads/writes past the last tile touch ring slots nobody consumes.
#define CCVM_SLOT_BEGIN() __builtin_amdgcn_sched_barrier(0)
#define CCVM_SLOT_END(VALU_PER_MFMA)                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 2 * NA; ++i_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             /* 1 MFMA      */ \
        __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0); /* VALU        */ \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             /* 1 DS read   */ \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);             /* 1 DS write  */ \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);             /* 1 VMEM read */ \
    }                                                                                   \
    __builtin_amdgcn_sched_barrier(0)
    auto iteration = [&](const Frags<NA>& cur, Frags<NA>& nxt, Staged& g, int t, auto with_noise) {
        constexpr bool WN = decltype(with_noise)::value;
        constexpr int V = (24 * NPH) / (2 * NA) + 2;  // VALU ops offered per MFMA gap
        const int k4 = min(t + 4, last) * KT;
        float* wbase = lds + ((t + 2) % NSTAGE) * STAGE;
        const float* q4 = gQ + (size_t)k4 * ld;
        const int rstage = (t + 1) % NSTAGE;
#define CCVM_READS(SL) if constexpr (!(ABL & 4)) read_frags_part(nxt, rstage, SL)
        CCVM_SLOT_BEGIN();
        CCVM_READS(0);
        if constexpr (WN) { noise_begin(t); noise_rounds(0, 2); }
        mfma_range(cur, 0, 2);
        CCVM_SLOT_END(V);
        CCVM_READS(1);
        if constexpr (!(ABL & 2)) {
#pragma unroll
            for (int n = 0; n < NA; ++n)
                *reinterpret_cast<f32x4*>(wbase + n * A_TILE + sa_off) = g.ra[n] * a.in_scale + a.in_shift;
        }
        if constexpr (WN) noise_rounds(2, 4);
        mfma_range(cur, 2, 4);
        CCVM_SLOT_END(V);
        CCVM_READS(2);
        if constexpr (!(ABL & 2)) {
            *reinterpret_cast<f32x4*>(wbase + sb_off) = g.rq[0];
            *reinterpret_cast<f32x4*>(wbase + sb_off + 8 * BN) = g.rq[1];
        }
        if constexpr (WN) noise_rounds(4, 6);
        mfma_range(cur, 4, 6);
        CCVM_SLOT_END(V);
        CCVM_READS(3);
        if constexpr (!(ABL & 2)) {
            *reinterpret_cast<f32x4*>(wbase + sb_off + 16 * BN) = g.rq[2];
            *reinterpret_cast<f32x4*>(wbase + sb_off + 24 * BN) = g.rq[3];
        }
        if constexpr (WN) noise_rounds(6, 8);
        mfma_range(cur, 6, 8);
        CCVM_SLOT_END(V);
        CCVM_READS(4);
        if constexpr (!(ABL & 1)) {
            g.ra[0] = *reinterpret_cast<const f32x4*>(gA0 + k4);
            if constexpr (NA == 2) g.ra[1] = *reinterpret_cast<const f32x4*>(gA1 + k4);
            g.rq[0] = *reinterpret_cast<const f32x4*>(q4);
        }
        if constexpr (WN) noise_rounds(8, 10);
        mfma_range(cur, 8, 10);
        CCVM_SLOT_END(V);
        CCVM_READS(5);
        if constexpr (!(ABL & 1)) {
            g.rq[1] = *reinterpret_cast<const f32x4*>(q4 + q_step);
            g.rq[2] = *reinterpret_cast<const f32x4*>(q4 + 2 * q_step);
            g.rq[3] = *reinterpret_cast<const f32x4*>(q4 + 3 * q_step);
        }
        if constexpr (WN) noise_radius();
        mfma_range(cur, 10, 12);
        CCVM_SLOT_END(V);
        CCVM_READS(6);
        if constexpr (WN) noise_end(t);
        mfma_range(cur, 12, 14);
        CCVM_SLOT_END(V);
        CCVM_READS(7);
        mfma_range(cur, 14, 16);
        CCVM_SLOT_END(V);
        if constexpr (!(ABL & 32)) __syncthreads();
    };
#undef CCVM_SLOT_BEGIN
#undef CCVM_SLOT_END
#undef CCVM_READS
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    int t = 0;
    if (gen_noise) {
        const int tn = min(nkt, 16);
        for (; t + 1 < tn; t += 2) {
            iteration(f0, f1, gs0, t, Yes{});
            iteration(f1, f0, gs1, t + 1, Yes{});
        }
    }
    const int noise_done = t;
    for (; t + 1 < nkt; t += 2) {
        iteration(f0, f1, gs0, t, No{});
        iteration(f1, f0, gs1, t + 1, No{});
    }
    if (t < nkt) iteration(f0, f1, gs0, t, No{});
    if (gen_noise) {
        for (int r = noise_done; r < 16; ++r) make_noise(r);
    }

    if constexpr (ABL & 16) {  // ablation: keep the accumulators live, skip the real epilogue
        float sum = vj;
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[n][r] + e0[r] + (HAS_E1 ? e1[r] : 0.0f);
        if (sum == 123.456f) a.o0[0] = sum;
        return;
    }

    // ---- epilogue in the accumulator layout ---------------------------------------
    if constexpr (MODE == MODE_ENERGY) {
        // partial over this wave's 32 columns of (1/2 (x@Q)[b,j] + V[j]) * x[b,j]
        float part[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = col_ok ? (0.5f * acc[0][r] + vj) * e0[r] : 0.0f;
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
        // results first (pure arithmetic on registers), stores afterwards: no load ever waits
        // behind a store
        float r0v[16], r1v[16], r2v[16], r3v[16], r4v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool ok = col_ok && (b < a.B);
            float n0 = 0.0f, n1 = 0.0f, n0n = 0.0f;
            if constexpr (NOISY) {
                if (a.replay) {
                    if (ok) {
                        const size_t widx = (size_t)j * a.B + b;
                        n0 = a.w0[widx];
                        if constexpr (MODE == MODE_DL) n1 = a.w1[widx];
                        if constexpr (MODE == MODE_MF)
                            if (a.s.mf.has_next) n0n = a.w0n[widx];
                    }
                } else {
                    // written by this same thread in noise_end: no barrier needed
                    n0 = lds_noise[(0 * 16 + r) * NTHREADS + tid];
                    if constexpr (MODE == MODE_DL) n1 = lds_noise[(1 * 16 + r) * NTHREADS + tid];
                    if constexpr (MODE == MODE_MF) n0n = lds_noise[(1 * 16 + r) * NTHREADS + tid];
                }
            }

            if constexpr (MODE == MODE_DL) {
                const DlScalars& k = a.s.dl;
                const float c = e0[r], s = e1[r];
                const float c2 = c * c, s2 = s * s;
                const float diff = k.g2 * __builtin_sqrtf(c2 + s2 + 0.5f);
                const float fbk = k.a_v * vj;
                r0v[r] = c + (k.a_q * acc[0][r] + fbk + k.dt * ((k.pm_c - c2 - s2) * c)) + diff * (n0 * k.w_c);
                r1v[r] = s + (k.a_q * acc[1][r] + fbk + k.dt * ((k.pm_s - c2 - s2) * s)) + diff * (n1 * k.w_s);
            } else if constexpr (MODE == MODE_MF) {
                const MfScalars& k = a.s.mf;
                const float mu = e0[r], sg = e1[r];
                const float wdot = n0 * k.inv_sdt;
                const float mu2 = mu * mu;
                const float term1 = (k.a0 - k.g2 * mu2) * mu;
                float fb = k.f_q * acc[0][r] + k.f_v * vj;
                if constexpr (ADAM) {
                    const AdamScalars& ad = a.ad;
                    const float m = ad.beta1 * e2[r] + ad.one_m_beta1 * fb;
                    const float mhat = m * ad.inv_bc1;
                    float upd;
                    if (ad.use_v) {
                        const float v = ad.beta2 * e3[r] + ad.one_m_beta2 * (fb * fb);
                        const float vhat = v * ad.inv_bc2;
                        upd = ad.alpha * (mhat / (__builtin_sqrtf(vhat) + ad.eps));
                        r4v[r] = v;
                    } else {
                        upd = ad.alpha * mhat;
                    }
                    r3v[r] = m;
                    fb = ad.add_assign ? fb + upd : upd;
                }
                const float sh = sg - 0.5f;
                const float dsig = 2.0f * (k.a0 - 3.0f * k.g2 * mu2) * sg - 2.0f * k.j_i * (sh * sh) + (k.one_j + 2.0f * k.g2 * mu2);
                const float diffusion = k.sqrt_j * sh * wdot;
                const float mun = mu + k.dt * (term1 + fb + diffusion);
                r0v[r] = mun;
                r1v[r] = sg + k.dt * dsig;
                r2v[r] = clampf(mun + k.k_next * n0n, -k.S, k.S);
            } else if constexpr (MODE == MODE_LANGEVIN) {
                const LvScalars& k = a.s.lv;
                const float c = e0[r];
                float g = k.g_q * acc[0][r] + k.g_v * vj;
                if constexpr (ADAM) {
                    const AdamScalars& ad = a.ad;
                    const float m = ad.beta1 * e2[r] + ad.one_m_beta1 * g;
                    const float mhat = m * ad.inv_bc1;
                    float upd;
                    if (ad.use_v) {
                        const float v = ad.beta2 * e3[r] + ad.one_m_beta2 * (g * g);
                        const float vhat = v * ad.inv_bc2;
                        upd = ad.alpha * (mhat / (__builtin_sqrtf(vhat) + ad.eps));
                        r4v[r] = v;
                    } else {
                        upd = ad.alpha * mhat;
                    }
                    r3v[r] = m;
                    g = ad.add_assign ? g + upd : upd;
                }
                float x = c + k.dt_fs * g + k.w * n0;
                if (k.use_pump) x += k.dt * ((k.pm - c * c) * c);
                r0v[r] = clampf(x, -k.S, k.S);
            } else if constexpr (MODE == MODE_GD) {
                const PpScalars& k = a.s.pp;
                r0v[r] = clampf(e0[r] - k.step * (acc[0][r] + vj), k.lo, k.hi);
            } else if constexpr (MODE == MODE_ADAMPP) {
                const PpScalars& k = a.s.pp;
                const float g = acc[0][r] + vj;
                r0v[r] = clampf(e0[r] - k.step * (g / (fabsf(g) + k.eps)), k.lo, k.hi);
            } else if constexpr (MODE == MODE_AFFINE) {
                const PpScalars& k = a.s.pp;  // step = f_q, eps = f_v
                r0v[r] = k.step * acc[0][r] + k.eps * vj;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const size_t idx = ebase + (size_t)((r & 3) + 8 * (r >> 2)) * ld;
            if (col_ok && b < a.B) {
                if constexpr (MODE == MODE_DL) {
                    a.o0[idx] = r0v[r];
                    a.o1[idx] = r1v[r];
                } else if constexpr (MODE == MODE_MF) {
                    a.st0[idx] = r0v[r];
                    a.st1[idx] = r1v[r];
                    if (a.s.mf.has_next) a.o0[idx] = r2v[r];
                } else {
                    a.o0[idx] = r0v[r];
                }
                if constexpr (ADAM) {
                    a.am[idx] = r3v[r];
                    if (a.ad.use_v) a.av[idx] = r4v[r];
                }
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
