// Device kernels of the CCVM dynamics engine (gfx950 / MI355X only).
//
// One Euler-Maruyama step of every solver is  X' = f(X, A(X) @ Q, noise)  with B independent
// rows and one dense N x N coupling matrix.  `step_kernel` is that whole step in one launch:
// an fp32-MFMA GEMM (v_mfma_f32_32x32x2_f32, exact f32) fed from an LDS ring, and an epilogue
// that applies the solver's drift/diffusion/clamp with in-kernel counter-based noise, in the
// MFMA accumulator layout (no LDS round trip for the result).
//
// Workgroup = 512 threads = 8 waves, wave-specialised (one producer + one consumer per SIMD):
//   consumers (waves 0-3): 32 batch rows x 128 columns; wave w owns columns [32w, 32w+32) and
//     NA accumulators of 32x32 (DL: c and s share the Q fragments).  Per K tile of 32 they read
//     next tile's fragments from the ring (conflict-free ds_read_b128 / ds_read_b32) in the
//     issue gaps of this tile's 16*NA MFMAs, then run the epilogue.
//   producers (waves 4-7): stream tiles L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA: no
//     VGPRs, no ds_write), four tiles ahead in a 4-stage ring, and generate the Threefry
//     normals for the consumers' accumulator elements into LDS.
// Why this shape (all measured with tools/ablate.hip, see docs/kernel-step.md): a single wave issues in
// order, so any LDS/VMEM/VALU issue stall delays its next MFMA; ds_write_b128 staging from
// VGPRs blocked the SIMD's MFMA for its 13-cycle data transfer; L2-hit latency under load is
// ~1.5 us, so >= 3 tiles must be in flight.
// One s_barrier per K tile.  The A tile is XOR-swizzled in LDS through the per-lane SOURCE
// address of the DMA (dest is lane-linear by hardware); the input map x*scale+shift is folded
// into the epilogue as  scale*(x@Q) + shift*colsum(Q).  Inside a K tile lane-half h owns
// k in [16h, 16h+16); the k order differs from the reference's BLAS, which is inside the
// stated fp32 tolerance (docs/parity.md).
#pragma once
#include "ccvm_common.h"

namespace ccvm {

constexpr int BM = 32;        // batch rows per workgroup
constexpr int BN = 128;       // output columns per workgroup
constexpr int KT = 32;        // K tile
constexpr int NTHREADS = 256; // consumer threads (the tile mapping)
constexpr int NPW = 4;        // producer waves (one per SIMD)
constexpr int WG_THREADS = NTHREADS + 64 * NPW;
constexpr int A_TILE = BM * KT;   // floats, 128-B rows, XOR-swizzled 16-B chunks
constexpr int Q_TILE = KT * BN;   // floats, 512-B rows, linear

struct StepArgs {
    const float* Q;
    const float* V;
    const float* qsum;  // column sums of Q (length ld): the affine input map's constant term
    const float* a0;    // GEMM input 0 (pitched B x N)
    const float* a1;    // GEMM input 1 (DL: s)
    float* o0;          // DL: c'; MF: next measured amplitude; LV/GD/ADAMPP: x'; ENERGY: partials
    float* o1;          // DL: s'
    float* st0;         // MF: mu (in place)
    float* st1;         // MF: sigma (in place)
    float* carry;       // MF: this step's normals, written by the previous step (or mf_prepare), in place
    float* am;          // Adam first moment (in place)
    float* av;          // Adam second moment (in place)
    const float* w0;    // REPLAY: this step's [N][B] block
    const float* w1;    // REPLAY: DL second stream
    const float* w0n;   // REPLAY (MF): next step's block
    const float* s_cols; // per-variable saturation S_j (length ld) or NULL: then the scalars carry S.
                         // With s_cols the scalars are built for S = 1, Q is the row-scaled copy
                         // Q[k][j] / S_k, and the epilogue applies 1 / S_j and clamps to +-S_j.
    unsigned long long* dbg;  // ablation stamps only
    uint64_t seed;
    int64_t row_offset;
    int step;
    int replay;
    int B, N, ld;
    int wld;  // REPLAY: pitch of the noise blocks (ccvm_noise::w_ld; >= B)
    int nrb, ncb;       // row blocks, column blocks
    int xr, xc;         // tiles of one XCD form an xr x xc rectangle; xr = 0: no rectangle, runs of a blocked order with
                        // super-columns of xc column blocks (xc = 0: row-major)
    int ks;             // host only: the tile shape this launch plan uses (template parameter KS)
    float in_scale, in_shift;  // GEMM input = x * in_scale + in_shift
    union {
        DlScalars dl;
        MfScalars mf;
        LvScalars lv;
        PpScalars pp;
    } s;
    AdamScalars ad;
};

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

// Diagnostic stamp.  NOTE: the lgkmcnt(0) it needs also drains LDS-DMA in flight (LDS-DMA counts
// on lgkmcnt as well as vmcnt), so on the producer side it serialises the DMA it brackets.
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ unsigned long long stamp_delta(unsigned long long a, unsigned long long b) { return b - a; }

// Sixteenths of a step's noise units made in the prologue by the consumer / by the producer wave
// (the remainder is spread over the main loop).  Tuned with tools/ablate.hip.
// Wave priorities (s_setprio) of the producer / consumer waves, for tools/ablate.hip.  Both 0: while the
// producers still made noise inside the main loop a raised producer priority paid; with the noise in the
// prologue ANY producer priority above the consumers' costs ~1 us per step at N = 1000 (35.4 vs 34.5),
// and a raised consumer priority changes nothing.
#ifndef CCVM_CONSUMER_PRIO
#define CCVM_CONSUMER_PRIO 0
#endif
#ifndef CCVM_PRODUCER_PRIO
#define CCVM_PRODUCER_PRIO 0
#endif
#ifndef CCVM_NOISE_PROLOGUE_C
#define CCVM_NOISE_PROLOGUE_C 12
#endif
#ifndef CCVM_NOISE_PROLOGUE_P
#define CCVM_NOISE_PROLOGUE_P 4
#endif

// s_waitcnt vmcnt(min(tiles, K) * P): the counter is an instruction immediate, so a wave-uniform value
// walks down a chain of compile-time cases (producer waves only; scalar compares and branches).
template <int P, int K>
__device__ __forceinline__ void wait_vm_tiles(int tiles) {
    if constexpr (K == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (tiles >= K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K * P) : "memory");
        else wait_vm_tiles<P, K - 1>(tiles);
    }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_cvoid;

// ABL: ablation bits for tools/ablate.hip (0 in the product): 1 no DMA loads, 4 no fragment
// reads, 8 no MFMA, 16 no epilogue, 32 no loop barrier (timing only), 64 no noise,
// 128 s_memtime stamps of the consumer loop into a.dbg (diagnostic build: shares, not run time).
//
// KS (1, 2 or 4): in-workgroup split of K.  KS = 1: tile 32 x 128, consumer wave w = column strip w.
// KS = 2: tile 32 x 64 for grids that would leave CUs idle (e.g. N = 500, B = 1000: 128 tiles of
// 32 x 128); consumer wave w = column strip (w & 1), K half (w >> 1): within every K tile it runs
// the k-steps [8 kh, 8 kh + 8) of each lane-half.  After the loop the two halves swap 8 accumulator
// registers through LDS, so each wave ends up with the full sum of 8 of the 16 rows and the
// epilogue work stays balanced.
// KS = 4: tile 32 x 32 for batches that leave three quarters of the chip idle even with KS = 2 (B = 129 ... 256 at
// N = 1000: 64 tiles of 32 x 128); consumer wave w = K quarter w of the one column strip: k-steps [4 w, 4 w + 4) of
// each lane-half.  After the loop every wave leaves its 16 partial registers in LDS and sums the four partials of
// registers 4 w ... 4 w + 3 in a fixed order; the epilogue (4 elements per lane) stays with the consumers.
// VS: per-variable saturation (StepArgs::s_cols).  A template parameter, not a run-time test of the
// pointer: with the run-time form the scalar path of the N = 500 kernels lost 0.4 us per step.
// RING: LDS ring depth = DMA prefetch distance in tiles (0 = the default of 4; even, >= 4; a tuning knob
// for tools/ablate_mid.hip).  Deeper rings are SLOWER for the short-tile kernels (Langevin N = 500,
// 32 x 64 tiles: 8.14 / 8.30 / 8.49 / 8.73 us per step at depth 4 / 6 / 8 / 10): those kernels are bound by
// the per-CU LDS-DMA rate (192 KB per workgroup and step at ~68 GB/s per CU), not by its latency, and
// more tiles queued in front only delay the first one.
template <int MODE, bool ADAM, int ABL = 0, int KS = 1, bool VS = false, int RING = 0>
__global__ __launch_bounds__(WG_THREADS) void step_kernel(const StepArgs a) {
    static_assert(KS == 1 || KS == 2 || KS == 4, "KS");
    static_assert(!VS || MODE == MODE_MF || MODE == MODE_LANGEVIN, "per-variable saturation: MF and Langevin steps");
    constexpr int NA = (MODE == MODE_DL) ? 2 : 1;
    constexpr bool NOISY = (MODE == MODE_DL || MODE == MODE_MF || MODE == MODE_LANGEVIN);
    constexpr int BNT = BN / KS;             // columns per workgroup
    constexpr int QT = KT * BNT;             // floats of a Q tile
    constexpr int STAGE = NA * A_TILE + QT;
    constexpr int NM = 16 / KS;              // k-steps per consumer wave, lane-half and K tile
    constexpr int NQA = 4 / KS;              // b128 A-fragment reads per accumulator and tile
    constexpr int NG = 8 / KS;               // fragment read groups per tile
    constexpr int NR = 16 / KS;              // accumulator registers a wave finishes (epilogue rows)
    constexpr int NSTAGE = RING ? RING : 4;
    static_assert(NSTAGE >= 4 && NSTAGE % 2 == 0, "ring depth: the consumer loop alternates two fragment sets");
    // DL: (W_c, W_s) per accumulator register a wave finishes (NR of the 16: the split-K shapes keep their LDS small
    // enough for two workgroups per CU -- 32 x 64 tiles: 80 KB instead of 96 -- which pays where a CU runs several
    // workgroups one after the other)
    constexpr int NOISE_LDS = NOISY ? NA * NR * NTHREADS : 0;
    // one array (a second __shared__ object can de-pipeline the loop, guide section 5)
    __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE + NOISE_LDS];
    float* const lds_noise = lds + NSTAGE * STAGE;

    // tid / wave index the 256-thread tile mapping; a producer thread shares the mapping of the
    // consumer thread 256 below it (it makes that thread's noise)
    const int tid = threadIdx.x & (NTHREADS - 1);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= NTHREADS;
    const int cs = (KS == 1) ? wave : (KS == 2) ? (wave & 1) : 0;   // column strip of 32
    const int kh = (KS == 1) ? 0 : (KS == 2) ? (wave >> 1) : wave;  // K half / quarter
    const int R0 = kh * NR;                         // first accumulator register this wave finishes

    // Blocks b and b+8 share an XCD (round-robin dispatch; speed only, never correctness).  When the
    // grid divides evenly, each XCD gets an xr x xc rectangle of tiles chosen on the host to
    // minimise the bytes its private L2 must hold (xr A row-blocks + xc Q column panels).
    int rb, cb;
    if (a.xr > 0) {
        const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int regions_c = a.ncb / a.xc;
        rb = (x / regions_c) * a.xr + i / a.xc;
        cb = (x % regions_c) * a.xc + i % a.xc;
    } else {
        // No rectangle divides the grid: every XCD takes a contiguous run of tiles in a BLOCKED order (round 5) -- super-
        // columns of a.xc column blocks, walked row block by row block -- so the workgroups an XCD runs at a time share
        // a.xc Q panels and a few A row blocks in its L2 instead of one A row block and a whole row of Q panels (row-major
        // order, a.xc = 0: at N = 2000 every XCD pulled all of Q once per row of tiles).
        const int tile = xcd_remap(blockIdx.x, a.nrb * a.ncb);
        if (a.xc > 0) {
            const int per = a.nrb * a.xc;           // tiles of a full super-column
            const int sc = tile / per, rem = tile - sc * per;
            const int w = min(a.xc, a.ncb - sc * a.xc);  // (the last one may be narrower)
            rb = rem / w;
            cb = sc * a.xc + rem - rb * w;
        } else {
            rb = tile / a.ncb;
            cb = tile - rb * a.ncb;
        }
    }
    const int row0 = rb * BM, col0 = cb * BNT;
    const int ld = a.ld;
    const int j = col0 + 32 * cs + l31;  // this lane's output column
    const bool col_ok = j < a.N;
    const int nkt = (a.N + KT - 1) / KT;
    const int last = nkt - 1;
    const bool fused = NOISY && !a.replay;  // noise source: fused generator vs replayed normals
    // what the producers generate: MF makes the NEXT step's normals only (none after the last step)
    const bool gen_noise = fused && (MODE != MODE_MF || a.s.mf.has_next);
    // accumulator register r of lane (half, l31) is element (row0 + erow(r), j):
    auto erow = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * half; };

    // Noise for accumulator registers R0 .. R0 + NR - 1 of consumer thread `tid` (made by that thread
    // and by its producer twin 256 threads above, into lds_noise; consumed in the epilogue).
    // DL: one unit per register (its (W_c, W_s) pair).  One-stream solvers: registers 2i and 2i+1
    // are adjacent rows and share a unit (normal_two_rows); MF generates the NEXT step's normals
    // (this step's arrive through the carry buffer).
    // Every VALU instruction issued while the matrix pipe is busy costs matrix time (measured:
    // +4.1 us per DL step at N = 1000 when all units ran inside the main loop), so the units are
    // made in the prologue, while the first ring tiles are still in flight and the MFMA pipe is
    // idle anyway: units [0, NPRO_C) by the consumer, [NPRO_C, NPRO_C + NPRO_P) by the producer
    // after it has issued the first NSTAGE tiles, the rest (if any) one per main-loop iteration.
    constexpr int NUNIT = ((MODE == MODE_DL) ? 16 : 8) / KS;  // noise work units per lane and step
    // split tuned on the same box (tools/ablate.hip, bench.py): 12/4 for the 32 x 128 tiles (N = 1000:
    // -0.3 us against 8/8), an even split for the short 32 x 64 split-K kernels
    constexpr int NPRO_C = NUNIT * (KS == 1 ? CCVM_NOISE_PROLOGUE_C : 8) / 16;
    constexpr int NPRO_P = NUNIT * (KS == 1 ? CCVM_NOISE_PROLOGUE_P : 8) / 16;
    static_assert(NPRO_C + NPRO_P <= NUNIT, "noise split");
    auto make_noise = [&](int u) {
        if constexpr (NOISY && !(ABL & 64)) {
            if constexpr (MODE == MODE_DL) {
                const int r = R0 + u;
                const NormalPair p = normal_pair(a.seed, a.row_offset + row0 + erow(r), a.step, j);
                lds_noise[(0 * NR + u) * NTHREADS + tid] = p.n0;
                lds_noise[(1 * NR + u) * NTHREADS + tid] = p.n1;
            } else {
                const int r = R0 + 2 * u;  // rows b (even) and b + 1
                const int st = (MODE == MODE_MF) ? a.step + 1 : a.step;
                const NormalPair p = normal_two_rows(a.seed, a.row_offset + row0 + erow(r), st, j);
                lds_noise[(2 * u) * NTHREADS + tid] = p.n0;
                lds_noise[(2 * u + 1) * NTHREADS + tid] = p.n1;
            }
        }
    };

    // ---- epilogue, shared by both kinds of wave -------------------------------------------------
    // For the solver steps the producer waves, idle once the last tile is in the ring, take half of
    // the epilogue: the consumer hands the finished sums of its upper H accumulator registers and
    // their operands to its producer twin through the (then idle) ring, and both run
    // `run_epilogue` on H registers.  Two waves per SIMD also hide each other's VALU dependency
    // stalls; with the consumers alone the 16-element epilogue cost 2.2 us of the 35 us step.
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    constexpr bool SHARE = NOISY && !(ABL & 16) && KS != 4;
    constexpr int H = SHARE ? NR / 2 : NR;  // epilogue elements per thread
    static_assert(H % 4 == 0, "element i uses lane offset eoff[i & 3]");
    constexpr bool HAS_E0 = (MODE != MODE_AFFINE);
    constexpr bool HAS_E1 = (MODE == MODE_DL || MODE == MODE_MF);
    constexpr int HAND_OFF = 4096;  // floats from the ring start: behind the KS = 2 exchange area
    constexpr int HAND_VALUES = NA * H + H + (HAS_E1 ? H : 0) + (ADAM ? 2 * H : 0) + (MODE == MODE_MF ? H : 0);
    static_assert(!SHARE || HAND_OFF + HAND_VALUES * NTHREADS <= NSTAGE * STAGE, "hand-over area fits in the ring");
    float* const hand = lds + HAND_OFF;  // [value][tid]
    // element address = [uniform: array + (row0 + 8 * (r >> 2)) * ld]  +  [lane: eoff[r & 3]]
    // (scalar base + constant 32-bit lane offset: no per-access address VALU)
    unsigned eoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) eoff[i] = (unsigned)((i + 4 * half) * ld + j);
    auto gofs = [&](int r) { return (size_t)(row0 + 8 * (r >> 2)) * ld; };  // uniform part
    const float vj = col_ok ? a.V[j] : 0.0f;
    const float shift_j = a.in_shift * a.qsum[j];  // shift * colsum(Q)[j]
    // per-variable saturation: the column's bound (its reciprocal, the 1 / S_j factor of the feedback
    // term, is formed in the epilogue: a division here would put a vmcnt(0) wait for all of these loads
    // in front of the producers' first DMA)
    float sat_j = 1.0f;
    if constexpr (VS) sat_j = col_ok ? a.s_cols[j] : 1.0f;

    // Registers R0 + ibase + ii, ii = 0 .. H - 1.  hf: affine-folded GEMM sums; he0/he1: old state;
    // he2/he3: Adam moments; hcar: MF's normals of this step.
    // The new state is written once and next read by the following launch (mostly from other CUs):
    // non-temporal stores shorten the drain of the tail (-0.5 us per step at the headline shape).
#ifdef CCVM_ABL_NO_STORES  // tools/ablate.hip: the epilogue's arithmetic without its stores
    auto st_nt = [](float* p, float x) { if (x == 123.456f) __builtin_nontemporal_store(x, p); };
#else
    auto st_nt = [](float* p, float x) { __builtin_nontemporal_store(x, p); };
#endif
    auto run_epilogue = [&](const float (&hf)[NA][H], const float (&he0)[H], const float (&he1)[H],
                            const float (&he2)[H], const float (&he3)[H], const float (&hcar)[H], int ibase) {
        // every operand is already in registers, so each result is stored as soon as it is
        // computed: no load ever waits behind a store.  `element(ii, ok, fused)` handles one
        // register; the common case (whole row block inside the batch, fused noise) runs it under
        // ONE column mask with no per-element branches, edge blocks and replay mode take the
        // general path.
        float inv_sat_j = 1.0f;
        if constexpr (VS) inv_sat_j = 1.0f / sat_j;
        auto element = [&](int ii, bool ok, auto fused_tag) {
            constexpr bool FUSED = decltype(fused_tag)::value;
            const int r = R0 + ibase + ii;
            const size_t gb = gofs(r);        // uniform
            const unsigned lo = eoff[ii & 3];  // per lane (H and ibase are multiples of 4)
            float n0 = 0.0f, n1 = 0.0f, n0n = 0.0f;
            if constexpr (NOISY) {
                if constexpr (FUSED) {
                    // written in the prologue by this thread or its twin
                    if constexpr (MODE == MODE_MF) {
                        n0 = hcar[ii];  // this step's normal, generated one step ago
                        n0n = a.s.mf.has_next ? lds_noise[(ibase + ii) * NTHREADS + tid] : 0.0f;
                    } else {
                        n0 = lds_noise[(0 * NR + ibase + ii) * NTHREADS + tid];
                        if constexpr (MODE == MODE_DL) n1 = lds_noise[(1 * NR + ibase + ii) * NTHREADS + tid];
                    }
                } else if (ok) {
                    const size_t widx = (size_t)j * a.wld + row0 + erow(r);
                    n0 = a.w0[widx];
                    if constexpr (MODE == MODE_DL) n1 = a.w1[widx];
                    if constexpr (MODE == MODE_MF)
                        if (a.s.mf.has_next) n0n = a.w0n[widx];
                }
            }

            // Adam preconditioning of the feedback term g (MF / Langevin variants)
            auto adam = [&](float g) {
                if constexpr (ADAM) {
                    float m, v;
                    const float out = adam_precondition(a.ad, g, he2[ii], he3[ii], m, v);
                    if (ok) {
                        st_nt(&(a.am + gb)[lo], m);
                        if (a.ad.use_v) st_nt(&(a.av + gb)[lo], v);
                    }
                    return out;
                } else {
                    return g;
                }
            };

            if constexpr (MODE == MODE_DL) {
                float cn, sn;
                dl_update(a.s.dl, he0[ii], he1[ii], hf[0][ii], hf[NA - 1][ii], vj, n0, n1, cn, sn);
                if (ok) {
                    st_nt(&(a.o0 + gb)[lo], cn);
                    st_nt(&(a.o1 + gb)[lo], sn);
                }
            } else if constexpr (MODE == MODE_MF) {
                const MfScalars& k = a.s.mf;
                const float bound = VS ? sat_j : k.S;
                float fq = __builtin_fmaf(k.f_q, hf[0][ii], k.f_v * vj);
                if constexpr (VS) fq *= inv_sat_j;
                const float fb = adam(fq);
                float mun, sgn;
                mf_update(k, he0[ii], he1[ii], fb, n0, mun, sgn);
                if (ok) {
                    st_nt(&(a.st0 + gb)[lo], mun);
                    st_nt(&(a.st1 + gb)[lo], sgn);
                    if (k.has_next) {
                        st_nt(&(a.o0 + gb)[lo], clampf(__builtin_fmaf(k.k_next, n0n, mun), -bound, bound));
                        if constexpr (FUSED) st_nt(&(a.carry + gb)[lo], n0n);  // next step's normal
                    }
                }
            } else if constexpr (MODE == MODE_LANGEVIN) {
                const LvScalars& k = a.s.lv;
                float gq = __builtin_fmaf(k.g_q, hf[0][ii], k.g_v * vj);
                if constexpr (VS) gq *= inv_sat_j;
                const float g = adam(gq);
                const float x = lv_update(k, he0[ii], g, n0, VS ? sat_j : k.S);
                if (ok) st_nt(&(a.o0 + gb)[lo], x);
            } else if constexpr (MODE == MODE_GD) {
                const PpScalars& k = a.s.pp;
                if (ok) st_nt(&(a.o0 + gb)[lo], clampf(__builtin_fmaf(-k.step, hf[0][ii] + vj, he0[ii]), k.lo, k.hi));
            } else if constexpr (MODE == MODE_ADAMPP) {
                const PpScalars& k = a.s.pp;
                const float g = hf[0][ii] + vj;
                if (ok) st_nt(&(a.o0 + gb)[lo], clampf(__builtin_fmaf(-k.step, g / (fabsf(g) + k.eps), he0[ii]), k.lo, k.hi));
            } else if constexpr (MODE == MODE_ASGDPP) {
                const PpScalars& k = a.s.pp;  // eps = 1 - lambd * lr (the decay of the parameter)
                const float g = hf[0][ii] + vj;
                if (ok) st_nt(&(a.o0 + gb)[lo], clampf(__builtin_fmaf(-k.step, g, k.eps * he0[ii]), k.lo, k.hi));
            } else if constexpr (MODE == MODE_AFFINE) {
                const PpScalars& k = a.s.pp;  // step = f_q, eps = f_v
                if (ok) st_nt(&(a.o0 + gb)[lo], __builtin_fmaf(k.step, hf[0][ii], k.eps * vj));
            }
        };
        const bool whole_block = row0 + BM <= a.B;  // wave-uniform
        if (whole_block && (fused || !NOISY)) {
            if (col_ok) {
#pragma unroll
                for (int ii = 0; ii < H; ++ii) element(ii, true, Yes{});
            }
        } else if (fused || !NOISY) {
#pragma unroll
            for (int ii = 0; ii < H; ++ii) element(ii, col_ok && (row0 + erow(R0 + ibase + ii) < a.B), Yes{});
        } else {
#pragma unroll
            for (int ii = 0; ii < H; ++ii) element(ii, col_ok && (row0 + erow(R0 + ibase + ii) < a.B), No{});
        }
    };

    if (producer) {
        // =========================== producer waves ====================================
        if (CCVM_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(CCVM_PRODUCER_PRIO);
        // DMA pieces (1 KiB = one wave instruction) of a tile: 4 per A tile (8 rows x 128 B) then
        // 16 / KS of Q (1024 / (4 BNT) rows x 4 BNT B), dealt round-robin to the NPW producer waves.
        // LDS destination = piece base + lane * 16 (hardware); the A tile's swizzle (chunk c of row
        // r stored at position c ^ ((r >> 1) & 7)) is applied to the per-lane SOURCE address.
        constexpr int NPIECE = 4 * NA + 16 / KS;
        constexpr int PMAX = (NPIECE + NPW - 1) / NPW;
        constexpr int LPR = BNT / 4;           // lanes per Q row
        constexpr int RPP = 64 / LPR;          // Q rows per piece
        const int pw = __builtin_amdgcn_readfirstlane((threadIdx.x - NTHREADS) >> 6);  // 0 .. NPW-1
        const int plane = threadIdx.x & 63;
        // Addressing: wave-uniform 64-bit base (SGPRs, advanced per tile by SALU) + a per-lane
        // 32-bit byte offset that never changes -> no per-tile VALU (every VALU op costs matrix time).
        unsigned voff[PMAX];
        int dst[PMAX];
        bool is_a[PMAX];
        const char* sbase_g[PMAX];
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            const int p = min(pw + NPW * i, NPIECE - 1);  // wave-uniform; a wave without piece i repeats the last
            is_a[i] = p < 4 * NA;
            if (is_a[i]) {
                const int n = p >> 2, g = p & 3;
                const int r = 8 * g + (plane >> 3), pos = plane & 7;
                sbase_g[i] = reinterpret_cast<const char*>(((n == 0) ? a.a0 : a.a1) + (size_t)row0 * ld);
                voff[i] = (unsigned)((r * ld + 4 * (pos ^ ((r >> 1) & 7))) * 4);
                dst[i] = n * A_TILE + g * 256;
            } else {
                const int q = p - 4 * NA;
                sbase_g[i] = reinterpret_cast<const char*>(a.Q + col0);
                voff[i] = (unsigned)(((RPP * q + plane / LPR) * ld + 4 * (plane % LPR)) * 4);
                dst[i] = NA * A_TILE + q * 256;
            }
        }
        auto dma_tile = [&](int kt) {
            if constexpr (ABL & 1) return;
            const int k0 = kt * KT;
            float* sbase = lds + (kt % NSTAGE) * STAGE;
#pragma unroll
            for (int i = 0; i < PMAX; ++i) {
                const char* tb = sbase_g[i] + (is_a[i] ? (size_t)k0 * 4 : (size_t)k0 * ld * 4);  // scalar
                // saddr form: SGPR base + 32-bit lane offset, LDS destination base in M0 (written in
                // the same statement: the compiler does not preserve M0 around asm).  vmcnt is
                // counted by hand in publish().
                const unsigned ldst = (unsigned)(size_t)(lds_void*)(sbase + dst[i]);
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(voff[i]), "s"(tb), "s"(ldst)
                    : "memory");
            }
        };
        // Only real tiles are issued (`issued` of them so far: no duplicate loads of the last tile into
        // unread slots, 12 % of the L2 reads at N = 1000 with a ring of 4).  Tile kt must have landed
        // before the barrier that precedes its first fragment read: with `later` tiles issued after it
        // that is vmcnt(later * PMAX), at most NSTAGE - 2 of them in the steady state.  No lgkmcnt
        // wait: the noise ds_writes are only consumed behind the final barrier.
        int issued = 0;
        auto issue_next = [&]() {
            if (issued < nkt) {  // wave-uniform
                dma_tile(issued);
                ++issued;
            }
        };
        auto publish = [&](int landed) {  // tiles 0 .. landed are in the ring when the barrier opens
            wait_vm_tiles<PMAX, NSTAGE - 2>(issued - 1 - landed);
            if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
        };
#pragma unroll
        for (int kt = 0; kt < NSTAGE; ++kt) issue_next();
        if (gen_noise) {
#pragma unroll
            for (int u = NPRO_C; u < NPRO_C + NPRO_P; ++u) make_noise(u);
        }
        publish(1);  // tiles 0, 1 visible
        // the consumers read tile 0's fragments right after that barrier: slot 0 may only be
        // refilled (with tile NSTAGE) once they are done
        if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nkt; ++t) {
            issue_next();  // tile t + NSTAGE into the slot of tile t, whose fragments are already in registers
            constexpr int NLOOP = NUNIT - NPRO_C - NPRO_P;
            if (gen_noise && t < NLOOP) make_noise(NPRO_C + NPRO_P + t);
            publish(t + 2);  // tile t + 2 visible
        }
        if (gen_noise)
            for (int u = NPRO_C + NPRO_P + min(nkt, NUNIT - NPRO_C - NPRO_P); u < NUNIT; ++u) make_noise(u);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // noise visible to the consumers' epilogue; ring idle
        if constexpr (SHARE) {
            if constexpr (KS == 2) __builtin_amdgcn_s_barrier();  // the consumers' K-half exchange
            __syncthreads();  // the consumers' hand-over is in the ring (fence + barrier)
            float hf[NA][H], he0[H], he1[H], he2[H], he3[H], hcar[H];
            int v = 0;
#pragma unroll
            for (int n = 0; n < NA; ++n)
#pragma unroll
                for (int ii = 0; ii < H; ++ii) hf[n][ii] = hand[(v++) * NTHREADS + tid];
#pragma unroll
            for (int ii = 0; ii < H; ++ii) {
                he0[ii] = hand[(v++) * NTHREADS + tid];
                he1[ii] = HAS_E1 ? hand[(v++) * NTHREADS + tid] : 0.0f;
                he2[ii] = ADAM ? hand[(v++) * NTHREADS + tid] : 0.0f;
                he3[ii] = ADAM ? hand[(v++) * NTHREADS + tid] : 0.0f;
                hcar[ii] = (MODE == MODE_MF) ? hand[(v++) * NTHREADS + tid] : 0.0f;
            }
            run_epilogue(hf, he0, he1, he2, he3, hcar, H);
        }
        return;
    }

    // ============================= consumer waves =======================================
    // ---- epilogue operands, fetched now so their latency hides under the whole GEMM -----
    // e0/e1: old state at the element (DL: c, s; MF: mu, sigma; others: x); e2/e3: Adam moments;
    // ecar: MF's normals of this step.  Index i is accumulator register R0 + i.  Rows >= B and
    // columns >= N are inside the padded arrays.
    float e0[NR], e1[NR], e2[NR], e3[NR], ecar[NR];
    {
        const float* p0 = (MODE == MODE_MF) ? a.st0 : a.a0;
        const float* p1 = (MODE == MODE_MF) ? a.st1 : a.a1;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = R0 + i;  // R0 is wave-uniform: addresses stay scalar + lane offset
            if constexpr (HAS_E0) e0[i] = (p0 + gofs(r))[eoff[i & 3]];
            if constexpr (HAS_E1) e1[i] = (p1 + gofs(r))[eoff[i & 3]];
            if constexpr (MODE == MODE_MF) ecar[i] = a.replay ? 0.0f : (a.carry + gofs(r))[eoff[i & 3]];
        }
    }
    // fragment read offsets inside a stage (A: swizzled chunk position)
    const int sw = (l31 >> 1) & 7;
    const int fa = l31 * KT;
    const int fb = NA * A_TILE + (16 * half + kh * NM) * BNT + 32 * cs + l31;
    struct FragsT {
        f32x4 a[NA][NQA];  // k-steps kh*NM .. kh*NM + NM - 1 of this lane-half, 4 per b128
        float b[NM];
    };
    // per-stage fragment base pointers: inside the loop every LDS address is base + immediate
    const float* stA[NSTAGE];
    const float* stB[NSTAGE];
#pragma unroll
    for (int st = 0; st < NSTAGE; ++st) {
        stA[st] = lds + st * STAGE + fa;
        stB[st] = lds + st * STAGE + fb;
    }
    // group g of a tile's fragments (NG groups): keeps the LDS queue shallow
    auto read_part = [&](FragsT& f, int st, int g) {  // st is a compile-time constant at every call
        if (g < NQA * NA) {
            const int n = g / NQA, ql = g % NQA;
            f.a[n][ql] = *reinterpret_cast<const f32x4*>(stA[st] + n * A_TILE + 4 * ((4 * half + kh * NQA + ql) ^ sw));
        }
        f.b[2 * g] = stB[st][(2 * g) * BNT];
        f.b[2 * g + 1] = stB[st][(2 * g + 1) * BNT];
    };

    f32x16 acc[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    auto mfma_range = [&](const FragsT& f, int m0, int m1) {
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

    FragsT f0, f1;
    if (CCVM_CONSUMER_PRIO) __builtin_amdgcn_s_setprio(CCVM_CONSUMER_PRIO);
    if (gen_noise) {  // this thread's share of the noise, under the first tiles' flight time
#pragma unroll
        for (int u = 0; u < NPRO_C; ++u) make_noise(u);
    }
    __syncthreads();  // tiles 0, 1 are in the ring (fence + barrier: LDS reads stay below it)
#pragma unroll
    for (int g = 0; g < NG; ++g) read_part(f0, 0, g);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the loop starts with settled LDS counters
    if constexpr (!(ABL & 32)) __syncthreads();  // slot 0 may now be refilled

    // iteration t: the NM * NA MFMAs of tile t (fragments in registers) with the fragment reads of
    // tile t+1 in their issue gaps -- slot by slot (2 k-steps each), order pinned, and none in the
    // last two slots so no read latency is exposed at the barrier.
    unsigned long long c_work = 0, c_bar = 0, c_last = 0;
    auto c_iteration = [&](const FragsT& cur, FragsT& nxt, auto stage_tag) {
        constexpr int rstage = decltype(stage_tag)::value;
        constexpr int NSLOT = NM / 2;
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 4)) {
                // NG read groups over slots 0 .. NSLOT-3: slots 0,1 carry two
                if (sl < 2) {
                    if (2 * sl < NG) read_part(nxt, rstage, 2 * sl);
                    if (2 * sl + 1 < NG) read_part(nxt, rstage, 2 * sl + 1);
                }
                else if (sl + 2 < NG) read_part(nxt, rstage, sl + 2);
            }
            mfma_range(cur, 2 * sl, 2 * sl + 2);
#pragma unroll
            for (int i_ = 0; i_ < 2 * NA; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS reads
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL & 128) {
            const unsigned long long s1 = stamp();
            __syncthreads();
            const unsigned long long s2 = stamp();
            c_bar += stamp_delta(s1, s2);
            c_work += stamp_delta(c_last, s1);
            c_last = s2;
        } else if constexpr (!(ABL & 32)) __syncthreads();
    };
    {
        if constexpr (ABL & 128) c_last = stamp();
        // iteration t reads the fragments of tile t+1 from ring stage (t+1) % NSTAGE: unrolled by the ring
        // depth so that every stage index is a compile-time constant (LDS addresses = base + immediate);
        // the two fragment sets alternate (NSTAGE is even)
        auto iteration = [&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            using Stage = std::integral_constant<int, (i + 1) % NSTAGE>;
            if constexpr (i % 2 == 0) c_iteration(f0, f1, Stage{});
            else c_iteration(f1, f0, Stage{});
        };
        int t = 0;
        for (; t + NSTAGE - 1 < nkt; t += NSTAGE) unroll_indices(iteration, std::make_integer_sequence<int, NSTAGE>{});
        unroll_indices([&](auto i_tag) {  // the remaining nkt % NSTAGE tiles
            if (t < nkt) {
                iteration(i_tag);
                ++t;
            }
        }, std::make_integer_sequence<int, NSTAGE - 1>{});
    }
    if constexpr (ADAM) {
        // Adam moments: fetched here, not at kernel start (holding 2 x NR more registers through the
        // main loop spills at two waves per SIMD); the latency hides under the final barrier
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = R0 + i;
            e2[i] = (a.am + gofs(r))[eoff[i & 3]];
            e3[i] = a.ad.use_v ? (a.av + gofs(r))[eoff[i & 3]] : 0.0f;
        }
    }
    __syncthreads();  // producers' noise is complete, no DMA in flight: the ring is free
    if constexpr (ABL & 128) {
        if (threadIdx.x == 0) {
            unsigned long long* d = a.dbg + (size_t)blockIdx.x * 8;
            d[4] = c_work; d[5] = c_bar;
        }
    }

    // ---- KS = 2: swap halves so this wave holds the full K sum of registers R0 .. R0 + 7 ----
    // fin[n][i] = total of accumulator register R0 + i
    float fin[NA][NR];
    if constexpr (KS == 1) {
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
            for (int i = 0; i < NR; ++i) fin[n][i] = acc[n][i];
    } else if constexpr (KS == 4) {
        // every wave leaves its 16 partial registers per accumulator: [source quarter][n][register quad][lane][4]
        f32x4* xq = reinterpret_cast<f32x4*>(lds);
        auto quad = [&](int src, int n, int q) { return ((src * NA + n) * 4 + q) * 64 + lane; };
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[n][4 * q + i];
                xq[quad(kh, n, q)] = v;
            }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            const f32x4 p0 = xq[quad(0, n, kh)], p1 = xq[quad(1, n, kh)], p2 = xq[quad(2, n, kh)], p3 = xq[quad(3, n, kh)];
#pragma unroll
            for (int i = 0; i < 4; ++i) fin[n][i] = ((p0[i] + p1[i]) + p2[i]) + p3[i];  // K quarters in order
        }
    } else {
        float* xb = lds;  // [kh][cs][n][8][64]
        auto slot = [&](int k_, int n, int i) { return (((k_ * 2 + cs) * NA + n) * 8 + i) * 64 + lane; };
        if (kh == 0) {
#pragma unroll
            for (int n = 0; n < NA; ++n)
#pragma unroll
                for (int i = 0; i < 8; ++i) xb[slot(0, n, i)] = acc[n][8 + i];  // give rows 8..15
        } else {
#pragma unroll
            for (int n = 0; n < NA; ++n)
#pragma unroll
                for (int i = 0; i < 8; ++i) xb[slot(1, n, i)] = acc[n][i];      // give rows 0..7
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int n = 0; n < NA; ++n)
#pragma unroll
                for (int i = 0; i < 8; ++i) fin[n][i] = acc[n][i] + xb[slot(1, n, i)];       // k half 0 + k half 1
        } else {
#pragma unroll
            for (int n = 0; n < NA; ++n)
#pragma unroll
                for (int i = 0; i < 8; ++i) fin[n][i] = xb[slot(0, n, i)] + acc[n][8 + i];   // k half 0 + k half 1
        }
    }

    // the affine input map, folded:  (x*scale + shift) @ Q = scale * (x @ Q) + shift * colsum(Q)
#pragma unroll
    for (int n = 0; n < NA; ++n)
#pragma unroll
        for (int i = 0; i < NR; ++i) fin[n][i] = __builtin_fmaf(a.in_scale, fin[n][i], shift_j);

    if constexpr (ABL & 16) {  // ablation: keep the accumulators live, skip the real epilogue
        float sum = vj;
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
            for (int i = 0; i < NR; ++i) sum += fin[n][i] + e0[i] + (HAS_E1 ? e1[i] : 0.0f);
        if (sum == 123.456f) a.o0[0] = sum;
        return;
    }

    // ---- epilogue in the accumulator layout ---------------------------------------
    if constexpr (MODE == MODE_ENERGY) {
        // partial over this wave's 32 columns of (1/2 (x@Q)[b,j] + V[j]) * x[b,j]
        float part[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float p = col_ok ? (0.5f * fin[0][i] + vj) * e0[i] : 0.0f;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) p += __shfl_xor(p, off, 64);
            part[i] = p;
        }
        if (l31 == 0) {
            // o0: [column strips of 32][rows_pad] partial sums
            const int strip = cb * (BNT / 32) + cs;
            const int rows_pad = a.nrb * BM;
#pragma unroll
            for (int i = 0; i < NR; ++i) a.o0[(size_t)strip * rows_pad + row0 + erow(R0 + i)] = part[i];
        }
        return;
    } else {
        float hf[NA][H], he0[H], he1[H], he2[H], he3[H], hcar[H];
#pragma unroll
        for (int ii = 0; ii < H; ++ii) {
#pragma unroll
            for (int n = 0; n < NA; ++n) hf[n][ii] = fin[n][ii];
            he0[ii] = HAS_E0 ? e0[ii] : 0.0f;
            he1[ii] = HAS_E1 ? e1[ii] : 0.0f;
            he2[ii] = ADAM ? e2[ii] : 0.0f;
            he3[ii] = ADAM ? e3[ii] : 0.0f;
            hcar[ii] = (MODE == MODE_MF) ? ecar[ii] : 0.0f;
        }
        if constexpr (SHARE) {
            // registers R0 + H .. R0 + 2H - 1 and their operands go to the producer twin (same tid
            // mapping), in the order it reads them back
            int v = 0;
#pragma unroll
            for (int n = 0; n < NA; ++n)
#pragma unroll
                for (int ii = 0; ii < H; ++ii) hand[(v++) * NTHREADS + tid] = fin[n][H + ii];
#pragma unroll
            for (int ii = 0; ii < H; ++ii) {
                hand[(v++) * NTHREADS + tid] = e0[H + ii];
                if constexpr (HAS_E1) hand[(v++) * NTHREADS + tid] = e1[H + ii];
                if constexpr (ADAM) {
                    hand[(v++) * NTHREADS + tid] = e2[H + ii];
                    hand[(v++) * NTHREADS + tid] = e3[H + ii];
                }
                if constexpr (MODE == MODE_MF) hand[(v++) * NTHREADS + tid] = ecar[H + ii];
            }
            __syncthreads();
        }
        run_epilogue(hf, he0, he1, he2, he3, hcar, 0);
    }
}

// 32 x 32 split-K tiles (KS = 4) of the solver steps: instantiated in ccvm_tile4_{dl,mf,lv}.hip
void tile4_launch_dl(const StepArgs& a, int grid, hipStream_t st);
void tile4_launch_mf(const StepArgs& a, int grid, bool adam, bool per_variable_s, hipStream_t st);
void tile4_launch_lv(const StepArgs& a, int grid, bool adam, bool per_variable_s, hipStream_t st);
template <int MODE>
void launch_tile4(const StepArgs& a, int grid, bool adam, bool vs, hipStream_t st) {
    if constexpr (MODE == MODE_DL) {
        hipLaunchKernelGGL((step_kernel<MODE, false, 0, 4>), dim3(grid), dim3(WG_THREADS), 0, st, a);
    } else {
        if (adam) {
            if (vs) hipLaunchKernelGGL((step_kernel<MODE, true, 0, 4, true>), dim3(grid), dim3(WG_THREADS), 0, st, a);
            else hipLaunchKernelGGL((step_kernel<MODE, true, 0, 4>), dim3(grid), dim3(WG_THREADS), 0, st, a);
        } else {
            if (vs) hipLaunchKernelGGL((step_kernel<MODE, false, 0, 4, true>), dim3(grid), dim3(WG_THREADS), 0, st, a);
            else hipLaunchKernelGGL((step_kernel<MODE, false, 0, 4>), dim3(grid), dim3(WG_THREADS), 0, st, a);
        }
    }
}

#ifndef CCVM_STEP_KERNEL_ONLY  // the instantiation units ccvm_tile4_*.hip take the step kernel only
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

// Per-variable saturation: Qs[k][j] = Q[k][j] / S_k (the input map's 1 / S_k folded into the rows).
__global__ void scale_rows_kernel(const float* __restrict__ Q, const float* __restrict__ s_cols,
                                  float* __restrict__ Qs, int N, int ld) {
    const size_t total = (size_t)ld * ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / ld);
        Qs[i] = (k < N) ? Q[i] / s_cols[k] : 0.0f;
    }
}

// x = clamp(x, -S_j, S_j) / y = 0.5 * x / S_j * (u - l) + 0.5 * (u + l) with a per-variable S.
__global__ void clamp_cols_kernel(float* x, int B, int N, int ld, const float* __restrict__ s_cols) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        float* p = x + (size_t)r * ld + c;
        *p = clampf(*p, -s_cols[c], s_cols[c]);
    }
}

__global__ void change_variables_cols_kernel(const float* x, float* y, int B, int N, int ld,
                                             const float* __restrict__ s_cols, float ul, float half_up) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        const size_t idx = (size_t)r * ld + c;
        y[idx] = change_var(x[idx], s_cols[c], ul, half_up);
    }
}

// The two above with one bound / saturation per trajectory AND variable (pitched [rows][ld] arrays):
// the reference passes any non-1-D tensor S straight through to the elementwise ops
// (dl_solver.py:843-848), and its clamp takes tensor bounds (torch.clamp(c, lower, upper)).
__global__ void clamp_full_kernel(float* x, int B, int N, int ld, const float* __restrict__ lo,
                                  const float* __restrict__ hi) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        const size_t idx = (size_t)r * ld + c;
        x[idx] = clampf(x[idx], lo[idx], hi[idx]);
    }
}

__global__ void change_variables_full_kernel(const float* x, float* y, int B, int N, int ld,
                                             const float* __restrict__ s_full, float ul, float half_up) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        const size_t idx = (size_t)r * ld + c;
        y[idx] = change_var(x[idx], s_full[idx], ul, half_up);
    }
}

// ---- one saturation per trajectory AND variable inside the loop (MF / Langevin / pumped Langevin) -------------
// The reference passes a 2-D tensor S straight through (mf_solver.py:834-839 and the same lines of the other
// solvers), so 1 / S_bk sits inside the GEMM's input map per ELEMENT and cannot be folded into Q (per-variable
// S: the row-scaled copy Qs) or into a scalar.  Rare input, composed path ("support on a
// slower path or reject loudly"): per step the MODE_AFFINE launch of step_kernel contracts the pre-scaled input
// xs = x / S, and one elementwise kernel does noise, Adam, update, clamp and the next xs.  Same pinned update
// helpers and noise definition as the fused kernels.
__global__ void fulls_scale_kernel(const float* __restrict__ x, const float* __restrict__ s_full, float* xs, int B,
                                   int N, int ld) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        const size_t idx = (size_t)r * ld + c;
        xs[idx] = x[idx] / s_full[idx];
    }
}

struct FullSArgs {
    float* x0;            // Langevin: c; MF: mu (in place)
    float* x1;            // MF: sigma (in place)
    float* mt;            // MF: clamped measured amplitude fed to the next step (in/out)
    float* xs;            // GEMM input of the next step: x / S (Langevin) or mt / S (MF)
    const float* y;       // f_q * (A(xs) @ Q) + f_v * V of this step (MODE_AFFINE), scalars built for S = 1
    const float* s_full;  // pitched saturation
    float* am;
    float* av;
    const float* w0;      // REPLAY: this step's [N][B] block
    const float* w0n;     // REPLAY (MF): next step's block
    uint64_t seed;
    int64_t row_offset;
    int step, B, N, ld;
    AdamScalars ad;
    int adam;
};

__device__ __forceinline__ float fulls_adam(const FullSArgs& a, float g, size_t idx) {
    if (!a.adam) return g;
    float m, v;
    const float out = adam_precondition(a.ad, g, a.am[idx], a.ad.use_v ? a.av[idx] : 0.0f, m, v);
    a.am[idx] = m;
    if (a.ad.use_v) a.av[idx] = v;
    return out;
}

__global__ void fulls_langevin_update_kernel(const FullSArgs a, const LvScalars k) {
    const size_t total = (size_t)a.B * a.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / a.N), j = (int)(i - (size_t)b * a.N);
        const size_t idx = (size_t)b * a.ld + j;
        const float S = a.s_full[idx];
        const float n0 = a.w0 ? a.w0[(size_t)j * a.B + b] : normal_single(a.seed, a.row_offset + b, a.step, j);
        const float g = fulls_adam(a, a.y[idx] / S, idx);
        const float x = lv_update(k, a.x0[idx], g, n0, S);
        a.x0[idx] = x;
        a.xs[idx] = x / S;
    }
}

// MF first step of a chunk: mt = clamp(mu + k0 W_step0, -S, S) (mf_solver.py:551-554), xs = mt / S.
__global__ void fulls_mf_prepare_kernel(const FullSArgs a, float k0) {
    const size_t total = (size_t)a.B * a.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / a.N), j = (int)(i - (size_t)b * a.N);
        const size_t idx = (size_t)b * a.ld + j;
        const float S = a.s_full[idx];
        const float n0 = a.w0 ? a.w0[(size_t)j * a.B + b] : normal_single(a.seed, a.row_offset + b, a.step, j);
        const float mt = clampf(__builtin_fmaf(k0, n0, a.x0[idx]), -S, S);
        a.mt[idx] = mt;
        a.xs[idx] = mt / S;
    }
}

__global__ void fulls_mf_update_kernel(const FullSArgs a, const MfScalars k) {
    const size_t total = (size_t)a.B * a.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / a.N), j = (int)(i - (size_t)b * a.N);
        const size_t idx = (size_t)b * a.ld + j;
        const float S = a.s_full[idx];
        const float n0 = a.w0 ? a.w0[(size_t)j * a.B + b] : normal_single(a.seed, a.row_offset + b, a.step, j);
        const float fb = fulls_adam(a, a.y[idx] / S, idx);
        float mun, sgn;
        mf_update(k, a.x0[idx], a.x1[idx], fb, n0, mun, sgn);
        a.x0[idx] = mun;
        a.x1[idx] = sgn;
        if (k.has_next) {  // the last step's input is what mu_tilde_out returns: no new measurement after it
            const float n1 = a.w0n ? a.w0n[(size_t)j * a.B + b] : normal_single(a.seed, a.row_offset + b, a.step + 1, j);
            const float mt = clampf(__builtin_fmaf(k.k_next, n1, mun), -S, S);
            a.mt[idx] = mt;
            a.xs[idx] = mt / S;
        }
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
        y[idx] = change_var(x[idx], S, ul, half_up);
    }
}

// Measured amplitude of step `step` from the current mu (start of an MF chunk):
//   mu_tilde_c = clamp(mu + k * W, -S, S)    (reference mf_solver.py:551-554)
// Also seeds the carry buffer with this step's normals (fused mode).
__global__ void mf_prepare_kernel(const float* mu, float* out, float* carry, int B, int N, int ld,
                                  float k, float S, const float* s_cols, uint64_t seed, int64_t row_offset,
                                  int step, const float* w0, int wld) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / N), j = (int)(i - (size_t)b * N);
        const float n0 = w0 ? w0[(size_t)j * wld + b] : normal_single(seed, row_offset + b, step, j);
        const size_t idx = (size_t)b * ld + j;
        const float bound = s_cols ? s_cols[j] : S;
        out[idx] = clampf(mu[idx] + k * n0, -bound, bound);
        if (!w0) carry[idx] = n0;
    }
}

// w1 != NULL: the DL pair (W_c, W_s) per element; w1 == NULL: the one-stream normal (rows paired).
__global__ void philox_fill_kernel(uint64_t seed, int64_t row_offset, int step, int B, int N,
                                   float* w0, float* w1) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / B), b = (int)(i - (size_t)j * B);
        if (w1) {
            const NormalPair p = normal_pair(seed, row_offset + b, step, j);
            w0[i] = p.n0;
            w1[i] = p.n1;
        } else {
            w0[i] = normal_single(seed, row_offset + b, step, j);
        }
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

// Finalize, step 1 (one pass over the state): optional fit_to_constraints clamp IN PLACE on the state
// (dl_solver.py:567: scalar bounds, or -S .. S with a per-variable / per-element saturation) and the change of
// variables x = 0.5 * y / S * (u - l) + 0.5 * (u + l) (dl_solver.py:219-235, same operation order as
// change_variables_kernel) into `x`, which may alias the state.  do_cv == 0 copies (x != state) or
// leaves the state as the variables.
__global__ void finalize_prepare_kernel(float* state, float* x, int B, int N, int ld, int do_clamp, float clo,
                                        float chi, int do_cv, float S, const float* __restrict__ s_cols,
                                        const float* __restrict__ s_full, float ul, float half_up) {
    const size_t total = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / N), c = (int)(i - (size_t)r * N);
        const size_t idx = (size_t)r * ld + c;
        float y = state[idx];
        const float sat = s_full ? s_full[idx] : s_cols ? s_cols[c] : S;
        if (do_clamp) {
            y = (s_cols || s_full) ? clampf(y, -sat, sat) : clampf(y, clo, chi);
            state[idx] = y;
        }
        if (do_cv) y = change_var(y, sat, ul, half_up);
        if (do_cv || x != state) x[idx] = y;
    }
}

// Success statistics of a batch (solution.py:65-146), one workgroup, fixed-order reductions:
//   found_b = -obj_b;  best = max_b found_b;  gap_b = (optimal - found_b) * 100 / |found_b|   (fp32, as
//   torch evaluates it for a float32 tensor and Python-float scalars);  within[k] = #{b : gap_b <= thr_k},
//   thr = 0.1, 1, 2, 3, 4, 5, 10 (percent).  NaN objective values count in no threshold and make
//   `best` NaN, like torch.max.
struct ObjectiveStats {
    float best_objective_value;
    int within[7];
    int rows;
    int nonfinite;  // rows whose objective value is NaN or infinite (a diverged run)
};
__global__ __launch_bounds__(1024) void objective_stats_kernel(const float* __restrict__ obj, int B, float optimal,
                                                                ObjectiveStats* out) {
    __shared__ float s_best[1024];
    __shared__ int s_cnt[9][1024];
    const float thr[7] = {0.1f, 1.0f, 2.0f, 3.0f, 4.0f, 5.0f, 10.0f};
    float best = -INFINITY;
    int cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // 7 thresholds, NaN seen, non-finite rows
    for (int b = threadIdx.x; b < B; b += 1024) {
        const float found = -obj[b];
        const float gap = (optimal - found) * 100.0f / fabsf(found);
#pragma unroll
        for (int k = 0; k < 7; ++k) cnt[k] += (gap <= thr[k]) ? 1 : 0;
        if (found != found) cnt[7] = 1;
        else best = fmaxf(best, found);
        if (!(fabsf(found) <= 3.402823466e38f)) cnt[8] += 1;
    }
    s_best[threadIdx.x] = best;
#pragma unroll
    for (int k = 0; k < 9; ++k) s_cnt[k][threadIdx.x] = cnt[k];
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_best[threadIdx.x] = fmaxf(s_best[threadIdx.x], s_best[threadIdx.x + w]);
#pragma unroll
            for (int k = 0; k < 9; ++k) s_cnt[k][threadIdx.x] += s_cnt[k][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out->best_objective_value = s_cnt[7][0] ? __builtin_nanf("") : s_best[0];
#pragma unroll
        for (int k = 0; k < 7; ++k) out->within[k] = s_cnt[k][0];
        out->rows = B;
        out->nonfinite = s_cnt[8][0];
    }
}

// Column sums of Q in two deterministic passes: part[s][j] over row slice s, then a fixed-order
// sum over slices (no atomics: results must not depend on arrival order).
constexpr int QSUM_SLICES = 32;
__global__ void qsum_partial_kernel(const float* __restrict__ Q, int N, int ld, float* __restrict__ part) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (j >= ld) return;
    const int per = (N + QSUM_SLICES - 1) / QSUM_SLICES;
    const int k0 = s * per, k1 = min(N, k0 + per);
    float acc = 0.0f;
    for (int k = k0; k < k1; ++k) acc += Q[(size_t)k * ld + j];
    part[(size_t)s * ld + j] = acc;
}
__global__ void qsum_final_kernel(const float* __restrict__ part, int ld, float* __restrict__ qsum) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    float acc = 0.0f;
    for (int s = 0; s < QSUM_SLICES; ++s) acc += part[(size_t)s * ld + j];
    qsum[j] = acc;
}

// LBFGS post-processor, one iteration of one row per workgroup (torch.optim.LBFGS(lr, max_iter=1) from a
// fresh state is one steepest-descent step of length lr * min(1, 1 / |g|_1); it returns without moving
// when max|g| <= 1e-7 (tolerance_grad) or g.g < 1e-9 (tolerance_change)):
//   x <- clamp(x - lr * min(1, 1 / |g|_1) * g, lo, hi).   Fixed-order reductions: deterministic.
__global__ __launch_bounds__(256) void lbfgs_row_kernel(float* x, const float* g, int N, int ld, float lr,
                                                        float lo, float hi) {
    __shared__ float red[3][256];
    const size_t row = (size_t)blockIdx.x * ld;
    float l1 = 0.0f, mx = 0.0f, sq = 0.0f;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float v = g[row + j];
        l1 += fabsf(v);
        mx = fmaxf(mx, fabsf(v));
        sq = __builtin_fmaf(v, v, sq);
    }
    red[0][threadIdx.x] = l1; red[1][threadIdx.x] = mx; red[2][threadIdx.x] = sq;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + w]);
            red[2][threadIdx.x] += red[2][threadIdx.x + w];
        }
        __syncthreads();
    }
    l1 = red[0][0]; mx = red[1][0]; sq = red[2][0];
    const bool moves = (mx > 1e-7f) && (sq >= 1e-9f);
    const float t = moves ? fminf(1.0f, 1.0f / l1) * lr : 0.0f;
    for (int j = threadIdx.x; j < N; j += 256) x[row + j] = clampf(__builtin_fmaf(-t, g[row + j], x[row + j]), lo, hi);
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

#endif  // CCVM_STEP_KERNEL_ONLY

}  // namespace ccvm
