// Column-slab persistent kernel for SMALL batches above N = 256: a whole chunk of time steps in ONE launch with
// the coupling matrix resident in the register files of (up to) the whole chip.
//
// Why: the tile kernel's unit is a 32-row MFMA tile, so a step of B <= 32 rows at N = 1000 runs on 16 of the
// 256 CUs and costs what a step of 512 rows costs (18 us: every workgroup streams its K x 64 panel of Q from L2
// and runs the full K loop); the column-cluster kernel's unit is 32-48 rows x 64 columns.  With few rows the
// contraction x @ Q is a batch of matrix-VECTOR products: what matters is that Q (4 N^2 bytes, 4 MB at N = 1000,
// 16 MB at N = 2000) is not moved at all and that every CU takes part.
//
// Here a CLUSTER of G = ceil(N / C) workgroups owns R = 4 RG batch rows for the whole launch (<= 4096 steps).
// Member m owns C = 4, 8, 16 or 32 output columns: it keeps the K x C slab of Q in the REGISTERS of its four
// waves (NQ = K C / 256 VGPRs per lane, loaded once per launch -- the 128 MB of register files are the largest
// and the only free-to-read memory on the chip; N = 1000 spread over 250 members is 16 registers per lane) and
// owns the state of its R x C elements in registers.  Per step it needs the cluster's whole GEMM input (R x K;
// DL: two planes), which travels as the column-cluster kernel's 8-byte {value, tag} packets (flag in data, sc1
// write-through stores, sc1 loads, ping-pong buffers, bounded waits: ccvm_cluster.h and MI355X_MICROARCH.md,
// "visibility" -- the hand-off is store -> load, nothing drains or polls a counter).
//
// Contraction: v_mfma_f32_4x4x1_16B_f32, sixteen independent 4 x 4 outer products per instruction.  The K range
// is split over the four waves (one per SIMD) and, inside a wave, over the instruction's blocks: block
// b = kr * CGRP + cg handles k residue class kr (of KRES = 16 / CGRP) and column group cg (of CGRP = C / 4); the four
// rows of a row group are the A operand, broadcast over the column groups (CBSZ / ABID, as in ccvm_persist.h:
// one ds_read_b32 of the staged input feeds CGRP MFMAs).  The matvec rows are then reduced at WAVEFRONT level
// (log2 KRES butterfly steps over the k residues) and over the four waves through LDS (4 x R x C floats); the
// owners of the elements (2 rows x 1 column per lane) add the four partial sums in a fixed order and run the
// pinned update arithmetic of ccvm_common.h.  Work per member and step: R C K MACs at the fp32 matrix rate, R K
// packets fetched, R C packets published -- at R = 4, N = 1000 that is 32 KB in, 128 bytes out and 128 cycles of
// MFMA: the step is bound by the hand-off latency across the chip (~1 us, MI355X_MICROARCH.md price list:
// "handoff-1to1", "allgather"), not by any bandwidth.
//
// Exchange layout: [cluster][plane][row group][k][4 rows] packets, so a member's publish region per row group
// is one contiguous run of C x 32 bytes (C = 4: exactly one 128-byte line) written by 16-byte stores
// {x_row, tag, x_row+1, tag}, and a reader's 16-byte loads are contiguous over a whole (plane, row group) block.
// The blocks of a step stream through a ring of two LDS slots: while the waves contract block u, the loads of blocks
// u + 1 and u + 2 are in flight (one barrier per block), so the rows per cluster are bounded by the owners' registers
// (two pairs of rows per lane: R C <= 1024), not by LDS.
//
// Placement: a cluster's members are confined to the smallest group of XCDs that holds them (1, 2, 4 or all 8; blocks
// b, b + 8, ... share an XCD: speed only): inside one XCD the exchange stays in that L2 (a hand-off round trip of ~1000
// cycles against ~3500 across the fabric); N > 1024 needs 63 members of 32 columns = two XCDs.  Workgroups are
// dispatched in order and the whole grid is resident (one workgroup per CU); every spin is bounded all the same
// (status word + LDS flag, as in ccvm_cluster.h).
//
// Same noise definition (global row, column, step), folded affine input map and pinned update arithmetic as the
// other three kernel families; only the summation order of the contraction differs.
#pragma once
#include "ccvm_persist.h"

namespace ccvm {

constexpr int SL_THREADS = 256;          // four waves, one per SIMD
constexpr int SL_NW = 4;
constexpr unsigned SL_XE = 8;            // bytes per exchanged element: {value, tag}
constexpr int SL_XS_FLOATS = 16384;      // staged GEMM input: a ring of two (plane, row group) blocks of K x 4 floats (64 KB)
constexpr int SL_RED_FLOATS = 8192;      // partial sums of the four waves: 4 x planes x RG x C x 4 floats (32 KB)
constexpr int SL_MIN_N = 257, SL_MAX_N = 2048;
constexpr int SL_MAX_RC = 256;           // RG x C: a lane owns up to two pairs of rows at one column (512 pairs per member)
constexpr int SL_MAX_B = 512;            // batches the path is ever considered for (workspace sizing)
// (the bound of a wait: SlabArgs::spin_limit, ticks of the 100 MHz reference clock -- ccvm_abi.hip: spin_ticks)

struct SlabArgs {
    const float* Q;      // [ld][ld] (the row-scaled copy with a per-variable saturation)
    const float* V;
    const float* qsum;
    float* x0;           // DL, Langevin: c;  MF: mu   (pitched, in/out; owner-only data)
    float* x1;           // DL: s;  MF: sigma
    float* xt;           // MF: measured amplitude fed to the LAST step of this launch (out, may be NULL)
    float* xb0;          // exchange buffers: GEMM input of even / odd steps, [cluster][plane][RG][K][4 rows] packets;
    float* xb1;          //   before the call: zero, tags of the columns nobody owns (k >= G C) 0xFFFFFFFF (slab_init_kernel)
    float* am;           // Adam moments (in/out)
    float* av;
    const float* table;  // [nsteps][TABLE_WORDS] schedule rows
    const float* w0;     // REPLAY noise for the chunk: [nsteps][N][B]
    const float* w1;
    unsigned* status;    // 0 = ok; set to 1 when a bounded spin gave up
    unsigned long long* dbg;  // ablation stamps only: [grid][16]
    uint64_t seed;
    int64_t row_offset;
    int step0, nsteps;
    int replay, adam;
    int B, N, ld;
    int wld;             // REPLAY: pitch of the noise blocks (ccvm_noise::w_ld; >= B)
    int nclusters, G, RG;  // clusters of G members; a cluster owns 4 RG batch rows
    int span;            // XCDs a cluster's members are confined to (1, 2, 4, ... nxcd): speed only
    int nxcd;            // XCDs of the device (blocks b and b + nxcd share one)
    int delay_fabric;    // x 64 cycles: what every wave of a cluster that spans XCDs waits before its first loads of a step
    int delay_fixed;     // 1: no calibration of that delay (CCVM_AMD_SLAB_DELAY given)
    int drop;            // fault injection (tests only): this many workgroups are left out of the launch
    unsigned spin_limit; // how long a wave retries before it gives up a wait, in ticks of s_memrealtime (100 MHz; ccvm_abi.hip: spin_ticks)
    float in_scale, in_shift;
    float k_first;       // MF: sqrt(1 / (4 j_step0)) / sqrt(dt)
    float S;             // MF: clamp of the measured amplitude
    const float* s_cols; // per-variable saturation S_j (length ld) or NULL
    AdamConsts ad;
};

// Ablation bits for tools/slab_ablate.hip (0 in the product; timing only, results are wrong): 1 no MFMA, 2 no noise,
// 4 no tag checks, 64 s_memtime stamps of wave 0 into a.dbg[block][0..7]: noise, wait for the first unit, rest of the
// fetch + staging, wait at B1, contraction + reductions, wait at B2, update + publish, retry rounds.
#ifndef CCVM_SLAB_ABL
#define CCVM_SLAB_ABL 0
#endif
#ifndef CCVM_SL_SLEEP
#define CCVM_SL_SLEEP 2
#endif
#ifndef CCVM_SL_CALIBRATE
#define CCVM_SL_CALIBRATE 1
#endif
#ifndef CCVM_SL_DELAY
#define CCVM_SL_DELAY 12   // x 64 cycles: what a wave without owners sleeps before its first loads of a step
#endif

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));

constexpr int sl_log2(int x) { return x <= 1 ? 0 : 1 + sl_log2(x / 2); }

// Butterfly steps of the wavefront-level reduction (every lane ends with the sum; probed: tools/permlane_probe.hip).
// Inside a row of 16 lanes: DPP row rotations fused into the add; across rows: gfx950's v_permlane16/32_swap (the
// second operand goes through an empty asm: given the same SSA value twice hipcc 7.2 assumes both results equal).
template <int ROR>
__device__ __forceinline__ float sl_add_ror(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROR, 0xf, 0xf, false));
}
__device__ __forceinline__ float sl_add_swap16(float v) {
    unsigned x = __float_as_uint(v), y = x;
    asm volatile("" : "+v"(y));
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sl_add_swap32(float v) {
    unsigned x = __float_as_uint(v), y = x;
    asm volatile("" : "+v"(y));
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// CGRP = C / 4 column groups per member (1, 2, 4, 8); NQ = Q registers per lane; K = 64 NQ / CGRP >= N rounded up to C
// CAL: the launch calibrates its fetch delay (clusters that span XCDs, launches of 512 steps or more); a template
// parameter because the calibration's state costs the other launches 2-9 % when it is merely branched around
// (same-box A/B: DL N = 300 B = 8 2.06 -> 2.25 us per step)
template <int MODE, int CGRP, int NQ, bool CAL>
__global__ __launch_bounds__(SL_THREADS) void slab_kernel(const SlabArgs a) {
    static_assert(MODE == MODE_DL || MODE == MODE_MF || MODE == MODE_LANGEVIN, "slab kernel: solver loops only");
    static_assert(CGRP == 1 || CGRP == 2 || CGRP == 4 || CGRP == 8, "column groups per member");
    static_assert(NQ % CGRP == 0 && NQ <= 256, "Q registers per lane (one wave per SIMD: 512 registers)");
    constexpr int C = 4 * CGRP;
    constexpr int NA = NQ / CGRP;            // A registers per wave and block (16 k each)
    constexpr int KW = 16 * NA;              // k range of a wave
    constexpr int K = SL_NW * KW;
    constexpr int CBSZ = sl_log2(CGRP);
    constexpr int NPL = (MODE == MODE_DL) ? 2 : 1;
    constexpr int NLD = K / 128;             // 16-byte loads per lane and (plane, row group) block: 2 K loads / 256 lanes
    constexpr int PPL = SL_MAX_RC * 2 / SL_THREADS;  // row pairs a lane may own (2 RG C pairs over 256 lanes)
    static_assert(K * 4 <= SL_XS_FLOATS / 2, "a block of the staged input fits a ring slot");
    __shared__ __attribute__((aligned(16))) float lds[SL_XS_FLOATS + SL_RED_FLOATS + 4];
    float* const xs = lds;                       // ring of two blocks [K][4 rows]
    float* const red = lds + SL_XS_FLOATS;       // [wave][plane][rg][C][4 rows]
    constexpr int DEAD = SL_XS_FLOATS + SL_RED_FLOATS;
    constexpr int SLOT = SL_XS_FLOATS / 2;

    // ---- who am I -----------------------------------------------------------------------------------
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = a.G, RG = a.RG;
    // Placement (speed only): blocks b, b + nxcd, ... share an XCD; a cluster's members are confined to a group of
    // `span` XCDs (1: its exchange stays in one L2; nxcd: round-robin over the chip), the groups take the clusters in turn
    const int xcd = blockIdx.x % a.nxcd, idx = blockIdx.x / a.nxcd;
    const int slot_in_group = idx * a.span + xcd % a.span;  // position inside the group's run of members
    const int cluster = (slot_in_group / G) * (a.nxcd / a.span) + xcd / a.span;
    const int member = slot_in_group % G;
    if (cluster >= a.nclusters) return;  // a whole cluster out of range: nobody waits for it
    const int N = a.N, ld = a.ld;
    const int col0 = member * C;
    const int crow0 = cluster * 4 * RG;
    if (tid == 0) lds[DEAD] = 0.0f;

    // ---- Q slab, resident in registers for the whole launch -----------------------------------------
    // lane = 4 (kr CGRP + cg) + j.  MFMA q = a CGRP + m of a wave covers, in residue kr, k = kw0 + 16 a + kr CGRP + m:
    // its B operand is Q[k][col0 + 4 cg + j], its A operand x[row j][k] comes from block (kr, cg = m) of A register a
    // -- so lane (kr, cg, j) reads x[row j][kw0 + 16 a + kr CGRP + cg], i.e. float kw0 * 4 + 64 a + lane of the block:
    // one conflict-free ds_read_b32 per A register (with k = ... + m KRES + kr the eight column groups of C = 32 put two
    // lanes of a half-wave on every bank: SQ_LDS_BANK_CONFLICT 256 cycles per CU and step)
    const int blk = lane >> 2, j4 = lane & 3;
    const int kr = blk / CGRP, cg = blk % CGRP;
    const int kw0 = wave * KW;
    float qf[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int k = kw0 + (q / CGRP) * 16 + kr * CGRP + (q % CGRP);
        qf[q] = (k < ld) ? a.Q[(size_t)k * ld + col0 + 4 * cg + j4] : 0.0f;  // (col0 + C <= ld: C divides 128)
    }
    const int a_off = kw0 * 4 + lane;

    // ---- owners: pair t (t = tid, tid + 256; t < 2 RG C) = rows 2 h, 2 h + 1 of row group t / (2 C) at column col.
    // 2 C divides 256, so a lane's pairs share the column (and h): only their row groups differ.
    const int EP = 2 * RG * C;
    const int oh = tid & 1, oc = (tid >> 1) % C;
    const int col = col0 + oc;
    const bool col_ok = col < N;
    const float vj = col_ok ? a.V[col] : 0.0f;
    const float shift_j = a.in_shift * a.qsum[col];
    const float sat_j = (a.s_cols && col_ok) ? a.s_cols[col] : 1.0f;
    const float inv_sat_j = a.s_cols ? 1.0f / sat_j : 1.0f;
    bool owner[PPL];
    int org[PPL], brow[PPL][2];
    bool ok[PPL][2];
    float s0[PPL][2], s1[PPL][2], mt[PPL][2], wc[PPL][2], am[PPL][2], av[PPL][2];
    auto gidx = [&](int p, int e) { return (size_t)brow[p][e] * ld + col; };
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int t = tid + SL_THREADS * p;
        owner[p] = t < EP;
        org[p] = t / (2 * C);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            brow[p][e] = crow0 + 4 * org[p] + 2 * oh + e;
            ok[p][e] = owner[p] && col_ok && brow[p][e] < a.B;
            s0[p][e] = ok[p][e] ? a.x0[gidx(p, e)] : 0.0f;
            s1[p][e] = (MODE != MODE_LANGEVIN && ok[p][e]) ? a.x1[gidx(p, e)] : 0.0f;
            mt[p][e] = wc[p][e] = 0.0f;
            am[p][e] = (a.adam && ok[p][e]) ? a.am[gidx(p, e)] : 0.0f;
            av[p][e] = (a.adam && a.ad.use_v && ok[p][e]) ? a.av[gidx(p, e)] : 0.0f;
        }
    }

    // ---- exchange buffers ---------------------------------------------------------------------------
    const size_t xbytes = (size_t)a.nclusters * NPL * RG * K * 4 * SL_XE;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(a.xb0, 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(a.xb1, 0, (int)xbytes, 0x00020000);
    constexpr int SC1 = 16;  // aux bits of the buffer builtins: sc1
    constexpr unsigned blk_bytes = (unsigned)K * 4 * SL_XE;              // one (plane, row group) block
    const unsigned cbase = (unsigned)(cluster * NPL * RG) * blk_bytes;   // this cluster's blocks
    const int TU = NPL * RG;                                             // fetch units = blocks, plane-major

    unsigned long long t_last = 0, seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto mark = [&](unsigned long long& acc) {
        if constexpr (CCVM_SLAB_ABL & 64) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            acc += t - t_last;
            t_last = t;
        }
    };

    // publish pair p's two elements of plane pl with tag `tag` into buffer `par`: one 16-byte store
    const unsigned pub_off = cbase + (unsigned)col * 4 * SL_XE + (unsigned)oh * 16;
    auto publish = [&](int p, int par, const float (&x)[2], unsigned tag, int pl) {
        if (!owner[p]) return;
        const u32x4s v = {__builtin_bit_cast(unsigned, ok[p][0] ? x[0] : 0.0f), tag,
                          __builtin_bit_cast(unsigned, ok[p][1] ? x[1] : 0.0f), tag};
        __builtin_amdgcn_raw_buffer_store_b128(v, par ? rs1 : rs0, pub_off, (pl * RG + org[p]) * (int)blk_bytes, SC1);
    };

    // one-stream normals of pair p's two rows at `step` (it = index inside the launch)
    auto stream_normals = [&](int p, int step, int it, float* out) {
        if (a.replay) {
#pragma unroll
            for (int e = 0; e < 2; ++e) out[e] = ok[p][e] ? a.w0[((size_t)it * N + col) * a.wld + brow[p][e]] : 0.0f;
        } else {
            const NormalPair n = normal_two_rows(a.seed, a.row_offset + brow[p][0], step, col);
            out[0] = n.n0;
            out[1] = n.n1;
        }
    };
    // DL: the (W_c, W_s) pairs of pair p's two elements
    auto pair_normals = [&](int p, int step, int it, float* n0, float* n1) {
        if (a.replay) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const size_t wi = ((size_t)it * N + col) * a.wld + brow[p][e];
                n0[e] = ok[p][e] ? a.w0[wi] : 0.0f;
                n1[e] = ok[p][e] ? a.w1[wi] : 0.0f;
            }
        } else {
            NormalPair pa, pb;
            normal_pair_x2(a.seed, a.row_offset + brow[p][0], a.row_offset + brow[p][1], step, col, pa, pb);
            n0[0] = pa.n0; n1[0] = pa.n1; n0[1] = pb.n0; n1[1] = pb.n1;
        }
    };

    // ---- first input: x(step0) ----------------------------------------------------------------------
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        if (!owner[p]) continue;
        if constexpr (MODE == MODE_MF) {
            stream_normals(p, a.step0, 0, wc[p]);  // mf_solver.py:551-554 for the first step of the launch
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float bound = a.s_cols ? sat_j : a.S;
                mt[p][e] = ok[p][e] ? clampf(__builtin_fmaf(a.k_first, wc[p][e], s0[p][e]), -bound, bound) : 0.0f;
            }
            publish(p, 0, mt[p], (unsigned)a.step0 + 1u, 0);
        } else {
            publish(p, 0, s0[p], (unsigned)a.step0 + 1u, 0);
            if constexpr (MODE == MODE_DL) publish(p, 0, s1[p], (unsigned)a.step0 + 1u, 1);
        }
    }

    // schedule rows through the scalar cache (the table is written by an earlier kernel and never here)
    struct Row { float w[TABLE_WORDS]; };
    typedef const __attribute__((address_space(4))) float* table_ptr;
    const table_ptr table = (table_ptr)(size_t)a.table;
    auto load_row = [&](int i) {
        Row r;
#pragma unroll
        for (int k = 0; k < TABLE_WORDS; ++k) r.w[k] = table[(size_t)i * TABLE_WORDS + k];
        return r;
    };

    // ---- fetch: one unit = one (plane, row group) block = NLD 16-byte loads per lane, staged 1:1 into a ring slot ---
    u32x4s wa[NLD], wb[NLD];
    const unsigned ld_off = cbase + (unsigned)tid * 16u;
    auto issue = [&](int b, int par, u32x4s (&w)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            w[j] = __builtin_amdgcn_raw_buffer_load_b128(par ? rs1 : rs0, ld_off + (unsigned)b * blk_bytes, j * SL_THREADS * 16, SC1);
    };
    // every packet of the unit carries the awaited tag (a stale tag is always smaller, the columns nobody owns carry
    // 0xFFFFFFFF: ccvm_cluster.h, exchange_init_kernel)
    auto arrived = [&](unsigned want, const u32x4s (&w)[NLD]) {
        unsigned lo = want;
#pragma unroll
        for (int j = 0; j < NLD; ++j) asm("v_min3_u32 %0, %1, %2, %3" : "=v"(lo) : "v"(lo), "v"(w[j][1]), "v"(w[j][3]));
        return __builtin_amdgcn_ballot_w64(lo != want) == 0;
    };
    // what a wave sleeps before its first loads of a step (x 64 cycles): waves without owners start with what the
    // generator takes; self-tuning below
    int delay = (wave * 64 >= EP) ? CCVM_SL_DELAY : 0;
    // across the fabric the packets take longer, and longer with more blocks per step (static sweeps of tools/slab_ablate,
    // best delay in units: one block 20-28, two 40-56, four 56+; N = 2000 B = 32: 10.9 us per step without, 7.2 with 56)
    const int delay_own = delay;   // what the wave sleeps for the generator's sake
    if (a.span > 1) delay += a.delay_fabric;
    // Across the fabric the best delay moves with the shape, the solver and the clocks (24-72 units; a wrong one costs
    // 30-50 %), and no closed-loop rule survives the coupling between members (below).  So a launch of 512 steps or
    // more CALIBRATES: candidate delays 72, 64, ... 16 units for sixteen steps each (after a warm-up group), timed with
    // s_memtime over the last twelve; the fastest candidate serves the rest of the launch (open loop: nothing to
    // ratchet; the members of a cluster are in lockstep, see the same step times and choose alike).  144 of <= 4096
    // steps run on candidates.
    constexpr int CAL_STEPS = 16, CAL_CANDS = 8;  // + one warm-up group: the first steps of a launch are not typical
    constexpr bool calibrate = CAL;
    unsigned long long cal_mark = 0, cal_best_t = ~0ull;
    int cal_best = delay;
    // Inside an XCD the delay tunes itself, conservatively: a miss adds three units, only 1024 clean steps take one off,
    // and it never exceeds what a miss costs there (a round trip: ~16 units).  Across the fabric it stays what it is:
    // any rule that adds delay after a miss RATCHETS in a coupled cluster -- the member with the smallest delay asks
    // first, misses because its peers are still sleeping in THEIR delays, adds its units, and the next smallest takes
    // its place: with +4 per miss / -1 per 8 clean steps the delays of a 250-member cluster climbed to 192 units
    // (N = 1000, B = 4: 2.2 -> 9.4 us per step), with +2 / -1 per 1024 to whatever cap they were given.
    const int up = a.span > 1 ? 0 : 3, down_mask = 1023;
    const int delay_cap = a.span > 1 ? 72 : 24;
    bool retried = false;
    bool dead = false;
    const unsigned k_spin = a.spin_limit;
    auto await = [&](int b, int par, unsigned want, u32x4s (&w)[NLD]) {
        if constexpr (CCVM_SLAB_ABL & 4) return;
        if (__builtin_expect(arrived(want, w), 1)) return;
        if (dead) return;  // this wave gave up on an earlier unit of the step: on to the barriers
        unsigned spins = 0;
        const unsigned long long t_wait = wall_clock64();
#pragma nounroll
        do {
            // give up: when the wait's bound is spent (SlabArgs::spin_limit), or as soon as another wave of the workgroup
            // has (volatile: the flag is written without a barrier in between)
            const bool peer_gave_up = (spins & 63u) == 63u && *reinterpret_cast<volatile float*>(lds + DEAD) != 0.0f;
            ++spins;
            if ((unsigned)(wall_clock64() - t_wait) > k_spin || peer_gave_up) {
                if (lane == 0) {
                    __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lds[DEAD] = 1.0f;  // read by everyone behind the step's last barrier
                }
                dead = true;
                break;
            }
            if constexpr (CCVM_SLAB_ABL & 64) seg[7] += 1;
            retried = true;
            __builtin_amdgcn_s_sleep(CCVM_SL_SLEEP);
            // only the pieces that have not arrived travel again (a retry of the whole unit is 8 MB per round and
            // chip at N = 1000): the lanes of an arrived piece load from beyond the buffer (range-checked: zeros, no
            // traffic) and keep what they have -- no branch around the loads (the wait counts at a join would
            // serialise the pieces: one round trip EACH)
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const bool have = min(w[j][1], w[j][3]) == want;
                const u32x4s nw = __builtin_amdgcn_raw_buffer_load_b128(
                    par ? rs1 : rs0, have ? 0xFFFFFFF0u : ld_off + (unsigned)b * blk_bytes, j * SL_THREADS * 16, SC1);
                w[j][0] = have ? w[j][0] : nw[0];
                w[j][1] = have ? w[j][1] : nw[1];
                w[j][2] = have ? w[j][2] : nw[2];
                w[j][3] = have ? w[j][3] : nw[3];
            }
        } while (!arrived(want, w));
    };
    float* const st_dst = xs + tid * 2;
    auto stage = [&](int slot, const u32x4s (&w)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            // (scalars first: __builtin_bit_cast of a vector ELEMENT expression reads element 0, hipcc 7.2)
            const unsigned u0 = w[j][0], u1 = w[j][2];
            const f32x2s v = {__uint_as_float(u0), __uint_as_float(u1)};
            *reinterpret_cast<f32x2s*>(st_dst + slot * SLOT + j * SL_THREADS * 2) = v;
        }
    };
    // partial sums of this wave's k range for block b, staged in ring slot `slot` -> red[wave][b]
    auto contract = [&](int slot, int b) {
        const float* xsb = xs + slot * SLOT + a_off;
        float af[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) af[i] = xsb[i * 64];
        f32x4v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (CCVM_SLAB_ABL & 1) {
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i & 3][0] += af[i] * qf[i];
        } else {
            mfma_chain<CBSZ, CGRP>(af, qf, acc, std::make_integer_sequence<int, NQ>{});
        }
        f32x4v sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        // wavefront-level reduction over the k residues (lanes that differ in kr only; strides 4 CGRP .. 32)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = sum[r];
            if constexpr (CGRP == 1) v = sl_add_ror<4>(v);
            if constexpr (CGRP <= 2) v = sl_add_ror<8>(v);
            if constexpr (CGRP <= 4) v = sl_add_swap16(v);
            v = sl_add_swap32(v);
            sum[r] = v;
        }
        if (lane < 4 * CGRP)  // kr == 0: lane = column inside the member
            *reinterpret_cast<f32x4v*>(red + ((size_t)(wave * TU + b) * C + lane) * 4) = sum;
    };

    Row rnext = load_row(0);
    __syncthreads();  // DEAD is initialised
    if constexpr (CCVM_SLAB_ABL & 64) { unsigned long long dummy = 0; mark(dummy); }

    for (int it = 0; it < a.nsteps; ++it) {
        const int step = a.step0 + it;
        const int par = it & 1;
        const unsigned want = (unsigned)step + 1u;
        const bool has_next = it + 1 < a.nsteps;
        const Row rcur = rnext;
        const float* trow = rcur.w;

        // ---- this step's / the next step's normals first, THEN the loads: every member publishes at about the same
        // time, and loads issued right behind the own publish mostly meet the peers' previous packets -- each such
        // miss costs a whole round trip (1.0-1.4 retries per step when the loads went first).  The waves without
        // owners sleep as long as the generator takes.
        float nz[PPL][2], nz1[PPL][2];
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            nz[p][0] = nz[p][1] = nz1[p][0] = nz1[p][1] = (CCVM_SLAB_ABL & 2) ? 0.25f : 0.0f;
            if constexpr (!(CCVM_SLAB_ABL & 2)) {
                if (owner[p]) {
                    if constexpr (MODE == MODE_DL) {
                        pair_normals(p, step, it, nz[p], nz1[p]);
                    } else if constexpr (MODE == MODE_MF) {
                        if (has_next) stream_normals(p, step + 1, it + 1, nz[p]);
                    } else {
                        stream_normals(p, step, it, nz[p]);
                    }
                }
            }
        }
        if constexpr (calibrate) if (it <= CAL_STEPS * (CAL_CANDS + 1)) {
            const int g = it / CAL_STEPS - 1, r = it % CAL_STEPS;  // group -1: warm-up on the first candidate
            if (r == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                if (g > 0 && now - cal_mark < cal_best_t) {  // candidate g - 1, its steps 4 .. 15
                    cal_best_t = now - cal_mark;
                    cal_best = delay;
                }
                // from the longest delay down: the long ones never miss, so the cluster keeps its rhythm between them
                delay = g < CAL_CANDS ? delay_own + 72 - 8 * (g < 0 ? 0 : g) : cal_best;
            }
            if (r == 4) cal_mark = __builtin_amdgcn_s_memtime();
        }
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(1);
        rnext = load_row(min(it + 1, a.nsteps - 1));
        retried = false;
        issue(0, par, wa);
        if (TU > 1) issue(1, par, wb);
        mark(seg[0]);

        // ---- the cluster's GEMM input, block by block through the two-slot ring: while the waves contract block u, the
        // loads of blocks u + 1 and u + 2 are in flight.  One barrier per block (the block is staged by all four waves);
        // slot u % 2 is staged again only behind the barrier of block u + 1, which every wave reaches after it has
        // finished contracting block u.
        auto unit = [&](int u, int slot, u32x4s (&w)[NLD]) {
            await(u, par, want, w);
            if (u == 0) {
                mark(seg[1]);
                // the delay before the first loads of the next steps (see `up`, `down_mask`, `delay_cap` above).  Timing
                // only: the result does not depend on it.
                if (retried) delay = min(delay + up, delay_cap);
                else if ((it & down_mask) == down_mask && delay > 0) delay -= 1;
            }
            stage(slot, w);
            if (u + 2 < TU) issue(u + 2, par, w);
            mark(seg[2]);
            __syncthreads();
            mark(seg[3]);
            contract(slot, u);
            mark(seg[4]);
        };
        for (int u = 0; u < TU; u += 2) {
            unit(u, 0, wa);
            if (u + 1 < TU) unit(u + 1, 1, wb);
        }
        __syncthreads();  // the four waves' partial sums of every block are in LDS
        if (lds[DEAD] != 0.0f) return;  // a bounded wait gave up: the whole workgroup leaves (the host recovers)
        mark(seg[5]);

        // ---- the owners' update and the next input -----------------------------------------------------
        AdamScalars ad;
        ad.beta1 = a.ad.beta1; ad.one_m_beta1 = a.ad.one_m_beta1; ad.inv_bc1 = trow[12];
        ad.beta2 = a.ad.beta2; ad.one_m_beta2 = a.ad.one_m_beta2; ad.inv_bc2 = trow[13];
        ad.alpha = a.ad.alpha; ad.eps = a.ad.eps; ad.use_v = a.ad.use_v; ad.add_assign = a.ad.add_assign;
        const unsigned tag = has_next ? (unsigned)step + 2u : 0u;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            if (!owner[p]) continue;
            float qx[2], qx1[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float t[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    float part[SL_NW];
#pragma unroll
                    for (int w = 0; w < SL_NW; ++w)
                        part[w] = red[((size_t)(w * TU + pl * RG + org[p]) * C + oc) * 4 + 2 * oh + e];
                    t[pl] = (part[0] + part[1]) + (part[2] + part[3]);
                }
                qx[e] = __builtin_fmaf(a.in_scale, t[0], shift_j);
                qx1[e] = __builtin_fmaf(a.in_scale, t[NPL - 1], shift_j);
            }
            auto adam = [&](float gr, int e) {
                if (!a.adam) return gr;
                float m, v;
                const float out = adam_precondition(ad, gr, am[p][e], av[p][e], m, v);
                am[p][e] = m;
                av[p][e] = v;
                return out;
            };
            if constexpr (MODE == MODE_DL) {
                const DlScalars k = *reinterpret_cast<const DlScalars*>(trow);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float cn, sn;
                    dl_update(k, s0[p][e], s1[p][e], qx[e], qx1[e], vj, nz[p][e], nz1[p][e], cn, sn);
                    s0[p][e] = cn;
                    s1[p][e] = sn;
                }
                publish(p, par ^ 1, s0[p], tag, 0);
                publish(p, par ^ 1, s1[p], tag, 1);
            } else if constexpr (MODE == MODE_MF) {
                const MfScalars k = *reinterpret_cast<const MfScalars*>(trow);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float bound = a.s_cols ? sat_j : k.S;
                    const float fb = adam(__builtin_fmaf(k.f_q, qx[e], k.f_v * vj) * inv_sat_j, e);
                    float mun, sgn;
                    mf_update(k, s0[p][e], s1[p][e], fb, wc[p][e], mun, sgn);
                    s0[p][e] = mun;
                    s1[p][e] = sgn;
                    // the last step's input is what mu_tilde_out returns: no new measurement after it
                    const bool nxt = has_next;
                    mt[p][e] = nxt ? clampf(__builtin_fmaf(k.k_next, nz[p][e], s0[p][e]), -bound, bound) : mt[p][e];
                    wc[p][e] = nxt ? nz[p][e] : wc[p][e];
                }
                publish(p, par ^ 1, mt[p], tag, 0);
            } else {
                const LvScalars k = *reinterpret_cast<const LvScalars*>(trow);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float gr = adam(__builtin_fmaf(k.g_q, qx[e], k.g_v * vj) * inv_sat_j, e);
                    s0[p][e] = lv_update(k, s0[p][e], gr, nz[p][e], a.s_cols ? sat_j : k.S);
                }
                publish(p, par ^ 1, s0[p], tag, 0);
            }
        }
        mark(seg[6]);
    }
    if constexpr (CCVM_SLAB_ABL & 64) {
        if (tid == 0)
            for (int k = 0; k < 8; ++k) a.dbg[(size_t)blockIdx.x * 16 + k] = seg[k];
    }
    if constexpr (CCVM_SLAB_ABL & 8) {  // the delay every wave ended with (no stamps: the timing is the product's)
        if (lane == 0) a.dbg[(size_t)blockIdx.x * 16 + 8 + wave] = (unsigned long long)delay;
    }

    // ---- write the state back (owner-only data: plain stores) ---------------------------------------
#pragma unroll
    for (int p = 0; p < PPL; ++p)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (!ok[p][e]) continue;
            a.x0[gidx(p, e)] = s0[p][e];
            if constexpr (MODE == MODE_DL) a.x1[gidx(p, e)] = s1[p][e];
            if constexpr (MODE == MODE_MF) {
                a.x1[gidx(p, e)] = s1[p][e];
                if (a.xt) a.xt[gidx(p, e)] = mt[p][e];
            }
            if (a.adam) {
                a.am[gidx(p, e)] = am[p][e];
                if (a.ad.use_v) a.av[gidx(p, e)] = av[p][e];
            }
        }
}

// ---- host side: the shapes that exist, and the plan for (B, N) -------------------------------------------
// K = 64 NQ / CGRP = every multiple of 128 from 384 to 2048; NQ <= 256 registers per lane
struct SlabPlan {
    int ok;        // 0: the slab path does not serve this shape
    int cgrp, nq;  // template parameters
    int K;
    int rg;        // row groups of 4 per cluster
    int G, nclusters;
    int span;      // XCDs a cluster's members are confined to
    int grid;
    double est_us; // the model's time per step (slab_step_estimate_us)
};

// The chip as the launch policy sees it (queried once per device by the ABI; tests pass their own)
struct ChipGeometry {
    int cus;   // compute units
    int xcds;  // XCDs (L2 domains); blocks b and b + xcds share one
};

inline int slab_k_for(int N) {
    // every multiple of 128 (round 5; before: 512, 768, 1024, 1280, 1536, 2048 -- N = 300 contracted K = 512)
    return (N >= 1 && N <= 2048) ? (N <= 384 ? 384 : (N + 127) / 128 * 128) : 0;
}

// Every feasible (rows per cluster, XCDs per cluster, member width) is priced with a small model of a step fitted to
// the measurements of profiles/r03_small_batch.md, and the cheapest wins:
//   hand-off + local chain: 1.0 us inside an XCD, 2.0 over two, 3.0 over four or eight, + 6 ns per member (a round
//     trip is ~1000 cycles inside an L2, ~3500 across the fabric; N = 1000, B = 4: 32 members x 32 columns in one XCD
//     1.9 us per step, 250 x 4 over the chip 2.5-3.0; DL N = 2000, B = 4: 63 x 32 over two XCDs 6.2, 250 x 8 7.3);
//   per (plane, row group) block: NQ MFMAs x 10 cycles + ~450 cycles of operand reads, reductions and barrier, plus
//     the block's fetch (0.25 us per 1024 columns);
//   the owners' noise + update: 0.7 us per pair of rows a lane owns.
// force_cgrp / force_rg (tuning): 0 = choose.  Member widths: K C / 64 registers per lane hold the slab, at most 256.
inline double slab_step_estimate_us(int planes, int rg, int span, int nq, int K, int C, int G) {
    const double base = (span == 1 ? 1.0 : span == 2 ? 2.0 : 3.0) + 0.006 * G;
    const double block = (nq * 10.0 + 450.0) / 2400.0 + 0.25 * K / 1024.0;
    const int pairs = (2 * rg * C + SL_THREADS - 1) / SL_THREADS;
    return base + planes * rg * block + 0.7 * (pairs - 1);
}

inline SlabPlan slab_plan(int B, int N, int planes, const ChipGeometry& chip, int force_cgrp = 0, int force_rg = 0) {
    SlabPlan best{};
    if (N < SL_MIN_N || N > SL_MAX_N || B < 1 || B > SL_MAX_B || chip.cus < 8 || chip.xcds < 1) return best;
    if (chip.cus % chip.xcds) return best;
    const int K = slab_k_for(N);
    const int cus_per_xcd = chip.cus / chip.xcds;
    int spans[8], nspans = 0;  // powers of two that divide the XCD count, then the whole chip
    for (int sp = 1; sp < chip.xcds && nspans < 7 && chip.xcds % sp == 0; sp *= 2) spans[nspans++] = sp;
    spans[nspans++] = chip.xcds;
    double best_us = 0.0;
    for (int rg = 1; rg <= SL_MAX_RC / 4; ++rg) {
        if (force_rg && rg != force_rg) continue;
        const int nclusters = (B + 4 * rg - 1) / (4 * rg);
        if (rg > 1 && (B + 4 * (rg - 1) - 1) / (4 * (rg - 1)) == nclusters) continue;  // same clusters, more padding
        for (int si = 0; si < nspans; ++si) {
            const int span = spans[si];
            const int groups = chip.xcds / span;
            const int per_group = (nclusters + groups - 1) / groups;  // clusters a group of XCDs holds
            for (int cgrp = 1; cgrp <= 8; cgrp *= 2) {
                if (force_cgrp && cgrp != force_cgrp) continue;
                const int C = 4 * cgrp, nq = K * cgrp / 64;
                if (nq > 256 || rg * C > SL_MAX_RC) continue;
                const int G = (N + C - 1) / C;
                if ((long)per_group * G > (long)span * cus_per_xcd) continue;
                const double us = slab_step_estimate_us(planes, rg, span, nq, K, C, G);
                if (best.ok && us >= best_us) continue;
                best.ok = 1; best.cgrp = cgrp; best.nq = nq; best.K = K; best.rg = rg; best.G = G;
                best.nclusters = nclusters; best.span = span;
                best.grid = (per_group * G + span - 1) / span * chip.xcds;
                best_us = us;
                best.est_us = us;
            }
        }
    }
    return best;
}

// Static fetch delay of clusters that span XCDs (x 64 cycles).  Best values of static sweeps with the library
// (CCVM_AMD_SLAB_DELAY, tools/delay_sweep.sh; the time per step is within 5 % over +-8 units around them, 30-50 % worse
// at 16 or 96): one block per step 24-32 (PL N = 2000 B = 8); two blocks 40-56 (DL N = 1200 / 1500 / 2000 at 4 rows:
// 40 / 48 / 56; PL N = 2000 B = 32: 56; Langevin N = 1500 B = 32 prefers 24); four blocks 56-72.
inline int slab_fabric_delay(int planes, int rg, int K) {
    const int d = 24 + 16 * (planes * rg - 1) * K / 1024;
    return d < 72 ? d : 72;
}

inline size_t slab_exchange_bytes(int B, int N, int planes) {
    if (N < SL_MIN_N || N > SL_MAX_N || B > SL_MAX_B) return 0;
    // rows of all clusters < B + 4 RG <= B + 256; two buffers of [rows / 4][K][4 rows] packets per plane
    return 2 * (size_t)(B + 256) * planes * slab_k_for(N) * SL_XE;
}

void slab_launch_dl(const SlabArgs& a, const SlabPlan& p, hipStream_t st);
void slab_launch_mf(const SlabArgs& a, const SlabPlan& p, hipStream_t st);
void slab_launch_lv(const SlabArgs& a, const SlabPlan& p, hipStream_t st);

inline bool slab_calibrates(const SlabArgs& a) {
    return CCVM_SL_CALIBRATE && a.span > 1 && a.nsteps >= 512 && !a.delay_fixed;
}

template <int MODE, int CGRP, bool CAL>
void launch_slab_nq(const SlabArgs& a, const SlabPlan& p, int grid, hipStream_t st) {
    const dim3 g(grid), b(SL_THREADS);
    switch (p.K) {
        case 384: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 6 * CGRP, CAL>), g, b, 0, st, a); break;
        case 512: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 8 * CGRP, CAL>), g, b, 0, st, a); break;
        case 640: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 10 * CGRP, CAL>), g, b, 0, st, a); break;
        case 768: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 12 * CGRP, CAL>), g, b, 0, st, a); break;
        case 896: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 14 * CGRP, CAL>), g, b, 0, st, a); break;
        case 1024: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 16 * CGRP, CAL>), g, b, 0, st, a); break;
        case 1152: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 18 * CGRP, CAL>), g, b, 0, st, a); break;
        case 1280: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 20 * CGRP, CAL>), g, b, 0, st, a); break;
        case 1408: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 22 * CGRP, CAL>), g, b, 0, st, a); break;
        case 1536: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 24 * CGRP, CAL>), g, b, 0, st, a); break;
        case 1664: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 26 * CGRP, CAL>), g, b, 0, st, a); break;
        case 1792: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 28 * CGRP, CAL>), g, b, 0, st, a); break;
        case 1920: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 30 * CGRP, CAL>), g, b, 0, st, a); break;
        case 2048: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 32 * CGRP, CAL>), g, b, 0, st, a); break;
        default: break;
    }
}

template <int MODE, bool CAL>
void launch_slab_cal(const SlabArgs& a, const SlabPlan& p, hipStream_t st) {
    const int grid = p.grid - a.drop;
    switch (p.cgrp) {
        case 1: launch_slab_nq<MODE, 1, CAL>(a, p, grid, st); break;
        case 2: launch_slab_nq<MODE, 2, CAL>(a, p, grid, st); break;
        case 4: launch_slab_nq<MODE, 4, CAL>(a, p, grid, st); break;
        case 8: launch_slab_nq<MODE, 8, CAL>(a, p, grid, st); break;
        default: break;
    }
}

// the calibrating instantiations live in translation units of their own (ccvm_slab_*_cal.hip)
void slab_launch_dl_cal(const SlabArgs& a, const SlabPlan& p, hipStream_t st);
void slab_launch_mf_cal(const SlabArgs& a, const SlabPlan& p, hipStream_t st);
void slab_launch_lv_cal(const SlabArgs& a, const SlabPlan& p, hipStream_t st);

}  // namespace ccvm
