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
// {x_row, tag, x_row+1, tag}, and a reader's 16-byte loads are contiguous over the whole input.
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
constexpr int SL_XS_FLOATS = 32768;      // staged GEMM input: planes x RG x K x 4 floats (128 KB)
constexpr int SL_RED_FLOATS = 4096;      // partial sums of the four waves: 4 x planes x RG x C x 4 floats
constexpr int SL_MIN_N = 257, SL_MAX_N = 2048;
constexpr int SL_MAX_RC = 128;           // RG x C: two rows x one column per lane -> at most 256 owners
constexpr int SL_MAX_B = 512;            // batches the path is ever considered for (workspace sizing)
constexpr unsigned SL_SPIN_LIMIT = 1u << 22;

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
    int nclusters, G, RG;  // clusters of G members; a cluster owns 4 RG batch rows
    int span;            // XCDs a cluster's members are confined to (1, 2, 4, ... nxcd): speed only
    int nxcd;            // XCDs of the device (blocks b and b + nxcd share one)
    int drop;            // fault injection (tests only): this many workgroups are left out of the launch
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
template <int MODE, int CGRP, int NQ>
__global__ __launch_bounds__(SL_THREADS) void slab_kernel(const SlabArgs a) {
    static_assert(MODE == MODE_DL || MODE == MODE_MF || MODE == MODE_LANGEVIN, "slab kernel: solver loops only");
    static_assert(CGRP == 1 || CGRP == 2 || CGRP == 4 || CGRP == 8, "column groups per member");
    static_assert(NQ % CGRP == 0 && NQ <= 256, "Q registers per lane (one wave per SIMD: 512 registers)");
    constexpr int C = 4 * CGRP;
    constexpr int KRES = 16 / CGRP;          // k residues inside one MFMA
    constexpr int NA = NQ / CGRP;            // A registers per wave, row group and plane (16 k each)
    constexpr int KW = 16 * NA;              // k range of a wave
    constexpr int K = SL_NW * KW;
    constexpr int CBSZ = sl_log2(CGRP);
    constexpr int NPL = (MODE == MODE_DL) ? 2 : 1;
    constexpr int NLD = K / 128;             // 16-byte loads per lane and (plane, row group) block: 2 K loads / 256 lanes
    __shared__ __attribute__((aligned(16))) float lds[SL_XS_FLOATS + SL_RED_FLOATS + 4];
    float* const xs = lds;                       // [plane][rg][K][4 rows]
    float* const red = lds + SL_XS_FLOATS;       // [wave][plane][rg][C][4 rows]
    constexpr int DEAD = SL_XS_FLOATS + SL_RED_FLOATS;

    // ---- who am I -----------------------------------------------------------------------------------
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = a.G, RG = a.RG;
    // Placement (speed only): blocks b, b + nxcd, ... share an XCD; a cluster's members are confined to a group of
    // `span` XCDs (1: its exchange stays in one L2; nxcd: round-robin over the chip), the groups take the clusters in turn
    const int xcd = blockIdx.x % a.nxcd, idx = blockIdx.x / a.nxcd;
    const int slot = idx * a.span + xcd % a.span;          // position inside the group's run of members
    const int cluster = (slot / G) * (a.nxcd / a.span) + xcd / a.span;
    const int member = slot % G;
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

    // ---- owners: lane t < 2 RG C owns rows 2 h, 2 h + 1 of row group org at column col ------------------
    const int EP = 2 * RG * C;
    const bool owner = tid < EP;
    const int oh = tid & 1, oc = (tid >> 1) % C, org = tid / (2 * C);
    const int col = col0 + oc;
    const bool col_ok = owner && col < N;
    const float vj = col_ok ? a.V[col] : 0.0f;
    const float shift_j = owner ? a.in_shift * a.qsum[col] : 0.0f;
    const float sat_j = (a.s_cols && col_ok) ? a.s_cols[col] : 1.0f;
    const float inv_sat_j = a.s_cols ? 1.0f / sat_j : 1.0f;
    int brow[2];
    bool ok[2];
    float s0[2], s1[2], mt[2], wc[2], am[2], av[2];
    auto gidx = [&](int e) { return (size_t)brow[e] * ld + col; };
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        brow[e] = crow0 + 4 * org + 2 * oh + e;
        ok[e] = col_ok && brow[e] < a.B;
        s0[e] = ok[e] ? a.x0[gidx(e)] : 0.0f;
        s1[e] = (MODE != MODE_LANGEVIN && ok[e]) ? a.x1[gidx(e)] : 0.0f;
        mt[e] = wc[e] = 0.0f;
        am[e] = (a.adam && ok[e]) ? a.am[gidx(e)] : 0.0f;
        av[e] = (a.adam && a.ad.use_v && ok[e]) ? a.av[gidx(e)] : 0.0f;
    }

    // ---- exchange buffers ---------------------------------------------------------------------------
    const size_t xbytes = (size_t)a.nclusters * NPL * RG * K * 4 * SL_XE;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(a.xb0, 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(a.xb1, 0, (int)xbytes, 0x00020000);
    constexpr int SC1 = 16;  // aux bits of the buffer builtins: sc1
    constexpr unsigned blk_bytes = (unsigned)K * 4 * SL_XE;              // one (plane, row group) block
    const unsigned cbase = (unsigned)(cluster * NPL * RG) * blk_bytes;   // this cluster's blocks
    const int TU = NPL * RG;                                             // fetch units = blocks

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

    // publish this lane's two elements of plane pl with tag `tag` into buffer `par`: one 16-byte store
    const unsigned pub_off = cbase + (unsigned)(org * K + col) * 4 * SL_XE + (unsigned)oh * 16;
    auto publish = [&](int par, const float (&x)[2], unsigned tag, int pl) {
        if (!owner) return;
        const u32x4s v = {__builtin_bit_cast(unsigned, ok[0] ? x[0] : 0.0f), tag,
                          __builtin_bit_cast(unsigned, ok[1] ? x[1] : 0.0f), tag};
        __builtin_amdgcn_raw_buffer_store_b128(v, par ? rs1 : rs0, pub_off, pl * RG * (int)blk_bytes, SC1);
    };

    // one-stream normals of this lane's two rows at `step` (it = index inside the launch)
    auto stream_normals = [&](int step, int it, float* out) {
        if (a.replay) {
#pragma unroll
            for (int e = 0; e < 2; ++e) out[e] = ok[e] ? a.w0[((size_t)it * N + col) * a.B + brow[e]] : 0.0f;
        } else {
            const NormalPair p = normal_two_rows(a.seed, a.row_offset + brow[0], step, col);
            out[0] = p.n0;
            out[1] = p.n1;
        }
    };
    // DL: the (W_c, W_s) pairs of this lane's two elements
    auto pair_normals = [&](int step, int it, float* n0, float* n1) {
        if (a.replay) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const size_t wi = ((size_t)it * N + col) * a.B + brow[e];
                n0[e] = ok[e] ? a.w0[wi] : 0.0f;
                n1[e] = ok[e] ? a.w1[wi] : 0.0f;
            }
        } else {
            NormalPair pa, pb;
            normal_pair_x2(a.seed, a.row_offset + brow[0], a.row_offset + brow[1], step, col, pa, pb);
            n0[0] = pa.n0; n1[0] = pa.n1; n0[1] = pb.n0; n1[1] = pb.n1;
        }
    };

    // ---- first input: x(step0) ----------------------------------------------------------------------
    if (owner) {
        if constexpr (MODE == MODE_MF) {
            stream_normals(a.step0, 0, wc);  // mf_solver.py:551-554 for the first step of the launch
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float bound = a.s_cols ? sat_j : a.S;
                mt[e] = ok[e] ? clampf(__builtin_fmaf(a.k_first, wc[e], s0[e]), -bound, bound) : 0.0f;
            }
            publish(0, mt, (unsigned)a.step0 + 1u, 0);
        } else {
            publish(0, s0, (unsigned)a.step0 + 1u, 0);
            if constexpr (MODE == MODE_DL) publish(0, s1, (unsigned)a.step0 + 1u, 1);
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

    // ---- fetch: one unit = one (plane, row group) block = NLD 16-byte loads per lane, staged 1:1 into xs -----------
    u32x4s wa[NLD], wb[NLD];
    const unsigned ld_off = cbase + (unsigned)tid * 16u;
    auto issue = [&](int b, int par, u32x4s (&w)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            w[j] = __builtin_amdgcn_raw_buffer_load_b128(par ? rs1 : rs0, ld_off + (unsigned)b * blk_bytes, j * SL_THREADS * 16, SC1);
    };
    // every packet of the unit carries the awaited tag (a stale tag is always smaller, the columns nobody owns carry
    // 0xFFFFFFFF: ccvm_cluster.h, slab_init_kernel)
    auto arrived = [&](unsigned want, const u32x4s (&w)[NLD]) {
        unsigned lo = want;
#pragma unroll
        for (int j = 0; j < NLD; ++j) asm("v_min3_u32 %0, %1, %2, %3" : "=v"(lo) : "v"(lo), "v"(w[j][1]), "v"(w[j][3]));
        return __builtin_amdgcn_ballot_w64(lo != want) == 0;
    };
    // what a wave sleeps before its first loads of a step (x 64 cycles): waves without owners start with what the
    // generator takes; self-tuning below
    int delay = (wave * 64 >= EP) ? CCVM_SL_DELAY : 0;
    if (a.span > 1) delay += 28;  // across the fabric the packets take ~2000 cycles longer (converged values measured)
    const int up = a.span > 1 ? 4 : 3, down_mask = a.span > 1 ? 7 : 1023;
    bool retried = false;
    bool dead = false;
    auto await = [&](int b, int par, unsigned want, u32x4s (&w)[NLD]) {
        if constexpr (CCVM_SLAB_ABL & 4) return;
        if (__builtin_expect(arrived(want, w), 1)) return;
        unsigned spins = 0;
#pragma nounroll
        do {
            if (++spins > SL_SPIN_LIMIT) {
                if (lane == 0) {
                    __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lds[DEAD] = 1.0f;  // read by everyone behind the next barrier
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
    auto stage = [&](int b, const u32x4s (&w)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            // (scalars first: __builtin_bit_cast of a vector ELEMENT expression reads element 0, hipcc 7.2)
            const unsigned u0 = w[j][0], u1 = w[j][2];
            const f32x2s v = {__uint_as_float(u0), __uint_as_float(u1)};
            *reinterpret_cast<f32x2s*>(st_dst + (size_t)b * (K * 4) + j * SL_THREADS * 2) = v;
        }
    };

    Row rnext = load_row(0);
    __syncthreads();  // xs is zeroed, DEAD is initialised
    if constexpr (CCVM_SLAB_ABL & 64) { unsigned long long dummy = 0; mark(dummy); }

    for (int it = 0; it < a.nsteps; ++it) {
        const int step = a.step0 + it;
        const int par = it & 1;
        const unsigned want = (unsigned)step + 1u;
        const bool has_next = it + 1 < a.nsteps;
        const Row rcur = rnext;
        const float* trow = rcur.w;

        // ---- phase A: the cluster's GEMM input of this step -> LDS --------------------------------------
        // This step's / the next step's normals first, THEN the loads: every member publishes at about the same
        // time, and loads issued right behind the own publish mostly meet the peers' previous packets -- each such
        // miss costs a whole round trip across the chip (1.0-1.4 retries per step when the loads went first).  The
        // waves without owners sleep as long as the generator takes.
        float nz[2] = {0.0f, 0.0f}, nz1[2] = {0.0f, 0.0f};
        if constexpr (CCVM_SLAB_ABL & 2) {
            nz[0] = nz[1] = nz1[0] = nz1[1] = 0.25f;
        } else if (owner) {
            if constexpr (MODE == MODE_DL) {
                pair_normals(step, it, nz, nz1);
            } else if constexpr (MODE == MODE_MF) {
                if (has_next) stream_normals(step + 1, it + 1, nz);
            } else {
                stream_normals(step, it, nz);
            }
        }
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(1);
        rnext = load_row(min(it + 1, a.nsteps - 1));
        retried = false;
        issue(0, par, wa);
        mark(seg[0]);
        for (int u = 0; u < TU && !dead; u += 2) {
            if (u + 1 < TU) issue(u + 1, par, wb);
            await(u, par, want, wa);
            if (u == 0) {
                mark(seg[1]);
                // Self-tuning delay (x 64 cycles) before the first loads of a step: a miss costs a round trip (~1000
                // cycles inside an XCD, ~3500 across the fabric) plus the retry's bookkeeping, waiting a little too
                // long costs 64 cycles a unit.  Inside an XCD arrival times are steady and a cluster advances at the pace
                // of its slowest wave (~100 waves probing downwards: somebody always misses), so a miss adds three units
                // and only 1024 clean steps take one off; across the fabric the arrival times wander and the delay has to
                // follow them: four units up, one off every eight clean steps (measured, us per step, static / slow /
                // fast: N = 1000 B = 32 in XCDs 1.92 / 1.99 / 2.17; N = 2000 B = 32 over two XCDs 10.5 / 7.2 / 6.4; N = 1000
                // B = 4 over the chip 3.0 / 2.7 / 2.2).  Timing only: the result does not depend on it.
                if (retried) delay = min(delay + up, 192);
                else if ((it & down_mask) == down_mask && delay > 0) delay -= 1;
            }
            stage(u, wa);
            if (u + 2 < TU) issue(u + 2, par, wa);
            if (u + 1 < TU) {
                await(u + 1, par, want, wb);
                stage(u + 1, wb);
            }
        }
        mark(seg[2]);
        __syncthreads();  // B1: the input is staged
        if (lds[DEAD] != 0.0f) return;
        mark(seg[3]);

        // ---- phase B: partial sums of this wave's k range, every plane and row group --------------------
        for (int b = 0; b < NPL * RG; ++b) {
            const float* xsb = xs + (size_t)b * (K * 4) + a_off;
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
                *reinterpret_cast<f32x4v*>(red + ((size_t)(wave * NPL * RG + b) * C + lane) * 4) = sum;
        }
        mark(seg[4]);
        __syncthreads();  // B2: the four waves' partial sums are in LDS; xs may be overwritten
        mark(seg[5]);

        // ---- phase C: the owners' update and the next input -------------------------------------------
        if (owner) {
            float qx[2], qx1[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float t[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    float p[SL_NW];
#pragma unroll
                    for (int w = 0; w < SL_NW; ++w)
                        p[w] = red[((size_t)(w * NPL * RG + pl * RG + org) * C + oc) * 4 + 2 * oh + e];
                    t[pl] = (p[0] + p[1]) + (p[2] + p[3]);
                }
                qx[e] = __builtin_fmaf(a.in_scale, t[0], shift_j);
                qx1[e] = __builtin_fmaf(a.in_scale, t[NPL - 1], shift_j);
            }
            AdamScalars ad;
            ad.beta1 = a.ad.beta1; ad.one_m_beta1 = a.ad.one_m_beta1; ad.inv_bc1 = trow[12];
            ad.beta2 = a.ad.beta2; ad.one_m_beta2 = a.ad.one_m_beta2; ad.inv_bc2 = trow[13];
            ad.alpha = a.ad.alpha; ad.eps = a.ad.eps; ad.use_v = a.ad.use_v; ad.add_assign = a.ad.add_assign;
            auto adam = [&](float gr, int e) {
                if (!a.adam) return gr;
                float m, v;
                const float out = adam_precondition(ad, gr, am[e], av[e], m, v);
                am[e] = m;
                av[e] = v;
                return out;
            };
            const unsigned tag = has_next ? (unsigned)step + 2u : 0u;
            if constexpr (MODE == MODE_DL) {
                const DlScalars k = *reinterpret_cast<const DlScalars*>(trow);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float cn, sn;
                    dl_update(k, s0[e], s1[e], qx[e], qx1[e], vj, nz[e], nz1[e], cn, sn);
                    s0[e] = cn;
                    s1[e] = sn;
                }
                publish(par ^ 1, s0, tag, 0);
                publish(par ^ 1, s1, tag, 1);
            } else if constexpr (MODE == MODE_MF) {
                const MfScalars k = *reinterpret_cast<const MfScalars*>(trow);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float bound = a.s_cols ? sat_j : k.S;
                    const float fb = adam(__builtin_fmaf(k.f_q, qx[e], k.f_v * vj) * inv_sat_j, e);
                    float mun, sgn;
                    mf_update(k, s0[e], s1[e], fb, wc[e], mun, sgn);
                    s0[e] = mun;
                    s1[e] = sgn;
                    // the last step's input is what mu_tilde_out returns: no new measurement after it
                    const bool nxt = k.has_next;
                    mt[e] = nxt ? clampf(__builtin_fmaf(k.k_next, nz[e], s0[e]), -bound, bound) : mt[e];
                    wc[e] = nxt ? nz[e] : wc[e];
                }
                publish(par ^ 1, mt, tag, 0);
            } else {
                const LvScalars k = *reinterpret_cast<const LvScalars*>(trow);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float gr = adam(__builtin_fmaf(k.g_q, qx[e], k.g_v * vj) * inv_sat_j, e);
                    s0[e] = lv_update(k, s0[e], gr, nz[e], a.s_cols ? sat_j : k.S);
                }
                publish(par ^ 1, s0, tag, 0);
            }
        }
        mark(seg[6]);
    }
    if constexpr (CCVM_SLAB_ABL & 64) {
        if (tid == 0)
            for (int k = 0; k < 8; ++k) a.dbg[(size_t)blockIdx.x * 16 + k] = seg[k];
    }

    // ---- write the state back (owner-only data: plain stores) ---------------------------------------
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (!ok[e]) continue;
        a.x0[gidx(e)] = s0[e];
        if constexpr (MODE == MODE_DL) a.x1[gidx(e)] = s1[e];
        if constexpr (MODE == MODE_MF) {
            a.x1[gidx(e)] = s1[e];
            if (a.xt) a.xt[gidx(e)] = mt[e];
        }
        if (a.adam) {
            a.am[gidx(e)] = am[e];
            if (a.ad.use_v) a.av[gidx(e)] = av[e];
        }
    }
}

// ---- host side: the shapes that exist, and the plan for (B, N) -------------------------------------------
// K = 64 NQ / CGRP in {512, 768, 1024, 1280, 1536, 2048}; NQ <= 256 registers per lane
struct SlabPlan {
    int ok;        // 0: the slab path does not serve this shape
    int cgrp, nq;  // template parameters
    int K;
    int rg;        // row groups of 4 per cluster
    int G, nclusters;
    int span;      // XCDs a cluster's members are confined to
    int grid;
};

// The chip as the launch policy sees it (queried once per device by the ABI; tests pass their own)
struct ChipGeometry {
    int cus;   // compute units
    int xcds;  // XCDs (L2 domains); blocks b and b + xcds share one
};

inline int slab_k_for(int N) {
    const int ks[6] = {512, 768, 1024, 1280, 1536, 2048};
    for (int i = 0; i < 6; ++i)
        if (N <= ks[i]) return ks[i];
    return 0;
}

// Fewest rows per cluster first (the fetched input per member and step is R K packets, the contraction R C K MACs);
// for that row count the fewest XCDs per cluster (measured, N = 1000, B = 4, us per step: 32 members x 32 columns
// inside an XCD 1.89; 63 x 16 / 125 x 8 / 250 x 4 over the chip 3.3 / 3.3 / 3.0: across the fabric a hand-off round
// trip is ~3500 cycles against ~1000 inside an L2), then the narrowest member.  force_cgrp / force_rg (tuning): 0 =
// choose.  Member widths: K C / 64 registers per lane hold the slab, at most 256 (one wave per SIMD).
inline SlabPlan slab_plan(int B, int N, int planes, const ChipGeometry& chip, int force_cgrp = 0, int force_rg = 0) {
    SlabPlan p{};
    if (N < SL_MIN_N || N > SL_MAX_N || B < 1 || B > SL_MAX_B || chip.cus < 8 || chip.xcds < 1) return p;
    if (chip.cus % chip.xcds) return p;
    const int K = slab_k_for(N);
    const int cus_per_xcd = chip.cus / chip.xcds;
    for (int rg = 1; rg <= 32; ++rg) {
        if (force_rg && rg != force_rg) continue;
        if (planes * rg * K * 4 > SL_XS_FLOATS) break;
        const int nclusters = (B + 4 * rg - 1) / (4 * rg);
        int spans[8], nspans = 0;  // powers of two that divide the XCD count, then the whole chip
        for (int sp = 1; sp < chip.xcds && nspans < 7 && chip.xcds % sp == 0; sp *= 2) spans[nspans++] = sp;
        spans[nspans++] = chip.xcds;
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
                p.ok = 1; p.cgrp = cgrp; p.nq = nq; p.K = K; p.rg = rg; p.G = G; p.nclusters = nclusters;
                p.span = span;
                p.grid = (per_group * G + span - 1) / span * chip.xcds;
                return p;
            }
        }
    }
    return p;
}

inline size_t slab_exchange_bytes(int B, int N, int planes) {
    if (N < SL_MIN_N || N > SL_MAX_N || B > SL_MAX_B) return 0;
    // rows of all clusters < B + 4 RG <= B + 128; two buffers of [rows / 4][K][4 rows] packets per plane
    return 2 * (size_t)(B + 128) * planes * slab_k_for(N) * SL_XE;
}

void slab_launch_dl(const SlabArgs& a, const SlabPlan& p, hipStream_t st);
void slab_launch_mf(const SlabArgs& a, const SlabPlan& p, hipStream_t st);
void slab_launch_lv(const SlabArgs& a, const SlabPlan& p, hipStream_t st);

template <int MODE, int CGRP>
void launch_slab_nq(const SlabArgs& a, const SlabPlan& p, int grid, hipStream_t st) {
    const dim3 g(grid), b(SL_THREADS);
    switch (p.K) {
        case 512: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 8 * CGRP>), g, b, 0, st, a); break;
        case 768: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 12 * CGRP>), g, b, 0, st, a); break;
        case 1024: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 16 * CGRP>), g, b, 0, st, a); break;
        case 1280: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 20 * CGRP>), g, b, 0, st, a); break;
        case 1536: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 24 * CGRP>), g, b, 0, st, a); break;
        case 2048: hipLaunchKernelGGL((slab_kernel<MODE, CGRP, 32 * CGRP>), g, b, 0, st, a); break;
        default: break;
    }
}

template <int MODE>
void launch_slab(const SlabArgs& a, const SlabPlan& p, hipStream_t st) {
    const int grid = p.grid - a.drop;
    switch (p.cgrp) {
        case 1: launch_slab_nq<MODE, 1>(a, p, grid, st); break;
        case 2: launch_slab_nq<MODE, 2>(a, p, grid, st); break;
        case 4: launch_slab_nq<MODE, 4>(a, p, grid, st); break;
        case 8: launch_slab_nq<MODE, 8>(a, p, grid, st); break;
        default: break;
    }
}

}  // namespace ccvm
