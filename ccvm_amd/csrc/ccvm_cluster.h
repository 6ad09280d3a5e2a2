// Column-cluster persistent kernel for the mid-size regime (256 < N <= 768): a whole chunk of time
// steps in ONE launch, every solver (DL, MF, Langevin / pumped Langevin) and Adam variant.
//
// Why: at N = 500, B = 1000 a per-step launch of the tile kernel takes 8.2 us for 3.1 us of MFMA work
// (tools/ablate_mid.hip): the kernel boundary (~1.8 us), the first tiles' fill, the store drain and a
// Q panel re-streamed L2 -> LDS every step (192 KB per workgroup and step) are paid once per STEP.  The
// persistent row-owner kernel (ccvm_persist.h) cannot take over: above N = 256 a workgroup can hold
// neither Q's fragments in registers nor all of Q in LDS.
//
// Here a CLUSTER of G = ceil(N / 64) workgroups owns 32 batch rows (48 above K = 512) for the whole
// launch.  Member m keeps the Q panel of its 64 output columns resident (k < 512 in LDS: <= 128 KB; k >= 512,
// for K = 640 / 768, as B fragments in the MFMA waves' registers; loaded once per launch), owns the
// elements (row, its 64 columns) in registers, and per step needs the cluster's full GEMM input rows
// (rows x K; DL: two planes, c and s), of which the other members produce (G - 1) / G.
//
// Exchange: flag in data.  Every exchanged element is an 8-byte packet {value, tag} (tag = global step
// number of the GEMM it feeds + 1; buffers zeroed before the call), written by ONE lane with ONE 8-byte
// sc1 (write-through, agent scope) store and read as half of a 16-byte sc1 load: a reader that finds the
// expected tag has the value stored with it (gfx950 inter-workgroup rules, MI355X_MICROARCH.md,
// visibility).  No drain, no counter, no poll: the hand-off chain is store -> load instead of
// store -> drain -> counter -> poll -> load (round 2's first version: three memory round trips, ~2.7 us,
// longer than a phase).  Ping-pong buffers (input of even / odd steps); a member publishes input j + 1
// only after it has consumed input j from every member, so nobody overwrites input j (with j + 2) while
// anybody still reads it, and every stale tag a reader can meet is SMALLER than the one it waits for
// (min over the tags == expected  <=>  everything arrived).
//
// Roles.  512 threads: waves 0-3 (one per SIMD) do nothing but MFMAs, LDS operand reads, the update and
// the publish stores; waves 4-7 (their SIMD siblings) fetch the exchanged rows, check the tags (bounded
// retries), and stage them into the LDS operand ring.  Measured with s_memtime stamps on the
// one-role version: issuing the exchange loads (64 B/clk per CU through the texture path: 32-64 KB per
// phase) blocked the issuing wave for 400-1100 cycles per phase, the staging and its barrier another
// ~700, all of it with the matrix pipe idle; a sibling wave's VMEM / LDS issue and its waiting cost the
// MFMA wave nothing (only its VALU instructions do: tools/coissue.hip; they are kept to a v_min3_u32 per
// 16 bytes and the staging moves).
// Latency hiding: the rows are two (K <= 512) or three independent row sets of 16 (v_mfma_f32_16x16x4_f32
// tiles); PHASES cycle through the sets, so a set's new input has the other sets' phases to travel.
//
// Operand ring: A chunks of 128 k through three LDS buffers, chunk n of the launch in buffer n % 3; a
// phase has NC = K / 128 chunks (DL: 2 K / 128: plane c, then plane s, against the same panel).  Barrier
// B_c precedes chunk c.  Fetch waves: behind B_c they stage chunk c + 2 (its buffer held chunk c - 1), so
// chunk c + 1 is always complete when chunk c starts (the MFMA waves read operands half a chunk ahead, in
// the issue shadow of the MFMAs); they request the next phase's input in PAIRS of chunks (8 loads per
// wave: never more between two barriers than a chunk of MFMAs covers), pair k behind B_max(L0 + k, 2 k - 1),
// and behind the last barrier check pair 0 and stage the next phase's chunks 0 and 1 while the MFMA waves
// run the last chunk and the update.
//
// Placement and deadlock freedom: blocks b, b + 8, ... share an XCD, a cluster is G consecutive indices of
// one XCD (its exchange stays in that XCD's L2); with 11-12 members, when the clusters do not pack XCD by
// XCD but fit the chip, G consecutive blocks (members on all XCDs, exchange over the fabric).  Every
// LAUNCH's grid fits the chip it is planned for (one workgroup per CU): a batch of more clusters than that
// runs as one launch per round of resident clusters (launch_cluster below; ClusterArgs::cluster0), one
// after the other in stream order -- clusters never interact, and all of a round's clusters finish
// together, so nothing is lost against workgroups trickling in as others leave, and no progress
// argument rests on the order in which a dispatcher places workgroups (rounds 2-5 ran them in ONE launch
// and relied on in-order dispatch: VERDICT r5).  Every spin is bounded all the same: on a timeout the
// fetch wave sets the launch's status word and an LDS flag, the workgroup leaves behind its next barrier
// (the host raises).
//
// Per MFMA wave: 16 of the member's 64 columns, every row set, full K.  MFMA t of a chunk takes
// k = 128 c + 32 g + t for lane group g, so a lane's 32 operands of a chunk are contiguous: 8
// ds_read_b128 per operand and chunk.  Conflict-free LDS image for ds_read_b128 (MI355X_MICROARCH.md,
// LDS: four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32; bank row = 16 slots
// of 16 bytes): a group mixes rows {0-3,12-15} of lane group g with rows {4-11} of lane group g + 1,
// whose k segments sit 8 slots apart -> A rows are stored in the order pi(r) (rows {0-3,12-15} on even
// positions, {4-11} on odd ones; row stride == 1 slot mod 16), and the Q panel keeps each lane group's k
// in its own 128-float block ([g][chunk][t]).  History (profiles/r02_langevin_n500_b1000_pmc.json):
// ds_read_b32 with row stride 4 (mod 64) floats: SQ_LDS_BANK_CONFLICT a quarter of the kernel; plain
// ds_read_b128 image: every read 2-way conflicted.  Same noise definition, folded affine map and pinned
// update arithmetic as the other two kernels; only the summation order of the contraction differs.
#pragma once
#include "ccvm_persist.h"

namespace ccvm {

constexpr int CL_COLS = 64;     // output columns per workgroup (cluster member)
constexpr int CL_ROWS = 16;     // rows per row set (one 16x16x4 tile height); two sets per cluster
constexpr int CL_KC = 128;      // K chunk staged through LDS per barrier
constexpr int CL_MIN_N = 257, CL_MAX_N = 768;
constexpr int CL_LDS_K = 512;   // k < 512 of a member's panel lives in LDS, the rest (K = 640, 768) in the MFMA waves' registers
constexpr int CL_THREADS = 512; // 4 MFMA waves + 4 fetch waves
// (the bound of a wait: ClusterArgs::spin_limit, ticks of the 100 MHz reference clock -- ccvm_abi.hip: spin_ticks.  A count
// of retries bounds nothing: a retry is 0.3 us on an idle chip and several us when every wave of a cluster retries)
constexpr unsigned CL_XE = 8;   // bytes per exchanged element: {value, tag}

struct ClusterArgs {
    const float* Q;      // [ld][ld] (the row-scaled copy with a per-variable saturation)
    const float* V;
    const float* qsum;
    float* x0;           // DL, Langevin: c;  MF: mu   (pitched, in/out; owner-only data)
    float* x1;           // DL: s;  MF: sigma
    float* xt;           // MF: measured amplitude fed to the LAST step of this launch (out, may be NULL)
    float* xb0;          // exchange buffers: GEMM input of even / odd steps, [clusters][planes][32 rows][ld] packets
    float* xb1;          //   {value, tag} (planes: DL c and s, else one); zeroed by the host before the call
    float* am;           // Adam moments (in/out)
    float* av;
    const float* table;  // [nsteps][TABLE_WORDS] schedule rows (the persistent kernel's tables)
    const float* w0;     // REPLAY noise for the chunk: [nsteps][N][B]
    const float* w1;     // DL: the second Wiener stream
    unsigned* status;    // 0 = ok; set to 1 when a bounded spin gave up
    unsigned long long* dbg;  // ablation stamps only: [grid][16]
    uint64_t seed;
    int64_t row_offset;
    int step0, nsteps;
    int replay;
    int B, N, ld;
    int wld;             // REPLAY: pitch of the noise blocks (ccvm_noise::w_ld; >= B)
    int nclusters, G;
    int spread;          // 1: a cluster = G consecutive blocks (members on all XCDs); 0: a cluster stays in one XCD
    int drop;            // fault injection (tests only): this many workgroups are left out of the launch
    unsigned spin_limit; // how long a wave retries before it gives up a wait, in ticks of s_memrealtime (100 MHz; ccvm_abi.hip: spin_ticks)
    int cluster0;        // first cluster of THIS launch (a batch of several rounds is one launch per round)
    int cus, xcds;       // host only: the chip the launch is planned for
    int half_off;        // host only (tuning): 1 = the full kernel also where the half-chunk variant applies
    int sets;            // host only: row sets of 16 per cluster (2; K = 640 / 768: 3, or 2 where clusters of 32 rows fit the chip)
    float in_scale, in_shift;
    float k_first;       // MF: sqrt(1 / (4 j_step0)) / sqrt(dt)
    float S;             // MF: clamp of the measured amplitude
    const float* s_cols; // per-variable saturation S_j (length ld) or NULL
    AdamConsts ad;
};

// Ablation bits for tools/cluster_ablate.hip (0 in the product; timing only, results are wrong): 1 no MFMA,
// 2 no noise, 4 no tag checks, 8 no exchange (no loads, no tag checks, no publish), 16 no publish stores (with 4),
// 32 no LDS operand reads, 64 s_memtime stamps:
// a.dbg[block][0..7] MFMA wave 0: waiting at B_0, first operand read + chunks, update + publish, waiting at the inner
// barriers (total; at B_1, B_2, B_3 and later), the publish stores;
// fetch wave 4: [8] waiting at B_0, [9] staging, loads and the inner barriers, [12] tag check + staging of the next
// phase's first chunks, [13] retry rounds.
#ifndef CCVM_CLUSTER_ABL
#define CCVM_CLUSTER_ABL 0
#endif
#ifndef CCVM_CL_SLEEP
#define CCVM_CL_SLEEP 4
#endif
#ifndef CCVM_CL_FETCH_PRIO
#define CCVM_CL_FETCH_PRIO 1
#endif
#ifndef CCVM_CL_STAGE_ASM
#define CCVM_CL_STAGE_ASM 0
#endif
// Round 6: the fetch waves make the NEXT phase's normals.  A fetch wave is done with its staging
// 1200 cycles before the next phase's first barrier and then only waits, while its MFMA twin spends 60 % of its
// noise + update section in the generator, alone on the SIMD's VALU -- one vector instruction per FOUR cycles, where two
// waves together issue one per TWO (MI355X_MICROARCH.md, per-instruction constants).  With this on, fetch wave w makes
// the Threefry / Box-Muller pairs of MFMA wave w's elements of the next phase (the same calls: bit-identical results)
// into a 4 KB LDS block behind the phase's last inner barrier; the MFMA wave takes its four normals out of that block in
// front of the NEXT phase's last inner barrier (so block and reader are always a barrier apart) and keeps only the update
// and the publish.  One-stream solvers with the panel in LDS (K <= 512: the block fits the 6 KB the panel and the ring
// leave), fused noise.
// MEASURED AND LEFT OFF (same-box A/B of the library, profiles/r06_ab_fetch_noise.txt; stamps:
// profiles/r06_cluster_fetch_noise.txt, tools/cluster_fetch_noise.sh): Langevin N = 500, B = 1000 4.85 -> 5.04 us per step, MF
// 5.03 -> 5.09, Langevin N = 300 3.36 -> 3.57; results bit-identical (tests/test_gpu_cluster.py green either way).  Per
// phase the MFMA wave's noise + update section shrinks by 637 cycles (1364 -> 727; MF 1794 -> 928), but the fetch wave's
// "check pair 0 + stage + normals" grows by 865 (1195 -> 2060) -- 6.6 cycles per generator instruction although its twin
// runs its own ~120 update / publish instructions in the same window: the two waves of a SIMD issued one VALU instruction
// per ~4 cycles BETWEEN them, not one per 2 -- and the MFMA waves now wait for it at the next phase's first barrier (236
// -> 628 cycles) and lose 85 cycles of matrix time to its first instructions.  A sibling's VALU work is paid at full
// price on this chip; it can only shrink.
#ifndef CCVM_CL_FETCH_NOISE
#define CCVM_CL_FETCH_NOISE 0
#endif
// MFMAs per operand-prefetch unit: 32 = a whole chunk ahead (2 x 64 operand registers), 16 = half a chunk (2 x 32
// registers; 512 cycles of MFMAs still cover the LDS latency: N = 500 Langevin 5.00 -> 4.91 us / step, N = 640 9.55 ->
// 9.30, DL 18.8 -> 18.5, and 64 registers freed)
#ifndef CCVM_CL_STAMP_TID
#define CCVM_CL_STAMP_TID 256  // the fetch wave whose stamps are reported (first lane of wave 4 .. 7)
#endif
#ifndef CCVM_CL_UNIT
#define CCVM_CL_UNIT 16
#endif

__device__ __forceinline__ unsigned long long cl_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
typedef unsigned u32x2c __attribute__((ext_vector_type(2)));

// KCH = K / 128 (3 .. 6): K = ld = N rounded up to 128.  K <= 512: two row sets of 16 per cluster; K = 640 / 768: three
// (a cluster has 9-12 members there: with 48 rows each, B = 1000 = 21 clusters still fit the chip at once), and the
// panel's k >= 512 are B fragments in the MFMA waves' registers (32 / 64 VGPRs), loaded once per launch.
// REPLAY is a template parameter, not a run-time branch: with the replay loads in the same code as the fused
// noise hipcc guards the registers they share with s_waitcnt vmcnt(0) in BOTH paths.
//
// HALF (round 5; N mod 128 in 1 .. 64, i.e. an odd number of members): the last chunk of a plane holds 64 real k only.
// Its single member publishes column 16 w + j of its 64 at position 32 w + j of the chunk and the panel's rows follow
// (physical k = 32 g + t of that chunk holds logical k = 16 g + t for t < 16, nothing for t >= 16), so that the real k
// are every lane group's FIRST sixteen operands: the chunk's second half-unit of MFMAs is all padding and is left out
// (K in steps of 64 instead of 128: 1/6 of a step's MFMAs at K = 320, 1/12 at K = 704).  Fetch waves, ring and
// barriers are the full kernel's; the results differ from its results in summation order only (the omitted products are zeros,
// the last chunk's real k meet the accumulators in another order).
// SETS (round 5): row sets of 16 per cluster.  K <= 512: two.  K = 640 / 768: three where the batch needs them to fit the
// chip (B = 1000: 21 clusters of 48 rows), two where clusters of 32 rows fit as well (B <= 672 ... 896 by member count):
// a step is then two phases instead of three (ccvm_abi.hip: cluster_rows).
template <int MODE, bool ADAM, int KCH, bool REPLAY, bool HALF, int SETS>
__device__ __forceinline__ void cluster_body(const ClusterArgs& a) {
    static_assert(MODE == MODE_DL || MODE == MODE_MF || MODE == MODE_LANGEVIN, "cluster kernel: solver loops only");
    static_assert(!(ADAM && MODE == MODE_DL), "DL has no Adam variant");
    static_assert(KCH >= 3 && KCH <= 6, "K = 384 ... 768");
    constexpr int K = KCH * CL_KC;
    static_assert(SETS == 2 || (SETS == 3 && KCH > 4), "row sets per cluster");
    constexpr int NSETS = SETS;                    // row sets of 16 per cluster (ccvm_abi.hip: cluster_rows)
    constexpr int CROWS = NSETS * CL_ROWS;
    constexpr int KL = (KCH > 4) ? 4 : KCH;        // panel chunks in LDS; chunks KL .. KCH-1 in registers
    // DL contracts two input planes (c, then s) against the same panel: 2 KCH chunks per phase, two accumulator sets
    constexpr int NPL = (MODE == MODE_DL) ? 2 : 1;
    constexpr int NC = NPL * KCH;
    constexpr int QS = 512 + 4;             // panel row stride (floats): == 4 (mod 32); a 128-float block per lane group
    constexpr int AS = CL_KC + 4;           // A chunk row stride (floats): == 1 slot (mod 16)
    constexpr int ABUF = CL_ROWS * AS + 4;
    constexpr int QPANEL = CL_COLS * QS + 4;
    // position of row r in an A buffer -- rows 0-3, 12-15 -> 0, 2, .., 14; rows 4-11 -> 1, 3, .., 15
    auto rowpos = [](int r) { return (r < 4) ? 2 * r : (r < 12) ? 2 * (r - 4) + 1 : 2 * (r - 8); };
    // position of Q[k][.] inside a panel column: k = 128 c + 32 g + t  ->  128 g + 32 c + t
    auto kpos = [](int k) { return 128 * ((k >> 5) & 3) + 32 * (k >> 7) + (k & 31); };
    // the Q row held at physical k (HALF: the last chunk's operand t of lane group g is row 16 g + t of its 64; t >= 16: none)
    auto ksrc = [](int k) {
        if (!HALF || k < CL_KC * (KCH - 1)) return k;
        return ((k & 31) < 16) ? CL_KC * (KCH - 1) + 16 * ((k >> 5) & 3) + (k & 31) : -1;
    };
    // one array (a second __shared__ object can de-pipeline the loop, guide section 5)
    constexpr bool FN = CCVM_CL_FETCH_NOISE && MODE != MODE_DL && !REPLAY && KCH <= 4 && !(CCVM_CLUSTER_ABL & 2);
    constexpr int NZB = FN ? 4 * 64 * 4 : 0;  // [MFMA wave][lane][4 rows]: the fetch waves' normals of one phase
    __shared__ __attribute__((aligned(16))) float lds[QPANEL + 3 * ABUF + 4 + NZB];
    float* const nzb = lds + QPANEL + 3 * ABUF + 4;
    float* const qp = lds;                    // [64 columns][512 + 4]
    float* const abuf = lds + QPANEL;         // 3 x [16 rows][128 + 4]
    // lds[DEAD] != 0: a bounded spin gave up.  Written by a fetch wave before a barrier, read by everyone behind it
    // (plain LDS accesses: a generic or volatile access would wait for the exchange loads in flight)
    constexpr int DEAD = QPANEL + 3 * ABUF;

    // ---- who am I -----------------------------------------------------------------------------------
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = a.G;
    // Placement.  Default: blocks b, b + 8, ... share an XCD, so a cluster (consecutive indices of one XCD) keeps its
    // exchange in one L2.  a.spread: consecutive blocks form a cluster, i.e. its members sit on all XCDs and the
    // exchange crosses the fabric (sc1 = agent scope: still coherent) -- for member counts that do not pack into an
    // XCD's 32 CUs (11-12 members: 2 clusters of 12 use 24 CUs per XCD, 21 clusters need 252 of the chip's 256).
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int cluster = a.cluster0 + (a.spread ? blockIdx.x / G : (idx / G) * 8 + xcd);
    const int member = a.spread ? blockIdx.x % G : idx % G;
    if (cluster >= a.nclusters) return;  // a whole cluster out of range: nobody waits for it
    const int N = a.N, ld = a.ld;
    const int col0 = member * CL_COLS;
    const int crow0 = cluster * CROWS;             // first batch row of the cluster
    const int xrow0 = crow0 * NPL;                 // its first row in the exchange buffers: + CROWS plane + 16 set + row
    const int nphases = NSETS * a.nsteps;          // phase P: step P / NSETS of the launch, row set P % NSETS
    if (tid == 0) lds[DEAD] = 0.0f;

    // ---- Q panel, resident for the whole launch: qp[c][k] = Q[k][col0 + c] --------------------------
    {
        const int c = tid & 63, kk = tid >> 6;
#pragma unroll 8
        for (int k = kk; k < KL * CL_KC; k += CL_THREADS / 64) {
            const int ks = ksrc(k);
            qp[c * QS + kpos(k)] = (ks >= 0) ? a.Q[(size_t)ks * ld + col0 + c] : 0.0f;
        }
    }
    // buffer descriptors of the two exchange buffers (wave-uniform: kernel arguments only)
    const size_t xbytes = (size_t)(a.nclusters * CROWS * NPL) * ld * CL_XE;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(a.xb0, 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(a.xb1, 0, (int)xbytes, 0x00020000);
    constexpr int SC1 = 16;  // aux bits of the buffer builtins: sc1
    constexpr bool NO_XCHG = (CCVM_CLUSTER_ABL & 8) != 0;
    unsigned long long t_last = 0;
    auto mark = [&](unsigned long long& acc) {
        if constexpr (CCVM_CLUSTER_ABL & 64) {
            const unsigned long long t = cl_stamp();
            acc += t - t_last;
            t_last = t;
        }
    };
    __syncthreads();  // the panel is in LDS, DEAD is initialised

    if (wave >= 4) {
        // =============================== fetch waves =====================================================
        // piece j (0, 1) of a chunk: row hr + 8 j, floats 4 hq .. 4 hq + 3 of the chunk's 128 = two 16-byte loads
        // {x0, tag, x1, tag}, {x2, tag, x3, tag}
        // a little above the MFMA waves: a fetch wave's few instructions go first when both are ready (DL N = 500:
        // 10.9 -> 10.4 us / step, Langevin 5.13 -> 4.93; priorities 1 and 3 measure the same)
        __builtin_amdgcn_s_setprio(CCVM_CL_FETCH_PRIO);
        const int ht = tid - 256, hr = ht >> 5, hq = ht & 31;
        const unsigned ld_off = (unsigned)(((size_t)(xrow0 + hr) * ld + 4 * hq) * CL_XE);
        // the columns >= 64 G of a plane's last chunk are never published: their tags are ignored, their values stay 0
        // (HALF: those are the positions t >= 16 of every lane group's 32)
        const unsigned pad_tag = (HALF ? ((4 * hq) & 16) != 0 : CL_KC * (KCH - 1) + 4 * hq >= CL_COLS * G) ? 0xFFFFFFFFu : 0u;
        // The fetch waves' VALU instructions are the expensive ones: next to a wave that issues MFMAs back to back a
        // sibling's VALU instruction gets a slot about once per MFMA (~32 cycles; stamps: 60 of them took 1900
        // cycles), so staging and checking must not need any beyond the tag minima: staging addresses are
        // registers per (buffer, piece) selected at compile time (the phase loop is unrolled over the ring's
        // period), values go to LDS with ds_write2_b32 straight from the loaded {x, tag, x, tag} registers.
        typedef __attribute__((address_space(3))) float lds_float;
        unsigned stb[3][2];  // LDS byte addresses
#pragma unroll
        for (int bf = 0; bf < 3; ++bf) {
            stb[bf][0] = (unsigned)(size_t)(lds_float*)(abuf + bf * ABUF + rowpos(hr) * AS + 4 * hq);
            stb[bf][1] = (unsigned)(size_t)(lds_float*)(abuf + bf * ABUF + rowpos(hr + 8) * AS + 4 * hq);
        }
        // the compiler does not track the inline-asm LDS writes: every barrier of the fetch waves drains them itself
        auto fetch_barrier = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        };
        u32x4c w[NC][2][2];  // chunk cc of a phase = chunk cc % KCH of plane cc / KCH
        unsigned long long hseg[6] = {0, 0, 0, 0, 0, 0};
        // An input travels in PAIRS of chunks (8 loads per wave): never more loads between two barriers than the MFMA
        // waves' chunk covers -- issuing a whole input at once (16 loads per wave, 64 KB per workgroup at the
        // texture path's 64 B/clk) held the fetch waves ~1800 cycles and the MFMA waves waited for them at the next
        // barrier.  Pair k = chunks [2 k, min(2 k + 2, NC)).
        constexpr int NPAIR = (NC + 1) / 2;
        auto load_pair = [&](int s, int par, auto k_tag) {
            if constexpr (NO_XCHG) return;
            constexpr int C0 = 2 * decltype(k_tag)::value, C1 = (C0 + 2 < NC) ? C0 + 2 : NC;
#pragma unroll
            for (int cc = C0; cc < C1; ++cc)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        w[cc][j][h] = __builtin_amdgcn_raw_buffer_load_b128(
                            par ? rs1 : rs0, ld_off,
                            ((CROWS * (cc / KCH) + CL_ROWS * s + 8 * j) * ld + CL_KC * (cc % KCH) + 2 * h) * CL_XE, SC1);
        };
        auto arrived = [&](unsigned want, auto k_tag) {
            constexpr int C0 = 2 * decltype(k_tag)::value, C1 = (C0 + 2 < NC) ? C0 + 2 : NC;
            unsigned lo = 0xFFFFFFFFu, lo_pad = 0xFFFFFFFFu;  // tags of full chunks / of a plane's last chunk
#pragma unroll
            for (int cc = C0; cc < C1; ++cc)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned t0 = w[cc][j][h][1], t1 = w[cc][j][h][3];
                        // one v_min3_u32 per packet (left to itself hipcc builds a tree of v_min_u32 + v_min3_u32,
                        // half as many instructions again)
                        if (cc % KCH == KCH - 1) asm("v_min3_u32 %0, %1, %2, %3" : "=v"(lo_pad) : "v"(lo_pad), "v"(t0), "v"(t1));
                        else asm("v_min3_u32 %0, %1, %2, %3" : "=v"(lo) : "v"(lo), "v"(t0), "v"(t1));
                    }
            lo = min(lo, lo_pad | pad_tag);
            // (a lane whose pieces are all padding keeps lo = ~0)
            return __builtin_amdgcn_ballot_w64(lo != want && lo != 0xFFFFFFFFu) == 0;
        };
        // wait (bounded) until pair k of input `want` of set s is complete in w
        const unsigned k_spin = a.spin_limit;
        bool gave_up = false;  // this wave gave up a wait: its later waits of the phase return at once (the workgroup
                               // leaves behind the next B_0; without this every pair of the phase waited out the bound again)
        auto await_pair = [&](int s, int par, unsigned want, auto k_tag) {
            if constexpr (NO_XCHG || (CCVM_CLUSTER_ABL & 4)) return;
            // the first check stands alone: straight-line code whose wait counts leave the younger loads (the
            // input's other pairs) in flight; the merged counts of a loop header would wait for everything
            if (__builtin_expect(arrived(want, k_tag), 1)) return;
            if (gave_up) return;
            const unsigned long long t_wait = wall_clock64();
#pragma nounroll
            do {
                if constexpr (CCVM_CLUSTER_ABL & 64) hseg[5] += 1;
                if ((unsigned)(wall_clock64() - t_wait) > k_spin) {
                    if (lane == 0) {
                        __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        lds[DEAD] = 1.0f;  // read by everyone behind the next B_0
                    }
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(CCVM_CL_SLEEP);
                load_pair(s, par, k_tag);
            } while (!arrived(want, k_tag));
        };
        auto stage = [&](int cc, int buf) {  // chunk cc of the input in w -> A buffer buf (both compile-time after unrolling)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#if CCVM_CL_STAGE_ASM
                // (hipcc turns four scalar stores into v_mov x 4 + ds_write_b128: hence the asm)
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1\n\tds_write2_b32 %0, %3, %4 offset0:2 offset1:3"
                             :: "v"(stb[buf][j]), "v"(w[cc][j][0][0]), "v"(w[cc][j][0][2]), "v"(w[cc][j][1][0]),
                                "v"(w[cc][j][1][2]) : "memory");
#else
                const u32x4c x = {w[cc][j][0][0], w[cc][j][0][2], w[cc][j][1][0], w[cc][j][1][2]};
                *reinterpret_cast<__attribute__((address_space(3))) u32x4c*>(stb[buf][j]) = x;
#endif
            }
        };

        // FN: the normals of phase P for the twin MFMA wave (tid - 256 = its thread: same lane -> rows and column)
        auto make_normals = [&](int P) {
            if constexpr (FN) {
                const int fs = P % NSETS, fj = P / NSETS;            // the phase's row set and step of the launch
                const int fstep = a.step0 + fj + (MODE == MODE_MF ? 1 : 0);  // MF: the NEXT step's normals (mf_solver.py:551-554)
                const int fg = lane >> 4, fcol = col0 + 16 * (wave - 4) + (lane & 15);
                const int64_t r0 = a.row_offset + crow0 + CL_ROWS * fs + 4 * fg;
                NormalPair pa, pb;
                normal_two_rows_x2(a.seed, r0, r0 + 2, fstep, fcol, pa, pb);
                *reinterpret_cast<f32x4c*>(nzb + ((wave - 4) * 64 + lane) * 4) = f32x4c{pa.n0, pa.n1, pb.n0, pb.n1};
            }
        };
        make_normals(0);
        // first input: whatever has not landed yet (the peers may not even run yet) is fetched again by await_pair
        unroll_indices([&](auto k_tag) { load_pair(0, 0, k_tag); }, std::make_integer_sequence<int, NPAIR>{});
        await_pair(0, 0, (unsigned)a.step0 + 1u, std::integral_constant<int, 0>{});
        stage(0, 0);
        stage(1, 1);
        // Pair k of the NEXT phase's input is requested behind barrier B_max(L0 + k, 2 k - 1): its peers stored it at
        // the end of their previous phase, >= L0 chunks ago, and its registers are free -- this phase's chunks 2 k and
        // 2 k + 1 are staged behind B_(2 k - 2) and B_(2 k - 1), the latter earlier in the same interval.
        // (three row sets: the input was stored a whole phase earlier and is requested from the phase's start: N = 640
        // 9.55 -> 8.86 us / step against L0 = 2, and what makes clusters spread over the XCDs pay: N = 768 12.1 -> 10.8)
        constexpr int L0 = (NSETS == 3) ? 0 : (NC == 3) ? 1 : 2;
        auto pair_behind = [](int c) {  // the pair requested behind B_c, or -1
            for (int k = 0; k < NPAIR; ++k)
                if (((L0 + k > 2 * k - 1) ? L0 + k : 2 * k - 1) == c) return k;
            return -1;
        };
        if constexpr (CCVM_CLUSTER_ABL & 64) t_last = cl_stamp();
        bool dead = false;
        // one phase; B0 = buffer of its chunk 0 (compile time)
        auto phase = [&](int P, auto b0_tag) {
            constexpr int B0 = decltype(b0_tag)::value;
            const int cs = P % NSETS, cj = P / NSETS, cpar = cj & 1;  // this phase's set, input number, exchange buffer
            const unsigned cwant = (unsigned)(a.step0 + cj) + 1u;
            const bool next = P + 1 < nphases;
            const int ns = (P + 1) % NSETS, nj = (P + 1) / NSETS;      // the next phase's set and input number
            const unsigned nwant = (unsigned)(a.step0 + nj) + 1u;
            unroll_indices([&](auto c_tag) {
                constexpr int c = decltype(c_tag)::value;
                if (dead) return;
                fetch_barrier();  // B_c.  B_0: chunks 0, 1 of this phase are staged
                if constexpr (c == 0) {
                    if (lds[DEAD] != 0.0f) { dead = true; return; }
                    mark(hseg[0]);
                }
                if constexpr (c + 2 < NC) {
                    // chunk c + 2 into the buffer of chunk c - 1 (done: every MFMA wave is past B_c)
                    if constexpr (c % 2 == 0) await_pair(cs, cpar, cwant, std::integral_constant<int, (c + 2) / 2>{});
                    stage(c + 2, (B0 + c + 2) % 3);
                }
                if constexpr (constexpr int kp = pair_behind(c); kp >= 0) {
                    // unconditional (behind the last phase it fetches packets nobody looks at): with a branch around
                    // the loads the wait counts of the tag checks that follow must assume the shorter path and
                    // would wait for these very loads
                    load_pair(ns, nj & 1, std::integral_constant<int, kp>{});
                }
                if constexpr (c == NC - 1) {
                    mark(hseg[1]);
                    if (next) {
                        await_pair(ns, nj & 1, nwant, std::integral_constant<int, 0>{});
                        stage(0, (B0 + NC) % 3);      // buffers of chunks NC-3, NC-2 of this phase: done before B_(NC-1)
                        stage(1, (B0 + NC + 1) % 3);
                        make_normals(P + 1);           // (every MFMA wave took this phase's normals in front of B_(NC-1))
                    }
                    mark(hseg[4]);
                }
            }, std::make_integer_sequence<int, NC>{});
        };
        // the ring's period: chunk 0 of phase P sits in buffer (NC P) % 3
        for (int P = 0;;) {
            phase(P, std::integral_constant<int, 0>{});
            if (dead || ++P >= nphases) break;
            phase(P, std::integral_constant<int, NC % 3>{});
            if (dead || ++P >= nphases) break;
            phase(P, std::integral_constant<int, (2 * NC) % 3>{});
            if (dead || ++P >= nphases) break;
        }
        if constexpr (CCVM_CLUSTER_ABL & 64) {
            if (tid == CCVM_CL_STAMP_TID)
                for (int k = 0; k < 6; ++k) a.dbg[(size_t)blockIdx.x * 16 + 8 + k] = hseg[k];
        }
        return;
    }

    // ================================== MFMA waves =======================================================
    const int g = lane >> 4;      // lane group: k segment of the operands, row quad of the results
    const int c16 = lane & 15;
    const int col = col0 + 16 * wave + c16;        // this lane's output column
    const bool col_ok = col < N;
    const float vj = col_ok ? a.V[col] : 0.0f;
    const float shift_j = a.in_shift * a.qsum[col];
    const float sat_j = (a.s_cols && col_ok) ? a.s_cols[col] : 1.0f;
    const float inv_sat_j = a.s_cols ? 1.0f / sat_j : 1.0f;

    // ---- this lane's elements: set s, i = 0..3 -> batch row crow0 + 16 s + 4 g + i, column col ------
    int brow[NSETS][4];
    bool ok[NSETS][4];
    float s0[NSETS][4], s1[NSETS][4], mt[NSETS][4], wc[NSETS][4], am[NSETS][4], av[NSETS][4];
    // (a cluster's last rows can lie beyond the padded arrays when it owns 48: every access is guarded by ok)
    auto gidx = [&](int s, int i) { return (size_t)brow[s][i] * ld + col; };
#pragma unroll
    for (int s = 0; s < NSETS; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            brow[s][i] = crow0 + CL_ROWS * s + 4 * g + i;
            ok[s][i] = col_ok && brow[s][i] < a.B;
            s0[s][i] = ok[s][i] ? a.x0[gidx(s, i)] : 0.0f;
            s1[s][i] = (MODE != MODE_LANGEVIN && ok[s][i]) ? a.x1[gidx(s, i)] : 0.0f;
            mt[s][i] = wc[s][i] = am[s][i] = av[s][i] = 0.0f;
            if constexpr (ADAM) {
                am[s][i] = ok[s][i] ? a.am[gidx(s, i)] : 0.0f;
                av[s][i] = (a.ad.use_v && ok[s][i]) ? a.av[gidx(s, i)] : 0.0f;
            }
        }

    // one-stream normals of this lane's four rows of set s at `step` (it = index inside the launch)
    auto stream_normals = [&](int s, int step, int it, float* out) {
        if constexpr (CCVM_CLUSTER_ABL & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) out[i] = 0.25f;
        } else if constexpr (REPLAY) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                out[i] = ok[s][i] ? a.w0[((size_t)it * N + col) * a.wld + brow[s][i]] : 0.0f;
        } else {
            NormalPair pa, pb;
            normal_two_rows_x2(a.seed, a.row_offset + brow[s][0], a.row_offset + brow[s][2], step, col, pa, pb);
            out[0] = pa.n0; out[1] = pa.n1; out[2] = pb.n0; out[3] = pb.n1;
        }
    };

    // DL: the (W_c, W_s) pairs of this lane's four elements of set s
    auto pair_normals = [&](int s, int step, int it, float* n0, float* n1) {
        if constexpr (CCVM_CLUSTER_ABL & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) n0[i] = n1[i] = 0.25f;
        } else if constexpr (REPLAY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t wi = ((size_t)it * N + col) * a.wld + brow[s][i];
                n0[i] = ok[s][i] ? a.w0[wi] : 0.0f;
                n1[i] = ok[s][i] ? a.w1[wi] : 0.0f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i += 2) {  // two generator calls in lockstep
                NormalPair pa, pb;
                normal_pair_x2(a.seed, a.row_offset + brow[s][i], a.row_offset + brow[s][i + 1], step, col, pa, pb);
                n0[i] = pa.n0; n1[i] = pa.n1; n0[i + 1] = pb.n0; n1[i + 1] = pb.n1;
            }
        }
    };

    // Publish set s's new GEMM input x[i] (rows 4 g + i, column col) with tag `tag` into exchange buffer `par`: this
    // lane's elements are its own 8-byte packets; a store instruction writes 4 rows x 16 columns x 8 bytes = four
    // whole 128-byte lines.  Never waited for.
    // (HALF: the last member's columns go to the first sixteen positions of each lane group's 32, see above)
    const int pcol = (HALF && member == G - 1) ? col0 + 32 * wave + c16 : col;
    const unsigned pub_off = (unsigned)(((size_t)(xrow0 + 4 * g) * ld + pcol) * CL_XE);
    auto publish = [&](int s, int par, const float (&x)[4], unsigned tag, int plane = 0) {
        if constexpr (NO_XCHG || (CCVM_CLUSTER_ABL & 16)) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x2c v = {__builtin_bit_cast(unsigned, ok[s][i] ? x[i] : 0.0f), tag};
            __builtin_amdgcn_raw_buffer_store_b64(v, par ? rs1 : rs0, pub_off,
                                                  (CROWS * plane + CL_ROWS * s + i) * ld * CL_XE, SC1);
        }
    };

    // ---- first inputs: x(step0) of every set ---------------------------------------------------------
#pragma unroll
    for (int s = 0; s < NSETS; ++s) {
        if constexpr (MODE == MODE_MF) {
            stream_normals(s, a.step0, 0, wc[s]);  // mf_solver.py:551-554 for the first step of the launch
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float bound = a.s_cols ? sat_j : a.S;
                mt[s][i] = ok[s][i] ? clampf(__builtin_fmaf(a.k_first, wc[s][i], s0[s][i]), -bound, bound) : 0.0f;
            }
            publish(s, 0, mt[s], (unsigned)a.step0 + 1u);
        } else {
            publish(s, 0, s0[s], (unsigned)a.step0 + 1u);
            if constexpr (MODE == MODE_DL) publish(s, 0, s1[s], (unsigned)a.step0 + 1u, 1);
        }
    }

    // operand read addresses: A row c16, B column 16 wave + c16, k segment g; operand m of a chunk at [m]
    const float* const a_rd = abuf + rowpos(c16) * AS + 32 * g;
    const float* const b_rd = qp + (16 * wave + c16) * QS + 128 * g;  // chunk c's B operands start at + 32 c
    // operands are read one UNIT (UM MFMAs' worth: a chunk or half a chunk) ahead, double-buffered
    // (MF + Adam at K = 768 with fused noise, the tightest variant: quarter-chunk units -- 2 x 16 operand registers less,
    // 256 cycles of MFMAs still cover the LDS latency -- instead of 10 spilled registers)
    constexpr int UM = (ADAM && MODE == MODE_MF && KCH == 6 && !REPLAY) ? 8 : CCVM_CL_UNIT, UPC = 32 / UM;
    // units of a plane (HALF: the last chunk's second half is never computed) and of a phase
    constexpr int UPP = KCH * UPC - (HALF ? UPC / 2 : 0), NU = NPL * UPP;
    static_assert(UM == 32 || UM == 16 || UM == 8, "operand unit: a chunk, half or a quarter of a chunk");
    float bq[2][UM];  // B operands of a unit; the first unit's now
    auto read_ops = [&](float (&dst)[UM], const float* src) {  // the UM operands of one unit
#pragma unroll
        for (int q = 0; q < UM / 4; ++q) {
            const f32x4c v = *reinterpret_cast<const f32x4c*>(src + 4 * q);
            dst[4 * q] = v[0]; dst[4 * q + 1] = v[1]; dst[4 * q + 2] = v[2]; dst[4 * q + 3] = v[3];
        }
    };
    // the panel's chunks KL .. KCH-1 (k >= 512): this lane's B fragments, for the whole launch
    float breg[(KCH > KL) ? KCH - KL : 1][32];
    if constexpr (KCH > KL) {
#pragma unroll
        for (int j = 0; j < KCH - KL; ++j)
#pragma unroll
            for (int m = 0; m < 32; ++m)
                breg[j][m] = (HALF && KL + j == KCH - 1)
                                 ? ((m < 16) ? a.Q[(size_t)(CL_KC * (KL + j) + 16 * g + m) * ld + col] : 0.0f)
                                 : a.Q[(size_t)(CL_KC * (KL + j) + 32 * g + m) * ld + col];
    }
    // The first unit's B operands are read during the previous phase's last unit (the panel never changes), into the
    // half of the double buffer that unit does not compute from -- possible when the units of an iteration are even
    // in number (the halves' roles are compile-time constants); else every phase reads them itself, next to its first
    // A operands.
    constexpr bool XPHASE = (NSETS * NU) % 2 == 0;
    if constexpr (XPHASE) read_ops(bq[0], b_rd);

    // schedule rows through the scalar cache into SGPRs (constant address space: the table is written by an earlier
    // kernel and never here): as vector loads the current and the next row held 16-32 VGPRs for the whole iteration
    struct Row { float w[TABLE_WORDS]; };
    typedef const __attribute__((address_space(4))) float* table_ptr;
    const table_ptr table = (table_ptr)(size_t)a.table;
    auto load_row = [&](int i) {
        Row r;
#pragma unroll
        for (int k = 0; k < TABLE_WORDS; ++k) r.w[k] = table[(size_t)i * TABLE_WORDS + k];
        return r;
    };
    Row rnext = load_row(0);
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int b0 = 0;  // buffer of chunk 0 of the current phase
    if constexpr (CCVM_CLUSTER_ABL & 64) t_last = cl_stamp();

    for (int it = 0; it < a.nsteps; ++it) {
        const int step = a.step0 + it;
        const Row rcur = rnext;
        const float* trow = rcur.w;
        const bool has_next = it + 1 < a.nsteps;
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            __syncthreads();  // B_0: the fetch waves have staged this phase's chunks 0 and 1
            // (looking at the flag only behind the chunks, or making the ring addresses below before the barrier, each
            // measured 2 % SLOWER: the MFMA waves gain nothing from starting earlier than the fetch waves' staging)
            if (lds[DEAD] != 0.0f) return;
            mark(seg[0]);
            // chunk c of this phase sits in buffer (b0 + c) % 3
            const float* const ab[3] = {a_rd + b0 * ABUF, a_rd + ((b0 + 1) % 3) * ABUF, a_rd + ((b0 + 2) % 3) * ABUF};
            b0 = (b0 + NC) % 3;

            // ---- acc[plane] = X_plane[set rows][:] @ Q[:, this wave's 16 columns] ------------------------
            f32x4c acc[NPL][2];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) acc[pl][0] = acc[pl][1] = f32x4c{0.0f, 0.0f, 0.0f, 0.0f};
            float aq[2][UM];
            f32x4c fnz = {0.0f, 0.0f, 0.0f, 0.0f};  // FN: the phase's normals, out of the fetch waves' block
            __builtin_amdgcn_sched_barrier(0);
            // operand double buffer: unit n of the iteration (n = NU s + u) computes from [n & 1] (s is a constant
            // after unrolling); without the cross-phase read every phase starts at [0]
            const int ub0 = XPHASE ? (NU * s) & 1 : 0;
            read_ops(aq[ub0], ab[0]);  // unit 0: the one exposed LDS latency
            if constexpr (!XPHASE) read_ops(bq[0], b_rd);
            __builtin_amdgcn_sched_barrier(0);
            unroll_indices([&](auto u_tag) {
                constexpr int u = decltype(u_tag)::value;
                constexpr int pl = u / UPP;                  // plane
                constexpr int pc = (u % UPP) / UPC;          // panel chunk: from LDS (pc < KL) or registers
                constexpr int h = (u % UPP) % UPC;           // unit inside the chunk
                constexpr int c = pl * KCH + pc;             // chunk of the phase
                constexpr int npc = ((u + 1) % UPP) / UPC, nh = ((u + 1) % UPP) % UPC;  // the next unit's chunk and position
                constexpr int nc = ((u + 1) / UPP) * KCH + npc;
                // operands to read during this unit: the next unit's A (if it belongs to this phase) and B (if it lives
                // in LDS and belongs to this phase or the cross-phase read is on)
                constexpr bool RD_A = u + 1 < NU;
                constexpr bool RD_B = npc < KL && (u + 1 < NU || XPHASE);
                const int cb = (ub0 + u) & 1, nb = cb ^ 1;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < UM; ++m) {
                    // order pinned below: hipcc otherwise sinks every read to its use and waits for it there
                    if constexpr (!(CCVM_CLUSTER_ABL & 32)) {
                        if (m % 4 == 0) {  // one b128 per operand and four MFMAs
                            if constexpr (RD_B) {
                                const f32x4c vb = *reinterpret_cast<const f32x4c*>(b_rd + 32 * npc + UM * nh + m);
                                bq[nb][m] = vb[0]; bq[nb][m + 1] = vb[1]; bq[nb][m + 2] = vb[2]; bq[nb][m + 3] = vb[3];
                            }
                            if constexpr (RD_A) {
                                const f32x4c va = *reinterpret_cast<const f32x4c*>(ab[nc % 3] + UM * nh + m);
                                aq[nb][m] = va[0]; aq[nb][m + 1] = va[1]; aq[nb][m + 2] = va[2]; aq[nb][m + 3] = va[3];
                            }
                        }
                    } else {
                        bq[nb][m] = bq[cb][m];
                        aq[nb][m] = aq[cb][m];
                    }
                    const float bop = (pc < KL) ? bq[cb][m] : breg[(pc < KL) ? 0 : pc - KL][UM * h + m];
                    if constexpr (CCVM_CLUSTER_ABL & 1) {
                        acc[pl][0][m & 3] += aq[cb][m] * bop;  // keeps the operands live
                    } else {
                        acc[pl][m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[cb][m], bop, acc[pl][m & 1], 0, 0, 0);
                    }
                }
                constexpr int NRD = (RD_A ? 1 : 0) + (RD_B ? 1 : 0);  // the next unit's operand reads per four MFMAs
                if constexpr (NRD > 0) {
#pragma unroll
                    for (int m = 0; m < UM / 4; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);  // the next unit's b128 reads
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);    // 3 MFMAs
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (nc != c && u + 1 < NU) {
                    mark(seg[1]);
                    if constexpr (FN && nc == NC - 1) fnz = *reinterpret_cast<const f32x4c*>(nzb + tid * 4);  // this phase's normals
                    __syncthreads();  // B_(c+1): chunk c's buffer may be refilled; chunk c + 2 is staged
                    if constexpr (CCVM_CLUSTER_ABL & 64) {
                        const unsigned long long before = t_last;
                        mark(seg[3]);
                        seg[(c < 2) ? 4 + c : 6] += t_last - before;
                    }
                }
            }, std::make_integer_sequence<int, NU>{});
            mark(seg[1]);
            // the next step's schedule row: requested here, ahead of ~1000 cycles of VALU work without a single LDS
            // wait -- scalar loads share lgkmcnt with LDS and return out of order, so requested at the top of the
            // iteration the barrier's s_waitcnt lgkmcnt(0) sat out the scalar cache's latency (~600 cycles per step)
            if (s == NSETS - 1) rnext = load_row(min(it + 1, a.nsteps - 1));
            // ---- this step's / the next step's normals -------------------------------------------
            float nz[4] = {0.0f, 0.0f, 0.0f, 0.0f}, nz1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (FN) {
                nz[0] = fnz[0]; nz[1] = fnz[1]; nz[2] = fnz[2]; nz[3] = fnz[3];
            } else if constexpr (MODE == MODE_DL) {
                pair_normals(s, step, it, nz, nz1);
            } else if constexpr (MODE == MODE_MF) {
                if (has_next) stream_normals(s, step + 1, it + 1, nz);
            } else {
                stream_normals(s, step, it, nz);
            }
            float qx[4], qx1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qx[i] = __builtin_fmaf(a.in_scale, acc[0][0][i] + acc[0][1][i], shift_j);
                qx1[i] = __builtin_fmaf(a.in_scale, acc[NPL - 1][0][i] + acc[NPL - 1][1][i], shift_j);
            }

            AdamScalars ad;
            if constexpr (ADAM) {
                ad.beta1 = a.ad.beta1; ad.one_m_beta1 = a.ad.one_m_beta1; ad.inv_bc1 = trow[12];
                ad.beta2 = a.ad.beta2; ad.one_m_beta2 = a.ad.one_m_beta2; ad.inv_bc2 = trow[13];
                ad.alpha = a.ad.alpha; ad.eps = a.ad.eps; ad.use_v = a.ad.use_v; ad.add_assign = a.ad.add_assign;
            }
            auto adam = [&](float gr, int i) {
                if constexpr (ADAM) {
                    float m, v;
                    const float out = adam_precondition(ad, gr, am[s][i], av[s][i], m, v);
                    am[s][i] = m;  // lanes outside B x N carry don't-care values: never published (0) nor written back
                    av[s][i] = v;
                    return out;
                } else {
                    return gr;
                }
            };

            // ---- update (pinned arithmetic of ccvm_common.h) ---------------------------------------------
            if constexpr (MODE == MODE_DL) {
                const DlScalars k = *reinterpret_cast<const DlScalars*>(trow);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float cn, sn;
                    dl_update(k, s0[s][i], s1[s][i], qx[i], qx1[i], vj, nz[i], nz1[i], cn, sn);
                    s0[s][i] = cn;  // lanes outside B x N carry don't-care values: never published (0) nor written back
                    s1[s][i] = sn;
                }
                const unsigned tag = has_next ? (unsigned)step + 2u : 0u;
                mark(seg[2]);
                publish(s, (it + 1) & 1, s0[s], tag);
                publish(s, (it + 1) & 1, s1[s], tag, 1);
            } else if constexpr (MODE == MODE_MF) {
                const MfScalars k = *reinterpret_cast<const MfScalars*>(trow);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float bound = a.s_cols ? sat_j : k.S;
                    const float fb = adam(__builtin_fmaf(k.f_q, qx[i], k.f_v * vj) * inv_sat_j, i);
                    float mun, sgn;
                    mf_update(k, s0[s][i], s1[s][i], fb, wc[s][i], mun, sgn);
                    s0[s][i] = mun;
                    s1[s][i] = sgn;
                    // the last step's input is what mu_tilde_out returns: no new measurement after it
                    const bool nxt = has_next;
                    mt[s][i] = nxt ? clampf(__builtin_fmaf(k.k_next, nz[i], s0[s][i]), -bound, bound) : mt[s][i];
                    wc[s][i] = nxt ? nz[i] : wc[s][i];
                }
                // published every step (no branch around the stores); after the last step with tag 0, which nobody
                // waits for
                mark(seg[2]);
                publish(s, (it + 1) & 1, mt[s], has_next ? (unsigned)step + 2u : 0u);
            } else {
                const LvScalars k = *reinterpret_cast<const LvScalars*>(trow);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gr = adam(__builtin_fmaf(k.g_q, qx[i], k.g_v * vj) * inv_sat_j, i);
                    s0[s][i] = lv_update(k, s0[s][i], gr, nz[i], a.s_cols ? sat_j : k.S);
                }
                mark(seg[2]);
                publish(s, (it + 1) & 1, s0[s], has_next ? (unsigned)step + 2u : 0u);
            }
            mark(seg[7]);
        }
    }

    if constexpr (CCVM_CLUSTER_ABL & 64) {
        if (tid == 0)
            for (int k = 0; k < 8; ++k) a.dbg[(size_t)blockIdx.x * 16 + k] = seg[k];
        // the other MFMA waves' wait at B_0 (slots 10, 11, 14) and wave 3's noise + update (15)
        if (lane == 0 && wave > 0) a.dbg[(size_t)blockIdx.x * 16 + (wave == 1 ? 10 : wave == 2 ? 11 : 14)] = seg[0];
        if (lane == 0 && wave == 3) a.dbg[(size_t)blockIdx.x * 16 + 15] = seg[2];
    }
    // ---- write the state back (owner-only data: plain stores) ---------------------------------------
#pragma unroll
    for (int s = 0; s < NSETS; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!ok[s][i]) continue;
            a.x0[gidx(s, i)] = s0[s][i];
            if constexpr (MODE == MODE_DL) a.x1[gidx(s, i)] = s1[s][i];
            if constexpr (MODE == MODE_MF) {
                a.x1[gidx(s, i)] = s1[s][i];
                if (a.xt) a.xt[gidx(s, i)] = mt[s][i];
            }
            if constexpr (ADAM) {
                a.am[gidx(s, i)] = am[s][i];
                if (a.ad.use_v) a.av[gidx(s, i)] = av[s][i];
            }
        }
}

template <int MODE, bool ADAM, int KCH, bool REPLAY>
__global__ __launch_bounds__(CL_THREADS) void cluster_kernel(const ClusterArgs a) {
    cluster_body<MODE, ADAM, KCH, REPLAY, false, (KCH > 4) ? 3 : 2>(a);
}

// K = 128 KCH - 64 (an odd number of members)
template <int MODE, bool ADAM, int KCH, bool REPLAY>
__global__ __launch_bounds__(CL_THREADS) void cluster_kernel_half(const ClusterArgs a) {
    cluster_body<MODE, ADAM, KCH, REPLAY, true, (KCH > 4) ? 3 : 2>(a);
}

// K = 640 / 768 in clusters of 32 rows (two row sets)
template <int MODE, bool ADAM, int KCH, bool REPLAY>
__global__ __launch_bounds__(CL_THREADS) void cluster_kernel_2sets(const ClusterArgs a) {
    static_assert(KCH > 4, "K <= 512 always runs two row sets");
    cluster_body<MODE, ADAM, KCH, REPLAY, false, 2>(a);
}
template <int MODE, bool ADAM, int KCH, bool REPLAY>
__global__ __launch_bounds__(CL_THREADS) void cluster_kernel_half_2sets(const ClusterArgs a) {
    static_assert(KCH > 4, "K <= 512 always runs two row sets");
    cluster_body<MODE, ADAM, KCH, REPLAY, true, 2>(a);
}

// K > 512 is served for every solver variant: with half-chunk operand units the MFMA waves' registers suffice (the
// tightest, MF + Adam at K = 768: 255 of 256)
constexpr bool cluster_wide_ok(int, bool) { return true; }
// the half-chunk variant serves N mod 128 in 1 .. 64 (an odd number of 64-column members); `off`: tuning, the full kernel
inline bool cluster_half(int N, int off) { return !off && (((N + CL_COLS - 1) / CL_COLS) & 1) != 0; }

void cluster_launch_dl(const ClusterArgs& a, hipStream_t st);
void cluster_launch_mf(const ClusterArgs& a, bool adam, hipStream_t st);
void cluster_launch_lv(const ClusterArgs& a, bool adam, hipStream_t st);

template <int MODE, bool ADAM, bool REPLAY>
void launch_cluster_variant(const ClusterArgs& a, int grid, hipStream_t st) {
    const int kch = a.ld / CL_KC;
    if (kch > 4 && a.sets == 2) {
        if (cluster_half(a.N, a.half_off)) {
            if (kch == 5) hipLaunchKernelGGL((cluster_kernel_half_2sets<MODE, ADAM, 5, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
            else hipLaunchKernelGGL((cluster_kernel_half_2sets<MODE, ADAM, 6, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
        } else {
            if (kch == 5) hipLaunchKernelGGL((cluster_kernel_2sets<MODE, ADAM, 5, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
            else hipLaunchKernelGGL((cluster_kernel_2sets<MODE, ADAM, 6, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
        }
        return;
    }
    if (cluster_half(a.N, a.half_off)) {
        if (kch == 3) hipLaunchKernelGGL((cluster_kernel_half<MODE, ADAM, 3, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
        else if (kch == 4) hipLaunchKernelGGL((cluster_kernel_half<MODE, ADAM, 4, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
        else if (kch == 5) hipLaunchKernelGGL((cluster_kernel_half<MODE, ADAM, 5, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
        else if (kch == 6) hipLaunchKernelGGL((cluster_kernel_half<MODE, ADAM, 6, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
        return;
    }
    if (kch == 3) hipLaunchKernelGGL((cluster_kernel<MODE, ADAM, 3, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
    else if (kch == 4) hipLaunchKernelGGL((cluster_kernel<MODE, ADAM, 4, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
    // K = 640 / 768: three row sets + 32 / 64 registers of Q per MFMA wave
    if constexpr (cluster_wide_ok(MODE, ADAM)) {
        if (kch == 5) hipLaunchKernelGGL((cluster_kernel<MODE, ADAM, 5, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
        else if (kch == 6) hipLaunchKernelGGL((cluster_kernel<MODE, ADAM, 6, REPLAY>), dim3(grid), dim3(CL_THREADS), 0, st, a);
    }
}

// clusters of one launch: all of them where they fit the chip (spread: by construction), else whole clusters per XCD
inline int cluster_round(const ClusterArgs& a) {
    if (a.spread) return a.nclusters;
    const int xcds = a.xcds > 0 ? a.xcds : 8, cus = a.cus > 0 ? a.cus : 256;
    const int per_xcd = cus / xcds / a.G;
    return per_xcd > 0 ? xcds * per_xcd : xcds;  // (a member count beyond an XCD's CUs is not planned: want_cluster)
}

template <int MODE>
void launch_cluster(const ClusterArgs& args, bool adam, hipStream_t st) {
    const int per_round = cluster_round(args);
    for (int c0 = 0; c0 < args.nclusters; c0 += per_round) {
        ClusterArgs a = args;
        a.cluster0 = c0;
        const int count = args.nclusters - c0 < per_round ? args.nclusters - c0 : per_round;
        const bool last = c0 + per_round >= args.nclusters;
        const int grid = (a.spread ? count * a.G : ((count + 7) / 8) * 8 * a.G) - (last ? a.drop : 0);
        if constexpr (MODE == MODE_DL) {
            if (a.replay) launch_cluster_variant<MODE, false, true>(a, grid, st);
            else launch_cluster_variant<MODE, false, false>(a, grid, st);
        } else if (adam) {
            if (a.replay) launch_cluster_variant<MODE, true, true>(a, grid, st);
            else launch_cluster_variant<MODE, true, false>(a, grid, st);
        } else {
            if (a.replay) launch_cluster_variant<MODE, false, true>(a, grid, st);
            else launch_cluster_variant<MODE, false, false>(a, grid, st);
        }
    }
}

}  // namespace ccvm
