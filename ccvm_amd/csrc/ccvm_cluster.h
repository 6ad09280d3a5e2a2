// Column-cluster persistent kernel for the mid-size regime (256 < N <= 512): a whole chunk of time
// steps in ONE launch for the one-stream solvers (MF, Langevin / pumped Langevin, Adam variants).
//
// Why: at N = 500, B = 1000 a per-step launch of the tile kernel takes 8.2 us for 3.1 us of MFMA work
// (tools/ablate_mid.hip): the kernel boundary (~1.8 us), the first tiles' fill, the store drain and a
// Q panel re-streamed L2 -> LDS every step (192 KB per workgroup and step) are paid once per STEP.  The
// persistent row-owner kernel (ccvm_persist.h) cannot take over: above N = 256 a workgroup can hold
// neither Q's fragments in registers nor all of Q in LDS.
//
// Here a CLUSTER of G = ceil(N / 64) workgroups owns 32 batch rows for the whole launch.  Member m
// keeps the Q panel of its 64 output columns resident in LDS (K x 64 floats, <= 128 KB, loaded once
// per launch), owns the elements (row, its 64 columns) in registers, and per step needs the cluster's
// full GEMM input rows (32 x K), of which the other members produce 7/8.  Exchange, per the gfx950
// inter-workgroup rules (MI355X_MICROARCH.md, visibility, valid forms: row 1 of the sc1 table):
//   * every member publishes its 64 new input columns with 16-byte sc1 (write-through) stores, each
//     128-byte line written whole by one store instruction; every storing wave drains (vmcnt(0)), the
//     workgroup barriers, ONE lane adds 1 to the cluster's counter (agent-scope atomic);
//   * a reader polls that counter with sc1 loads from ONE lane (bounded spin), the workgroup barriers,
//     then EVERY load of the exchanged bytes is a 16-byte sc1 buffer load to registers.
//   Ping-pong buffers (input of even / odd steps): one counter add per step is the only barrier.
// Latency hiding: the 32 rows are TWO independent row sets of 16 (v_mfma_f32_16x16x4_f32 tiles).  While
// the MFMAs of one set run, the other set's new input (published by the peers one phase earlier) is
// already travelling into registers, so the exchange latency (~1.3-1.6 us) hides behind ~2 us of
// matrix work instead of adding to it.
//
// Deadlock freedom does not need the whole grid resident: workgroups are dispatched in order, a
// cluster's members are consecutive in their XCD's dispatch order (blocks b, b + 8, ... share an XCD:
// speed only), so at most one cluster per XCD is ever partially resident and every complete cluster
// runs to the end of the launch without waiting for anything unplaced.  Every spin is bounded all the
// same: on a timeout the workgroup sets the launch's status word and leaves (the host raises).
//
// Per wave: 16 of the member's 64 columns, both row sets, full K.  K order is natural (k = 4 t + g for
// MFMA t, lane group g).  LDS operand layout: row strides == 4 (mod 32) floats and rows 8..15 of every
// 16 shifted by 2 floats, so the 32 lanes of a half-wave (16 rows x 2 k residues) of every ds_read_b32 hit
// 32 distinct banks (4 r + 2 (r >> 3) + g; with the plain stride-4 layout rows r and r + 8 collided and
// SQ_LDS_BANK_CONFLICT was a quarter of the kernel: profiles/r02_langevin_n500_b1000_pmc.json history).  Same noise definition, folded affine map and pinned update
// arithmetic as the other two kernels; only the summation order of the contraction differs.
#pragma once
#include "ccvm_persist.h"

namespace ccvm {

constexpr int CL_COLS = 64;     // output columns per workgroup (cluster member)
constexpr int CL_ROWS = 16;     // rows per row set (one 16x16x4 tile height); two sets per cluster
constexpr int CL_KC = 128;      // K chunk staged through LDS per barrier
constexpr int CL_MIN_N = 257, CL_MAX_N = 512;
constexpr unsigned CL_SPIN_LIMIT = 1u << 22;  // x ~0.3 us per poll: ~1 s

struct ClusterArgs {
    const float* Q;      // [ld][ld] (the row-scaled copy with a per-variable saturation)
    const float* V;
    const float* qsum;
    float* x0;           // Langevin: c;  MF: mu        (pitched, in/out; owner-only data)
    float* x1;           // MF: sigma
    float* xt;           // MF: measured amplitude fed to the LAST step of this launch (out, may be NULL)
    float* xb0;          // exchange buffers: GEMM input of even / odd steps of this launch (pitched, zeroed
    float* xb1;          //   by the host before the launch; columns >= 64 G are never written)
    float* am;           // Adam moments (in/out)
    float* av;
    const float* table;  // [nsteps][TABLE_WORDS] schedule rows (the persistent kernel's tables)
    const float* w0;     // REPLAY noise for the chunk: [nsteps][N][B]
    unsigned* sync;      // [nclusters][2][32] counters (one 128-byte line each), zeroed before the launch
    unsigned* status;    // 0 = ok; set to 1 when a bounded spin gave up
    unsigned long long* dbg;  // ablation stamps only
    uint64_t seed;
    int64_t row_offset;
    int step0, nsteps;
    int replay;
    int B, N, ld;
    int nclusters, G;
    float in_scale, in_shift;
    float k_first;       // MF: sqrt(1 / (4 j_step0)) / sqrt(dt)
    float S;             // MF: clamp of the measured amplitude
    const float* s_cols; // per-variable saturation S_j (length ld) or NULL
    AdamConsts ad;
};

// Ablation bits for tools/cluster_ablate.hip (0 in the product; timing only, results are wrong): 1 no MFMA,
// 2 no noise, 4 no waiting at the polls, 8 no exchange loads, 16 no publish (stores, drain, signal),
// 32 no LDS operand reads.
#ifndef CCVM_CLUSTER_ABL
#define CCVM_CLUSTER_ABL 0
#endif

// Where the exchange rides on the chunk barriers of the NEXT phase (K = 512: chunks 0..3; tuned with
// tools/cluster_ablate.hip): the previous phase's publish is drained and signalled after chunk CCVM_CL_X, the
// other set's next input is polled for and fetched after chunk CCVM_CL_Y (> X: a workgroup must signal before
// it polls, or every member waits for signals nobody has sent).  CCVM_CL_WAVESIG: every storing wave signals
// for itself right after its own drain (counter target 4 G per input) instead of one lane behind a barrier.
// Measured at N = 500, B = 1000 (us per step): (X, Y) = (0, 1) 6.61, (0, 2) 6.46, (1, 2) 6.52; per-wave
// signalling and the poll's sleep length (0 / 1 / 4) change nothing.
#ifndef CCVM_CL_X
#define CCVM_CL_X 0
#endif
#ifndef CCVM_CL_Y
#define CCVM_CL_Y 2
#endif
// CCVM_CL_WIDE = 1: operands read with ds_read_b128 -- MFMA t of a chunk takes k = 128 c + 32 g + t for lane group g
// (each lane's 32 operands of a chunk are contiguous: 8 reads instead of 32); 0: natural k = 4 t + g with
// ds_read_b32 and the shifted conflict-free layout.  Same sums either way up to the order of the additions.
// Conflict-free image for ds_read_b128 (MI355X_MICROARCH.md, LDS: four 16-lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31} and the same + 32; bank row = 16 slots of 16 bytes): a group mixes rows {0-3,12-15} of lane
// group g with rows {4-11} of lane group g + 1, whose k segments sit 8 slots apart -> rows are stored in the order
// pi(r) (rows {0-3,12-15} on even positions, {4-11} on odd ones; row stride == 1 slot mod 16), and the Q panel keeps
// each lane group's k in its own 128-float block ([g][chunk][t]).  With the plain image every read was 2-way
// conflicted (SQ_LDS_BANK_CONFLICT 2081 cycles per CU and step).
#ifndef CCVM_CL_WIDE
#define CCVM_CL_WIDE 1
#endif
#ifndef CCVM_CL_SLEEP
#define CCVM_CL_SLEEP 4
#endif
// CCVM_CL_LL = 1: flag-in-data exchange.  Every exchanged element is an 8-byte pair {value, tag} (tag = global step
// number of the GEMM it feeds + 1; buffers zeroed before the call), written by ONE lane with ONE 8-byte sc1 store and
// read as half of a 16-byte sc1 load: a reader that finds the expected tag has the value that was stored with it.
// No drain, no counter, no poll, no barrier for the hand-off: the chain store -> counter -> poll -> load (three
// memory round trips) becomes store -> load.  A lane whose packets have not all landed re-issues its loads (bounded).
// Ping-pong safety is the counter protocol's: a member publishes input j + 1 only after it has consumed input j from
// every member, so nobody overwrites input j (with j + 2) while anybody still reads it, and every stale tag a reader
// can meet is SMALLER than the one it waits for (min over the tags == expected  <=>  all arrived).
#ifndef CCVM_CL_LL
#define CCVM_CL_LL 1
#endif
// LL: the other set's input is fetched after chunk CCVM_CL_LLY of a phase (its stores were issued at the end of the
// previous phase)
#ifndef CCVM_CL_LLY
#define CCVM_CL_LLY 1
#endif
#ifndef CCVM_CL_WAVESIG
#define CCVM_CL_WAVESIG 0
#endif

// ABL bit 64 (diagnostic build): s_memtime stamps around the phase's segments, accumulated per workgroup into
// a.dbg[block][8]: 0 stage (incl. waiting for the input's loads), 1 chunks before X, 2 drain wait, 3 barrier at X,
// 4 chunks X..Y, 5 poll wait, 6 barrier at Y + load issue, 7 remaining chunks + epilogue + publish.
__device__ __forceinline__ unsigned long long cl_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));

// KCH = K / 128 (3 or 4): K = ld = N rounded up to 128.
// REPLAY is a template parameter, not a run-time branch: with the replay loads in the same code as the fused
// noise hipcc guards the registers they share with s_waitcnt vmcnt(0) in BOTH paths, which made every epilogue wait
// for the other set's input in flight (~0.5 us per phase) -- the very latency the two row sets exist to hide.
template <int MODE, bool ADAM, int KCH, bool REPLAY>
__global__ __launch_bounds__(256) void cluster_kernel(const ClusterArgs a) {
    static_assert(MODE == MODE_MF || MODE == MODE_LANGEVIN, "cluster kernel: one-stream solvers");
    static_assert(KCH == 3 || KCH == 4, "K = 384 or 512");
    constexpr int K = KCH * CL_KC;
    // panel row stride (floats): == 4 (mod 32); the wide image keeps a 128-float block per lane group also at K = 384
    constexpr int QS = (CCVM_CL_WIDE ? 512 : K) + 4;
    constexpr int AS = CL_KC + 4;    // A chunk row stride (floats): == 4 (mod 32)
    constexpr int ABUF = CL_ROWS * AS + 4;  // + the 2-float shift of rows 8..15
    // bank de-conflicting shift of row / column r (ds_read_b32 layout only: b128 accesses must stay 16-byte aligned
    // and are conflict-free per group of 8 consecutive lanes with the stride alone)
    auto shift = [](int r) { return CCVM_CL_WIDE ? 0 : 2 * ((r >> 3) & 1); };
    // wide image: position of row r in an A buffer -- rows 0-3, 12-15 -> 0, 2, .., 14; rows 4-11 -> 1, 3, .., 15
    auto rowpos = [](int r) {
        if constexpr (!CCVM_CL_WIDE) return r;
        return (r < 4) ? 2 * r : (r < 12) ? 2 * (r - 4) + 1 : 2 * (r - 8);
    };
    // wide image: position of Q[k][.] inside a panel column: k = 128 c + 32 g + t  ->  128 g + 32 c + t
    auto kpos = [](int k) {
        if constexpr (!CCVM_CL_WIDE) return k;
        return 128 * ((k >> 5) & 3) + 32 * (k >> 7) + (k & 31);
    };
    constexpr int TS = CL_COLS + 4;  // publish tile row stride
    // one array (a second __shared__ object can de-pipeline the loop, guide section 5)
    constexpr int QPANEL = CL_COLS * QS + 4;
    __shared__ __attribute__((aligned(16))) float lds[QPANEL + 3 * ABUF + CL_ROWS * TS + 4];
    float* const qp = lds;                    // [64 columns][K + 4] (+ shift)
    float* const abuf = lds + QPANEL;         // 3 x [16 rows][128 + 4] (+ shift)
    float* const tile = abuf + 3 * ABUF;      // [16 rows][64 + 4]: a set's new input on its way out
    // lds[DEAD] != 0: a bounded spin gave up.  Written by thread 0 before a barrier, read by everyone behind it
    // (plain LDS accesses: a generic or volatile access would wait for the exchange loads in flight)
    constexpr int DEAD = QPANEL + 3 * ABUF + CL_ROWS * TS;

    // ---- who am I -----------------------------------------------------------------------------------
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;      // lane group: k residue of the operands, row quad of the results
    const int c16 = lane & 15;
    const int G = a.G;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int cluster = (idx / G) * 8 + xcd;
    const int member = idx % G;
    if (cluster >= a.nclusters) return;  // a whole cluster out of range: nobody waits for it
    const int N = a.N, ld = a.ld;
    const int col0 = member * CL_COLS;
    const int col = col0 + 16 * wave + c16;        // this lane's output column
    const bool col_ok = col < N;
    const int crow0 = cluster * 2 * CL_ROWS;       // first batch row of the cluster
    unsigned* const ctr0 = a.sync + (size_t)cluster * 64;  // counters of row set 0 / 1: own 128-byte lines
    if (tid == 0) lds[DEAD] = 0.0f;

    // ---- Q panel, resident for the whole launch: qp[c][k] = Q[k][col0 + c] --------------------------
    {
        const int c = tid & 63, kk = tid >> 6;
#pragma unroll 8
        for (int k = kk; k < K; k += 4) qp[c * QS + shift(c) + kpos(k)] = a.Q[(size_t)k * ld + col0 + c];
    }
    const float vj = col_ok ? a.V[col] : 0.0f;
    const float shift_j = a.in_shift * a.qsum[col];
    const float sat_j = (a.s_cols && col_ok) ? a.s_cols[col] : 1.0f;
    const float inv_sat_j = a.s_cols ? 1.0f / sat_j : 1.0f;

    // ---- this lane's elements: set s, i = 0..3 -> batch row crow0 + 16 s + 4 g + i, column col ------
    int brow[2][4];
    bool ok[2][4];
    size_t gidx[2][4];
    float s0[2][4], s1[2][4], mt[2][4], wc[2][4], am[2][4], av[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            brow[s][i] = crow0 + CL_ROWS * s + 4 * g + i;
            ok[s][i] = col_ok && brow[s][i] < a.B;
            gidx[s][i] = (size_t)brow[s][i] * ld + col;  // inside the padded arrays for every lane
            s0[s][i] = a.x0[gidx[s][i]];
            s1[s][i] = (MODE == MODE_MF) ? a.x1[gidx[s][i]] : 0.0f;
            mt[s][i] = wc[s][i] = am[s][i] = av[s][i] = 0.0f;
            if constexpr (ADAM) {
                am[s][i] = a.am[gidx[s][i]];
                av[s][i] = a.ad.use_v ? a.av[gidx[s][i]] : 0.0f;
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
                out[i] = ok[s][i] ? a.w0[((size_t)it * N + col) * a.B + brow[s][i]] : 0.0f;
        } else {
            NormalPair pa, pb;
            normal_two_rows_x2(a.seed, a.row_offset + brow[s][0], a.row_offset + brow[s][2], step, col, pa, pb);
            out[0] = pa.n0; out[1] = pa.n1; out[2] = pb.n0; out[3] = pb.n1;
        }
    };

    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
    auto mark = [&](int k) {
        if constexpr (CCVM_CLUSTER_ABL & 64) {
            const unsigned long long t = cl_stamp();
            seg[k] += t - t_last;
            t_last = t;
        }
    };
    // ---- exchange plumbing --------------------------------------------------------------------------
    // buffer descriptors of the two exchange buffers (wave-uniform: kernel arguments only)
    constexpr unsigned XE = CCVM_CL_LL ? 8 : 4;  // bytes per exchanged element
    const size_t xbytes = (size_t)(a.nclusters * 2 * CL_ROWS) * ld * XE;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(a.xb0, 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(a.xb1, 0, (int)xbytes, 0x00020000);
    constexpr int SC1 = 16;  // aux bits of the buffer builtins: sc1

    // Publish set s's new GEMM input x[i] (this lane: rows 4 g + i, column 16 wave + c16 of the member's 64)
    // into exchange buffer `par`: through an LDS tile so that each lane stores 16 bytes and every 128-byte
    // line is written whole by one instruction.  The stores are NOT waited for here: `signal` (drain by every
    // storing wave, barrier, ONE counter add) runs one chunk into the next phase, behind ~0.6 us of MFMAs.
    typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
    // LL: this lane's elements are its own 8-byte packets: rows 4 g + i, column `col` -- a store instruction writes
    // 4 rows x 16 columns x 8 bytes = four whole 128-byte lines; no LDS tile, no barrier
    const unsigned pub_off = (unsigned)(((size_t)(crow0 + 4 * g) * ld + col) * XE);
    auto publish_stores = [&](int s, int par, const float (&x)[4], unsigned tag) {
        if constexpr (CCVM_CL_LL) {
            if constexpr (CCVM_CLUSTER_ABL & 16) return;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x2c v = {__builtin_bit_cast(unsigned, ok[s][i] ? x[i] : 0.0f), tag};
                __builtin_amdgcn_raw_buffer_store_b64(v, par ? rs1 : rs0, pub_off, (CL_ROWS * s + i) * ld * XE, SC1);
            }
            return;
        }
        if constexpr (CCVM_CLUSTER_ABL & 16) { __syncthreads(); return; }
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[(4 * g + i) * TS + 16 * wave + c16] = ok[s][i] ? x[i] : 0.0f;
        __syncthreads();
        const int r = tid >> 4, q4 = tid & 15;
        const u32x4c v = *reinterpret_cast<const u32x4c*>(tile + r * TS + 4 * q4);
        const unsigned off = (unsigned)(((size_t)(crow0 + CL_ROWS * s + r) * ld + col0 + 4 * q4) * sizeof(float));
        __builtin_amdgcn_raw_buffer_store_b128(v, par ? rs1 : rs0, off, 0, SC1);
    };
    auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };  // inline asm: never elided
    auto signal = [&](int s) {  // behind drain() of every wave and a workgroup barrier (or per wave, after its drain)
        if constexpr (CCVM_CLUSTER_ABL & 16) return;
        if (CCVM_CL_WAVESIG ? lane == 0 : tid == 0) __hip_atomic_fetch_add(ctr0 + 32 * s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // wait (one lane, bounded) until every member has published input number j of set s
    auto poll = [&](int s, int j) {
        if constexpr (CCVM_CLUSTER_ABL & (4 | 16)) return;
        if (tid == 0) {
            const unsigned want = (unsigned)G * (unsigned)(j + 1) * (CCVM_CL_WAVESIG ? 4u : 1u);
            unsigned spins = 0;
            while (__hip_atomic_load(ctr0 + 32 * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > CL_SPIN_LIMIT) {
                    __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lds[DEAD] = 1.0f;
                    break;
                }
                __builtin_amdgcn_s_sleep(CCVM_CL_SLEEP);
            }
        }
    };

    // the full input rows of set s from buffer `par`, in flight into registers: chunk c, piece j of this
    // lane = row (tid + 256 j) / 32, floats 4 ((tid + 256 j) % 32) of the chunk's 128
    // LL: piece (c, j) is two 16-byte packets pairs {x0, tag, x1, tag}, {x2, tag, x3, tag}
    struct AReg { f32x4c v[KCH][2]; u32x4c w[CCVM_CL_LL ? KCH : 1][2][2]; };
    const unsigned ld_off = (unsigned)(((size_t)(crow0 + (tid >> 5)) * ld + 4 * (tid & 31)) * XE);
    auto load_a = [&](AReg& ar, int s, int par) {
        if constexpr (CCVM_CLUSTER_ABL & 8) return;
        if constexpr (CCVM_CL_LL) {
#pragma unroll
            for (int c = 0; c < KCH; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        ar.w[c][j][h] = __builtin_amdgcn_raw_buffer_load_b128(
                            par ? rs1 : rs0, ld_off, ((CL_ROWS * s + 8 * j) * ld + CL_KC * c + 2 * h) * XE, SC1);
            return;
        }
#pragma unroll
        for (int c = 0; c < KCH; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = tid + 256 * j, r = p >> 5, q4 = p & 31;
                const unsigned off = (unsigned)(((size_t)(crow0 + CL_ROWS * s + r) * ld + CL_KC * c + 4 * q4) * sizeof(float));
                const u32x4c raw = __builtin_amdgcn_raw_buffer_load_b128(par ? rs1 : rs0, off, 0, SC1);
                ar.v[c][j] = __builtin_bit_cast(f32x4c, raw);
            }
    };
    // LL: have all packets of this lane's pieces arrived?  (stale tags are smaller than `want`, see CCVM_CL_LL; the
    // columns >= 64 G of the last chunk are never published: their tags are ignored, their values stay 0)
    const unsigned pad_tag = (CL_KC * (KCH - 1) + 4 * (tid & 31) >= CL_COLS * G) ? 0xFFFFFFFFu : 0u;
    auto arrived = [&](const AReg& ar, unsigned want) {
        unsigned lo = 0xFFFFFFFFu;
#pragma unroll
        for (int c = 0; c < KCH; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned t0 = ar.w[c][j][h][1], t1 = ar.w[c][j][h][3];
                    lo = (c == KCH - 1) ? min(lo, min(t0 | pad_tag, t1 | pad_tag)) : min(lo, min(t0, t1));
                }
        return __builtin_amdgcn_ballot_w64(lo != want) == 0;
    };
    // LL: wait (bounded) until set s's input `want` is complete in ar -- normally it is when the phase starts
    auto await_a = [&](AReg& ar, int s, int par, unsigned want) {
        if constexpr (!CCVM_CL_LL || (CCVM_CLUSTER_ABL & (4 | 8 | 16))) return;
        // the first check stands alone (straight-line code behind the previous phase's publish stores, which stay in
        // flight: inside the retry loop the merged wait counts would drain them)
        if (__builtin_expect(arrived(ar, want), 1)) return;
        unsigned spins = 0;
        do {
            if constexpr (CCVM_CLUSTER_ABL & 64) seg[7] += 1;
            if (++spins > (CL_SPIN_LIMIT >> 3)) {  // ~2 us per round
                if (lane == 0) {
                    __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lds[DEAD] = 1.0f;  // read by everyone behind the next barrier
                }
                break;
            }
            __builtin_amdgcn_s_sleep(CCVM_CL_SLEEP);
            load_a(ar, s, par);
        } while (!arrived(ar, want));
    };
    auto stage_a = [&](const AReg& ar, int c) {  // chunk c -> A buffer c % 3
        float* dst = abuf + (c % 3) * ABUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = tid + 256 * j, r = p >> 5, q4 = p & 31;
            float* d = dst + rowpos(r) * AS + shift(r) + 4 * q4;
            if constexpr (CCVM_CL_LL) {
                static_assert(CCVM_CL_WIDE, "LL exchange: wide image only");
                const u32x4c lo = ar.w[c][j][0], hi = ar.w[c][j][1];
                const u32x4c x = {lo[0], lo[2], hi[0], hi[2]};
                *reinterpret_cast<f32x4c*>(d) = __builtin_bit_cast(f32x4c, x);
            } else if constexpr (CCVM_CL_WIDE) {
                *reinterpret_cast<f32x4c*>(d) = ar.v[c][j];
            } else {  // rows 8..15 sit 8 bytes off the 16-byte grid: two 8-byte writes
                typedef float f32x2c __attribute__((ext_vector_type(2)));
                *reinterpret_cast<f32x2c*>(d) = f32x2c{ar.v[c][j][0], ar.v[c][j][1]};
                *reinterpret_cast<f32x2c*>(d + 2) = f32x2c{ar.v[c][j][2], ar.v[c][j][3]};
            }
        }
    };

    // ---- first inputs: x(step0) of both sets ---------------------------------------------------------
    __syncthreads();  // the panel is in LDS, `dead` is initialised
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if constexpr (MODE == MODE_MF) {
            stream_normals(s, a.step0, 0, wc[s]);  // mf_solver.py:551-554 for the first step of the launch
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float bound = a.s_cols ? sat_j : a.S;
                mt[s][i] = ok[s][i] ? clampf(__builtin_fmaf(a.k_first, wc[s][i], s0[s][i]), -bound, bound) : 0.0f;
            }
            publish_stores(s, 0, mt[s], (unsigned)a.step0 + 1u);
        } else {
            publish_stores(s, 0, s0[s], (unsigned)a.step0 + 1u);
        }
        if constexpr (!CCVM_CL_LL) {
            drain();
            if (CCVM_CL_WAVESIG) signal(s);
            __syncthreads();
            if (!CCVM_CL_WAVESIG) signal(s);
        }
    }

    AReg ar[2];
    if constexpr (!CCVM_CL_LL) {
        poll(0, 0);
        __syncthreads();
        if (lds[DEAD] != 0.0f) return;
    }
    load_a(ar[0], 0, 0);  // LL: whatever has not landed yet is fetched again by await_a

    // operand read addresses: A row c16, B column 16 wave + c16, k residue g
    // operand m of a chunk sits at rd[OPS * m]: natural order k = 4 m + g, or (wide) k = 32 g + m, contiguous per lane
    // chunk c's B operands start at b_rd + BCH * c
    constexpr int BCH = CCVM_CL_WIDE ? 32 : CL_KC;
    const float* const a_rd = abuf + rowpos(c16) * AS + shift(c16) + (CCVM_CL_WIDE ? 32 : 1) * g;
    const float* const b_rd = qp + (16 * wave + c16) * QS + shift(c16) + (CCVM_CL_WIDE ? 128 : 1) * g;
    float bq[2][32];  // B operands of a chunk (double-buffered across chunks); the first chunk's now
    auto read_ops = [&](float (&dst)[32], const float* src) {  // 32 operands of one chunk
        if constexpr (CCVM_CL_WIDE) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4c v = *reinterpret_cast<const f32x4c*>(src + 4 * q);
                dst[4 * q] = v[0]; dst[4 * q + 1] = v[1]; dst[4 * q + 2] = v[2]; dst[4 * q + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int m = 0; m < 32; ++m) dst[m] = src[4 * m];
        }
    };
    read_ops(bq[0], b_rd);

    struct Row { float w[TABLE_WORDS]; };
    Row rnext = *reinterpret_cast<const Row*>(a.table);
    bool pending = false;  // publish stores of the previous phase not yet drained / signalled
    if constexpr (CCVM_CLUSTER_ABL & 64) t_last = cl_stamp();

    for (int it = 0; it < a.nsteps; ++it) {
        const int step = a.step0 + it;
        const Row rcur = rnext;
        const float* trow = rcur.w;
        rnext = *reinterpret_cast<const Row*>(a.table + (size_t)min(it + 1, a.nsteps - 1) * TABLE_WORDS);
        const bool has_next = it + 1 < a.nsteps;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // ---- stage this set's first chunks (its input arrived in registers during the previous phase) ----
            if constexpr (CCVM_CL_LL) mark(6);
            await_a(ar[s], s, it & 1, (unsigned)step + 1u);
            if constexpr (CCVM_CL_LL) mark(0);
            stage_a(ar[s], 0);
            stage_a(ar[s], 1);
            stage_a(ar[s], 2);
            const int os = s ^ 1;                       // the other set ...
            const int oj = (s == 0) ? it : it + 1;      // ... needs input number oj next
            const bool fetch = (s == 0) || has_next;
            __syncthreads();
            if constexpr (CCVM_CL_LL) {
                if (lds[DEAD] != 0.0f) return;
                mark(1);
            } else {
                mark(0);
            }

            // ---- acc = X[set rows][:] @ Q[:, this wave's 16 columns] ------------------------------------
            f32x4c acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
            float aq[2][32];
            __builtin_amdgcn_sched_barrier(0);
            read_ops(aq[0], a_rd);  // chunk 0 (buffer 0): the one exposed LDS latency
            __builtin_amdgcn_sched_barrier(0);
            unroll_indices([&](auto c_tag) {
                constexpr int c = decltype(c_tag)::value;
                constexpr int cb = c & 1, nb = cb ^ 1;
                if constexpr (c == 1 && KCH == 4) stage_a(ar[s], 3);  // buffer 0 is free: every wave has passed M(0)'s barrier
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 32; ++m) {
                    // next chunk's operands in the issue shadow of this chunk's MFMAs (after the last chunk:
                    // the B operands of the next phase's first chunk -- the panel never changes); order pinned
                    // below: hipcc otherwise sinks every read to its use and waits for it there
                    if constexpr (!(CCVM_CLUSTER_ABL & 32)) {
                        if constexpr (CCVM_CL_WIDE) {
                            if (m % 4 == 0) {  // one b128 per operand and four MFMAs
                                const f32x4c vb = *reinterpret_cast<const f32x4c*>(b_rd + BCH * ((c + 1) % KCH) + m);
                                bq[nb][m] = vb[0]; bq[nb][m + 1] = vb[1]; bq[nb][m + 2] = vb[2]; bq[nb][m + 3] = vb[3];
                                if constexpr (c + 1 < KCH) {
                                    const f32x4c va = *reinterpret_cast<const f32x4c*>(a_rd + ((c + 1) % 3) * ABUF + m);
                                    aq[nb][m] = va[0]; aq[nb][m + 1] = va[1]; aq[nb][m + 2] = va[2]; aq[nb][m + 3] = va[3];
                                }
                            }
                        } else {
                            bq[nb][m] = b_rd[BCH * ((c + 1) % KCH) + 4 * m];
                            if constexpr (c + 1 < KCH) aq[nb][m] = a_rd[((c + 1) % 3) * ABUF + 4 * m];
                        }
                    } else {
                        bq[nb][m] = bq[cb][m];
                        aq[nb][m] = aq[cb][m];
                    }
                    if constexpr (CCVM_CLUSTER_ABL & 1) {
                        acc0[m & 3] += aq[cb][m] * bq[cb][m];  // keeps the operands live
                    } else {
                        if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[cb][m], bq[cb][m], acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[cb][m], bq[cb][m], acc0, 0, 0, 0);
                    }
                }
                constexpr int NRD = (c + 1 < KCH) ? 2 : 1;  // next chunk's operand reads per MFMA (wide: per 4 MFMAs)
                if constexpr (CCVM_CL_WIDE) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);  // the next chunk's b128 reads
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);    // 3 MFMAs
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < 32; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);  // its DS reads
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // The exchange rides on the chunk barriers, one phase's latencies behind the MFMAs of the next:
                //   after chunk CX: the previous phase's publish stores have had ~0.5 us -> drain, barrier, signal;
                //   after chunk CY: the peers' signals (same point of THEIR phase) have had time to arrive ->
                //     poll, barrier, and the other set's next input starts travelling into registers, with the
                //     remaining chunk(s) and the epilogue (~1.2 us) to land before it is staged.
                constexpr int CX = (KCH == 4) ? CCVM_CL_X : 0, CY = (KCH == 4) ? CCVM_CL_Y : 1;
                static_assert(CX < CY && CY + 1 < KCH, "signal before poll; both on chunk barriers");
                if constexpr (CCVM_CL_LL) {
                    // the peers stored this input at the end of THEIR previous phase, (LLY + 1) chunks ago; it has
                    // the remaining chunks and the epilogue to travel
                    constexpr int LLY = CCVM_CL_LLY < KCH ? CCVM_CL_LLY : KCH - 1;
                    if constexpr (c == LLY) {
                        mark(2);
                        if (fetch) load_a(ar[os], os, oj & 1);
                        mark(3);
                    }
                    if constexpr (c + 1 == KCH) mark(4);
                    if constexpr (c + 1 < KCH) __syncthreads();  // chunk c's buffer may be refilled; chunk c + 1 is complete
                } else if constexpr (c == CX) {
                    mark(1);
                    if (pending) drain();
                    mark(2);
                    if (CCVM_CL_WAVESIG && pending) signal(os);
                    __syncthreads();
                    if (!CCVM_CL_WAVESIG && pending) signal(os);
                    pending = false;
                    mark(3);
                } else if constexpr (c == CY) {
                    mark(4);
                    if (fetch) poll(os, oj);
                    mark(5);
                    __syncthreads();
                    if (lds[DEAD] != 0.0f) return;
                    if (fetch) load_a(ar[os], os, oj & 1);
                    mark(6);
                } else if constexpr (c + 1 < KCH) {
                    __syncthreads();  // chunk c's buffer may be refilled; chunk c + 1 is complete
                }
            }, std::make_integer_sequence<int, KCH>{});
            // KCH chunks flip bq an odd number of times when KCH is odd: bring the next phase's first
            // chunk back to bq[0]
            if constexpr (KCH % 2 == 1) {
#pragma unroll
                for (int m = 0; m < 32; ++m) bq[0][m] = bq[1][m];
            }

            // ---- this step's / the next step's normals (made at the START of the phase they measured 0.2 us
            // per step slower: the ~150 VALU instructions cost the same there and hid nothing) ---------------
            float nz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (MODE == MODE_MF) {
                if (has_next) stream_normals(s, step + 1, it + 1, nz);
            } else {
                stream_normals(s, step, it, nz);
            }
            float qx[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) qx[i] = __builtin_fmaf(a.in_scale, acc0[i] + acc1[i], shift_j);

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
            if constexpr (MODE == MODE_MF) {
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
                    const bool nxt = k.has_next;
                    mt[s][i] = nxt ? clampf(__builtin_fmaf(k.k_next, nz[i], s0[s][i]), -bound, bound) : mt[s][i];
                    wc[s][i] = nxt ? nz[i] : wc[s][i];
                }
                // LL: published every step (a conditional store makes the next tag check drain it); after the last
                // step with tag 0, which nobody waits for
                if (CCVM_CL_LL || has_next) publish_stores(s, (it + 1) & 1, mt[s], has_next ? (unsigned)step + 2u : 0u);
            } else {
                const LvScalars k = *reinterpret_cast<const LvScalars*>(trow);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gr = adam(__builtin_fmaf(k.g_q, qx[i], k.g_v * vj) * inv_sat_j, i);
                    s0[s][i] = lv_update(k, s0[s][i], gr, nz[i], a.s_cols ? sat_j : k.S);
                }
                if (CCVM_CL_LL || has_next) publish_stores(s, (it + 1) & 1, s0[s], has_next ? (unsigned)step + 2u : 0u);
            }
            if constexpr (CCVM_CL_LL) {
                mark(5);
                __syncthreads();               // the A buffers may be restaged: every wave is past its last chunk
                __builtin_amdgcn_sched_barrier(0);  // no part of the next tag check (a wait for the loads) up here
            } else {
                if (has_next) pending = true;  // drained and signalled one chunk into the next phase
                else __syncthreads();          // the A buffers may be restaged: every wave is past its last chunk
            }
            if constexpr (!CCVM_CL_LL) mark(7);
        }
    }

    if constexpr (CCVM_CLUSTER_ABL & 64) {
        if (tid == 0)
            for (int k = 0; k < 8; ++k) a.dbg[(size_t)blockIdx.x * 8 + k] = seg[k];
    }
    // ---- write the state back (owner-only data: plain stores) ---------------------------------------
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!ok[s][i]) continue;
            a.x0[gidx[s][i]] = s0[s][i];
            if constexpr (MODE == MODE_MF) {
                a.x1[gidx[s][i]] = s1[s][i];
                if (a.xt) a.xt[gidx[s][i]] = mt[s][i];
            }
            if constexpr (ADAM) {
                a.am[gidx[s][i]] = am[s][i];
                if (a.ad.use_v) a.av[gidx[s][i]] = av[s][i];
            }
        }
}

void cluster_launch_mf(const ClusterArgs& a, bool adam, hipStream_t st);
void cluster_launch_lv(const ClusterArgs& a, bool adam, hipStream_t st);

template <int MODE, bool ADAM, bool REPLAY>
void launch_cluster_variant(const ClusterArgs& a, int grid, hipStream_t st) {
    if (a.ld == 512) hipLaunchKernelGGL((cluster_kernel<MODE, ADAM, 4, REPLAY>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((cluster_kernel<MODE, ADAM, 3, REPLAY>), dim3(grid), dim3(256), 0, st, a);
}

template <int MODE>
void launch_cluster(const ClusterArgs& a, bool adam, hipStream_t st) {
    const int grid = ((a.nclusters + 7) / 8) * 8 * a.G;
    if (adam) {
        if (a.replay) launch_cluster_variant<MODE, true, true>(a, grid, st);
        else launch_cluster_variant<MODE, true, false>(a, grid, st);
    } else {
        if (a.replay) launch_cluster_variant<MODE, false, true>(a, grid, st);
        else launch_cluster_variant<MODE, false, false>(a, grid, st);
    }
}

}  // namespace ccvm
