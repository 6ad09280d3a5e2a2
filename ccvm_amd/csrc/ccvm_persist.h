// Row-owner persistent kernel for small problems (N <= 256; DL and Langevin without Adam: N <= 320, Langevin + Adam 288, MF 272): a whole chunk of time steps in ONE
// launch, for every solver of the family (DL, MF, Langevin / pumped Langevin, Adam variants).
//
// At these sizes a per-step launch is latency-bound (N=100, B=1000: ~0.3 us of arithmetic per step
// against ~10 us of launch + pipeline fill) and batch rows never interact, so a workgroup can own its
// rows for the entire trajectory with no inter-workgroup traffic at all.
//
// Matrix instruction: v_mfma_f32_4x4x1_16B_f32 -- sixteen independent 4x4 outer products per
// instruction, D[reg r][lane l] += A[lane 4*(l/4) + r] * B[lane l] (layout probed on the chip,
// tools/mfma4x4_probe.hip).  With B = Q[k][column of lane l] and A = X[row r][k] replicated over the
// blocks, one instruction is one k-step of (4 rows) x (64 columns): the same 256 flop/cycle/CU as the
// big tiles, but with M = 4, so a wave owns FEW rows and the batch spreads over all 1024 SIMDs
// (B=1000 DL rows on the 16-row tile of v_mfma_f32_16x16x4_f32 filled 125 of 256 CUs).
//
//   shape (CW, NCG): a wave covers CW columns (16 / 32 / 64; 64 < N <= 96 also three 32-column waves) and RG = 64 / CW row groups of 4 MFMA
//     rows; N > 64 uses NCG = 2 waves side by side (128 columns; N > 128: 3, N > 192: 4) which exchange the state
//     through a double-buffered LDS tile and one barrier per step; N <= 64 is ONE wave per row set.
//   RU (2 or 4) = MFMA rows in use per group: 4 when that still gives every SIMD a wave, else 2
//     (half the per-step VALU chain per wave, twice the waves).
//   KH (1 or 2; 2 with NCG = 2 or 4, round 3) = split of K over two waves: wave (cg, kh) contracts half of the
//     k-steps for all four rows, the halves swap the partial sums of two rows each through LDS (one more barrier)
//     and each finishes -- normals, update, publish -- the rows it received: half the per-step chain per wave
//     WITHOUT idle MFMA rows.  A workgroup is then ONE row set of four waves, and two workgroups share a CU: a
//     wave alone on its SIMD stalls a third of the step (LDS round trip, barrier, dependent-issue latency:
//     2250 cycles for 1540 of issue at DL N = 100), with a second, independent wave those stalls are filled.
//   Rows of a group: DL (c_b0, s_b0, c_b1, s_b1) -- both quadratures of an element and its
//     (W_c, W_s) noise pair live in ONE lane; one-stream solvers: 4 consecutive batch rows, adjacent
//     rows sharing a generator call exactly as in the tile kernel.
//   Q fragments (Q[k][col], k < 16 * NCH, NCH = ceil(N / 16) a template parameter) are loaded once and
//     stay in registers; the state makes a round trip through LDS each step because it is the MFMA A
//     operand.  The contraction is straight-line code: with run-time chunk branches hipcc shuffled
//     the accumulators AGPR <-> VGPR around every branch.
//   Per-step schedule scalars come from a table built on the device in fp64 (schedule kernels below).
//
// Same noise definition, folded affine map and pinned update arithmetic (ccvm_kernels.h helpers) as
// step_kernel; only the summation order of the contraction differs (inside the stated tolerance).
#pragma once
#include "ccvm_common.h"

namespace ccvm {

constexpr int TABLE_WORDS = 16;  // fp32 words per schedule-table row: solver scalars, [12..13] Adam bias corrections

struct AdamConsts {
    float beta1, one_m_beta1, beta2, one_m_beta2, alpha, eps;
    int use_v, add_assign;
};

struct PersistArgs {
    const float* Q;
    const float* V;
    const float* qsum;
    float* x0;          // DL: c;  MF: mu;  Langevin: c      (pitched, in/out)
    float* x1;          // DL: s;  MF: sigma
    float* xt;          // MF: measured amplitude fed to the LAST step of this launch (may be NULL)
    float* am;          // Adam moments (in/out)
    float* av;
    const float* table; // [nsteps][TABLE_WORDS]
    const float* w0;    // REPLAY noise for the chunk: [nsteps][N][B]
    const float* w1;
    uint64_t seed;
    int64_t row_offset;
    int step0, nsteps;
    int replay;
    int B, N, ld;
    int wld;             // REPLAY: pitch of the noise blocks (rows of the WHOLE batch, ccvm_noise::w_ld; >= B)
    float in_scale, in_shift;
    float k_first;      // MF: sqrt(1 / (4 j_step0)) / sqrt(dt)
    float S;            // MF: clamp of the measured amplitude
    const float* s_cols; // per-variable saturation S_j (length ld) or NULL (see StepArgs::s_cols)
    int ru_override;     // host only: 2 / 4 forces the rows in use per group (tuning), 0 = by batch size
    int kh_override;     // host only: 1 / 2 forces the K split off / on (tuning), 0 = by batch size
    int simds;           // host only: SIMDs of the chip the shape is planned for (4 per CU; 0 = 1024)
    int pw_override;     // host only: 1 / 2 forces the noise producer waves off / on (tuning), 0 = by shape and batch size
    int rsw_override;    // host only: 1 / 2 forces the row sets per six-wave workgroup (tuning), 0 = by batch size
    int cw_override;     // host only: 32 / 64 forces three 32-column / two 64-column waves side by side at 64 < N <= 96 (tuning)
    int xs_override;     // host only: 1 / 2 forces equal halves / the unequal K split of the five-waves-side-by-side shape (tuning), 0 = default
    AdamConsts ad;
    unsigned long long* dbg;  // tools/persist_ablate.hip, CCVM_PERSIST_ABL & 16: s_memtime sums, [grid][16]
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

// Ablation bits for tools/persist_ablate.hip (0 in the product): 1 no MFMA, 2 no noise, 4 no update
// arithmetic, 8 no barrier / LDS round trip, 16 s_memtime stamps of the step's segments (consumer wave 0 and
// producer wave 0 of every workgroup, summed over the launch into PersistArgs::dbg).
#ifndef CCVM_PERSIST_ABL
#define CCVM_PERSIST_ABL 0
#endif

// acc[k & 3] += A(af[k / KC], block k % KC of each row group) x B(qf[k]) for k = 0 .. sizeof...(K) - 1
// (CBSZ / ABID are instruction immediates, hence the pack expansion)
template <int CBSZ, int KC, int... K>
__device__ __forceinline__ void mfma_chain(const float* af, const float* qf, f32x4v* acc,
                                           std::integer_sequence<int, K...>) {
    ((acc[K & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(af[K / KC], qf[K], acc[K & 3], CBSZ, K % KC, 0)), ...);
}

// the same for k = OFF .. OFF + sizeof...(I) - 1
template <int CBSZ, int KC, int OFF, int... I>
__device__ __forceinline__ void mfma_chain_at(const float* af, const float* qf, f32x4v* acc,
                                              std::integer_sequence<int, I...>) {
    ((acc[(OFF + I) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(af[(OFF + I) / KC], qf[OFF + I], acc[(OFF + I) & 3], CBSZ,
                                                              (OFF + I) % KC, 0)), ...);
}

// K split: the k-steps OFF .. OFF + sizeof...(I) - 1 of one half; qf holds that half's fragments from index 0
template <int CBSZ, int KC, int OFF, int... I>
__device__ __forceinline__ void mfma_chain_half(const float* af, const float* qf, f32x4v* acc,
                                                std::integer_sequence<int, I...>) {
    ((acc[I & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(af[(OFF + I) / KC], qf[I], acc[I & 3], CBSZ, (OFF + I) % KC, 0)), ...);
}

// K split with part of the fragments in LDS (QL): the k-steps OFF + IB .. OFF + IB + 3 of one half, their fragments in q
template <int CBSZ, int KC, int OFF, int IB, int... J>
__device__ __forceinline__ void mfma_group4(const float* af, f32x4v q, f32x4v* acc, std::integer_sequence<int, J...>) {
    ((acc[(IB + J) & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(af[(OFF + IB + J) / KC], q[J], acc[(IB + J) & 3], CBSZ,
                                                            (OFF + IB + J) % KC, 0)), ...);
}

// K tail (measured and left off): the last chunk of 16 k-steps runs only the groups of four that hold
// a k < N (N = 100: 100 MFMAs per
// step instead of 112; wave-uniform scalar branches behind the straight-line part)
// the next step's normals between this step's LDS publish and its barrier (waves side by side): measured, DL N = 100
// B = 1000, three alternating runs: 0.982 / 1.026 / 0.984 us per step without, 1.014 / 1.016 / 1.018 with -- off.
// (A third placement -- the next step's generator calls in the same straight-line code as this step's update, four
// independent dependency chains instead of two -- measured 0.971 / 0.973 / 0.967 without, 1.000 / 0.991 / 0.995 with:
// the step is not waiting on VALU dependencies either.)
#ifndef CCVM_PERSIST_NOISE_AHEAD
#define CCVM_PERSIST_NOISE_AHEAD 0
#endif
#ifndef CCVM_PERSIST_KTAIL
#define CCVM_PERSIST_KTAIL 0   // measured: DL N = 100 0.967 / 0.980 us per step without, 0.978 / 0.956 with -- no gain, off
#endif
// ... but for ONE-wave row sets (N <= 64) the step is a single wave's serial chain, and every MFMA left out is 10 cycles
// off it (N = 20: 20 instead of 32 per step): on by default there (round 6)
#ifndef CCVM_PERSIST_KTAIL1
#define CCVM_PERSIST_KTAIL1 1
#endif
// THREE waves side by side for 128 < N <= 192 (round 6): with four, the fourth wave of N <= 192 owns no real column at all
// -- it runs its MFMAs on zero fragments and its generator calls for nothing (ccvm_persist_launch.h: persist_shape).
// Same-box A/B of the library (profiles/r06_ab_persist_ncg3.txt, r06_ab_persist_kh_small.txt; bit-identical results: a
// column's arithmetic does not depend on how many waves stand next to its own): at B >= 768 nothing changes -- within 2 %
// either way; the fullest SIMD of a CU holds as many waves as before -- but small batches on whole chains, one workgroup
// per CU, are 2-9 % faster without it (MF N = 144, B <= 512: 1.15 -> 1.04 us per step; DL 0.90 -> 0.88; Langevin 0.95 -> 0.92).
#ifndef CCVM_PERSIST_QL_AHEAD
#define CCVM_PERSIST_QL_AHEAD 2   // fragments in LDS (QL): groups of four read ahead of their MFMAs (3: two registers spilled at N > 288)
#endif
#ifndef CCVM_PERSIST_NCG3
#define CCVM_PERSIST_NCG3 1
#endif

__device__ __forceinline__ unsigned long long persist_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// PW = 1 (round 6): NOISE PRODUCER waves.  The normals of step t + 1 do not depend on the state, and below N = 64 (one
// wave per row set) a batch of 1000 leaves half of the chip's SIMDs without a wave while every wave issues ONE serial
// chain per step -- LDS read -> MFMAs -> generator (~65 dependent VALU instructions) -> update -> LDS write.  With PW a
// workgroup holds as many producer waves as consumer waves: producer p has consumer p's lane -> element map, makes the
// noise set of the NEXT step (the same generator calls: bit-identical results) into a double-buffered LDS slot and
// meets the consumers at ONE workgroup barrier per step; the consumer's chain loses the generator (and, in replay
// mode, the global loads of the noise blocks).
// threads of a workgroup: row sets per workgroup (RSWO, or what fills four wave slots) x waves per row set
constexpr int persist_block_threads(int ncg, int kh, int pw, int rswo) {
    const int wps = ncg * kh * (1 + pw), rsw = rswo ? rswo : (wps > 4 ? 1 : 4 / wps);
    return 64 * wps * rsw < 256 ? 256 : 64 * wps * rsw;
}

// RSWO = 2 (round 6): TWO row sets per workgroup where a row set is six waves (three side by side x two K halves) -- twelve
// waves are three per SIMD, where two six-wave workgroups on a CU put four on two of its SIMDs and two on the others
// (HW_ID of every wave: tools/simd_probe.hip, profiles/r06_simd_probe.txt) and a step costs what the fullest SIMD issues.
// QL > 0 (round 6, FIVE waves side by side: 256 < N <= 320): the LAST QL k-steps of a wave's K half keep their fragments in LDS
// -- ten waves are three on some SIMD, 168 registers each, and 160 fragments + the working set do not fit; the fragments of a
// wave's own column, [k / 4][lane][k % 4], read back four at a time (one ds_read_b128 per four MFMAs, conflict-free) a few
// groups ahead of their MFMAs.
// XS > 0 (five waves side by side only): UNEQUAL K split.  The ten waves of a workgroup land {0, 4, 8} {1, 5, 9} {2, 6} {3, 7} on
// the four SIMDs, in every workgroup (HW_ID of every wave: tools/simd_probe.hip, profiles/r06_simd_probe.txt), and a step
// costs what the fullest SIMD issues: with equal halves two SIMDs issue three half chains and two SIMDs two.  So the waves
// that triple up take LESS of K: column groups 0 ... 3 are split [0, XS) | [XS, K) with the short part on waves 0, 1, 4, 5 and
// the long one on waves 2, 3, 6, 7 (alone in pairs on their SIMDs), column group 4 in equal halves on waves 8, 9 -- every
// SIMD then issues (2 XS + K / 2) = 2 (K - XS) k-steps + its waves' updates.  The short parts live in registers entirely, the
// long ones keep KR = K / 2 - QL in registers like the equal halves and K - XS - KR in LDS.
template <int MODE, bool ADAM, int CW, int NCG, int NCH, int RU, int KH = 1, int PW = 0, int RSWO = 0, int QL = 0, int XS = 0>
__global__ __launch_bounds__(persist_block_threads(NCG, KH, PW, RSWO)) void persist_kernel(const PersistArgs a) {
    static_assert(XS == 0 || (KH == 2 && XS % 4 == 0 && ((NCG == 5 && QL > 0 && XS <= 8 * NCH - QL) || (NCG == 3 && QL == 0 && RSWO == 0 && CW == 64))),
                  "unequal K split: five side by side (the short part in registers, the long one partly in LDS) or three (six waves, {0, 4} {1, 5} {2} {3}: all in registers)");
    static_assert(KH == 1 || (KH == 2 && NCG >= 2 && NCG <= 5 && RU == 4), "K split: waves side by side, all four rows in use");
    static_assert(QL == 0 || (KH == 2 && QL % 4 == 0 && PW == 0 && RSWO == 0), "fragments in LDS: K split only, whole groups of four");
    static_assert(PW == 0 || (PW == 1 && NCG * KH <= 4), "producer waves: at most eight waves per workgroup");
    static_assert(RSWO == 0 || (PW == 0 && 64 * NCG * KH * RSWO <= 1024), "row sets per workgroup: at most sixteen waves, no producers");
    static_assert(MODE == MODE_DL || MODE == MODE_MF || MODE == MODE_LANGEVIN, "persistent kernel: solver loops only");
    static_assert(!(ADAM && MODE == MODE_DL), "DL has no Adam variant (dl_solver.py:571-769 is unreachable)");
    static_assert((CW == 16 || CW == 32 || CW == 64) && (NCG == 1 || (NCG >= 2 && NCG <= 5 && CW == 64) || (NCG == 3 && CW == 32)), "shape");
    static_assert(KH == 1 || CW == 64, "K split: 64-column waves");
    static_assert(RU == 2 || RU == 4, "rows in use per group");
    constexpr int RG = 64 / CW;                                // row groups per wave
    static_assert(NCH >= 1 && 16 * NCH <= CW * NCG, "K chunks vs shape");
    constexpr int KMAX = 16 * NCH;                             // K in use: 16 * ceil(N / 16), compile time so
                                                               // that the contraction is straight-line code
    constexpr int ROWS = RU * RG;                              // MFMA rows per workgroup
    constexpr int NEG = (MODE == MODE_DL) ? RU / 2 : RU;       // batch rows (elements) per lane position of a row group
    constexpr int NE = NEG / KH;                               // of which this wave finishes NE (K split: half)
    constexpr int BR = NEG * RG;                               // batch rows per row set
    constexpr int LDX = CW * NCG + 8;                          // LDS row stride: == 8 (mod 32), the 4 rows x 8 k
                                                               // of a half-wave read hit 32 distinct banks
    constexpr int KC = CW / 4;                                 // k-steps fed by one A register (blocks per row group)
    constexpr int CBSZ = (CW == 64) ? 4 : (CW == 32) ? 3 : 2;  // log2(KC)
    // Row sets per workgroup: a workgroup is four waves = one per SIMD where the row sets divide four, i.e. two two-wave
    // sets (N > 64) or four one-wave sets (three side by side: one row set of three waves; more than four: one row set).
    // Smaller workgroups landed unevenly on the SIMDs (DL N=100: 1.46 vs 0.92 us/step; N=64: 0.89 vs 0.63) and a SIMD with
    // two of these waves takes twice as long.  The sets of a workgroup share nothing (N > 64: but the barrier).
    constexpr int WPS = NCG * KH * (1 + PW);                   // waves per row set, producers included
    constexpr int RSW = RSWO ? RSWO : (WPS > 4) ? 1 : 4 / WPS; // (four waves side by side x two K halves: eight waves)
    constexpr int NWC = RSW * NCG * KH;                        // consumer waves per workgroup
    constexpr int PXF = (KH == 2) ? 2 * NCG * 2 * 64 : 0;      // K split: [kh][cg][2 rows][lane] partial sums, per row set
    constexpr int NV = ((MODE == MODE_DL) ? 2 : 1) * NE;       // normals per lane and step
    constexpr int NZF = PW ? 2 * NWC * NV * 64 : 0;            // producer waves: [step parity][consumer][value][lane]
    // producer waves also relay the schedule rows: ring[step & 3][TABLE_WORDS], the words a solver reads (in fours)
    constexpr int ROWW = ADAM ? 16 : (MODE == MODE_MF) ? 12 : 8;  // (Adam's bias corrections are words 12, 13)
    // (one wave per row set, and Langevin -- one word of its row changes -- with two side by side: measured, us per step at
    // B = 1000 without / with the relay: DL N = 20 0.435 / 0.359, MF 0.449 / 0.411, Langevin 0.427 / 0.371, DL N = 64 0.647 /
    // 0.593, Langevin N = 100 0.658 / 0.595; but MF N = 100 0.80 / 0.84, DL N = 128 0.92 / 0.94: their steps are long enough
    // for the scalar load and the row's vector registers cost more than they save)
    constexpr bool RELAY = PW && (NCG == 1 || MODE == MODE_LANGEVIN);
    constexpr int RING = RELAY ? 4 * TABLE_WORDS : 0;
    // words of a row that are the same in every step of a run (ccvm_schedule.h) -- DL: dt, 2 g; MF: g^2, f_q, f_v, 1 / sqrt(dt),
    // dt, S, has_next (unused here), + the unused words 14, 15; Langevin: all but the pump term
    constexpr unsigned ROW_SAME = (MODE == MODE_DL) ? 0x030u : (MODE == MODE_MF) ? 0xCD8Eu : 0xC0FBu;
    constexpr int PXA = (RSWO ? RSW : 1) * PXF;                // (without RSWO a K-split workgroup is one row set)
    constexpr int QLH = (XS && QL) ? 16 * NCH - XS - (8 * NCH - QL) : 0; // unequal split: the long parts' fragments in LDS (K - XS - KR)
    constexpr int QTF = XS ? (4 * QLH + 2 * QL) * 64 : QL * 64 * NWC;  // fragments kept in LDS: [consumer wave][k / 4][lane][k % 4]
    __shared__ __attribute__((aligned(16))) float xs_all[RSW * 2 * ROWS * LDX + PXA + NZF + RING + QTF];
    float* const nzl = xs_all + RSW * 2 * ROWS * LDX + PXA;
    float* const ring = nzl + NZF;
    float* const qtail = ring + RING;

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Four-wave workgroups put consumers and producers on DIFFERENT SIMDs (wave w -> SIMD w mod 4), and a chip-filling
    // grid runs two workgroups per CU: the second round of workgroups swaps the roles' wave slots, so a SIMD holds one
    // consumer and one producer instead of two consumers (DL N = 100, B = 1000, whole chains: 1.055 us per step without
    // the swap).  Which CU a workgroup lands on is the dispatcher's business: a wrong guess costs time, nothing else.
    const bool swap_roles = PW && WPS <= 4 && ((blockIdx.x / (a.simds > 0 ? a.simds / 4 : 256)) & 1);
    const bool producer = PW && ((wave_all >= NWC) != swap_roles);  // (wave-uniform)
    const int wave = (PW && wave_all >= NWC) ? wave_all - NWC : wave_all;  // a producer has its consumer's index and element map
    const int set = wave / (NCG * KH), wis = wave % (NCG * KH);  // row set of the workgroup, wave inside the row set
    // (unequal split: waves 0 1 4 5 = the short parts of column groups 0 1 2 3, waves 2 3 6 7 their long parts, 8 9 = the halves of 4)
    // (three side by side: waves 0 1 = the short parts of column groups 0 1, waves 2 3 their long parts -- alone on their SIMDs --, 4 5 the halves of 2)
    constexpr int NUW = 2 * (NCG - 1);  // waves of the unequally split column groups
    const int cg = XS ? (wis < NUW ? ((wis >> 2) * 2 + (wis & 1)) : NCG - 1) : wis % NCG;
    const int kh = XS ? (wis < NUW ? ((wis >> 1) & 1) : (wis & 1)) : (KH == 2) ? wis / NCG : 0;  // K half (wave-uniform)
    const int role = XS ? (wis < NUW ? kh : 2 + kh) : kh;  // unequal split: 0 short, 1 long, 2 / 3 the equal halves
    float* const xs = xs_all + set * (2 * ROWS * LDX);
    float* const px = xs_all + RSW * 2 * ROWS * LDX + (RSWO ? set : 0) * PXF;
    const int rs = lane / CW;            // row group
    const int col = cg * CW + (lane % CW);
    const int i4 = lane & 3;             // the A-operand row this lane supplies to its block
    const int bg = (lane % CW) >> 2;     // block inside the row group: the k residue this lane supplies
    const int N = a.N, ld = a.ld;
    const bool col_ok = col < N;

    // ---- Q fragments, resident for the whole launch (rows >= N of the pitched Q are zero) ------
    // (K split: half kh holds the fragments of k = KSPLIT kh ... KSPLIT kh + KSPLIT - 1, those below KMAX)
    constexpr int KSPLIT = (KH == 2) ? (KMAX / 2 + 3) / 4 * 4 : KMAX;  // k-steps per half, a multiple of 4
    constexpr int KQ = (KH == 2) ? KSPLIT : KMAX;
    const int koff = XS ? (role == 0 ? 0 : role == 1 ? XS : (role - 2) * KSPLIT) : kh * KSPLIT;
    constexpr int KR = KQ - QL;  // fragments in registers: the first KR k-steps of the wave's range
    static_assert(KR >= 4, "fragments in LDS: some stay in registers");
    constexpr int KRL = (XS && !QL) ? KMAX - XS : KR;               // the long parts' fragments in registers (no LDS: all of them)
    constexpr int QF = KRL > KR ? KRL : KR;                         // fragment registers of the kernel: the largest role's
    const int nreg = (XS && role == 0) ? XS : (XS && role == 1) ? KRL : KR;  // (the short parts: all of them)
    const int nlds = XS ? (role == 0 ? 0 : role == 1 ? QLH : QL) : QL;  // this wave's fragments in LDS
    float qf[QF];
#pragma unroll
    for (int k = 0; k < QF; ++k) qf[k] = 0.0f;
    if (!producer) {
#pragma unroll
        for (int k = 0; k < QF; ++k) qf[k] = (k < nreg && koff + k < KMAX) ? a.Q[(size_t)(koff + k) * ld + col] : 0.0f;
    }
    // this lane's fragments of group g: qt[g * 256 .. + 3] (unequal split: the four long parts first, then column group 4's halves)
    float* const qt = qtail + (XS ? (role == 1 ? ((wis >> 2) * 2 + (wis & 1)) * (QLH * 64) : 4 * (QLH * 64) + (wis & 1) * (QL * 64))
                                  : wave * (QL * 64)) + lane * 4;
    if constexpr (QL > 0) {
        constexpr int GMAX = (XS ? (QLH > QL ? QLH : QL) : QL) / 4;
#pragma unroll 1
        for (int g = 0; g < GMAX; ++g) {
            if (4 * g >= nlds) break;
            f32x4v q;
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = (koff + nreg + 4 * g + j < KMAX) ? a.Q[(size_t)(koff + nreg + 4 * g + j) * ld + col] : 0.0f;
            *reinterpret_cast<f32x4v*>(qt + g * 256) = q;  // (read back by this lane only: no barrier)
        }
    }
    const float vj = col_ok ? a.V[col] : 0.0f;
    const float shift_j = a.in_shift * a.qsum[col];  // shift * colsum(Q)[j]
    const float sat_j = (a.s_cols && col_ok) ? a.s_cols[col] : 1.0f;  // per-variable saturation
    const float inv_sat_j = a.s_cols ? 1.0f / sat_j : 1.0f;

    // ---- this lane's elements: batch rows brow[e] at column col -------------------------------
    const int row0 = (blockIdx.x * RSW + set) * BR;
    int brow[NE];
    bool ok[NE];
    size_t gidx[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        brow[e] = row0 + rs * NEG + kh * NE + e;
        ok[e] = col_ok && brow[e] < a.B;
        gidx[e] = (size_t)brow[e] * ld + col;  // inside the padded arrays for every lane
    }
    float s0[NE], s1[NE];   // DL: c, s;  MF: mu, sigma;  Langevin: c, -
    float mt[NE], wc[NE];   // MF: measured amplitude (GEMM input) and the normals that made it
    float am[NE], av[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) s0[e] = s1[e] = mt[e] = wc[e] = am[e] = av[e] = 0.0f;
    if (!producer) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            s0[e] = a.x0[gidx[e]];
            s1[e] = (MODE == MODE_LANGEVIN) ? 0.0f : a.x1[gidx[e]];
            if constexpr (ADAM) {
                am[e] = a.am[gidx[e]];
                av[e] = a.ad.use_v ? a.av[gidx[e]] : 0.0f;
            }
        }
    }

    // one-stream normals of this lane's rows at `step` (it = index inside the launch, for replay)
    auto stream_normals = [&](int step, int it, float* out) {
        if (a.replay) {
#pragma unroll
            for (int e = 0; e < NE; ++e) out[e] = ok[e] ? a.w0[((size_t)it * N + col) * a.wld + brow[e]] : 0.0f;
        } else {
            if constexpr (NE == 4) {  // rows (0,1) and (2,3): two generator calls in lockstep
                NormalPair pa, pb;
                normal_two_rows_x2(a.seed, a.row_offset + brow[0], a.row_offset + brow[2], step, col, pa, pb);
                out[0] = pa.n0; out[1] = pa.n1; out[2] = pb.n0; out[3] = pb.n1;
            } else {                  // rows (0,1): local row 0 is even
                const NormalPair p = normal_two_rows(a.seed, a.row_offset + brow[0], step, col);
                out[0] = p.n0;
                out[1] = p.n1;
            }
        }
    };

    // the noise set a step consumes -- DL: (W_c, W_s) per element; MF: the NEXT step's normals; Langevin: this step's
    auto gen_noise = [&](int step, int it, float* nz0, float* nz1) {
        if constexpr (MODE == MODE_DL) {
            if (a.replay) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const size_t w = ((size_t)it * N + col) * a.wld + brow[e];
                    nz0[e] = ok[e] ? a.w0[w] : 0.0f;
                    nz1[e] = ok[e] ? a.w1[w] : 0.0f;
                }
            } else {
                if constexpr (CCVM_PERSIST_ABL & 2) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) nz0[e] = nz1[e] = 0.25f;
                } else if constexpr (NE == 2) {  // two generator calls in lockstep
                    NormalPair pa, pb;
                    normal_pair_x2(a.seed, a.row_offset + brow[0], a.row_offset + brow[1], step, col, pa, pb);
                    nz0[0] = pa.n0; nz1[0] = pa.n1; nz0[1] = pb.n0; nz1[1] = pb.n1;
                } else {
                    const NormalPair p = normal_pair(a.seed, a.row_offset + brow[0], step, col);
                    nz0[0] = p.n0;
                    nz1[0] = p.n1;
                }
            }
        } else if constexpr (MODE == MODE_MF) {
            if (it + 1 < a.nsteps) stream_normals(step + 1, it + 1, nz0);  // (the launch's last step draws none)
        } else {
            stream_normals(step, it, nz0);
        }
    };

    // ---- producer waves: the noise set of step it + 1 while the consumers run step it ------------------------------
    // slot (parity of it, consumer wave): NV values per lane.  The consumers read slot it & 1 during step it and pass the
    // step's last barrier only after the update that consumed it, so the write of step it + 2's set behind that barrier
    // cannot overtake a read; a producer executes exactly the consumers' barriers (one per step, two with the K split).
    unsigned long long st_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;  // (ablation stamps)
    auto stamp = [&](int k) {
        if constexpr (CCVM_PERSIST_ABL & 16) {
            const unsigned long long t = persist_stamp();
            st_sum[k] += t - st_last;
            st_last = t;
        }
    };
    auto stamps_out = [&](int base) {
        if constexpr (CCVM_PERSIST_ABL & 16) {
            if (wave == 0 && lane == 0 && a.dbg)
                for (int k = 0; k < 8; ++k) a.dbg[(size_t)blockIdx.x * 16 + base + k] = st_sum[k];
        }
    };
    if constexpr (PW) {
        if (producer) {
            float o0[NE], o1[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) o0[e] = o1[e] = 0.0f;
            auto put = [&](int parity) {
                float* d = nzl + (parity * NWC + wave) * (NV * 64) + lane;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    d[e * 64] = o0[e];
                    if constexpr (MODE == MODE_DL) d[(NE + e) * 64] = o1[e];
                }
            };
            // The schedule rows travel with the noise: producer wave 0 reads row it + 3 (an ordinary vector load, one word
            // per lane) at the top of step it and puts the row it read a step earlier into ring[(it + 2) & 3] behind its
            // generator calls; the consumers take ring[it & 3] out of LDS with their A operands.  Why: a consumer's own
            // scalar load of the next row has to land by the step's next LDS wait (scalar loads and LDS share lgkmcnt, SMEM
            // returns out of order: every lgkmcnt wait drains it), ~150 cycles at N = 20 -- an L2 hit makes that, a line
            // from the Infinity Cache (545 cycles: the table was written by a kernel on another XCD) does not.  The
            // ablation harness, whose 256 KB table every L2 holds after the first launch, ran this loop at 0.30 us per step
            // where the engine, walking through the rows of a whole run, took 0.43 (round 6: tools/persist_vs_abi.hip).
            const bool relay = RELAY && wave == 0 && lane < TABLE_WORDS;
            auto row_word = [&](int it) {  // word `lane` of the row of step it (clamped: the launch's last rows again)
                return relay ? a.table[(size_t)min(it, a.nsteps - 1) * TABLE_WORDS + lane] : 0.0f;
            };
            if (relay) {
                ring[0 * TABLE_WORDS + lane] = row_word(0);
                ring[1 * TABLE_WORDS + lane] = row_word(1);
            }
            float row_carry = row_word(2);  // -> ring[2] during step 0
            gen_noise(a.step0, 0, o0, o1);
            put(0);
            __syncthreads();
            if constexpr (CCVM_PERSIST_ABL & 16) st_last = persist_stamp();
            for (int it = 0; it < a.nsteps; ++it) {
                const float row_ahead = row_word(it + 3);
                if (it + 1 < a.nsteps) {
                    gen_noise(a.step0 + it + 1, it + 1, o0, o1);
                    put((it + 1) & 1);
                }
                // (slot (it + 2) & 3 held step it - 2's row: every consumer is past that step's last barrier)
                if (relay) ring[((it + 2) & 3) * TABLE_WORDS + lane] = row_carry;
                row_carry = row_ahead;
                stamp(0);
                if constexpr (KH == 2) __syncthreads();
                __syncthreads();
                stamp(1);
            }
            stamps_out(8);
            return;
        }
    }

    // ---- GEMM input of the first step into LDS buffer 0 ----------------------------------------
    if constexpr (MODE == MODE_MF) {
        stream_normals(a.step0, 0, wc);  // mf_solver.py:551-554 for the first step of the launch
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float bound = a.s_cols ? sat_j : a.S;
            mt[e] = ok[e] ? clampf(__builtin_fmaf(a.k_first, wc[e], s0[e]), -bound, bound) : 0.0f;
        }
    }
    auto publish = [&](float* buf) {  // this lane's GEMM-input values into an LDS state tile
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if constexpr (MODE == MODE_DL) {
                buf[(rs * RU + 2 * (kh * NE + e)) * LDX + col] = s0[e];
                buf[(rs * RU + 2 * (kh * NE + e) + 1) * LDX + col] = s1[e];
            } else if constexpr (MODE == MODE_MF) {
                buf[(rs * RU + kh * NE + e) * LDX + col] = mt[e];
            } else {
                buf[(rs * RU + kh * NE + e) * LDX + col] = s0[e];
            }
        }
    };
    publish(xs);
    if constexpr (NCG > 1 || PW) __syncthreads();

    const int arow = rs * RU + (i4 < RU ? i4 : 0);  // rows >= RU of a group are unused: any finite row
    int cur = 0;
    // schedule row of the step, fetched one step ahead (scalar loads share lgkmcnt with the LDS reads:
    // a row requested at the top of the step would be waited for together with the A operands)
    struct Row { float w[TABLE_WORDS]; };
    Row rnext = *reinterpret_cast<const Row*>(a.table);
    const Row first_row = rnext;  // (producer waves: the words of a row that never change are taken from here)
    float nzn0[NE], nzn1[NE];  // the next step's normals (NOISE_AHEAD)
#pragma unroll
    for (int e = 0; e < NE; ++e) nzn0[e] = nzn1[e] = 0.0f;
    // H: this wave's K half, a compile-time constant inside (the MFMA's block select is an immediate): the wave-uniform
    // branch is taken once, in front of the whole loop
    auto run_steps = [&](auto h_tag) {
    constexpr int ROLE = decltype(h_tag)::value;  // (unequal split: 0 short, 1 long, 2 / 3 the equal halves; else the K half)
    constexpr int H = ROLE & 1;
    // this wave's k-steps [K0, K1), of which the first KRR have their fragments in registers
    constexpr int K0 = XS ? (ROLE == 0 ? 0 : ROLE == 1 ? XS : H * KSPLIT) : H * KSPLIT;
    constexpr int K1 = XS ? (ROLE == 0 ? XS : ROLE == 1 ? KMAX : (H * KSPLIT + KSPLIT < KMAX ? H * KSPLIT + KSPLIT : KMAX))
                          : (K0 + KSPLIT < KMAX) ? K0 + KSPLIT : KMAX;
    constexpr int KRR = (XS && ROLE == 0) ? XS : (XS && ROLE == 1) ? KRL : KR;
    for (int it = 0; it < a.nsteps; ++it) {
        const int step = a.step0 + it;
        const Row rcur = rnext;  // (producer waves: unused -- the row comes out of the ring below)
        // ---- acc[r] = X[row rs*RU + r][:] @ Q[:, col] ---------------------------------------------
        const float* xb = xs + cur * ROWS * LDX + arow * LDX + bg;
        f32x4v acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        // A operand through the broadcast controls: with CBSZ = log2(CW / 4) the blocks of a row group
        // all take their A rows from block ABID of that group, so lane (block bg, row i4) only has to
        // hold X[row i4][k] for the k == bg (mod KC) -- ONE ds_read_b32 per lane feeds KC MFMAs (a
        // row-broadcast b128 per 4 k-steps left one read in flight and exposed the LDS latency 28 times).
        float af[16 * NCH / KC];
#pragma unroll
        for (int c = K0 / KC; c <= (K1 - 1) / KC; ++c) af[c] = xb[c * KC];
        float nzp[PW ? NV : 1];  // producer waves: this step's noise set comes out of its LDS slot with the A operands
        float rowv[RELAY ? ROWW : 1];  // ... and so does its schedule row (the same address in every lane: a broadcast read)
        if constexpr (PW) {
            const float* nzr = nzl + ((it & 1) * NWC + wave) * (NV * 64) + lane;
#pragma unroll
            for (int v = 0; v < NV; ++v) nzp[v] = nzr[v * 64];
        }
        if constexpr (RELAY) {
            // (only the words that change from step to step: the others come from the launch's first row, below)
            const float* rr = ring + (it & 3) * TABLE_WORDS;
#pragma unroll
            for (int w = 0; w < ROWW; ++w) rowv[w] = ((ROW_SAME >> w) & 1u) ? 0.0f : rr[w];
        }
        if constexpr (RELAY) {
            // the words that are the same in every row of a run (dt, g, the feedback coefficients, S: ccvm_schedule.h) come
            // from the launch's first row in scalar registers, so that only the words that change hold vector registers
            // (MF: 12 words of the row in VGPRs put the K-split + producers kernel past 128 at N > 96)
#pragma unroll
            for (int w = 0; w < ROWW; ++w)
                if ((ROW_SAME >> w) & 1u) rowv[w] = first_row.w[w];
        }
        const float* trow = RELAY ? rowv : rcur.w;
        __builtin_amdgcn_sched_barrier(0);  // all reads in flight before anything else (one latency, not NCH)
        stamp(0);
        // ---- this step's normals -- DL: (W_c, W_s) per element; MF: the NEXT step's; Langevin: this step's.
        // One-wave row sets make them HERE, while the A reads are in flight (they do not depend on them:
        // DL N=20 0.53 -> 0.46 us/step, N=64 0.63 -> 0.56); with two or four waves side by side the same
        // order was 3-12 % slower (same-box A/B), so those make them after the contraction.
        constexpr bool NOISE_FIRST = (NCG == 1) && !PW;
        // waves side by side: the NEXT step's normals are made between this step's LDS publish and its barrier (below),
        // where the faster wave of a row set would only wait; nzn0 / nzn1 carry them over
        constexpr bool NOISE_AHEAD = CCVM_PERSIST_NOISE_AHEAD && NCG > 1 && MODE != MODE_MF && !PW;
        float nz0[NE], nz1[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) nz0[e] = nz1[e] = 0.0f;
        auto make_step_noise = [&](int step, int it) { gen_noise(step, it, nz0, nz1); };
        if constexpr (NOISE_FIRST) make_step_noise(step, it);
        if constexpr (PW) {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                nz0[e] = nzp[e];
                if constexpr (MODE == MODE_DL) nz1[e] = nzp[NE + e];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(1);
        if constexpr (!(CCVM_PERSIST_ABL & 1)) {
            if constexpr (CCVM_PERSIST_KTAIL || (CCVM_PERSIST_KTAIL1 && NCG == 1)) {
                constexpr int FULL = 16 * (NCH - 1);
                mfma_chain<CBSZ, KC>(af, qf, acc, std::make_integer_sequence<int, FULL + 4>{});
                const int rem = N - FULL;  // 1 .. 16 k-steps of the last chunk are real
                if (rem > 4) mfma_chain_at<CBSZ, KC, FULL + 4>(af, qf, acc, std::make_integer_sequence<int, 4>{});
                if (rem > 8) mfma_chain_at<CBSZ, KC, FULL + 8>(af, qf, acc, std::make_integer_sequence<int, 4>{});
                if (rem > 12) mfma_chain_at<CBSZ, KC, FULL + 12>(af, qf, acc, std::make_integer_sequence<int, 4>{});
            } else if constexpr (KH == 2 && (QL > 0 || XS > 0)) {
                static_assert(XS || K1 - K0 == KQ, "fragments in LDS: equal halves");
                static_assert((K1 - K0 - KRR) % 4 == 0 && K1 - K0 >= KRR, "fragments in LDS: whole groups of four");
                constexpr int G = (K1 - K0 - KRR) / 4, AHEAD = CCVM_PERSIST_QL_AHEAD;  // groups of four k-steps out of LDS, read AHEAD groups ahead of their MFMAs
                f32x4v qb[AHEAD];
#pragma unroll
                for (int g = 0; g < AHEAD && g < G; ++g) qb[g] = *reinterpret_cast<const f32x4v*>(qt + g * 256);
                mfma_chain_half<CBSZ, KC, K0>(af, qf, acc, std::make_integer_sequence<int, KRR>{});  // (covers the first reads)
                unroll_indices([&](auto g_tag) {
                    constexpr int g = decltype(g_tag)::value;
                    mfma_group4<CBSZ, KC, K0, KRR + 4 * g>(af, qb[g % AHEAD], acc, std::make_integer_sequence<int, 4>{});
                    if constexpr (g + AHEAD < G) {
                        qb[g % AHEAD] = *reinterpret_cast<const f32x4v*>(qt + (g + AHEAD) * 256);
                        __builtin_amdgcn_sched_barrier(0);  // (keeps the reads where they are: hoisted, they would all be live at once)
                    }
                }, std::make_integer_sequence<int, G>{});
            } else if constexpr (KH == 2) {
                mfma_chain_half<CBSZ, KC, K0>(af, qf, acc, std::make_integer_sequence<int, K1 - K0>{});
            } else {
                mfma_chain<CBSZ, KC>(af, qf, acc, std::make_integer_sequence<int, 16 * NCH>{});
            }
        } else {
#pragma unroll
            for (int c = K0 / KC; c <= (K1 - 1) / KC; ++c) acc[c & 3][0] += af[c] * qf[c - K0 / KC];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!RELAY) rnext = *reinterpret_cast<const Row*>(a.table + (size_t)min(it + 1, a.nsteps - 1) * TABLE_WORDS);
        // K split: the partial sums of the twin's rows leave before this wave makes its normals (the LDS write and the
        // twin's arrival at the barrier run under them)
        float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (KH == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) part[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
            px[((H * NCG + cg) * 2 + 0) * 64 + lane] = part[2 * (1 - H)];
            px[((H * NCG + cg) * 2 + 1) * 64 + lane] = part[2 * (1 - H) + 1];
        }
        stamp(2);  // (the partial sums above wait for the MFMAs)
        if constexpr (NOISE_AHEAD) {
            if (it == 0) {
                make_step_noise(step, it);
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) { nz0[e] = nzn0[e]; nz1[e] = nzn1[e]; }
            }
        } else if constexpr (!NOISE_FIRST && !PW) {
            make_step_noise(step, it);
        }
        stamp(3);
        float qx[4];
        if constexpr (KH == 2) {
            // the halves swap partial sums: this wave finishes rows 2 H, 2 H + 1 (its NE elements) and hands the other
            // two to its twin; sum = (half 0) + (half 1)
            __syncthreads();
            stamp(4);
            const float o0 = px[(((1 - H) * NCG + cg) * 2 + 0) * 64 + lane], o1 = px[(((1 - H) * NCG + cg) * 2 + 1) * 64 + lane];
            const float t0 = (H == 0) ? part[0] + o0 : o0 + part[2], t1 = (H == 0) ? part[1] + o1 : o1 + part[3];
            qx[0] = __builtin_fmaf(a.in_scale, t0, shift_j);
            qx[1] = __builtin_fmaf(a.in_scale, t1, shift_j);
            qx[2] = qx[3] = 0.0f;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                qx[r] = __builtin_fmaf(a.in_scale, (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]), shift_j);
        }

        AdamScalars ad;
        if constexpr (ADAM) {
            ad.beta1 = a.ad.beta1; ad.one_m_beta1 = a.ad.one_m_beta1; ad.inv_bc1 = trow[12];
            ad.beta2 = a.ad.beta2; ad.one_m_beta2 = a.ad.one_m_beta2; ad.inv_bc2 = trow[13];
            ad.alpha = a.ad.alpha; ad.eps = a.ad.eps; ad.use_v = a.ad.use_v; ad.add_assign = a.ad.add_assign;
        }
        auto adam = [&](float g, int e) {
            if constexpr (ADAM) {
                float m, v;
                const float out = adam_precondition(ad, g, am[e], av[e], m, v);
                am[e] = ok[e] ? m : am[e];
                av[e] = ok[e] ? v : av[e];
                return out;
            } else {
                return g;
            }
        };

        // ---- update ------------------------------------------------------------------------------
        // The elements of a lane are independent dependency chains and this wave is usually alone on
        // its SIMD (a dependent VALU instruction issues every ~8 cycles, an independent one every 4):
        // all updates together, branch-free, so the scheduler interleaves them.
        if constexpr (MODE == MODE_DL) {
            const DlScalars k = *reinterpret_cast<const DlScalars*>(trow);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                float cn, sn;
                if constexpr (CCVM_PERSIST_ABL & 4) {
                    cn = s0[e] + qx[2 * e] * nz0[e];
                    sn = s1[e] + qx[2 * e + 1] * nz1[e];
                } else {
                    dl_update(k, s0[e], s1[e], qx[2 * e], qx[2 * e + 1], vj, nz0[e], nz1[e], cn, sn);
                }
                s0[e] = ok[e] ? cn : s0[e];
                s1[e] = ok[e] ? sn : s1[e];
            }
        } else if constexpr (MODE == MODE_MF) {
            const MfScalars k = *reinterpret_cast<const MfScalars*>(trow);
            const float* wn = nz0;  // the next step's normals
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float bound = a.s_cols ? sat_j : k.S;
                const float fb = adam(__builtin_fmaf(k.f_q, qx[e], k.f_v * vj) * inv_sat_j, e);
                float mun, sgn;
                mf_update(k, s0[e], s1[e], fb, wc[e], mun, sgn);
                s0[e] = ok[e] ? mun : s0[e];
                s1[e] = ok[e] ? sgn : s1[e];
                // the last step's input is what mu_tilde_out returns: no new measurement after it
                const bool nxt = ok[e] && it + 1 < a.nsteps;
                mt[e] = nxt ? clampf(__builtin_fmaf(k.k_next, wn[e], s0[e]), -bound, bound) : mt[e];
                wc[e] = nxt ? wn[e] : wc[e];
            }
        } else {
            const LvScalars k = *reinterpret_cast<const LvScalars*>(trow);
            const float* n0 = nz0;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float g = adam(__builtin_fmaf(k.g_q, qx[e], k.g_v * vj) * inv_sat_j, e);
                const float x = lv_update(k, s0[e], g, n0[e], a.s_cols ? sat_j : k.S);
                s0[e] = ok[e] ? x : s0[e];
            }
        }
        if constexpr (!(CCVM_PERSIST_ABL & 8)) {
            cur ^= 1;
            stamp(5);
            publish(xs + cur * ROWS * LDX);
            if constexpr (NOISE_AHEAD) {
                if (it + 1 < a.nsteps) {
                    make_step_noise(step + 1, it + 1);
#pragma unroll
                    for (int e = 0; e < NE; ++e) { nzn0[e] = nz0[e]; nzn1[e] = nz1[e]; }
                }
            }
            // one-wave sets: a wave's LDS instructions execute in order, its reads see its own writes
            stamp(6);
            if constexpr (NCG > 1 || PW) __syncthreads();
            stamp(7);
        }
    }
    };  // run_steps
    if constexpr (CCVM_PERSIST_ABL & 16) st_last = persist_stamp();
    if constexpr (XS > 0) {
        if (role == 0) run_steps(std::integral_constant<int, 0>{});
        else if (role == 1) run_steps(std::integral_constant<int, 1>{});
        else if (role == 2) run_steps(std::integral_constant<int, 2>{});
        else run_steps(std::integral_constant<int, 3>{});
    } else if constexpr (KH == 2) {
        if (kh == 0) run_steps(std::integral_constant<int, 0>{});
        else run_steps(std::integral_constant<int, 1>{});
    } else {
        run_steps(std::integral_constant<int, 0>{});
    }

    stamps_out(0);
    // ---- write the state back -----------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (!ok[e]) continue;
        a.x0[gidx[e]] = s0[e];
        if constexpr (MODE != MODE_LANGEVIN) a.x1[gidx[e]] = s1[e];
        if constexpr (MODE == MODE_MF)
            if (a.xt) a.xt[gidx[e]] = mt[e];
        if constexpr (ADAM) {
            a.am[gidx[e]] = am[e];
            if (a.ad.use_v) a.av[gidx[e]] = av[e];
        }
    }
}

static_assert(sizeof(DlScalars) == 32 && sizeof(LvScalars) == 32 && sizeof(MfScalars) == 48,
              "schedule table rows: solver scalars in words 0..11");

}  // namespace ccvm
