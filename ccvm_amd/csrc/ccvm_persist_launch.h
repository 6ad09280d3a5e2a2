// Host-side launch of the persistent kernel: shape dispatch by N.  The kernel instantiations
// (8 + 8 shapes x 2 row fillings per solver variant) are spread over five translation units
// (ccvm_persist_*.hip) so that they compile in parallel; each defines one persist_launch_* entry.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "ccvm_persist.h"
#include "ccvm_persist_model.h"
#include "ccvm_plan_model.h"

namespace ccvm {

constexpr int PERSIST_MAX_N = 256;
// FIVE waves side by side x two K halves (round 6): 256 < N <= 320, part of every wave's fragments in LDS (ccvm_persist.h: QL).
// Ten waves are three on some SIMD -- 168 registers -- and what the working set leaves of them holds a wave's first KR
// fragments, the other 8 NCH - KR live in LDS, 160 KB for all ten waves + the state: DL and Langevin / pumped Langevin
// without Adam keep 104 in registers and reach N = 320 (56 in LDS); Langevin + Adam keeps 92 and reaches N = 288; MF 84 and
// N = 272; MF + Adam (72) nothing.  Beyond that the column-cluster kernel stays (code objects' counts: tests/test_launch_policy.py).
constexpr int PERSIST_WIDE_MAX_N = 320;
constexpr int persist_wide_kr(int solver, bool adam) { return solver == 1 ? (adam ? 0 : 84) : (solver == 2 && adam) ? 92 : 104; }
constexpr int persist_wide_max_nch(int solver, bool adam) { return solver == 1 ? (adam ? 0 : 17) : (solver == 2 && adam) ? 18 : 20; }
// the unequal K split's short part (ccvm_persist.h: XS): (12 K - V) / 32 k-steps balance the SIMDs' issue (V: a wave's update,
// ~350 cycles), a multiple of four, and all of them in registers (<= 104) -- then swept +-4 / 8 around that on the chip
// (profiles/r06_ab_persist_xs_delta.txt): 96 at 17 K chunks, the cap from 18 on
// (MF and Langevin + Adam: the unequal form keeps four fragments fewer in registers -- 80 / 88 -- than their equal halves, and
// their short parts are all of those)
#ifndef CCVM_PERSIST_XS_DELTA
#define CCVM_PERSIST_XS_DELTA 0   // (tuning builds: the short part longer / shorter by this many k-steps)
#endif
constexpr int persist_wide_kr_xs(int solver, bool adam) { return solver == 1 ? (adam ? 0 : 80) : (solver == 2 && adam) ? 88 : 104; }
constexpr int persist_wide_xs(int solver, bool adam, int nch) {
    const int x = (nch == 17 ? 96 : 104) + CCVM_PERSIST_XS_DELTA, cap = persist_wide_kr_xs(solver, adam);
    return x > cap ? cap : x;
}
// ... and of three waves side by side x two K halves in ONE six-wave workgroup per CU -- {0, 4} {1, 5} {2} {3} on the SIMDs: the
// waves alone on theirs take the long parts, all fragments in registers (K / 4 - V / 16, a multiple of four; swept on the
// chip, profiles/r06_ab_persist_xs3_delta.txt: the table for DL and MF, eight more for Langevin, whose update is the shortest)
#ifndef CCVM_PERSIST_XS3_DELTA
#define CCVM_PERSIST_XS3_DELTA 0   // (tuning builds)
#endif
constexpr int persist_xs3(int solver, int nch) {
    return (nch == 9 ? 16 : nch == 10 ? 20 : nch == 11 ? 24 : 28) + (solver == 2 ? 8 : 0) + CCVM_PERSIST_XS3_DELTA;
}
inline bool persist_wide_ok(int solver, bool adam, int N) { return N > PERSIST_MAX_N && (N + 15) / 16 <= persist_wide_max_nch(solver, adam); }
// its estimate: rounds of one row set (two DL rows, four of a one-stream solver) per CU x the measured round
inline double persist_wide_us(int solver, bool adam, int B, int N, int cus) {
    const int nch = (N + 15) / 16, rows = solver == 0 ? 2 : 4, sets = (B + rows - 1) / rows;
    const int k = nch < 17 ? 0 : nch > 20 ? 3 : nch - 17;
    const double round = solver == 1 ? PERSIST_WIDE_ROUND_MF_US : PERSIST_WIDE_ROUND_US[solver == 0 ? 0 : 1][k] * (adam ? PERSIST_WIDE_ADAM : 1.0);
    return (double)((sets + cus - 1) / cus) * round;
}

void persist_launch_dl(const PersistArgs& a, hipStream_t st);
void persist_launch_mf(const PersistArgs& a, hipStream_t st);
void persist_launch_mf_adam(const PersistArgs& a, hipStream_t st);
void persist_launch_lv(const PersistArgs& a, hipStream_t st);
void persist_launch_lv_adam(const PersistArgs& a, hipStream_t st);

// The launch shape of the persistent kernel for (solver, B, N): ONE definition, used by the launcher below and
// by ccvm_describe_launch (so that the name a benchmark line reports is the instantiation that runs).
struct PersistShape {
    int cw, ncg, nch, ru, grid, kh, pw, threads;
    int rsw;        // 2: two six-wave row sets per workgroup (three side by side x two K halves), else 0
    int xs;         // > 0: the unequal K split's short part (three side by side, one six-wave workgroup per CU: persist_xs3)
    double est_us;  // the variant model's estimate (one wave per row set, N <= 64: ccvm_persist_model.h); five side by side: rounds x
                    // the measured round (ccvm_plan_model.h); else 0
};
constexpr int PERSIST_PW_MAX_NCG = 2;  // noise producer waves are instantiated for one and two waves side by side
// kh_override: 1 / 2 forces the K split off / on (where the shape has one), 0 = by batch size
// solver: 0 DL, 1 MF, 2 Langevin / pumped Langevin (the C ABI's numbering)
// pw_override: 1 / 2 forces the noise producer waves off / on (where the shape has them), 0 = by shape and batch size
// rsw_override: 1 / 2 forces the row sets per workgroup of six-wave row sets, 0 = by batch size
// cw_override: 32 / 64 forces the wave shape at 64 < N <= 96 (three 32-column waves of eight rows / two 64-column waves of four)
inline PersistShape persist_shape(int solver, bool adam, int B, int N, int ru_override, int kh_override = 0, int simds = 1024,
                                  int pw_override = 0, int rsw_override = 0, int cw_override = 0) {
    const bool dl = solver == 0;
    if (simds <= 0) simds = 1024;
    PersistShape s;
    s.nch = (N + 15) / 16 > 20 ? 20 : (N + 15) / 16;                 // K chunks of 16
    s.cw = s.nch == 1 ? 16 : s.nch == 2 ? 32 : 64;                   // columns a wave covers
    s.ncg = s.nch <= 4 ? 1 : s.nch <= 8 ? 2 : (CCVM_PERSIST_NCG3 && s.nch <= 12) ? 3 : s.nch <= 16 ? 4 : 5;  // waves side by side
    // 64 < N <= 96 (round 6): THREE 32-column waves side by side, two row groups (eight MFMA rows) each -- 96 columns instead
    // of 128 for the same rows, three waves where two pairs stood; whole chains, producers next to them.  Bit-identical to
    // the 64-column waves' whole chains.  Same-box sweep (profiles/r06_ab_persist_cw32.txt, tools/ab_persist_cw32.sh): mostly a
    // wash -- every wave's chain is as long as before -- but two kinds of cells gain: (1) a batch that needs TWO eight-wave
    // workgroups of the wide shape on a CU and ONE six-wave workgroup of this one (DL N = 70, B = 1000: 0.765 -> 0.72 us per
    // step; Langevin B = 1500 / 2000: 0.78 -> 0.70, + Adam 1.07 -> 0.96; MF: +-1 %, + Adam worse: left alone); (2) the Adam
    // variants while every two-row set has a CU (Langevin + Adam N = 70, B <= 1000: 0.82 -> 0.69; MF + Adam 0.94 -> 0.85).
    // Elsewhere the wide shape stays.
    bool narrow = false;
    int narrow_ru = 0;  // the rows in use and producers that come with the rule (0: as forced / by the generic rules)
    if (s.nch == 5 || s.nch == 6) {
        if (cw_override == 32) {
            narrow = true;
        } else if (cw_override == 0 && !ru_override && !kh_override && !pw_override) {
            const int rows64 = dl ? 2 : 4, cus = simds / 4;
            const int sets64 = (B + rows64 - 1) / rows64, sets32 = (B + 2 * rows64 - 1) / (2 * rows64);
            if (solver != 1 && sets64 > cus && sets32 <= cus) { narrow = true; narrow_ru = 4; }
            else if (adam && sets64 <= cus) { narrow = true; narrow_ru = 2; }
        }
    }
    if (narrow) { s.cw = 32; s.ncg = 3; }
    const int br4 = (dl ? 2 : 4) * (64 / s.cw);                      // batch rows per row set at RU = 4
    s.ru = ((B + br4 - 1) / br4) * s.ncg >= 768 ? 4 : 2;
    if (ru_override == 2 || ru_override == 4) s.ru = ru_override;
    if (narrow_ru) s.ru = narrow_ru;
    // Two waves side by side (64 < N <= 128): K can be split over two waves instead of idling two MFMA rows -- twice the
    // waves at half the chain each.  A step costs what the fullest SIMD issues: with w = waves per SIMD at four rows in
    // use, ceil(w) whole chains against ceil(2 w) half chains (DL N = 100, us per step, whole / split: B = 1500, w = 1.46:
    // 1.56 / 1.24; B = 2000, w = 1.95: 1.56 / 1.60; B = 4000: 2.84 / 2.93; Langevin B = 3000, w = 1.46: 1.61 / 1.32) --
    // and a wave alone on its SIMD stalls a third of its step (LDS round trip, barrier, dependent issue), which a
    // second, independent half-chain wave fills: w <= 1 takes the split too (Langevin B = 1000 0.82 -> 0.68, MF 0.94 ->
    // 0.84, DL B <= 512 0.79 -> 0.66, DL B = 1000 0.95 -> 0.94).
    s.kh = 1;
    if (s.ncg >= 2 && !narrow) {
        const int waves4 = ((B + br4 - 1) / br4) * s.ncg;
        const int whole = (waves4 + simds - 1) / simds, halves = (2 * waves4 + simds - 1) / simds;
        // Four waves side by side (128 < N <= 256): from this many K chunks on the unsplit kernel needs more than 256
        // VGPRs (its Q fragments alone are 16 NCH), i.e. a SIMD holds ONE of its waves; the split kernel (half the
        // fragments) fits two, and wins at every batch (N = 256: DL 3.54 -> 2.88 us per step at B = 1000, 14.2 -> 11.5
        // at B = 4000; Langevin 1.89 -> 1.48, MF 2.30 -> 1.69; N = 240 DL -12 %, N = 224 MF -18 %).  Below that the
        // unsplit kernel holds two waves itself and the rule above decides (N = 208 Langevin, B = 2000: 2.30 unsplit
        // vs 2.61 split).  Register counts from the code objects of this build (hipcc 7.2): DL 258 at NCH = 13, MF 272
        // at 11, MF + Adam 270 at 10, Langevin 272 at 14, Langevin + Adam 266 at 12.
        const int lone_from = dl ? 13 : solver == 1 ? (adam ? 10 : 11) : (adam ? 12 : 14);
        const bool lone = s.ncg >= 3 && s.nch >= lone_from;
        // (three waves side by side: the "three halves < two wholes" case does not pay -- DL N = 176 / 192, B = 1000: 2.26 / 2.37
        // us per step split, 1.96 / 2.08 whole; Langevin N = 176, B = 2000: 2.39 / 2.02 -- profiles/r06_ab_persist_ncg3.txt)
        const bool fewer_rounds = s.ncg != 3 && halves < 2 * whole;
        if ((fewer_rounds || waves4 <= simds || lone) && !(ru_override == 2 || ru_override == 4)) s.kh = 2;
        // Small batches, three / four waves side by side (round 6): while every row set at TWO rows in use still has a CU of
        // its own, whole chains over two rows beat both the K split and four rows -- one wave per SIMD either way, and the
        // wave's VALU part (generator, update, publish) is half as long (us per step, split / two rows: DL N = 144 1.05 /
        // 0.88 up to B = 256, N = 224 1.32 / 1.20; Langevin N = 144 1.07 / 0.92 up to B = 512, N = 200 1.30 / 1.17; MF N = 144 1.25 /
        // 1.04, + Adam 1.54 / 1.25; one row set more than CUs: 1.49 -- and from N = 241 the unsplit kernel's registers
        // turn it around, N = 256: 1.42 / 1.59: profiles/r06_ab_persist_kh_small.txt).
        const int sets2 = (B + br4 / 2 - 1) / (br4 / 2);
        // (three side by side, DL and Langevin: the six-wave workgroup's unequal K split below is 3 % faster still -- DL N = 144
        // 0.88 -> 0.86, Langevin 0.92 -> 0.89 -- and MF 2 % slower: 1.04 / 1.06; profiles/r06_ab_persist_xs3.txt)
        const bool two_rows_alone = s.ncg >= 3 && s.nch <= 14 && sets2 <= simds / 4 && ru_override != 4 &&
                                    !(s.ncg == 3 && solver != 1 && !ru_override && kh_override != 1);
        if (two_rows_alone) { s.kh = 1; s.ru = 2; }
        if (kh_override == 1) s.kh = 1;
        if (kh_override == 2) s.kh = 2;
        if (s.kh == 2) s.ru = 4;
    }
    // Three waves side by side (128 < N <= 192), batches of more row sets than CUs (round 6): six whole-chain waves of two
    // workgroups cannot put fewer than two on some SIMD of their CU, and two six-wave K-split workgroups land four / four /
    // two / two (the second starts on the SIMDs the first doubled up on: HW_ID of every wave, tools/simd_probe.hip,
    // profiles/r06_simd_probe.txt) -- and a step costs what the fullest SIMD issues.  TWO six-wave row sets in ONE workgroup
    // are twelve waves, three half chains on every SIMD: a round of such workgroups takes 1.6 x a round of whole chains
    // and holds twice the rows (us per step, before / after: DL N = 144, B = 1000 1.72 -> 1.39, B = 2000
    // 3.36 -> 2.77, B = 4000 6.57 -> 5.55; Langevin N = 160, B = 2000 1.90 -> 1.54; MF N = 176, B = 2000 2.64 -> 1.83; MF + Adam
    // N = 160, B = 2000 3.19 -> 2.07 -- but B = 1500 DL, three rounds of whole chains against two of these: 2.58 / 2.82:
    // profiles/r06_ab_persist_rsw.txt).  Twelve waves need <= 168 VGPRs: not MF + Adam from 11 K chunks (tests/test_launch_policy.py).
    s.rsw = 0;
    const bool rsw_fits = !(solver == 1 && adam && s.nch >= 11);
    if (s.ncg == 3 && !narrow && rsw_fits) {
        const int cus = simds / 4, sets4 = (B + br4 - 1) / br4;
        const int r1 = (sets4 + cus - 1) / cus, r2 = (sets4 + 2 * cus - 1) / (2 * cus);  // rounds of one / two row sets per CU
        const bool pays = 8 * r2 < 5 * r1 && kh_override != 1 && !(ru_override == 2 || ru_override == 4);
        if (rsw_override == 2 ? (s.kh == 2 || pays) : (rsw_override == 0 && pays)) { s.kh = 2; s.ru = 4; s.rsw = 2; }
    }
    // (one six-wave workgroup per CU -- the K split, every row set with a CU of its own: the unequal split; with more
    // workgroups on a CU its 200+ registers would keep the second one out)
    s.xs = 0;
    if (s.ncg == 3 && !narrow && s.kh == 2 && s.rsw == 0 && (B + br4 - 1) / br4 <= simds / 4) s.xs = persist_xs3(solver, s.nch);
    if (s.ncg == 5) { s.kh = 2; s.ru = 4; }  // five side by side: the K split only (ten waves, fragments partly in LDS)
    const double wide_est = s.ncg == 5 ? persist_wide_us(solver, adam, B, N, simds / 4) : 0.0;
    // Noise producer waves (ccvm_persist.h, PW).
    // One wave per row set (N <= 64): four variants -- two or four rows in use, with or without producers -- and which
    // one is fastest depends on how many ROUNDS of waves the fullest SIMD holds (a consumer next to its producer costs
    // little more than a consumer; a second consumer costs nearly a whole chain).  The estimate of each variant is the
    // model fitted to the sweep of round 6 (ccvm_persist_model.h, generated by tools/fit_persist_model.py: within 3 %
    // of the best variant in all 144 cells, at most 2.7 % behind: profiles/r06_persist_policy.md); the overrides pin their dimension.
    s.pw = 0;
    s.est_us = wide_est;
    if (s.ncg == 1) {
        const int k4 = (N + 3) / 4;
        double best = 1e30;
        for (int ru = 2; ru <= 4; ru += 2) {
            if ((ru_override == 2 || ru_override == 4) && ru != ru_override) continue;
            for (int pw = 0; pw <= 1; ++pw) {
                if (pw_override && pw != pw_override - 1) continue;
                const int per = br4 * ru / 4;
                const long waves = (long)((B + per - 1) / per) * (1 + pw);
                const int r = (int)((waves + simds - 1) / simds);
                const double* c = PERSIST_MODEL[solver < 0 ? 0 : solver > 2 ? 2 : solver][ru / 2 - 1][pw];
                const double est = c[0] + c[1] * k4 + (r - 1) * (c[2] + c[3] * k4) + (r > 1 ? c[4] : 0.0);
                if (est < best) { best = est; s.ru = ru; s.pw = pw; }
            }
        }
        s.est_us = best < 1e30 ? best : 0.0;
    } else if (narrow) {
        if (pw_override == 2 || narrow_ru) s.pw = 1;
    } else if (s.ncg == 2 && s.kh == 2) {
        // Two waves side by side: producers next to the K split while a SIMD holds at most two half-chain consumers (DL
        // N = 100: B = 1000 0.95 -> 0.87 us per step, B = 1500 1.26 -> 1.35; Langevin B = 2000 0.98 -> 0.92, B = 3000 1.30 ->
        // 1.42) -- and while two of the eight-wave workgroups still fit a CU where the batch needs them to: above 128
        // VGPRs a SIMD holds three waves, not four (MF at NCH = 8: 132, MF + Adam from NCH = 7, Langevin + Adam at NCH = 8;
        // MF, B = 1500 at N = 128: 1.20 -> 1.70).
        const int sets = (B + br4 - 1) / br4;
        // largest NCH whose K-split + producers kernel needs <= 128 VGPRs (the code objects' counts: tests/test_launch_policy.py)
        const int two_fit_to = solver == 1 ? (adam ? 6 : 7) : (solver == 2 && adam) ? 7 : 8;
        if (4 * sets <= 2 * simds && (4 * sets <= simds || s.nch <= two_fit_to)) s.pw = 1;
    }
    if (pw_override == 1) s.pw = 0;
    // (two waves side by side: producers only next to the K split -- with whole chains the consumers and producers of a
    // four-wave workgroup sit on different SIMDs and two such workgroups per CU cost more than they save: DL N = 100,
    // B = 1000 1.06 us per step against 0.93 without, role swap or not)
    if (pw_override == 2 && (s.ncg == 1 || narrow || (s.ncg == 2 && s.kh == 2))) s.pw = 1;
    const int wps = s.ncg * s.kh * (1 + s.pw);                       // waves per row set
    const int sets = s.rsw ? s.rsw : wps > 4 ? 1 : 4 / wps;          // row sets per workgroup
    const int per = br4 * s.ru / 4 * sets;                           // batch rows per workgroup
    s.grid = (B + per - 1) / per;
    s.threads = 64 * wps * sets;                                     // (256 up to four waves per row set, but 192 for three)
    return s;
}

// Shape by N (columns a wave covers x waves side by side) and rows in use per 4-row group: 4 when
// that still gives (nearly) every one of the 1024 SIMDs a wave, else 2 (shorter per-step chain per
// wave, twice the waves).  PersistArgs::ru_override (CCVM_AMD_PERSIST_RU=2|4, read by the ABI) forces one.
template <int MODE, bool ADAM, int CW, int NCG, int NCH>
void launch_persist_shape(const PersistArgs& a, hipStream_t st) {
    const PersistShape sh = persist_shape(MODE == MODE_DL ? 0 : MODE == MODE_MF ? 1 : 2, ADAM, a.B, a.N, a.ru_override,
                                          a.kh_override, a.simds, a.pw_override, a.rsw_override, a.cw_override);  // sh.cw == CW etc. by construction
    constexpr bool NARROW = CW == 32 && NCG == 3;  // (whole chains only)
    const dim3 block(sh.threads);  // row sets of NCG x KH (x 2 with producers) waves per workgroup (ccvm_persist.h: RSW)
    if constexpr (NCG <= PERSIST_PW_MAX_NCG || NARROW) {
        if (sh.pw) {  // as many producer waves as consumer waves
            if constexpr (NCG >= 2 && !NARROW) {  // (next to the K split only: persist_shape)
                hipLaunchKernelGGL((persist_kernel<MODE, ADAM, CW, NCG, NCH, 4, 2, 1>), dim3(sh.grid), block, 0, st, a);
            } else {
                if (sh.ru == 4)
                    hipLaunchKernelGGL((persist_kernel<MODE, ADAM, CW, NCG, NCH, 4, 1, 1>), dim3(sh.grid), block, 0, st, a);
                else
                    hipLaunchKernelGGL((persist_kernel<MODE, ADAM, CW, NCG, NCH, 2, 1, 1>), dim3(sh.grid), block, 0, st, a);
            }
            return;
        }
    }
    if constexpr (NCG == 3 && !NARROW && !(MODE == MODE_MF && ADAM && NCH >= 11)) {  // (MF + Adam from 11 K chunks: > 168 VGPRs, persist_shape)
        if (sh.kh == 2 && sh.rsw == 2) {  // two row sets of six waves
            hipLaunchKernelGGL((persist_kernel<MODE, ADAM, CW, NCG, NCH, 4, 2, 0, 2>), dim3(sh.grid), block, 0, st, a);
            return;
        }
    }
    if constexpr (NCG == 3 && !NARROW) {
        if (sh.kh == 2 && sh.rsw == 0 && sh.xs > 0 && a.xs_override != 1) {  // one six-wave workgroup per CU: the unequal K split
            hipLaunchKernelGGL((persist_kernel<MODE, ADAM, CW, NCG, NCH, 4, 2, 0, 0, 0, persist_xs3(MODE == MODE_DL ? 0 : MODE == MODE_MF ? 1 : 2, NCH)>),
                               dim3(sh.grid), block, 0, st, a);
            return;
        }
    }
    if constexpr (NCG >= 2 && !NARROW) {
        if (sh.kh == 2) {  // one row set of NCG x 2 waves
            hipLaunchKernelGGL((persist_kernel<MODE, ADAM, CW, NCG, NCH, 4, 2>), dim3(sh.grid), block, 0, st, a);
            return;
        }
    }
    if (sh.ru == 4)
        hipLaunchKernelGGL((persist_kernel<MODE, ADAM, CW, NCG, NCH, 4>), dim3(sh.grid), block, 0, st, a);
    else
        hipLaunchKernelGGL((persist_kernel<MODE, ADAM, CW, NCG, NCH, 2>), dim3(sh.grid), block, 0, st, a);
}

template <int MODE, bool ADAM>
void launch_persist(const PersistArgs& a, hipStream_t st) {
    switch ((a.N + 15) / 16) {  // K chunks of 16
        case 1: launch_persist_shape<MODE, ADAM, 16, 1, 1>(a, st); break;
        case 2: launch_persist_shape<MODE, ADAM, 32, 1, 2>(a, st); break;
        case 3: launch_persist_shape<MODE, ADAM, 64, 1, 3>(a, st); break;
        case 4: launch_persist_shape<MODE, ADAM, 64, 1, 4>(a, st); break;
        case 5:
        case 6: {
            const bool narrow = persist_shape(MODE == MODE_DL ? 0 : MODE == MODE_MF ? 1 : 2, ADAM, a.B, a.N, a.ru_override, a.kh_override,
                                              a.simds, a.pw_override, a.rsw_override, a.cw_override).cw == 32;
            if ((a.N + 15) / 16 == 5) {
                if (narrow) launch_persist_shape<MODE, ADAM, 32, 3, 5>(a, st);
                else launch_persist_shape<MODE, ADAM, 64, 2, 5>(a, st);
            } else {
                if (narrow) launch_persist_shape<MODE, ADAM, 32, 3, 6>(a, st);
                else launch_persist_shape<MODE, ADAM, 64, 2, 6>(a, st);
            }
            break;
        }
        case 7: launch_persist_shape<MODE, ADAM, 64, 2, 7>(a, st); break;
        case 8: launch_persist_shape<MODE, ADAM, 64, 2, 8>(a, st); break;
        case 9: launch_persist_shape<MODE, ADAM, 64, CCVM_PERSIST_NCG3 ? 3 : 4, 9>(a, st); break;
        case 10: launch_persist_shape<MODE, ADAM, 64, CCVM_PERSIST_NCG3 ? 3 : 4, 10>(a, st); break;
        case 11: launch_persist_shape<MODE, ADAM, 64, CCVM_PERSIST_NCG3 ? 3 : 4, 11>(a, st); break;
        case 12: launch_persist_shape<MODE, ADAM, 64, CCVM_PERSIST_NCG3 ? 3 : 4, 12>(a, st); break;
        case 13: launch_persist_shape<MODE, ADAM, 64, 4, 13>(a, st); break;
        case 14: launch_persist_shape<MODE, ADAM, 64, 4, 14>(a, st); break;
        case 15: launch_persist_shape<MODE, ADAM, 64, 4, 15>(a, st); break;
        case 16: launch_persist_shape<MODE, ADAM, 64, 4, 16>(a, st); break;
        default:
            {  // five waves side by side (persist_wide_ok: the planner asks only for what is instantiated here)
                constexpr int SOLVER = MODE == MODE_DL ? 0 : MODE == MODE_MF ? 1 : 2;
                constexpr int KR = persist_wide_kr(SOLVER, ADAM), MAXCH = persist_wide_max_nch(SOLVER, ADAM);
                const PersistShape sh = persist_shape(SOLVER, ADAM, a.B, a.N, 0, 0, a.simds);
                const dim3 grid(sh.grid), block(sh.threads);
                const int nch = (a.N + 15) / 16;
                if (nch > MAXCH) {  // (want_persist never asks: a launch that would silently do nothing must not pass)
                    std::fprintf(stderr, "ccvm: no row-owner kernel for N = %d of this solver variant\n", a.N);
                    std::abort();
                }
                const bool uneq = a.xs_override != 1;  // (1: equal halves, tuning)
#define CCVM_WIDE_CASE(NCHV, COND)                                                                                                        \
    if constexpr (MAXCH >= NCHV) {                                                                                                        \
        if (COND) {                                                                                                                       \
            bool launched = false;                                                                                                        \
            if constexpr (persist_wide_xs(SOLVER, ADAM, NCHV) > 0) {                                                                      \
                if (uneq) {                                                                                                               \
                    hipLaunchKernelGGL((persist_kernel<MODE, ADAM, 64, 5, NCHV, 4, 2, 0, 0, 8 * NCHV - persist_wide_kr_xs(SOLVER, ADAM),          \
                                                       persist_wide_xs(SOLVER, ADAM, NCHV)>),                                              \
                                       grid, block, 0, st, a);                                                                            \
                    launched = true;                                                                                                      \
                }                                                                                                                         \
            }                                                                                                                             \
            if (!launched)                                                                                                                \
                hipLaunchKernelGGL((persist_kernel<MODE, ADAM, 64, 5, NCHV, 4, 2, 0, 0, 8 * NCHV - KR>), grid, block, 0, st, a);           \
        }                                                                                                                                 \
    }
                CCVM_WIDE_CASE(17, nch == 17)
                CCVM_WIDE_CASE(18, nch == 18)
                CCVM_WIDE_CASE(19, nch == 19)
                CCVM_WIDE_CASE(20, nch >= 20)
#undef CCVM_WIDE_CASE
            }
            break;
    }
}

}  // namespace ccvm
