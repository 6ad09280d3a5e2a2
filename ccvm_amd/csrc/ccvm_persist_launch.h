// Host-side launch of the persistent kernel: shape dispatch by N.  The kernel instantiations
// (8 + 8 shapes x 2 row fillings per solver variant) are spread over five translation units
// (ccvm_persist_*.hip) so that they compile in parallel; each defines one persist_launch_* entry.
#pragma once
#include <cstdlib>

#include "ccvm_persist.h"

namespace ccvm {

constexpr int PERSIST_MAX_N = 256;

void persist_launch_dl(const PersistArgs& a, hipStream_t st);
void persist_launch_mf(const PersistArgs& a, hipStream_t st);
void persist_launch_mf_adam(const PersistArgs& a, hipStream_t st);
void persist_launch_lv(const PersistArgs& a, hipStream_t st);
void persist_launch_lv_adam(const PersistArgs& a, hipStream_t st);

// The launch shape of the persistent kernel for (solver, B, N): ONE definition, used by the launcher below and
// by ccvm_describe_launch (so that the name a benchmark line reports is the instantiation that runs).
struct PersistShape {
    int cw, ncg, nch, ru, grid, kh, pw, threads;
};
constexpr int PERSIST_PW_MAX_NCG = 2;  // noise producer waves are instantiated for one and two waves side by side
// kh_override: 1 / 2 forces the K split off / on (where the shape has one), 0 = by batch size
// solver: 0 DL, 1 MF, 2 Langevin / pumped Langevin (the C ABI's numbering)
// pw_override: 1 / 2 forces the noise producer waves off / on (where the shape has them), 0 = by shape and batch size
inline PersistShape persist_shape(int solver, bool adam, int B, int N, int ru_override, int kh_override = 0, int simds = 1024,
                                  int pw_override = 0) {
    const bool dl = solver == 0;
    if (simds <= 0) simds = 1024;
    PersistShape s;
    s.nch = (N + 15) / 16 > 16 ? 16 : (N + 15) / 16;                 // K chunks of 16
    s.cw = s.nch == 1 ? 16 : s.nch == 2 ? 32 : 64;                   // columns a wave covers
    s.ncg = s.nch <= 4 ? 1 : s.nch <= 8 ? 2 : 4;                     // waves side by side
    const int br4 = (dl ? 2 : 4) * (64 / s.cw);                      // batch rows per row set at RU = 4
    s.ru = ((B + br4 - 1) / br4) * s.ncg >= 768 ? 4 : 2;
    if (ru_override == 2 || ru_override == 4) s.ru = ru_override;
    // Two waves side by side (64 < N <= 128): K can be split over two waves instead of idling two MFMA rows -- twice the
    // waves at half the chain each.  A step costs what the fullest SIMD issues: with w = waves per SIMD at four rows in
    // use, ceil(w) whole chains against ceil(2 w) half chains (DL N = 100, us per step, whole / split: B = 1500, w = 1.46:
    // 1.56 / 1.24; B = 2000, w = 1.95: 1.56 / 1.60; B = 4000: 2.84 / 2.93; Langevin B = 3000, w = 1.46: 1.61 / 1.32) --
    // and a wave alone on its SIMD stalls a third of its step (LDS round trip, barrier, dependent issue), which a
    // second, independent half-chain wave fills: w <= 1 takes the split too (Langevin B = 1000 0.82 -> 0.68, MF 0.94 ->
    // 0.84, DL B <= 512 0.79 -> 0.66, DL B = 1000 0.95 -> 0.94).
    s.kh = 1;
    if (s.ncg >= 2) {
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
        const bool lone = s.ncg == 4 && s.nch >= lone_from;
        if ((halves < 2 * whole || waves4 <= simds || lone) && !(ru_override == 2 || ru_override == 4)) s.kh = 2;
        if (kh_override == 1) s.kh = 1;
        if (kh_override == 2) s.kh = 2;
        if (s.kh == 2) s.ru = 4;
    }
    // Noise producer waves (ccvm_persist.h, PW): one-wave row sets whose consumers AND producers all find a SIMD of
    // their own -- the chain of a step loses the generator.  (policy: tools/time_small.py A/B, round 6)
    s.pw = 0;
    if (s.ncg == 1) {
        const int consumers = (B + br4 * s.ru / 4 - 1) / (br4 * s.ru / 4);
        if (consumers <= simds) s.pw = 1;  // (up to one consumer and one producer per SIMD)
    } else if (s.ncg == 2 && s.kh == 2) {
        const int consumers = ((B + br4 - 1) / br4) * s.ncg * 2;  // half-chain waves
        if (consumers <= 2 * simds) s.pw = 1;
    }
    if (pw_override == 1) s.pw = 0;
    // (two waves side by side: producers only next to the K split -- with whole chains the consumers and producers of a
    // four-wave workgroup sit on different SIMDs and two such workgroups per CU cost more than they save: DL N = 100,
    // B = 1000 1.06 us per step against 0.93 without, role swap or not)
    if (pw_override == 2 && (s.ncg == 1 || (s.ncg == 2 && s.kh == 2))) s.pw = 1;
    const int wps = s.ncg * s.kh * (1 + s.pw);                       // waves per row set
    const int sets = wps > 4 ? 1 : 4 / wps;                          // row sets per workgroup
    const int per = br4 * s.ru / 4 * sets;                           // batch rows per workgroup
    s.grid = (B + per - 1) / per;
    s.threads = wps > 4 ? 64 * wps : 256;
    return s;
}

// Shape by N (columns a wave covers x waves side by side) and rows in use per 4-row group: 4 when
// that still gives (nearly) every one of the 1024 SIMDs a wave, else 2 (shorter per-step chain per
// wave, twice the waves).  PersistArgs::ru_override (CCVM_AMD_PERSIST_RU=2|4, read by the ABI) forces one.
template <int MODE, bool ADAM, int CW, int NCG, int NCH>
void launch_persist_shape(const PersistArgs& a, hipStream_t st) {
    const PersistShape sh = persist_shape(MODE == MODE_DL ? 0 : MODE == MODE_MF ? 1 : 2, ADAM, a.B, a.N, a.ru_override,
                                          a.kh_override, a.simds, a.pw_override);  // sh.cw == CW etc. by construction
    const dim3 block(sh.threads);  // 4 / NCG row sets of NCG waves per workgroup (ccvm_persist.h)
    if constexpr (NCG <= PERSIST_PW_MAX_NCG) {
        if (sh.pw) {  // as many producer waves as consumer waves
            if constexpr (NCG >= 2) {  // (next to the K split only: persist_shape)
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
    if constexpr (NCG >= 2) {
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
        case 5: launch_persist_shape<MODE, ADAM, 64, 2, 5>(a, st); break;
        case 6: launch_persist_shape<MODE, ADAM, 64, 2, 6>(a, st); break;
        case 7: launch_persist_shape<MODE, ADAM, 64, 2, 7>(a, st); break;
        case 8: launch_persist_shape<MODE, ADAM, 64, 2, 8>(a, st); break;
        case 9: launch_persist_shape<MODE, ADAM, 64, 4, 9>(a, st); break;
        case 10: launch_persist_shape<MODE, ADAM, 64, 4, 10>(a, st); break;
        case 11: launch_persist_shape<MODE, ADAM, 64, 4, 11>(a, st); break;
        case 12: launch_persist_shape<MODE, ADAM, 64, 4, 12>(a, st); break;
        case 13: launch_persist_shape<MODE, ADAM, 64, 4, 13>(a, st); break;
        case 14: launch_persist_shape<MODE, ADAM, 64, 4, 14>(a, st); break;
        case 15: launch_persist_shape<MODE, ADAM, 64, 4, 15>(a, st); break;
        default: launch_persist_shape<MODE, ADAM, 64, 4, 16>(a, st); break;
    }
}

}  // namespace ccvm
