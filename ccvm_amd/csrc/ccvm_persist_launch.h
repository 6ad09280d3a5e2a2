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
    int cw, ncg, nch, ru, grid;
};
inline PersistShape persist_shape(bool dl, int B, int N, int ru_override) {
    PersistShape s;
    s.nch = (N + 15) / 16 > 16 ? 16 : (N + 15) / 16;                 // K chunks of 16
    s.cw = s.nch == 1 ? 16 : s.nch == 2 ? 32 : 64;                   // columns a wave covers
    s.ncg = s.nch <= 4 ? 1 : s.nch <= 8 ? 2 : 4;                     // waves side by side
    const int br4 = (dl ? 2 : 4) * (64 / s.cw);                      // batch rows per row set at RU = 4
    s.ru = ((B + br4 - 1) / br4) * s.ncg >= 768 ? 4 : 2;
    if (ru_override == 2 || ru_override == 4) s.ru = ru_override;
    const int per = br4 * s.ru / 4 * (4 / s.ncg);                    // batch rows per workgroup
    s.grid = (B + per - 1) / per;
    return s;
}

// Shape by N (columns a wave covers x waves side by side) and rows in use per 4-row group: 4 when
// that still gives (nearly) every one of the 1024 SIMDs a wave, else 2 (shorter per-step chain per
// wave, twice the waves).  PersistArgs::ru_override (CCVM_AMD_PERSIST_RU=2|4, read by the ABI) forces one.
template <int MODE, bool ADAM, int CW, int NCG, int NCH>
void launch_persist_shape(const PersistArgs& a, hipStream_t st) {
    const PersistShape sh = persist_shape(MODE == MODE_DL, a.B, a.N, a.ru_override);  // sh.cw == CW etc. by construction
    const dim3 block(256);  // 4 / NCG row sets of NCG waves per workgroup (ccvm_persist.h)
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
