// Persistent streamed-Q tile kernel: MODE_LANGEVIN instantiations (Langevin and pumped Langevin, with and without the
// Adam preconditioner; see ccvm_ptile.h).
#define CCVM_STEP_KERNEL_ONLY
#include "ccvm_ptile.h"

namespace ccvm {
void ptile_launch_lv(const PtileArgs& a, bool adam, hipStream_t st) {
    const dim3 grid(a.nrb * a.ncb), block(WG_THREADS);
    if (a.s_cols) {  // per-variable saturation
        if (adam) hipLaunchKernelGGL((ptile_kernel<MODE_LANGEVIN, true, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((ptile_kernel<MODE_LANGEVIN, false, false, true>), grid, block, 0, st, a);
    } else {
        if (adam) hipLaunchKernelGGL((ptile_kernel<MODE_LANGEVIN, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((ptile_kernel<MODE_LANGEVIN, false>), grid, block, 0, st, a);
    }
}
}  // namespace ccvm
