// Persistent streamed-Q tile kernel: MODE_MF instantiations (with and without the Adam preconditioner; see ccvm_ptile.h).
#define CCVM_STEP_KERNEL_ONLY
#include "ccvm_ptile.h"

namespace ccvm {
void ptile_launch_mf(const PtileArgs& a, bool adam, hipStream_t st) {
    const dim3 grid(a.nrb * a.ncb), block(WG_THREADS);
    if (a.s_cols) {  // per-variable saturation
        if (adam) hipLaunchKernelGGL((ptile_kernel<MODE_MF, true, false, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((ptile_kernel<MODE_MF, false, false, true>), grid, block, 0, st, a);
    } else {
        if (adam) hipLaunchKernelGGL((ptile_kernel<MODE_MF, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((ptile_kernel<MODE_MF, false>), grid, block, 0, st, a);
    }
}
}  // namespace ccvm
