// Persistent streamed-Q tile kernel: MODE_LANGEVIN instantiations (Langevin and pumped Langevin, with and without the
// Adam preconditioner; see ccvm_ptile.h).
#define CCVM_STEP_KERNEL_ONLY
#include "ccvm_ptile.h"

namespace ccvm {
void ptile_launch_lv(const PtileArgs& a, bool adam, hipStream_t st) {
    if (adam) hipLaunchKernelGGL((ptile_kernel<MODE_LANGEVIN, true>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, st, a);
    else hipLaunchKernelGGL((ptile_kernel<MODE_LANGEVIN, false>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, st, a);
}
}  // namespace ccvm
