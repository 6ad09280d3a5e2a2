// Persistent streamed-Q tile kernel: MODE_DL instantiation (see ccvm_ptile.h).
#define CCVM_STEP_KERNEL_ONLY
#include "ccvm_ptile.h"

namespace ccvm {
void ptile_launch_dl(const PtileArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((ptile_kernel<MODE_DL>), dim3(a.nrb * a.ncb), dim3(WG_THREADS), 0, st, a);
}
}  // namespace ccvm
