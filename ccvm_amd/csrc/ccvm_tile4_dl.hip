// Per-step tile kernel, 32 x 32 split-K tiles (KS = 4): MODE_DL instantiations (see ccvm_kernels.h).
#define CCVM_STEP_KERNEL_ONLY
#include "ccvm_kernels.h"

namespace ccvm {
void tile4_launch_dl(const StepArgs& a, int grid, hipStream_t st) { launch_tile4<MODE_DL>(a, grid, false, false, st); }
}  // namespace ccvm
