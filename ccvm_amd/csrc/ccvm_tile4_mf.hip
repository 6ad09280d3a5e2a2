// Per-step tile kernel, 32 x 32 split-K tiles (KS = 4): MODE_MF instantiations (see ccvm_kernels.h).
#define CCVM_STEP_KERNEL_ONLY
#include "ccvm_kernels.h"

namespace ccvm {
void tile4_launch_mf(const StepArgs& a, int grid, bool adam, bool per_variable_s, hipStream_t st) {
    launch_tile4<MODE_MF>(a, grid, adam, per_variable_s, st);
}
}  // namespace ccvm
