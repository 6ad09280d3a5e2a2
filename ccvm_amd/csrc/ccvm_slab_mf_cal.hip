// Column-slab persistent kernel instantiations: MODE_MF, launches that calibrate their fetch delay (see ccvm_slab.h).
#include "ccvm_slab.h"

namespace ccvm {
void slab_launch_mf_cal(const SlabArgs& a, const SlabPlan& p, hipStream_t st) { launch_slab_cal<MODE_MF, true>(a, p, st); }
}  // namespace ccvm
