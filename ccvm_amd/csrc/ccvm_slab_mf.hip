// Column-slab persistent kernel instantiations: MODE_MF (see ccvm_slab.h).
#include "ccvm_slab.h"

namespace ccvm {
void slab_launch_mf(const SlabArgs& a, const SlabPlan& p, hipStream_t st) { launch_slab<MODE_MF>(a, p, st); }
}  // namespace ccvm
