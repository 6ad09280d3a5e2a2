// Column-slab persistent kernel instantiations: MODE_DL (see ccvm_slab.h).
#include "ccvm_slab.h"

namespace ccvm {
void slab_launch_dl(const SlabArgs& a, const SlabPlan& p, hipStream_t st) { launch_slab<MODE_DL>(a, p, st); }
}  // namespace ccvm
