// Column-slab persistent kernel instantiations: MODE_LANGEVIN (see ccvm_slab.h).
#include "ccvm_slab.h"

namespace ccvm {
void slab_launch_lv(const SlabArgs& a, const SlabPlan& p, hipStream_t st) { launch_slab<MODE_LANGEVIN>(a, p, st); }
}  // namespace ccvm
