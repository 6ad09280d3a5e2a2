// Column-slab persistent kernel instantiations: MODE_LANGEVIN (see ccvm_slab.h).
#include "ccvm_slab.h"

namespace ccvm {
void slab_launch_lv(const SlabArgs& a, const SlabPlan& p, hipStream_t st) {
    if (slab_calibrates(a)) slab_launch_lv_cal(a, p, st);
    else launch_slab_cal<MODE_LANGEVIN, false>(a, p, st);
}
}  // namespace ccvm
