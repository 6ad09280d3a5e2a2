// Column-cluster persistent kernel instantiations: MODE_LANGEVIN (see ccvm_cluster.h).
#include "ccvm_cluster.h"

namespace ccvm {
void cluster_launch_lv(const ClusterArgs& a, bool adam, hipStream_t st) { launch_cluster<MODE_LANGEVIN>(a, adam, st); }
}  // namespace ccvm
