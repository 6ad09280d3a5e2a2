// Column-cluster persistent kernel instantiations: MODE_MF (see ccvm_cluster.h).
#include "ccvm_cluster.h"

namespace ccvm {
void cluster_launch_mf(const ClusterArgs& a, bool adam, hipStream_t st) { launch_cluster<MODE_MF>(a, adam, st); }
}  // namespace ccvm
