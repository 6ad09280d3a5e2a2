// Column-cluster persistent kernel instantiations: MODE_DL (see ccvm_cluster.h).
#include "ccvm_cluster.h"

namespace ccvm {
void cluster_launch_dl(const ClusterArgs& a, hipStream_t st) { launch_cluster<MODE_DL>(a, false, st); }
}  // namespace ccvm
