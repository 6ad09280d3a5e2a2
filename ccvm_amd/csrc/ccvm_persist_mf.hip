// Persistent-kernel instantiations: MODE_MF (see ccvm_persist_launch.h).
#include "ccvm_persist_launch.h"

namespace ccvm {
void persist_launch_mf(const PersistArgs& a, hipStream_t st) { launch_persist<MODE_MF, false>(a, st); }
}  // namespace ccvm
