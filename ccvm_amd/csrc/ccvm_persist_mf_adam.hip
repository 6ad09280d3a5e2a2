// Persistent-kernel instantiations: MODE_MF, Adam variant (see ccvm_persist_launch.h).
#include "ccvm_persist_launch.h"

namespace ccvm {
void persist_launch_mf_adam(const PersistArgs& a, hipStream_t st) { launch_persist<MODE_MF, true>(a, st); }
}  // namespace ccvm
