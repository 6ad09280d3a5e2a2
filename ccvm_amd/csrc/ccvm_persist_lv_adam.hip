// Persistent-kernel instantiations: MODE_LANGEVIN, Adam variant (see ccvm_persist_launch.h).
#include "ccvm_persist_launch.h"

namespace ccvm {
void persist_launch_lv_adam(const PersistArgs& a, hipStream_t st) { launch_persist<MODE_LANGEVIN, true>(a, st); }
}  // namespace ccvm
