// Persistent-kernel instantiations: MODE_LANGEVIN (see ccvm_persist_launch.h).
#include "ccvm_persist_launch.h"

namespace ccvm {
void persist_launch_lv(const PersistArgs& a, hipStream_t st) { launch_persist<MODE_LANGEVIN, false>(a, st); }
}  // namespace ccvm
