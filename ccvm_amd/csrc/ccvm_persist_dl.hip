// Persistent-kernel instantiations: MODE_DL (see ccvm_persist_launch.h).
#include "ccvm_persist_launch.h"

namespace ccvm {
void persist_launch_dl(const PersistArgs& a, hipStream_t st) { launch_persist<MODE_DL, false>(a, st); }
}  // namespace ccvm
