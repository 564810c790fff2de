// unit_persist.hip -- explicit instantiations of cg_persist_kernel<KQ, SHARD>: the whole X-solve as one persistent kernel
// (kernel_units.hpp: one translation unit per heavy kernel family, compiled in parallel).
#define TRMF_UNIT 3
#include "kernel_units.hpp"

namespace trmf {
TRMF_UNIT_PERSIST(TRMF_DEFINE_KERNEL)
}  // namespace trmf
