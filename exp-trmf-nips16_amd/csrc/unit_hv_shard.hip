// unit_hv_shard.hip -- explicit instantiations of hv_tile_kernel<MODE, KQ, true>: the launch-per-step CG sharded over time
// (kernel_units.hpp: one translation unit per heavy kernel family, compiled in parallel).
#define TRMF_UNIT 2
#include "kernel_units.hpp"

namespace trmf {
TRMF_UNIT_HV(TRMF_DEFINE_KERNEL, true, 256)
}  // namespace trmf
