// unit_hv_rep.hip -- explicit instantiations of hv_tile_kernel<MODE, KQ, false>: the launch-per-step CG of one rank / the replicated form
// (kernel_units.hpp: one translation unit per heavy kernel family, compiled in parallel).
#define TRMF_UNIT 2
#include "kernel_units.hpp"

namespace trmf {
TRMF_UNIT_HV(TRMF_DEFINE_KERNEL, false, 256)
}  // namespace trmf
