// unit_hv_wide.hip -- explicit instantiations of hv_tile_kernel<MODE, KQ, false, 512>: the launch-per-step CG of one rank over wide tiles (512 threads per workgroup,
// one workgroup per CU) -- the bit-identical twin of cg_persist_kernel<KQ, false, 512>
// (kernel_units.hpp: one translation unit per heavy kernel family, compiled in parallel).
#define TRMF_UNIT 2
#include "kernel_units.hpp"

namespace trmf {
TRMF_UNIT_HV(TRMF_DEFINE_KERNEL, false, 512)
}  // namespace trmf
