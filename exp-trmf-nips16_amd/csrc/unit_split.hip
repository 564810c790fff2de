// unit_split.hip -- explicit instantiations of the split path of long rows (gram_kernels.hpp "split rows"): the per-item partial
// Gram kernel and the second halves of the F-solve / X-side Gram kernels fed by summed partials
// (kernel_units.hpp: one translation unit per heavy kernel family, compiled in parallel).
#define TRMF_UNIT 1
#include "kernel_units.hpp"

namespace trmf {
TRMF_UNIT_SPLIT(TRMF_DEFINE_KERNEL)
}  // namespace trmf
