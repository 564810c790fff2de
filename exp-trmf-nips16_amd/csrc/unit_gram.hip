// unit_gram.hip -- explicit instantiations of the F-solve kernel of this element type, gram_x_kernel, loss_kernel
// (kernel_units.hpp: one translation unit per heavy kernel family, compiled in parallel).
#define TRMF_UNIT 1
#include "kernel_units.hpp"

namespace trmf {
TRMF_UNIT_GRAM(TRMF_DEFINE_KERNEL)
}  // namespace trmf
