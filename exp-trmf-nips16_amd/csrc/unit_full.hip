// unit_full.hip -- explicit instantiations of the full-observation path's MFMA kernels (spmm_rows, dense_tn_mfma, small_gram_mfma,
// chol_wave, apply_shared_mfma; kernel_units.hpp: one translation unit per heavy kernel family, compiled in parallel).
#define TRMF_UNIT 4
#include "kernel_units.hpp"

namespace trmf {
TRMF_UNIT_FULL(TRMF_DEFINE_KERNEL)
}  // namespace trmf
