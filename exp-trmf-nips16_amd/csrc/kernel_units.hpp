// kernel_units.hpp -- which translation unit compiles which kernel instantiations (round 5).
//
// Rounds 1-4 built each library from ONE translation unit: ~150 kernel instantiations, 5-6 minutes per library.  The four heavy
// template families are now instantiated explicitly in units of their own (csrc/unit_*.hip), compiled in parallel; the main
// unit (trmf_abi.hip: the host code and every other kernel) sees their declarations only and references their host stubs.  All units of a library are linked into one .so; each registers its own code object with the runtime, and a stub's
// address is the kernel's handle in every unit (checked on the GPU: scripts/ubench/tu_split).
//
//   unit_hv_rep.hip    hv_tile_kernel<MODE, KQ, false>   (32)  one rank / replicated CG, launch per step
//   unit_hv_wide.hip   hv_tile_kernel<MODE, KQ, false, 512> (32)  the same over wide tiles (512 threads; one rank)
//   unit_hv_shard.hip  hv_tile_kernel<MODE, KQ, true>    (32)  time-sharded CG, launch per step
//   unit_persist.hip   cg_persist_kernel<KQ, SHARD, NTH> (24)  the persistent CG kernel (256 threads: one rank / sharded; 512: one rank, wide tiles)
//   unit_gram.hip      fsolve_quad / fsolve_mfma (8), gram_x_kernel (16), loss_kernel (4)
//   unit_split.hip     the split path of long rows (round 6): gram_part_kernel (8), fsolve_*_long_kernel (8), gram_x_long_kernel (16)
//   unit_full.hip      the MFMA kernels of the full-observation path (20)
//
// A unit defines TRMF_UNIT before including this file; the non-template kernels of the shared headers are compiled by the main
// unit only (#if !defined(TRMF_UNIT) around them).
#pragma once

// (a unit includes only the headers of its own family: TRMF_UNIT = 1 gram, 2 hv_tile, 3 persist, 4 full; the kernels of these families
// have their BODIES only where TRMF_UNIT_BODIES is defined -- the main unit sees declarations, so that it neither compiles them
// nor runs them through the optimiser as `extern template` would (available_externally bodies: 3 of its 3.5 minutes))
#if defined(TRMF_UNIT) || defined(TRMF_SINGLE_UNIT)
#define TRMF_UNIT_BODIES 1
#endif
#if !defined(TRMF_UNIT) || TRMF_UNIT == 1
#include "gram_kernels.hpp"
#endif
#if !defined(TRMF_UNIT) || TRMF_UNIT == 2
#include "cg_kernels.hpp"
#endif
#if defined(TRMF_UNIT) && TRMF_UNIT == 4
#include "cg_kernels.hpp"
#include "gram_kernels.hpp"
#include "full_kernels.hpp"
#endif
#if !defined(TRMF_UNIT)
#include "cg_persist_args.hpp"     // the declaration only: the body is unit_persist.hip's business
#elif TRMF_UNIT == 3
#include "cg_persist.hpp"
#endif
#if defined(TRMF_SINGLE_UNIT)
#include "cg_persist.hpp"
#endif

namespace trmf {

#define TRMF_HV_SIG (XParams, XState *, HvVecs, TileShard, int, int, const uint32_t *, const real *, const real *, const double *, double *, const PeerTable *, int, int)
#define TRMF_HV_KQS(X, MODE, SHARD, NTH)                           \
    X void hv_tile_kernel<MODE, 8, SHARD, NTH> TRMF_HV_SIG;        \
    X void hv_tile_kernel<MODE, 16, SHARD, NTH> TRMF_HV_SIG;       \
    X void hv_tile_kernel<MODE, 24, SHARD, NTH> TRMF_HV_SIG;       \
    X void hv_tile_kernel<MODE, 32, SHARD, NTH> TRMF_HV_SIG;       \
    X void hv_tile_kernel<MODE, 40, SHARD, NTH> TRMF_HV_SIG;       \
    X void hv_tile_kernel<MODE, 48, SHARD, NTH> TRMF_HV_SIG;       \
    X void hv_tile_kernel<MODE, 56, SHARD, NTH> TRMF_HV_SIG;       \
    X void hv_tile_kernel<MODE, 64, SHARD, NTH> TRMF_HV_SIG;
#define TRMF_UNIT_HV(X, SHARD, NTH)                                                                                   \
    TRMF_HV_KQS(X, HV_GRAD, SHARD, NTH) TRMF_HV_KQS(X, HV_CG_FIRST, SHARD, NTH) TRMF_HV_KQS(X, HV_CG_STEP, SHARD, NTH) TRMF_HV_KQS(X, HV_PLAIN, SHARD, NTH)

#define TRMF_PERSIST_SIG (XParams, XState *, PersistArgs)
#define TRMF_PERSIST_KQS(X, SHARD, NTH)                                  \
    X void cg_persist_kernel<8, SHARD, NTH> TRMF_PERSIST_SIG;            \
    X void cg_persist_kernel<16, SHARD, NTH> TRMF_PERSIST_SIG;           \
    X void cg_persist_kernel<24, SHARD, NTH> TRMF_PERSIST_SIG;           \
    X void cg_persist_kernel<32, SHARD, NTH> TRMF_PERSIST_SIG;           \
    X void cg_persist_kernel<40, SHARD, NTH> TRMF_PERSIST_SIG;           \
    X void cg_persist_kernel<48, SHARD, NTH> TRMF_PERSIST_SIG;           \
    X void cg_persist_kernel<56, SHARD, NTH> TRMF_PERSIST_SIG;           \
    X void cg_persist_kernel<64, SHARD, NTH> TRMF_PERSIST_SIG;
#define TRMF_UNIT_PERSIST(X) TRMF_PERSIST_KQS(X, false, 256) TRMF_PERSIST_KQS(X, true, 256) TRMF_PERSIST_KQS(X, false, 512)

#define TRMF_FSOLVE_SIG (const uint32_t *, const uint32_t *, const real *, const real *, real *, uint32_t, uint32_t, int, real, uint32_t, uint32_t)
#define TRMF_FSOLVE_LONG_SIG (SplitRows, real *, int, real)
#if defined(TRMF_F32)
#define TRMF_FSOLVE_ONE(X, NT, KMAX) X void fsolve_quad_kernel<NT, KMAX, 0> TRMF_FSOLVE_SIG;
#define TRMF_FSOLVE_LONG_ONE(X, NT, KMAX) X void fsolve_quad_long_kernel<NT, KMAX> TRMF_FSOLVE_LONG_SIG;
#else
#define TRMF_FSOLVE_ONE(X, NT, KMAX) X void fsolve_mfma_kernel<NT, KMAX> TRMF_FSOLVE_SIG;
#define TRMF_FSOLVE_LONG_ONE(X, NT, KMAX) X void fsolve_mfma_long_kernel<NT, KMAX> TRMF_FSOLVE_LONG_SIG;
#endif
#define TRMF_GRAMX_SIG (const uint32_t *, const uint32_t *, const real *, const real *, real *, real *, uint32_t, uint32_t, int, uint32_t, size_t, uint32_t, const uint32_t *)
// the split path of long rows (gram_kernels.hpp "split rows"): partial Grams per item, then the row kernels' second halves
#define TRMF_GRAMX_LONG_SIG (SplitRows, real *, real *, int, size_t)
#define TRMF_GRAM_PART_SIG (const uint32_t *, const real *, const real *, const uint32_t *, uint32_t, uint32_t, real *, uint32_t, uint32_t)
#define TRMF_SPLIT_NT(X, NT)                                               \
    X void gram_part_kernel<NT, true> TRMF_GRAM_PART_SIG;                  \
    X void gram_part_kernel<NT, false> TRMF_GRAM_PART_SIG;                 \
    X void gram_x_long_kernel<NT, true, true> TRMF_GRAMX_LONG_SIG;         \
    X void gram_x_long_kernel<NT, true, false> TRMF_GRAMX_LONG_SIG;        \
    X void gram_x_long_kernel<NT, false, true> TRMF_GRAMX_LONG_SIG;        \
    X void gram_x_long_kernel<NT, false, false> TRMF_GRAMX_LONG_SIG;
#define TRMF_UNIT_SPLIT(X)                                                                                                   \
    TRMF_FSOLVE_LONG_ONE(X, 1, 8) TRMF_FSOLVE_LONG_ONE(X, 1, 16) TRMF_FSOLVE_LONG_ONE(X, 2, 24) TRMF_FSOLVE_LONG_ONE(X, 2, 32) \
    TRMF_FSOLVE_LONG_ONE(X, 3, 40) TRMF_FSOLVE_LONG_ONE(X, 3, 48) TRMF_FSOLVE_LONG_ONE(X, 4, 56) TRMF_FSOLVE_LONG_ONE(X, 4, 64) \
    TRMF_SPLIT_NT(X, 1) TRMF_SPLIT_NT(X, 2) TRMF_SPLIT_NT(X, 3) TRMF_SPLIT_NT(X, 4)
#define TRMF_GRAMX_NT(X, NT)                                     \
    X void gram_x_kernel<NT, true, true> TRMF_GRAMX_SIG;         \
    X void gram_x_kernel<NT, true, false> TRMF_GRAMX_SIG;        \
    X void gram_x_kernel<NT, false, true> TRMF_GRAMX_SIG;        \
    X void gram_x_kernel<NT, false, false> TRMF_GRAMX_SIG;
#define TRMF_LOSS_SIG (const uint32_t *, const uint32_t *, const real *, const real *, const real *, double *, uint32_t, uint32_t, uint32_t)
#define TRMF_UNIT_GRAM(X)                                                                                                    \
    TRMF_FSOLVE_ONE(X, 1, 8) TRMF_FSOLVE_ONE(X, 1, 16) TRMF_FSOLVE_ONE(X, 2, 24) TRMF_FSOLVE_ONE(X, 2, 32)                   \
    TRMF_FSOLVE_ONE(X, 3, 40) TRMF_FSOLVE_ONE(X, 3, 48) TRMF_FSOLVE_ONE(X, 4, 56) TRMF_FSOLVE_ONE(X, 4, 64)                  \
    TRMF_GRAMX_NT(X, 1) TRMF_GRAMX_NT(X, 2) TRMF_GRAMX_NT(X, 3) TRMF_GRAMX_NT(X, 4)                                         \
    X void loss_kernel<1> TRMF_LOSS_SIG; X void loss_kernel<2> TRMF_LOSS_SIG; X void loss_kernel<3> TRMF_LOSS_SIG; X void loss_kernel<4> TRMF_LOSS_SIG;

// the full-observation path's MFMA kernels (full_kernels.hpp).  On their own they compile in seconds; in one module with the
// non-template kernels of cg_kernels.hpp / generic_kernels.hpp LLVM's CodeGenPrepare spent 175 s of the main unit's 180 s.
#define TRMF_APPLY_SHARED_SIG (XParams, const XState *, int, const real *, const real *, const real *, const real *, const real *, int, real *, int, double *, int, int, int)
#define TRMF_FULL_NT(X, NT)                                                                                                              \
    X void spmm_rows_kernel<NT>(const uint32_t *, const uint32_t *, const real *, const real *, real *, uint32_t, uint32_t, uint32_t, uint32_t);  \
    X void spmm_part_kernel<NT>(const uint32_t *, const real *, const real *, const uint32_t *, uint32_t, uint32_t, real *, uint32_t);  \
    X void dense_tn_mfma_kernel<NT>(const real *, int, int, const real *, double *);                                                    \
    X void small_gram_mfma_kernel<NT>(const real *, int, int, double *);                                                                \
    X void chol_wave_kernel<NT>(const real *, real *, int);                                                                             \
    X void apply_shared_mfma_kernel<NT> TRMF_APPLY_SHARED_SIG;
#define TRMF_UNIT_FULL(X) TRMF_FULL_NT(X, 1) TRMF_FULL_NT(X, 2) TRMF_FULL_NT(X, 3) TRMF_FULL_NT(X, 4)

#define TRMF_DEFINE_KERNEL template __global__

}  // namespace trmf
