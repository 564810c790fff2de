// cg_persist_args.hpp -- what the HOST needs of the persistent CG kernel (cg_persist.hpp): its argument block, table geometry
// and LDS need, and the kernel's declaration.  The main translation unit includes only this file, so that work on the kernel's
// body recompiles unit_persist.hip alone (kernel_units.hpp).
#pragma once

#include "cg_kernels.hpp"

namespace trmf {

constexpr int kLLWords = 8;                          // 8-byte words per record: 4 doubles as (payload half, tag) pairs
constexpr long long kPersistTimeoutTicks = 200000000;   // 2 s of the 100 MHz wall clock
constexpr int kSc1 = 16;                             // aux bit of raw buffer loads / stores: sc1 (device-coherent, around the L2)

typedef unsigned int pu4 __attribute__((ext_vector_type(4)));

struct PersistArgs {
    real *W;                       // T x KP temporal factor: operand of the gradient, updated in place when the step is accepted
    const real *Bv, *G;
    const uint32_t *lag_set;
    const real *theta;
    unsigned long long *hll;       // exchanged vector rows in tagged form: [2 parities][T x KP elements][2 * sizeof(real) bytes]
    unsigned long long *ll;        // records: [2 parities][tiles][kLLWords]
    uint32_t epoch0;               // tag of this launch's first exchange (tags of a buffer only grow; never 0)
    int TI, maxcg;
    XState *log_x;                 // iteration record written by the accept phase (or null)
    double *log_n;
    // several ranks (SHARD): this rank runs the tiles [sh.tile0, sh.tile0 + sh.ntiles) of sh.nbt; `ll` / `hll` are its OWN copies in
    // its IPC-exported arena, peer_ll / peer_hll the other ranks' copies (null for the own rank)
    TileShard sh;
    unsigned long long *peer_ll[kMaxPeers];
    unsigned long long *peer_hll[kMaxPeers];
    long long timeout_ticks;       // bound of every poll (100 MHz ticks; kPersistTimeoutTicks unless TRMF_PERSIST_TIMEOUT_MS says otherwise)
    long long *prof;               // -DTRMF_PERSIST_PROF builds only: cycle stamps of the phases (tile 0 and the middle tile)
    int fail_tile, fail_x;         // test hook (TRMF_TEST + TRMF_PERSIST_FAIL=tile:exchange): that tile never publishes its record of
                                   // that exchange (-2: of the final one, the acceptance test's) -- every workgroup's poll runs into its bound
};
constexpr int kFailFinal = -2;
constexpr int kProfSlots = 8, kProfIters = 32;

constexpr int kPersistMaxTiles = 512;                 // 32 poll chunks of 16 records (one bit each); also the co-residency ceiling of the chip
__host__ __device__ inline size_t persist_lds_bytes(int TI, int midx, int KP, int nlag, int k, int tiles) {
    const size_t vec = ((size_t)(TI + 2 * midx) * KP * sizeof(real) + 15) / 16 * 16;     // d, r, H d on the staged rows
    const size_t own = ((size_t)TI * KP * sizeof(real) + 15) / 16 * 16;                   // s, g on the own rows
    const size_t res = ((size_t)(TI + midx) * hv_res_pitch(k) * sizeof(double) + 15) / 16 * 16;
    const size_t th = ((size_t)nlag * KP * (sizeof(double) + sizeof(real)) + (size_t)nlag * sizeof(int) + 15) / 16 * 16;
    return 3 * vec + 2 * own + res + th + (size_t)tiles * 4 * sizeof(double);             // + the collected records
}

template <int KQ, bool SHARD>
__global__ void cg_persist_kernel(XParams p, XState *__restrict__ st, PersistArgs a);      // defined in cg_persist.hpp, instantiated in unit_persist.hip

}  // namespace trmf
