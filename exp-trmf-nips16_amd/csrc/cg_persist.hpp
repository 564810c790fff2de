// cg_persist.hpp -- the whole X-solve of the fused path as ONE persistent kernel (round 4).
//
// The fused path of cg_kernels.hpp spends one launch per CG step: hv_tile_kernel re-reads the tile's cached Grams (64 MB per
// launch at config 3), re-stages Theta, re-loads four vectors on 2.3x halo-amplified rows and pays a kernel boundary; the
// profile (profiles/r03_hv_tile_ablation.txt) prices a working step at 20.7 us of which only the arithmetic (~6 us) is
// intrinsic.  Here every tile's workgroup stays resident for the whole solve (gradient, CG, closing step, H s, accept):
//
//   * the thread's slice of the cached Gram is loaded ONCE and lives in registers (the same 160 VGPRs hv_tile_kernel fills per
//     launch); the CG vectors d, r, H d (staged rows incl. halo) and s, g (own rows) live in LDS; Theta is staged once;
//   * workgroups exchange through memory WITHOUT fences: everything published is stored write-through (sc1) and read
//     around the L2 (sc1), so no buffer_wbl2 / buffer_inv is ever needed (an agent-scope release fence costs one L2
//     write-back PER WORKGROUP: 47-95 us per grid barrier, profiles/r03_gridsync_ubench.txt; this protocol 3-5 us,
//     profiles/r04_gridsync2_ubench.txt);
//   * there is no barrier object and no ordering assumption at all: EVERYTHING a workgroup publishes is self-validating.  A
//     value travels as 8-byte words (32-bit payload, 32-bit epoch tag); 8-byte stores are single-copy atomic, so a reader that
//     sees the tag sees the payload (the "LL" protocol of the collective libraries).  A tile's RECORD of a step (its three dot
//     products) is eight such words and is its own arrival flag: every workgroup polls all records of the step and sums them
//     in the single fixed order of hv_tile_kernel.  The tile's rows of H d (of g, of s) travel the same way, and the
//     neighbours poll the midx rows they stage as halo.  (A first version stored the rows plainly (sc1), "waited" for them with a
//     workgroup-scope release fence and read them once the owner's record had arrived: the rows were occasionally STALE -- that
//     fence emits no s_waitcnt vmcnt(0) on gfx950, the stores were never drained before the record went out, and only the latency of
//     the all-to-all had hidden it.  Found by the bit-identity test; profiles/r04_persist_notes.txt.  The price list of
//     /opt/skills/guides/MI355X_MICROARCH.md has both forms -- "handoff-flag" with an explicit asm vmcnt(0) drain, and data-tagged
//     granules at about half its latency; the granules need no drain and no ordering argument, so they are what is used.)  Buffers
//     ping-pong by exchange parity: a workgroup can be at most one exchange ahead of any other.
//
// Several ranks (SHARD, one process per GPU; round 4): the same kernel, one launch per rank over the rank's block of tiles.  The
// exchange is transport-agnostic: a record is stored into EVERY rank's copy of the record table, a row within midx of the
// rank's block boundary also into the neighbouring rank's copy of the row table (system-scope stores into IPC-mapped, uncached
// arenas), and every rank polls its own copies.  All ranks sum all records in the single-GPU order: bit-identical iterates for
// any number of ranks, one kernel per solve per rank, no launch, no collective and no host involvement inside the solve.
//
// Across launches (SHARD): exchange 0 of launch N + 1 stores into the parity-0 slots a slower peer may still be polling for the
// last exchange of launch N.  Nothing in this file orders that: the host does -- xsolve_persist() ends with the all-gather of W
// (a collective every rank enters only after its kernel has finished), so no rank launches solve N + 1 before every rank has
// left solve N.  On one GPU consecutive launches are ordered by the stream.
//
// Same tiles, same per-thread element mapping, same arithmetic and summation order as hv_tile_kernel / cg_close_kernel /
// accept_tile_kernel: the iterates are BIT-IDENTICAL to the launch-per-step path (tests/test_gpu_parity.py compares them).
// Needs every workgroup co-resident (the host checks the grid against the occupancy the runtime reports; a plain launch has the
// same residency as a cooperative one and costs 15-19 us less host time, guide row "coop-launch") and the GPU to itself: used with one
// rank per GPU; every poll is bounded (2 s), a timeout sets XState::p2p_error and ends the kernel.
#pragma once

#include <type_traits>

#include "cg_persist_args.hpp"

namespace trmf {

template <typename R>
__device__ __forceinline__ R buffer_load_sc1(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
    if constexpr (sizeof(R) == 4) return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off, 0, kSc1));
    else return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b64(rsrc, byte_off, 0, kSc1));
}
template <typename R>
__device__ __forceinline__ void buffer_store_sc1(__amdgpu_buffer_rsrc_t rsrc, int byte_off, R x) {
    if constexpr (sizeof(R) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), rsrc, byte_off, 0, kSc1);
    else {
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, x), rsrc, byte_off, 0, kSc1);
    }
}

// 16-byte unit of tagged vector elements {payload, tag, payload, tag}: two fp32 elements, or the two halves of one fp64 element
template <typename R> __device__ __forceinline__ pu4 ll_pack(const R *src, int u, unsigned int tag) {
    if constexpr (sizeof(R) == 4) return pu4{__builtin_bit_cast(unsigned int, src[2 * u]), tag, __builtin_bit_cast(unsigned int, src[2 * u + 1]), tag};
    else {
        const unsigned long long bits = (unsigned long long)__double_as_longlong((double)src[u]);
        return pu4{(unsigned int)bits, tag, (unsigned int)(bits >> 32), tag};
    }
}
template <typename R> __device__ __forceinline__ void ll_unpack(const pu4 &w, R *dst) {
    // (components copied to scalars first: __builtin_bit_cast applied directly to `w.z` read component 0 -- both elements came
    // out as w.x; seen in the ISA, hipcc of ROCm 7.2)
    if constexpr (sizeof(R) == 4) { const unsigned int p0 = w.x, p1 = w.z; dst[0] = __builtin_bit_cast(R, p0); dst[1] = __builtin_bit_cast(R, p1); }
    else dst[0] = (R)__longlong_as_double((long long)(((unsigned long long)w.z << 32) | (unsigned long long)w.x));
}

template <int KQ, bool SHARD, int NTH>
__global__ __launch_bounds__(NTH, NTH == 256 ? 2 : 1) void cg_persist_kernel(XParams p, XState *__restrict__ st, PersistArgs a) {
    constexpr int kAuxLd = SHARD ? (kSc1 | 1) : kSc1;        // polls: device scope on one GPU, system scope when peers store into the arena
    extern __shared__ __attribute__((aligned(16))) unsigned char hv_smem[];
    __shared__ double smem[256];
    constexpr int NW = NTH / 64;                             // wavefronts of the workgroup
    __shared__ double keep[8];          // uniform scalars of the solve (f, |g|, ...): parked in LDS, not in registers, between their uses
    __shared__ int s_fail;
    constexpr int VEC = hv_vec(KQ);
    constexpr int NT_T = (KQ + kTile - 1) / kTile;
    constexpr int KP = kTile * NT_T;
    constexpr int RPITCH = KQ;
    // LDS layout of the fp64 residual rows and of lambdaAR * Theta (this kernel only; the arithmetic does not see it).  The adjoint reads a
    // thread's VEC values of either as 16-byte vectors; with VEC = 4 (fp32, rank <= 40) that was two ds_read_b128 at a 32-byte lane
    // stride -- a 16-lane group then covers 32 sixteen-byte slots of a 16-slot bank row: 2-way conflicts by construction, 3/4 of the
    // kernel's bank-conflict cycles.  The values are therefore kept in NPL planes of EPP values per thread: plane h of
    // all rows first, so that the wave's read of a plane is one contiguous range again.
    constexpr int NPL = VEC * 8 >= 32 ? 2 : 1, EPP = VEC / NPL, RPP = RPITCH / NPL, TPP = KP / NPL;
    const int tid = threadIdx.x;
    const int k = p.k, T = p.T, Hh = p.midx, nlag = p.nlag, TI = a.TI;
    const int rowsV = TI + 2 * Hh, rowsR = TI + Hh, nV = rowsV * KP, nTh = nlag * k;
    const int tile = (SHARD ? a.sh.tile0 : 0) + (int)blockIdx.x, nbt = SHARD ? a.sh.nbt : (int)gridDim.x;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;   // writes this rank's XState
    const int i0 = tile * TI, i1 = min(i0 + TI, T);
    const int own_n = (i1 - i0) * KP, own0 = Hh * KP;       // own rows: staged elements [own0, own0 + own_n)
    const int sz = (int)sizeof(real);
    // ---- LDS ----
    const size_t vecb = ((size_t)nV * sizeof(real) + 15) / 16 * 16, ownb = ((size_t)TI * KP * sizeof(real) + 15) / 16 * 16;
    real *vs = reinterpret_cast<real *>(hv_smem);                       // operand of the product: w, then d, then s (staged rows)
    real *rst = reinterpret_cast<real *>(hv_smem + vecb);               // residual r (staged rows)
    real *hst = reinterpret_cast<real *>(hv_smem + 2 * vecb);           // H d (staged rows: own part local, halo from the neighbours)
    real *sown = reinterpret_cast<real *>(hv_smem + 3 * vecb);          // step s (own rows)
    real *gown = reinterpret_cast<real *>(hv_smem + 3 * vecb + ownb);   // gradient g (own rows)
    double *rs = reinterpret_cast<double *>(hv_smem + 3 * vecb + 2 * ownb);
    double *thd = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(rs) + (((size_t)rowsR * RPITCH * sizeof(double) + 15) / 16 * 16));
    real *thp = reinterpret_cast<real *>(thd + (size_t)nlag * KP);
    // the lag set as the two element offsets the phases need (lag * KP into the staged operand, lag * RPP into a plane of the residual rows),
    // two spare entries each: a loop over pairs of lags reads the NEXT pair's offsets while it works on the current one
    int *lagv = reinterpret_cast<int *>(thp + (size_t)nlag * KP);
    int *lagq = lagv + nlag + 2;
    double *recs = reinterpret_cast<double *>(hv_smem + (((size_t)(reinterpret_cast<unsigned char *>(lagq + nlag + 2) - hv_smem) + 15) / 16 * 16));   // [tiles][4]
    if (tid == 0) s_fail = 0;
    // an earlier solve of this session ran into a poll bound: the iterates are void until the host has recovered (session.hpp,
    // persist_recover); do not spend another bound per queued solve
    if (__hip_atomic_load(&st->p2p_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
#if defined(TRMF_PERSIST_PROF)
    const int prof_sel = blockIdx.x == 0 ? 0 : (int)blockIdx.x == (int)gridDim.x / 2 ? 1 : -1;
    auto stamp = [&](int iter, int slot) {
        if (a.prof && tid == 0 && prof_sel >= 0 && iter < kProfIters) a.prof[((size_t)prof_sel * kProfIters + iter) * kProfSlots + slot] = wall_clock64();
    };
    // every tile, CG iteration 5: [kProfAll + 2 * tile] = collect done, [+1] = record published
    const size_t kProfAll = (size_t)2 * kProfIters * kProfSlots;
    auto stamp_all = [&](int iter, int which) { if (a.prof && tid == 0 && iter == 5) a.prof[kProfAll + 2 * blockIdx.x + which] = wall_clock64(); };
#else
    auto stamp = [&](int, int) {};
    auto stamp_all = [&](int, int) {};
#endif
    stamp(0, 0);

    // ---- the Gram slice: requested first, consumed last (columns [t0, t0+VEC) of timestamp row i0+lr, all KQ rows) ----
    const int tpr = (k + VEC - 1) / VEC;
    const int lr = tid / tpr, t0 = (tid - lr * tpr) * VEC;
    const bool lane_on = lr < TI;
    const int lrc = lane_on ? lr : TI - 1;
    // requests in consumption order (vmcnt retires in order): the operand rows of the gradient (w) first, then the 64 MB Gram
    // stream -- the staging and the AR phases of the gradient run underneath it
    const size_t vec_bytes = (size_t)T * KP * sizeof(real);
    const int vbyte0 = (i0 - Hh) * KP * sz;
    constexpr int kThRegs = NTH == 256 ? kHvThetaRegs : 2;
    real thr[kThRegs];
    int lagr = 0;
    if (nlag > 0) {                  // Theta / the lag set first: their staging must not queue behind the Gram stream
#pragma unroll
        for (int m = 0; m < kThRegs; m++) thr[m] = a.theta[min(tid + NTH * m, nTh - 1)];
        lagr = (int)a.lag_set[min(tid, nlag - 1)];
    }
    // operand elements requested ahead of the Gram stream: nV <= 12 x 256 at every fused shape; the wide tile (more threads, fewer
    // elements per thread) needs 8, and its fp64 rank-40 instantiation has no register to spare
    constexpr int kOpRegs = NTH == 256 ? kHvOperandRegs : 8;
    real vr[kOpRegs];
    {
        const __amdgpu_buffer_rsrc_t v_rsrc = buffer_rsrc(a.W, vec_bytes);
#pragma unroll
        for (int m = 0; m < kOpRegs; m++) vr[m] = buffer_load_real(v_rsrc, vbyte0 + (tid + NTH * m) * sz);   // zeros outside [0, T)
    }
    __builtin_amdgcn_sched_barrier(0);
    GramVec<VEC> gq[KQ];
    {
        const __amdgpu_buffer_rsrc_t g_rsrc = buffer_rsrc(a.G + (size_t)i0 * p.gstride, 0x7fffffff);
        const int g_voff = (int)(((uint32_t)(min(i0 + lrc, T - 1) - i0) * (uint32_t)p.gstride + (uint32_t)t0) * sizeof(real));
        const int rowbytes = k * (int)sizeof(real);
        int g_soff = 0;
#pragma unroll
        for (int j = 0; j < KQ; j++) {
            gq[j] = gram_load<VEC>(g_rsrc, g_voff, g_soff);
            g_soff += (j + 1 < k) ? rowbytes : 0;           // j >= k: a finite duplicate, multiplied by a zero pad
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- Theta / lag set -> LDS, once ----
    if (nlag > 0) {
        auto put = [&](int e, real th) {
            const int tt = e / nlag, l = e - tt * nlag;
            thp[l * KP + colpos(tt, NT_T)] = th;
            thd[(((tt % VEC) / EPP) * nlag + l) * TPP + (tt / VEC) * EPP + tt % EPP] = p.lambdaAR * (double)th;
        };
#pragma unroll
        for (int m = 0; m < kThRegs; m++)
            if (tid + NTH * m < nTh) put(tid + NTH * m, thr[m]);
#pragma nounroll
        for (int e = tid + NTH * kThRegs; e < nTh; e += NTH) put(e, a.theta[e]);         // many lags x high rank only
#pragma nounroll
        for (int e = tid; e < nlag * (KP - k); e += NTH) {
            const int l = e / (KP - k), tt = k + (e - l * (KP - k));
            thp[l * KP + colpos(tt, NT_T)] = 0;
            thd[(((tt % VEC) / EPP) * nlag + l) * TPP + (tt / VEC) * EPP + tt % EPP] = 0;
        }
        if (tid < nlag) { lagv[tid] = lagr * KP; lagq[tid] = lagr * RPP; }
#pragma nounroll
        for (int e = tid + NTH; e < nlag; e += NTH) { const int lg = (int)a.lag_set[e]; lagv[e] = lg * KP; lagq[e] = lg * RPP; }
        if (tid < 2) { lagv[nlag + tid] = 0; lagq[nlag + tid] = 0; }
    }
    for (int e = tid; e < nV; e += NTH) { rst[e] = 0; hst[e] = 0; }
    for (int e = tid; e < TI * KP; e += NTH) { sown[e] = 0; gown[e] = 0; }

    // (descriptors of the exchanged vectors: staged element e = vector element (i0 - midx) * KP + e; offsets outside the
    // vector read 0 -- a negative offset wraps to a huge unsigned -- the clipping hv_tile_kernel relies on)
    const bool ar_on = nlag > 0 && p.lambdaAR > 0;

    // ---- exchange: publish this tile's record of exchange x, collect everybody's ----
    auto publish = [&](int x, double v0, double v1, double v2, double v3, bool final_x = false) {       // thread 0, after the tile's sc1 stores have been waited for
        if (a.fail_tile == tile && a.fail_x == (final_x ? kFailFinal : x)) return;    // test hook: this record never arrives
        const size_t ri = ((size_t)(x & 1) * nbt + tile) * kLLWords;
        const unsigned long long tag = (unsigned long long)(a.epoch0 + (uint32_t)x) << 32;
        const double v[4] = {v0, v1, v2, v3};
        for (int r = 0; r < (SHARD ? a.sh.world : 1); r++) {
            unsigned long long *rec = ((SHARD && r != a.sh.rank) ? a.peer_ll[r] : a.ll) + ri;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v[q]);
                if constexpr (SHARD) {
                    __hip_atomic_store(rec + 2 * q, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(rec + 2 * q + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    __hip_atomic_store(rec + 2 * q, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(rec + 2 * q + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    // Collect exchange x: the records of ALL tiles -> LDS, and (halo) the midx rows on either side of the tile -> dst_staged.
    // Every thread polling every record (the first version) put ~100k pollers on the memory system and an exchange took 10 us
    // (profiles/r04_persist_notes.txt); now
    //   records: wave w owns the chunks w, w + 4, ... of 16 records; four lanes read the four 16-byte quarters of a record -- a
    //     wavefront load is 16 whole records, 1 KB contiguous -- and a quarter that carries the tag is parked in LDS at once (it
    //     validates itself); a chunk is re-read only until all of its quarters have arrived;
    //   halo rows: every thread polls a few 16-byte units of the tagged rows, all requested before the first check;
    // every wave does both in the same pass (requests of both first, then the checks), so an exchange costs one store-to-load
    // latency through memory, not a chain of them.
    // Then every thread sums the records in hv_tile_kernel's order (thread t adds records t, t + 256, ...; fixed-order block
    // sums).  Returns false when a poll timed out (the whole grid then winds down).
    constexpr int EB = 2 * (int)sizeof(real), PER = 16 / EB;                  // tagged vector rows: bytes per element, elements per 16-byte unit
    const size_t hll_elems = (size_t)T * KP;
    const __amdgpu_buffer_rsrc_t ll_rsrc = buffer_rsrc(a.ll, (size_t)2 * nbt * kLLWords * 8);
    const int nchunks = (nbt + 15) / 16;
    auto timed_out = [&](long long &t_start) -> bool {
        const long long now = wall_clock64();
        if (t_start == 0) { t_start = now; return false; }
        return now - t_start > a.timeout_ticks || __hip_atomic_load(&st->p2p_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    };
    auto collect = [&](int x, int NF, double (&sum)[4], bool halo, real *dst_staged) -> bool {
        const uint32_t tag = a.epoch0 + (uint32_t)x;
        // (the record indices of the wave's chunks are otherwise hoisted out of the CG loop and kept for the whole kernel -- the rank-40
        // instantiations have no register for that; an opaque copy of the thread index keeps them local to the call)
        int tid_c = tid;
        asm volatile("" : "+v"(tid_c));
        const int base = (x & 1) * nbt, wave = tid_c >> 6, lane = tid_c & 63;
        // records: wave w owns the chunks w, w + 4, ... (16 records each; lane = (record of the chunk, 16-byte quarter))
        constexpr int NCW = kPersistMaxTiles / 16 / NW;          // chunks per wave at most (8): polled in groups of NC
        constexpr int NC = (KQ * VEC * (int)sizeof(real) / 4 >= 160) ? 2 : 3;   // (the 160-register Gram slices leave room for two requests in flight only)
        const int q = lane & 3, r = lane >> 2;
        uint32_t cpend = 0;                                      // wave-uniform: bit j = chunk wave + 4 j still incomplete
#pragma unroll
        for (int j = 0; j < NCW; j++) if (wave + NW * j < nchunks) cpend |= 1u << j;
        // halo rows: unit u = tid + 256 m of the 2 * midx rows on either side of the tile; rows outside [0, T) are zeros
        constexpr int NU = 3;
        const int units = halo ? 2 * own0 / PER : 0;
        const __amdgpu_buffer_rsrc_t rs_ = buffer_rsrc(reinterpret_cast<const unsigned char *>(a.hll) + (size_t)(x & 1) * hll_elems * EB, hll_elems * EB);
        for (int u0 = 0; u0 < units || cpend; u0 += NTH * NU) {           // (more than 768 halo units: further rounds, records done by then)
            int es[NU]; bool need[NU];
#pragma unroll
            for (int m = 0; m < NU; m++) {
                const int u = u0 + tid + NTH * m, e = u * PER;
                es[m] = e < own0 ? e : e + TI * KP;
                const int i = i0 - Hh + es[m] / KP;
                need[m] = u < units && i >= 0 && i < T;
                if (u < units && !need[m]) {
#pragma unroll
                    for (int c = 0; c < PER; c++) dst_staged[es[m] + c] = 0;
                }
            }
            long long t_start = 0;
            int passes = 0;
            for (;;) {
                // the requests of a group first, then its checks: one memory round trip per group (two groups only while more
                // than four of the wave's chunks are incomplete -- the first pass of a large grid)
                bool pending = false;
#pragma unroll
                for (int g0 = 0; g0 < NCW; g0 += NC) {
                    if (g0 > 0 && !((cpend >> g0) & ((1u << NC) - 1u))) continue;           // wave-uniform
                    pu4 wc[NC], wh[NU];
#pragma unroll
                    for (int j = 0; j < NC; j++)
                        if ((cpend >> (g0 + j)) & 1u)
                            wc[j] = __builtin_amdgcn_raw_buffer_load_b128(ll_rsrc, (base + min(16 * (wave + NW * (g0 + j)) + r, nbt - 1)) * 64 + q * 16, 0, kAuxLd);
                    if (g0 == 0) {
#pragma unroll
                        for (int m = 0; m < NU; m++)
                            if (need[m]) wh[m] = __builtin_amdgcn_raw_buffer_load_b128(rs_, vbyte0 * 2 + es[m] * EB, 0, kAuxLd);
                    }
#pragma unroll
                    for (int j = 0; j < NC; j++) {
                        if (!((cpend >> (g0 + j)) & 1u)) continue;                 // wave-uniform
                        const int rec = 16 * (wave + NW * (g0 + j)) + r;
                        const bool valid = rec < nbt, ok = !valid || (wc[j].y == tag && wc[j].w == tag);
                        if (valid && ok) recs[rec * 4 + q] = __longlong_as_double((long long)(((unsigned long long)wc[j].z << 32) | (unsigned long long)wc[j].x));
                        if (__all(ok)) cpend &= ~(1u << (g0 + j));
                    }
                    if (g0 == 0) {
#pragma unroll
                        for (int m = 0; m < NU; m++) {
                            if (!need[m]) continue;
                            if (wh[m].y == tag && wh[m].w == tag) { ll_unpack<real>(wh[m], dst_staged + es[m]); need[m] = false; }
                            else pending = true;
                        }
                    }
                }
                if (!pending && !cpend) break;
                if ((++passes & 63) == 0 && timed_out(t_start)) { atomicExch(&s_fail, cpend ? 1 : 2); cpend = 0; break; }   // 1: records missing, 2: halo rows
                __builtin_amdgcn_s_sleep(0);
            }
        }
        __syncthreads();
        double acc[4] = {0, 0, 0, 0};
        for (int i = tid; i < nbt; i += NTH) {
            const VecOf<double, 4> v = *reinterpret_cast<const VecOf<double, 4> *>(recs + i * 4);
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] += v.v[q];
        }
        block_allsum3<NW>(acc[0], acc[1], acc[2], smem);
        if (NF > 3) acc[3] = block_allsum<NW>(acc[3], smem);
        sum[0] = acc[0]; sum[1] = acc[1]; sum[2] = acc[2]; sum[3] = acc[3];
        if (s_fail) {                                        // read after the barriers of the block sums: uniform in the workgroup
            if (tid == 0) {
                if (!__hip_atomic_exchange(&st->p2p_error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {   // the first workgroup to give up says where
                    st->p2p_diag[0] = x; st->p2p_diag[1] = tile; st->p2p_diag[2] = s_fail; st->p2p_diag[3] = 0;
                }
            }
            return false;
        }
        return true;
    };
    // the tile's own rows of a vector (LDS) -> memory in tagged form: 16-byte units {payload, tag, payload, tag} -- two fp32
    // elements, or the two halves of one fp64 element -- stored write-through and never waited for
    auto publish_rows = [&](int x, const real *src_own) {
        const uint32_t tag = a.epoch0 + (uint32_t)x;
        const __amdgpu_buffer_rsrc_t rs_ = buffer_rsrc(reinterpret_cast<const unsigned char *>(a.hll) + ((size_t)(x & 1) * hll_elems + (size_t)i0 * KP) * EB,
                                                       (size_t)own_n * EB);
        constexpr int kAuxSt = SHARD ? (kSc1 | 1) : kSc1;
        for (int u = tid; u < own_n / PER; u += NTH) {
            __builtin_amdgcn_raw_buffer_store_b128(ll_pack<real>(src_own, u, tag), rs_, u * 16, 0, kAuxSt);
        }
        if constexpr (SHARD) {
            // rows within midx of this rank's block boundary are the neighbouring rank's halo: also into its copy
            const size_t off = ((size_t)(x & 1) * hll_elems + (size_t)i0 * KP) * EB;
            const bool lo = a.sh.rank > 0 && i0 < a.sh.row_b + Hh, hi = a.sh.rank + 1 < a.sh.world && i1 > a.sh.row_e - Hh;
            if (lo || hi) {
                const __amdgpu_buffer_rsrc_t rlo = buffer_rsrc(reinterpret_cast<const unsigned char *>(lo ? a.peer_hll[a.sh.rank - 1] : a.hll) + off, (size_t)own_n * EB);
                const __amdgpu_buffer_rsrc_t rhi = buffer_rsrc(reinterpret_cast<const unsigned char *>(hi ? a.peer_hll[a.sh.rank + 1] : a.hll) + off, (size_t)own_n * EB);
                for (int u = tid; u < own_n / PER; u += NTH) {
                    const int i = i0 + (u * PER) / KP;
                    const pu4 w = ll_pack<real>(src_own, u, tag);
                    if (lo && i < a.sh.row_b + Hh) __builtin_amdgcn_raw_buffer_store_b128(w, rlo, u * 16, 0, kAuxSt);
                    if (hi && i >= a.sh.row_e - Hh) __builtin_amdgcn_raw_buffer_store_b128(w, rhi, u * 16, 0, kAuxSt);
                }
            }
        }
    };

    // ---- (2) AR residuals of rows [i0, i0+TI+midx) from the staged operand (hv_tile_kernel phase 2, verbatim) ----
    // (The CG loop around these phases invites the compiler to hoist every tid-derived index -- rows, columns, LDS offsets of each
    // unrolled work item -- out of the loop and keep it in registers for the whole kernel: with the Gram slice resident there is
    // no room for that.  Each phase therefore derives its indices from an OPAQUE copy of the thread index.)
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    // Work items a thread may carry through the lag loop TOGETHER (their LDS round trips overlap): two, like hv_tile_kernel's kHvResU.  A
    // wavefront takes the second one only if any of its lanes has one (wave-uniform): at 672 items on 512 threads that is wavefronts 0-2,
    // which used to run a second round of eight dependent trips alone while the other five idled at the barrier.
    // (one where the Gram slice leaves no room: fp64 at rank 33..40 -- 160 registers of slice, 8 per staged quad)
    constexpr int kResU = (sizeof(real) == 8 && KQ * VEC * (int)sizeof(real) / 4 >= 160) ? 1 : 2;
    auto ar_residuals = [&](double &ar2) {
        constexpr int NG = KP / 4;
        const int items = rowsR * NG;
        const int tid = opaque((int)threadIdx.x);
#pragma nounroll
        for (int it0 = 0; it0 < items; it0 += NTH * kResU) {
            // a wavefront none of whose lanes has a work item in this round leaves (wave-uniform)
            const int wave_first = __builtin_amdgcn_readfirstlane(it0 + (tid & ~63));
            if (wave_first >= items) break;
            const bool two = kResU == 2 && wave_first + NTH < items;           // wave-uniform: some lane of this wavefront has a second item
            int vb[kResU], pg[kResU];
            bool on[kResU];
            double res[kResU][4];
#pragma unroll
            for (int u = 0; u < kResU; u++) {
                const int it = it0 + tid + NTH * u;
                const int rr = it / NG, g = it - rr * NG, i = i0 + rr;
                on[u] = it < items && i >= Hh && i < T;
                vb[u] = on[u] ? (rr + Hh) * KP + 4 * g : Hh * KP;
                pg[u] = on[u] ? 4 * g : 0;
                const Quad<real> x4 = *reinterpret_cast<const Quad<real> *>(vs + vb[u]);
#pragma unroll
                for (int c = 0; c < 4; c++) res[u][c] = (double)x4.v[c];
            }
            // Two lags per trip, lags in order (the reference's summation order, trmf.cpp:110-113).  The operand row of a lag sits at an
            // offset that is itself read from LDS: as a plain loop every lag cost TWO dependent LDS round trips (offset, then operand;
            // seen in the ISA: ds_read2_b32 -> s_waitcnt -> v_mul_lo -> ds_read_b128 -> s_waitcnt).  The offsets are now stored
            // pre-multiplied and the next pair's are requested behind this pair's operands: one round trip per pair.
            auto lag_loop = [&](auto uc) {
                constexpr int U = decltype(uc)::value;
                int bn0 = lagv[0], bn1 = lagv[1];
                int l = 0;
#pragma nounroll
                for (; l + 1 < nlag; l += 2) {
                    const int b0 = bn0, b1 = bn1;
                    Quad<real> tha[U], thb[U], xa[U], xb[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        tha[u] = *reinterpret_cast<const Quad<real> *>(thp + l * KP + pg[u]);
                        thb[u] = *reinterpret_cast<const Quad<real> *>(thp + (l + 1) * KP + pg[u]);
                        xa[u] = *reinterpret_cast<const Quad<real> *>(vs + vb[u] - b0);
                        xb[u] = *reinterpret_cast<const Quad<real> *>(vs + vb[u] - b1);
                    }
                    bn0 = lagv[l + 2]; bn1 = lagv[l + 3];
#pragma unroll
                    for (int u = 0; u < U; u++) {
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const real prod = tha[u].v[c] * xa[u].v[c];
                            res[u][c] -= (double)prod;
                        }
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const real prod = thb[u].v[c] * xb[u].v[c];
                            res[u][c] -= (double)prod;
                        }
                    }
                }
                if (l < nlag) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const Quad<real> tha = *reinterpret_cast<const Quad<real> *>(thp + l * KP + pg[u]);
                        const Quad<real> xa = *reinterpret_cast<const Quad<real> *>(vs + vb[u] - bn0);
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const real prod = tha.v[c] * xa.v[c];
                            res[u][c] -= (double)prod;
                        }
                    }
                }
            };
            if (kResU == 2 && two) lag_loop(std::integral_constant<int, kResU>{});
            else lag_loop(std::integral_constant<int, 1>{});
#pragma unroll
            for (int u = 0; u < kResU; u++) {
                const int it = it0 + tid + NTH * u;
                const int rr = it / NG, g = it - rr * NG;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int tl = collog(4 * g + c, NT_T);
                    const double rv2 = on[u] ? res[u][c] : 0.0;
                    if (it < items && tl < k) {
                        if (rr < TI) ar2 += rv2 * rv2;
                        rs[(((tl % VEC) / EPP) * rowsR + rr) * RPP + (tl / VEC) * EPP + tl % EPP] = rv2;
                    }
                }
            }
        }
    };
    // ---- (3) out = lambdaI*v + lambdaAR*AR'(v) + G.v for the thread's row and VEC columns (hv_tile_kernel phase 3); each
    //      finished element goes to `emit(rr, i, c, tcol, tpos, x, acc, od)` ----
    auto product = [&](auto &&emit) {
        // (row and column group re-derived from the thread index on every call -- one integer division -- instead of living in three
        // registers for the whole kernel: the fp32 rank-40 wide instantiation spilled exactly the clamped row)
        const int tidp = opaque((int)threadIdx.x), lrp = tidp / tpr;
        const bool lane_on_p = lrp < TI;
        const int rr = lane_on_p ? lrp : TI - 1, i = i0 + rr, t0 = (tidp - lrp * tpr) * VEC;
        const bool live = lane_on_p && i < T;
        const real *vi = vs + (rr + Hh) * KP;
        double od[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) {
            const real xc = vi[colpos(min(t0 + c, k - 1), NT_T)];
            real o;
            if (p.lambdaI == 0) o = 0;
            else if (p.lambdaI == 1) o = xc;
            else o = (real)(p.lambdaI * (double)xc);
            od[c] = (double)o;
        }
        if (ar_on) {
            {
                const double *r0p = rs + rr * RPP + (t0 / VEC) * EPP;
#pragma unroll
                for (int h = 0; h < NPL; h++) {
                    const VecOf<double, EPP> r0 = *reinterpret_cast<const VecOf<double, EPP> *>(r0p + h * rowsR * RPP);
#pragma unroll
                    for (int e = 0; e < EPP; e++) od[h * EPP + e] += p.lambdaAR * r0.v[e];
                }
            }
            // (lags in pairs, the next pair's residual-row offsets requested ahead: see ar_residuals)
            const double *rrow = rs + rr * RPP + (t0 / VEC) * EPP;            // plane 0; plane h: + h * rowsR * RPP
            const double *trow = thd + (t0 / VEC) * EPP;                       // plane 0 of lag 0; plane h of lag l: + (h * nlag + l) * TPP
            const int rplane = rowsR * RPP, tplane = nlag * TPP;
            int qn0 = lagq[0], qn1 = lagq[1];
            int l = 0;
#pragma nounroll
            for (; l + 1 < nlag; l += 2) {
                const int q0 = qn0, q1 = qn1;
                VecOf<double, EPP> ra[NPL], ta[NPL], rb[NPL], tb[NPL];
#pragma unroll
                for (int h = 0; h < NPL; h++) {
                    ra[h] = *reinterpret_cast<const VecOf<double, EPP> *>(rrow + h * rplane + q0);
                    ta[h] = *reinterpret_cast<const VecOf<double, EPP> *>(trow + h * tplane + l * TPP);
                    rb[h] = *reinterpret_cast<const VecOf<double, EPP> *>(rrow + h * rplane + q1);
                    tb[h] = *reinterpret_cast<const VecOf<double, EPP> *>(trow + h * tplane + (l + 1) * TPP);
                }
                qn0 = lagq[l + 2]; qn1 = lagq[l + 3];
#pragma unroll
                for (int c = 0; c < VEC; c++) od[c] -= ra[c / EPP].v[c % EPP] * ta[c / EPP].v[c % EPP];
#pragma unroll
                for (int c = 0; c < VEC; c++) od[c] -= rb[c / EPP].v[c % EPP] * tb[c / EPP].v[c % EPP];
            }
            if (l < nlag) {
                VecOf<double, EPP> ra[NPL], ta[NPL];
#pragma unroll
                for (int h = 0; h < NPL; h++) {
                    ra[h] = *reinterpret_cast<const VecOf<double, EPP> *>(rrow + h * rplane + qn0);
                    ta[h] = *reinterpret_cast<const VecOf<double, EPP> *>(trow + h * tplane + l * TPP);
                }
#pragma unroll
                for (int c = 0; c < VEC; c++) od[c] -= ra[c / EPP].v[c % EPP] * ta[c / EPP].v[c % EPP];
            }
        }
        double acc[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) acc[c] = 0;
        const real *vip = vi;
        real vcur[4], vnext[4];
#pragma unroll
        for (int u = 0; u < 4; u++) vcur[u] = vip[colpos(u, NT_T)];
#pragma unroll
        for (int j0 = 0; j0 < KQ; j0 += 4) {
            asm volatile("" : "+v"(vip));
            if (j0 + 4 < KQ) {
#pragma unroll
                for (int u = 0; u < 4; u++) vnext[u] = vip[colpos(j0 + 4 + u, NT_T)];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double vj = (double)vcur[u];
                // the resident slice is loop-invariant for the CG loop: without this fence the compiler hoists its CONVERSION to
                // double out of the loop and keeps a second, twice as large copy of the slice in registers (64 of the 185
                // non-Gram registers at rank 8; hundreds of spilled values at rank 40)
#pragma unroll
                for (int c = 0; c < VEC; c++) asm volatile("" : "+v"(gq[j0 + u].c[c]));
#pragma unroll
                for (int c = 0; c < VEC; c++) acc[c] += (double)gq[j0 + u].c[c] * vj;
            }
#pragma unroll
            for (int c = 0; c < VEC; c++) asm volatile("" : "+v"(acc[c]));
            if (j0 + 4 < KQ) {
#pragma unroll
                for (int u = 0; u < 4; u++) { asm volatile("" : "+v"(vnext[u])); vcur[u] = vnext[u]; }
            }
        }
        // column indices and the operand values are re-derived here (from an opaque copy, so that nothing of them stays in
        // registers across the two loops above)
        const int t0b = opaque(t0);
#pragma unroll
        for (int c = 0; c < VEC; c++) {
            const int tcol = min(t0b + c, k - 1), tpos = colpos(tcol, NT_T);
            if (live && t0b + c < k) emit(rr, i, tcol, tpos, vi[tpos], acc[c], od[c]);
        }
    };
    // =========================== gradient: g = H w - b, <g,g>, AR / ridge sums, w.(Gw) - 2 b.w ===========================
    double ar2 = 0, vv = 0, dot = 0, lq = 0;
    {
        auto stage_w = [&](int e, real x) {
            if ((uint32_t)(e - own0) < (uint32_t)own_n) vv += (double)x * (double)x;
            if (e < nV) vs[e] = x;
        };
#pragma unroll
        for (int m = 0; m < kOpRegs; m++) stage_w(tid + NTH * m, vr[m]);
        const __amdgpu_buffer_rsrc_t v_rsrc = buffer_rsrc(a.W, vec_bytes);
#pragma nounroll
        for (int e = tid + NTH * kOpRegs; e < nV; e += NTH) stage_w(e, buffer_load_real(v_rsrc, vbyte0 + e * sz));   // very long halos only
    }
    __syncthreads();
    stamp(0, 1);
    if (ar_on) ar_residuals(ar2);
    __syncthreads();
    stamp(0, 2);
    product([&](int rr, int i, int tcol, int tpos, real x, double ac, double od) {
        const double bb = (double)a.Bv[(size_t)i * KP + tcol];
        lq += (double)x * (ac - 2.0 * bb);
        ac -= bb;
        const real oc = (real)(od + ac);
        gown[rr * KP + tpos] = oc;
        dot += (double)oc * (double)oc;
    });
    block_allsum3<NW>(ar2, vv, dot, smem);
    lq = block_allsum<NW>(lq, smem);                                // (its barriers also order gown before publish_rows)
    int xi = 0;                                                 // exchanges so far: index, parity and tag of the next one (identical in every workgroup)
    stamp(0, 3);
    if (tid == 0) publish(xi, ar2, vv, dot, lq);
    publish_rows(xi, gown);
    double sum[4];
    if (!collect(xi, 4, sum, true, vs)) return;            // + the neighbours' rows of g
    stamp(0, 4);
    xi++;
    // f, |g|, tolerances (rf_tron.h:154-169, 424-439)
    const real ggr = (real)sum[2];
    const real cgtol = (real)(p.eps_cg * sqrt((double)ggr));
    {
        double f = 0.5 * (p.trYTY + sum[3]);
        if (p.lambdaI > 0) f += 0.5 * p.lambdaI * (double)(real)sum[1];
        if (p.nlag > 0 && p.lambdaAR > 0) f += 0.5 * p.lambdaAR * sum[0];
        if (tid == 0) { keep[0] = f; keep[1] = sqrt((double)ggr); keep[2] = (double)ggr; }   // f, |g|, r^T r of the last completed iteration
    }
    double rho_prev_d = (double)ggr, rho_d = rho_prev_d;
    if (lead) st->rho_hist[0] = rho_prev_d;                     // r^T r of iteration 0 = g^T g (the launch-per-step path stores it likewise)
    bool stopped = cg_stopped(ggr, cgtol);
    int stop_it = stopped ? 0 : kCgRunning, cg_iter = stopped ? 0 : 1;

    // =========================== CG ===========================
    real alpha = 0;
    int it = 0;
    // iteration 0: s = 0, r = d = -g on every staged row (the halo rows of g arrived with the records)
    for (int e = tid; e < own_n; e += NTH) vs[own0 + e] = gown[e];
    __syncthreads();
    for (int e = tid; e < nV; e += NTH) { const real x = -vs[e]; vs[e] = x; rst[e] = x; }
    __syncthreads();
    for (;;) {
        if (it > 0) {
            // the three dot products of iteration it-1 -> alpha, r^T r, the stop test, beta (rf_tron.h:444-446, 460, 495-497)
            if (!collect(xi, 3, sum, true, hst)) return;    // + the halo rows of H d(it-1)
            stamp(it + 1, 0);
            stamp_all(it, 0);
            xi++;
            const real rho_prev = (real)rho_prev_d;
            alpha = rho_prev / (real)sum[0];
            const double ad = (double)alpha;
            rho_d = fmax(rho_prev_d - 2.0 * ad * sum[1] + ad * ad * sum[2], 0.0);
            const real rho = (real)rho_d;
            stopped = it == a.maxcg || cg_stopped(rho, cgtol);
            if (lead) st->rho_hist[it] = rho_d;
            if (stopped) { stop_it = it; if (tid == 0) keep[2] = rho_d; break; }
            cg_iter = it + 1;
            const real beta = rho / rho_prev;
            const real tmp = beta - (real)1.0, nalpha = -alpha;
            rho_prev_d = rho_d;
            // element-wise, no sums: any assignment of elements to threads gives the same vectors.  One 16-byte vector of neighbouring
            // elements per thread and array, two trips' worth requested before the first use (all of nV, own0, own_n are multiples of
            // 16): 7 trips of three dependent LDS round trips each became 2 trips of one (the ISA of the scalar form: ds_read -> branch
            // -> ds_read -> wait -> ds_write -> ds_read x2 -> wait -> ds_write x2 per element).
            {
                constexpr int EV = 16 / (int)sizeof(real);
                typedef VecOf<real, EV> V;
                const int tq = opaque((int)threadIdx.x) * EV;
#pragma nounroll
                for (int e0 = 0; e0 < nV; e0 += 2 * EV * NTH) {
                    V xq[2], hq[2], rq[2], sq[2];
                    int e[2]; bool in[2], own[2];
#pragma unroll
                    for (int m = 0; m < 2; m++) {
                        e[m] = e0 + tq + EV * NTH * m;
                        in[m] = e[m] < nV;
                        const int ec = in[m] ? e[m] : 0;
                        own[m] = in[m] && (uint32_t)(ec - own0) < (uint32_t)own_n;
                        xq[m] = *reinterpret_cast<const V *>(vs + ec);
                        hq[m] = *reinterpret_cast<const V *>(hst + ec);
                        rq[m] = *reinterpret_cast<const V *>(rst + ec);
                        sq[m] = *reinterpret_cast<const V *>(sown + (own[m] ? ec - own0 : 0));
                    }
#pragma unroll
                    for (int m = 0; m < 2; m++) {
                        if (!in[m]) continue;
                        V rn, xn, sn;
#pragma unroll
                        for (int c = 0; c < EV; c++) {
                            real x = xq[m].v[c];
                            sn.v[c] = fma(alpha, x, sq[m].v[c]);                         // s += alpha d      (rf_tron.h:461)
                            rn.v[c] = fma(nalpha, hq[m].v[c], rq[m].v[c]);               // r -= alpha Hd     (rf_tron.h:489-490)
                            x = fma(tmp, x, x); xn.v[c] = x + rn.v[c];                   // d = beta d + r    (rf_tron.h:497-499)
                        }
                        if (own[m]) *reinterpret_cast<V *>(sown + (e[m] - own0)) = sn;
                        *reinterpret_cast<V *>(rst + e[m]) = rn;
                        *reinterpret_cast<V *>(vs + e[m]) = xn;
                    }
                }
            }
            __syncthreads();
            stamp(it + 1, 1);
        } else if (stopped) break;                              // the gradient already meets the tolerance
        double d0 = 0, rhd = 0, hh = 0, unused = 0;
        if (ar_on) ar_residuals(unused);
        __syncthreads();
        stamp(it + 1, 2);
        product([&](int rr, int i, int tcol, int tpos, real x, double ac, double od) {
            const real oc = (real)(od + ac);
            hst[own0 + rr * KP + tpos] = oc;
            d0 += (double)x * (double)oc;                                                // <d,Hd>
            rhd += (double)rst[own0 + rr * KP + tpos] * (double)oc;                      // <r,Hd>
            hh += (double)oc * (double)oc;                                               // <Hd,Hd>
        });
        stamp(it + 1, 3);
        block_allsum3<NW>(d0, rhd, hh, smem);
        stamp(it + 1, 4);
        if (tid == 0) publish(xi, d0, rhd, hh, 0);
        stamp_all(it, 1);
        publish_rows(xi, hst + own0);
        stamp(it + 1, 5);
        it++;
    }

    // =========================== close the last completed iteration (cg_close_kernel) ===========================
    // s += alpha d, r' = r - alpha Hd (own rows; stop_it == 0: s = 0, r = -g), sums <g,s>, <s,r'>, <s,s>
    __syncthreads();
    double gs = 0, sr = 0, ss = 0;
    for (int e = tid; e < own_n; e += NTH) {
        real snew = sown[e], rnew = rst[own0 + e];
        if (stop_it >= 1) { snew = fma(alpha, vs[own0 + e], snew); rnew = fma(-alpha, hst[own0 + e], rnew); sown[e] = snew; }
        gs += (double)gown[e] * (double)snew; sr += (double)snew * (double)rnew; ss += (double)snew * (double)snew;
    }
    block_allsum3<NW>(gs, sr, ss, smem);
    // a.direct (diagnostics; TRMF_TEST + TRMF_CG_DIRECT): one more operator pass evaluates s^T H s and the residual -g - H s directly.
    // Otherwise (round 6) this is the solve's LAST exchange: the CG's own recurrence r = -g - H s gives s^T H s = -s^T (g + r) -- what the
    // reference's prered is made of (rf_tron.h:189-190) -- so f(w + s) - f(w) = g.s + 1/2 s.Hs = 1/2 (g.s - s.r): actred = prered, one
    // pass (~12 us at config 3) and the halo rows of s saved.  How far the recurrence is from the direct residual at the stopping
    // step is on record (tests/test_gpu_fullsize.py, < 1e-2 of rho; the exit is at 10 % of |g|).
    const bool direct = a.direct != 0;
    if (tid == 0) publish(xi, gs, sr, ss, 0, !direct);
    if (direct) publish_rows(xi, sown);
    double cs[4];
    if (!collect(xi, 3, cs, direct, vs)) return;           // + the halo rows of s (vs: the direction is no longer needed)
    xi++;
    if (tid == 0) { keep[3] = cs[0]; keep[4] = cs[1]; keep[5] = cs[2]; }     // parked across the product below (registers)

    // =========================== H s and the acceptance test (hv_tile_kernel<HV_PLAIN>, accept_tile_kernel) ===========================
    double ps[4] = {-1.0, 0, 0, 0};
    if (direct) {
        for (int e = tid; e < own_n; e += NTH) vs[own0 + e] = sown[e];
        for (int e = own_n + tid; e < TI * KP; e += NTH) vs[own0 + e] = 0;     // a short last tile: rows past T
        __syncthreads();
        double sHs = 0, unused2 = 0, rdir = 0, unused3 = 0;
        if (ar_on) ar_residuals(unused2);
        __syncthreads();
        product([&](int rr, int i, int tcol, int tpos, real x, double ac, double od) {
            const real oc = (real)(od + ac);
            sHs += (double)x * (double)oc;
            // the residual of the step evaluated directly, -g - H s: the CG's own r^T r is the recurrence rho - 2 alpha <r,Hd> +
            // alpha^2 <Hd,Hd>, and the stop test reads it (its drift against the direct norm is on record now)
            const double rt = (double)gown[rr * KP + tpos] + (double)oc;
            rdir += rt * rt;
        });
        block_allsum3<NW>(sHs, rdir, unused3, smem);
        if (tid == 0) publish(xi, rdir, 0, sHs, 0, true);
        if (!collect(xi, 3, ps, false, nullptr)) return;
    } else __syncthreads();                                                      // keep[] (thread 0) before everybody reads it
    const double gsr = (double)(real)keep[3], srr = (double)(real)keep[4];       // BLAS dots in val_type (rf_tron.h:186-187)
    const double snorm = sqrt((double)(real)keep[5]);
    const double prered = -0.5 * (gsr - srr);                                    // rf_tron.h:190
    const double actred = direct ? -(gsr + 0.5 * ps[2]) : prered;                // f - f(w+s): directly / through the recurrence
    const double f = keep[0], gnorm = keep[1], rho_stop = keep[2];               // (written before many barriers ago)
    const double fnew = f - actred;
    const bool accept = actred > 1e-4 * prered;                                  // eta0, rf_tron.h:222
    if (accept)
        for (int e = tid; e < own_n; e += NTH) { real *wp = a.W + (size_t)i0 * KP + e; *wp = *wp + sown[e]; }   // w_new = w + s (rf_tron.h:183-184)
    stamp(kProfIters - 1, 0);
    if (lead) {
        const double rho = (double)(real)rho_stop;
        st->f = f; st->fnew = fnew; st->gnorm = gnorm; st->cgtol = cgtol; st->gs = gsr; st->sr = srr;
        st->prered = prered; st->actred = actred; st->accepted = accept ? 1 : 0; st->cg_iter = cg_iter;
        st->stop_it = stop_it; st->r_parity = stop_it & 1; st->cg_rnorm = sqrt(rho); st->rho_direct = ps[0];
        double delta = fmin(gnorm, snorm);                                       // trust-region bound of the TRON line: see accept_kernel
        const double curv = fnew - f - gsr;
        const double al = curv <= 0 ? 4.0 : fmax(0.25, -0.5 * (gsr / curv));
        if (actred < 1e-4 * prered) delta = fmin(fmax(al, 0.25) * snorm, 0.5 * delta);
        else if (actred < 0.25 * prered) delta = fmax(0.25 * delta, fmin(al * snorm, 0.5 * delta));
        else if (actred < 0.75 * prered) delta = fmax(0.25 * delta, fmin(al * snorm, 4.0 * delta));
        else delta = fmax(delta, fmin(al * snorm, 4.0 * delta));
        st->delta = delta;
        if (a.log_x) {
            XState *lx = a.log_x;
            lx->f = f; lx->fnew = fnew; lx->gnorm = gnorm; lx->cg_rnorm = sqrt(rho);
            lx->actred = actred; lx->prered = prered; lx->gs = gsr; lx->sr = srr;
            lx->cgtol = cgtol; lx->cg_iter = cg_iter; lx->accepted = accept ? 1 : 0; lx->delta = delta; lx->rho_direct = ps[0];
            a.log_n[0] = a.log_n[1] = a.log_n[2] = -1.0;
        }
    }
}

}  // namespace trmf
