// session.hpp -- host side of the MI355X TRMF solver: HBM-resident problem state and the outer ALS
// loop (trmf.cpp:599-694) expressed as asynchronous kernel launches on one HIP stream.
//
// HBM layout (all resident for the lifetime of a session):
//   Yc_*   CSC of Y viewed as CSR over items   (F-solve rows):  ptr u32[n+1], idx u32[nnz], val[nnz]
//   Yr_*   CSR of Y over timestamps            (X-side rows):   ptr u32[T+1], idx u32[nnz], val[nnz]
//   W      T x KP, H  n x KP   (KP = k rounded up to 16, zero padded, row-major)
//   theta  |L| x k column-major (as the ABI delivers it)
//   G      T x k x k   cached per-timestamp Gram,  Bv  T x KP rhs,  lossrow  T doubles
//   CG     g, s, r/r1, d0/d1, Hd/Hd1, w_new, arbase: T x KP each; partial-sum arrays
//
// With several ranks (one process per GPU) the nnz-heavy kernels (F-solve, X-side Gram, loss) run
// on this rank's contiguous row block and the results are all-gathered; the CG itself runs
// replicated and bit-identical on every rank (DESIGN.md "Multi-GPU").
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <vector>

#include "../../include/trmf_abi.h"
#include "kernel_units.hpp"      // cg_kernels.hpp, cg_persist.hpp, gram_kernels.hpp + which unit compiles which instantiation
#include "comm.hpp"
#include "full_kernels.hpp"
#include "generic_kernels.hpp"
#include "common.hpp"
#include "device_pool.hpp"
#include "resident_kernels.hpp"
#include "theta_kernels.hpp"

namespace trmf {

std::shared_ptr<Comm> active_comm();   // trmf_abi.hip

// Stream that zero-fills of fresh buffers are ordered on (the owning session's solver stream, set for the duration of
// every entry point that allocates).  hipMemset runs on the NULL stream and may return before the fill has executed, and
// the solver's stream is non-blocking (never ordered against the NULL stream): filling ON the solver stream orders the
// fill before every later user without a device-wide wait per buffer (round 2: one hipDeviceSynchronize per buffer,
// ~25 of them per create / append_rows).
inline hipStream_t &fill_stream() { static thread_local hipStream_t s = nullptr; return s; }
struct FillStreamScope {
    hipStream_t prev;
    explicit FillStreamScope(hipStream_t s) : prev(fill_stream()) { fill_stream() = s; }
    ~FillStreamScope() { fill_stream() = prev; }
};

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;       // elements in use / allocated (a buffer that shrinks or regrows within cap is reused)
    DevicePool *pool = nullptr;  // where `p` came from (device_pool.hpp: slabs shared by the sessions of this process)
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { pool->free(p); p = nullptr; n = 0; cap = 0; } }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(pool, o.pool); }
    int alloc(size_t count, bool zero = true) {
        const size_t want = std::max<size_t>(count, 1);
        if (!p || want > cap) {
            release();
            pool = &DevicePool::current();
            p = static_cast<T *>(pool->alloc(want * sizeof(T)));
            if (!p) { set_error("device allocation of " + std::to_string(want * sizeof(T)) + " bytes failed"); return kFail; }
            cap = want;
        }
        n = count;
        if (zero) {
            if (fill_stream()) TRMF_HIP_CHECK(hipMemsetAsync(p, 0, want * sizeof(T), fill_stream()));
            else {              // no session stream known: fill on the NULL stream and wait for it
                TRMF_HIP_CHECK(hipMemset(p, 0, want * sizeof(T)));
                TRMF_HIP_CHECK(hipDeviceSynchronize());
            }
        }
        return 0;
    }
    // Host array -> this buffer through the library's pinned ring (device_pool.hpp), ordered on the owning session's stream;
    // returns once the SOURCE has been read (the caller's array may go away), not when the bytes have landed.
    int upload(const T *src, size_t count) {
        if (alloc(count, false)) return kFail;
        if (!count) return 0;
        if (fill_stream()) return HostStager::current().h2d(p, src, count * sizeof(T), fill_stream());
        TRMF_HIP_CHECK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
};

struct DeviceIterLog {       // one per ALS iteration, filled on device
    double normF, normX, normLV;
    XState x;
};

struct PhaseEvents { hipEvent_t f0, fk0, fk1, f1, xg1, x1, lv1; };   // xg1: end of the X-side Gram build (start of the CG)

struct TrmfSessionImpl {
    // problem
    int T = 0, n = 0, k = 0, KP = 0, NT = 0, KMAX = 0, nlag = 0, midx = 0;
    uint64_t nnz = 0;
    double lambdaI = 0, lambdaAR = 0, lambdaLag = 0;
    int period_W = 1, period_H = 1, period_Lag = 2, verbose = 0;
    bool log_norms = true;       // ||.||^2 records of the iteration log (the reference: only under verbose)
    int max_cg_iter = 20;        // 10 * 2, trmf.h:90-93 folded by trmf.cpp:603-606
    double eps_cg = 0.1;
    int iter = 0;                // ALS iterations done so far
    // distribution
    std::shared_ptr<Comm> comm;      // shared with the library: outlives trmf_dist_finalize() while the session lives
    std::vector<uint64_t> fbounds, xbounds;   // row partitions of items / timestamps
    // device
    hipStream_t stream = nullptr;
    DevBuf<uint32_t> Yc_ptr, Yc_idx, Yr_ptr, Yr_idx, lag_set, lag_steps;   // lag_steps: ar_lag_steps() of the lag set
    int nsteps = 0;
    DevBuf<real> Yc_val, Yr_val, W, H, theta, G, Bv, g, s, r, r1, d0, d1, Hd, Hd1, w_new;
    DevBuf<double> lossrow, partials, theta_part;
    // full-observation path (missing == 0)
    bool full = false, dense = false;
    DevBuf<real> Yd_tn, Yd_nt;                // dense Y as T x n and as n x T (both row-major)
    DevBuf<real> Bf, GSf, GSx, Uf;            // F-side right-hand sides (n x KP), shared Grams (k x k), Cholesky factor of GSf
    DevBuf<double> gemm_part, sgram_part;
    double trYTY = 0;
    static constexpr int kGemmChunks = 32, kSmallGramBlocks = 256;
    DevBuf<XState> xstate;
    DevBuf<DeviceIterLog> log;
    static constexpr int kLogCap = 4096;
    std::vector<PhaseEvents> events;
    static constexpr int kEventRing = 64;
    int nbe = 1, nba = 1, rpb = 1;            // grids of the elementwise / apply kernels
    bool generic = false;                     // 64 < k <= 256: generic_kernels.hpp for the Grams / the F-solve, unfused CG
    DevBuf<real> gen_scratch, theta_scratch;  // k x k systems of the generic F-solve; |L| x |L| systems of long lag sets
    bool gpacked = false;                     // unfused path: G holds upper triangles (packed_gram_elems(k) per timestamp), apply_kernel<true>
    int tile_TI = 0, nbt = 1;                 // fused Hv kernel: timestamps per tile (0 = unfused path), tiles of the problem
    // fused path: per-tile records of each launch (cg_kernels.hpp "per-tile partial records") in three message buffers:
    // CG launches of even / odd iteration, gradient + plain launch.  tsh: the one-rank view (one slot, every tile);
    // tsh_rank: this rank's block of tiles when the CG is sharded over time (ts_possible).
    DevBuf<double> xmsg_own[3];               // backing store of the messages unless they live in the peer-to-peer arena
    double *xm[3] = {nullptr, nullptr, nullptr};
    TileShard tsh{}, tsh_rank{};
    // peer-to-peer exchange (TRMF_CG=p2p; cg_kernels.hpp "peer-to-peer form of the exchange"): messages + flag words of
    // this rank in one IPC-exported arena, the peers' arenas opened, the pointer table in device memory
    struct P2p {
        bool on = false;                      // arena allocated, exported, mapped by every peer, and the trial exchange passed on EVERY rank
        void *arena = nullptr;
        size_t bytes = 0;
        std::vector<void *> peer;             // opened arenas of the other ranks (own slot: nullptr)
        unsigned long long epoch[3] = {0, 0, 0};
        double *msg[3] = {nullptr, nullptr, nullptr};   // this rank's three messages inside the arena
        size_t ext_off = 0, ext_bytes = 0;    // tail of the arena: the persistent kernel's record table + tagged vector rows (cg_persist.hpp, SHARD)
        size_t ext_ll_bytes = 0;
        std::string note;                     // why the peer-to-peer transport is unavailable (empty: available or not tried)
    } p2p;
    bool p2p_use = false;                     // transport of the CURRENT X-solve (select_transport)
    DevBuf<PeerTable> peer_table;
    std::vector<uint64_t> tbounds;            // tile-aligned timestamp partition of the time-sharded CG
    bool ts_possible = false;
    // Form of the multi-GPU X-solve: the CG replicated on every rank, or sharded over time with the
    // per-launch exchange through the communicator or peer to peer.  Forced by TRMF_CG, else measured once: every candidate
    // runs two X phases (the second timed on every rank), the slowest rank's time decides.  The peer-to-peer transport is
    // a candidate whenever its set-up (IPC arenas + a trial exchange with a short bound) succeeded on every rank.
    enum { kXRep = 0, kXTsComm = 1, kXTsP2p = 2, kXTsPersist = 3, kXForms = 4 };
    int x_form = kXRep;                       // the decided form; -1 while the candidates are being measured
    std::vector<int> x_cands;
    int x_calls = 0, cg_pred = 4;
    float x_ms[kXForms] = {0, 0, 0, 0};       // X phase of the measured call of each candidate (this rank)
    double x_ms_all[kXForms] = {0, 0, 0, 0};     // ... the slowest rank's (after the decision)
    hipEvent_t ts0 = nullptr, ts1 = nullptr;
    int ar_TI = 64, nbar = 1;                 // unfused path: timestamps per ar_tile_kernel workgroup, its partial-sum slots
    DevBuf<real> arbase;                      // lambdaI*v + lambdaAR*AR'(v) between ar_tile_kernel and apply_kernel
    XParams xp{};

    ~TrmfSessionImpl() {
        // nothing of this session may still be running when its buffers go back to the pool and its stream to the cache
        if (stream) (void)hipStreamSynchronize(stream);
        if (side) (void)hipStreamSynchronize(side);
        for (auto &e : events) {
            hipEvent_t all[] = {e.f0, e.fk0, e.fk1, e.f1, e.xg1, e.x1, e.lv1};
            for (hipEvent_t ev : all) if (ev) (void)hipEventDestroy(ev);
        }
        for (hipEvent_t ev : {gx0, gx1, gx2, fs0, fs1, fs2, ts0, ts1}) if (ev) (void)hipEventDestroy(ev);
        release_p2p();
        for (hipEvent_t ev : {ov_b, ov_c[0], ov_c[1], ov_c[2], ov_c[3]}) if (ev) (void)hipEventDestroy(ev);
        if (side) (void)hipStreamDestroy(side);
        StreamCache::release(stream);
    }
    // Knobs that exist for the tests and the measurement scripts (forced failures, forced forms, ablations) are read only when
    // TRMF_TEST is set; INTEGRATION.md lists the production knobs.
    static bool test_knobs() { static const bool on = getenv("TRMF_TEST") != nullptr; return on; }
    static const char *test_env(const char *name) { return test_knobs() ? getenv(name) : nullptr; }
    void release_p2p() {
        for (void *q : p2p.peer) if (q) (void)hipIpcCloseMemHandle(q);
        p2p.peer.clear();
        if (p2p.arena) (void)hipFree(p2p.arena);
        p2p.arena = nullptr; p2p.on = false; p2p_use = false; pbase_override = nullptr;
        for (int m = 0; m < 3; m++) p2p.msg[m] = nullptr;
    }
    // test hook TRMF_P2P_FAIL=<stage>[:rank] (stage: alloc | export | open | fence): the set-up fails there (on that rank only)
    bool p2p_forced_failure(const char *stage) const {
        const char *e = test_env("TRMF_P2P_FAIL");
        if (!e) return false;
        const std::string v(e);
        const size_t c = v.find(':');
        if (v.substr(0, c) != stage) return false;
        return c == std::string::npos || atoi(v.c_str() + c + 1) == comm->rank;
    }
    // One arena per rank: [message 0 | message 1 | message 2 | flag words: 3 messages x world source ranks x 64 bytes].
    // COLLECTIVE, and the outcome is an agreement: every rank allocates (uncached device memory: peers store into it while
    // local kernels poll it -- without that allocation flavour there is NO peer-to-peer transport), exports its
    // handle, the handles and an ok flag travel through the communicator, every rank maps the other arenas, the ok flags
    // travel again, and one flags-only exchange with a SHORT bound (200 ms) runs as a trial.  If any step failed on any
    // rank, every rank releases what it has and the session goes on with the communicator transport (p2p.note says why;
    // one line on stderr under verbose or TRMF_P2P_VERBOSE).  `required` (TRMF_CG=p2p: explicitly requested) turns
    // "unavailable" into an error instead.  Returns kFail only for that and for a failing communicator.
    int setup_p2p(size_t msg_doubles, bool required, size_t ext_ll_bytes = 0, size_t ext_hll_bytes = 0) {
        const int W_ = comm->world, me = comm->rank;
        auto unavailable = [&](const std::string &why) -> int {          // taken by EVERY rank together
            release_p2p();
            p2p.note = why;
            if (required) { set_error("TRMF_CG=p2p: " + why); return kFail; }
            if (me == 0 && (verbose || getenv("TRMF_P2P_VERBOSE")))
                fprintf(stderr, ">> peer-to-peer exchange unavailable (%s): the time-sharded CG uses the communicator\n", why.c_str());
            return 0;
        };
        p2p.note.clear();
        if (W_ > kMaxPeers) return unavailable("more than 8 ranks");
        constexpr size_t kSlot = 128;                                     // [0..63] IPC handle, [64] ok flag
        static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle slot");
        DevBuf<unsigned char> slots;
        if (slots.alloc(kSlot * W_)) return kFail;
        std::vector<unsigned char> all(kSlot * W_);
        auto agree = [&](const unsigned char *mine, int *who_failed) -> int {   // all-gather of one slot per rank; ok = byte 64
            TRMF_HIP_CHECK(hipMemcpyAsync(slots.p + kSlot * me, mine, kSlot, hipMemcpyHostToDevice, stream));
            if (comm->allgather_slots(slots.p, kSlot, stream)) return kFail;
            TRMF_HIP_CHECK(hipStreamSynchronize(stream));
            TRMF_HIP_CHECK(hipMemcpy(all.data(), slots.p, all.size(), hipMemcpyDeviceToHost));
            *who_failed = -1;
            for (int r = 0; r < W_; r++) if (!all[kSlot * r + 64]) { *who_failed = r; break; }
            return 0;
        };
        // ---- stage 1: arena + handle ----
        const size_t msg_bytes = (msg_doubles * sizeof(double) + 255) / 256 * 256, flag_bytes = (size_t)3 * W_ * kFlagStride * sizeof(unsigned long long);
        p2p.ext_off = (3 * msg_bytes + flag_bytes + 255) / 256 * 256;
        p2p.ext_ll_bytes = (ext_ll_bytes + 255) / 256 * 256;
        p2p.ext_bytes = p2p.ext_ll_bytes + ext_hll_bytes;
        p2p.bytes = p2p.ext_off + p2p.ext_bytes;
        unsigned char mine[kSlot] = {0};
        bool ok = !p2p_forced_failure("alloc") && hipExtMallocWithFlags(&p2p.arena, p2p.bytes, hipDeviceMallocUncached) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); p2p.arena = nullptr; }
        if (ok) {
            ok = hipMemsetAsync(p2p.arena, 0, p2p.bytes, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
            hipIpcMemHandle_t h;
            ok = ok && !p2p_forced_failure("export") && hipIpcGetMemHandle(&h, p2p.arena) == hipSuccess;
            if (ok) std::memcpy(mine, &h, sizeof h); else (void)hipGetLastError();
        }
        mine[64] = ok ? 1 : 0;
        int bad = -1;
        if (agree(mine, &bad)) return kFail;
        if (bad >= 0) return unavailable("rank " + std::to_string(bad) + " could not allocate / export an uncached IPC arena");
        const std::vector<unsigned char> handles = all;
        // ---- stage 2: map the peers' arenas ----
        p2p.peer.assign(W_, nullptr);
        PeerTable tab{};
        ok = !p2p_forced_failure("open");
        for (int r = 0; r < W_ && ok; r++) {
            unsigned char *base = (unsigned char *)p2p.arena;
            if (r != me) {
                hipIpcMemHandle_t h;
                std::memcpy(&h, handles.data() + kSlot * r, sizeof h);
                if (hipIpcOpenMemHandle(&p2p.peer[r], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); p2p.peer[r] = nullptr; ok = false; break; }
                base = (unsigned char *)p2p.peer[r];
            }
            for (int m = 0; m < 3; m++) {
                tab.msg[m][r] = reinterpret_cast<double *>(base + m * msg_bytes);
                tab.flags[m][r] = reinterpret_cast<unsigned long long *>(base + 3 * msg_bytes) + (size_t)m * W_ * kFlagStride;
            }
        }
        std::memset(mine, 0, sizeof mine); mine[64] = ok ? 1 : 0;
        if (agree(mine, &bad)) return kFail;          // also: nobody starts writing into a peer before every rank has opened every arena
        if (bad >= 0) return unavailable("rank " + std::to_string(bad) + " could not map a peer's arena");
        if (peer_table.upload(&tab, 1)) return kFail;
        for (int m = 0; m < 3; m++) { p2p.msg[m] = tab.msg[m][me]; p2p.epoch[m] = 0; }
        // ---- stage 3: a trial exchange (flags only) with a short bound ----
        TileShard sh{}; sh.rank = me; sh.world = W_;
        hipLaunchKernelGGL(xchg_sync_kernel, dim3(1), dim3(256), 0, stream, peer_table.p, 1, ++p2p.epoch[1], xstate.p, -1, sh, 0, KP, 0,
                           (real *)nullptr, (real *)nullptr, (real *)nullptr, kP2pTrialTicks);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
        XState hx;
        if (ok) { TRMF_HIP_CHECK(hipMemcpy(&hx, xstate.p, sizeof hx, hipMemcpyDeviceToHost)); ok = !hx.p2p_error; }
        if (p2p_forced_failure("fence")) ok = false;
        std::memset(mine, 0, sizeof mine); mine[64] = ok ? 1 : 0;
        if (agree(mine, &bad)) return kFail;
        if (bad >= 0) {
            TRMF_HIP_CHECK(hipMemset(&xstate.p->p2p_error, 0, sizeof(int)));      // the trial's failure is not the session's
            return unavailable("the trial flag exchange timed out on rank " + std::to_string(bad));
        }
        p2p.on = true;
        return 0;
    }
    // message buffers / partial-sum base of the transport the next X-solve uses
    void select_transport(bool use_p2p) {
        p2p_use = use_p2p && p2p.on;
        for (int m = 0; m < 3; m++) xm[m] = p2p_use ? p2p.msg[m] : xmsg_own[m].p;
        pbase_override = (p2p_use && uts) ? p2p.msg[1] : nullptr;
    }

    // base of the partial-sum arrays: the session's own buffer, or -- peer-to-peer time-sharded unfused CG -- message 1 of the arena
    double *pbase_override = nullptr;
    double *pbase() { return pbase_override ? pbase_override : partials.p; }
    double *P(int slot) { return pbase() + (size_t)slot * xp.pstride; }

    // ---------------------------------------------------------------------------------------------
    // Factors carry one extra all-zero row at index `rows` (operand of masked-out MFMA lanes).  The ABI's rows x k
    // arrays cross PCIe as they are; padding and column interleaving happen on the device (a host loop took 1.5 s for
    // the 512 MB item factor of config 5).
    // `raw` (the unpadded copy) must stay alive until the pad kernel has run: a member of the session, released by
    // finish_setup() after the set-up's one synchronisation.
    int upload_padded(DevBuf<real> &dst, DevBuf<real> &raw, const real *src, size_t rows) {
        if (raw.upload(src, rows * (size_t)k) || dst.alloc((rows + 1) * (size_t)KP, false)) return kFail;
        const size_t N = (rows + 1) * (size_t)KP;
        hipLaunchKernelGGL(factor_pad_kernel, dim3((unsigned)std::min<size_t>(4096, (N + 255) / 256)), dim3(256), 0, stream,
                           raw.p, rows, k, KP, NT, dst.p);
        TRMF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    int download_padded(const DevBuf<real> &src, real *dst, size_t rows) {
        if (rows == 0) return 0;
        DevBuf<real> raw;
        const size_t N = rows * (size_t)k;
        if (raw.alloc(N, false)) return kFail;
        hipLaunchKernelGGL(factor_unpad_kernel, dim3((unsigned)std::min<size_t>(4096, (N + 255) / 256)), dim3(256), 0, stream,
                           src.p, rows, k, KP, NT, raw.p);
        TRMF_HIP_CHECK(hipGetLastError());
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        TRMF_HIP_CHECK(hipMemcpy(dst, raw.p, N * sizeof(real), hipMemcpyDeviceToHost));
        return 0;
    }
    // All three factors -> host staging in one stream-ordered batch (pinned memory of the library when they fit, ordinary
    // memory otherwise); nothing of the caller's is touched.  commit() then copies them out: c_trmf_train's outputs change
    // together or not at all (the reference's contract for a failed call, trmf.cpp:632-634).
    struct StagedFactors {
        unsigned char *base = nullptr;
        std::unique_lock<std::mutex> lease;
        std::vector<unsigned char> fallback;
        size_t bW = 0, bH = 0, bL = 0;
        void commit(void *Wout, void *Hout, void *Lout) const {
            HostStager::parallel_copy(Wout, base, bW);
            HostStager::parallel_copy(Hout, base + bW, bH);
            if (bL) std::memcpy(Lout, base + bW + bH, bL);
        }
    };
    int download_staged(StagedFactors &sf) {
        sf.bW = (size_t)T * k * sizeof(real); sf.bH = (size_t)n * k * sizeof(real); sf.bL = (size_t)nlag * k * sizeof(real);
        const size_t total = sf.bW + sf.bH + sf.bL;
        sf.base = HostStager::current().staging(total, sf.lease);
        if (!sf.base) {
            try { sf.fallback.resize(total); } catch (const std::bad_alloc &) { set_error("host staging of the factors: out of memory"); return kFail; }
            sf.base = sf.fallback.data();
        }
        DevBuf<real> rawW, rawH;
        if (rawW.alloc((size_t)T * k, false) || rawH.alloc((size_t)n * k, false)) return kFail;
        auto unpad = [&](const DevBuf<real> &src, size_t rows, real *dst) {
            const size_t N = rows * (size_t)k;
            if (N) hipLaunchKernelGGL(factor_unpad_kernel, dim3((unsigned)std::min<size_t>(4096, (N + 255) / 256)), dim3(256), 0, stream, src.p, rows, k, KP, NT, dst);
        };
        unpad(W, T, rawW.p); unpad(H, n, rawH.p);
        TRMF_HIP_CHECK(hipGetLastError());
        if (sf.bW) TRMF_HIP_CHECK(hipMemcpyAsync(sf.base, rawW.p, sf.bW, hipMemcpyDeviceToHost, stream));
        if (sf.bH) TRMF_HIP_CHECK(hipMemcpyAsync(sf.base + sf.bW, rawH.p, sf.bH, hipMemcpyDeviceToHost, stream));
        if (sf.bL) TRMF_HIP_CHECK(hipMemcpyAsync(sf.base + sf.bW + sf.bH, theta.p, sf.bL, hipMemcpyDeviceToHost, stream));
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        if (test_env("TRMF_FAIL_DOWNLOAD")) { set_error("download failure forced by TRMF_FAIL_DOWNLOAD"); return kFail; }   // test hook
        return 0;
    }
    // the ABI's 64-bit pointer arrays are narrowed on their way through the pinned ring (nnz < 2^32 is checked at the boundary)
    int upload_ptr32(DevBuf<uint32_t> &dst, const size_t *src, size_t count) {
        if (dst.alloc(count, false)) return kFail;
        return HostStager::current().h2d_narrow(dst.p, (const uint64_t *)src, count, stream);
    }
    // sum of squares of a device value array, fp64 (fixed order): the kernel is enqueued here, the partial sums are read by
    // finish_sum_squares() after the caller's next synchronisation of the stream
    static constexpr int kSumsqBlocks = 1024;
    DevBuf<double> sumsq_part;
    int launch_sum_squares(const real *dv, size_t count) {
        if (sumsq_part.alloc(kSumsqBlocks)) return kFail;
        hipLaunchKernelGGL(sumsq_values_kernel, dim3(kSumsqBlocks), dim3(256), 0, stream, dv, count, sumsq_part.p);
        TRMF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    int finish_sum_squares(double *out) {
        std::vector<double> h(kSumsqBlocks);
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        TRMF_HIP_CHECK(hipMemcpy(h.data(), sumsq_part.p, kSumsqBlocks * sizeof(double), hipMemcpyDeviceToHost));
        double acc = 0;
        for (double x : h) acc += x;
        *out = acc;
        return 0;
    }
    int device_sum_squares(const real *dv, size_t count, double *out) { return launch_sum_squares(dv, count) || finish_sum_squares(out) ? kFail : 0; }

    // Host copies of the two pointer arrays (8 bytes per row/column): row partitions, the byte model of
    // fsolve_bytes(), and the merged pointers of append_rows() are derived from them.
    std::vector<uint64_t> host_row_ptr, host_col_ptr;
    double ysq_acc = 0;          // sum of y^2 over every entry uploaded so far (fp64)

    // ---- set-up (trmf_session_create; the first part of every c_trmf_train call) ----------------------------------------
    // Everything is enqueued on the session's stream and the host waits ONCE, at the end: the caller's arrays travel through the
    // library's pinned ring (device_pool.hpp), the 64-bit pointers are narrowed on the way, the factors are padded / interleaved
    // and sum y^2 is formed on the device; device memory comes from the process-level pool, the stream from the stream cache.
    // Round 4 took 0.12 s for config 3 here (a hipMalloc + a synchronous pageable hipMemcpy per array, ~10 stream
    // synchronisations, ~30 hipFree at the end); profiles/r05_oneshot.txt has the split now.
    double t_upload_s = 0;       // seconds of create() spent reading the caller's arrays (TrmfTrainProfile.upload_s)
    double bytes_uploaded = 0;
    DevBuf<real> raw_W, raw_H;   // unpadded factor uploads: alive until the set-up's synchronisation
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    size_t footprint_estimate() const {
        const size_t sz = sizeof(real), NVb = (size_t)(T + 1) * KP * sz, NHb = (size_t)(n + 1) * KP * sz;
        size_t b = dense ? 2 * (size_t)T * n * sz + (full ? (size_t)kGemmChunks * std::max(T, n) * KP * sz : 0)
                         : 2 * (size_t)nnz * (4 + sz) + ((size_t)T + n + 2) * 4;
        b += NVb + NHb + (size_t)(T + n) * k * sz;              // factors and their unpadded upload copies
        b += 11 * NVb + 4 * NVb;                                 // CG vectors, rhs; tagged rows of the persistent kernel
        if (!full) b += (size_t)T * k * k * sz;                  // Gram cache (packed on the unfused path: an upper bound)
        if (full) b += NHb;
        if (generic) b += (size_t)std::min(kGenBlocks, std::max(n, 1)) * k * k * sz;
        b += (size_t)kLogCap * sizeof(DeviceIterLog) + ((size_t)32 << 20);
        return b + b / 16;
    }
    int create(const PyMatrix *Y, const uint32_t *lags, uint32_t lag_size, const PyMatrix *Wm,
               const PyMatrix *Hm, const PyMatrix *LVm) {
        T = (int)Y->rows; n = (int)Y->cols; k = (int)Wm->cols; nnz = Y->nnz;
        KP = padded_rank(k); NT = KP / kTile; KMAX = ((k + 7) / 8) * 8;
        generic = k > kMaxRank;
        nlag = (int)lag_size; midx = nlag ? (int)lags[nlag - 1] : 0;
        comm = active_comm();
        if (const char *e = test_env("TRMF_DEBUG_ABLATE")) dbg_flags = atoi(e);
        if (StreamCache::acquire(&stream)) return kFail;
        FillStreamScope fill(stream);
        dense = Y->type != TRMF_SPARSE;
        DevicePool::current().reserve(footprint_estimate());

        const double tu0 = now_s();
        if (!dense) {
            host_row_ptr.assign(Y->row_ptr, Y->row_ptr + (size_t)T + 1);
            host_col_ptr.assign(Y->col_ptr, Y->col_ptr + (size_t)n + 1);
            // both orientations cross PCIe as the caller holds them: the CSC's entry order IS the F-solve's summation order
            // (the reference's, trmf.cpp:369-397), and a caller's arrays need not be the canonical transpose of its CSR (the
            // reference's own coo path keeps duplicate entries apart, rf_util.py:98-118) -- deriving one orientation from the
            // other on the device would save 80 MB = 1.7 ms of the ring's time at config 3 and give up that guarantee
            if (upload_ptr32(Yc_ptr, Y->col_ptr, (size_t)n + 1)) return kFail;      // CSC first: the first F-solve needs it
            if (Yc_idx.upload(Y->row_idx, nnz)) return kFail;
            if (Yc_val.upload((const real *)Y->val, nnz)) return kFail;
            if (upload_ptr32(Yr_ptr, Y->row_ptr, (size_t)T + 1)) return kFail;
            if (Yr_idx.upload(Y->col_idx, nnz)) return kFail;
            if (Yr_val.upload((const real *)Y->val_t, nnz)) return kFail;
            if (launch_sum_squares(Yr_val.p, nnz)) return kFail;
            bytes_uploaded += 2.0 * (double)nnz * (4 + sizeof(real)) + 8.0 * ((double)T + n + 2);
        } else {
            // dense Y (only legal with missing == 0): keep both orientations, like CSR + CSC
            std::vector<real> tn;
            ysq_acc = dense_rows_to_rowmajor(Y, tn);
            if (Yd_tn.upload(tn.data(), tn.size()) || Yd_nt.alloc((size_t)T * n, false)) return kFail;
            launch_transpose(Yd_tn.p, T, n, Yd_nt.p);
            bytes_uploaded += (double)tn.size() * sizeof(real);
        }
        if (lag_set.upload(lags, nlag)) return kFail;
        {
            const std::vector<uint32_t> steps = ar_lag_steps(lags, nlag);
            nsteps = (int)steps.size();
            if (lag_steps.upload(steps.data(), steps.size())) return kFail;
        }
        if (upload_padded(W, raw_W, (const real *)Wm->val, T)) return kFail;
        if (upload_padded(H, raw_H, (const real *)Hm->val, n)) return kFail;
        if (theta.upload((const real *)LVm->val, (size_t)nlag * k)) return kFail;
        bytes_uploaded += ((double)T + n + nlag) * k * sizeof(real);
        t_upload_s = now_s() - tu0;

        if (xstate.alloc(1) || log.alloc(kLogCap)) return kFail;
        if (full && (Bf.alloc((size_t)n * KP) || GSf.alloc((size_t)k * k) || Uf.alloc((size_t)k * k) || GSx.alloc((size_t)k * k + kHvGramPad) ||
                     sgram_part.alloc((size_t)kSmallGramBlocks * k * k)))
            return kFail;
        if (comm->world > 1 && count_ranks_per_device()) return kFail;
        if (alloc_time_scratch()) return kFail;

        events.resize(kEventRing);
        for (auto &e : events) {
            hipEvent_t *all[] = {&e.f0, &e.fk0, &e.fk1, &e.f1, &e.xg1, &e.x1, &e.lv1};
            for (hipEvent_t *ev : all) { *ev = nullptr; TRMF_HIP_CHECK(hipEventCreate(ev)); }
        }
        for (hipEvent_t *ev : {&gx0, &gx1, &gx2, &fs0, &fs1, &fs2, &ts0, &ts1}) TRMF_HIP_CHECK(hipEventCreate(ev));
        if (gramx_times.alloc((size_t)8 * comm->world)) return kFail;
        if (comm->world == 1) { gramx_mode = kGramxShard; fs_mode = kShardOn; }     // nothing to decide
        if (const char *e = test_env("TRMF_GRAMX")) gramx_mode = (e[0] == 'r') ? kGramxReplicate : kGramxShard;
        if (const char *e = test_env("TRMF_FSHARD")) fs_mode = (e[0] == 'r') ? kShardOff : kShardOn;
        // the set-up's one synchronisation: uploads landed, factors padded, sum y^2 formed
        if (!dense) { if (finish_sum_squares(&ysq_acc)) return kFail; }
        else TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        set_trYTY();
        raw_W.release(); raw_H.release(); sumsq_part.release();
        if (comm->world > 1) {
            // one small gather now: the communicator's connections are set up before the ALS loop (and before the timed
            // gathers of the shard decisions).  The values are this rank's own zeros, the buffer is rewritten before use.
            std::vector<uint64_t> off(comm->world + 1);
            for (int r = 0; r <= comm->world; r++) off[r] = (uint64_t)r * 2 * sizeof(double);
            if (comm->allgatherv(gramx_times.p, off.data(), stream)) return kFail;
            TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        }
        created = true;
        return autotune();
    }
    bool created = false;        // create() has finished: append_rows() re-tunes, alloc_time_scratch() inside create() does not

    static double sum_squares(const real *v, uint64_t count) {
        double acc = 0;
        for (uint64_t e = 0; e < count; e++) acc += (double)v[e] * (double)v[e];
        return acc;
    }
    // rows of a dense PyMatrix (either memory order) as one row-major block; returns the sum of squares
    static double dense_rows_to_rowmajor(const PyMatrix *Y, std::vector<real> &tn) {
        const size_t R = Y->rows, C = Y->cols;
        const real *v = (const real *)Y->val;
        tn.resize(R * C);
        double acc = 0;
        if (Y->type == TRMF_DENSE_ROWMAJOR) {
            std::memcpy(tn.data(), v, R * C * sizeof(real));
            for (size_t e = 0; e < R * C; e++) acc += (double)v[e] * (double)v[e];
        } else {
            for (size_t j = 0; j < R; j++)
                for (size_t i = 0; i < C; i++) { const real y = v[i * R + j]; tn[j * C + i] = y; acc += (double)y * (double)y; }
        }
        return acc;
    }
    void launch_transpose(const real *src, int rows, int cols, real *dst) {
        if (rows > 0 && cols > 0)
            hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, stream, src, rows, cols, dst);
    }
    // full: do_dot_product(Y, Y) in val_type (trmf.cpp:184); observed-entries path: kept in double, it is the
    // constant of  loss(w) = sum y^2 + sum_i (w_i^T G_i w_i - 2 b_i.w_i)
    void set_trYTY() { trYTY = (full || dense) ? (double)(real)ysq_acc : ysq_acc; xp.trYTY = trYTY; }

    // Everything whose size depends on the number of timestamps T (and the row partitions): called by create()
    // and again by append_rows().
    int alloc_time_scratch() {
        const size_t NV = (size_t)T * KP;
        if (full && dense && gemm_part.alloc((size_t)kGemmChunks * (size_t)std::max(T, n) * KP)) return kFail;
        if (Bv.alloc(NV) || g.alloc(NV) || s.alloc(NV) || r.alloc(NV) ||
            d0.alloc(NV) || d1.alloc(NV) || Hd.alloc(NV) || r1.alloc(NV) || Hd1.alloc(NV) || w_new.alloc(NV) ||
            lossrow.alloc(T))
            return kFail;
        const int nchunk = std::max(1, (T - midx + kThetaChunk - 1) / kThetaChunk);
        const int npairs = nlag * (nlag + 1) / 2 + nlag;
        if (theta_part.alloc((size_t)k * nchunk * std::max(npairs, 1))) return kFail;
        if (nlag && allow_dyn_lds(theta_gram_kernel, theta_gram_lds(), "Theta Gram (max lag too large)")) return kFail;
        if (nlag && theta_solve_lds() > kLdsMax) {          // long lag sets: the |L| x |L| systems in global scratch
            if (theta_scratch.alloc((size_t)k * ((size_t)nlag * nlag + nlag), false)) return kFail;
        } else if (nlag && allow_dyn_lds(theta_solve_kernel, theta_solve_lds(), "Theta solve (too many lags)")) return kFail;
        if (generic) {
            if (gen_scratch.alloc((size_t)kGenBlocks * k * k, false)) return kFail;
            if (allow_dyn_lds(gram_generic_kernel<true>, gram_generic_lds(k), "generic F-solve") ||
                allow_dyn_lds(gram_generic_kernel<false>, gram_generic_lds(k), "generic Gram build")) return kFail;
        }

        nbe = (int)std::min<size_t>(kMaxPartials, (NV + 255) / 256);
        rpb = std::max(1, 256 / k);
        nba = std::min(kMaxPartials, (T + rpb - 1) / rpb);
        if (full && !generic) {      // apply_shared_mfma_kernel: a 16-row tile per wavefront and pass, <= 2 workgroups per CU resident, equal passes
            const int blocks = ((T + kApplyTile - 1) / kApplyTile + 3) / 4, passes = (blocks + 511) / 512;
            nba = std::max(1, (blocks + passes - 1) / passes);
            const size_t need = apply_shared_lds_bytes(KP);
            int rc = 0;
            switch (NT) {
                case 1: rc = allow_dyn_lds(apply_shared_mfma_kernel<1>, need, "shared-Gram product"); break;
                case 2: rc = allow_dyn_lds(apply_shared_mfma_kernel<2>, need, "shared-Gram product"); break;
                case 3: rc = allow_dyn_lds(apply_shared_mfma_kernel<3>, need, "shared-Gram product"); break;
                default: rc = allow_dyn_lds(apply_shared_mfma_kernel<4>, need, "shared-Gram product"); break;
            }
            if (rc) return kFail;
        }
        tile_TI = 0; nbt = 1; persist_state = 0; persist_shard_state = 0; persist_failed = false; persist_note.clear(); snap_iter = -1;
        {   // fused Hv tile: one timestamp row per 16-byte Gram column group, if the AR halo fits a modest LDS budget
            int TI = hv_tile_rows(k);
            if (const char *e = test_env("TRMF_HV_TI")) TI = std::max(1, std::min(TI, atoi(e)));   // experiments
            // the tile kernel addresses the CG vectors with 32-bit byte offsets through buffer descriptors
            const bool fits32 = (uint64_t)(T + 1) * KP * sizeof(real) < 0x7fffffffull;
            if (!generic && fits32 && hv_tile_lds_bytes(TI, midx, KP, nlag, k) <= 48 * 1024 && !test_env("TRMF_NO_HV_TILE")) {
                tile_TI = TI;
                nbt = (T + TI - 1) / TI;                     // one tile per workgroup
            }
        }
        // The cached Grams: k x k per timestamp for the fused kernel; the unfused path's product streams them once per CG
        // step and nothing else (1.64 GB per step at config 5), so there only the upper triangle is kept (packed_gram_elems)
        gpacked = !full && !generic && tile_TI == 0 && !test_env("TRMF_GRAM_FULL");
        const size_t gelems = gpacked ? packed_gram_elems(k) : (size_t)k * k;
        if (G.alloc((full ? 1 : (size_t)T * gelems) + kHvGramPad)) return kFail;
        if (gpacked) {
            const size_t need = (size_t)apply_stages(k) * 512 * sizeof(real);
            if (allow_dyn_lds(apply_kernel<true, 5>, need, "packed cached-Gram product") ||
                allow_dyn_lds(apply_kernel<true, 10>, need, "packed cached-Gram product") ||
                allow_dyn_lds(apply_kernel<true, 17>, need, "packed cached-Gram product"))
                return kFail;
        }
        if (setup_tile_messages()) return kFail;
        {   // unfused path: timestamps per AR tile.  One workgroup per CU (LDS); a tile costs ~(TI + 2 midx) staged rows,
            // (TI + midx) residual rows and TI output rows, and the grid runs in ceil(tiles * column groups / CUs) rounds:
            // take the tile count with the cheapest schedule among those whose halo fits the 150 KB LDS budget (tiles of at
            // most 1024 - midx timestamps keep all residuals in registers and need half the LDS: ar_tile_one_pass).
            hipDeviceProp_t prop;
            int dev = 0;
            TRMF_HIP_CHECK(hipGetDevice(&dev));
            TRMF_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
            const int cus = std::max(1, prop.multiProcessorCount), groups = KP / kArCols;
            int ti_max = (T + kArU - 1) / kArU * kArU;
            while (ti_max > kArU && ar_tile_lds_bytes(ti_max, midx, nlag) > 150 * 1024) ti_max = std::max(kArU, (ti_max / 2 + kArU - 1) / kArU * kArU);
            while (ti_max + kArU <= T && ar_tile_lds_bytes(ti_max + kArU, midx, nlag) <= 150 * 1024) ti_max += kArU;
            const int nt_min = (T + ti_max - 1) / ti_max;
            double best = 0;
            for (int nt = nt_min; nt <= 4 * nt_min + 1; nt++) {
                const int ti = ((T + nt - 1) / nt + kArU - 1) / kArU * kArU;
                const int tiles = (T + ti - 1) / ti;
                // a round is dominated by the tile's serial chain (measured: 15.2 us at TI=280, 16.0 us at TI=416), rows add little
                const double cost = (double)((tiles * groups + cus - 1) / cus) * (3.0 * ti + 3.0 * midx + 2000);
                if (best == 0 || cost < best) { best = cost; ar_TI = ti; }
            }
            if (const char *e = test_env("TRMF_AR_TI")) ar_TI = std::max(kArU, atoi(e) / kArU * kArU);   // experiments
            const size_t need = ar_tile_lds_bytes(ar_TI, midx, nlag);
            if (allow_dyn_lds(ar_tile_kernel<AR_PLAIN>, need, "AR operator (max lag too large)") ||
                allow_dyn_lds(ar_tile_kernel<AR_CG_STEP>, need, "AR operator (max lag too large)"))
                return kFail;
            nbar = ((T + ar_TI - 1) / ar_TI) * (KP / kArCols);
            if (arbase.alloc(NV)) return kFail;
        }
        // with several ranks the apply kernel's partial slots are divided among them: kShardSlots in all, so that a rank's
        // share of the rows still launches enough workgroups to stream its Grams at full rate (128 of 1024 slots per rank
        // on 8 GPUs ran config 5's product at half the bandwidth)
        xp.pstride = std::max(std::max(kMaxPartials, comm->world > 1 ? kShardSlots : 0), std::max(nbt, nbar));
        if (partials.alloc((size_t)P_NSLOTS * xp.pstride)) return kFail;
        xp.T = T; xp.k = k; xp.KP = KP; xp.NT = NT; xp.nlag = nlag; xp.midx = midx;
        xp.lambdaI = lambdaI; xp.lambdaAR = lambdaAR; xp.eps_cg = eps_cg;
        xp.full = full ? 1 : 0; xp.gstride = full ? 0 : gelems; xp.trYTY = trYTY;

        fbounds.resize(comm->world + 1); xbounds.resize(comm->world + 1); fcut.clear();
        if (!dense) {
            partition_by_nnz<uint64_t>((uint64_t)n, host_col_ptr.data(), comm->world, fbounds.data());
            partition_by_nnz<uint64_t>((uint64_t)T, host_row_ptr.data(), comm->world, xbounds.data());
        } else {
            for (int r = 0; r <= comm->world; r++) {
                fbounds[r] = (uint64_t)n * r / comm->world;
                xbounds[r] = (uint64_t)T * r / comm->world;
            }
        }
        if (decide_cg_shard()) return kFail;
        init_x_forms();
        return 0;
    }

    // Message buffers of the fused path and the tile partition of the time-sharded CG: rank r owns
    // the tiles [r * tpr, (r + 1) * tpr) -- a contiguous block of timestamps -- and needs, per launch, the other
    // ranks' tile records (three scalars per CG step) and midx rows of halo from each neighbour.  Possible when every
    // rank holds at least one tile and at least midx timestamps (halo rows then come from the direct neighbours only).
    int setup_tile_messages() {
        tsh = TileShard{};
        tsh.rank = 0; tsh.world = 1; tsh.tile0 = 0; tsh.ntiles = nbt; tsh.nbt = nbt; tsh.tpr = std::max(nbt, 1);
        tsh.row_b = 0; tsh.row_e = T; tsh.slot_dbl = (unsigned)nbt * kRecDoubles; tsh.edge_off_dbl = tsh.slot_dbl;
        tsh_rank = tsh;
        ts_possible = false;
        tbounds.assign(comm->world + 1, (uint64_t)T);
        tbounds[0] = 0;
        size_t doubles = (size_t)std::max(nbt, 1) * kRecDoubles;
        const int W_ = comm->world;
        if (tile_TI > 0 && W_ > 1 && !full) {
            const int tpr = (nbt + W_ - 1) / W_;
            const long long last_rows = (long long)T - (long long)(W_ - 1) * tpr * tile_TI;
            if ((long long)(W_ - 1) * tpr < nbt && (long long)tpr * tile_TI >= midx && last_rows >= std::max(midx, 1)) {
                ts_possible = true;
                const size_t edge_bytes = (size_t)2 * kEdgeVecs * midx * KP * sizeof(real);
                tsh_rank.rank = comm->rank; tsh_rank.world = W_; tsh_rank.tpr = tpr;
                tsh_rank.tile0 = comm->rank * tpr; tsh_rank.ntiles = std::min(nbt, (comm->rank + 1) * tpr) - tsh_rank.tile0;
                tsh_rank.row_b = tsh_rank.tile0 * tile_TI; tsh_rank.row_e = std::min(T, (tsh_rank.tile0 + tsh_rank.ntiles) * tile_TI);
                tsh_rank.edge_off_dbl = (unsigned)tpr * kRecDoubles;
                tsh_rank.slot_dbl = tsh_rank.edge_off_dbl + (unsigned)((edge_bytes + 15) / 16 * 2);
                for (int r = 1; r < W_; r++) tbounds[r] = (uint64_t)std::min<long long>(T, (long long)r * tpr * tile_TI);
                doubles = std::max(doubles, (size_t)W_ * tsh_rank.slot_dbl);
            }
        }
        release_p2p();
        for (int m = 0; m < 3; m++) { if (xmsg_own[m].alloc(doubles)) return kFail; xm[m] = xmsg_own[m].p; }
        // peer-to-peer arena: required under TRMF_CG=p2p, otherwise tried (and silently dropped where it does not work) so
        // that the measure-once rule can consider it; never with TRMF_CG=timeshard|replicate or TRMF_NO_P2P
        const char *e = getenv("TRMF_CG");
        const bool ptables = nbt <= kPersistMaxTiles;             // the persistent kernel's tables ride in the same arena
        if (ts_possible && ((e && e[0] == 'p') || (!e && !getenv("TRMF_NO_P2P"))) &&
            setup_p2p(doubles, e != nullptr, ptables ? (size_t)2 * nbt * kLLWords * 8 : 0, ptables ? (size_t)2 * T * KP * 2 * sizeof(real) : 0)) return kFail;
        return 0;
    }
    // candidates of the X-solve's form, called once the geometry (tiles, uts) and the peer-to-peer arena are settled
    void init_x_forms() {
        x_cands.clear(); x_calls = 0; x_form = kXRep;
        for (int f = 0; f < kXForms; f++) { x_ms[f] = 0; x_ms_all[f] = 0; }
        const char *e = getenv("TRMF_CG");
        const bool fused_ts = tile_TI > 0 && ts_possible;
        if (!fused_ts && !uts) return;                                   // one rank, or nothing time-sharded: no choice
        if (fused_ts && e && e[0] == 'p' && e[1] == 'e') {              // "persist": one persistent kernel per rank
            if (!persist_usable_shard()) { x_form = kXTsP2p; persist_note = "TRMF_CG=persist: tiles not co-resident / tables missing, peer-to-peer launches instead"; return; }
            x_form = kXTsPersist; return;
        }
        if (e && e[0] == 'p') { x_form = kXTsP2p; return; }              // set-up succeeded, or create() has failed already
        if (e && e[0] == 't') { x_form = kXTsComm; return; }
        if (fused_ts && e && e[0] == 'r') { x_form = kXRep; return; }
        if (fused_ts) x_cands.push_back(kXRep);
        x_cands.push_back(kXTsComm);
        if (p2p.on) x_cands.push_back(kXTsP2p);
        // the persistent kernel across ranks: measured in the set-up iterations only (a trial that times out -- workgroups of several
        // ranks that share ONE device and do not fit together -- costs the iteration it ran in, which autotune() undoes)
        const char *at = getenv("TRMF_AUTOTUNE");
        if (fused_ts && p2p.on && !(at && atoi(at) == 0) && !test_env("TRMF_NO_PERSIST_SHARD") && persist_usable_shard()) {
            if (max_ranks_per_device == 1) x_cands.push_back(kXTsPersist);
            else persist_note = std::to_string(max_ranks_per_device) + " ranks share one device: the persistent-kernel form is not tried";
        }
        x_form = x_cands.size() == 1 ? x_cands[0] : -1;
    }
    static const char *x_form_name(int f) {
        return f == kXRep ? "replicated" : f == kXTsComm ? "time-sharded (communicator)" : f == kXTsP2p ? "time-sharded (peer to peer)"
             : f == kXTsPersist ? "time-sharded (one persistent kernel per rank, peer to peer)" : "measuring";
    }

    // ---- per-series affine transform of a resident dense Y (trmf_session_set_series_transform) -------------------------
    // rolling_validate(transform=True) -- the paper scripts' setting -- refits a NormalizedTransform on every growing
    // prefix (trmf.py:82-96, 237-249), which rescales EVERY entry of Y.  The session therefore keeps the raw matrix and
    // re-derives both training orientations from it on the device; only the 2n coefficients cross PCIe per window.
    DevBuf<real> Yraw;
    DevBuf<real> tr_a, tr_b;
    DevBuf<double> tr_part;
    bool has_transform = false;
    int set_series_transform(const real *a, const real *b) {
        if (!dense) { set_error("set_series_transform: needs a dense Y (missing == 0)"); return kFail; }
        FillStreamScope fill(stream);
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        if (Yraw.n != (size_t)T * n) {              // first call: the resident copy is still the raw matrix
            if (has_transform) { set_error("set_series_transform: raw matrix lost"); return kFail; }
            if (Yraw.alloc((size_t)T * n, false)) return kFail;
            TRMF_HIP_CHECK(hipMemcpyAsync(Yraw.p, Yd_tn.p, (size_t)T * n * sizeof(real), hipMemcpyDeviceToDevice, stream));
        }
        std::vector<real> one((size_t)n, real(1)), zero((size_t)n, real(0));
        if (tr_a.upload(a ? a : one.data(), n) || tr_b.upload(b ? b : zero.data(), n)) return kFail;
        has_transform = true;
        return apply_series_transform();
    }
    int apply_series_transform() {
        const int nb = 1024;
        if (tr_part.alloc(nb)) return kFail;
        hipLaunchKernelGGL(affine_columns_kernel, dim3(nb), dim3(256), 0, stream, Yraw.p, (size_t)T, n, tr_a.p, tr_b.p, Yd_tn.p, tr_part.p);
        launch_transpose(Yd_tn.p, T, n, Yd_nt.p);
        TRMF_HIP_CHECK(hipGetLastError());
        std::vector<double> part(nb);
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        TRMF_HIP_CHECK(hipMemcpy(part.data(), tr_part.p, nb * sizeof(double), hipMemcpyDeviceToHost));
        ysq_acc = 0;
        for (double v : part) ysq_acc += v;
        set_trYTY();
        return 0;
    }

    // ---- append new timestamps (trmf_session_append_rows; the rolling-window caller trmf.py:303-329) ---------------
    // Ynew: Tn x n block of NEW timestamps, same storage class as the session's Y.  Only that block (plus, for a
    // sparse Y, one 4-byte pointer per item) crosses PCIe: the CSR gains rows at its end, the CSC -- whose columns
    // each gain entries at their tails -- is rebuilt on the device from the old CSC and the block's CSC, a dense Y's
    // n x T copy is re-strided on the device, and W is extended by the AR recursion with the current Theta
    // (Model.latent_forecast, trmf.py:170-181, the reference's warm start :237-246).  H and Theta are kept.
    // The iteration counter restarts (a new train() call in the reference, trmf.cpp:647).
    int append_rows(const PyMatrix *Yn) {
        const int Tn = (int)Yn->rows, T0 = T;
        if ((int)Yn->cols != n) { set_error("append_rows: column count differs from the session's"); return kFail; }
        if ((Yn->type != TRMF_SPARSE) != dense) { set_error("append_rows: storage class (sparse/dense) differs from the session's"); return kFail; }
        if ((uint64_t)(T0 + Tn + 1) >= (1ull << 24) || (uint64_t)(T0 + Tn + 1) * KP * sizeof(real) > 0xffffffffull ||
            nnz + Yn->nnz >= (1ull << 32)) { set_error("append_rows: problem would exceed 32-bit device indices"); return kFail; }
        if (Tn <= 0) return 0;
        FillStreamScope fill(stream);
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        const int T1 = T0 + Tn;
        // Failure-atomic: everything new is built in locals (device buffers, host pointer arrays, sums) and swapped into
        // the session only after every allocation, copy and kernel of the step has succeeded; a failed call leaves the
        // session exactly as it was.
        std::vector<uint64_t> new_row_ptr, new_col_ptr;
        uint64_t new_nnz = nnz;
        double new_ysq = ysq_acc;
        DevBuf<uint32_t> ptr2, idx2, cptr2, cidx2; DevBuf<real> val2, cval2, tn2, nt2, raw2, W2;
        if (!dense) {
            const uint64_t nz0 = nnz, nzn = Yn->nnz, nz1 = nz0 + nzn;
            // CSR: old rows keep their place, the block's rows follow
            new_row_ptr = host_row_ptr;
            for (int i = 1; i <= Tn; i++) new_row_ptr.push_back(nz0 + Yn->row_ptr[i]);
            if (upload_ptr32(ptr2, (const size_t *)new_row_ptr.data(), (size_t)T1 + 1) || idx2.alloc(nz1, false) || val2.alloc(nz1, false)) return kFail;
            TRMF_HIP_CHECK(hipMemcpyAsync(idx2.p, Yr_idx.p, nz0 * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
            TRMF_HIP_CHECK(hipMemcpyAsync(val2.p, Yr_val.p, nz0 * sizeof(real), hipMemcpyDeviceToDevice, stream));
            if (nzn) {
                TRMF_HIP_CHECK(hipMemcpyAsync(idx2.p + nz0, Yn->col_idx, nzn * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                TRMF_HIP_CHECK(hipMemcpyAsync(val2.p + nz0, Yn->val_t, nzn * sizeof(real), hipMemcpyHostToDevice, stream));
            }
            // CSC: column j = its old entries, then the block's entries of that column (timestamps shifted by T0)
            new_col_ptr = host_col_ptr;
            std::vector<uint32_t> np32((size_t)n + 1);
            for (int j = 0; j <= n; j++) { new_col_ptr[j] += Yn->col_ptr[j]; np32[j] = (uint32_t)new_col_ptr[j]; }
            DevBuf<uint32_t> wptr, widx; DevBuf<real> wval;
            if (cptr2.upload(np32.data(), np32.size()) || cidx2.alloc(nz1, false) || cval2.alloc(nz1, false) ||
                upload_ptr32(wptr, Yn->col_ptr, (size_t)n + 1) || widx.upload(Yn->row_idx, nzn) || wval.upload((const real *)Yn->val, nzn))
                return kFail;
            hipLaunchKernelGGL(csc_append_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, Yc_ptr.p, Yc_idx.p, Yc_val.p, wptr.p, widx.p,
                               wval.p, cptr2.p, cidx2.p, cval2.p, n, (uint32_t)T0);
            TRMF_HIP_CHECK(hipGetLastError());
            TRMF_HIP_CHECK(hipStreamSynchronize(stream));
            new_nnz = nz1;
            if (device_sum_squares(val2.p, new_nnz, &new_ysq)) return kFail;
        } else {
            std::vector<real> blk;
            new_ysq += dense_rows_to_rowmajor(Yn, blk);
            if (tn2.alloc((size_t)T1 * n, false) || nt2.alloc((size_t)T1 * n, false)) return kFail;
            TRMF_HIP_CHECK(hipMemcpyAsync(tn2.p, Yd_tn.p, (size_t)T0 * n * sizeof(real), hipMemcpyDeviceToDevice, stream));
            TRMF_HIP_CHECK(hipMemcpyAsync(tn2.p + (size_t)T0 * n, blk.data(), blk.size() * sizeof(real), hipMemcpyHostToDevice, stream));
            if (has_transform) {                        // the block is RAW data: grow the raw copy, re-derive below
                if (raw2.alloc((size_t)T1 * n, false)) return kFail;
                TRMF_HIP_CHECK(hipMemcpyAsync(raw2.p, Yraw.p, (size_t)T0 * n * sizeof(real), hipMemcpyDeviceToDevice, stream));
                TRMF_HIP_CHECK(hipMemcpyAsync(raw2.p + (size_t)T0 * n, blk.data(), blk.size() * sizeof(real), hipMemcpyHostToDevice, stream));
            }
            launch_transpose(tn2.p, T1, n, nt2.p);
            TRMF_HIP_CHECK(hipGetLastError());
            TRMF_HIP_CHECK(hipStreamSynchronize(stream));      // blk is a local: its copies must have left the host
            new_nnz = (uint64_t)T1 * n;
        }
        {   // W: T0 rows kept, Tn rows rolled out by the AR model, one all-zero row at the end
            if (W2.alloc((size_t)(T1 + 1) * KP)) return kFail;
            TRMF_HIP_CHECK(hipMemcpyAsync(W2.p, W.p, (size_t)T0 * KP * sizeof(real), hipMemcpyDeviceToDevice, stream));
            hipLaunchKernelGGL(latent_forecast_kernel, dim3(1), dim3(64), 0, stream, W2.p, T0, T1, KP, NT, k, lag_set.p, nlag, theta.p);
            TRMF_HIP_CHECK(hipGetLastError());
            TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        }
        // ---- commit ----
        if (!dense) {
            Yr_ptr.swap(ptr2); Yr_idx.swap(idx2); Yr_val.swap(val2);
            Yc_ptr.swap(cptr2); Yc_idx.swap(cidx2); Yc_val.swap(cval2);
            host_row_ptr.swap(new_row_ptr); host_col_ptr.swap(new_col_ptr);
        } else {
            Yd_tn.swap(tn2); Yd_nt.swap(nt2);
            if (has_transform) Yraw.swap(raw2);
        }
        W.swap(W2);
        nnz = new_nnz; ysq_acc = new_ysq;
        T = T1;
        if (dense && has_transform && apply_series_transform()) return kFail;   // current coefficients over the grown raw matrix
        set_trYTY();
        iter = 0;
        if (gramx_mode != kGramxReplicate && comm->world > 1 && !test_env("TRMF_GRAMX")) { gramx_mode = kGramxMeasure; gramx_calls = 0; }
        if (alloc_time_scratch()) return kFail;
        TRMF_HIP_CHECK(hipDeviceSynchronize());
        return autotune();
    }

    // ---- all-gather helpers ------------------------------------------------------------------------
    int gather_rows(void *dbuf, const std::vector<uint64_t> &bounds, size_t row_bytes) {
        if (comm->world == 1 && !comm->call_when_single) return 0;
        std::vector<uint64_t> off(bounds.size());
        for (size_t i = 0; i < bounds.size(); i++) off[i] = bounds[i] * row_bytes;
        return comm->allgatherv(dbuf, off.data(), stream);
    }

    // ---- F-solve (trmf.cpp:654-663 -> 369-397) -------------------------------------------------------
    template <int NT_, int KMAX_> int launch_fsolve_mfma(uint32_t rb, uint32_t re) {
        const uint32_t rows = re - rb;
        if (rows == 0) return 0;
#if !defined(TRMF_F32)
        hipLaunchKernelGGL((fsolve_mfma_kernel<NT_, KMAX_>), dim3((rows + 3) / 4), dim3(256), 0, stream,
                           Yc_ptr.p, Yc_idx.p, Yc_val.p, W.p, H.p, rb, re, k, (real)lambdaI, (uint32_t)T);
#endif
        return 0;
    }
    template <int NT_, int KMAX_> int launch_fsolve_quad(uint32_t rb, uint32_t re) {
        const uint32_t rows = re - rb;
        if (rows == 0) return 0;
#if defined(TRMF_F32)
        const dim3 grid((rows + 15) / 16), block(256);
#define TRMF_LAUNCH_QUAD(ABL)                                                                          \
        hipLaunchKernelGGL((fsolve_quad_kernel<NT_, KMAX_, ABL>), grid, block, 0, stream, Yc_ptr.p,    \
                           Yc_idx.p, Yc_val.p, W.p, H.p, rb, re, k, (real)lambdaI, (uint32_t)T)
#if defined(TRMF_ABLATION)
        if (NT_ == 3 && KMAX_ == 40 && dbg_flags) {
            switch (dbg_flags) {
                case 1: TRMF_LAUNCH_QUAD(1); break;
                case 2: TRMF_LAUNCH_QUAD(2); break;
                case 4: TRMF_LAUNCH_QUAD(4); break;
                case 6: TRMF_LAUNCH_QUAD(6); break;
                default: TRMF_LAUNCH_QUAD(7); break;
            }
            return 0;
        }
#endif
        TRMF_LAUNCH_QUAD(0);
#undef TRMF_LAUNCH_QUAD
#endif
        return 0;
    }
    // fp32: four systems per wavefront (fsolve_quad_kernel); fp64: one system per wavefront, factorised in the MFMA
    // accumulator layout (fsolve_mfma_kernel)
    // X-side Gram build across ranks: sharded rows + all-gather of G (64 MB at config 3) pays only when a
    // rank's share of the gather is cheaper than the rows it no longer computes -- true on 8 GPUs, not on 2.
    // First call measures (kernel and gather time of every rank, exchanged through the communicator so that
    // all ranks take the same decision); TRMF_GRAMX=shard|replicate overrides.
    enum { kGramxMeasure = 0, kGramxShard = 1, kGramxReplicate = 2 };
    int gramx_mode = kGramxMeasure, gramx_calls = 0;
    hipEvent_t gx0 = nullptr, gx1 = nullptr, gx2 = nullptr;
    DevBuf<double> gramx_times;
    int dbg_flags = 0;           // TRMF_DEBUG_ABLATE: bit0 skip Gram, bit1 skip factorisation, bit2 skip back-solve
    int launch_fsolve_rows(uint32_t rb, uint32_t re) {
        if (generic) {
            if (re > rb)
                hipLaunchKernelGGL(gram_generic_kernel<true>, dim3(std::min<uint32_t>(kGenBlocks, re - rb)), dim3(256), gram_generic_lds(k), stream,
                                   Yc_ptr.p, Yc_idx.p, Yc_val.p, W.p, rb, re, k, KP, NT, (real)lambdaI, gen_scratch.p, (size_t)0, H.p);
            return 0;
        }
#define TRMF_FSOLVE_SWITCH(FN)                                                       \
        switch (KMAX) {                                                              \
            case 8:  FN<1, 8>(rb, re); break;                                        \
            case 16: FN<1, 16>(rb, re); break;                                       \
            case 24: FN<2, 24>(rb, re); break;                                       \
            case 32: FN<2, 32>(rb, re); break;                                       \
            case 40: FN<3, 40>(rb, re); break;                                       \
            case 48: FN<3, 48>(rb, re); break;                                       \
            case 56: FN<4, 56>(rb, re); break;                                       \
            case 64: FN<4, 64>(rb, re); break;                                       \
            default: set_error("unsupported rank"); return kFail;                    \
        }
        if (sizeof(real) == 4) { TRMF_FSOLVE_SWITCH(launch_fsolve_quad) }
        else { TRMF_FSOLVE_SWITCH(launch_fsolve_mfma) }
#undef TRMF_FSOLVE_SWITCH
        return 0;
    }
    // Sharding a phase over the ranks pays only when the all-gather of its result costs less than the rows a rank no
    // longer computes (true for the F-solve at config 3 on 4 and 8 GPUs, not on 2).  Measure-once rule, shared by the
    // F-solve and the X-side Gram build: the first two calls run sharded, the second is timed on every rank (kernel, gather); the
    // times are exchanged through the communicator and every rank takes the same decision.
    enum { kShardMeasure = 0, kShardOn = 1, kShardOff = 2 };
    int decide_shard(hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, const char *what) {
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        float tk = 0, tg = 0;
        TRMF_HIP_CHECK(hipEventElapsedTime(&tk, e0, e1));
        TRMF_HIP_CHECK(hipEventElapsedTime(&tg, e1, e2));
        const double mine[2] = {(double)tk, (double)tg};
        TRMF_HIP_CHECK(hipMemcpy(gramx_times.p + 2 * comm->rank, mine, sizeof mine, hipMemcpyHostToDevice));
        std::vector<uint64_t> off(comm->world + 1);
        for (int r = 0; r <= comm->world; r++) off[r] = (uint64_t)r * sizeof mine;
        if (comm->allgatherv(gramx_times.p, off.data(), stream)) return -1;
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<double> all((size_t)2 * comm->world);
        TRMF_HIP_CHECK(hipMemcpy(all.data(), gramx_times.p, all.size() * sizeof(double), hipMemcpyDeviceToHost));
        double t_all_rows = 0, t_sharded = 0;
        for (int r = 0; r < comm->world; r++) {
            t_all_rows += all[2 * r];                                        // one GPU doing every rank's rows
            t_sharded = std::max(t_sharded, all[2 * r] + all[2 * r + 1]);    // slowest rank: its rows + the gather
        }
        const int mode = (t_all_rows < 0.95 * t_sharded) ? kShardOff : kShardOn;
        if (verbose && comm->rank == 0)
            fprintf(stderr, ">> %s: all rows %.3f ms vs sharded %.3f ms -> %s\n", what, t_all_rows, t_sharded,
                    mode == kShardOff ? "replicated" : "sharded");
        return mode;
    }
    int fs_mode = kShardMeasure, fs_calls = 0;
    hipEvent_t fs0 = nullptr, fs1 = nullptr, fs2 = nullptr;
    // Overlapped all-gather of H (large item factors: config 5's is 512 MB): the rank's rows are solved in C launches of
    // equal nnz; chunk c of every rank's block is gathered on a side stream while launch c + 1 runs, the last chunk follows on
    // the solver stream, which then waits for the side stream -- only the last chunk's gather is exposed.  C = 2..4 by size
    // (one chunk per 16 MB of the rank's block); below kOverlapBytes per rank the extra launches' tails cost more than the
    // gather they hide (config 4).  TRMF_FOVERLAP=0 switches it off, =2..4 forces that many chunks at any size.
    static constexpr uint64_t kOverlapBytes = 16ull << 20;
    static constexpr int kMaxChunks = 4;
    hipStream_t side = nullptr;
    hipEvent_t ov_b = nullptr, ov_c[kMaxChunks] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<uint64_t> fcut;               // (world x (chunks + 1)) rows: chunk c of rank r = [fcut[r*(C+1)+c], fcut[r*(C+1)+c+1])
    int fchunks = 0;
    // The chunk count must be the SAME on every rank (each chunk is one collective): it is derived from the LARGEST block of
    // the partition, a number every rank computes from the same bounds -- not from the rank's own row count, which differs
    // between the ranks of an nnz-balanced partition (near 16 / 48 / 64 MiB the ranks disagreed).
    int overlap_chunks() {
        if (comm->world <= 1 || full || host_col_ptr.empty()) return 0;
        if (const char *e = test_env("TRMF_FOVERLAP")) { const int c = atoi(e); return c <= 0 ? 0 : std::max(2, std::min(kMaxChunks, c)); }
        uint64_t rows = 0;
        for (int r = 0; r < comm->world; r++) rows = std::max<uint64_t>(rows, fbounds[r + 1] - fbounds[r]);
        uint64_t thresh = kOverlapBytes;
        if (const char *e = test_env("TRMF_FOVERLAP_BYTES")) thresh = std::max<uint64_t>(1, strtoull(e, nullptr, 10));   // tests: the threshold at small sizes
        const uint64_t bytes = rows * KP * sizeof(real);
        return bytes >= thresh ? (int)std::max<uint64_t>(2, std::min<uint64_t>(kMaxChunks, bytes / thresh)) : 0;
    }
    int fsolve(PhaseEvents &ev) {
        // the SECOND call is the measured one: the first carries one-time costs on both sides of the comparison (code
        // object load of the kernel, connection set-up inside the first collective)
        if (fs_mode == kShardMeasure && fs_calls == 2) {
            const int m = decide_shard(fs0, fs1, fs2, "F-solve");
            if (m < 0) return kFail;
            fs_mode = m;
        }
        const bool replicate = fs_mode == kShardOff, measure = fs_mode == kShardMeasure && fs_calls == 1;
        const uint32_t rb = replicate ? 0u : (uint32_t)fbounds[comm->rank];
        const uint32_t re = replicate ? (uint32_t)n : (uint32_t)fbounds[comm->rank + 1];
        const int C = fs_mode == kShardOn ? overlap_chunks() : 0;
        if (C >= 2) {
            const int W_ = comm->world;
            if (!side) {
                TRMF_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
                TRMF_HIP_CHECK(hipEventCreateWithFlags(&ov_b, hipEventDisableTiming));
                for (hipEvent_t &e : ov_c) TRMF_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            if (fchunks != C || fcut.size() != (size_t)W_ * (C + 1)) {
                fchunks = C; fcut.resize((size_t)W_ * (C + 1));
                for (int r = 0; r < W_; r++) {
                    const uint64_t n0 = host_col_ptr[fbounds[r]], n1 = host_col_ptr[fbounds[r + 1]];
                    fcut[(size_t)r * (C + 1)] = fbounds[r]; fcut[(size_t)r * (C + 1) + C] = fbounds[r + 1];
                    for (int c = 1; c < C; c++)
                        fcut[(size_t)r * (C + 1) + c] = (uint64_t)(std::lower_bound(host_col_ptr.begin() + fbounds[r], host_col_ptr.begin() + fbounds[r + 1],
                                                                                   n0 + (n1 - n0) * c / C) - host_col_ptr.begin());
                }
            }
            const uint64_t rowbytes = (uint64_t)KP * sizeof(real);
            const uint64_t *mine = fcut.data() + (size_t)comm->rank * (C + 1);
            TRMF_HIP_CHECK(hipEventRecord(ev.fk0, stream));
            for (int c = 0; c < C; c++) {
                if (launch_fsolve_rows((uint32_t)mine[c], (uint32_t)mine[c + 1])) return kFail;
                if (c + 1 < C) TRMF_HIP_CHECK(hipEventRecord(ov_c[c], stream));
            }
            TRMF_HIP_CHECK(hipEventRecord(ev.fk1, stream));
            TRMF_HIP_CHECK(hipGetLastError());
            fs_calls++;
            std::vector<uint64_t> gb(W_), ge(W_);
            for (int c = 0; c < C; c++) {                      // chunk c: on the side stream under launch c + 1; the last one on the solver stream
                for (int r = 0; r < W_; r++) { gb[r] = fcut[(size_t)r * (C + 1) + c] * rowbytes; ge[r] = fcut[(size_t)r * (C + 1) + c + 1] * rowbytes; }
                if (c + 1 < C) {
                    TRMF_HIP_CHECK(hipStreamWaitEvent(side, ov_c[c], 0));
                    if (comm->allgatherv_ranges(H.p, gb.data(), ge.data(), side)) return kFail;
                } else {
                    TRMF_HIP_CHECK(hipEventRecord(ov_b, side));
                    if (comm->allgatherv_ranges(H.p, gb.data(), ge.data(), stream)) return kFail;
                    TRMF_HIP_CHECK(hipStreamWaitEvent(stream, ov_b, 0));
                }
            }
            return 0;
        }
        if (measure) TRMF_HIP_CHECK(hipEventRecord(fs0, stream));
        TRMF_HIP_CHECK(hipEventRecord(ev.fk0, stream));
        if (launch_fsolve_rows(rb, re)) return kFail;
        TRMF_HIP_CHECK(hipEventRecord(ev.fk1, stream));
        TRMF_HIP_CHECK(hipGetLastError());
        fs_calls++;
        if (replicate) return 0;                                    // every rank solved every row: nothing to gather
        if (measure) TRMF_HIP_CHECK(hipEventRecord(fs1, stream));
        if (gather_rows(H.p, fbounds, (size_t)KP * sizeof(real))) return kFail;
        if (measure) TRMF_HIP_CHECK(hipEventRecord(fs2, stream));
        return 0;
    }

    // ---- X-side Gram cache / loss ---------------------------------------------------------------------
    template <int NT_> void launch_gram_x(uint32_t rb, uint32_t re) {
        if (re <= rb) return;
        const dim3 grid((re - rb + 3) / 4), block(256);
#define TRMF_LAUNCH_GRAM_X(PAD, PACKED)                                                                                      \
    hipLaunchKernelGGL((gram_x_kernel<NT_, PAD, PACKED>), grid, block, 0, stream, Yr_ptr.p, Yr_idx.p, Yr_val.p, H.p, G.p, Bv.p, \
                       rb, re, k, (uint32_t)n, xp.gstride)
        if (rhs_pad_ok<NT_>(k)) {     // rhs accumulated by the MFMAs in the panel's pad columns
            if (gpacked) TRMF_LAUNCH_GRAM_X(true, true); else TRMF_LAUNCH_GRAM_X(true, false);
        } else {
            if (gpacked) TRMF_LAUNCH_GRAM_X(false, true); else TRMF_LAUNCH_GRAM_X(false, false);
        }
#undef TRMF_LAUNCH_GRAM_X
    }
    void launch_gram_x_rows(uint32_t rb, uint32_t re) {
        if (generic) {
            if (re > rb)
                hipLaunchKernelGGL(gram_generic_kernel<false>, dim3(std::min<uint32_t>(4096, re - rb)), dim3(256), gram_generic_lds(k), stream,
                                   Yr_ptr.p, Yr_idx.p, Yr_val.p, H.p, rb, re, k, KP, NT, real(0), G.p, xp.gstride, Bv.p);
            return;
        }
        switch (NT) {
            case 1: launch_gram_x<1>(rb, re); break;
            case 2: launch_gram_x<2>(rb, re); break;
            case 3: launch_gram_x<3>(rb, re); break;
            default: launch_gram_x<4>(rb, re); break;
        }
    }
    template <int NT_> void launch_loss(const real *Wv, uint32_t rb, uint32_t re) {
        if (re > rb)
            hipLaunchKernelGGL((loss_kernel<NT_>), dim3(re - rb), dim3(256), 0, stream, Yr_ptr.p, Yr_idx.p,
                               Yr_val.p, H.p, Wv, lossrow.p, rb, re, (uint32_t)n);
    }
    int gram_x(bool timeshard = false) {
        if (timeshard || uts) {     // time-sharded CG: a rank only ever reads the Grams / right-hand sides of its own timestamps
            const uint32_t rb = (uint32_t)(uts ? ush.row_b : tsh_rank.row_b), re = (uint32_t)(uts ? ush.row_e : tsh_rank.row_e);
            launch_gram_x_rows(rb, re);
            TRMF_HIP_CHECK(hipGetLastError());
            return 0;
        }
        if (!cg_shard && gramx_mode == kGramxMeasure && gramx_calls == 2 && gramx_decide()) return kFail;   // second call measured, see fsolve()
        const bool replicate = gramx_mode == kGramxReplicate && !cg_shard;
        const bool measure = gramx_mode == kGramxMeasure && !cg_shard && gramx_calls == 1;
        const uint32_t rb = replicate ? 0u : (uint32_t)xbounds[comm->rank];
        const uint32_t re = replicate ? (uint32_t)T : (uint32_t)xbounds[comm->rank + 1];
        if (measure) TRMF_HIP_CHECK(hipEventRecord(gx0, stream));
        launch_gram_x_rows(rb, re);
        TRMF_HIP_CHECK(hipGetLastError());
        gramx_calls++;
        if (replicate) return 0;                                    // every rank built every row: nothing to gather
        if (cg_shard) return 0;                                     // sharded Gram product: a rank only ever reads its own G / b rows
        if (measure) TRMF_HIP_CHECK(hipEventRecord(gx1, stream));
        if (gather_rows(G.p, xbounds, xp.gstride * sizeof(real))) return kFail;
        if (gather_rows(Bv.p, xbounds, (size_t)KP * sizeof(real))) return kFail;
        if (measure) TRMF_HIP_CHECK(hipEventRecord(gx2, stream));
        return 0;
    }
    // One-time decision after the second (measured, sharded) build.  Rank r publishes (kernel ms, gather ms);
    // after the exchange every rank evaluates the same rule on the same numbers.
    int gramx_decide() {
        const int m = decide_shard(gx0, gx1, gx2, "X-side Gram build");
        if (m < 0) return kFail;
        gramx_mode = m == kShardOff ? kGramxReplicate : kGramxShard;
        return 0;
    }
    int loss(const real *Wv, bool all_rows) {
        const uint32_t rb = all_rows ? 0u : (uint32_t)xbounds[comm->rank];
        const uint32_t re = all_rows ? (uint32_t)T : (uint32_t)xbounds[comm->rank + 1];
        if (generic) {
            if (re > rb) hipLaunchKernelGGL(loss_generic_kernel, dim3(re - rb), dim3(256), 0, stream, Yr_ptr.p, Yr_idx.p, Yr_val.p, H.p, Wv, lossrow.p, rb, re, KP);
        } else
        switch (NT) {
            case 1: launch_loss<1>(Wv, rb, re); break;
            case 2: launch_loss<2>(Wv, rb, re); break;
            case 3: launch_loss<3>(Wv, rb, re); break;
            default: launch_loss<4>(Wv, rb, re); break;
        }
        TRMF_HIP_CHECK(hipGetLastError());
        return all_rows ? 0 : gather_rows(lossrow.p, xbounds, sizeof(double));
    }

    // ---- full-observation path (missing == 0): trmf.cpp:299-351 and 155-215 -----------------------------
    template <int NT_> void launch_spmm(const uint32_t *ptr, const uint32_t *idx, const real *val, const real *X,
                                        real *out, uint32_t rb, uint32_t re, uint32_t zero_row) {
        if (re > rb)
            hipLaunchKernelGGL((spmm_rows_kernel<NT_>), dim3((re - rb + 3) / 4), dim3(256), 0, stream, ptr, idx, val, X,
                               out, rb, re, zero_row);
    }
    template <int NT_> void launch_dense_tn(const real *A, int K, int M, const real *B, real *out) {
        // contraction chunks: enough workgroups (64 output rows each) to fill the chip even when there are only a few
        // hundred output rows (Y^T W of a tall series matrix), at least 64 contracted rows per chunk, within the
        // partial buffer (kGemmChunks * max(n,T) rows)
        const int xb = (M + 63) / 64;
        const long long cap = (long long)kGemmChunks * std::max(n, T) / std::max(M, 1);
        const int nchunk = (int)std::max<long long>(1, std::min<long long>({cap, (long long)std::max(1, K / 64), (1024 + xb - 1) / xb}));
        hipLaunchKernelGGL((dense_tn_mfma_kernel<NT_>), dim3(xb, nchunk), dim3(256), 0, stream, A, K, M, B, gemm_part.p);
        if (nchunk <= 16)       // few chunks: a thread per output; many (tall contraction, few outputs): a wavefront per output
            hipLaunchKernelGGL(dense_tn_reduce_flat_kernel, dim3((unsigned)(((size_t)M * KP + 255) / 256)), dim3(256), 0, stream,
                               gemm_part.p, nchunk, M, KP, NT, k, out);
        else
            hipLaunchKernelGGL(dense_tn_reduce_kernel, dim3((unsigned)(((size_t)M * KP + 3) / 4)), dim3(256), 0, stream,
                               gemm_part.p, nchunk, M, KP, NT, k, out);
    }
    int y_times_factor(bool transposed, const real *X, real *out, uint32_t rb, uint32_t re) {
        if (generic) {
            if (!dense) {
                if (re > rb)
                    hipLaunchKernelGGL(spmm_generic_kernel, dim3(std::min<uint32_t>(4096, re - rb)), dim3(256), 0, stream, transposed ? Yc_ptr.p : Yr_ptr.p,
                                       transposed ? Yc_idx.p : Yr_idx.p, transposed ? Yc_val.p : Yr_val.p, X, out, rb, re, k, KP, NT);
            } else {
                const int K = transposed ? T : n, M = transposed ? n : T;
                hipLaunchKernelGGL(dense_tn_generic_kernel, dim3(std::min(4096, std::max(1, M))), dim3(256), 0, stream, transposed ? Yd_tn.p : Yd_nt.p, K, M, X, out, k, KP, NT);
            }
            TRMF_HIP_CHECK(hipGetLastError());
            return 0;
        }
        if (!dense) {
            const uint32_t *ptr = transposed ? Yc_ptr.p : Yr_ptr.p, *idx = transposed ? Yc_idx.p : Yr_idx.p;
            const real *val = transposed ? Yc_val.p : Yr_val.p;
            const uint32_t zr = (uint32_t)(transposed ? T : n);
            switch (NT) {
                case 1: launch_spmm<1>(ptr, idx, val, X, out, rb, re, zr); break;
                case 2: launch_spmm<2>(ptr, idx, val, X, out, rb, re, zr); break;
                case 3: launch_spmm<3>(ptr, idx, val, X, out, rb, re, zr); break;
                default: launch_spmm<4>(ptr, idx, val, X, out, rb, re, zr); break;
            }
        } else {
            const real *A = transposed ? Yd_tn.p : Yd_nt.p;     // K x M row-major with K the contracted dim
            const int K = transposed ? T : n, M = transposed ? n : T;
            switch (NT) {
                case 1: launch_dense_tn<1>(A, K, M, X, out); break;
                case 2: launch_dense_tn<2>(A, K, M, X, out); break;
                case 3: launch_dense_tn<3>(A, K, M, X, out); break;
                default: launch_dense_tn<4>(A, K, M, X, out); break;
            }
        }
        TRMF_HIP_CHECK(hipGetLastError());
        return 0;
    }
    template <int NT_> void launch_small_gram(const real *A, int rows, int nb) {
        hipLaunchKernelGGL((small_gram_mfma_kernel<NT_>), dim3(nb), dim3(256), 0, stream, A, rows, k, sgram_part.p);
    }
    int small_gram(const real *A, int rows, real lambda, real *GS) {
        if (generic) {
            hipLaunchKernelGGL(small_gram_generic_kernel, dim3(k), dim3(256), 0, stream, A, rows, k, KP, NT, lambda, GS);
            return 0;
        }
        // one partial per wavefront (4 per workgroup), at least 64 rows each, kSmallGramBlocks slots in all
        const int nb = std::max(1, std::min(kSmallGramBlocks / 4, rows / 256));
        switch (NT) {
            case 1: launch_small_gram<1>(A, rows, nb); break;
            case 2: launch_small_gram<2>(A, rows, nb); break;
            case 3: launch_small_gram<3>(A, rows, nb); break;
            default: launch_small_gram<4>(A, rows, nb); break;
        }
        hipLaunchKernelGGL(small_gram_reduce_kernel, dim3((k * k + 3) / 4), dim3(256), 0, stream, sgram_part.p, nb * 4, k, lambda, GS);
        return 0;
    }
    int fsolve_full(PhaseEvents &ev) {
        const uint32_t rb = (uint32_t)fbounds[comm->rank], re = (uint32_t)fbounds[comm->rank + 1];
        TRMF_HIP_CHECK(hipEventRecord(ev.fk0, stream));
        if (y_times_factor(true, W.p, Bf.p, dense ? 0u : rb, dense ? (uint32_t)n : re)) return kFail;   // Y^T W
        small_gram(W.p, T, (real)lambdaI, GSf.p);                                                       // W^T W + lambda I
        if (re > rb && generic) {
            hipLaunchKernelGGL(chol_generic_kernel, dim3(1), dim3(256), 0, stream, GSf.p, Uf.p, k);
            const int nrows = (int)(re - rb);
            hipLaunchKernelGGL(solve_rows_generic_kernel, dim3(std::min(2048, nrows)), dim3(256), (size_t)k * sizeof(real), stream, Uf.p, Bf.p + (size_t)rb * KP,
                               H.p + (size_t)rb * KP, nrows, k, KP, NT);
        } else if (re > rb) {
            const size_t ulds = (size_t)k * k * sizeof(real);          // <= 32 KB
            if (test_env("TRMF_CHOL_WORKGROUP")) hipLaunchKernelGGL(chol_shared_kernel, dim3(1), dim3(256), ulds, stream, GSf.p, Uf.p, k);
            else switch (NT) {
                case 1: hipLaunchKernelGGL(chol_wave_kernel<1>, dim3(1), dim3(64), 0, stream, GSf.p, Uf.p, k); break;
                case 2: hipLaunchKernelGGL(chol_wave_kernel<2>, dim3(1), dim3(64), 0, stream, GSf.p, Uf.p, k); break;
                case 3: hipLaunchKernelGGL(chol_wave_kernel<3>, dim3(1), dim3(64), 0, stream, GSf.p, Uf.p, k); break;
                default: hipLaunchKernelGGL(chol_wave_kernel<4>, dim3(1), dim3(64), 0, stream, GSf.p, Uf.p, k); break;
            }
            const int nrows = (int)(re - rb), nblk = std::max(1, std::min(2048, (nrows + 3) / 4));
            hipLaunchKernelGGL(solve_rows_kernel, dim3(nblk), dim3(256), ulds, stream, Uf.p, Bf.p + (size_t)rb * KP,
                               H.p + (size_t)rb * KP, nrows, k, KP, NT);
        }
        TRMF_HIP_CHECK(hipEventRecord(ev.fk1, stream));
        TRMF_HIP_CHECK(hipGetLastError());
        return gather_rows(H.p, fbounds, (size_t)KP * sizeof(real));
    }
    int xprepare_full() {        // init() of arr_ls_fY_IX, trmf.cpp:183-187
        const uint32_t rb = (uint32_t)xbounds[comm->rank], re = (uint32_t)xbounds[comm->rank + 1];
        if (y_times_factor(false, H.p, Bv.p, dense ? 0u : rb, dense ? (uint32_t)T : re)) return kFail;  // Y H
        if (!dense && gather_rows(Bv.p, xbounds, (size_t)KP * sizeof(real))) return kFail;
        small_gram(H.p, n, real(0), GSx.p);                                                             // H^T H
        return 0;
    }

    // ---- X-solve (trmf.cpp:665-674 -> rf_tron.h:134-254) -----------------------------------------------
    // Fused path (the AR halo fits LDS): hv_tile_kernel in its four roles, one launch per CG iteration.
    template <int MODE, bool SHARD> void launch_hv_tile_as(const HvVecs &a, int it, int last, const double *rec_in, double *rec_out) {
        const size_t lds = hv_tile_lds_bytes(tile_TI, midx, KP, nlag, k);
        const TileShard &sh = SHARD ? tsh_rank : tsh;
        const PeerTable *pt = (SHARD && p2p_use) ? peer_table.p : nullptr;
        const int mi = rec_out == xm[0] ? 0 : rec_out == xm[1] ? 1 : 2;
#define TRMF_LAUNCH_HV_KQ(KQ)                                                                                        \
        hipLaunchKernelGGL((hv_tile_kernel<MODE, KQ, SHARD>), dim3(sh.ntiles), dim3(256), lds, stream, xp, xstate.p, a, sh, it, last,  \
                           lag_set.p, theta.p, Gmat(), rec_in, rec_out, pt, mi, tile_TI)
        switch (hv_kq(k) / 8) {
            case 1: TRMF_LAUNCH_HV_KQ(8); break;
            case 2: TRMF_LAUNCH_HV_KQ(16); break;
            case 3: TRMF_LAUNCH_HV_KQ(24); break;
            case 4: TRMF_LAUNCH_HV_KQ(32); break;
            case 5: TRMF_LAUNCH_HV_KQ(40); break;
            case 6: TRMF_LAUNCH_HV_KQ(48); break;
            case 7: TRMF_LAUNCH_HV_KQ(56); break;
            default: TRMF_LAUNCH_HV_KQ(64); break;
        }
#undef TRMF_LAUNCH_HV_KQ
    }
    template <int MODE> void launch_hv_tile(bool shard, const HvVecs &a, int it, int last, const double *rec_in, double *rec_out) {
        if (shard) launch_hv_tile_as<MODE, true>(a, it, last, rec_in, rec_out);
        else launch_hv_tile_as<MODE, false>(a, it, last, rec_in, rec_out);
    }
    // time-sharded CG: exchange the slots of a message (tile records + edge rows of every rank), then copy the
    // neighbours' edge rows of up to three vectors to their natural rows of the local vectors
    // one exchange of message `mi` after a launch (`it`: the CG launch index, -1 for the gradient / plain launch) and the
    // unpacking of the neighbours' edge rows of nvec vectors: through the communicator (in-place all-gather of the slots +
    // halo_unpack_kernel) or peer to peer (the launch wrote into the peers' arenas; xchg_sync_kernel raises / awaits the flags)
    // Peer to peer, before the first launch of a solve that writes into the peers' messages: wait until every peer has
    // finished the previous solve.  Inside a solve a rank is never more than one exchange ahead of a peer and consecutive
    // exchanges use different records / arrays; across the solve boundary nothing else orders the ranks once the F-solve
    // is replicated (no all-gather between two X-solves), and the gradient launch of the next solve would overwrite sums
    // the slower rank's acceptance test has yet to read (seen as ranks disagreeing on |g| and on the CG's stop: timeouts).
    void p2p_fence(const TileShard &sh) {
        hipLaunchKernelGGL(xchg_sync_kernel, dim3(1), dim3(256), 0, stream, peer_table.p, 1, ++p2p.epoch[1], xstate.p, -1, sh, 0, KP, 0,
                           (real *)nullptr, (real *)nullptr, (real *)nullptr, kP2pTimeoutTicks);
    }
    int exchange(int mi, int it, int nvec, real *v0, real *v1, real *v2) {
        const int edgeN = midx * KP;
        if (p2p_use) {
            hipLaunchKernelGGL(xchg_sync_kernel, dim3(1), dim3(256), 0, stream, peer_table.p, mi, ++p2p.epoch[mi], xstate.p, it, tsh_rank,
                               edgeN, KP, nvec, v0, v1, v2, kP2pTimeoutTicks);
            return 0;
        }
        if (comm->allgather_slots(xm[mi], (size_t)tsh_rank.slot_dbl * sizeof(double), stream)) return kFail;
        if (nvec > 0 && edgeN > 0)
            hipLaunchKernelGGL(halo_unpack_kernel, dim3(std::max(1, std::min(8, (edgeN + 255) / 256))), dim3(256), 0, stream, xm[mi],
                               tsh_rank, edgeN, KP, nvec, v0, v1, v2);
        return 0;
    }
    // The fused X-solve: gradient launch, CG launches (one per iteration, the closing one also forms w_new and the sums
    // of the acceptance test), plain launch H s, accept.  shard: every launch runs this rank's tiles only and is
    // followed by the exchange of its message; the host then follows the CG's progress (the stop is detected on the
    // device) so that no exchange is issued for an iteration that will not run: it enqueues as many iterations as the
    // previous solve needed, reads XState::stop_it back, and goes on two at a time.  All ranks derive identical scalars
    // from identical records, so they take identical decisions (the collectives match).
    int xsolve_fused(bool shard, int maxcg, XState *log_x, double *log_n) {
        real *dbuf[2] = {d0.p, d1.p}, *rbuf[2] = {r.p, r1.p}, *hbuf[2] = {Hd.p, Hd1.p};
        double *mg = xm[2], *mc[2] = {xm[0], xm[1]};
        HvVecs a{};
        a.v = W.p; a.out = g.p; a.Bv = Bv.p;
        launch_hv_tile<HV_GRAD>(shard, a, 0, 0, nullptr, mg);                  // gradient, <g,g>, AR/ridge sums
        if (shard && exchange(2, -1, 1, g.p, nullptr, nullptr)) return kFail;
        a = HvVecs{};
        a.v = g.p; a.s = s.p; a.d_out = dbuf[0]; a.r_out = rbuf[0]; a.out = hbuf[0];
        launch_hv_tile<HV_CG_FIRST>(shard, a, 0, 0, mg, mc[0]);                // f, |g|, cgtol; s = 0, r = d = -g; H d
        if (shard && exchange(0, 0, 3, dbuf[0], rbuf[0], hbuf[0])) return kFail;
        // host-followed progress only where an exchange costs a collective; peer to peer (and on one rank) the launches of
        // iterations that will not run are no-ops on the device and everything is enqueued at once
        const bool follow = shard && !p2p_use;
        int upto = follow ? std::min(maxcg, std::max(1, cg_pred)) : maxcg;
        for (int it = 1; it <= maxcg; it++) {                              // launch `maxcg` only closes the last iteration
            a.v = dbuf[(it - 1) & 1]; a.r_in = rbuf[(it - 1) & 1]; a.hd_in = hbuf[(it - 1) & 1];
            a.d_out = dbuf[it & 1]; a.r_out = rbuf[it & 1]; a.out = hbuf[it & 1];
            launch_hv_tile<HV_CG_STEP>(shard, a, it, it == maxcg ? 1 : 0, mc[(it - 1) & 1], mc[it & 1]);
            if (!shard) continue;
            if (exchange(it & 1, it, 3, dbuf[it & 1], rbuf[it & 1], hbuf[it & 1])) return kFail;
            if (!follow) continue;
            if (it == upto && it < maxcg) {                                // has the CG stopped?  (identical on every rank)
                int stop = kCgRunning;
                TRMF_HIP_CHECK(hipMemcpyAsync(&stop, &xstate.p->stop_it, sizeof(int), hipMemcpyDeviceToHost, stream));
                TRMF_HIP_CHECK(hipStreamSynchronize(stream));
                if (stop != kCgRunning) { cg_pred = stop; break; }
                upto = std::min(maxcg, upto + 2);
            } else if (it == maxcg) cg_pred = maxcg;
        }
        // close the last completed iteration: s, w_new = w + s, the sums of the acceptance test (+ the edge rows of s)
        const TileShard &sh = shard ? tsh_rank : tsh;
        const PeerTable *pt = (shard && p2p_use) ? peer_table.p : nullptr;
        if (shard)
            hipLaunchKernelGGL(cg_close_kernel<true>, dim3(sh.ntiles), dim3(256), 0, stream, xp, xstate.p, sh, tile_TI, mc[0], mc[1], dbuf[0],
                               dbuf[1], rbuf[0], rbuf[1], hbuf[0], hbuf[1], s.p, g.p, W.p, w_new.p, mg, pt);
        else
            hipLaunchKernelGGL(cg_close_kernel<false>, dim3(sh.ntiles), dim3(256), 0, stream, xp, xstate.p, sh, tile_TI, mc[0], mc[1], dbuf[0],
                               dbuf[1], rbuf[0], rbuf[1], hbuf[0], hbuf[1], s.p, g.p, W.p, w_new.p, mg, pt);
        if (shard && exchange(2, -1, 1, s.p, nullptr, nullptr)) return kFail;   // halo rows of s
        a = HvVecs{};
        a.v = s.p; a.out = hbuf[0];
        launch_hv_tile<HV_PLAIN>(shard, a, 0, 0, nullptr, mg);                 // H s, <s,Hs> (fields [0..2] of the same records)
        if (shard && exchange(2, -1, 0, nullptr, nullptr, nullptr)) return kFail;
        const int nb = (int)std::min<size_t>(kMaxPartials, ((size_t)(sh.row_e - sh.row_b) * KP + 255) / 256);
        hipLaunchKernelGGL(accept_tile_kernel, dim3(std::max(nb, 1)), dim3(256), 0, stream, xp, xstate.p, mg, sh,
                           shard ? 1 : 0, w_new.p, W.p, log_x, log_n);
        TRMF_HIP_CHECK(hipGetLastError());
        if (shard && gather_rows(W.p, tbounds, (size_t)KP * sizeof(real))) return kFail;   // the F-solve gathers rows of all of W
        return 0;
    }
    // ---- the X-solve as ONE persistent kernel (cg_persist.hpp) ---------------------------------------------------------------
    // One rank per GPU and the GPU to itself (world == 1): every tile's workgroup stays resident for the whole solve.  Needs all
    // workgroups co-resident (checked against the occupancy the runtime reports) and the LDS of the resident
    // vectors; otherwise -- or with TRMF_PERSIST=0 -- the launch-per-step path runs.  Bit-identical results either way.
    DevBuf<unsigned long long> ll_rec, ll_vec;   // tagged records / tagged vector rows (zero = never a valid tag)
    DevBuf<long long> persist_prof;           // -DTRMF_PERSIST_PROF builds: phase stamps of the last solve (printed by sync())
    uint32_t persist_epoch = 1;
    int persist_state = 0;                    // 0: not examined yet, 1: usable, -1: not
    template <int KQ, bool SHARD = false> int persist_prepare(size_t lds) {
        const void *fn = reinterpret_cast<const void *>(&cg_persist_kernel<KQ, SHARD>);
        if (lds > kLdsMax) return 0;
        if (lds > kLdsDefault && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        // (256-thread blocks are admitted per CU up to min(API answer, 8, 800 / (ceil(sgprs / 16) * 16 + 16)), same guide: 6 at this
        // kernel's ~106 SGPRs -- the register-bound answer of 2..3 is always the smaller one; capped anyway)
        return std::min(per_cu, 4) * prop.multiProcessorCount;
    }
    template <int KQ, bool SHARD = false> int persist_launch(const PersistArgs &pa, size_t lds) {
        // a plain launch: the grid was checked against the occupancy in persist_prepare(); hipLaunchCooperativeKernel gives the same
        // residency for 15-19 us more host time per launch (MI355X guide, "coop-launch")
        hipLaunchKernelGGL((cg_persist_kernel<KQ, SHARD>), dim3(SHARD ? tsh_rank.ntiles : nbt), dim3(256), lds, stream, xp, xstate.p, pa);
        TRMF_HIP_CHECK(hipGetLastError());
        return 0;
    }
#define TRMF_PERSIST_SWITCH(CALL)                  \
        switch (hv_kq(k) / 8) {                    \
            case 1: CALL(8); break;                \
            case 2: CALL(16); break;               \
            case 3: CALL(24); break;               \
            case 4: CALL(32); break;               \
            case 5: CALL(40); break;               \
            case 6: CALL(48); break;               \
            case 7: CALL(56); break;               \
            default: CALL(64); break;              \
        }
    bool persist_usable(int maxcg) {
        if (persist_state == 0) {
            persist_state = -1;
            const char *e = getenv("TRMF_PERSIST");
            const size_t lds = persist_lds_bytes(tile_TI, midx, KP, nlag, k, nbt);
            // one rank, or several ranks each with a device of its own running the REPLICATED CG (every rank all tiles, no interaction
            // between the ranks' kernels); never where ranks share a device (their workgroups would have to be co-resident)
            if ((comm->world == 1 || max_ranks_per_device == 1) && tile_TI > 0 && nbt <= kPersistMaxTiles && maxcg <= kCgHistCap && !(e && atoi(e) == 0)) {
                int slots = 0;
#define TRMF_PERSIST_PREP(KQV) slots = persist_prepare<KQV>(lds)
                TRMF_PERSIST_SWITCH(TRMF_PERSIST_PREP)
#undef TRMF_PERSIST_PREP
                if (slots >= nbt && ll_rec.alloc((size_t)2 * nbt * kLLWords) == 0 &&
                    ll_vec.alloc((size_t)2 * T * KP * (2 * sizeof(real) / sizeof(unsigned long long))) == 0) persist_state = 1;
            }
        }
        return persist_state == 1;
    }
    // How many ranks drive the device that hosts the most of them (1 on a real node: one process per GPU).  Persistent kernels of
    // several processes on ONE device only make progress while all of them are scheduled at once.  Measured with processes standing
    // in for GPUs (profiles/r04_persist_notes.txt): 2 processes fine; 4 fine while their other kernels are short, but time-sliced to
    // ~7 s per solve when every rank also runs the full F-solve (a 30 s poll bound lets it finish: slow progress, no lost data); 8
    // processes ~30 s per solve; even 2 processes occasionally miss the 2 s bound (it won the measurement and then timed out in a later
    // solve of the full-size test).  Where ranks SHARE a device the measure-once rule therefore leaves that form out (TRMF_CG=persist
    // still forces it); with a device per rank every poll stays bounded (2 s) and a trial that times out only loses the candidate.
    int max_ranks_per_device = 1;
    int count_ranks_per_device() {
        int dev = 0;
        hipDeviceProp_t prop;
        TRMF_HIP_CHECK(hipGetDevice(&dev));
        TRMF_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        const int W_ = comm->world;
        long long mine[8] = {prop.pciDomainID, prop.pciBusID, prop.pciDeviceID, 0, 0, 0, 0, 0};
        DevBuf<long long> ids;
        if (ids.alloc((size_t)8 * W_)) return kFail;
        TRMF_HIP_CHECK(hipMemcpyAsync(ids.p + 8 * comm->rank, mine, sizeof mine, hipMemcpyHostToDevice, stream));
        if (comm->allgather_slots(ids.p, sizeof mine, stream)) return kFail;
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<long long> all((size_t)8 * W_);
        TRMF_HIP_CHECK(hipMemcpy(all.data(), ids.p, all.size() * sizeof(long long), hipMemcpyDeviceToHost));
        max_ranks_per_device = 1;
        for (int a = 0; a < W_; a++) {
            int same = 0;
            for (int b = 0; b < W_; b++) same += all[8 * a] == all[8 * b] && all[8 * a + 1] == all[8 * b + 1] && all[8 * a + 2] == all[8 * b + 2];
            max_ranks_per_device = std::max(max_ranks_per_device, same);
        }
        return 0;
    }
    // several ranks: every rank's tiles in one persistent kernel, tables in the IPC arenas (cg_persist.hpp, SHARD)
    int persist_shard_state = 0;
    bool persist_failed = false;
    std::string persist_note;
    bool persist_usable_shard() {
        if (persist_shard_state == 0) {
            persist_shard_state = -1;
            const int maxcg = (int)std::min<long long>(max_cg_iter, (long long)T * k);
            if (p2p.on && p2p.ext_bytes > 0 && tile_TI > 0 && ts_possible && nbt <= kPersistMaxTiles && maxcg <= kCgHistCap && !full) {
                const size_t lds = persist_lds_bytes(tile_TI, midx, KP, nlag, k, nbt);
                int slots = 0;
#define TRMF_PERSIST_PREP_S(KQV) slots = persist_prepare<KQV, true>(lds)
                TRMF_PERSIST_SWITCH(TRMF_PERSIST_PREP_S)
#undef TRMF_PERSIST_PREP_S
                // against the LARGEST block of the partition (the last rank may own fewer tiles): every rank must reach the same
                // answer, or the ranks' candidate lists -- and with them the collectives of decide_x_form() -- differ
                if (slots >= tsh_rank.tpr) persist_shard_state = 1;
            }
        }
        return persist_shard_state == 1;
    }
    int xsolve_persist(int maxcg, XState *log_x, double *log_n, bool shard = false) {
        PersistArgs pa{};
        pa.W = W.p; pa.Bv = Bv.p; pa.G = Gmat(); pa.lag_set = lag_set.p; pa.theta = theta.p;
        pa.hll = ll_vec.p; pa.ll = ll_rec.p;
        if (shard) {
            pa.sh = tsh_rank;
            unsigned char *own = (unsigned char *)p2p.arena + p2p.ext_off;
            pa.ll = reinterpret_cast<unsigned long long *>(own);
            pa.hll = reinterpret_cast<unsigned long long *>(own + p2p.ext_ll_bytes);
            for (int r = 0; r < comm->world; r++) {
                unsigned char *pr = r == comm->rank ? nullptr : (unsigned char *)p2p.peer[r] + p2p.ext_off;
                pa.peer_ll[r] = reinterpret_cast<unsigned long long *>(pr);
                pa.peer_hll[r] = pr ? reinterpret_cast<unsigned long long *>(pr + p2p.ext_ll_bytes) : nullptr;
            }
        }
        pa.timeout_ticks = kPersistTimeoutTicks;
        if (const char *e = getenv("TRMF_PERSIST_TIMEOUT_MS")) pa.timeout_ticks = std::max(1ll, atoll(e)) * 100000ll;
        pa.epoch0 = persist_epoch; pa.TI = tile_TI; pa.maxcg = maxcg; pa.log_x = log_x; pa.log_n = log_n;
        pa.fail_tile = -1; pa.fail_x = -1;
        if (const char *e = test_env("TRMF_PERSIST_FAIL")) {           // "<tile>:<exchange>" (exchange -2: the final one)
            pa.fail_tile = atoi(e);
            if (const char *c = strchr(e, ':')) pa.fail_x = atoi(c + 1);
        }
        persist_epoch += (uint32_t)maxcg + 8;
#if defined(TRMF_PERSIST_PROF)
        if (test_env("TRMF_PERSIST_PROF")) {
            if (!persist_prof.p && persist_prof.alloc((size_t)2 * kProfIters * kProfSlots + 2 * (size_t)kPersistMaxTiles)) return kFail;
            pa.prof = persist_prof.p;
        }
#endif
        const size_t lds = persist_lds_bytes(tile_TI, midx, KP, nlag, k, nbt);
#define TRMF_PERSIST_GO(KQV) if (shard ? persist_launch<KQV, true>(pa, lds) : persist_launch<KQV, false>(pa, lds)) return kFail
        TRMF_PERSIST_SWITCH(TRMF_PERSIST_GO)
#undef TRMF_PERSIST_GO
        // the F-solve gathers rows of all of W -- and this collective is what keeps a fast rank's next solve out of the record
        // slots a slow rank is still polling (cg_persist.hpp, "Across launches")
        if (shard && gather_rows(W.p, tbounds, (size_t)KP * sizeof(real))) return kFail;
        return 0;
    }
#undef TRMF_PERSIST_SWITCH

    // Which form of the X-solve?  Measured once, like the other shard decisions: each candidate of init_x_forms() runs two X
    // phases, the second timed; the times of every rank are exchanged and the slowest rank's time decides.  All forms of the
    // fused path give bit-identical iterates (the unfused transports likewise among themselves), so switching between
    // iterations is free.  The measurement starts only once the X-side Gram build has taken its own decision (ADVICE r3:
    // timing the replicated form while the Gram build was still in ITS measuring mode biased the comparison).
    int decide_x_form() {
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        double mine[8] = {(double)x_ms[0], (double)x_ms[1], (double)x_ms[2], (double)x_ms[3], 0, 0, 0, 0};
        TRMF_HIP_CHECK(hipMemcpy(gramx_times.p + 8 * comm->rank, mine, sizeof mine, hipMemcpyHostToDevice));
        std::vector<uint64_t> off(comm->world + 1);
        for (int r = 0; r <= comm->world; r++) off[r] = (uint64_t)r * sizeof mine;
        if (comm->allgatherv(gramx_times.p, off.data(), stream)) return kFail;
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<double> all((size_t)8 * comm->world);
        TRMF_HIP_CHECK(hipMemcpy(all.data(), gramx_times.p, all.size() * sizeof(double), hipMemcpyDeviceToHost));
        int best = x_cands[0];
        for (int f : x_cands) {
            x_ms_all[f] = 0;
            for (int r = 0; r < comm->world; r++) x_ms_all[f] = std::max(x_ms_all[f], all[8 * r + f]);
            if (x_ms_all[f] < x_ms_all[best]) best = f;
        }
        x_form = best;
        if (verbose && comm->rank == 0) {
            fprintf(stderr, ">> X-solve:");
            for (int f : x_cands) fprintf(stderr, " %s %.3f ms;", x_form_name(f), x_ms_all[f]);
            fprintf(stderr, " -> %s\n", x_form_name(x_form));
        }
        return 0;
    }
    // Multi-GPU, unfused path: shard the cached-Gram product of every CG step.  It pays when the
    // rows a rank no longer streams (T k^2 s (1 - 1/N) bytes at ~4 TB/s) outweigh an all-gather of T KP s bytes per
    // step (latency ~40 us + bytes over the rank's xGMI links); the fused one-launch-per-step path is faster
    // replicated at the sizes it covers.  TRMF_CG=shard|replicate overrides.
    bool cg_shard = false;
    static constexpr int kShardSlots = 4096;
    int apply_slots = 1;         // partial-sum slots (= workgroups of apply_kernel) per rank when sharded
    // Time-sharded UNFUSED CG (round 3): like the fused path's, a rank owns a contiguous block of AR
    // tiles -- its timestamps -- and runs every kernel of the solve on that block only: ar_tile_kernel (vector updates, AR
    // operator), apply_kernel (cached-Gram product), the element-wise kernels.  Per step the ranks exchange the midx first /
    // last rows of d, r and H d (edge_pack_kernel -> one all-gather of equal slots -> halo_unpack_kernel) and their slots of
    // the partial-sum arrays, in one grouped round; nothing T-sized is gathered (the sharded Gram product above gathers the
    // rows of H d, T KP values, every step).  The host follows the CG's stop as in the fused path.
    bool uts = false;
    TileShard ush{};                          // rank / world / rows / edge-slot geometry (no records: the unfused kernels keep arrays)
    std::vector<uint64_t> ubounds;            // AR-tile-aligned timestamp partition
    DevBuf<double> umsg;                      // edge message: world slots of 2 sides x 3 vectors x midx rows
    double *umsg_ptr = nullptr;               // = umsg.p (communicator transport)
    unsigned u_exchanges = 0;                 // peer to peer: exchanges issued so far (selects the edge message, 0 or 2)
    int u_tile0 = 0, u_ntiles = 0, u_tpr = 0, wn_slots = 1;
    int decide_cg_shard() {
        cg_shard = false; uts = false;
        apply_slots = std::max(1, std::min(nba, kShardSlots / std::max(1, comm->world)));
        const int W_ = comm->world;
        if (W_ <= 1 || tile_TI > 0 || full) return 0;
        const char *e = getenv("TRMF_CG");
        const int tiles = (T + ar_TI - 1) / ar_TI, tpr = (tiles + W_ - 1) / W_;
        const long long last_rows = (long long)T - (long long)(W_ - 1) * tpr * ar_TI;
        const bool can_uts = (long long)(W_ - 1) * tpr < tiles && (long long)tpr * ar_TI >= midx && last_rows >= std::max(midx, 1) &&
                             (long long)tiles * (KP / kArCols) <= xp.pstride;
        if (can_uts && !(e && (e[0] == 's' || e[0] == 'r'))) {
            uts = true;
            u_tpr = tpr; u_tile0 = comm->rank * tpr; u_ntiles = std::min(tiles, (comm->rank + 1) * tpr) - u_tile0;
            ubounds.assign(W_ + 1, (uint64_t)T);
            for (int r = 0; r < W_; r++) ubounds[r] = (uint64_t)std::min<long long>(T, (long long)r * tpr * ar_TI);
            ush = TileShard{};
            ush.rank = comm->rank; ush.world = W_; ush.row_b = (int)ubounds[comm->rank]; ush.row_e = (int)ubounds[comm->rank + 1];
            const size_t edge_bytes = (size_t)2 * kEdgeVecs * midx * KP * sizeof(real);
            ush.edge_off_dbl = 0; ush.slot_dbl = (unsigned)((edge_bytes + 15) / 16 * 2);
            wn_slots = std::max(1, std::min(nbe, kMaxPartials / W_));
            if (umsg.alloc((size_t)W_ * std::max(1u, ush.slot_dbl))) return kFail;
            umsg_ptr = umsg.p;
            // peer to peer: the edge messages and the partial-sum arrays live in the IPC-exported arena -- required under
            // TRMF_CG=p2p, otherwise tried so that the measure-once rule can consider it (init_x_forms)
            if ((e && e[0] == 'p') || (!e && !getenv("TRMF_NO_P2P"))) {
                release_p2p();
                if (setup_p2p(std::max((size_t)W_ * ush.slot_dbl, (size_t)P_NSLOTS * xp.pstride), e != nullptr)) return kFail;
                u_exchanges = 0;
            }
            return 0;
        }
        const double N = W_, sz = sizeof(real);
        const double t_saved = (double)T * (double)xp.gstride * sz * (1.0 - 1.0 / N) / 4e12;
        const double t_gather = 40e-6 + (double)T * KP * sz * (1.0 - 1.0 / N) / ((N - 1.0) * 50e9);
        cg_shard = t_saved > 2.0 * t_gather;
        if (e && (e[0] == 's' || e[0] == 'r')) cg_shard = (e[0] == 's');
        return 0;
    }
    // one grouped exchange of the time-sharded unfused CG: edge rows of nvec vectors + this rank's slots of partial arrays
    // (kind 0: apply_kernel's slots, 1: ar_tile_kernel's, 2: wnew_kernel's)
    struct PartialRef { int slot, kind; };
    int uts_exchange(int it, int nvec, real *v0, real *v1, real *v2, std::initializer_list<PartialRef> arrays) {
        const int W_ = comm->world, edgeN = midx * KP;
        const bool edges = nvec > 0 && edgeN > 0;
        if (p2p_use) {
            PushList pl{};
            const int tiles = (T + ar_TI - 1) / ar_TI, groups = KP / kArCols;
            for (const PartialRef &a : arrays) {
                const int b = a.kind == 0 ? comm->rank * apply_slots : a.kind == 1 ? u_tile0 * groups : comm->rank * wn_slots;
                const int c = a.kind == 0 ? apply_slots : a.kind == 1 ? (std::min(tiles, u_tile0 + u_ntiles) - u_tile0) * groups : wn_slots;
                pl.slot[pl.n] = a.slot; pl.begin[pl.n] = b; pl.count[pl.n] = c; pl.n++;
            }
            const int mi = (u_exchanges++ & 1) ? 2 : 0;      // edge messages alternate (identical count on every rank: all enqueue alike)
            hipLaunchKernelGGL(uts_push_kernel, dim3(8), dim3(256), 0, stream, peer_table.p, mi, xstate.p, it, ush, xp.pstride, pl, edgeN, KP,
                               edges ? nvec : 0, v0, v1, v2);
            hipLaunchKernelGGL(xchg_sync_kernel, dim3(1), dim3(256), 0, stream, peer_table.p, mi, ++p2p.epoch[mi], xstate.p, it, ush, edgeN, KP,
                               edges ? nvec : 0, v0, v1, v2, kP2pTimeoutTicks);
            TRMF_HIP_CHECK(hipGetLastError());
            return 0;
        }
        if (edges)
            hipLaunchKernelGGL(edge_pack_kernel, dim3(std::max(1, std::min(8, (edgeN + 255) / 256))), dim3(256), 0, stream, umsg_ptr, ush, edgeN, KP,
                               nvec, v0, v1, v2);
        std::vector<uint64_t> off[3];
        for (int kind = 0; kind < 3; kind++) {
            off[kind].resize(W_ + 1);
            for (int r = 0; r <= W_; r++) {
                const uint64_t slots = kind == 0 ? (uint64_t)r * apply_slots
                                     : kind == 1 ? (uint64_t)std::min<long long>((long long)r * u_tpr, (T + ar_TI - 1) / ar_TI) * (KP / kArCols)
                                                 : (uint64_t)r * wn_slots;
                off[kind][r] = slots * sizeof(double);
            }
        }
        if (comm->group_begin()) return kFail;
        int rc = edges ? comm->allgather_slots(umsg_ptr, (size_t)ush.slot_dbl * sizeof(double), stream) : 0;
        for (const PartialRef &a : arrays)
            if (rc == 0) rc = comm->allgatherv(P(a.slot), off[a.kind].data(), stream);
        if (comm->group_end()) return kFail;
        if (rc) return rc;
        if (edges)
            hipLaunchKernelGGL(halo_unpack_kernel, dim3(std::max(1, std::min(8, (edgeN + 255) / 256))), dim3(256), 0, stream, umsg_ptr, ush, edgeN,
                               KP, nvec, v0, v1, v2);
        return 0;
    }
    // Unfused path (long lag sets): one operator application = ar_tile_kernel (AR + ridge part -> arbase) followed by
    // apply_kernel (+ cached-Gram product, dot-product partials).
    //   cg_it < 0: plain product of `av.v` (gradient at w when minus_b, H s)
    //   cg_it = 0: first CG product H d0 (d = av.v, residual rvec)
    //   cg_it >= 1: the whole CG iteration (ar_tile_kernel<AR_CG_STEP> closes iteration cg_it-1 and forms the new
    //               direction av.d_out / residual av.r_out, apply_kernel multiplies it)
    int hv(const ArVecs &av, int cg_it, int last, int minus_b, real *out, int dot_mode) {
        XState *st = xstate.p;
        double *Pb = pbase();
        const dim3 ar_grid(uts ? u_ntiles : (T + ar_TI - 1) / ar_TI, KP / kArCols);
        const int ar_tile0 = uts ? u_tile0 : 0;
        const size_t ar_lds = ar_tile_lds_bytes(ar_TI, midx, nlag);
        const int ndot = (cg_shard || uts) ? comm->world * apply_slots : nba;     // entries of the apply partial arrays
        if (cg_it >= 1)
            hipLaunchKernelGGL((ar_tile_kernel<AR_CG_STEP>), ar_grid, dim3(kArThreads), ar_lds, stream, xp, st, av, ndot, cg_it, last,
                               lag_set.p, lag_steps.p, nsteps, theta.p, arbase.p, Pb, ar_TI, ar_tile0);
        else
            hipLaunchKernelGGL((ar_tile_kernel<AR_PLAIN>), ar_grid, dim3(kArThreads), ar_lds, stream, xp, st, av, ndot, 0, 0,
                               lag_set.p, lag_steps.p, nsteps, theta.p, arbase.p, Pb, ar_TI, ar_tile0);
        if (last) return 0;                                              // the closing launch has no product
        const real *operand = cg_it >= 1 ? av.d_out : av.v;
        const real *resid = cg_it >= 1 ? av.r_out : av.r_in;
        // shared Gram, or the packed Grams of one row group, staged per workgroup
        const size_t ap_lds = full ? (size_t)k * k * sizeof(real) : gpacked ? (size_t)apply_stages(k) * 512 * sizeof(real) : 0;
        auto launch_apply = [&](int blocks, int row_b, int rows, int slot_b) {
            if (k > kApplyThreadPerColumn || (generic && full)) {   // very wide ranks (a workgroup per timestamp walks the columns); also the shared
                                                                    // Gram of the full-observation path above rank 64 (read from L2, not staged in LDS)
                hipLaunchKernelGGL(apply_wide_kernel, dim3(blocks), dim3(256), (size_t)k * sizeof(real), stream, xp, st, cg_it, operand, resid, arbase.p,
                                   Gmat(), Bv.p, minus_b, out, dot_mode, P(P_DOT), row_b, rows, slot_b);
                return;
            }
#define TRMF_LAUNCH_APPLY_PACKED(NS)                                                                                          \
    hipLaunchKernelGGL((apply_kernel<true, NS>), dim3(blocks), dim3(256), ap_lds, stream, xp, st, cg_it, operand, resid, arbase.p, \
                       Gmat(), Bv.p, minus_b, out, dot_mode, P(P_DOT), rpb, row_b, rows, slot_b)
            if (gpacked) {
                switch (apply_stages(k)) {
                    case 5: TRMF_LAUNCH_APPLY_PACKED(5); break;
                    case 10: TRMF_LAUNCH_APPLY_PACKED(10); break;
                    default: TRMF_LAUNCH_APPLY_PACKED(17); break;
                }
            } else
#undef TRMF_LAUNCH_APPLY_PACKED
                hipLaunchKernelGGL(apply_kernel<false>, dim3(blocks), dim3(256), ap_lds, stream, xp, st, cg_it, operand, resid, arbase.p,
                                   Gmat(), Bv.p, minus_b, out, dot_mode, P(P_DOT), rpb, row_b, rows, slot_b);
        };
        if (full && !generic && !test_env("TRMF_NO_APPLY_SHARED")) {       // one Gram for every timestamp: the product runs on the matrix pipe (never sharded)
#define TRMF_LAUNCH_APPLY_SHARED(NTV)                                                                                           \
    hipLaunchKernelGGL((apply_shared_mfma_kernel<NTV>), dim3(nba), dim3(256), apply_shared_lds_bytes(KP), stream, xp, st, cg_it,  \
                       operand, resid, arbase.p, Gmat(), Bv.p, minus_b, out, dot_mode, P(P_DOT), 0, T, 0)
            switch (NT) {
                case 1: TRMF_LAUNCH_APPLY_SHARED(1); break;
                case 2: TRMF_LAUNCH_APPLY_SHARED(2); break;
                case 3: TRMF_LAUNCH_APPLY_SHARED(3); break;
                default: TRMF_LAUNCH_APPLY_SHARED(4); break;
            }
#undef TRMF_LAUNCH_APPLY_SHARED
            return 0;
        }
        if (uts) {
            // this rank's timestamps only; then one grouped exchange: the edge rows the next kernel stages as halo and the
            // rank's slots of the partial sums
            launch_apply(apply_slots, ush.row_b, ush.row_e - ush.row_b, comm->rank * apply_slots);
            TRMF_HIP_CHECK(hipGetLastError());
            const int c0 = P_CG0 + 3 * (cg_it & 1);
            if (cg_it >= 1) return uts_exchange(cg_it, 3, av.d_out, av.r_out, out, {{c0, 0}, {c0 + 1, 0}, {c0 + 2, 0}});
            if (cg_it == 0) return uts_exchange(0, 1, out, nullptr, nullptr, {{c0, 0}, {c0 + 1, 0}, {c0 + 2, 0}});
            if (minus_b) return uts_exchange(-1, 1, out, nullptr, nullptr, {{P_DOT, 0}, {P_LQ, 0}, {P_AR, 1}, {P_VV, 1}});   // gradient
            return uts_exchange(-1, 0, nullptr, nullptr, nullptr, {{P_DOT, 0}});                                              // H s
        }
        if (!cg_shard) {
            launch_apply(nba, 0, T, 0);
            return 0;
        }
        // Gram product on this rank's timestamps only; its rows of `out` and its slots of the partial sums are
        // all-gathered (one grouped round), so every rank continues with identical vectors and scalars
        const int rb = (int)xbounds[comm->rank], re = (int)xbounds[comm->rank + 1];
        launch_apply(apply_slots, rb, re - rb, comm->rank * apply_slots);
        TRMF_HIP_CHECK(hipGetLastError());
        std::vector<uint64_t> poff(comm->world + 1);
        for (int r = 0; r <= comm->world; r++) poff[r] = (uint64_t)r * apply_slots * sizeof(double);
        if (comm->group_begin()) return kFail;
        int rc = gather_rows(out, xbounds, (size_t)KP * sizeof(real));
        if (cg_it >= 0) {
            for (int a3 = 0; a3 < 3 && rc == 0; a3++) rc = comm->allgatherv(P(P_CG0 + 3 * (cg_it & 1) + a3), poff.data(), stream);
        } else {
            if (rc == 0) rc = comm->allgatherv(P(P_DOT), poff.data(), stream);
            if (rc == 0 && minus_b) rc = comm->allgatherv(P(P_LQ), poff.data(), stream);
        }
        if (comm->group_end()) return kFail;
        return rc;
    }
    const real *Gmat() const { return full ? GSx.p : G.p; }      // shared H^T H or the per-timestamp cache

    hipEvent_t xg1_event = nullptr;           // this iteration's PhaseEvents::xg1 (set by run())
    int xsolve(XState *log_x = nullptr, double *log_n = nullptr) {   // log_*: record written by the accept kernel
        XState *st = xstate.p;
        const int maxcg = (int)std::min<long long>(max_cg_iter, (long long)T * k);   // trmf.cpp:523-526
        const bool fused = tile_TI > 0 && maxcg <= kCgHistCap;
        bool timed = false;
        int form = x_form;
        const bool choice = (fused && ts_possible) || (!fused && uts);
        if (choice && form < 0) {
            // the replicated form's Gram build takes its own measure-once decision first (second call measured)
            if (fused && !cg_shard && gramx_mode == kGramxMeasure && gramx_calls == 2 && gramx_decide()) return kFail;
            if (fused && gramx_mode == kGramxMeasure && !cg_shard) form = x_cands[0];
            else if (x_calls == 2 * (int)x_cands.size()) { if (decide_x_form()) return kFail; form = x_form; }
            else { form = x_cands[x_calls / 2]; timed = (x_calls & 1) != 0; x_calls++; }
        }
        if (!choice) form = kXRep;
        const bool shard = fused && form != kXRep;
        const int timed_form = form;
        const bool measuring = choice && x_form < 0;
        select_transport(form == kXTsP2p);
        if (timed) TRMF_HIP_CHECK(hipEventRecord(ts0, stream));
        if (full) {
            if (xprepare_full()) return kFail;                                 // b = Y H, shared Gram H^T H
        } else {
            if (gram_x(shard)) return kFail;                                   // G, b
        }
        if (xg1_event) TRMF_HIP_CHECK(hipEventRecord(xg1_event, stream));
        if (p2p_use && (fused ? shard : uts)) p2p_fence(fused ? tsh_rank : ush);
        auto end_timed = [&]() -> int {
            if (!timed) return 0;
            TRMF_HIP_CHECK(hipEventRecord(ts1, stream));
            TRMF_HIP_CHECK(hipStreamSynchronize(stream));
            TRMF_HIP_CHECK(hipEventElapsedTime(&x_ms[timed_form], ts0, ts1));
            return 0;
        };
        if (fused) {
            if (!shard && persist_usable(maxcg)) { if (xsolve_persist(maxcg, log_x, log_n)) return kFail; }
            else if (form == kXTsPersist) {
                if (xsolve_persist(maxcg, log_x, log_n, true)) return kFail;
                if (measuring) {
                    // a trial that timed out (ranks sharing one device whose workgroups do not fit together) must not fail the session:
                    // the candidate loses, the error flag is cleared, the set-up iterations' effect on the factors is undone anyway
                    TRMF_HIP_CHECK(hipStreamSynchronize(stream));
                    int err = 0;
                    TRMF_HIP_CHECK(hipMemcpy(&err, &xstate.p->p2p_error, sizeof(int), hipMemcpyDeviceToHost));
                    if (err) {
                        if (getenv("TRMF_P2P_VERBOSE")) {
                            XState hx;
                            TRMF_HIP_CHECK(hipMemcpy(&hx, xstate.p, sizeof hx, hipMemcpyDeviceToHost));
                            fprintf(stderr, "[persist trial] rank %d: error %d, exchange %lld, tile %lld, missing %lld (1 records, 2 halo rows) %lld (tiles %d..%d of %d)\n",
                                    comm->rank, err, hx.p2p_diag[0], hx.p2p_diag[1], hx.p2p_diag[2], hx.p2p_diag[3], tsh_rank.tile0, tsh_rank.tile0 + tsh_rank.ntiles, nbt);
                        }
                        TRMF_HIP_CHECK(hipMemset(&xstate.p->p2p_error, 0, sizeof(int)));
                        x_ms[kXTsPersist] = 1e9f; persist_failed = true;
                        persist_note = "the persistent kernel's trial timed out";
                    }
                }
            } else if (xsolve_fused(shard, maxcg, log_x, log_n)) return kFail;
            if (timed && form == kXTsPersist && persist_failed) { timed = false; TRMF_HIP_CHECK(hipEventRecord(ts1, stream)); }
            return end_timed();
        }
        real *dbuf[2] = {d0.p, d1.p}, *rbuf[2] = {r.p, r1.p}, *hbuf[2] = {Hd.p, Hd1.p};
        double *Pb = pbase();                                            // after select_transport(): the arena's arrays when peer to peer
        const int ndot = (cg_shard || uts) ? comm->world * apply_slots : nba;     // entries of the apply partial arrays
        // element ranges of the element-wise kernels: everything, or (time-sharded) this rank's timestamps -- cg_init_kernel
        // also covers the halo rows, whose gradient the exchange after the gradient product has delivered
        const size_t NV = (size_t)T * KP;
        const size_t own_b = uts ? (size_t)ush.row_b * KP : 0, own_e = uts ? (size_t)ush.row_e * KP : NV;
        const size_t halo_b = uts ? (size_t)std::max(0, ush.row_b - midx) * KP : 0, halo_e = uts ? (size_t)std::min(T, ush.row_e + midx) * KP : NV;
        const int nbw = uts ? wn_slots : nbe, npw = uts ? comm->world * wn_slots : nbe;
        ArVecs av{};
        av.v = W.p;
        if (hv(av, -1, 0, 1, g.p, 0)) return kFail;                      // gradient, <g,g>, AR/ridge sums
        hipLaunchKernelGGL(cg_init_kernel, dim3(nbe), dim3(256), 0, stream, xp, st, Pb, nbar, ndot, g.p,
                           s.p, rbuf[0], dbuf[0], halo_b, halo_e);       // f, |g|, cgtol, rho[0]; s = 0, r = d = -g
        av = ArVecs{};
        av.v = dbuf[0]; av.r_in = rbuf[0];
        if (hv(av, 0, 0, 0, hbuf[0], 1)) return kFail;                   // H d0 and its three dot products
        const bool follow_u = uts && !p2p_use;                            // peer to peer: everything is enqueued at once, as on one GPU
        int upto = follow_u ? std::min(maxcg, std::max(1, cg_pred)) : maxcg;
        for (int it = 1; it <= maxcg; it++) {                            // launch `maxcg` only closes the last iteration
            av.v = dbuf[(it - 1) & 1]; av.r_in = rbuf[(it - 1) & 1]; av.hd_in = hbuf[(it - 1) & 1];
            av.s = s.p; av.d_out = dbuf[it & 1]; av.r_out = rbuf[it & 1];
            if (hv(av, it, it == maxcg ? 1 : 0, 0, hbuf[it & 1], 1)) return kFail;
            if (!follow_u) continue;
            if (it == upto && it < maxcg) {                              // time-sharded: follow the stop (identical on every rank)
                int stop = kCgRunning;
                TRMF_HIP_CHECK(hipMemcpyAsync(&stop, &xstate.p->stop_it, sizeof(int), hipMemcpyDeviceToHost, stream));
                TRMF_HIP_CHECK(hipStreamSynchronize(stream));
                if (stop != kCgRunning) { cg_pred = stop; break; }
                upto = std::min(maxcg, upto + 2);
            } else if (it == maxcg) cg_pred = maxcg;
        }
        hipLaunchKernelGGL(wnew_kernel, dim3(nbw), dim3(256), 0, stream, xp, st, W.p, s.p, g.p, rbuf[0], rbuf[1], w_new.p, Pb, own_b, own_e,
                           uts ? comm->rank * wn_slots : 0);
        if (uts && uts_exchange(-1, 1, s.p, nullptr, nullptr, {{P_GS, 2}, {P_SR, 2}, {P_SS, 2}})) return kFail;
        av = ArVecs{};
        av.v = s.p;
        if (hv(av, -1, 0, 0, hbuf[0], 1)) return kFail;                  // H s, <s,Hs>
        hipLaunchKernelGGL(accept_kernel, dim3(nbw), dim3(256), 0, stream, xp, st, Pb, npw, ndot, (const double *)nullptr, w_new.p,
                           W.p, log_x, log_n, own_b, own_e);
        TRMF_HIP_CHECK(hipGetLastError());
        if (uts && gather_rows(W.p, ubounds, (size_t)KP * sizeof(real))) return kFail;   // the F-solve gathers rows of all of W
        return end_timed();
    }

    // Dynamic LDS above the 64 KB every launch may use needs an explicit opt-in per kernel (gfx950: up to 160 KB
    // per workgroup); anything larger is an unsupported problem, reported instead of a failed launch.
    static constexpr size_t kLdsDefault = 64 * 1024, kLdsMax = 160 * 1024;
    template <typename Fn> int allow_dyn_lds(Fn fn, size_t bytes, const char *what) {
        if (bytes <= kLdsDefault) return 0;
        if (bytes > kLdsMax) { set_error(std::string(what) + ": needs more than 160 KB of LDS per workgroup"); return kFail; }
        TRMF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        return 0;
    }

    // ---- Theta solve (trmf.cpp:677-689 -> 455-484) ------------------------------------------------------
    size_t theta_gram_lds() const { return theta_gram_lds_bytes(midx); }
    size_t theta_solve_lds() const { return (size_t)(nlag * nlag + nlag) * sizeof(real); }
    int theta_solve() {
        if (nlag == 0) return 0;
        const int nchunk = std::max(1, (T - midx + kThetaChunk - 1) / kThetaChunk);
        const int npairs = nlag * (nlag + 1) / 2 + nlag;
        const size_t lds1 = theta_gram_lds();
        hipLaunchKernelGGL(theta_gram_kernel, dim3(k, nchunk), dim3(256), lds1, stream, W.p, T, KP, lag_set.p,
                           nlag, midx, npairs, theta_part.p);
        const size_t lds2 = theta_scratch.p ? 0 : theta_solve_lds();
        hipLaunchKernelGGL(theta_solve_kernel, dim3(k), dim3(256), lds2, stream, theta_part.p, nchunk, nlag,
                           npairs, lambdaLag, theta.p, theta_scratch.p);
        TRMF_HIP_CHECK(hipGetLastError());
        return 0;
    }

    // ---- ||.||^2 into a log slot ----------------------------------------------------------------------------
    int log_norm(const real *v, size_t count, double *dst) {
        const int nb = (int)std::max<size_t>(1, std::min<size_t>(kMaxPartials, (count + 255) / 256));    // count == 0 (no lags): one block, sum 0
        hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, stream, v, count, P(P_DOT));
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(256), 0, stream, P(P_DOT), nb, dst);
        return 0;
    }

    double host_double(const double *dptr) {
        double v = 0;
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(&v, dptr, sizeof(double), hipMemcpyDeviceToHost);
        return v;
    }

    // ---- measure-once decisions, taken BEFORE the first iteration --------------------------------------------------------
    // With several ranks three things are decided by measurement (F rows sharded or not, X-side Gram rows sharded or not, the
    // form of the X-solve); each needs a couple of ordinary iterations and host synchronisations.  autotune() runs those
    // iterations right after the session is built (and after append_rows, which changes the geometry) on the real problem,
    // then puts W, H and Theta back and resets the iteration counter: the ALS loop proper never synchronises with the host,
    // and its timed window -- wherever a caller places it -- contains no measuring iterations.  Every form
    // computes the same iterates, so the decisions change speed only.  TRMF_AUTOTUNE=0 leaves the decisions to the first
    // iterations of run() as in round 3.
    static constexpr int kAutotuneMax = 14;
    int tuned_iters = 0;
    bool decisions_pending() const {
        if (comm->world <= 1 || full) return false;
        if (fs_mode == kShardMeasure && period_H > 0) return true;
        if (period_W <= 0) return false;
        if (x_form < 0) return true;
        const bool rep_gram = tile_TI > 0 ? x_form == kXRep : (!uts && !cg_shard);
        return rep_gram && !cg_shard && gramx_mode == kGramxMeasure;
    }
    int autotune() {
        tuned_iters = 0;
        if (const char *e = getenv("TRMF_AUTOTUNE")) if (atoi(e) == 0) return 0;
        if (!decisions_pending()) return 0;
        DevBuf<real> W0, H0, T0;
        const size_t nw = (size_t)(T + 1) * KP, nh = (size_t)(n + 1) * KP, nt = (size_t)nlag * k;
        if (W0.alloc(nw, false) || H0.alloc(nh, false) || T0.alloc(nt, false)) return kFail;
        TRMF_HIP_CHECK(hipMemcpyAsync(W0.p, W.p, nw * sizeof(real), hipMemcpyDeviceToDevice, stream));
        TRMF_HIP_CHECK(hipMemcpyAsync(H0.p, H.p, nh * sizeof(real), hipMemcpyDeviceToDevice, stream));
        if (nt) TRMF_HIP_CHECK(hipMemcpyAsync(T0.p, theta.p, nt * sizeof(real), hipMemcpyDeviceToDevice, stream));
        const int v = verbose, it0 = iter;
        const bool ln = log_norms;
        verbose = 0; log_norms = false;
        int rc = 0;
        for (; tuned_iters < kAutotuneMax && decisions_pending() && rc == 0; tuned_iters++) rc = run(1);
        verbose = v; log_norms = ln; iter = it0;
        if (rc) return kFail;
        TRMF_HIP_CHECK(hipMemcpyAsync(W.p, W0.p, nw * sizeof(real), hipMemcpyDeviceToDevice, stream));
        TRMF_HIP_CHECK(hipMemcpyAsync(H.p, H0.p, nh * sizeof(real), hipMemcpyDeviceToDevice, stream));
        if (nt) TRMF_HIP_CHECK(hipMemcpyAsync(theta.p, T0.p, nt * sizeof(real), hipMemcpyDeviceToDevice, stream));
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        if (sync()) return kFail;
        if (verbose && comm->rank == 0) fprintf(stderr, ">> %s\n", describe().c_str());
        return 0;
    }
    // what this session runs, in one line (trmf_session_describe; bench.py's config.parallelism)
    std::string describe() const {
        char buf[640];
        if (comm->world <= 1) {
            snprintf(buf, sizeof buf, "1 rank; X-solve %s", generic ? "unfused; generic kernels for rank > 64 (Gram build, F-solve)"
                     : tile_TI <= 0 ? "unfused (AR tile + cached-Gram product per CG step)"
                     : persist_state == 1 ? "fused, one persistent kernel per solve" : persist_state == 0 ? "fused (not run yet)" : "fused, one launch per CG step");
            return persist_note.empty() ? std::string(buf) : std::string(buf) + " (" + persist_note + ")";
        }
        const bool fused = tile_TI > 0;
        std::string x;
        if (full) x = "full-observation path (shared Gram)";
        else if (fused && ts_possible) x = std::string("fused CG ") + x_form_name(x_form) + (x_form == kXRep && persist_state == 1 ? " (one persistent kernel per solve on every rank)" : "");
        else if (fused) x = "fused CG replicated (too few tiles to shard over time)";
        else if (uts) x = std::string("unfused CG ") + x_form_name(x_form);
        else x = cg_shard ? "unfused CG, cached-Gram product sharded (H d rows gathered per step)" : "unfused CG replicated";
        std::string meas;
        for (int f : x_cands) {
            char t[96];
            snprintf(t, sizeof t, "%s%s %.3f ms", meas.empty() ? "" : ", ", x_form_name(f), x_ms_all[f]);
            meas += t;
        }
        const bool rep_gram = !full && (fused ? x_form == kXRep : (!uts && !cg_shard));
        snprintf(buf, sizeof buf, "%d ranks; F rows %s%s; X-side Gram rows %s; %s%s%s%s; peer-to-peer transport %s%s%s; decided in %d set-up iterations",
                 comm->world, fs_mode == kShardOff ? "replicated" : fs_mode == kShardOn ? "sharded + all-gather of H" : "sharded (undecided)",
                 (fs_mode == kShardOn && fchunks >= 2) ? " in overlapped chunks" : "",
                 !rep_gram ? "own timestamps only (never gathered)" : gramx_mode == kGramxReplicate ? "replicated" : gramx_mode == kGramxShard ? "sharded + all-gather of G" : "sharded (undecided)",
                 x.c_str(), meas.empty() ? "" : " [slowest rank's X phase: ", meas.c_str(), meas.empty() ? "" : "]",
                 p2p.on ? "available" : "unavailable", p2p.note.empty() ? "" : ": ", p2p.note.c_str(), tuned_iters);
        return persist_note.empty() ? std::string(buf) : std::string(buf) + " (" + persist_note + ")";
    }

    // ---- the ALS loop (trmf.cpp:647-693) --------------------------------------------------------------------
    int run(int iters) {
        if (comm->world == 1 && snap_iter < 0 && period_W > 0 && iters > 0 && tile_TI > 0 && !persist_failed) {
            // the persistent kernel is about to be used: the state to come back to if it times out (persist_recover)
            const int maxcg = (int)std::min<long long>(max_cg_iter, (long long)T * k);
            FillStreamScope fill(stream);                 // (persist_usable() allocates the exchange tables on first use)
            if (maxcg <= kCgHistCap && persist_usable(maxcg) && take_snapshot()) return kFail;
        }
        for (int it = 0; it < iters; it++) {
            const int iter1 = ++iter;                       // 1-based like the reference
            DeviceIterLog *L = log.p + ((iter1 - 1) % kLogCap);
            PhaseEvents &ev = events[(iter1 - 1) % kEventRing];
            static const DeviceIterLog blank = [] { DeviceIterLog b; std::memset(&b, 0, sizeof b); b.normF = b.normX = b.normLV = -1; return b; }();
            const bool doF = period_H > 0 && (iter1 % period_H) == 0;
            const bool doX = period_W > 0 && (iter1 % period_W) == 0;
            const bool doL = period_Lag > 0 && (iter1 % period_Lag) == 0;
            // quiet runs: accept_kernel writes the whole record; otherwise a blank record is uploaded and the
            // phases fill it in (two small copies per iteration on the stream)
            const bool device_log = doX && !log_norms && !verbose;
            if (!device_log) TRMF_HIP_CHECK(hipMemcpyAsync(L, &blank, sizeof blank, hipMemcpyHostToDevice, stream));
            TRMF_HIP_CHECK(hipEventRecord(ev.f0, stream));
            if (doF) {
                if (full ? fsolve_full(ev) : fsolve(ev)) return kFail;
                if (log_norms || verbose) log_norm(H.p, (size_t)n * KP, &L->normF);
                if (verbose) fprintf(stderr, ">> iter %d F %g\n", iter1, host_double(&L->normF));
            } else {
                TRMF_HIP_CHECK(hipEventRecord(ev.fk0, stream));
                TRMF_HIP_CHECK(hipEventRecord(ev.fk1, stream));
            }
            TRMF_HIP_CHECK(hipEventRecord(ev.f1, stream));
            if (!doX) TRMF_HIP_CHECK(hipEventRecord(ev.xg1, stream));
            if (doX) {
                xg1_event = ev.xg1;
                const int xrc = xsolve(device_log ? &L->x : nullptr, device_log ? &L->normF : nullptr);
                xg1_event = nullptr;
                if (xrc) return kFail;
                if (log_norms || verbose) log_norm(W.p, (size_t)T * KP, &L->normX);
                if (!device_log)
                    TRMF_HIP_CHECK(hipMemcpyAsync(&L->x, xstate.p, sizeof(XState), hipMemcpyDeviceToDevice, stream));
                if (verbose) {
                    fprintf(stderr, ">> iter %d X %g\n", iter1, host_double(&L->normX));
                    if (verbose >= 2) {
                        XState hx;
                        (void)hipMemcpy(&hx, xstate.p, sizeof hx, hipMemcpyDeviceToHost);
                        fprintf(stdout, "iter  1 act %5.3e pre %5.3e delta %5.3e f %5.3e |g| %5.3e CG %3d |g| %5.3e\n",
                                hx.actred, hx.prered, hx.delta, hx.f, hx.gnorm, hx.cg_iter, hx.cg_rnorm);
                        fflush(stdout);
                    }
                }
            }
            TRMF_HIP_CHECK(hipEventRecord(ev.x1, stream));
            if (doL) {
                if (verbose) {
                    log_norm(theta.p, (size_t)nlag * k, &L->normLV);
                    fprintf(stderr, ">> iter %d LV(%d %d) %g\n", iter1, nlag, k, host_double(&L->normLV));
                }
                if (theta_solve()) return kFail;
                if (log_norms || verbose) log_norm(theta.p, (size_t)nlag * k, &L->normLV);
                if (verbose) fprintf(stderr, ">> iter %d LV %g\n", iter1, host_double(&L->normLV));
            }
            TRMF_HIP_CHECK(hipEventRecord(ev.lv1, stream));
        }
        return 0;
    }

    // ---- recovery when the persistent kernel's co-residency assumption breaks (one rank; VERDICT / ADVICE r4) -----------------
    // The one-GPU X-solve is ONE kernel whose workgroups wait for each other; if something else holds compute units (a second
    // process on the GPU) a poll runs into its bound, the kernel ends with XState::p2p_error set and every later persistent
    // launch returns at once.  The iterates since then are void -- and a timeout in the last exchange can leave W half-updated
    // -- so the session keeps a snapshot of (W, H, Theta, iteration counter) as of its last CHECKED synchronisation: sync()
    // restores it, switches to the launch-per-step path (bit-identical iterates, no co-residency needed) for the rest of the
    // session's life and repeats the iterations since the snapshot.  Cost: three device-to-device copies per sync()
    // (18 MB at config 3, ~10 us), only while the persistent kernel is in use.
    DevBuf<real> snapW, snapH, snapT;
    int snap_iter = -1;
    int take_snapshot() {
        const size_t nw = (size_t)(T + 1) * KP, nh = (size_t)(n + 1) * KP, nt = (size_t)nlag * k;
        FillStreamScope fill(stream);
        if (snapW.alloc(nw, false) || snapH.alloc(nh, false) || snapT.alloc(nt, false)) return kFail;
        TRMF_HIP_CHECK(hipMemcpyAsync(snapW.p, W.p, nw * sizeof(real), hipMemcpyDeviceToDevice, stream));
        TRMF_HIP_CHECK(hipMemcpyAsync(snapH.p, H.p, nh * sizeof(real), hipMemcpyDeviceToDevice, stream));
        if (nt) TRMF_HIP_CHECK(hipMemcpyAsync(snapT.p, theta.p, nt * sizeof(real), hipMemcpyDeviceToDevice, stream));
        snap_iter = iter;
        return 0;
    }
    int persist_recover(const XState &hx) {
        if (snap_iter < 0) { set_error("persistent CG kernel timed out and no snapshot exists"); return kFail; }
        const int redo = iter - snap_iter;
        if (verbose || getenv("TRMF_P2P_VERBOSE"))
            fprintf(stderr, ">> persistent CG kernel: a poll ran into its bound (exchange %lld, tile %lld, %s missing; is another process using the GPU?): "
                    "repeating %d iteration(s) with one launch per CG step\n", hx.p2p_diag[0], hx.p2p_diag[1], hx.p2p_diag[2] == 1 ? "records" : "halo rows", redo);
        persist_state = -1; persist_failed = true;
        persist_note = "the persistent kernel ran into a poll bound after iteration " + std::to_string(snap_iter) + ": one launch per CG step since";
        const size_t nw = (size_t)(T + 1) * KP, nh = (size_t)(n + 1) * KP, nt = (size_t)nlag * k;
        TRMF_HIP_CHECK(hipMemsetAsync(&xstate.p->p2p_error, 0, sizeof(int), stream));
        TRMF_HIP_CHECK(hipMemcpyAsync(W.p, snapW.p, nw * sizeof(real), hipMemcpyDeviceToDevice, stream));
        TRMF_HIP_CHECK(hipMemcpyAsync(H.p, snapH.p, nh * sizeof(real), hipMemcpyDeviceToDevice, stream));
        if (nt) TRMF_HIP_CHECK(hipMemcpyAsync(theta.p, snapT.p, nt * sizeof(real), hipMemcpyDeviceToDevice, stream));
        iter = snap_iter;
        if (run(redo)) return kFail;
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        snapW.release(); snapH.release(); snapT.release(); snap_iter = -1;
        return 0;
    }

    // recover = false: report a timed-out persistent kernel instead of repeating its iterations (session teardown)
    int sync(bool recover = true) {
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
#if defined(TRMF_PERSIST_PROF)
        if (persist_prof.p) {
            std::vector<long long> hp((size_t)2 * kProfIters * kProfSlots + 2 * (size_t)kPersistMaxTiles);
            TRMF_HIP_CHECK(hipMemcpy(hp.data(), persist_prof.p, hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
            for (int sel = 0; sel < 2; sel++) {
                const long long t0 = hp[(size_t)sel * kProfIters * kProfSlots];
                for (int i = 0; i < kProfIters; i++) {
                    const long long *r = hp.data() + ((size_t)sel * kProfIters + i) * kProfSlots;
                    if (!r[0] && !r[3]) continue;
                    fprintf(stderr, "PERSIST_PROF tile %s row %2d:", sel ? "mid" : "0  ", i);
                    for (int q = 0; q < kProfSlots; q++) fprintf(stderr, " %8.2f", r[q] ? (r[q] - t0) * 0.01 : 0.0);
                    fprintf(stderr, "  us\n");
                }
            }
            {   // CG iteration 5 of every tile, relative to the earliest collect: when its collect finished, when it published
                const long long *q = hp.data() + (size_t)2 * kProfIters * kProfSlots;
                long long base = 0;
                for (int t = 0; t < nbt; t++) if (q[2 * t] && (!base || q[2 * t] < base)) base = q[2 * t];
                fprintf(stderr, "PERSIST_TILES");
                for (int t = 0; t < nbt; t++) fprintf(stderr, " %.2f/%.2f", (q[2 * t] - base) * 0.01, (q[2 * t + 1] - base) * 0.01);
                fprintf(stderr, "\n");
            }
        }
#endif
        if (persist_state == 1 && comm->world == 1) {          // one rank: a timed-out solve is repeated on the launch-per-step path
            XState hx;
            TRMF_HIP_CHECK(hipMemcpy(&hx, xstate.p, sizeof hx, hipMemcpyDeviceToHost));
            if (hx.p2p_error && recover) return persist_recover(hx);
            if (hx.p2p_error) { set_error("persistent CG kernel: an exchange between workgroups timed out"); return kFail; }
            if (recover && snap_iter >= 0 && snap_iter != iter && take_snapshot()) return kFail;
        } else if (persist_state == 1 || (persist_shard_state == 1 && x_form == kXTsPersist)) {   // several ranks: a bounded poll of the persistent CG kernel ran out (the GPU must not hang)
            XState hx;
            TRMF_HIP_CHECK(hipMemcpy(&hx, xstate.p, sizeof hx, hipMemcpyDeviceToHost));
            if (hx.p2p_error) { set_error("persistent CG kernel: an exchange between workgroups timed out (is another process using the GPU? TRMF_PERSIST=0 selects the launch-per-step path)"); return kFail; }
        }
        if (p2p.on) {               // a bounded wait of the peer-to-peer exchange ran out: the factors are not to be trusted
            XState hx;
            TRMF_HIP_CHECK(hipMemcpy(&hx, xstate.p, sizeof hx, hipMemcpyDeviceToHost));
            if (hx.p2p_error) {
                char msg[256];
                snprintf(msg, sizeof msg, "time-sharded CG: a peer-to-peer exchange timed out (TRMF_CG=p2p): rank %d, message %lld, launch %lld, "
                         "peer %lld: expected epoch %lld, flag %lld", comm->rank, hx.p2p_diag[0] / 1000000, hx.p2p_diag[0] / 1000 % 1000 - 1,
                         hx.p2p_diag[0] % 1000, hx.p2p_diag[1], hx.p2p_diag[2]);
                set_error(msg);
                if (test_env("TRMF_P2P_DEBUG")) {       // what this rank derived its stop decisions from
                    std::vector<double> hp((size_t)P_NSLOTS * xp.pstride);
                    TRMF_HIP_CHECK(hipMemcpy(hp.data(), pbase(), hp.size() * sizeof(double), hipMemcpyDeviceToHost));
                    fprintf(stderr, "[p2p debug] rank %d: %s\n  stop_it %d cg_iter %d rho %.17g %.17g %.17g gnorm %.17g cgtol %.9g\n", comm->rank, msg,
                            hx.stop_it, hx.cg_iter, hx.rho_hist[0], hx.rho_hist[1], hx.rho_hist[2], hx.gnorm, (double)hx.cgtol);
                    for (int a = 0; a < P_NSLOTS; a++) {
                        double t = 0; int nz = 0;
                        for (int q = 0; q < xp.pstride; q++) { t += hp[(size_t)a * xp.pstride + q]; nz += hp[(size_t)a * xp.pstride + q] != 0; }
                        fprintf(stderr, "  partial array %2d: sum %.17g (%d nonzero)\n", a, t, nz);
                    }
                }
                return kFail;
            }
        }
        return 0;
    }

    int stats(TrmfIterStats *out, int cap) {
        if (sync()) return kFail;
        const int avail = std::min(iter, std::min(kLogCap, kEventRing));
        const int cnt = std::min(cap, avail);
        for (int q = 0; q < cnt; q++) {
            const int it0 = iter - cnt + q;                 // 0-based iteration index
            DeviceIterLog hl;
            TRMF_HIP_CHECK(hipMemcpy(&hl, log.p + (it0 % kLogCap), sizeof hl, hipMemcpyDeviceToHost));
            PhaseEvents &ev = events[it0 % kEventRing];
            TrmfIterStats &o = out[q];
            o.normF = hl.normF; o.normX = hl.normX; o.normLV = hl.normLV;
            o.f = hl.x.f; o.fnew = hl.x.fnew; o.actred = hl.x.actred; o.prered = hl.x.prered;
            o.gnorm = hl.x.gnorm; o.cg_rnorm = hl.x.cg_rnorm; o.cg_iter = hl.x.cg_iter; o.accepted = hl.x.accepted; o.delta = hl.x.delta;
            o.cg_rnorm_direct = hl.x.rho_direct >= 0 ? std::sqrt(hl.x.rho_direct) : -1.0;
            o.ms_F = o.ms_X = o.ms_LV = o.ms_F_kernel = o.ms_X_gram = 0;
            (void)hipEventElapsedTime(&o.ms_X_gram, ev.f1, ev.xg1);
            (void)hipEventElapsedTime(&o.ms_F, ev.f0, ev.f1);
            (void)hipEventElapsedTime(&o.ms_F_kernel, ev.fk0, ev.fk1);
            (void)hipEventElapsedTime(&o.ms_X, ev.f1, ev.x1);
            (void)hipEventElapsedTime(&o.ms_LV, ev.x1, ev.lv1);
        }
        return cnt;
    }

    // J = 0.5*sum_Omega (Y - w.h)^2 + 0.5*lambdaI(|W|^2+|H|^2) + 0.5*lambdaAR*AR(W;Theta)  
    double objective() {
        if (full) return NAN;       // defined for the observed-entries objective only
        XState *st = xstate.p;
        if (loss(W.p, true)) return NAN;
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(256), 0, stream, lossrow.p, T, &st->loss1);
        const double l = host_double(&st->loss1);
        {   // AR and ridge sums of W: the AR tile kernel's partials (its operator output goes to scratch)
            ArVecs av{};
            av.v = W.p;
            hipLaunchKernelGGL((ar_tile_kernel<AR_PLAIN>), dim3((T + ar_TI - 1) / ar_TI, KP / kArCols), dim3(kArThreads),
                               ar_tile_lds_bytes(ar_TI, midx, nlag), stream, xp, st, av, 0, 0, 0, lag_set.p, lag_steps.p, nsteps,
                               theta.p, arbase.p, pbase(), ar_TI, 0);
        }
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(256), 0, stream, P(P_AR), nbar, &st->gs);
        const double ar = host_double(&st->gs);
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(256), 0, stream, P(P_VV), nbar, &st->gs);
        const double w2 = host_double(&st->gs);
        log_norm(H.p, (size_t)n * KP, &st->gs);
        const double h2 = host_double(&st->gs);
        return 0.5 * l + 0.5 * lambdaI * (w2 + h2) + 0.5 * lambdaAR * ar;
    }

    // algorithmic bytes of one F-solve launch on this rank
    double fsolve_bytes() const {
        return fs_mode == kShardOff ? bytes_for_rows(0, (uint64_t)n) : bytes_for_rows(fbounds[comm->rank], fbounds[comm->rank + 1]);
    }
    double bytes_for_rows(uint64_t rb, uint64_t re) const {
        const double sz = sizeof(real);
        const double nz = host_col_ptr.empty() ? 0.0 : (double)(host_col_ptr[re] - host_col_ptr[rb]);
        const double rows = (double)(re - rb);
        return nz * (4.0 + sz + k * sz) + (rows + 1) * 8.0 + rows * k * sz;
    }
};

}  // namespace trmf
