// session_transport.hpp -- how the ranks of a session exchange data: all-gathers through the communicator, the peer-to-peer
// arenas (IPC-exported uncached memory, set up as a collective trial with a safe fallback), the per-launch exchanges of the
// time-sharded CG, and the measure-once rule "shard a phase or replicate it" (DESIGN.md section 6).
#pragma once

#include <initializer_list>

#include "session_state.hpp"

namespace trmf {

struct SessionTransport : SessionState {
    void release_p2p() {
        if (!p2p.loopback) for (void *q : p2p.peer) if (q) (void)hipIpcCloseMemHandle(q);
        p2p.loopback = false;
        p2p.peer.clear();
        // ranks as threads of one process: the peers hold RAW pointers into this arena; nobody frees before everybody's queued work
        // (which may still store into a peer) has drained -- every rank comes through here at the same point of the set-up / teardown
        // (every rank, whether or not its own allocation succeeded: the barrier is the group's)
        if (comm && comm->in_process() && comm->world > 1) { if (stream) (void)hipStreamSynchronize(stream); (void)comm->barrier(); }
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
        if (comm->solo()) {
            // loop-back (SoloComm): one uncached arena, every "peer" pointer is this rank's own copy; no handles, no trial.  Only the
            // persistent-kernel form runs on it (its solo mode waits for nobody); the launch-per-step exchanges would wait for flags
            // no peer raises, so without TRMF_CG=persist the transport stays unavailable as before.
            const char *e = getenv("TRMF_CG");
            if (!(e && e[0] == 'p' && e[1] == 'e')) return unavailable("no peers under the solo communicator");
            const size_t msg_bytes = (msg_doubles * sizeof(double) + 255) / 256 * 256, flag_bytes = (size_t)3 * W_ * kFlagStride * sizeof(unsigned long long);
            p2p.ext_off = (3 * msg_bytes + flag_bytes + 255) / 256 * 256;
            p2p.ext_ll_bytes = (ext_ll_bytes + 255) / 256 * 256;
            p2p.ext_bytes = p2p.ext_ll_bytes + ext_hll_bytes;
            p2p.bytes = p2p.ext_off + p2p.ext_bytes;
            if (hipExtMallocWithFlags(&p2p.arena, p2p.bytes, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); p2p.arena = nullptr; return unavailable("no uncached arena"); }
            TRMF_HIP_CHECK(hipMemsetAsync(p2p.arena, 0, p2p.bytes, stream));
            p2p.peer.assign(W_, nullptr);
            p2p.loopback = true;
            PeerTable tab{};
            for (int r = 0; r < W_; r++) {
                unsigned char *base = (unsigned char *)p2p.arena;
                if (r != me) p2p.peer[r] = p2p.arena;
                for (int m = 0; m < 3; m++) {
                    tab.msg[m][r] = reinterpret_cast<double *>(base + m * msg_bytes);
                    tab.flags[m][r] = reinterpret_cast<unsigned long long *>(base + 3 * msg_bytes) + (size_t)m * W_ * kFlagStride;
                }
            }
            if (peer_table.upload(&tab, 1)) return kFail;
            for (int m = 0; m < 3; m++) { p2p.msg[m] = tab.msg[m][me]; p2p.epoch[m] = 0; }
            p2p.on = true;
            return 0;
        }
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
            const bool inproc = comm->in_process();          // threads of one process: the pointer itself travels (comm.hpp)
            ok = ok && !p2p_forced_failure("export") && (inproc || hipIpcGetMemHandle(&h, p2p.arena) == hipSuccess);
            if (ok && inproc) std::memcpy(mine, &p2p.arena, sizeof(void *));
            else if (ok) std::memcpy(mine, &h, sizeof h);
            else (void)hipGetLastError();
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
            if (r != me && comm->in_process()) {
                void *q = nullptr;
                std::memcpy(&q, handles.data() + kSlot * r, sizeof(void *));
                const int mydev = comm->device_of(me), rdev = comm->device_of(r);
                if (rdev != mydev) {                        // another device of this process: direct access over xGMI
                    const hipError_t e = hipDeviceEnablePeerAccess(rdev, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); ok = false; break; }
                    (void)hipGetLastError();
                }
                p2p.peer[r] = q;
                base = (unsigned char *)q;
            } else if (r != me) {
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
        if (comm->in_process()) p2p.loopback = true;        // raw pointers: nothing to close in release_p2p()
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
    double *pbase() { return pbase_override ? pbase_override : partials.p; }
    double *P(int slot) { return pbase() + (size_t)slot * xp.pstride; }

    // ---- all-gather helpers ------------------------------------------------------------------------
    int gather_rows(void *dbuf, const std::vector<uint64_t> &bounds, size_t row_bytes) {
        if (comm->world == 1 && !comm->call_when_single) return 0;
        std::vector<uint64_t> off(bounds.size());
        for (size_t i = 0; i < bounds.size(); i++) off[i] = bounds[i] * row_bytes;
        return comm->allgatherv(dbuf, off.data(), stream);
    }
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
};

}  // namespace trmf
