// session_xphase.hpp -- the X-solve (trmf.cpp:665-674 -> rf_tron.h:134-254) and the Theta-solve: the X-side Gram cache, the
// forms of the CG (one persistent kernel per solve; one launch per step, replicated or sharded over time; the unfused
// two-kernel step for long lag sets), their geometry and the measure-once choice between them.
#pragma once

#include "session_fphase.hpp"

namespace trmf {

struct SessionXPhase : SessionFPhase {
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

    // ---- X-side Gram cache / loss ---------------------------------------------------------------------
    template <int NT_> void launch_gram_x(uint32_t rb, uint32_t re) {
        if (re <= rb) return;
        // skewed row lengths (session_state.hpp: LongRows::skewed): one wavefront per WORKGROUP, so that a finished row's slot is free at once
        const uint32_t wpb = longX.skewed ? 1u : 4u;
        const dim3 grid((re - rb + wpb - 1) / wpb), block(64 * wpb), lblock(256);
        uint32_t lo = 0, hi = 0;
        if (longX.any()) longX.range(rb, re, lo, hi);     // split timestamps: partial Grams per item, then their sums -> G_i / b_i
        const dim3 lgrid((hi - lo + 3) / 4);
        // longest rows first (LongRows::d_order) when the launch covers every timestamp; a rank's block runs in index order
        const uint32_t *xorder = (longX.skewed && longX.d_order.p && rb == 0 && re == (uint32_t)T) ? longX.d_order.p : nullptr;
#define TRMF_LAUNCH_GRAM_X(PAD, PACKED)                                                                                      \
    do {                                                                                                                     \
        if (hi > lo) {                                                                                                       \
            launch_gram_part<NT_, PAD>(longX, lo, hi, Yr_idx.p, Yr_val.p, H.p, (uint32_t)n);                                 \
            hipLaunchKernelGGL((gram_x_long_kernel<NT_, PAD, PACKED>), lgrid, lblock, 0, stream, split_view(longX, lo, hi), G.p, Bv.p, k, xp.gstride); \
        }                                                                                                                    \
        hipLaunchKernelGGL((gram_x_kernel<NT_, PAD, PACKED>), grid, block, 0, stream, Yr_ptr.p, Yr_idx.p, Yr_val.p, H.p, G.p, Bv.p, \
                           rb, re, k, (uint32_t)n, xp.gstride, longX.thresh, xorder);                                        \
    } while (0)
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
    int xprepare_full() {        // init() of arr_ls_fY_IX, trmf.cpp:183-187
        const uint32_t rb = (uint32_t)xbounds[comm->rank], re = (uint32_t)xbounds[comm->rank + 1];
        if (y_times_factor(false, H.p, Bv.p, dense ? 0u : rb, dense ? (uint32_t)T : re)) return kFail;  // Y H
        if (!dense && gather_rows(Bv.p, xbounds, (size_t)KP * sizeof(real))) return kFail;
        small_gram(H.p, n, real(0), GSx.p, stream);                                                     // H^T H
        return 0;
    }

    // ---- X-solve (trmf.cpp:665-674 -> rf_tron.h:134-254) -----------------------------------------------
    // Fused path (the AR halo fits LDS): hv_tile_kernel in its four roles, one launch per CG iteration.
    template <int MODE, bool SHARD> void launch_hv_tile_as(const HvVecs &a, int it, int last, const double *rec_in, double *rec_out) {
        const size_t lds = hv_tile_lds_bytes(tile_TI, midx, KP, nlag, k);
        const TileShard &sh = SHARD ? tsh_rank : tsh;
        const PeerTable *pt = (SHARD && p2p_use) ? peer_table.p : nullptr;
        const int mi = rec_out == xm[0] ? 0 : rec_out == xm[1] ? 1 : 2;
        // (wide tiles -- 512 threads, one rank only: session.hpp, choose_tile() -- have their own instantiations, unit_hv_wide.hip)
#define TRMF_LAUNCH_HV_KQ(KQ)                                                                                        \
        if (!SHARD && tile_nth == 512)                                                                               \
            hipLaunchKernelGGL((hv_tile_kernel<MODE, KQ, false, 512>), dim3(sh.ntiles), dim3(512), lds, stream, xp, xstate.p, a, sh, it, last,  \
                               lag_set.p, theta.p, Gmat(), rec_in, rec_out, pt, mi, tile_TI);                         \
        else                                                                                                         \
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
        else if (tile_nth == 512)
            hipLaunchKernelGGL((cg_close_kernel<false, 512>), dim3(sh.ntiles), dim3(512), 0, stream, xp, xstate.p, sh, tile_TI, mc[0], mc[1], dbuf[0],
                               dbuf[1], rbuf[0], rbuf[1], hbuf[0], hbuf[1], s.p, g.p, W.p, w_new.p, mg, pt);
        else
            hipLaunchKernelGGL(cg_close_kernel<false>, dim3(sh.ntiles), dim3(256), 0, stream, xp, xstate.p, sh, tile_TI, mc[0], mc[1], dbuf[0],
                               dbuf[1], rbuf[0], rbuf[1], hbuf[0], hbuf[1], s.p, g.p, W.p, w_new.p, mg, pt);
        // the records of the closing launch (+ under cg_direct the halo rows of s, operand of the pass below)
        if (shard && exchange(2, -1, cg_direct ? 1 : 0, cg_direct ? s.p : nullptr, nullptr, nullptr)) return kFail;
        if (cg_direct) {                                                           // diagnostics: s^T H s by one more operator pass (session_state.hpp)
            a = HvVecs{};
            a.v = s.p; a.out = hbuf[0];
            launch_hv_tile<HV_PLAIN>(shard, a, 0, 0, nullptr, mg);                 // H s, <s,Hs> (fields [0..2] of the same records)
            if (shard && exchange(2, -1, 0, nullptr, nullptr, nullptr)) return kFail;
        }
        const int nb = (int)std::min<size_t>(kMaxPartials, ((size_t)(sh.row_e - sh.row_b) * KP + 255) / 256);
        if (!shard && tile_nth == 512)                 // (the records are summed in the tiles' own order: thread stride = workgroup size)
            hipLaunchKernelGGL(accept_tile_kernel<512>, dim3(std::max(nb, 1)), dim3(512), 0, stream, xp, xstate.p, mg, sh, 0, w_new.p, W.p, log_x, log_n, cg_direct ? 1 : 0);
        else
            hipLaunchKernelGGL(accept_tile_kernel<256>, dim3(std::max(nb, 1)), dim3(256), 0, stream, xp, xstate.p, mg, sh,
                               shard ? 1 : 0, w_new.p, W.p, log_x, log_n, cg_direct ? 1 : 0);
        TRMF_HIP_CHECK(hipGetLastError());
        if (shard && gather_rows(W.p, tbounds, (size_t)KP * sizeof(real))) return kFail;   // the F-solve gathers rows of all of W
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
#define TRMF_PERSIST_PREP(KQV) slots = tile_nth == 512 ? persist_prepare<KQV, false, 512>(lds) : persist_prepare<KQV>(lds)
                TRMF_PERSIST_SWITCH(TRMF_PERSIST_PREP)
#undef TRMF_PERSIST_PREP
                if (slots >= nbt && ll_rec.alloc((size_t)2 * nbt * kLLWords) == 0 &&
                    ll_vec.alloc((size_t)2 * T * KP * (2 * sizeof(real) / sizeof(unsigned long long))) == 0) persist_state = 1;
            }
        }
        return persist_state == 1;
    }
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
        pa.direct = cg_direct ? 1 : 0;
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
        unsigned long long *emu_done = nullptr;
        if (shard && comm->solo()) {
            // loop-back measurement (solo communicator): one workgroup on the side stream plays the other ranks
            if (!side) TRMF_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            if (!emu_ready) { TRMF_HIP_CHECK(hipEventCreateWithFlags(&emu_ready, hipEventDisableTiming)); TRMF_HIP_CHECK(hipEventCreateWithFlags(&emu_end, hipEventDisableTiming)); }
            PeerEmuArgs ea{};
            ea.ll = pa.ll; ea.hll = pa.hll; ea.epoch0 = pa.epoch0; ea.nbt = nbt; ea.tile0 = tsh_rank.tile0; ea.ntiles = tsh_rank.ntiles;
            ea.row_b = tsh_rank.row_b; ea.row_e = tsh_rank.row_e; ea.T = T; ea.KP = KP; ea.midx = midx; ea.max_x = maxcg + 6; ea.elem_bytes = (int)sizeof(real);
            // the done word: the first flag word of the arena's message area (unused by the persistent form)
            emu_done = reinterpret_cast<unsigned long long *>((unsigned char *)p2p.arena + p2p.ext_off - 256);
            ea.done = emu_done;
            TRMF_HIP_CHECK(hipEventRecord(emu_ready, stream));               // everything the solve reads is in place
            TRMF_HIP_CHECK(hipStreamWaitEvent(side, emu_ready, 0));
            hipLaunchKernelGGL(persist_peer_emulator_kernel, dim3(1), dim3(256), 0, side, ea);
            TRMF_HIP_CHECK(hipGetLastError());
        }
#define TRMF_PERSIST_GO(KQV) if (shard ? persist_launch<KQV, true>(pa, lds) : tile_nth == 512 ? persist_launch<KQV, false, 512>(pa, lds) : persist_launch<KQV, false>(pa, lds)) return kFail
        TRMF_PERSIST_SWITCH(TRMF_PERSIST_GO)
#undef TRMF_PERSIST_GO
        if (emu_done) {
            hipLaunchKernelGGL(store_u64_kernel, dim3(1), dim3(1), 0, stream, emu_done, (unsigned long long)pa.epoch0);   // ends the emulator
            TRMF_HIP_CHECK(hipEventRecord(emu_end, side));
            TRMF_HIP_CHECK(hipStreamWaitEvent(stream, emu_end, 0));           // the next solve's tables are not touched by a late emulator
        }
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
        if (xg1_event) TRMF_EVREC(xg1_event, stream);
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
        // one rank: stay kCgLook steps ahead of the device and stop enqueuing once the CG has stopped (session_state.hpp, cg_note)
        const bool follow_note = comm->world == 1 && maxcg > kCgLook + 1 && maxcg < 128 && !test_env("TRMF_NO_CG_FOLLOW") && ensure_cg_note() == 0;
        const unsigned int seq = follow_note ? (++cg_seq & 0xffffffu) : 0u;
        bool note_live = follow_note;
        for (int it = 1; it <= maxcg; it++) {                            // launch `maxcg` only closes the last iteration
            av.v = dbuf[(it - 1) & 1]; av.r_in = rbuf[(it - 1) & 1]; av.hd_in = hbuf[(it - 1) & 1];
            av.s = s.p; av.d_out = dbuf[it & 1]; av.r_out = rbuf[it & 1];
            av.note = follow_note ? cg_note : nullptr; av.note_seq = seq;
            if (hv(av, it, it == maxcg ? 1 : 0, 0, hbuf[it & 1], 1)) return kFail;
            if (note_live && it < maxcg && it > kCgLook) {
                // wait until step it - kCgLook has been decided; the device is then still kCgLook steps (>= 100 us of work) behind the queue's end
                const unsigned int need = (unsigned int)(it - kCgLook);
                const double t_wait = now_s();
                bool stopped = false;
                for (unsigned int spins = 0;; spins++) {
                    const unsigned int v = __atomic_load_n(cg_note, __ATOMIC_ACQUIRE);
                    if ((v >> 8) == seq) {
                        if (v & 1u) { stopped = true; break; }
                        if (((v >> 1) & 0x7fu) >= need) break;
                    }
                    if ((spins & 1023u) == 1023u && now_s() - t_wait > 2.0) { note_live = false; break; }   // a stuck device is the synchronisation's problem, not this loop's
                    __builtin_ia32_pause();
                }
                if (stopped) break;
            }
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
        if (uts && uts_exchange(-1, cg_direct ? 1 : 0, cg_direct ? s.p : nullptr, nullptr, nullptr, {{P_GS, 2}, {P_SR, 2}, {P_SS, 2}})) return kFail;
        if (cg_direct) {                                                 // diagnostics: H s, <s,Hs> by one more operator application
            av = ArVecs{};
            av.v = s.p;
            if (hv(av, -1, 0, 0, hbuf[0], 1)) return kFail;
        }
        hipLaunchKernelGGL(accept_kernel, dim3(nbw), dim3(256), 0, stream, xp, st, Pb, npw, ndot, (const double *)nullptr, w_new.p,
                           W.p, log_x, log_n, own_b, own_e, cg_direct ? 1 : 0);
        TRMF_HIP_CHECK(hipGetLastError());
        if (uts && gather_rows(W.p, ubounds, (size_t)KP * sizeof(real))) return kFail;   // the F-solve gathers rows of all of W
        return end_timed();
    }

    // ---- Theta solve (trmf.cpp:677-689 -> 455-484) ------------------------------------------------------
    size_t theta_gram_lds() const { return theta_gram_lds_bytes(midx); }
    size_t theta_solve_lds() const { return (size_t)(nlag * nlag + nlag) * sizeof(real); }
    int theta_solve(hipStream_t stream) {
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
};

}  // namespace trmf
