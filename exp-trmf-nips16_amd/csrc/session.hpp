// session.hpp -- host side of the MI355X TRMF solver: set-up of the HBM-resident problem (asynchronous, one synchronisation),
// growth by appended rows, the measure-once decisions, the outer ALS loop (trmf.cpp:599-694) as asynchronous kernel launches
// on one HIP stream, recovery from a timed-out persistent kernel, statistics.  State and the phases live in the layers below
// (session_state.hpp lists them).
//
// With several ranks (one process per GPU) the nnz-heavy kernels (F-solve, X-side Gram, loss) run on this rank's contiguous row
// block; the CG runs replicated or sharded over time (DESIGN.md "Multi-GPU").
#pragma once

#include "session_xphase.hpp"

extern "C" char **environ;      // (the TRMF_* environment is part of the key of the decision cache)

namespace trmf {

struct TrmfSessionImpl : SessionXPhase {
    ~TrmfSessionImpl() {
        // nothing of this session may still be running when its buffers go back to the pool and its stream to the cache
        if (stream) (void)hipStreamSynchronize(stream);
        if (side) (void)hipStreamSynchronize(side);
        if (aux_theta) (void)hipStreamSynchronize(aux_theta);
        for (auto &e : events) {
            hipEvent_t all[] = {e.f0, e.fk0, e.fk1, e.f1, e.xg1, e.x1, e.lv1};
            for (hipEvent_t ev : all) if (ev) (void)hipEventDestroy(ev);
        }
        for (hipEvent_t ev : {gx0, gx1, gx2, fs0, fs1, fs2, ts0, ts1}) if (ev) (void)hipEventDestroy(ev);
        release_p2p();
        for (hipEvent_t ev : {ov_b, ov_c[0], ov_c[1], ov_c[2], ov_c[3], emu_ready, emu_end}) if (ev) (void)hipEventDestroy(ev);
        if (side) (void)hipStreamDestroy(side);
        for (hipEvent_t ev : {theta_fork, theta_done}) if (ev) (void)hipEventDestroy(ev);
        if (cg_note) (void)hipHostFree(cg_note);
        StreamCache::release(aux_theta);
        StreamCache::release(stream);
    }

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
        SyncStreamOnExit drain(stream);
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
        SyncStreamOnExit drain(stream);
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
            // dense Y (only legal with missing == 0): keep both orientations, like CSR + CSC.  The caller's array goes through the
            // pinned ring as it is -- a column-major T x n array IS the row-major n x T orientation -- the other orientation and
            // sum y^2 are formed on the device (round 4 made a host copy and a host pass: 27 of config 1's 30 ms per one-shot call)
            const size_t N = (size_t)T * n;
            if (Yd_tn.alloc(N, false) || Yd_nt.alloc(N, false)) return kFail;
            if (Y->type == TRMF_DENSE_ROWMAJOR) {
                if (HostStager::current().h2d(Yd_tn.p, Y->val, N * sizeof(real), stream)) return kFail;
                launch_transpose(Yd_tn.p, T, n, Yd_nt.p);
            } else {
                if (HostStager::current().h2d(Yd_nt.p, Y->val, N * sizeof(real), stream)) return kFail;
                launch_transpose(Yd_nt.p, n, T, Yd_tn.p);
            }
            if (launch_sum_squares(Yd_tn.p, N)) return kFail;
            bytes_uploaded += (double)N * sizeof(real);
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
        if (finish_sum_squares(&ysq_acc)) return kFail;
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
        // the Theta-solve's own stream, where it will be used: acquired here, not inside somebody's timed iterations (the first
        // hipStreamCreate of a process costs ~20 ms -- it showed up as 3 ms per iteration in a 6-iteration config-5 bench line)
        if (nlag > 0 && theta_overlap_pays() && !test_env("TRMF_NO_OVERLAP")) (void)ensure_aux();
        created = true;
        return autotune();
    }

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
            // one k x k slot per workgroup of the generic F-solve: min(kGenBlocks, item rows) of them (kGenBlocks * k^2 at rank 1024
            // was 2 GiB in fp32 / 4 GiB in fp64 whatever the number of rows)
            if (gen_scratch.alloc((size_t)std::min(kGenBlocks, std::max(n, 1)) * k * k, false)) return kFail;
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
        {   // fused Hv tile: one timestamp row per 16-byte Gram column group, if the AR halo fits a modest LDS budget.
            // Tile geometry.  Narrow: 256 threads, hv_tile_rows(k) timestamps (25 at k = 40), up to two workgroups per CU.  Wide (round 5;
            // one rank only): when the narrow tiles outnumber the CUs -- config 3: 400 tiles on 256 CUs, 144 CUs carry two workgroups
            // and the persistent kernel's pass is as slow as those -- and ceil(T / CUs) timestamps fit a 512-thread workgroup, ONE
            // workgroup per CU with that many timestamps (config 3: 250 tiles of 40): the halo rows are staged once per CU instead
            // of twice and every CU carries the same load (pass 16.7 -> 14.7 us, profiles/r05_wide_tiles.txt).  Both CG forms
            // (persistent kernel, launch per step) have both geometries and are bit-identical to each other WITHIN a geometry; several
            // ranks always use the narrow one (a rank owns 1/N of the tiles), so TRMF_TEST + TRMF_TILE=narrow is what reproduces an
            // N-rank run bit for bit on one GPU.
            int TI = hv_tile_rows(k);
            tile_nth = 256;
            if (const char *e = test_env("TRMF_HV_TI")) TI = std::max(1, std::min(TI, atoi(e)));   // experiments
            // the tile kernel addresses the CG vectors with 32-bit byte offsets through buffer descriptors
            const bool fits32 = (uint64_t)(T + 1) * KP * sizeof(real) < 0x7fffffffull;
            const bool tiles_ok = !generic && fits32 && !test_env("TRMF_NO_HV_TILE");
            const char *tk = test_env("TRMF_TILE");                 // narrow | wide (wide: also where the rule would not choose it)
            if (tiles_ok && comm->world == 1 && !(tk && tk[0] == 'n') && !test_env("TRMF_HV_TI")) {
                hipDeviceProp_t prop;
                int dev = 0;
                TRMF_HIP_CHECK(hipGetDevice(&dev));
                TRMF_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
                const int cus = std::max(1, prop.multiProcessorCount), rows_wide = hv_tile_rows(k, 512), need = (T + cus - 1) / cus;
                int wide_TI = 0;
                // (the wide geometry pays through the persistent kernel; where that cannot run -- switched off, a CG cap beyond its
                // history, LDS -- the launch-per-step path is faster on narrow tiles: 975 against 959 iter/s at config 3)
                const int maxcg = (int)std::min<long long>(max_cg_iter, (long long)T * k);
                const char *pe = getenv("TRMF_PERSIST");
                const bool persist_wanted = !(pe && atoi(pe) == 0) && maxcg <= kCgHistCap;
                if (tk && tk[0] == 'w') wide_TI = std::min(rows_wide, std::max(need, TI + 1));
                else if (persist_wanted && (T + TI - 1) / TI > cus && need <= rows_wide &&
                         persist_lds_bytes(need, midx, KP, nlag, k, (T + need - 1) / need) <= kLdsMax) wide_TI = need;
                if (wide_TI > 0 && hv_tile_lds_bytes(wide_TI, midx, KP, nlag, k) <= 64 * 1024) { TI = wide_TI; tile_nth = 512; }
            }
            if (tiles_ok && hv_tile_lds_bytes(TI, midx, KP, nlag, k) <= (tile_nth == 512 ? 64 : 48) * 1024) {
                tile_TI = TI;
                nbt = (T + TI - 1) / TI;                     // one tile per workgroup
            } else tile_nth = 256;
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
        return setup_split_rows();
    }
    // Long rows of both orientations (session_state.hpp "split rows"): lists, items and the slab of partial Grams.  Observed-entries
    // path with the register-tiled kernels only: the full-observation path has one shared Gram, the generic kernels (rank > 64) already
    // give a row a whole workgroup.
    int setup_split_rows() {
        longF.clear(); longX.clear();
        part_slab.release(); part_stride = 0;
        if (dense || generic) return 0;
        hipDeviceProp_t prop;
        int dev = 0;
        TRMF_HIP_CHECK(hipGetDevice(&dev));
        TRMF_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        const int waves = std::max(1, prop.multiProcessorCount) * 12;        // three wavefronts per SIMD: what the Gram kernels are built for
        if (build_long_rows(longF, host_col_ptr, (size_t)n, (sizeof(real) == 4 && !full) ? 512u : 2048u, waves) ||
            build_long_rows(longX, host_row_ptr, (size_t)T, 2048u, waves, true)) return kFail;
        const uint32_t items = std::max(longF.nitems, longX.nitems);
        if (!items) return 0;
        if (full) {                      // sparse Y with missing == 0: the items' partial rows of Y^T W / Y H (spmm_part_kernel), KP values each
            part_stride = (uint32_t)KP;
            return part_slab.alloc((size_t)items * part_stride, false);
        }
        switch (NT) {
            case 1: part_stride = split_part_reals<1>(true); break;
            case 2: part_stride = split_part_reals<2>(true); break;
            case 3: part_stride = split_part_reals<3>(true); break;
            default: part_stride = split_part_reals<4>(true); break;
        }
        return part_slab.alloc((size_t)items * part_stride, false);
    }
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
        {   // the grown generation of buffers in ONE slab (the pool returns the previous generation's slab once it is wholly free)
            const int Tsave = T; const uint64_t nzsave = nnz;
            T = T1; nnz = dense ? (uint64_t)T1 * n : nnz + Yn->nnz;
            const size_t fp = footprint_estimate();
            T = Tsave; nnz = nzsave;
            DevicePool::current().reserve(fp);
        }
        // Failure-atomic: everything new is built in locals (device buffers, host pointer arrays, sums) and swapped into
        // the session only after every allocation, copy and kernel of the step has succeeded; a failed call leaves the
        // session exactly as it was.
        std::vector<uint64_t> new_row_ptr, new_col_ptr;
        uint64_t new_nnz = nnz;
        double new_ysq = ysq_acc;
        DevBuf<uint32_t> ptr2, idx2, cptr2, cidx2; DevBuf<real> val2, cval2, tn2, nt2, raw2, W2;
        SyncStreamOnExit drain(stream);
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
            SyncStreamOnExit drain_block(stream);
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
            (void)dense_rows_to_rowmajor(Yn, blk);
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
            if (device_sum_squares(tn2.p, new_nnz, &new_ysq)) return kFail;    // over the grown matrix, like a fresh session's
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
    bool decisions_pending() const {
        if (comm->world <= 1 || full) return false;
        if (fs_mode == kShardMeasure && period_H > 0) return true;
        if (period_W <= 0) return false;
        if (x_form < 0) return true;
        const bool rep_gram = tile_TI > 0 ? x_form == kXRep : (!uts && !cg_shard);
        return rep_gram && !cg_shard && gramx_mode == kGramxMeasure;
    }
    // Process-level cache of the decisions, keyed by everything they depend on (shape, element type, communicator, the TRMF_*
    // environment): the second session of a grid_search over the same problem -- c_trmf_train creates one per call -- takes the
    // first one's decisions instead of measuring again (<= 14 iterations).  Every rank makes the same sequence of calls, so the
    // caches agree; a cached form that this session cannot run (the peer-to-peer arena's trial failed this time) is ignored.
    struct TunedDecisions {
        int fs_mode, gramx_mode, x_form, tuned_iters;
        float x_ms[kXForms]; double x_ms_all[kXForms];
        std::vector<int> x_cands;
        std::string persist_note;
    };
    static std::map<std::string, TunedDecisions> &decision_cache() { static std::map<std::string, TunedDecisions> c; return c; }
    static std::mutex &decision_mu() { static std::mutex m; return m; }
    std::string decision_key() const {
        std::string key = std::to_string(T) + "," + std::to_string(n) + "," + std::to_string(k) + "," + std::to_string(nnz) + "," + std::to_string(nlag) + "," +
                          std::to_string(midx) + "," + std::to_string(sizeof(real)) + "," + std::to_string(full) + std::to_string(dense) + "," + std::to_string(comm->id) + "," +
                          std::to_string(comm->world) + "," + std::to_string(period_W > 0) + std::to_string(period_H > 0);
        for (char **e = ::environ; e && *e; e++) if (!strncmp(*e, "TRMF_", 5)) { key += ";"; key += *e; }
        return key;
    }
    int autotune() {
        tuned_iters = 0; decisions_from_cache = false;
        if (const char *e = getenv("TRMF_AUTOTUNE")) if (atoi(e) == 0) return 0;
        if (!decisions_pending()) return 0;
        const std::string key = decision_key();
        {
            std::lock_guard<std::mutex> lk(decision_mu());
            auto it = decision_cache().find(key);
            if (it != decision_cache().end() && it->second.x_cands == x_cands) {
                const TunedDecisions &d = it->second;
                fs_mode = d.fs_mode; gramx_mode = d.gramx_mode; x_form = d.x_form; persist_note = d.persist_note;
                for (int f = 0; f < kXForms; f++) { x_ms[f] = d.x_ms[f]; x_ms_all[f] = d.x_ms_all[f]; }
                x_calls = 2 * (int)x_cands.size();
                decisions_from_cache = true;
                if (!decisions_pending()) {
                    if (verbose && comm->rank == 0) fprintf(stderr, ">> %s\n", describe().c_str());
                    return 0;
                }
                decisions_from_cache = false;          // (incomplete entry: measure)
            }
        }
        DevBuf<real> W0, H0, T0;
        SyncStreamOnExit drain(stream);
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
        if (!decisions_pending()) {
            TunedDecisions d{fs_mode, gramx_mode, x_form, tuned_iters, {}, {}, x_cands, persist_note};
            for (int f = 0; f < kXForms; f++) { d.x_ms[f] = x_ms[f]; d.x_ms_all[f] = x_ms_all[f]; }
            std::lock_guard<std::mutex> lk(decision_mu());
            decision_cache()[key] = d;
        }
        if (verbose && comm->rank == 0) fprintf(stderr, ">> %s\n", describe().c_str());
        return 0;
    }
    // what this session runs, in one line (trmf_session_describe; bench.py's config.parallelism)
    // which rows are cut into items (session_state.hpp "split rows"); empty when none are
    std::string describe_split() const {
        if (!longF.any() && !longX.any()) return "";
        char t[224];
        snprintf(t, sizeof t, "; split rows: F %zu rows in %u items (>= %u entries), X %zu rows in %u items (>= %u entries)", longF.rows.size(), longF.nitems,
                 longF.thresh, longX.rows.size(), longX.nitems, longX.thresh);
        return t;
    }
    std::string describe() const { return describe_forms() + describe_split(); }
    std::string describe_forms() const {
        char buf[640];
        if (comm->world <= 1) {
            snprintf(buf, sizeof buf, "1 rank; X-solve %s", generic ? "unfused; generic kernels for rank > 64 (Gram build, F-solve)"
                     : tile_TI <= 0 ? "unfused (AR tile + cached-Gram product per CG step)"
                     : persist_state == 1 ? "fused, one persistent kernel per solve" : persist_state == 0 ? "fused (not run yet)" : "fused, one launch per CG step");
            std::string d(buf);
            if (!generic && tile_TI > 0) { snprintf(buf, sizeof buf, "; %d tiles of %d timestamps, %d threads", nbt, tile_TI, tile_nth); d += buf; }
            return persist_note.empty() ? d : d + " (" + persist_note + ")";
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
        snprintf(buf, sizeof buf, "%d ranks; F rows %s%s; X-side Gram rows %s; %s%s%s%s; peer-to-peer transport %s%s%s; decided %s%d set-up iterations",
                 comm->world, fs_mode == kShardOff ? "replicated" : fs_mode == kShardOn ? "sharded + all-gather of H" : "sharded (undecided)",
                 (fs_mode == kShardOn && fchunks >= 2) ? " in overlapped chunks" : "",
                 !rep_gram ? "own timestamps only (never gathered)" : gramx_mode == kGramxReplicate ? "replicated" : gramx_mode == kGramxShard ? "sharded + all-gather of G" : "sharded (undecided)",
                 x.c_str(), meas.empty() ? "" : " [slowest rank's X phase: ", meas.c_str(), meas.empty() ? "" : "]",
                 p2p.on ? "available" : "unavailable", p2p.note.empty() ? "" : ": ", p2p.note.c_str(),
                 decisions_from_cache ? "by an earlier session of this process (cached), " : "in ", tuned_iters);
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
            ev_on = ev_period > 0 && (iter1 % ev_period) == 0;
            if (ev_valid.size() != (size_t)kEventRing) ev_valid.assign(kEventRing, 0);
            ev_valid[(iter1 - 1) % kEventRing] = ev_on ? 1 : 0;
            static const DeviceIterLog blank = [] { DeviceIterLog b; std::memset(&b, 0, sizeof b); b.normF = b.normX = b.normLV = -1; return b; }();
            const bool doF = period_H > 0 && (iter1 % period_H) == 0;
            const bool doX = period_W > 0 && (iter1 % period_W) == 0;
            const bool doL = period_Lag > 0 && (iter1 % period_Lag) == 0;
            // quiet runs: accept_kernel writes the whole record; otherwise a blank record is uploaded and the
            // phases fill it in (two small copies per iteration on the stream)
            const bool device_log = doX && !log_norms && !verbose;
            if (!device_log) TRMF_HIP_CHECK(hipMemcpyAsync(L, &blank, sizeof blank, hipMemcpyHostToDevice, stream));
            TRMF_EVREC(ev.f0, stream);
            if (doF) {
                if (full ? fsolve_full(ev) : fsolve(ev)) return kFail;
                if (log_norms || verbose) log_norm(H.p, (size_t)n * KP, &L->normF);
                if (verbose) fprintf(stderr, ">> iter %d F %g\n", iter1, host_double(&L->normF));
            } else {
                TRMF_EVREC(ev.fk0, stream);
                TRMF_EVREC(ev.fk1, stream);
            }
            TRMF_EVREC(ev.f1, stream);
            if (!doX) TRMF_EVREC(ev.xg1, stream);
            if (doX) {
                if (join_theta()) return kFail;              // the X-solve reads Theta (and writes W, which the Theta-solve reads)
                xg1_event = ev_on ? ev.xg1 : nullptr;
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
            TRMF_EVREC(ev.x1, stream);
            if (doL) {
                if (verbose) {
                    log_norm(theta.p, (size_t)nlag * k, &L->normLV);
                    fprintf(stderr, ">> iter %d LV(%d %d) %g\n", iter1, nlag, k, host_double(&L->normLV));
                }
                // not the last iteration of this call, nothing reads Theta in stream order, long enough to pay: under the next F-solve
                const bool under_f = nlag > 0 && it + 1 < iters && overlap_ok();
                if (under_f) {
                    if (join_theta()) return kFail;
                    TRMF_HIP_CHECK(hipEventRecord(theta_fork, stream));
                    TRMF_HIP_CHECK(hipStreamWaitEvent(aux_theta, theta_fork, 0));
                    if (theta_solve(aux_theta)) return kFail;
                    TRMF_EVREC(ev.lv1, aux_theta);
                    TRMF_HIP_CHECK(hipEventRecord(theta_done, aux_theta));
                    theta_pending = true;
                    continue;
                }
                if (join_theta() || theta_solve(stream)) return kFail;
                if (log_norms || verbose) log_norm(theta.p, (size_t)nlag * k, &L->normLV);
                if (verbose) fprintf(stderr, ">> iter %d LV %g\n", iter1, host_double(&L->normLV));
            }
            TRMF_EVREC(ev.lv1, stream);
        }
        return join_theta();
    }
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
    int mark() {
        if (sync()) return kFail;
        const size_t nw = (size_t)(T + 1) * KP, nh = (size_t)(n + 1) * KP, nt = (size_t)nlag * k;
        FillStreamScope fill(stream);
        if (markW.alloc(nw, false) || markH.alloc(nh, false) || markT.alloc(nt, false)) return kFail;
        TRMF_HIP_CHECK(hipMemcpyAsync(markW.p, W.p, nw * sizeof(real), hipMemcpyDeviceToDevice, stream));
        TRMF_HIP_CHECK(hipMemcpyAsync(markH.p, H.p, nh * sizeof(real), hipMemcpyDeviceToDevice, stream));
        if (nt) TRMF_HIP_CHECK(hipMemcpyAsync(markT.p, theta.p, nt * sizeof(real), hipMemcpyDeviceToDevice, stream));
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        mark_iter = iter;
        return 0;
    }
    int rewind() {
        if (mark_iter < 0) { set_error("rewind: no mark has been set"); return kFail; }
        const size_t nw = (size_t)(T + 1) * KP, nh = (size_t)(n + 1) * KP, nt = (size_t)nlag * k;
        if (markW.n != nw || markH.n != nh) { set_error("rewind: the session has grown since the mark (append_rows)"); return kFail; }
        if (sync()) return kFail;
        TRMF_HIP_CHECK(hipMemcpyAsync(W.p, markW.p, nw * sizeof(real), hipMemcpyDeviceToDevice, stream));
        TRMF_HIP_CHECK(hipMemcpyAsync(H.p, markH.p, nh * sizeof(real), hipMemcpyDeviceToDevice, stream));
        if (nt) TRMF_HIP_CHECK(hipMemcpyAsync(theta.p, markT.p, nt * sizeof(real), hipMemcpyDeviceToDevice, stream));
        iter = mark_iter;
        if (snap_iter >= 0 && take_snapshot()) return kFail;       // the recovery snapshot follows the rewound state
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
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
            o.ms_F = o.ms_X = o.ms_LV = o.ms_F_kernel = o.ms_X_gram = -1;
            if (ev_valid.size() != (size_t)kEventRing || !ev_valid[it0 % kEventRing]) continue;      // no phase events in that iteration (ev_period)
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
};

}  // namespace trmf
