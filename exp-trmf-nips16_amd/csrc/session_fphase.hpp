// session_fphase.hpp -- the F-solve (trmf.cpp:654-663 -> 369-397; full-observation form :299-351): kernel dispatch by rank
// and element type, row sharding across ranks with the all-gather of H, the overlapped chunked gather of large factors.
#pragma once

#include "session_transport.hpp"

namespace trmf {

struct SessionFPhase : SessionTransport {
    // ---- F-solve (trmf.cpp:654-663 -> 369-397) -------------------------------------------------------
    // ---- split rows: partial Grams of the items of the long rows in [rb, re) of one orientation (gram_kernels.hpp "split rows") ----
    template <int NT_, bool PAD_> void launch_gram_part(const LongRows &L, uint32_t lo, uint32_t hi, const uint32_t *idx, const real *val,
                                                        const real *X, uint32_t zero_row) {
        const uint32_t i0 = L.first[lo], i1 = L.first[hi];
        if (i1 <= i0) return;
        hipLaunchKernelGGL((gram_part_kernel<NT_, PAD_>), dim3((i1 - i0 + 3) / 4), dim3(256), 0, stream, idx, val, X, L.d_items.p, i0, i1,
                           part_slab.p, part_stride, zero_row);
        // each row's partials summed into its first item's slot (rows of one item: nothing to do -- the launch is skipped when all are)
        if (i1 - i0 > hi - lo) {
            const uint32_t bpr = (uint32_t)((part_stride * sizeof(real) / 16 + 255) / 256);
            hipLaunchKernelGGL(split_reduce_kernel, dim3(bpr * (hi - lo)), dim3(256), 0, stream, L.d_first.p, part_slab.p, part_stride, lo, hi, bpr);
        }
    }
    template <int NT_, int KMAX_> int launch_fsolve_mfma(uint32_t rb, uint32_t re) {
        const uint32_t rows = re - rb;
        if (rows == 0) return 0;
#if !defined(TRMF_F32)
        if (longF.any()) {
            uint32_t lo, hi;
            longF.range(rb, re, lo, hi);
            if (hi > lo) {
                launch_gram_part<NT_, false>(longF, lo, hi, Yc_idx.p, Yc_val.p, W.p, (uint32_t)T);
                hipLaunchKernelGGL((fsolve_mfma_long_kernel<NT_, KMAX_>), dim3((hi - lo + 3) / 4), dim3(256), 0, stream, split_view(longF, lo, hi), H.p, k, (real)lambdaI);
            }
        }
        hipLaunchKernelGGL((fsolve_mfma_kernel<NT_, KMAX_>), dim3((rows + 3) / 4), dim3(256), 0, stream,
                           Yc_ptr.p, Yc_idx.p, Yc_val.p, W.p, H.p, rb, re, k, (real)lambdaI, (uint32_t)T, longF.thresh - 1u);
#endif
        return 0;
    }
    template <int NT_, int KMAX_> int launch_fsolve_quad(uint32_t rb, uint32_t re) {
        const uint32_t rows = re - rb;
        if (rows == 0) return 0;
#if defined(TRMF_F32)
        if (longF.any()) {
            uint32_t lo, hi;
            longF.range(rb, re, lo, hi);
            if (hi > lo) {
                launch_gram_part<NT_, (KMAX_ <= kTile * NT_ - 8)>(longF, lo, hi, Yc_idx.p, Yc_val.p, W.p, (uint32_t)T);
                hipLaunchKernelGGL((fsolve_quad_long_kernel<NT_, KMAX_>), dim3((hi - lo + 15) / 16), dim3(256), 0, stream, split_view(longF, lo, hi), H.p, k, (real)lambdaI);
            }
        }
        if (longF.any()) { uint32_t lo, hi; longF.range(rb, re, lo, hi); if (hi - lo == rows) return 0; }     // every row of the range is split: nothing left for the row kernel
        const dim3 grid((rows + 15) / 16), block(256);
#define TRMF_LAUNCH_QUAD(ABL)                                                                          \
        hipLaunchKernelGGL((fsolve_quad_kernel<NT_, KMAX_, ABL>), grid, block, 0, stream, Yc_ptr.p,    \
                           Yc_idx.p, Yc_val.p, W.p, H.p, rb, re, k, (real)lambdaI, (uint32_t)T, longF.thresh - 1u)
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
            TRMF_EVREC(ev.fk0, stream);
            for (int c = 0; c < C; c++) {
                if (launch_fsolve_rows((uint32_t)mine[c], (uint32_t)mine[c + 1])) return kFail;
                if (c + 1 < C) TRMF_HIP_CHECK(hipEventRecord(ov_c[c], stream));
            }
            TRMF_EVREC(ev.fk1, stream);
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
        TRMF_EVREC(ev.fk0, stream);
        if (launch_fsolve_rows(rb, re)) return kFail;
        TRMF_EVREC(ev.fk1, stream);
        TRMF_HIP_CHECK(hipGetLastError());
        fs_calls++;
        if (replicate) return 0;                                    // every rank solved every row: nothing to gather
        if (measure) TRMF_HIP_CHECK(hipEventRecord(fs1, stream));
        if (gather_rows(H.p, fbounds, (size_t)KP * sizeof(real))) return kFail;
        if (measure) TRMF_HIP_CHECK(hipEventRecord(fs2, stream));
        return 0;
    }

    // ---- full-observation path (missing == 0): trmf.cpp:299-351 and 155-215 -----------------------------
    // sparse Y times a factor (full-observation path with a sparse Y): rows of the orientation `L` describes; its split rows go through
    // spmm_part_kernel (one wavefront per item) + spmm_reduce_kernel (item order), the others through the row kernel
    template <int NT_> void launch_spmm(const LongRows &L, const uint32_t *ptr, const uint32_t *idx, const real *val, const real *X,
                                        real *out, uint32_t rb, uint32_t re, uint32_t zero_row) {
        if (re <= rb) return;
        if (L.any()) {
            uint32_t lo, hi;
            L.range(rb, re, lo, hi);
            const uint32_t i0 = L.first[lo], i1 = L.first[hi];
            if (i1 > i0) {
                hipLaunchKernelGGL((spmm_part_kernel<NT_>), dim3((i1 - i0 + 3) / 4), dim3(256), 0, stream, idx, val, X, L.d_items.p, i0, i1, part_slab.p, zero_row);
                hipLaunchKernelGGL(spmm_reduce_kernel, dim3((unsigned)(((size_t)(hi - lo) * KP + 255) / 256)), dim3(256), 0, stream, L.d_rows.p, L.d_first.p,
                                   part_slab.p, out, lo, hi, KP);
            }
        }
        hipLaunchKernelGGL((spmm_rows_kernel<NT_>), dim3((re - rb + 3) / 4), dim3(256), 0, stream, ptr, idx, val, X, out, rb, re, zero_row, L.thresh);
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
                case 1: launch_spmm<1>(transposed ? longF : longX, ptr, idx, val, X, out, rb, re, zr); break;
                case 2: launch_spmm<2>(transposed ? longF : longX, ptr, idx, val, X, out, rb, re, zr); break;
                case 3: launch_spmm<3>(transposed ? longF : longX, ptr, idx, val, X, out, rb, re, zr); break;
                default: launch_spmm<4>(transposed ? longF : longX, ptr, idx, val, X, out, rb, re, zr); break;
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
    template <int NT_> void launch_small_gram(const real *A, int rows, int nb, hipStream_t stream) {
        hipLaunchKernelGGL((small_gram_mfma_kernel<NT_>), dim3(nb), dim3(256), 0, stream, A, rows, k, sgram_part.p);
    }
    int small_gram(const real *A, int rows, real lambda, real *GS, hipStream_t stream) {
        if (generic) {
            hipLaunchKernelGGL(small_gram_generic_kernel, dim3(k), dim3(256), 0, stream, A, rows, k, KP, NT, lambda, GS);
            return 0;
        }
        // one partial per wavefront (4 per workgroup), at least 64 rows each, kSmallGramBlocks slots in all
        const int nb = std::max(1, std::min(kSmallGramBlocks / 4, rows / 256));
        switch (NT) {
            case 1: launch_small_gram<1>(A, rows, nb, stream); break;
            case 2: launch_small_gram<2>(A, rows, nb, stream); break;
            case 3: launch_small_gram<3>(A, rows, nb, stream); break;
            default: launch_small_gram<4>(A, rows, nb, stream); break;
        }
        hipLaunchKernelGGL(small_gram_reduce_kernel, dim3((k * k + 3) / 4), dim3(256), 0, stream, sgram_part.p, nb * 4, k, lambda, GS);
        return 0;
    }
    int fsolve_full(PhaseEvents &ev) {
        const uint32_t rb = (uint32_t)fbounds[comm->rank], re = (uint32_t)fbounds[comm->rank + 1];
        TRMF_EVREC(ev.fk0, stream);
        // (W^T W + lambda I and its factor do not depend on Y^T W, but running them beside it on a second stream was measured and is
        // not faster: the fork / join events cost more than the ~60 us chain they would hide; profiles/r05_streams.txt)
        hipStream_t gs = stream;
        if (y_times_factor(true, W.p, Bf.p, dense ? 0u : rb, dense ? (uint32_t)n : re)) return kFail;   // Y^T W
        small_gram(W.p, T, (real)lambdaI, GSf.p, gs);                                                   // W^T W + lambda I
        if (re > rb && generic) {
            hipLaunchKernelGGL(chol_generic_kernel, dim3(1), dim3(256), 0, stream, GSf.p, Uf.p, k);
            const int nrows = (int)(re - rb);
            hipLaunchKernelGGL(solve_rows_generic_kernel, dim3(std::min(2048, nrows)), dim3(256), (size_t)k * sizeof(real), stream, Uf.p, Bf.p + (size_t)rb * KP,
                               H.p + (size_t)rb * KP, nrows, k, KP, NT);
        } else if (re > rb) {
            const size_t ulds = (size_t)k * k * sizeof(real);          // <= 32 KB
            if (test_env("TRMF_CHOL_WORKGROUP")) hipLaunchKernelGGL(chol_shared_kernel, dim3(1), dim3(256), ulds, gs, GSf.p, Uf.p, k);
            else switch (NT) {
                case 1: hipLaunchKernelGGL(chol_wave_kernel<1>, dim3(1), dim3(64), 0, gs, GSf.p, Uf.p, k); break;
                case 2: hipLaunchKernelGGL(chol_wave_kernel<2>, dim3(1), dim3(64), 0, gs, GSf.p, Uf.p, k); break;
                case 3: hipLaunchKernelGGL(chol_wave_kernel<3>, dim3(1), dim3(64), 0, gs, GSf.p, Uf.p, k); break;
                default: hipLaunchKernelGGL(chol_wave_kernel<4>, dim3(1), dim3(64), 0, gs, GSf.p, Uf.p, k); break;
            }
            const int nrows = (int)(re - rb), nblk = std::max(1, std::min(2048, (nrows + 3) / 4));
            hipLaunchKernelGGL(solve_rows_kernel, dim3(nblk), dim3(256), ulds, stream, Uf.p, Bf.p + (size_t)rb * KP,
                               H.p + (size_t)rb * KP, nrows, k, KP, NT);
        }
        TRMF_EVREC(ev.fk1, stream);
        TRMF_HIP_CHECK(hipGetLastError());
        return gather_rows(H.p, fbounds, (size_t)KP * sizeof(real));
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
