// generic_kernels.hpp -- both training paths for ranks the register-tiled kernels do not cover (64 < k <= 1024).
//
// The reference allocates a k x k scratch per thread for ANY k (trmf.cpp:362-365) and a grid search over the rank
// (python/trmf/trmf.py:331-346) may well walk past 64; the drop-in must compute there, not answer "[ERR MSG]" (VERDICT r3).
// These kernels are deliberately plain -- one workgroup per row, the k x k system accumulated in (L2-resident) global scratch,
// a workgroup-wide Cholesky -- and slow next to gram_kernels.hpp; they exist for coverage, not for the roofline.  Same
// arithmetic as the reference where it prescribes one: the Gram and the right-hand side accumulate in val_type, one observed
// entry after the other in CSR order (trmf.cpp:382-389: fused multiply-adds), upper triangle mirrored, lambda added to the
// diagonal afterwards (:390-394), posv('U') (rf_matrix.h:3008-3014) as a right-looking Cholesky + two substitutions.
// The X-solve of such ranks runs the unfused CG (ar_tile_kernel + apply_kernel<false>, any KP) on the Grams built here.
#pragma once

#include "cg_kernels.hpp"

namespace trmf {

constexpr int kMaxRankGeneric = 1024;    // the k x k scratch slots in HBM bound it (gram_generic_lds(k): ~37 KB of LDS at rank 1024), nothing else does
constexpr int kApplyThreadPerColumn = 256;   // apply_kernel: one thread per (timestamp, column) up to here, apply_wide_kernel beyond
constexpr int kGenChunk = 8;             // observed entries staged per pass
constexpr int kGenBlocks = 512;          // workgroups (and k x k scratch slots) of the F-solve

// SOLVE = true : F-solve of item rows [row_begin, row_end): h_i = (sum x x^T + lambda I)^-1 sum y x; empty rows untouched
//                (trmf.cpp:374).  `A` = scratch, (k*k) reals per workgroup; `out` = H (rows x KP, column-interleaved).
// SOLVE = false: X-side cache of timestamps [row_begin, row_end): G_i = sum h h^T (full k x k, `gs` elements apart) into `A`,
//                b_i = sum y h into `out` = Bv (logical column order).
template <bool SOLVE>
__global__ __launch_bounds__(256) void gram_generic_kernel(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
                                                           const real *__restrict__ val, const real *__restrict__ X,
                                                           uint32_t row_begin, uint32_t row_end, int k, int KP, int NT,
                                                           real lambda, real *__restrict__ A_base, size_t gs,
                                                           real *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_smem[];
    real *xs = reinterpret_cast<real *>(gen_smem);          // [kGenChunk][k] gathered factor rows, logical column order
    real *ys = xs + (size_t)kGenChunk * k;                    // [kGenChunk] observation values
    real *bl = ys + kGenChunk;                                // [k] right-hand side / solution
    const int tid = threadIdx.x;
    for (uint32_t row = row_begin + blockIdx.x; row < row_end; row += gridDim.x) {
        const uint32_t p0 = ptr[row], p1 = ptr[row + 1];
        if (SOLVE && p1 == p0) continue;                      // uniform: no observation of this item
        real *A = SOLVE ? A_base + (size_t)blockIdx.x * k * k : A_base + (size_t)row * gs;
        for (int e = tid; e < k * k; e += 256) A[e] = 0;
        for (int t = tid; t < k; t += 256) bl[t] = 0;
        __syncthreads();
        for (uint32_t c0 = p0; c0 < p1; c0 += kGenChunk) {
            const int cnt = (int)min((uint32_t)kGenChunk, p1 - c0);
            for (int e = tid; e < cnt * k; e += 256) {
                const int c = e / k, t = e - c * k;
                xs[c * k + t] = X[(size_t)idx[c0 + c] * KP + colpos(t, NT)];
            }
            if (tid < cnt) ys[tid] = val[c0 + tid];
            __syncthreads();
            for (int e = tid; e < k * k; e += 256) {          // upper triangle, entry by entry in CSR order
                const int a = e / k, b = e - a * k;
                if (a > b) continue;
                real acc = A[e];
                for (int c = 0; c < cnt; c++) acc = fma(xs[c * k + a], xs[c * k + b], acc);
                A[e] = acc;
            }
            for (int t = tid; t < k; t += 256) {
                real acc = bl[t];
                for (int c = 0; c < cnt; c++) acc = fma(ys[c], xs[c * k + t], acc);
                bl[t] = acc;
            }
            __syncthreads();
        }
        if (!SOLVE) {
            for (int e = tid; e < k * k; e += 256) {          // mirror
                const int a = e / k, b = e - a * k;
                if (a > b) A[e] = A[b * k + a];
            }
            for (int t = tid; t < k; t += 256) out[(size_t)row * KP + t] = bl[t];
            __syncthreads();
            continue;
        }
        for (int t = tid; t < k; t += 256) A[t * k + t] += lambda;
        __syncthreads();
        // upper Cholesky A = U^T U in place (right-looking, one column per thread), then U^T z = b, U x = z
        for (int j = 0; j < k; j++) {
            const real ajj = sqrt(A[j * k + j]);
            __syncthreads();
            for (int c = j + tid; c < k; c += 256) A[j * k + c] = (c == j) ? ajj : A[j * k + c] / ajj;
            __syncthreads();
            for (int c = j + 1 + tid; c < k; c += 256) {
                const real ujc = A[j * k + c];
                for (int s = j + 1; s <= c; s++) A[s * k + c] -= A[j * k + s] * ujc;
            }
            __syncthreads();
        }
        for (int q = 0; q < k; q++) {
            if (tid == 0) bl[q] = bl[q] / A[q * k + q];
            __syncthreads();
            for (int i = q + 1 + tid; i < k; i += 256) bl[i] -= A[q * k + i] * bl[q];
            __syncthreads();
        }
        for (int q = k - 1; q >= 0; q--) {
            if (tid == 0) bl[q] = bl[q] / A[q * k + q];
            __syncthreads();
            for (int i = tid; i < q; i += 256) bl[i] -= A[i * k + q] * bl[q];
            __syncthreads();
        }
        for (int t = tid; t < k; t += 256) out[(size_t)row * KP + colpos(t, NT)] = bl[t];
        __syncthreads();
    }
}
inline size_t gram_generic_lds(int k) { return ((size_t)kGenChunk * k + kGenChunk + k) * sizeof(real); }

// out = base + G_i v (- b_i) for ranks above 256 (apply_kernel gives a thread to every column of a timestamp): one workgroup per
// timestamp (grid-stride), the operand row in LDS, a thread walks columns t, t + 256, ...  Same interface, same products in the
// same order s = 0 .. k-1, same partial-sum slots as apply_kernel<false>.
__global__ __launch_bounds__(256) void apply_wide_kernel(XParams p, const XState *__restrict__ st, int cg_it,
                                                         const real *__restrict__ v, const real *__restrict__ rvec,
                                                         const real *__restrict__ base, const real *__restrict__ G,
                                                         const real *__restrict__ Bv, int minus_b, real *__restrict__ out, int dot_mode,
                                                         double *__restrict__ Pdot, int row0, int nrows, int slot0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_smem[];
    __shared__ double smem[256];
    real *vs = reinterpret_cast<real *>(wide_smem);          // [k] operand row, logical column order
    const bool cg = cg_it >= 0;
    if (cg && st->stop_it <= cg_it) return;
    const int k = p.k, KP = p.KP, NT = p.NT;
    double dot = 0, lq = 0, rhd = 0, hh = 0;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int i = row0 + r;
        __syncthreads();
        for (int t = threadIdx.x; t < k; t += 256) vs[t] = v[(size_t)i * KP + colpos(t, NT)];
        __syncthreads();
        const real *Gi = G + (size_t)i * p.gstride;
        for (int t = threadIdx.x; t < k; t += 256) {
            const int tp = colpos(t, NT);
            double acc = 0;
            for (int s = 0; s < k; s++) acc += (double)Gi[(size_t)s * k + t] * (double)vs[s];
            const real x = vs[t];
            if (minus_b) {
                const double bb = (double)Bv[(size_t)i * KP + t];
                lq += (double)x * (acc - 2.0 * bb);
                acc -= bb;
            }
            const real o = (real)((double)base[(size_t)i * KP + tp] + acc);
            out[(size_t)i * KP + tp] = o;
            dot += (double)(dot_mode ? x : o) * (double)o;
            if (cg) { rhd += (double)rvec[(size_t)i * KP + tp] * (double)o; hh += (double)o * (double)o; }
        }
    }
    if (cg) {
        block_allsum3(dot, rhd, hh, smem);
        if (threadIdx.x == 0) {
            double *Po = Pdot + (size_t)(P_CG0 - P_DOT + 3 * (cg_it & 1)) * p.pstride + slot0 + blockIdx.x;
            Po[0] = dot; Po[(size_t)p.pstride] = rhd; Po[2 * (size_t)p.pstride] = hh;
        }
        return;
    }
    dot = block_allsum(dot, smem);
    lq = block_allsum(lq, smem);
    if (threadIdx.x == 0) { Pdot[slot0 + blockIdx.x] = dot; Pdot[(P_LQ - P_DOT) * (size_t)p.pstride + slot0 + blockIdx.x] = lq; }
}

// ---- full-observation path (missing == 0; trmf.cpp:299-351, 155-215) for ranks above 64 -------------------------------------------
// out[row][t] = sum_e val[e] * X[idx[e]][t]   (sparse Y times a factor; `out` rows x KP in LOGICAL column order like spmm_rows_kernel's;
// val_type multiply-adds in entry order: the reference's gmat_x_dmat loop)
__global__ __launch_bounds__(256) void spmm_generic_kernel(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
                                                           const real *__restrict__ val, const real *__restrict__ X, real *__restrict__ out,
                                                           uint32_t row_begin, uint32_t row_end, int k, int KP, int NT) {
    for (uint32_t row = row_begin + blockIdx.x; row < row_end; row += gridDim.x) {
        const uint32_t p0 = ptr[row], p1 = ptr[row + 1];
        for (int t = threadIdx.x; t < k; t += 256) {
            const int tp = colpos(t, NT);
            real acc = 0;
            for (uint32_t e = p0; e < p1; e++) acc = fma(val[e], X[(size_t)idx[e] * KP + tp], acc);
            out[(size_t)row * KP + t] = acc;
        }
    }
}
// C[m][t] = sum_j A[j][m] * B[j][t]   (dense A: K x M row-major, B: K x KP factor; the reference's gemm, rf_matrix.h:3182-3216;
// products widened, sum in double, rounded once)
__global__ __launch_bounds__(256) void dense_tn_generic_kernel(const real *__restrict__ A, int K, int M, const real *__restrict__ B,
                                                               real *__restrict__ out, int k, int KP, int NT) {
    for (int m = blockIdx.x; m < M; m += gridDim.x)
        for (int t = threadIdx.x; t < k; t += 256) {
            const int tp = colpos(t, NT);
            double acc = 0;
            for (int j = 0; j < K; j++) acc += (double)A[(size_t)j * M + m] * (double)B[(size_t)j * KP + tp];
            out[(size_t)m * KP + t] = (real)acc;
        }
}
// GS = A^T A (+ lambda on the diagonal, trmf.cpp:322-324): workgroup a, thread b
__global__ __launch_bounds__(256) void small_gram_generic_kernel(const real *__restrict__ A, int rows, int k, int KP, int NT, real lambda,
                                                                 real *__restrict__ GS) {
    const int a = blockIdx.x, ap = colpos(a, NT);
    for (int b = threadIdx.x; b < k; b += 256) {
        const int bp = colpos(b, NT);
        double acc = 0;
        for (int r = 0; r < rows; r++) acc += (double)A[(size_t)r * KP + ap] * (double)A[(size_t)r * KP + bp];
        real v = (real)acc;
        if (a == b) v += lambda;
        GS[(size_t)a * k + b] = v;
    }
}
// upper Cholesky of ONE k x k matrix in global memory (posv 'U' of trmf.cpp:333), one workgroup
__global__ __launch_bounds__(256) void chol_generic_kernel(const real *__restrict__ GS, real *__restrict__ U, int k) {
    const int tid = threadIdx.x;
    for (int e = tid; e < k * k; e += 256) U[e] = GS[e];
    __syncthreads();
    for (int j = 0; j < k; j++) {
        const real ajj = sqrt(U[j * k + j]);
        __syncthreads();
        for (int c = j + tid; c < k; c += 256) U[j * k + c] = (c == j) ? ajj : U[j * k + c] / ajj;
        __syncthreads();
        for (int c = j + 1 + tid; c < k; c += 256) {
            const real ujc = U[j * k + c];
            for (int s = j + 1; s <= c; s++) U[s * k + c] -= U[j * k + s] * ujc;
        }
        __syncthreads();
    }
}
// H[i][:] = (U^T U)^-1 b_i: a workgroup per right-hand side (grid-stride), unknowns in LDS, column-oriented substitutions
__global__ __launch_bounds__(256) void solve_rows_generic_kernel(const real *__restrict__ U, const real *__restrict__ Brows, real *__restrict__ out,
                                                                 int rows, int k, int KP, int NT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sr_smem[];
    real *x = reinterpret_cast<real *>(sr_smem);
    const int tid = threadIdx.x;
    for (int i = blockIdx.x; i < rows; i += gridDim.x) {
        __syncthreads();
        for (int t = tid; t < k; t += 256) x[t] = Brows[(size_t)i * KP + t];
        __syncthreads();
        for (int q = 0; q < k; q++) {
            if (tid == 0) x[q] = x[q] / U[q * k + q];
            __syncthreads();
            for (int r = q + 1 + tid; r < k; r += 256) x[r] -= U[q * k + r] * x[q];
            __syncthreads();
        }
        for (int q = k - 1; q >= 0; q--) {
            if (tid == 0) x[q] = x[q] / U[q * k + q];
            __syncthreads();
            for (int r = tid; r < q; r += 256) x[r] -= U[r * k + q] * x[q];
            __syncthreads();
        }
        for (int t = tid; t < k; t += 256) out[(size_t)i * KP + colpos(t, NT)] = x[t];
    }
}

// squared residuals of one timestamp row (trmf_session_objective): sum (y - w.h)^2, products in val_type, sum in double
__global__ __launch_bounds__(256) void loss_generic_kernel(const uint32_t *__restrict__ ptr, const uint32_t *__restrict__ idx,
                                                           const real *__restrict__ val, const real *__restrict__ Hf,
                                                           const real *__restrict__ W, double *__restrict__ lossrow,
                                                           uint32_t row_begin, uint32_t row_end, int KP) {
    __shared__ double smem[256];
    const uint32_t row = row_begin + blockIdx.x;
    if (row >= row_end) return;
    const real *w = W + (size_t)row * KP;
    double l = 0;
    for (uint32_t e = ptr[row] + threadIdx.x; e < ptr[row + 1]; e += 256) {
        const real *h = Hf + (size_t)idx[e] * KP;
        real dot = 0;
        for (int t = 0; t < KP; t++) dot = fma(w[t], h[t], dot);             // pad columns are zero on both sides
        const double r = (double)val[e] - (double)dot;
        l += r * r;
    }
    l = block_allsum(l, smem);
    if (threadIdx.x == 0) lossrow[row] = l;
}

}  // namespace trmf
