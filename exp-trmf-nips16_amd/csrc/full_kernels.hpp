// full_kernels.hpp -- the full-observation path (missing == 0): every entry of Y is observed, zeros
// included.  Reference: l2r_ls_fY_IX_chol (trmf.cpp:299-351) for the F-solve and arr_ls_fY_IX
// (trmf.cpp:155-215) for the X-solve objective.  With Omega = everything the per-row Gram is the same
// k x k matrix for every row, so
//     F-solve:  H = (Y^T W) (W^T W + lambda I)^-1      one Cholesky, n pairs of triangular solves
//     X-solve:  grad = base + W (H^T H) - Y H,  Hv = base + S (H^T H)   (shared Gram, stride 0)
// Y may be sparse (zeros are observed zeros; products via the CSR/CSC streams) or dense (both
// orientations are kept in HBM, like the sparse form keeps CSR and CSC).
#pragma once

#include "cg_kernels.hpp"
#include "common.hpp"
#include "gram_kernels.hpp"

namespace trmf {

// ---- out[row][:] = sum_j val * X[idx][:]  (sparse Y times a factor): one wavefront per row ----------
// `out` is rows x KP in LOGICAL column order (it is a right-hand side, not a factor).
template <int NT>
__global__ __launch_bounds__(256) void spmm_rows_kernel(const uint32_t *__restrict__ ptr,
                                                        const uint32_t *__restrict__ idx,
                                                        const real *__restrict__ val,
                                                        const real *__restrict__ X,
                                                        real *__restrict__ out, uint32_t row_begin,
                                                        uint32_t row_end, uint32_t zero_row) {
    constexpr int KP = kTile * NT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    const uint32_t row = row_begin + blockIdx.x * 4u + (uint32_t)wave;
    if (row >= row_end) return;
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ptr[row]);
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ptr[row + 1]);
    GramState<NT> st;
    st.clear();
    real nowq[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) nowq[q] = 0;
    if (p1 > p0)
        gram_ring<NT, kRingDepth, false, false>(st, idx, val, X, zero_row, 4u, lane, nowq, GramDesc{p0, p1, 0},
                                                SingleRowStream{4u * kRingDepth}, [](int) {});
#pragma unroll
    for (int q = 0; q < NT; q++) {
        real v = st.b[q];
        v += __shfl_xor(v, 16, kWave);
        v += __shfl_xor(v, 32, kWave);
        if (g == 0) out[(size_t)row * KP + kTile * q + c] = v;
    }
}

// ---- C[m][:] = sum_j A[j][m] * B[j][:]  (dense A: K x M row-major; B: factor, K x KP interleaved) -----
// One thread per output row m, KP accumulators; the K range is split over gridDim.y chunks whose
// partials are summed in fixed order by dense_tn_reduce_kernel.  B rows are wave-uniform (scalar loads).
template <int NT>
__global__ __launch_bounds__(256) void dense_tn_kernel(const real *__restrict__ A, int K, int M,
                                                       const real *__restrict__ B,
                                                       double *__restrict__ part) {
    constexpr int KP = kTile * NT;
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int nchunk = gridDim.y, ch = blockIdx.y;
    const int j0 = (int)((long long)K * ch / nchunk), j1 = (int)((long long)K * (ch + 1) / nchunk);
    double acc[KP];
#pragma unroll
    for (int t = 0; t < KP; t++) acc[t] = 0;
    if (m < M) {
        for (int j = j0; j < j1; j++) {
            const double a = (double)A[(size_t)j * M + m];
            const real *brow = B + (size_t)j * KP;
#pragma unroll
            for (int t = 0; t < KP; t++) acc[t] += a * (double)brow[t];
        }
        double *dst = part + ((size_t)ch * M + m) * KP;
#pragma unroll
        for (int t = 0; t < KP; t++) dst[t] = acc[t];
    }
}
// out (M x KP, LOGICAL columns) = sum over chunks; positions of B's interleaved layout mapped back.
// One wavefront per output element: lanes stride over the chunks, then a fixed-order butterfly -- the
// chunk count goes up to ~1000 for tall-skinny products (few output rows, long contraction).
__global__ __launch_bounds__(256) void dense_tn_reduce_kernel(const double *__restrict__ part, int nchunk,
                                                              int M, int KP, int NT, int k,
                                                              real *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= (size_t)M * KP) return;
    const int m = (int)(e / KP), tp = (int)(e - (size_t)m * KP);
    double acc = 0;
    for (int ch = lane; ch < nchunk; ch += 64) acc += part[((size_t)ch * M + m) * KP + tp];
    acc = wave_butterfly_sum(acc);
    const int t = collog(tp, NT);
    if (lane == 0) out[(size_t)m * KP + t] = (t < k) ? (real)acc : real(0);
}

// ---- small Gram: GS (k x k, logical) = A^T A (+ lambda I) over the rows of a factor ------------------
// grid = nblk workgroups, each over a contiguous row chunk staged kSgRows rows at a time in LDS; thread e
// owns entries e, e+256, ... of the k x k result and adds the staged rows in order.  Partials are reduced
// in fixed order by small_gram_reduce_kernel.
constexpr int kSgRows = 32;
__global__ __launch_bounds__(256) void small_gram_kernel(const real *__restrict__ A, int rows, int KP,
                                                         int NT, int k, double *__restrict__ part) {
    __shared__ real srow[kSgRows][64];
    const int nblk = gridDim.x;
    const int r0 = (int)((long long)rows * blockIdx.x / nblk), r1 = (int)((long long)rows * (blockIdx.x + 1) / nblk);
    double acc[16];
#pragma unroll
    for (int u = 0; u < 16; u++) acc[u] = 0;
    for (int rb = r0; rb < r1; rb += kSgRows) {
        const int nr = min(kSgRows, r1 - rb);
        __syncthreads();
        for (int x = threadIdx.x; x < nr * k; x += 256) {
            const int rr = x / k, t = x - rr * k;
            srow[rr][t] = A[(size_t)(rb + rr) * KP + colpos(t, NT)];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int e = threadIdx.x + 256 * u;
            if (e < k * k) {
                const int s = e / k, t = e % k;
                double a2 = acc[u];
                for (int rr = 0; rr < nr; rr++) a2 += (double)srow[rr][s] * (double)srow[rr][t];
                acc[u] = a2;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int e = threadIdx.x + 256 * u;
        if (e < k * k) part[(size_t)blockIdx.x * k * k + e] = acc[u];
    }
}
// one wavefront per entry: lanes stride over the workgroup partials, fixed-order butterfly
__global__ __launch_bounds__(256) void small_gram_reduce_kernel(const double *__restrict__ part, int nblk,
                                                                int k, real lambda, real *__restrict__ GS) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= k * k) return;
    double acc = 0;
    for (int b = lane; b < nblk; b += 64) acc += part[(size_t)b * k * k + e];
    acc = wave_butterfly_sum(acc);
    real v = (real)acc;
    if (e / k == e % k) v += lambda;                                         // trmf.cpp:322-324
    if (lane == 0) GS[e] = v;
}

// ---- shared-matrix solve: H[i][:] = GS^-1 b_i for every row (posv with n right-hand sides) -------------
// Each workgroup first factorises the k x k matrix in LDS (upper Cholesky, val_type, as posv 'U'),
// then every thread solves one row by forward / backward substitution; its vector lives in an LDS
// column (conflict-free).  The workgroup size (= rows per workgroup, 64..256) is chosen by the host so
// that the dynamic LDS, solve_shared_lds_bytes(), stays within the 64 KB every launch may use without a
// function attribute: 256 rows up to k = 27 (fp64) / 51 (fp32), 64 rows at k = 64 fp64.
__host__ __device__ inline size_t solve_shared_lds_bytes(int k, int rows_per_block) {
    return ((size_t)k * k + (size_t)rows_per_block * k) * sizeof(real);
}
inline int solve_shared_rows_per_block(int k) {
    for (int r : {256, 128, 64})
        if (solve_shared_lds_bytes(k, r) <= 64 * 1024) return r;
    return 0;       // cannot happen for k <= kMaxRank
}
__global__ __launch_bounds__(256) void solve_shared_kernel(const real *__restrict__ GS,
                                                           const real *__restrict__ Brows,
                                                           real *__restrict__ out, int rows, int k,
                                                           int KP, int NT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_raw[];
    const int nthr = (int)blockDim.x;                    // rows per workgroup
    real *U = reinterpret_cast<real *>(ss_raw);          // k x k
    real *xs = U + k * k;                                // k x nthr, xs[p * nthr + tid]
    for (int e = threadIdx.x; e < k * k; e += nthr) U[e] = GS[e];
    __syncthreads();
    for (int j = 0; j < k; j++) {                        // same loop as theta_solve_kernel
        const real ajj = sqrt(U[j * k + j]);
        __syncthreads();
        for (int c = j + threadIdx.x; c < k; c += nthr) U[j * k + c] = (c == j) ? ajj : U[j * k + c] / ajj;
        __syncthreads();
        for (int s = j + 1; s < k; s++) {
            const real ujs = U[j * k + s];
            for (int c = s + threadIdx.x; c < k; c += nthr) U[s * k + c] -= ujs * U[j * k + c];
        }
        __syncthreads();
    }
    const int i = blockIdx.x * nthr + threadIdx.x;
    if (i >= rows) return;
    real *x = xs + threadIdx.x;
    for (int p = 0; p < k; p++) x[p * nthr] = Brows[(size_t)i * KP + p];
    for (int p = 0; p < k; p++) {                        // U^T z = b
        real s = x[p * nthr];
        for (int q = 0; q < p; q++) s -= U[q * k + p] * x[q * nthr];
        x[p * nthr] = s / U[p * k + p];
    }
    for (int p = k - 1; p >= 0; p--) {                   // U x = z
        real s = x[p * nthr];
        for (int q = p + 1; q < k; q++) s -= U[p * k + q] * x[q * nthr];
        x[p * nthr] = s / U[p * k + p];
    }
    for (int p = 0; p < k; p++) out[(size_t)i * KP + colpos(p, NT)] = x[p * nthr];
}

}  // namespace trmf
