// theta_kernels.hpp -- lag-weight (Theta) update on device: trmf.cpp:447-484
// (l2r_autoregressive_solver::{lagged_inner_product, solve}).
//
// Per latent dimension t: |L| x |L| Gram of lagged inner products of the series W[:,t] over
// i in [midx, T) with double accumulators, + lambdaLag on the diagonal, Cholesky solve in val_type.
// Kept on device so that the ALS loop never round-trips W to the host.
#pragma once

#include "common.hpp"

namespace trmf {

constexpr int kThetaChunk = 512;    // timestamps per workgroup of theta_gram_kernel (2048 -> 512: 92 us -> ~40 us at config 3)
constexpr int kMaxLags = 128;

// pair index p in [0, npairs): p < nlag -> rhs entry y[p] = <s_i, s_{i-L_p}>;
// otherwise the upper-triangle entry (a, b), a <= b, in row-major order.
__device__ __forceinline__ void theta_decode_pair(int p, int nlag, int &a, int &b, bool &rhs) {
    if (p < nlag) { rhs = true; a = p; b = p; return; }
    rhs = false;
    int q = p - nlag, row = 0, len = nlag;
    while (q >= len) { q -= len; row++; len--; }
    a = row; b = row + q;
}

// grid (k, nchunk), dynamic LDS = (kThetaChunk + midx) * sizeof(real)
__global__ __launch_bounds__(256) void theta_gram_kernel(const real *__restrict__ W, int T, int KP,
                                                         const uint32_t *__restrict__ lag_set,
                                                         int nlag, int midx, int npairs,
                                                         double *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    real *series = reinterpret_cast<real *>(smem_raw);
    const int t = blockIdx.x, ch = blockIdx.y, nchunk = gridDim.y;
    const int i0 = midx + ch * kThetaChunk;
    const int i1 = min(T, i0 + kThetaChunk);
    const int lo = i0 - midx;                           // first timestamp staged
    const int tp = colpos(t, KP / kTile);                 // column-interleaved factor layout
    for (int i = lo + threadIdx.x; i < i1; i += 256) series[i - lo] = W[(size_t)i * KP + tp];
    __syncthreads();
    for (int p = threadIdx.x; p < npairs; p += 256) {
        int a, b; bool rhs;
        theta_decode_pair(p, nlag, a, b, rhs);
        const int la = rhs ? 0 : (int)lag_set[a];
        const int lb = (int)lag_set[b];
        double acc = 0;
        for (int i = i0; i < i1; i++) {
            const real prod = series[i - la - lo] * series[i - lb - lo];   // val_type product
            acc += (double)prod;                                           // double accumulate
        }
        part[((size_t)t * nchunk + ch) * npairs + p] = acc;
    }
}

// one wavefront per latent dimension; dynamic LDS = (nlag*nlag + nlag) * sizeof(real)
__global__ __launch_bounds__(64) void theta_solve_kernel(const double *__restrict__ part, int nchunk,
                                                         int nlag, int npairs, double lambdaLag,
                                                         real *__restrict__ theta) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    real *A = reinterpret_cast<real *>(smem_raw);       // nlag x nlag, element (i,j) at A[i*nlag+j]
    real *y = A + nlag * nlag;
    const int t = blockIdx.x, lane = threadIdx.x;
    for (int p = lane; p < npairs; p += 64) {
        double acc = 0;
#pragma unroll 8
        for (int ch = 0; ch < nchunk; ch++) acc += part[((size_t)t * nchunk + ch) * npairs + p];   // fixed order
        int a, b; bool rhs;
        theta_decode_pair(p, nlag, a, b, rhs);
        if (rhs) y[a] = (real)acc;
        else {
            real v = (real)acc;                                              // trmf.cpp:473
            if (a == b) v = (real)((double)v + lambdaLag);                   // trmf.cpp:480
            A[a * nlag + b] = v;
            A[b * nlag + a] = v;
        }
    }
    __syncthreads();
    // upper Cholesky A = U^T U, row by row (posv 'U', rf_matrix.h:3008-3014).  One wavefront: LDS accesses
    // retire in program order, the barriers are wave-local.  The trailing update of step j runs over all
    // (s, c) pairs at once (s > j, c >= s), 64 per pass -- same operations on every element, in the same j
    // order, as the row-by-row loop.
    for (int j = 0; j < nlag; j++) {
        const real ajj = sqrt(A[j * nlag + j]);
        __syncthreads();
        for (int c = j + lane; c < nlag; c += 64) A[j * nlag + c] = (c == j) ? ajj : A[j * nlag + c] / ajj;
        __syncthreads();
        const int m = nlag - 1 - j;                     // trailing dimension
        for (int e = lane; e < m * m; e += 64) {
            const int s = j + 1 + e / m, c = j + 1 + e % m;
            if (c >= s) A[s * nlag + c] -= A[j * nlag + s] * A[j * nlag + c];
        }
        __syncthreads();
    }
    // substitutions, column-oriented: after step q every remaining unknown has had its U(.,.)*z_q term removed:
    // |L| steps of one parallel pass instead of |L|^2/2 dependent LDS round trips on a single lane.  Forward:
    // the same subtractions in the same order as the row-oriented loop; backward: the terms of a row are
    // subtracted in descending instead of ascending q (a last-bit difference in Theta)
    for (int q = 0; q < nlag; q++) {                    // U^T z = y
        if (lane == 0) y[q] = y[q] / A[q * nlag + q];
        __syncthreads();
        for (int i = q + 1 + lane; i < nlag; i += 64) y[i] -= A[q * nlag + i] * y[q];
        __syncthreads();
    }
    for (int q = nlag - 1; q >= 0; q--) {               // U x = z
        if (lane == 0) y[q] = y[q] / A[q * nlag + q];
        __syncthreads();
        for (int i = lane; i < q; i += 64) y[i] -= A[i * nlag + q] * y[q];
        __syncthreads();
    }
    for (int a = lane; a < nlag; a += 64) theta[(size_t)t * nlag + a] = y[a];
}

}  // namespace trmf
