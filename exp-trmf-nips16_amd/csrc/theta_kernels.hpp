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
constexpr int kMaxLags = 1024;      // beyond ~140 (fp64) / 200 (fp32) lags the |L| x |L| systems of theta_solve_kernel no longer fit LDS: global scratch

// pair index p in [0, npairs): p < nlag -> rhs entry y[p] = <s_i, s_{i-L_p}>;
// otherwise the upper-triangle entry (a, b), a <= b, in row-major order.
__device__ __forceinline__ void theta_decode_pair(int p, int nlag, int &a, int &b, bool &rhs) {
    if (p < nlag) { rhs = true; a = p; b = p; return; }
    rhs = false;
    int q = p - nlag, row = 0, len = nlag;
    while (q >= len) { q -= len; row++; len--; }
    a = row; b = row + q;
}

// grid (k, nchunk), 256 threads, dynamic LDS = theta_gram_lds_bytes(midx).
// Register tiling: a thread owns a 4 x 4 block of lag pairs (a in ta, b in tb, tb >= ta; the "zero lag" series s_i
// against a block of four lags for the right-hand side) and, when there are fewer blocks than threads, one of S time
// slices of the chunk -- per timestamp it reads 8 series values from LDS for 16 products (0.5 reads per product instead
// of 2).  Products are rounded to val_type, sums are double (trmf.cpp:447-453); the S slice sums of a block are added
// in fixed order through LDS.
constexpr int kThetaTile = 4;
constexpr int kThetaRows = 4;     // consecutive timestamps per pass of the sliding-window path
__host__ __device__ inline size_t theta_gram_lds_bytes(int midx) {
    return ((size_t)(kThetaChunk + midx + kThetaRows) * sizeof(real) + 15) / 16 * 16 + (size_t)256 * 16 * sizeof(double);   // + window slack
}
__global__ __launch_bounds__(256) void theta_gram_kernel(const real *__restrict__ W, int T, int KP,
                                                         const uint32_t *__restrict__ lag_set,
                                                         int nlag, int midx, int npairs,
                                                         double *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    real *series = reinterpret_cast<real *>(smem_raw);
    double *red = reinterpret_cast<double *>(smem_raw + (((size_t)(kThetaChunk + midx + kThetaRows) * sizeof(real) + 15) / 16 * 16));
    const int t = blockIdx.x, ch = blockIdx.y, nchunk = gridDim.y;
    const int i0 = midx + ch * kThetaChunk;
    const int i1 = min(T, i0 + kThetaChunk);
    const int lo = i0 - midx;                           // first timestamp staged
    const int tp = colpos(t, KP / kTile);                 // column-interleaved factor layout
    for (int i = lo + threadIdx.x; i < i1; i += 256) series[i - lo] = W[(size_t)i * KP + tp];
    __syncthreads();
    const int NA = (nlag + kThetaTile - 1) / kThetaTile;
    const int ntiles = NA * (NA + 1) / 2 + NA;          // upper-triangle blocks, then the rhs blocks
    const int S = max(1, 256 / ntiles);                 // time slices per block of pairs
    double *out = part + ((size_t)t * nchunk + ch) * npairs;
    for (int base = 0; base < ntiles; base += 256 / S) {
        const int tile = base + (int)threadIdx.x / S, slice = (int)threadIdx.x % S;
        const bool live = tile < ntiles && (int)threadIdx.x < (256 / S) * S;
        int ta = 0, tb = 0;
        bool rhs = false;
        if (live) {
            if (tile >= NA * (NA + 1) / 2) { rhs = true; tb = tile - NA * (NA + 1) / 2; }
            else { int q = tile, len = NA; while (q >= len) { q -= len; ta++; len--; } tb = ta + q; }
        }
        int la[kThetaTile], lb[kThetaTile];
#pragma unroll
        for (int u = 0; u < kThetaTile; u++) {
            la[u] = rhs ? 0 : (int)lag_set[min(kThetaTile * ta + u, nlag - 1)];
            lb[u] = (int)lag_set[min(kThetaTile * tb + u, nlag - 1)];
        }
        double acc[kThetaTile][kThetaTile];
#pragma unroll
        for (int u = 0; u < kThetaTile; u++)
#pragma unroll
            for (int v = 0; v < kThetaTile; v++) acc[u][v] = 0;
        // Blocks whose four lags are consecutive integers on both sides (all of the paper's lag set, every 1..|L| set):
        // a thread then takes kThetaRows consecutive timestamps per pass -- the operands of neighbouring rows and lags
        // overlap, so two windows of kThetaRows + 3 LDS reads feed 16 kThetaRows products (0.22 reads per product
        // instead of 0.5; the kernel is LDS-bound).  Other blocks take the general loop below.
        bool runs = live;
#pragma unroll
        for (int u = 1; u < kThetaTile; u++)
            runs = runs && (rhs || la[u] == la[0] + u) && lb[u] == lb[0] + u;
        if (runs) {
            constexpr int R = kThetaRows, Wn = R + kThetaTile - 1;
            for (int i = i0 + slice * R; i < i1; i += S * R) {
                real wa[Wn], wb[Wn];
                const real *pa = series + (i - la[0] - (kThetaTile - 1) - lo), *pb = series + (i - lb[0] - (kThetaTile - 1) - lo);
#pragma unroll
                for (int m = 0; m < Wn; m++) { wb[m] = pb[m]; wa[m] = rhs ? series[i - lo + min(m, R - 1)] : pa[m]; }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (i + r >= i1) break;
#pragma unroll
                    for (int u = 0; u < kThetaTile; u++) {
                        if (rhs && u > 0) break;
                        const real su = rhs ? wa[r] : wa[r - u + kThetaTile - 1];       // s_{i+r-la[u]}
#pragma unroll
                        for (int v = 0; v < kThetaTile; v++) {
                            const real prod = su * wb[r - v + kThetaTile - 1];           // val_type product
                            acc[u][v] += (double)prod;                                 // double accumulate
                        }
                    }
                }
            }
        } else if (live) {
            const int nrow = rhs ? 1 : kThetaTile;
            for (int i = i0 + slice; i < i1; i += S) {
                real sa[kThetaTile], sb[kThetaTile];
#pragma unroll
                for (int u = 0; u < kThetaTile; u++) { sa[u] = series[i - la[u] - lo]; sb[u] = series[i - lb[u] - lo]; }
#pragma unroll
                for (int u = 0; u < kThetaTile; u++)
                    if (u < nrow) {
#pragma unroll
                        for (int v = 0; v < kThetaTile; v++) {
                            const real prod = sa[u] * sb[v];                           // val_type product
                            acc[u][v] += (double)prod;                                 // double accumulate
                        }
                    }
            }
        }
        // slice sums -> LDS -> the slice-0 thread of a block adds them in order and stores the block's pairs
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kThetaTile; u++)
#pragma unroll
            for (int v = 0; v < kThetaTile; v++) red[(size_t)threadIdx.x * 16 + u * kThetaTile + v] = acc[u][v];
        __syncthreads();
        if (live && slice == 0) {
#pragma unroll
            for (int u = 0; u < kThetaTile; u++)
#pragma unroll
                for (int v = 0; v < kThetaTile; v++) {
                    const int a = kThetaTile * ta + u, b = kThetaTile * tb + v;
                    if (b >= nlag || (!rhs && (a >= nlag || a > b)) || (rhs && u > 0)) continue;
                    double sum = 0;
                    for (int sl = 0; sl < S; sl++) sum += red[(size_t)(threadIdx.x + sl) * 16 + u * kThetaTile + v];
                    // pair index: p < nlag -> rhs entry y[b]; otherwise upper-triangle (a, b) in row-major order
                    const int p = rhs ? b : nlag + a * nlag - a * (a - 1) / 2 + (b - a);
                    out[p] = sum;
                }
        }
    }
}

// one workgroup per latent dimension; dynamic LDS = (nlag*nlag + nlag) * sizeof(real).  All 256 threads add up the
// time-chunk partials (a single wavefront streaming nchunk * npairs doubles is latency-bound: 125 us of the former
// 187 us at 48 lags); the |L| x |L| solve itself is one wavefront's work, the other three retire after the sum.
// `scratch` != null: the systems live there ((nlag*nlag + nlag) reals per latent dimension, L2-resident) instead of LDS -- lag sets
// too long for LDS (the reference has no limit on |L|: trmf.cpp:425-484); same code, workgroup barriers order the accesses.
__global__ __launch_bounds__(256) void theta_solve_kernel(const double *__restrict__ part, int nchunk,
                                                         int nlag, int npairs, double lambdaLag,
                                                         real *__restrict__ theta, real *__restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    real *A = scratch ? scratch + (size_t)blockIdx.x * ((size_t)nlag * nlag + nlag) : reinterpret_cast<real *>(smem_raw);   // nlag x nlag, (i,j) at A[i*nlag+j]
    real *y = A + nlag * nlag;
    const int t = blockIdx.x, lane = threadIdx.x;
    for (int p = threadIdx.x; p < npairs; p += 256) {
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
    if (threadIdx.x >= 64) return;                      // the barriers below only count the wavefront that is left
    // upper Cholesky A = U^T U, row by row (posv 'U', rf_matrix.h:3008-3014).  One wavefront: LDS accesses
    // retire in program order, the barriers are wave-local.  Same operations on every element, in the same j order,
    // as the row-by-row loop.
    for (int j = 0; j < nlag; j++) {
        const real ajj = sqrt(A[j * nlag + j]);
        __syncthreads();
        for (int c = j + lane; c < nlag; c += 64) A[j * nlag + c] = (c == j) ? ajj : A[j * nlag + c] / ajj;
        __syncthreads();
        // trailing update, a lane per column (two for more than 64 lags): rows s = j+1 .. c of its column
        for (int c = j + 1 + lane; c < nlag; c += 64) {
            const real ujc = A[j * nlag + c];
            int s = j + 1;
            for (; s + 3 <= c; s += 4) {                 // four independent rows per pass: reads first, then writes
                real u[4], a[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { u[q] = A[j * nlag + s + q]; a[q] = A[(s + q) * nlag + c]; }
#pragma unroll
                for (int q = 0; q < 4; q++) A[(s + q) * nlag + c] = a[q] - u[q] * ujc;
            }
            for (; s <= c; s++) A[s * nlag + c] -= A[j * nlag + s] * ujc;
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
