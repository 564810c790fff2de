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
// The dense contraction of the full-observation path (gmat_x_dmat / dmat_x_dmat -> BLAS gemm in the reference,
// rf_matrix.h:3182-3216) on the matrix pipe: a wavefront owns a 16-row tile of C and all NT column tiles; per
// K-slice of 4 the A operand is one 64-byte segment per 16-lane group (A is row-major with m contiguous) and
// the B operand is the lane's NT adjacent values of the column-interleaved factor row -- both straight from
// global memory in fragment layout, no LDS.  A is streamed exactly once (the roofline of this product: the
// bytes of Y); B is small and cache-resident.  The K range is split over gridDim.y chunks (enough workgroups
// to fill the chip when C has only a few hundred rows); the chunk partials are summed in fixed order by
// dense_tn_reduce_kernel.  Products and sums inside a chunk are val_type fused multiply-adds in K order
// (like a gemm micro-kernel); across chunks the sum is fp64.
template <int NT>
__global__ __launch_bounds__(256) void dense_tn_mfma_kernel(const real *__restrict__ A, int K, int M,
                                                            const real *__restrict__ B,
                                                            double *__restrict__ part) {
    constexpr int KP = kTile * NT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int m0 = (blockIdx.x * 4 + wave) * kTile;
    if (m0 >= M) return;                                // wave-uniform
    const int nchunk = gridDim.y, ch = blockIdx.y;
    const int j0 = (int)((long long)K * ch / nchunk), j1 = (int)((long long)K * (ch + 1) / nchunk);
    typename Mfma16<real>::acc_t acc[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) acc[q] = typename Mfma16<real>::acc_t{0, 0, 0, 0};
    const int mrow = min(m0 + c, M - 1);                // rows past M shadow the last one (never stored)
    constexpr int U = 4;                                // K-slices requested together (loads in flight before the first MFMA)
    for (int j = j0; j < j1; j += 4 * U) {
        real a[U];
        RealVec<NT> bv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int jj = j + 4 * u + g;
            const bool valid = jj < j1;
            const size_t jr = (size_t)(valid ? jj : j0);
            a[u] = A[jr * M + mrow];
            bv[u] = *reinterpret_cast<const RealVec<NT> *>(B + jr * KP + NT * c);
            if (!valid) a[u] = 0;                       // a slice past the chunk end contributes nothing
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q = 0; q < NT; q++) acc[q] = Mfma16<real>::mma(a[u], bv[u].v[q], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < NT; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + Mfma16<real>::row(lane, r);
            if (m < M) part[((size_t)ch * M + m) * KP + NT * c + q] = (double)acc[q][r];
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

// few chunks (many output rows, short contraction): one thread per output element, same fixed chunk order
__global__ __launch_bounds__(256) void dense_tn_reduce_flat_kernel(const double *__restrict__ part, int nchunk,
                                                                   int M, int KP, int NT, int k,
                                                                   real *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)M * KP) return;
    const int m = (int)(e / KP), t = (int)(e - (size_t)m * KP);          // t: logical column of the output
    const int tp = colpos(t, NT);
    double acc = 0;
    for (int ch = 0; ch < nchunk; ch++) acc += part[((size_t)ch * M + m) * KP + tp];
    out[e] = (t < k) ? (real)acc : real(0);
}

// ---- small Gram: GS (k x k, logical) = A^T A (+ lambda I) over the rows of a factor ------------------
// A SYRK over contiguous factor rows on the matrix pipe: every wavefront takes a contiguous chunk of rows, four
// rows per MFMA K-slice, operands loaded in fragment layout (the lane's NT adjacent values of a row), upper
// tiles only; its k x k partial (both triangles, fp64) goes to slot blockIdx.x * 4 + wave and the slots are
// reduced in fixed order by small_gram_reduce_kernel.
template <int NT>
__global__ __launch_bounds__(256) void small_gram_mfma_kernel(const real *__restrict__ A, int rows, int k,
                                                              double *__restrict__ part) {
    constexpr int KP = kTile * NT, NTRI = NT * (NT + 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int slot = blockIdx.x * 4 + wave, nslot = gridDim.x * 4;
    const int r0 = (int)((long long)rows * slot / nslot), r1 = (int)((long long)rows * (slot + 1) / nslot);
    typename Mfma16<real>::acc_t acc[NTRI];
#pragma unroll
    for (int t = 0; t < NTRI; t++) acc[t] = typename Mfma16<real>::acc_t{0, 0, 0, 0};
    constexpr int U = 2;
    for (int r = r0; r < r1; r += 4 * U) {
        RealVec<NT> x[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int rr = r + 4 * u + g;
            const bool valid = rr < r1;
            x[u] = *reinterpret_cast<const RealVec<NT> *>(A + (size_t)(valid ? rr : r0) * KP + NT * c);
            if (!valid) {
#pragma unroll
                for (int q = 0; q < NT; q++) x[u].v[q] = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            int t = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ti++)
#pragma unroll
                for (int tj = ti; tj < NT; tj++, t++) acc[t] = Mfma16<real>::mma(x[u].v[ti], x[u].v[tj], acc[t]);
        }
    }
    double *dst = part + (size_t)slot * k * k;
    int t = 0;
#pragma unroll
    for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int tj = ti; tj < NT; tj++, t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int s = kTile * ti + Mfma16<real>::row(lane, r), tc = kTile * tj + c;
                if (s < k && tc < k) {
                    dst[s * k + tc] = (double)acc[t][r];
                    if (ti != tj) dst[tc * k + s] = (double)acc[t][r];
                }
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

// ---- shared-matrix solve: H[i][:] = GS^-1 b_i for every row (posv with n right-hand sides, trmf.cpp:333) ----
// chol_shared_kernel: ONE workgroup factorises the k x k matrix in LDS (upper Cholesky in val_type, as posv 'U')
// and leaves U in global memory.  solve_rows_kernel: every workgroup copies U to LDS; a wavefront owns one
// right-hand side at a time with one unknown per lane, and both substitutions are column-oriented -- after step q
// every remaining lane has had its U(.,.) * z_q term removed -- so a row costs 2k broadcast + FMA steps on 64 lanes
// instead of k^2 dependent LDS round trips on one thread.  Forward: the same subtractions in the same order as the
// row-oriented loop; backward: a row's terms are subtracted in descending instead of ascending order (last bit).
__global__ __launch_bounds__(256) void chol_shared_kernel(const real *__restrict__ GS, real *__restrict__ Uout, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_raw[];
    real *U = reinterpret_cast<real *>(ss_raw);          // k x k
    for (int e = threadIdx.x; e < k * k; e += 256) U[e] = GS[e];
    __syncthreads();
    for (int j = 0; j < k; j++) {                        // same loop as theta_solve_kernel
        const real ajj = sqrt(U[j * k + j]);
        __syncthreads();
        for (int c = j + threadIdx.x; c < k; c += 256) U[j * k + c] = (c == j) ? ajj : U[j * k + c] / ajj;
        __syncthreads();
        const int m = k - 1 - j;                        // trailing dimension: all (s, c) pairs at once, 256 per pass
        for (int e = threadIdx.x; e < m * m; e += 256) {
            const int s = j + 1 + e / m, c = j + 1 + e % m;
            if (c >= s) U[s * k + c] -= U[j * k + s] * U[j * k + c];
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < k * k; e += 256) Uout[e] = U[e];
}
// dynamic LDS = k * k * sizeof(real); k <= 64: lane p holds unknown p
__global__ __launch_bounds__(256) void solve_rows_kernel(const real *__restrict__ Ug, const real *__restrict__ Brows,
                                                         real *__restrict__ out, int rows, int k, int KP, int NT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_raw[];
    real *U = reinterpret_cast<real *>(ss_raw);
    for (int e = threadIdx.x; e < k * k; e += 256) U[e] = Ug[e];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = lane < k ? lane : k - 1;               // idle lanes shadow the last unknown
    const real upp = U[p * k + p];
    for (int i = blockIdx.x * 4 + wave; i < rows; i += gridDim.x * 4) {
        real x = Brows[(size_t)i * KP + p];
        for (int q = 0; q < k; q++) {                    // U^T z = b
            const real zq = lane_bcast(x, q) / U[q * k + q];
            if (lane == q) x = zq;
            else if (lane > q) x -= U[q * k + p] * zq;
        }
        for (int q = k - 1; q >= 0; q--) {               // U x = z
            const real xq = lane_bcast(x, q) / U[q * k + q];
            if (lane == q) x = xq;
            else if (lane < q) x -= U[p * k + q] * xq;
        }
        (void)upp;
        if (lane < k) out[(size_t)i * KP + colpos(lane, NT)] = x;
    }
}

}  // namespace trmf
