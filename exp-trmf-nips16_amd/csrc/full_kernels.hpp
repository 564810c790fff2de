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
#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int NT>
__global__ void spmm_rows_kernel(const uint32_t *__restrict__ ptr,
                                                        const uint32_t *__restrict__ idx,
                                                        const real *__restrict__ val,
                                                        const real *__restrict__ X,
                                                        real *__restrict__ out, uint32_t row_begin,
                                                        uint32_t row_end, uint32_t zero_row, uint32_t long_thresh);
// a split row's item (gram_kernels.hpp "split rows"): the same sum over entries [items[2 i], items[2 i + 1]) -> part[i][0 .. KP)
template <int NT>
__global__ void spmm_part_kernel(const uint32_t *__restrict__ idx, const real *__restrict__ val, const real *__restrict__ X,
                                 const uint32_t *__restrict__ items, uint32_t item_begin, uint32_t item_end, real *__restrict__ part,
                                 uint32_t zero_row);
#else
// ITEM: one wavefront per item of a split row instead of per row
template <int NT, bool ITEM>
__device__ __forceinline__ void spmm_body(const uint32_t *__restrict__ ptr,
                                                        const uint32_t *__restrict__ idx,
                                                        const real *__restrict__ val,
                                                        const real *__restrict__ X,
                                                        real *__restrict__ out, uint32_t row_begin,
                                                        uint32_t row_end, uint32_t zero_row, uint32_t long_thresh) {
    constexpr int KP = kTile * NT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    const uint32_t row = row_begin + blockIdx.x * 4u + (uint32_t)wave;
    if (row >= row_end) return;
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ptr[ITEM ? 2 * row : row]);
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ptr[ITEM ? 2 * row + 1 : row + 1]);
    if (!ITEM && p1 - p0 >= long_thresh) return;        // a split row: spmm_part_kernel + spmm_reduce_kernel write it
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
template <int NT>
__global__ __launch_bounds__(256) void spmm_rows_kernel(const uint32_t *__restrict__ ptr,
                                                        const uint32_t *__restrict__ idx,
                                                        const real *__restrict__ val,
                                                        const real *__restrict__ X,
                                                        real *__restrict__ out, uint32_t row_begin,
                                                        uint32_t row_end, uint32_t zero_row, uint32_t long_thresh) {
    spmm_body<NT, false>(ptr, idx, val, X, out, row_begin, row_end, zero_row, long_thresh);
}
template <int NT>
__global__ __launch_bounds__(256) void spmm_part_kernel(const uint32_t *__restrict__ idx, const real *__restrict__ val,
                                                        const real *__restrict__ X, const uint32_t *__restrict__ items,
                                                        uint32_t item_begin, uint32_t item_end, real *__restrict__ part, uint32_t zero_row) {
    spmm_body<NT, true>(items, idx, val, X, part, item_begin, item_end, zero_row, 0u);
}
#endif

#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
// out[row of long-row position li][:] = the row's item partials summed in item order (one thread per column)
__global__ __launch_bounds__(256) void spmm_reduce_kernel(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ first,
                                                          const real *__restrict__ part, real *__restrict__ out, uint32_t begin,
                                                          uint32_t end, int KP) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t li = begin + (uint32_t)(e / (size_t)KP);
    if (li >= end) return;
    const int col = (int)(e % (size_t)KP);
    real acc = 0;
    for (uint32_t it = first[li]; it < first[li + 1]; it++) acc += part[(size_t)it * KP + col];
    out[(size_t)rows[li] * KP + col] = acc;
}
#endif

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
#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int NT>
__global__ void dense_tn_mfma_kernel(const real *__restrict__ A, int K, int M,
                                                            const real *__restrict__ B,
                                                            double *__restrict__ part);
#else
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
#endif
// out (M x KP, LOGICAL columns) = sum over chunks; positions of B's interleaved layout mapped back.
// One wavefront per output element: lanes stride over the chunks, then a fixed-order butterfly -- the
// chunk count goes up to ~1000 for tall-skinny products (few output rows, long contraction).
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
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
#endif

// few chunks (many output rows, short contraction): one thread per output element, same fixed chunk order
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
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
#endif

// ---- small Gram: GS (k x k, logical) = A^T A (+ lambda I) over the rows of a factor ------------------
// A SYRK over contiguous factor rows on the matrix pipe: every wavefront takes a contiguous chunk of rows, four
// rows per MFMA K-slice, operands loaded in fragment layout (the lane's NT adjacent values of a row), upper
// tiles only; its k x k partial (both triangles, fp64) goes to slot blockIdx.x * 4 + wave and the slots are
// reduced in fixed order by small_gram_reduce_kernel.
#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int NT>
__global__ void small_gram_mfma_kernel(const real *__restrict__ A, int rows, int k,
                                                              double *__restrict__ part);
#else
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
#endif
// one wavefront per entry: lanes stride over the workgroup partials, fixed-order butterfly
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
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
#endif

// ---- shared-matrix solve: H[i][:] = GS^-1 b_i for every row (posv with n right-hand sides, trmf.cpp:333) ----
// chol_shared_kernel: ONE workgroup factorises the k x k matrix in LDS (upper Cholesky in val_type, as posv 'U')
// and leaves U in global memory.  solve_rows_kernel: every workgroup copies U to LDS; a wavefront owns one
// right-hand side at a time with one unknown per lane, and both substitutions are column-oriented -- after step q
// every remaining lane has had its U(.,.) * z_q term removed -- so a row costs 2k broadcast + FMA steps on 64 lanes
// instead of k^2 dependent LDS round trips on one thread.  Forward: the same subtractions in the same order as the
// row-oriented loop; backward: a row's terms are subtracted in descending instead of ascending order (last bit).
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
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
#endif
// chol_wave_kernel: the same factorisation by ONE wavefront without LDS or barriers.  Lane c keeps column c of the
// matrix in registers (KMAX values; rows and columns >= k padded with the identity, so no step needs a guard); step
// j scales row j and subtracts u_js * u_jc from every later row s, u_js arriving as a scalar through v_readlane
// with a compile-time lane -- the loops are fully unrolled.  Same operations per element as chol_shared_kernel
// (sqrt, divide, one fused multiply-subtract), ~4x faster at k = 60 (70 -> 18 us): the workgroup version is a chain
// of 3 k barriers and LDS round trips.  Writes U (upper part, zeros below).
#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int NT>
__global__ void chol_wave_kernel(const real *__restrict__ GS, real *__restrict__ Uout, int k);
#else
template <int NT>
__global__ __launch_bounds__(64) void chol_wave_kernel(const real *__restrict__ GS, real *__restrict__ Uout, int k) {
    constexpr int KMAX = kTile * NT;
    const int c = threadIdx.x;
    real a[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; s++) a[s] = (s < k && c < k) ? GS[(size_t)s * k + c] : (s == c ? real(1) : real(0));
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
        const real ajj = sqrt(lane_bcast(a[j], j));
        const real u = (c == j) ? ajj : a[j] / ajj;
        a[j] = u;
#pragma unroll
        for (int s = j + 1; s < KMAX; s++) {
            a[s] = fma(-lane_bcast(u, s), u, a[s]);
            if ((s & 15) == 15) __builtin_amdgcn_sched_barrier(0);       // bound the scalar operands in flight
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (c < k) {
#pragma unroll
        for (int s = 0; s < KMAX; s++)
            if (s < k) Uout[(size_t)s * k + c] = s <= c ? a[s] : real(0);
    }
}
#endif
// dynamic LDS = k * k * sizeof(real); k <= 64: lane p holds unknown p
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
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
#endif

// ---- unfused CG with ONE Gram for every timestamp: out = base + V G (- B), on the matrix pipe -------------------
// apply_kernel (cg_kernels.hpp) spends two LDS reads per multiply-add when the Gram is shared; here the product of a
// 16-timestamp tile of the operand with the k x k Gram is KP/4 x NT fp64 MFMAs (fp32 sessions: operands widened, so
// the accumulation stays in double as in apply_kernel).  The Gram is staged once per workgroup in LDS in VECTOR
// POSITION order on both axes (zero in the pad positions), so tiles of the column-interleaved vectors are used as
// they lie in memory.  A wavefront owns a tile: coalesced rows -> its LDS slab -> A fragments; base / residual /
// right-hand side of the tile are requested before the MFMA loop and consumed in the accumulator layout.
// Same outputs and partial-sum slots as apply_kernel (summation order differs).
constexpr int kApplyTile = 16;
__device__ __forceinline__ void wave_slab_sync() {      // LDS operations of one wavefront retire in order: a fence, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__host__ __device__ constexpr int apply_gram_pitch(int KP) { return (KP % 32 == 16) ? KP : KP + 16; }   // doubles; B-fragment reads conflict-free
__host__ __device__ constexpr int apply_tile_pitch(int KP) { return KP + 4; }                            // reals; A-fragment reads conflict-free
__host__ __device__ constexpr size_t apply_shared_lds_bytes(int KP) {
    return (size_t)KP * apply_gram_pitch(KP) * sizeof(double) + (size_t)4 * kApplyTile * apply_tile_pitch(KP) * sizeof(real);
}
#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int NT>
__global__ void apply_shared_mfma_kernel(XParams p, const XState *__restrict__ st, int cg_it,
                                                                const real *__restrict__ v, const real *__restrict__ rvec,
                                                                const real *__restrict__ base, const real *__restrict__ G,
                                                                const real *__restrict__ Bv, int minus_b,
                                                                real *__restrict__ out, int dot_mode,
                                                                double *__restrict__ Pdot, int row0, int nrows, int slot0);
#else
template <int NT>
__global__ __launch_bounds__(256) void apply_shared_mfma_kernel(XParams p, const XState *__restrict__ st, int cg_it,
                                                                const real *__restrict__ v, const real *__restrict__ rvec,
                                                                const real *__restrict__ base, const real *__restrict__ G,
                                                                const real *__restrict__ Bv, int minus_b,
                                                                real *__restrict__ out, int dot_mode,
                                                                double *__restrict__ Pdot, int row0, int nrows, int slot0) {
    constexpr int KP = kTile * NT, GP = apply_gram_pitch(KP), TP = apply_tile_pitch(KP), EPL = kApplyTile * KP / 64;
    typedef Mfma16<double> M;
    extern __shared__ __attribute__((aligned(16))) unsigned char apply_lds[];
    __shared__ double smem[256];
    const bool cg = cg_it >= 0;
    if (cg && st->stop_it <= cg_it) return;               // see apply_kernel
    double *Gs = reinterpret_cast<double *>(apply_lds);                                   // Gs[position kk][position cc]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = p.k, c = lane & 15, g = lane >> 4;
    real *tile = reinterpret_cast<real *>(Gs + (size_t)KP * GP) + (size_t)wave * kApplyTile * TP;
    const int ntiles = (nrows + kApplyTile - 1) / kApplyTile, stride = gridDim.x * 4;
    real raw[EPL];
    auto load_raw = [&](int tidx) {                       // the tile's rows, coalesced
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int idx = e * 64 + lane, rr = idx / KP, i = tidx * kApplyTile + rr;
            raw[e] = i < nrows ? v[(size_t)(row0 + i) * KP + (idx - rr * KP)] : real(0);
        }
    };
    int tidx = blockIdx.x * 4 + wave;
    if (tidx < ntiles) load_raw(tidx);
    for (int e = threadIdx.x; e < KP * KP; e += 256) {
        const int kk = e / KP, cc = e - kk * KP, s = collog(kk, NT), t = collog(cc, NT);
        Gs[kk * GP + cc] = (s < k && t < k) ? (double)G[(size_t)s * k + t] : 0.0;
    }
    __syncthreads();
    double dot = 0, lq = 0, rhd = 0, hh = 0;
    for (; tidx < ntiles; tidx += stride) {
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int idx = e * 64 + lane, rr = idx / KP;
            tile[rr * TP + (idx - rr * KP)] = raw[e];
        }
        wave_slab_sync();
        if (tidx + stride < ntiles) load_raw(tidx + stride);
        // operands of the epilogue, in the accumulator layout (row g + 4r, position 16 nt + c)
        real ob[NT][4], rv[NT][4];
        double bb[NT][4];
        bool on[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = tidx * kApplyTile + M::row(lane, r), pc = kTile * nt + c, t = collog(pc, NT);
                const size_t row = (size_t)(row0 + i) * KP;
                on[nt][r] = i < nrows && t < k;
                ob[nt][r] = on[nt][r] ? base[row + pc] : real(0);               // lambdaI*v + lambdaAR*AR'(v), ar_tile_kernel
                rv[nt][r] = (on[nt][r] && cg) ? rvec[row + pc] : real(0);
                bb[nt][r] = (on[nt][r] && minus_b) ? (double)Bv[row + t] : 0.0;
            }
        typename M::acc_t acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt] = typename M::acc_t{0, 0, 0, 0};
#pragma unroll 4
        for (int ks = 0; ks < KP / 4; ks++) {
            const double a = (double)tile[c * TP + 4 * ks + g];
            const double *brow = Gs + (size_t)(4 * ks + g) * GP + c;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = M::mma(a, brow[kTile * nt], acc[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (!on[nt][r]) continue;
                const int lrow = M::row(lane, r), pc = kTile * nt + c;
                const real x = tile[lrow * TP + pc];
                double a = acc[nt][r];
                if (minus_b) {
                    lq += (double)x * (a - 2.0 * bb[nt][r]);                     // w.(Gw) - 2 b.w
                    a -= bb[nt][r];
                }
                const real o = (real)((double)ob[nt][r] + a);
                out[(size_t)(row0 + tidx * kApplyTile + lrow) * KP + pc] = o;
                dot += (double)(dot_mode ? x : o) * (double)o;
                if (cg) {
                    rhd += (double)rv[nt][r] * (double)o;                        // <r,Hd>
                    hh += (double)o * (double)o;                                 // <Hd,Hd>
                }
            }
        wave_slab_sync();                                 // the slab is rewritten by the next tile
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
#endif

}  // namespace trmf
