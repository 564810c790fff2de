// resident_kernels.hpp -- device-side maintenance of an HBM-resident problem when new timestamps arrive
// (trmf_session_append_rows): the rolling-window caller of the reference (python/trmf/trmf.py:303-329) retrains
// on a prefix that grows by one window at a time and warm-starts W by the AR recursion (trmf.py:170-181,
// 237-246).  Only the new window crosses PCIe; everything else is rebuilt here.
#pragma once

#include "common.hpp"

namespace trmf {

// dst (cols x rows, row-major) = transpose of src (rows x cols, row-major); 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const real *__restrict__ src, int rows, int cols,
                                                        real *__restrict__ dst) {
    __shared__ real tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int dy = threadIdx.y; dy < 32; dy += 8) {
        const int r = r0 + dy, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[dy][threadIdx.x] = src[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int dy = threadIdx.y; dy < 32; dy += 8) {
        const int c = c0 + dy, r = r0 + threadIdx.x;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[threadIdx.x][dy];
    }
}

// CSC of the grown matrix: column j = its old entries followed by the new block's entries of that column
// (their timestamps shifted by T0, so the column stays sorted when both parts were).  One wavefront per column.
__global__ __launch_bounds__(256) void csc_append_kernel(const uint32_t *__restrict__ optr, const uint32_t *__restrict__ oidx,
                                                         const real *__restrict__ oval, const uint32_t *__restrict__ wptr,
                                                         const uint32_t *__restrict__ widx, const real *__restrict__ wval,
                                                         const uint32_t *__restrict__ nptr, uint32_t *__restrict__ nidx,
                                                         real *__restrict__ nval, int ncols, uint32_t T0) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= ncols) return;
    const uint32_t o0 = optr[j], on = optr[j + 1] - o0, w0 = wptr[j], wn = wptr[j + 1] - w0, d0 = nptr[j];
    for (uint32_t e = lane; e < on; e += 64) { nidx[d0 + e] = oidx[o0 + e]; nval[d0 + e] = oval[o0 + e]; }
    for (uint32_t e = lane; e < wn; e += 64) { nidx[d0 + on + e] = widx[w0 + e] + T0; nval[d0 + on + e] = wval[w0 + e]; }
}

// W[i][t] = sum_l W[i - L_l][t] * Theta(l, t) for i in [T0, T1), in timestamp order (each row feeds the next
// ones): Model.latent_forecast, trmf.py:170-181 -- an elementwise product rounded to val_type followed by a
// sequential sum over the lags, as NumPy evaluates `(Wnew[i - lags, :] * lag_val).sum(axis=0)`; no fused
// multiply-add.  One thread per latent dimension (k <= 64); a thread only ever reads its own column.
__global__ __launch_bounds__(64) void latent_forecast_kernel(real *__restrict__ W, int T0, int T1, int KP, int NT, int k,
                                                             const uint32_t *__restrict__ lag_set, int nlag,
                                                             const real *__restrict__ theta) {
#pragma clang fp contract(off)      // product and sum are rounded separately, like the NumPy expression (hipcc contracts by default)
    const int t = threadIdx.x;
    if (t >= k) return;
    const int tp = colpos(t, NT);
    for (int i = T0; i < T1; i++) {
        real acc = 0;
        for (int l = 0; l < nlag; l++) {
            const int src = i - (int)lag_set[l];
            const real w = src >= 0 ? W[(size_t)src * KP + tp] : real(0);
            const real prod = w * theta[(size_t)t * nlag + l];
            acc = acc + prod;
        }
        W[(size_t)i * KP + tp] = acc;
    }
}

// Factor layout conversion on the device (the ABI delivers rows x k row-major; HBM holds (rows+1) x KP with the
// column-interleaved layout of common.hpp, pads and the extra row zero).  One thread per element of the padded table.
__global__ __launch_bounds__(256) void factor_pad_kernel(const real *__restrict__ src, size_t rows, int k, int KP, int NT,
                                                         real *__restrict__ dst) {
    const size_t N = (rows + 1) * (size_t)KP;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < N; e += (size_t)gridDim.x * 256) {
        const size_t i = e / KP;
        const int t = collog((int)(e - i * KP), NT);
        dst[e] = (i < rows && t < k) ? src[i * (size_t)k + t] : real(0);
    }
}
__global__ __launch_bounds__(256) void factor_unpad_kernel(const real *__restrict__ src, size_t rows, int k, int KP, int NT,
                                                           real *__restrict__ dst) {
    const size_t N = rows * (size_t)k;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < N; e += (size_t)gridDim.x * 256) {
        const size_t i = e / k;
        const int t = (int)(e - i * k);
        dst[e] = src[i * (size_t)KP + colpos(t, NT)];
    }
}
// 64-bit row pointers of the ABI narrowed to the 32-bit device form
__global__ __launch_bounds__(256) void narrow_ptr_kernel(const uint64_t *__restrict__ src, size_t count, uint32_t *__restrict__ dst) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (size_t)gridDim.x * 256) dst[e] = (uint32_t)src[e];
}
// sum of squares of a value array (fp64, fixed-order per-workgroup partials)
__global__ __launch_bounds__(256) void sumsq_values_kernel(const real *__restrict__ v, size_t count, double *__restrict__ Psq) {
    __shared__ double sm[4];
    double acc = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (size_t)gridDim.x * 256) acc += (double)v[e] * (double)v[e];
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) Psq[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// dst[j][i] = raw[j][i] * a[i] + b[i]: the per-series affine map of NormalizedTransform.preprocess
// (python/trmf/trmf.py:82-96).  NumPy evaluates `Y * a + b` in the common dtype of Y and the coefficients -- which
// are fitted from Y and therefore have its dtype -- with the product and the sum rounded separately: val_type
// arithmetic, no fused multiply-add.  rows x cols row-major; also emits per-workgroup partial sums of dst^2.
__global__ __launch_bounds__(256) void affine_columns_kernel(const real *__restrict__ raw, size_t rows, int cols,
                                                             const real *__restrict__ a, const real *__restrict__ b,
                                                             real *__restrict__ dst, double *__restrict__ Psq) {
#pragma clang fp contract(off)
    __shared__ double sm[4];
    const size_t N = rows * (size_t)cols;
    double acc = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < N; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e % (size_t)cols);
        const real prod = raw[e] * a[i];
        const real y = prod + b[i];
        dst[e] = y;
        acc += (double)y * (double)y;
    }
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) Psq[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

}  // namespace trmf
