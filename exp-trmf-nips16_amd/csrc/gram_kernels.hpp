// gram_kernels.hpp -- the sparse normal-equation kernels (gfx950 / CDNA4, wave64, MFMA 16x16x4).
//
// What the reference does per observed entry (trmf.cpp:382-389): gather one k-vector x_j and add
// the rank-1 update x_j x_j^T (upper triangle) and y*x_j.  Here four observed entries form one
// K=4 slice of an exact-f32 (or f64) MFMA:  A_tile(ti,tj) += P^T P  with P the 4 x 16 panel of
// gathered factor slices, so the per-row Gram is a dense SYRK over the gathered panel with the
// contraction running over the row's observed entries in CSR order.  v_mfma_f32_16x16x4_f32 is
// bit-for-bit an fmaf chain in k order (MI355X guide section 3), i.e. the same summation order as
// the reference's sequential loop.  The factor is stored in HBM with leading dimension
// KP = 16*NT (zero padded), so every operand load is one 64-byte segment per 16-lane group and
// needs no masking.
//
//   fsolve_kernel   one wavefront per item row: Gram in MFMA accumulators -> LDS -> one factor
//                   column per lane in registers -> right-looking Cholesky with v_readlane
//                   broadcasts -> forward/backward substitution -> row of F.  (trmf.cpp:369-397)
//   gram_x_kernel   one 4-wave workgroup per timestamp row: Gram + rhs + loss of the X-side
//                   sub-problem, cached in HBM for the CG (replaces the per-Hv re-streaming of
//                   trmf.cpp:269-288).
//   loss_kernel     sum of squared residuals per timestamp row (trmf.cpp:231-245, loss part).
#pragma once

#include <type_traits>
#include <utility>

#include "common.hpp"

namespace trmf {

// ---- MFMA 16x16x4 traits ------------------------------------------------------------------------
template <typename T> struct Mfma16;
template <> struct Mfma16<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = 4*(lane>>4) + r
    static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <> struct Mfma16<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D layout differs: col = lane & 15, row = (lane>>4) + 4*r
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

// ---- cross-lane helpers -------------------------------------------------------------------------
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src_lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// 1/sqrt(p): v_rsq_f32 + one Newton step (<= ~1 ulp); f64 uses the correctly rounded forms.
__device__ __forceinline__ float inv_sqrt(float p) {
    const float r = __builtin_amdgcn_rsqf(p);
    const float h = 0.5f * p * r;
    return fmaf(r, fmaf(-h, r, 0.5f), r);
}
__device__ __forceinline__ double inv_sqrt(double p) { return 1.0 / sqrt(p); }

template <typename T> __device__ __forceinline__ T row16_allsum(T v) {   // sum over a 16-lane row
    v += __shfl_xor(v, 1, kWave);
    v += __shfl_xor(v, 2, kWave);
    v += __shfl_xor(v, 4, kWave);
    v += __shfl_xor(v, 8, kWave);
    return v;
}
template <typename T> __device__ __forceinline__ T wave_allsum(T v) {
    v = row16_allsum(v);
    v += __shfl_xor(v, 16, kWave);
    v += __shfl_xor(v, 32, kWave);
    return v;
}

// ---- Gram accumulation over one CSR row ----------------------------------------------------------
template <int NT> struct GramState {
    typename Mfma16<real>::acc_t acc[NT * (NT + 1) / 2];   // upper tiles, row-major over (ti<=tj)
    real b[NT];                                            // rhs partial of this lane group
    double loss;
};

// Lane (g = lane>>4, c = lane&15) handles observed entry 4*group+g and factor columns 16q+c.
// `wsub`/`nsub`: this wavefront takes groups wsub, wsub+nsub, ... of the row.
template <int NT, bool DO_MMA, bool WITH_LOSS>
__device__ __forceinline__ void gram_row(GramState<NT> &st, const uint32_t *__restrict__ idx,
                                         const real *__restrict__ val, const real *__restrict__ X,
                                         uint32_t p0, uint32_t p1, int wsub, int nsub, int lane,
                                         const real (&wq)[NT]) {
    constexpr int KP = kTile * NT;
    const int g = lane >> 4, c = lane & 15;
    const uint32_t step = 4u * (uint32_t)nsub;
    uint32_t base = p0 + 4u * (uint32_t)wsub;

    // software pipeline: indices two groups ahead, factor slices one group ahead
    uint32_t j1 = 0; real y0 = 0, y1 = 0; bool v1 = false;
    real x0[NT], x1[NT];
    {
        const uint32_t p = base + g;
        const bool v = p < p1;
        const uint32_t j = v ? idx[p] : 0u;
        y0 = v ? val[p] : real(0);
#pragma unroll
        for (int q = 0; q < NT; q++) {
            const real x = X[(size_t)j * KP + kTile * q + c];
            x0[q] = v ? x : real(0);
        }
        const uint32_t pn = base + step + g;
        v1 = pn < p1;
        j1 = v1 ? idx[pn] : 0u;
        y1 = v1 ? val[pn] : real(0);
    }
    while (base < p1) {
#pragma unroll
        for (int q = 0; q < NT; q++) {
            const real x = X[(size_t)j1 * KP + kTile * q + c];
            x1[q] = v1 ? x : real(0);
        }
        const uint32_t p2 = base + 2u * step + g;
        const bool v2 = p2 < p1;
        const uint32_t j2 = v2 ? idx[p2] : 0u;
        const real y2 = v2 ? val[p2] : real(0);

        // ---- consume group 0 ----
#pragma unroll
        for (int q = 0; q < NT; q++) st.b[q] = fma(y0, x0[q], st.b[q]);
        if (DO_MMA) {
            int t = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ti++)
#pragma unroll
                for (int tj = ti; tj < NT; tj++, t++)
                    st.acc[t] = Mfma16<real>::mma(x0[ti], x0[tj], st.acc[t]);
        }
        if (WITH_LOSS) {
            real d = 0;
#pragma unroll
            for (int q = 0; q < NT; q++) d = fma(wq[q], x0[q], d);
            d = row16_allsum(d);
            const real res = y0 - d;                      // trmf.cpp:238 (val_type arithmetic)
            st.loss += (double)res * (double)res;         // invalid entries: y0 = x0 = 0 -> 0
        }
        // ---- rotate ----
#pragma unroll
        for (int q = 0; q < NT; q++) x0[q] = x1[q];
        y0 = y1; j1 = j2; y1 = y2; v1 = v2;
        base += step;
    }
}

// ---- F-solve: one wavefront per item row -----------------------------------------------------------
// KMAX: static bound of the factorisation loops, k <= KMAX <= 16*NT (KMAX = k rounded up to 8).
template <int NT, int KMAX>
__global__ __launch_bounds__(256) void fsolve_kernel(const uint32_t *__restrict__ ptr,
                                                     const uint32_t *__restrict__ idx,
                                                     const real *__restrict__ val,
                                                     const real *__restrict__ X,
                                                     real *__restrict__ F, uint32_t row_begin,
                                                     uint32_t row_end, int k, real lambda) {
    constexpr int KP = kTile * NT, LD = KP + 1;
    __shared__ real lds[4][KP * LD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t row = row_begin + blockIdx.x * 4u + (uint32_t)wave;
    if (row >= row_end) return;                         // wave-uniform; no block barrier below
    const uint32_t p0 = ptr[row], p1 = ptr[row + 1];
    if (p0 == p1) return;                               // trmf.cpp:374: empty rows stay untouched

    GramState<NT> st;
#pragma unroll
    for (int t = 0; t < NT * (NT + 1) / 2; t++) st.acc[t] = typename Mfma16<real>::acc_t{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NT; q++) st.b[q] = 0;
    st.loss = 0;
    real nowq[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) nowq[q] = 0;
    gram_row<NT, true, false>(st, idx, val, X, p0, p1, 0, 1, lane, nowq);

    const int g = lane >> 4, c = lane & 15;
    // rhs: fold the 4 lane groups; afterwards lane t owns b[t] = st.b[t>>4]
    real bz = 0;
#pragma unroll
    for (int q = 0; q < NT; q++) {
        real v = st.b[q];
        v += __shfl_xor(v, 16, kWave);
        v += __shfl_xor(v, 32, kWave);
        if (g == q) bz = v;
    }

    // accumulators -> LDS slab (upper tiles only)
    real *S = lds[wave];
    {
        int t = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int tj = ti; tj < NT; tj++, t++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    S[(kTile * ti + Mfma16<real>::row(lane, r)) * LD + kTile * tj + c] = st.acc[t][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // one column per lane: a[s] = A[s][col], valid for s <= col (upper triangle); + lambda on diag
    const int col = lane < KP ? lane : KP - 1;
    real a[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; s++) {
        a[s] = S[s * LD + col];
        if (s == lane) a[s] += lambda;                  // trmf.cpp:393
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // right-looking Cholesky A = U^T U, forward substitution fused (bz -> z = U^-T b)
    real dinv = 0;
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
        if (j < k) {
            const real inv = inv_sqrt(lane_bcast(a[j], j));
            const real u = lane > j ? a[j] * inv : real(0);     // row j of U, strictly right of diag
            const real zj = lane_bcast(bz, j) * inv;
            bz = fma(-u, zj, bz);
            if (lane == j) { bz = zj; dinv = inv; }
            S[j * LD + col] = u;                                // row layout for the back solve
#pragma unroll
            for (int s = j + 1; s < KMAX; s++) a[s] = fma(-lane_bcast(u, s), u, a[s]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // one row of U per lane (a[t] = U[lane][t], t > lane), then column-oriented back substitution
    const int rw = lane < KMAX ? lane : KMAX - 1;
#pragma unroll
    for (int t = 0; t < KMAX; t++) a[t] = S[rw * LD + t];
    real x = 0;
#pragma unroll
    for (int t = KMAX - 1; t >= 0; t--) {
        if (t < k) {
            const real xt = lane_bcast(bz * dinv, t);
            if (lane == t) x = xt;
            bz = fma(-a[t], xt, bz);        // lanes >= t hold dead values from here on
        }
    }
    if (lane < k) F[(size_t)row * KP + lane] = x;
}

#if defined(TRMF_F32)
// ---- F-solve, quad form (fp32): one wavefront per FOUR item rows ------------------------------------
// The O(k^3) part of the solve is a chain of rank-1 updates whose operands must be broadcast across
// lanes.  With one system per wavefront (fsolve_kernel above) every broadcast is a v_readlane that
// feeds a single FMA per lane, and the kernel is VALU-issue bound (~3000 of its ~3500 instructions).
// Here the four 16-lane rows of a wavefront each own one system: lane (grp, c) holds columns
// {c, 16+c, 32+c, ...} of system `grp`, one ds_swizzle row-broadcast serves all four systems and
// feeds up to NT FMAs per lane, and the back substitution reduces inside the 16-lane row with DPP.
// The four Grams are still accumulated one after the other with the full-wave MFMA of gram_row().
template <int... Is, typename Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, Fn &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename Fn> __device__ __forceinline__ void static_for(Fn &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
// value of lane SRC (0..15) of the caller's own 16-lane row (ds_swizzle bit mode, no LDS memory)
template <int SRC> __device__ __forceinline__ float row_bcast(float v) {
    constexpr int pattern = (SRC << 5) | 0x10;          // and_mask = 0x10, or_mask = SRC, xor_mask = 0
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), pattern));
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_allsum_dpp(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror
    v += dpp_mov<0x140>(v);     // row_mirror
    return v;
}

template <int NT, int KMAX>
__global__ __launch_bounds__(256) void fsolve_quad_kernel(const uint32_t *__restrict__ ptr,
                                                          const uint32_t *__restrict__ idx,
                                                          const float *__restrict__ val,
                                                          const float *__restrict__ X,
                                                          float *__restrict__ F, uint32_t row_begin,
                                                          uint32_t row_end, int k, float lambda) {
    static_assert(sizeof(real) == 4, "quad F-solve is the fp32 path");
    constexpr int KP = kTile * NT, LDC = KP + 4;        // column-major slab: S[col * LDC + row]
    __shared__ __attribute__((aligned(16))) float lds[4][KP * LDC];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 4, c = lane & 15;
    const uint32_t row0 = row_begin + (blockIdx.x * 4u + (uint32_t)wave) * 4u;
    if (row0 >= row_end) return;                        // wave-uniform; no block barrier below
    float *S = lds[wave];
    typedef float f4 __attribute__((ext_vector_type(4)));

    // areg[q][s] = A[s][16q + c] of this lane row's system; only s <= 16q+15 is ever touched
    float areg[NT][KMAX];
    float bz[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) {
        bz[q] = 0;
#pragma unroll
        for (int s = 0; s < KMAX; s++) areg[q][s] = (s == kTile * q + c) ? 1.0f : 0.0f;   // idle rows: identity
    }
    bool mine = false;

#pragma unroll
    for (int sys = 0; sys < 4; sys++) {
        const uint32_t row = row0 + (uint32_t)sys;
        uint32_t p0 = 0, p1 = 0;
        if (row < row_end) { p0 = ptr[row]; p1 = ptr[row + 1]; }
        if (p0 == p1) continue;                         // trmf.cpp:374 (wave-uniform)
        GramState<NT> st;
#pragma unroll
        for (int t = 0; t < NT * (NT + 1) / 2; t++) st.acc[t] = typename Mfma16<real>::acc_t{0, 0, 0, 0};
        real nowq[NT];
#pragma unroll
        for (int q = 0; q < NT; q++) { st.b[q] = 0; nowq[q] = 0; }
        st.loss = 0;
        gram_row<NT, true, false>(st, idx, val, X, p0, p1, 0, 1, lane, nowq);
#pragma unroll
        for (int q = 0; q < NT; q++) {
            st.b[q] += __shfl_xor(st.b[q], 16, kWave);
            st.b[q] += __shfl_xor(st.b[q], 32, kWave);
        }
        // + lambda on the diagonal (trmf.cpp:393), then accumulators -> slab, 4 consecutive rows per store
        {
            int t = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ti++)
#pragma unroll
                for (int tj = ti; tj < NT; tj++, t++) {
                    f4 v = st.acc[t];
                    if (ti == tj) {
#pragma unroll
                        for (int r = 0; r < 4; r++) if (c == 4 * grp + r) v[r] += lambda;
                    }
                    *reinterpret_cast<f4 *>(&S[(kTile * tj + c) * LDC + kTile * ti + 4 * grp]) = v;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (grp == sys) {
            mine = true;
#pragma unroll
            for (int q = 0; q < NT; q++) {
                bz[q] = st.b[q];
                constexpr int dummy = 0; (void)dummy;
#pragma unroll
                for (int s4 = 0; s4 < KMAX / 4; s4++) {
                    if (4 * s4 <= kTile * q + 15) {
                        const f4 v = *reinterpret_cast<const f4 *>(&S[(kTile * q + c) * LDC + 4 * s4]);
                        areg[q][4 * s4 + 0] = v[0]; areg[q][4 * s4 + 1] = v[1];
                        areg[q][4 * s4 + 2] = v[2]; areg[q][4 * s4 + 3] = v[3];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- four right-looking Cholesky factorisations side by side, forward substitution fused ----
    float dinv[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) dinv[q] = 0;
    static_for<KMAX>([&](auto J) {
        constexpr int j = decltype(J)::value, qj = j >> 4, cj = j & 15;
        if (j < k) {
            const float inv = inv_sqrt(row_bcast<cj>(areg[qj][j]));
            float u[NT];
#pragma unroll
            for (int q = qj; q < NT; q++) {
                u[q] = areg[q][j] * inv;
                if (q == qj && c <= cj) u[q] = 0;       // row j of U, strictly right of the diagonal
                areg[q][j] = u[q];
            }
            const float zj = row_bcast<cj>(bz[qj]) * inv;
#pragma unroll
            for (int q = qj; q < NT; q++) bz[q] = fmaf(-u[q], zj, bz[q]);
            if (c == cj) { bz[qj] = zj; dinv[qj] = inv; }
            static_for<KMAX>([&](auto Sx) {
                constexpr int s = decltype(Sx)::value, qs = s >> 4, cs = s & 15;
                if constexpr (s > j) {
                    const float us = row_bcast<cs>(u[qs]);
#pragma unroll
                    for (int q = qs; q < NT; q++) areg[q][s] = fmaf(-us, u[q], areg[q][s]);
                }
            });
        }
    });

    // ---- back substitution U x = z, row-oriented, reduction inside the 16-lane row ----
    float x[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) x[q] = 0;
    static_for<KMAX>([&](auto Jr) {
        constexpr int j = KMAX - 1 - decltype(Jr)::value, qj = j >> 4, cj = j & 15;
        if (j < k) {
            float part = 0;
#pragma unroll
            for (int q = qj; q < NT; q++) part = fmaf(areg[q][j], x[q], part);
            const float sum = row16_allsum_dpp(part);
            const float xv = (bz[qj] - sum) * dinv[qj];
            if (c == cj) x[qj] = xv;
        }
    });
    if (mine) {
#pragma unroll
        for (int q = 0; q < NT; q++)
            if (kTile * q + c < k) F[(size_t)(row0 + grp) * KP + kTile * q + c] = x[q];
    }
}

#endif  // TRMF_F32

// ---- X-side Gram cache: one workgroup (4 waves) per timestamp row ---------------------------------
template <int NT>
__global__ __launch_bounds__(256) void gram_x_kernel(const uint32_t *__restrict__ ptr,
                                                     const uint32_t *__restrict__ idx,
                                                     const real *__restrict__ val,
                                                     const real *__restrict__ Hf,
                                                     const real *__restrict__ W,
                                                     real *__restrict__ G, real *__restrict__ Bv,
                                                     double *__restrict__ lossrow,
                                                     uint32_t row_begin, uint32_t row_end, int k) {
    constexpr int KP = kTile * NT, LD = KP + 1;
    __shared__ real S[KP * LD];
    __shared__ real Sb[KP];
    __shared__ double Sl[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    const uint32_t row = row_begin + blockIdx.x;
    if (row >= row_end) return;
    const uint32_t p0 = ptr[row], p1 = ptr[row + 1];

    GramState<NT> st;
#pragma unroll
    for (int t = 0; t < NT * (NT + 1) / 2; t++) st.acc[t] = typename Mfma16<real>::acc_t{0, 0, 0, 0};
    real wq[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) { st.b[q] = 0; wq[q] = W[(size_t)row * KP + kTile * q + c]; }
    st.loss = 0;
    gram_row<NT, true, true>(st, idx, val, Hf, p0, p1, wave, 4, lane, wq);

#pragma unroll
    for (int q = 0; q < NT; q++) {
        st.b[q] += __shfl_xor(st.b[q], 16, kWave);
        st.b[q] += __shfl_xor(st.b[q], 32, kWave);
    }
    double l = (c == 0) ? st.loss : 0.0;               // every lane of a 16-row holds the same residual
    l = wave_allsum(l);
    if (lane == 0) Sl[wave] = l;

    for (int w = 0; w < 4; w++) {                       // ordered accumulation: deterministic
        if (wave == w) {
            int t = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ti++)
#pragma unroll
                for (int tj = ti; tj < NT; tj++, t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int o = (kTile * ti + Mfma16<real>::row(lane, r)) * LD + kTile * tj + c;
                        S[o] = (w == 0) ? st.acc[t][r] : S[o] + st.acc[t][r];
                    }
            if (g == 0) {
#pragma unroll
                for (int q = 0; q < NT; q++) {
                    const int o = kTile * q + c;
                    Sb[o] = (w == 0) ? st.b[q] : Sb[o] + st.b[q];
                }
            }
        }
        __syncthreads();
    }
    real *Grow = G + (size_t)row * k * k;
    for (int e = threadIdx.x; e < k * k; e += 256) {
        const int s = e / k, t = e - s * k;
        Grow[e] = (s <= t) ? S[s * LD + t] : S[t * LD + s];
    }
    if ((int)threadIdx.x < KP) Bv[(size_t)row * KP + threadIdx.x] = ((int)threadIdx.x < k) ? Sb[threadIdx.x] : real(0);
    if (threadIdx.x == 0) lossrow[row] = (Sl[0] + Sl[1]) + (Sl[2] + Sl[3]);
}

// ---- loss only (f(w_new) of the TRON acceptance test, rf_tron.h:191) ------------------------------
template <int NT>
__global__ __launch_bounds__(256) void loss_kernel(const uint32_t *__restrict__ ptr,
                                                   const uint32_t *__restrict__ idx,
                                                   const real *__restrict__ val,
                                                   const real *__restrict__ Hf,
                                                   const real *__restrict__ W,
                                                   double *__restrict__ lossrow,
                                                   uint32_t row_begin, uint32_t row_end) {
    constexpr int KP = kTile * NT;
    __shared__ double Sl[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15;
    const uint32_t row = row_begin + blockIdx.x;
    if (row >= row_end) return;
    const uint32_t p0 = ptr[row], p1 = ptr[row + 1];
    GramState<NT> st;
    real wq[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) { st.b[q] = 0; wq[q] = W[(size_t)row * KP + kTile * q + c]; }
    st.loss = 0;
    gram_row<NT, false, true>(st, idx, val, Hf, p0, p1, wave, 4, lane, wq);
    double l = (c == 0) ? st.loss : 0.0;
    l = wave_allsum(l);
    if (lane == 0) Sl[wave] = l;
    __syncthreads();
    if (threadIdx.x == 0) lossrow[row] = (Sl[0] + Sl[1]) + (Sl[2] + Sl[3]);
}

}  // namespace trmf
