// cg_kernels.hpp -- the X-solve (temporal factor) on device: TRON reduced to one truncated CG pass
// (rf_tron.h:134-254, 412-505 with the fold of trmf.cpp:603-606), the AR + ridge operator
// (trmf.cpp:70-149) and the cached-Gram Hessian product (replaces trmf.cpp:269-288).
//
// Control flow lives on the device.  Every scalar of the CG (alpha, beta, r^T r, the stop test,
// the accept decision) is re-derived inside each kernel from fixed-order per-block partial sums
// written by the previous kernel, so
//   * the host enqueues the whole solve without a single synchronisation,
//   * every workgroup -- and, with several GPUs, every rank -- derives bit-identical scalars,
//   * a stop is "sticky": all later (already enqueued) iterations do nothing.  The CG records
//     the ITERATION INDEX of the stopping launch (XState::stop_it); launch `it` returns early only
//     when stop_it < it, a value that can only have been written by an EARLIER launch -- the
//     stopping launch itself always sees "not stopped yet" in every one of its workgroups, whatever
//     order they are dispatched in, and closes s and r for all of its rows.
//
// Vectors are T x KP row-major (KP = padded rank); pad columns are zero and stay zero.
#pragma once

#include <vector>

#include "common.hpp"

namespace trmf {

constexpr int kCgHistCap = 64;    // CG iterations the fused path can record (the reference folds to 20)

struct XState {
    double f, fnew, gnorm, cg_rnorm, actred, prered, gs, sr, delta;   // delta: trust-region bound after the step (rf_tron.h:195-215)
    double loss0, loss1;          // sum of squared residuals at w and at w_new (reduce_rows_kernel)
    real cgtol;
    int cg_iter, accepted;
    // fused CG path (hv_tile_kernel, HV_CG_*): r^T r of every iteration, the iteration whose launch detected the
    // stop (kCgRunning until then), buffer that holds the final r
    double rho_hist[kCgHistCap + 2];
    int stop_it, r_parity;
    double rho_direct;            // |-g - H s|^2 of the step evaluated DIRECTLY (persistent kernel; -1 elsewhere): the CG's own r^T r is a recurrence
    int p2p_error;                // a peer-to-peer exchange of the time-sharded CG timed out (sticky; reported by sync())
    long long p2p_diag[4];        // the first timeout: message index, launch (it), peer, expected epoch, flag value seen
};
constexpr int kCgRunning = 0x7fffffff;

// partial-sum arrays: Pbase[slot * kMaxPartials + block]
enum PartialSlot { P_AR = 0, P_VV = 1, P_DOT = 2, P_RR0 = 3, P_RR1 = 4, P_GS = 5, P_SR = 6, P_LQ = 7,
                   P_CG0 = 8,    // fused CG path: <d,Hd>, <r,Hd>, <Hd,Hd> of even iterations (P_CG0..+2), odd ones (+3..+5)
                   P_SS = 14,    // <s,s> (step norm of the TRON line)
                   P_NSLOTS = 15 };

struct XParams {
    int T, k, KP, NT, nlag, midx;      // NT = KP/16: vectors use the column-interleaved layout (colpos)
    double lambdaI, lambdaAR, eps_cg;
    // full-observation path (missing == 0, trmf.cpp:155-215): one shared Gram H^T H for every
    // timestamp (gstride == 0) and fun = base + 0.5*(tr Y^T Y + sum_i w_i^T G w_i - 2 b_i.w_i)
    int full;
    int pstride;                       // entries between the partial-sum arrays (>= kMaxPartials, >= tile count)
    size_t gstride;                    // elements between consecutive per-timestamp Grams (k*k, packed_gram_elems(k), or 0)
    double trYTY;
};

// Fixed-order block reductions (blockDim.x == 256 = 4 wavefronts).  Butterfly inside the wavefront
// (no barrier), then the four wave sums are combined in a fixed order: deterministic, identical in
// every thread, two barriers instead of a nine-barrier LDS tree.
// (Round 5: the xor butterfly `v += __shfl_xor(v, m)` for m = 1 .. 32 compiled to twelve ds_bpermute_b32 per sum -- LDS-pipe round trips
// in a dependent chain, ~0.25 us per block sum, twice per CG pass.  The same tree -- pairs, quads, halves of a 16-lane row, rows, (r0 + r1)
// + (r2 + r3) -- through DPP row operations and four lane reads: every add has the operands of the butterfly's, possibly swapped, and
// IEEE addition is commutative, so the result is BIT-IDENTICAL to the butterfly's in every lane.)
template <int CTRL> __device__ __forceinline__ double dpp_move_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo));
}
__device__ __forceinline__ double read_lane_f64(double v, int lane) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, lane);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}
__device__ __forceinline__ double wave_butterfly_sum(double v) {
    v += dpp_move_f64<0xB1>(v);        // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_move_f64<0x4E>(v);        // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_move_f64<0x141>(v);       // row_half_mirror: the other quad of the 8-lane half
    v += dpp_move_f64<0x140>(v);       // row_mirror: the other half of the 16-lane row
    const double r0 = read_lane_f64(v, 0), r1 = read_lane_f64(v, 16), r2 = read_lane_f64(v, 32), r3 = read_lane_f64(v, 48);
    // lanes of rows 0 / 1 hold (r0 + r1) after the butterfly's xor-16 step, rows 2 / 3 (r2 + r3); its xor-32 step adds the two
    return (r0 + r1) + (r2 + r3);
}
// NW = wavefronts of the workgroup: 4 (256 threads, every kernel but the wide tiles) or 8 (512 threads: the wide tiles of
// cg_persist_kernel / hv_tile_kernel / cg_close_kernel); wave sums combined pairwise in index order
template <int NW>
__device__ __forceinline__ double wave_sums_fixed(const double *w) {
    static_assert(NW == 4 || NW == 8, "256 or 512 threads");
    if constexpr (NW == 4) return (w[0] + w[1]) + (w[2] + w[3]);
    else return ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
}
template <int NW = 4>
__device__ __forceinline__ double block_allsum(double v, double *smem /* >= 16 doubles */) {
    v = wave_butterfly_sum(v);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = wave_sums_fixed<NW>(smem);
    __syncthreads();
    return r;
}
// three sums sharing the barriers
template <int NW = 4>
__device__ __forceinline__ void block_allsum3(double &a, double &b, double &c, double *smem /* >= 24 doubles */) {
    a = wave_butterfly_sum(a); b = wave_butterfly_sum(b); c = wave_butterfly_sum(c);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smem[w] = a; smem[NW + w] = b; smem[2 * NW + w] = c; }
    __syncthreads();
    a = wave_sums_fixed<NW>(smem);
    b = wave_sums_fixed<NW>(smem + NW);
    c = wave_sums_fixed<NW>(smem + 2 * NW);
    __syncthreads();
}
// Sum of a partial array written by a producer kernel with `np` blocks; identical in every block.
__device__ __forceinline__ double sum_partials(const double *__restrict__ P, int np, double *smem) {
    double v = 0;
    for (int i = threadIdx.x; i < np; i += 256) v += P[i];
    return block_allsum(v, smem);
}
// two partial arrays of the same length in one pass
__device__ __forceinline__ void sum_partials2(const double *__restrict__ P, const double *__restrict__ Q, int np,
                                              double &sp, double &sq, double *smem) {
    double a = 0, b = 0, c = 0;
    for (int i = threadIdx.x; i < np; i += 256) { a += P[i]; b += Q[i]; }
    block_allsum3(a, b, c, smem);
    sp = a; sq = b;
}
__device__ __forceinline__ bool cg_stopped(real rho, real cgtol) {           // rf_tron.h:444-446
    return sqrt((double)rho) <= (double)cgtol;
}

// ---- single-block reduction of a per-row array into a scalar ----------------------------------------
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const double *__restrict__ src, int n,
                                                          double *__restrict__ dst) {
    __shared__ double smem[256];
    double v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += src[i];
    v = block_allsum(v, smem);
    if (threadIdx.x == 0) *dst = v;
}
#endif

// ---- ||v||_F^2 partials (verbose log lines trmf.cpp:661,672,687) -----------------------------------
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const real *__restrict__ v, size_t n,
                                                            double *__restrict__ P) {
    __shared__ double smem[256];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc += (double)v[i] * (double)v[i];
    acc = block_allsum(acc, smem);
    if (threadIdx.x == 0) P[blockIdx.x] = acc;
}
#endif

// ---- AR + ridge part of the operator, tiled over (time x column group) in LDS ------------------------------
// base = lambdaI*v + lambdaAR*AR'(AR(v)) for the unfused path (lag sets whose reach does not fit hv_tile_kernel's
// whole-row tiles: the paper scripts' lags reach back 191 timestamps).  The AR operator never mixes latent
// dimensions, so a workgroup owns TI consecutive timestamps of kArCols neighbouring columns: it stages the
// operand rows [i0-midx, i0+TI+midx) of those columns once (every element of v is read from global memory
// ~(TI+2 midx)/TI times instead of 2|L|+1 times), forms the residuals of rows [i0, i0+TI+midx) there
// (the halo is recomputed, not exchanged) and applies the adjoint to its own rows.  Arithmetic and rounding
// sequence of trmf.cpp:99-149 (residual in double, every accumulation into the result rounded to val_type).
// FUSE_DIR: the operand is the new direction d + (beta-1) d + r (rf_tron.h:494-502), written once for the own rows.
// Partial sums of r^2 (AR part of fun) and v^2 (ridge part) go to slot (tile0 + blockIdx.x) * gridDim.y + blockIdx.y (tile-major:
// the slots of a rank's tiles are one contiguous range).
constexpr int kArCols = 8;
constexpr int kArPitch = kArCols + 1; // LDS row pitch in elements: a thread owns kArU = 8 CONSECUTIVE rows of one column, so the 8 row groups of a
                                      // wavefront sit 8 rows apart -- with 9 elements per row they fall into different bank groups (fp32 and fp64)
constexpr int kArU = 8;               // rows per thread
constexpr int kArRun = 4;             // consecutive lags handled as one sliding window: 11 LDS reads for 32 products instead of 32
constexpr int kArThreads = 1024;      // one workgroup per CU (its LDS tile is ~118 KB at the paper's lag set)
// One pass (every residual of the tile in registers at once) lets the residual rows reuse the operand rows' LDS.
__host__ __device__ inline bool ar_tile_one_pass(int TI, int midx) { return TI + midx <= kArThreads / kArCols * kArU; }
__host__ __device__ inline size_t ar_tile_lds_bytes(int TI, int midx, int nlag) {
    const size_t vbytes = ((size_t)(TI + 2 * midx + kArU) * kArPitch * sizeof(real) + 15) / 16 * 16;
    const size_t rbytes = (size_t)(TI + midx + kArU) * kArPitch * sizeof(double);
    const size_t rows = ar_tile_one_pass(TI, midx) ? (vbytes > rbytes ? vbytes : rbytes) : vbytes + rbytes;
    return rows + ((size_t)nlag * kArCols * sizeof(real) + 15) / 16 * 16;
}
// The lag set as a list of steps in lag order: entry = (index of the first lag) * 2 + (1: that lag and the next three are
// consecutive integers, handled as one window; 0: a single lag).  Same accumulation order as a plain loop over the lags.
inline std::vector<uint32_t> ar_lag_steps(const uint32_t *lags, int nlag) {
    std::vector<uint32_t> steps;
    for (int l = 0; l < nlag;) {
        bool run = l + kArRun <= nlag;
        for (int j = 1; run && j < kArRun; j++) run = lags[l + j] == lags[l] + (uint32_t)j;
        steps.push_back((uint32_t)l * 2 + (run ? 1 : 0));
        l += run ? kArRun : 1;
    }
    return steps;
}
// fixed-order sum over a workgroup of kArThreads threads (16 wavefronts); identical in every thread
__device__ __forceinline__ double block_allsum_wide(double v, double *smem /* >= 16 doubles */) {
    v = wave_butterfly_sum(v);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int w = 0; w < kArThreads / 64; w++) r += smem[w];
    __syncthreads();
    return r;
}
__device__ __forceinline__ void block_allsum3_wide(double &a, double &b, double &c, double *smem /* >= 48 doubles */) {
    a = wave_butterfly_sum(a); b = wave_butterfly_sum(b); c = wave_butterfly_sum(c);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smem[w] = a; smem[16 + w] = b; smem[32 + w] = c; }
    __syncthreads();
    a = 0; b = 0; c = 0;
#pragma unroll
    for (int i = 0; i < kArThreads / 64; i++) { a += smem[i]; b += smem[16 + i]; c += smem[32 + i]; }
    __syncthreads();
}
// MODE AR_PLAIN: base = AR/ridge part for the operand v (gradient at w, H s, first CG product).
// MODE AR_CG_STEP: CG iteration it >= 1 of the unfused path, the counterpart of hv_tile_kernel<HV_CG_STEP>: the three
//   dot products <d,Hd>, <r,Hd>, <Hd,Hd> of iteration it-1 (per-workgroup partials written by apply_kernel, `np` of
//   them) give alpha, r^T r of the new residual by the recurrence rho' = rho - 2 alpha <r,Hd> + alpha^2 <Hd,Hd>, the
//   stop test and beta; the staged rows are updated on the fly (s += alpha d and r' = r - alpha Hd on the own rows,
//   d' = r' + beta d on every staged row: the halo is recomputed from the previous iteration's buffers, which are
//   ping-pong so that no workgroup overwrites what a neighbour still reads).  A stop is recorded as the iteration
//   index (XState::stop_it, see hv_tile_kernel); the stopping launch closes s and r and skips the operator.
enum ArMode { AR_PLAIN = 0, AR_CG_STEP = 1 };
struct ArVecs {
    const real *v;        // PLAIN: operand;  CG_STEP: previous direction d
    const real *r_in;     // CG_STEP: residual of the previous iteration
    const real *hd_in;    // CG_STEP: H d of the previous iteration
    real *s;              // CG_STEP: the step (own rows, in place)
    real *d_out, *r_out;  // CG_STEP: new direction / residual
    unsigned int *note;   // CG_STEP: pinned host word the deciding workgroup reports (solve, step, stopped) to, or null (xsolve)
    unsigned int note_seq;
};
template <int MODE>
__global__ __launch_bounds__(kArThreads) void ar_tile_kernel(XParams p, XState *__restrict__ st, ArVecs a, int np, int it, int last,
                                                             const uint32_t *__restrict__ lag_set,
                                                             const uint32_t *__restrict__ steps, int nsteps,
                                                             const real *__restrict__ theta,
                                                             real *__restrict__ base, double *__restrict__ Pbase, int TI,
                                                             int tile0) {
    // tile0: index of the first time tile of this launch (a rank of the time-sharded unfused CG launches its own tiles only);
    // the partial sums go to slot (tile index) * gridDim.y + column group, so that a rank's slots are one contiguous range
    extern __shared__ __attribute__((aligned(16))) unsigned char ar_smem[];
    __shared__ double smem[48];
    constexpr bool STEP = MODE == AR_CG_STEP;
    const int tid = threadIdx.x, T = p.T, KP = p.KP, Hh = p.midx, nlag = p.nlag;
    const int rowsV = TI + 2 * Hh, rowsR = TI + Hh;
    // vs[row][col] (operand rows) and rs[row][col] (residual rows), pitch kArPitch, kArU spare rows each; in the
    // one-pass form rs REUSES the memory of vs (every thread holds its residuals in registers across a barrier)
    const bool one_pass = ar_tile_one_pass(TI, Hh);
    const size_t vbytes = ((size_t)(rowsV + kArU) * kArPitch * sizeof(real) + 15) / 16 * 16;
    const size_t rbytes = (size_t)(rowsR + kArU) * kArPitch * sizeof(double);
    real *vs = reinterpret_cast<real *>(ar_smem);
    double *rs = reinterpret_cast<double *>(ar_smem + (one_pass ? 0 : vbytes));
    real *ths = reinterpret_cast<real *>(ar_smem + (one_pass ? (vbytes > rbytes ? vbytes : rbytes) : vbytes + rbytes));   // ths[l][col]
    if (STEP && st->stop_it < it) return;               // an EARLIER launch ended the CG
    const int i0 = (blockIdx.x + tile0) * TI, i1 = min(i0 + TI, T), c0 = blockIdx.y * kArCols;
    // The workgroup is a serial chain  partial sums -> alpha, beta -> operand rows -> residuals -> adjoint: the operand
    // loads do not depend on the scalars, so the first batch is requested before anything else (and every later batch
    // before the previous one is processed).
    constexpr int kBatch = 4;
    struct Batch { real v[kBatch], hd[kBatch], r[kBatch], s[kBatch]; };
    const int nV = rowsV * kArCols;
    auto request = [&](int e0) {
        Batch b;
#pragma unroll
        for (int m = 0; m < kBatch; m++) {
            const int e = e0 + m * kArThreads + tid, rr = e / kArCols, cc = e - rr * kArCols, i = i0 - Hh + rr;
            b.v[m] = 0; b.hd[m] = 0; b.r[m] = 0; b.s[m] = 0;
            if (e < nV && i >= 0 && i < T) {
                const size_t ge = (size_t)i * KP + c0 + cc;
                b.v[m] = a.v[ge];
                if (STEP) {
                    b.hd[m] = a.hd_in[ge]; b.r[m] = a.r_in[ge];
                    if (i >= i0 && i < i1) b.s[m] = a.s[ge];
                }
            }
        }
        return b;
    };
    Batch cur = request(0);
    // CG step: this thread's entries of the three dot-product arrays, requested with the first batch -- the Theta staging
    // below drains every outstanding load (loop + LDS store), so anything requested after it costs a second round trip
    double pre[3] = {0, 0, 0};
    if (STEP && tid < np) {
        const double *Pp = Pbase + (size_t)(P_CG0 + 3 * ((it - 1) & 1)) * p.pstride;
        pre[0] = Pp[tid]; pre[1] = Pp[(size_t)p.pstride + tid]; pre[2] = Pp[2 * (size_t)p.pstride + tid];
    }
    for (int e = tid; e < nlag * kArCols; e += kArThreads) {
        const int l = e / kArCols, cc = e - l * kArCols, t = collog(c0 + cc, p.NT);
        ths[e] = t < p.k ? theta[(size_t)t * nlag + l] : real(0);
    }
    real tmp = 0, alpha = 0, nalpha = 0;
    bool stopped = false;
    if (STEP) {
        const double *Pp = Pbase + (size_t)(P_CG0 + 3 * ((it - 1) & 1)) * p.pstride;
        double dHd = pre[0], rHd = pre[1], HH = pre[2];
        for (int i = tid + kArThreads; i < np; i += kArThreads) { dHd += Pp[i]; rHd += Pp[(size_t)p.pstride + i]; HH += Pp[2 * (size_t)p.pstride + i]; }
        block_allsum3_wide(dHd, rHd, HH, smem);
        const double rho_prev_d = st->rho_hist[it - 1];
        const real rho_prev = (real)rho_prev_d;
        alpha = rho_prev / (real)dHd;                                            // rf_tron.h:460
        nalpha = -alpha;
        const double ad = (double)alpha;
        const double rho_d = fmax(rho_prev_d - 2.0 * ad * rHd + ad * ad * HH, 0.0);   // |r - alpha Hd|^2
        const real rho = (real)rho_d;
        stopped = last || cg_stopped(rho, st->cgtol);                            // top of iteration `it`, rf_tron.h:444-446
        tmp = rho / rho_prev - (real)1.0;                                        // rf_tron.h:495-497
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
            st->rho_hist[it] = rho_d;
            if (stopped) { st->stop_it = it; st->r_parity = it & 1; }
            else st->cg_iter = it + 1;
            if (a.note) __hip_atomic_store(a.note, (a.note_seq << 8) | ((unsigned int)it << 1) | (stopped ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // (1) operand rows -> LDS (zeros outside [0, T)); CG step: s, r, d of the own rows go out
    double vv = 0, ar2 = 0;
    for (int e0 = 0; e0 < nV; e0 += kBatch * kArThreads) {
        Batch nxt{};
        if (e0 + kBatch * kArThreads < nV) nxt = request(e0 + kBatch * kArThreads);
#pragma unroll
        for (int m = 0; m < kBatch; m++) {
            const int e = e0 + m * kArThreads + tid, rr = e / kArCols, cc = e - rr * kArCols, i = i0 - Hh + rr;
            if (e >= nV) break;
            real x = 0;
            if (i >= 0 && i < T) {
                const size_t ge = (size_t)i * KP + c0 + cc;
                x = cur.v[m];
                const bool own = i >= i0 && i < i1;
                if (STEP) {
                    const real rnew = fma(nalpha, cur.hd[m], cur.r[m]);            // r -= alpha Hd     (rf_tron.h:489-490)
                    if (own) a.s[ge] = fma(alpha, x, cur.s[m]);                    // s += alpha d      (rf_tron.h:461)
                    x = fma(tmp, x, x); x = x + rnew;                              // d = beta d + r    (rf_tron.h:497-499)
                    if (own) { a.r_out[ge] = rnew; a.d_out[ge] = x; }
                }
                if (own) vv += (double)x * (double)x;
            }
            vs[rr * kArPitch + cc] = x;
        }
        cur = nxt;
    }
    if (STEP && stopped) return;                        // s and r are final; no further product
    __syncthreads();
    const bool ar_on = nlag > 0 && p.lambdaAR > 0;
    // (2) residuals of rows [i0, i0+TI+midx) (trmf.cpp:110-113 / 136-139): r = x_i - sum_l Theta_l x_{i-L_l}, lags in
    //     order.  A thread keeps ONE column and kArU consecutive rows; over a run of kArRun consecutive lags the
    //     operands of its rows overlap, so one window of kArU + kArRun - 1 LDS reads feeds kArU * kArRun products.
    //     The step list and the lag offsets are wave-uniform (scalar loads).
    constexpr int kGroups = kArThreads / kArCols, kWin = kArU + kArRun - 1;
    const int cc = tid % kArCols, g = tid / kArCols;
    auto residuals = [&](int rb, double (&res)[kArU]) {
        const real *own = vs + (size_t)(rb + Hh) * kArPitch + cc;
#pragma unroll
        for (int u = 0; u < kArU; u++) res[u] = (double)own[u * kArPitch];
        for (int sidx = 0; sidx < nsteps; sidx++) {
            const uint32_t step = steps[sidx];
            const int l = (int)(step >> 1);
            const real *win = own - (ptrdiff_t)lag_set[l] * kArPitch;
            if (step & 1) {
                real x[kWin], th[kArRun];
#pragma unroll
                for (int m = 0; m < kWin; m++) x[m] = win[(m - (kArRun - 1)) * kArPitch];      // row rb + m - 3 - L
#pragma unroll
                for (int j = 0; j < kArRun; j++) th[j] = ths[(l + j) * kArCols + cc];
#pragma unroll
                for (int j = 0; j < kArRun; j++)
#pragma unroll
                    for (int u = 0; u < kArU; u++) {
                        const real prod = th[j] * x[u - j + kArRun - 1];
                        res[u] -= (double)prod;
                    }
            } else {
                const real th = ths[l * kArCols + cc];
#pragma unroll
                for (int u = 0; u < kArU; u++) {
                    const real prod = th * win[u * kArPitch];
                    res[u] -= (double)prod;
                }
            }
        }
    };
    auto keep_residuals = [&](int rb, const double (&res)[kArU]) {
#pragma unroll
        for (int u = 0; u < kArU; u++) {
            const int rr = rb + u, i = i0 + rr;
            if (rr < rowsR) {
                const double rv = (i >= Hh && i < T) ? res[u] : 0.0;          // rows outside [midx, T): exact zeros
                if (rr < TI) ar2 += rv * rv;
                rs[(size_t)rr * kArPitch + cc] = rv;
            }
        }
    };
    real xown[kArU];                                    // one pass: the thread's own operand rows, for step (3)
#pragma unroll
    for (int u = 0; u < kArU; u++) xown[u] = 0;
    if (one_pass) {
        const int rb = g * kArU;
        double res[kArU];
        const bool act = ar_on && rb < rowsR;
        if (rb < i1 - i0) {
#pragma unroll
            for (int u = 0; u < kArU; u++) xown[u] = vs[(size_t)(rb + u + Hh) * kArPitch + cc];
        }
        if (act) residuals(rb, res);
        __syncthreads();                                // every read of the operand rows is done: their memory becomes rs
        if (act) keep_residuals(rb, res);
    } else if (ar_on) {
        for (int rb = g * kArU; rb < rowsR; rb += kGroups * kArU) {
            double res[kArU];
            residuals(rb, res);
            keep_residuals(rb, res);
        }
    }
    __syncthreads();
    // (3) base = lambdaI*v + lambdaAR*AR'(r) for the own rows: same rounding sequence as the reference's
    //     time-ordered scatter loop (own-row term first, then the lags in order).  Residual rows outside [midx, T)
    //     were stored as exact zeros, so the lagged terms need no range test.
    for (int rb = g * kArU; rb < i1 - i0; rb += kGroups * kArU) {
        real o[kArU];
        const double *rown = rs + (size_t)rb * kArPitch + cc;
#pragma unroll
        for (int u = 0; u < kArU; u++) {
            const real x = one_pass ? xown[u] : vs[(size_t)(rb + u + Hh) * kArPitch + cc];
            if (p.lambdaI == 0) o[u] = 0;
            else if (p.lambdaI == 1) o[u] = x;
            else o[u] = (real)(p.lambdaI * (double)x);
            if (ar_on && i0 + rb + u >= Hh) o[u] = (real)((double)o[u] + p.lambdaAR * rown[u * kArPitch]);
        }
        if (ar_on) {
            for (int sidx = 0; sidx < nsteps; sidx++) {
                const uint32_t step = steps[sidx];
                const int l = (int)(step >> 1);
                const double *win = rown + (size_t)lag_set[l] * kArPitch;
                if (step & 1) {
                    double y[kWin], lth[kArRun];
#pragma unroll
                    for (int m = 0; m < kWin; m++) y[m] = win[m * kArPitch];                       // row rb + m + L
#pragma unroll
                    for (int j = 0; j < kArRun; j++) lth[j] = p.lambdaAR * (double)ths[(l + j) * kArCols + cc];
#pragma unroll
                    for (int j = 0; j < kArRun; j++)
#pragma unroll
                        for (int u = 0; u < kArU; u++) o[u] = (real)((double)o[u] - y[u + j] * lth[j]);
                } else {
                    const double lth = p.lambdaAR * (double)ths[l * kArCols + cc];
#pragma unroll
                    for (int u = 0; u < kArU; u++) o[u] = (real)((double)o[u] - win[u * kArPitch] * lth);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kArU; u++)
            if (rb + u < i1 - i0) base[(size_t)(i0 + rb + u) * KP + c0 + cc] = o[u];
    }
    ar2 = block_allsum_wide(ar2, smem);
    vv = block_allsum_wide(vv, smem);
    if (tid == 0) {
        const size_t slot = (size_t)(blockIdx.x + tile0) * gridDim.y + blockIdx.y;
        Pbase[P_AR * (size_t)p.pstride + slot] = ar2;
        Pbase[P_VV * (size_t)p.pstride + slot] = vv;
    }
}

// ---- out = lambdaI*v + lambdaAR*AR'(v) + G.v (- b) ; partial of <dotwith, out> ----------------------
// Multi-GPU (TrmfSessionImpl::cg_shard): every rank evaluates its own block of timestamps -- the k x k Gram per
// timestamp is the only HBM-sized stream of a CG step, and the Hessian is block-diagonal there (trmf.cpp:269-288)
// -- and the blocks of `out` and of the partial sums are all-gathered; everything else of the CG stays replicated.
// grad (trmf.cpp:99-123 + 247-267) when minus_b, Hessian-vector product (trmf.cpp:125-149 + 269-288)
// otherwise.  The AR + ridge part comes in as `base` (ar_tile_kernel); this kernel adds the cached-Gram product.
// One thread per (row, column); `rpb` rows per block; the row's v is staged in LDS.
// dot_mode 0: <out,out>   1: <v,out>
//
// PACKED (unfused path, per-timestamp Grams): G_i is symmetric and this product is its only reader, so gram_x_kernel keeps
// the upper triangle only -- element (s, t), s <= t, at s k - s (s - 1) / 2 + (t - s), packed_gram_elems(k) per
// timestamp -- and the stream of a CG step is halved (config 5: 1.64 -> 0.83 GB).  The rpb packed triangles of a row
// group are contiguous: the workgroup copies them to LDS with coalesced 16-byte loads, requested one group ahead (they
// are in flight under the previous group's products), and thread (row, t) reads column t above the diagonal and row t
// beyond it, in the same order s = 0 .. k-1 as the full form: the results are bit-identical.
__host__ __device__ inline size_t packed_gram_elems(int k) { return (((size_t)k * (k + 1) / 2) + 1) & ~(size_t)1; }
#ifndef TRMF_APPLY_ABL
#define TRMF_APPLY_ABL 0              // scripts/ubench/apply_packed.hip: 1 = no products, 2 = no LDS copy either
#endif
// STAGES = 16-byte loads per thread and row group >= rpb * packed_gram_elems(k) / 512 (17 at k = 64), a template
// parameter because the loads and the LDS copy must be straight-line code: with a (uniform) branch per load the
// compiler waits for each load before issuing the next (measured: 10.9 us per row group, 296 us per launch at config 5,
// slower than the full form).  Loads past the group are clamped to the last pair of G; the LDS buffer is STAGES * 512 reals.
__host__ __device__ inline int apply_stages(int k) {
    const size_t need = ((size_t)(256 / k) * packed_gram_elems(k) + 511) / 512;
    return need <= 5 ? 5 : need <= 10 ? 10 : 17;
}

template <bool PACKED, int STAGES = 1>
__global__ __launch_bounds__(256, 2) void apply_kernel(XParams p, const XState *__restrict__ st, int cg_it,
                                                    const real *__restrict__ v, const real *__restrict__ rvec,
                                                    const real *__restrict__ base,
                                                    const real *__restrict__ G,
                                                    const real *__restrict__ Bv, int minus_b,
                                                    real *__restrict__ out, int dot_mode,
                                                    double *__restrict__ Pdot, int rpb,
                                                    int row0, int nrows, int slot0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char apply_smem[];   // k*k reals when the Gram is shared (gstride == 0)
    __shared__ double smem[256];
    __shared__ real vs[256];
    // cg_it >= 0: this is H d of CG iteration cg_it (d = v, residual = rvec): nothing to do once the CG has stopped at
    // or before that iteration (stop_it is written by the ar_tile launch of the iteration, an EARLIER launch); the
    // partials are then <d,Hd>, <r,Hd>, <Hd,Hd> in the slots P_CG0 + 3 (cg_it & 1) .. + 2 (as hv_tile_kernel emits them)
    const bool cg = cg_it >= 0;
    if (cg && st->stop_it <= cg_it) return;
    const int k = p.k, KP = p.KP;
    const bool shared_gram = !PACKED && p.gstride == 0;          // full-observation path: one H^T H for every timestamp
    real *Gs = reinterpret_cast<real *>(apply_smem);
    if (shared_gram) {
        for (int e = threadIdx.x; e < k * k; e += 256) Gs[e] = G[e];
        __syncthreads();
    }
    const int lr = threadIdx.x / k, t = threadIdx.x - lr * k;     // t: logical column
    const int tp = colpos(t, p.NT);                                 // its position in a vector row
    const bool active_lane = lr < rpb;
    double dot = 0, lq = 0, rhd = 0, hh = 0;
    // rows [row0, row0 + nrows): all of them, or this rank's block when the Gram product is sharded across GPUs
    // (the partial sums then land in this rank's slot range [slot0, slot0 + gridDim.x) and are all-gathered)
    const int ngroups = (nrows + rpb - 1) / rpb;
    typedef real Pair __attribute__((ext_vector_type(2)));
    Pair stage[STAGES];
    const int Lp = (int)p.gstride;                                  // PACKED: elements per timestamp (even)
    // no predicate on the loads: the tail of a short last group re-reads the final pair of G
    const size_t lastpair = (size_t)p.T * (size_t)Lp - 2;
#define TRMF_APPLY_REQUEST(GRP)                                                                                   \
    {                                                                                                             \
        const size_t first = (size_t)(row0 + (GRP) * rpb) * (size_t)Lp + 2 * threadIdx.x;                         \
        _Pragma("unroll") for (int j = 0; j < STAGES; j++)                                                        \
            stage[j] = *reinterpret_cast<const Pair *>(G + min(first + 512 * j, lastpair));                       \
    }
    if constexpr (PACKED)
        if ((int)blockIdx.x < ngroups) TRMF_APPLY_REQUEST((int)blockIdx.x)
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int i = row0 + grp * rpb + lr;
        const bool active = active_lane && grp * rpb + lr < nrows;
        // every per-row operand is requested here, ahead of the next group's Grams: loads retire in order, a wait for one
        // of these issued after TRMF_APPLY_REQUEST would wait for the whole group
        real x = 0, o = 0, rv = 0, bv = 0;
        if (active) {
            x = v[(size_t)i * KP + tp];
            o = base[(size_t)i * KP + tp];                         // lambdaI*v + lambdaAR*AR'(v), ar_tile_kernel
            if (cg) rv = rvec[(size_t)i * KP + tp];
            if (minus_b) bv = Bv[(size_t)i * KP + t];
        }
        __syncthreads();
        vs[threadIdx.x] = x;
        if constexpr (PACKED) {
#pragma unroll
            for (int j = 0; j < STAGES; j++) {
                if (TRMF_APPLY_ABL == 2) { asm volatile("" ::"v"(stage[j])); continue; }
                *reinterpret_cast<Pair *>(Gs + 2 * threadIdx.x + 512 * j) = stage[j];
            }
        }
        __syncthreads();
        if constexpr (PACKED)
            if (grp + (int)gridDim.x < ngroups) TRMF_APPLY_REQUEST(grp + (int)gridDim.x)
        if (active) {
            // cached Gram: sum_s G_i[s][t] * v_i[s]
            const real *vi = vs + lr * k;
            double acc = 0;
            if constexpr (PACKED) {
                const real *U = Gs + lr * Lp;
                int up = t;                                         // (s, t) for s <= t: s k - s (s - 1) / 2 + t - s
                const int across = t * k - t * (t - 1) / 2 - t;     // (t, s) for s > t: this + s
                if (TRMF_APPLY_ABL) acc = (double)U[t];
                for (int s0 = 0; s0 < (TRMF_APPLY_ABL ? 0 : k); s0 += 8) {   // eight LDS reads in flight, then their products in order
                    real u[8], w[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int sc = min(s0 + j, k - 1);
                        u[j] = U[sc <= t ? up : across + sc];
                        w[j] = vi[sc];
                        up += k - 1 - sc;
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (s0 + j < k) acc += (double)u[j] * (double)w[j];
                }
            } else if (shared_gram) {
#pragma unroll 8
                for (int s = 0; s < k; s++) acc += (double)Gs[s * k + t] * (double)vi[s];
            } else {
                const real *Gi = G + (size_t)i * p.gstride + t;
#pragma unroll 8
                for (int s = 0; s < k; s++) acc += (double)Gi[(size_t)s * k] * (double)vi[s];
            }
            if (minus_b) {
                const double bb = (double)bv;
                lq += (double)x * (acc - 2.0 * bb);                          // w.(Gw) - 2 b.w
                acc -= bb;
            }
            o = (real)((double)o + acc);
            out[(size_t)i * KP + tp] = o;
            dot += (double)(dot_mode ? x : o) * (double)o;
            if (cg) {
                rhd += (double)rv * (double)o;                               // <r,Hd>
                hh += (double)o * (double)o;                                 // <Hd,Hd>
            }
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

// ---- fused Hessian-vector / gradient kernel, tiled over time in LDS --------------------------------
// One launch = [direction update (FUSE_DIR)] + AR residual + AR adjoint + cached-Gram product
// (ar_tile_kernel + apply_kernel above, same arithmetic and rounding sequence).  A workgroup owns
// TI consecutive timestamps: it stages the operand rows [i0-midx, i0+TI+midx) in LDS, forms the AR
// residuals of rows [i0, i0+TI+midx) there (the halo is recomputed, not exchanged), and multiplies its
// TI cached Grams.  Used when the halo fits LDS (hv_tile_lds_bytes); otherwise the two-kernel path runs.
//
// The kernel is LATENCY-bound, not bandwidth-bound: the launch is a single wave of workgroups, each a
// serial chain  partial sums -> operand -> residual -> Gram product.  So every global load of the chain
// is issued at the very top, in the order it will be consumed (vmcnt retires in order): CG partials,
// Theta/lag set, operand rows, and finally the thread's whole slice of the cached Gram (KQ 16-byte
// loads, one timestamp row x 16/sizeof(real) columns per thread).  The Gram -- the only HBM-sized
// stream of the CG, T*k*k values -- is then in flight for the entire launch while the chain runs on
// registers and LDS underneath it.  Barriers wait on lgkmcnt only, so the loads stay outstanding.
// dynamic LDS = hv_tile_lds_bytes(TI, midx, KP, nlag, k)
// Row pitch (in doubles) of the AR residual rows in LDS.  The adjoint phase reads them with one 16/32-byte vector per
// thread, thread = (timestamp row, column group): with the pitch equal to the bytes a row's threads actually read
// (k rounded up to the vector) the 64 threads of a wavefront read ONE contiguous 2 KB range -- conflict-free.  (Round 2
// used the padded rank: at k = 40 that is 384 bytes, rows two apart fall on the same banks, and 37 % of the LDS cycles
// of the kernel were bank conflicts, profiles/r02_pmc_hv_tile.txt.)
__host__ __device__ constexpr int hv_res_pitch(int k) { return (k + 7) / 8 * 8; }    // = KQ: a compile-time constant of the kernel
__host__ __device__ inline size_t hv_tile_lds_bytes(int TI, int midx, int KP, int nlag = 0, int k = 0) {
    const size_t a = ((size_t)(TI + 2 * midx) * KP * sizeof(real) + 15) / 16 * 16;
    const size_t b = ((size_t)(TI + midx) * (k > 0 ? hv_res_pitch(k) : KP) * sizeof(double) + 15) / 16 * 16;
    const size_t c = ((size_t)TI * KP * sizeof(real) + 15) / 16 * 16;                                  // own-row r (fused CG)
    return a + b + c + (size_t)nlag * KP * (sizeof(double) + sizeof(real)) + (size_t)nlag * sizeof(int);  // + lambdaAR*Theta, Theta, lag_set
}
constexpr int kHvThetaRegs = 3;                  // Theta elements per thread loaded ahead of the scalar prologue
#ifndef TRMF_HV_RESU
#define TRMF_HV_RESU 2
#endif
constexpr int kHvResU = TRMF_HV_RESU;                       // AR residual work items (4 columns each) a thread carries through the lag loop
constexpr int kHvOperandRegs = 12;               // operand elements per thread requested ahead of the Gram
#ifndef TRMF_HV_UPFRONT
#define TRMF_HV_UPFRONT 96
#endif
constexpr int kHvGramUpfront = TRMF_HV_UPFRONT;  // registers of the Gram slice requested before the LDS phases
constexpr int kHvGramPad = 4;                    // elements allocated past the Gram cache (vector tail reads)
// Gram columns per thread: one 16-byte load per Gram row up to rank 40, 8-byte loads above (the thread's
// slice, KQ loads, has to stay within ~160 of the 256 registers)
__host__ __device__ constexpr int hv_vec(int KQ) { return (KQ <= 40 ? 16 : 8) / (int)sizeof(real); }
__host__ __device__ constexpr int hv_kq(int k) { return (k + 7) / 8 * 8; }
__host__ __device__ inline int hv_tile_rows(int k, int threads = 256) { const int vec = hv_vec(hv_kq(k)); return threads / ((k + vec - 1) / vec); }

// VEC consecutive Gram entries through a raw buffer load: address = wave-uniform descriptor base (the
// tile's first Gram) + scalar offset (Gram row j) + one 32-bit lane offset -- no per-load vector address.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_rsrc(const void *base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(bytes < 0x7fffffffu ? bytes : 0x7fffffffu),
                                             0x00020000);
}
// element access at a BYTE offset; a byte offset at or past the descriptor's size reads 0 / drops the store
template <typename R = real>
__device__ __forceinline__ R buffer_load_real(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
    if constexpr (sizeof(R) == 4) return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off, 0, 0));
    else return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b64(rsrc, byte_off, 0, 0));
}
template <typename R = real>
__device__ __forceinline__ void buffer_store_real(__amdgpu_buffer_rsrc_t rsrc, int byte_off, R x) {
    if constexpr (sizeof(R) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, x), rsrc, byte_off, 0, 0);
    else {
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, x), rsrc, byte_off, 0, 0);
    }
}
template <int VEC> struct GramVec { real c[VEC]; };
template <int VEC>
__device__ __forceinline__ GramVec<VEC> gram_load(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
#if defined(TRMF_HV_ABL) && (TRMF_HV_ABL & 1)
    GramVec<VEC> z; for (int c = 0; c < VEC; c++) z.c[c] = (real)voff; return z;
#endif
    if constexpr (VEC * sizeof(real) == 16)
        return __builtin_bit_cast(GramVec<VEC>, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
    else
        return __builtin_bit_cast(GramVec<VEC>, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
}

// One kernel, four roles (compile-time, so that the 20 CG launches carry nothing of the others; a runtime
// flag in this kernel cost +0.3 ms per solve when measured):
//   HV_PLAIN     out = H v,            partial <v, Hv>                     (H s of the acceptance test)
//   HV_GRAD      out = H v - b,        partials <g,g>, AR/ridge sums, w.(Gw) - 2 b.w   (gradient at w)
//   HV_CG_FIRST  CG iteration 0: f, |g|, cgtol (rf_tron.h:154-169, 424-439); s = 0, r = d = -g; out = H d
//   HV_CG_STEP   CG iteration it >= 1, the WHOLE iteration in one launch (rf_tron.h:456-502):
//                   alpha = rho/<d,Hd>;  s += alpha d;  r -= alpha Hd;            (closes iteration it-1)
//                   rho' = rho - 2 alpha <r,Hd> + alpha^2 <Hd,Hd>                 (= r'^T r', no extra pass)
//                   stop if sqrt(rho') <= cgtol, else beta = rho'/rho, d' = r' + beta d, out = H d'
//                The three dot products come from the previous launch's per-tile partials (fixed order,
//                double), so every workgroup -- and every rank -- derives the same scalars; r' on the halo
//                rows is recomputed from r and Hd, nothing is exchanged.  `last` closes the final iteration.
// KQ = rank rounded up to 8 (Gram rows requested per thread).
enum HvMode { HV_PLAIN = 0, HV_GRAD = 1, HV_CG_FIRST = 2, HV_CG_STEP = 3 };

struct HvVecs {
    const real *v;        // PLAIN/GRAD: operand;  CG_FIRST: gradient g;  CG_STEP: previous direction d
    const real *r_in;     // CG_STEP: residual of the previous iteration
    const real *hd_in;    // CG_STEP: H d of the previous iteration
    real *s;              // CG_*: the step (own rows, in place)
    real *d_out;          // CG_*: new direction
    real *r_out;          // CG_*: new residual
    real *out;            // H v / gradient / H d
    const real *Bv;       // GRAD: right-hand sides
};

// ---- per-tile partial records and the time-sharded CG ------------------------------------------
// Every launch of the fused path leaves ONE 64-byte record per tile (tile-major, fixed order => every workgroup and
// every rank derives bit-identical scalars from them):
//     gradient launch      [0] AR residual^2   [1] <w,w>      [2] <g,g>      [3] w.(Gw) - 2 b.w
//     CG launch `it`       [0] <d,Hd>          [1] <r,Hd>     [2] <Hd,Hd>
//     cg_close_kernel      [4] <g,s>           [5] <s,r>      [6] <s,s>      (closes the last iteration: s, w_new = w + s)
//     plain launch (H s)   [0] AR residual^2   [1] <s,s>      [2] <s,Hs>     (same message as cg_close_kernel's)
// The records of a launch live in a MESSAGE buffer of `world` equal slots; slot r holds the records of rank r's tiles
// followed by the rank's EDGE rows -- the first and the last midx rows of its timestamp block of up to three vectors
// (d, r, H d of a CG launch; g of the gradient launch; s of cg_close_kernel).  With one rank there is one slot and
// no edges.  With several ranks every rank runs the tiles of its own contiguous block of timestamps (the Hessian is
// block-diagonal per timestamp, trmf.cpp:269-288; the AR stencil reaches midx rows, trmf.cpp:125-149; the CG needs
// three scalars per step, rf_tron.h:460-501): after each launch the slots are exchanged (one in-place all-gather of
// the message, Comm::allgather_slots) and halo_unpack_kernel copies the neighbours' edge rows to their natural rows
// of the local vectors, where the next launch stages them exactly as on one GPU.
constexpr int kRecDoubles = 8;
constexpr int kEdgeVecs = 3;
struct TileShard {
    int rank, world;
    int tile0, ntiles;        // this rank's tiles [tile0, tile0 + ntiles)   (grid of every launch)
    int nbt, tpr;             // tiles of the whole problem, tiles per slot (= nbt with one rank)
    int row_b, row_e;         // this rank's timestamps [row_b, row_e)
    unsigned slot_dbl;        // doubles per slot (records + edges, 16-byte multiple)
    unsigned edge_off_dbl;    // offset of the edge rows within a slot, in doubles
};
template <bool SHARD>
__device__ __forceinline__ size_t rec_index(const TileShard &sh, int t) {
    if constexpr (!SHARD) return (size_t)t * kRecDoubles;
    else {
        const int r = t / sh.tpr;
        return (size_t)r * sh.slot_dbl + (size_t)(t - r * sh.tpr) * kRecDoubles;
    }
}
// edge rows of this rank's slot: [side: first rows / last rows][vector][midx * KP]
__device__ __forceinline__ real *edge_base(double *msg, const TileShard &sh, int rank) {
    return reinterpret_cast<real *>(msg + (size_t)rank * sh.slot_dbl + sh.edge_off_dbl);
}
// The neighbours' edge rows -> their natural rows of the local vectors (rows [row_b - midx, row_b) from the LAST rows of
// rank - 1, rows [row_e, row_e + midx) from the FIRST rows of rank + 1).
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
__global__ __launch_bounds__(256) void halo_unpack_kernel(const double *__restrict__ msg, TileShard sh,
                                                          int edgeN /* midx * KP */, int KP, int nvec,
                                                          real *__restrict__ v0, real *__restrict__ v1, real *__restrict__ v2) {
    real *dst[kEdgeVecs] = {v0, v1, v2};
    for (int side = 0; side < 2; side++) {
        const int nb = side == 0 ? sh.rank - 1 : sh.rank + 1;          // neighbour
        if (nb < 0 || nb >= sh.world) continue;
        // left neighbour: its LAST rows (edge side 1) land below row_b; right neighbour: its FIRST rows (side 0) at row_e
        const real *src = reinterpret_cast<const real *>(msg + (size_t)nb * sh.slot_dbl + sh.edge_off_dbl) +
                          (size_t)(side == 0 ? 1 : 0) * kEdgeVecs * edgeN;
        const size_t row0 = side == 0 ? (size_t)sh.row_b * KP - edgeN : (size_t)sh.row_e * KP;
        for (int v = 0; v < nvec; v++)
            for (int e = blockIdx.x * 256 + threadIdx.x; e < edgeN; e += gridDim.x * 256) dst[v][row0 + e] = src[(size_t)v * edgeN + e];
    }
}
#endif

// Workgroup b -> tile, such that the workgroups an XCD receives (b % 8 == x on gfx950's round-robin dispatch) own one
// contiguous range of tiles.  A bijection on [0, n) for every n; changes only which workgroup does which tile.
__device__ __forceinline__ int xcd_contiguous_tile(int b, int n) {
    constexpr int kXcds = 8;
    const int x = b % kXcds, i = b / kXcds;
    const int base = n / kXcds, extra = n % kXcds;           // XCDs 0..extra-1 get base+1 workgroups
    return x * base + min(x, extra) + i;
}

// This rank's first / last midx rows of up to three vectors -> its slot of an edge message (the time-sharded UNFUSED CG: its
// kernels do not export edges themselves; hv_tile_kernel and cg_close_kernel do).
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
__global__ __launch_bounds__(256) void edge_pack_kernel(double *__restrict__ msg, TileShard sh, int edgeN, int KP, int nvec,
                                                        const real *__restrict__ v0, const real *__restrict__ v1,
                                                        const real *__restrict__ v2) {
    const real *src[kEdgeVecs] = {v0, v1, v2};
    real *edges = edge_base(msg, sh, sh.rank);
    for (int side = 0; side < 2; side++) {
        const size_t row0 = side == 0 ? (size_t)sh.row_b * KP : (size_t)sh.row_e * KP - edgeN;
        for (int v = 0; v < nvec; v++)
            for (int e = blockIdx.x * 256 + threadIdx.x; e < edgeN; e += gridDim.x * 256)
                edges[((size_t)side * kEdgeVecs + v) * edgeN + e] = src[v][row0 + e];
    }
}
#endif

// ---- peer-to-peer form of the exchange (TRMF_CG=p2p) ---------------------------------------------------------------------
// The all-gather of a message costs a collective launch (tens of microseconds) per CG step -- more than the step itself
// at config 4.  On a node whose GPUs map each other's memory the ranks write their part of a message straight into the
// peers' copies instead: every rank keeps its message buffers and a flag word per (message, source rank) in ONE arena
// (uncached device memory, exported with hipIpcGetMemHandle and opened by the peers); a launch stores its tile records
// into the copy of EVERY rank and its first / last midx rows into the copy of the neighbour that stages them; the small
// kernel that follows on the stream (all stores of the launch are complete and released by then) raises this rank's flag
// in every peer's arena to the message's epoch, waits until every peer's flag in the OWN arena has reached it, and
// unpacks the halo rows.  Epochs only grow; a message buffer is reused two launches later, which a peer cannot reach
// before it has consumed the previous content (it needs this rank's NEXT message to get there).  Every wait is bounded:
// a timeout sets XState::p2p_error (the solve is then reported as failed) instead of hanging the GPU.
constexpr int kMaxPeers = 8;
struct PeerTable {                              // lives in device memory (indexed by rank at run time)
    double *msg[3][kMaxPeers];                  // message m in the arena of rank r (own rank: the local copy)
    unsigned long long *flags[3][kMaxPeers];    // message m's flag words (one 64-byte line per source rank) in the arena of rank r
};
constexpr int kFlagStride = 8;                  // 64 bytes between the flags of different source ranks
constexpr long long kP2pTimeoutTicks = 1000000000;  // 10 s of the 100 MHz wall clock
constexpr long long kP2pTrialTicks = 20000000;      // 200 ms: the trial exchange of the set-up (a failure there only means "use the communicator")
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
__global__ __launch_bounds__(256) void xchg_sync_kernel(const PeerTable *__restrict__ pt, int mi, unsigned long long epoch, XState *__restrict__ st, int it,
                                                        TileShard sh, int edgeN, int KP, int nvec,
                                                        real *__restrict__ v0, real *__restrict__ v1, real *__restrict__ v2,
                                                        long long timeout_ticks) {
    if (it >= 0 && st->stop_it <= it) return;   // launch `it` left no message (the CG stopped at or before it): on every rank alike
    __shared__ int failed;
    const int tid = threadIdx.x;
    if (tid == 0) failed = st->p2p_error;
    __syncthreads();
    if (tid < sh.world && tid != sh.rank && !failed) {
        __hip_atomic_store(pt->flags[mi][tid] + (size_t)sh.rank * kFlagStride, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long *mine = pt->flags[mi][sh.rank] + (size_t)tid * kFlagStride;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            if (wall_clock64() - t0 > timeout_ticks) {
                if (!atomicExch(&failed, 1)) {
                    st->p2p_diag[0] = mi * 1000000ll + (long long)(it + 1) * 1000 + tid; st->p2p_diag[1] = (long long)epoch;
                    st->p2p_diag[2] = (long long)__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (failed) { if (tid == 0) st->p2p_error = 1; return; }
    const double *msg = pt->msg[mi][sh.rank];
    real *dst[kEdgeVecs] = {v0, v1, v2};
    for (int side = 0; side < 2; side++) {
        const int nb = side == 0 ? sh.rank - 1 : sh.rank + 1;
        if (nb < 0 || nb >= sh.world || edgeN == 0) continue;
        const real *src = reinterpret_cast<const real *>(msg + (size_t)nb * sh.slot_dbl + sh.edge_off_dbl) +
                          (size_t)(side == 0 ? 1 : 0) * kEdgeVecs * edgeN;
        const size_t row0 = side == 0 ? (size_t)sh.row_b * KP - edgeN : (size_t)sh.row_e * KP;
        for (int v = 0; v < nvec; v++)
            for (int e = tid; e < edgeN; e += 256) dst[v][row0 + e] = __builtin_nontemporal_load(src + (size_t)v * edgeN + e);
    }
}
#endif

// Peer-to-peer exchange of the time-sharded UNFUSED CG: its kernels keep their partial sums in arrays and do not export edges,
// so one small kernel pushes what the peers need after each operator application -- this rank's entries of up to four
// partial-sum arrays into every peer's copy of the arrays (message 1 of the arena), its first / last midx rows of up to three
// vectors into the edge message of the neighbour that stages them (messages 0 and 2 alternately: a peer may still be unpacking
// the previous exchange's edges when this rank, one operator application further, pushes the next) -- and xchg_sync_kernel
// follows (flags, halo rows).
struct PushList { int n; int slot[4], begin[4], count[4]; };
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
__global__ __launch_bounds__(256) void uts_push_kernel(const PeerTable *__restrict__ pt, int mi, const XState *__restrict__ st, int it,
                                                       TileShard sh, int pstride, PushList pl, int edgeN, int KP, int nvec,
                                                       const real *__restrict__ v0, const real *__restrict__ v1, const real *__restrict__ v2) {
    if (it >= 0 && st->stop_it <= it) return;           // the application left nothing (the CG stopped at or before it): on every rank alike
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsize = gridDim.x * 256;
    const double *mine = pt->msg[1][sh.rank];
    for (int a = 0; a < pl.n; a++) {
        const size_t base = (size_t)pl.slot[a] * pstride + pl.begin[a];
        for (int i = gtid; i < pl.count[a]; i += gsize) {
            const double v = mine[base + i];
            for (int r = 0; r < sh.world; r++)
                if (r != sh.rank) pt->msg[1][r][base + i] = v;
        }
    }
    const real *src[kEdgeVecs] = {v0, v1, v2};
    for (int side = 0; side < 2 && edgeN > 0; side++) {
        const int nb = side == 0 ? sh.rank - 1 : sh.rank + 1;          // the first rows are the lower neighbour's upper halo, the last rows the upper neighbour's
        if (nb < 0 || nb >= sh.world) continue;
        real *edges = edge_base(pt->msg[mi][nb], sh, sh.rank);
        const size_t row0 = side == 0 ? (size_t)sh.row_b * KP : (size_t)sh.row_e * KP - edgeN;
        for (int v = 0; v < nvec; v++)
            for (int e = gtid; e < edgeN; e += gsize) edges[((size_t)side * kEdgeVecs + v) * edgeN + e] = src[v][row0 + e];
    }
    __threadfence_system();
}
#endif

#if !defined(TRMF_UNIT_BODIES)     // the main translation unit sees the declaration only (kernel_units.hpp)
template <int MODE, int KQ, bool SHARD, int NTH = 256>
__global__ void hv_tile_kernel(XParams p, XState *__restrict__ st, HvVecs a, TileShard sh,
                                                         int it, int last,
                                                         const uint32_t *__restrict__ lag_set,
                                                         const real *__restrict__ theta,
                                                         const real *__restrict__ G,
                                                         const double *__restrict__ rec_in, double *__restrict__ rec_out,
                                                         const PeerTable *__restrict__ pt, int mi, int TI);
#else
template <int MODE, int KQ, bool SHARD, int NTH = 256>
__global__ __launch_bounds__(NTH, NTH == 256 ? 2 : 1) void hv_tile_kernel(XParams p, XState *__restrict__ st, HvVecs a, TileShard sh,
                                                         int it, int last,
                                                         const uint32_t *__restrict__ lag_set,
                                                         const real *__restrict__ theta,
                                                         const real *__restrict__ G,
                                                         const double *__restrict__ rec_in, double *__restrict__ rec_out,
                                                         const PeerTable *__restrict__ pt, int mi, int TI) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hv_smem[];
    __shared__ double smem[256];
    constexpr bool GRAD = MODE == HV_GRAD;
    constexpr bool CG = MODE == HV_CG_FIRST || MODE == HV_CG_STEP;
    constexpr int VEC = hv_vec(KQ);
    constexpr int NT_T = (KQ + kTile - 1) / kTile;
    constexpr int KP = kTile * NT_T;
    const int tid = threadIdx.x;
    const int k = p.k, T = p.T, Hh = p.midx, nlag = p.nlag;
    const int rowsV = TI + 2 * Hh, rowsR = TI + Hh, nV = rowsV * KP, nTh = nlag * k;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch; speed only, never correctness), so
    // workgroups of one XCD take CONSECUTIVE tiles -- the halo rows a tile reads (its neighbours' r, d, H d of the
    // previous launch) were then written through the same XCD's L2
    const int tile = (SHARD ? sh.tile0 : 0) + xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x);
    const int i0 = tile * TI, i1 = min(i0 + TI, T);          // one tile per workgroup
    const int np_in = sh.nbt;                                // records of the previous launch: every tile of the problem
    real *vs = reinterpret_cast<real *>(hv_smem);
    double *rs = reinterpret_cast<double *>(hv_smem + (((size_t)rowsV * KP * sizeof(real) + 15) / 16 * 16));
    constexpr int RPITCH = KQ;                              // = hv_res_pitch(k)
    real *rn = reinterpret_cast<real *>(reinterpret_cast<unsigned char *>(rs) + (((size_t)rowsR * RPITCH * sizeof(double) + 15) / 16 * 16));
    double *thd = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(rn) + (((size_t)TI * KP * sizeof(real) + 15) / 16 * 16));
    real *thp = reinterpret_cast<real *>(thd + (size_t)nlag * KP);
    int *lags = reinterpret_cast<int *>(thp + (size_t)nlag * KP);
    // rn[own row][pos]  = new residual of the tile's own rows (fused CG: <r,Hd>)
    // thd[l*KP + t]     = lambdaAR * Theta(l,t) in double, LOGICAL column order  (AR adjoint, phase 3)
    // thp[l*KP + pos]   = Theta(l, collog(pos)), POSITION order like the staged operand  (AR residual, phase 2)
    // both with the row stride KP so that a thread's 4 neighbouring columns are one aligned 16/32-byte read

    // sticky stop: an EARLIER launch ended the CG.  Launch `it` itself may store stop_it = it while some of its
    // workgroups have not started yet; they read either kCgRunning or `it`, never a value below `it`.
    if (MODE == HV_CG_STEP && st->stop_it < it) return;

    // ---- scalar prologue; Theta / lag set to LDS (their loads fly with the partials) ----
    real thr[kHvThetaRegs];
    int lagr = 0;
    if (nlag > 0) {
#pragma unroll
        for (int m = 0; m < kHvThetaRegs; m++) thr[m] = theta[min(tid + NTH * m, nTh - 1)];
        lagr = (int)lag_set[min(tid, nlag - 1)];
    }
    real tmp = 0, alpha = 0, nalpha = 0;
    bool stopped = false;
    // HV_CG_STEP: the three dot-product partial arrays are requested FIRST (vmcnt retires in order), then the
    // operand rows and the Gram slice, and only then reduced -- the whole request stream of the launch is
    // in flight before the first wait.  (More than 512 tiles: read in a loop after the requests instead.)
    constexpr int kEarlyPartials = 2;                       // x 256 threads
    const bool early = MODE == HV_CG_STEP && np_in <= NTH * kEarlyPartials;
    double pq[3][kEarlyPartials];
    if (MODE == HV_CG_STEP && early) {
#pragma unroll
        for (int m = 0; m < kEarlyPartials; m++) {
            const double *rec = rec_in + rec_index<SHARD>(sh, min(tid + NTH * m, np_in - 1));
#pragma unroll
            for (int a3 = 0; a3 < 3; a3++) pq[a3][m] = rec[a3];
        }
    }
    if (MODE == HV_CG_FIRST) {
        // f, |g|, tolerances from the gradient launch's partials (rf_tron.h:154-169, 424-439)
        double ar2 = 0, vv = 0, gg = 0, lq = 0;
        for (int i = tid; i < np_in; i += NTH) {
            const double *rec = rec_in + rec_index<SHARD>(sh, i);
            ar2 += rec[0]; vv += rec[1]; gg += rec[2]; lq += rec[3];
        }
        block_allsum3<NTH / 64>(ar2, vv, gg, smem);
        lq = block_allsum<NTH / 64>(lq, smem);
        const real ggr = (real)gg;                                           // BLAS dot in val_type
        const double gnorm = sqrt((double)ggr);
        const real cgtol = (real)(p.eps_cg * gnorm);                         // rf_tron.h:434
        stopped = cg_stopped(ggr, cgtol);                                    // rho[0] = r^T r = g^T g (rf_tron.h:439)
        if (blockIdx.x == 0 && tid == 0) {
            // loss = sum y^2 + sum_i (w_i^T G_i w_i - 2 b_i.w_i): the reference's own formula on the full path
            // (trmf.cpp:189-197); on the observed-entries path the same identity over the cached Grams
            double f = 0.5 * (p.trYTY + lq);
            if (p.lambdaI > 0) f += 0.5 * p.lambdaI * (double)(real)vv;      // trmf.cpp:73-75
            if (p.nlag > 0 && p.lambdaAR > 0) f += 0.5 * p.lambdaAR * ar2;   // trmf.cpp:94
            st->f = f; st->fnew = f; st->gnorm = gnorm; st->cgtol = cgtol; st->cg_rnorm = gnorm;
            st->cg_iter = stopped ? 0 : 1;          // iterations the CG is committed to so far (launch `it` raises it to it + 1)
            st->accepted = 0; st->rho_hist[0] = (double)ggr;
            st->stop_it = stopped ? 0 : kCgRunning; st->r_parity = 0;
        }
    }
    // ---- requests, in consumption order (vmcnt retires in order) ----
    // (a) operand rows: the staged rows [i0-midx, i0+TI+midx) are one contiguous range of the vector,
    //     stage element e = vector element (i0-midx)*KP + e.  Buffer descriptors do the clipping: an
    //     offset outside the vector reads 0 (its negative wraps to a huge unsigned), and the direction
    //     store goes through a descriptor that spans only the tile's own rows.
    const int sz = (int)sizeof(real);
    const size_t vec_bytes = (size_t)T * KP * sizeof(real);
    const __amdgpu_buffer_rsrc_t v_rsrc = buffer_rsrc(a.v, vec_bytes);
    const __amdgpu_buffer_rsrc_t r_rsrc = buffer_rsrc(MODE == HV_CG_STEP ? a.r_in : a.v, vec_bytes);
    const __amdgpu_buffer_rsrc_t h_rsrc = buffer_rsrc(MODE == HV_CG_STEP ? a.hd_in : a.v, vec_bytes);
    // own-row descriptors: element eo of the tile's own rows; anything outside reads 0 / is dropped
    const size_t own_bytes = (size_t)(i1 - i0) * KP * sizeof(real);
    const __amdgpu_buffer_rsrc_t s_rsrc = buffer_rsrc(CG ? a.s + (size_t)i0 * KP : nullptr, CG ? own_bytes : 0);
    const __amdgpu_buffer_rsrc_t do_rsrc = buffer_rsrc(CG ? a.d_out + (size_t)i0 * KP : nullptr, CG ? own_bytes : 0);
    const __amdgpu_buffer_rsrc_t ro_rsrc = buffer_rsrc(CG ? a.r_out + (size_t)i0 * KP : nullptr, CG ? own_bytes : 0);
    const int vbyte0 = ((i0 - Hh) * KP + tid) * sz;          // wraps below zero for the first tiles: reads 0
    const int obyte0 = (tid - Hh * KP) * sz;                 // the same element in own-row coordinates
    real vr[kHvOperandRegs], rv[kHvOperandRegs], hr[kHvOperandRegs], sr[kHvOperandRegs];
#pragma unroll
    for (int m = 0; m < kHvOperandRegs; m++) {
        vr[m] = buffer_load_real(v_rsrc, vbyte0 + NTH * m * sz);
        if (MODE == HV_CG_STEP) {
            rv[m] = buffer_load_real(r_rsrc, vbyte0 + NTH * m * sz);
            hr[m] = buffer_load_real(h_rsrc, vbyte0 + NTH * m * sz);
            sr[m] = buffer_load_real(s_rsrc, obyte0 + NTH * m * sz);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // (b) the thread's slice of the cached Gram: columns [t0, t0+VEC) of timestamp row i0+lr, all KQ rows
    const int tpr = (k + VEC - 1) / VEC;
    const int lr = tid / tpr, t0 = (tid - lr * tpr) * VEC;
    const bool lane_on = lr < TI;
    const int lrc = lane_on ? lr : TI - 1;                  // idle lanes shadow a live one (no branches)
    // The first KA Gram rows (kHvGramUpfront = 96 registers) are requested here, the rest after phase 2: the AR phases
    // need registers for their independent LDS reads (measured: 96 up front beats 128, 144 and 160; 32..96 are equal --
    // the stream is bandwidth-bound once it has started, it only must not start late).
    constexpr int KA = (KQ * VEC * (int)sizeof(real) / 4 <= kHvGramUpfront) ? KQ : kHvGramUpfront / (VEC * (int)sizeof(real) / 4);
    GramVec<VEC> gq[KQ];
    const __amdgpu_buffer_rsrc_t g_rsrc = buffer_rsrc(G + (size_t)i0 * p.gstride, 0x7fffffff);
    // (Measured and rejected in round 3: reading only the UPPER triangle -- blocks below the thread's own loaded as their
    // mirror G[t0 + u][VEC jq ..] and transposed in registers -- halves the HBM stream of a launch (64 -> 35 MB at config 3) and is
    // bit-identical, but a mirror load touches VEC different Gram rows per thread instead of one: the wavefront's requests fall
    // into 4x as many cache lines and the launch went from 14.3 to 20.2 us.)
    const int g_voff = (int)(((uint32_t)(min(i0 + lrc, T - 1) - i0) * (uint32_t)p.gstride + (uint32_t)t0) * sizeof(real));
    const int rowbytes = k * (int)sizeof(real);
    int g_soff = 0;
#pragma unroll
    for (int j = 0; j < KA; j++) {
        gq[j] = gram_load<VEC>(g_rsrc, g_voff, g_soff);
        g_soff += (j + 1 < k) ? rowbytes : 0;               // j >= k: a finite duplicate, multiplied by a zero pad
    }
    __builtin_amdgcn_sched_barrier(0);                      // keep the requests above everything that follows
    if (MODE == HV_CG_STEP && early) {
        // the partials become visible to the compiler only here: it otherwise starts their sum right behind their loads
        // (`v_add_f64 x, 0` in the loading block) and the wait for them lands in front of every request above
#pragma unroll
        for (int m = 0; m < kEarlyPartials; m++)
#pragma unroll
            for (int a3 = 0; a3 < 3; a3++) asm volatile("" : "+v"(pq[a3][m]));
    }
    // Theta / lag set -> LDS only now: the staging has loops, and with it above the requests the compiler drained every
    // outstanding load (s_waitcnt vmcnt(0)) before issuing the first operand load -- one memory round trip per launch
    if (nlag > 0) {
        auto put = [&](int e, real th) {
            const int tt = e / nlag, l = e - tt * nlag;
            thp[l * KP + colpos(tt, NT_T)] = th;
            thd[l * KP + tt] = p.lambdaAR * (double)th;
        };
#pragma unroll
        for (int m = 0; m < kHvThetaRegs; m++)
            if (tid + NTH * m < nTh) put(tid + NTH * m, thr[m]);
#pragma nounroll
        for (int e = tid + NTH * kHvThetaRegs; e < nTh; e += NTH) put(e, theta[e]);
#pragma nounroll
        for (int e = tid; e < nlag * (KP - k); e += NTH) {                  // pad columns: exact zeros
            const int l = e / (KP - k), tt = k + (e - l * (KP - k));
            thp[l * KP + colpos(tt, NT_T)] = 0;
            thd[l * KP + tt] = 0;
        }
        if (tid < nlag) lags[tid] = lagr;
#pragma nounroll
        for (int e = tid + NTH; e < nlag; e += NTH) lags[e] = (int)lag_set[e];
    }

    if (MODE == HV_CG_STEP) {
        double dHd = 0, rHd = 0, HH = 0;
        if (early) {
#pragma unroll
            for (int m = 0; m < kEarlyPartials; m++)
                if (tid + NTH * m < np_in) { dHd += pq[0][m]; rHd += pq[1][m]; HH += pq[2][m]; }
        } else {
            for (int i = tid; i < np_in; i += NTH) {
                const double *rec = rec_in + rec_index<SHARD>(sh, i);
                dHd += rec[0]; rHd += rec[1]; HH += rec[2];
            }
        }
        block_allsum3<NTH / 64>(dHd, rHd, HH, smem);
        const double rho_prev_d = st->rho_hist[it - 1];
        const real rho_prev = (real)rho_prev_d;
        alpha = rho_prev / (real)dHd;                                        // rf_tron.h:460
        nalpha = -alpha;
        const double ad = (double)alpha;
        const double rho_d = fmax(rho_prev_d - 2.0 * ad * rHd + ad * ad * HH, 0.0);   // |r - alpha Hd|^2
        const real rho = (real)rho_d;
        stopped = last || cg_stopped(rho, st->cgtol);                        // top of iteration `it`, rf_tron.h:444-446
        const real beta = rho / rho_prev;                                    // rf_tron.h:495
        tmp = beta - (real)1.0;                                              // rf_tron.h:497
        if (blockIdx.x == 0 && tid == 0) {
            st->rho_hist[it] = rho_d;
            if (stopped) { st->stop_it = it; st->r_parity = it & 1; }
            else st->cg_iter = it + 1;                                       // nobody reads cg_iter during the solve
        }
    }

    const uint32_t own_n = (uint32_t)((i1 - i0) * KP);
    // time-sharded CG: does this tile hold any of the rank's first / last midx rows (which the neighbours stage as halo)?
    const int edgeN = Hh * KP;
    const bool edge_tile = SHARD && edgeN > 0 && (i0 < sh.row_b + Hh || i1 > sh.row_e - Hh);
    // the first rows go to the copy of the rank that stages them as its upper halo (rank - 1), the last rows to rank + 1's;
    // without peer-to-peer access both land in the local message, which the all-gather then distributes
    const bool p2p = SHARD && pt != nullptr;
    real *edges_lo = SHARD ? edge_base(p2p && sh.rank > 0 ? pt->msg[mi][sh.rank - 1] : rec_out, sh, sh.rank) : nullptr;
    real *edges_hi = SHARD ? edge_base(p2p && sh.rank + 1 < sh.world ? pt->msg[mi][sh.rank + 1] : rec_out, sh, sh.rank) : nullptr;
    auto edge_put = [&](int vec, int ge /* element of the T x KP vector, an own row of this tile */, real x) {
        const uint32_t lo = (uint32_t)(ge - sh.row_b * KP), hi = (uint32_t)(ge - (sh.row_e - Hh) * KP);
        if (lo < (uint32_t)edgeN) edges_lo[(size_t)vec * edgeN + lo] = x;
        if (hi < (uint32_t)edgeN) edges_hi[(size_t)(kEdgeVecs + vec) * edgeN + hi] = x;
    };
    // a tile's record: into the local message, and with peer-to-peer access into the copy of every other rank
    auto put_record = [&](int first, double a0, double a1, double a2, double a3, bool four) {
        const size_t ri = rec_index<SHARD>(sh, tile) + first;
        for (int r = 0; r < (p2p ? sh.world : 1); r++) {
            double *dst = (p2p ? pt->msg[mi][r] : rec_out) + ri;
            dst[0] = a0; dst[1] = a1; dst[2] = a2;
            if (four) dst[3] = a3;
        }
    };
    // The launch that detects the stop (at the top of iteration `it`, or `it` is the iteration cap) does nothing more: the
    // last completed iteration is closed -- s += alpha d, w_new = w + s, the sums of the acceptance test -- by
    // cg_close_kernel, which the host enqueues after the CG launches.  (Closing it here, as round 2 did for s and r and an
    // earlier round-3 version did for everything, keeps six more vectors' worth of state alive next to the Gram slice: the
    // kernel spilled and every CG launch became 3 us slower.)
    if (MODE == HV_CG_STEP && stopped) return;

    // (1) operand rows -> LDS (zeros outside [0,T) stay zeros through every update below)
    double ar2 = 0, vv = 0, dot = 0, lq = 0, rhd = 0, hh = 0;
    auto operand = [&](int e, real x, real rx, real hx, real sx) {
        const int eo = e - Hh * KP;                                        // index within the tile's own rows
        if (CG) {
            real rnew, snew;
            if (MODE == HV_CG_FIRST) { x = -x; rnew = x; snew = 0; }       // s = 0, r = d = -g
            else {
                snew = fma(alpha, x, sx);                                  // s += alpha d      (rf_tron.h:461)
                rnew = fma(nalpha, hx, rx);                                // r -= alpha Hd     (rf_tron.h:489-490)
                x = fma(tmp, x, x); x = x + rnew;                          // d = beta d + r    (rf_tron.h:497-499)
            }
            buffer_store_real(s_rsrc, eo * sz, snew);                      // all three dropped outside the own rows
            buffer_store_real(ro_rsrc, eo * sz, rnew);
            buffer_store_real(do_rsrc, eo * sz, x);
            if ((uint32_t)eo < own_n) {
                rn[eo] = rnew;
                if (edge_tile) { edge_put(0, i0 * KP + eo, x); edge_put(1, i0 * KP + eo, rnew); }
            }
        } else {
            if ((uint32_t)eo < own_n) vv += (double)x * (double)x;
        }
        if (e < nV) vs[e] = x;
    };
#pragma unroll
    for (int m = 0; m < kHvOperandRegs; m++)
        operand(tid + NTH * m, vr[m], MODE == HV_CG_STEP ? rv[m] : real(0), MODE == HV_CG_STEP ? hr[m] : real(0),
                MODE == HV_CG_STEP ? sr[m] : real(0));
#pragma nounroll
    for (int e = tid + NTH * kHvOperandRegs; e < nV; e += NTH) {           // very long halos only
        const int vb = vbyte0 + (e - tid) * sz, ob = obyte0 + (e - tid) * sz;
        operand(e, buffer_load_real(v_rsrc, vb), MODE == HV_CG_STEP ? buffer_load_real(r_rsrc, vb) : real(0),
                MODE == HV_CG_STEP ? buffer_load_real(h_rsrc, vb) : real(0),
                MODE == HV_CG_STEP ? buffer_load_real(s_rsrc, ob) : real(0));
    }
    if (MODE == HV_CG_FIRST && stopped) return;             // the gradient already meets the tolerance: s = 0 is written, no product
#if defined(TRMF_HV_ABL) && (TRMF_HV_ABL & 2)
    const bool ar_on = false;
#else
    const bool ar_on = nlag > 0 && p.lambdaAR > 0;
#endif
    __syncthreads();
    // (2) AR residuals of rows [i0, i0+TI+midx)  (trmf.cpp:110-113 / 136-139), stored to rs in LOGICAL
    //     column order.  A work item is (row, 4 neighbouring positions): per lag it costs two 16-byte LDS
    //     reads (operand, Theta in position order) for four residuals; kHvResU items advance together.
    if (ar_on) {
        constexpr int NG = KP / 4;
        const int items = rowsR * NG;
#pragma nounroll
        for (int it0 = 0; it0 < items; it0 += NTH * kHvResU) {
            int vb[kHvResU], pg[kHvResU];
            bool on[kHvResU];
            double res[kHvResU][4];
#pragma unroll
            for (int u = 0; u < kHvResU; u++) {
                const int it = it0 + tid + NTH * u;
                const int rr = it / NG, g = it - rr * NG, i = i0 + rr;
                on[u] = it < items && i >= Hh && i < T;
                vb[u] = on[u] ? (rr + Hh) * KP + 4 * g : Hh * KP;
                pg[u] = on[u] ? 4 * g : 0;
                const Quad<real> x4 = *reinterpret_cast<const Quad<real> *>(vs + vb[u]);
#pragma unroll
                for (int c = 0; c < 4; c++) res[u][c] = (double)x4.v[c];
            }
#pragma unroll 2
            for (int l = 0; l < nlag; l++) {
                const int back = lags[l] * KP;
                const real *thl = thp + l * KP;
#pragma unroll
                for (int u = 0; u < kHvResU; u++) {
                    const Quad<real> th4 = *reinterpret_cast<const Quad<real> *>(thl + pg[u]);
                    const Quad<real> x4 = *reinterpret_cast<const Quad<real> *>(vs + vb[u] - back);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const real prod = th4.v[c] * x4.v[c];
                        res[u][c] -= (double)prod;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kHvResU; u++) {
                const int it = it0 + tid + NTH * u;
                const int rr = it / NG, g = it - rr * NG;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int tl = collog(4 * g + c, NT_T);
                    const double rv2 = on[u] ? res[u][c] : 0.0;           // rows outside [midx,T): exact zeros
                    if (it < items && tl < k) {
                        if (rr < TI) ar2 += rv2 * rv2;
                        rs[rr * RPITCH + tl] = rv2;
                    }
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = KA; j < KQ; j++) {                         // rest of the Gram slice: requested only now -- phase 2 needs the registers for its
                                                            // independent LDS reads -- and lands under the AR adjoint
        gq[j] = gram_load<VEC>(g_rsrc, g_voff, g_soff);
        g_soff += (j + 1 < k) ? rowbytes : 0;
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // (3) out = lambdaI*v + lambdaAR*AR'(v) + G.v (- b) for the thread's row and VEC columns
    {
        const int rr = lrc, i = i0 + rr;
        const bool live = lane_on && i < T;
        const real *vi = vs + (rr + Hh) * KP;
        int tcol[VEC], tpos[VEC];
        real x[VEC];
        double od[VEC];                                     // lambdaI*v + lambdaAR*AR'(v), carried in double
#pragma unroll
        for (int c = 0; c < VEC; c++) {
            tcol[c] = min(t0 + c, k - 1);
            tpos[c] = colpos(tcol[c], NT_T);
            x[c] = vi[tpos[c]];
            real o;
            if (p.lambdaI == 0) o = 0;
            else if (p.lambdaI == 1) o = x[c];
            else o = (real)(p.lambdaI * (double)x[c]);
            od[c] = (double)o;
        }
        if (ar_on) {
            // residual rows outside [midx, T) were stored as exact zeros by phase 2, so neither the own-row
            // term nor the lagged terms need a range test; lambdaAR*Theta comes pre-multiplied from LDS
            {
                const VecOf<double, VEC> r0 = *reinterpret_cast<const VecOf<double, VEC> *>(rs + rr * RPITCH + t0);
#pragma unroll
                for (int c = 0; c < VEC; c++) od[c] += p.lambdaAR * r0.v[c];
            }
#pragma unroll 4
            for (int l = 0; l < nlag; l++) {                // VEC neighbouring logical columns: aligned vector reads
                const VecOf<double, VEC> r4 = *reinterpret_cast<const VecOf<double, VEC> *>(rs + (rr + lags[l]) * RPITCH + t0);
                const VecOf<double, VEC> t4 = *reinterpret_cast<const VecOf<double, VEC> *>(thd + l * KP + t0);
#pragma unroll
                for (int c = 0; c < VEC; c++) od[c] -= r4.v[c] * t4.v[c];
            }
        }
        // cached Gram last: by now the slice has (mostly) arrived.  Four Gram rows per step; the operand
        // values of the next step are read from LDS while this step's FMAs run.  The empty asm statements
        // are ordering fences on VALUES (accumulators, the prefetched operands, the LDS pointer): without
        // them the scheduler hoists every LDS read and conversion of the 40-row loop to the top and the
        // register file -- most of which the slice already holds -- overflows into scratch.
        double acc[VEC];
#pragma unroll
        for (int c = 0; c < VEC; c++) acc[c] = 0;
        const real *vip = vi;
        real vcur[4], vnext[4];
#pragma unroll
        for (int u = 0; u < 4; u++) vcur[u] = vip[colpos(u, NT_T)];  // logical column j sits at position colpos(j)
#pragma unroll
        for (int j0 = 0; j0 < KQ; j0 += 4) {
            asm volatile("" : "+v"(vip));
            if (j0 + 4 < KQ) {
#pragma unroll
                for (int u = 0; u < 4; u++) vnext[u] = vip[colpos(j0 + 4 + u, NT_T)];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double vj = (double)vcur[u];                          // pad columns hold 0
#pragma unroll
                for (int c = 0; c < VEC; c++) acc[c] += (double)gq[j0 + u].c[c] * vj;
            }
#pragma unroll
            for (int c = 0; c < VEC; c++) asm volatile("" : "+v"(acc[c]));
            if (j0 + 4 < KQ) {
#pragma unroll
                for (int u = 0; u < 4; u++) { asm volatile("" : "+v"(vnext[u])); vcur[u] = vnext[u]; }
            }
        }
#pragma unroll
        for (int c = 0; c < VEC; c++) {
            if (live && t0 + c < k) {
                double ac = acc[c];
                if (GRAD) {
                    const double bb = (double)a.Bv[(size_t)i * KP + tcol[c]];
                    lq += (double)x[c] * (ac - 2.0 * bb);                    // w.(Gw) - 2 b.w
                    ac -= bb;
                }
                const real oc = (real)(od[c] + ac);
                a.out[(size_t)i * KP + tpos[c]] = oc;
                if (edge_tile && (CG || GRAD)) edge_put(CG ? 2 : 0, i * KP + tpos[c], oc);   // H d of a CG launch / the gradient
                dot += (double)(GRAD ? oc : x[c]) * (double)oc;      // <g,g> for the gradient, <v,Hv> otherwise
                if (CG) {
                    rhd += (double)rn[rr * KP + tpos[c]] * (double)oc;       // <r,Hd>
                    hh += (double)oc * (double)oc;                           // <Hd,Hd>
                }
            }
        }
    }
    if (CG) {
        block_allsum3<NTH / 64>(dot, rhd, hh, smem);
        if (threadIdx.x == 0) put_record(0, dot, rhd, hh, 0, false);
        if (p2p) __threadfence_system();
        return;
    }
    block_allsum3<NTH / 64>(ar2, vv, dot, smem);
    if (GRAD) lq = block_allsum<NTH / 64>(lq, smem);
    if (threadIdx.x == 0) put_record(0, ar2, vv, dot, lq, GRAD);     // [2]: <g,g> of the gradient launch, <v,Hv> of the plain one
    if (p2p) __threadfence_system();
}
#endif

// ---- CG initialisation: f, |g|, tolerances; s = 0, r = -g, d = r  (rf_tron.h:154-169, 424-439) ----
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
__global__ __launch_bounds__(256) void cg_init_kernel(XParams p, XState *__restrict__ st,
                                                      double *__restrict__ Pbase, int np_base,
                                                      int np_dot, const real *__restrict__ g,
                                                      real *__restrict__ s, real *__restrict__ r,
                                                      real *__restrict__ d, size_t e_begin, size_t e_end) {
    __shared__ double smem[256];
    const double ar2 = sum_partials(Pbase + P_AR * (size_t)p.pstride, np_base, smem);
    const double vv = sum_partials(Pbase + P_VV * (size_t)p.pstride, np_base, smem);
    const double gg = sum_partials(Pbase + P_DOT * (size_t)p.pstride, np_dot, smem);
    const double lq = sum_partials(Pbase + P_LQ * (size_t)p.pstride, np_dot, smem);
    const real ggr = (real)gg;                                               // BLAS dot in val_type
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // loss = sum y^2 + sum_i (w_i^T G_i w_i - 2 b_i.w_i): the reference's own formula on the full path
        // (trmf.cpp:189-197); on the observed-entries path the same identity over the cached Grams replaces its
        // pass over the residuals (trmf.cpp:231-245) -- f only feeds the TRON line, not the iterates
        double f = 0.5 * (p.trYTY + lq);
        if (p.lambdaI > 0) f += 0.5 * p.lambdaI * (double)(real)vv;          // trmf.cpp:73-75
        if (p.nlag > 0 && p.lambdaAR > 0) f += 0.5 * p.lambdaAR * ar2;       // trmf.cpp:94
        const double gnorm = sqrt((double)ggr);
        st->f = f; st->fnew = f; st->gnorm = gnorm;
        st->cgtol = (real)(p.eps_cg * gnorm);                                // rf_tron.h:434
        st->cg_rnorm = gnorm;
        const bool stopped = cg_stopped(ggr, (real)(p.eps_cg * gnorm));
        st->cg_iter = stopped ? 0 : 1;              // iterations the CG is committed to so far (launch `it` raises it to it + 1)
        st->accepted = 0;
        st->rho_hist[0] = (double)ggr;                                       // rho[0] = r^T r = g^T g (rf_tron.h:439)
        st->stop_it = stopped ? 0 : kCgRunning;
        st->r_parity = 0;
    }
    // elements [e_begin, e_end): everything, or a rank's timestamps plus the halo rows whose gradient it has been sent
    for (size_t e = e_begin + (size_t)blockIdx.x * 256 + threadIdx.x; e < e_end; e += (size_t)gridDim.x * 256) {
        const real gv = g[e];
        s[e] = 0; r[e] = -gv; d[e] = -gv;
    }
}
#endif

// ---- w_new = w + s ; partials <g,s>, <s,r>  (rf_tron.h:183-190) --------------------------------------
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
__global__ __launch_bounds__(256) void wnew_kernel(XParams p, const XState *__restrict__ st,
                                                   const real *__restrict__ w,
                                                   const real *__restrict__ s,
                                                   const real *__restrict__ g,
                                                   const real *__restrict__ r_even,
                                                   const real *__restrict__ r_odd,
                                                   real *__restrict__ w_new,
                                                   double *__restrict__ Pbase, size_t e_begin, size_t e_end, int slot0) {
    __shared__ double smem[256];
    const real *__restrict__ r = st->r_parity ? r_odd : r_even;   // the launch that stopped wrote it
    double gs = 0, sr = 0, ss = 0;
    for (size_t e = e_begin + (size_t)blockIdx.x * 256 + threadIdx.x; e < e_end; e += (size_t)gridDim.x * 256) {
        const real sv = s[e];
        w_new[e] = w[e] + sv;
        gs += (double)g[e] * (double)sv;
        sr += (double)sv * (double)r[e];
        ss += (double)sv * (double)sv;
    }
    block_allsum3(gs, sr, ss, smem);
    if (threadIdx.x == 0) {
        Pbase[P_GS * (size_t)p.pstride + slot0 + blockIdx.x] = gs;
        Pbase[P_SR * (size_t)p.pstride + slot0 + blockIdx.x] = sr;
        Pbase[P_SS * (size_t)p.pstride + slot0 + blockIdx.x] = ss;
    }
}
#endif

// ---- acceptance test and commit (rf_tron.h:191-229) -------------------------------------------------
// The X sub-problem is exactly quadratic, so f(w+s) - f(w) = g.s + 1/2 s.Hs.  The reference evaluates
// fun(w+s) with another full pass over the observations (rf_tron.h:191 -> trmf.cpp:231-245); here one
// extra Hessian-vector product H s (cached Grams, no gather) gives the same reduction without the
// cancellation of subtracting two large objective values.  Every block derives the same decision;
// block 0 records the TRON line values.
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
__global__ __launch_bounds__(256) void accept_kernel(XParams p, XState *__restrict__ st,
                                                     const double *__restrict__ Pbase, int np,
                                                     int np_dot, const double *__restrict__ Prr_final,
                                                     const real *__restrict__ w_new,
                                                     real *__restrict__ w,
                                                     XState *__restrict__ log_x, double *__restrict__ log_norms,
                                                     size_t e_begin, size_t e_end, int direct) {
    __shared__ double smem[256];
    const double gs = (double)(real)sum_partials(Pbase + P_GS * (size_t)p.pstride, np, smem);
    const double sr = (double)(real)sum_partials(Pbase + P_SR * (size_t)p.pstride, np, smem);
    const double sHs = direct ? sum_partials(Pbase + P_DOT * (size_t)p.pstride, np_dot, smem) : 0.0;
    const double snorm = sqrt((double)(real)sum_partials(Pbase + P_SS * (size_t)p.pstride, np, smem));
    const double rho = Prr_final ? (double)(real)sum_partials(Prr_final, np, smem)
                                 : (double)(real)st->rho_hist[st->cg_iter];              // fused CG path
    const double f = st->f;
    const double prered = -0.5 * (gs - sr);                                  // rf_tron.h:190
    // direct (diagnostics): s^T H s from one more operator pass; else through the CG's recurrence r = -g - H s, i.e. s^T H s =
    // -s^T (g + r) and f - f(w+s) = prered (cg_persist.hpp has the argument)
    const double actred = direct ? -(gs + 0.5 * sHs) : prered;
    const double fnew = f - actred;
    const bool accept = actred > 1e-4 * prered;                              // eta0, rf_tron.h:222
    if (accept) {
        for (size_t e = e_begin + (size_t)blockIdx.x * 256 + threadIdx.x; e < e_end; e += (size_t)gridDim.x * 256)
            w[e] = w_new[e];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {          // fields no block reads in this kernel
        st->fnew = fnew; st->gs = gs; st->sr = sr;
        st->prered = prered; st->actred = actred;
        st->accepted = accept ? 1 : 0;
        st->cg_rnorm = sqrt(rho);
        // trust-region bound the reference prints (rf_tron.h:195-215; it never constrains the step here: the folded
        // parameters of trmf.cpp:603-606 run a pure CG pass): delta0 = |g|, first iteration min(delta, |s|), then the
        // update by the ratio of actual to predicted reduction
        double delta = fmin(st->gnorm, snorm);
        const double curv = fnew - f - gs;
        const double alpha = curv <= 0 ? 4.0 : fmax(0.25, -0.5 * (gs / curv));
        if (actred < 1e-4 * prered) delta = fmin(fmax(alpha, 0.25) * snorm, 0.5 * delta);
        else if (actred < 0.25 * prered) delta = fmax(0.25 * delta, fmin(alpha * snorm, 0.5 * delta));
        else if (actred < 0.75 * prered) delta = fmax(0.25 * delta, fmin(alpha * snorm, 4.0 * delta));
        else delta = fmax(delta, fmin(alpha * snorm, 4.0 * delta));
        st->delta = delta;
        st->rho_direct = -1.0;
        if (log_x) {                                    // iteration record written here: no copies on the stream
            log_x->f = f; log_x->fnew = fnew; log_x->gnorm = st->gnorm; log_x->cg_rnorm = sqrt(rho);
            log_x->actred = actred; log_x->prered = prered; log_x->gs = gs; log_x->sr = sr;
            log_x->cgtol = st->cgtol; log_x->cg_iter = st->cg_iter; log_x->accepted = accept ? 1 : 0; log_x->delta = delta; log_x->rho_direct = -1.0;
            log_norms[0] = log_norms[1] = log_norms[2] = -1.0;      // ||.||^2 lines are off in this mode
        }
    }
}
#endif

// ---- closing the CG of the fused path (rf_tron.h:460-461, 183-190) ---------------------------------------------------
// The CG launches stop at the top of iteration `stop_it` without touching anything.  This kernel closes iteration
// stop_it - 1: alpha = rho / <d,Hd> from that launch's records (same fixed-order sum as a CG launch makes),
// s += alpha d, r' = r - alpha Hd (needed only for <s,r'>), w_new = w + s, and the per-tile sums <g,s>, <s,r'>, <s,s>
// into fields [4..6] of the records of the gradient / plain message; with several ranks also the first / last midx rows of
// s (edge vector 0).  One workgroup per tile, own rows only; stop_it = 0 (the gradient met the tolerance): s = 0.
template <bool SHARD, int NTH = 256>
__global__ __launch_bounds__(NTH) void cg_close_kernel(XParams p, const XState *__restrict__ st, TileShard sh, int TI,
                                                      const double *__restrict__ msg_even, const double *__restrict__ msg_odd,
                                                      const real *__restrict__ d_even, const real *__restrict__ d_odd,
                                                      const real *__restrict__ r_even, const real *__restrict__ r_odd,
                                                      const real *__restrict__ h_even, const real *__restrict__ h_odd,
                                                      real *__restrict__ s, const real *__restrict__ g, const real *__restrict__ w,
                                                      real *__restrict__ w_new, double *__restrict__ rec_out,
                                                      const PeerTable *__restrict__ pt) {
    __shared__ double smem[256];
    const int tid = threadIdx.x, it = st->stop_it, KP = p.KP, Hh = p.midx;
    const int tile = (SHARD ? sh.tile0 : 0) + (int)blockIdx.x;
    const int i0 = tile * TI, i1 = min(i0 + TI, p.T);
    const int par = it >= 1 ? (it - 1) & 1 : 0;         // it == 0: r = -g sits in the even buffer, d and H d are not used
    const real *dv = par ? d_odd : d_even, *rv = par ? r_odd : r_even, *hv = par ? h_odd : h_even;
    // the tile's own elements are requested BEFORE the records are summed (the loads do not depend on alpha): a thread
    // keeps up to kPer of them in registers, longer tiles go through the loop below
    constexpr int kPer = 6;
    const int e0 = i0 * KP + tid, e1 = i1 * KP;
    real xd[kPer], xh[kPer], xr[kPer], xs[kPer], xg[kPer], xw[kPer];
#pragma unroll
    for (int m = 0; m < kPer; m++) {
        const int e = min(e0 + NTH * m, e1 - 1);
        xd[m] = dv[e]; xh[m] = hv[e]; xr[m] = rv[e]; xs[m] = s[e]; xg[m] = g[e]; xw[m] = w[e];
    }
    real alpha = 0;
    if (it >= 1) {                                      // uniform: every workgroup (and rank) sees the same stop_it
        const double *msg = par ? msg_odd : msg_even;
        double dHd = 0;
        for (int i = tid; i < sh.nbt; i += NTH) dHd += msg[rec_index<SHARD>(sh, i)];
        dHd = block_allsum<NTH / 64>(dHd, smem);
        alpha = (real)st->rho_hist[it - 1] / (real)dHd;                     // rf_tron.h:460
    }
    const bool p2p = SHARD && pt != nullptr;
    const int edgeN = Hh * KP;
    real *edges_lo = SHARD ? edge_base(p2p && sh.rank > 0 ? pt->msg[2][sh.rank - 1] : rec_out, sh, sh.rank) : nullptr;
    real *edges_hi = SHARD ? edge_base(p2p && sh.rank + 1 < sh.world ? pt->msg[2][sh.rank + 1] : rec_out, sh, sh.rank) : nullptr;
    double gs = 0, sr = 0, ss = 0;
    auto close_elem = [&](int e, real dx, real hx, real rx, real sx, real gx, real wx) {
        real snew = sx, rnew = rx;                      // it == 0: s = 0 and r = -g as the first launch left them
        if (it >= 1) { snew = fma(alpha, dx, sx); rnew = fma(-alpha, hx, rx); s[e] = snew; }   // rf_tron.h:461, 489-490
        w_new[e] = wx + snew;                           // rf_tron.h:183-184
        gs += (double)gx * (double)snew; sr += (double)snew * (double)rnew; ss += (double)snew * (double)snew;
        if (SHARD) {
            const uint32_t lo = (uint32_t)(e - sh.row_b * KP), hi = (uint32_t)(e - (sh.row_e - Hh) * KP);
            if (lo < (uint32_t)edgeN) edges_lo[lo] = snew;
            if (hi < (uint32_t)edgeN) edges_hi[(size_t)kEdgeVecs * edgeN + hi] = snew;
        }
    };
#pragma unroll
    for (int m = 0; m < kPer; m++)
        if (e0 + NTH * m < e1) close_elem(e0 + NTH * m, xd[m], xh[m], xr[m], xs[m], xg[m], xw[m]);
    for (int e = e0 + NTH * kPer; e < e1; e += NTH) close_elem(e, dv[e], hv[e], rv[e], s[e], g[e], w[e]);
    block_allsum3<NTH / 64>(gs, sr, ss, smem);
    if (tid == 0) {
        const size_t ri = rec_index<SHARD>(sh, tile) + 4;
        for (int r = 0; r < (p2p ? sh.world : 1); r++) {
            double *dst = (p2p ? pt->msg[2][r] : rec_out) + ri;
            dst[0] = gs; dst[1] = sr; dst[2] = ss;
        }
    }
    if (p2p) __threadfence_system();
}

// ---- acceptance test and commit of the fused path: sums of the per-tile records -----------------------------------
// <g,s>, <s,r>, <s,s> (cg_close_kernel) and <s,Hs> (plain launch) sit in the same records; with several ranks a rank
// commits its own timestamps [row_b, row_e) (the rows of W are all-gathered next).
#if !defined(TRMF_UNIT)      // compiled by the main translation unit only (kernel_units.hpp)
template <int NTH = 256>
__global__ __launch_bounds__(NTH) void accept_tile_kernel(XParams p, XState *__restrict__ st, const double *__restrict__ msgP,
                                                          TileShard sh, int sharded, const real *__restrict__ w_new,
                                                          real *__restrict__ w, XState *__restrict__ log_x,
                                                          double *__restrict__ log_norms, int direct) {
    __shared__ double smem[256];
    double gs_d = 0, sr_d = 0, ss_d = 0, sHs = 0;
    for (int i = threadIdx.x; i < sh.nbt; i += NTH) {
        const size_t ri = sharded ? rec_index<true>(sh, i) : rec_index<false>(sh, i);
        gs_d += msgP[ri + 4]; sr_d += msgP[ri + 5]; ss_d += msgP[ri + 6];
        if (direct) sHs += msgP[ri + 2];
    }
    block_allsum3<NTH / 64>(gs_d, sr_d, ss_d, smem);
    sHs = block_allsum<NTH / 64>(sHs, smem);
    const double gs = (double)(real)gs_d, sr = (double)(real)sr_d;          // BLAS dots in val_type (rf_tron.h:186-187)
    const double snorm = sqrt((double)(real)ss_d);
    const double rho = (double)(real)st->rho_hist[st->cg_iter];
    const double f = st->f;
    const double prered = -0.5 * (gs - sr);                                  // rf_tron.h:190
    const double actred = direct ? -(gs + 0.5 * sHs) : prered;               // f - f(w+s): directly / through the recurrence (accept_kernel)
    const double fnew = f - actred;
    const bool accept = actred > 1e-4 * prered;                              // eta0, rf_tron.h:222
    if (accept) {
        const size_t e0 = (size_t)sh.row_b * p.KP, e1 = (size_t)sh.row_e * p.KP;
        for (size_t e = e0 + (size_t)blockIdx.x * NTH + threadIdx.x; e < e1; e += (size_t)gridDim.x * NTH) w[e] = w_new[e];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {          // fields no block reads in this kernel
        st->fnew = fnew; st->gs = gs; st->sr = sr;
        st->prered = prered; st->actred = actred;
        st->accepted = accept ? 1 : 0;
        st->cg_rnorm = sqrt(rho);
        st->rho_direct = -1.0;
        double delta = fmin(st->gnorm, snorm);          // trust-region bound of the TRON line: see accept_kernel
        const double curv = fnew - f - gs;
        const double alpha = curv <= 0 ? 4.0 : fmax(0.25, -0.5 * (gs / curv));
        if (actred < 1e-4 * prered) delta = fmin(fmax(alpha, 0.25) * snorm, 0.5 * delta);
        else if (actred < 0.25 * prered) delta = fmax(0.25 * delta, fmin(alpha * snorm, 0.5 * delta));
        else if (actred < 0.75 * prered) delta = fmax(0.25 * delta, fmin(alpha * snorm, 4.0 * delta));
        else delta = fmax(delta, fmin(alpha * snorm, 4.0 * delta));
        st->delta = delta;
        if (log_x) {                                    // iteration record written here: no copies on the stream
            log_x->f = f; log_x->fnew = fnew; log_x->gnorm = st->gnorm; log_x->cg_rnorm = sqrt(rho);
            log_x->actred = actred; log_x->prered = prered; log_x->gs = gs; log_x->sr = sr;
            log_x->cgtol = st->cgtol; log_x->cg_iter = st->cg_iter; log_x->accepted = accept ? 1 : 0; log_x->delta = delta; log_x->rho_direct = -1.0;
            log_norms[0] = log_norms[1] = log_norms[2] = -1.0;      // ||.||^2 lines are off in this mode
        }
    }
}
#endif

}  // namespace trmf
