/*
 * trmf_oracle.c -- TEST INFRASTRUCTURE ONLY (see trmf_oracle.h for the rules).
 *
 * Plain-C restatement of the TRMF ALS hot path.  Each function cites the reference lines it
 * follows (paths relative to /root/reference/python/trmf/corelib/).  Arithmetic types follow the
 * reference: `oreal` where the reference uses val_type, `double` where it uses double.
 *
 * Known, deliberate deviations from the reference build (all last-bit class; tolerances are stated
 * in tests/):
 *   - BLAS {s,d}dot over long vectors: accumulated here in double and rounded to oreal once
 *     (OpenBLAS/MKL accumulate in SIMD partial sums of val_type; order is library-dependent).
 *   - LAPACK posv('U'): restated as the unblocked upper Cholesky (LAPACK potf2 'U' loop order)
 *     followed by the two triangular solves, all in oreal.
 */
#include "trmf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_sizeof_real(void) { return (int)sizeof(oreal); }

static void set_threads(int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);   /* trmf.cpp:636 */
#else
    (void)threads;
#endif
}

/* ---- BLAS-1 stand-ins (rf_matrix.h:2495-2509) ------------------------------------------------ */
static oreal dot_r(size_t n, const oreal *x, const oreal *y) {
    double acc = 0.0;
#pragma omp parallel for reduction(+:acc) schedule(static)
    for (size_t i = 0; i < n; i++) acc += (double)x[i] * (double)y[i];
    return (oreal)acc;
}
static void axpy_r(size_t n, oreal a, const oreal *x, oreal *y) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) y[i] += a * x[i];
}

/* ---- posv('U') restatement (rf_matrix.h:3008-3014) ------------------------------------------- */
/* A: n x n symmetric (row-major == col-major), only the upper triangle (in the col-major sense:
 * element (i,j), i<=j, stored at A[j*n+i]) is referenced, exactly as LAPACK does with uplo='U'.
 * Because the caller mirrors the matrix, A[j*n+i] == A[i*n+j] on entry. Returns 0 on success. */
static int chol_solve_upper(oreal *A, oreal *b, size_t n, size_t nrhs, size_t ldb) {
#define U(i, j) A[(j) * n + (i)]
    for (size_t j = 0; j < n; j++) {
        oreal ajj = U(j, j);
        for (size_t p = 0; p < j; p++) ajj -= U(p, j) * U(p, j);
        if (!(ajj > 0)) return (int)(j + 1);
        ajj = (oreal)sqrt((double)ajj);
        U(j, j) = ajj;
        for (size_t c = j + 1; c < n; c++) {
            oreal s = U(j, c);
            for (size_t p = 0; p < j; p++) s -= U(p, j) * U(p, c);
            U(j, c) = s / ajj;
        }
    }
    for (size_t r = 0; r < nrhs; r++) {
        oreal *x = b + r * ldb;
        /* U^T z = b */
        for (size_t i = 0; i < n; i++) {
            oreal s = x[i];
            for (size_t p = 0; p < i; p++) s -= U(p, i) * x[p];
            x[i] = s / U(i, i);
        }
        /* U x = z */
        for (size_t ii = n; ii-- > 0;) {
            oreal s = x[ii];
            for (size_t p = ii + 1; p < n; p++) s -= U(ii, p) * x[p];
            x[ii] = s / U(ii, ii);
        }
    }
#undef U
    return 0;
}

/* ---- F-solve: trmf.cpp:369-397 ---------------------------------------------------------------- */
void oracle_fsolve_sparse(uint64_t nrows, const uint64_t *ptr, const uint32_t *idx,
                          const oreal *val, const oreal *X, uint64_t k, double lambda_d,
                          oreal *F, int threads) {
    set_threads(threads);
    const oreal lambda = (oreal)lambda_d;                 /* trmf.cpp:358,537: cast to val_type */
#pragma omp parallel
    {
        oreal *Hess = (oreal *)malloc(sizeof(oreal) * k * k);
#pragma omp for schedule(dynamic, 64)
        for (uint64_t i = 0; i < nrows; i++) {
            if (ptr[i + 1] == ptr[i]) continue;             /* :374 empty rows untouched */
            oreal *y = F + i * k;                           /* :379 rhs accumulates in the output */
            memset(Hess, 0, sizeof(oreal) * k * k);
            memset(y, 0, sizeof(oreal) * k);
            for (uint64_t p = ptr[i]; p != ptr[i + 1]; p++) {
                const oreal *xj = X + (size_t)idx[p] * k;
                const oreal v = val[p];
                for (uint64_t s = 0; s < k; s++) {          /* :384-388 upper triangle only */
                    y[s] += v * xj[s];
                    for (uint64_t t = s; t < k; t++) Hess[s * k + t] += xj[s] * xj[t];
                }
            }
            for (uint64_t s = 0; s < k; s++) {              /* :390-394 mirror, then +lambda */
                for (uint64_t t = 0; t < s; t++) Hess[s * k + t] = Hess[t * k + s];
                Hess[s * k + s] += lambda;
            }
            chol_solve_upper(Hess, y, k, 1, k);             /* :395, info ignored */
        }
        free(Hess);
    }
}

/* ---- AR + ridge term: trmf.cpp:70-149 (arr_base_IX) ------------------------------------------ */
#define TH(l, t) theta[(size_t)(t) * nlag + (l)]           /* col-major |L| x k, rf_matrix.h:1273 */

static double base_fun(uint64_t T, uint64_t k, const oreal *W, const uint32_t *lag_set,
                       uint32_t nlag, const oreal *theta, double lambdaI, double lambdaAR) {
    double f = 0;
    if (lambdaI > 0) f += 0.5 * lambdaI * (double)dot_r(T * k, W, W);      /* :73-75 */
    if (lag_set != NULL && nlag > 0 && lambdaAR > 0) {
        const uint64_t midx = lag_set[nlag - 1];                           /* :79 */
        double AR_val = 0;
#pragma omp parallel for reduction(+:AR_val) schedule(static)
        for (uint64_t i = midx; i < T; i++) {
            double tmp = 0;
            for (uint64_t t = 0; t < k; t++) {
                double residual = W[i * k + t];
                for (uint32_t l = 0; l < nlag; l++)
                    residual -= (double)(oreal)(TH(l, t) * W[(i - lag_set[l]) * k + t]);
                tmp += residual * residual;
            }
            AR_val += tmp;
        }
        f += 0.5 * lambdaAR * AR_val;
    }
    return f;
}

/* grad (:99-123) and Hv (:125-149) share one body: OUT = lambdaI*IN + lambdaAR * AR-operator(IN) */
static void base_apply(uint64_t T, uint64_t k, const oreal *IN, const uint32_t *lag_set,
                       uint32_t nlag, const oreal *theta, double lambdaI, double lambdaAR,
                       oreal *OUT) {
    const size_t N = (size_t)T * k;
    if (lambdaI == 0) {                                                    /* rf_matrix.h:972-1003 */
        memset(OUT, 0, sizeof(oreal) * N);
    } else if (lambdaI == 1) {
        if (OUT != IN) memcpy(OUT, IN, sizeof(oreal) * N);
    } else {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < N; i++) OUT[i] = (oreal)(lambdaI * (double)IN[i]);
    }
    if (lag_set != NULL && nlag > 0 && lambdaAR > 0) {
        const uint64_t midx = lag_set[nlag - 1];
#pragma omp parallel for schedule(static)
        for (uint64_t t = 0; t < k; t++) {                 /* parallel over latent dims only */
            for (uint64_t i = midx; i < T; i++) {
                double residual = IN[i * k + t];
                for (uint32_t l = 0; l < nlag; l++)
                    residual -= (double)(oreal)(TH(l, t) * IN[(i - lag_set[l]) * k + t]);
                OUT[i * k + t] = (oreal)((double)OUT[i * k + t] + lambdaAR * residual);
                for (uint32_t l = 0; l < nlag; l++) {
                    size_t o = (size_t)(i - lag_set[l]) * k + t;
                    OUT[o] = (oreal)((double)OUT[o] - lambdaAR * residual * (double)TH(l, t));
                }
            }
        }
    }
}

/* ---- X-solve loss part, sparse: trmf.cpp:231-288 (arr_ls_pY_IX) ------------------------------- */
double oracle_xfun_sparse(uint64_t T, const uint64_t *ptr, const uint32_t *idx, const oreal *val,
                          const oreal *H, uint64_t k, const oreal *W,
                          const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                          double lambdaI, double lambdaAR) {
    double f = 0;
#pragma omp parallel for schedule(dynamic, 32) reduction(+:f)
    for (uint64_t i = 0; i < T; i++) {
        const oreal *wi = W + i * k;
        for (uint64_t p = ptr[i]; p != ptr[i + 1]; p++) {
            const oreal *hj = H + (size_t)idx[p] * k;
            oreal d = 0;                                   /* :238 BLAS dot in val_type */
            for (uint64_t t = 0; t < k; t++) d += wi[t] * hj[t];
            double residual = (double)(oreal)(val[p] - d);
            f += residual * residual;
        }
    }
    f *= 0.5;
    f += base_fun(T, k, W, lag_set, nlag, theta, lambdaI, lambdaAR);
    return f;
}

void oracle_xgrad_sparse(uint64_t T, const uint64_t *ptr, const uint32_t *idx, const oreal *val,
                         const oreal *H, uint64_t k, const oreal *W,
                         const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                         double lambdaI, double lambdaAR, oreal *G) {
    base_apply(T, k, W, lag_set, nlag, theta, lambdaI, lambdaAR, G);
#pragma omp parallel for schedule(dynamic, 32)
    for (uint64_t i = 0; i < T; i++) {
        const oreal *wi = W + i * k;
        oreal *gi = G + i * k;
        for (uint64_t p = ptr[i]; p != ptr[i + 1]; p++) {
            const oreal *hj = H + (size_t)idx[p] * k;
            double residual = -(double)val[p];                               /* :258 */
            for (uint64_t t = 0; t < k; t++) residual += (double)(oreal)(wi[t] * hj[t]);
            for (uint64_t t = 0; t < k; t++)
                gi[t] = (oreal)((double)gi[t] + residual * (double)hj[t]);   /* :263 */
        }
    }
}

void oracle_xhv_sparse(uint64_t T, const uint64_t *ptr, const uint32_t *idx,
                       const oreal *H, uint64_t k, const oreal *S,
                       const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                       double lambdaI, double lambdaAR, oreal *HS) {
    base_apply(T, k, S, lag_set, nlag, theta, lambdaI, lambdaAR, HS);
#pragma omp parallel for schedule(dynamic, 32)
    for (uint64_t i = 0; i < T; i++) {
        const oreal *si = S + i * k;
        oreal *hsi = HS + i * k;
        for (uint64_t p = ptr[i]; p != ptr[i + 1]; p++) {
            const oreal *hj = H + (size_t)idx[p] * k;
            double residual = 0;                                             /* :279 */
            for (uint64_t t = 0; t < k; t++) residual += (double)(oreal)(si[t] * hj[t]);
            for (uint64_t t = 0; t < k; t++)
                hsi[t] = (oreal)((double)hsi[t] + residual * (double)hj[t]);
        }
    }
}

/* ---- generic "function" object so the CG driver serves both the sparse and the full path ----- */
typedef struct xfunc {
    uint64_t T, n, k;
    const uint64_t *ptr; const uint32_t *idx; const oreal *val;  /* sparse path */
    const oreal *H;
    const uint32_t *lag_set; uint32_t nlag; const oreal *theta;
    double lambdaI, lambdaAR;
    /* full path (arr_ls_fY_IX, trmf.cpp:155-215) */
    int full;
    double trYTY; const oreal *YH; const oreal *HTH;
} xfunc;

static double xf_fun(const xfunc *F, const oreal *W) {
    if (!F->full)
        return oracle_xfun_sparse(F->T, F->ptr, F->idx, F->val, F->H, F->k, W, F->lag_set, F->nlag,
                                  F->theta, F->lambdaI, F->lambdaAR);
    const uint64_t T = F->T, k = F->k;
    double f = base_fun(T, k, W, F->lag_set, F->nlag, F->theta, F->lambdaI, F->lambdaAR);
    f += 0.5 * F->trYTY;                                                     /* :192 */
    double *WTW = (double *)calloc(k * k, sizeof(double));
    for (uint64_t i = 0; i < T; i++)
        for (uint64_t a = 0; a < k; a++)
            for (uint64_t b = 0; b < k; b++) WTW[a * k + b] += (double)W[i * k + a] * (double)W[i * k + b];
    double tr = 0;
    for (uint64_t a = 0; a < k * k; a++) tr += (double)(oreal)WTW[a] * (double)F->HTH[a];
    free(WTW);
    f += 0.5 * (double)(oreal)tr;                                            /* :194 */
    f -= (double)dot_r(T * k, F->YH, W);                                     /* :195 */
    return f;
}

static void xf_WxHTH_add(const xfunc *F, const oreal *S, oreal *OUT) {       /* OUT += S * HTH */
    const uint64_t T = F->T, k = F->k;
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < T; i++)
        for (uint64_t b = 0; b < k; b++) {
            double acc = 0;
            for (uint64_t a = 0; a < k; a++) acc += (double)S[i * k + a] * (double)F->HTH[a * k + b];
            OUT[i * k + b] = (oreal)((double)OUT[i * k + b] + acc);
        }
}

static void xf_grad(const xfunc *F, const oreal *W, oreal *G) {
    if (!F->full) {
        oracle_xgrad_sparse(F->T, F->ptr, F->idx, F->val, F->H, F->k, W, F->lag_set, F->nlag,
                            F->theta, F->lambdaI, F->lambdaAR, G);
        return;
    }
    base_apply(F->T, F->k, W, F->lag_set, F->nlag, F->theta, F->lambdaI, F->lambdaAR, G);
    axpy_r((size_t)F->T * F->k, (oreal)-1.0, F->YH, G);                      /* :205 */
    xf_WxHTH_add(F, W, G);                                                   /* :206 */
}

static void xf_Hv(const xfunc *F, const oreal *S, oreal *HS) {
    if (!F->full) {
        oracle_xhv_sparse(F->T, F->ptr, F->idx, F->H, F->k, S, F->lag_set, F->nlag, F->theta,
                          F->lambdaI, F->lambdaAR, HS);
        return;
    }
    base_apply(F->T, F->k, S, F->lag_set, F->nlag, F->theta, F->lambdaI, F->lambdaAR, HS);
    xf_WxHTH_add(F, S, HS);                                                  /* :213 */
}

/* ---- TRON reduced to one CG pass: rf_tron.h:134-254 (tron_trustregion), :412-505 (trcg) ------- */
static void tron_solve(const xfunc *F, oreal *w, int max_cg_iter, double eps_cg, double eps,
                       OracleXStats *st) {
    const size_t n = (size_t)F->T * F->k;
    const double eta0 = 1e-4;
    oreal *s = (oreal *)calloc(n, sizeof(oreal)), *r = (oreal *)calloc(n, sizeof(oreal));
    oreal *w_new = (oreal *)calloc(n, sizeof(oreal)), *g = (oreal *)calloc(n, sizeof(oreal));
    oreal *d = (oreal *)calloc(n, sizeof(oreal)), *Hd = (oreal *)calloc(n, sizeof(oreal));
    if (max_cg_iter > 0 && (size_t)max_cg_iter >= n) max_cg_iter = (int)n;   /* trmf.cpp:523-526 */

    double f = xf_fun(F, w);
    xf_grad(F, w, g);
    double gnorm1 = sqrt((double)dot_r(n, g, g));                            /* :164 */
    double gnorm = gnorm1;
    int search = (gnorm <= eps * gnorm1) ? 0 : 1;                            /* :169 */
    OracleXStats z; memset(&z, 0, sizeof z); z.f = f; z.fnew = f; z.gnorm = gnorm;

    /* max_tron_iter == 1 after the fold (trmf.cpp:603-606).  A rejected step would repeat the
     * identical computation (pure_cg ignores delta), so retries are capped. */
    int iter = 1, tries = 0;
    while (iter <= 1 && search && tries < 2) {
        tries++;
        /* ---- trcg, rf_tron.h:412-505, pure_cg = true ---- */
        for (size_t i = 0; i < n; i++) { s[i] = 0; r[i] = -g[i]; d[i] = r[i]; }
        const oreal cgtol = (oreal)(eps_cg * sqrt((double)dot_r(n, g, g)));  /* :434 */
        int cg_iter = 0;
        oreal rTr = dot_r(n, r, r);
        double cg_rnorm = 0;
        for (;;) {
            cg_rnorm = sqrt((double)dot_r(n, r, r));                         /* :444 */
            if (cg_rnorm <= (double)cgtol) break;
            if (max_cg_iter > 0 && cg_iter >= max_cg_iter) break;
            cg_iter++;
            xf_Hv(F, d, Hd);
            oreal alpha = rTr / dot_r(n, d, Hd);                             /* :460 */
            axpy_r(n, alpha, d, s);
            alpha = -alpha;
            axpy_r(n, alpha, Hd, r);
            oreal rnew = dot_r(n, r, r);
            oreal beta = rnew / rTr;
            oreal tmp = beta - (oreal)1.0;                                   /* :497-501 */
            axpy_r(n, tmp, d, d);
            axpy_r(n, (oreal)1.0, r, d);
            rTr = rnew;
        }
        /* ---- back in tron_trustregion ---- */
        memcpy(w_new, w, sizeof(oreal) * n);
        axpy_r(n, (oreal)1.0, s, w_new);
        double gs = (double)dot_r(n, g, s);
        double prered = -0.5 * (gs - (double)dot_r(n, s, r));                /* :190 */
        double fnew = xf_fun(F, w_new);
        double actred = f - fnew;
        z.cg_iter = cg_iter; z.cg_rnorm = cg_rnorm; z.f = f; z.fnew = fnew;
        z.actred = actred; z.prered = prered; z.gnorm = gnorm;
        if (actred > eta0 * prered) {                                        /* :222 */
            iter++;
            memcpy(w, w_new, sizeof(oreal) * n);
            f = fnew;
            z.accepted = 1;
            /* the reference recomputes grad here (:229) but never uses it: skipped */
        }
        if (f < -1.0e+32) break;
        if (fabs(actred) <= 0 && prered <= 0) break;
        if (fabs(actred) <= 1.0e-12 * fabs(f) && fabs(prered) <= 1.0e-12 * fabs(f)) break;
    }
    if (st) *st = z;
    free(s); free(r); free(w_new); free(g); free(d); free(Hd);
}

void oracle_xsolve_sparse(uint64_t T, const uint64_t *ptr, const uint32_t *idx, const oreal *val,
                          const oreal *H, uint64_t k, oreal *W,
                          const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                          double lambdaI, double lambdaAR, int max_cg_iter, double eps_cg,
                          int threads, OracleXStats *stats) {
    set_threads(threads);
    xfunc F; memset(&F, 0, sizeof F);
    F.T = T; F.k = k; F.ptr = ptr; F.idx = idx; F.val = val; F.H = H;
    F.lag_set = lag_set; F.nlag = nlag; F.theta = theta; F.lambdaI = lambdaI; F.lambdaAR = lambdaAR;
    tron_solve(&F, W, max_cg_iter, eps_cg, 0.1, stats);
}

/* ---- Theta solve: trmf.cpp:447-484 ------------------------------------------------------------- */
void oracle_theta_solve(uint64_t T, uint64_t k, const oreal *W, const uint32_t *lag_set,
                        uint32_t nlag, double lambdaLag, oreal *theta, int threads) {
    set_threads(threads);
    if (nlag == 0) return;
    const uint64_t start = lag_set[nlag - 1], end = T;
#pragma omp parallel
    {
        oreal *series = (oreal *)malloc(sizeof(oreal) * T);
        oreal *Hess = (oreal *)malloc(sizeof(oreal) * nlag * nlag);
#pragma omp for schedule(static)
        for (uint64_t t = 0; t < k; t++) {
            for (uint64_t i = 0; i < T; i++) series[i] = W[i * k + t];
            oreal *y = theta + (size_t)t * nlag;           /* column t of col-major lag_val */
            for (uint32_t a = 0; a < nlag; a++) {
                const uint64_t la = lag_set[a];
                double acc = 0;                            /* :447-453 double accumulators */
                for (uint64_t i = start; i < end; i++) acc += (double)(oreal)(series[i] * series[i - la]);
                y[a] = (oreal)acc;
                for (uint32_t b = a; b < nlag; b++) {
                    const uint64_t lb = lag_set[b];
                    double h = 0;
                    for (uint64_t i = start; i < end; i++)
                        h += (double)(oreal)(series[i - la] * series[i - lb]);
                    Hess[a * nlag + b] = (oreal)h;
                }
            }
            for (uint32_t a = 0; a < nlag; a++) {          /* :476-481 */
                for (uint32_t b = 0; b < a; b++) Hess[a * nlag + b] = Hess[b * nlag + a];
                Hess[a * nlag + a] = (oreal)((double)Hess[a * nlag + a] + lambdaLag);
            }
            chol_solve_upper(Hess, y, nlag, 1, nlag);
        }
        free(series); free(Hess);
    }
}

/* ---- Global objective in fp64 ------------------------------------------------------------------ */
double oracle_objective_sparse(uint64_t T, uint64_t n, const uint64_t *ptr, const uint32_t *idx,
                               const oreal *val, const oreal *W, const oreal *H, uint64_t k,
                               const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                               double lambdaI, double lambdaAR) {
    double loss = 0, w2 = 0, h2 = 0, ar = 0;
#pragma omp parallel for schedule(dynamic, 32) reduction(+:loss)
    for (uint64_t i = 0; i < T; i++)
        for (uint64_t p = ptr[i]; p != ptr[i + 1]; p++) {
            const oreal *hj = H + (size_t)idx[p] * k;
            double d = (double)val[p];
            for (uint64_t t = 0; t < k; t++) d -= (double)W[i * k + t] * (double)hj[t];
            loss += d * d;
        }
#pragma omp parallel for reduction(+:w2) schedule(static)
    for (size_t i = 0; i < (size_t)T * k; i++) w2 += (double)W[i] * (double)W[i];
#pragma omp parallel for reduction(+:h2) schedule(static)
    for (size_t i = 0; i < (size_t)n * k; i++) h2 += (double)H[i] * (double)H[i];
    if (nlag > 0) {
        const uint64_t midx = lag_set[nlag - 1];
#pragma omp parallel for reduction(+:ar) schedule(static)
        for (uint64_t i = midx; i < T; i++)
            for (uint64_t t = 0; t < k; t++) {
                double res = (double)W[i * k + t];
                for (uint32_t l = 0; l < nlag; l++)
                    res -= (double)TH(l, t) * (double)W[(i - lag_set[l]) * k + t];
                ar += res * res;
            }
    }
    return 0.5 * loss + 0.5 * lambdaI * (w2 + h2) + 0.5 * lambdaAR * ar;
}

/* ---- Full-observation helpers: trmf.cpp:155-215, 299-351 --------------------------------------- */
/* Y (rows x cols) in any PyMatrix form; element access for the dense forms, zeros = observed 0
 * for the sparse form (exactly what gmat_x_dmat / do_dot_product do on a sparse Y). */
static void full_YH(const OracleMatrix *Y, int transposed, const oreal *H, uint64_t k, oreal *YH) {
    /* YH = op(Y) * H, op(Y) is R x C */
    const uint64_t R = transposed ? Y->cols : Y->rows, C = transposed ? Y->rows : Y->cols;
    if (Y->type == 3) {
        const uint64_t *ptr = transposed ? Y->col_ptr : Y->row_ptr;
        const uint32_t *idx = transposed ? Y->row_idx : Y->col_idx;
        const oreal *val = (const oreal *)(transposed ? Y->val : Y->val_t);
#pragma omp parallel for schedule(dynamic, 64)
        for (uint64_t i = 0; i < R; i++)
            for (uint64_t t = 0; t < k; t++) {
                double acc = 0;
                for (uint64_t p = ptr[i]; p != ptr[i + 1]; p++) acc += (double)val[p] * (double)H[(size_t)idx[p] * k + t];
                YH[i * k + t] = (oreal)acc;
            }
        return;
    }
    const oreal *v = (const oreal *)Y->val;
    const int rowmajor = (Y->type == 1);
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < R; i++)
        for (uint64_t t = 0; t < k; t++) {
            double acc = 0;
            for (uint64_t j = 0; j < C; j++) {
                uint64_t r = transposed ? j : i, c = transposed ? i : j;
                oreal y = rowmajor ? v[r * Y->cols + c] : v[c * Y->rows + r];
                acc += (double)y * (double)H[j * k + t];
            }
            YH[i * k + t] = (oreal)acc;
        }
}
static void full_gram(const oreal *H, uint64_t n, uint64_t k, oreal *HTH) {   /* HTH = H^T H */
    double *acc = (double *)calloc(k * k, sizeof(double));
    for (uint64_t j = 0; j < n; j++)
        for (uint64_t a = 0; a < k; a++)
            for (uint64_t b = 0; b < k; b++) acc[a * k + b] += (double)H[j * k + a] * (double)H[j * k + b];
    for (uint64_t a = 0; a < k * k; a++) HTH[a] = (oreal)acc[a];
    free(acc);
}
static double full_trYTY(const OracleMatrix *Y) {
    const oreal *v = (const oreal *)(Y->type == 3 ? Y->val_t : Y->val);
    return (double)dot_r((size_t)Y->nnz, v, v);
}

/* ---- Driver: trmf.cpp:599-725 -------------------------------------------------------------------- */
static int check_dimension(const OracleMatrix *Y, const OracleMatrix *W, const OracleMatrix *H,
                           const OracleMatrix *LV, uint32_t lag_size) {       /* trmf.cpp:561-596 */
    int pass = 1;
    if (Y->rows != W->rows) { fprintf(stderr, "[ERR MSG]: Y.rows (%ld) != W.rows (%ld)\n", (long)Y->rows, (long)W->rows); pass = 0; }
    if (Y->cols != H->rows) { fprintf(stderr, "[ERR MSG]: Y.cols (%ld) != H.rows (%ld)\n", (long)Y->cols, (long)H->rows); pass = 0; }
    if (W->cols != H->cols) { fprintf(stderr, "[ERR MSG]: W.cols (%ld) != H.cols (%ld)\n", (long)W->cols, (long)H->cols); pass = 0; }
    if (lag_size != LV->rows) { fprintf(stderr, "[ERR MSG]: lag_set.size(%ld) != lag_val.rows(%ld)\n", (long)lag_size, (long)LV->rows); pass = 0; }
    if (W->cols != LV->cols) { fprintf(stderr, "[ERR MSG]: W.cols(%ld) != lag_val.cols(%ld)\n", (long)W->cols, (long)LV->cols); pass = 0; }
    if (W->type != 1) { fprintf(stderr, "[ERR MSG]: W should be rowmajored\n"); pass = 0; }
    if (H->type != 1) { fprintf(stderr, "[ERR MSG]: H should be rowmajored\n"); pass = 0; }
    if (LV->type != 2) { fprintf(stderr, "[ERR MSG]: lag_val should be colmajored\n"); pass = 0; }
    return pass;
}

void oracle_trmf_train_log(const OracleMatrix *Y, const uint32_t *lag_set, uint32_t lag_size,
                           OracleMatrix *Wm, OracleMatrix *Hm, OracleMatrix *LVm, int warm_start,
                           double lambdaI, double lambdaAR, double lambdaLag,
                           int32_t max_iter, int32_t period_W, int32_t period_H, int32_t period_Lag,
                           int32_t threads, int32_t missing, int32_t verbose, OracleIterLog *log) {
    /* defaults not exposed through the ABI: trmf.h:85-96, folded by trmf.cpp:603-606 */
    const double eps = 0.1, eps_cg = 0.1;
    const int max_cg_iter = 10 * 2;
    if (!warm_start) return;        /* quirk Q1: the cold path never updates the caller */
    if (!check_dimension(Y, Wm, Hm, LVm, lag_size)) return;
    if (missing && Y->type != 3) { fprintf(stderr, "[ERR MSG]: missing!=0 needs a sparse Y\n"); return; }
    set_threads(threads);

    const uint64_t T = Y->rows, n = Y->cols, k = Wm->cols;
    oreal *W = (oreal *)Wm->val, *H = (oreal *)Hm->val, *theta = (oreal *)LVm->val;
    oreal *YHt = NULL, *YH = NULL, *G1 = NULL, *G2 = NULL;
    double trYTY = 0;
    if (!missing) {
        YHt = (oreal *)malloc(sizeof(oreal) * n * k); YH = (oreal *)malloc(sizeof(oreal) * T * k);
        G1 = (oreal *)malloc(sizeof(oreal) * k * k);  G2 = (oreal *)malloc(sizeof(oreal) * k * k);
        trYTY = full_trYTY(Y);
    }

    for (int iter = 1; iter <= max_iter; iter++) {
        OracleIterLog L; memset(&L, 0, sizeof L); L.normF = L.normX = L.normLV = -1;
        if (period_H > 0 && (iter % period_H) == 0) {                          /* trmf.cpp:654-663 */
            if (missing) {
                oracle_fsolve_sparse(n, Y->col_ptr, Y->row_idx, (const oreal *)Y->val, W, k, lambdaI, H, threads);
            } else {                                                           /* :319-337 */
                full_YH(Y, 1, W, k, YHt);
                full_gram(W, T, k, G1);
                for (uint64_t t = 0; t < k; t++) G1[t * k + t] += (oreal)lambdaI;
                memcpy(H, YHt, sizeof(oreal) * n * k);
                chol_solve_upper(G1, H, k, n, k);
            }
            L.normF = (double)dot_r((size_t)n * k, H, H);
            if (verbose) fprintf(stderr, ">> iter %d F %g\n", iter, L.normF);
        }
        if (period_W > 0 && (iter % period_W) == 0) {                          /* :665-674 */
            xfunc F; memset(&F, 0, sizeof F);
            F.T = T; F.n = n; F.k = k; F.H = H; F.lag_set = lag_set; F.nlag = lag_size; F.theta = theta;
            F.lambdaI = lambdaI; F.lambdaAR = lambdaAR;
            if (missing) { F.ptr = Y->row_ptr; F.idx = Y->col_idx; F.val = (const oreal *)Y->val_t; }
            else { F.full = 1; full_YH(Y, 0, H, k, YH); full_gram(H, n, k, G2); F.trYTY = trYTY; F.YH = YH; F.HTH = G2; }
            tron_solve(&F, W, max_cg_iter, eps_cg, eps, &L.x);
            L.normX = (double)dot_r((size_t)T * k, W, W);
            if (verbose) fprintf(stderr, ">> iter %d X %g\n", iter, L.normX);
            if (verbose >= 2)
                fprintf(stdout, "iter  1 act %5.3e pre %5.3e f %5.3e |g| %5.3e CG %3d |g| %5.3e\n",
                        L.x.actred, L.x.prered, L.x.f, L.x.gnorm, L.x.cg_iter, L.x.cg_rnorm);
        }
        if (period_Lag > 0 && (iter % period_Lag) == 0) {                      /* :677-689 */
            oracle_theta_solve(T, k, W, lag_set, lag_size, lambdaLag, theta, threads);
            L.normLV = (double)dot_r((size_t)lag_size * k, theta, theta);
            if (verbose) fprintf(stderr, ">> iter %d LV %g\n", iter, L.normLV);
        }
        if (log) log[iter - 1] = L;
    }
    free(YHt); free(YH); free(G1); free(G2);
}

void oracle_trmf_train(const OracleMatrix *Y, const uint32_t *lag_set, uint32_t lag_size,
                       OracleMatrix *W, OracleMatrix *H, OracleMatrix *lag_val, int warm_start,
                       double lambdaI, double lambdaAR, double lambdaLag,
                       int32_t max_iter, int32_t period_W, int32_t period_H, int32_t period_Lag,
                       int32_t threads, int32_t missing, int32_t verbose) {
    oracle_trmf_train_log(Y, lag_set, lag_size, W, H, lag_val, warm_start, lambdaI, lambdaAR,
                          lambdaLag, max_iter, period_W, period_H, period_Lag, threads, missing,
                          verbose, NULL);
}
