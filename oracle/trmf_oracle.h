/*
 * trmf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C + OpenMP) of the TRMF alternating-least-squares hot path of
 * rofuyu/exp-trmf-nips16 (python/trmf/corelib/{trmf.cpp,rf_tron.h}).  It exists so that the HIP
 * product path can be checked on a GPU box where the reference sources do not exist.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product library
 * (exp-trmf-nips16_amd/trmf/corelib/trmf_float{32,64}.so) never links, loads or calls it.
 *
 * Parity status: PINNED against the real reference built by `make -C oracle ref`
 * (oracle/_ref/trmf_float{32,64}.so) by tests/test_oracle_vs_ref.py and by the committed golden
 * vectors under tests/golden/ (generated from oracle/_ref by tests/golden/make_golden.py).  The
 * reference itself ships no tests or golden vectors.
 *
 * Built twice: -DORACLE_REAL=float -> libtrmf_oracle_f32.so, -DORACLE_REAL=double -> ..._f64.so.
 * Naming follows the reference's C++ core: Y is T x n (rows = timestamps), W is T x k (temporal
 * factor X^T), H is n x k (item factor F), lag_val is |L| x k COLUMN-major (Theta).
 */
#ifndef TRMF_ORACLE_H
#define TRMF_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifndef ORACLE_REAL
#define ORACLE_REAL float
#endif
typedef ORACLE_REAL oreal;

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as the reference's PyMatrix (rf_matrix.h:3398-3416): sizeof == 80. */
typedef struct {
    uint64_t rows, cols, nnz;
    uint64_t *row_ptr;   /* CSR: row_ptr, col_idx, val_t */
    uint64_t *col_ptr;   /* CSC: col_ptr, row_idx, val   */
    uint32_t *row_idx;
    uint32_t *col_idx;
    void *val;
    void *val_t;
    int32_t type;        /* 1 dense row-major, 2 dense col-major, 3 sparse, 4 eye */
} OracleMatrix;

/* Per-X-solve diagnostics, mirroring the TRON info line (rf_tron.h:219). */
typedef struct {
    int32_t cg_iter;
    int32_t accepted;
    double f, fnew, actred, prered, gnorm, cg_rnorm;
} OracleXStats;

/* Per-ALS-iteration log, mirroring the verbose>=1 stderr lines (trmf.cpp:661,672,687). */
typedef struct {
    double normF;      /* ||H||^2 after the F(H)-solve   (-1 if the phase did not run) */
    double normX;      /* ||W||^2 after the X(W)-solve   */
    double normLV;     /* ||Theta||^2 after the lag solve */
    OracleXStats x;
} OracleIterLog;

int oracle_sizeof_real(void);

/* F-solve, trmf.cpp:369-397 (l2r_ls_pY_IX_chol::solve). CSR over `nrows` items; X is the gathered
 * factor (rows indexed by idx[]); F (nrows x k) is overwritten row by row; empty rows untouched. */
void oracle_fsolve_sparse(uint64_t nrows, const uint64_t *ptr, const uint32_t *idx,
                          const oreal *val, const oreal *X, uint64_t k, double lambda,
                          oreal *F, int threads);

/* X-solve pieces, trmf.cpp:70-149 + 231-288 (arr_base_IX + arr_ls_pY_IX). CSR over T timestamps. */
double oracle_xfun_sparse(uint64_t T, const uint64_t *ptr, const uint32_t *idx, const oreal *val,
                          const oreal *H, uint64_t k, const oreal *W,
                          const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                          double lambdaI, double lambdaAR);
void oracle_xgrad_sparse(uint64_t T, const uint64_t *ptr, const uint32_t *idx, const oreal *val,
                         const oreal *H, uint64_t k, const oreal *W,
                         const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                         double lambdaI, double lambdaAR, oreal *G);
void oracle_xhv_sparse(uint64_t T, const uint64_t *ptr, const uint32_t *idx,
                       const oreal *H, uint64_t k, const oreal *S,
                       const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                       double lambdaI, double lambdaAR, oreal *HS);
/* TRON-as-CG, rf_tron.h:134-254 + 412-505 with the parameter fold of trmf.cpp:603-606. */
void oracle_xsolve_sparse(uint64_t T, const uint64_t *ptr, const uint32_t *idx, const oreal *val,
                          const oreal *H, uint64_t k, oreal *W,
                          const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                          double lambdaI, double lambdaAR, int max_cg_iter, double eps_cg,
                          int threads, OracleXStats *stats);

/* Theta solve, trmf.cpp:455-484 (l2r_autoregressive_solver::solve). theta is |L| x k col-major. */
void oracle_theta_solve(uint64_t T, uint64_t k, const oreal *W, const uint32_t *lag_set,
                        uint32_t nlag, double lambdaLag, oreal *theta, int threads);

/* Global objective J in fp64 (SURVEY.md section 8(d) parity gate):
 * 0.5*sum_Omega (Y - w.h)^2 + 0.5*lambdaI*(|W|^2+|H|^2) + 0.5*lambdaAR*AR(W;Theta). */
double oracle_objective_sparse(uint64_t T, uint64_t n, const uint64_t *ptr, const uint32_t *idx,
                               const oreal *val, const oreal *W, const oreal *H, uint64_t k,
                               const uint32_t *lag_set, uint32_t nlag, const oreal *theta,
                               double lambdaI, double lambdaAR);

/* Whole driver with the reference's signature (trmf.h:203-210 / trmf.cpp:696-725). */
void oracle_trmf_train(const OracleMatrix *Y, const uint32_t *lag_set, uint32_t lag_size,
                       OracleMatrix *W, OracleMatrix *H, OracleMatrix *lag_val, int warm_start,
                       double lambdaI, double lambdaAR, double lambdaLag,
                       int32_t max_iter, int32_t period_W, int32_t period_H, int32_t period_Lag,
                       int32_t threads, int32_t missing, int32_t verbose);
/* Same, additionally filling log[0..max_iter-1] (may be NULL). */
void oracle_trmf_train_log(const OracleMatrix *Y, const uint32_t *lag_set, uint32_t lag_size,
                           OracleMatrix *W, OracleMatrix *H, OracleMatrix *lag_val, int warm_start,
                           double lambdaI, double lambdaAR, double lambdaLag,
                           int32_t max_iter, int32_t period_W, int32_t period_H, int32_t period_Lag,
                           int32_t threads, int32_t missing, int32_t verbose, OracleIterLog *log);

#ifdef __cplusplus
}
#endif
#endif
