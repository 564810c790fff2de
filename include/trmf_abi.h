/*
 * trmf_abi.h -- C ABI of the MI355X-native TRMF ALS solver (drop-in for rofuyu/exp-trmf-nips16).
 *
 * Two shared objects export this interface, one per element type, found by the reference's own
 * loader glob (python/trmf/trmf.py:21-22,77-80; python/trmf/rf_util.py:19-32):
 *
 *     <pkg>/trmf/corelib/trmf_float32.so      (TRMF_REAL = float)
 *     <pkg>/trmf/corelib/trmf_float64.so      (TRMF_REAL = double)
 *
 * Section 1 is the reference's boundary, byte-for-byte (every declaration cites the reference
 * file:line it replaces).  Section 2 is build-owned (no reference counterpart): device control,
 * a resident-session API so that callers can keep Y and the factors in HBM across calls
 * (SURVEY.md section 8(f) rank 4, and what bench.py times), and the multi-GPU bootstrap.
 *
 * Everything is plain C: pointers, sizes, PODs.  No torch / C++ types cross this boundary.
 * There is no CPU fallback behind it: if no HIP device is usable the entry points print a
 * diagnostic on stderr and leave the outputs untouched (c_trmf_train is `void`, like the
 * reference's) or return a negative error code (section 2).
 */
#ifndef TRMF_ABI_H
#define TRMF_ABI_H

#include <stddef.h>
#include <stdint.h>

/* Only the entry points below are exported; everything else in the libraries has hidden
 * visibility so that trmf_float32.so and trmf_float64.so can live in one process. */
#if defined(__GNUC__) || defined(__clang__)
#define TRMF_API __attribute__((visibility("default")))
#else
#define TRMF_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Section 1 -- the reference boundary
 * ---------------------------------------------------------------------------------------------- */

/* Matrix type tags: reference rf_matrix.h:3400-3405. */
enum {
    TRMF_DENSE_ROWMAJOR = 1,
    TRMF_DENSE_COLMAJOR = 2,
    TRMF_SPARSE = 3,
    TRMF_EYE = 4
};

/* PyMatrix: reference rf_matrix.h:3407-3415 (C side) and rf_util.py:35-52 (ctypes mirror).
 * Natural alignment, sizeof == 80.  Non-owning views on caller (NumPy) buffers.
 *   dense : only `val` (rows*cols elements, row- or column-major per `type`)
 *   sparse: CSR = (row_ptr[rows+1], col_idx[nnz], val_t[nnz]);
 *           CSC = (col_ptr[cols+1], row_idx[nnz], val[nnz]); both always present.        */
typedef struct {
    uint64_t rows, cols, nnz;
    size_t *row_ptr;
    size_t *col_ptr;
    uint32_t *row_idx;
    uint32_t *col_idx;
    void *val;
    void *val_t;
    int32_t type;
} PyMatrix;

/* c_trmf_train: reference trmf.h:203-210, trmf.cpp:696-725; ctypes prototype trmf.py:23-43.
 *
 *   Y        T x n observations (rows = timestamps). SPARSE required when missing != 0.
 *   lag_set  ascending uint32[lag_size]
 *   W        T x k DENSE_ROWMAJOR   (temporal factor, in/out)
 *   H        n x k DENSE_ROWMAJOR   (item factor, in/out)
 *   lag_val  lag_size x k DENSE_COLMAJOR (AR weights Theta, in/out)
 *
 * One ALS iteration = H(F)-solve, W(X)-solve, Theta-solve every period_Lag iterations
 * (trmf.cpp:647-693).  Outputs are written in place; Y and lag_set are never written.
 * Dimension/layout violations print the reference's "[ERR MSG]" lines on stderr and return
 * without touching the outputs (trmf.cpp:561-596,632-634).  So does every later failure (device
 * error, out of memory): the three factors are staged on the host and committed together or not at all.  warm_start == 0 reproduces the
 * reference's behaviour: the reference rebuilds W, H and lag_val as
 * PRIVATE random matrices of matching shapes before its dimension check (trmf.cpp:547-558:
 * std::mt19937 seeded 0, uniform / normal draws in doubles cast to the element type), trains
 * those -- the ">> iter" lines under verbose describe that run -- and discards them: the
 * caller's arrays are not updated and no "[ERR MSG]" about the caller's shapes can appear.
 * Here the same generator calls give the same private starting point, the training runs on
 * the device, and nothing is written back (tests/test_abi.py replays a capture of the
 * reference's lines).  `threads` is accepted
 * and ignored (no OpenMP on the device path).
 *
 * Checks beyond the reference's (each prints one "[ERR MSG]: ..." line and returns like a
 * dimension error; the reference aborts on an assert, reads out of bounds or has no such limit):
 *   - missing != 0 with a dense Y                  (reference: assert, rf_matrix.h:180)
 *   - lag_set not ascending, max lag >= rows of Y  (reference: out-of-bounds reads)
 *   - rank k outside 1..1024, more than 1024 lags (ranks up to 64 and lag sets whose |L| x |L| systems fit LDS run the
 *     register-tiled kernels; beyond that generic kernels compute the same thing slowly -- csrc/generic_kernels.hpp --
 *     where rounds 1-3 answered "[ERR MSG]")
 *   - a SPARSE Y with nnz >= 2^32; Y with >= 2^24 - 1 rows or columns, or a factor table
 *     (rows+1) x 16*ceil(k/16) elements above 4 GiB     (32-bit device offsets)
 *   - a lag reach / lag count whose LDS tiles exceed 160 KB per workgroup (session creation)
 *
 * Deliberate differences inside the X-solve (documented in DESIGN.md section 1): the TRON loop of
 * the reference (rf_tron.h:134-254, folded to ONE outer iteration by trmf.cpp:603-606) retries a
 * rejected step without changing anything (quirk Q3); here a rejected step leaves W unchanged
 * and the iteration goes on.  The acceptance test uses the exact quadratic identity
 * f(w+s) - f(w) = g.s + s.Hs/2 instead of a second pass over the observations, so `actred`
 * agrees with the reference's to rounding, not bit for bit.
 * verbose >= 1 prints the reference's parameter dump on stdout and the per-half-step
 * ">> iter i F|X|LV v" lines on stderr; verbose >= 2 adds the TRON line on stdout.          */
TRMF_API void c_trmf_train(const PyMatrix *pyY, uint32_t *py_lag_set, uint32_t py_lag_size,
                  PyMatrix *pyW, PyMatrix *pyH, PyMatrix *pylag_val, int warm_start,
                  double lambdaI, double lambdaAR, double lambdaLag,
                  int32_t max_iter, int32_t period_W, int32_t period_H, int32_t period_Lag,
                  int32_t threads, int32_t missing, int32_t verbose);

/* ------------------------------------------------------------------------------------------------
 * Section 2 -- build-owned additions (no reference counterpart)
 * ---------------------------------------------------------------------------------------------- */

/* Where the wall time of the LAST completed c_trmf_train call of this process went (round 5).  The reference's entry wraps
 * the caller's buffers zero-copy (trmf.cpp:696-725); this one has to put the problem into HBM and bring the factors back:
 *   setup_s     validation, device memory (a process-level pool: device_mallocs counts the hipMalloc calls of THIS call, 0
 *               once the pool holds the shape), the caller's arrays through a pinned ring (upload_s of it: the time the host
 *               spent reading them; bytes_h2d), padding / narrowing on the device, events, the set-up's one synchronisation
 *   compute_s   max_iter ALS iterations enqueued + the final synchronisation
 *   download_s  factors -> pinned staging -> the caller's arrays, committed only when ALL of them have arrived
 *   teardown_s  releasing the session (buffers back to the pool, no hipFree)
 * total_s is measured around the whole call; the four parts add up to it. */
typedef struct {
    double total_s, setup_s, upload_s, compute_s, download_s, teardown_s;
    double bytes_h2d, bytes_d2h;
    int32_t iters, device_mallocs, pool_reused, failed;
} TrmfTrainProfile;
/* 0 and *out filled when a c_trmf_train call has completed in this process (and library), -1 otherwise. */
TRMF_API int32_t trmf_last_train_profile(TrmfTrainProfile *out);
/* Give back what the library keeps between calls on the selected device: pooled device memory (when no session is alive),
 * idle streams, the pinned upload ring (24 MB) and the pinned download staging, the idle worker threads / communicators of the TRMF_DEVICES mode.  TRMF_POOL_MAX_MB (default 8192, 0 = keep
 * nothing) bounds the pool.  Returns 0, or -1 when the selected device cannot be made current (trmf_last_error() says why). */
TRMF_API int32_t trmf_release_cached(void);

/* sizeof(element type) of this library: 4 or 8. */
TRMF_API int32_t trmf_sizeof_real(void);
/* Number of visible HIP devices (0 if none / runtime unusable). */
TRMF_API int32_t trmf_device_count(void);
/* Select the HIP device used by subsequent calls from this process (default 0). 0 on success. */
TRMF_API int32_t trmf_set_device(int32_t device);
/* Free bytes of HBM on the selected device right now (-1 if no device): sizing aid, and what the leak test reads. */
TRMF_API int64_t trmf_device_free_bytes(void);
/* Last error text of section-2 calls (static storage, never NULL). */
TRMF_API const char *trmf_last_error(void);

/* Per-iteration record filled by a session (mirrors the reference's verbose output:
 * trmf.cpp:661,672,687 and the TRON line rf_tron.h:219). Values are -1 when a phase did not run. */
typedef struct {
    double normF, normX, normLV;           /* ||H||^2, ||W||^2, ||Theta||^2 after each phase   */
    double f, fnew, actred, prered;        /* X-subproblem objective before/after, reductions  */
    double gnorm, cg_rnorm;
    int32_t cg_iter, accepted;
    float ms_F, ms_X, ms_LV;               /* HIP-event time of each phase on the solver stream; all ms_* are -1 for an
                                              iteration that carried no events (trmf_session_set_timing)         */
    float ms_F_kernel;                     /* HIP-event time of the F-solve kernel alone       */
    double delta;                          /* trust-region bound of the TRON line (rf_tron.h:195-219) */
    double cg_rnorm_direct;                /* |-g - H s| of the step evaluated directly; cg_rnorm is the CG's recurrence
                                              (-1 where not evaluated: only the one-GPU persistent kernel does)   */
    float ms_X_gram;                       /* HIP-event time from the start of the X phase to the end of its Gram build
                                              (gram_x_kernel + gathers); ms_X - ms_X_gram is the CG solve              */
    float reserved_;
} TrmfIterStats;

typedef struct TrmfSession TrmfSession;

/* Upload Y (both orientations), lag_set and the initial factors to HBM and build all device
 * scratch.  Same argument meaning and validation as c_trmf_train.  Returns NULL on failure
 * (see trmf_last_error()).  When a communicator is active (trmf_dist_init*), every rank must
 * call this with identical inputs; rows are partitioned inside.                               */
TRMF_API TrmfSession *trmf_session_create(const PyMatrix *Y, const uint32_t *lag_set, uint32_t lag_size,
                                 const PyMatrix *W, const PyMatrix *H, const PyMatrix *lag_val,
                                 double lambdaI, double lambdaAR, double lambdaLag,
                                 int32_t period_W, int32_t period_H, int32_t period_Lag,
                                 int32_t missing, int32_t verbose);
/* Enqueue `iters` further ALS iterations (iteration numbering continues across calls, so the
 * period_* gating matches one long c_trmf_train run).  Asynchronous unless verbose > 0.        */
TRMF_API int32_t trmf_session_run(TrmfSession *s, int32_t iters);
/* Switch the ||H||^2, ||W||^2, ||Theta||^2 records of TrmfIterStats on (default) or off.  The
 * reference evaluates these norms only under `verbose` (trmf.cpp:659-688); with the switch off the
 * fields read -1 and an iteration is about 30 us shorter.  c_trmf_train follows `verbose`.        */
TRMF_API int32_t trmf_session_log_norms(TrmfSession *s, int32_t on);
/* Which iterations carry the HIP events behind TrmfIterStats.ms_*: 1 = every iteration (default), N > 1 = the iterations whose
 * 1-based index is a multiple of N, 0 = none.  The seven event records of an iteration are barrier packets between kernels that
 * would otherwise be dispatched back to back: 21-26 us per iteration on an MI355X (2.5 % of a config-3 iteration, 22 % of a
 * config-2 one).  An iteration without events reports ms_* = -1; everything else of its record is unaffected.  c_trmf_train
 * records none.  No reference counterpart.                                                                                      */
TRMF_API int32_t trmf_session_set_timing(TrmfSession *s, int32_t period);
/* Append Ynew (Tn x n, NEW timestamps, same storage class -- sparse or dense -- as the session's Y) to the resident
 * problem: the rolling-window caller of the reference (python/trmf/trmf.py:303-329) retrains on a prefix that
 * grows by one window and warm-starts from the previous model (trmf.py:237-246).  Only the block (and, for a
 * sparse Y, one 4-byte pointer per item) is uploaded: CSR rows are appended, the CSC is rebuilt on the device,
 * a dense Y's second orientation is re-strided on the device, W gains Tn rows by the AR recursion with the
 * current lag weights (Model.latent_forecast, trmf.py:170-181); H and lag_val are kept.  The iteration counter
 * restarts at 1 like a fresh c_trmf_train call (period_* gating, trmf.cpp:647).  Blocking.  With a communicator
 * active every rank must call it with the same block. */
TRMF_API int32_t trmf_session_append_rows(TrmfSession *s, const PyMatrix *Ynew);
/* Train on a[i] * y + b[i] of series (column) i instead of the raw dense Y the session holds (missing == 0 only):
 * the per-window NormalizedTransform of the reference's rolling_validate(transform=True) (python/trmf/trmf.py:82-96,
 * 237-249, 253-257), which rescales every entry of the growing prefix.  The session keeps the raw matrix (rows passed
 * to create / append_rows are always RAW) and re-derives both training orientations on the device.  The
 * coefficients are arrays of n values of THIS library's element type (they are fitted from Y and have its dtype);
 * y * a + b is evaluated in that type with the product and the sum rounded separately, as NumPy evaluates it; only
 * the 2n coefficients are uploaded.  a == NULL / b == NULL mean 1 / 0.  May be called again at any time (e.g. after
 * append_rows, with the coefficients refitted on the grown prefix). */
TRMF_API int32_t trmf_session_set_series_transform(TrmfSession *s, const void *a /* n */, const void *b /* n */);
/* Checkpoint of the session's (W, H, lag_val, iteration counter) ON THE DEVICE, and the way back to it: after rewind the
 * session continues exactly as it did after mark (the solver is deterministic).  One mark per session (a new one replaces the
 * old); append_rows invalidates it.  Both block.  No reference counterpart (bench.py repeats its timed window with these). */
TRMF_API int32_t trmf_session_mark(TrmfSession *s);
TRMF_API int32_t trmf_session_rewind(TrmfSession *s);
/* Number of timestamps currently held by the session (rows of W). */
TRMF_API int32_t trmf_session_rows(TrmfSession *s);
/* Block until all enqueued work of the session has finished. */
TRMF_API int32_t trmf_session_sync(TrmfSession *s);
/* Copy the current factors back into caller PyMatrix views (same shapes as at create). */
TRMF_API int32_t trmf_session_download(TrmfSession *s, PyMatrix *W, PyMatrix *H, PyMatrix *lag_val);
/* Copy the stats of the last min(cap, iterations-run) iterations, oldest first. Returns count. */
TRMF_API int32_t trmf_session_stats(TrmfSession *s, TrmfIterStats *out, int32_t cap);
/* Global objective J of the current factors, evaluated on device in fp64. */
TRMF_API double trmf_session_objective(TrmfSession *s);
/* Algorithmic bytes of ONE F-solve launch on this rank: nnz*(4+s+k*s) + (rows+1)*8 + rows*k*s
 * over the item rows this rank owns. */
TRMF_API double trmf_session_fsolve_bytes(TrmfSession *s);
/* One line of text: what the session runs -- with several ranks, which phases are sharded, the form of the X-solve chosen
 * by the measure-once rule (replicated / time-sharded through the communicator / time-sharded peer to peer) with the
 * slowest rank's measured X-phase time of every candidate, and whether the peer-to-peer transport is available (and why
 * not).  Writes at most cap bytes including the terminating NUL; returns the length of the full text. */
TRMF_API int32_t trmf_session_describe(TrmfSession *s, char *buf, int32_t cap);
TRMF_API void trmf_session_destroy(TrmfSession *s);

/* --- multi-GPU (one process per GPU; RCCL all-gathers over xGMI) ---------------------------- */
#define TRMF_UNIQUE_ID_BYTES 128
/* Rank 0: create an RCCL unique id; the caller broadcasts the bytes to all ranks
 * (e.g. with torch.distributed) and every rank passes them to trmf_dist_init. */
TRMF_API int32_t trmf_dist_get_unique_id(void *out_id /* TRMF_UNIQUE_ID_BYTES */);
/* world <= 64 (slots of the staged all-gather); larger worlds are refused here. */
TRMF_API int32_t trmf_dist_init(int32_t rank, int32_t world, const void *id /* TRMF_UNIQUE_ID_BYTES */);
/* Host-staged communicator for tests and RCCL-less setups: `allgatherv` must gather, in place,
 * the byte ranges [offsets[r], offsets[r+1]) of `buf` owned by each rank r. Returns 0 on success. */
typedef int32_t (*trmf_allgatherv_fn)(void *buf, const uint64_t *offsets, int32_t world,
                                      void *ctx);
TRMF_API int32_t trmf_dist_init_callback(int32_t rank, int32_t world, trmf_allgatherv_fn allgatherv,
                                void *ctx);
/* Measurement aid: act as rank `rank` of `world` WITHOUT peers (every gather is skipped).  A session created under it
 * runs exactly the kernels that rank would run, alone on the GPU -- its times are that rank's compute share; the factors
 * are NOT a solution (the other ranks' blocks stay stale).  Used by scripts/shard_compute_times.py. */
TRMF_API int32_t trmf_dist_init_solo(int32_t rank, int32_t world);
TRMF_API int32_t trmf_dist_rank(void);
TRMF_API int32_t trmf_dist_world(void);
TRMF_API void trmf_dist_finalize(void);
/* Contiguous row partition balanced by nnz: bounds[0]=0 <= ... <= bounds[world]=nrows. */
TRMF_API int32_t trmf_partition_by_nnz(uint64_t nrows, const size_t *ptr, int32_t world,
                              uint64_t *bounds /* world+1 */);

#ifdef __cplusplus
}
#endif
#endif /* TRMF_ABI_H */
