// trmf_abi.hip -- extern "C" entry points of trmf_float{32,64}.so (see include/trmf_abi.h).
//
// Section 1: c_trmf_train, the reference's own boundary (trmf.h:203-210, trmf.cpp:696-725).
// Section 2: build-owned session / device / multi-GPU entry points.
//
// There is deliberately no host implementation of the solver behind these symbols: if the HIP
// runtime reports no usable device the calls fail loudly.
#include "../../include/trmf_abi.h"

#include <memory>
#include <mutex>
#include <random>

#include "session_group.hpp"

namespace trmf {

static std::string g_last_error = "";
static std::mutex g_err_mu;
void set_error(const std::string &msg) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_last_error = msg;
}

static int g_device = 0;
static std::mutex g_prof_mu;
static TrmfTrainProfile g_last_profile;
static bool g_have_profile = false;
static std::shared_ptr<Comm> g_self = std::make_shared<SelfComm>();
static std::shared_ptr<Comm> g_comm;
// A session shares ownership of the communicator it was created under, so trmf_dist_finalize() before
// trmf_session_destroy() leaves the session's communicator alive until the session goes away.
// (a worker thread of an in-process session group has its own communicator and device: session_group.hpp)
std::shared_ptr<Comm> active_comm() { return tl_comm() ? tl_comm() : g_comm ? g_comm : g_self; }
std::string SessionGroup::trmf_last_error_text() { std::lock_guard<std::mutex> lk(g_err_mu); return g_last_error; }

static bool bind_device() {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        set_error("no HIP device visible (the MI355X TRMF solver has no CPU fallback)");
        return false;
    }
    if (hipSetDevice(tl_device() >= 0 ? tl_device() : g_device) != hipSuccess) {
        set_error("hipSetDevice failed");
        return false;
    }
    return true;
}
// Every entry point that touches the device selects the library's device (trmf_set_device) for its duration and
// puts the caller's current device back on return.
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    DeviceGuard() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = bind_device();
    }
    ~DeviceGuard() { if (prev >= 0 && prev != g_device) (void)hipSetDevice(prev); }
};

// check_dimension, trmf.cpp:561-596 (same messages, same order)
static bool check_dimension(const PyMatrix *Y, const PyMatrix *W, const PyMatrix *H, const PyMatrix *LV,
                            uint32_t lag_size) {
    bool pass = true;
    if (Y->rows != W->rows) { fprintf(stderr, "[ERR MSG]: Y.rows (%ld) != W.rows (%ld)\n", (long)Y->rows, (long)W->rows); pass = false; }
    if (Y->cols != H->rows) { fprintf(stderr, "[ERR MSG]: Y.cols (%ld) != H.rows (%ld)\n", (long)Y->cols, (long)H->rows); pass = false; }
    if (W->cols != H->cols) { fprintf(stderr, "[ERR MSG]: W.cols (%ld) != H.cols (%ld)\n", (long)W->cols, (long)H->cols); pass = false; }
    if (lag_size != LV->rows) { fprintf(stderr, "[ERR MSG]: lag_set.size(%ld) != lag_val.rows(%ld)\n", (long)lag_size, (long)LV->rows); pass = false; }
    if (W->cols != LV->cols) { fprintf(stderr, "[ERR MSG]: W.cols(%ld) != lag_val.cols(%ld)\n", (long)W->cols, (long)LV->cols); pass = false; }
    if (W->type != TRMF_DENSE_ROWMAJOR) { fprintf(stderr, "[ERR MSG]: W should be rowmajored\n"); pass = false; }
    if (H->type != TRMF_DENSE_ROWMAJOR) { fprintf(stderr, "[ERR MSG]: H should be rowmajored\n"); pass = false; }
    if (LV->type != TRMF_DENSE_COLMAJOR) { fprintf(stderr, "[ERR MSG]: lag_val should be colmajored\n"); pass = false; }
    return pass;
}

// Limits of the device path (documented in DESIGN.md); violations are reported, never worked around.
static bool check_device_limits(const PyMatrix *Y, const PyMatrix *W, uint32_t lag_size, int missing) {
    bool pass = true;
    // the reference asserts (aborts) on a dense Y with missing != 0 (rf_matrix.h:180); here: diagnostic
    if (missing && Y->type != TRMF_SPARSE) { fprintf(stderr, "[ERR MSG]: missing!=0 requires a sparse Y\n"); pass = false; }
    if (Y->type != TRMF_SPARSE && Y->type != TRMF_DENSE_ROWMAJOR && Y->type != TRMF_DENSE_COLMAJOR) { fprintf(stderr, "[ERR MSG]: unsupported Y matrix type %d\n", (int)Y->type); pass = false; }
    // register-tiled kernels up to rank 64, generic kernels (csrc/generic_kernels.hpp) up to 1024, on both training paths
    if (W->cols < 1 || W->cols > (uint64_t)kMaxRankGeneric) { fprintf(stderr, "[ERR MSG]: rank k=%ld outside the supported range 1..%d\n", (long)W->cols, kMaxRankGeneric); pass = false; }
    if (lag_size > (uint32_t)kMaxLags) { fprintf(stderr, "[ERR MSG]: |lag_set|=%u exceeds the supported %d\n", lag_size, kMaxLags); pass = false; }
    if ((Y->type == TRMF_SPARSE && Y->nnz >= (1ull << 32)) || Y->rows >= (1ull << 31) || Y->cols >= (1ull << 31)) { fprintf(stderr, "[ERR MSG]: problem exceeds 32-bit device indices\n"); pass = false; }
    // gathered factor rows are addressed with 32-bit byte offsets (gram_ring): tables up to 4 GiB
    const uint64_t rowbytes = (uint64_t)padded_rank((int)W->cols) * sizeof(real);
    if ((Y->rows + 1) * rowbytes > 0xffffffffull || (Y->cols + 1) * rowbytes > 0xffffffffull ||
        Y->rows + 1 >= (1ull << 24) || Y->cols + 1 >= (1ull << 24)) {
        fprintf(stderr, "[ERR MSG]: factor tables exceed the 4 GiB / 2^24 rows of 32-bit gather offsets\n"); pass = false;
    }
    return pass;
}

// The reference's dimension check plus this build's own limits (include/trmf_abi.h lists them); diagnostics on stderr.
static bool validate_problem(const PyMatrix *Y, const uint32_t *lag_set, uint32_t lag_size, const PyMatrix *W,
                             const PyMatrix *H, const PyMatrix *LV, int32_t missing) {
    if (!check_dimension(Y, W, H, LV, lag_size)) { set_error("dimension check failed"); return false; }
    if (!check_device_limits(Y, W, lag_size, missing)) { set_error("unsupported problem"); return false; }
    for (uint32_t l = 1; l < lag_size; l++)
        if (lag_set[l] < lag_set[l - 1]) { fprintf(stderr, "[ERR MSG]: lag_set must be ascending\n"); set_error("lag_set not ascending"); return false; }
    if (lag_size && lag_set[lag_size - 1] >= Y->rows) { fprintf(stderr, "[ERR MSG]: max lag >= number of timestamps\n"); set_error("lag too large"); return false; }
    return true;
}

// one session on the calling thread's device under the calling thread's communicator (the problem has been validated)
static TrmfSessionImpl *build_session(const PyMatrix *Y, const uint32_t *lag_set, uint32_t lag_size,
                                      const PyMatrix *W, const PyMatrix *H, const PyMatrix *LV,
                                      double lambdaI, double lambdaAR, double lambdaLag, int32_t period_W,
                                      int32_t period_H, int32_t period_Lag, int32_t missing, int32_t verbose) {
    if (!bind_device()) return nullptr;
    std::unique_ptr<TrmfSessionImpl> s(new TrmfSessionImpl());
    s->lambdaI = lambdaI; s->lambdaAR = lambdaAR; s->lambdaLag = lambdaLag;
    s->period_W = period_W; s->period_H = period_H; s->period_Lag = period_Lag; s->verbose = verbose;
    s->full = (missing == 0);
    if (s->create(Y, lag_set, lag_size, W, H, LV)) {
        const std::string why = SessionGroup::trmf_last_error_text();
        active_comm()->abort();          // before the half-built session is torn down (its teardown passes the group's barrier)
        set_error(why);
        return nullptr;
    }
    return s.release();
}

// What a TrmfSession* points to: ONE session on the library's device -- or, under TRMF_DEVICES / TRMF_GPUS (session_group.hpp), one
// session per listed device, each on a worker thread of this process, joined by an in-process communicator.
struct SessionHandle {
    TrmfSessionImpl *single = nullptr;
    SessionGroup *group = nullptr;
    TrmfSessionImpl *first() const { return group ? group->impl[0] : single; }
    // f on every rank's session (one rank: on the calling thread)
    int all(const std::function<int(TrmfSessionImpl *)> &f) const {
        return group ? group->on_all([&](int r) { return f(group->impl[r]); }) : f(single);
    }
    int rank0(const std::function<int(TrmfSessionImpl *)> &f) const {
        return group ? group->on_rank0([&](int) { return f(group->impl[0]); }) : f(single);
    }
    ~SessionHandle() {
        if (group) SessionGroup::release(group);      // the sessions go; threads + communicators wait for the next call (session_group.hpp)
        else if (single) { (void)single->sync(false); delete single; }
    }
};

static SessionHandle *make_session(const PyMatrix *Y, const uint32_t *lag_set, uint32_t lag_size,
                                   const PyMatrix *W, const PyMatrix *H, const PyMatrix *LV,
                                   double lambdaI, double lambdaAR, double lambdaLag, int32_t period_W,
                                   int32_t period_H, int32_t period_Lag, int32_t missing, int32_t verbose) {
    if (!validate_problem(Y, lag_set, lag_size, W, H, LV, missing)) return nullptr;
    if (!bind_device()) { fprintf(stderr, "[ERR MSG]: %s\n", trmf_last_error()); return nullptr; }
    std::unique_ptr<SessionHandle> h(new SessionHandle());
    std::string why;
    const std::vector<int> devs = (g_comm || tl_comm()) ? std::vector<int>() : inproc_devices(&why);
    if (!why.empty()) { set_error(why); fprintf(stderr, "[ERR MSG]: %s\n", why.c_str()); return nullptr; }
    if (devs.empty()) {
        h->single = build_session(Y, lag_set, lag_size, W, H, LV, lambdaI, lambdaAR, lambdaLag, period_W, period_H, period_Lag, missing, verbose);
        if (!h->single) { fprintf(stderr, "[ERR MSG]: %s\n", trmf_last_error()); return nullptr; }
        return h.release();
    }
    h->group = SessionGroup::acquire(devs);
    if (!h->group) { fprintf(stderr, "[ERR MSG]: %s\n", trmf_last_error()); return nullptr; }
    int rc = 0;
    if (rc == 0)
        rc = h->group->on_all([&](int r) {          // every rank uploads the whole problem to its device and builds its session
            h->group->impl[r] = build_session(Y, lag_set, lag_size, W, H, LV, lambdaI, lambdaAR, lambdaLag, period_W, period_H, period_Lag, missing,
                                              r == 0 ? verbose : 0);        // the reference's log lines once
            return h->group->impl[r] ? 0 : kFail;
        });
    if (rc) { fprintf(stderr, "[ERR MSG]: %s\n", trmf_last_error()); return nullptr; }    // (~SessionHandle tears the group down)
    if (verbose) fprintf(stderr, ">> %d ranks as threads of this process on devices TRMF_DEVICES; communicator: %s\n", h->group->world(), h->group->comm_kind.c_str());
    return h.release();
}

}  // namespace trmf

using namespace trmf;

extern "C" {

struct TrmfSession { SessionHandle h; };        // opaque to callers; never instantiated as such

// ------------------------------------------------------------------------------------------------
// Section 1
// ------------------------------------------------------------------------------------------------
void c_trmf_train(const PyMatrix *pyY, uint32_t *py_lag_set, uint32_t py_lag_size, PyMatrix *pyW,
                  PyMatrix *pyH, PyMatrix *pylag_val, int warm_start, double lambdaI, double lambdaAR,
                  double lambdaLag, int32_t max_iter, int32_t period_W, int32_t period_H,
                  int32_t period_Lag, int32_t threads, int32_t missing, int32_t verbose) {
    if (verbose > 0) {   // parameter dump, trmf.cpp:607-629 (after the fold of :603-606)
        fprintf(stdout, ">> param.solver_type %d\n", missing ? 31 : 30);
        fprintf(stdout, ">> param.max_iter %d\n", max_iter);
        fprintf(stdout, ">> param.lambdaI %g\n", lambdaI);
        fprintf(stdout, ">> param.lambdaAR %g\n", lambdaAR);
        fprintf(stdout, ">> param.lambdaLag %g\n", lambdaLag);
        fprintf(stdout, ">> param.period_W %d\n", period_W);
        fprintf(stdout, ">> param.period_H %d\n", period_H);
        fprintf(stdout, ">> param.period_Lag %d\n", period_Lag);
        fprintf(stdout, ">> param.threads %d\n", threads);
        fprintf(stdout, ">> param.verbose %d\n", verbose);
        fprintf(stdout, ">> param.eps %g\n", 0.1);
        fprintf(stdout, ">> param.eps_cg %g\n", 0.1);
        fprintf(stdout, ">> param.max_tron_iter %d\n", 1);
        fprintf(stdout, ">> param.max_cg_iter %d\n", 20);
        fprintf(stdout, ">> prob.lag_size %ld:  ", (long)py_lag_size);
        for (uint32_t i = 0; i < py_lag_size; i++) fprintf(stdout, " %d", (int)py_lag_set[i]);
        fprintf(stdout, "\n");
        fflush(stdout);
    }
    // Quirk Q1: with warm_start == 0 the reference rebuilds W, H and lag_val as PRIVATE random matrices of
    // matching shapes before its dimension check (trmf_initialization, trmf.cpp:547-558, 719-722: rng_t = std::mt19937 seeded 0,
    // a fresh uniform_real_distribution<double>(0,1) / normal_distribution<double>(0,1) per element, cast to val_type,
    // rf_matrix.h:56-68, 740-753), trains those -- its ">> iter" lines under verbose describe that run -- and discards them:
    // the caller's arrays come back unchanged and no "[ERR MSG]" about the caller's shapes can appear.  Same here: the same
    // generator calls in the same order produce the same starting point (libstdc++ on both sides), the training runs on the
    // device on private copies, and nothing is written back.
    std::vector<real> cold_W, cold_H, cold_LV;
    PyMatrix privW, privH, privLV;
    if (!warm_start) {
        const size_t m = pyY->rows, n = pyY->cols, k = pyW->cols;
        std::mt19937 rng(0);
        cold_W.resize(m * k); cold_H.resize(n * k); cold_LV.resize((size_t)py_lag_size * k);
        for (real &x : cold_W) x = (real)std::uniform_real_distribution<double>(0.0, 1.0)(rng);
        for (real &x : cold_H) x = (real)std::uniform_real_distribution<double>(0.0, 1.0)(rng);
        for (real &x : cold_LV) x = (real)std::normal_distribution<double>(0.0, 1.0)(rng);
        auto view = [](std::vector<real> &buf, size_t rows, size_t cols, int32_t type) {
            PyMatrix v{};
            v.rows = rows; v.cols = cols; v.nnz = rows * cols; v.val = buf.data(); v.type = type;
            return v;
        };
        privW = view(cold_W, m, k, TRMF_DENSE_ROWMAJOR);
        privH = view(cold_H, n, k, TRMF_DENSE_ROWMAJOR);
        privLV = view(cold_LV, py_lag_size, k, TRMF_DENSE_COLMAJOR);
        pyW = &privW; pyH = &privH; pylag_val = &privLV;       // from here on the call works on the private model
    }
    DeviceGuard guard;                               // device of this library for the call, the caller's afterwards
    // where the call's wall time goes (trmf_last_train_profile): set-up / compute / download / teardown
    const double t0 = TrmfSessionImpl::now_s();
    TrmfTrainProfile prof{};
    const DevicePool::Stats ps0 = guard.ok ? DevicePool::current().stats() : DevicePool::Stats{};
    SessionHandle *s = make_session(pyY, py_lag_set, py_lag_size, pyW, pyH, pylag_val, lambdaI, lambdaAR,
                                    lambdaLag, period_W, period_H, period_Lag, missing, verbose);
    if (!s) {                                        // diagnostics already on stderr; outputs untouched
        prof.failed = 1; prof.total_s = prof.setup_s = TrmfSessionImpl::now_s() - t0;
        std::lock_guard<std::mutex> lk(g_prof_mu); g_last_profile = prof; g_have_profile = true;
        return;
    }
    (void)s->all([&](TrmfSessionImpl *t) {
        t->log_norms = verbose > 0;                  // the norm lines exist only under verbose (trmf.cpp:659-688)
        t->ev_period = 0;                            // nobody reads per-phase times of a one-shot call (they cost 21-26 us per iteration)
        return 0;
    });
    const double t1 = TrmfSessionImpl::now_s();
    int rc = s->all([&](TrmfSessionImpl *t) { const int r = t->run(max_iter); return r ? r : t->sync(); });
    const double t2 = TrmfSessionImpl::now_s();
    // the factors come back through host staging and are committed together: a failure anywhere -- on ANY rank -- leaves W, H and
    // lag_val as the caller passed them (the reference's contract for a failed call, trmf.cpp:632-634).  Staged and committed on rank
    // 0's thread (the staging lease is a mutex: locked and released by one thread); every rank holds the same factors.
    size_t d2h = 0;
    if (rc == 0)
        rc = s->rank0([&](TrmfSessionImpl *t) {
            TrmfSessionImpl::StagedFactors staged;
            const int r = t->download_staged(staged);
            if (r == 0) { staged.commit(pyW->val, pyH->val, pylag_val->val); d2h = staged.bW + staged.bH + staged.bL; }
            return r;
        });
    if (rc) fprintf(stderr, "[ERR MSG]: device failure, outputs untouched: %s\n", trmf_last_error());
    const double t3 = TrmfSessionImpl::now_s();
    prof.upload_s = s->first()->t_upload_s; prof.bytes_h2d = s->first()->bytes_uploaded;
    prof.bytes_d2h = (double)d2h;
    delete s;
    const double t4 = TrmfSessionImpl::now_s();
    const DevicePool::Stats ps1 = DevicePool::current().stats();
    prof.total_s = t4 - t0; prof.setup_s = t1 - t0; prof.compute_s = t2 - t1; prof.download_s = t3 - t2; prof.teardown_s = t4 - t3;
    prof.iters = max_iter; prof.device_mallocs = (int32_t)(ps1.hip_mallocs - ps0.hip_mallocs); prof.pool_reused = (int32_t)(ps1.reused - ps0.reused);
    prof.failed = rc != 0;
    std::lock_guard<std::mutex> lk(g_prof_mu); g_last_profile = prof; g_have_profile = true;
}

// ------------------------------------------------------------------------------------------------
// Section 2
// ------------------------------------------------------------------------------------------------
int32_t trmf_sizeof_real(void) { return (int32_t)sizeof(real); }

int32_t trmf_last_train_profile(TrmfTrainProfile *out) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_have_profile || !out) return kFail;
    *out = g_last_profile;
    return 0;
}
int32_t trmf_release_cached(void) {
    SessionGroup::drop_idle();           // worker threads + communicators of the TRMF_DEVICES mode kept between calls
    DeviceGuard guard;
    if (!guard.ok) return kFail;
    DevicePool::current().trim();
    StreamCache::drop_idle();
    HostStager::current().release_staging();
    HostStager::current().release_ring();
    return 0;
}

int32_t trmf_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

int32_t trmf_set_device(int32_t device) {
    int cnt = trmf_device_count();
    if (device < 0 || device >= cnt) { set_error("device index out of range"); return kFail; }
    g_device = device;
    return hipSetDevice(device) == hipSuccess ? 0 : kFail;
}

int64_t trmf_device_free_bytes(void) {
    DeviceGuard guard;
    if (!guard.ok) return -1;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return -1;
    return (int64_t)free_b;
}

const char *trmf_last_error(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_err_mu);
    copy = g_last_error;
    return copy.c_str();
}

TrmfSession *trmf_session_create(const PyMatrix *Y, const uint32_t *lag_set, uint32_t lag_size,
                                 const PyMatrix *W, const PyMatrix *H, const PyMatrix *lag_val,
                                 double lambdaI, double lambdaAR, double lambdaLag, int32_t period_W,
                                 int32_t period_H, int32_t period_Lag, int32_t missing, int32_t verbose) {
    DeviceGuard guard;
    return reinterpret_cast<TrmfSession *>(make_session(Y, lag_set, lag_size, W, H, lag_val, lambdaI, lambdaAR,
                                                        lambdaLag, period_W, period_H, period_Lag, missing, verbose));
}
#define HND(s) reinterpret_cast<SessionHandle *>(s)

int32_t trmf_session_run(TrmfSession *s, int32_t iters) {
    if (!s) return kFail;
    DeviceGuard guard;
    return guard.ok ? HND(s)->all([&](TrmfSessionImpl *t) { return t->run(iters); }) : kFail;
}
int32_t trmf_session_log_norms(TrmfSession *s, int32_t on) {
    if (!s) return kFail;
    return HND(s)->all([&](TrmfSessionImpl *t) { t->log_norms = on != 0; return 0; });
}
int32_t trmf_session_set_timing(TrmfSession *s, int32_t period) {
    if (!s || period < 0) return kFail;
    return HND(s)->all([&](TrmfSessionImpl *t) { t->ev_period = period; return 0; });
}
int32_t trmf_session_sync(TrmfSession *s) {
    if (!s) return kFail;
    DeviceGuard guard;
    return guard.ok ? HND(s)->all([&](TrmfSessionImpl *t) { return t->sync(); }) : kFail;
}
int32_t trmf_session_mark(TrmfSession *s) {
    if (!s) return kFail;
    DeviceGuard guard;
    return guard.ok ? HND(s)->all([&](TrmfSessionImpl *t) { return t->mark(); }) : kFail;
}
int32_t trmf_session_rewind(TrmfSession *s) {
    if (!s) return kFail;
    DeviceGuard guard;
    return guard.ok ? HND(s)->all([&](TrmfSessionImpl *t) { return t->rewind(); }) : kFail;
}
int32_t trmf_session_append_rows(TrmfSession *s, const PyMatrix *Ynew) {
    if (!s || !Ynew) { set_error("null session or block"); return kFail; }
    DeviceGuard guard;
    return guard.ok ? HND(s)->all([&](TrmfSessionImpl *t) { return t->append_rows(Ynew); }) : kFail;
}
int32_t trmf_session_rows(TrmfSession *s) { return s ? HND(s)->first()->T : kFail; }
int32_t trmf_session_set_series_transform(TrmfSession *s, const void *a, const void *b) {
    if (!s) { set_error("null session"); return kFail; }
    DeviceGuard guard;
    return guard.ok ? HND(s)->all([&](TrmfSessionImpl *t) { return t->set_series_transform((const real *)a, (const real *)b); }) : kFail;
}

int32_t trmf_session_download(TrmfSession *s, PyMatrix *W, PyMatrix *H, PyMatrix *lag_val) {
    if (!s) return kFail;
    DeviceGuard guard;
    if (!guard.ok) return kFail;
    if (HND(s)->all([&](TrmfSessionImpl *t) { return t->sync(); })) return kFail;       // every rank holds the same factors: rank 0 answers
    return HND(s)->rank0([&](TrmfSessionImpl *t) -> int {
        if (W && (W->rows != (uint64_t)t->T || W->cols != (uint64_t)t->k || W->type != TRMF_DENSE_ROWMAJOR)) { set_error("W shape/layout mismatch"); return kFail; }
        if (H && (H->rows != (uint64_t)t->n || H->cols != (uint64_t)t->k || H->type != TRMF_DENSE_ROWMAJOR)) { set_error("H shape/layout mismatch"); return kFail; }
        if (lag_val && (lag_val->rows != (uint64_t)t->nlag || lag_val->cols != (uint64_t)t->k || lag_val->type != TRMF_DENSE_COLMAJOR)) { set_error("lag_val shape/layout mismatch"); return kFail; }
        if (W && t->download_padded(t->W, (real *)W->val, t->T)) return kFail;
        if (H && t->download_padded(t->H, (real *)H->val, t->n)) return kFail;
        if (lag_val && t->nlag)
            TRMF_HIP_CHECK(hipMemcpy(lag_val->val, t->theta.p, sizeof(real) * (size_t)t->nlag * t->k, hipMemcpyDeviceToHost));
        return 0;
    });
}

int32_t trmf_session_stats(TrmfSession *s, TrmfIterStats *out, int32_t cap) {
    if (!(s && out && cap > 0)) return 0;
    DeviceGuard guard;
    if (!guard.ok) return 0;
    if (HND(s)->group && HND(s)->all([&](TrmfSessionImpl *t) { return t->sync(); })) return 0;
    int cnt = 0;
    (void)HND(s)->rank0([&](TrmfSessionImpl *t) { cnt = t->stats(out, cap); return cnt < 0 ? kFail : 0; });
    return cnt < 0 ? 0 : cnt;
}
double trmf_session_objective(TrmfSession *s) {
    if (!s) return NAN;
    DeviceGuard guard;
    if (!guard.ok) return NAN;
    double J = NAN;
    (void)HND(s)->all([&](TrmfSessionImpl *t) { const double v = t->objective(); if (t->comm->rank == 0) J = v; return 0; });
    return J;
}
double trmf_session_fsolve_bytes(TrmfSession *s) { return s ? HND(s)->first()->fsolve_bytes() : 0.0; }
int32_t trmf_session_describe(TrmfSession *s, char *buf, int32_t cap) {
    if (!s) return kFail;
    std::string d = HND(s)->first()->describe();
    if (HND(s)->group) d += "; ranks are threads of this process (TRMF_DEVICES), communicator: " + HND(s)->group->comm_kind;
    if (buf && cap > 0) { std::strncpy(buf, d.c_str(), (size_t)cap - 1); buf[cap - 1] = 0; }
    return (int32_t)d.size();
}
void trmf_session_destroy(TrmfSession *s) {
    if (!s) return;
    DeviceGuard guard;
    delete HND(s);
}

// ---- multi-GPU ----------------------------------------------------------------------------------
int32_t trmf_dist_get_unique_id(void *out_id) {
    RcclApi &api = rccl_api();
    if (!api.load()) return kFail;
    RcclApi::UniqueId id;
    const int rc = api.GetUniqueId(&id);
    if (rc != 0) { set_error(std::string("ncclGetUniqueId: ") + api.GetErrorString(rc)); return kFail; }
    std::memcpy(out_id, &id, TRMF_UNIQUE_ID_BYTES);
    return 0;
}

int32_t trmf_dist_init(int32_t rank, int32_t world, const void *id_bytes) {
    if (world < 1 || rank < 0 || rank >= world) { set_error("bad rank/world"); return kFail; }
    if (world > kMaxWorld) { set_error("more ranks than the staged gather supports (64)"); return kFail; }
    DeviceGuard guard;
    if (!guard.ok) return kFail;
    RcclApi &api = rccl_api();
    if (!api.load()) return kFail;
    RcclApi::UniqueId id;
    std::memcpy(&id, id_bytes, TRMF_UNIQUE_ID_BYTES);
    std::shared_ptr<RcclComm> c = std::make_shared<RcclComm>();
    c->rank = rank; c->world = world; c->device = tl_device() >= 0 ? tl_device() : g_device;
    const int rc = api.CommInitRank(&c->comm, world, id, rank);
    if (rc != 0) { set_error(std::string("ncclCommInitRank: ") + api.GetErrorString(rc)); c->comm = nullptr; return kFail; }
    g_comm = std::move(c);
    return 0;
}

int32_t trmf_dist_init_callback(int32_t rank, int32_t world, trmf_allgatherv_fn fn, void *ctx) {
    if (world < 1 || rank < 0 || rank >= world || !fn) { set_error("bad rank/world/callback"); return kFail; }
    if (world > kMaxWorld) { set_error("more ranks than the staged gather supports (64)"); return kFail; }
    std::shared_ptr<CallbackComm> c = std::make_shared<CallbackComm>();
    c->rank = rank; c->world = world; c->fn = fn; c->ctx = ctx;
    g_comm = std::move(c);
    return 0;
}

int32_t trmf_dist_init_solo(int32_t rank, int32_t world) {
    if (world < 1 || rank < 0 || rank >= world || world > kMaxWorld) { set_error("bad rank/world"); return kFail; }
    std::shared_ptr<SoloComm> c = std::make_shared<SoloComm>();
    c->rank = rank; c->world = world;
    g_comm = std::move(c);
    return 0;
}

int32_t trmf_dist_rank(void) { return active_comm()->rank; }
int32_t trmf_dist_world(void) { return active_comm()->world; }
void trmf_dist_finalize(void) { g_comm.reset(); }

int32_t trmf_partition_by_nnz(uint64_t nrows, const size_t *ptr, int32_t world, uint64_t *bounds) {
    if (world < 1 || !ptr || !bounds) { set_error("bad arguments"); return kFail; }
    partition_by_nnz<size_t>(nrows, ptr, world, bounds);
    return 0;
}

}  // extern "C"
