// dev_buf.hpp -- device buffers of a session: pooled allocation, stream-ordered zero fill and upload (device_pool.hpp), and the
// small PODs the session keeps per iteration.
#pragma once

#include <algorithm>
#include <string>

#include "kernel_units.hpp"      // cg_kernels.hpp (XState), cg_persist_args.hpp, gram_kernels.hpp + which unit compiles which instantiation
#include "device_pool.hpp"

namespace trmf {

// Stream that zero-fills of fresh buffers are ordered on (the owning session's solver stream, set for the duration of
// every entry point that allocates).  hipMemset runs on the NULL stream and may return before the fill has executed, and
// the solver's stream is non-blocking (never ordered against the NULL stream): filling ON the solver stream orders the
// fill before every later user without a device-wide wait per buffer (round 2: one hipDeviceSynchronize per buffer,
// ~25 of them per create / append_rows).
inline hipStream_t &fill_stream() { static thread_local hipStream_t s = nullptr; return s; }
struct FillStreamScope {
    hipStream_t prev;
    explicit FillStreamScope(hipStream_t s) : prev(fill_stream()) { fill_stream() = s; }
    ~FillStreamScope() { fill_stream() = prev; }
};

// Pool blocks are reused without device synchronisation (device_pool.hpp "Ordering"): code that enqueues work on LOCAL buffers
// declares one of these AFTER them, so that on every early return -- a failed check between the enqueue and the function's own
// synchronisation -- the stream is drained before the locals go back to the pool (hipFree used to synchronise implicitly).
struct SyncStreamOnExit {
    hipStream_t s;
    explicit SyncStreamOnExit(hipStream_t st) : s(st) {}
    ~SyncStreamOnExit() { if (s) (void)hipStreamSynchronize(s); }
};

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;       // elements in use / allocated (a buffer that shrinks or regrows within cap is reused)
    DevicePool *pool = nullptr;  // where `p` came from (device_pool.hpp: slabs shared by the sessions of this process)
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { pool->free(p); p = nullptr; n = 0; cap = 0; } }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(pool, o.pool); }
    int alloc(size_t count, bool zero = true) {
        const size_t want = std::max<size_t>(count, 1);
        if (!p || want > cap) {
            release();
            pool = &DevicePool::current();
            p = static_cast<T *>(pool->alloc(want * sizeof(T)));
            if (!p) { set_error("device allocation of " + std::to_string(want * sizeof(T)) + " bytes failed"); return kFail; }
            cap = want;
        }
        n = count;
        if (zero) {
            if (fill_stream()) TRMF_HIP_CHECK(hipMemsetAsync(p, 0, want * sizeof(T), fill_stream()));
            else {              // no session stream known: fill on the NULL stream and wait for it
                TRMF_HIP_CHECK(hipMemset(p, 0, want * sizeof(T)));
                TRMF_HIP_CHECK(hipDeviceSynchronize());
            }
        }
        return 0;
    }
    // Host array -> this buffer through the library's pinned ring (device_pool.hpp), ordered on the owning session's stream;
    // returns once the SOURCE has been read (the caller's array may go away), not when the bytes have landed.
    int upload(const T *src, size_t count) {
        if (alloc(count, false)) return kFail;
        if (!count) return 0;
        if (fill_stream()) return HostStager::current().h2d(p, src, count * sizeof(T), fill_stream());
        TRMF_HIP_CHECK(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
};

struct DeviceIterLog {       // one per ALS iteration, filled on device
    double normF, normX, normLV;
    XState x;
};

struct PhaseEvents { hipEvent_t f0, fk0, fk1, f1, xg1, x1, lv1; };   // xg1: end of the X-side Gram build (start of the CG)

}  // namespace trmf
