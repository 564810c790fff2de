// device_pool.hpp -- process-level resources behind the one-shot entry point (round 5).
//
// The reference's c_trmf_train wraps the caller's buffers zero-copy (trmf.cpp:696-725); here every call has to put the problem
// into HBM first, and at config 3 that set-up used to cost 0.12 s around 10-40 ms of compute.  Measured on the box
// (scripts/ubench/h2d.hip, profiles/r05_h2d_ubench.txt):
//   * hipMemcpy from pageable caller memory runs at 2.6 GB/s the first time a host range is seen (the runtime pins it page by
//     page: 66 ms for the 160 MB of the two orientations of Y), hipHostRegister + copy 17.6 ms, a ring of pinned chunks filled by
//     two host threads 3.6-3.9 ms (43-47 GB/s, close to the link's 56 GB/s)                              -> HostStager
//   * hipFree costs 0.1-0.2 ms apiece (it synchronises the device): 6.5 ms for a session's ~30 buffers; hipMalloc is cheap but
//     every fresh range is mapped on first touch                                                          -> DevicePool
//   * the first hipStreamCreate of a process costs 19.6 ms, pinned host memory 0.2 ms per MB             -> StreamCache, the ring
//     and the download staging are allocated once per process
// None of this is on the path of a resident session's iterations; it is what a caller of the reference API pays per call.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

#include "common.hpp"

namespace trmf {

// ---------------------------------------------------------------------------------------------------------------------------
// DevicePool: device allocations of all sessions of this process on one device.  Slabs (one hipMalloc each) are divided into
// blocks kept in address order; a request takes the smallest free block that fits (best fit, the remainder split off), a freed
// block merges with its free neighbours -- so the second call of a grid_search finds every buffer of the first, and a long-lived
// session beside many short ones of changing shapes does not make the pool grow without bound.  When the last live block goes
// (no session left) slabs above the cache cap (TRMF_POOL_MAX_MB, default 8192; 0 = plain hipMalloc / hipFree, no caching) or
// fragmented over several slabs are released and the next session gets ONE slab of everything the last one needed.  reserve()
// lets a session announce its footprint so that the first call allocates one slab.
// A session that GROWS (append_rows: every window's buffers are larger than every free block, so each generation gets a slab of its
// own) must not keep its earlier generations' slabs (31.6 GB in 527 slabs around 0.4 GB of live data after 200 windows):
// wholly free slabs are returned to the runtime whenever a NEW slab is needed (no free block fits the request, or reserve() announces
// a footprint that none can hold) -- they are evidently too small -- and before an allocation fails.
// Ordering: a block is reused without any device synchronisation.  Safe because every user of a session's buffers is enqueued
// on that session's stream (or follows a synchronisation of it), sessions synchronise their stream before they release, and
// temporaries of asynchronous set-up code are kept until the set-up's final synchronisation.
class DevicePool {
public:
    static DevicePool &current() {
        static std::mutex mu;
        static std::map<int, DevicePool *> pools;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        DevicePool *&p = pools[dev];
        if (!p) p = new DevicePool();         // lives for the process: device memory is returned by the runtime at exit
        return *p;
    }
    struct Stats { uint64_t hip_mallocs = 0, reused = 0, live = 0, slab_bytes = 0, slabs = 0, live_bytes = 0, slab_frees = 0; };

    void *alloc(size_t bytes) {
        const size_t need = round_up(std::max<size_t>(bytes, 1));
        std::lock_guard<std::mutex> lk(mu_);
        if (cap_bytes_ == 0) {                                    // caching off
            void *p = nullptr;
            if (hipMalloc(&p, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            st_.hip_mallocs++; st_.live++;
            return p;
        }
        auto it = free_.lower_bound(std::make_pair(need, (unsigned char *)nullptr));      // best fit
        if (it == free_.end()) {
            // a new slab is needed: slabs that are wholly free are evidently too small for this request -- a growing session's
            // previous generations -- and go back to the runtime first (see the class comment)
            if (st_.live > 0) (void)release_idle_slabs();
            const size_t want = std::max(need, std::max(hint_, (size_t)(8u << 20)));
            hint_ = 0;
            unsigned char *base = nullptr;
            size_t got = want;
            if (hipMalloc((void **)&base, want) != hipSuccess) {
                (void)hipGetLastError();
                got = need;
                if (want == need || hipMalloc((void **)&base, need) != hipSuccess) {
                    // out of device memory: give back every slab nobody uses and try once more before reporting failure
                    (void)hipGetLastError();
                    if (release_idle_slabs() == 0 || hipMalloc((void **)&base, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
                }
            }
            st_.hip_mallocs++; st_.slabs++; st_.slab_bytes += got;
            slabs_.push_back(base); slab_sizes_[base] = got;
            blocks_[base] = Block{got, true, base};
            it = free_.emplace(got, base).first;
        } else st_.reused++;
        unsigned char *p = it->second;
        free_.erase(it);
        Block &b = blocks_[p];
        if (b.size - need >= kSplitMin) {                          // split the remainder off
            unsigned char *rest = p + need;
            blocks_[rest] = Block{b.size - need, true, b.slab};
            free_.emplace(b.size - need, rest);
            b.size = need;
        }
        b.free = false;
        st_.live++; st_.live_bytes += b.size;
        return p;
    }
    void free(void *ptr) {
        if (!ptr) return;
        std::lock_guard<std::mutex> lk(mu_);
        if (cap_bytes_ == 0) { st_.live--; (void)hipFree(ptr); return; }
        auto it = blocks_.find((unsigned char *)ptr);
        // not ours, or released twice: the counters stay as they are (a double release used to underflow `live` and switch the
        // consolidation at quiescence off for the rest of the process)
        if (it == blocks_.end() || it->second.free) return;
        st_.live--; st_.live_bytes -= it->second.size;
        it->second.free = true;
        auto nx = std::next(it);                                    // merge with the free neighbours of the same slab
        if (nx != blocks_.end() && nx->second.free && nx->second.slab == it->second.slab && it->first + it->second.size == nx->first) {
            free_.erase(std::make_pair(nx->second.size, nx->first));
            it->second.size += nx->second.size;
            blocks_.erase(nx);
        }
        if (it != blocks_.begin()) {
            auto pv = std::prev(it);
            if (pv->second.free && pv->second.slab == it->second.slab && pv->first + pv->second.size == it->first) {
                free_.erase(std::make_pair(pv->second.size, pv->first));
                pv->second.size += it->second.size;
                blocks_.erase(it);
                it = pv;
            }
        }
        free_.emplace(it->second.size, it->first);
        if (st_.live == 0 && (slabs_.size() > 1 || st_.slab_bytes > cap_bytes_)) {
            const size_t total = st_.slab_bytes;
            release_all();
            if (total <= cap_bytes_) hint_ = total;                 // the next session gets one slab of everything the last one needed
        }
    }
    // the next slab is at least this large (a session's estimate of its footprint)
    void reserve(size_t bytes) {
        std::lock_guard<std::mutex> lk(mu_);
        if (cap_bytes_ == 0) return;
        const size_t room = free_.empty() ? 0 : free_.rbegin()->first;
        if (st_.live == 0 && !slabs_.empty() && room < bytes) release_all();     // idle and too small: one slab of the right size instead
        else if (room < bytes) (void)release_idle_slabs();                       // a growing session: its previous generation's slab
        if (room < bytes) hint_ = std::max(hint_, round_up(bytes));
    }
    void trim() {
        std::lock_guard<std::mutex> lk(mu_);
        if (st_.live == 0) release_all();
    }
    Stats stats() {
        std::lock_guard<std::mutex> lk(mu_);
        return st_;
    }

private:
    struct Block { size_t size; bool free; unsigned char *slab; };
    static constexpr size_t kSplitMin = 4096;
    DevicePool() {
        cap_bytes_ = (size_t)8192 << 20;
        if (const char *e = getenv("TRMF_POOL_MAX_MB")) cap_bytes_ = (size_t)std::max(0ll, atoll(e)) << 20;
    }
    static size_t round_up(size_t b) { return (b + 255) / 256 * 256; }
    void release_all() {
        for (unsigned char *s : slabs_) (void)hipFree(s);
        slabs_.clear(); slab_sizes_.clear(); free_.clear(); blocks_.clear();
        st_.slab_bytes = 0; st_.slabs = 0;
    }
    size_t slab_size(unsigned char *base) const { auto it = slab_sizes_.find(base); return it == slab_sizes_.end() ? 0 : it->second; }
    // a wholly free slab (one free block spanning it) back to the runtime
    void release_slab(unsigned char *base) {
        auto it = blocks_.find(base);
        if (it == blocks_.end() || !it->second.free || it->second.size != slab_size(base)) return;
        free_.erase(std::make_pair(it->second.size, base));
        st_.slab_bytes -= it->second.size; st_.slabs--; st_.slab_frees++;
        blocks_.erase(it);
        slab_sizes_.erase(base);
        slabs_.erase(std::find(slabs_.begin(), slabs_.end(), base));
        (void)hipFree(base);
    }
    size_t release_idle_slabs() {
        size_t n = 0;
        for (size_t i = slabs_.size(); i-- > 0;) {
            unsigned char *base = slabs_[i];
            auto it = blocks_.find(base);
            if (it != blocks_.end() && it->second.free && it->second.size == slab_size(base)) { release_slab(base); n++; }
        }
        return n;
    }
    std::mutex mu_;
    std::vector<unsigned char *> slabs_;
    std::map<unsigned char *, size_t> slab_sizes_;
    std::map<unsigned char *, Block> blocks_;                      // every block of every slab, by address
    std::set<std::pair<size_t, unsigned char *>> free_;            // the free ones, by size
    size_t hint_ = 0, cap_bytes_ = 0;
    Stats st_;
};

// ---------------------------------------------------------------------------------------------------------------------------
// StreamCache: non-blocking streams handed from one session to the next (the first hipStreamCreate of a process costs ~20 ms).
// A stream is returned only after its owner has synchronised it.
class StreamCache {
public:
    static int acquire(hipStream_t *out) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lk(mu());
            auto &v = idle()[dev];
            if (!v.empty()) { *out = v.back(); v.pop_back(); return 0; }
        }
        TRMF_HIP_CHECK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
        return 0;
    }
    static void release(hipStream_t s) {
        if (!s) return;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu());
        auto &v = idle()[dev];
        if (v.size() < 8) v.push_back(s); else (void)hipStreamDestroy(s);
    }
    static void drop_idle() {
        std::lock_guard<std::mutex> lk(mu());
        for (auto &kv : idle()) { for (hipStream_t s : kv.second) (void)hipStreamDestroy(s); kv.second.clear(); }
    }
private:
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::map<int, std::vector<hipStream_t>> &idle() { static std::map<int, std::vector<hipStream_t>> m; return m; }
};

// ---------------------------------------------------------------------------------------------------------------------------
// HostStager: caller memory <-> device through pinned memory owned by the library.
//   h2d(): a ring of kSlots pinned chunks; two host threads copy (or convert) the caller's bytes into free chunks while the
//          calling thread enqueues one asynchronous copy per filled chunk on the given stream.  Returns when the SOURCE has been
//          read completely (the caller's array may go away); the device side completes in stream order.
//   staging(): one pinned buffer that grows to the largest download so far (capped), for the all-or-nothing commit of
//          c_trmf_train's outputs.
// One transfer at a time per process (a mutex); sessions of different threads serialise their uploads.
class HostStager {
public:
    static constexpr size_t kChunk = (size_t)4 << 20;
    static constexpr int kSlots = 6, kThreads = 2;
    static constexpr size_t kStagingCap = (size_t)1 << 30;      // larger downloads stage through ordinary host memory (config 5: 538 MB)
    static HostStager &current() {              // one per device (its events belong to that device)
        static std::mutex mu;
        static std::map<int, HostStager *> all;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        HostStager *&s = all[dev];
        if (!s) s = new HostStager();
        return *s;
    }

    // fill(dst_chunk, byte_offset, byte_count) writes the bytes [offset, offset + count) of the device image into dst_chunk
    template <typename Fill> int h2d_fill(void *dst, size_t bytes, hipStream_t stream, Fill fill) {
        if (bytes == 0) return 0;
        std::lock_guard<std::mutex> lk(mu_);
        if (ensure_ring()) return kFail;
        for (int j = 0; j < kSlots; j++)                       // chunks still in flight from the previous transfer
            if (busy_[j]) { TRMF_HIP_CHECK(hipEventSynchronize(ev_[j])); busy_[j] = false; }
        const size_t nchunks = (bytes + kChunk - 1) / kChunk;
        std::vector<std::atomic<int>> filled(nchunks), released(nchunks);
        for (size_t c = 0; c < nchunks; c++) { filled[c].store(0, std::memory_order_relaxed); released[c].store(0, std::memory_order_relaxed); }
        std::atomic<size_t> next{0};
        std::atomic<int> abort{0};
        auto worker = [&]() {
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= nchunks || abort.load(std::memory_order_relaxed)) break;
                if (c >= (size_t)kSlots)
                    while (!released[c - kSlots].load(std::memory_order_acquire)) { if (abort.load(std::memory_order_relaxed)) return; std::this_thread::yield(); }
                const size_t off = c * kChunk, len = std::min(kChunk, bytes - off);
                fill(ring_ + (c % kSlots) * kChunk, off, len);
                filled[c].store(1, std::memory_order_release);
            }
        };
        const int nth = nchunks >= 4 ? kThreads : nchunks >= 2 ? 1 : 0;      // small transfers: the calling thread alone
        std::vector<std::thread> th;
        for (int w = 0; w < nth; w++) th.emplace_back(worker);
        int rc = 0;
        size_t done = 0;
        for (size_t c = 0; c < nchunks && rc == 0; c++) {
            if (nth == 0) worker();                            // (fills every chunk on the first call: nchunks <= 1 here)
            while (!filled[c].load(std::memory_order_acquire)) std::this_thread::yield();
            const size_t off = c * kChunk, len = std::min(kChunk, bytes - off);
            const int slot = (int)(c % kSlots);
            if (hipMemcpyAsync((unsigned char *)dst + off, ring_ + slot * kChunk, len, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipEventRecord(ev_[slot], stream) != hipSuccess) { rc = kFail; break; }
            busy_[slot] = true;
            // hand back the chunks whose copies have left the host (oldest first), keeping half the ring in flight
            while (done + kSlots / 2 <= c && c + 1 < nchunks) {
                if (hipEventSynchronize(ev_[done % kSlots]) != hipSuccess) { rc = kFail; break; }
                busy_[done % kSlots] = false;
                released[done].store(1, std::memory_order_release);
                done++;
            }
        }
        if (rc) { abort.store(1); set_error(std::string("host-to-device staging failed: ") + hipGetErrorString(hipGetLastError())); }
        for (size_t c = done; c < nchunks; c++) released[c].store(1, std::memory_order_release);
        for (auto &t : th) t.join();
        bytes_h2d += bytes;
        return rc;
    }
    int h2d(void *dst, const void *src, size_t bytes, hipStream_t stream) {
        const unsigned char *s = (const unsigned char *)src;
        return h2d_fill(dst, bytes, stream, [s](unsigned char *chunk, size_t off, size_t len) { std::memcpy(chunk, s + off, len); });
    }
    // 64-bit row / column pointers of the ABI -> the 32-bit pointers the kernels read (nnz < 2^32 is checked at the boundary)
    int h2d_narrow(uint32_t *dst, const uint64_t *src, size_t count, hipStream_t stream) {
        return h2d_fill(dst, count * sizeof(uint32_t), stream, [src](unsigned char *chunk, size_t off, size_t len) {
            uint32_t *o = reinterpret_cast<uint32_t *>(chunk);
            const uint64_t *s = src + off / sizeof(uint32_t);
            for (size_t e = 0; e < len / sizeof(uint32_t); e++) o[e] = (uint32_t)s[e];
        });
    }
    // pinned staging of at least `bytes`, leased to the caller until `lease` is released (nullptr: too large or the allocation
    // failed -- the caller falls back to ordinary memory)
    unsigned char *staging(size_t bytes, std::unique_lock<std::mutex> &lease) {
        if (bytes > kStagingCap) return nullptr;
        lease = std::unique_lock<std::mutex>(stage_mu_);
        if (bytes > stage_cap_) {
            if (stage_) (void)hipHostFree(stage_);
            stage_ = nullptr; stage_cap_ = 0;
            const size_t want = std::max<size_t>((bytes + (bytes >> 2) + 4095) / 4096 * 4096, (size_t)1 << 20);
            if (hipHostMalloc((void **)&stage_, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); stage_ = nullptr; lease.unlock(); return nullptr; }
            stage_cap_ = want;
        }
        return stage_;
    }
    void release_staging() {
        std::lock_guard<std::mutex> lk(stage_mu_);
        if (stage_) (void)hipHostFree(stage_);
        stage_ = nullptr; stage_cap_ = 0;
    }
    // the upload ring (24 MB of pinned memory) and its events; the next upload allocates them again
    void release_ring() {
        std::lock_guard<std::mutex> lk(mu_);
        if (!ring_) return;
        for (int j = 0; j < kSlots; j++) {
            if (busy_[j]) { (void)hipEventSynchronize(ev_[j]); busy_[j] = false; }
            if (ev_[j]) (void)hipEventDestroy(ev_[j]);
            ev_[j] = nullptr;
        }
        (void)hipHostFree(ring_);
        ring_ = nullptr;
    }
    // host copy with the stager's worker count (committing staged outputs into the caller's arrays)
    static void parallel_copy(void *dst, const void *src, size_t bytes) {
        if (bytes < ((size_t)4 << 20)) { std::memcpy(dst, src, bytes); return; }
        const int parts = bytes >= ((size_t)64 << 20) ? 4 : 2;
        const size_t piece = (bytes / parts + 63) / 64 * 64;
        std::vector<std::thread> th;
        for (int q = 1; q < parts; q++) {
            const size_t off = std::min(bytes, q * piece), len = std::min(bytes, (q + 1) * piece) - off;
            if (len) th.emplace_back([=] { std::memcpy((unsigned char *)dst + off, (const unsigned char *)src + off, len); });
        }
        std::memcpy(dst, src, std::min(bytes, piece));
        for (auto &t : th) t.join();
    }
    std::atomic<uint64_t> bytes_h2d{0};

private:
    int ensure_ring() {
        if (ring_) return 0;
        TRMF_HIP_CHECK(hipHostMalloc((void **)&ring_, kChunk * kSlots, hipHostMallocDefault));
        for (int j = 0; j < kSlots; j++) { TRMF_HIP_CHECK(hipEventCreateWithFlags(&ev_[j], hipEventDisableTiming)); busy_[j] = false; }
        return 0;
    }
    std::mutex mu_, stage_mu_;
    unsigned char *ring_ = nullptr, *stage_ = nullptr;
    size_t stage_cap_ = 0;
    hipEvent_t ev_[kSlots] = {};
    bool busy_[kSlots] = {};
};

}  // namespace trmf
