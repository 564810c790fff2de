// comm.hpp -- the only communication the sharded ALS loop needs: an in-place all-gather of
// contiguous, possibly uneven, row blocks of a device buffer.
//
//   SelfComm      world == 1: nothing to do.
//   RcclComm      one process per GPU, RCCL over xGMI.  librccl is dlopen()ed on first use so a
//                 single-GPU caller never loads it, and so that a process that already holds an
//                 RCCL (e.g. through torch.distributed) shares that copy.  Uneven blocks travel as ONE
//                 ncclAllGather of equal slots: every rank copies its block into its slot of a staging
//                 buffer (slot = largest block), the slots are gathered in place, and a small kernel
//                 moves the other ranks' blocks to where they belong -- all links of the fully
//                 connected xGMI node carry data at once.  TRMF_RCCL_GATHER=broadcast selects the
//                 earlier form instead (`world` grouped in-place ncclBroadcast calls, root = owner).
//   CallbackComm  host-staged gather through a caller-supplied function (tests drive it with
//                 torch.distributed/gloo; also usable where RCCL is unavailable).  Same slot staging
//                 and unpack kernel as RcclComm, so two ranks on one GPU exercise them.
#pragma once

#include <atomic>

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/trmf_abi.h"
#include "common.hpp"

namespace trmf {

struct Comm {
    int rank = 0, world = 1;
    // distinguishes communicators in the process-level cache of measure-once decisions (session.hpp): a new communicator
    // (other connections, maybe other devices) never inherits what was measured under an old one
    const uint64_t id = [] { static std::atomic<uint64_t> next{1}; return next.fetch_add(1); }();
    bool call_when_single = false;   // issue the (degenerate) gather even for world == 1
    virtual ~Comm() {}
    // Gather in place: rank r owns bytes [off[r], off[r+1]) of dbuf (device memory).
    int allgatherv(void *dbuf, const uint64_t *off, hipStream_t stream) { return allgatherv_ranges(dbuf, off, off + 1, stream); }
    // The same with an arbitrary (disjoint) byte range [begin[r], end[r]) per rank -- e.g. the first half of every rank's
    // block, gathered while the second half is still being computed.
    virtual int allgatherv_ranges(void *dbuf, const uint64_t *begin, const uint64_t *end, hipStream_t stream) = 0;
    // Several gathers issued between group_begin() and group_end() may be fused into one transfer round.
    virtual int group_begin() { return 0; }
    virtual int group_end() { return 0; }
    // Gather in place with EQUAL slots: rank r owns bytes [r * slot, (r + 1) * slot) of dbuf.  No staging, no unpack:
    // the per-step exchange of the time-sharded CG (one message slot per rank: tile records + edge rows).
    virtual int allgather_slots(void *dbuf, size_t slot_bytes, hipStream_t stream) = 0;
    virtual bool solo() const { return false; }      // SoloComm: rank r of N without peers (measurement aid)
    // The ranks are threads of ONE process (session_group.hpp: TRMF_DEVICES behind the unchanged c_trmf_train): device pointers of
    // a peer are valid here as they are -- same device, or another device after hipDeviceEnablePeerAccess -- so the peer-to-peer
    // arenas are exchanged as raw pointers instead of IPC handles (which a process cannot open on itself).
    virtual bool in_process() const { return false; }
    virtual int device_of(int /*rank*/) const { return -1; }
    // all ranks of an in-process group meet here (no-op elsewhere): frees of memory that peers store into are ordered behind it
    virtual int barrier() { return 0; }
    // this rank is about to give up (in-process groups: breaks the group's barrier so that no peer waits for it -- or mistakes a barrier
    // this rank passes on its way out for the one of the collective they are in)
    virtual void abort() {}
};

// ---- equal-slot staging shared by the communicators -----------------------------------------------
constexpr int kMaxWorld = 64;
struct GatherOffsets { uint64_t b[kMaxWorld], e[kMaxWorld]; };
// block (x, r): bytes [b[r], e[r]) of the result come from slot r of the staging buffer; VEC bytes per thread
template <typename V>
__global__ __launch_bounds__(256) void gather_unpack_kernel(const unsigned char *__restrict__ stage,
                                                            unsigned char *__restrict__ dbuf, GatherOffsets o,
                                                            uint64_t slot, int rank) {
    const int r = blockIdx.y;
    if (r == rank) return;
    const uint64_t n = (o.e[r] - o.b[r]) / sizeof(V);
    const V *src = reinterpret_cast<const V *>(stage + (uint64_t)r * slot);
    V *dst = reinterpret_cast<V *>(dbuf + o.b[r]);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}
struct StagePool {
    // `last` is only ever COMPARED, never passed to the runtime: the pool lives in the communicator, which outlives the
    // sessions (and their streams) it serves, so the handle may belong to a stream that no longer exists.
    struct Buf { unsigned char *p; size_t cap; bool busy; hipStream_t last; };
    std::deque<Buf> bufs;                    // stable addresses: gathers of an open group keep pointers into it
    ~StagePool() { for (Buf &b : bufs) (void)hipFree(b.p); }
    Buf *acquire(size_t bytes, hipStream_t stream) {
        // a free buffer this stream used last (reuse on one stream is ordered by the stream itself) ...
        for (Buf &b : bufs)
            if (!b.busy && b.cap >= bytes && b.last == stream) return &b;
        // ... else, while the pool is small, a fresh one (two streams of one session -- the F-solve's overlapped gather -- keep
        // their own buffers and never wait for each other) ...
        if (bufs.size() >= 8)
            for (Buf &b : bufs)
                if (!b.busy && b.cap >= bytes) {
                    // ... else one another stream used last (a previous session under this communicator): wait for the whole
                    // device -- set-up path -- instead of synchronising a handle that may be destroyed
                    if (hipDeviceSynchronize() != hipSuccess) { set_error("device synchronisation failed"); return nullptr; }
                    b.last = stream;
                    return &b;
                }
        Buf b{nullptr, (bytes + 4095) / 4096 * 4096, false, stream};
        if (hipMalloc((void **)&b.p, b.cap) != hipSuccess) { set_error("hipMalloc of a gather staging buffer failed"); return nullptr; }
        bufs.push_back(b);
        return &bufs.back();
    }
};
struct StagedGather {                       // one gather between pack and unpack
    unsigned char *dbuf; GatherOffsets o; uint64_t slot; StagePool::Buf *buf; hipStream_t stream;
};
inline uint64_t gather_slot_bytes(const uint64_t *begin, const uint64_t *end, int world) {
    uint64_t slot = 0;
    for (int r = 0; r < world; r++) slot = std::max<uint64_t>(slot, end[r] - begin[r]);
    return (slot + 15) / 16 * 16;
}
inline int gather_pack(StagedGather &g, void *dbuf, const uint64_t *begin, const uint64_t *end, int rank, int world, StagePool &pool,
                       hipStream_t stream) {
    if (world > kMaxWorld) { set_error("more ranks than the staged gather supports"); return kFail; }
    g.dbuf = (unsigned char *)dbuf; g.slot = gather_slot_bytes(begin, end, world); g.stream = stream;
    for (int r = 0; r < world; r++) { g.o.b[r] = begin[r]; g.o.e[r] = end[r]; }
    g.buf = nullptr;
    if (g.slot == 0) return 0;                  // nothing anywhere: the callers stop here
    g.buf = pool.acquire(g.slot * world, stream);
    if (!g.buf) return kFail;
    const uint64_t mine = end[rank] - begin[rank];
    if (mine) TRMF_HIP_CHECK(hipMemcpyAsync(g.buf->p + (uint64_t)rank * g.slot, g.dbuf + begin[rank], mine, hipMemcpyDeviceToDevice, stream));
    return 0;
}
inline int gather_unpack(const StagedGather &g, int rank, int world) {
    uint64_t align = (uint64_t)(uintptr_t)g.dbuf, most = 0;
    for (int r = 0; r < world; r++) align |= g.o.b[r] | g.o.e[r];
    for (int r = 0; r < world; r++) if (r != rank) most = std::max<uint64_t>(most, g.o.e[r] - g.o.b[r]);
    if (most == 0) return 0;
    const int vec = (align % 16 == 0) ? 16 : (align % 8 == 0) ? 8 : (align % 4 == 0) ? 4 : 1;
    const dim3 grid((unsigned)std::min<uint64_t>(1024, (most / vec + 255) / 256), (unsigned)world);
    switch (vec) {
        case 16: hipLaunchKernelGGL(gather_unpack_kernel<uint4>, grid, dim3(256), 0, g.stream, g.buf->p, g.dbuf, g.o, g.slot, rank); break;
        case 8: hipLaunchKernelGGL(gather_unpack_kernel<uint64_t>, grid, dim3(256), 0, g.stream, g.buf->p, g.dbuf, g.o, g.slot, rank); break;
        case 4: hipLaunchKernelGGL(gather_unpack_kernel<uint32_t>, grid, dim3(256), 0, g.stream, g.buf->p, g.dbuf, g.o, g.slot, rank); break;
        default: hipLaunchKernelGGL(gather_unpack_kernel<unsigned char>, grid, dim3(256), 0, g.stream, g.buf->p, g.dbuf, g.o, g.slot, rank); break;
    }
    TRMF_HIP_CHECK(hipGetLastError());
    return 0;
}

struct SelfComm : Comm {
    int allgatherv_ranges(void *, const uint64_t *, const uint64_t *, hipStream_t) override { return 0; }
    int allgather_slots(void *, size_t, hipStream_t) override { return 0; }
};

// Measurement aid: rank r of a world of N with NO peers -- every gather is skipped.  A session under it runs exactly the
// kernels rank r would run (its item rows, its timestamps, its tiles) with the GPU to itself, so their times are a rank's
// compute share undisturbed by other processes; the factors it produces are NOT a solution (the other ranks' blocks are stale).
// The persistent-kernel form of the time-sharded CG (TRMF_CG=persist) runs under it in LOOP-BACK: the "peers' arenas" are this
// rank's own arena, and one workgroup on a side stream plays the other ranks (persist_peer_emulator_kernel, cg_persist_args.hpp):
// the rank's kernel -- unchanged -- does all of ITS work (its tiles' chain, the poll of the full record table, its publishes into
// N table copies) and is answered one memory round trip later: the per-step cost of a rank at N-way sharding, measured alone
// (VERDICT r4 item 2).
struct SoloComm : Comm {
    int allgatherv_ranges(void *, const uint64_t *, const uint64_t *, hipStream_t) override { return 0; }
    int allgather_slots(void *, size_t, hipStream_t) override { return 0; }
    bool solo() const override { return true; }
};

struct CallbackComm : Comm {
    trmf_allgatherv_fn fn = nullptr;
    void *ctx = nullptr;
    std::vector<unsigned char> host;
    StagePool pool;
    int allgatherv_ranges(void *dbuf, const uint64_t *begin, const uint64_t *end, hipStream_t stream) override {
        StagedGather g;
        if (gather_pack(g, dbuf, begin, end, rank, world, pool, stream)) return kFail;
        if (g.slot == 0) return 0;
        const uint64_t total = g.slot * world, mine = end[rank] - begin[rank];
        if (host.size() < total) host.resize(total);
        std::vector<uint64_t> eq(world + 1);
        for (int r = 0; r <= world; r++) eq[r] = (uint64_t)r * g.slot;
        if (mine)
            TRMF_HIP_CHECK(hipMemcpyAsync(host.data() + eq[rank], g.buf->p + eq[rank], mine, hipMemcpyDeviceToHost, stream));
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        if (fn(host.data(), eq.data(), world, ctx) != 0) { set_error("allgatherv callback failed"); return kFail; }
        for (int r = 0; r < world; r++) {
            if (r == rank || end[r] == begin[r]) continue;
            TRMF_HIP_CHECK(hipMemcpyAsync(g.buf->p + eq[r], host.data() + eq[r], end[r] - begin[r], hipMemcpyHostToDevice, stream));
        }
        if (gather_unpack(g, rank, world)) return kFail;
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        return 0;
    }
    int allgather_slots(void *dbuf, size_t slot, hipStream_t stream) override {
        if (slot == 0 || world == 1) return 0;
        const uint64_t total = (uint64_t)slot * world;
        if (host.size() < total) host.resize(total);
        std::vector<uint64_t> eq(world + 1);
        for (int r = 0; r <= world; r++) eq[r] = (uint64_t)r * slot;
        unsigned char *d = (unsigned char *)dbuf;
        TRMF_HIP_CHECK(hipMemcpyAsync(host.data() + eq[rank], d + eq[rank], slot, hipMemcpyDeviceToHost, stream));
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        if (fn(host.data(), eq.data(), world, ctx) != 0) { set_error("allgatherv callback failed"); return kFail; }
        if (rank > 0) TRMF_HIP_CHECK(hipMemcpyAsync(d, host.data(), eq[rank], hipMemcpyHostToDevice, stream));
        if (rank + 1 < world)
            TRMF_HIP_CHECK(hipMemcpyAsync(d + eq[rank + 1], host.data() + eq[rank + 1], total - eq[rank + 1], hipMemcpyHostToDevice, stream));
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        return 0;
    }
};

// ---- ranks as threads of one process --------------------------------------------------------------------------------------
// ThreadGroup: what the N rank threads of a session group share -- a reusable barrier that a failing rank can break (every
// waiter then returns an error instead of hanging), and a table of the device pointers of the collective in flight.
struct ThreadGroup {
    int world = 1;
    std::vector<int> device;                 // HIP device of every rank (duplicates allowed: virtual ranks on one device)
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    std::atomic<int> failed{0};
    std::vector<void *> ptr;
    explicit ThreadGroup(std::vector<int> devs) : world((int)devs.size()), device(std::move(devs)), ptr(world, nullptr) {}
    void fail() { failed.store(1); std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
    int barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed.load()) return kFail;
        const unsigned long long gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); return 0; }
        cv.wait(lk, [&] { return generation != gen || failed.load(); });
        return generation != gen ? 0 : kFail;
    }
};
// ThreadComm: the all-gathers of comm.hpp between the rank threads of a ThreadGroup -- every rank publishes its buffer, waits for
// the others' blocks to be complete (stream synchronisation + barrier), PULLS the other ranks' blocks with device-to-device copies
// (peer copies between devices) and meets the others again before anybody's buffer may change.  Host-synchronous like
// CallbackComm; the stream-ordered alternative on a real node is RcclComm per thread (session_group.hpp tries that first when
// every rank has a device of its own).
struct ThreadComm : Comm {
    std::shared_ptr<ThreadGroup> grp;
    bool in_process() const override { return true; }
    int device_of(int r) const override { return grp->device[r]; }
    int barrier() override {
        if (grp->barrier()) { set_error("another rank of the in-process group failed"); return kFail; }
        return 0;
    }
    void abort() override { grp->fail(); }
    int pull(void *dbuf, const uint64_t *begin, const uint64_t *end, hipStream_t stream) {
        auto bail = [&](const char *what) { grp->fail(); set_error(what); return kFail; };
        {
            std::lock_guard<std::mutex> lk(grp->mu);
            grp->ptr[rank] = dbuf;
        }
        if (hipStreamSynchronize(stream) != hipSuccess) return bail("in-process gather: stream synchronisation failed");
        if (barrier()) return kFail;
        for (int r = 0; r < world; r++) {
            if (r == rank || end[r] == begin[r]) continue;
            unsigned char *dst = (unsigned char *)dbuf + begin[r];
            const unsigned char *src = (const unsigned char *)grp->ptr[r] + begin[r];
            const hipError_t e = grp->device[r] == grp->device[rank]
                ? hipMemcpyAsync(dst, src, end[r] - begin[r], hipMemcpyDeviceToDevice, stream)
                : hipMemcpyPeerAsync(dst, grp->device[rank], src, grp->device[r], end[r] - begin[r], stream);
            if (e != hipSuccess) return bail("in-process gather: device-to-device copy failed");
        }
        if (hipStreamSynchronize(stream) != hipSuccess) return bail("in-process gather: stream synchronisation failed");
        return barrier();
    }
    int allgatherv_ranges(void *dbuf, const uint64_t *begin, const uint64_t *end, hipStream_t stream) override {
        return pull(dbuf, begin, end, stream);
    }
    int allgather_slots(void *dbuf, size_t slot, hipStream_t stream) override {
        if (slot == 0) return 0;
        std::vector<uint64_t> b(world), e(world);
        for (int r = 0; r < world; r++) { b[r] = (uint64_t)r * slot; e[r] = b[r] + slot; }
        return pull(dbuf, b.data(), e.data(), stream);
    }
};

// Minimal view of the RCCL C API (rccl.h), resolved at run time.
struct RcclApi {
    typedef struct { char internal[128]; } UniqueId;
    typedef void *CommT;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(CommT *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(CommT) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int /*dtype*/, int /*root*/, CommT, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t /*count per rank*/, int /*dtype*/, CommT, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *handle = nullptr;

    bool load() {
        if (handle) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { set_error(std::string("cannot dlopen librccl: ") + dlerror()); return false; }
        bool ok = true;
        auto sym = [&](const char *s) { void *p = dlsym(handle, s); if (!p) ok = false; return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) set_error("librccl is missing expected symbols");
        return ok;
    }
};
static_assert(TRMF_UNIQUE_ID_BYTES == 128, "ncclUniqueId is 128 bytes");

inline RcclApi &rccl_api() { static RcclApi api; return api; }

struct RcclComm : Comm {
    RcclApi::CommT comm = nullptr;
    RcclComm() { call_when_single = true; }   // keeps the RCCL call path testable on a 1-GPU box
    int device = 0;                           // HIP device the communicator was created on
    std::shared_ptr<ThreadGroup> grp;         // set when the ranks are threads of this process (session_group.hpp)
    bool in_process() const override { return grp != nullptr; }
    int device_of(int r) const override { return grp ? grp->device[r] : -1; }
    int barrier() override {
        if (grp && grp->barrier()) { set_error("another rank of the in-process group failed"); return kFail; }
        return 0;
    }
    void abort() override { if (grp) grp->fail(); }
    ~RcclComm() override {
        if (!comm) return;
        int prev = -1;
        if (hipGetDevice(&prev) == hipSuccess && prev != device) (void)hipSetDevice(device);
        rccl_api().CommDestroy(comm);
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    }
    StagePool pool;
    std::vector<StagedGather> pending;       // gathers issued inside a group: unpacked after ncclGroupEnd
    bool in_group = false;
    const bool by_broadcast = [] { const char *e = getenv("TRMF_RCCL_GATHER"); return e && e[0] == 'b'; }();
    int group_begin() override {
        if (rccl_api().GroupStart() != 0) return kFail;
        in_group = true;
        return 0;
    }
    int group_end() override {
        const int rc = rccl_api().GroupEnd();
        in_group = false;
        int bad = 0;
        for (const StagedGather &g : pending) { if (rc == 0 && gather_unpack(g, rank, world)) bad = 1; g.buf->busy = false; }
        pending.clear();
        if (rc != 0) { set_error(std::string("RCCL group failed: ") + rccl_api().GetErrorString(rc)); return kFail; }
        return bad ? kFail : 0;
    }
    int allgatherv_ranges(void *dbuf, const uint64_t *begin, const uint64_t *end, hipStream_t stream) override {
        RcclApi &api = rccl_api();
        constexpr int kNcclInt8 = 0;                     // ncclInt8 / ncclChar
        if (by_broadcast) return allgatherv_broadcast(dbuf, begin, end, stream);
        // Equal slots cost world x (largest block) of staging per gather in flight.  With a very skewed partition (row
        // counts of an nnz-balanced split can differ widely) that can be several times the gathered data itself: beyond
        // 4x the payload (and 64 MiB) the in-place form -- one broadcast per owner, no extra memory -- is used instead.
        {
            uint64_t payload = 0;
            for (int r = 0; r < world; r++) payload += end[r] - begin[r];
            const uint64_t stage = gather_slot_bytes(begin, end, world) * (uint64_t)world;
            if (stage > 4 * payload && stage > (64ull << 20)) return allgatherv_broadcast(dbuf, begin, end, stream);
        }
        // equal slots, one collective: in place in the staging buffer (send = own slot of the receive buffer)
        StagedGather g;
        if (gather_pack(g, dbuf, begin, end, rank, world, pool, stream)) return kFail;
        if (g.slot == 0) return 0;
        const int rc = api.AllGather(g.buf->p + (uint64_t)rank * g.slot, g.buf->p, g.slot, kNcclInt8, comm, stream);
        if (rc != 0) { set_error(std::string("RCCL all-gather failed: ") + api.GetErrorString(rc)); return kFail; }
        if (in_group) { g.buf->busy = true; pending.push_back(g); return 0; }    // the collective starts at ncclGroupEnd
        return gather_unpack(g, rank, world);
    }
    int allgather_slots(void *dbuf, size_t slot, hipStream_t stream) override {
        if (slot == 0) return 0;
        RcclApi &api = rccl_api();
        const int rc = api.AllGather((unsigned char *)dbuf + (size_t)rank * slot, dbuf, slot, /* ncclInt8 */ 0, comm, stream);
        if (rc != 0) { set_error(std::string("RCCL all-gather failed: ") + api.GetErrorString(rc)); return kFail; }
        return 0;
    }
    int allgatherv_broadcast(void *dbuf, const uint64_t *begin, const uint64_t *end, hipStream_t stream) {
        RcclApi &api = rccl_api();
        constexpr int kNcclInt8 = 0;
        int rc = in_group ? 0 : api.GroupStart();
        for (int r = 0; r < world && rc == 0; r++) {
            const uint64_t bytes = end[r] - begin[r];
            if (bytes == 0) continue;
            char *p = (char *)dbuf + begin[r];
            rc = api.Broadcast(p, p, bytes, kNcclInt8, r, comm, stream);
        }
        if (!in_group) { const int rc2 = api.GroupEnd(); if (rc == 0) rc = rc2; }
        if (rc != 0) { set_error(std::string("RCCL all-gather failed: ") + api.GetErrorString(rc)); return kFail; }
        return 0;
    }
};

// Contiguous row partition balanced by nnz (host logic, also exported for tests).
template <typename PtrT>
inline void partition_by_nnz(uint64_t nrows, const PtrT *ptr, int world, uint64_t *bounds) {
    const uint64_t total = nrows ? (uint64_t)ptr[nrows] - (uint64_t)ptr[0] : 0;
    bounds[0] = 0;
    uint64_t row = 0;
    for (int r = 1; r < world; r++) {
        // first row whose prefix nnz reaches r/world of the total (ties -> even row split)
        const uint64_t target = (uint64_t)ptr[0] + (total * (uint64_t)r) / (uint64_t)world;
        uint64_t lo = row, hi = nrows;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if ((uint64_t)ptr[mid] < target) lo = mid + 1; else hi = mid;
        }
        if (total == 0) lo = (nrows * (uint64_t)r) / (uint64_t)world;
        row = lo;
        bounds[r] = row;
    }
    bounds[world] = nrows;
}

}  // namespace trmf
