// comm.hpp -- the only communication the sharded ALS loop needs: an in-place all-gather of
// contiguous, possibly uneven, row blocks of a device buffer.
//
//   SelfComm      world == 1: nothing to do.
//   RcclComm      one process per GPU, RCCL over xGMI.  librccl is dlopen()ed on first use so a
//                 single-GPU caller never loads it, and so that a process that already holds an
//                 RCCL (e.g. through torch.distributed) shares that copy.  Uneven blocks are
//                 gathered as `world` grouped ncclBroadcast calls (in place, root = owner): on a
//                 fully connected xGMI node each is one direct write per peer link.
//   CallbackComm  host-staged gather through a caller-supplied function (tests drive it with
//                 torch.distributed/gloo; also usable where RCCL is unavailable).
#pragma once

#include <dlfcn.h>

#include <cstring>
#include <vector>

#include "../../include/trmf_abi.h"
#include "common.hpp"

namespace trmf {

struct Comm {
    int rank = 0, world = 1;
    bool call_when_single = false;   // issue the (degenerate) gather even for world == 1
    virtual ~Comm() {}
    // Gather in place: rank r owns bytes [off[r], off[r+1]) of dbuf (device memory).
    virtual int allgatherv(void *dbuf, const uint64_t *off, hipStream_t stream) = 0;
    // Several gathers issued between group_begin() and group_end() may be fused into one transfer round.
    virtual int group_begin() { return 0; }
    virtual int group_end() { return 0; }
};

struct SelfComm : Comm {
    int allgatherv(void *, const uint64_t *, hipStream_t) override { return 0; }
};

struct CallbackComm : Comm {
    trmf_allgatherv_fn fn = nullptr;
    void *ctx = nullptr;
    std::vector<unsigned char> host;
    int allgatherv(void *dbuf, const uint64_t *off, hipStream_t stream) override {
        const uint64_t total = off[world];
        if (host.size() < total) host.resize(total);
        const uint64_t b0 = off[rank], b1 = off[rank + 1];
        if (b1 > b0)
            TRMF_HIP_CHECK(hipMemcpyAsync(host.data() + b0, (char *)dbuf + b0, b1 - b0, hipMemcpyDeviceToHost, stream));
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        if (fn(host.data(), off, world, ctx) != 0) { set_error("allgatherv callback failed"); return kFail; }
        for (int r = 0; r < world; r++) {
            if (r == rank || off[r + 1] == off[r]) continue;
            TRMF_HIP_CHECK(hipMemcpyAsync((char *)dbuf + off[r], host.data() + off[r], off[r + 1] - off[r],
                                          hipMemcpyHostToDevice, stream));
        }
        TRMF_HIP_CHECK(hipStreamSynchronize(stream));
        return 0;
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
    ~RcclComm() override {
        if (!comm) return;
        int prev = -1;
        if (hipGetDevice(&prev) == hipSuccess && prev != device) (void)hipSetDevice(device);
        rccl_api().CommDestroy(comm);
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    }
    int group_begin() override { return rccl_api().GroupStart() == 0 ? 0 : kFail; }
    int group_end() override {
        const int rc = rccl_api().GroupEnd();
        if (rc != 0) { set_error(std::string("RCCL group failed: ") + rccl_api().GetErrorString(rc)); return kFail; }
        return 0;
    }
    int allgatherv(void *dbuf, const uint64_t *off, hipStream_t stream) override {
        RcclApi &api = rccl_api();
        constexpr int kNcclInt8 = 0;                     // ncclInt8 / ncclChar
        int rc = api.GroupStart();
        for (int r = 0; r < world && rc == 0; r++) {
            const uint64_t bytes = off[r + 1] - off[r];
            if (bytes == 0) continue;
            char *p = (char *)dbuf + off[r];
            rc = api.Broadcast(p, p, bytes, kNcclInt8, r, comm, stream);
        }
        const int rc2 = api.GroupEnd();
        if (rc == 0) rc = rc2;
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
