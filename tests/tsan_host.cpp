// Host-only ThreadSanitizer run of the library's multi-threaded host code: the process-level device pool, the
// stream cache and the pinned upload ring with its two copy threads (exp-trmf-nips16_amd/csrc/device_pool.hpp), and the in-process
// communicator of the session groups -- ThreadGroup's breakable barrier and ThreadComm's pull all-gather (csrc/comm.hpp) -- with the
// HIP runtime replaced by host stand-ins (malloc / memcpy / no-op events).  Built and run by tests/test_tsan_host.py, which cuts the
// classes out of the headers; any data race TSan reports fails the test.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
typedef int hipError_t;
const int hipSuccess = 0;
typedef struct FakeStream { int id; } *hipStream_t;
typedef struct FakeEvent { int id; } *hipEvent_t;
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n); return *p ? 0 : 1; }
inline hipError_t hipFree(void *p) { free(p); return 0; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n); return *p ? 0 : 1; }
inline hipError_t hipHostFree(void *p) { free(p); return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "fake"; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new FakeEvent{0}; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new FakeStream{0}; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }   // "the device" copies at once
inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
namespace trmf {
constexpr int kFail = -1;
inline void set_error(const std::string &) {}
}
#define TRMF_HIP_CHECK(expr) do { if ((expr) != hipSuccess) return ::trmf::kFail; } while (0)
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <set>
#include <thread>
#include <vector>
#include TSAN_POOL_HEADER      // DevicePool, StreamCache, HostStager
#include TSAN_COMM_HEADER      // Comm, ThreadGroup, ThreadComm

using namespace trmf;

static int pool_and_ring(int threads, int rounds) {
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            std::mt19937 rng(100 + t);
            hipStream_t s = nullptr;
            if (StreamCache::acquire(&s)) { bad++; return; }
            for (int r = 0; r < rounds; r++) {
                DevicePool::current().reserve((size_t)(8 + rng() % 24) << 20);
                std::vector<std::pair<unsigned char *, size_t>> mine;
                for (int b = 0; b < 12; b++) {
                    const size_t n = (rng() % 10 == 0) ? ((size_t)(5 + rng() % 20) << 20) : (rng() % 200000) + 1;     // some transfers span the whole ring
                    unsigned char *d = (unsigned char *)DevicePool::current().alloc(n);
                    if (!d) { bad++; return; }
                    std::vector<unsigned char> src(n);
                    for (size_t i = 0; i < n; i += 4099) src[i] = (unsigned char)(i * 31 + t);
                    if (HostStager::current().h2d(d, src.data(), n, s)) { bad++; return; }
                    for (size_t i = 0; i < n; i += 4099) if (d[i] != (unsigned char)(i * 31 + t)) { bad++; return; }
                    mine.emplace_back(d, n);
                }
                {   // the all-or-nothing download: a leased staging buffer, committed by the parallel copy
                    std::unique_lock<std::mutex> lease;
                    const size_t n = ((size_t)1 << 20) + (rng() % (6u << 20));
                    unsigned char *st = HostStager::current().staging(n, lease);
                    if (st) {
                        memset(st, t + 1, n);
                        std::vector<unsigned char> out(n);
                        HostStager::parallel_copy(out.data(), st, n);
                        if (out[0] != t + 1 || out[n - 1] != t + 1) bad++;
                    }
                }
                for (auto &kv : mine) DevicePool::current().free(kv.first);
                if (r % 5 == 4) { HostStager::current().release_ring(); DevicePool::current().trim(); }      // trmf_release_cached() from one thread while others work
            }
            StreamCache::release(s);
        });
    for (auto &t : th) t.join();
    return bad.load();
}

static int group_gathers(int world, int rounds, bool break_it) {
    std::vector<int> devs(world, 0);
    auto grp = std::make_shared<ThreadGroup>(devs);
    std::atomic<int> bad{0}, failed_as_expected{0};
    std::vector<std::thread> th;
    const size_t block = 4096;
    for (int r = 0; r < world; r++)
        th.emplace_back([&, r] {
            ThreadComm c;
            c.rank = r; c.world = world; c.grp = grp;
            std::vector<unsigned char> buf(block * world);
            std::vector<uint64_t> off(world + 1);
            for (int q = 0; q <= world; q++) off[q] = (uint64_t)q * block;
            hipStream_t s = nullptr;
            for (int it = 0; it < rounds; it++) {
                memset(buf.data(), 0xee, buf.size());
                memset(buf.data() + off[r], (unsigned char)(it * 7 + r), block);
                if (break_it && it == rounds / 2 && r == world - 1) { grp->fail(); failed_as_expected++; return; }      // a rank dies outside a collective
                const int rc = (it & 1) ? c.allgather_slots(buf.data(), block, s) : c.allgatherv(buf.data(), off.data(), s);
                if (rc) { if (break_it) failed_as_expected++; else bad++; return; }
                for (int q = 0; q < world; q++)
                    if (buf[off[q]] != (unsigned char)(it * 7 + q) || buf[off[q] + block - 1] != (unsigned char)(it * 7 + q)) { bad++; return; }
            }
        });
    for (auto &t : th) t.join();
    if (break_it && failed_as_expected.load() != world) return 1 + bad.load();        // every rank must come back (nobody waits for the dead one)
    return bad.load();
}

int main() {
    int bad = pool_and_ring(4, 20);
    printf("pool + ring + staging on 4 threads: %s\n", bad ? "FAILED" : "ok");
    for (int w : {2, 3, 4, 8}) {
        const int b = group_gathers(w, 200, false);
        printf("in-process gathers, %d ranks: %s\n", w, b ? "FAILED" : "ok");
        bad += b;
    }
    const int b = group_gathers(4, 50, true);
    printf("a rank fails: every other rank returns an error: %s\n", b ? "FAILED" : "ok");
    return bad + b ? 1 : 0;
}
