// h2d.hip -- what the set-up of a one-shot c_trmf_train call is made of (round 5): host-to-device copies of pageable
// caller memory by four routes, and the fixed costs of the runtime objects a session creates.
//   hipcc -O3 --offload-arch=gfx950 -pthread h2d.hip -o h2d && ./h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    CK(hipSetDevice(0));
    { void *w; double t = now(); CK(hipMalloc(&w, 1 << 20)); printf("first hipMalloc (runtime init)      %8.2f ms\n", 1e3 * (now() - t)); CK(hipFree(w)); }
    hipStream_t s; { double t = now(); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); printf("hipStreamCreate                      %8.3f ms\n", 1e3 * (now() - t)); }
    { std::vector<hipEvent_t> ev(400); double t = now(); for (auto &e : ev) CK(hipEventCreate(&e)); printf("400 x hipEventCreate                 %8.3f ms\n", 1e3 * (now() - t));
      t = now(); for (auto &e : ev) CK(hipEventDestroy(e)); printf("400 x hipEventDestroy                %8.3f ms\n", 1e3 * (now() - t)); }
    for (size_t mb : {1, 16, 40, 256}) {
        void *p[30]; double t = now();
        for (int i = 0; i < 30; i++) CK(hipMalloc(&p[i], mb << 20));
        double t1 = now();
        for (int i = 0; i < 30; i++) CK(hipFree(p[i]));
        printf("30 x hipMalloc(%3zu MB) %8.3f ms   30 x hipFree %8.3f ms\n", mb, 1e3 * (t1 - t), 1e3 * (now() - t1));
    }
    { void *p; double t = now(); CK(hipMalloc(&p, (size_t)400 << 20)); double t1 = now(); CK(hipMemsetAsync(p, 0, (size_t)400 << 20, s)); CK(hipStreamSynchronize(s));
      printf("one hipMalloc(400 MB) %8.3f ms, memset of it %8.3f ms\n", 1e3 * (t1 - t), 1e3 * (now() - t1)); CK(hipFree(p)); }
    for (size_t mb : {8, 64}) { void *h; double t = now(); CK(hipHostMalloc(&h, mb << 20, hipHostMallocDefault)); double t1 = now(); memset(h, 1, mb << 20); double t2 = now(); CK(hipHostFree(h));
      printf("hipHostMalloc(%2zu MB) %8.3f ms, first touch %8.3f ms, hipHostFree %8.3f ms\n", mb, 1e3 * (t1 - t), 1e3 * (t2 - t1), 1e3 * (now() - t2)); }
    const size_t N = (size_t)160 << 20;
    std::vector<char> host(N, 3);
    void *d; CK(hipMalloc(&d, N));
    for (int rep = 0; rep < 3; rep++) {
        double t = now(); CK(hipMemcpy(d, host.data(), N, hipMemcpyHostToDevice)); double dt = now() - t;
        printf("hipMemcpy pageable 160 MB            %8.3f ms  %6.1f GB/s\n", 1e3 * dt, N / dt / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); CK(hipMemcpyAsync(d, host.data(), N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double dt = now() - t;
        printf("hipMemcpyAsync pageable 160 MB       %8.3f ms  %6.1f GB/s\n", 1e3 * dt, N / dt / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); CK(hipHostRegister(host.data(), N, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpyAsync(d, host.data(), N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now();
        CK(hipHostUnregister(host.data())); double t3 = now();
        printf("register %7.3f + copy %7.3f (%5.1f GB/s) + unregister %7.3f = %8.3f ms\n", 1e3 * (t1 - t), 1e3 * (t2 - t1), N / (t2 - t1) / 1e9, 1e3 * (t3 - t2), 1e3 * (t3 - t));
    }
    // staged: ring of pinned chunks, `nth` threads fill, main thread enqueues
    for (size_t chunk_mb : {2, 4, 8}) for (int nth : {1, 2, 4, 8}) {
        const size_t C = chunk_mb << 20; const int slots = 8;
        char *pin; CK(hipHostMalloc((void **)&pin, C * slots, hipHostMallocDefault)); memset(pin, 0, C * slots);
        hipEvent_t ev[slots]; for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            const size_t nchunks = (N + C - 1) / C;
            std::vector<std::atomic<int>> filled(nchunks); for (auto &f : filled) f = 0;
            std::vector<std::atomic<int>> freed(nchunks); for (auto &f : freed) f = 0;
            std::atomic<size_t> next{0};
            double t = now();
            std::vector<std::thread> th;
            for (int w = 0; w < nth; w++) th.emplace_back([&] {
                for (;;) {
                    const size_t c = next.fetch_add(1); if (c >= nchunks) break;
                    if (c >= (size_t)slots) while (!freed[c - slots].load(std::memory_order_acquire)) std::this_thread::yield();
                    const size_t off = c * C, len = std::min(C, N - off);
                    memcpy(pin + (c % slots) * C, host.data() + off, len);
                    filled[c].store(1, std::memory_order_release);
                }
            });
            size_t waited = 0;
            for (size_t c = 0; c < nchunks; c++) {
                while (!filled[c].load(std::memory_order_acquire)) std::this_thread::yield();
                const size_t off = c * C, len = std::min(C, N - off);
                (void)hipMemcpyAsync((char *)d + off, pin + (c % slots) * C, len, hipMemcpyHostToDevice, s);
                (void)hipEventRecord(ev[c % slots], s);
                // free the slots whose copies have completed (oldest first)
                while (waited + slots / 2 <= c) { (void)hipEventSynchronize(ev[waited % slots]); freed[waited].store(1, std::memory_order_release); waited++; }
            }
            CK(hipStreamSynchronize(s));
            for (; waited < nchunks; waited++) freed[waited].store(1);
            for (auto &x : th) x.join();
            best = std::min(best, now() - t);
        }
        printf("staged %zu MB chunks, %d threads: %8.3f ms  %6.1f GB/s\n", chunk_mb, nth, 1e3 * best, N / best / 1e9);
        for (auto &e : ev) (void)hipEventDestroy(e);
        CK(hipHostFree(pin));
    }
    // device to host, 18 MB
    { const size_t M = (size_t)18 << 20; std::vector<char> back(M);
      for (int rep = 0; rep < 3; rep++) { double t = now(); CK(hipMemcpy(back.data(), d, M, hipMemcpyDeviceToHost)); double dt = now() - t; printf("hipMemcpy D2H pageable 18 MB         %8.3f ms  %6.1f GB/s\n", 1e3 * dt, M / dt / 1e9); }
      char *pin; CK(hipHostMalloc((void **)&pin, M, hipHostMallocDefault));
      for (int rep = 0; rep < 3; rep++) { double t = now(); CK(hipMemcpyAsync(pin, d, M, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double t1 = now(); memcpy(back.data(), pin, M); double dt = now() - t;
        printf("D2H pinned 18 MB %8.3f ms (%5.1f GB/s) + memcpy out %8.3f ms\n", 1e3 * (t1 - t), M / (t1 - t) / 1e9, 1e3 * (dt - (t1 - t))); }
      CK(hipHostFree(pin)); }
    CK(hipFree(d));
    return 0;
}
