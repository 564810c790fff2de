// Can kernels of TWO PROCESSES exchange flags through IPC-mapped device memory while both are running?  (The transport of
// the time-sharded CG's peer-to-peer form: a rank's kernel writes its tile records / edge rows into the peers' message
// buffers and raises a flag; the peer's next launch waits for it.)  One GPU is enough to answer the software question:
//   ipc_flag server <file>   allocates the buffer, publishes its IPC handle, plays "ping"
//   ipc_flag client <file>   opens the handle, plays "pong"
// Every spin is bounded (2 s) so that a platform without concurrent execution reports a timeout instead of hanging.
// Build: hipcc --offload-arch=gfx950 -O3 ipc_flag.hip -o ipc_flag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <unistd.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ bool wait_ge(volatile unsigned long long *f, unsigned long long v, long long limit_ticks) {
    const long long t0 = wall_clock64();
    while (__atomic_load_n((unsigned long long *)f, __ATOMIC_RELAXED) < v) {
        if (wall_clock64() - t0 > limit_ticks) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return true;
}
// role 0 (server): raise f[0] = i, wait for f[1] = i;  role 1 (client): wait for f[0] = i, raise f[1] = i
__global__ void pingpong(unsigned long long *f, int role, int rounds, long long *out) {
    const long long limit = 200000000;     // 2 s of the 100 MHz wall clock
    long long t0 = wall_clock64();
    int done = 0;
    for (int i = 1; i <= rounds; i++) {
        if (role == 0) {
            __atomic_store_n(&f[0], (unsigned long long)i, __ATOMIC_RELEASE);
            if (!wait_ge(&f[8], i, limit)) break;
        } else {
            if (!wait_ge(&f[0], i, limit)) break;
            __atomic_store_n(&f[8], (unsigned long long)i, __ATOMIC_RELEASE);
        }
        done = i;
    }
    out[0] = done; out[1] = wall_clock64() - t0;
}

int main(int argc, char **argv) {
    if (argc < 3) { printf("usage: ipc_flag server|client <handle file> [uncached]\n"); return 2; }
    const bool server = !strcmp(argv[1], "server");
    const bool uncached = argc > 3 && !strcmp(argv[3], "uncached");
    const int rounds = 1000;
    unsigned long long *f = nullptr;
    long long *out = nullptr;
    CK(hipMalloc(&out, 16));
    if (server) {
        if (uncached) CK(hipExtMallocWithFlags((void **)&f, 4096, hipDeviceMallocUncached));
        else CK(hipMalloc(&f, 4096));
        CK(hipMemset(f, 0, 4096));
        CK(hipDeviceSynchronize());
        hipIpcMemHandle_t h;
        CK(hipIpcGetMemHandle(&h, f));
        FILE *fp = fopen(argv[2], "wb"); fwrite(&h, sizeof h, 1, fp); fclose(fp);
        char ready[512]; snprintf(ready, sizeof ready, "%s.ready", argv[2]);
        fp = fopen(ready, "w"); fclose(fp);
    } else {
        char ready[512]; snprintf(ready, sizeof ready, "%s.ready", argv[2]);
        for (int i = 0; i < 200 && access(ready, F_OK) != 0; i++) usleep(50000);
        hipIpcMemHandle_t h;
        FILE *fp = fopen(argv[2], "rb"); if (!fp) { printf("no handle file\n"); return 1; }
        if (fread(&h, sizeof h, 1, fp) != 1) return 1;
        fclose(fp);
        CK(hipIpcOpenMemHandle((void **)&f, h, hipIpcMemLazyEnablePeerAccess));
    }
    pingpong<<<1, 1>>>(f, server ? 0 : 1, rounds, out);
    CK(hipDeviceSynchronize());
    long long h[2];
    CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    printf("%s (%s memory): %lld of %d rounds, %.2f us per round trip\n", server ? "server" : "client", uncached ? "uncached" : "hipMalloc",
           h[0], rounds, h[0] ? 0.01 * (double)h[1] / (double)h[0] : 0.0);
    if (!server) CK(hipIpcCloseMemHandle(f));
    else { usleep(300000); CK(hipFree(f)); }
    return h[0] == rounds ? 0 : 3;
}
