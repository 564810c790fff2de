// Grid-wide barrier on gfx950 WITHOUT a release/acquire fence (round 4).  scripts/ubench/gridsync.hip measured 47-95 us per
// barrier: its __threadfence() is an agent-scope release = one L2 write-back (buffer_wbl2) PER WORKGROUP.  Here everything
// the barrier publishes is stored write-through (relaxed agent-scope atomic stores: sc1) and read around the L2 (relaxed
// agent-scope atomic loads), so the barrier itself needs no cache maintenance: s_waitcnt vmcnt(0) + one relaxed atomic add
// + a relaxed poll.  Also: what a no-op launch of the CG kernel's shape costs (400 x 256 threads, 48 KB of dynamic LDS).
// Build: hipcc --offload-arch=gfx950 -O3 gridsync2.hip -o gridsync2
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T> __device__ __forceinline__ void st_dev(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T ld_dev(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// flat monotone counter; POLL_ALL: every wavefront's lane 0 polls (no second __syncthreads)
#define LD(p) (INV ? *(p) : ld_dev(p))
template <bool POLL_ALL, int NDATA, bool INV>
__global__ __launch_bounds__(256, 2) void k_fast(float *out, double *rec, unsigned *ctr, int iters, int *bad) {
    float v = threadIdx.x;
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        float *o = out + (size_t)(it & 1) * 8 * 512 * 256;          // ping-pong: a workgroup is at most one barrier ahead
        double *rc = rec + (size_t)(it & 1) * 512 * 8;
        for (int m = 0; m < NDATA; m++) st_dev(&o[((size_t)m * nb + b) * 256 + t], (float)it);    // "halo rows"
        if (t == 0) { st_dev(&rc[b * 8 + 0], (double)it); st_dev(&rc[b * 8 + 1], 1.0); st_dev(&rc[b * 8 + 2], 2.0); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // own stores have left the CU
        __syncthreads();
        const unsigned target = (unsigned)(it + 1) * nb;
        if (POLL_ALL) {
            if (t == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((t & 63) == 0) {
                long spins = 0;
                while (ld_dev(ctr) < target && ++spins < (1L << 22)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1L << 22)) *bad = 1;
            }
        } else {
            if (t == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                long spins = 0;
                while (ld_dev(ctr) < target && ++spins < (1L << 22)) __builtin_amdgcn_s_sleep(1);
                if (spins >= (1L << 22)) *bad = 1;
            }
            __syncthreads();
        }
        // consume: every workgroup sums all records (as the CG does) and reads a neighbour's rows
        double s = 0;
        if (INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");             // buffer_inv sc1: then plain loads
        for (int i = t; i < nb; i += 256) s += LD(&rc[i * 8 + 0]) + LD(&rc[i * 8 + 1]) + LD(&rc[i * 8 + 2]);
        if (t < nb && LD(&rc[t * 8 + 0]) != (double)it) *bad = 2;          // stale record?
#pragma unroll
        for (int m = 0; m < NDATA; m++) {
            const float x = LD(&o[((size_t)m * nb + (b + 1) % nb) * 256 + t]);
            if (x != (float)it) *bad = 3;                                        // stale row?
            v += x * 1e-9f + (float)s * 1e-20f;
        }
    }
    out[(size_t)b * 256 + t] = v;
}

// two-level: one counter per XCD group (b % 8), then a global one -- fewer atomics on one address
template <bool INV>
__global__ __launch_bounds__(256, 2) void k_two(float *out, double *rec, unsigned *ctr, int iters, int *bad) {
    float v = threadIdx.x;
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    const int x = b % 8, members = nb / 8 + (x < nb % 8 ? 1 : 0);
    unsigned *cx = ctr + 64 + 32 * x, *cg = ctr;
    for (int it = 0; it < iters; it++) {
        float *o = out + (size_t)(it & 1) * 8 * 512 * 256;
        double *rc = rec + (size_t)(it & 1) * 512 * 8;
        for (int m = 0; m < 3; m++) st_dev(&o[((size_t)m * nb + b) * 256 + t], (float)it);
        if (t == 0) { st_dev(&rc[b * 8 + 0], (double)it); st_dev(&rc[b * 8 + 1], 1.0); st_dev(&rc[b * 8 + 2], 2.0); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (t == 0) {
            const unsigned old = __hip_atomic_fetch_add(cx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (unsigned)(it + 1) * members) __hip_atomic_fetch_add(cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            while (ld_dev(cg) < (unsigned)(it + 1) * 8 && ++spins < (1L << 22)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1L << 22)) *bad = 1;
        }
        __syncthreads();
        if (INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double s = 0;
        for (int i = t; i < nb; i += 256) s += LD(&rc[i * 8 + 0]) + LD(&rc[i * 8 + 1]) + LD(&rc[i * 8 + 2]);
        if (t < nb && LD(&rc[t * 8 + 0]) != (double)it) *bad = 2;
        for (int m = 0; m < 3; m++) {
            const float x = LD(&o[((size_t)m * nb + (b + 1) % nb) * 256 + t]);
            if (x != (float)it) *bad = 3;
            v += x * 1e-9f + (float)s * 1e-20f;
        }
    }
    out[(size_t)b * 256 + t] = v;
}

__global__ __launch_bounds__(256, 2) void k_noop(const int *flag, float *out) {
    extern __shared__ float lds[];
    if (*flag < 5) return;
    lds[threadIdx.x] = 1; __syncthreads(); out[threadIdx.x] = lds[255 - threadIdx.x];
}
__global__ __launch_bounds__(256, 2) void k_noop_s(const int *flag, float *out) {
    if (*flag < 5) return;
    out[threadIdx.x] = 1;
}

int main() {
    const int iters = 500;
    float *d; hipMalloc(&d, (size_t)2 * 8 * 512 * 256 * sizeof(float)); hipMemset(d, 0, (size_t)2 * 8 * 512 * 256 * sizeof(float));
    double *rec; hipMalloc(&rec, 2 * 512 * 64); hipMemset(rec, 0, 2 * 512 * 64);
    unsigned *ctr; hipMalloc(&ctr, 4096);
    int *bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    int *flag; hipMalloc(&flag, 4); hipMemset(flag, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms; int hb = 0;
    auto run = [&](const char *name, void *fn, int blocks) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(ctr, 0, 4096); hipMemset(bad, 0, 4);
            int it = iters; void *args[] = {&d, &rec, &ctr, &it, &bad};
            hipEventRecord(a);
            hipError_t e = hipLaunchCooperativeKernel(fn, dim3(blocks), dim3(256), args, 0, 0);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("%-34s %3d blocks: %s  %.2f us per barrier  (bad=%d)\n", name, blocks, hipGetErrorString(e), ms * 1e3 / iters, hb);
        }
    };
    for (int blocks : {400, 256, 79, 512}) {
        run("flat, thread 0 polls, 3 rows", (void *)k_fast<false, 3, false>, blocks);
        run("flat, thread 0 polls, 3 rows, inv", (void *)k_fast<false, 3, true>, blocks);
        run("two-level, sc1 loads", (void *)k_two<false>, blocks);
        run("two-level, buffer_inv + plain loads", (void *)k_two<true>, blocks);
    }
    // no-op launches back to back
    hipFuncSetAttribute((const void *)k_noop, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_noop, dim3(400), dim3(256), 48 * 1024, 0, flag, d);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("no-op launch, 400 x 256, 48 KB LDS : %.2f us each\n", ms * 1e3 / 200);
        hipEventRecord(a);
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_noop_s, dim3(400), dim3(256), 0, 0, flag, d);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("no-op launch, 400 x 256, no LDS    : %.2f us each\n", ms * 1e3 / 200);
        hipEventRecord(a);
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_noop_s, dim3(1), dim3(64), 0, 0, flag, d);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("no-op launch, 1 x 64               : %.2f us each\n", ms * 1e3 / 200);
    }
    return 0;
}
