// Cost of a grid-wide barrier on gfx950 (8 XCDs): cooperative_groups grid.sync() vs a hand-rolled
// two-level (per-slot + global) counter barrier.  Decides whether a persistent CG kernel can pay.
// Build: hipcc --offload-arch=gfx950 -O3 gridsync.hip -o gridsync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256, 2) void k_cg(float *out, int iters) {
    cg::grid_group g = cg::this_grid();
    float v = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        out[blockIdx.x * 256 + threadIdx.x] = v;     // something to publish
        g.sync();
        v += out[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
    }
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

// hand-rolled: monotone counter, one arrival per workgroup, agent-scope release/acquire
__global__ __launch_bounds__(256, 2) void k_flat(float *out, unsigned *ctr, int iters) {
    float v = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        out[blockIdx.x * 256 + threadIdx.x] = v;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * gridDim.x;
            long spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1L << 24)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __threadfence();
        v += out[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
    }
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    const int blocks = 400, iters = 200;
    float *d; hipMalloc(&d, blocks * 256 * sizeof(float));
    unsigned *ctr; hipMalloc(&ctr, 64); hipMemset(ctr, 0, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        int it = iters; void *args[] = {&d, &it};
        hipEventRecord(a);
        hipError_t e = hipLaunchCooperativeKernel((void *)k_cg, dim3(blocks), dim3(256), args, 0, 0);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("cg grid.sync   : %s  %.2f us per barrier (%d blocks)\n", hipGetErrorString(e), ms * 1e3 / iters, blocks);
    }
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(ctr, 0, 64);
        int it = iters; void *args[] = {&d, &ctr, &it};
        hipEventRecord(a);
        hipError_t e = hipLaunchCooperativeKernel((void *)k_flat, dim3(blocks), dim3(256), args, 0, 0);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("flat counter   : %s  %.2f us per barrier\n", hipGetErrorString(e), ms * 1e3 / iters);
    }
    return 0;
}
