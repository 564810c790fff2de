// Throughput of cross-lane primitives on gfx950 (per CU), to choose the broadcast mechanism of the
// in-register Cholesky.  Build: hipcc --offload-arch=gfx950 -O3 xlane.hip -o xlane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters) {
    float v = threadIdx.x * 0.5f, acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int it = 0; it < iters; it++) {
#define STEP(u, ACC)                                                                                   \
        {                                                                                              \
            float b;                                                                                   \
            if (MODE == 0) b = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), ((u) << 5) | 0x10)); \
            else if (MODE == 1) b = __int_as_float(__builtin_amdgcn_ds_bpermute(((threadIdx.x & 48) | (u)) << 2, __float_as_int(v))); \
            else if (MODE == 2) b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (u))); \
            else if (MODE == 3) b = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55 * ((u) & 3), 0xf, 0xf, true)); \
            else if (MODE == 4) b = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true)); \
            else b = v;                                                                                \
            ACC = fmaf(b, v, ACC);                                                                     \
        }
        STEP(0, acc0) STEP(1, acc1) STEP(2, acc2) STEP(3, acc3) STEP(4, acc0) STEP(5, acc1) STEP(6, acc2) STEP(7, acc3)
        STEP(8, acc0) STEP(9, acc1) STEP(10, acc2) STEP(11, acc3) STEP(12, acc0) STEP(13, acc1) STEP(14, acc2) STEP(15, acc3)
        v += 1e-9f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}

template <int MODE> void run(const char *name, int blocks_per_cu) {
    int iters = 2000, blocks = 256 * blocks_per_cu;
    float *d; hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(d, 10);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops_per_cu = (double)blocks_per_cu * 4 /*waves*/ * iters * 16;   // wave-instructions per CU
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s blocks/CU=%d  %.3f ms  ->  %.2f cycles per wave-instr per CU (@2.4GHz), %.2f per SIMD\n", name, blocks_per_cu, ms, cyc / ops_per_cu, 4 * cyc / ops_per_cu);
    hipFree(d);
}

int main() {
    for (int bpc : {1, 3}) {
        run<0>("ds_swizzle", bpc); run<1>("ds_bpermute", bpc); run<2>("v_readlane", bpc);
        run<3>("dpp quad_perm", bpc); run<4>("dpp row_mirror", bpc); run<5>("fma only", bpc);
    }
    return 0;
}
