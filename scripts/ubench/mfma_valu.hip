// Do fp32 MFMA (v_mfma_f32_16x16x4_f32) and VALU instructions of DIFFERENT or the SAME wavefront overlap on a
// gfx950 SIMD?  Decides whether the Gram kernels are bound by MFMA time or by MFMA + VALU issue time.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: 6 MFMA per step; 1: NV VALU FMAs per step; 2: both interleaved in one wave
template <int MODE, int NV> __global__ __launch_bounds__(256) void k(float *out, int iters) {
    f4 acc[6];
    for (int t = 0; t < 6; t++) acc[t] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 0.25f, b = threadIdx.x * 0.5f;
    float v[NV];
    for (int u = 0; u < NV; u++) v[u] = (float)u;
    for (int it = 0; it < iters; it++) {
        if (MODE != 1) {
#pragma unroll
            for (int t = 0; t < 6; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int u = 0; u < NV; u++) v[u] = fmaf(v[u], 1.0001f, a);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0;
    for (int t = 0; t < 6; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int u = 0; u < NV; u++) s += v[u];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NV> float run(int blocks_per_cu, int iters) {
    int blocks = 256 * blocks_per_cu;
    float *d; hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, NV><<<blocks, 256>>>(d, 10);
    hipEventRecord(a);
    k<MODE, NV><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    return ms;
}

template <int NV> void report(int bpc) {
    const int iters = 4000;
    float m = run<0, NV>(bpc, iters), v = run<1, NV>(bpc, iters), both = run<2, NV>(bpc, iters);
    double cyc = 1e-3 * 2.4e9 / ((double)iters * bpc);    // cycles per step per wave-slot on a SIMD
    printf("waves/SIMD=%d  NV=%2d : mfma %.1f  valu %.1f  both %.1f cycles per step (sum %.1f, max %.1f)\n", bpc, NV,
           m * cyc, v * cyc, both * cyc, (m + v) * cyc, (m > v ? m : v) * cyc);
}

int main() {
    for (int bpc : {1, 2, 4}) { report<12>(bpc); report<24>(bpc); report<48>(bpc); }
    return 0;
}
