// Issue cost of v_pk_fma_f32 against v_fma_f32 on gfx950 (is packed f32 really two FMAs per issue slot?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters) {
    f2 v[12]; float a = threadIdx.x * 0.25f;
    for (int u = 0; u < 12; u++) v[u] = f2{(float)u, (float)u + 0.5f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 12; u++) {
            if (MODE == 0) { v[u].x = fmaf(v[u].x, 1.0001f, a); }                       // 12 scalar FMAs
            else if (MODE == 1) { v[u].x = fmaf(v[u].x, 1.0001f, a); v[u].y = fmaf(v[u].y, 1.0001f, a); }   // 24 scalar
            else { const f2 m = {1.0001f, 1.0001f}, c = {a, a}; v[u] = __builtin_elementwise_fma(v[u], m, c); }  // 12 packed
        }
        asm volatile("" : "+v"(a));
    }
    float s = 0; for (int u = 0; u < 12; u++) s += v[u].x + v[u].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> float run(int bpc, int iters) {
    int blocks = 256 * bpc; float *d; (void)hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(d, 10);
    (void)hipEventRecord(a); k<MODE><<<blocks, 256>>>(d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); (void)hipFree(d); return ms;
}
int main() {
    const int iters = 4000;
    for (int bpc : {1, 4}) {
        double cyc = 1e-3 * 2.16e9 / ((double)iters * bpc);
        printf("waves/SIMD=%d: 12 v_fma %.1f cyc/step, 24 v_fma %.1f, 12 v_pk_fma %.1f\n", bpc, run<0>(bpc, iters) * cyc, run<1>(bpc, iters) * cyc, run<2>(bpc, iters) * cyc);
    }
    return 0;
}
