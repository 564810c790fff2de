// fp64 on a gfx950 SIMD: issue cost of v_mfma_f64_16x16x4_f64, of v_fma_f64, and whether the two overlap (same wave /
// different waves of one SIMD).  Also the accuracy of v_rcp_f64 with 0 / 1 / 2 Newton steps.  Decides how the rank-64
// fp64 F-solve (config 5) should split its work between the matrix pipe and the vector ALUs.
// Build: hipcc --offload-arch=gfx950 -O3 f64_pipe.hip -o f64_pipe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int NM, int NV> __global__ __launch_bounds__(256) void k(double *out, int iters) {
    d4 acc[NM];
    for (int t = 0; t < NM; t++) acc[t] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 0.25, b = threadIdx.x * 0.5;
    double v[NV];
    for (int u = 0; u < NV; u++) v[u] = (double)u;
    for (int it = 0; it < iters; it++) {
        if (MODE != 1) {
#pragma unroll
            for (int t = 0; t < NM; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int u = 0; u < NV; u++) v[u] = fma(v[u], 1.0001, a);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    double s = 0;
    for (int t = 0; t < NM; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int u = 0; u < NV; u++) s += v[u];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NM, int NV> float run(int blocks_per_cu, int iters) {
    int blocks = 256 * blocks_per_cu;
    double *d; hipMalloc(&d, blocks * 256 * sizeof(double));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, NM, NV><<<blocks, 256>>>(d, 10);
    hipEventRecord(a);
    k<MODE, NM, NV><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    return ms;
}

template <int MODE> __global__ __launch_bounds__(256) void kclk(double *out, long long *clk, int iters) {
    d4 acc[6];
    for (int t = 0; t < 6; t++) acc[t] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 0.25, b = threadIdx.x * 0.5, v[24];
    for (int u = 0; u < 24; u++) v[u] = (double)u;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 6; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 24; u++) v[u] = fma(v[u], 1.0001, a);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int t = 0; t < 6; t++) s += acc[t][0];
    for (int u = 0; u < 24; u++) s += v[u];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int MODE> void clock_probe(int bpc) {
    int blocks = 256 * bpc, iters = 4000;
    double *d; long long *c; hipMalloc(&d, blocks * 256 * sizeof(double)); hipMalloc(&c, 16);
    kclk<MODE><<<blocks, 256>>>(d, c, iters); hipDeviceSynchronize();
    kclk<MODE><<<blocks, 256>>>(d, c, iters); hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    printf("%s load, %d waves/SIMD: %lld shader ticks in %lld wall ticks (100 MHz) -> %.3f GHz; %.1f shader cycles per %s\n",
           MODE == 0 ? "f64 MFMA" : "f64 FMA", bpc, h[0], h[1], 0.1 * (double)h[0] / (double)h[1],
           (double)h[0] / ((double)iters * (MODE == 0 ? 6 : 24) * bpc), MODE == 0 ? "MFMA (SIMD level)" : "FMA (SIMD level)");
    hipFree(d); hipFree(c);
}

static double g_ghz = 2.4;
template <int NM, int NV> void report(int bpc) {
    const int iters = 2000;
    float m = run<0, NM, NV>(bpc, iters), v = run<1, NM, NV>(bpc, iters), both = run<2, NM, NV>(bpc, iters);
    double cyc = 1e-3 * g_ghz * 1e9 / ((double)iters * bpc);    // cycles per step per wave-slot on a SIMD
    printf("waves/SIMD=%d  MFMA=%d VALU=%2d : mfma %.1f (%.1f/MFMA)  valu %.1f (%.1f/FMA)  both %.1f cycles per step (sum %.1f, max %.1f)\n",
           bpc, NM, NV, m * cyc, m * cyc / NM, v * cyc, v * cyc / NV, both * cyc, (m + v) * cyc, (m > v ? m : v) * cyc);
}

__global__ void rcp_k(const double *x, double *r0, double *r1, double *r2, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    r0[i] = r;
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    r1[i] = r;
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    r2[i] = r;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    g_ghz = prop.clockRate * 1e-6;
    printf("clock %.2f GHz (nominal), %d CUs\n", g_ghz, prop.multiProcessorCount);
    for (int bpc : {1, 2}) { report<6, 12>(bpc); report<6, 48>(bpc); report<6, 96>(bpc); }
    report<10, 24>(1);
    for (int bpc : {3, 4, 8}) report<6, 24>(bpc);
    // real shader clock under each load: s_memtime ticks (shader cycles) against the 100 MHz wall clock
    clock_probe<0>(2); clock_probe<1>(2);
    const int n = 1 << 20;
    std::vector<double> x(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    rcp_k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    std::vector<double> r0(n), r1(n), r2(n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        long double t = 1.0L / (long double)x[i];
        e0 = std::fmax(e0, (double)fabsl(((long double)r0[i] - t) / t));
        e1 = std::fmax(e1, (double)fabsl(((long double)r1[i] - t) / t));
        e2 = std::fmax(e2, (double)fabsl(((long double)r2[i] - t) / t));
    }
    printf("v_rcp_f64 max relative error: raw %.3e  +1 Newton %.3e  +2 Newton %.3e  (eps = 1.11e-16)\n", e0, e1, e2);
    return 0;
}
