// f64_mfma_shapes.hip -- issue cost of the two fp64 MFMA shapes of gfx950 on one SIMD: v_mfma_f64_16x16x4_f64 (2048 flop) and
// v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4: 512 flop).  The fp64 F-solve's Gram accumulation uses the first at ~100 cycles per
// instruction = 20 flop/clk/SIMD (profiles/r03_f64_pipe_ubench.txt), below the fp64 vector ALU's 26-40; would the small shape
// reach the pipe's nominal rate?  Build: hipcc --offload-arch=gfx950 -O3 f64_mfma_shapes.hip -o f64_mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int N> __global__ __launch_bounds__(256) void k(double *out, long long *clk, int iters) {
    d4 acc16[N]; double acc4[N];
    for (int t = 0; t < N; t++) { acc16[t] = d4{0, 0, 0, 0}; acc4[t] = 0; }
    double a = threadIdx.x * 0.25, b = threadIdx.x * 0.5;
    const long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < N; t++) {
            if (SHAPE == 16) acc16[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[t], 0, 0, 0);
            else acc4[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[t], 0, 0, 0);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    const long long c1 = clock64();
    double s = 0;
    for (int t = 0; t < N; t++) s += acc16[t][0] + acc4[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int SHAPE, int N> void run(int bpc) {
    const int blocks = 256 * bpc, iters = 4000;
    double *d; long long *c; (void)hipMalloc(&d, blocks * 256 * sizeof(double)); (void)hipMalloc(&c, 8);
    k<SHAPE, N><<<blocks, 256>>>(d, c, 10); (void)hipDeviceSynchronize();
    k<SHAPE, N><<<blocks, 256>>>(d, c, iters); (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / ((double)iters * N * bpc);          // shader cycles per instruction at the SIMD (bpc wavefronts share it)
    const double flop = SHAPE == 16 ? 2048.0 : 512.0;
    printf("v_mfma_f64_%s, %d independent accumulators, %d wavefront(s)/SIMD: %6.1f cycles per instruction = %5.1f flop/clk/SIMD = %5.1f TFLOP/s on 1024 SIMDs at 2.4 GHz\n",
           SHAPE == 16 ? "16x16x4   " : "4x4x4_4b  ", N, bpc, per, flop / per, flop / per * 1024 * 2.4e9 / 1e12);
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    for (int bpc : {1, 2, 4}) { run<16, 6>(bpc); run<4, 6>(bpc); run<4, 16>(bpc); }
    return 0;
}
