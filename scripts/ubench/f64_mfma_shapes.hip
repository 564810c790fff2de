// f64_mfma_shapes.hip -- issue cost of the two fp64 MFMA shapes of gfx950 on one SIMD: v_mfma_f64_16x16x4_f64 (2048 flop) and
// v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4: 512 flop).  The fp64 F-solve's Gram accumulation uses the first at ~100 cycles per
// instruction = 20 flop/clk/SIMD (profiles/r03_f64_pipe_ubench.txt), below the fp64 vector ALU's 26-40; would the small shape
// reach the pipe's nominal rate?  Build: hipcc --offload-arch=gfx950 -O3 f64_mfma_shapes.hip -o f64_mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// SHAPE 16: the large shape; 4: the small shape, one A / B register pair for every instruction; 5: the small shape with 8 different
// A and 4 different B registers in rotation (what a Gram accumulation issues: operand reads from many registers)
template <int SHAPE, int N, bool RANDOM_OPERANDS = false> __global__ __launch_bounds__(256) void k(double *out, long long *clk, int iters) {
    d4 acc16[N]; double acc4[N];
    for (int t = 0; t < N; t++) { acc16[t] = d4{0, 0, 0, 0}; acc4[t] = 0; }
    double a = threadIdx.x * 0.25, b = threadIdx.x * 0.5;
    double av[8], bv[4];
    for (int q = 0; q < 8; q++) av[q] = a + q;
    for (int q = 0; q < 4; q++) bv[q] = b - q;
    if (RANDOM_OPERANDS) {              // full-entropy mantissas in [-1, 1): what real factors look like (the clock is data-dependent)
        unsigned long long h = 0x9e3779b97f4a7c15ull * (threadIdx.x + 256ull * blockIdx.x + 1);
        auto rnd = [&]() { h ^= h << 13; h ^= h >> 7; h ^= h << 17; return (double)(long long)h * (1.0 / 9223372036854775808.0); };
        a = rnd(); b = rnd();
        for (int q = 0; q < 8; q++) av[q] = rnd();
        for (int q = 0; q < 4; q++) bv[q] = rnd();
    }
    const long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < N; t++) {
            if (SHAPE == 16) acc16[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[t], 0, 0, 0);
            else if (SHAPE == 17) acc16[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t % 4], av[4 + (t / 4) % 4], acc16[t], 0, 0, 0);   // operands in rotation
            else if (SHAPE == 4) acc4[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[t], 0, 0, 0);
            else acc4[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[t % 8], bv[(t / 8) % 4], acc4[t], 0, 0, 0);
        }
        asm volatile("" : "+v"(a), "+v"(b));
#pragma unroll
        for (int q = 0; q < 8; q++) asm volatile("" : "+v"(av[q]));
#pragma unroll
        for (int q = 0; q < 4; q++) asm volatile("" : "+v"(bv[q]));
    }
    const long long c1 = clock64();
    double s = 0;
    for (int t = 0; t < N; t++) s += acc16[t][0] + acc4[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int SHAPE, int N, bool RND = false> void run(int bpc) {
    const int blocks = 256 * bpc, iters = 4000;
    double *d; long long *c; (void)hipMalloc(&d, blocks * 256 * sizeof(double)); (void)hipMalloc(&c, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<SHAPE, N, RND><<<blocks, 256>>>(d, c, 10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<SHAPE, N, RND><<<blocks, 256>>>(d, c, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double flop = (SHAPE == 16 || SHAPE == 17) ? 2048.0 : 512.0;   // per instruction
    const double wave_cycles = (double)h / ((double)iters * N);         // one wavefront's own clock per instruction it issued
    const double tflops = flop * iters * N * (blocks * 4.0) / (ms * 1e-3) / 1e12;      // WALL clock over the whole grid: what counts
    if (RND) printf("[random operands] ");
    printf("v_mfma_f64_%s, %2d independent accumulators, %d workgroup(s)/CU: %6.1f cycles per instruction in one wavefront; whole grid %6.1f TFLOP/s by wall clock = %5.1f flop/clk/SIMD at 2.4 GHz\n",
           SHAPE == 16 ? "16x16x4   " : SHAPE == 17 ? "16x16x4 (operands in rotation)" : SHAPE == 4 ? "4x4x4_4b  " : "4x4x4_4b (operands in rotation)", N, bpc, wave_cycles, tflops, tflops * 1e12 / (1024 * 2.4e9));
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    for (int bpc : {1, 2, 4}) { run<16, 6>(bpc); run<4, 6>(bpc); run<4, 16>(bpc); run<5, 40>(bpc); run<16, 6, true>(bpc); run<5, 40, true>(bpc); run<16, 10, true>(bpc); run<16, 16, true>(bpc); run<17, 10, true>(bpc); }
    return 0;
}
