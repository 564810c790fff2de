// Real shader clock and cycles per v_mfma_f32_16x16x4_f32 under a sustained fp32 MFMA load, by operand data (zeros / ones /
// random): s_memtime ticks (shader cycles) against the 100 MHz wall clock, over ~1 ms.  The Gram kernels' MFMA floor is priced on this.
// Build: hipcc --offload-arch=gfx950 -O3 f32_clock.hip -o f32_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const float *in, float *out, long long *stamps, int iters) {
    f4 acc[6];
    for (int t = 0; t < 6; t++) acc[t] = f4{0, 0, 0, 0};
    float a[4], b[4];
    for (int u = 0; u < 4; u++) { a[u] = in[(threadIdx.x * 8 + u) & 4095]; b[u] = in[(threadIdx.x * 8 + 4 + u) & 4095]; }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int t = 0; t < 6; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], (t & 1) ? b[u] : a[(u + t) & 3], acc[t], 0, 0, 0);
            asm volatile("" : "+v"(a[u]), "+v"(b[u]));
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int t = 0; t < 6; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = w1 - w0; }
}
int main() {
    float *in, *out; long long *st; float h[4096];
    hipMalloc(&in, sizeof(h)); hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&st, 16);
    for (int mode = 0; mode < 3; mode++)
        for (int bpc : {1, 2, 8}) {
            for (int i = 0; i < 4096; i++) h[i] = mode == 0 ? 0.f : mode == 1 ? 1.f : (float)((rand() % 2001) - 1000) * 1e-3f;
            hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
            const int iters = 40000 / bpc;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                k<<<256 * bpc, 256>>>(in, out, st, iters);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            long long s[2]; hipMemcpy(s, st, 16, hipMemcpyDeviceToHost);
            const double ghz = (double)s[0] / ((double)s[1] * 10.0);
            printf("data %s, %d waves/SIMD: kernel %.3f ms; shader clock %.3f GHz; %.1f shader cycles per MFMA (SIMD level), %.1f ns\n",
                   mode == 0 ? "zeros " : mode == 1 ? "ones  " : "random", bpc, ms, ghz,
                   (double)s[0] / ((double)iters * 24 * bpc), (double)s[1] * 10.0 / ((double)iters * 24 * bpc));
        }
    return 0;
}
