// v_mfma_f32_4x4x1_16b_f32: (1) operand / result layout, (2) issue cost next to plain v_fma_f32.
// Expected layout (16 independent 4x4 blocks, K = 1): lane l -> block l/4; A operand of lane 4b+i = A_b[i];
// B operand of lane 4b+j = B_b[j]; result VGPR i of lane 4b+j = D_b[i][j] = C + A_b[i] * B_b[j].
// Build: hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout(float *out) {
    const int l = threadIdx.x;
    const float a = 1.0f + l, b = 100.0f + 3.0f * l;
    f4 c = {0.5f, 0.5f, 0.5f, 0.5f};
    f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) out[l * 4 + i] = d[i];
}

template <int MODE> __global__ __launch_bounds__(256) void rate(float *out, int iters) {
    f4 acc[8];
    for (int t = 0; t < 8; t++) acc[t] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 0.25f, b = threadIdx.x * 0.5f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[t][i] = fmaf(a, b, acc[t][i]);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0;
    for (int t = 0; t < 8; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> float run(int bpc, int iters) {
    int blocks = 256 * bpc;
    float *d; (void)hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    rate<MODE><<<blocks, 256>>>(d, 10);
    (void)hipEventRecord(a);
    rate<MODE><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(d);
    return ms;
}

int main() {
    float *d, h[256];
    (void)hipMalloc(&d, sizeof h);
    layout<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            const int blk = l / 4, j = l % 4;
            const float expect = 0.5f + (1.0f + 4 * blk + i) * (100.0f + 3.0f * (4 * blk + j));
            if (fabsf(h[l * 4 + i] - expect) > 1e-3f) bad++;
        }
    printf("layout D[vgpr i][lane 4b+j] = C + A[lane 4b+i] * B[lane 4b+j]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) for (int l = 0; l < 8; l++) printf("lane %d: %g %g %g %g\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    const int iters = 4000;
    for (int bpc : {1, 2, 4}) {
        const double cyc = 1e-3 * 2.4e9 / ((double)iters * bpc);
        printf("waves/SIMD=%d: 8 x mfma_4x4x1 = %.1f cycles ; the same 32 FMAs as v_fma_f32 = %.1f cycles (per wave, at 2.4 GHz)\n", bpc,
               run<0>(bpc, iters) * cyc, run<1>(bpc, iters) * cyc);
    }
    return 0;
}
