// Feasibility of an fp32-accurate Gram on the bf16 matrix pipe: x = x1 + x2 + x3 (three bf16 parts, 8 mantissa bits
// each: an exact split of an fp32 value), products a_i * b_j are exact in fp32, the MFMA accumulates in fp32.
//   (1) layout + accuracy: G = P^T P for a random 16-column panel of K rows via v_mfma_f32_16x16x32_bf16 with
//       6 (i + j <= 4) and 9 split terms, against an fp64 reference and an fp32 fmaf chain;
//   (2) issue cost of the 36 / 54 bf16 MFMAs that replace the 48 f32 MFMAs of two 16-entry iterations (k = 40).
// Build: hipcc --offload-arch=gfx950 -O3 bf16x3.hip -o bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __host__ inline unsigned short bf16_trunc(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
__device__ __host__ inline float bf16_val(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
__device__ __host__ inline void split3(float x, unsigned short (&p)[3]) {
    p[0] = bf16_trunc(x); float r = x - bf16_val(p[0]);
    p[1] = bf16_trunc(r); r = r - bf16_val(p[1]);
    p[2] = bf16_trunc(r);
}

// P: K x 16 row-major (K multiple of 32).  out[row][col] = sum_k P[k][row] P[k][col]
template <int TERMS> __global__ void gram16(const float *P, int K, float *out) {
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    f4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        union { bf8 v; unsigned short s[8]; } part[3];
        for (int e = 0; e < 8; e++) {
            unsigned short p[3];
            split3(P[(size_t)(k0 + 8 * g + e) * 16 + c], p);
            for (int i = 0; i < 3; i++) part[i].s[e] = p[i];
        }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                if (TERMS == 9 || i + j <= 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(part[i].v, part[j].v, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) out[(4 * g + r) * 16 + c] = acc[r];
}

template <int MODE> __global__ __launch_bounds__(256) void rate(float *out, int iters) {
    f4 acc[6];
    for (int t = 0; t < 6; t++) acc[t] = f4{0, 0, 0, 0};
    union { bf8 v; float f[4]; } a, b;
    for (int i = 0; i < 4; i++) { a.f[i] = threadIdx.x * 0.25f + i; b.f[i] = threadIdx.x * 0.5f - i; }
    float fa = threadIdx.x * 0.25f, fb = threadIdx.x * 0.5f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {            // 32 entries, f32 pipe: 8 groups x 6 tiles
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int t = 0; t < 6; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[t], 0, 0, 0);
        } else {                    // 32 entries, bf16 pipe: MODE split terms x 6 tiles
#pragma unroll
            for (int u = 0; u < MODE; u++)
#pragma unroll
                for (int t = 0; t < 6; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc[t], 0, 0, 0);
        }
        asm volatile("" : "+v"(fa), "+v"(fb));
    }
    float s = 0;
    for (int t = 0; t < 6; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
// 16 entries on the K=16 form: 36 x v_mfma_f32_16x16x16_bf16
__global__ __launch_bounds__(256) void rate16(float *out, int iters) {
    f4 acc[6];
    for (int t = 0; t < 6; t++) acc[t] = f4{0, 0, 0, 0};
    union { s4 v; float f[2]; } a, b;
    for (int i = 0; i < 2; i++) { a.f[i] = threadIdx.x * 0.25f + i; b.f[i] = threadIdx.x * 0.5f - i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 6; u++)
#pragma unroll
            for (int t = 0; t < 6; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.v, b.v, acc[t], 0, 0, 0);
        asm volatile("" : "+v"(a.f[0]), "+v"(b.f[0]));
    }
    float s = 0;
    for (int t = 0; t < 6; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
float run16(int bpc, int iters) {
    int blocks = 256 * bpc; float *d; (void)hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    rate16<<<blocks, 256>>>(d, 10);
    (void)hipEventRecord(a); rate16<<<blocks, 256>>>(d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); (void)hipFree(d); return ms;
}
template <int MODE> float run(int bpc, int iters) {
    int blocks = 256 * bpc; float *d; (void)hipMalloc(&d, blocks * 256 * sizeof(float));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    rate<MODE><<<blocks, 256>>>(d, 10);
    (void)hipEventRecord(a); rate<MODE><<<blocks, 256>>>(d, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); (void)hipFree(d); return ms;
}

int main() {
    const int K = 1024;
    std::vector<float> P(K * 16);
    srand(1);
    for (auto &x : P) x = (float)rand() / RAND_MAX * 2.0f - 0.7f;
    float *dP, *dO; (void)hipMalloc(&dP, P.size() * 4); (void)hipMalloc(&dO, 256 * 4);
    (void)hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(256, 0); std::vector<float> chain(256, 0);
    for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) {
        double s = 0; float f = 0;
        for (int k = 0; k < K; k++) { s += (double)P[k * 16 + r] * P[k * 16 + c]; f = fmaf(P[k * 16 + r], P[k * 16 + c], f); }
        ref[r * 16 + c] = s; chain[r * 16 + c] = f;
    }
    float h[256];
    auto report = [&](const char *name, const float *v) {
        double worst = 0, scale = 0;
        for (int i = 0; i < 256; i++) { worst = fmax(worst, fabs(v[i] - ref[i])); scale = fmax(scale, fabs(ref[i])); }
        printf("%-28s max |err| / max |G| = %.3e\n", name, worst / scale);
    };
    report("fp32 fmaf chain (reference)", chain.data());
    gram16<6><<<1, 64>>>(dP, K, dO); (void)hipMemcpy(h, dO, sizeof h, hipMemcpyDeviceToHost); report("bf16 x 3, 6 split terms", h);
    gram16<9><<<1, 64>>>(dP, K, dO); (void)hipMemcpy(h, dO, sizeof h, hipMemcpyDeviceToHost); report("bf16 x 3, 9 split terms", h);
    const int iters = 2000;
    for (int bpc : {1, 2, 3}) {
        const double cyc = 1e-3 * 2.4e9 / ((double)iters * bpc);
        printf("waves/SIMD=%d, 32 entries of a k=40 Gram: f32 pipe (48 MFMA) %.0f cycles | bf16 pipe 6 terms (36 MFMA) %.0f | 9 terms (54 MFMA) %.0f\n",
               bpc, run<0>(bpc, iters) * cyc, run<6>(bpc, iters) * cyc, run<9>(bpc, iters) * cyc);
        printf("              16 entries on v_mfma_f32_16x16x16_bf16, 6 terms (36 MFMA): %.0f cycles\n", run16(bpc, iters) * cyc);
    }
    return 0;
}
