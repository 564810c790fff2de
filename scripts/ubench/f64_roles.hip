// f64_roles.hip -- VERDICT r4 item 8: would a two-role workgroup (some wavefronts accumulate Grams on the fp64 matrix pipe,
// the others factorise on the fp64 vector ALUs, "separate issue budgets across SIMDs") beat one-role-per-wavefront?
// The same total work per workgroup (4 wavefronts = one per SIMD; NM MFMAs + NV FMAs per wavefront and step) is issued
//   (a) mixed:  every wavefront issues its NM MFMAs and its NV FMAs                (what fsolve_mfma_kernel does)
//   (b) roles:  wavefronts 0,1 issue 2*NM MFMAs each, wavefronts 2,3 issue 2*NV FMAs each
// at 1, 2 and 3 workgroups per CU.  A SIMD issues ONE instruction stream per cycle whatever its kind (r03_f64_pipe_ubench:
// MFMA and FMA of one SIMD do not overlap), so (b) can only win if the per-SIMD loads happen to balance -- and then it equals (a).
// Build: hipcc --offload-arch=gfx950 -O3 f64_roles.hip -o f64_roles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, bool ROLES> __global__ __launch_bounds__(256) void k(double *out, int iters) {
    d4 acc[2 * NM];
    for (int t = 0; t < 2 * NM; t++) acc[t] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 0.25, b = threadIdx.x * 0.5, v[2 * NV];
    for (int u = 0; u < 2 * NV; u++) v[u] = (double)u;
    const int wave = threadIdx.x >> 6;
    const bool do_m = !ROLES || wave < 2, do_v = !ROLES || wave >= 2;
    constexpr int M = ROLES ? 2 * NM : NM, V = ROLES ? 2 * NV : NV;
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int t = 0; t < M; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int u = 0; u < V; u++) v[u] = fma(v[u], 1.0001, a);
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    double s = 0;
    for (int t = 0; t < 2 * NM; t++) s += acc[t][0];
    for (int u = 0; u < 2 * NV; u++) s += v[u];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM, int NV, bool ROLES> float run(int bpc, int iters) {
    int blocks = 256 * bpc;
    double *d; (void)hipMalloc(&d, blocks * 256 * sizeof(double));
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<NM, NV, ROLES><<<blocks, 256>>>(d, 10);
    (void)hipEventRecord(a);
    k<NM, NV, ROLES><<<blocks, 256>>>(d, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(d);
    return ms;
}
template <int NM, int NV> void both() {
    const int iters = 4000;
    for (int bpc : {1, 2, 3}) {
        const float m = run<NM, NV, false>(bpc, iters), r = run<NM, NV, true>(bpc, iters);
        // cycles per step and SIMD at 2.4 GHz, per workgroup resident on the CU
        printf("NM=%2d NV=%2d  %d workgroup(s)/CU: mixed %8.1f cycles/step   roles %8.1f cycles/step   (roles / mixed = %.2f)\n", NM, NV, bpc,
               m * 1e-3 * 2.4e9 / iters / bpc, r * 1e-3 * 2.4e9 / iters / bpc, r / m);
    }
}
int main() {
    printf("per wavefront and step: NM fp64 MFMAs (16x16x4) + NV fp64 FMAs; roles: waves 0,1 -> 2 NM MFMAs, waves 2,3 -> 2 NV FMAs\n");
    // (NV <= 24: the roles form carries 2 NV doubles per lane; larger counts spill and measure the spills)
    both<6, 12>();      // MFMA-heavy (the Gram phase)
    both<4, 24>();      // the factorisation's own mix, ~2 : 1 in cycles (8k MFMA cycles to 4.4k FMA cycles per system)
    both<2, 24>();      // ~balanced in cycles (100 per MFMA, ~3.2-8 per FMA)
    both<1, 24>();      // FMA-heavy (the substitutions)
    return 0;
}
