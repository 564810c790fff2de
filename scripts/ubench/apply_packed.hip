// The cached-Gram product of the unfused X-solve at config 5's shape (T = 50 000, k = 64, fp64): apply_kernel<false> on
// full k x k Grams (1.64 GB per launch) against apply_kernel<true, 17> on packed upper triangles (0.83 GB), several grids,
// and -- compiled with -DTRMF_APPLY_ABL=1|2 -- the packed kernel without its products / without its LDS copy.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DTRMF_APPLY_ABL=n] apply_packed.hip -o apply_packed
#define TRMF_REAL double
#include "../../exp-trmf-nips16_amd/csrc/cg_kernels.hpp"
#include <cstdio>
#include <cstdlib>
namespace trmf { void set_error(const std::string &) {} }
using namespace trmf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 50000, k = 64, KP = 64;
    XParams p{};
    p.T = T; p.k = k; p.KP = KP; p.NT = 4; p.nlag = 0; p.midx = 0; p.pstride = 4096;
    XState hs{}; hs.stop_it = kCgRunning;
    XState *st; CK(hipMalloc(&st, sizeof(XState))); CK(hipMemcpy(st, &hs, sizeof(XState), hipMemcpyHostToDevice));
    const size_t NV = (size_t)(T + 1) * KP, full = (size_t)k * k, packed = packed_gram_elems(k);
    double *v, *r, *base, *out, *G, *Bv, *P;
    CK(hipMalloc(&v, NV * 8)); CK(hipMalloc(&r, NV * 8)); CK(hipMalloc(&base, NV * 8)); CK(hipMalloc(&out, NV * 8)); CK(hipMalloc(&Bv, NV * 8));
    CK(hipMalloc(&G, (size_t)T * full * 8 + 4096)); CK(hipMalloc(&P, (size_t)P_NSLOTS * p.pstride * 8));
    CK(hipMemset(v, 0, NV * 8)); CK(hipMemset(r, 0, NV * 8)); CK(hipMemset(base, 0, NV * 8)); CK(hipMemset(G, 0, (size_t)T * full * 8));
    const int rpb = 256 / k;
    const size_t lds = (size_t)apply_stages(k) * 512 * 8;
    CK(hipFuncSetAttribute((const void *)apply_kernel<true, 17>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](bool pk, int blocks) {
        p.gstride = pk ? packed : full;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(a));
            for (int it = 0; it < 20; it++) {
                if (pk) hipLaunchKernelGGL((apply_kernel<true, 17>), dim3(blocks), dim3(256), lds, 0, p, st, it, v, r, base, G, Bv, 0, out, 1, P + (size_t)P_DOT * p.pstride, rpb, 0, T, 0);
                else hipLaunchKernelGGL(apply_kernel<false>, dim3(blocks), dim3(256), 0, 0, p, st, it, v, r, base, G, Bv, 0, out, 1, P + (size_t)P_DOT * p.pstride, rpb, 0, T, 0);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        const double bytes = (double)T * (pk ? packed : full) * 8;
        printf("ABL=%d %-6s grid %5d : %7.1f us per launch, %.2f TB/s of Gram bytes\n", TRMF_APPLY_ABL, pk ? "packed" : "full", blocks, ms * 1e3 / 20,
               bytes / (ms / 20 * 1e-3) / 1e12);
    };
    if (!TRMF_APPLY_ABL) run(false, 1024);
    for (int blocks : {256, 512, 1024, 2048, 4096}) run(true, blocks);
    return 0;
}
