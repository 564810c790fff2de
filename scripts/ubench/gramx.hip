// The X-side Gram build (gram_x_kernel<3, true, false>, fp32) at config 3's shape on a synthetic CSC: kernel time against the
// number of resident wavefronts per SIMD (capped through dynamic LDS: the kernel itself uses none), with every gathered row
// inside a window of factor rows (argument 4: cache-resident gather) and with a zero / all-ones factor (argument 5: the
// data-dependent power draw of the MFMAs).  Results and what was tried on top of it: profiles/r04_gramx_ubench.txt.
// usage: gramx [T n density window hmode]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form gramx.hip -o gramx
#define TRMF_REAL float
#define TRMF_F32 1
#include "../../exp-trmf-nips16_amd/csrc/gram_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <cstring>
namespace trmf { void set_error(const std::string &) {} }
using namespace trmf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 10000, n = argc > 2 ? atoi(argv[2]) : 100000, k = 40, KP = 48;
    const double dens = argc > 3 ? atof(argv[3]) : 0.01;
    std::mt19937_64 rng(1);
    std::vector<uint32_t> ptr(T + 1, 0), idx;
    std::vector<float> val;
    std::geometric_distribution<long> gap(dens);
    for (int t = 0; t < T; t++) {
        for (long i = gap(rng); i < n; i += 1 + gap(rng)) { idx.push_back((uint32_t)i); val.push_back((float)(rng() % 1000) * 1e-3f); }
        ptr[t + 1] = (uint32_t)idx.size();
    }
    const size_t nnz = idx.size();
    const int window = argc > 4 ? atoi(argv[4]) : 0;          // > 0: every gathered row inside a window of that many factor rows (cache-resident gather)
    if (window > 0) for (auto &j : idx) j %= (uint32_t)window;
    std::vector<float> H((size_t)(n + 1) * KP, 0.f);
    const int hmode = argc > 5 ? atoi(argv[5]) : 0;           // 1: the factor all zero, 2: all ones (data-dependent power draw of the MFMAs)
    if (hmode == 2) std::fill(H.begin(), H.end(), 1.0f);
    if (hmode == 0) for (int i = 0; i < n; i++) for (int c = 0; c < k; c++) H[(size_t)i * KP + colpos(c, 3)] = (float)((rng() % 2001) - 1000.0) * 1e-3f;
    uint32_t *dptr, *didx; float *dval, *dH, *dG, *dB;
    CK(hipMalloc(&dptr, (T + 1) * 4)); CK(hipMalloc(&didx, nnz * 4 + 64)); CK(hipMalloc(&dval, nnz * 4 + 64));
    CK(hipMalloc(&dH, H.size() * 4)); CK(hipMalloc(&dG, (size_t)T * k * k * 4)); CK(hipMalloc(&dB, (size_t)T * KP * 4));
    CK(hipMemcpy(dptr, ptr.data(), (T + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(didx, idx.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dval, val.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice));
    auto kern = gram_x_kernel<3, true, false>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double BG = (double)nnz * (4 + 4 + k * 4) + (double)T * (k * k + k) * 4;
    printf("T=%d n=%d nnz=%zu  B_G=%.3f GB\n", T, n, nnz, BG * 1e-9);
    std::vector<float> ref((size_t)T * k * k), got((size_t)T * k * k);
    const int ldss[] = {0, 32 * 1024, 80 * 1024, 160 * 1024};
    for (int lds : ldss) {
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 12; rep++) {
            float ms;
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3((T + 3) / 4), dim3(256), lds, 0, dptr, didx, dval, dH, dG, dB, 0u, (uint32_t)T, k, (uint32_t)n, (size_t)k * k);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            if (rep >= 2) { best = std::min(best, ms); sum += ms; }
        }
        CK(hipMemcpy(got.data(), dG, got.size() * 4, hipMemcpyDeviceToHost));
        if (lds == 0) ref = got;
        const bool same = memcmp(ref.data(), got.data(), got.size() * 4) == 0;
        const int wgs = lds ? std::min(8, 160 * 1024 / lds) : 8;
        printf("dynamic LDS %6d B (<= %d wavefronts per SIMD): avg %.1f us  min %.1f us  -> %.2f TB/s of B_G  %s\n", lds, wgs,
               1e3 * sum / 10, 1e3 * best, BG / (sum / 10 * 1e-3) * 1e-12, same ? "bit-identical" : "DIFFERENT");
    }
    return 0;
}
