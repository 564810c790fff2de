// common.hpp -- shared definitions for the gfx950 TRMF kernels and the host session.
//
// Element type: the library is compiled twice, -DTRMF_REAL=float and -DTRMF_REAL=double, like the
// reference's corelib Makefile (python/trmf/corelib/Makefile:32-33).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#ifndef TRMF_REAL
#define TRMF_REAL float
#endif

namespace trmf {

using real = TRMF_REAL;

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kTile = 16;          // MFMA 16x16x4 tile edge
constexpr int kMaxRank = 64;       // k <= 64: one wavefront lane per factor column in the solve
constexpr int kMaxPartials = 1024; // entries of every per-block partial-sum array

// Padded leading dimension of every factor / CG vector in HBM: k rounded up to the MFMA tile.
__host__ __device__ constexpr int padded_rank(int k) { return ((k + kTile - 1) / kTile) * kTile; }

#define TRMF_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::trmf::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));             \
            return ::trmf::kFail;                                                             \
        }                                                                                     \
    } while (0)

constexpr int kFail = -1;

void set_error(const std::string &msg);   // defined in trmf_abi.hip

}  // namespace trmf
