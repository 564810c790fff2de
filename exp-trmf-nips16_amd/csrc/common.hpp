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

// Column-interleaved factor layout.  Every T x KP / n x KP matrix in HBM stores logical column t at
// position colpos(t): the NT = KP/16 columns {c, 16+c, 32+c, ...} that one MFMA lane needs sit next to
// each other, so a lane fetches its operand slices of a gathered row with ONE vector load.
__host__ __device__ constexpr int colpos(int t, int NT) { return NT * (t & 15) + (t >> 4); }
__host__ __device__ constexpr int collog(int p, int NT) { return kTile * (p % NT) + p / NT; }

// aligned packs for vector LDS access (ds_read_b64 / ds_read_b128)
template <typename T> struct alignas(16) Quad { T v[4]; };
template <typename T, int N> struct alignas((N * sizeof(T)) % 16 == 0 ? 16 : 8) VecOf { T v[N]; };

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
