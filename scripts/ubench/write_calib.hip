// WRITE_SIZE calibration: known byte counts written in the access patterns the F-solve epilogue uses.
//   mode 0: 16 B per lane, fully coalesced (global_store_dwordx4)            -- the reference pattern
//   mode 1: 12 B per lane, lanes contiguous (global_store_dwordx3): a 16-lane row writes 192 contiguous bytes
//   mode 2: 4 B per lane, three dword stores 64 B apart per 16-lane row (the non-interleaved layout)
// Every mode writes rows*192 bytes (rows = 100000 -> 19.2 MB).  Run under
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -- ./write_calib     (and a second pass with FETCH_SIZE)
// Build: hipcc --offload-arch=gfx950 -O3 write_calib.hip -o write_calib
#include <hip/hip_runtime.h>
#include <cstdio>

struct V3 { float v[3]; };

template <int MODE> __global__ __launch_bounds__(256) void wr(float *out, int rows) {
    const int gl = blockIdx.x * 256 + threadIdx.x;          // one 16-lane group per row
    const int row = gl >> 4, c = gl & 15;
    if (row >= rows) return;
    float *p = out + (size_t)row * 48;
    if (MODE == 0) {
        if (c < 12) *reinterpret_cast<float4 *>(p + 4 * c) = make_float4(c, row, 1.f, 2.f);
    } else if (MODE == 1) {
        V3 o; o.v[0] = c; o.v[1] = row; o.v[2] = 1.f;
        *reinterpret_cast<V3 *>(p + 3 * c) = o;
    } else {
        p[c] = c; p[16 + c] = row; p[32 + c] = 1.f;
    }
}

int main() {
    const int rows = 100000;
    float *d; hipMalloc(&d, (size_t)rows * 48 * 4);
    const int blocks = (rows * 16 + 255) / 256;
    for (int rep = 0; rep < 3; rep++) {
        wr<0><<<blocks, 256>>>(d, rows);
        wr<1><<<blocks, 256>>>(d, rows);
        wr<2><<<blocks, 256>>>(d, rows);
    }
    hipDeviceSynchronize();
    printf("wrote %.1f MB per kernel\n", rows * 192.0 / 1e6);
    return 0;
}
