// f64_4x4_layout.hip -- lane layout of v_mfma_f64_4x4x4_4b_f64 (4 blocks of D[4x4] += A[4x4] B[4x4]) found by probing:
// for every pair of lanes (la, lb) with A = e_la, B = e_lb, which D lane lights up.
// Build: hipcc --offload-arch=gfx950 -O3 f64_4x4_layout.hip -o f64_4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const double *a, const double *b, double *d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
    static int hit[64][64];
    for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) {
        for (int l = 0; l < 64; l++) { ha[l] = l == la; hb[l] = l == lb; }
        (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(da, db, dd); (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
        int h = -1, cnt = 0; for (int l = 0; l < 64; l++) if (hd[l] != 0) { h = l; cnt++; }
        hit[la][lb] = cnt == 1 ? h : (cnt == 0 ? -1 : -2);
    }
    for (int la = 0; la < 64; la++) {
        printf("A lane %2d x B lane -> D lane:", la);
        for (int lb = 0; lb < 64; lb++) if (hit[la][lb] != -1) printf(" %d->%d", lb, hit[la][lb]);
        printf("\n");
    }
    return 0;
}
