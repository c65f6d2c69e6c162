// Layout check of v_mfma_f64_16x16x4_f64 on gfx950 (MI355X_MICROARCH/cdna_hip_programming: f64 C/D map differs from f32):
//   A operand: lane l holds A[i = l & 15][k = l >> 4] ; B operand: B[k = l >> 4][j = l & 15]
//   C/D: 4 doubles per lane, col = l & 15, row = (l >> 4) + 4 * reg
// Exact integer data, asymmetric operands.  Build+run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_layout_test.hip -o /tmp/mf && /tmp/mf
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* C) {
    int l = threadIdx.x;
    d4 c = {0, 0, 0, 0};
    for (int kk = 0; kk < 4; kk++) {
        double a = A[(l & 15) * 16 + 4 * kk + (l >> 4)];
        double b = B[(4 * kk + (l >> 4)) * 16 + (l & 15)];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main() {
    double hA[256], hB[256], hC[256], ref[256];
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { hA[i * 16 + j] = (i * 7 + j * 3) % 11 - 5; hB[i * 16 + j] = (i * 5 + j * 13) % 17 - 8; }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int t = 0; t < 16; t++) s += hA[i * 16 + t] * hB[t * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dC; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 2048);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC); hipMemcpy(hC, dC, 2048, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; i++) if (hC[i] != ref[i]) bad++;
    printf("mfma_f64_16x16x4 layout check: %d mismatches of 256\n", bad);
    return bad != 0;
}
