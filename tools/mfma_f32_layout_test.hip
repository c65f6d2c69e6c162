// Layout check of v_mfma_f32_16x16x4_f32 on gfx950, the instruction of the fp32 Riccati sweep (hs_mfma.hpp MfmaT<float>):
//   A operand: lane l holds A[i = l & 15][k = l >> 4] ; B operand: B[k = l >> 4][j = l & 15]   (as the fp64 instruction)
//   C/D: 4 floats per lane, col = l & 15, row = 4 * (l >> 4) + reg                               (NOT the fp64 map, (l >> 4) + 4 * reg)
// Exact integer data, asymmetric operands; prints which of the two row maps the hardware uses.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/mfma_f32_layout_test.hip -o /tmp/mf32 && /tmp/mf32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* Cacc) {
    int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    for (int kk = 0; kk < 4; kk++) {
        float a = A[(l & 15) * 16 + 4 * kk + (l >> 4)];
        float b = B[(4 * kk + (l >> 4)) * 16 + (l & 15)];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) Cacc[l * 4 + r] = c[r];     // raw accumulators: lane-major
}
int main() {
    float hA[256], hB[256], acc[256], ref[256];
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { hA[i * 16 + j] = (float)((i * 7 + j * 3) % 11 - 5); hB[i * 16 + j] = (float)((i * 5 + j * 13) % 17 - 8); }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float s = 0; for (int t = 0; t < 16; t++) s += hA[i * 16 + t] * hB[t * 16 + j]; ref[i * 16 + j] = s; }
    float *dA, *dB, *dC; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC); hipMemcpy(acc, dC, 1024, hipMemcpyDeviceToHost);
    int bad32 = 0, bad64 = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        if (acc[l * 4 + r] != ref[(4 * (l >> 4) + r) * 16 + (l & 15)]) bad32++;
        if (acc[l * 4 + r] != ref[((l >> 4) + 4 * r) * 16 + (l & 15)]) bad64++;
    }
    printf("mfma_f32_16x16x4 layout check: row = 4*(l>>4)+r: %d mismatches ; row = (l>>4)+4*r: %d mismatches (of 256)\n", bad32, bad64);
    return bad32 != 0;
}
