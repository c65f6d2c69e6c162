// Probe (GPU box): operand layout and issue rate of v_mfma_f64_4x4x4_4b_f64 against v_mfma_f64_16x16x4_f64 on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f64_4x4.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_layout(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    double acc = 0.0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], acc, 0, 0, 0);
    d[l] = acc;
}
template <int KIND> __global__ void k_rate(double* out, int iters, unsigned long long* cyc) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0; double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0); }
        else if (KIND == 1) { s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0); s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0); s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0); s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0); }
        else { s0 = __builtin_fma(a, b, s0); s1 = __builtin_fma(a, b, s1); s2 = __builtin_fma(a, b, s2); s3 = __builtin_fma(a, b, s3); }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3;
}
int main() {
    double *a, *b, *d; hipMalloc(&a, 64 * 8); hipMalloc(&b, 64 * 8); hipMalloc(&d, 64 * 8);
    std::vector<double> ha(64), hb(64), hd(64);
    // layout: A one-hot at lane la, B all ones in 'every lane' -> which output lanes light up tells (block, row) of lane la; and the transpose for B
    printf("A one-hot (B = lane index + 1): lane -> [output lane: value]\n");
    for (int la = 0; la < 64; la += 21) {
        for (int i = 0; i < 64; i++) { ha[i] = (i == la) ? 1.0 : 0.0; hb[i] = i + 1; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, d); hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("A lane %2d:", la); for (int i = 0; i < 64; i++) if (hd[i] != 0.0) printf(" [%d: B lane %d]", i, (int)hd[i] - 1); printf("\n");
    }
    unsigned long long* cyc; hipMalloc(&cyc, 8); double* out; hipMalloc(&out, (size_t)1024 * 512 * 8);      // 1024 blocks x at most 512 threads
    const int iters = 20000; unsigned long long hc;
    for (int kind = 0; kind < 3; kind++) for (int waves = 1; waves <= 2; waves++) {      // one wave per SIMD (256 threads = 4 waves on a CU's 4 SIMDs), then two per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        dim3 g(256 * 4), blk(256 * waves);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k_rate<0>, g, blk, 0, 0, out, iters, cyc); else if (kind == 1) hipLaunchKernelGGL(k_rate<1>, g, blk, 0, 0, out, iters, cyc); else hipLaunchKernelGGL(k_rate<2>, g, blk, 0, 0, out, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        const char* nm[] = {"mfma_f64_16x16x4", "mfma_f64_4x4x4_4b", "v_fma_f64"};
        printf("%-18s %d wave(s)/SIMD: %.1f cycles per instruction per wave (clock64 of wave 0), launch %.3f ms\n", nm[kind], waves, (double)hc / (4.0 * iters), ms);
    }
    return 0;
}
