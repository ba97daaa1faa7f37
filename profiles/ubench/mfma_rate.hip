// micro-benchmark: issue interval of v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32 on gfx950 (cycles per instruction per
// SIMD), with NACC independent accumulators and W waves per SIMD; plus the f64 C/D layout check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k64(double a, double b, int iters, double *out, long long *cyc) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-9, bv = b;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void k32(float a, float b, int iters, float *out, long long *cyc) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
    float av = a + threadIdx.x * 1e-6f, bv = b;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// mixed: VALU fp64 FMAs interleaved with f64 MFMAs: do the pipes overlap?
__global__ void kmix(double a, double b, int iters, int nvalu, double *out, long long *cyc) {
    f64x4 acc = {0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-9, bv = b, x = a, y = b;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        for (int v = 0; v < nvalu; ++v) x = fma(x, y, av);
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void klayout(const double *A, const double *B, double *D) {  // A[16][4], B[4][16] row-major -> D[16][16]
    const int l = threadIdx.x;
    f64x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[((l >> 4) + 4 * v) * 16 + (l & 15)] = c[v];
}
int main() {
    double *out; long long *cyc; float *outf;
    hipMalloc(&out, 1 << 24); hipMalloc(&outf, 1 << 24); hipMalloc(&cyc, 1 << 20);
    const int iters = 2000;
    auto report = [&](const char *name, int blocks, int nmfma) {
        hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        // clock64 = s_memtime counts at 100 MHz on gfx9? report raw ticks per MFMA and let the ratio speak
        printf("%-40s ticks per MFMA (per wave) %.3f\n", name, s / blocks / iters / nmfma);
    };
    for (int wpb : {64, 256, 512}) {  // 1, 4, 8 waves per block; one block per CU (256 blocks) -> waves per SIMD = wpb/256
        char nm[128];
        hipLaunchKernelGGL(k64<1>, dim3(256), dim3(wpb), 0, 0, 1.0, 2.0, iters, out, cyc); snprintf(nm, 128, "f64 16x16x4, 1 acc, %d thr/CU", wpb); report(nm, 256, 1);
        hipLaunchKernelGGL(k64<4>, dim3(256), dim3(wpb), 0, 0, 1.0, 2.0, iters, out, cyc); snprintf(nm, 128, "f64 16x16x4, 4 acc, %d thr/CU", wpb); report(nm, 256, 4);
        hipLaunchKernelGGL(k32<1>, dim3(256), dim3(wpb), 0, 0, 1.0f, 2.0f, iters, outf, cyc); snprintf(nm, 128, "f32 16x16x4, 1 acc, %d thr/CU", wpb); report(nm, 256, 1);
        hipLaunchKernelGGL(k32<4>, dim3(256), dim3(wpb), 0, 0, 1.0f, 2.0f, iters, outf, cyc); snprintf(nm, 128, "f32 16x16x4, 4 acc, %d thr/CU", wpb); report(nm, 256, 4);
    }
    for (int nv : {0, 4, 8, 16, 32}) {
        char nm[128];
        hipLaunchKernelGGL(kmix, dim3(256), dim3(512), 0, 0, 1.0, 2.0, iters, nv, out, cyc); snprintf(nm, 128, "f64 mfma + %d dependent v_fma_f64, 2 w/SIMD", nv); report(nm, 256, 1);
        hipLaunchKernelGGL(kmix, dim3(256), dim3(1024), 0, 0, 1.0, 2.0, iters, nv, out, cyc); snprintf(nm, 128, "f64 mfma + %d dependent v_fma_f64, 4 w/SIMD", nv); report(nm, 256, 1);
    }
    // wall-clock calibration of the tick: time a long kernel
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k64<4>, dim3(256), dim3(256), 0, 0, 1.0, 2.0, 200000, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    printf("calibration: %.3f ms wall for %lld ticks -> %.1f MHz tick; f64 MFMA: %.2f ns each per SIMD (1 wave/SIMD, 4 acc)\n", ms, h[0], h[0] / ms / 1e3, ms * 1e6 / (200000.0 * 4));
    // layout
    std::vector<double> A(64), B(64), D(256), R(256, 0);
    for (int i = 0; i < 64; ++i) { A[i] = (i * 7 % 11) - 3; B[i] = (i * 5 % 13) + 0.5 * (i % 3); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(klayout, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("f64 layout (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4v col=l&15): %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return 0;
}
