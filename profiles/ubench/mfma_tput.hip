// wall-clock throughput of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 / v_fma_f64 at 1..8 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k64(double a, double b, int iters, double *out) {
    f64x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k32(float a, float b, int iters, float *out) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0, 0, 0, 0};
    float av = a + threadIdx.x * 1e-6f, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) kfma(double a, double b, int iters, double *out) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], b, a);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// both at once in one wave: 1 f64 MFMA + NV independent fp64 FMAs per iteration
template <int NV>
__global__ void __launch_bounds__(256) kboth(double a, double b, int iters, double *out) {
    f64x4 acc[2];
    acc[0] = {0, 0, 0, 0}; acc[1] = {0, 0, 0, 0};
    double x[NV];
    for (int i = 0; i < NV; ++i) x[i] = a + i + threadIdx.x;
    double av = a + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
        acc[it & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b, acc[it & 1], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] = fma(x[i], b, a);
    }
    double s = acc[0][0] + acc[1][1];
    for (int i = 0; i < NV; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *out; hipMalloc(&out, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads (4 waves = 1 per SIMD), wps blocks per CU
        const int blocks = 256 * wps;
        float ms;
        hipLaunchKernelGGL(k64, dim3(blocks), dim3(256), 0, 0, 1.0, 2.0, 10, out); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k64, dim3(blocks), dim3(256), 0, 0, 1.0, 2.0, iters, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("f64 mfma  %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wps, ms, 2048.0 * 4 * iters * blocks * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 / (4.0 * iters * wps));
        hipEventRecord(e0); hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, 1.0f, 2.0f, iters, (float *) out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("f32 mfma  %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD)\n", wps, ms, 2048.0 * 4 * iters * blocks * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 / (4.0 * iters * wps));
        hipEventRecord(e0); hipLaunchKernelGGL(kfma, dim3(blocks), dim3(256), 0, 0, 1.0, 0.5, iters, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("v_fma_f64 %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  (%.1f cycles per instruction per SIMD)\n", wps, ms, 128.0 * 8 * iters * blocks * 4 / ms / 1e9, ms * 1e-3 * 2.4e9 / (8.0 * iters * wps));
        hipEventRecord(e0); hipLaunchKernelGGL(kboth<8>, dim3(blocks), dim3(256), 0, 0, 1.0, 0.5, iters, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("1 mfma64 + 8 v_fma_f64 per iter, %d waves/SIMD: %8.3f ms (%.1f cycles per iteration per SIMD; alone: 64 and 32)\n", wps, ms, ms * 1e-3 * 2.4e9 / (1.0 * iters * wps));
        hipEventRecord(e0); hipLaunchKernelGGL(kboth<16>, dim3(blocks), dim3(256), 0, 0, 1.0, 0.5, iters, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("1 mfma64 + 16 v_fma_f64 per iter, %d waves/SIMD: %8.3f ms (%.1f cycles per iteration per SIMD; alone: 64 and 64)\n", wps, ms, ms * 1e-3 * 2.4e9 / (1.0 * iters * wps));
    }
    return 0;
}
