// Does the bf16 MFMA (v_mfma_f32_16x16x32_bf16, the real matrix cores) overlap with independent vector work of the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int NV>
__global__ void __launch_bounds__(256) k(float a, float b, int iters, float *out) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0, 0, 0, 0};
    float x[NV];
    for (int i = 0; i < NV; ++i) x[i] = a + i + threadIdx.x;
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16) (a + i * 0.125f + threadIdx.x * 1e-3f); bv[i] = (__bf16) (b + i); }
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) x[i] = fmaf(x[i], b, a);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NV; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
void run(float *out, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        float ms[3];
        for (int m = 0; m < 3; ++m) {
            auto launch = [&](int it) {
                if (m == 0) hipLaunchKernelGGL((k<0, NV>), dim3(blocks), dim3(256), 0, 0, 1.0f, 0.5f, it, out);
                if (m == 1) hipLaunchKernelGGL((k<1, NV>), dim3(blocks), dim3(256), 0, 0, 1.0f, 0.5f, it, out);
                if (m == 2) hipLaunchKernelGGL((k<2, NV>), dim3(blocks), dim3(256), 0, 0, 1.0f, 0.5f, it, out);
            };
            launch(10); hipDeviceSynchronize();
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[m], e0, e1);
        }
        printf("%d waves/SIMD: 4 bf16 MFMA %.3f ms (%.0f TFLOP/s) | %d v_fma_f32 %.3f ms | both %.3f ms  (sum %.3f, max %.3f) -> hidden %.0f %% of the shorter\n", wps, ms[0],
               16384.0 * 4 * iters * blocks * 4 / ms[0] / 1e9, NV, ms[1], ms[2], ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1],
               100.0 * (ms[0] + ms[1] - ms[2]) / (ms[0] < ms[1] ? ms[0] : ms[1]));
    }
}
int main() {
    float *out; hipMalloc(&out, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run<16>(out, e0, e1);
    run<32>(out, e0, e1);
    return 0;
}
