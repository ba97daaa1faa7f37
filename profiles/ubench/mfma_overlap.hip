// Does the f32 MFMA overlap with independent VALU work of the same SIMD?  Three kernels at 8 waves per SIMD, wall clock:
//   A: ITER x (4 independent v_mfma_f32_16x16x4_f32)            B: ITER x (32 independent v_fma_f32)           C: both, interleaved
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float a, float b, int iters, float *out) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0, 0, 0, 0};
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = a + i + threadIdx.x;
    const float av = a + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[i], 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = fmaf(x[i], b, a);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *out; hipMalloc(&out, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        float ms[3];
        for (int m = 0; m < 3; ++m) {
            auto launch = [&](int it) {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, 1.0f, 0.5f, it, out);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, 1.0f, 0.5f, it, out);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, 1.0f, 0.5f, it, out);
            };
            launch(10); hipDeviceSynchronize();
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[m], e0, e1);
        }
        printf("%d waves/SIMD: 4 MFMA %.3f ms | 32 v_fma_f32 %.3f ms | both %.3f ms  (sum %.3f, max %.3f)  -> overlap %.0f %%\n", wps, ms[0], ms[1], ms[2],
               ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1], 100.0 * (ms[0] + ms[1] - ms[2]) / (ms[0] < ms[1] ? ms[0] : ms[1]));
    }
    return 0;
}
