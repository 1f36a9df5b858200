// micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 for one wave per SIMD (developer aid)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // j-major: 4 dependent MFMAs per accumulator
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        } else if (MODE == 1) {  // s-major: accumulators alternate
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        } else {  // one accumulator only
#pragma unroll
            for (int s = 0; s < 16; ++s) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * 256] = (float)(t1 - t0) / (iters * 16.0f);
}
int main() {
    float* d; hipMalloc(&d, (1024 * 256 + 1) * 4);
    for (int blocks : {64, 256, 512}) {
        float r[3];
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms[3];
        for (int m = 0; m < 3; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (m == 0) k<0><<<blocks, 256>>>(d, 2000, 1.f, 1.f);
                if (m == 1) k<1><<<blocks, 256>>>(d, 2000, 1.f, 1.f);
                if (m == 2) k<2><<<blocks, 256>>>(d, 2000, 1.f, 1.f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[m], e0, e1);
            }
            hipMemcpy(&r[m], d + blocks * 256, 4, hipMemcpyDeviceToHost);
        }
        printf("blocks %d: cycles/MFMA (s_memtime ticks) j-major %.1f  s-major %.1f  single-acc %.1f | ms %.3f %.3f %.3f -> TFLOP/s %.1f %.1f %.1f\n", blocks, r[0], r[1], r[2],
               ms[0], ms[1], ms[2], blocks * 4.0 * 2000 * 16 * 4096 / ms[0] / 1e9, blocks * 4.0 * 2000 * 16 * 4096 / ms[1] / 1e9, blocks * 4.0 * 2000 * 16 * 4096 / ms[2] / 1e9);
    }
    return 0;
}
