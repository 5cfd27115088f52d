// Pure-MFMA ceiling on this box: f32 32x32x2 and bf16 32x32x16, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void k_f32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k_bf16(float* out, int iters, short v) {
    f32x16 acc[NACC];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = v; b[i] = v; }
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double timeit(F f) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms * 1e-3;
}
int main() {
    float* out; hipMalloc(&out, 256 * 16 * 256 * 4 * sizeof(float));
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps) {            // waves per SIMD = blocks/CU (256-thread blocks)
        const int blocks = 256 * wps;
        double t = timeit([&] { hipLaunchKernelGGL(k_f32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
        double fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 2;
        printf("f32 32x32x2  waves/SIMD %d: %.1f TFLOP/s\n", wps, fl / t / 1e12);
        t = timeit([&] { hipLaunchKernelGGL(k_bf16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, (short)0x3f80); });
        fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("bf16 32x32x16 waves/SIMD %d: %.1f TFLOP/s\n", wps, fl / t / 1e12);
    }
    return 0;
}
