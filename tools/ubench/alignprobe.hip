// Is a 16-byte buffer load legal at an 8-byte (or 4-byte) aligned address on gfx950, and what does it cost?
#include <hip/hip_runtime.h>
#include <cstdio>
struct W4 { unsigned a, b, c, d; };
__global__ void probe(const float* src, float* out, int bytes, int shift_bytes, int iters) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
    float acc = 0.f;
    unsigned vo = (unsigned)((blockIdx.x * 256 + threadIdx.x) * 16 + shift_bytes);
    for (int i = 0; i < iters; ++i) {
        auto v = __builtin_bit_cast(W4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + i * (256 * 256 * 16), 0, 0));
        acc += __uint_as_float(v.a) + __uint_as_float(v.b) + __uint_as_float(v.c) + __uint_as_float(v.d);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const int n = 256 * 256 * 4 * 64 + 64;
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = (float)(i % 97);
    float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 256 * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    float ho[4];
    for (int shift : {0, 8, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<<<256, 256>>>(d, o, n * 4, shift, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) probe<<<256, 256>>>(d, o, n * 4, shift, 64);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        double want = 0; for (int i = 0; i < 64; ++i) for (int j = 0; j < 4; ++j) want += h[(size_t)i * 256 * 256 * 4 + shift / 4 + j];
        printf("shift %d B: lane0 sum %.1f (expected %.1f), %.3f ms per launch (%.1f GB/s)\n", shift, ho[0], want, ms / 10, 256.0 * 256 * 16 * 64 / (ms / 10) / 1e6);
    }
    return 0;
}
