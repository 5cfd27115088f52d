// bf16 storage: a tap shifted by one element makes a vector of bf16 positions start at a 2-BYTE aligned address.
// Are raw-buffer b32 / b64 / b128 loads and global loads legal there on gfx950 (values correct), what do they cost, and
// what does the bounds check do with a vector that starts 2 bytes in front of the descriptor?
// build: hipcc --offload-arch=gfx950 -O3 -o align2probe align2probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct W4 { unsigned a, b, c, d; };
struct W2 { unsigned a, b; };

template <int WIDTH>
__global__ void probe(const unsigned short* src, unsigned* out, int bytes, int shift_bytes, int iters) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), 0, bytes, 0x00020000);
    unsigned acc = 0;
    const unsigned vo = (unsigned)((blockIdx.x * 256 + threadIdx.x) * WIDTH + shift_bytes);
    for (int i = 0; i < iters; ++i) {
        const unsigned o = vo + i * (256 * 256 * WIDTH);
        if (WIDTH == 16) {
            auto v = __builtin_bit_cast(W4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
            acc += v.a + v.b + v.c + v.d;
        } else if (WIDTH == 8) {
            auto v = __builtin_bit_cast(W2, __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 0));
            acc += v.a + v.b;
        } else {
            acc += __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void edge(const unsigned short* src, unsigned* out, int bytes) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src) + 64, 0, bytes, 0x00020000);
    const auto v4 = __builtin_bit_cast(W4, __builtin_amdgcn_raw_buffer_load_b128(rs, 0xfffffffeu, 0, 0));   // 2 B in front
    const auto v2 = __builtin_bit_cast(W2, __builtin_amdgcn_raw_buffer_load_b64(rs, 0xfffffffeu, 0, 0));
    const unsigned v1 = __builtin_amdgcn_raw_buffer_load_b32(rs, 0xfffffffeu, 0, 0);
    const auto t4 = __builtin_bit_cast(W4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)bytes - 14u, 0, 0));   // last word 2 B beyond
    const unsigned s0 = __builtin_amdgcn_raw_buffer_load_b16(rs, 2, 0, 0);
    const unsigned s1 = __builtin_amdgcn_raw_buffer_load_b16(rs, (unsigned)bytes, 0, 0);
    const auto g4 = *reinterpret_cast<const W4*>(reinterpret_cast<const unsigned char*>(src) + 130);         // global load, 2-byte aligned
    if (threadIdx.x == 0) {
        out[0] = v4.a; out[1] = v4.b; out[2] = v4.c; out[3] = v4.d; out[4] = v2.a; out[5] = v2.b; out[6] = v1;
        out[7] = t4.a; out[8] = t4.b; out[9] = t4.c; out[10] = t4.d; out[11] = s0; out[12] = s1;
        out[13] = g4.a; out[14] = g4.b; out[15] = g4.c; out[16] = g4.d;
    }
}

int main() {
    const size_t n = (size_t)256 * 256 * 8 * 64 + 256;              // bf16 elements
    unsigned short* h = (unsigned short*)malloc(n * 2);
    for (size_t i = 0; i < n; ++i) h[i] = (unsigned short)(i * 7 + 1);
    unsigned short* d; unsigned* o;
    hipMalloc(&d, n * 2); hipMalloc(&o, 256 * 256 * 4);
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    unsigned ho[32];
    auto word = [&](size_t byte) { return (unsigned)h[byte / 2] | ((unsigned)h[byte / 2 + 1] << 16); };
    auto run = [&](auto kern, int width, int shift) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        kern<<<256, 256>>>(d, o, (int)(n * 2), shift, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) kern<<<256, 256>>>(d, o, (int)(n * 2), shift, 64);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
        unsigned want = 0;
        for (int i = 0; i < 64; ++i) for (int j = 0; j < width / 4; ++j) want += word((size_t)i * 256 * 256 * width + shift + 4 * j);
        printf("b%-3d shift %2d B: lane0 sum %08x (expected %08x) %s  %.3f ms (%.1f GB/s)\n", width * 8, shift, ho[0], want,
               ho[0] == want ? "OK " : "BAD", ms / 10, 256.0 * 256 * width * 64 / (ms / 10) / 1e6);
    };
    for (int shift : {0, 4, 2, 6, 14}) run(probe<16>, 16, shift);
    for (int shift : {0, 4, 2, 6}) run(probe<8>, 8, shift);
    for (int shift : {0, 2}) run(probe<4>, 4, shift);
    edge<<<1, 64>>>(d, o, 1024);
    hipMemcpy(ho, o, 17 * 4, hipMemcpyDeviceToHost);
    printf("b128 starting 2 B in front: %08x %08x %08x %08x (in-range words would be %08x.. )\n", ho[0], ho[1], ho[2], ho[3], word(128 + 2));
    printf("b64  starting 2 B in front: %08x %08x; b32: %08x (elements 0: %04x 1: %04x)\n", ho[4], ho[5], ho[6], h[64], h[65]);
    printf("b128 ending 2 B beyond    : %08x %08x %08x %08x (words at -14: %08x %08x %08x, last element %04x)\n", ho[7], ho[8], ho[9], ho[10],
           word(128 + 1024 - 14), word(128 + 1024 - 10), word(128 + 1024 - 6), h[64 + 511]);
    printf("b16 at 2: %04x (expected %04x); b16 at num_records: %04x\n", ho[11], h[65], ho[12]);
    printf("global b128 at +130 B: %08x %08x %08x %08x (expected %08x %08x ..)\n", ho[13], ho[14], ho[15], ho[16], word(130), word(134));
    return 0;
}
