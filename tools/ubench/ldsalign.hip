// Are ds_read_b128 / ds_read_b64 legal at 4-BYTE aligned LDS addresses on gfx950 (values correct), and what do they cost
// next to aligned ones?  (A planar bf16 patch read with stride-2 windows starts every window at a multiple of 4 bytes.)
// build: hipcc --offload-arch=gfx950 -O3 -o ldsalign ldsalign.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct W4 { unsigned a, b, c, d; };
struct W2 { unsigned a, b; };

template <int WIDTH>
__global__ __launch_bounds__(256) void probe(unsigned* out, int lane_stride, int shift, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned sm[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = i * 2654435761u + 12345u;
    __syncthreads();
    unsigned acc = 0;
    unsigned addr = (unsigned)(size_t)(sm) + (threadIdx.x & 63) * lane_stride + shift + (threadIdx.x >> 6) * 4096;
    for (int i = 0; i < iters; ++i) {
        const unsigned a = addr + ((i & 7) << 10);
        if (WIDTH == 16) {
            W4 v;
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.a + v.b + v.c + v.d;
        } else if (WIDTH == 8) {
            W2 v;
            asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.a + v.b;
        } else {
            unsigned v;
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// throughput variant: 8 independent reads in flight per iteration
template <int WIDTH>
__global__ __launch_bounds__(256) void stream(unsigned* out, int lane_stride, int shift, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned sm[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = i;
    __syncthreads();
    unsigned acc = 0;
    unsigned addr = (unsigned)(size_t)(sm) + (threadIdx.x & 63) * lane_stride + shift + (threadIdx.x >> 6) * 4096;
    for (int i = 0; i < iters; ++i) {
        if (WIDTH == 16) {
            W4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u]) : "v"(addr), "n"(u * 1024) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u].a + v[u].b + v[u].c + v[u].d;
        } else {
            unsigned v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[u]) : "v"(addr), "n"((u >> 2) * 1024 + (u & 3) * 4) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += v[u];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    unsigned* o; hipMalloc(&o, 1024 * 256 * 4);
    unsigned ho[64];
    auto word = [](unsigned byte) { return (byte / 4) * 2654435761u + 12345u; };
    for (int width : {16, 8, 4})
        for (int shift : {0, 4, 8, 12}) {
            const int ls = 16;
            if (width == 16) probe<16><<<1, 256>>>(o, ls, shift, 1);
            else if (width == 8) probe<8><<<1, 256>>>(o, ls, shift, 1);
            else probe<4><<<1, 256>>>(o, ls, shift, 1);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(ho, o, 64 * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l) {
                unsigned want = 0;
                for (int j = 0; j < width / 4; ++j) want += word(l * ls + shift + 4 * j);
                bad += want != ho[l];
            }
            printf("ds_read_b%-3d shift %2d: %s (%d bad lanes) %s\n", width * 8, shift, bad ? "BAD" : "OK ", bad, hipGetErrorString(e));
        }
    // timing: 1024 workgroups x 256 threads, windows 4 bytes apart per lane (the stride-2 bf16 window) and 16 apart
    for (int ls : {16, 4})
        for (int shift : {0, 4})
            for (int width : {16, 4}) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                auto go = [&]() { if (width == 16) stream<16><<<1024, 256>>>(o, ls, shift, 2000); else stream<4><<<1024, 256>>>(o, ls, shift, 2000); };
                go(); hipDeviceSynchronize();
                hipEventRecord(e0); go(); hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = 1024.0 * 256 * 2000 * 128;
                printf("%s lane stride %2d shift %d: %.3f ms  %.1f TB/s aggregate (%.1f B/clk/CU at 2.4 GHz)\n", width == 16 ? "8 x b128" : "32 x b32", ls, shift, ms,
                       bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
            }
    return 0;
}
