// How fast can a (64 channels x 96 positions) output tile be written to a channel-major tensor whose channel rows are
// 1.18 MB apart (the Conv3d_1a output, 604 MB for b = 8)?  Three store patterns, same bytes:
//   0  the MFMA accumulator layout as store_acc writes it: dword per lane, 32 consecutive positions of one channel per
//      half-wave (two 128-byte runs per instruction)
//   1  float4 per lane, 8 lanes per 128-byte run (LDS-transposed per wave): 8 channel rows per instruction
//   2  float4 per lane, 24 lanes per 384-byte run (LDS-transposed per workgroup): the whole 96-position row at once
// Measured on MI355X: 121 / 119 / 132 us for the 604 MB -- 5.0 TB/s whatever the pattern: the dword-per-lane accumulator
// layout of store_acc is NOT what limits the convolution epilogues.
// build: hipcc --offload-arch=gfx950 -O3 -o storepat storepat.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int C = 64, T = 128, H = 48, W = 48, B = 8;
constexpr long CS = (long)T * H * W, BS = (long)C * CS;

template <int PAT>
__global__ __launch_bounds__(384) void wr(float* y) {
    // tile = (b, t pair, h pair): 2 x 2 x 48 positions; 6 waves x 32 positions, 64 channels
    const int tiles_h = H / 2, tiles_t = T / 2;
    int bid = blockIdx.x;
    bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int th = bid % tiles_h; bid /= tiles_h;
    const int tt = bid % tiles_t; const int b = bid / tiles_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* yb = y + b * BS;
    const float v = (float)tid;
    if (PAT == 0) {
        const int n = wave * 32 + (lane & 31);                 // position in tile: run (n / 96), offset n % 96
        const long pos = ((long)(2 * tt + n / 96) * H + 2 * th) * W + n % 96;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                yb[m * CS + pos] = v + r;
            }
    } else if (PAT == 1) {
        // wave owns 32 positions x 64 channels = 2048 floats = 512 float4: 8 per lane; lane -> (channel group, quad)
        const int n0 = wave * 32;
        const long pos0 = ((long)(2 * tt + n0 / 96) * H + 2 * th) * W + n0 % 96;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int item = k * 64 + lane, m = item >> 3, q = item & 7;
            *reinterpret_cast<float4*>(yb + m * CS + pos0 + 4 * q) = make_float4(v, v + 1, v + 2, v + k);
        }
    } else {
        // workgroup owns 2 runs x 64 channels x 96 positions = 3072 float4: 8 per thread; thread -> (run, channel, quad of 24)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int item = k * 384 + tid, run = item / 1536, rem = item - run * 1536, m = rem / 24, q = rem - m * 24;
            const long pos0 = ((long)(2 * tt + run) * H + 2 * th) * W;
            *reinterpret_cast<float4*>(yb + m * CS + pos0 + 4 * q) = make_float4(v, v + 1, v + 2, v + k);
        }
    }
}

int main() {
    float* y;
    const size_t bytes = (size_t)B * BS * 4;
    hipMalloc(&y, bytes);
    const int grid = B * (T / 2) * (H / 2);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int pat = 0; pat < 3; ++pat) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            for (int i = 0; i < 10; ++i) {
                if (pat == 0) hipLaunchKernelGGL(wr<0>, dim3(grid), dim3(384), 0, 0, y);
                else if (pat == 1) hipLaunchKernelGGL(wr<1>, dim3(grid), dim3(384), 0, 0, y);
                else hipLaunchKernelGGL(wr<2>, dim3(grid), dim3(384), 0, 0, y);
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("pattern %d: %.1f us per launch, %.2f TB/s\n", pat, ms * 100.f, (double)bytes / (ms * 1e-4) / 1e12);
        }
    }
    return 0;
}
