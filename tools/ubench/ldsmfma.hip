// How close to the matrix-pipe ceiling does the direct convolution's inner structure get?  One "K step" = 9 tap groups of
// RA + RB ds_read_b128 fragment reads and WM x WN v_mfma_f32_32x32x16_bf16, 8 (or 4) waves per workgroup, one barrier per step.
// Variants: 0 MFMAs only; 1 reads then MFMAs (no prefetch); 2 reads of group g+1 issued before the MFMAs of group g;
// 3 reads of group g+1 interleaved one by one between the MFMAs of group g; 4 = 3 without the per-step barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int VAR, int WM, int WN, int NT>
__global__ __launch_bounds__(NT) void k(float* out, const unsigned* fill, int steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 96 * 304 / 4 + 564 * 48 / 4; i += NT) reinterpret_cast<unsigned*>(lds)[i] = fill[i % 4096];
    __syncthreads();
    const unsigned char* A = lds;
    const unsigned char* X = lds + 96 * 304;
    f32x16 acc[WM][WN];
    for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int arow = (lane & 31) * 304 + (lane >> 5) * 16;
    const int xrow = (wave * WN * 32 + (lane & 31) + 25) * 48 + (lane >> 5) * 16;
    auto fetch = [&](int g9, bf16x8 (&av)[WM], bf16x8 (&bv)[WN]) {
        const int dh = g9 / 3, dw = g9 - dh * 3;
#pragma unroll
        for (int i = 0; i < WM; ++i) av[i] = *reinterpret_cast<const bf16x8*>(A + i * 32 * 304 + arow + g9 * 32);
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(X + xrow + (j * 32 + (dh - 1) * 24 + (dw - 1)) * 48);
    };
    for (int s = 0; s < steps; ++s) {
        bf16x8 avA[WM], bvA[WN], avB[WM], bvB[WN];
        if (VAR >= 2) fetch(0, avA, bvA);
#pragma unroll
        for (int g9 = 0; g9 < 9; ++g9) {
            bf16x8 (&av)[WM] = (VAR >= 2 && (g9 & 1)) ? avB : avA;
            bf16x8 (&bv)[WN] = (VAR >= 2 && (g9 & 1)) ? bvB : bvA;
            bf16x8 (&avn)[WM] = (g9 & 1) ? avA : avB;
            bf16x8 (&bvn)[WN] = (g9 & 1) ? bvA : bvB;
            if (VAR == 0) { if (s == 0 && g9 == 0) fetch(0, avA, bvA); }
            if (VAR == 1) fetch(g9, avA, bvA);
            if (VAR == 2 && g9 < 8) fetch(g9 + 1, avn, bvn);
            if (VAR >= 3 && g9 < 8) {
                const int dh = (g9 + 1) / 3, dw = (g9 + 1) - dh * 3;
                int n = 0;
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                        if (n < WM) avn[n] = *reinterpret_cast<const bf16x8*>(A + n * 32 * 304 + arow + (g9 + 1) * 32);
                        else if (n < WM + WN) bvn[n - WM] = *reinterpret_cast<const bf16x8*>(X + xrow + ((n - WM) * 32 + (dh - 1) * 24 + (dw - 1)) * 48);
                        ++n;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (VAR != 4) __syncthreads();
    }
    float sum = 0;
    for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * NT + tid] = sum;
}

template <int VAR, int WM, int WN, int NT>
void run(const char* name, float* out, const unsigned* fill, int wgs_per_cu) {
    const int steps = 2000, lds = 96 * 304 + 564 * 48 + 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<VAR, WM, WN, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256 * wgs_per_cu;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k<VAR, WM, WN, NT>), dim3(blocks), dim3(NT), lds, 0, out, fill, 10);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL((k<VAR, WM, WN, NT>), dim3(blocks), dim3(NT), lds, 0, out, fill, steps);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double fl = (double)blocks * (NT / 64) * steps * 9.0 * WM * WN * 2.0 * 32 * 32 * 16;
    printf("%-44s WM %d WN %d waves/WG %d WG/CU %d: %7.1f TFLOP/s  (%.2f us per step)\n", name, WM, WN, NT / 64, wgs_per_cu,
           fl / (ms * 1e-3) / 1e12, ms * 1e3 / steps);
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 512 * sizeof(float));
    unsigned* fill; hipMalloc(&fill, 4096 * 4);
    unsigned h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) {            // random bf16 pairs in [-2, 2)
        auto bf = [] { float f = (rand() / (float)RAND_MAX) * 4.f - 2.f; unsigned u; __builtin_memcpy(&u, &f, 4); return u >> 16; };
        h[i] = bf() | (bf() << 16);
    }
    hipMemcpy(fill, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 3, 2, 512>("MFMAs only", out, fill, 1);
    run<1, 3, 2, 512>("reads, then MFMAs", out, fill, 1);
    run<2, 3, 2, 512>("next group's reads before the MFMAs", out, fill, 1);
    run<3, 3, 2, 512>("next group's reads between the MFMAs", out, fill, 1);
    run<4, 3, 2, 512>("... and no barrier per step", out, fill, 1);
    run<0, 3, 1, 512>("MFMAs only", out, fill, 1);
    run<1, 3, 1, 512>("reads, then MFMAs", out, fill, 1);
    run<3, 3, 1, 512>("next group's reads between the MFMAs", out, fill, 1);
    run<1, 2, 1, 512>("reads, then MFMAs", out, fill, 2);
    run<3, 2, 1, 512>("next group's reads between the MFMAs", out, fill, 2);
    run<1, 3, 2, 256>("reads, then MFMAs", out, fill, 1);
    run<3, 3, 2, 256>("next group's reads between the MFMAs", out, fill, 1);
    run<3, 3, 2, 256>("next group's reads between the MFMAs", out, fill, 2);
    run<3, 3, 4, 256>("next group's reads between the MFMAs", out, fill, 1);
    return 0;
}
