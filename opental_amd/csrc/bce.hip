// opental_amd/csrc/bce.hip -- the start / end boundary losses of the training step for gfx950 (MI355X).
//
// Replaces calc_bce_loss (AFSD/thumos14/train.py:152-161, AFSD/anet/train.py:136-144) on the two channel halves of a
// boundary feature map: for each half h and every (sample b, frame t)
//     m = mean_c tanh(x[b, h*C/2 + c, t]),   loss_h = mean_{b,t} BCE(m, mask_h[b, t])
// -- per call the reference runs 2 permute copies, 2 tanh, 2 means, 2 mask slices and 2 BCE kernels forward and as many
// backward; the step calls it three times.  Here ONE launch reads the (B, C, T) map in place (coalesced along t, no
// permuted copy), writes the per-(b, t) loss terms and the complete gradient d loss_h / d x; the autograd node only sums
// the terms and scales the two gradient halves by the incoming scalars.
// BCE follows torch.nn.functional.binary_cross_entropy: logs clamped at -100, gradient (m - y) / max(m (1 - m), 1e-12).
#include "common.h"

namespace {

// grid: (ceil(T / 16), B, 2 halves); 256 threads = 16 frames x 16 channel groups (the maps are small -- (8,512,256),
// (8,1024,64) -- so the channel loop is spread over threads: one thread per frame walking 256 channels twice was a
// 0.5 ms serial chain of dependent strided loads)
__global__ __launch_bounds__(256) void boundary_bce_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_cs,
                                                           const float* __restrict__ mask, int64_t m_bs, int64_t m_rs, int row0,
                                                           int m_step, float* __restrict__ terms, float* __restrict__ dx,
                                                           int B, int C, int T) {
    __shared__ float part[16][17];
    const int tl = threadIdx.x & 15, cg = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + tl, b = blockIdx.y, h = blockIdx.z;
    const int half = C >> 1;
    const bool live = t < T;
    const float* xp = x + (int64_t)b * x_bs + (int64_t)h * half * x_cs + (live ? t : 0);
    float s = 0.f;
    if (live)
        for (int c = cg; c < half; c += 16) s += tanhf(xp[(int64_t)c * x_cs]);
    part[cg][tl] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += part[k][tl];             // fixed order: every thread of a frame gets the same mean
    if (!live) return;
    const float m = tot / (float)half;
    const float y = mask[(int64_t)b * m_bs + (int64_t)(row0 + h) * m_rs + (int64_t)t * m_step];
    if (cg == 0) {
        const float lm = fmaxf(logf(m), -100.f), l1m = fmaxf(logf(1.f - m), -100.f);
        terms[((int64_t)b * 2 + h) * T + t] = -(y * lm + (1.f - y) * l1m);
    }
    // d mean-loss / d x = (m - y) / max(m (1 - m), 1e-12) / (B T) / half * (1 - tanh(x)^2)
    const float gm = (m - y) / fmaxf(m * (1.f - m), 1e-12f) / ((float)B * (float)T) / (float)half;
    float* dp = dx + ((int64_t)b * C + (int64_t)h * half) * T + t;
    for (int c = cg; c < half; c += 16) {
        const float th = tanhf(xp[(int64_t)c * x_cs]);
        dp[(int64_t)c * T] = gm * (1.f - th * th);
    }
}

}  // namespace

extern "C" int otal_boundary_bce(const float* x, int64_t x_batch_stride, int64_t x_channel_stride, const float* mask,
                                 int64_t mask_batch_stride, int64_t mask_row_stride, int mask_row0, int mask_step, float* terms,
                                 float* dx, int B, int C, int T, void* stream) {
    if (!x || !mask || !terms || !dx) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || (C & 1) || mask_step <= 0 || mask_row0 < 0) return OTAL_E_SHAPE;
    hipLaunchKernelGGL(boundary_bce_kernel, dim3((T + 15) / 16, B, 2), dim3(256), 0, (hipStream_t)stream, x, x_batch_stride,
                       x_channel_stride, mask, mask_batch_stride, mask_row_stride, mask_row0, mask_step, terms, dx, B, C, T);
    return otal_launch_status();
}
