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
    // A thread's channels (cg, cg + 16, ..: up to BCE_MAXI of them) are fetched in rounds of eight loads issued together and their
    // tanh kept in registers for the gradient pass below.  (The plain loop -- load, tanh, add, one channel at a time, then the
    // same again for the gradient -- was 2 x 32 dependent round trips per thread in a 19 us kernel; channels beyond the register
    // budget fall back to it.)  Same values, same order of additions.
    constexpr int BCE_MAXI = 32;
    float th[BCE_MAXI];
    const int nit = live ? (half - cg + 15) / 16 : 0;         // this thread's channel count
    float s = 0.f;
#pragma unroll
    for (int i0 = 0; i0 < BCE_MAXI; i0 += 8) {
        if (i0 >= nit) { for (int u = 0; u < 8; ++u) th[i0 + u] = 0.f; continue; }      // (uniform in practice: a whole round less)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[(int64_t)(cg + 16 * min(i0 + u, max(nit - 1, 0))) * x_cs];      // clamped: always readable
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            th[i0 + u] = tanhf(v[u]);
            if (i0 + u < nit) s += th[i0 + u];
        }
    }
    for (int i = BCE_MAXI; i < nit; ++i) s += tanhf(xp[(int64_t)(cg + 16 * i) * x_cs]);
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
#pragma unroll
    for (int i = 0; i < BCE_MAXI; ++i)
        if (i < nit) dp[(int64_t)(cg + 16 * i) * T] = gm * (1.f - th[i] * th[i]);
    for (int i = BCE_MAXI; i < nit; ++i) {
        const float t1 = tanhf(xp[(int64_t)(cg + 16 * i) * x_cs]);
        dp[(int64_t)(cg + 16 * i) * T] = gm * (1.f - t1 * t1);
    }
}


// ---- the tails of the three boundary losses of a training step in one launch each way.
// forward: out[h] = sum_i weight[i] * mean_{b,t} terms_i[b][h][t]   (h = start / end; i = the frame-level map and the two
// level-0 proposal maps, weights 1, 0.1, 0.1: AFSD/thumos14/train.py:193-201) -- ONE workgroup, fixed summation order.
// backward: out_i[b][c][t] = dx_i[b][c][t] * (weight[i] * g[half of c])  for all three maps (the stored d loss_h / d x,
// scaled by the incoming gradients of the two combined losses).
constexpr int BL_MAX = 4;
struct BlItems { int n; const float* terms[BL_MAX]; const float* dx[BL_MAX]; float* out[BL_MAX]; float weight[BL_MAX]; int T[BL_MAX]; int C[BL_MAX]; };

__global__ __launch_bounds__(256) void boundary_finish_kernel(BlItems it, int B, float* __restrict__ out) {
    // One workgroup, six sums (2 halves x up to 3 maps), each the pairwise halving tree over the 256 thread partials
    // (r[t] += r[t + s], s = 128 .. 1) as in the first version -- which walked it with nine barriers per sum and fetched a
    // thread's terms one dependent load at a time.  Here: eight loads per round, the cross-wave levels (s = 128, 64) summed by
    // wave 0 from LDS and the in-wave levels by shuffles: three barriers per sum, the same additions in the same order.
    __shared__ float red[256];
    const int t = threadIdx.x;
    for (int h = 0; h < 2; ++h) {
        float total = 0.f;
        for (int i = 0; i < it.n; ++i) {
            const int T = it.T[i], cnt = B * T;
            const float* __restrict__ src = it.terms[i];
            float acc = 0.f;
            for (int e0 = t; e0 < cnt; e0 += 8 * 256) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u * 256, cnt - 1);           // clamped: the surplus values are not added
                    const int b = e / T, tt = e - b * T;
                    v[u] = src[((size_t)b * 2 + h) * T + tt];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * 256 < cnt) acc += v[u];
            }
            red[t] = acc;
            __syncthreads();
            if (t < 64) {
                float r = (red[t] + red[t + 128]) + (red[t + 64] + red[t + 192]);       // s = 128, then s = 64
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) r += __shfl_down(r, s, 64);
                if (t == 0) red[0] = r;
            }
            __syncthreads();
            total += it.weight[i] * (red[0] / (float)cnt);
            __syncthreads();
        }
        if (t == 0) out[h] = total;
    }
}

// grid (ceil(max elements / 256), n)
__global__ __launch_bounds__(256) void boundary_scale_kernel(BlItems it, int B, const float* __restrict__ g_start,
                                                             const float* __restrict__ g_end) {
    const int i = blockIdx.y;
    const int T = it.T[i], C = it.C[i];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * C * T) return;
    const int c = (int)((e / T) % C);
    const float g = c < C / 2 ? (g_start ? *g_start : 0.f) : (g_end ? *g_end : 0.f);
    it.out[i][e] = it.dx[i][e] * (it.weight[i] * g);
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

extern "C" int otal_boundary_finish(int n, const float* const* terms, const float* weights, const int* T, int B, float* out2,
                                    void* stream) {
    if (!terms || !weights || !T || !out2) return OTAL_E_NULL;
    if (n < 1 || n > BL_MAX || B <= 0) return OTAL_E_SHAPE;
    BlItems it = {};
    it.n = n;
    for (int i = 0; i < n; ++i) {
        if (!terms[i] || T[i] <= 0) return OTAL_E_SHAPE;
        it.terms[i] = terms[i]; it.weight[i] = weights[i]; it.T[i] = T[i];
    }
    hipLaunchKernelGGL(boundary_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, it, B, out2);
    return otal_launch_status();
}

extern "C" int otal_boundary_scale(int n, const float* const* dx, float* const* out, const float* weights, const int* C,
                                   const int* T, int B, const float* g_start, const float* g_end, void* stream) {
    if (!dx || !out || !weights || !C || !T) return OTAL_E_NULL;
    if (n < 1 || n > BL_MAX || B <= 0) return OTAL_E_SHAPE;
    BlItems it = {};
    it.n = n;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        if (!dx[i] || !out[i] || T[i] <= 0 || C[i] <= 0 || (C[i] & 1)) return OTAL_E_SHAPE;
        it.dx[i] = dx[i]; it.out[i] = out[i]; it.weight[i] = weights[i]; it.T[i] = T[i]; it.C[i] = C[i];
        const int64_t cnt = (int64_t)B * C[i] * T[i];
        most = cnt > most ? cnt : most;
    }
    hipLaunchKernelGGL(boundary_scale_kernel, dim3((unsigned)((most + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, it, B,
                       g_start, g_end);
    return otal_launch_status();
}
