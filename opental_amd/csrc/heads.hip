// opental_amd/csrc/heads.hip -- the tails of the detection heads for gfx950 (MI355X): everything between the head
// convolutions and the output dict of CoarsePyramid.forward, in one launch forward and one backward.
//
// Replaces, per pyramid level and head, AFSD/thumos14/BDNet.py:337-353,:399-412 (and AFSD/anet/BDNet.py:307-320,:366-376):
//   loc      = ScaleExp_l(conv)            .view(b,2,-1).permute(0,2,1).contiguous()   [* fpn_stride_l in the ANet model]
//   conf/act/prop_*/center = conv          .view(b,C,-1).permute(0,2,1).contiguous()
// and DirichletLayer.compute_uncertainty (BDNet.py:538-556): u = K / sum_k (exp(clamp(z_k, -10, 10)) + 1)
// -- ~25 ATen launches forward (exp, mul, 7 permute copies, 2 x {clamp, exp, add, sum, div}) and as many backward.
// The maps are small ((B, C <= 150, 126..189)): one thread per (item, sample, anchor) walks the channels; reads are
// coalesced along the anchors, the channel-last writes are 4..600 bytes per thread (L2-resident).
#include "common.h"

namespace {

constexpr int MAX_ITEMS = 8;
struct HeadItems {
    int n_items;
    int C[MAX_ITEMS];
    int mode[MAX_ITEMS];               // 0 permute, 1 exp(scale_l * x) * stride_l, 2 permute + Dirichlet uncertainty
    const float* raw[MAX_ITEMS];       // (B, C, N)
    float* out[MAX_ITEMS];             // (B, N, C)
    float* unct[MAX_ITEMS];            // (B, N) for mode 2
    // backward
    const float* dout[MAX_ITEMS];      // (B, N, C) or null
    const float* dunct[MAX_ITEMS];     // (B, N) or null
    float* draw[MAX_ITEMS];            // (B, C, N)
};
struct HeadLevels { int nlev; int lev[OTAL_MAX_LEVELS + 1]; float stride[OTAL_MAX_LEVELS]; };

__device__ __forceinline__ int level_of(const HeadLevels& L, int n) {
    int l = 0;
#pragma unroll
    for (int j = 1; j < OTAL_MAX_LEVELS; ++j)
        if (j < L.nlev && n >= L.lev[j]) l = j;
    return l;
}

// grid: (ceil(N / 256), B, n_items)
__global__ __launch_bounds__(256) void heads_fwd_kernel(HeadItems it, HeadLevels L, const float* __restrict__ scales, int B, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, i = blockIdx.z;
    if (n >= N) return;
    const int C = it.C[i], mode = it.mode[i];
    const float* x = it.raw[i] + (size_t)b * C * N + n;
    float* y = it.out[i] + ((size_t)b * N + n) * C;
    if (mode == 1) {
        const int l = level_of(L, n);
        const float s = scales[l], f = L.stride[l];
        for (int c = 0; c < C; ++c) y[c] = expf(x[(size_t)c * N] * s) * f;
        return;
    }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = x[(size_t)c * N];
        y[c] = v;
        if (mode == 2) sum += expf(fminf(fmaxf(v, -10.f), 10.f)) + 1.f;
    }
    if (mode == 2) it.unct[i][(size_t)b * N + n] = (float)C / sum;
}

// grid: (ceil(N / 256), B, n_items) + one extra z-slice (z == n_items) of a single workgroup for the ScaleExp gradients
__global__ __launch_bounds__(256) void heads_bwd_kernel(HeadItems it, HeadLevels L, const float* __restrict__ scales,
                                                        float* __restrict__ dscales, int B, int N) {
    const int i = blockIdx.z;
    if (i == it.n_items) {
        // d scale_l = sum over (b, n in level l, c) of dout * out * raw, summed in a fixed order (deterministic)
        if (blockIdx.x != 0 || blockIdx.y != 0) return;
        __shared__ float red[256];
        for (int l = 0; l < L.nlev; ++l) {
            float acc = 0.f;
            for (int j = 0; j < it.n_items; ++j) {
                if (it.mode[j] != 1 || !it.dout[j]) continue;
                const int C = it.C[j], t = L.lev[l + 1] - L.lev[l], cnt = B * t * C;
                for (int e = threadIdx.x; e < cnt; e += 256) {
                    const int c = e % C, q = e / C, nn = L.lev[l] + q % t, b = q / t;
                    const size_t o = ((size_t)b * N + nn) * C + c;
                    acc += it.dout[j][o] * it.out[j][o] * it.raw[j][((size_t)b * C + c) * N + nn];
                }
            }
            red[threadIdx.x] = acc;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
                __syncthreads();
            }
            if (threadIdx.x == 0) dscales[l] = red[0];
            __syncthreads();
        }
        return;
    }
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= N) return;
    const int C = it.C[i], mode = it.mode[i];
    float* dx = it.draw[i] + (size_t)b * C * N + n;
    const float* dy = it.dout[i] ? it.dout[i] + ((size_t)b * N + n) * C : nullptr;
    if (mode == 1) {
        const float s = scales[level_of(L, n)];
        const float* y = it.out[i] + ((size_t)b * N + n) * C;
        for (int c = 0; c < C; ++c) dx[(size_t)c * N] = dy ? dy[c] * y[c] * s : 0.f;
        return;
    }
    float coef = 0.f;                    // d unct / d z_c = -K / S^2 * exp(z_c) inside the clamp, 0 outside
    const float* x = it.raw[i] + (size_t)b * C * N + n;
    if (mode == 2 && it.dunct[i]) {
        const float u = it.unct[i][(size_t)b * N + n];          // K / S
        coef = -it.dunct[i][(size_t)b * N + n] * u * u / (float)C;
    }
    for (int c = 0; c < C; ++c) {
        float g = dy ? dy[c] : 0.f;
        if (coef != 0.f) {
            const float v = x[(size_t)c * N];
            if (v > -10.f && v < 10.f) g += coef * expf(v);
        }
        dx[(size_t)c * N] = g;
    }
}

int fill(HeadItems& it, HeadLevels& L, int n_items, const int* channels, const int* modes, int B, int N, int nlev, const int* lev,
         const float* strides) {
    if (n_items < 1 || n_items > MAX_ITEMS || B <= 0 || N <= 0) return OTAL_E_SHAPE;
    if (nlev < 1 || nlev > OTAL_MAX_LEVELS || !lev || lev[0] != 0 || lev[nlev] != N) return OTAL_E_LEVELS;
    it = HeadItems{};
    it.n_items = n_items;
    for (int i = 0; i < n_items; ++i) {
        if (channels[i] <= 0 || modes[i] < 0 || modes[i] > 2) return OTAL_E_SHAPE;
        it.C[i] = channels[i]; it.mode[i] = modes[i];
    }
    L.nlev = nlev;
    for (int i = 0; i <= OTAL_MAX_LEVELS; ++i) L.lev[i] = lev[i <= nlev ? i : nlev];
    for (int i = 0; i < nlev; ++i) if (L.lev[i + 1] <= L.lev[i]) return OTAL_E_LEVELS;
    for (int i = 0; i < OTAL_MAX_LEVELS; ++i) L.stride[i] = (strides && i < nlev) ? strides[i] : 1.f;
    return 0;
}

}  // namespace

extern "C" int otal_head_outputs_fwd(int n_items, const int* channels, const int* modes, const float* const* raw, float* const* out,
                                     float* const* unct, const float* scales, int B, int N, int nlev, const int* lev,
                                     const float* level_strides, void* stream) {
    if (!channels || !modes || !raw || !out || !unct) return OTAL_E_NULL;
    HeadItems it;
    HeadLevels L;
    if (int e = fill(it, L, n_items, channels, modes, B, N, nlev, lev, level_strides)) return e;
    for (int i = 0; i < n_items; ++i) {
        if (!raw[i] || !out[i] || (modes[i] == 2 && !unct[i]) || (modes[i] == 1 && !scales)) return OTAL_E_NULL;
        it.raw[i] = raw[i]; it.out[i] = out[i]; it.unct[i] = unct[i];
    }
    hipLaunchKernelGGL(heads_fwd_kernel, dim3((N + 255) / 256, B, n_items), dim3(256), 0, (hipStream_t)stream, it, L, scales, B, N);
    return otal_launch_status();
}

extern "C" int otal_head_outputs_bwd(int n_items, const int* channels, const int* modes, const float* const* raw,
                                     const float* const* out, const float* const* unct, const float* const* dout,
                                     const float* const* dunct, float* const* draw, const float* scales, float* dscales, int B,
                                     int N, int nlev, const int* lev, const float* level_strides, void* stream) {
    if (!channels || !modes || !raw || !out || !unct || !dout || !dunct || !draw) return OTAL_E_NULL;
    HeadItems it;
    HeadLevels L;
    if (int e = fill(it, L, n_items, channels, modes, B, N, nlev, lev, level_strides)) return e;
    bool any_exp = false;
    for (int i = 0; i < n_items; ++i) {
        if (!raw[i] || !out[i] || !draw[i] || (modes[i] == 2 && !unct[i])) return OTAL_E_NULL;
        it.raw[i] = raw[i]; it.out[i] = const_cast<float*>(out[i]); it.unct[i] = const_cast<float*>(unct[i]);
        it.dout[i] = dout[i]; it.dunct[i] = dunct[i]; it.draw[i] = draw[i];
        any_exp = any_exp || modes[i] == 1;
    }
    if (any_exp && (!scales || !dscales)) return OTAL_E_NULL;
    hipLaunchKernelGGL(heads_bwd_kernel, dim3((N + 255) / 256, B, n_items + (any_exp ? 1 : 0)), dim3(256), 0, (hipStream_t)stream,
                       it, L, scales, dscales, B, N);
    return otal_launch_status();
}
