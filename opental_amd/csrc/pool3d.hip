// opental_amd/csrc/pool3d.hip -- MaxPool3dSamePadding forward / backward for (B,C,T,H,W) maps.
//
// Replaces MaxPool3dSamePadding (AFSD/common/layers.py:9-35): F.pad with ZEROS (not -inf) followed
// by nn.MaxPool3d, i.e. two ATen launches and a padded copy per pool in the reference.  Here the
// padding is virtual (out-of-range taps contribute the value 0.0), the forward records which tap
// won (uint8; 255 = a padded zero won, its gradient is dropped exactly as the reference's slice
// of the padded gradient drops it), and the backward is a deterministic gather over the <= 27
// windows that cover an input element -- no atomics.  Ties keep the first tap in (t,h,w) scan
// order, as aten::max_pool3d does.  HBM-bound: each element is read / written once.
#include "common.h"

#include "conv_index.h"

namespace {

struct PoolGeom {
    int B, C, Ti, Hi, Wi, To, Ho, Wo;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int64_t x_bs, x_cs, y_bs, y_cs;
    FastDiv fWo, fHo, fWi, fHi;
};

// KT..SW > 0: compile-time kernel / stride (the four pools of I3D); 0: read them from the geometry.
template <int KT, int KH, int KW, int ST, int SH, int SW>
struct PoolShape {
    __device__ __forceinline__ static int kt(const PoolGeom& g) { return KT ? KT : g.kt; }
    __device__ __forceinline__ static int kh(const PoolGeom& g) { return KT ? KH : g.kh; }
    __device__ __forceinline__ static int kw(const PoolGeom& g) { return KT ? KW : g.kw; }
    __device__ __forceinline__ static int st(const PoolGeom& g) { return KT ? ST : g.st; }
    __device__ __forceinline__ static int sh(const PoolGeom& g) { return KT ? SH : g.sh; }
    __device__ __forceinline__ static int sw(const PoolGeom& g) { return KT ? SW : g.sw; }
};

// grid: x over output positions of one (b,c) plane, y = b*C + c
template <int KT, int KH, int KW, int ST, int SH, int SW>
__global__ __launch_bounds__(256) void maxpool3d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            unsigned char* __restrict__ arg, PoolGeom g) {
    using S = PoolShape<KT, KH, KW, ST, SH, SW>;
    const int P = g.To * g.Ho * g.Wo;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    uint32_t q = fd_div(g.fWo, p);
    const int wo = p - q * g.Wo;
    const uint32_t q2 = fd_div(g.fHo, q);
    const int ho = q - q2 * g.Ho, to = (int)q2;
    const float* xb = x + (int64_t)b * g.x_bs + (int64_t)c * g.x_cs;
    const int kt = S::kt(g), kh = S::kh(g), kw = S::kw(g);
    float best = 0.f;
    int win = 255;
    bool first = true;
#pragma unroll
    for (int dt = 0; dt < kt; ++dt) {
        const int ti = to * S::st(g) + dt - g.pt;
#pragma unroll
        for (int dh = 0; dh < kh; ++dh) {
            const int hi = ho * S::sh(g) + dh - g.ph;
#pragma unroll
            for (int dw = 0; dw < kw; ++dw) {
                const int wi = wo * S::sw(g) + dw - g.pw;
                const bool in = (unsigned)ti < (unsigned)g.Ti && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                const float v = in ? xb[((int64_t)ti * g.Hi + hi) * g.Wi + wi] : 0.f;
                if (first || v > best || v != v) {
                    best = v;
                    win = in ? (dt * kh + dh) * kw + dw : 255;
                    first = false;
                }
            }
        }
    }
    y[(int64_t)b * g.y_bs + (int64_t)c * g.y_cs + p] = best;
    arg[(int64_t)bc * P + p] = (unsigned char)win;
}

// dx[b,c,i] (+)= sum over outputs o whose recorded winner is i of dy[b,c,o]; fixed (dt,dh,dw) order
template <int KT, int KH, int KW, int ST, int SH, int SW>
__global__ __launch_bounds__(256) void maxpool3d_bwd_kernel(const float* __restrict__ dy,
                                                            const unsigned char* __restrict__ arg,
                                                            float* __restrict__ dx, PoolGeom g, int accumulate,
                                                            const float* __restrict__ emask,
                                                            const float* __restrict__ escale) {
    using S = PoolShape<KT, KH, KW, ST, SH, SW>;
    const int Pi = g.Ti * g.Hi * g.Wi;
    const int Po = g.To * g.Ho * g.Wo;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Pi) return;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    uint32_t q = fd_div(g.fWi, p);
    const int wi = p - q * g.Wi;
    const uint32_t q2 = fd_div(g.fHi, q);
    const int hi = q - q2 * g.Hi, ti = (int)q2;
    const float* dyb = dy + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs;
    const unsigned char* ab = arg + (int64_t)bc * Po;
    const int kt = S::kt(g), kh = S::kh(g), kw = S::kw(g);
    const int st = S::st(g), sh = S::sh(g), sw = S::sw(g);
    // Branch-free gather over the windows that can cover (ti,hi,wi): per axis at most ceil(k/s) outputs,
    // o = (i+p)/s - c for c = 0..ceil(k/s)-1, reached through tap (i+p) - o*s when that tap is < k.
    // The winner byte and dy are loaded UNCONDITIONALLY from a clamped index and selected afterwards, so
    // all loads are in flight together (a guarded load makes hipcc branch and wait per tap).
    // Descending output index == ascending tap: fixed order -> deterministic sums.
    const int tn = ti + g.pt, hn = hi + g.ph, wn = wi + g.pw;
    constexpr int CT = KT ? (KT + ST - 1) / ST : 0, CH = KT ? (KH + SH - 1) / SH : 0, CW = KT ? (KW + SW - 1) / SW : 0;
    const int ct = KT ? CT : (kt + st - 1) / st, ch = KT ? CH : (kh + sh - 1) / sh, cw = KT ? CW : (kw + sw - 1) / sw;
    const int to0 = tn / st, ho0 = hn / sh, wo0 = wn / sw;
    float acc = 0.f;
#pragma unroll
    for (int a_ = 0; a_ < ct; ++a_) {
        const int to = to0 - a_, dt = tn - to * st;
        const bool okt = to >= 0 && to < g.To && dt < kt;
        const int toc = min(max(to, 0), g.To - 1);
#pragma unroll
        for (int b_ = 0; b_ < ch; ++b_) {
            const int ho = ho0 - b_, dh = hn - ho * sh;
            const bool okh = okt && ho >= 0 && ho < g.Ho && dh < kh;
            const int hoc = min(max(ho, 0), g.Ho - 1);
#pragma unroll
            for (int c_ = 0; c_ < cw; ++c_) {
                const int wo = wo0 - c_, dw = wn - wo * sw;
                const bool ok = okh && wo >= 0 && wo < g.Wo && dw < kw;
                const int woc = min(max(wo, 0), g.Wo - 1);
                const int o = (toc * g.Ho + hoc) * g.Wo + woc;
                const int a = ab[o];
                const float d = dyb[o];
                acc += (ok && a == (dt * kh + dh) * kw + dw) ? d : 0.f;
            }
        }
    }
    const int64_t off = (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + p;
    if (emask) acc = emask[off] > 0.f ? acc * escale[c] : 0.f;   // ReLU/BN backward of the pooled layer
    dx[off] = accumulate ? dx[off] + acc : acc;
}

int fill(PoolGeom& g, const int* d, const int64_t* s) {
    // d: B,C, Ti,Hi,Wi, To,Ho,Wo, kt,kh,kw, st,sh,sw, pt,ph,pw
    g.B = d[0]; g.C = d[1]; g.Ti = d[2]; g.Hi = d[3]; g.Wi = d[4]; g.To = d[5]; g.Ho = d[6]; g.Wo = d[7];
    g.kt = d[8]; g.kh = d[9]; g.kw = d[10]; g.st = d[11]; g.sh = d[12]; g.sw = d[13];
    g.pt = d[14]; g.ph = d[15]; g.pw = d[16];
    for (int i = 0; i < 14; ++i) if (d[i] <= 0) return OTAL_E_SHAPE;
    if (g.kt * g.kh * g.kw > 254) return OTAL_E_UNSUPPORTED;
    if ((int64_t)g.B * g.C > 65535) return OTAL_E_UNSUPPORTED;
    if ((int64_t)g.Ti * g.Hi * g.Wi >= (1LL << 31)) return OTAL_E_SHAPE;
    g.x_bs = s[0]; g.x_cs = s[1]; g.y_bs = s[2]; g.y_cs = s[3];
    g.fWo = make_fastdiv(g.Wo); g.fHo = make_fastdiv(g.Ho); g.fWi = make_fastdiv(g.Wi); g.fHi = make_fastdiv(g.Hi);
    return 0;
}

#define OTAL_POOL_DISPATCH(KERNEL, GRID, ...)                                                                   \
    do {                                                                                                         \
        const int kk = g.kt * 100 + g.kh * 10 + g.kw, ss = g.st * 100 + g.sh * 10 + g.sw;                        \
        if (kk == 133 && ss == 122) hipLaunchKernelGGL((KERNEL<1, 3, 3, 1, 2, 2>), GRID, dim3(256), 0, st_, __VA_ARGS__); \
        else if (kk == 333 && ss == 111) hipLaunchKernelGGL((KERNEL<3, 3, 3, 1, 1, 1>), GRID, dim3(256), 0, st_, __VA_ARGS__); \
        else if (kk == 333 && ss == 222) hipLaunchKernelGGL((KERNEL<3, 3, 3, 2, 2, 2>), GRID, dim3(256), 0, st_, __VA_ARGS__); \
        else if (kk == 222 && ss == 222) hipLaunchKernelGGL((KERNEL<2, 2, 2, 2, 2, 2>), GRID, dim3(256), 0, st_, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<0, 0, 0, 0, 0, 0>), GRID, dim3(256), 0, st_, __VA_ARGS__);              \
    } while (0)

}  // namespace

extern "C" int otal_maxpool3d_fwd(const int* geom, const int64_t* strides, const float* x, float* y,
                                  unsigned char* argtap, void* stream) {
    if (!geom || !strides || !x || !y || !argtap) return OTAL_E_NULL;
    PoolGeom g;
    if (int e = fill(g, geom, strides)) return e;
    hipStream_t st_ = (hipStream_t)stream;
    const dim3 grid((g.To * g.Ho * g.Wo + 255) / 256, g.B * g.C);
    OTAL_POOL_DISPATCH(maxpool3d_fwd_kernel, grid, x, y, argtap, g);
    return otal_launch_status();
}

extern "C" int otal_maxpool3d_bwd(const int* geom, const int64_t* strides, const float* dy,
                                  const unsigned char* argtap, float* dx, int accumulate,
                                  const float* out_mask, const float* out_scale, void* stream) {
    if (!geom || !strides || !dy || !dx || !argtap) return OTAL_E_NULL;
    if ((out_mask == nullptr) != (out_scale == nullptr)) return OTAL_E_NULL;
    PoolGeom g;
    if (int e = fill(g, geom, strides)) return e;
    hipStream_t st_ = (hipStream_t)stream;
    const dim3 grid((g.Ti * g.Hi * g.Wi + 255) / 256, g.B * g.C);
    OTAL_POOL_DISPATCH(maxpool3d_bwd_kernel, grid, dy, argtap, dx, g, accumulate, out_mask, out_scale);
    return otal_launch_status();
}
