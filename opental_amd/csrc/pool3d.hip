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

namespace {

struct PoolGeom {
    int B, C, Ti, Hi, Wi, To, Ho, Wo;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int64_t x_bs, x_cs, y_bs, y_cs;
};

__global__ __launch_bounds__(256) void maxpool3d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            unsigned char* __restrict__ arg, PoolGeom g) {
    const int64_t P = (int64_t)g.To * g.Ho * g.Wo;
    const int64_t total = (int64_t)g.B * g.C * P;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t r = idx;
        const int wo = (int)(r % g.Wo); r /= g.Wo;
        const int ho = (int)(r % g.Ho); r /= g.Ho;
        const int to = (int)(r % g.To); r /= g.To;
        const int c = (int)(r % g.C);
        const int b = (int)(r / g.C);
        const float* xb = x + (int64_t)b * g.x_bs + (int64_t)c * g.x_cs;
        float best = 0.f;
        int win = 255;
        bool first = true;
        for (int dt = 0; dt < g.kt; ++dt) {
            const int ti = to * g.st + dt - g.pt;
            for (int dh = 0; dh < g.kh; ++dh) {
                const int hi = ho * g.sh + dh - g.ph;
                for (int dw = 0; dw < g.kw; ++dw) {
                    const int wi = wo * g.sw + dw - g.pw;
                    const bool in = (unsigned)ti < (unsigned)g.Ti && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
                    const float v = in ? xb[((int64_t)ti * g.Hi + hi) * g.Wi + wi] : 0.f;
                    if (first || v > best || v != v) {
                        best = v;
                        win = in ? (dt * g.kh + dh) * g.kw + dw : 255;
                        first = false;
                    }
                }
            }
        }
        y[(int64_t)b * g.y_bs + (int64_t)c * g.y_cs + ((int64_t)to * g.Ho + ho) * g.Wo + wo] = best;
        arg[idx] = (unsigned char)win;
    }
}

// dx[b,c,i] (+)= sum over outputs o whose recorded winner is i of dy[b,c,o]; fixed (dt,dh,dw) order
__global__ __launch_bounds__(256) void maxpool3d_bwd_kernel(const float* __restrict__ dy,
                                                            const unsigned char* __restrict__ arg,
                                                            float* __restrict__ dx, PoolGeom g, int accumulate) {
    const int64_t Pi = (int64_t)g.Ti * g.Hi * g.Wi;
    const int64_t Po = (int64_t)g.To * g.Ho * g.Wo;
    const int64_t total = (int64_t)g.B * g.C * Pi;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t r = idx;
        const int wi = (int)(r % g.Wi); r /= g.Wi;
        const int hi = (int)(r % g.Hi); r /= g.Hi;
        const int ti = (int)(r % g.Ti); r /= g.Ti;
        const int c = (int)(r % g.C);
        const int b = (int)(r / g.C);
        const float* dyb = dy + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs;
        const unsigned char* ab = arg + ((int64_t)b * g.C + c) * Po;
        float acc = 0.f;
        for (int dt = 0; dt < g.kt; ++dt) {
            const int tn = ti + g.pt - dt;
            if (tn < 0 || tn % g.st) continue;
            const int to = tn / g.st;
            if (to >= g.To) continue;
            for (int dh = 0; dh < g.kh; ++dh) {
                const int hn = hi + g.ph - dh;
                if (hn < 0 || hn % g.sh) continue;
                const int ho = hn / g.sh;
                if (ho >= g.Ho) continue;
                for (int dw = 0; dw < g.kw; ++dw) {
                    const int wn = wi + g.pw - dw;
                    if (wn < 0 || wn % g.sw) continue;
                    const int wo = wn / g.sw;
                    if (wo >= g.Wo) continue;
                    const int64_t o = ((int64_t)to * g.Ho + ho) * g.Wo + wo;
                    if (ab[o] == (dt * g.kh + dh) * g.kw + dw) acc += dyb[o];
                }
            }
        }
        const int64_t off = (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + ((int64_t)ti * g.Hi + hi) * g.Wi + wi;
        dx[off] = accumulate ? dx[off] + acc : acc;
    }
}

int fill(PoolGeom& g, const int* d, const int64_t* s) {
    // d: B,C, Ti,Hi,Wi, To,Ho,Wo, kt,kh,kw, st,sh,sw, pt,ph,pw
    g.B = d[0]; g.C = d[1]; g.Ti = d[2]; g.Hi = d[3]; g.Wi = d[4]; g.To = d[5]; g.Ho = d[6]; g.Wo = d[7];
    g.kt = d[8]; g.kh = d[9]; g.kw = d[10]; g.st = d[11]; g.sh = d[12]; g.sw = d[13];
    g.pt = d[14]; g.ph = d[15]; g.pw = d[16];
    for (int i = 0; i < 14; ++i) if (d[i] <= 0) return OTAL_E_SHAPE;
    if (g.kt * g.kh * g.kw > 254) return OTAL_E_UNSUPPORTED;
    g.x_bs = s[0]; g.x_cs = s[1]; g.y_bs = s[2]; g.y_cs = s[3];
    return 0;
}

}  // namespace

extern "C" int otal_maxpool3d_fwd(const int* geom, const int64_t* strides, const float* x, float* y,
                                  unsigned char* argtap, void* stream) {
    if (!geom || !strides || !x || !y || !argtap) return OTAL_E_NULL;
    PoolGeom g;
    if (int e = fill(g, geom, strides)) return e;
    const int64_t total = (int64_t)g.B * g.C * g.To * g.Ho * g.Wo;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool3d_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, argtap, g);
    return otal_launch_status();
}

extern "C" int otal_maxpool3d_bwd(const int* geom, const int64_t* strides, const float* dy,
                                  const unsigned char* argtap, float* dx, int accumulate, void* stream) {
    if (!geom || !strides || !dy || !dx || !argtap) return OTAL_E_NULL;
    PoolGeom g;
    if (int e = fill(g, geom, strides)) return e;
    const int64_t total = (int64_t)g.B * g.C * g.Ti * g.Hi * g.Wi;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool3d_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, argtap, dx, g, accumulate);
    return otal_launch_status();
}
