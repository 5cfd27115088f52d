// opental_amd/csrc/conv_index.h -- geometry of the implicit-GEMM convolution, shared verbatim by
// the gfx950 kernels (conv_gemm.hip) and by a plain-C++ CPU harness (tests/cpu_conv_index.cpp) that
// checks this index math against torch on the CPU.  No HIP types in here.
//
// One geometry serves Conv1d (H = W = 1; reference Unit1D, AFSD/common/layers.py:178-214) and
// Conv3d (reference Unit3D, AFSD/common/i3d_backbone.py:7-87, layers.py:106-175).  TF-"SAME"
// padding is never materialised: `pt/ph/pw` are the FRONT pads and out-of-range taps read zero.
// A level table turns a stride-1 Conv1d over a packed (B,C,sum t_l) pyramid buffer into six
// independent SAME-padded convolutions in one launch (taps never cross a level boundary).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define OTAL_HD __host__ __device__ __forceinline__
#else
#define OTAL_HD inline
#endif

#define OTAL_CONV_MAX_LEVELS 8

struct ConvGeom {
    int B, Cin, Cout;
    int Ti, Hi, Wi;            // input extent
    int To, Ho, Wo;            // output extent
    int kt, kh, kw;            // kernel
    int st, sh, sw;            // stride
    int pt, ph, pw;            // front padding (SAME split: total // 2)
    int64_t x_bs, x_cs;        // input  batch / channel stride in elements (spatial is dense)
    int64_t y_bs, y_cs;        // output batch / channel stride in elements
    int nlev;                  // > 1: packed 1-D levels (H = W = 1, stride 1, Ti == To)
    int lev[OTAL_CONV_MAX_LEVELS + 1];
};

// Division by a launch-invariant divisor without the ~40-instruction runtime divide: round-up
// multiplier method (Granlund-Montgomery), exact for every 32-bit numerator.  Built on the host.
struct FastDiv {
    uint32_t d, mul, sh, one;   // one = 0xffffffff marks d == 1 (identity), else 0; branch-free at run time
};
OTAL_HD FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.one = 0;
    if (d <= 1) { f.d = 1; f.mul = 0; f.sh = 0; f.one = 0xffffffffu; return f; }
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;                       // s = ceil(log2 d), 1..32
    f.mul = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
    f.sh = s - 1;
    return f;
}
OTAL_HD uint32_t otal_umulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
OTAL_HD uint32_t fd_div(const FastDiv& f, uint32_t n) {
    const uint32_t t = otal_umulhi(f.mul, n);
    const uint32_t q = (t + ((n - t) >> 1)) >> f.sh;
    return (q & ~f.one) | (n & f.one);
}

struct PosDec { int b, t, h, w; };          // a decomposed spatial position
struct TapDec { int c, dt, dh, dw; };       // channel + kernel tap

OTAL_HD int conv_kvol(const ConvGeom& g) { return g.kt * g.kh * g.kw; }
OTAL_HD int conv_out_positions(const ConvGeom& g) { return g.To * g.Ho * g.Wo; }
OTAL_HD int conv_in_positions(const ConvGeom& g) { return g.Ti * g.Hi * g.Wi; }

OTAL_HD PosDec dec_pos(int n, int T, int H, int W) {   // n = ((b*T + t)*H + h)*W + w
    PosDec p;
    p.w = n % W; n /= W;
    p.h = n % H; n /= H;
    p.t = n % T; p.b = n / T;
    return p;
}
OTAL_HD TapDec dec_tap(const ConvGeom& g, int k) {     // k = ((c*kt + dt)*kh + dh)*kw + dw
    TapDec d;
    d.dw = k % g.kw; k /= g.kw;
    d.dh = k % g.kh; k /= g.kh;
    d.dt = k % g.kt; d.c = k / g.kt;
    return d;
}

struct ConvFastDiv { FastDiv kw, kh, kt, Wo, Ho, To, Wi, Hi, Ti, P; };   // P = To*Ho*Wo
inline ConvFastDiv make_conv_fastdiv(const ConvGeom& g) {
    ConvFastDiv f;
    f.kw = make_fastdiv(g.kw); f.kh = make_fastdiv(g.kh); f.kt = make_fastdiv(g.kt);
    f.Wo = make_fastdiv(g.Wo); f.Ho = make_fastdiv(g.Ho); f.To = make_fastdiv(g.To);
    f.Wi = make_fastdiv(g.Wi); f.Hi = make_fastdiv(g.Hi); f.Ti = make_fastdiv(g.Ti);
    f.P = make_fastdiv((uint32_t)(g.To * g.Ho * g.Wo));
    return f;
}
OTAL_HD PosDec dec_pos_fd(uint32_t n, const FastDiv& T, const FastDiv& H, const FastDiv& W) {
    PosDec p;
    uint32_t q = fd_div(W, n); p.w = (int)(n - q * W.d); n = q;
    q = fd_div(H, n); p.h = (int)(n - q * H.d); n = q;
    q = fd_div(T, n); p.t = (int)(n - q * T.d); p.b = (int)q;
    return p;
}
OTAL_HD TapDec dec_tap_fd(const ConvFastDiv& f, uint32_t k) {
    TapDec d;
    uint32_t q = fd_div(f.kw, k); d.dw = (int)(k - q * f.kw.d); k = q;
    q = fd_div(f.kh, k); d.dh = (int)(k - q * f.kh.d); k = q;
    q = fd_div(f.kt, k); d.dt = (int)(k - q * f.kt.d); d.c = (int)q;
    return d;
}

// [lo,hi) of the level that column t belongs to (whole axis when nlev <= 1)
OTAL_HD void level_bounds(const ConvGeom& g, int t, int extent, int& lo, int& hi) {
    lo = 0; hi = extent;
    if (g.nlev > 1) {
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < OTAL_CONV_MAX_LEVELS; ++j)
            if (j < g.nlev && t >= g.lev[j]) { lo = g.lev[j]; hi = g.lev[j + 1]; }
    }
}

// forward / weight-gradient gather: which input element does tap `k` of output position `o` read?
OTAL_HD bool conv_src_of_output(const ConvGeom& g, const PosDec& o, const TapDec& k, int64_t& off) {
    const int ti = o.t * g.st + k.dt - g.pt;
    const int hi = o.h * g.sh + k.dh - g.ph;
    const int wi = o.w * g.sw + k.dw - g.pw;
    int lo, up;
    level_bounds(g, o.t, g.Ti, lo, up);
    if (ti < lo || ti >= up || (unsigned)hi >= (unsigned)g.Hi || (unsigned)wi >= (unsigned)g.Wi) return false;
    off = (int64_t)o.b * g.x_bs + (int64_t)k.c * g.x_cs + ((int64_t)ti * g.Hi + hi) * g.Wi + wi;
    return true;
}

OTAL_HD bool div_stride(int num, int s, int extent, int& q) {
    if (num < 0) return false;
    if (s == 1) q = num;
    else if (s == 2) { if (num & 1) return false; q = num >> 1; }
    else { if (num % s) return false; q = num / s; }
    return q < extent;
}

// branch-free form for strides 1 and 2 (all the model has): num = q * s exactly, 0 <= q < extent
OTAL_HD bool div_stride12(int num, int s, int extent, int& q) {
    const int sm1 = s - 1;                   // 0 or 1
    q = num >> sm1;
    return (num >= 0) & ((num & sm1) == 0) & (q < extent);
}

// data-gradient gather: which output-gradient element reaches input position `i` through tap `k`
// (k.c is the OUTPUT channel here)?
OTAL_HD bool conv_src_of_input(const ConvGeom& g, const PosDec& i, const TapDec& k, int64_t& off) {
    int to, ho, wo;
    if (!div_stride(i.t + g.pt - k.dt, g.st, g.To, to)) return false;
    if (!div_stride(i.h + g.ph - k.dh, g.sh, g.Ho, ho)) return false;
    if (!div_stride(i.w + g.pw - k.dw, g.sw, g.Wo, wo)) return false;
    if (g.nlev > 1) {   // packed levels: stride 1, the output column must sit in the same level
        int lo, up;
        level_bounds(g, i.t, g.Ti, lo, up);
        if (to < lo || to >= up) return false;
    }
    off = (int64_t)i.b * g.y_bs + (int64_t)k.c * g.y_cs + ((int64_t)to * g.Ho + ho) * g.Wo + wo;
    return true;
}

OTAL_HD int64_t conv_out_offset(const ConvGeom& g, const PosDec& o, int co) {
    return (int64_t)o.b * g.y_bs + (int64_t)co * g.y_cs + ((int64_t)o.t * g.Ho + o.h) * g.Wo + o.w;
}
OTAL_HD int64_t conv_in_offset(const ConvGeom& g, const PosDec& i, int ci) {
    return (int64_t)i.b * g.x_bs + (int64_t)ci * g.x_cs + ((int64_t)i.t * g.Hi + i.h) * g.Wi + i.w;
}

// SAME padding split of the reference (layers.py:198-210, i3d_backbone.py:46-72)
inline void same_pad(int size, int k, int s, int& front, int& out) {
    int total = (size % s == 0) ? (k - s) : (k - size % s);
    if (total < 0) total = 0;
    front = total / 2;
    out = (size + total - k) / s + 1;
}
