// opental_amd/csrc/pool3d.hip -- MaxPool3dSamePadding forward / backward for (B,C,T,H,W) maps.
//
// Replaces MaxPool3dSamePadding (AFSD/common/layers.py:9-35): F.pad with ZEROS (not -inf) followed
// by nn.MaxPool3d, i.e. two ATen launches and a padded copy per pool in the reference.  Here the
// padding is virtual (out-of-range taps contribute the value 0.0), the forward records which tap
// won (uint8; 255 = a padded zero won, its gradient is dropped exactly as the reference's slice
// of the padded gradient drops it), and the backward is a deterministic gather over the <= 27
// windows that cover an input element -- no atomics.  Ties keep the first tap in (t,h,w) scan
// order, as aten::max_pool3d does.
//
// Both directions stage their operand planes in LDS (zero halo = the reference's zero padding): a block owns a
// run of t-planes of one (b,c) slab, copies the planes it needs with coalesced loads ONCE, and every tap /
// covering-window read is an LDS read.  The first version read each tap from global memory (27 scalar loads per
// output, 54 per input gradient) and was bound by the texture addresser (~14 CU cycles per wave load), not by
// HBM: 5.8 ms per training step; see profiles/.  The per-element kernels remain as the fallback for planes that
// do not fit the LDS budget.
#include "common.h"
#include <cstdlib>

#include "conv_index.h"

namespace {

struct PoolGeom {
    int B, C, Ti, Hi, Wi, To, Ho, Wo;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int64_t x_bs, x_cs, y_bs, y_cs;
    FastDiv fWo, fHo, fWi, fHi;
    // LDS-staged kernels
    int HL, WL;            // forward: staged input plane incl. halo  ((Ho-1)*sh + kh, (Wo-1)*sw + kw)
    int HLo, WLo, ho_min, wo_min;   // backward: staged dy/arg plane incl. halo, first staged ho / wo (<= 0)
    FastDiv fPo, fPi, fPL, fWL, fPLo, fWLo;   // Ho*Wo, Hi*Wi, HL*WL, WL, HLo*WLo, WLo
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
    // unroll factors of the window loops: the window itself, or 1 for the run-time-sized instance (nothing to unroll there:
    // a bare `#pragma unroll` on a loop with a run-time trip count only earns a -Wpass-failed warning)
    static constexpr int UT = KT ? KT : 1, UH = KT ? KH : 1, UW = KT ? KW : 1;
    static constexpr int UCT = KT ? (KT + ST - 1) / ST : 1, UCH = KT ? (KH + SH - 1) / SH : 1, UCW = KT ? (KW + SW - 1) / SW : 1;
};

// ---- bf16-STORED pool tensors (template flag H of the row-per-thread and strided kernels; ops.HALF_STORAGE).  A max-pool
// commutes with the monotonic rounding, and every consumer of a pooled backbone map rounds its operand to bf16 anyway, so
// the forward values do not change; backward sums are formed in fp32 and rounded (to nearest even) once, at the store.
__device__ __forceinline__ float h2f_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h2f_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned f2h_pair_exact(float lo, float hi) {      // both values ARE bf16 values: no rounding
    return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}
__device__ __forceinline__ unsigned f2h_pair_rne(float lo, float hi) {        // v_cvt_pk_bf16_f32
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2));
}
// a run of N (even) consecutive elements of a fp32 / bf16 tensor <-> registers; `base` is the tensor seen as float*,
// `off` the ELEMENT offset of the run (a multiple of VW; the caller guarantees the alignment of VW elements)
// `ok` false: the run reads as zeros.  The address must be VALID either way (callers clamp it): the load is unconditional and
// the value is masked afterwards -- `ok ? load : 0` makes the compiler put every load into its own branch with a full
// s_waitcnt behind it, one exposed memory round trip per piece (tools/isa_loads.sh shows it).
template <int N, int VW, bool H>
__device__ __forceinline__ void load_run(const float* base, int64_t off, bool ok, float (&a)[N]) {
    static_assert(VW == 4 || VW == 2, "vector width");
    const unsigned keep = ok ? ~0u : 0u;
    if constexpr (H) {
        const unsigned short* src = reinterpret_cast<const unsigned short*>(base) + off;
#pragma unroll
        for (int q = 0; q < N / VW; ++q) {
            if constexpr (VW == 4) {
                uint2 t = *reinterpret_cast<const uint2*>(src + 4 * q);
                t.x &= keep; t.y &= keep;
                a[4 * q] = h2f_lo(t.x); a[4 * q + 1] = h2f_hi(t.x); a[4 * q + 2] = h2f_lo(t.y); a[4 * q + 3] = h2f_hi(t.y);
            } else {
                const unsigned t = *reinterpret_cast<const unsigned*>(src + 2 * q) & keep;
                a[2 * q] = h2f_lo(t); a[2 * q + 1] = h2f_hi(t);
            }
        }
    } else {
        const unsigned* src = reinterpret_cast<const unsigned*>(base + off);
#pragma unroll
        for (int q = 0; q < N / VW; ++q) {
            if constexpr (VW == 4) {
                const uint4 t4 = *reinterpret_cast<const uint4*>(src + 4 * q);
                a[4 * q] = __uint_as_float(t4.x & keep); a[4 * q + 1] = __uint_as_float(t4.y & keep);
                a[4 * q + 2] = __uint_as_float(t4.z & keep); a[4 * q + 3] = __uint_as_float(t4.w & keep);
            } else {
                const uint2 t2 = *reinterpret_cast<const uint2*>(src + 2 * q);
                a[2 * q] = __uint_as_float(t2.x & keep); a[2 * q + 1] = __uint_as_float(t2.y & keep);
            }
        }
    }
}
// the same run as RAW words: loaded now, unpacked where it is used -- any arithmetic on a loaded value (even the bf16 -> fp32
// shift) inside a conditional block makes the compiler wait for the load at the end of that block
template <int N, bool H>
struct RawRun { unsigned w[H ? N / 2 : N]; };
template <int N, int VW, bool H>
__device__ __forceinline__ void load_raw(const float* base, int64_t off, RawRun<N, H>& r) {
    constexpr int NW = H ? N / 2 : N, PW = H ? VW / 2 : VW;        // words in all / per piece
    const unsigned* src = H ? reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(base) + off)
                            : reinterpret_cast<const unsigned*>(base + off);
#pragma unroll
    for (int q = 0; q < NW / PW; ++q) {
        if constexpr (PW == 4) {
            const uint4 t = *reinterpret_cast<const uint4*>(src + 4 * q);
            r.w[4 * q] = t.x; r.w[4 * q + 1] = t.y; r.w[4 * q + 2] = t.z; r.w[4 * q + 3] = t.w;
        } else if constexpr (PW == 2) {
            const uint2 t = *reinterpret_cast<const uint2*>(src + 2 * q);
            r.w[2 * q] = t.x; r.w[2 * q + 1] = t.y;
        } else {
            r.w[q] = src[q];
        }
    }
}
template <int N, bool H>
__device__ __forceinline__ void unpack_run(const RawRun<N, H>& r, float (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if constexpr (H) a[i] = (i & 1) ? h2f_hi(r.w[i >> 1]) : h2f_lo(r.w[i >> 1]);
        else a[i] = __uint_as_float(r.w[i]);
    }
}
template <int N, int VW, bool H, bool EXACT>
__device__ __forceinline__ void store_run(float* base, int64_t off, const float (&a)[N]) {
    if constexpr (H) {
        unsigned short* dst = reinterpret_cast<unsigned short*>(base) + off;
#pragma unroll
        for (int q = 0; q < N / VW; ++q) {
            if constexpr (VW == 4) {
                uint2 t;
                t.x = EXACT ? f2h_pair_exact(a[4 * q], a[4 * q + 1]) : f2h_pair_rne(a[4 * q], a[4 * q + 1]);
                t.y = EXACT ? f2h_pair_exact(a[4 * q + 2], a[4 * q + 3]) : f2h_pair_rne(a[4 * q + 2], a[4 * q + 3]);
                *reinterpret_cast<uint2*>(dst + 4 * q) = t;
            } else {
                *reinterpret_cast<unsigned*>(dst + 2 * q) = EXACT ? f2h_pair_exact(a[2 * q], a[2 * q + 1]) : f2h_pair_rne(a[2 * q], a[2 * q + 1]);
            }
        }
    } else {
        float* dst = base + off;
#pragma unroll
        for (int q = 0; q < N / VW; ++q) {
            if constexpr (VW == 4) *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
            else *reinterpret_cast<float2*>(dst + 2 * q) = make_float2(a[2 * q], a[2 * q + 1]);
        }
    }
}

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
#pragma unroll S::UT
    for (int dt = 0; dt < kt; ++dt) {
        const int ti = to * S::st(g) + dt - g.pt;
#pragma unroll S::UH
        for (int dh = 0; dh < kh; ++dh) {
            const int hi = ho * S::sh(g) + dh - g.ph;
#pragma unroll S::UW
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
#pragma unroll S::UCT
    for (int a_ = 0; a_ < ct; ++a_) {
        const int to = to0 - a_, dt = tn - to * st;
        const bool okt = to >= 0 && to < g.To && dt < kt;
        const int toc = min(max(to, 0), g.To - 1);
#pragma unroll S::UCH
        for (int b_ = 0; b_ < ch; ++b_) {
            const int ho = ho0 - b_, dh = hn - ho * sh;
            const bool okh = okt && ho >= 0 && ho < g.Ho && dh < kh;
            const int hoc = min(max(ho, 0), g.Ho - 1);
#pragma unroll S::UCW
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


// ---- LDS-staged forward: block = TT output t-planes of one (b,c) slab
template <int KT, int KH, int KW, int ST, int SH, int SW>
__global__ __launch_bounds__(256) void maxpool3d_fwd_lds_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                unsigned char* __restrict__ arg, PoolGeom g, int TT) {
    using S = PoolShape<KT, KH, KW, ST, SH, SW>;
    extern __shared__ float sm[];
    const int kt = S::kt(g), kh = S::kh(g), kw = S::kw(g);
    const int st = S::st(g), sh = S::sh(g), sw = S::sw(g);
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const int to0 = blockIdx.x * TT;
    const int tt = min(TT, g.To - to0);                     // output planes of this block
    const int TL = (tt - 1) * st + kt;                      // staged input planes
    const int PL = g.HL * g.WL;
    const float* xb = x + (int64_t)b * g.x_bs + (int64_t)c * g.x_cs;
    const int ti0 = to0 * st - g.pt;
    for (int i = threadIdx.x; i < TL * PL; i += 256) {
        const uint32_t tl = fd_div(g.fPL, (uint32_t)i);
        const uint32_t r = i - tl * PL;
        const uint32_t hl = fd_div(g.fWL, r);
        const int wl = (int)(r - hl * g.WL);
        const int ti = ti0 + (int)tl, hi = (int)hl - g.ph, wi = wl - g.pw;
        const bool in = (unsigned)ti < (unsigned)g.Ti && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
        const int64_t off = in ? ((int64_t)ti * g.Hi + hi) * g.Wi + wi : 0;
        const float v = xb[off];
        sm[i] = in ? v : 0.f;
    }
    __syncthreads();
    const int Po = g.Ho * g.Wo;
    const int P = g.To * Po;
    for (int i = threadIdx.x; i < tt * Po; i += 256) {
        const uint32_t tq = fd_div(g.fPo, (uint32_t)i);
        const uint32_t r = i - tq * Po;
        const uint32_t ho = fd_div(g.fWo, r);
        const int wo = (int)(r - ho * g.Wo);
        const float* base = sm + ((int)tq * st * g.HL + (int)ho * sh) * g.WL + wo * sw;
        float best = 0.f;
        int win = 0;
        bool first = true;
#pragma unroll S::UT
        for (int dt = 0; dt < kt; ++dt)
#pragma unroll S::UH
            for (int dh = 0; dh < kh; ++dh)
#pragma unroll S::UW
                for (int dw = 0; dw < kw; ++dw) {
                    const float v = base[(dt * g.HL + dh) * g.WL + dw];
                    if (first || v > best || v != v) {
                        best = v;
                        win = (dt * kh + dh) * kw + dw;
                        first = false;
                    }
                }
        // a padded zero that won is recorded as 255 (its gradient is dropped)
        const int wdt = win / (kh * kw), wr = win - wdt * (kh * kw), wdh = wr / kw, wdw = wr - wdh * kw;
        const int ti = (to0 + (int)tq) * st - g.pt + wdt, hi = (int)ho * sh - g.ph + wdh, wi = wo * sw - g.pw + wdw;
        const bool in = (unsigned)ti < (unsigned)g.Ti && (unsigned)hi < (unsigned)g.Hi && (unsigned)wi < (unsigned)g.Wi;
        const int p = to0 * Po + i;
        y[(int64_t)b * g.y_bs + (int64_t)c * g.y_cs + p] = best;
        arg[(int64_t)bc * P + p] = (unsigned char)(in ? win : 255);
    }
}

// ---- LDS-staged backward: block = TI input t-planes of one (b,c) slab; dy (zero halo) and the winner bytes of
// every output plane that can cover them are staged, then each dx element gathers its <= 27 covering windows.
template <int KT, int KH, int KW, int ST, int SH, int SW>
__global__ __launch_bounds__(256) void maxpool3d_bwd_lds_kernel(const float* __restrict__ dy,
                                                                const unsigned char* __restrict__ arg,
                                                                float* __restrict__ dx, PoolGeom g, int accumulate,
                                                                const float* __restrict__ emask,
                                                                const float* __restrict__ escale, int TI, int TLo_max) {
    using S = PoolShape<KT, KH, KW, ST, SH, SW>;
    extern __shared__ float sm[];
    const int kt = S::kt(g), kh = S::kh(g), kw = S::kw(g);
    const int st = S::st(g), sh = S::sh(g), sw = S::sw(g);
    constexpr int CT = KT ? (KT + ST - 1) / ST : 0, CH = KT ? (KH + SH - 1) / SH : 0, CW = KT ? (KW + SW - 1) / SW : 0;
    const int ct = KT ? CT : (kt + st - 1) / st, ch = KT ? CH : (kh + sh - 1) / sh, cw = KT ? CW : (kw + sw - 1) / sw;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const int ti0 = blockIdx.x * TI;
    const int tin = min(TI, g.Ti - ti0);
    const int toA = (ti0 + g.pt) / st - (ct - 1);           // first output plane that may cover the tile (may be < 0)
    const int toB = (ti0 + tin - 1 + g.pt) / st;            // last
    const int TLo = toB - toA + 1;
    const int PLo = g.HLo * g.WLo;
    unsigned char* sa = reinterpret_cast<unsigned char*>(sm + TLo_max * PLo);
    const int Po = g.Ho * g.Wo;
    const float* dyb = dy + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs;
    const unsigned char* ab = arg + (int64_t)bc * g.To * Po;
    for (int i = threadIdx.x; i < TLo * PLo; i += 256) {
        const uint32_t tl = fd_div(g.fPLo, (uint32_t)i);
        const uint32_t r = i - tl * PLo;
        const uint32_t hl = fd_div(g.fWLo, r);
        const int wl = (int)(r - hl * g.WLo);
        const int to = toA + (int)tl, ho = (int)hl + g.ho_min, wo = wl + g.wo_min;
        const bool in = (unsigned)to < (unsigned)g.To && (unsigned)ho < (unsigned)g.Ho && (unsigned)wo < (unsigned)g.Wo;
        const int o = in ? (to * g.Ho + ho) * g.Wo + wo : 0;
        const float d = dyb[o];
        const unsigned char a = ab[o];
        sm[i] = in ? d : 0.f;
        sa[i] = a;
    }
    __syncthreads();
    const int Pi = g.Hi * g.Wi;
    for (int i = threadIdx.x; i < tin * Pi; i += 256) {
        const uint32_t tq = fd_div(g.fPi, (uint32_t)i);
        const uint32_t r = i - tq * Pi;
        const uint32_t hi = fd_div(g.fWi, r);
        const int wi = (int)(r - hi * g.Wi);
        const int tn = ti0 + (int)tq + g.pt, hn = (int)hi + g.ph, wn = wi + g.pw;
        const int to0 = tn / st, ho0 = hn / sh, wo0 = wn / sw;
        float acc = 0.f;
#pragma unroll S::UCT
        for (int a_ = 0; a_ < ct; ++a_) {
            const int to = to0 - a_, dt = tn - to * st;
#pragma unroll S::UCH
            for (int b_ = 0; b_ < ch; ++b_) {
                const int ho = ho0 - b_, dh = hn - ho * sh;
#pragma unroll S::UCW
                for (int c_ = 0; c_ < cw; ++c_) {
                    const int wo = wo0 - c_, dw = wn - wo * sw;
                    const int idx = ((to - toA) * g.HLo + (ho - g.ho_min)) * g.WLo + (wo - g.wo_min);
                    const bool ok = dt < kt && dh < kh && dw < kw;
                    const float d = sm[idx];
                    const int a = sa[idx];
                    acc += (ok && a == (dt * kh + dh) * kw + dw) ? d : 0.f;     // halo dy is 0: no range test
                }
            }
        }
        const int p = ti0 * Pi + i;
        const int64_t off = (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + p;
        if (emask) acc = emask[off] > 0.f ? acc * escale[c] : 0.f;   // ReLU/BN backward of the pooled layer
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
    if ((int64_t)g.B * g.C > 65535) return OTAL_E_UNSUPPORTED;
    if ((int64_t)g.Ti * g.Hi * g.Wi >= (1LL << 31)) return OTAL_E_SHAPE;
    g.x_bs = s[0]; g.x_cs = s[1]; g.y_bs = s[2]; g.y_cs = s[3];
    g.fWo = make_fastdiv(g.Wo); g.fHo = make_fastdiv(g.Ho); g.fWi = make_fastdiv(g.Wi); g.fHi = make_fastdiv(g.Hi);
    g.HL = (g.Ho - 1) * g.sh + g.kh; g.WL = (g.Wo - 1) * g.sw + g.kw;
    const int ch = (g.kh + g.sh - 1) / g.sh, cw = (g.kw + g.sw - 1) / g.sw;
    g.ho_min = -(ch - 1); g.wo_min = -(cw - 1);
    g.HLo = (g.Hi - 1 + g.ph) / g.sh - g.ho_min + 1; g.WLo = (g.Wi - 1 + g.pw) / g.sw - g.wo_min + 1;
    g.fPo = make_fastdiv((uint32_t)(g.Ho * g.Wo)); g.fPi = make_fastdiv((uint32_t)(g.Hi * g.Wi));
    g.fPL = make_fastdiv((uint32_t)(g.HL * g.WL)); g.fWL = make_fastdiv((uint32_t)g.WL);
    g.fPLo = make_fastdiv((uint32_t)(g.HLo * g.WLo)); g.fWLo = make_fastdiv((uint32_t)g.WLo);
    return 0;
}

// ---- 3x3x3 / stride 1 / pad 1 on P x P planes (the nine Inception branch pools: P = 12, 6, 3), SEPARABLE.  max over the 3x3x3 window = max_dt PM[t+dt-1][h][w], PM[t'][h][w] = max_dh
// RM[t'][h+dh-1][w], RM[t'][h''][w] = max_dw x[t'][h''][w+dw-1] (zero halo everywhere): three 1-D maxima of three values
// instead of 27 compare/select pairs per output (the flat kernels are VALU-bound: 156 us for the 254 MB of Mixed_3b's
// pool, a third of the HBM rate).  Taking the FIRST maximum at every stage selects the lexicographically first
// (dt,dh,dw) among the window's maxima -- exactly the flat scan's winner.  Each stage's tap is a property of its own
// cell, so the byte of position p holds tapW of RM[p] (bits 1:0), tapH of PM[p] (3:2) and tapT of out[p] (5:4), and the
// backward pass routes gradients back through the three stages: 9 candidate reads per input element instead of 27,
// summed in ascending tap order per stage (deterministic; association differs from a flat scan by fp32 rounding only).
__device__ __forceinline__ void first_max3(float a, float b, float c, float& v, int& k) {
    v = a; k = 0;
    if (b > v || b != b) { v = b; k = 1; }
    if (c > v || c != c) { v = c; k = 2; }
}

template <int P>
__global__ __launch_bounds__(256) void maxpool333_sep_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 unsigned char* __restrict__ arg, PoolGeom g, int TT, int vec) {
    extern __shared__ float sm[];
    constexpr int Q = P + 2, PL = Q * Q, PP = P * P, CR = Q * P;
    const int tid = threadIdx.x;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const int to0 = blockIdx.x * TT;
    const int tt = min(TT, g.To - to0);
    const int TL = tt + 2;
    float* xs = sm;                                              // [TL][Q][Q] input with zero halo; later pm [TL][P][P]
    float* rm = sm + (TT + 2) * PL;                              // [TL][Q][P] row maxima
    unsigned char* tw = reinterpret_cast<unsigned char*>(rm + (TT + 2) * CR);   // [TL][Q][P]
    unsigned char* th = tw + (TT + 2) * CR;                      // [TL][P][P]
    const float* xb = x + (int64_t)b * g.x_bs + (int64_t)c * g.x_cs;
    constexpr int VW = P % 4 == 0 ? 4 : (P % 2 == 0 ? 2 : 1);          // interior rows as float4 / float2 pieces
    if (VW > 1 && vec) {
        // zero everything (halo included), then overwrite the interior with vector loads: 1.4 float4 loads per thread and
        // tile instead of 7.7 dword loads (the scalar staging made the kernel texture-addresser-bound)
        for (int i = tid; i < TL * PL; i += 256) xs[i] = 0.f;
        __syncthreads();
        constexpr int NQ = P / VW;
        for (int i = tid; i < TL * P * NQ; i += 256) {
            const int tl = i / (P * NQ), r = i - tl * (P * NQ), hi = r / NQ, q = r - hi * NQ;
            const int ti = to0 - 1 + tl;
            if ((unsigned)ti >= (unsigned)g.Ti) continue;
            float* dst = xs + (tl * Q + hi + 1) * Q + q * VW + 1;
            if constexpr (VW == 4) {
                const float4 v = *reinterpret_cast<const float4*>(xb + (ti * P + hi) * P + q * 4);
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            } else {
                const float2 v = *reinterpret_cast<const float2*>(xb + (ti * P + hi) * P + q * 2);
                dst[0] = v.x; dst[1] = v.y;
            }
        }
    } else {
        for (int i = tid; i < TL * PL; i += 256) {
            const int tl = i / PL, r = i - tl * PL, hl = r / Q, wl = r - hl * Q;
            const int ti = to0 - 1 + tl, hi = hl - 1, wi = wl - 1;
            const bool in = (unsigned)ti < (unsigned)g.Ti && (unsigned)hi < (unsigned)P && (unsigned)wi < (unsigned)P;
            const float v = xb[in ? (ti * P + hi) * P + wi : 0];
            xs[i] = in ? v : 0.f;
        }
    }
    __syncthreads();
    {   // rows: one thread per (row, column) of a plane, walking the planes
        constexpr int G = 256 / CR;
        const int grp = tid / CR, cell = tid - grp * CR;
        const int hl = cell / P, w = cell - hl * P;
        if (grp < G)
            for (int tl = grp; tl < TL; tl += G) {
                const float* r = xs + (tl * Q + hl) * Q + w;
                float v; int k;
                first_max3(r[0], r[1], r[2], v, k);
                rm[(tl * Q + hl) * P + w] = v;
                tw[(tl * Q + hl) * P + w] = (unsigned char)k;
            }
    }
    __syncthreads();
    constexpr int G2 = 256 / PP;
    const int grp = tid / PP, cell = tid - grp * PP;
    const int h = cell / P, w = cell - h * P;
    float* pm = xs;
    if (grp < G2)
        for (int tl = grp; tl < TL; tl += G2) {
            const float* r = rm + (tl * Q + h) * P + w;
            float v; int k;
            first_max3(r[0], r[P], r[2 * P], v, k);
            pm[tl * PP + cell] = v;
            th[tl * PP + cell] = (unsigned char)k;
        }
    __syncthreads();
    const int Pn = g.To * PP;
    if (grp < G2)
        for (int tq = grp; tq < tt; tq += G2) {
            const float* r = pm + tq * PP + cell;
            float v; int k;
            first_max3(r[0], r[PP], r[2 * PP], v, k);
            const int p = (to0 + tq) * PP + cell;
            y[(int64_t)b * g.y_bs + (int64_t)c * g.y_cs + p] = v;
            arg[(int64_t)bc * Pn + p] = (unsigned char)(tw[((tq + 1) * Q + h + 1) * P + w] | (th[(tq + 1) * PP + cell] << 2) | (k << 4));
        }
}

// ---- the same pooling with one ROW per thread (planes of 12 x 12 and 6 x 6).  The cell-per-thread kernel above spends
// ~45 instructions per element on scalar LDS traffic (three passes of 3 reads + 2 writes of one value each) and is VALU /
// LDS-issue bound at 2.5 - 3 TB/s.  Here a thread owns a whole row of P values: the w pass runs in registers, the h and t
// passes exchange whole rows through LDS with 16-byte (8-byte for P = 6) accesses -- six times fewer LDS instructions,
// ~18 VALU instructions per element -- and the row is loaded and stored as vectors.  A workgroup covers TT output planes
// of one (sample, channel) plus one halo plane on each side (256 / P rows).  Same values and tap bytes as the kernel
// above: every stage is the FIRST maximum of three with the zero padding taking part.
// H: x and y are STORED as bf16 (8-byte / 4-byte row pieces); the maxima are taken on the exact fp32 images of the bf16 values.
template <int P, bool H = false>
__global__ __launch_bounds__(256) void maxpool333_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  unsigned char* __restrict__ arg, PoolGeom g, int TT) {
    constexpr int VW = P % 4 == 0 ? 4 : 2, NV = P / VW;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const int to0 = blockIdx.x * TT;
    const int tt = min(TT, g.To - to0);
    const int rows = (tt + 2) * P;
    float* srm = sm;                                             // [rows][P] row maxima
    float* spm = sm + (TT + 2) * P * P;                          // [rows][P] plane maxima
    const int r = tid;
    const bool act = r < rows;
    const int tl = r / P, h = r - tl * P;
    const int ti = to0 - 1 + tl;
    float v[P + 2];
    v[0] = 0.f; v[P + 1] = 0.f;
    {
        const bool in = act && (unsigned)ti < (unsigned)g.Ti;
        float row[P];
        load_run<P, VW, H>(x, (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + ((int64_t)(in ? ti : 0) * P + h) * P, in, row);
#pragma unroll
        for (int w = 0; w < P; ++w) v[1 + w] = row[w];
    }
    auto put = [&](float* dst, const float (&a)[P]) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if constexpr (VW == 4) *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
            else *reinterpret_cast<float2*>(dst + 2 * q) = make_float2(a[2 * q], a[2 * q + 1]);
        }
    };
    auto get = [&](const float* src, bool ok, float (&a)[P]) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if constexpr (VW == 4) {
                const float4 t4 = ok ? *reinterpret_cast<const float4*>(src + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                a[4 * q] = t4.x; a[4 * q + 1] = t4.y; a[4 * q + 2] = t4.z; a[4 * q + 3] = t4.w;
            } else {
                const float2 t2 = ok ? *reinterpret_cast<const float2*>(src + 2 * q) : make_float2(0.f, 0.f);
                a[2 * q] = t2.x; a[2 * q + 1] = t2.y;
            }
        }
    };
    float rm[P], pm[P];
    unsigned tw = 0, th = 0;
#pragma unroll
    for (int w = 0; w < P; ++w) {
        int k;
        first_max3(v[w], v[w + 1], v[w + 2], rm[w], k);
        tw |= (unsigned)k << (2 * w);
    }
    if (act) put(srm + r * P, rm);
    __syncthreads();
    if (act) {
        float up[P], dn[P];
        get(srm + (r - 1) * P, h > 0, up);
        get(srm + (r + 1) * P, h < P - 1, dn);
#pragma unroll
        for (int w = 0; w < P; ++w) {
            int k;
            first_max3(up[w], rm[w], dn[w], pm[w], k);
            th |= (unsigned)k << (2 * w);
        }
        put(spm + r * P, pm);
    }
    __syncthreads();
    if (act && tl >= 1 && tl <= tt) {
        float before[P], after[P], out[P];
        get(spm + (r - P) * P, true, before);
        get(spm + (r + P) * P, true, after);
        unsigned bytes[P / 4 + 1] = {};
#pragma unroll
        for (int w = 0; w < P; ++w) {
            int k;
            first_max3(before[w], pm[w], after[w], out[w], k);
            const unsigned byte = ((tw >> (2 * w)) & 3u) | (((th >> (2 * w)) & 3u) << 2) | ((unsigned)k << 4);
            bytes[w >> 2] |= byte << (8 * (w & 3));
        }
        const int64_t p = ((int64_t)(to0 + tl - 1) * P + h) * P;
        store_run<P, VW, H, true>(y, (int64_t)b * g.y_bs + (int64_t)c * g.y_cs + p, out);
        unsigned char* ab = arg + (int64_t)bc * g.To * P * P + p;
        if constexpr (P % 4 == 0) {
#pragma unroll
            for (int q = 0; q < P / 4; ++q) reinterpret_cast<unsigned*>(ab)[q] = bytes[q];
        } else {                                                // P = 6: rows of 6 bytes, 2-byte aligned
#pragma unroll
            for (int q = 0; q < P / 2; ++q)
                reinterpret_cast<unsigned short*>(ab)[q] = (unsigned short)((bytes[q >> 1] >> (16 * (q & 1))) & 0xffffu);
        }
    }
}

constexpr int POOL_SEP_ELEMS = 1152;     // input elements per backward workgroup (8 planes of 12x12, 32 of 6x6, 128 of 3x3)
// V4: every tensor is 16-byte aligned and the plane size a multiple of 4 -> float4 / uchar4 global accesses (the scalar
// version issued 27 memory instructions per thread and tile: texture-addresser-bound at 2.5 TB/s)
template <int P, bool V4>
__global__ __launch_bounds__(256) void maxpool333_sep_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                 float* __restrict__ dx, PoolGeom g, int accumulate,
                                                                 const float* __restrict__ emask, const float* __restrict__ escale) {
    extern __shared__ float sm[];
    constexpr int PP = P * P, TI = POOL_SEP_ELEMS / PP, VW = V4 ? 4 : 1, J = (POOL_SEP_ELEMS / VW + 255) / 256;
    static_assert(!V4 || PP % 4 == 0, "vector path: planes of a multiple of 4 elements");
    const int tid = threadIdx.x;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const int ti0 = blockIdx.x * TI;
    const int tin = min(TI, g.Ti - ti0);
    const int n = tin * PP;
    float* dys = sm;                                             // [tin + 2][PP]: dy of output planes ti0-1 .. ti0+tin
    float* gp = dys + (TI + 2) * PP;                             // [tin][PP] gradient w.r.t. the plane maxima
    float* gr = gp + TI * PP;                                    // [tin][PP] gradient w.r.t. the row maxima
    unsigned char* tp = reinterpret_cast<unsigned char*>(gr + TI * PP);         // [tin + 2][PP] tap bytes (0xff: no plane)
    const float* dyb = dy + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs;
    const unsigned char* ab = arg + (int64_t)bc * g.To * PP;
    const int64_t xoff = (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + ti0 * PP;
    // epilogue operands first: their latency hides behind the three LDS stages
    float mk[J][VW], old[J][VW];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int i = min((tid + 256 * j) * VW, n - VW);
        if constexpr (V4) {
            const float4 m4 = emask ? *reinterpret_cast<const float4*>(emask + xoff + i) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 o4 = accumulate ? *reinterpret_cast<const float4*>(dx + xoff + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            mk[j][0] = m4.x; mk[j][1] = m4.y; mk[j][2] = m4.z; mk[j][3] = m4.w;
            old[j][0] = o4.x; old[j][1] = o4.y; old[j][2] = o4.z; old[j][3] = o4.w;
        } else {
            mk[j][0] = emask ? emask[xoff + i] : 1.f;
            old[j][0] = accumulate ? dx[xoff + i] : 0.f;
        }
    }
    const int first = (ti0 - 1) * PP, total = g.To * PP;
    for (int i = tid * VW; i < (tin + 2) * PP; i += 256 * VW) {
        const int o = first + i;
        const bool in = o >= 0 && o < total;                     // whole planes: a vector never straddles the tensor's ends
        if constexpr (V4) {
            const float4 d4 = in ? *reinterpret_cast<const float4*>(dyb + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            const unsigned t4 = in ? *reinterpret_cast<const unsigned*>(ab + o) : 0xffffffffu;
            *reinterpret_cast<float4*>(dys + i) = d4;
            *reinterpret_cast<unsigned*>(tp + i) = t4;
        } else {
            dys[i] = in ? dyb[in ? o : 0] : 0.f;
            tp[i] = in ? ab[in ? o : 0] : (unsigned char)0xff;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {                         // through the t stage: out planes t+1-dt, dt = 0, 1, 2
        float s = 0.f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int j = i + (2 - dt) * PP;
            s += ((tp[j] >> 4) & 3) == dt ? dys[j] : 0.f;
        }
        gp[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {                         // through the h stage: plane-max cells h+1-dh of the same plane
        const int r = i % PP, h2 = r / P;
        float s = 0.f;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hh = h2 + 1 - dh;
            const int j = i + (1 - dh) * P;
            if ((unsigned)hh < (unsigned)P) s += ((tp[j + PP] >> 2) & 3) == dh ? gp[j] : 0.f;
        }
        gr[i] = s;
    }
    __syncthreads();
    const float esc = emask ? escale[c] : 1.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {                                // through the w stage: row-max cells w+1-dw of the same row
        const int i0 = (tid + 256 * j) * VW;
        if (i0 >= n) break;
        float res[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            const int i = i0 + e;
            const int w_in = i % P;
            float s = 0.f;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int ww = w_in + 1 - dw;
                const int q = i + (1 - dw);
                if ((unsigned)ww < (unsigned)P) s += (tp[q + PP] & 3) == dw ? gr[q] : 0.f;
            }
            if (emask) s = mk[j][e] > 0.f ? s * esc : 0.f;       // ReLU/BN backward of the pooled layer
            res[e] = old[j][e] + s;
        }
        if constexpr (V4) *reinterpret_cast<float4*>(dx + xoff + i0) = make_float4(res[0], res[1], res[2], res[3]);
        else dx[xoff + i0] = res[0];
    }
}

// Backward of the same pools with one ROW of the input per thread: the t stage reads the dy / tap rows of the three output
// planes that can point into the row (staged once per workgroup in LDS), the h stage exchanges the plane-stage gradient
// rows through LDS, the w stage and the fused ReLU / BN mask run in registers; every global access is a vector.  Same
// sums in the same order as maxpool333_sep_bwd_kernel (ascending tap per stage).
// H: dy, dx and the mask are STORED as bf16; an accumulating call reads the bf16 dx, adds this pool's contribution in fp32 and
// rounds once (the module-input gradient has two producers: the fused 1x1 data gradient stores, this kernel adds).
template <int P, bool H = false>
__global__ __launch_bounds__(256) void maxpool333_rows_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                  float* __restrict__ dx, PoolGeom g, int TI, int accumulate,
                                                                  const float* __restrict__ emask, const float* __restrict__ escale) {
    constexpr int VW = P % 4 == 0 ? 4 : 2, NV = P / VW, TB = P % 4 == 0 ? P : 8;      // tap row pitch in LDS (bytes)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const int ti0 = blockIdx.x * TI;
    const int tin = min(TI, g.Ti - ti0);
    const int rows = (tin + 2) * P;
    float* sdy = sm;                                             // [rows][P] dy of output planes ti0-1 .. ti0+tin
    float* sgp = sm + (TI + 2) * P * P;                          // [rows][P] gradient w.r.t. the plane maxima
    unsigned char* stp = reinterpret_cast<unsigned char*>(sgp + (TI + 2) * P * P);      // [rows][TB] tap bytes (0xff: no plane)
    const int r = tid;
    const bool act = r < rows;
    const int tl = r / P, h = r - tl * P;
    const int to = ti0 - 1 + tl;                                 // the output plane this thread stages / the input plane it owns
    const bool inside = act && tl >= 1 && tl <= tin;
    const int64_t xoff = (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + ((int64_t)(inside ? to : 0) * P + h) * P;
    auto getg = [&](const float* src, bool ok, float (&a)[P]) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if constexpr (VW == 4) {
                const float4 t4 = ok ? *reinterpret_cast<const float4*>(src + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                a[4 * q] = t4.x; a[4 * q + 1] = t4.y; a[4 * q + 2] = t4.z; a[4 * q + 3] = t4.w;
            } else {
                const float2 t2 = ok ? *reinterpret_cast<const float2*>(src + 2 * q) : make_float2(0.f, 0.f);
                a[2 * q] = t2.x; a[2 * q + 1] = t2.y;
            }
        }
    };
    auto put = [&](float* dst, const float (&a)[P]) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if constexpr (VW == 4) *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
            else *reinterpret_cast<float2*>(dst + 2 * q) = make_float2(a[2 * q], a[2 * q + 1]);
        }
    };
    auto taps = [&](const unsigned char* src, unsigned (&t)[3]) {   // a tap row as bytes packed into words
        if constexpr (P % 4 == 0) {
#pragma unroll
            for (int q = 0; q < P / 4; ++q) t[q] = reinterpret_cast<const unsigned*>(src)[q];
        } else {
            t[0] = reinterpret_cast<const unsigned*>(src)[0]; t[1] = reinterpret_cast<const unsigned*>(src)[1]; t[2] = 0;
        }
    };
    // epilogue operands first: their latency hides behind the LDS stages
    RawRun<P, H> mkr = {}, oldr = {};                                  // (threads outside the tile never use them; xoff is clamped)
    if (emask) load_raw<P, VW, H>(emask, xoff, mkr);                   // uniform conditions, unconditional loads, no use until the end
    if (accumulate) load_raw<P, VW, H>(dx, xoff, oldr);
    {   // stage this thread's dy row and tap row
        float d[P];
        const bool in = act && (unsigned)to < (unsigned)g.To;
        const int64_t p = ((int64_t)(in ? to : 0) * P + h) * P;
        load_run<P, VW, H>(dy, (int64_t)b * g.y_bs + (int64_t)c * g.y_cs + p, in, d);
        if (act) put(sdy + r * P, d);
        const unsigned char* ab = arg + (int64_t)bc * g.To * P * P + p;
        if (act) {
            if constexpr (P % 4 == 0) {
#pragma unroll
                for (int q = 0; q < P / 4; ++q)
                    reinterpret_cast<unsigned*>(stp + r * TB)[q] = reinterpret_cast<const unsigned*>(ab)[q] | (in ? 0u : 0xffffffffu);
            } else {
                const unsigned short* a16 = reinterpret_cast<const unsigned short*>(ab);    // (clamped address: always readable)
                const unsigned none = in ? 0u : 0xffffffffu;
                reinterpret_cast<unsigned*>(stp + r * TB)[0] = ((unsigned)a16[0] | ((unsigned)a16[1] << 16)) | none;
                reinterpret_cast<unsigned*>(stp + r * TB)[1] = ((unsigned)a16[2] | 0xffff0000u) | none;
            }
        }
    }
    __syncthreads();
    auto tap_of = [&](const unsigned (&t)[3], int w) -> unsigned { return (t[w >> 2] >> (8 * (w & 3))) & 0xffu; };
    float gp[P];
    unsigned town[3];
    if (inside) {   // through the t stage: output planes to + 1 - dt, dt = 0, 1, 2
        float d0[P], d1[P], d2[P];
        unsigned t0[3], t2[3];
        getg(sdy + (r + P) * P, true, d0); getg(sdy + r * P, true, d1); getg(sdy + (r - P) * P, true, d2);
        taps(stp + (r + P) * TB, t0); taps(stp + r * TB, town); taps(stp + (r - P) * TB, t2);
#pragma unroll
        for (int w = 0; w < P; ++w) {
            float s_ = 0.f;
            s_ += ((tap_of(t0, w) >> 4) & 3u) == 0u ? d0[w] : 0.f;
            s_ += ((tap_of(town, w) >> 4) & 3u) == 1u ? d1[w] : 0.f;
            s_ += ((tap_of(t2, w) >> 4) & 3u) == 2u ? d2[w] : 0.f;
            gp[w] = s_;
        }
        put(sgp + r * P, gp);
    }
    __syncthreads();
    if (!inside) return;
    float gr[P];
    {   // through the h stage: plane-max cells h + 1 - dh of the same plane
        float gu[P], gd[P];
        unsigned tu[3] = {}, td[3] = {};
        getg(sgp + (r + 1) * P, h + 1 < P, gd);                 // dh = 0: row h + 1
        getg(sgp + (r - 1) * P, h >= 1, gu);                    // dh = 2: row h - 1
        if (h + 1 < P) taps(stp + (r + 1) * TB, td);
        if (h >= 1) taps(stp + (r - 1) * TB, tu);
#pragma unroll
        for (int w = 0; w < P; ++w) {
            float s_ = 0.f;
            if (h + 1 < P) s_ += ((tap_of(td, w) >> 2) & 3u) == 0u ? gd[w] : 0.f;
            s_ += ((tap_of(town, w) >> 2) & 3u) == 1u ? gp[w] : 0.f;
            if (h >= 1) s_ += ((tap_of(tu, w) >> 2) & 3u) == 2u ? gu[w] : 0.f;
            gr[w] = s_;
        }
    }
    const float esc = emask ? escale[c] : 1.f;
    float res[P], mk[P], old[P];
    unpack_run<P, H>(mkr, mk);
    unpack_run<P, H>(oldr, old);
#pragma unroll
    for (int w = 0; w < P; ++w) {   // through the w stage: row-max cells w + 1 - dw of the same row
        float s_ = 0.f;
        if (w + 1 < P) s_ += (tap_of(town, w + 1 < P ? w + 1 : w) & 3u) == 0u ? gr[w + 1 < P ? w + 1 : w] : 0.f;
        s_ += (tap_of(town, w) & 3u) == 1u ? gr[w] : 0.f;
        if (w >= 1) s_ += (tap_of(town, w >= 1 ? w - 1 : 0) & 3u) == 2u ? gr[w >= 1 ? w - 1 : 0] : 0.f;
        if (emask) s_ = mk[w] > 0.f ? s_ * esc : 0.f;            // ReLU / BN backward of the pooled layer
        res[w] = old[w] + s_;
    }
    store_run<P, VW, H, false>(dx, xoff, res);
}

// ---- MaxPool3d_2a / 3a (kernel (1,3,3), stride (1,2,2)) and MaxPool3d_4a ((3,3,3) / (2,2,2)): SAME padding = one zero
// plane / row / column at the END of each strided axis, planes with even height and a width that is a multiple of 4.
// The generic kernels issue one dword load per tap (9 or 27 per output, every second one wasted by the stride):
// texture-addresser-bound at 3 of 8 TB/s.  Here a thread owns TWO neighbouring outputs and reads each of its input rows
// as one aligned float4 + one dword (forward), or owns a 2 x 4 block of inputs and reads the outputs that can point into
// it (backward: 6 per candidate output plane); results are identical to the generic kernels (same scan order, same
// first-maximum rule, padding zeros take part, winner-is-padding = 255).  KT = 1 (stride 1 in t) or 3 (stride 2 in t).
__device__ __forceinline__ void scan_tap(float v, bool in, int tap, bool first, float& best, int& win) {
    if (first || v > best || v != v) { best = v; win = in ? tap : 255; }
}

// signbits (optional): one byte per 2 x 4 block of INPUT elements -- bit i*4+j = (x[row 2a+i][col 4m+j] > 0) -- at
// ((bc*Ti + t)*(Hi/2) + a)*(Wi/4) + m: the ReLU mask of the producing layer, which the backward kernel then reads as one
// byte per thread instead of two float4 of the 4-byte activations (604 MB -> 19 MB for MaxPool3d_2a).
// XH: the pool INPUT is stored as bf16 (the producing convolution wrote it so): rows are read as 8 + 2 bytes.
__device__ __forceinline__ float bf16_bits_to_float(unsigned h) { return __uint_as_float(h << 16); }
// YH (with XH): the pool OUTPUT is stored as bf16 too -- the winners are bf16 values already, nothing is rounded.
template <int KT, bool XH, bool YH = false>
__global__ __launch_bounds__(256) void maxpoolk33_s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                unsigned char* __restrict__ arg, PoolGeom g, FastDiv fW2,
                                                                unsigned char* __restrict__ signbits) {
    constexpr int ST = KT == 3 ? 2 : 1;
    const int W2 = g.Wo >> 1;
    const int p2 = blockIdx.x * 256 + threadIdx.x;
    if (p2 >= g.To * g.Ho * W2) return;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const uint32_t q = fd_div(fW2, p2);
    const int m = p2 - q * W2;
    const uint32_t t = fd_div(g.fHo, q);
    const int ho = q - t * g.Ho;
    const int64_t xo = (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + ((int64_t)t * ST * g.Hi + 2 * ho) * g.Wi + 4 * m;
    const float* xr = x + xo;
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(x) + xo;
    const bool cin = 4 * m + 4 < g.Wi;                 // the fifth column exists (else it is the zero pad)
    float4 v[KT][3];
    float e[KT][3];
    bool rin[KT][3];
#pragma unroll
    for (int dt = 0; dt < KT; ++dt)
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            rin[dt][dh] = (int)t * ST + dt < g.Ti && 2 * ho + dh < g.Hi;
            // branch-free: rows / columns outside the tensor are read from a clamped address and masked to the zero pad, so all
            // 2 KT x 3 loads of a thread are in flight together (a conditional load costs one exposed round trip EACH: 18 for KT = 3)
            const int roff = (min(dt, g.Ti - 1 - (int)t * ST) * g.Hi + min(dh, g.Hi - 1 - 2 * ho)) * g.Wi;
            if constexpr (XH) {
                const unsigned short* r = xh + roff;
                uint2 w = *reinterpret_cast<const uint2*>(r);
                const unsigned keep = rin[dt][dh] ? ~0u : 0u;
                w.x &= keep; w.y &= keep;
                v[dt][dh] = make_float4(bf16_bits_to_float(w.x & 0xffffu), bf16_bits_to_float(w.x >> 16),
                                        bf16_bits_to_float(w.y & 0xffffu), bf16_bits_to_float(w.y >> 16));
                e[dt][dh] = bf16_bits_to_float((unsigned)r[cin ? 4 : 0] & ((rin[dt][dh] && cin) ? 0xffffu : 0u));
            } else {
                const float* r = xr + roff;
                const float4 w = *reinterpret_cast<const float4*>(r);
                const unsigned keep = rin[dt][dh] ? ~0u : 0u, keep5 = (rin[dt][dh] && cin) ? ~0u : 0u;
                v[dt][dh] = make_float4(__uint_as_float(__float_as_uint(w.x) & keep), __uint_as_float(__float_as_uint(w.y) & keep),
                                        __uint_as_float(__float_as_uint(w.z) & keep), __uint_as_float(__float_as_uint(w.w) & keep));
                e[dt][dh] = __uint_as_float(__float_as_uint(r[cin ? 4 : 0]) & keep5);
            }
        }
    float b0 = 0.f, b1 = 0.f;
    int w0 = 255, w1 = 255;
#pragma unroll
    for (int dt = 0; dt < KT; ++dt)
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int tap = (dt * 3 + dh) * 3;
            const bool first = dt == 0 && dh == 0, in = rin[dt][dh];
            scan_tap(v[dt][dh].x, in, tap + 0, first, b0, w0);
            scan_tap(v[dt][dh].y, in, tap + 1, false, b0, w0);
            scan_tap(v[dt][dh].z, in, tap + 2, false, b0, w0);
            scan_tap(v[dt][dh].z, in, tap + 0, first, b1, w1);
            scan_tap(v[dt][dh].w, in, tap + 1, false, b1, w1);
            scan_tap(e[dt][dh], in && cin, tap + 2, false, b1, w1);
        }
    const int p = ((int)t * g.Ho + ho) * g.Wo + 2 * m;
    if constexpr (YH) {
        static_assert(XH, "a bf16 pool output needs a bf16 input (the values are copied, not rounded)");
        *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(y) + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs + p) = f2h_pair_exact(b0, b1);
    } else {
        *reinterpret_cast<float2*>(y + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs + p) = make_float2(b0, b1);
    }
    *reinterpret_cast<unsigned short*>(arg + (int64_t)bc * g.To * g.Ho * g.Wo + p) = (unsigned short)(w0 | (w1 << 8));
    if (signbits) {             // this thread read the input planes t*ST .. t*ST + ST - 1, rows 2 ho and 2 ho + 1, columns 4m .. 4m+3 in full
#pragma unroll
        for (int dt = 0; dt < ST; ++dt) {
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 r = v[dt][i];
                bits |= (unsigned)(r.x > 0.f) << (i * 4) | (unsigned)(r.y > 0.f) << (i * 4 + 1) | (unsigned)(r.z > 0.f) << (i * 4 + 2) |
                        (unsigned)(r.w > 0.f) << (i * 4 + 3);
            }
            signbits[(((int64_t)bc * g.Ti + (int)t * ST + dt) * (g.Hi >> 1) + ho) * (g.Wi >> 2) + m] = (unsigned char)bits;
        }
    }
}

// XH: dx is STORED as bf16 (round to nearest even -- the rounding the consuming weight-gradient GEMM applies anyway);
// no accumulate, mask from the sign bits only.
// DYH: the incoming gradient dy is STORED as bf16.
template <int KT, bool XH, bool DYH = false>
__global__ __launch_bounds__(256) void maxpoolk33_s2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                float* __restrict__ dx, PoolGeom g, int accumulate,
                                                                const float* __restrict__ emask, const float* __restrict__ escale,
                                                                FastDiv fW4, FastDiv fH2, const unsigned char* __restrict__ signbits) {
    const int W4 = g.Wi >> 2, H2 = g.Hi >> 1;
    const int p4 = blockIdx.x * 256 + threadIdx.x;
    if (p4 >= g.Ti * H2 * W4) return;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const uint32_t q = fd_div(fW4, p4);
    const int m = p4 - q * W4;
    const int t = (int)fd_div(fH2, q);
    const int a = q - t * H2;
    const int64_t xo = (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + ((int64_t)t * g.Hi + 2 * a) * g.Wi + 4 * m;
    float4 mk[2], old[2];
    // the producer's ReLU mask as one byte per thread (written by the forward kernel): loaded here, expanded only after the
    // gradient loads below have been issued (expanding it here put a full s_waitcnt in front of them)
    unsigned signbyte = 0u;
    if (signbits) signbyte = signbits[(((int64_t)bc * g.Ti + t) * H2 + a) * W4 + m];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!signbits) mk[i] = (!XH && emask) ? *reinterpret_cast<const float4*>(emask + xo + i * g.Wi) : make_float4(1.f, 1.f, 1.f, 1.f);
        old[i] = (!XH && accumulate) ? *reinterpret_cast<const float4*>(dx + xo + i * g.Wi) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // candidate output planes, ascending tap (dt) order: KT = 1: plane t (dt 0);  KT = 3, stride 2: plane t/2 with
    // dt = t & 1, then -- for even t -- plane t/2 - 1 with dt = 2
    constexpr int NP = KT == 3 ? 2 : 1;
    float D[NP][2][3];
    int A[NP][2][3];
    int base[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const int to = KT == 1 ? t : (t >> 1) - pl;
        const bool pin = KT == 1 || (pl == 0 ? true : ((t & 1) == 0 && to >= 0));
        base[pl] = KT == 1 ? 0 : 9 * (pl == 0 ? (t & 1) : 2);
        const int64_t dyo = (int64_t)b * g.y_bs + (int64_t)c * g.y_cs + (int64_t)(pin ? to : 0) * g.Ho * g.Wo;
        const float* dyb = dy + dyo;
        const unsigned short* dyh = reinterpret_cast<const unsigned short*>(dy) + dyo;
        const unsigned char* ab = arg + ((int64_t)bc * g.To + (pin ? to : 0)) * g.Ho * g.Wo;
        // D[r][k], A[r][k]: output rows a-1 (r = 0) and a (r = 1), output columns 2m-1, 2m, 2m+1
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // columns 2m, 2m+1 as one aligned float2 / ushort, column 2m-1 on its own: 4 loads per row instead of 6
            const int ho = a - 1 + r;
            const bool rin = pin && ho >= 0;
            const int o = rin ? ho * g.Wo + 2 * m : 0;
            const bool lin = rin && m > 0;
            float2 d2;
            float d0;
            if constexpr (DYH) {
                const unsigned w2 = *reinterpret_cast<const unsigned*>(dyh + o);
                d2 = make_float2(h2f_lo(w2), h2f_hi(w2));
                d0 = h2f_lo(dyh[lin ? o - 1 : o]);
            } else {
                d2 = *reinterpret_cast<const float2*>(dyb + o);
                d0 = dyb[lin ? o - 1 : o];
            }
            const unsigned t2 = *reinterpret_cast<const unsigned short*>(ab + o);
            const int t0 = ab[lin ? o - 1 : o];
            D[pl][r][0] = lin ? d0 : 0.f;  A[pl][r][0] = lin ? t0 : 255;
            D[pl][r][1] = rin ? d2.x : 0.f; A[pl][r][1] = rin ? (int)(t2 & 255u) : 255;
            D[pl][r][2] = rin ? d2.y : 0.f; A[pl][r][2] = rin ? (int)(t2 >> 8) : 255;
        }
    }
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;   // ascending tap order per input element
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const int tb = base[pl];
        auto hit = [&](int r, int k, int tap) { return A[pl][r][k] == tb + tap ? D[pl][r][k] : 0.f; };
        s0.x += ((hit(1, 1, 0) + hit(1, 0, 2)) + hit(0, 1, 6)) + hit(0, 0, 8);
        s0.y += hit(1, 1, 1) + hit(0, 1, 7);
        s0.z += ((hit(1, 2, 0) + hit(1, 1, 2)) + hit(0, 2, 6)) + hit(0, 1, 8);
        s0.w += hit(1, 2, 1) + hit(0, 2, 7);
        s1.x += hit(1, 1, 3) + hit(1, 0, 5);
        s1.y += hit(1, 1, 4);
        s1.z += hit(1, 2, 3) + hit(1, 1, 5);
        s1.w += hit(1, 2, 4);
    }
    if (signbits) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            mk[i] = make_float4((float)((signbyte >> (i * 4)) & 1u), (float)((signbyte >> (i * 4 + 1)) & 1u),
                                (float)((signbyte >> (i * 4 + 2)) & 1u), (float)((signbyte >> (i * 4 + 3)) & 1u));
    }
    if (emask || signbits) {
        const float esc = escale[c];
        s0.x = mk[0].x > 0.f ? s0.x * esc : 0.f; s0.y = mk[0].y > 0.f ? s0.y * esc : 0.f;
        s0.z = mk[0].z > 0.f ? s0.z * esc : 0.f; s0.w = mk[0].w > 0.f ? s0.w * esc : 0.f;
        s1.x = mk[1].x > 0.f ? s1.x * esc : 0.f; s1.y = mk[1].y > 0.f ? s1.y * esc : 0.f;
        s1.z = mk[1].z > 0.f ? s1.z * esc : 0.f; s1.w = mk[1].w > 0.f ? s1.w * esc : 0.f;
    }
    if constexpr (XH) {
        unsigned short* dh = reinterpret_cast<unsigned short*>(dx) + xo;
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        auto pk = [](float lo, float hi) { const f2 f = {lo, hi}; return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2)); };
        *reinterpret_cast<uint2*>(dh) = make_uint2(pk(s0.x, s0.y), pk(s0.z, s0.w));
        *reinterpret_cast<uint2*>(dh + g.Wi) = make_uint2(pk(s1.x, s1.y), pk(s1.z, s1.w));
    } else {
        *reinterpret_cast<float4*>(dx + xo) = make_float4(old[0].x + s0.x, old[0].y + s0.y, old[0].z + s0.z, old[0].w + s0.w);
        *reinterpret_cast<float4*>(dx + xo + g.Wi) = make_float4(old[1].x + s1.x, old[1].y + s1.y, old[1].z + s1.z, old[1].w + s1.w);
    }
}

// ---- the (1,3,3)/(1,2,2) pools on bf16-STORED tensors, EIGHT input columns per thread.  With bf16 rows the two-output form above
// moves half the bytes per thread for the same instruction count and stops being bandwidth-bound (MaxPool3d_2a: 135 us for
// 434 MB where the fp32 form took 138 us for 793 MB); here a thread owns four neighbouring outputs and reads each of its three
// input rows as ONE 16-byte load + the ninth column (forward), or owns a 2 x 8 input block and the <= 5 x 2 outputs that can
// point into it (backward).  Same scan order, first-maximum rule, winner bytes, sign bits and -- backward -- the same
// ascending-tap summation order per input element as the kernels above: results are bit-identical to them.  Wi % 8 == 0.
__global__ __launch_bounds__(256) void maxpool133_s2_w8_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                   unsigned char* __restrict__ arg, PoolGeom g, FastDiv fW4,
                                                                   unsigned char* __restrict__ signbits) {
    const int W4 = g.Wo >> 2;
    const int p4 = blockIdx.x * 256 + threadIdx.x;
    if (p4 >= g.To * g.Ho * W4) return;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const uint32_t q = fd_div(fW4, p4);
    const int m = p4 - q * W4;
    const uint32_t t = fd_div(g.fHo, q);
    const int ho = q - t * g.Ho;
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(x) + (int64_t)b * g.x_bs + (int64_t)c * g.x_cs +
                               ((int64_t)t * g.Hi + 2 * ho) * g.Wi + 8 * m;
    const bool cin = 8 * m + 8 < g.Wi;                 // the ninth column exists (else it is the zero pad)
    float v[3][9];
    bool rin[3];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
        rin[dh] = 2 * ho + dh < g.Hi;          // loads from clamped addresses, masked afterwards: all six in flight together
        const unsigned short* r = xh + (rin[dh] ? dh * g.Wi : 0);
        uint4 w = *reinterpret_cast<const uint4*>(r);
        const unsigned keep = rin[dh] ? ~0u : 0u;
        w.x &= keep; w.y &= keep; w.z &= keep; w.w &= keep;
        v[dh][0] = h2f_lo(w.x); v[dh][1] = h2f_hi(w.x); v[dh][2] = h2f_lo(w.y); v[dh][3] = h2f_hi(w.y);
        v[dh][4] = h2f_lo(w.z); v[dh][5] = h2f_hi(w.z); v[dh][6] = h2f_lo(w.w); v[dh][7] = h2f_hi(w.w);
        v[dh][8] = h2f_lo((unsigned)r[cin ? 8 : 0] & ((rin[dh] && cin) ? 0xffffu : 0u));
    }
    float best[4];
    int win[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        best[o] = 0.f; win[o] = 255;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int j = 2 * o + dw;
                const bool in = rin[dh] && (j < 8 || cin);
                scan_tap(v[dh][j], in, dh * 3 + dw, dh == 0 && dw == 0, best[o], win[o]);
            }
    }
    const int p = ((int)t * g.Ho + ho) * g.Wo + 4 * m;
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs + p) =
        make_uint2(f2h_pair_exact(best[0], best[1]), f2h_pair_exact(best[2], best[3]));
    *reinterpret_cast<unsigned*>(arg + (int64_t)bc * g.To * g.Ho * g.Wo + p) =
        (unsigned)win[0] | ((unsigned)win[1] << 8) | ((unsigned)win[2] << 16) | ((unsigned)win[3] << 24);
    if (signbits) {             // rows 2 ho, 2 ho + 1, columns 8m .. 8m+7: two of the 2 x 4 blocks, adjacent bytes
        unsigned bits[2] = {0u, 0u};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) bits[j >> 2] |= (unsigned)(v[i][j] > 0.f) << (i * 4 + (j & 3));
        *reinterpret_cast<unsigned short*>(signbits + (((int64_t)bc * g.Ti + (int)t) * (g.Hi >> 1) + ho) * (g.Wi >> 2) + 2 * m) =
            (unsigned short)(bits[0] | (bits[1] << 8));
    }
}

// The same forward for inputs that are NON-NEGATIVE (the output of a conv + ReLU: what all three strided pools of the
// backbone read; the caller says so, io bit 2).  For non-negative bf16 values the BIT PATTERNS are ordered keys, so a window's
// scan collapses into unsigned maxima of composites (bits << 16 | 15 - tap): the larger value wins, equal values keep the
// EARLIER tap (the larger low field) -- exactly the first-maximum rule of scan_tap -- and the padding (zeros that are never
// larger than tap 0, which is always inside) can never win, so no winner is 255.  Per output 9 v_lshl_or / v_and_or + 4
// v_max3_u32 instead of 9 x (compare, NaN compare, two selects): 298 -> ~150 VALU instructions per thread, same bytes.
// A -0.0 (which a ReLU does not produce) is read as +0.0; NaN cannot occur behind fmaxf(x, 0).  Winners, values and sign bits
// are bit-identical to maxpool133_s2_w8_fwd_kernel on such inputs (tests/test_half_chain_gpu.py).
typedef unsigned short pool_u16x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void maxpool133_s2_w8_nn_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                      unsigned char* __restrict__ arg, PoolGeom g, FastDiv fW4,
                                                                      unsigned char* __restrict__ signbits) {
    const int W4 = g.Wo >> 2;
    const int p4 = blockIdx.x * 256 + threadIdx.x;
    if (p4 >= g.To * g.Ho * W4) return;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const uint32_t q = fd_div(fW4, p4);
    const int m = p4 - q * W4;
    const uint32_t t = fd_div(g.fHo, q);
    const int ho = q - t * g.Ho;
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(x) + (int64_t)b * g.x_bs + (int64_t)c * g.x_cs +
                               ((int64_t)t * g.Hi + 2 * ho) * g.Wi + 8 * m;
    const bool cin = 8 * m + 8 < g.Wi;                 // the ninth column exists (else it is the zero pad)
    unsigned W[3][5];                                   // four words of two columns each + the ninth column (low half)
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
        // branch-free: a row / column outside the tensor is read from a clamped address and masked to the zero pad, so the six
        // loads of a thread are all in flight together (a conditional load costs one exposed memory round trip EACH)
        const bool rin = 2 * ho + dh < g.Hi;
        const unsigned short* r = xh + (rin ? dh * g.Wi : 0);
        const uint4 w = *reinterpret_cast<const uint4*>(r);
        const unsigned keep = rin ? 0x7fff7fffu : 0u;
        W[dh][0] = w.x & keep; W[dh][1] = w.y & keep; W[dh][2] = w.z & keep; W[dh][3] = w.w & keep;
        W[dh][4] = (unsigned)r[cin ? 8 : 0] & ((rin && cin) ? 0x7fffu : 0u);
    }
    unsigned best[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        unsigned rowmax[3];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const unsigned c0 = (W[dh][o] << 16) | (unsigned)(15 - (dh * 3 + 0));
            const unsigned c1 = (W[dh][o] & 0xffff0000u) | (unsigned)(15 - (dh * 3 + 1));
            const unsigned c2 = (W[dh][o + 1] << 16) | (unsigned)(15 - (dh * 3 + 2));
            rowmax[dh] = max(max(c0, c1), c2);
        }
        best[o] = max(max(rowmax[0], rowmax[1]), rowmax[2]);
    }
    const int p = ((int)t * g.Ho + ho) * g.Wo + 4 * m;
    // values: the high halves of the composites; winners: 15 - low nibble (no byte exceeds 15: no borrow between bytes)
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs + p) =
        make_uint2(__builtin_amdgcn_perm(best[1], best[0], 0x07060302u), __builtin_amdgcn_perm(best[3], best[2], 0x07060302u));
    const unsigned lows = __builtin_amdgcn_perm(best[1], best[0], 0x0c0c0400u) | __builtin_amdgcn_perm(best[3], best[2], 0x04000c0cu);
    *reinterpret_cast<unsigned*>(arg + (int64_t)bc * g.To * g.Ho * g.Wo + p) = 0x0f0f0f0fu - lows;
    if (signbits) {             // rows 2 ho, 2 ho + 1, columns 8m .. 8m+7: two of the 2 x 4 blocks, adjacent bytes
        unsigned rowbits[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned acc = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const pool_u16x2 one = {1, 1};
                const unsigned f = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(pool_u16x2, W[i][k]), one));
                acc |= ((f | (f >> 15)) & 3u) << (2 * k);       // bit 2k: column 2k > 0, bit 2k+1: column 2k+1 > 0
            }
            rowbits[i] = acc;
        }
        const unsigned b0 = (rowbits[0] & 15u) | ((rowbits[1] & 15u) << 4), b1 = (rowbits[0] >> 4) | (rowbits[1] & 0xf0u);
        *reinterpret_cast<unsigned short*>(signbits + (((int64_t)bc * g.Ti + (int)t) * (g.Hi >> 1) + ho) * (g.Wi >> 2) + 2 * m) =
            (unsigned short)(b0 | (b1 << 8));
    }
}

// ---- MaxPool3d_4a ((3,3,3) / (2,2,2)) backward on bf16-STORED 12-wide planes, sign-bit mask: a thread owns input planes 2k
// and 2k+1, rows 2a and 2a+1, all twelve columns, and reads the output rows that can point into them -- planes k (dt = 0 for
// plane 2k, dt = 1 for plane 2k+1) and k-1 (dt = 2, plane 2k only), rows a and a-1, six columns each: 16 row loads + 6 sign
// bytes for 96 bytes of dx, where the 2 x 4-block form issues 17 / 9 loads for 16 bytes.  Per input element the same hits are
// added in the same (ascending tap) order as maxpoolk33_s2_bwd_kernel<3, true, true>: bit-identical
// (tests/test_half_chain_gpu.py::test_strided_pools_on_bf16_tensors).  72 -> 56 us for the model's (8,480,128,12,12) map.
// (The forward twin -- one output row per thread, nine 24-byte row loads -- was built too and is SLOWER than the two-output
// form, 72.6 vs 62.9 us: a third of the load instructions, but a third of the threads as well.  Not kept.)
// Wi == 12, Hi even, Ti even, To == Ti / 2.
__global__ __launch_bounds__(256) void maxpool333_s2_w12_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                    float* __restrict__ dx, PoolGeom g, const float* __restrict__ escale,
                                                                    const unsigned char* __restrict__ signbits) {
    const int H2 = g.Hi >> 1;
    const int r = blockIdx.x * 256 + threadIdx.x;            // (k, a)
    if (r >= g.To * H2) return;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const int k = (int)fd_div(g.fHo, (uint32_t)r);           // Ho == Hi / 2
    const int a = r - k * H2;
    const unsigned short* dyh = reinterpret_cast<const unsigned short*>(dy) + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs;
    const unsigned char* ab = arg + (int64_t)bc * g.To * g.Ho * 6;
    // raw rows [pl][rr]: pl 0 = output plane k, 1 = plane k-1; rr 0 = output row a-1, 1 = row a (clamped addresses, masked below)
    uint2 d8[2][2];
    unsigned d4[2][2], t4[2][2], t2[2][2];
    unsigned sraw[2][3] = {};
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int o = (max(k - pl, 0) * g.Ho + max(a - 1 + rr, 0)) * 6;
            d8[pl][rr] = *reinterpret_cast<const uint2*>(dyh + o);
            d4[pl][rr] = *reinterpret_cast<const unsigned*>(dyh + o + 4);
            t4[pl][rr] = *reinterpret_cast<const unsigned*>(ab + o);                 // (6-byte rows: 2-byte aligned -- global loads, any alignment)
            t2[pl][rr] = *reinterpret_cast<const unsigned short*>(ab + o + 4);
        }
#pragma unroll
    for (int q = 0; q < 2; ++q) {                           // (no branch around these loads: without sign bits they read winner bytes nobody uses)
        const unsigned char* sp = signbits ? signbits + (((int64_t)bc * g.Ti + 2 * k + q) * H2 + a) * 3 : ab;
#pragma unroll
        for (int m = 0; m < 3; ++m) sraw[q][m] = sp[m];
    }
    __builtin_amdgcn_sched_barrier(0);                      // all loads issued before the first value is used
    float D[2][2][6];
    int A[2][2][6];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const bool in = (k - pl >= 0) && (a - 1 + rr >= 0);
            const unsigned keep = in ? ~0u : 0u;
            const unsigned w0 = d8[pl][rr].x & keep, w1 = d8[pl][rr].y & keep, w2 = d4[pl][rr] & keep;
            const unsigned tl = t4[pl][rr] | ~keep, th = t2[pl][rr] | ~keep;
            D[pl][rr][0] = h2f_lo(w0); D[pl][rr][1] = h2f_hi(w0); D[pl][rr][2] = h2f_lo(w1);
            D[pl][rr][3] = h2f_hi(w1); D[pl][rr][4] = h2f_lo(w2); D[pl][rr][5] = h2f_hi(w2);
            A[pl][rr][0] = (int)(tl & 255u); A[pl][rr][1] = (int)((tl >> 8) & 255u); A[pl][rr][2] = (int)((tl >> 16) & 255u);
            A[pl][rr][3] = (int)((tl >> 24) & 255u); A[pl][rr][4] = (int)(th & 255u); A[pl][rr][5] = (int)((th >> 8) & 255u);
        }
    // hit(pl, rr, w, tap): output (plane k - pl, row a - 1 + rr, column w) chose `tap` (w = -1: the column left of the plane)
    auto hit = [&](int pl, int rr, int w, int tap) __attribute__((always_inline)) -> float {
        return (w >= 0 && A[pl][rr][w >= 0 ? w : 0] == tap) ? D[pl][rr][w >= 0 ? w : 0] : 0.f;
    };
    // contribution of one (output plane, dt) to input row i (0: row 2a, 1: row 2a+1), column j
    auto contrib = [&](int pl, int base, int i, int j) __attribute__((always_inline)) -> float {
        const int w = j >> 1;
        if (i == 0) {
            if ((j & 1) == 0) return ((hit(pl, 1, w, base + 0) + hit(pl, 1, w - 1, base + 2)) + hit(pl, 0, w, base + 6)) + hit(pl, 0, w - 1, base + 8);
            return hit(pl, 1, w, base + 1) + hit(pl, 0, w, base + 7);
        }
        if ((j & 1) == 0) return hit(pl, 1, w, base + 3) + hit(pl, 1, w - 1, base + 5);
        return hit(pl, 1, w, base + 4);
    };
    const float esc = signbits ? escale[c] : 1.f;
    unsigned short* dh = reinterpret_cast<unsigned short*>(dx) + (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + ((int64_t)(2 * k) * g.Hi + 2 * a) * 12;
#pragma unroll
    for (int q = 0; q < 2; ++q)                             // input plane 2k + q
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float sv[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                float v;
                if (q == 0) { v = 0.f + contrib(0, 0, i, j); v += contrib(1, 18, i, j); }      // dt = 0 from plane k, then dt = 2 from plane k-1
                else v = 0.f + contrib(0, 9, i, j);                                             // dt = 1 from plane k
                if (signbits) v = ((sraw[q][j >> 2] >> (i * 4 + (j & 3))) & 1u) ? v * esc : 0.f;
                sv[j] = v;
            }
            unsigned short* row = dh + ((int64_t)q * g.Hi + i) * 12;
            *reinterpret_cast<uint4*>(row) = make_uint4(f2h_pair_rne(sv[0], sv[1]), f2h_pair_rne(sv[2], sv[3]), f2h_pair_rne(sv[4], sv[5]), f2h_pair_rne(sv[6], sv[7]));
            *reinterpret_cast<uint2*>(row + 8) = make_uint2(f2h_pair_rne(sv[8], sv[9]), f2h_pair_rne(sv[10], sv[11]));
        }
}

__global__ __launch_bounds__(256) void maxpool133_s2_w8_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                   float* __restrict__ dx, PoolGeom g, const float* __restrict__ escale,
                                                                   FastDiv fW8, FastDiv fH2, const unsigned char* __restrict__ signbits) {
    const int W8 = g.Wi >> 3, H2 = g.Hi >> 1;
    const int p8 = blockIdx.x * 256 + threadIdx.x;
    if (p8 >= g.Ti * H2 * W8) return;
    const int bc = blockIdx.y;
    const int b = bc / g.C, c = bc - b * g.C;
    const uint32_t q = fd_div(fW8, p8);
    const int m = p8 - q * W8;
    const int t = (int)fd_div(fH2, q);
    const int a = q - t * H2;
    const int64_t xo = (int64_t)b * g.x_bs + (int64_t)c * g.x_cs + ((int64_t)t * g.Hi + 2 * a) * g.Wi + 8 * m;
    unsigned bits = 0xffffu;
    if (signbits) bits = *reinterpret_cast<const unsigned short*>(signbits + (((int64_t)bc * g.Ti + t) * H2 + a) * (g.Wi >> 2) + 2 * m);
    const unsigned short* dyh = reinterpret_cast<const unsigned short*>(dy) + (int64_t)b * g.y_bs + (int64_t)c * g.y_cs + (int64_t)t * g.Ho * g.Wo;
    const unsigned char* ab = arg + ((int64_t)bc * g.To + t) * g.Ho * g.Wo;
    // D[r][k], A[r][k]: output rows a-1 (r = 0) and a (r = 1), output columns 4m-1 (k = 0) .. 4m+3 (k = 4)
    float D[2][5];
    int A[2][5];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        // clamped addresses + bit masks (not selects, which the compiler turns into a branch around the load with a full wait
        // behind it): the eight loads of a thread are in flight together
        const int ho = a - 1 + r;
        const bool rin = ho >= 0;
        const int o = max(ho, 0) * g.Wo + 4 * m;
        const unsigned keep = rin ? ~0u : 0u;
        uint2 d4 = *reinterpret_cast<const uint2*>(dyh + o);
        d4.x &= keep; d4.y &= keep;
        const unsigned t4 = *reinterpret_cast<const unsigned*>(ab + o) | ~keep;
        const bool lin = rin && m > 0;
        const int ol = o - (m > 0 ? 1 : 0);
        D[r][0] = h2f_lo((unsigned)dyh[ol] & (lin ? 0xffffu : 0u)); A[r][0] = (int)((unsigned)ab[ol] | (lin ? 0u : 255u));
        D[r][1] = h2f_lo(d4.x); A[r][1] = (int)(t4 & 255u);
        D[r][2] = h2f_hi(d4.x); A[r][2] = (int)((t4 >> 8) & 255u);
        D[r][3] = h2f_lo(d4.y); A[r][3] = (int)((t4 >> 16) & 255u);
        D[r][4] = h2f_hi(d4.y); A[r][4] = (int)(t4 >> 24);
    }
    auto hit = [&](int r, int k, int tap) { return A[r][k] == tap ? D[r][k] : 0.f; };
    float s[2][8];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {        // ascending tap order per input element, as maxpoolk33_s2_bwd_kernel sums them
        s[0][2 * qq] = ((hit(1, qq + 1, 0) + hit(1, qq, 2)) + hit(0, qq + 1, 6)) + hit(0, qq, 8);
        s[0][2 * qq + 1] = hit(1, qq + 1, 1) + hit(0, qq + 1, 7);
        s[1][2 * qq] = hit(1, qq + 1, 3) + hit(1, qq, 5);
        s[1][2 * qq + 1] = hit(1, qq + 1, 4);
    }
    if (signbits) {
        const float esc = escale[c];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[i][j] = ((bits >> ((j >> 2) * 8 + i * 4 + (j & 3))) & 1u) ? s[i][j] * esc : 0.f;
    }
    unsigned short* dh = reinterpret_cast<unsigned short*>(dx) + xo;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        *reinterpret_cast<uint4*>(dh + (int64_t)i * g.Wi) = make_uint4(f2h_pair_rne(s[i][0], s[i][1]), f2h_pair_rne(s[i][2], s[i][3]),
                                                                      f2h_pair_rne(s[i][4], s[i][5]), f2h_pair_rne(s[i][6], s[i][7]));
}

// 1: the (1,3,3)/(1,2,2) pools, 3: the (3,3,3)/(2,2,2) pool, 0: not one of them (or misaligned operands)
static inline int strided_k33_kind(const PoolGeom& g, const void* x, const void* y) {
    if (!(g.kh == 3 && g.kw == 3 && g.sh == 2 && g.sw == 2 && g.pt == 0 && g.ph == 0 && g.pw == 0 && g.Hi % 2 == 0 &&
          g.Wi % 4 == 0 && g.Ho == g.Hi / 2 && g.Wo == g.Wi / 2 && g.x_bs % 4 == 0 && g.x_cs % 4 == 0 && g.y_bs % 2 == 0 &&
          g.y_cs % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0) ||
        OTAL_OPT("OTAL_POOL_NO133", 0))
        return 0;
    if (g.kt == 1 && g.st == 1 && g.To == g.Ti) return 1;
    if (g.kt == 3 && g.st == 2 && g.Ti % 2 == 0 && g.To == g.Ti / 2) return 3;
    return 0;
}

constexpr size_t POOL_LDS_BUDGET = 48 * 1024;
// the Inception branch pools: 3x3x3, stride 1, pad 1, square planes of side 12 / 6 / 3 with T unchanged
static inline bool is_333_s1(const PoolGeom& g) {
    return g.kt == 3 && g.kh == 3 && g.kw == 3 && g.st == 1 && g.sh == 1 && g.sw == 1 && g.pt == 1 && g.ph == 1 && g.pw == 1 &&
           g.Hi == g.Wi && (g.Hi == 12 || g.Hi == 6 || g.Hi == 3) && g.To == g.Ti && g.Ho == g.Hi && g.Wo == g.Wi &&
           (int64_t)g.Ti * g.Hi * g.Wi < (1LL << 30);
}
// output planes per block (forward): ~4096 outputs, staged input planes within the LDS budget; 0 = does not fit
int fwd_planes(const PoolGeom& g, size_t& lds) {
    int tt = 4096 / (g.Ho * g.Wo);
    if (tt < 1) tt = 1;
    if (tt > g.To) tt = g.To;
    for (; tt >= 1; --tt) {
        lds = (size_t)((tt - 1) * g.st + g.kt) * g.HL * g.WL * sizeof(float);
        if (lds <= POOL_LDS_BUDGET) return tt;
    }
    return 0;
}
// input planes per block (backward) + the largest number of output planes a block stages
int bwd_planes(const PoolGeom& g, int& tlo_max, size_t& lds) {
    const int tile_elems = OTAL_OPT("OTAL_POOL_TILE_G", 4096);
    int ti = tile_elems / (g.Hi * g.Wi);
    if (ti < 1) ti = 1;
    if (ti > g.Ti) ti = g.Ti;
    const int ct = (g.kt + g.st - 1) / g.st;
    for (; ti >= 1; --ti) {
        tlo_max = (ti - 1 + g.st - 1) / g.st + ct + 1;      // >= toB - toA + 1 for every tile origin
        lds = (size_t)tlo_max * g.HLo * g.WLo * (sizeof(float) + 1) + 16;
        if (lds <= POOL_LDS_BUDGET) return ti;
    }
    return 0;
}

#define OTAL_POOL_DISPATCH(KERNEL, GRID, LDS, ...)                                                              \
    do {                                                                                                         \
        const int kk = g.kt * 100 + g.kh * 10 + g.kw, ss = g.st * 100 + g.sh * 10 + g.sw;                        \
        if (kk == 133 && ss == 122) hipLaunchKernelGGL((KERNEL<1, 3, 3, 1, 2, 2>), GRID, dim3(256), LDS, st_, __VA_ARGS__); \
        else if (kk == 333 && ss == 111) hipLaunchKernelGGL((KERNEL<3, 3, 3, 1, 1, 1>), GRID, dim3(256), LDS, st_, __VA_ARGS__); \
        else if (kk == 333 && ss == 222) hipLaunchKernelGGL((KERNEL<3, 3, 3, 2, 2, 2>), GRID, dim3(256), LDS, st_, __VA_ARGS__); \
        else if (kk == 222 && ss == 222) hipLaunchKernelGGL((KERNEL<2, 2, 2, 2, 2, 2>), GRID, dim3(256), LDS, st_, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<0, 0, 0, 0, 0, 0>), GRID, dim3(256), LDS, st_, __VA_ARGS__);            \
    } while (0)

}  // namespace

// io (forward): bit 0 -- x is stored as bf16, bit 1 -- y is stored as bf16 (only together with bit 0)
static int pool_fwd(const int* geom, const int64_t* strides, const float* x, float* y, unsigned char* argtap,
                    unsigned char* signbits, void* stream, int io = 0) {
    if (!geom || !strides || !x || !y || !argtap) return OTAL_E_NULL;
    const bool nonneg = (io & 4) != 0;      // the caller guarantees x >= +0 (a conv + ReLU output): ordered-key kernels
    io &= 3;
    if (io != 0 && io != 1 && io != 3) return OTAL_E_UNSUPPORTED;
    PoolGeom g;
    if (int e = fill(g, geom, strides)) return e;
    hipStream_t st_ = (hipStream_t)stream;
    if ((signbits || io == 1) && !strided_k33_kind(g, x, y)) return OTAL_E_UNSUPPORTED;
    if (const int kind = strided_k33_kind(g, x, y)) {
        const int n2 = g.To * g.Ho * (g.Wo / 2);
        const dim3 grid((n2 + 255) / 256, g.B * g.C);
        const FastDiv fW2 = make_fastdiv((uint32_t)(g.Wo / 2));
        if (io == 3 && kind == 1 && g.Wi % 8 == 0 && g.x_bs % 8 == 0 && g.x_cs % 8 == 0 && g.y_bs % 4 == 0 && g.y_cs % 4 == 0 &&
            (reinterpret_cast<uintptr_t>(argtap) & 3) == 0 && (!signbits || (reinterpret_cast<uintptr_t>(signbits) & 1) == 0) &&
            !OTAL_OPT("OTAL_POOL_NOW8", 0)) {           // eight input columns per thread
            const int n4 = g.To * g.Ho * (g.Wo / 4);
            if (nonneg && !OTAL_OPT("OTAL_POOL_NOKEYS", 0))
                hipLaunchKernelGGL(maxpool133_s2_w8_nn_fwd_kernel, dim3((n4 + 255) / 256, g.B * g.C), dim3(256), 0, st_, x, y, argtap, g,
                                   make_fastdiv((uint32_t)(g.Wo / 4)), signbits);
            else
                hipLaunchKernelGGL(maxpool133_s2_w8_fwd_kernel, dim3((n4 + 255) / 256, g.B * g.C), dim3(256), 0, st_, x, y, argtap, g,
                                   make_fastdiv((uint32_t)(g.Wo / 4)), signbits);
        } else if (io == 3) {   // bf16 in, bf16 out
            if (kind == 1) hipLaunchKernelGGL((maxpoolk33_s2_fwd_kernel<1, true, true>), grid, dim3(256), 0, st_, x, y, argtap, g, fW2, signbits);
            else hipLaunchKernelGGL((maxpoolk33_s2_fwd_kernel<3, true, true>), grid, dim3(256), 0, st_, x, y, argtap, g, fW2, signbits);
        } else if (io == 1) {   // bf16-stored input (8-byte rows), fp32 output: the (1,3,3)/(1,2,2) pools
            if (kind != 1) return OTAL_E_UNSUPPORTED;
            hipLaunchKernelGGL((maxpoolk33_s2_fwd_kernel<1, true>), grid, dim3(256), 0, st_, x, y, argtap, g, fW2, signbits);
        } else if (kind == 1) hipLaunchKernelGGL((maxpoolk33_s2_fwd_kernel<1, false>), grid, dim3(256), 0, st_, x, y, argtap, g, fW2, signbits);
        else hipLaunchKernelGGL((maxpoolk33_s2_fwd_kernel<3, false>), grid, dim3(256), 0, st_, x, y, argtap, g, fW2, signbits);
        return otal_launch_status();
    }
    if (is_333_s1(g) && !OTAL_OPT("OTAL_POOL_NOLDS", 0)) {
        const int tile_elems = OTAL_OPT("OTAL_POOL_TILE", 1152);
        const int P = g.Hi, Q = P + 2;
        int tt = tile_elems / (P * P);
        tt = tt < 1 ? 1 : (tt > g.To ? g.To : tt);
        auto need = [&](int t) { return (size_t)(t + 2) * (Q * Q + Q * P) * sizeof(float) + (size_t)(t + 2) * (Q * P + P * P); };
        while (tt > 1 && need(tt) > POOL_LDS_BUDGET) --tt;
        const dim3 grid((g.To + tt - 1) / tt, g.B * g.C);
        const int vec = (g.x_bs % 4 == 0 && g.x_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? 1 : 0;
        const bool vy = g.y_bs % 4 == 0 && g.y_cs % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(argtap) & 3) == 0;
        if (io == 3) {          // bf16 in / out: the row-per-thread kernels (12 x 12, 6 x 6 planes; 16-byte aligned channel planes)
            const bool ok = (P == 12 || P == 6) && g.x_bs % 8 == 0 && g.x_cs % 8 == 0 && g.y_bs % 8 == 0 && g.y_cs % 8 == 0 &&
                            ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 &&
                            (reinterpret_cast<uintptr_t>(argtap) & 3) == 0;
            if (!ok) return OTAL_E_UNSUPPORTED;
            const int TT = 256 / P - 2;
            const dim3 rgrid((g.To + TT - 1) / TT, g.B * g.C);
            const size_t lds = (size_t)2 * (TT + 2) * P * P * sizeof(float);
            if (P == 12) hipLaunchKernelGGL((maxpool333_rows_fwd_kernel<12, true>), rgrid, dim3(256), lds, st_, x, y, argtap, g, TT);
            else hipLaunchKernelGGL((maxpool333_rows_fwd_kernel<6, true>), rgrid, dim3(256), lds, st_, x, y, argtap, g, TT);
            return otal_launch_status();
        }
        if (vec && vy && (P == 12 || P == 6) && !OTAL_OPT("OTAL_POOL_NOROWS", 0)) {     // one row per thread
            const int TT = 256 / P - 2;
            const dim3 rgrid((g.To + TT - 1) / TT, g.B * g.C);
            const size_t lds = (size_t)2 * (TT + 2) * P * P * sizeof(float);
            if (P == 12) hipLaunchKernelGGL((maxpool333_rows_fwd_kernel<12>), rgrid, dim3(256), lds, st_, x, y, argtap, g, TT);
            else hipLaunchKernelGGL((maxpool333_rows_fwd_kernel<6>), rgrid, dim3(256), lds, st_, x, y, argtap, g, TT);
            return otal_launch_status();
        }
        if (P == 12) hipLaunchKernelGGL(maxpool333_sep_fwd_kernel<12>, grid, dim3(256), need(tt), st_, x, y, argtap, g, tt, vec);
        else if (P == 6) hipLaunchKernelGGL(maxpool333_sep_fwd_kernel<6>, grid, dim3(256), need(tt), st_, x, y, argtap, g, tt, vec);
        else hipLaunchKernelGGL(maxpool333_sep_fwd_kernel<3>, grid, dim3(256), need(tt), st_, x, y, argtap, g, tt, vec);
        return otal_launch_status();
    }
    if (io) return OTAL_E_UNSUPPORTED;      // bf16 tensors: the strided 3x3 pools and the 12 x 12 / 6 x 6 branch pools only
    size_t lds = 0;
    // staging pays when the taps overlap (stride 1: every input is read kvol times); the strided pools read each input
    // ~2 times and were measured faster with direct loads (r01: 230 vs 514 us for the 1x3x3 / (1,2,2) pool)
    const bool overlap = g.st == 1 && g.sh == 1 && g.sw == 1;
    const int tt = (OTAL_OPT("OTAL_POOL_NOLDS", 0) || !overlap) ? 0 : fwd_planes(g, lds);
    if (tt > 0) {
        const dim3 grid((g.To + tt - 1) / tt, g.B * g.C);
        OTAL_POOL_DISPATCH(maxpool3d_fwd_lds_kernel, grid, lds, x, y, argtap, g, tt);
    } else {
        const dim3 grid((g.To * g.Ho * g.Wo + 255) / 256, g.B * g.C);
        OTAL_POOL_DISPATCH(maxpool3d_fwd_kernel, grid, 0, x, y, argtap, g);
    }
    return otal_launch_status();
}

// io (backward): bit 0 -- dx is stored as bf16, bit 1 -- dy is stored as bf16, bit 2 -- out_mask is a bf16 tensor
static int pool_bwd(const int* geom, const int64_t* strides, const float* dy, const unsigned char* argtap, float* dx,
                    int accumulate, const float* out_mask, const float* out_scale, const unsigned char* signbits, void* stream,
                    int io = 0) {
    if (!geom || !strides || !dy || !dx || !argtap) return OTAL_E_NULL;
    if (((out_mask == nullptr) && (signbits == nullptr)) != (out_scale == nullptr)) return OTAL_E_NULL;
    if (out_mask && signbits) return OTAL_E_NULL;
    const bool all_half = (io & 3) == 3 && (!out_mask || (io & 4));
    if (io != 0 && io != 1 && !all_half) return OTAL_E_UNSUPPORTED;
    PoolGeom g;
    if (int e = fill(g, geom, strides)) return e;
    hipStream_t st_ = (hipStream_t)stream;
    const int kind = (!out_mask || (reinterpret_cast<uintptr_t>(out_mask) & 15) == 0) ? strided_k33_kind(g, dx, dy) : 0;
    if ((signbits || io == 1) && !kind) return OTAL_E_UNSUPPORTED;
    if (io == 1 && kind != 1) return OTAL_E_UNSUPPORTED;
    if (io && kind && (accumulate || out_mask)) return OTAL_E_UNSUPPORTED;      // bf16-stored dx of a strided pool: plain store, sign-bit mask
    if (kind) {
        const int n4 = g.Ti * (g.Hi / 2) * (g.Wi / 4);
        const dim3 grid((n4 + 255) / 256, g.B * g.C);
        const FastDiv fW4 = make_fastdiv((uint32_t)(g.Wi / 4)), fH2 = make_fastdiv((uint32_t)(g.Hi / 2));
        if (all_half && kind == 1 && g.Wi % 8 == 0 && g.x_bs % 8 == 0 && g.x_cs % 8 == 0 && g.y_bs % 4 == 0 && g.y_cs % 4 == 0 &&
            (reinterpret_cast<uintptr_t>(argtap) & 3) == 0 && (!signbits || (reinterpret_cast<uintptr_t>(signbits) & 1) == 0) &&
            !OTAL_OPT("OTAL_POOL_NOW8", 0)) {
            const int n8 = g.Ti * (g.Hi / 2) * (g.Wi / 8);
            hipLaunchKernelGGL(maxpool133_s2_w8_bwd_kernel, dim3((n8 + 255) / 256, g.B * g.C), dim3(256), 0, st_, dy, argtap, dx, g, out_scale,
                               make_fastdiv((uint32_t)(g.Wi / 8)), fH2, signbits);
        } else if (all_half && kind == 1) hipLaunchKernelGGL((maxpoolk33_s2_bwd_kernel<1, true, true>), grid, dim3(256), 0, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale, fW4, fH2, signbits);
        else if (all_half && g.Wi == 12 && g.x_bs % 4 == 0 && g.x_cs % 4 == 0 && g.y_bs % 2 == 0 && g.y_cs % 2 == 0 && !OTAL_OPT("OTAL_POOL_NOW12", 0))
            hipLaunchKernelGGL(maxpool333_s2_w12_bwd_kernel, dim3((g.To * (g.Hi / 2) + 255) / 256, g.B * g.C), dim3(256), 0, st_, dy, argtap, dx, g, out_scale, signbits);
        else if (all_half) hipLaunchKernelGGL((maxpoolk33_s2_bwd_kernel<3, true, true>), grid, dim3(256), 0, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale, fW4, fH2, signbits);
        else if (io == 1) hipLaunchKernelGGL((maxpoolk33_s2_bwd_kernel<1, true>), grid, dim3(256), 0, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale, fW4, fH2, signbits);
        else if (kind == 1) hipLaunchKernelGGL((maxpoolk33_s2_bwd_kernel<1, false>), grid, dim3(256), 0, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale, fW4, fH2, signbits);
        else hipLaunchKernelGGL((maxpoolk33_s2_bwd_kernel<3, false>), grid, dim3(256), 0, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale, fW4, fH2, signbits);
        return otal_launch_status();
    }
    if (is_333_s1(g) && !OTAL_OPT("OTAL_POOL_NOLDS", 0)) {
        const int PP = g.Hi * g.Wi, ti = POOL_SEP_ELEMS / PP;
        const size_t l3 = (size_t)((ti + 2) * PP + 2 * ti * PP) * sizeof(float) + (size_t)(ti + 2) * PP;
        const dim3 grid((g.Ti + ti - 1) / ti, g.B * g.C);
        const bool v4 = g.Hi != 3 && g.x_bs % 4 == 0 && g.x_cs % 4 == 0 && g.y_bs % 4 == 0 && g.y_cs % 4 == 0 &&
                        ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(out_mask) |
                          reinterpret_cast<uintptr_t>(argtap)) & 15) == 0 && ((int64_t)g.To * PP) % 4 == 0;
        const bool v2 = g.x_bs % 4 == 0 && g.x_cs % 4 == 0 && g.y_bs % 4 == 0 && g.y_cs % 4 == 0 &&
                        ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(out_mask) |
                          reinterpret_cast<uintptr_t>(argtap)) & 15) == 0;
        if (io) {               // bf16 dy / dx / mask: the row-per-thread kernels (12 x 12, 6 x 6 planes; 16-byte aligned channel planes)
            const bool ok = all_half && (g.Hi == 12 || g.Hi == 6) && g.x_bs % 8 == 0 && g.x_cs % 8 == 0 && g.y_bs % 8 == 0 && g.y_cs % 8 == 0 &&
                            ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(out_mask) |
                              reinterpret_cast<uintptr_t>(argtap)) & 15) == 0;
            if (!ok) return OTAL_E_UNSUPPORTED;
            const int P = g.Hi, TIr = 256 / P - 2, TB = P == 12 ? 12 : 8;
            const dim3 rgrid((g.Ti + TIr - 1) / TIr, g.B * g.C);
            const size_t lds = (size_t)2 * (TIr + 2) * P * P * sizeof(float) + (size_t)(TIr + 2) * P * TB;
            if (P == 12) hipLaunchKernelGGL((maxpool333_rows_bwd_kernel<12, true>), rgrid, dim3(256), lds, st_, dy, argtap, dx, g, TIr, accumulate, out_mask, out_scale);
            else hipLaunchKernelGGL((maxpool333_rows_bwd_kernel<6, true>), rgrid, dim3(256), lds, st_, dy, argtap, dx, g, TIr, accumulate, out_mask, out_scale);
            return otal_launch_status();
        }
        if (v2 && (g.Hi == 12 || g.Hi == 6) && !OTAL_OPT("OTAL_POOL_NOROWS", 0)) {      // one input row per thread
            const int P = g.Hi, TIr = 256 / P - 2, TB = P == 12 ? 12 : 8;
            const dim3 rgrid((g.Ti + TIr - 1) / TIr, g.B * g.C);
            const size_t lds = (size_t)2 * (TIr + 2) * P * P * sizeof(float) + (size_t)(TIr + 2) * P * TB;
            if (P == 12) hipLaunchKernelGGL((maxpool333_rows_bwd_kernel<12>), rgrid, dim3(256), lds, st_, dy, argtap, dx, g, TIr, accumulate, out_mask, out_scale);
            else hipLaunchKernelGGL((maxpool333_rows_bwd_kernel<6>), rgrid, dim3(256), lds, st_, dy, argtap, dx, g, TIr, accumulate, out_mask, out_scale);
            return otal_launch_status();
        }
        if (g.Hi == 12 && v4) hipLaunchKernelGGL((maxpool333_sep_bwd_kernel<12, true>), grid, dim3(256), l3, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale);
        else if (g.Hi == 6 && v4) hipLaunchKernelGGL((maxpool333_sep_bwd_kernel<6, true>), grid, dim3(256), l3, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale);
        else if (g.Hi == 12) hipLaunchKernelGGL((maxpool333_sep_bwd_kernel<12, false>), grid, dim3(256), l3, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale);
        else if (g.Hi == 6) hipLaunchKernelGGL((maxpool333_sep_bwd_kernel<6, false>), grid, dim3(256), l3, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale);
        else hipLaunchKernelGGL((maxpool333_sep_bwd_kernel<3, false>), grid, dim3(256), l3, st_, dy, argtap, dx, g, accumulate, out_mask, out_scale);
        return otal_launch_status();
    }
    if (io) return OTAL_E_UNSUPPORTED;
    size_t lds = 0;
    int tlo_max = 0;
    const int ti = OTAL_OPT("OTAL_POOL_NOLDS", 0) ? 0 : bwd_planes(g, tlo_max, lds);
    if (ti > 0) {
        const dim3 grid((g.Ti + ti - 1) / ti, g.B * g.C);
        OTAL_POOL_DISPATCH(maxpool3d_bwd_lds_kernel, grid, lds, dy, argtap, dx, g, accumulate, out_mask, out_scale, ti, tlo_max);
    } else {
        const dim3 grid((g.Ti * g.Hi * g.Wi + 255) / 256, g.B * g.C);
        OTAL_POOL_DISPATCH(maxpool3d_bwd_kernel, grid, 0, dy, argtap, dx, g, accumulate, out_mask, out_scale);
    }
    return otal_launch_status();
}

extern "C" int otal_maxpool3d_fwd(const int* geom, const int64_t* strides, const float* x, float* y,
                                  unsigned char* argtap, void* stream) {
    return pool_fwd(geom, strides, x, y, argtap, nullptr, stream);
}
extern "C" int otal_maxpool3d_bwd(const int* geom, const int64_t* strides, const float* dy,
                                  const unsigned char* argtap, float* dx, int accumulate,
                                  const float* out_mask, const float* out_scale, void* stream) {
    return pool_bwd(geom, strides, dy, argtap, dx, accumulate, out_mask, out_scale, nullptr, stream);
}
// The producing layer's ReLU mask as sign bits (strided 3x3 pools only): bytes the _signbits entry points need, 0 = unsupported
extern "C" size_t otal_maxpool3d_signbits_bytes(const int* geom, const int64_t* strides) {
    if (!geom || !strides) return 0;
    PoolGeom g;
    if (fill(g, geom, strides)) return 0;
    if (!strided_k33_kind(g, nullptr, nullptr)) return 0;
    return (size_t)g.B * g.C * g.Ti * (g.Hi / 2) * (g.Wi / 4);
}
extern "C" int otal_maxpool3d_fwd_signbits(const int* geom, const int64_t* strides, const float* x, float* y,
                                           unsigned char* argtap, unsigned char* signbits, void* stream) {
    if (!signbits) return OTAL_E_NULL;
    return pool_fwd(geom, strides, x, y, argtap, signbits, stream);
}
extern "C" int otal_maxpool3d_bwd_signbits(const int* geom, const int64_t* strides, const float* dy, const unsigned char* argtap,
                                           float* dx, int accumulate, const unsigned char* signbits, const float* out_scale,
                                           void* stream) {
    if (!signbits || !out_scale) return OTAL_E_NULL;
    return pool_bwd(geom, strides, dy, argtap, dx, accumulate, nullptr, out_scale, signbits, stream);
}
// The same pair with the LARGE tensor (the pool input x, its gradient dx) stored as bf16 -- (1,3,3)/(1,2,2) pools behind a
// convolution that writes bf16 (otal_conv_fwd precision bit 2): MaxPool3d_2a reads 302 MB instead of 604 MB.
extern "C" int otal_maxpool3d_fwd_signbits_h(const int* geom, const int64_t* strides, const void* x_bf16, float* y,
                                             unsigned char* argtap, unsigned char* signbits, void* stream) {
    if (!signbits) return OTAL_E_NULL;
    return pool_fwd(geom, strides, static_cast<const float*>(x_bf16), y, argtap, signbits, stream, 1);
}
extern "C" int otal_maxpool3d_bwd_signbits_h(const int* geom, const int64_t* strides, const float* dy, const unsigned char* argtap,
                                             void* dx_bf16, const unsigned char* signbits, const float* out_scale, void* stream) {
    if (!signbits || !out_scale) return OTAL_E_NULL;
    return pool_bwd(geom, strides, dy, argtap, static_cast<float*>(dx_bf16), 0, nullptr, out_scale, signbits, stream, 1);
}
/* The general bf16-storage forms (ABI 23): `io` says which tensors are STORED as bf16 (pointers through the float* parameters,
 * strides in elements).  fwd io: bit 0 x, bit 1 y (1: the form above; 3: both -- the strided 3x3 pools and the 3x3x3 branch pools
 * on 12 x 12 / 6 x 6 planes).  bwd io: bit 0 dx, bit 1 dy, bit 2 out_mask (1: the form above; 3 / 7: all of them -- strided pools
 * with the sign-bit mask and a plain store; branch pools with a bf16 mask tensor, and `accumulate` = read dx, add in fp32, round
 * once).  signbits / out_mask nullable as in the fp32 entry points.  Anything else: OTAL_E_UNSUPPORTED. */
extern "C" int otal_maxpool3d_fwd_io(const int* geom, const int64_t* strides, const void* x, void* y, unsigned char* argtap,
                                     unsigned char* signbits, int io, void* stream) {
    return pool_fwd(geom, strides, static_cast<const float*>(x), static_cast<float*>(y), argtap, signbits, stream, io);
}
extern "C" int otal_maxpool3d_bwd_io(const int* geom, const int64_t* strides, const void* dy, const unsigned char* argtap, void* dx,
                                     int accumulate, const void* out_mask, const float* out_scale, const unsigned char* signbits,
                                     int io, void* stream) {
    return pool_bwd(geom, strides, static_cast<const float*>(dy), argtap, static_cast<float*>(dx), accumulate,
                    static_cast<const float*>(out_mask), out_scale, signbits, stream, io);
}
