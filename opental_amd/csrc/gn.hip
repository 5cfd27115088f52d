// opental_amd/csrc/gn.hip -- GroupNorm(G, C) + ReLU, forward and backward, for (B,C,T) maps.
//
// Replaces nn.GroupNorm(32, C) + nn.ReLU after every Unit1D/Unit3D of the pyramid
// (AFSD/thumos14/BDNet.py:67-103,:129-203,:274-284): 2 ATen launches per block in the reference.
// HBM-bound.  A workgroup owns one (sample, group): its cpg*T floats are contiguous in the
// (B,C,T) layout, are read from HBM once with coalesced loads, kept in LDS for the two-pass
// statistics (mean, then centred variance -- no E[x^2]-E[x]^2 cancellation), normalised, and
// written once.  A level table gives each packed pyramid level its own statistics, so one
// launch normalises all six levels of a level-batched tower.
// Backward also emits, per (sample, channel), the partial sums of d_gamma, d_beta and of the
// gradient w.r.t. the convolution bias (= sum_t dx); the tiny reduction over the batch is done by
// the caller.  Everything is computed in a fixed order (deterministic).
#include "common.h"

namespace {

struct GnLevels { int nlev; int lev[OTAL_MAX_LEVELS + 1]; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-wide sum of a per-thread value; result broadcast to all threads.  red: 8 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Visit every (channel c, frame t) of a level of `len` frames with `nlanes` lanes, WITHOUT an integer division per element
// (the flat form `c = i / len` costs ~40 VALU instructions per element on this hardware -- a third of these kernels):
// long levels: channel by channel, lanes along t; short levels: a lane keeps its t and walks the channels, nlanes / len
// channels per trip (one division per level).
template <typename F>
__device__ __forceinline__ void for_level(int lane, int nlanes, int cpg, int len, F f) {
    if (len >= nlanes) {
        for (int c = 0; c < cpg; ++c)
            for (int t = lane; t < len; t += nlanes) f(c, t);
    } else {
        const int cpl = nlanes / len, cl = lane / len, t = lane - cl * len;
        if (cl < cpl)
            for (int c = cl; c < cpg; c += cpl) f(c, t);
    }
}

// global -> LDS copy of a group's map (n floats) by the whole workgroup: float4 pieces when the source is 16-byte aligned
__device__ __forceinline__ void stage_map(float* dst, const float* __restrict__ src, int n, int tid) {
    if (((reinterpret_cast<uintptr_t>(src) | (uintptr_t)(n * 4)) & 15) == 0) {
        for (int i = tid; i < (n >> 2); i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int i = tid; i < n; i += 256) dst[i] = src[i];
    }
}

// x,y: (B,C,T); stats out: (B,G,nlev,2) = {mean, rstd}
// (pair launches, otal_gn_relu_*_pair: grid.y = 2 and `alt` carries the second problem's tensors -- same shapes and options)
struct GnFwdAlt { const float* x; const float* gamma; const float* beta; float* y; float* stats; };
__global__ __launch_bounds__(256) void gn_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          float* __restrict__ stats, int C, int T, int G,
                                                          float eps, int relu, GnLevels L, GnFwdAlt alt,
                                                          int64_t y_bs, int64_t y_cs) {
    if (blockIdx.y) { x = alt.x; gamma = alt.gamma; beta = alt.beta; y = alt.y; stats = alt.stats; }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* buf = reinterpret_cast<float*>(smem);
    float* red = buf + (size_t)(C / G) * T;                      // 8 floats
    float* gam = red + 8;                                        // the group's affine parameters
    const int tid = threadIdx.x;
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int cpg = C / G;
    float* bet = gam + cpg;
    const int64_t base = ((int64_t)b * C + (int64_t)g * cpg) * T;
    const int64_t ybase = (int64_t)b * y_bs + (int64_t)g * cpg * y_cs;      // the destination may be a slice of a wider buffer
    const int n = cpg * T;
    // (one workgroup per (sample, group), a few microseconds long: every dependent memory round trip counts.  The affine
    //  parameters are fetched into registers and parked in LDS only AFTER the map's loads have been issued, and the map
    //  travels as 16-byte pieces where its alignment allows)
    float gv = 0.f, bv = 0.f;
    if (tid < cpg) { gv = gamma[g * cpg + tid]; bv = beta[g * cpg + tid]; }
    stage_map(buf, x + base, n, tid);
    if (tid < cpg) { gam[tid] = gv; bet[tid] = bv; }
    __syncthreads();
    if (L.nlev == 1) {          // one level: the whole workgroup reduces it
        const int cnt = n;
        float s = 0.f;
        for (int i = tid; i < cnt; i += 256) s += buf[i];
        const float mean = block_sum(s, red, tid) / (float)cnt;
        float q = 0.f;
        for (int i = tid; i < cnt; i += 256) { const float d = buf[i] - mean; q += d * d; }
        const float var = block_sum(q, red, tid) / (float)cnt;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (tid == 0) {
            stats[((int64_t)b * G + g) * 2 + 0] = mean;
            stats[((int64_t)b * G + g) * 2 + 1] = rstd;
        }
        for_level(tid, 256, cpg, T, [&](int c, int t) {
            float v = (buf[c * T + t] - mean) * rstd * gam[c] + bet[c];
            if (relu) v = fmaxf(v, 0.f);
            y[ybase + (int64_t)c * y_cs + t] = v;
        });
        return;
    }
    // level-packed maps: one WAVE per level (levels l = wave, wave + 4, ...), wave-local reductions only -- the 12
    // block-wide reductions of the six-level map (two barriers each) were most of this latency-bound kernel (one workgroup
    // per (sample, group), 2016 elements)
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll 1
    for (int l = wave; l < L.nlev; l += 4) {
        const int lo = L.lev[l], len = L.lev[l + 1] - lo;
        const int cnt = cpg * len;
        float s = 0.f;
        for_level(lane, 64, cpg, len, [&](int c, int t) { s += buf[c * T + lo + t]; });
        const float mean = __shfl(wave_sum(s), 0, 64) / (float)cnt;
        float q = 0.f;
        for_level(lane, 64, cpg, len, [&](int c, int t) { const float d = buf[c * T + lo + t] - mean; q += d * d; });
        const float var = __shfl(wave_sum(q), 0, 64) / (float)cnt;
        const float rstd = 1.0f / sqrtf(var + eps);
        if (lane == 0) {
            stats[(((int64_t)b * G + g) * L.nlev + l) * 2 + 0] = mean;
            stats[(((int64_t)b * G + g) * L.nlev + l) * 2 + 1] = rstd;
        }
        for_level(lane, 64, cpg, len, [&](int c, int t) {
            float v = (buf[c * T + lo + t] - mean) * rstd * gam[c] + bet[c];
            if (relu) v = fmaxf(v, 0.f);
            y[ybase + (int64_t)c * y_cs + lo + t] = v;
        });
    }
}

// dx: (B,C,T); partial: (B,3,C) = {sum dyh*xhat, sum dyh, sum dx} (channel-contiguous rows: summing over b leaves
// d_gamma, d_beta and the bias gradient as three contiguous vectors)
struct GnBwdAlt { const float* dy; const float* x; const float* gamma; const float* beta; const float* stats; float* dx; float* partial; int64_t dy_bs; };
// otal_gn_relu_bwd_sum: the output gradient is the SUM of up to three terms, each a (B,C,T_i <= T) map with its own batch /
// channel strides (a level slice of a packed buffer, a map that covers only the first level): what autograd used to add with
// one elementwise launch per term, read here while the map is staged (term order = summation order).
struct GnTerms { int n; int Tn[3]; const float* p[3]; int64_t bs[3]; int64_t cs[3]; };
__global__ __launch_bounds__(256) void gn_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ stats, float* __restrict__ dx,
                                                          float* __restrict__ partial, int C, int T, int G, int relu,
                                                          GnLevels L, int64_t dy_bs, int keep_dx, GnBwdAlt alt, GnTerms terms) {
    if (blockIdx.y) { dy = alt.dy; x = alt.x; gamma = alt.gamma; beta = alt.beta; stats = alt.stats; dx = alt.dx; partial = alt.partial; dy_bs = alt.dy_bs; }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cpg = C / G;
    float* xb = reinterpret_cast<float*>(smem);          // x, later xhat
    float* gb = xb + (size_t)cpg * T;                    // dy masked (dyh), later dx
    // keep_dx: a third array holds dx for the channel sums at the end (the host sets it when the LDS allows; otherwise those
    // sums re-read dx from global memory: four dependent round trips per wave at the tail of a latency-bound kernel)
    float* dxl = gb + (size_t)cpg * T;
    float* red = dxl + (keep_dx ? (size_t)cpg * T : 0);  // 8 floats
    float* gam = red + 8;                                // the group's affine parameters
    float* bet = gam + cpg;
    const int tid = threadIdx.x;
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int64_t base = ((int64_t)b * C + (int64_t)g * cpg) * T;
    const int n = cpg * T;
    float gv = 0.f, bv = 0.f;                                               // (see the forward: registers first, LDS after the map loads)
    if (tid < cpg) { gv = gamma[g * cpg + tid]; bv = beta[g * cpg + tid]; }
    const int64_t dbase = (int64_t)b * dy_bs + (int64_t)g * cpg * T;         // dy may be a channel slice of a wider map
    // level l is handled by the whole workgroup when it is the only one, else by wave l % 4 on its own (see the forward)
    const bool solo = L.nlev == 1;
    const int wv = tid >> 6, ln = tid & 63;
    const int first = solo ? tid : ln, stride = solo ? 256 : 64;
    // the statistics of this wave's levels (at most GN_LPW = 2 per wave for the six-level map; further ones are fetched in the
    // loop): fetched together with the maps, not one dependent round trip per level behind the barrier
    constexpr int GN_LPW = 2;
    float2 st_pre[GN_LPW];
#pragma unroll
    for (int k = 0; k < GN_LPW; ++k) {
        const int l = (solo ? 0 : wv) + k * (solo ? 1 : 4);
        st_pre[k] = make_float2(0.f, 0.f);
        if (l < L.nlev) {
            st_pre[k].x = stats[(((int64_t)b * G + g) * L.nlev + l) * 2 + 0];
            st_pre[k].y = stats[(((int64_t)b * G + g) * L.nlev + l) * 2 + 1];
        }
    }
    if (terms.n > 0) {          // summed terms: rows of the group wave by wave, lanes along the positions; all loads of a trip together
        for (int c = wv; c < cpg; c += 4)
            for (int t0 = ln; t0 < T; t0 += 128) {
                float v[2][3], xv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + 64 * u;
                    const int tc = t < T ? t : T - 1;
                    xv[u] = x[base + (int64_t)c * T + tc];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const bool ok = k < terms.n && t < terms.Tn[k];
                        const int tk = k < terms.n ? (tc < terms.Tn[k] ? tc : terms.Tn[k] - 1) : 0;
                        const float* q = k < terms.n ? terms.p[k] + (int64_t)b * terms.bs[k] + (int64_t)(g * cpg + c) * terms.cs[k] + tk : x + base;
                        const float w = *q;
                        v[u][k] = ok ? w : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + 64 * u;
                    if (t < T) { xb[c * T + t] = xv[u]; gb[c * T + t] = (v[u][0] + v[u][1]) + v[u][2]; }
                }
            }
    } else if (((reinterpret_cast<uintptr_t>(x + base) | reinterpret_cast<uintptr_t>(dy + dbase) | (uintptr_t)(n * 4)) & 15) == 0) {
        for (int i = tid; i < (n >> 2); i += 256) {          // both maps in one trip: their loads are in flight together
            const float4 xv = reinterpret_cast<const float4*>(x + base)[i], dv = reinterpret_cast<const float4*>(dy + dbase)[i];
            reinterpret_cast<float4*>(xb)[i] = xv;
            reinterpret_cast<float4*>(gb)[i] = dv;
        }
    } else {
        for (int i = tid; i < n; i += 256) { xb[i] = x[base + i]; gb[i] = dy[dbase + i]; }
    }
    if (tid < cpg) { gam[tid] = gv; bet[tid] = bv; }
    __syncthreads();
    int kl = 0;
#pragma unroll 1
    for (int l = solo ? 0 : wv; l < L.nlev; l += solo ? 1 : 4, ++kl) {
        const int lo = L.lev[l], len = L.lev[l + 1] - lo;
        const int cnt = cpg * len;
        float mean, rstd;
        if (kl == 0) { mean = st_pre[0].x; rstd = st_pre[0].y; }
        else if (kl == 1) { mean = st_pre[1].x; rstd = st_pre[1].y; }
        else {
            mean = stats[(((int64_t)b * G + g) * L.nlev + l) * 2 + 0];
            rstd = stats[(((int64_t)b * G + g) * L.nlev + l) * 2 + 1];
        }
        float s1 = 0.f, s2 = 0.f;
        for_level(first, stride, cpg, len, [&](int c, int t) {
            const int p = c * T + lo + t;
            const float ga = gam[c];
            const float xh = (xb[p] - mean) * rstd;
            float d = gb[p];
            if (relu && !(xh * ga + bet[c] > 0.f)) d = 0.f;
            xb[p] = xh;
            gb[p] = d;
            const float dg = d * ga;
            s1 += dg;
            s2 += dg * xh;
        });
        float m1, m2;
        if (solo) {
            m1 = block_sum(s1, red, tid) / (float)cnt;
            m2 = block_sum(s2, red, tid) / (float)cnt;       // block_sum's barriers also make every thread's xb/gb writes visible
        } else {
            m1 = __shfl(wave_sum(s1), 0, 64) / (float)cnt;   // a wave re-reads only what it wrote itself
            m2 = __shfl(wave_sum(s2), 0, 64) / (float)cnt;
        }
        for_level(first, stride, cpg, len, [&](int c, int t) {
            const int p = c * T + lo + t;
            const float v = rstd * (gb[p] * gam[c] - m1 - xb[p] * m2);
            dx[base + (int64_t)c * T + lo + t] = v;
            if (keep_dx) dxl[p] = v;
        });
    }
    __syncthreads();            // xb / gb of every level (and the dx stores the sums below re-read) are complete
    // per-channel sums over all t (all levels), fixed order: one wave per channel, lanes stride t
    const int wave = tid >> 6, lane = tid & 63;
    for (int c = wave; c < cpg; c += 4) {
        const int ch = g * cpg + c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        if (keep_dx) {                                       // (two loops: a pointer select would turn the read into a flat load)
            for (int t = lane; t < T; t += 64) {
                const float d = gb[c * T + t], xh = xb[c * T + t];
                a0 += d * xh;
                a1 += d;
                a2 += dxl[c * T + t];
            }
        } else {
            for (int t = lane; t < T; t += 64) {
                const float d = gb[c * T + t], xh = xb[c * T + t];
                a0 += d * xh;
                a1 += d;
                a2 += dx[base + (int64_t)c * T + t];
            }
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) {
            float* p = partial + (int64_t)b * 3 * C + ch;
            p[0] = a0; p[C] = a1; p[2 * C] = a2;
        }
    }
}

// gfx950 has 160 KB of LDS per CU and one workgroup may own all of it; allocations above the 64 KB default need the
// kernel's dynamic-LDS cap raised once (the 768-frame ActivityNet maps: 16 channels x 768 x 2 arrays = 96 KB).
constexpr size_t LDS_MAX = 160 * 1024;
template <typename K> int allow_large_lds(K kernel, size_t lds, bool& done) {
    if (lds <= 64 * 1024 || done) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)LDS_MAX) != hipSuccess) { (void)hipGetLastError(); return OTAL_E_UNSUPPORTED; }
    done = true;
    return 0;
}

int fill_levels(GnLevels& L, int T, int nlev, const int* lev) {
    if (nlev <= 1 || !lev) { L.nlev = 1; L.lev[0] = 0; for (int i = 1; i <= OTAL_MAX_LEVELS; ++i) L.lev[i] = T; return 0; }
    if (nlev > OTAL_MAX_LEVELS || lev[0] != 0 || lev[nlev] != T) return OTAL_E_LEVELS;
    L.nlev = nlev;
    for (int i = 0; i <= OTAL_MAX_LEVELS; ++i) L.lev[i] = lev[i <= nlev ? i : nlev];
    for (int i = 0; i < nlev; ++i) if (L.lev[i + 1] <= L.lev[i]) return OTAL_E_LEVELS;
    return 0;
}

}  // namespace

extern "C" int otal_gn_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                int B, int C, int T, int G, float eps, int relu, int nlev, const int* lev,
                                void* stream) {
    if (!x || !gamma || !beta || !y || !stats) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G) return OTAL_E_SHAPE;
    GnLevels L;
    if (int e = fill_levels(L, T, nlev, lev)) return e;
    const size_t lds = (size_t)(C / G) * T * 4 + 64 + (size_t)(C / G) * 8;
    if (lds > LDS_MAX) return OTAL_E_UNSUPPORTED;
    static bool large_ok = false;
    if (int e = allow_large_lds(gn_relu_fwd_kernel, lds, large_ok)) return e;
    hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(B * G), dim3(256), lds, (hipStream_t)stream,
                       x, gamma, beta, y, stats, C, T, G, eps, relu, L, GnFwdAlt{}, (int64_t)C * T, (int64_t)T);
    return otal_launch_status();
}

extern "C" int otal_gn_relu_fwd_to(const float* x, const float* gamma, const float* beta, float* y, int64_t y_bs, int64_t y_cs,
                                   float* stats, int B, int C, int T, int G, float eps, int relu, int nlev, const int* lev,
                                   void* stream) {
    if (!x || !gamma || !beta || !y || !stats) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G || y_cs < T || y_bs < 0) return OTAL_E_SHAPE;
    GnLevels L;
    if (int e = fill_levels(L, T, nlev, lev)) return e;
    const size_t lds = (size_t)(C / G) * T * 4 + 64 + (size_t)(C / G) * 8;
    if (lds > LDS_MAX) return OTAL_E_UNSUPPORTED;
    static bool large_ok = false;
    if (int e = allow_large_lds(gn_relu_fwd_kernel, lds, large_ok)) return e;
    hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(B * G), dim3(256), lds, (hipStream_t)stream,
                       x, gamma, beta, y, stats, C, T, G, eps, relu, L, GnFwdAlt{}, y_bs, y_cs);
    return otal_launch_status();
}

extern "C" int otal_gn_relu_fwd_pair(const float* const* x, const float* const* gamma, const float* const* beta, float* const* y,
                                     float* const* stats, int B, int C, int T, int G, float eps, int relu, int nlev,
                                     const int* lev, void* stream) {
    if (!x || !gamma || !beta || !y || !stats) return OTAL_E_NULL;
    for (int i = 0; i < 2; ++i) if (!x[i] || !gamma[i] || !beta[i] || !y[i] || !stats[i]) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G) return OTAL_E_SHAPE;
    GnLevels L;
    if (int e = fill_levels(L, T, nlev, lev)) return e;
    const size_t lds = (size_t)(C / G) * T * 4 + 64 + (size_t)(C / G) * 8;
    if (lds > 64 * 1024) return OTAL_E_UNSUPPORTED;
    const GnFwdAlt alt = {x[1], gamma[1], beta[1], y[1], stats[1]};
    hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(B * G, 2), dim3(256), lds, (hipStream_t)stream,
                       x[0], gamma[0], beta[0], y[0], stats[0], C, T, G, eps, relu, L, alt, (int64_t)C * T, (int64_t)T);
    return otal_launch_status();
}

extern "C" int otal_gn_relu_fwd_pair_to(const float* const* x, const float* const* gamma, const float* const* beta,
                                        float* const* y, int64_t y_bs, int64_t y_cs, float* const* stats, int B, int C, int T,
                                        int G, float eps, int relu, int nlev, const int* lev, void* stream) {
    if (!x || !gamma || !beta || !y || !stats) return OTAL_E_NULL;
    for (int i = 0; i < 2; ++i) if (!x[i] || !gamma[i] || !beta[i] || !y[i] || !stats[i]) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G || y_cs < T || y_bs < 0) return OTAL_E_SHAPE;
    GnLevels L;
    if (int e = fill_levels(L, T, nlev, lev)) return e;
    const size_t lds = (size_t)(C / G) * T * 4 + 64 + (size_t)(C / G) * 8;
    if (lds > 64 * 1024) return OTAL_E_UNSUPPORTED;
    const GnFwdAlt alt = {x[1], gamma[1], beta[1], y[1], stats[1]};
    hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(B * G, 2), dim3(256), lds, (hipStream_t)stream,
                       x[0], gamma[0], beta[0], y[0], stats[0], C, T, G, eps, relu, L, alt, y_bs, y_cs);
    return otal_launch_status();
}

extern "C" int otal_gn_relu_bwd(const float* dy, int64_t dy_batch_stride, const float* x, const float* gamma, const float* beta,
                                const float* stats, float* dx, float* partial, int B, int C, int T, int G,
                                int relu, int nlev, const int* lev, void* stream) {
    if (!dy || !x || !gamma || !beta || !stats || !dx || !partial) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G) return OTAL_E_SHAPE;
    if (dy_batch_stride == 0) dy_batch_stride = (int64_t)C * T;
    if (dy_batch_stride < (int64_t)C * T) return OTAL_E_SHAPE;
    GnLevels L;
    if (int e = fill_levels(L, T, nlev, lev)) return e;
    size_t lds = (size_t)(C / G) * T * 12 + 64 + (size_t)(C / G) * 8;
    const int keep_dx = lds <= LDS_MAX;
    if (!keep_dx) lds -= (size_t)(C / G) * T * 4;
    if (lds > LDS_MAX) return OTAL_E_UNSUPPORTED;
    static bool large_ok = false;
    if (int e = allow_large_lds(gn_relu_bwd_kernel, lds, large_ok)) return e;
    hipLaunchKernelGGL(gn_relu_bwd_kernel, dim3(B * G), dim3(256), lds, (hipStream_t)stream,
                       dy, x, gamma, beta, stats, dx, partial, C, T, G, relu, L, dy_batch_stride, keep_dx, GnBwdAlt{}, GnTerms{});
    return otal_launch_status();
}

extern "C" int otal_gn_relu_bwd_sum(int n_terms, const float* const* dy, const int64_t* dy_bs, const int64_t* dy_cs, const int* dy_T,
                                    const float* x, const float* gamma, const float* beta, const float* stats, float* dx,
                                    float* partial, int B, int C, int T, int G, int relu, int nlev, const int* lev, void* stream) {
    if (!dy || !dy_bs || !dy_cs || !dy_T || !x || !gamma || !beta || !stats || !dx || !partial) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G || n_terms < 1 || n_terms > 3) return OTAL_E_SHAPE;
    GnTerms tm = {};
    tm.n = n_terms;
    for (int k = 0; k < n_terms; ++k) {
        if (!dy[k]) return OTAL_E_NULL;
        if (dy_T[k] <= 0 || dy_T[k] > T || dy_cs[k] < dy_T[k] || dy_bs[k] < 0) return OTAL_E_SHAPE;
        tm.p[k] = dy[k]; tm.bs[k] = dy_bs[k]; tm.cs[k] = dy_cs[k]; tm.Tn[k] = dy_T[k];
    }
    GnLevels L;
    if (int e = fill_levels(L, T, nlev, lev)) return e;
    size_t lds = (size_t)(C / G) * T * 12 + 64 + (size_t)(C / G) * 8;
    const int keep_dx = lds <= LDS_MAX;
    if (!keep_dx) lds -= (size_t)(C / G) * T * 4;
    if (lds > LDS_MAX) return OTAL_E_UNSUPPORTED;
    static bool large_ok = false;
    if (int e = allow_large_lds(gn_relu_bwd_kernel, lds, large_ok)) return e;
    hipLaunchKernelGGL(gn_relu_bwd_kernel, dim3(B * G), dim3(256), lds, (hipStream_t)stream,
                       dy[0], x, gamma, beta, stats, dx, partial, C, T, G, relu, L, (int64_t)C * T, keep_dx, GnBwdAlt{}, tm);
    return otal_launch_status();
}

extern "C" int otal_gn_relu_bwd_pair(const float* const* dy, const int64_t* dy_batch_stride, const float* const* x,
                                     const float* const* gamma, const float* const* beta, const float* const* stats,
                                     float* const* dx, float* const* partial, int B, int C, int T, int G, int relu, int nlev,
                                     const int* lev, void* stream) {
    if (!dy || !dy_batch_stride || !x || !gamma || !beta || !stats || !dx || !partial) return OTAL_E_NULL;
    int64_t bs[2];
    for (int i = 0; i < 2; ++i) {
        if (!dy[i] || !x[i] || !gamma[i] || !beta[i] || !stats[i] || !dx[i] || !partial[i]) return OTAL_E_NULL;
        bs[i] = dy_batch_stride[i] ? dy_batch_stride[i] : (int64_t)C * T;
        if (bs[i] < (int64_t)C * T) return OTAL_E_SHAPE;
    }
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G) return OTAL_E_SHAPE;
    GnLevels L;
    if (int e = fill_levels(L, T, nlev, lev)) return e;
    size_t lds = (size_t)(C / G) * T * 12 + 64 + (size_t)(C / G) * 8;
    const int keep_dx = lds <= 64 * 1024;
    if (!keep_dx) lds -= (size_t)(C / G) * T * 4;
    if (lds > 64 * 1024) return OTAL_E_UNSUPPORTED;
    const GnBwdAlt alt = {dy[1], x[1], gamma[1], beta[1], stats[1], dx[1], partial[1], bs[1]};
    hipLaunchKernelGGL(gn_relu_bwd_kernel, dim3(B * G, 2), dim3(256), lds, (hipStream_t)stream,
                       dy[0], x[0], gamma[0], beta[0], stats[0], dx[0], partial[0], C, T, G, relu, L, bs[0], keep_dx, alt, GnTerms{});
    return otal_launch_status();
}

// ---- batch sums of the backward's partials, MANY layers per launch.  Item i: dst{0,1,2}[i][c] = sum_b partial[i][(b*3 + r)*C + c]
// in ascending b (deterministic).  The trainer defers these sums and runs them when a gradient bucket is flushed: 21
// reduce launches per step become 2-4, and the results land directly in the gradient arena.
namespace {
constexpr int SP_MAX = 32;
struct SumItems { const float* src[SP_MAX]; float* dst[SP_MAX][3]; int C[SP_MAX]; int B[SP_MAX]; };
__global__ __launch_bounds__(256) void sum_partials_kernel(const SumItems it) {
    const int i = blockIdx.y, r = blockIdx.z, c = blockIdx.x * 256 + threadIdx.x;
    const int C = it.C[i], B = it.B[i];
    if (c >= C || !it.dst[i][r]) return;
    const float* p = it.src[i] + (size_t)r * C + c;
    float v[8];
    float s = 0.f;
    for (int b0 = 0; b0 < B; b0 += 8) {            // eight loads in flight, added in ascending b
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = b0 + u < B ? p[(size_t)(b0 + u) * 3 * C] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (b0 + u < B) s += v[u];
    }
    it.dst[i][r][c] = s;
}
}  // namespace

extern "C" int otal_sum_partials(int n_items, const float* const* partial, float* const* dst0, float* const* dst1,
                                 float* const* dst2, const int* channels, const int* batches, void* stream) {
    if (!partial || !dst0 || !dst1 || !dst2 || !channels || !batches) return OTAL_E_NULL;
    if (n_items < 0) return OTAL_E_SHAPE;
    for (int i0 = 0; i0 < n_items; i0 += SP_MAX) {
        SumItems it = {};
        const int n = n_items - i0 < SP_MAX ? n_items - i0 : SP_MAX;
        int cmax = 0;
        for (int i = 0; i < n; ++i) {
            if (!partial[i0 + i] || channels[i0 + i] <= 0 || batches[i0 + i] <= 0) return OTAL_E_SHAPE;
            it.src[i] = partial[i0 + i];
            it.dst[i][0] = dst0[i0 + i]; it.dst[i][1] = dst1[i0 + i]; it.dst[i][2] = dst2[i0 + i];
            it.C[i] = channels[i0 + i]; it.B[i] = batches[i0 + i];
            cmax = channels[i0 + i] > cmax ? channels[i0 + i] : cmax;
        }
        hipLaunchKernelGGL(sum_partials_kernel, dim3((cmax + 255) / 256, n, 3), dim3(256), 0, (hipStream_t)stream, it);
        if (int e = otal_launch_status()) return e;
    }
    return 0;
}
