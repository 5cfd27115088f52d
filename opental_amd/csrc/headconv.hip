// opental_amd/csrc/headconv.hip -- the detection-head convolutions of one stage in ONE launch forward and TWO backward
// (gfx950 / MI355X).
//
// Replaces the skinny Unit1D heads of CoarsePyramid (AFSD/thumos14/BDNet.py:205-272 definitions, :337-353 coarse stage,
// :399-412 refined stage; AFSD/common/layers.py:178-214 Unit1D = SAME pad + nn.Conv1d + bias) and their autograd:
//   coarse stage : loc_head (2, k3) on loc_feat; conf_head (15, k3) and actionness_head (1, k3) on conf_feat
//   refined stage: prop_loc_head (2, k1), center_head (1, k3) on the loc proposal feature; prop_conf_head (15, k1),
//                  prop_actionness_head (1, k1) on the conf proposal feature
// Each is a GEMM with M = 1..15 output channels: on the matrix cores that is 2-47 % of one 32-row tile, and each head cost
// a forward launch plus SIX backward launches (weight transpose, tap table, data gradient, weight gradient, split-K
// reduce, bias sum) of 4-12 us each -- 49 launches and 0.38 ms per step for 56 MFLOP.  Here the heads of a stage are plain
// fp32 FMA loops over LDS-staged (B, C, T) windows, as the north star asks for the temporal heads:
//   * forward : a workgroup owns 8 positions of one sample and one head; the 512 x 10 input window is staged once in LDS,
//               thread (co, ci mod 16) walks its channels with its weights prefetched through registers, 16-lane shuffles
//               finish the sums;
//   * dgrad   : a workgroup owns 16 input channels of one sample and one input map; the output gradients of ALL heads on
//               that map (<= 21 rows, zero halo) and their weight slices sit in LDS; one coalesced store per element, the
//               heads' contributions summed in a fixed order;
//   * wgrad   : a workgroup owns 4 input channels of one input map; lanes = (head row, tap); x windows and the transposed
//               output gradients of all samples in LDS; every (row, channel, tap) is one serial fp32 sum over (b, t) --
//               no split-K slabs, no reduce launch; the bias gradients come out of the same launch.
// Level-packed maps: a tap never crosses a level boundary (same rule as the level-aware convolution kernels).
// Everything is summed in a fixed order (deterministic).
#include "common.h"

namespace {

constexpr int HC_MAX_HEADS = 8, HC_MAX_INPUTS = 4, HC_MAX_ROWS = 21;    // rows = output channels of all heads on one input
constexpr int HC_P = 8;                                                  // positions per forward workgroup

struct HcLevels { int nlev; int lev[OTAL_MAX_LEVELS + 1]; };
struct HcArgs {
    int n_heads, n_inputs, B, C, N;
    int in_idx[HC_MAX_HEADS], cout[HC_MAX_HEADS], k[HC_MAX_HEADS];
    int row0[HC_MAX_HEADS];                 // first row of the head among the rows of its input
    int rows[HC_MAX_INPUTS];                // rows on each input
    const float* x[HC_MAX_INPUTS];          // (B, C, N)
    const float* w[HC_MAX_HEADS];           // (cout, C, k)
    const float* bias[HC_MAX_HEADS];        // (cout) or null
    float* y[HC_MAX_HEADS];                 // (B, cout, N)
    const float* dy[HC_MAX_HEADS];          // (B, cout, N) or null (no gradient arrives: zeros)
    float* dx[HC_MAX_INPUTS];               // (B, C, N) or null
    float* dw[HC_MAX_HEADS];                // (cout, C, k)
    float* db[HC_MAX_HEADS];                // (cout) or null
    HcLevels L;
};

__device__ __forceinline__ void level_bounds(const HcLevels& L, int n, int& lo, int& hi) {
    lo = L.lev[0]; hi = L.lev[1];
#pragma unroll
    for (int j = 1; j < OTAL_MAX_LEVELS; ++j)
        if (j < L.nlev && n >= L.lev[j]) { lo = L.lev[j]; hi = L.lev[j + 1]; }
}

// Global -> LDS staging with U loads of a lane in flight before its first LDS store.  These kernels are pure latency: a plain
// `lds[f(e)] = global[g(e)]` loop keeps one load in flight per lane and pays one memory round trip per element it owns
// (the first version of the weight-gradient kernel spent 30 of its 40 us that way).
template <int U, typename Ld, typename St>
__device__ __forceinline__ void stage_batched(int tid, int total, Ld ld, St st) {
    for (int e0 = tid; e0 < total; e0 += U * 256) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 256;
            v[u] = e < total ? ld(e) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 256;
            if (e < total) st(e, v[u]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
// grid (ceil(N / 8), B, n_heads); LDS: C x 12 floats
__global__ __launch_bounds__(256) void head_convs_fwd_kernel(const HcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    constexpr int PITCH = HC_P + 4;                     // 10 columns used (one halo column each side); 48-byte rows:
                                                        // 16-byte aligned and conflict-free for ds_read_b128
    const int tid = threadIdx.x, h = blockIdx.z, b = blockIdx.y, n0 = blockIdx.x * HC_P;
    const int C = a.C, N = a.N, k = a.k[h], cout = a.cout[h];
    const float* __restrict__ x = a.x[a.in_idx[h]] + (size_t)b * C * N;
    // stage x[b][:, n0 - 1 .. n0 + 8] in one memory round trip (20 loads in flight per lane at C = 512); columns outside
    // [0, N) read as zero; columns 10, 11 of a row are never read
    stage_batched<24>(tid, C * (HC_P + 2),
                     [&](int e) { const int ci = e / (HC_P + 2), n = n0 - 1 + e - ci * (HC_P + 2);
                                  return (n >= 0 && n < N) ? x[(size_t)ci * N + n] : 0.f; },
                     [&](int e, float v) { const int ci = e / (HC_P + 2); hsm[ci * PITCH + e - ci * (HC_P + 2)] = v; });
    // validity of the left / right tap of each position (a tap stays inside its level)
    unsigned lmask = 0, rmask = 0;
#pragma unroll
    for (int p = 0; p < HC_P; ++p) {
        const int n = n0 + p;
        int lo, hi;
        level_bounds(a.L, n < N ? n : N - 1, lo, hi);
        lmask |= (unsigned)(n < N && n > lo) << p;
        rmask |= (unsigned)(n < N && n + 1 < hi) << p;
    }
    __syncthreads();
    const int cl = tid & 15;
    for (int cb = 0; cb < cout; cb += 16) {
        const int co = cb + (tid >> 4);
        const bool live = co < cout;
        const float* __restrict__ w = a.w[h] + (size_t)(live ? co : 0) * C * k;
        float acc[HC_P];
#pragma unroll
        for (int p = 0; p < HC_P; ++p) acc[p] = 0.f;
        // the lane's weights travel through registers one chunk (8 channel steps = 128 channels) ahead of their use:
        // loaded inside the loop they cost one L2 round trip per channel step
        constexpr int CH = 8;
        float wc[CH][3], wn[CH][3];
        auto load_w = [&](float (&dst)[CH][3], int ci0) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int ci = ci0 + 16 * i;
                const bool ok = live && ci < C;
#pragma unroll
                for (int t = 0; t < 3; ++t) dst[i][t] = (ok && t < k) ? w[ci * k + t] : 0.f;
            }
        };
        load_w(wc, cl);
        for (int ci0 = cl; ci0 < C; ci0 += 16 * CH) {
            if (ci0 + 16 * CH < C) load_w(wn, ci0 + 16 * CH);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int ci = ci0 + 16 * i;
                if (ci >= C) break;
                const float4* row = reinterpret_cast<const float4*>(hsm + ci * PITCH);
                float f[HC_P + 4];
#pragma unroll
                for (int q = 0; q < (HC_P + 4) / 4; ++q) {
                    const float4 v = row[q];
                    f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
                }
                if (k == 3) {
#pragma unroll
                    for (int p = 0; p < HC_P; ++p) {
                        acc[p] = fmaf(wc[i][0], (lmask >> p) & 1u ? f[p] : 0.f, acc[p]);
                        acc[p] = fmaf(wc[i][1], f[p + 1], acc[p]);
                        acc[p] = fmaf(wc[i][2], (rmask >> p) & 1u ? f[p + 2] : 0.f, acc[p]);
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < HC_P; ++p) acc[p] = fmaf(wc[i][0], f[p + 1], acc[p]);
                }
            }
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) wc[i][t] = wn[i][t];
        }
        float mine = 0.f;
#pragma unroll
        for (int p = 0; p < HC_P; ++p) {
            float v = acc[p];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
            if (cl == p) mine = v;
        }
        const int n = n0 + cl;
        if (live && cl < HC_P && n < N) a.y[h][((size_t)b * cout + co) * N + n] = mine + (a.bias[h] ? a.bias[h][co] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ data gradient
// grid (C / 16, B, n_inputs); LDS: rows x (N + 2) gradients (zero halo) + rows x 3 x 16 weights
__global__ __launch_bounds__(256) void head_convs_dgrad_kernel(const HcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    const int tid = threadIdx.x, j = blockIdx.z, b = blockIdx.y, c0 = blockIdx.x * 16;
    const int C = a.C, N = a.N, rows = a.rows[j], NP = N + 2;
    if (!a.dx[j]) return;
    float* g = hsm;                         // [rows][N + 2]
    float* wt = hsm + rows * NP;            // [rows][3][16]
    for (int h = 0; h < a.n_heads; ++h) {
        if (a.in_idx[h] != j) continue;
        const int cout = a.cout[h], k = a.k[h], r0 = a.row0[h];
        const float* dy = a.dy[h] ? a.dy[h] + (size_t)b * cout * N : nullptr;
        const float* wh = a.w[h];
        for (int co = tid >> 5; co < cout; co += 8)              // thread (row mod 8, column mod 32): no runtime division
            for (int q0 = 0; q0 < NP; q0 += 256) {
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int q = q0 + (tid & 31) + 32 * t, n = q - 1;
                    v[t] = (dy && q < NP && n >= 0 && n < N) ? dy[(size_t)co * N + n] : 0.f;
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int q = q0 + (tid & 31) + 32 * t;
                    if (q < NP) g[(r0 + co) * NP + q] = v[t];
                }
            }
        // k = 1: the single tap sits in the centre slot
        stage_batched<4>(tid, cout * 48,
                         [&](int e) { const int co = e / 48, r = e - co * 48, tap = r >> 4, ci = r & 15;
                                      if (k == 3) return wh[((size_t)co * C + c0 + ci) * 3 + tap];
                                      return tap == 1 ? wh[(size_t)co * C + c0 + ci] : 0.f; },
                         [&](int e, float v) { const int co = e / 48; wt[(r0 + co) * 48 + e - co * 48] = v; });
    }
    __syncthreads();
    const int ci = tid >> 4, nl = tid & 15;
    float* dx = a.dx[j] + ((size_t)b * C + c0 + ci) * N;
    for (int n = nl; n < N; n += 16) {
        int lo, hi;
        level_bounds(a.L, n, lo, hi);
        const bool lv = n > lo, rv = n + 1 < hi;
        float s = 0.f;
        for (int r = 0; r < rows; ++r) {
            const float* gr = g + r * NP + n;           // gr[0] = dy[n - 1], gr[1] = dy[n], gr[2] = dy[n + 1]
            const float* wr = wt + r * 48 + ci;
            // y[n'] = sum_tap w[tap] x[n' + tap - 1]  =>  dx[n] = w[0] dy[n + 1] + w[1] dy[n] + w[2] dy[n - 1]
            s = fmaf(wr[0], rv ? gr[2] : 0.f, s);
            s = fmaf(wr[16], gr[1], s);
            s = fmaf(wr[32], lv ? gr[0] : 0.f, s);
        }
        dx[n] = s;
    }
}

// ------------------------------------------------------------------------------------------------ weight / bias gradient
// grid (C / 4, 1, n_inputs); 256 threads = 4 input channels x 64 (row, tap) slots (rows <= 21)
// LDS: 3 x 4 x B x N windows of x (left / centre / right tap, level mask applied) + B x N x (rows | 1) transposed gradients
__global__ __launch_bounds__(256) void head_convs_wgrad_kernel(const HcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    const int tid = threadIdx.x, j = blockIdx.z, c0 = blockIdx.x * 4;
    const int B = a.B, C = a.C, N = a.N, rows = a.rows[j], RP = rows | 1, BN = B * N;
    float* xs = hsm;                        // [3 taps][4 ci][B * N]
    float* gt = hsm + 12 * BN;              // [B * N][RP]
    const float* __restrict__ x = a.x[j];
    // Centre windows and transposed gradients from HBM / L2.  Thread (b mod 8, n mod 32): no integer division per element
    // (a flat index cost three runtime divisions ~ 120 VALU instructions per element: 10 us of a 40 us kernel), sixteen
    // loads in flight per lane.
    const int bl = tid >> 5, nl = tid & 31;
    for (int b = bl; b < B; b += 8)
        for (int n0 = 0; n0 < N; n0 += 128) {
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int n = n0 + nl + 32 * t;
                    v[u][t] = n < N ? x[((size_t)b * C + c0 + u) * N + n] : 0.f;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int n = n0 + nl + 32 * t;
                    if (n < N) xs[(4 + u) * BN + b * N + n] = v[u][t];
                }
        }
    for (int h = 0; h < a.n_heads; ++h) {
        if (a.in_idx[h] != j) continue;
        const int cout = a.cout[h], r0 = a.row0[h];
        const float* dy = a.dy[h];
        for (int b = bl; b < B; b += 8)
            for (int cb = 0; cb < cout; cb += 4)
                for (int n0 = 0; n0 < N; n0 += 128) {
                    float v[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int n = n0 + nl + 32 * t;
                            v[u][t] = (dy && cb + u < cout && n < N) ? dy[((size_t)b * cout + cb + u) * N + n] : 0.f;
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int n = n0 + nl + 32 * t;     // lanes along n: coalesced reads, odd-pitch (conflict-free) writes
                            if (cb + u < cout && n < N) gt[(b * N + n) * RP + r0 + cb + u] = v[u][t];
                        }
                }
    }
    __syncthreads();
    // ... then the left / right windows from the centre one, with the level mask applied (LDS -> LDS)
    for (int b = bl; b < B; b += 8)
        for (int n = nl; n < N; n += 32) {
            int lo, hi;
            level_bounds(a.L, n, lo, hi);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = u * BN + b * N + n;
                xs[e] = n > lo ? xs[4 * BN + e - 1] : 0.f;
                xs[8 * BN + e] = n + 1 < hi ? xs[4 * BN + e + 1] : 0.f;
            }
        }
    __syncthreads();
    const int ci = tid >> 6, slot = tid & 63, r = slot / 3, tap = slot - r * 3;
    if (r >= rows) return;
    int h = 0;                              // the head that owns row r
    for (int q = 0; q < a.n_heads; ++q)
        if (a.in_idx[q] == j && r >= a.row0[q] && r < a.row0[q] + a.cout[q]) h = q;
    const int k = a.k[h], co = r - a.row0[h];
    // k = 3: tap t multiplies x[n + t - 1]; k = 1: only the centre window (handled by the lane with tap == 1)
    const bool wlane = k == 3 || tap == 1;
    const float* xw = xs + (tap * 4 + ci) * BN;
    const float* gr = gt + r;
    // sixteen positions per trip, four interleaved partial sums (combined in a fixed order at the end): the loop is LDS
    // latency, and one dependent fma chain with 5 reads per 4 positions left the four waves of the workgroup waiting
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int q = 0;
    if ((BN & 3) == 0) {                    // window rows are 16-byte aligned: four positions per LDS read of x
        for (; q + 16 <= BN; q += 16) {
            float4 xv[4];
            float gv[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const float4*>(xw + q + 4 * u);
#pragma unroll
            for (int u = 0; u < 16; ++u) gv[u] = gr[(q + u) * RP];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                p0 = fmaf(gv[4 * u], xv[u].x, p0); p1 = fmaf(gv[4 * u + 1], xv[u].y, p1);
                p2 = fmaf(gv[4 * u + 2], xv[u].z, p2); p3 = fmaf(gv[4 * u + 3], xv[u].w, p3);
                s0 += gv[4 * u]; s1 += gv[4 * u + 1]; s2 += gv[4 * u + 2]; s3 += gv[4 * u + 3];
            }
        }
    }
    for (; q < BN; ++q) {
        const float gq = gr[q * RP];
        p0 = fmaf(gq, xw[q], p0);
        s0 += gq;
    }
    const float acc = (p0 + p1) + (p2 + p3), sb = (s0 + s1) + (s2 + s3);
    if (wlane) a.dw[h][((size_t)co * C + c0 + ci) * k + (k == 3 ? tap : 0)] = acc;
    if (blockIdx.x == 0 && ci == 0 && tap == 1 && a.db[h]) a.db[h][co] = sb;
}

constexpr size_t HC_LDS_MAX = 160 * 1024;

int fill(HcArgs& a, int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize, int B, int C, int N, int nlev,
         const int* lev) {
    if (n_heads < 1 || n_heads > HC_MAX_HEADS || n_inputs < 1 || n_inputs > HC_MAX_INPUTS) return OTAL_E_SHAPE;
    if (B <= 0 || C <= 0 || N <= 0 || C % 16) return OTAL_E_SHAPE;
    if (!in_idx || !cout || !ksize) return OTAL_E_NULL;
    a = HcArgs{};
    a.n_heads = n_heads; a.n_inputs = n_inputs; a.B = B; a.C = C; a.N = N;
    for (int h = 0; h < n_heads; ++h) {
        if (in_idx[h] < 0 || in_idx[h] >= n_inputs || cout[h] <= 0 || (ksize[h] != 1 && ksize[h] != 3)) return OTAL_E_SHAPE;
        a.in_idx[h] = in_idx[h]; a.cout[h] = cout[h]; a.k[h] = ksize[h];
        a.row0[h] = a.rows[in_idx[h]];
        a.rows[in_idx[h]] += cout[h];
    }
    for (int j = 0; j < n_inputs; ++j)
        if (a.rows[j] > HC_MAX_ROWS) return OTAL_E_UNSUPPORTED;
    if (nlev <= 1 || !lev) {
        a.L.nlev = 1; a.L.lev[0] = 0;
        for (int i = 1; i <= OTAL_MAX_LEVELS; ++i) a.L.lev[i] = N;
    } else {
        if (nlev > OTAL_MAX_LEVELS || lev[0] != 0 || lev[nlev] != N) return OTAL_E_LEVELS;
        a.L.nlev = nlev;
        for (int i = 0; i <= OTAL_MAX_LEVELS; ++i) a.L.lev[i] = lev[i <= nlev ? i : nlev];
        for (int i = 0; i < nlev; ++i) if (a.L.lev[i + 1] <= a.L.lev[i]) return OTAL_E_LEVELS;
    }
    return 0;
}

size_t wgrad_lds(const HcArgs& a) {
    size_t m = 0;
    for (int j = 0; j < a.n_inputs; ++j) {
        const size_t v = ((size_t)12 * a.B * a.N + (size_t)a.B * a.N * (a.rows[j] | 1)) * 4;
        m = v > m ? v : m;
    }
    return m;
}

template <typename K> int allow_lds(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HC_LDS_MAX) != hipSuccess) {
        (void)hipGetLastError();
        return OTAL_E_UNSUPPORTED;
    }
    return 0;
}

}  // namespace

extern "C" int otal_head_convs_supported(int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize, int B, int C,
                                         int N) {
    HcArgs a;
    if (fill(a, n_heads, n_inputs, in_idx, cout, ksize, B, C, N, 1, nullptr)) return 0;
    int maxrows = 0;
    for (int j = 0; j < n_inputs; ++j) maxrows = a.rows[j] > maxrows ? a.rows[j] : maxrows;
    const size_t fwd = (size_t)C * (HC_P + 4) * 4, dg = ((size_t)maxrows * (N + 2) + (size_t)maxrows * 48) * 4;
    return fwd <= HC_LDS_MAX && dg <= HC_LDS_MAX && wgrad_lds(a) <= HC_LDS_MAX;
}

extern "C" int otal_head_convs_fwd(int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize,
                                   const float* const* x, const float* const* w, const float* const* bias, float* const* y, int B,
                                   int C, int N, int nlev, const int* lev, void* stream) {
    if (!x || !w || !bias || !y) return OTAL_E_NULL;
    HcArgs a;
    if (int e = fill(a, n_heads, n_inputs, in_idx, cout, ksize, B, C, N, nlev, lev)) return e;
    for (int j = 0; j < n_inputs; ++j) { if (!x[j]) return OTAL_E_NULL; a.x[j] = x[j]; }
    for (int h = 0; h < n_heads; ++h) {
        if (!w[h] || !y[h]) return OTAL_E_NULL;
        a.w[h] = w[h]; a.bias[h] = bias[h]; a.y[h] = y[h];
    }
    const size_t lds = (size_t)C * (HC_P + 4) * 4;
    if (lds > HC_LDS_MAX) return OTAL_E_UNSUPPORTED;
    if (int e = allow_lds(head_convs_fwd_kernel, lds)) return e;
    hipLaunchKernelGGL(head_convs_fwd_kernel, dim3((N + HC_P - 1) / HC_P, B, n_heads), dim3(256), lds, (hipStream_t)stream, a);
    return otal_launch_status();
}

// parts: bit 0 = the data gradients (dx), bit 1 = the weight / bias gradients (dw, db).  The two launches are independent
// (both read x, w, dy only), so a caller may issue them on different streams -- the data gradient on the chain that needs
// it, the weight gradients on its weight-gradient lane.
extern "C" int otal_head_convs_bwd_parts(int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize,
                                         const float* const* x, const float* const* w, const float* const* dy, float* const* dx,
                                         float* const* dw, float* const* db, int B, int C, int N, int nlev, const int* lev,
                                         int parts, void* stream);
extern "C" int otal_head_convs_bwd(int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize,
                                   const float* const* x, const float* const* w, const float* const* dy, float* const* dx,
                                   float* const* dw, float* const* db, int B, int C, int N, int nlev, const int* lev, void* stream) {
    return otal_head_convs_bwd_parts(n_heads, n_inputs, in_idx, cout, ksize, x, w, dy, dx, dw, db, B, C, N, nlev, lev, 3, stream);
}
extern "C" int otal_head_convs_bwd_parts(int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize,
                                         const float* const* x, const float* const* w, const float* const* dy, float* const* dx,
                                         float* const* dw, float* const* db, int B, int C, int N, int nlev, const int* lev,
                                         int parts, void* stream) {
    if (!x || !w || !dy || !dx || !dw || !db) return OTAL_E_NULL;
    if (parts < 1 || parts > 3) return OTAL_E_SHAPE;
    HcArgs a;
    if (int e = fill(a, n_heads, n_inputs, in_idx, cout, ksize, B, C, N, nlev, lev)) return e;
    bool any_dx = false;
    int maxrows = 0;
    for (int j = 0; j < n_inputs; ++j) {
        if (!x[j]) return OTAL_E_NULL;
        a.x[j] = x[j]; a.dx[j] = dx[j];
        any_dx = any_dx || dx[j];
        maxrows = a.rows[j] > maxrows ? a.rows[j] : maxrows;
    }
    for (int h = 0; h < n_heads; ++h) {
        if (!w[h] || !dw[h]) return OTAL_E_NULL;
        a.w[h] = w[h]; a.dy[h] = dy[h]; a.dw[h] = dw[h]; a.db[h] = db[h];
    }
    if (any_dx && (parts & 1)) {
        const size_t lds = ((size_t)maxrows * (N + 2) + (size_t)maxrows * 48) * 4;
        if (lds > HC_LDS_MAX) return OTAL_E_UNSUPPORTED;
        if (int e = allow_lds(head_convs_dgrad_kernel, lds)) return e;
        hipLaunchKernelGGL(head_convs_dgrad_kernel, dim3(C / 16, B, n_inputs), dim3(256), lds, (hipStream_t)stream, a);
        if (int e = otal_launch_status()) return e;
    }
    if (!(parts & 2)) return 0;
    const size_t lds = wgrad_lds(a);
    if (lds > HC_LDS_MAX) return OTAL_E_UNSUPPORTED;
    if (int e = allow_lds(head_convs_wgrad_kernel, lds)) return e;
    hipLaunchKernelGGL(head_convs_wgrad_kernel, dim3(C / 4, 1, n_inputs), dim3(256), lds, (hipStream_t)stream, a);
    return otal_launch_status();
}
