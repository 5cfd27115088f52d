// opental_amd/csrc/infer.hip -- inference post-processing of the OpenTAL detector on gfx950:
//   otal_decode_clips   : parse_output + decode_predictions + the per-class threshold test of
//                         `filtering` (AFSD/thumos14/test.py:79-162) for a batch of clips, one launch;
//   otal_softnms_classes: per (video, class) gather of the surviving candidates in the reference's
//                         order (clip-major, anchor-minor) followed by Gaussian Soft-NMS
//                         (softnms_v2, AFSD/common/segment_utils.py:128-162) -- one workgroup per
//                         (video, class), candidates resident in LDS for the whole greedy loop.
// The reference does all of this on the host: 15 boolean-mask kernels + syncs per clip, a .cpu()
// copy and a Python while-loop per class.  Here a whole batch of videos is two launches.
//
// Soft-NMS semantics kept bit for bit on the kept-index set: candidates with
// score >= thr are "undone"; each round takes the FIRST maximum among undone, marks it done, decays
// every other undone score by exp(-iou^2 / sigma) and drops those < thr; the loop stops when <= 1
// candidate is undone (so the last survivor is never kept) or top_k are done; rows come out in
// original index order with the decayed scores.  Latency-bound (min(top_k, N) dependent rounds), so
// the figure of merit is candidates/s, not bytes.
#include "common.h"

namespace {

constexpr int NMS_THREADS = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// one workgroup per clip; A anchors, K classes.  Outputs: seg (clip,A,2) seconds; score (clip,K,A);
// unct (clip,A); actn (clip,A); flag (clip,K,A) uint8.
__global__ __launch_bounds__(128) void decode_clips_kernel(
        const float* __restrict__ loc, const float* __restrict__ prop_loc, const float* __restrict__ priors,
        const float* __restrict__ conf, const float* __restrict__ prop_conf, const float* __restrict__ center,
        const float* __restrict__ act, const float* __restrict__ prop_act, const float* __restrict__ offsets,
        const float* __restrict__ fps, float* __restrict__ seg, float* __restrict__ score, float* __restrict__ unct,
        float* __restrict__ actn, unsigned char* __restrict__ flag, int A, int K, float clip_length,
        float conf_thresh) {
    const int c = blockIdx.x;
    for (int i = threadIdx.x; i < A; i += blockDim.x) {
        const size_t ai = (size_t)c * A + i;
        // decode_predictions (test.py:114-120)
        const float l0 = loc[ai * 2], l1 = loc[ai * 2 + 1];
        const float w = l0 + l1;
        const float r0 = 0.5f * w * prop_loc[ai * 2] + l0;
        const float r1 = 0.5f * w * prop_loc[ai * 2 + 1] + l1;
        const float pc = priors[i] * clip_length;
        float s0 = fminf(fmaxf(pc - r0, 0.f), clip_length);
        float s1 = fminf(fmaxf(pc + r1, 0.f), clip_length);
        seg[ai * 2] = (s0 + offsets[c]) / fps[c];
        seg[ai * 2 + 1] = (s1 + offsets[c]) / fps[c];
        // Dirichlet mean + uncertainty (BDNet.py:538-561), evidence = exp(clamp(logit, +-10))
        const float* cf = conf + ai * K;
        const float* pf = prop_conf + ai * K;
        float S = 0.f, PS = 0.f;
        for (int k = 0; k < K; ++k) {
            S += expf(fminf(fmaxf(cf[k], -10.f), 10.f)) + 1.0f;
            PS += expf(fminf(fmaxf(pf[k], -10.f), 10.f)) + 1.0f;
        }
        const float u = ((float)K / S + (float)K / PS) / 2.0f;
        const float an = (sigmoidf_(act[ai]) + sigmoidf_(prop_act[ai])) / 2.0f;
        const float ct = sigmoidf_(center[ai]);
        unct[ai] = u;
        actn[ai] = an;
        for (int k = 0; k < K; ++k) {
            const float a0 = (expf(fminf(fmaxf(cf[k], -10.f), 10.f)) + 1.0f) / S;
            const float a1 = (expf(fminf(fmaxf(pf[k], -10.f), 10.f)) + 1.0f) / PS;
            const float sc = (a0 + a1) / 2.0f * ct * an;
            const size_t o = ((size_t)c * K + k) * A + i;
            score[o] = sc;
            flag[o] = (sc > conf_thresh) && (an > 0.5f);       // filtering (test.py:143-147)
        }
    }
}

struct Best { float v; int i; int n; };   // max undone score, its lowest index, number of undone

__device__ __forceinline__ Best best_merge(Best a, Best b) {
    Best r;
    r.n = a.n + b.n;
    const bool take_b = (b.i >= 0) && (a.i < 0 || b.v > a.v || (b.v == a.v && b.i < a.i));
    r.v = take_b ? b.v : a.v;
    r.i = take_b ? b.i : a.i;
    return r;
}

// One workgroup per (video, class).  Working set: ts, te, sc (float), st (int: 0 dropped, 1 done, 2 undone),
// src (int: row in the clip-major candidate space) for up to `cap` candidates.
// BIG = false: the working set lives in LDS (`cap` = lds_cap rows); a video with more rows than that is left to the
//              BIG launch when one follows (`defer_big`), i.e. this workgroup returns at once.
// BIG = true : the working set lives in a caller-provided global scratch sized for every row of the video (five
//              arrays of total_clips*K*A words; (v,k) owns rows [(clip_start[v]*K + k*clips_v)*A, +clips_v*A)), so a
//              video of any length is handled (the reference's host loop has no limit either, test.py:165-200); only
//              the videos the LDS launch deferred are processed.  All traffic stays inside one workgroup, whose
//              barriers order its own global stores and loads.
template <bool BIG>
__global__ __launch_bounds__(NMS_THREADS) void softnms_classes_kernel(
        const float* __restrict__ seg, const float* __restrict__ score, const float* __restrict__ unct,
        const float* __restrict__ actn, const unsigned char* __restrict__ flag, const int* __restrict__ clip_start,
        float* __restrict__ out, int* __restrict__ counts, int* __restrict__ out_index, int A, int K, int lds_cap,
        float sigma, int top_k, float thr, int out_cols, float* __restrict__ scratch, long long scratch_rows, int defer_big) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ Best red[NMS_THREADS / 64];
    __shared__ int s_n, s_scan[NMS_THREADS / 64 + 1];
    const int v = blockIdx.x / K, k = blockIdx.x % K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = clip_start[v], c1 = clip_start[v + 1];
    const int total = (c1 - c0) * A;          // rows of this video in clip-major order
    if (BIG ? total <= lds_cap : (defer_big && total > lds_cap)) return;
    const int cap = BIG ? total : lds_cap;
    float* ts; float* te; float* sc; int* st; int* src;
    if constexpr (BIG) {
        const long long o = ((long long)c0 * K + (long long)k * (c1 - c0)) * A;
        ts = scratch + o;
        te = scratch + scratch_rows + o;
        sc = scratch + 2 * scratch_rows + o;
        st = reinterpret_cast<int*>(scratch + 3 * scratch_rows + o);
        src = reinterpret_cast<int*>(scratch + 4 * scratch_rows + o);
    } else {
        ts = reinterpret_cast<float*>(smem);
        te = ts + cap;
        sc = te + cap;
        st = reinterpret_cast<int*>(sc + cap);
        src = st + cap;
    }
    // ---- gather flagged candidates in index order (block scan over chunks of NMS_THREADS rows)
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int base = 0; base < total; base += NMS_THREADS) {
        const int r = base + tid;
        int f = 0;
        size_t so = 0;
        if (r < total) {
            const int c = c0 + r / A, i = r % A;
            so = ((size_t)c * K + k) * A + i;
            f = flag[so];
        }
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_scan[wave + 1] = __popcll(m);
        __syncthreads();
        if (tid == 0) {
            s_scan[0] = s_n;
            for (int w = 0; w < NMS_THREADS / 64; ++w) s_scan[w + 1] += s_scan[w];
            s_n = s_scan[NMS_THREADS / 64];
        }
        __syncthreads();
        if (f) {
            const int p = s_scan[wave] + before;
            if (p < cap) {
                const int c = c0 + r / A, i = r % A;
                ts[p] = seg[((size_t)c * A + i) * 2];
                te[p] = seg[((size_t)c * A + i) * 2 + 1];
                sc[p] = score[so];
                st[p] = score[so] >= thr ? 2 : 0;
                src[p] = c * A + i;
            }
        }
        __syncthreads();
    }
    const int n = min(s_n, cap);
    // ---- greedy loop
    int ndone = 0;
    for (;;) {
        Best b = {0.f, -1, 0};
        for (int i = tid; i < n; i += NMS_THREADS)
            if (st[i] == 2) {
                b.n += 1;
                if (b.i < 0 || sc[i] > b.v) { b.v = sc[i]; b.i = i; }     // strided ascending i: first max kept
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            Best t;
            t.v = __shfl_down(b.v, o, 64); t.i = __shfl_down(b.i, o, 64); t.n = __shfl_down(b.n, o, 64);
            b = best_merge(b, t);
        }
        if (lane == 0) red[wave] = b;
        __syncthreads();
        b = red[0];
#pragma unroll
        for (int w = 1; w < NMS_THREADS / 64; ++w) b = best_merge(b, red[w]);
        __syncthreads();
        if (b.n <= 1 || ndone >= top_k) break;                 // `undone.sum() > 1 and done.sum() < top_k`
        const int j = b.i;
        const float top_s = ts[j], top_e = te[j];
        const float width = fmaxf(top_e - top_s, 1e-5f);
        if (tid == 0) st[j] = 1;
        ++ndone;
        __syncthreads();
        for (int i = tid; i < n; i += NMS_THREADS)
            if (st[i] == 2) {
                const float a0 = ts[i], a1 = te[i];
                const float inter = fmaxf(fminf(a1, top_e) - fmaxf(a0, top_s), 0.f);
                const float iou = inter / (width + (a1 - a0) - inter);
                const float s = sc[i] * expf(-(iou * iou) / sigma);
                sc[i] = s;
                if (s < thr) st[i] = 0;
            }
        __syncthreads();
    }
    // ---- kept rows, original index order: [start, end, decayed score, unct, actionness]
    if (tid == 0) s_n = 0;
    __syncthreads();
    float* o = out + (size_t)blockIdx.x * top_k * out_cols;
    for (int base = 0; base < n; base += NMS_THREADS) {
        const int i = base + tid;
        const int f = (i < n) && st[i] == 1;
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_scan[wave + 1] = __popcll(m);
        __syncthreads();
        if (tid == 0) {
            s_scan[0] = s_n;
            for (int w = 0; w < NMS_THREADS / 64; ++w) s_scan[w + 1] += s_scan[w];
            s_n = s_scan[NMS_THREADS / 64];
        }
        __syncthreads();
        if (f) {
            const int p = s_scan[wave] + before;
            float* row = o + (size_t)p * out_cols;
            row[0] = ts[i]; row[1] = te[i]; row[2] = sc[i];
            if (out_cols > 3) row[3] = unct[src[i]];
            if (out_cols > 4) row[4] = actn[src[i]];
            if (out_index) out_index[(size_t)blockIdx.x * top_k + p] = src[i] - c0 * A;
        }
        __syncthreads();
    }
    if (tid == 0) counts[blockIdx.x] = s_n;
}

}  // namespace

extern "C" int otal_decode_clips(const float* loc, const float* prop_loc, const float* priors, const float* conf,
                                 const float* prop_conf, const float* center, const float* act,
                                 const float* prop_act, const float* offsets, const float* fps, float* seg,
                                 float* score, float* unct, float* actn, unsigned char* flag, int nclips, int A,
                                 int K, float clip_length, float conf_thresh, void* stream) {
    if (!loc || !prop_loc || !priors || !conf || !prop_conf || !center || !act || !prop_act || !offsets || !fps ||
        !seg || !score || !unct || !actn || !flag) return OTAL_E_NULL;
    if (nclips <= 0 || A <= 0 || K <= 0) return OTAL_E_SHAPE;
    hipLaunchKernelGGL(decode_clips_kernel, dim3(nclips), dim3(128), 0, (hipStream_t)stream, loc, prop_loc, priors,
                       conf, prop_conf, center, act, prop_act, offsets, fps, seg, score, unct, actn, flag, A, K,
                       clip_length, conf_thresh);
    return otal_launch_status();
}

constexpr size_t NMS_LDS_LIMIT = 150 * 1024;      // working set of ~7600 candidates; beyond that the scratch path runs

extern "C" size_t otal_softnms_scratch_bytes(int total_clips, int max_clips, int A, int K) {
    if (total_clips <= 0 || max_clips <= 0 || A <= 0 || K <= 0) return 0;
    if ((size_t)max_clips * A * 20 <= NMS_LDS_LIMIT) return 0;
    return (size_t)total_clips * K * A * 20;
}

extern "C" int otal_softnms_classes_ws(const float* seg, const float* score, const float* unct, const float* actn,
                                       const unsigned char* flag, const int* clip_start, int nvideos, int max_clips,
                                       int A, int K, float sigma, int top_k, float score_threshold, float* out,
                                       int* counts, int* out_index, int out_cols, void* scratch, size_t scratch_bytes,
                                       int total_clips, void* stream) {
    if (!seg || !score || !unct || !actn || !flag || !clip_start || !out || !counts) return OTAL_E_NULL;
    if (nvideos <= 0 || max_clips <= 0 || A <= 0 || K <= 0 || top_k <= 0 || out_cols < 3 || out_cols > 5)
        return OTAL_E_SHAPE;
    const size_t need = otal_softnms_scratch_bytes(total_clips, max_clips, A, K);
    const bool big = need != 0;
    if (big && (!scratch || scratch_bytes < need)) return OTAL_E_UNSUPPORTED;   // longest video exceeds LDS: scratch required
    const int lds_cap = big ? (int)(NMS_LDS_LIMIT / 20) : max_clips * A;
    const size_t lds = (size_t)lds_cap * 20;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(softnms_classes_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const long long rows = (long long)total_clips * K * A;
    hipLaunchKernelGGL(softnms_classes_kernel<false>, dim3(nvideos * K), dim3(NMS_THREADS), lds, (hipStream_t)stream, seg,
                       score, unct, actn, flag, clip_start, out, counts, out_index, A, K, lds_cap, sigma, top_k, score_threshold,
                       out_cols, (float*)nullptr, 0LL, big ? 1 : 0);
    if (big)
        hipLaunchKernelGGL(softnms_classes_kernel<true>, dim3(nvideos * K), dim3(NMS_THREADS), 0, (hipStream_t)stream, seg,
                           score, unct, actn, flag, clip_start, out, counts, out_index, A, K, lds_cap, sigma, top_k,
                           score_threshold, out_cols, (float*)scratch, rows, 1);
    return otal_launch_status();
}

extern "C" int otal_softnms_classes(const float* seg, const float* score, const float* unct, const float* actn,
                                    const unsigned char* flag, const int* clip_start, int nvideos, int max_clips,
                                    int A, int K, float sigma, int top_k, float score_threshold, float* out,
                                    int* counts, int* out_index, int out_cols, void* stream) {
    // no scratch: every video must fit the LDS working set (OTAL_E_UNSUPPORTED otherwise; use otal_softnms_classes_ws)
    return otal_softnms_classes_ws(seg, score, unct, actn, flag, clip_start, nvideos, max_clips, A, K, sigma, top_k,
                                   score_threshold, out, counts, out_index, out_cols, nullptr, 0,
                                   (size_t)max_clips * A * 20 <= NMS_LDS_LIMIT ? 1 : max_clips, stream);
}
