// opental_amd/csrc/loss.hip -- MultiSegmentLoss + EvidenceLoss + ActionnessLoss of the OpenTAL final recipe,
// forward AND backward, as ONE single-workgroup launch.
//
// Replaces (reference AFSD/thumos14): multisegment_loss.py:92-259 (anchor<->GT matching, GIoU / L1 / quality-BCE
// terms, normalisers), cls_loss.py:132-168,212-278 (EvidenceLoss 'log' with exp evidence and the influence-balanced
// 50-bin EMA re-weighting), cls_loss.py:120-129 (IoU calibration), cls_loss.py:288-339 (positive-unlabelled
// actionness BCE) and the autograd graph behind them -- ~400 ATen launches and ~60 host syncs per step in the
// reference, ~400 tiny launches in this package's own torch formulation.
//
// The problem is tiny (B*126 anchors, <= a few thousand) and full of global dependencies (positive counts
// normalise every term, the IBM histogram must be complete before any weight is read, the PU loss ranks all
// negatives), so it runs in ONE workgroup of 1024 threads: phases separated by __syncthreads(), per-anchor
// intermediates in a global scratch area, reductions in a fixed order (deterministic, unlike index_add_).
// Latency-bound by construction (SURVEY 8d puts the losses under "launch latency"); the win is 400 launches -> 1.
//
// Supported configurations: the final recipe -- evidence 'exp', loss_type 'log', os_head, size_average False, actionness
// rank-term weight 0 -- and, with cls_mode = 1, the as-shipped THUMOS14 dispatch (train.py:27-31 overwrites 'edl' with
// 'focal'): FocalLoss_Ori(balance_index 0, alpha 0.25, gamma 2) on softmax scores (cls_loss.py:6-78) instead
// of the evidential terms (no IBM, no IoU calibration).  Everything else stays on the torch formulation.
#include "common.h"

namespace {

constexpr int LT = 1024;            // threads of the single workgroup
constexpr int MAX_BINS = 64;
constexpr int MAX_A = 2048;         // anchors (B * K) the single-workgroup kernel accepts (LDS-resident scan / sort arrays)
constexpr int HSEG = LT / MAX_BINS; // histogram: anchors are cut in HSEG segments, thread = (segment, bin)
constexpr float F_EPS = 1.1920928955078125e-07f;     // torch.finfo(float32).eps

struct LossArgs {
    const float *loc, *conf, *prop_loc, *prop_conf, *center, *act, *prop_act, *priors, *gt;
    const unsigned char* gvalid;
    float* weight_accum;            // in/out, num_bins
    float* losses;                  // 7: l, c, prop_l, prop_c, ct, act, prop_act
    // gradients: dloc_l(A,2) dloc_ct(A,2) dprop_loc_pl(A,2) dprop_loc_ct(A,2) dconf(A,C) dprop_conf(A,C) dcenter(A) dact(A) dprop_act(A)
    float* grads;
    float* scratch;                 // per-anchor intermediates, SCR floats per anchor
    int B, K, C, G;
    float clip, overlap;
    int ibm_active, num_bins, iou_aware;
    float momentum;
    int cls_mode;                   // 0: EvidenceLoss (the OpenTAL recipe); 1: FocalLoss_Ori on softmax scores (as-shipped THUMOS14 dispatch)
    float focal_alpha;              // FocalLoss_Ori(balance_index=0, alpha): alpha for class 0, 1 - alpha for the others
};
constexpr int SCR = 12;             // loc_t0, loc_t1, conf_t, prop_conf_t, iou, prop_loc_t0, prop_loc_t1, ghat, slot, binpos, per, used

// ---- deterministic block reductions ---------------------------------------------------------------
// The sum is the pairwise halving tree over the thread index (r[t] += r[t + s], s = 512 .. 1) -- the order the first version
// of this kernel walked with one __syncthreads() per level.  This single-workgroup kernel is a chain of ~250 barriers of
// sixteen waves (a third of a microsecond each: most of its 111 us), so the tree is now evaluated with THREE: the levels that
// pair different waves (s >= 64) are summed by wave 0 from LDS in exactly that order, the levels inside a wave by DPP-free
// shuffles (lane t adds lane t + s).  Bit-identical to the barrier-per-level form.
__device__ float block_sum(float v, float* red) {
    static_assert(LT == 1024, "the cross-wave tree below is written for sixteen waves");
    const int t = threadIdx.x;
    __syncthreads();
    red[t] = v;
    __syncthreads();
    if (t < 64) {
        float y[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) y[w] = red[w * 64 + t] + red[(w + 8) * 64 + t];         // s = 512
        const float z0 = y[0] + y[4], z1 = y[1] + y[5], z2 = y[2] + y[6], z3 = y[3] + y[7];  // s = 256
        float r = (z0 + z2) + (z1 + z3);                                                     // s = 128, 64
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) r += __shfl_down(r, s, 64);                         // s = 32 .. 1 (lane t < s keeps the tree's value)
        if (t == 0) red[0] = r;
    }
    __syncthreads();
    const float r = red[0];
    __syncthreads();
    return r;
}

// torch.minimum / maximum backward: a tie sends half of the gradient to each side
__device__ __forceinline__ float dmin_da(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float dmax_da(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }

struct TIoU { float iou, inter, uni, d0, d1; };   // d*: d iou / d pred_*
__device__ TIoU tiou_grad(float p0, float p1, float t0, float t1) {
    TIoU r;
    r.inter = fminf(p0, t0) + fminf(p1, t1);
    r.uni = (t0 + t1) + (p0 + p1) - r.inter;
    const float uc = fmaxf(r.uni, F_EPS);
    r.iou = r.inter / uc;
    const float di0 = dmin_da(p0, t0), di1 = dmin_da(p1, t1);
    const float du0 = 1.f - di0, du1 = 1.f - di1;
    const float live = r.uni >= F_EPS ? 1.f : 0.f;       // clamp(min=eps) backward
    r.d0 = di0 / uc - r.inter * du0 * live / (uc * uc);
    r.d1 = di1 / uc - r.inter * du1 * live / (uc * uc);
    return r;
}

// EvidenceLoss statistics of one anchor row (cls_loss.py:132-160): per = log S - log alpha_y, IBM slot
struct Edl { float per, S, ay, ghat; int slot, binpos; };
__device__ Edl edl_row(const float* z, int C, int y, int num_bins) {
    Edl e;
    float S = 0.f, l1 = 0.f, ay = 1.f;
    for (int k = 0; k < C; ++k) {
        const float a = expf(fminf(fmaxf(z[k], -10.f), 10.f)) + 1.f;
        S += a;
        l1 += fabsf(z[k]);
        if (k == y) ay = a;
    }
    e.S = S; e.ay = ay;
    e.per = logf(S) - logf(ay);
    const float u = (float)C / S;
    const float gnorm = fabsf(1.f / ay - u);
    e.ghat = gnorm * l1;
    const long bins = (long)ceilf(gnorm * (float)num_bins);
    e.binpos = bins > 0 ? 1 : 0;
    long sl = (bins - 1) % num_bins;
    if (sl < 0) sl += num_bins;                           // python-style remainder: bin 0 reads slot -1
    e.slot = (int)sl;
    return e;
}

// ST (EvidenceLoss mode, A * C floats fit the dynamic LDS): the logits of a pass are copied to LDS with coalesced 16-byte loads,
// every class loop reads them there, the gradient rows are written over them in place (with the IoU-calibration term of the
// prop_conf pass folded in: same operands, same order of additions as the separate loop below) and leave with coalesced
// stores.  One anchor per thread reads its C = 21 logits at an 84-byte stride: every one of the ~230 load / store
// instructions per wave touched 64 different cache lines -- on ONE compute unit that was ~80 of the kernel's 107 us.
template <bool ST>
__global__ __launch_bounds__(LT) void detection_loss_kernel(const LossArgs a) {
    extern __shared__ __attribute__((aligned(16))) float zl[];       // ST: A * C floats (logits, then gradients, of the current pass)
    __shared__ float red[LT];
    __shared__ float wacc[MAX_BINS];
    __shared__ int icnt[4];
    // per-anchor arrays every thread scans (rank counting, histogram): LDS-resident, A <= MAX_A
    __shared__ float s_val[MAX_A];              // ghat (EDL passes) / actionness score (PU passes)
    __shared__ unsigned char s_slot[MAX_A];     // IBM slot
    __shared__ unsigned char s_flag[MAX_A];     // bit0: row counts for the histogram / is a negative
    __shared__ float h_tot[HSEG][MAX_BINS], h_cnt[HSEG][MAX_BINS];
    __shared__ unsigned long long skey[MAX_A];  // PU loss: (order-preserving score bits << 32 | anchor index), sorted ascending
    const int t = threadIdx.x;
    const int A = a.B * a.K, C = a.C;
    float* g_loc_l = a.grads;
    float* g_loc_ct = g_loc_l + 2 * A;
    float* g_pl_pl = g_loc_ct + 2 * A;
    float* g_pl_ct = g_pl_pl + 2 * A;
    float* g_conf = g_pl_ct + 2 * A;
    float* g_pconf = g_conf + (size_t)A * C;
    float* g_center = g_pconf + (size_t)A * C;
    float* g_act = g_center + A;
    float* g_pact = g_act + A;

    if (t < 4) icnt[t] = 0;
    if (t < a.num_bins) wacc[t] = a.weight_accum[t];
    __syncthreads();

    // ---- phase 1: matching (multisegment_loss.py:120-153), no gradient
    int npos = 0, nppos = 0;
    for (int i = t; i < A; i += LT) {
        const int b = i / a.K, k = i - b * a.K;
        const float c = a.priors[k];
        const float big = a.clip * 2.f;
        float best_area = 0.f;
        int best = 0;
        for (int g = 0; g < a.G; ++g) {
            const float left = (c - a.gt[(b * a.G + g) * 3]) * a.clip, right = (a.gt[(b * a.G + g) * 3 + 1] - c) * a.clip;
            float area = left + right;
            if (left < 0.f || right < 0.f || !a.gvalid[b * a.G + g]) area = big;
            if (g == 0 || area < best_area) { best_area = area; best = g; }     // first minimum, like torch.min
        }
        const float g0 = a.gt[(b * a.G + best) * 3], g1 = a.gt[(b * a.G + best) * 3 + 1], lab = a.gt[(b * a.G + best) * 3 + 2];
        const float lt0 = (c - g0) * a.clip, lt1 = (g1 - c) * a.clip;
        const int conf_t = best_area >= big ? 0 : (int)lab;
        const float p0 = a.loc[2 * i], p1 = a.loc[2 * i + 1];
        const TIoU q = tiou_grad(p0, p1, lt0, lt1);
        const int pconf_t = q.iou < a.overlap ? 0 : conf_t;
        const float w = p0 + p1;
        float* s = a.scratch + (size_t)i * SCR;
        s[0] = lt0; s[1] = lt1; s[2] = (float)conf_t; s[3] = (float)pconf_t; s[4] = q.iou;
        s[5] = (lt0 - p0) / (0.5f * w); s[6] = (lt1 - p1) / (0.5f * w);
        npos += conf_t > 0; nppos += pconf_t > 0;
    }
    atomicAdd(&icnt[0], npos);
    atomicAdd(&icnt[1], nppos);
    __syncthreads();
    const float Nf = (float)max(icnt[0], 1), PNf = (float)max(icnt[1], 1);

    // ---- phases 2-4 (conf) and 5-7 (prop_conf): EvidenceLoss with IBM re-weighting (cls_loss.py:132-168)
    float loss_cls[2];
    for (int pass = 0; pass < 2; ++pass) {
        const float* logits = pass == 0 ? a.conf : a.prop_conf;
        float* gout = pass == 0 ? g_conf : g_pconf;
        const float norm = pass == 0 ? Nf : PNf;
        if (a.cls_mode == 1) {
            // FocalLoss_Ori(gamma 2, size_average False) on F.softmax(logits) of the positive rows (cls_loss.py:6-78,
            // multisegment_loss.py:196-216): loss = -alpha_y (1 - pt)^2 log(pt), pt = p_y + 1e-6
            float part = 0.f;
            for (int i = t; i < A; i += LT) {
                const float* s = a.scratch + (size_t)i * SCR;
                const int tgt = (int)s[2 + pass];
                const float* z = logits + (size_t)i * C;
                float* gz = gout + (size_t)i * C;
                if (tgt > 0) {
                    const int y = tgt - 1;
                    float mx = z[0];
                    for (int k = 1; k < C; ++k) mx = fmaxf(mx, z[k]);
                    float S = 0.f;
                    for (int k = 0; k < C; ++k) S += expf(z[k] - mx);
                    const float py = expf(z[y] - mx) / S;
                    const float pt = py + 1e-6f;
                    const float al = y == 0 ? a.focal_alpha : 1.f - a.focal_alpha;
                    const float om = 1.f - pt, lg = logf(pt);
                    part += -(om * om) * (al * lg);
                    const float dpt = -al * (-2.f * om * lg + om * om / pt);       // d loss / d pt
                    for (int k = 0; k < C; ++k) {
                        const float pk = expf(z[k] - mx) / S;
                        gz[k] = dpt * py * ((k == y ? 1.f : 0.f) - pk) / norm;
                    }
                } else {
                    for (int k = 0; k < C; ++k) gz[k] = 0.f;
                }
            }
            loss_cls[pass] = block_sum(part, red) / norm;
            continue;
        }
        const bool staged = ST;
        const int AC = A * C;
        if constexpr (ST) {      // (16-byte pieces when the tensor allows, the tail and unaligned tensors element-wise)
            __syncthreads();     // the previous pass's copy-out has read zl
            const int n4 = (reinterpret_cast<uintptr_t>(logits) & 15) == 0 ? AC >> 2 : 0;
            for (int q = t; q < n4; q += LT) reinterpret_cast<float4*>(zl)[q] = reinterpret_cast<const float4*>(logits)[q];
            for (int q = 4 * n4 + t; q < AC; q += LT) zl[q] = logits[q];
            __syncthreads();
        }
        const bool cal = ST && pass == 1 && a.iou_aware;
        for (int i = t; i < A; i += LT) {
            float* s = a.scratch + (size_t)i * SCR;
            const int tgt = (int)s[2 + pass];
            const int y = max(tgt - 1, 0);
            const Edl e = edl_row(staged ? zl + (size_t)i * C : logits + (size_t)i * C, C, y, a.num_bins);
            s[7] = e.ghat; s[8] = (float)e.slot; s[9] = (float)e.binpos; s[10] = e.per;
            s_val[i] = e.ghat; s_slot[i] = (unsigned char)e.slot; s_flag[i] = (tgt > 0 && e.binpos) ? 1 : 0;
        }
        __syncthreads();
        if (a.ibm_active) {     // the 50-bin EMA, deterministic: thread = (segment of anchors, bin) sums its segment in
                                // index order, then one thread per bin adds the HSEG partials in segment order
            const int bin = t % MAX_BINS, seg = t / MAX_BINS;
            const int per_seg = (A + HSEG - 1) / HSEG;
            float tot = 0.f, cnt = 0.f;
            if (bin < a.num_bins) {
                const int hi = min(A, (seg + 1) * per_seg);
                for (int i = seg * per_seg; i < hi; ++i)
                    if (s_flag[i] && s_slot[i] == bin) { tot += s_val[i]; cnt += 1.f; }
            }
            h_tot[seg][bin] = tot; h_cnt[seg][bin] = cnt;
            __syncthreads();
            if (t < a.num_bins) {
                float tt = 0.f, cc = 0.f;
                for (int sg = 0; sg < HSEG; ++sg) { tt += h_tot[sg][t]; cc += h_cnt[sg][t]; }
                if (cc > 0.f) wacc[t] = a.momentum * wacc[t] + (1.f - a.momentum) * tt / fmaxf(cc, 1.f);
            }
            __syncthreads();
        }
        float part = 0.f, part_cal = 0.f;
        for (int i = t; i < A; i += LT) {
            const float* s = a.scratch + (size_t)i * SCR;
            const int tgt = (int)s[2 + pass];
            if constexpr (ST) {
                float* z = zl + (size_t)i * C;                  // logits in, gradient row out (in place: element k is read before it is written)
                const float wgt = (tgt > 0 && a.ibm_active) ? wacc[(int)s[8]] : 1.f;
                if (tgt > 0) part += wgt * s[10];
                const int y = tgt - 1;
                float S = 0.f, ay = 1.f;
                if (tgt > 0 || cal) {
                    for (int k = 0; k < C; ++k) {
                        const float al = expf(fminf(fmaxf(z[k], -10.f), 10.f)) + 1.f;
                        S += al;
                        if (k == y) ay = al;
                    }
                }
                float cg = 0.f;                                 // IoU calibration (the separate loop of the unstaged form, see there)
                if (cal) {
                    const int kk = i / a.B, bb = i - kk * a.B;
                    float iou = a.scratch[(size_t)(bb * a.K + kk) * SCR + 4];
                    if (iou < 0.f) iou = 1e-3f;
                    const float u = (float)C / S;
                    part_cal += -iou * logf(1.f - u) - (1.f - iou) * logf(u);
                    const float dreg_du = iou / (1.f - u) - (1.f - iou) / u;
                    cg = dreg_du * (-(float)C / (S * S));
                }
                for (int k = 0; k < C; ++k) {
                    const float zk = z[k];
                    const float da = (zk >= -10.f && zk <= 10.f) ? expf(zk) : 0.f;      // clamp backward is inclusive
                    float gk = tgt > 0 ? wgt * (1.f / S - (k == y ? 1.f / ay : 0.f)) * da / norm : 0.f;
                    if (cal) gk += cg * da / (float)A;
                    z[k] = gk;
                }
            } else {
            const float* z = logits + (size_t)i * C;
            float* gz = gout + (size_t)i * C;
            if (tgt > 0) {
                const float wgt = a.ibm_active ? wacc[(int)s[8]] : 1.f;
                part += wgt * s[10];
                const int y = tgt - 1;
                float S = 0.f, ay = 1.f;
                for (int k = 0; k < C; ++k) {
                    const float al = expf(fminf(fmaxf(z[k], -10.f), 10.f)) + 1.f;
                    S += al;
                    if (k == y) ay = al;
                }
                for (int k = 0; k < C; ++k) {
                    const float zk = z[k];
                    const float da = (zk >= -10.f && zk <= 10.f) ? expf(zk) : 0.f;      // clamp backward is inclusive
                    gz[k] = wgt * (1.f / S - (k == y ? 1.f / ay : 0.f)) * da / norm;
                }
            } else {
                for (int k = 0; k < C; ++k) gz[k] = 0.f;
            }
            }
        }
        if constexpr (ST) {
            __syncthreads();
            const int n4 = (reinterpret_cast<uintptr_t>(gout) & 15) == 0 ? AC >> 2 : 0;
            for (int q = t; q < n4; q += LT) reinterpret_cast<float4*>(gout)[q] = reinterpret_cast<const float4*>(zl)[q];
            for (int q = 4 * n4 + t; q < AC; q += LT) gout[q] = zl[q];
        }
        loss_cls[pass] = block_sum(part, red) / norm;
        if (cal) loss_cls[1] += block_sum(part_cal, red) / (float)A;
    }
    if (t < a.num_bins) a.weight_accum[t] = wacc[t];

    // ---- IoU calibration on prop_conf (cls_loss.py:120-129; pairing quirk of multisegment_loss.py:234-236)
    if (!ST && a.iou_aware) {
        float part = 0.f;
        for (int j = t; j < A; j += LT) {
            const int kk = j / a.B, bb = j - kk * a.B;                 // iou_pred.transpose(0,1).reshape(-1)[j]
            float iou = a.scratch[(size_t)(bb * a.K + kk) * SCR + 4];
            if (iou < 0.f) iou = 1e-3f;
            const float* z = a.prop_conf + (size_t)j * C;
            float S = 0.f;
            for (int k = 0; k < C; ++k) S += expf(fminf(fmaxf(z[k], -10.f), 10.f)) + 1.f;
            const float u = (float)C / S;
            part += -iou * logf(1.f - u) - (1.f - iou) * logf(u);
            const float dreg_du = iou / (1.f - u) - (1.f - iou) / u;
            float* gz = g_pconf + (size_t)j * C;
            for (int k = 0; k < C; ++k) {
                const float zk = z[k];
                const float da = (zk >= -10.f && zk <= 10.f) ? expf(zk) : 0.f;
                gz[k] += dreg_du * (-(float)C / (S * S)) * da / (float)A;
            }
        }
        loss_cls[1] += block_sum(part, red) / (float)A;
    }

    // ---- localisation (GIoU over positives), refined L1, quality BCE with the non-detached tIoU target
    float pl = 0.f, ppl = 0.f, pct = 0.f;
    for (int i = t; i < A; i += LT) {
        const float* s = a.scratch + (size_t)i * SCR;
        const bool pos = s[2] > 0.f, ppos = s[3] > 0.f;
        const float p0 = a.loc[2 * i], p1 = a.loc[2 * i + 1], t0 = s[0], t1 = s[1];
        float dl0 = 0.f, dl1 = 0.f, dct_l0 = 0.f, dct_l1 = 0.f, dct_p0 = 0.f, dct_p1 = 0.f, dcen = 0.f;
        if (pos) {
            const TIoU q = tiou_grad(p0, p1, t0, t1);
            const float hull = fmaxf(p0, t0) + fmaxf(p1, t1);
            const float hc = fmaxf(hull, F_EPS), hlive = hull >= F_EPS ? 1.f : 0.f;
            pl += 1.f - (q.iou - (hull - q.uni) / hc);
            const float dh0 = dmax_da(p0, t0), dh1 = dmax_da(p1, t1);
            const float du0 = 1.f - dmin_da(p0, t0), du1 = 1.f - dmin_da(p1, t1);
            dl0 = (-q.d0 + ((dh0 - du0) * hc - (hull - q.uni) * dh0 * hlive) / (hc * hc)) / Nf;
            dl1 = (-q.d1 + ((dh1 - du1) * hc - (hull - q.uni) * dh1 * hlive) / (hc * hc)) / Nf;
            // quality head: cur = 0.5 * w * prop_loc + loc
            const float w = p0 + p1, r0 = a.prop_loc[2 * i], r1 = a.prop_loc[2 * i + 1];
            const float c0 = 0.5f * w * r0 + p0, c1 = 0.5f * w * r1 + p1;
            const TIoU qq = tiou_grad(c0, c1, t0, t1);
            const float qv = fmaxf(qq.iou, 0.f), qlive = qq.iou >= 0.f ? 1.f : 0.f;
            const float x = a.center[i];
            const float ex = expf(-fabsf(x));
            pct += fmaxf(x, 0.f) - x * qv + log1pf(ex);
            const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
            dcen = ((x >= 0.f ? 1.f : 0.f) - qv - sg * ex / (1.f + ex)) / Nf;
            const float gq0 = -x * qlive * qq.d0 / Nf, gq1 = -x * qlive * qq.d1 / Nf;   // d loss / d cur
            dct_p0 = gq0 * 0.5f * w; dct_p1 = gq1 * 0.5f * w;
            const float common = 0.5f * (gq0 * r0 + gq1 * r1);
            dct_l0 = common + gq0; dct_l1 = common + gq1;
        }
        float dpp0 = 0.f, dpp1 = 0.f;
        if (ppos) {
            const float e0 = a.prop_loc[2 * i] - s[5], e1 = a.prop_loc[2 * i + 1] - s[6];
            ppl += fabsf(e0) + fabsf(e1);
            dpp0 = (e0 > 0.f ? 1.f : (e0 < 0.f ? -1.f : 0.f)) / PNf;
            dpp1 = (e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f)) / PNf;
        }
        g_loc_l[2 * i] = dl0; g_loc_l[2 * i + 1] = dl1;
        g_loc_ct[2 * i] = dct_l0; g_loc_ct[2 * i + 1] = dct_l1;
        g_pl_pl[2 * i] = dpp0; g_pl_pl[2 * i + 1] = dpp1;
        g_pl_ct[2 * i] = dct_p0; g_pl_ct[2 * i + 1] = dct_p1;
        g_center[i] = dcen;
    }
    const float loss_l = block_sum(pl, red) / Nf;
    const float loss_pl = block_sum(ppl, red) / PNf;
    const float loss_ct = block_sum(pct, red) / Nf;

    // ---- positive-unlabelled actionness BCE (cls_loss.py:288-339), rank term off (weight 0 in the final recipe)
    float loss_a[2];
    for (int pass = 0; pass < 2; ++pass) {
        const float* pred = pass == 0 ? a.act : a.prop_act;
        float* gout = pass == 0 ? g_act : g_pact;
        const int np = icnt[pass], nn = A - np;
        const int top_m = min(np, nn) - 1;
        // rank of every negative among the negatives (ascending score, ties by index) = its position after an exact
        // bitonic sort of 64-bit keys in LDS; positives and padding carry the largest key and sort behind them
        int used_cnt = 0;
        int n2 = 1;
        while (n2 < A) n2 <<= 1;
        __syncthreads();
        for (int i = t; i < n2; i += LT) {
            unsigned hi = 0xffffffffu;
            if (i < A && !(a.scratch[(size_t)i * SCR + 2 + pass] > 0.f)) {
                const unsigned u = __float_as_uint(pred[i]);
                hi = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
                if (hi == 0xffffffffu) hi = 0xfffffffeu;             // keep the sentinel unique to non-negatives
            }
            skey[i] = ((unsigned long long)hi << 32) | (unsigned)i;
            if (i < A) s_flag[i] = (hi == 0xffffffffu) ? 1 : (top_m > 0 ? 0 : 1);      // used: positives; negatives decided below
        }
        __syncthreads();
        if (top_m > 0) {
            // a stage of distance j < 64 exchanges keys inside 64-aligned groups of indices = inside one WAVE (i = t mod LT):
            // it needs the workgroup barrier only if its keys were written, or will next be read, across waves (the stage
            // before / after has distance >= 64, or the sort ends); otherwise the wave's own LDS order (+ a wave barrier) is
            // enough -- 14 workgroup barriers per sort of 1024 keys instead of 55
            for (int k = 2; k <= n2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = t; i < n2; i += LT) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long x = skey[i], y = skey[ixj];
                            const bool up = (i & k) == 0;
                            if ((x > y) == up) { skey[i] = y; skey[ixj] = x; }
                        }
                    }
                    const int jn = j > 1 ? (j >> 1) : k;                 // distance of the next stage (k = the last stage's successor reads everything)
                    if (j >= 64 || jn >= 64 || (k == n2 && j == 1)) __syncthreads();
                    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
                }
            for (int p_ = t; p_ < min(top_m, A); p_ += LT) {
                const unsigned long long kv = skey[p_];
                if ((unsigned)(kv >> 32) != 0xffffffffu) s_flag[(unsigned)kv] = 1;
            }
            __syncthreads();
        }
        for (int i = t; i < A; i += LT) {
            float* s = a.scratch + (size_t)i * SCR;
            const bool used = s_flag[i] != 0;
            s[11] = used ? 1.f : 0.f;
            used_cnt += used;
        }
        __syncthreads();
        if (t == 0) icnt[2 + pass] = 0;
        __syncthreads();
        atomicAdd(&icnt[2 + pass], used_cnt);
        __syncthreads();
        const float cntf = (float)icnt[2 + pass];
        float part = 0.f;
        for (int i = t; i < A; i += LT) {
            const float* s = a.scratch + (size_t)i * SCR;
            const float x = pred[i], y = s[2 + pass] > 0.f ? 1.f : 0.f;
            float gx = 0.f;
            if (s[11] > 0.f) {
                part += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
                gx = (1.f / (1.f + expf(-x)) - y) / cntf;
            }
            gout[i] = gx;
        }
        loss_a[pass] = block_sum(part, red) / cntf;
        __syncthreads();
    }

    if (t == 0) {
        a.losses[0] = loss_l; a.losses[1] = loss_cls[0]; a.losses[2] = loss_pl; a.losses[3] = loss_cls[1];
        a.losses[4] = loss_ct; a.losses[5] = loss_a[0]; a.losses[6] = loss_a[1];
    }
}


// =================================================================================================
// ActivityNet1.3 recipe (AFSD/anet/multisegment_loss.py:87-301, anet/cls_loss.py:78-296): every term is evaluated PER SAMPLE
// and normalised by that sample's own counts before the mean over the batch, so a workgroup owns ONE sample (K anchors,
// 189 for 768-frame clips; up to 150 classes) and a second tiny launch sums the B x 7 terms in sample order.  What differs
// from the THUMOS14 kernel above and is reproduced: per-level regression bounds on max(left, right) in the matching
// (:69-84, :156-166); refined-stage positives use min(overlap_thresh, best tIoU among the sample's positives) (:178-184);
// smooth-L1 for the refinement (:206); the influence-balanced weight is the closed form 1 / (|z|_1 exp(coeff g) + 1e-10)
// with |z|_1 NOT detached (cls_loss.py:137, :225-232; no EMA state); the actionness loss keeps its rank hinge (weight 0.1);
// the IoU calibration is each sample's own mean.  Gradients use the layout of otal_detection_loss, already divided by B.
// It replaced ~250 ATen launches of this package's own torch formulation of that file (2.4 ms of kernel time per step).
constexpr int LA = 1024;            // threads per sample: FOUR lanes per anchor -- the class loops (150 classes, five passes
constexpr int LQ = LA / 4;          // with an exp each) are split over them and finished with two shuffles; 256 anchors per sweep
constexpr int MAX_KA = 1024;        // anchors per sample
constexpr int MAX_LEVELS_A = 8;

struct AnetLossArgs {
    const float *loc, *conf, *prop_loc, *prop_conf, *center, *act, *prop_act, *priors, *gt;   // priors (K,2): centre, level
    const unsigned char* gvalid;
    float* terms;                   // (B, 8): the seven per-sample terms
    float* grads;
    int B, K, C, G;
    float clip, overlap;
    int ibm_active, iou_aware;
    float coeff, act_weight, act_margin;
    float lb[MAX_LEVELS_A], rb[MAX_LEVELS_A];
    int nlev;
};

__device__ __forceinline__ float quad_sum(float v) {        // sum over the four lanes of an anchor (same wave)
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}
__device__ float bsumA(float v, float* red) {
    const int t = threadIdx.x;
    __syncthreads();
    red[t] = v;
    __syncthreads();
    for (int s = LA / 2; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}
// maximum and the FIRST index that holds it (torch.max(dim) backward sends the gradient to one index; ties are measure-zero)
__device__ float bmaxA(float v, int idx, float* red, int* redi, int* arg) {
    const int t = threadIdx.x;
    __syncthreads();
    red[t] = v; redi[t] = idx;
    __syncthreads();
    for (int s = LA / 2; s > 0; s >>= 1) {
        if (t < s) {
            const float o = red[t + s];
            const int oi = redi[t + s];
            if (o > red[t] || (o == red[t] && oi < redi[t])) { red[t] = o; redi[t] = oi; }
        }
        __syncthreads();
    }
    const float r = red[0];
    if (arg) *arg = redi[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(LA) void detection_loss_anet_kernel(const AnetLossArgs a) {
    __shared__ float red[LA];
    __shared__ int redi[LA];
    __shared__ float s_lt0[MAX_KA], s_lt1[MAX_KA], s_iou[MAX_KA], s_pl0[MAX_KA], s_pl1[MAX_KA], s_pred[MAX_KA];
    __shared__ short s_ct[MAX_KA], s_pct[MAX_KA];
    __shared__ unsigned char s_used[MAX_KA];
    const int t = threadIdx.x, b = blockIdx.x;
    const int q4 = t >> 2, sub = t & 3;             // anchor slot of this lane and its share of the class / rank loops
    const bool lead = sub == 0;                     // per-anchor scalar work and partial sums: one lane of the four
    const int K = a.K, C = a.C, A = a.B * a.K;
    float* g_loc_l = a.grads;
    float* g_loc_ct = g_loc_l + 2 * (size_t)A;
    float* g_pl_pl = g_loc_ct + 2 * (size_t)A;
    float* g_pl_ct = g_pl_pl + 2 * (size_t)A;
    float* g_conf = g_pl_ct + 2 * (size_t)A;
    float* g_pconf = g_conf + (size_t)A * C;
    float* g_center = g_pconf + (size_t)A * C;
    float* g_act = g_center + A;
    float* g_pact = g_act + A;

    // ---- matching (anet/multisegment_loss.py:144-188), no gradient
    const float big = a.clip * 2.f;
    float npos_f = 0.f, my_best = -INFINITY;
    for (int k = t; k < K; k += LA) {
        const int i = b * K + k;
        const float c = a.priors[2 * k];
        int lvl = (int)a.priors[2 * k + 1];
        lvl = lvl < 0 ? 0 : (lvl >= a.nlev ? a.nlev - 1 : lvl);
        const float lb = a.lb[lvl], rb = a.rb[lvl];
        float best_area = 0.f;
        int best = 0;
        for (int g = 0; g < a.G; ++g) {
            const float left = (c - a.gt[(b * a.G + g) * 3]) * a.clip, right = (a.gt[(b * a.G + g) * 3 + 1] - c) * a.clip;
            const float far = fmaxf(left, right);
            float area = left + right;
            if (left < 0.f || right < 0.f || far <= lb || far > rb || !a.gvalid[b * a.G + g]) area = big;
            if (g == 0 || area < best_area) { best_area = area; best = g; }     // first minimum, like torch.min
        }
        const float g0 = a.gt[(b * a.G + best) * 3], g1 = a.gt[(b * a.G + best) * 3 + 1], lab = a.gt[(b * a.G + best) * 3 + 2];
        const float lt0 = (c - g0) * a.clip, lt1 = (g1 - c) * a.clip;
        const int conf_t = best_area >= big ? 0 : (int)lab;
        const float p0 = a.loc[2 * i], p1 = a.loc[2 * i + 1];
        const TIoU q = tiou_grad(p0, p1, lt0, lt1);
        const float w = p0 + p1;
        s_lt0[k] = lt0; s_lt1[k] = lt1; s_ct[k] = (short)conf_t; s_iou[k] = q.iou;
        s_pl0[k] = (lt0 - p0) / (0.5f * w); s_pl1[k] = (lt1 - p1) / (0.5f * w);
        if (conf_t > 0) { npos_f += 1.f; my_best = fmaxf(my_best, q.iou); }
    }
    const float npos = bsumA(npos_f, red);
    const float best_iou = bmaxA(my_best, t, red, redi, nullptr);
    const float thr = npos > 0.f ? fminf(best_iou, a.overlap) : a.overlap;
    float nppos_f = 0.f;
    for (int k = t; k < K; k += LA) {
        const int pct = s_iou[k] < thr ? 0 : (int)s_ct[k];
        s_pct[k] = (short)pct;
        nppos_f += pct > 0 ? 1.f : 0.f;
    }
    const float nppos = bsumA(nppos_f, red);
    const float Nf = fmaxf(npos, 1.f), PNf = fmaxf(nppos, 1.f);

    // ---- classification: EvidenceLoss 'log' with exp evidence (anet/cls_loss.py:120-141), per sample
    float loss_cls[2];
    for (int pass = 0; pass < 2; ++pass) {
        const float* logits = pass == 0 ? a.conf : a.prop_conf;
        float* gout = pass == 0 ? g_conf : g_pconf;
        const float norm = (pass == 0 ? Nf : PNf) * (float)a.B;
        float part = 0.f;
        for (int k0 = 0; k0 < K; k0 += LQ) {            // (uniform trip count: the shuffles below need all four lanes)
            const int k = k0 + q4;
            const bool live = k < K;
            const int i = b * K + (live ? k : 0);
            const int tgt = !live ? 0 : (pass == 0 ? (int)s_ct[k] : (int)s_pct[k]);
            const float* z = logits + (size_t)i * C;
            float* gz = gout + (size_t)i * C;
            const int y = tgt - 1;
            float S = 0.f, ay = 0.f, l1 = 0.f;
            if (tgt > 0)
                for (int c = sub; c < C; c += 4) {
                    const float al = expf(fminf(fmaxf(z[c], -10.f), 10.f)) + 1.f;
                    S += al;
                    l1 += fabsf(z[c]);
                    if (c == y) ay = al;
                }
            S = quad_sum(S); l1 = quad_sum(l1); ay = quad_sum(ay);
            if (tgt > 0) {
                const float per0 = logf(S) - logf(ay);
                float invD = 1.f, dfn = 0.f;            // per = per0 * invD;  d per / d |z|_1 = dfn
                if (a.ibm_active) {
                    const float e = expf(a.coeff * fabsf(1.f / ay - (float)C / S));       // exp(coeff * g), g detached
                    const float D = l1 * e + 1e-10f;
                    invD = 1.f / D;
                    dfn = -per0 * e / (D * D);
                }
                if (lead) part += per0 * invD;
                for (int c = sub; c < C; c += 4) {
                    const float zc = z[c];
                    const float da = (zc >= -10.f && zc <= 10.f) ? expf(zc) : 0.f;     // clamp backward is inclusive
                    const float sg = zc > 0.f ? 1.f : (zc < 0.f ? -1.f : 0.f);
                    gz[c] = ((1.f / S - (c == y ? 1.f / ay : 0.f)) * da * invD + dfn * sg) / norm;
                }
            } else if (live) {
                for (int c = sub; c < C; c += 4) gz[c] = 0.f;
            }
        }
        loss_cls[pass] = bsumA(part, red) / (pass == 0 ? Nf : PNf);
    }

    // ---- IoU calibration on prop_conf: this sample's mean over its K anchors (anet/multisegment_loss.py:259-261)
    if (a.iou_aware) {
        float part = 0.f;
        const float norm = (float)K * (float)a.B;
        for (int k0 = 0; k0 < K; k0 += LQ) {
            const int k = k0 + q4;
            const bool live = k < K;
            const int i = b * K + (live ? k : 0);
            float iou = live ? s_iou[k] : 0.f;
            if (iou < 0.f) iou = 1e-3f;
            const float* z = a.prop_conf + (size_t)i * C;
            float S = 0.f;
            if (live)
                for (int c = sub; c < C; c += 4) S += expf(fminf(fmaxf(z[c], -10.f), 10.f)) + 1.f;
            S = quad_sum(S);
            if (live) {
                const float u = (float)C / S;
                if (lead) part += -iou * logf(1.f - u) - (1.f - iou) * logf(u);
                const float dreg_du = iou / (1.f - u) - (1.f - iou) / u;
                float* gz = g_pconf + (size_t)i * C;
                for (int c = sub; c < C; c += 4) {
                    const float zc = z[c];
                    const float da = (zc >= -10.f && zc <= 10.f) ? expf(zc) : 0.f;
                    gz[c] += dreg_du * (-(float)C / (S * S)) * da / norm;
                }
            }
        }
        loss_cls[1] += bsumA(part, red) / (float)K;
    }

    // ---- localisation (GIoU over positives), refined smooth-L1, quality BCE with the non-detached tIoU target
    float pl = 0.f, ppl = 0.f, pct = 0.f;
    const float nN = Nf * (float)a.B, nPN = PNf * (float)a.B;
    for (int k = t; k < K; k += LA) {
        const int i = b * K + k;
        const bool pos = s_ct[k] > 0, ppos = s_pct[k] > 0;
        const float p0 = a.loc[2 * i], p1 = a.loc[2 * i + 1], t0 = s_lt0[k], t1 = s_lt1[k];
        float dl0 = 0.f, dl1 = 0.f, dct_l0 = 0.f, dct_l1 = 0.f, dct_p0 = 0.f, dct_p1 = 0.f, dcen = 0.f;
        if (pos) {
            const TIoU q = tiou_grad(p0, p1, t0, t1);
            const float hull = fmaxf(p0, t0) + fmaxf(p1, t1);
            const float hc = fmaxf(hull, F_EPS), hlive = hull >= F_EPS ? 1.f : 0.f;
            pl += 1.f - (q.iou - (hull - q.uni) / hc);
            const float dh0 = dmax_da(p0, t0), dh1 = dmax_da(p1, t1);
            const float du0 = 1.f - dmin_da(p0, t0), du1 = 1.f - dmin_da(p1, t1);
            dl0 = (-q.d0 + ((dh0 - du0) * hc - (hull - q.uni) * dh0 * hlive) / (hc * hc)) / nN;
            dl1 = (-q.d1 + ((dh1 - du1) * hc - (hull - q.uni) * dh1 * hlive) / (hc * hc)) / nN;
            const float w = p0 + p1, r0 = a.prop_loc[2 * i], r1 = a.prop_loc[2 * i + 1];
            const float c0 = 0.5f * w * r0 + p0, c1 = 0.5f * w * r1 + p1;
            const TIoU qq = tiou_grad(c0, c1, t0, t1);
            const float qv = fmaxf(qq.iou, 0.f), qlive = qq.iou >= 0.f ? 1.f : 0.f;
            const float x = a.center[i];
            const float ex = expf(-fabsf(x));
            pct += fmaxf(x, 0.f) - x * qv + log1pf(ex);
            const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
            dcen = ((x >= 0.f ? 1.f : 0.f) - qv - sg * ex / (1.f + ex)) / nN;
            const float gq0 = -x * qlive * qq.d0 / nN, gq1 = -x * qlive * qq.d1 / nN;
            dct_p0 = gq0 * 0.5f * w; dct_p1 = gq1 * 0.5f * w;
            const float common = 0.5f * (gq0 * r0 + gq1 * r1);
            dct_l0 = common + gq0; dct_l1 = common + gq1;
        }
        float dpp0 = 0.f, dpp1 = 0.f;
        if (ppos) {
            const float e0 = a.prop_loc[2 * i] - s_pl0[k], e1 = a.prop_loc[2 * i + 1] - s_pl1[k];
            const float d0 = fabsf(e0), d1 = fabsf(e1);
            ppl += (d0 < 1.f ? 0.5f * d0 * d0 : d0 - 0.5f) + (d1 < 1.f ? 0.5f * d1 * d1 : d1 - 0.5f);
            dpp0 = (d0 < 1.f ? e0 : (e0 > 0.f ? 1.f : -1.f)) / nPN;
            dpp1 = (d1 < 1.f ? e1 : (e1 > 0.f ? 1.f : -1.f)) / nPN;
        }
        g_loc_l[2 * i] = dl0; g_loc_l[2 * i + 1] = dl1;
        g_loc_ct[2 * i] = dct_l0; g_loc_ct[2 * i + 1] = dct_l1;
        g_pl_pl[2 * i] = dpp0; g_pl_pl[2 * i + 1] = dpp1;
        g_pl_ct[2 * i] = dct_p0; g_pl_ct[2 * i + 1] = dct_p1;
        g_center[i] = dcen;
    }
    const float loss_l = bsumA(pl, red) / Nf;
    const float loss_pl = bsumA(ppl, red) / PNf;
    const float loss_ct = bsumA(pct, red) / Nf;

    // ---- positive-unlabelled actionness BCE with the rank hinge (anet/cls_loss.py:249-296), per sample
    float loss_a[2];
    for (int pass = 0; pass < 2; ++pass) {
        const float* pred = (pass == 0 ? a.act : a.prop_act) + (size_t)b * K;
        float* gout = (pass == 0 ? g_act : g_pact) + (size_t)b * K;
        const short* tg = pass == 0 ? s_ct : s_pct;
        const int np = (int)(pass == 0 ? npos : nppos), nn = K - np;
        const int top_m = min(np, nn) - 1;
        __syncthreads();
        for (int k = t; k < K; k += LA) s_pred[k] = pred[k];
        __syncthreads();
        float used_f = 0.f, nmax = -INFINITY, pmax = -INFINITY;
        int nmax_i = 0x7fffffff;
        for (int k0 = 0; k0 < K; k0 += LQ) {
            const int k = k0 + q4;
            const bool live = k < K;
            const bool pos = live && tg[k] > 0;
            const float x = live ? s_pred[k] : 0.f;
            int rank = 0;
            if (live && !pos && top_m > 0)      // rank among the negatives: ascending score, ties by index; a quarter per lane
                for (int j = sub; j < K; j += 4)
                    if (!(tg[j] > 0)) rank += (s_pred[j] < x || (s_pred[j] == x && j < k)) ? 1 : 0;
            rank += __shfl_xor(rank, 1);
            rank += __shfl_xor(rank, 2);
            if (live && lead) {
                const bool used = pos || top_m <= 0 || rank < top_m;
                s_used[k] = used ? 1 : 0;
                used_f += used ? 1.f : 0.f;
                if (pos) pmax = fmaxf(pmax, x);
                else if (x > nmax) { nmax = x; nmax_i = k; }
            }
        }
        const float cnt = bsumA(used_f, red);
        int arg_n = 0;
        const float neg_max = bmaxA(nmax, nmax_i, red, redi, &arg_n);
        const float pos_max = bmaxA(pmax, t, red, redi, nullptr);
        // torch: where(neg, pred, -finfo.max).max(); with no negative / positive the hinge is switched off by top_m <= 0
        float hinge = 0.f, dh = 0.f;
        if (top_m > 0 && a.act_weight != 0.f) {
            const float v = a.act_margin - neg_max + pos_max;
            if (v >= 0.f) { hinge = a.act_weight * v; dh = -a.act_weight; }
        }
        const float normA = cnt * (float)a.B;
        float part = 0.f;
        for (int k = t; k < K; k += LA) {
            const float x = s_pred[k], y = tg[k] > 0 ? 1.f : 0.f;
            float gx = 0.f;
            if (s_used[k]) {
                part += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
                gx = (1.f / (1.f + expf(-x)) - y) / normA;
            }
            if (dh != 0.f && k == arg_n) gx += dh / normA;
            gout[k] = gx;
        }
        loss_a[pass] = (bsumA(part, red) + hinge) / cnt;
    }
    if (t == 0) {
        float* o = a.terms + 8 * b;
        o[0] = loss_l; o[1] = loss_cls[0]; o[2] = loss_pl; o[3] = loss_cls[1]; o[4] = loss_ct; o[5] = loss_a[0]; o[6] = loss_a[1];
    }
}

__global__ void detection_loss_anet_finish_kernel(const float* __restrict__ terms, float* __restrict__ losses, int B) {
    const int i = threadIdx.x;
    if (i >= 7) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += terms[8 * b + i];      // sample order, as the reference's python loop
    losses[i] = s / (float)B;
}

}  // namespace

extern "C" size_t otal_detection_loss_scratch_floats(int B, int K) { return (size_t)B * K * SCR; }

extern "C" size_t otal_detection_loss_grad_floats(int B, int K, int C) {
    const size_t A = (size_t)B * K;
    return 4 * 2 * A + 2 * A * C + 3 * A;
}

extern "C" int otal_detection_loss(const float* loc, const float* conf, const float* prop_loc, const float* prop_conf,
                                   const float* center, const float* act, const float* prop_act, const float* priors,
                                   const float* gt, const unsigned char* gvalid, float* weight_accum, int B, int K,
                                   int C, int G, float clip_length, float overlap_thresh, int ibm_active, int num_bins,
                                   float momentum, int iou_aware, int cls_mode, float focal_alpha, float* losses,
                                   float* grads, float* scratch, void* stream) {
    if (!loc || !conf || !prop_loc || !prop_conf || !center || !act || !prop_act || !priors || !gt || !gvalid ||
        !weight_accum || !losses || !grads || !scratch) return OTAL_E_NULL;
    if (B <= 0 || K <= 0 || C <= 0 || G <= 0) return OTAL_E_SHAPE;
    if (num_bins <= 0 || num_bins > MAX_BINS || (long)B * K > MAX_A || cls_mode < 0 || cls_mode > 1) return OTAL_E_UNSUPPORTED;
    LossArgs a;
    a.loc = loc; a.conf = conf; a.prop_loc = prop_loc; a.prop_conf = prop_conf; a.center = center; a.act = act;
    a.prop_act = prop_act; a.priors = priors; a.gt = gt; a.gvalid = gvalid; a.weight_accum = weight_accum;
    a.losses = losses; a.grads = grads; a.scratch = scratch;
    a.B = B; a.K = K; a.C = C; a.G = G; a.clip = clip_length; a.overlap = overlap_thresh;
    a.ibm_active = cls_mode == 0 ? ibm_active : 0; a.num_bins = num_bins; a.iou_aware = cls_mode == 0 ? iou_aware : 0;
    a.momentum = momentum; a.cls_mode = cls_mode; a.focal_alpha = focal_alpha;
    // staged logits: EvidenceLoss mode and A * C floats within the dynamic LDS this kernel may add to its ~40 KB of static arrays
    const size_t stage = (size_t)B * K * C * sizeof(float);
    constexpr size_t STAGE_MAX = 96 * 1024;
    static int staged_ok = -1;          // -1: not asked yet; the attribute is set once per process
    if (cls_mode == 0 && stage <= STAGE_MAX && !OTAL_OPT("OTAL_LOSS_NOSTAGE", 0)) {
        if (staged_ok < 0)
            staged_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(detection_loss_kernel<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)STAGE_MAX) == hipSuccess ? 1 : 0;
        if (staged_ok == 1) {
            hipLaunchKernelGGL(detection_loss_kernel<true>, dim3(1), dim3(LT), stage, (hipStream_t)stream, a);
            return otal_launch_status();
        }
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(detection_loss_kernel<false>, dim3(1), dim3(LT), 0, (hipStream_t)stream, a);
    return otal_launch_status();
}


extern "C" int otal_detection_loss_anet(const float* loc, const float* conf, const float* prop_loc, const float* prop_conf,
                                        const float* center, const float* act, const float* prop_act, const float* priors2,
                                        const float* gt, const unsigned char* gvalid, int B, int K, int C, int G,
                                        float clip_length, float overlap_thresh, const float* level_bounds, int nlev,
                                        int ibm_active, float ibm_coeff, int iou_aware, float act_weight, float act_margin,
                                        float* losses, float* grads, float* scratch, void* stream) {
    if (!loc || !conf || !prop_loc || !prop_conf || !center || !act || !prop_act || !priors2 || !gt || !gvalid ||
        !level_bounds || !losses || !grads || !scratch) return OTAL_E_NULL;
    if (B <= 0 || K <= 0 || C <= 0 || G <= 0 || nlev <= 0) return OTAL_E_SHAPE;
    if (K > MAX_KA || nlev > MAX_LEVELS_A) return OTAL_E_UNSUPPORTED;
    AnetLossArgs a;
    a.loc = loc; a.conf = conf; a.prop_loc = prop_loc; a.prop_conf = prop_conf; a.center = center; a.act = act;
    a.prop_act = prop_act; a.priors = priors2; a.gt = gt; a.gvalid = gvalid; a.terms = scratch; a.grads = grads;
    a.B = B; a.K = K; a.C = C; a.G = G; a.clip = clip_length; a.overlap = overlap_thresh;
    a.ibm_active = ibm_active; a.iou_aware = iou_aware; a.coeff = ibm_coeff; a.act_weight = act_weight; a.act_margin = act_margin;
    a.nlev = nlev;
    for (int l = 0; l < MAX_LEVELS_A; ++l) { a.lb[l] = l < nlev ? level_bounds[2 * l] : 0.f; a.rb[l] = l < nlev ? level_bounds[2 * l + 1] : 0.f; }
    hipLaunchKernelGGL(detection_loss_anet_kernel, dim3(B), dim3(LA), 0, (hipStream_t)stream, a);
    if (int e = otal_launch_status()) return e;
    hipLaunchKernelGGL(detection_loss_anet_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, losses, B);
    return otal_launch_status();
}

// ---- backward of the fused detection loss: the seven head gradients from the stored per-loss gradients and the incoming
// scalar gradients of the seven losses, in ONE launch (the autograd formulation is 9 multiplies and 2 adds = 11 launches).
//   grads layout (otal_detection_loss): dloc_l (2A) | dloc_ct (2A) | dpl_pl (2A) | dpl_ct (2A) | dconf (AC) | dpconf (AC) |
//                                        dcen (A) | dact (A) | dpact (A),  A = B K
//   g: device scalars {loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_act, loss_prop_act} (NULL = 0)
//   out layout: d_loc (2A) | d_prop_loc (2A) | d_conf (AC) | d_prop_conf (AC) | d_center (A) | d_act (A) | d_prop_act (A)
namespace {
struct LossG { const float* g[7]; };
__global__ __launch_bounds__(256) void detection_loss_bwd_kernel(const float* __restrict__ grads, LossG lg, float* __restrict__ out,
                                                                 int A, int C) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t A2 = 2 * (int64_t)A, AC = (int64_t)A * C;
    const int64_t total = 2 * A2 + 2 * AC + 3 * (int64_t)A;
    if (e >= total) return;
    auto gv = [&](int i) { return lg.g[i] ? *lg.g[i] : 0.f; };
    const float* dloc_l = grads, *dloc_ct = grads + A2, *dpl_pl = grads + 2 * A2, *dpl_ct = grads + 3 * A2;
    const float* dconf = grads + 4 * A2, *dpconf = dconf + AC, *dcen = dpconf + AC, *dact = dcen + A, *dpact = dact + A;
    float v;
    if (e < A2) v = dloc_l[e] * gv(0) + dloc_ct[e] * gv(4);
    else if (e < 2 * A2) v = dpl_pl[e - A2] * gv(2) + dpl_ct[e - A2] * gv(4);
    else if (e < 2 * A2 + AC) v = dconf[e - 2 * A2] * gv(1);
    else if (e < 2 * A2 + 2 * AC) v = dpconf[e - 2 * A2 - AC] * gv(3);
    else if (e < 2 * A2 + 2 * AC + A) v = dcen[e - 2 * A2 - 2 * AC] * gv(4);
    else if (e < 2 * A2 + 2 * AC + 2 * (int64_t)A) v = dact[e - 2 * A2 - 2 * AC - A] * gv(5);
    else v = dpact[e - 2 * A2 - 2 * AC - 2 * (int64_t)A] * gv(6);
    out[e] = v;
}
}  // namespace

extern "C" int otal_detection_loss_bwd(const float* grads, const float* const* g7, float* out, int B, int K, int C, void* stream) {
    if (!grads || !g7 || !out) return OTAL_E_NULL;
    if (B <= 0 || K <= 0 || C <= 0) return OTAL_E_SHAPE;
    LossG lg;
    for (int i = 0; i < 7; ++i) lg.g[i] = g7[i];
    const int A = B * K;
    const int64_t total = 4 * (int64_t)A + 2 * (int64_t)A * C + 3 * (int64_t)A;
    hipLaunchKernelGGL(detection_loss_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grads, lg,
                       out, A, C);
    return otal_launch_status();
}
