/* oracle/bmp_ref.c -- TEST INFRASTRUCTURE (CPU checker), never shipped or measured.
 *
 * Scalar C restatement of the reference's only native op, BoundaryMaxPooling:
 *   forward : AFSD/prop_pooling/boundary_max_pooling_kernel.cu:17-46
 *   backward: AFSD/prop_pooling/boundary_max_pooling_kernel.cu:48-82
 *   shapes / zero-init of outputs: AFSD/prop_pooling/boundary_max_pooling_cuda.cpp:26-31,43-47
 *
 * Semantics restated (not copied):
 *   - channel c uses the start window (seg[..,0:2]) when c <  C/2 and the end window
 *     (seg[..,2:4]) otherwise;
 *   - window bounds are float->int truncation, then clamped to [0, tscale-1];
 *   - the scan starts at l and only a strictly greater value replaces the running max,
 *     so ties keep the lowest index, NaN never replaces, and l > r yields in[l];
 *   - backward recomputes the argmax and adds grad_out into grad_in[argmax].  The reference
 *     does this with atomicAdd in undefined order; here contributions are added in ascending
 *     k, which is the order the HIP kernel uses too, so results are bit-comparable.
 *
 * `tscale_bwd`: the reference launcher passes grad_output.size(2) (= N) as tscale
 * (boundary_max_pooling_kernel.cu:121), which is only right when N == T.  Pass T for the
 * mathematically correct gradient, N to reproduce the reference's addressing bit for bit.
 *
 * Parity status: the reference ships no test for this op -> "parity unpinned" by the
 * reference; pinned by known-answer cases in tests/test_oracle_bmp.py.
 */
#include <stddef.h>
#include <math.h>

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void otal_oracle_bmp_fwd(const float* in, const float* seg, float* out,
                         int B, int C, int T, int N) {
    const int half = C / 2;
    for (int n = 0; n < B; ++n)
        for (int c = 0; c < C; ++c) {
            const float* row = in + ((size_t)n * C + c) * T;
            const int which = (c / half) * 2;
            for (int k = 0; k < N; ++k) {
                const float* s = seg + ((size_t)n * N + k) * 4 + which;
                int l = clampi((int)s[0], 0, T - 1);
                int r = clampi((int)s[1], 0, T - 1);
                float best = row[l];
                for (int i = l + 1; i <= r; ++i)
                    if (row[i] > best) best = row[i];
                out[((size_t)n * C + c) * N + k] = best;
            }
        }
}

/* grad_in must hold B*C*T floats and is zero-filled here. */
void otal_oracle_bmp_bwd(const float* gout, const float* in, const float* seg, float* gin,
                         int B, int C, int T, int N, int tscale_bwd) {
    const int half = C / 2;
    const int ts = tscale_bwd;
    for (size_t i = 0; i < (size_t)B * C * T; ++i) gin[i] = 0.f;
    for (int n = 0; n < B; ++n)
        for (int c = 0; c < C; ++c) {
            const size_t rowoff = ((size_t)n * C + c) * ts;
            const int which = (c / half) * 2;
            for (int k = 0; k < N; ++k) {
                const float* s = seg + ((size_t)n * N + k) * 4 + which;
                int l = clampi((int)s[0], 0, ts - 1);
                int r = clampi((int)s[1], 0, ts - 1);
                float best = in[rowoff + l];
                int arg = l;
                for (int i = l + 1; i <= r; ++i)
                    if (in[rowoff + i] > best) { best = in[rowoff + i]; arg = i; }
                gin[rowoff + arg] += gout[((size_t)n * C + c) * N + k];
            }
        }
}

/* Gaussian Soft-NMS, restating AFSD/common/segment_utils.py:128-162 (softnms_v2).
 * seg: (n,3) rows [start,end,score] (score column is decayed in place, like the reference);
 * done: n bytes out (1 = kept).  Returns the kept count.  Quirks kept on purpose:
 * the loop stops when <= 1 candidate is left, so the last survivor is never kept; the argmax
 * takes the first maximum; the kept rows are reported in original index order. */
int otal_oracle_softnms(float* seg, int n, float sigma, int top_k, float thr, unsigned char* done) {
    int ndone = 0, nundone = 0;
    /* undone mask lives in `done` as value 2 */
    for (int i = 0; i < n; ++i) {
        done[i] = (seg[3 * i + 2] >= thr) ? 2 : 0;
        nundone += done[i] == 2;
    }
    while (nundone > 1 && ndone < top_k) {
        int idx = -1;
        float best = 0.f;
        for (int i = 0; i < n; ++i)
            if (done[i] == 2 && (idx < 0 || seg[3 * i + 2] > best)) { idx = i; best = seg[3 * i + 2]; }
        done[idx] = 1; ++ndone; --nundone;
        const float ts = seg[3 * idx], te = seg[3 * idx + 1];
        float width = te - ts;
        if (width < 1e-5f) width = 1e-5f;
        for (int i = 0; i < n; ++i) {
            if (done[i] != 2) continue;
            float a = seg[3 * i], b = seg[3 * i + 1];
            float tt1 = a < ts ? ts : a;
            float tt2 = b > te ? te : b;
            float inter = tt2 - tt1;
            if (inter < 0.f) inter = 0.f;
            float iou = inter / (width + (b - a) - inter);
            seg[3 * i + 2] *= expf(-(iou * iou) / sigma);
            if (seg[3 * i + 2] < thr) { done[i] = 0; --nundone; }
        }
    }
    for (int i = 0; i < n; ++i) done[i] = done[i] == 1;
    return ndone;
}
