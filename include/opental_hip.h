/* include/opental_hip.h -- C ABI of libopental_hip.so (MI355X / gfx950, HIP).
 *
 * Drop-in boundary for the OpenTAL/AFSD detection hot path.  Every entry point takes plain
 * device pointers + sizes + a hipStream_t (passed as void*), launches asynchronously on that
 * stream, never synchronises, and returns
 *     0            success
 *    <0            argument error (OTAL_E_*), nothing was launched
 *    >0            a hipError_t from the launch
 * State: a call owns nothing persistent except three process-wide, mutex-guarded registries that exist for launch economy --
 * the named option switches (otal_set_option), the event ring of otal_stream_wait, and the record of DEFERRED split-K
 * reductions (otal_conv_defer_reduces .. otal_conv_flush_reduces).  The first two are safe from any number of host threads.
 * The deferred record is ONE list per process: concurrent calls cannot corrupt it, but "defer, launch weight gradients, flush"
 * is a protocol of one logical issuer at a time (a training step); leave it off (the default) to get launches that touch no
 * shared state at all.  Everything else -- workspaces, prologue regions, sign bits, winner bytes -- is caller-owned memory.
 * No exceptions cross this boundary; no torch types appear in it.  All tensors are contiguous,
 * channel-major, exactly as the reference lays them out: features (B,C,T) / (B,C,T,H,W),
 * proposals (B,N,4).
 *
 * Reference interfaces replaced (Cogito2012/OpenTAL):
 *   otal_bmp_fwd / otal_bmp_bwd    <- pybind module boundary_max_pooling_cuda.forward/backward,
 *                                     AFSD/prop_pooling/boundary_max_pooling_cuda.cpp:21-55,
 *                                     kernels AFSD/prop_pooling/boundary_max_pooling_kernel.cu:17-145
 *   (later sections of this header cite theirs next to each declaration)
 */
#ifndef OPENTAL_HIP_H
#define OPENTAL_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTAL_ABI_VERSION 24

/* argument errors */
#define OTAL_E_NULL      (-1)  /* null pointer */
#define OTAL_E_SHAPE     (-2)  /* non-positive / inconsistent size */
#define OTAL_E_ODD_C     (-3)  /* pooling needs an even channel count (two halves) */
#define OTAL_E_BATCH     (-4)  /* segments batch != feature batch (reference reads out of bounds) */
#define OTAL_E_DTYPE     (-5)  /* unsupported dtype code */
#define OTAL_E_LEVELS    (-6)  /* bad level table */
#define OTAL_E_UNSUPPORTED (-7)

/* dtype codes for feature tensors; proposals/segments are always float32 */
#define OTAL_F32  0
#define OTAL_BF16 1
#define OTAL_F16  2   /* otal_bmp_*: the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF (boundary_max_pooling_kernel.cu:99,:131) */
#define OTAL_F64  3   /* otal_bmp_* only */

#define OTAL_MAX_LEVELS 8

int otal_abi_version(void);
const char* otal_error_string(int code);
/* Named integer switches that select kernel variants (A/B tests, micro-benchmarks; e.g. "OTAL_CONV_NO1A").  A switch
 * starts from the environment variable of the same name, read once at its first use -- the launch path itself never
 * calls getenv; otal_set_option changes it at run time, otal_get_option reads it (dflt when it was never set). */
int otal_set_option(const char* name, int value);
int otal_get_option(const char* name, int dflt);
/* Stream fork / join for callers that spread independent launches of this library over several HIP streams (the host
 * side runs weight gradients beside the data-gradient chain): everything given to `waiter` after the call runs behind
 * what `signaler` has been given so far.  One hipEventRecord + hipStreamWaitEvent on an event of an internal ring
 * (no timing); legal while the streams are being captured into a hipGraph (the waiter joins the capture). */
int otal_stream_wait(void* waiter, void* signaler);

/* ------------------------------------------------------------------ BoundaryMaxPooling ----
 * out[n,c,k] = max_{i in [l,r]} in[n,c,i];  (l,r) = clamp(trunc(seg[n,k,2*(c>=C/2)+{0,1}]), 0, T-1);
 * strict '>' scan from l (ties keep the lowest index, NaN never replaces, l>r gives in[l]).
 * Replaces boundary_max_pooling_cuda.forward (boundary_max_pooling_cuda.cpp:21-34).
 * seg_batch must equal B. */
int otal_bmp_fwd(const void* in, const float* seg, void* out,
                 int B, int C, int T, int N, int seg_batch, int dtype, void* stream);

/* grad_in[n,c,argmax] += grad_out[n,c,k], contributions added in ascending k (deterministic; the
 * reference uses atomicAdd in undefined order).  grad_in is fully written (zeros included).
 * compat_ref_stride != 0 reproduces the reference launcher, which takes tscale from
 * grad_output.size(2) (= N) instead of T (boundary_max_pooling_kernel.cu:121): rows are then
 * addressed and clamped with stride N inside the same (B,C,T) buffers.
 * Replaces boundary_max_pooling_cuda.backward (boundary_max_pooling_cuda.cpp:36-50). */
int otal_bmp_bwd(const void* grad_out, const void* in, const float* seg, void* grad_in,
                 int B, int C, int T, int N, int seg_batch, int compat_ref_stride,
                 int dtype, void* stream);

/* Level-batched form: ONE launch pools every pyramid level.  `in` is (B,C,Ttot) holding the
 * levels side by side along T (level l = columns [t_start[l], t_start[l+1])), `seg` is
 * (B,Ntot,4) with each level's proposals in level-local coordinates (columns
 * [n_start[l], n_start[l+1])), out is (B,C,Ntot).  nlev <= OTAL_MAX_LEVELS; both tables have
 * nlev+1 entries.  With nlev == 1 this is otal_bmp_fwd.  Replaces the 6 per-level calls of
 * ProposalBranch.forward inside the level loop (AFSD/thumos14/BDNet.py:333,:386-389). */
int otal_bmp_fwd_levels(const void* in, const float* seg, void* out,
                        int B, int C, int nlev, const int* t_start, const int* n_start,
                        int dtype, void* stream);
int otal_bmp_bwd_levels(const void* grad_out, const void* in, const float* seg, void* grad_in,
                        int B, int C, int nlev, const int* t_start, const int* n_start,
                        int dtype, void* stream);

/* ------------------------------------------------------- implicit-GEMM convolution (MFMA) ----
 * One kernel family for Conv1d (H=W=1) and Conv3d, fp32 in / fp32 accumulate on
 * v_mfma_f32_32x32x2_f32.  Replaces, on the hot path, torch.nn.Conv1d inside Unit1D
 * (AFSD/common/layers.py:178-214), torch.nn.Conv3d inside Unit3D (AFSD/common/i3d_backbone.py:7-87,
 * AFSD/common/layers.py:106-175) and their autograd backward (cuDNN in the reference).
 *
 * geom   : 28 ints  B,Cin,Cout, Ti,Hi,Wi, To,Ho,Wo, kt,kh,kw, st,sh,sw, pt,ph,pw, nlev, lev[0..8]
 *          (pt/ph/pw = FRONT pad of the reference's SAME rule; out-of-range taps read zero;
 *          nlev > 1: level-packed stride-1 Conv1d, taps do not cross level boundaries)
 * strides: 4 int64  x batch stride, x channel stride, y batch stride, y channel stride (elements);
 *          pointers are pre-offset to the first channel, so channel slices of a concat buffer
 *          are read / written in place.
 * ws     : optional split-K workspace (otal_conv_workspace_bytes); a smaller one only lowers the
 *          split factor.  Split-K partials are reduced in a fixed order (deterministic).
 * mode   : 0 forward, 1 data gradient, 2 weight gradient.
 * precision: 0 = fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32, the parity path);
 *            1 = operands rounded to bf16 when staged into LDS, v_mfma_f32_32x32x16_bf16, fp32
 *                accumulation and fp32 tensors in HBM (the throughput path).
 *            bit 1 (value 2), otal_conv_dgrad only: `wt_packed` points at the FORWARD-layout weight
 *                (Cout, Cin, kvol); the launch re-orders it itself (no otal_conv_pack_wt call needed).
 *            bit 2 (value 4), with bit 0: the LARGE activation operand of the call is STORED as bf16 in HBM
 *                (otal_conv_fwd: y; otal_conv_wgrad / otal_conv_dgrad: dy) -- the pointer is passed through the float*
 *                parameter, strides stay in ELEMENTS.  Only geometries for which otal_conv_half_storage() returns 1
 *                (16-byte aligned pointer, y strides multiples of 8); anything else is OTAL_E_UNSUPPORTED.  The values
 *                are the ones the consumer's bf16 operand rounding produces from the fp32 tensor, so the forward
 *                results do not change; the backbone uses it for Conv3d_1a's output and its gradient (604 MB each).
 *            bit 3 (value 8), with bits 0 and 2 (ABI 23): the tensor on the layer's INPUT side is stored as bf16 as well
 *                (otal_conv_fwd: x; otal_conv_wgrad: x; otal_conv_dgrad: dx) -- every activation and data gradient between two
 *                backbone layers is then a bf16 tensor (half the bytes; operands, accumulators and results are those of the
 *                fp32-tensor kernels fed with the same bf16 values, outputs rounded to nearest even once).  No accumulate.
 *            bit 4 (value 16), otal_conv_dgrad with bit 3: out_mask is a bf16 tensor (the activation itself) -- required
 *                whenever out_mask is given together with bit 3.
 *                otal_conv_half_storage(geom, strides, mode, precision incl. bit 3) says whether a kernel exists
 *                (16-byte aligned pointers; batch / channel strides and positions per sample multiples of 8). */
int otal_conv_half_storage(const int* geom, const int64_t* strides, int mode, int precision);
size_t otal_conv_workspace_bytes(const int* geom, int mode);

/* y = act(scale[co] * conv(x, w) + shift[co]); scale/shift nullable (frozen BN folded, or bias). */
int otal_conv_fwd(const int* geom, const int64_t* strides, const float* x, const float* w,
                  const float* scale, const float* shift, float* y, int relu, int precision,
                  const void* prologue, void* ws, size_t ws_bytes, void* stream);

/* dx (+)= m * conv_transpose(dy, w); when out_mask/out_scale are given,
 * m = (out_mask[dx offset] > 0) * out_scale[ci]: the ReLU + frozen-BN backward of the layer that
 * PRODUCED x, folded into the store (out_mask has dx's layout), so the gradient handed to that layer
 * is already the gradient w.r.t. its convolution output.  wt_packed = otal_conv_pack_wt(w). */
int otal_conv_dgrad(const int* geom, const int64_t* strides, const float* dy, const float* wt_packed,
                    float* dx, int accumulate, const float* out_mask, const float* out_scale,
                    int precision, const void* prologue, void* ws, size_t ws_bytes, void* stream);

/* dw (+)= sum_{b,pos} dy[b,co,pos] * x[b,ci,pos*s + tap - pad] */
int otal_conv_wgrad(const int* geom, const int64_t* strides, const float* x, const float* dy,
                    float* dw, int accumulate, int precision, const void* prologue, void* ws, size_t ws_bytes,
                    void* stream);

/* PAIR launches (ABI 20): TWO problems of the same geometry, strides and options in ONE launch -- arrays of two pointers
 * each, results as of two separate calls (bit-identical: a workgroup never sees the other problem).  For the 1-D
 * temporal layers the model applies as siblings -- the loc / conf towers and the two ProposalBranches
 * (AFSD/thumos14/BDNet.py:64-113,:333-412 run them one after the other) -- whose launches sit at about twice the launch
 * floor: two of them in one grid cost what one does.  Only the geometries of the one-launch 1-D kernels (stride 1,
 * k = 1 / 3, H = W = 1, channels % 128 == 0, precision bit 0 set); anything else returns OTAL_E_UNSUPPORTED and the
 * caller issues the two launches itself.  otal_conv_dgrad_pair takes forward-layout weights (precision bit 1 implied). */
int otal_conv_fwd_pair(const int* geom, const int64_t* strides, const float* const* x, const float* const* w,
                       const float* const* scale, const float* const* shift, float* const* y, int relu, int precision,
                       const void* const* prologue, void* ws, size_t ws_bytes, void* stream);
int otal_conv_dgrad_pair(const int* geom, const int64_t* strides, const float* const* dy, const float* const* w,
                         float* const* dx, int precision, const void* const* prologue, void* ws, size_t ws_bytes, void* stream);
int otal_conv_wgrad_pair(const int* geom, const int64_t* strides, const float* const* x, const float* const* dy,
                         float* const* dw, int precision, void* ws, size_t ws_bytes, void* stream);

/* Persistent prologues.  A bf16 launch first builds its tables and (fwd / dgrad) re-packs the weights to bf16; by default
 * that happens inside every launch, in the workspace.  A caller that runs the same layers repeatedly (a training loop) can
 * own one region per (layer, mode) instead and pass it as `prologue` (NULL = build in the workspace):
 *   otal_conv_prologue_bytes : region size, 0 when this launch has no reusable prologue (generic kernel)
 *   otal_conv_prologue       : fill the region now; for fwd / dgrad also emit the descriptor (host memory,
 *                              otal_conv_prologue_desc_bytes) and return the workgroups it needs; wgrad regions hold a
 *                              geometry-only table and never need refreshing
 *   otal_conv_prologue_batch : after the weights changed, refresh n fwd / dgrad regions in ONE launch from the
 *                              descriptors copied to DEVICE memory; device_starts[n+1] = prefix sums of the workgroup
 *                              counts otal_conv_prologue returned, total_blocks = their sum
 * The regions must be refreshed before the first launch that follows a weight update. */
size_t otal_conv_prologue_bytes(const int* geom, const int64_t* strides, int mode, int precision);
size_t otal_conv_prologue_desc_bytes(void);
int otal_conv_prologue(const int* geom, const int64_t* strides, int mode, const float* w, int precision, void* region,
                       size_t region_bytes, void* host_desc, void* stream);
int otal_conv_prologue_batch(int n, const void* device_descs, const int* device_starts, int total_blocks, void* stream);

/* Deferred weight-gradient reductions.  Nobody reads a weight gradient before the optimizer / the gradient all-reduce, so
 * its split-K reduce need not follow its GEMM: after otal_conv_defer_reduces(1), otal_conv_wgrad launches whose reduction is
 * the plain fixed-order slab sum leave their slabs in the workspace and record the reduction instead of launching it;
 * otal_conv_deferred_end() != 0 then is the address one past those slabs -- the caller must hand every later launch workspace
 * BEHIND it until otal_conv_flush_reduces(stream) has run all recorded reductions (up to 24; more flush by themselves) as
 * ONE launch.  Launches with unaligned slabs reduce at once as before (deferred_end() == 0); the direct 3x3x3 kernels, whose
 * slabs needed a transposing reduce until round 6, now write slabs in dW's own layout and are recorded like the others.  otal_conv_defer_reduces(0) needs an empty record.  The summation order is the one of the
 * immediate reduce for >= 16 slabs (quarter sums in slab order, (q0+q1)+(q2+q3)); deterministic.  Process-wide record behind a
 * mutex: one logical issuer at a time (see the preamble).  Replaces nothing in the reference (autograd of nn.Conv1d / nn.Conv3d, i3d_backbone.py:33-43,
 * layers.py:187-192, has no split-K); it removes ~40 of this library's own launches per training step. */
int otal_conv_defer_reduces(int on);
size_t otal_conv_deferred_end(void);
int otal_conv_deferred_count(void);
int otal_conv_flush_reduces(void* stream);

/* (Cout,Cin,kvol) -> (Cin,Cout,kvol): the A operand of the data-gradient GEMM. */
int otal_conv_pack_wt(const float* w, float* wt, int Cout, int Cin, int kvol, void* stream);

/* ------------------------------------------------------------------ GroupNorm + ReLU ----
 * y = relu(GroupNorm_G(x) * gamma + beta) on (B,C,T); statistics per (sample, group, level)
 * (nlev <= 1 or lev == NULL: one level).  stats: (B,G,nlev,2) {mean, rstd}, kept for backward.
 * Replaces nn.GroupNorm(32, C) + nn.ReLU after each Unit1D / Unit3D of the pyramid
 * (AFSD/thumos14/BDNet.py:67-103,:129-203,:274-284). */
int otal_gn_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                     int B, int C, int T, int G, float eps, int relu, int nlev, const int* lev, void* stream);
/* dx and per-(sample,channel) partial sums partial[(b*3 + {0,1,2})*C + c] = {d_gamma, d_beta, sum_t dx}
 * (sum over b by the caller, e.g. otal_sum_partials; sum_t dx is the gradient of the preceding convolution's bias).
 * dy_batch_stride (elements; 0 = C*T): dy may be a channel slice of a wider (B, Ctot, T) map -- the gradient of a
 * torch.cat along the channels (BDNet.py:111) is read in place instead of through a contiguous copy. */
int otal_gn_relu_bwd(const float* dy, int64_t dy_batch_stride, const float* x, const float* gamma, const float* beta,
                     const float* stats, float* dx, float* partial, int B, int C, int T, int G,
                     int relu, int nlev, const int* lev, void* stream);
/* The same for TWO maps of one shape in one launch (ABI 20; see the convolution pair launches). */
int otal_gn_relu_fwd_pair(const float* const* x, const float* const* gamma, const float* const* beta, float* const* y,
                          float* const* stats, int B, int C, int T, int G, float eps, int relu, int nlev, const int* lev,
                          void* stream);
int otal_gn_relu_bwd_pair(const float* const* dy, const int64_t* dy_batch_stride, const float* const* x,
                          const float* const* gamma, const float* const* beta, const float* const* stats, float* const* dx,
                          float* const* partial, int B, int C, int T, int G, int relu, int nlev, const int* lev, void* stream);
/* Batch sums of MANY layers' partials in one launch: for item i, dst{0,1,2}[i][c] = sum_b partial[i][(b*3 + r)*C + c]
 * (r = 0,1,2; ascending b; a NULL dst row is skipped).  Replaces the per-layer torch sum over the batch that follows
 * every GroupNorm backward (21 per step); the destinations may be slices of a flat gradient arena. */
int otal_sum_partials(int n_items, const float* const* partial, float* const* dst0, float* const* dst1,
                      float* const* dst2, const int* channels, const int* batches, void* stream);

/* ------------------------------------------------------------------ MaxPool3dSamePadding ----
 * geom: 17 ints B,C, Ti,Hi,Wi, To,Ho,Wo, kt,kh,kw, st,sh,sw, pt,ph,pw (front pads; ZERO padding);
 * strides as for the convolution.  argtap: (B,C,To,Ho,Wo) uint8, OPAQUE to the caller: written by _fwd, consumed by
 * _bwd with the same geometry.  (Content: the winner's tap index (dt*kh+dh)*kw+dw, 255 = a padded zero won; for the
 * 3x3x3 / stride 1 / pad 1 pools on 12x12, 6x6 and 3x3 planes the maximum is taken separably and the byte of position p
 * holds three 2-bit axis taps: w stage bits 1:0, h stage 3:2, t stage 5:4 -- see csrc/pool3d.hip.)  In every case the
 * winner is the FIRST maximum in (dt,dh,dw) scan order with the zero padding taking part.
 * Replaces MaxPool3dSamePadding.forward (AFSD/common/layers.py:9-35) and its autograd backward. */
int otal_maxpool3d_fwd(const int* geom, const int64_t* strides, const float* x, float* y,
                       unsigned char* argtap, void* stream);
/* out_mask/out_scale (nullable pair): dx contribution *= (out_mask[dx offset] > 0) * out_scale[c]. */
int otal_maxpool3d_bwd(const int* geom, const int64_t* strides, const float* dy,
                       const unsigned char* argtap, float* dx, int accumulate,
                       const float* out_mask, const float* out_scale, void* stream);
/* The same pair with the producer's ReLU mask carried as SIGN BITS of the pool input instead of re-reading the fp32
 * activations in the backward pass (604 MB -> 19 MB for MaxPool3d_2a): _fwd_signbits also writes signbits
 * (otal_maxpool3d_signbits_bytes(), opaque), _bwd_signbits applies (bit set) * out_scale[c].  Strided 3x3 pools
 * ((1,3,3)/(1,2,2), (3,3,3)/(2,2,2)) only: _signbits_bytes returns 0 elsewhere and the calls OTAL_E_UNSUPPORTED. */
size_t otal_maxpool3d_signbits_bytes(const int* geom, const int64_t* strides);
int otal_maxpool3d_fwd_signbits(const int* geom, const int64_t* strides, const float* x, float* y,
                                unsigned char* argtap, unsigned char* signbits, void* stream);
int otal_maxpool3d_bwd_signbits(const int* geom, const int64_t* strides, const float* dy, const unsigned char* argtap,
                                float* dx, int accumulate, const unsigned char* signbits, const float* out_scale,
                                void* stream);
/* The same pair with the LARGE tensor stored as bf16: _fwd_signbits_h reads a bf16 pool input (written by otal_conv_fwd
 * with precision bit 2), _bwd_signbits_h writes dx as bf16 (round to nearest even; no accumulate) for otal_conv_wgrad
 * with precision bit 2.  (1,3,3)/(1,2,2) pools only. */
int otal_maxpool3d_fwd_signbits_h(const int* geom, const int64_t* strides, const void* x_bf16, float* y,
                                  unsigned char* argtap, unsigned char* signbits, void* stream);
int otal_maxpool3d_bwd_signbits_h(const int* geom, const int64_t* strides, const float* dy, const unsigned char* argtap,
                                  void* dx_bf16, const unsigned char* signbits, const float* out_scale, void* stream);
/* The general bf16-storage forms (ABI 23).  `io` says which tensors are STORED as bf16 (strides stay in elements).
 *   _fwd_io  io: bit 0 x, bit 1 y.  1 = _fwd_signbits_h; 3 = both: the strided 3x3 pools ((1,3,3)/(1,2,2), (3,3,3)/(2,2,2)) and
 *            the 3x3x3 / stride-1 branch pools on 12 x 12 and 6 x 6 planes.  A max-pool commutes with the monotonic bf16
 *            rounding and its winners are copied, not rounded: the output equals the fp32 pool's output rounded once.
 *            bit 2 (value 4, with 3; round 5): the caller guarantees x >= +0 everywhere -- the output of a conv + ReLU, which
 *            is what MaxPool3d_2a / 3a / 4a read (AFSD/common/i3d_backbone.py:194-244) -- and the (1,3,3)/(1,2,2) pools then
 *            run on ordered integer keys (the bf16 bit patterns themselves; v_max3_u32 over bits << 16 | tap priority):
 *            same winners, values and sign bits on such inputs; a -0.0 is read as +0.0, a NaN is not ordered.
 *   _bwd_io  io: bit 0 dx, bit 1 dy, bit 2 out_mask.  1 = _bwd_signbits_h; 3 / 7 = all of them: strided pools with the sign-bit
 *            mask and a plain store; branch pools with a bf16 out_mask tensor, `accumulate` = read the bf16 dx, add this
 *            pool's contribution in fp32, round to nearest even once (the second producer of an Inception module's input
 *            gradient, AFSD/common/i3d_backbone.py:116-121 under autograd).
 * signbits / out_mask / out_scale nullable as in the fp32 entry points; anything else is OTAL_E_UNSUPPORTED. */
int otal_maxpool3d_fwd_io(const int* geom, const int64_t* strides, const void* x, void* y, unsigned char* argtap,
                          unsigned char* signbits, int io, void* stream);
int otal_maxpool3d_bwd_io(const int* geom, const int64_t* strides, const void* dy, const unsigned char* argtap, void* dx,
                          int accumulate, const void* out_mask, const float* out_scale, const unsigned char* signbits, int io,
                          void* stream);
/* fp32 <-> bf16 storage conversion of a (B, C, P) map (dense positions, batch / channel strides in elements: channel slices of
 * a concat buffer): the boundary between the backbone's bf16-stored tensors and the fp32 tensors around them (the endpoints
 * handed to the pyramid, AFSD/thumos14/BDNet.py:307-308, and their gradients).  to_bf16 != 0: fp32 -> bf16, round to nearest
 * even; 0: bf16 -> fp32, exact.  P, the strides: multiples of 8; pointers 16-byte aligned. */
int otal_convert_storage(const void* src, int64_t src_bs, int64_t src_cs, void* dst, int64_t dst_bs, int64_t dst_cs,
                         int to_bf16, int B, int C, int P, void* stream);

/* ------------------------------------------------------------------ head output tails ----
 * Everything between the head convolutions and CoarsePyramid's outputs (AFSD/thumos14/BDNet.py:337-353,:399-412,:538-556;
 * AFSD/anet/BDNet.py:307-320,:366-376) for n_items <= 8 maps at once: raw[i] (B, channels[i], N) -> out[i] (B, N, channels[i]),
 *   modes[i] 0: the permute(0,2,1).contiguous() only;
 *            1: ScaleExp per pyramid level, exp(scales[l] * x) * level_strides[l] (level_strides null = 1: THUMOS14);
 *            2: permute + DirichletLayer.compute_uncertainty with evidence 'exp' into unct[i] (B, N).
 * lev[0..nlev]: anchor ranges of the levels along N.  _bwd: draw[i] (B, C, N) from dout[i] (B, N, C) and dunct[i] (B, N)
 * (either may be null = zero) and dscales[nlev] (deterministic single-workgroup reduction). */
int otal_head_outputs_fwd(int n_items, const int* channels, const int* modes, const float* const* raw, float* const* out,
                          float* const* unct, const float* scales, int B, int N, int nlev, const int* lev,
                          const float* level_strides, void* stream);
int otal_head_outputs_bwd(int n_items, const int* channels, const int* modes, const float* const* raw,
                          const float* const* out, const float* const* unct, const float* const* dout,
                          const float* const* dunct, float* const* draw, const float* scales, float* dscales, int B,
                          int N, int nlev, const int* lev, const float* level_strides, void* stream);

/* ------------------------------------------------------------------ boundary (start / end) losses ----
 * calc_bce_loss (AFSD/thumos14/train.py:152-161; anet/train.py:136-144) on both channel halves of a boundary feature
 * map x (B, C, T) read in place through its batch / channel strides (time stride 1): half h in {0, 1},
 * m = mean over the half's channels of tanh(x), loss_h = mean over (b, t) of BCE(m, mask[b, mask_row0 + h, t * mask_step]).
 * terms (B, 2, T): the per-(b, t) BCE values (loss_h = sum / (B T)); dx (B, C, T) contiguous: d loss_h / d x for the
 * channels of half h (scale each half by the incoming gradient of its loss). */
int otal_boundary_bce(const float* x, int64_t x_batch_stride, int64_t x_channel_stride, const float* mask,
                      int64_t mask_batch_stride, int64_t mask_row_stride, int mask_row0, int mask_step, float* terms,
                      float* dx, int B, int C, int T, void* stream);

/* The tails of the n (<= 4) boundary losses of a step (train.py:193-201: the frame-level map with weight 1 and the two
 * level-0 proposal maps with weight 0.1): out2[h] = sum_i weights[i] * mean_{b,t} terms_i[b][h][t] in one launch, and
 * the matching backward out_i[b][c][t] = dx_i[b][c][t] * weights[i] * (c < C_i / 2 ? *g_start : *g_end) for all maps in
 * one launch (dx_i: as written by otal_boundary_bce; g_start / g_end: device scalars, NULL = 0). */
int otal_boundary_finish(int n, const float* const* terms, const float* weights, const int* T, int B, float* out2, void* stream);
int otal_boundary_scale(int n, const float* const* dx, float* const* out, const float* weights, const int* C, const int* T,
                        int B, const float* g_start, const float* g_end, void* stream);

/* ------------------------------------------------------------------ proposal window indices ----
 * loc (B,Ntot,2) -> level-space windows seg (B,Ntot,4) and frame-space windows frame_seg (B,Ntot,4)
 * for all levels at once; bit-exact restatement of the no_grad block of CoarsePyramid.forward
 * (AFSD/thumos14/BDNet.py:355-384).  lev: nlev+1 column starts of the levels inside Ntot. */
int otal_proposal_windows(const float* loc, float* seg, float* frame_seg, int B, int nlev,
                          const int* lev, float frame_num, void* stream);

/* ------------------------------------------------------------------ inference post-processing ----
 * otal_decode_clips: parse_output + decode_predictions + the threshold test of filtering
 * (AFSD/thumos14/test.py:79-162) for `nclips` clips in one launch.  Inputs are the model outputs
 * (nclips,A,.) as the reference lays them out; offsets/fps: per-clip frame offset and sampling rate.
 * Outputs: seg (nclips,A,2) seconds, score (nclips,K,A), unct/actn (nclips,A), flag (nclips,K,A) uint8
 * = score > conf_thresh && actionness > 0.5. */
int otal_decode_clips(const float* loc, const float* prop_loc, const float* priors, const float* conf,
                      const float* prop_conf, const float* center, const float* act, const float* prop_act,
                      const float* offsets, const float* fps, float* seg, float* score, float* unct,
                      float* actn, unsigned char* flag, int nclips, int A, int K, float clip_length,
                      float conf_thresh, void* stream);
/* otal_softnms_classes: for every (video, class) gather the flagged candidates of the video's clips
 * [clip_start[v], clip_start[v+1]) in clip-major / anchor-minor order and run softnms_v2
 * (AFSD/common/segment_utils.py:128-162; get_video_detections, test.py:165-200).
 * out: (nvideos*K, top_k, out_cols) rows [start,end,decayed score,unct,actionness] in original index
 * order; counts: (nvideos*K); out_index (nullable): (nvideos*K, top_k) source row = clip*A + anchor,
 * relative to the video's first clip.  max_clips bounds the clips of one video. */
int otal_softnms_classes(const float* seg, const float* score, const float* unct, const float* actn,
                         const unsigned char* flag, const int* clip_start, int nvideos, int max_clips,
                         int A, int K, float sigma, int top_k, float score_threshold, float* out,
                         int* counts, int* out_index, int out_cols, void* stream);
/* A (video, class) problem keeps its candidates in LDS (<= ~7600 rows = 60 THUMOS14 windows).  The reference's host
 * loop has no such limit (test.py:165-200 handles a video of any length), so longer videos run from a caller-owned
 * global scratch: otal_softnms_scratch_bytes returns its size for a batch with `total_clips` clips in all and at most
 * `max_clips` per video (0 when every video fits LDS); otal_softnms_classes_ws = otal_softnms_classes + that scratch.
 * otal_softnms_classes itself returns OTAL_E_UNSUPPORTED when a scratch would be needed. */
size_t otal_softnms_scratch_bytes(int total_clips, int max_clips, int A, int K);
int otal_softnms_classes_ws(const float* seg, const float* score, const float* unct, const float* actn,
                            const unsigned char* flag, const int* clip_start, int nvideos, int max_clips,
                            int A, int K, float sigma, int top_k, float score_threshold, float* out,
                            int* counts, int* out_index, int out_cols, void* scratch, size_t scratch_bytes,
                            int total_clips, void* stream);

/* ------------------------------------------------------------------ detection-head convolutions ----
 * The skinny Unit1D heads of one CoarsePyramid stage (AFSD/thumos14/BDNet.py:205-272, :337-353, :399-412;
 * AFSD/common/layers.py:178-214: SAME pad + nn.Conv1d(512, cout, k) + bias, cout = 1 / 2 / num_classes, k = 1 / 3) as
 * fp32 FMA kernels over LDS-staged (B, C, N) windows: ONE launch for the forward of all heads of a stage, TWO for their
 * backward (data gradients summed per input map; weight + bias gradients without split-K).
 *   head h reads input map x[in_idx[h]] (B, C, N), has weights w[h] (cout[h], C, ksize[h]) and bias[h] (cout[h]) or NULL,
 *   writes y[h] (B, cout[h], N).  lev (nlev + 1 column starts, or nlev <= 1): taps never cross a level boundary.
 *   backward: dy[h] (B, cout[h], N) or NULL (= zeros); dx[j] (B, C, N) or NULL (not needed) receives the SUM over the
 *   heads on input j; dw[h] like w[h]; db[h] (cout[h]) or NULL.  Plain stores, nothing is accumulated into.
 * Limits: C % 16 == 0, at most 8 heads on 4 inputs, at most 21 output channels per input, ksize 1 or 3;
 * otal_head_convs_supported() != 0 says the launches fit (callers fall back to otal_conv_* otherwise). */
int otal_head_convs_supported(int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize, int B, int C, int N);
int otal_head_convs_fwd(int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize,
                        const float* const* x, const float* const* w, const float* const* bias, float* const* y, int B, int C,
                        int N, int nlev, const int* lev, void* stream);
int otal_head_convs_bwd(int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize,
                        const float* const* x, const float* const* w, const float* const* dy, float* const* dx,
                        float* const* dw, float* const* db, int B, int C, int N, int nlev, const int* lev, void* stream);
/* The same with the two launches selectable: parts bit 0 = data gradients, bit 1 = weight / bias gradients (independent:
 * a caller may put them on different streams). */
int otal_head_convs_bwd_parts(int n_heads, int n_inputs, const int* in_idx, const int* cout, const int* ksize,
                        const float* const* x, const float* const* w, const float* const* dy, float* const* dx,
                        float* const* dw, float* const* db, int B, int C, int N, int nlev, const int* lev, int parts, void* stream);

/* ------------------------------------------------------------------ gradient hand-over to the backbone ----
 * dst[b][c][t][s] (+)= (z[b][c][t][s] > 0 ? scale[c] : 0) * src[b][c][t][s]   (scale NULL: 1; accumulate != 0: +=).
 * Every tensor has its own element strides {batch, channel, frame} and unit stride along s (the H*W plane), so src may be
 * the swapped-role projection gradient laid out [(b, t)][c][s] and dst / z channel slices of larger buffers.
 * Replaces what autograd runs between the pyramid projections (AFSD/thumos14/BDNet.py:129-155, :310-319) and the last
 * Inception modules (AFSD/common/i3d_backbone.py:33-43 F.relu + frozen BatchNorm3d backward): relu backward, the
 * BatchNorm scale and the contiguous copy of the permuted gradient -- three passes over the map in ATen. */
int otal_masked_scale_copy(const float* src, const int64_t* src_strides, const float* z, const int64_t* z_strides,
                           const float* scale, float* dst, const int64_t* dst_strides, int accumulate, int B, int C,
                           int T, int S, void* stream);

/* ------------------------------------------------------------------ optimizer ----
 * torch.optim.Adam with L2 weight decay (AFSD/thumos14/train.py:321-323) over one flat fp32
 * arena; g is multiplied by grad_scale first (1/world_size after a sum all-reduce). */
int otal_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* The same update with the two bias corrections {1 - beta1^t, sqrt(1 - beta2^t)} read from DEVICE memory
 * (written by the host before the launch): no launch argument changes between steps, so a training step can be
 * captured once in a HIP graph and replayed.  Replaces the same reference lines as otal_adam_flat. */
int otal_adam_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                       float beta2, float eps, float weight_decay, const float* bias_corr, float grad_scale,
                       void* stream);

/* MultiSegmentLoss + EvidenceLoss ('log', exp evidence, IBM re-weighting, IoU calibration) + ActionnessLoss (PU BCE,
 * rank weight 0) of the OpenTAL final recipe, forward AND backward in one single-workgroup launch.
 * Replaces AFSD/thumos14/multisegment_loss.py:92-259, cls_loss.py:120-129,:132-168,:212-278,:288-339 and their autograd
 * graph.  All tensors fp32 contiguous: loc/prop_loc (B,K,2), conf/prop_conf (B,K,C), center/act/prop_act (B,K),
 * priors (K), gt (B,G,3) = padded [start,end,label] with gvalid (B,G) bytes; weight_accum (num_bins) is the IBM EMA
 * state, updated in place (conf first, then prop_conf, as the reference does).
 * losses: 7 floats {loc, conf, prop_loc, prop_conf (+IoU calibration), center, act, prop_act}, normalised as the
 * reference's forward.  grads (otal_detection_loss_grad_floats): d loss_i / d input of the term that owns it, in the
 * order dloc[loc term], dloc[center term], dprop_loc[prop_loc term], dprop_loc[center term] (each (B,K,2)),
 * dconf, dprop_conf ((B,K,C)), dcenter, dact, dprop_act ((B,K)).  scratch: otal_detection_loss_scratch_floats.
 * cls_mode 1 = the as-shipped THUMOS14 dispatch (AFSD/thumos14/train.py:27-31 overwrites cls_loss_type 'edl' with
 * 'focal'): the two classification terms are FocalLoss_Ori(balance_index 0, alpha = focal_alpha, gamma 2,
 * size_average False) on the softmax scores of the positive rows (cls_loss.py:6-78); no IBM, no IoU calibration. */
size_t otal_detection_loss_scratch_floats(int B, int K);
size_t otal_detection_loss_grad_floats(int B, int K, int C);
int otal_detection_loss(const float* loc, const float* conf, const float* prop_loc, const float* prop_conf,
                        const float* center, const float* act, const float* prop_act, const float* priors,
                        const float* gt, const unsigned char* gvalid, float* weight_accum, int B, int K, int C, int G,
                        float clip_length, float overlap_thresh, int ibm_active, int num_bins, float momentum,
                        int iou_aware, int cls_mode, float focal_alpha, float* losses, float* grads, float* scratch,
                        void* stream);

/* The same for the ActivityNet1.3 recipe (ABI 22): AFSD/anet/multisegment_loss.py:87-301 with anet/cls_loss.py:78-246
 * (EvidenceLoss 'log', exp evidence, the closed-form influence-balanced weight 1 / (|z|_1 exp(ibm_coeff g) + 1e-10) whose
 * |z|_1 carries gradient, IoU calibration as each sample's mean) and :249-296 (ActionnessLoss with its rank hinge:
 * act_weight, act_margin).  Every term is evaluated per sample, normalised by that sample's counts and averaged over the
 * batch: one workgroup per sample + a seven-sum launch.  priors2 (K,2) = [centre, pyramid level]; level_bounds (nlev,2) =
 * the per-level (lower, upper] bounds on max(left, right) in frames (multisegment_loss.py:11); K <= 1024, nlev <= 8.
 * losses, grads: as otal_detection_loss (gradients already divided by B; otal_detection_loss_bwd applies);
 * scratch: 8 * B floats.  No state (this recipe's IBM weight has no EMA). */
int otal_detection_loss_anet(const float* loc, const float* conf, const float* prop_loc, const float* prop_conf,
                             const float* center, const float* act, const float* prop_act, const float* priors2,
                             const float* gt, const unsigned char* gvalid, int B, int K, int C, int G,
                             float clip_length, float overlap_thresh, const float* level_bounds, int nlev,
                             int ibm_active, float ibm_coeff, int iou_aware, float act_weight, float act_margin,
                             float* losses, float* grads, float* scratch, void* stream);

/* Backward of otal_detection_loss in one launch: the gradients w.r.t. the seven head outputs from the stored per-loss
 * gradients (`grads` as written by otal_detection_loss) and the incoming gradients of the seven losses g7[i] (device
 * scalars in the order loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_act, loss_prop_act; NULL = 0).
 * out: d_loc (2A) | d_prop_loc (2A) | d_conf (A C) | d_prop_conf (A C) | d_center (A) | d_act (A) | d_prop_act (A), A = B K.
 * Replaces the autograd of the weighted loss sum over multisegment_loss.py:92-259 (nine multiplies, two adds). */
int otal_detection_loss_bwd(const float* grads, const float* const* g7, float* out, int B, int K, int C, void* stream);

/* Clip preparation on the device: uint8 frames (T',Hs,Ws,3) -> normalised fp32 batch (B,3,T,Ho,Wo).
 * Replaces AFSD/common/thumos_dataset.py:136-137,:246-262 and videotransforms.py:44-124 (temporal zero padding,
 * random / centre crop, horizontal flip, float(), (x/255)*2-1, THWC -> CTHW).  params: device array of B records
 * {int64 frame0 (element offset of the clip's first frame in `frames`), int32 valid_t, crop_i, crop_j, flip}
 * = 24 bytes each (8-byte aligned); the random decisions are taken on the host exactly as the reference takes them.
 * flip bit 0 = mirror along W; flip bit 1 = the frames past valid_t are 127.5 before normalisation (exactly 0.0 after it),
 * the ActivityNet loader's padding (AFSD/common/anet_dataset.py:226-229), instead of 0 (-1.0 after it). */
int otal_prepare_clips(const unsigned char* frames, const void* params, float* out, int B, int T, int Hs, int Ws,
                       int Ho, int Wo, void* stream);
/* The same, plus the self-supervised branch's SPLICED clips from the same upload: out_ssl[b,:,t] = clip b's frame
 * frame_map[b*T + t] (frames >= valid_t, or a negative entry, are the zero padding).  Replaces THUMOS_Dataset.augment_
 * (AFSD/common/thumos_dataset.py:187-228: two time segments of the normalised clip change places), whose decisions stay
 * on the host.  `out` or `out_ssl` may be null (not both); frame_map is required with out_ssl. */
int otal_prepare_clips_map(const unsigned char* frames, const void* params, const int* frame_map, float* out,
                           float* out_ssl, int B, int T, int Hs, int Ws, int Ho, int Wo, void* stream);

/* Sliding windows of the inference path: B windows cut out of planar uint8 videos (C,Tv,H,W) resident on the device
 * -> normalised fp32 batch (B,C,T,H,W) in one pass.  Replaces AFSD/thumos14/test.py:67-76 prepare_clip per window
 * (float(), (x/255)*2-1 as torch evaluates it on the GPU: x times the fp32 reciprocal of 255; zero padding of a short
 * last window AFTER normalisation, unsqueeze) and the torch.cat of the
 * batch; test_cross_data.py:80-89 prepare_anet_clip gives the same values (127.5 padded before normalisation = 0.0).
 * params: device array of B records {uint64 src (device address of frame `offset`, channel 0, of the window's video),
 * int32 chan_stride4 (Tv*H*W/4), int32 valid_t (frames available from `offset`, <= T)} = 16 bytes each; H*W % 4 == 0
 * and 4-byte aligned videos (else OTAL_E_UNSUPPORTED). */
int otal_prepare_windows(const void* params, float* out, int B, int C, int T, int H, int W, void* stream);

/* ------------------------------------------------------------------ pyramid glue (ABI 24) ---- */

/* Pyramid merge (AFSD/thumos14/BDNet.py:310-326) in one launch: p0 (B,C,t0), p1 (B,C,t0/2) -> packed[:, :, :t0] = p0 +
 * nearest-upsampled p1 (source index t/2), packed[:, :, t0:t0+t0/2] = p1 (packed has T positions per row), and
 * frame (B,C,t0*up) = nearest-upsampled level 0 (source index t/up: F.interpolate(., [frame_num, 1])).  Backward: da, db
 * (B,C,T) gradients w.r.t. packed (first two levels read; db may be null), dframe (B,C,t0*up), dnext (B,C,t0/2; may be
 * null) the data gradient of the stride-2 layer reading level 1 -> dp0, dp1 (fixed summation order). */
int otal_pyramid_merge_fwd(const float* p0, const float* p1, float* packed, float* frame, int B, int C, int t0, int T,
                           int up, void* stream);
int otal_pyramid_merge_bwd(const float* da, const float* db, const float* dframe, const float* dnext, float* dp0, float* dp1,
                           int B, int C, int t0, int T, int up, void* stream);
/* otal_gn_relu_fwd with a strided destination: y[b*y_bs + c*y_cs + t] (a level slice of a packed buffer, a channel
 * slice of a concatenation buffer -- the torch.cat of AFSD/thumos14/BDNet.py:111 without the copy). */
int otal_gn_relu_fwd_to(const float* x, const float* gamma, const float* beta, float* y, int64_t y_bs, int64_t y_cs,
                        float* stats, int B, int C, int T, int G, float eps, int relu, int nlev, const int* lev, void* stream);
int otal_gn_relu_fwd_pair_to(const float* const* x, const float* const* gamma, const float* const* beta, float* const* y,
                             int64_t y_bs, int64_t y_cs, float* const* stats, int B, int C, int T, int G, float eps, int relu,
                             int nlev, const int* lev, void* stream);

/* otal_gn_relu_bwd whose output gradient is the SUM of n_terms (1..3) maps (B,C,dy_T[k] <= T) with their own batch / channel
 * strides (positions >= dy_T[k] of term k add nothing; summation order = term order): the gradient adds autograd issues for a
 * tensor with several consumers, done while the map is staged. */
int otal_gn_relu_bwd_sum(int n_terms, const float* const* dy, const int64_t* dy_bs, const int64_t* dy_cs, const int* dy_T,
                         const float* x, const float* gamma, const float* beta, const float* stats, float* dx, float* partial,
                         int B, int C, int T, int G, int relu, int nlev, const int* lev, void* stream);

/* otal_bmp_fwd_levels / otal_bmp_bwd_levels (fp32) with the pooled tensor (out resp. grad_out) a channel slice of a wider
 * (B,Ctot,N) buffer: pooled_bs = elements between samples (>= C*N).  The torch.cat of AFSD/thumos14/BDNet.py:111 and the
 * slice of its gradient, without copies. */
int otal_bmp_fwd_levels_to(const float* in, const float* seg, float* out, int64_t pooled_bs, int B, int C, int nlev,
                           const int* t_start, const int* n_start, void* stream);
int otal_bmp_bwd_levels_from(const float* grad_out, int64_t pooled_bs, const float* in, const float* seg, float* grad_in, int B,
                             int C, int nlev, const int* t_start, const int* n_start, void* stream);

#ifdef __cplusplus
}
#endif
#endif
