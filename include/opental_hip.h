/* include/opental_hip.h -- C ABI of libopental_hip.so (MI355X / gfx950, HIP).
 *
 * Drop-in boundary for the OpenTAL/AFSD detection hot path.  Every entry point takes plain
 * device pointers + sizes + a hipStream_t (passed as void*), launches asynchronously on that
 * stream, never synchronises, owns no persistent state, and returns
 *     0            success
 *    <0            argument error (OTAL_E_*), nothing was launched
 *    >0            a hipError_t from the launch
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

#ifdef __cplusplus
extern "C" {
#endif

#define OTAL_ABI_VERSION 1

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

#define OTAL_MAX_LEVELS 8

int otal_abi_version(void);
const char* otal_error_string(int code);

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

#ifdef __cplusplus
}
#endif
#endif
