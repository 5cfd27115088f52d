// opental_amd/csrc/conv_gemm.hip -- implicit-GEMM convolution on the gfx950 matrix cores.
//
// One kernel family covers every dense convolution of the OpenTAL/AFSD path:
//   Conv3d of the I3D backbone (AFSD/common/i3d_backbone.py:7-87)      : FWD / DGRAD / WGRAD
//   Unit3D [1,6,6] / [1,3,3] pyramid projections (thumos14/BDNet.py:129-155)
//   Unit1D k=1/3, stride 1/2 of the temporal pyramid, towers, ProposalBranch (layers.py:178-214)
// as a GEMM  C[M][N] = A[M][K] * B[K][N]  with B (and for WGRAD also A) gathered on the fly:
//   FWD   : M=Cout  N=B*To*Ho*Wo  K=Cin*kvol   A=W          B=im2col(x)      C=y  (+scale/shift/ReLU)
//   DGRAD : M=Cin   N=B*Ti*Hi*Wi  K=Cout*kvol  A=W^T(packed) B=col2im-gather(dy) C=dx
//   WGRAD : M=Cout  N=Cin*kvol    K=B*To*Ho*Wo A=dy         B=im2col(x)^T    C=dW
// fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain), so the
// result is comparable with the reference at 1e-4 without any reduced-precision step.
//
// Tile: BM in {32,64,128} x BN=128, BK=16, 256 threads = 4 wave64.  Global -> registers -> LDS
// double buffering (one barrier per K step); LDS tiles are stored k-major ([BK][BM+2], [BK][BN+2])
// so every MFMA operand read is 32 consecutive floats per half-wave (conflict free) and the
// k-fast global tiles are written with a 2-bank skew (conflict free as well).
// SAME padding, strides, level-packed pyramids and channel-sliced (concat) tensors are folded
// into the gather (conv_index.h); the ReLU/frozen-BN backward of the PRODUCING layer is folded
// into the DGRAD store epilogue (out_mask/out_scale), so loaders read one tensor per operand.
// Split-K writes fp32 slabs that a second kernel reduces in a fixed order (deterministic).
#include "common.h"
#include "conv_index.h"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

// This file is compiled TWICE (csrc/build.py): part 0 = the C entry points, the dispatch and every kernel that works on fp32
// tensors; part 1 (conv_gemm_half.hip defines OTAL_CONV_PART 1 and includes this file) = the instantiations of the same
// kernel templates for bf16-STORED activations / gradients (template flag H), reached through otal_conv::launch_half().
// Two translation units halve the build's critical path; the kernels themselves exist once, as source.
#ifndef OTAL_CONV_PART
#define OTAL_CONV_PART 0
#endif

#include "conv1a_tile.h"
namespace otal_conv {
struct ConvArgs;
// part 1's dispatcher: the launch for bf16-stored tensors (a.half / a.xhalf set), or OTAL_E_UNSUPPORTED
__attribute__((visibility("hidden"))) int launch_half(int mode, ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t st);
}  // namespace otal_conv

namespace {

constexpr int NT = 256;
enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
enum { EPI_RELU = 1, EPI_ACCUM = 2, DBG_NOLOAD = 4, DBG_NOSTORE = 8, DBG_NOBARRIER = 16, EPI_NPAD8 = 32, EPI_PLAIN_GRID = 256 };   // DBG_*: ablation only (OTAL_CONV_DEBUG)

typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace
struct otal_conv::ConvArgs {
    ConvGeom g;
    ConvFastDiv fd;        // host-built exact fast-division constants (no runtime integer divides)
    const float* x;        // FWD/WGRAD: input activations
    const float* w;        // FWD: W (Cout, Cin*kvol); DGRAD: packed W^T (Cin, Cout*kvol)
    const float* dy;       // DGRAD/WGRAD: output gradient
    float* out;            // FWD: y; DGRAD: dx; WGRAD: dW (Cout, Cin*kvol)
    const float* scale;    // FWD epilogue: per-Cout scale (nullable -> 1)
    const float* shift;    // FWD epilogue: per-Cout shift / bias (nullable -> 0)
    float* slab;           // split-K workspace [splits][M][N] (nullable when splits == 1)
    const float* zero;     // address of a 16-byte device zero word (kernel argument: taking the address of a
                           // __device__ global inside the loop costs a GOT s_load + lgkmcnt(0) wait per element)
    const int2* tab;       // FWD/DGRAD tap table: per k {element offset relative to the anchor, validity bit pattern}
    const float* emask;    // DGRAD epilogue (nullable): dx *= (emask[off] > 0) * escale[ci]  -- the ReLU/BN
    const float* escale;   //   backward of the layer that PRODUCED this conv's input, fused into the store
    int M, N, K;
    int splits, k_per_split;   // k_per_split is a multiple of BK
    int flags;
    int a_vec4;            // A rows are 16-byte aligned and K % 4 == 0 -> float4 weight loads
    int prec;              // 0: fp32 MFMA (exact), 1: bf16 MFMA operands, fp32 accumulate
    // chunked bf16 path (conv_gemm_bf16c_kernel): K re-ordered k = ((c/8) * kvol + tap) * 8 + c%8, padded to Kp
    const int2* ctab;      // per 8-k chunk {byte offset (tap + first channel) relative to the anchor, validity bits}
    const unsigned short* wp;   // A operand packed by the prologue: bf16 [Mpad][Kp], zero padded
    unsigned src_bytes;    // extent of the gathered tensor (buffer descriptor num_records; also the OOB offset)
    unsigned wp_bytes;
    int Kp;
    // vector WGRAD path (conv_wgrad_bf16v_kernel)
    const int2* ptab;      // per CW-position group {byte offset of its window origin in x, t/h validity bits | row-edge flags}
    unsigned dy_bytes;     // extent of dy (buffer descriptor of the A operand)
    int P;                 // positions per sample (To*Ho*Wo), a multiple of 32 on this path
    int w_natural;         // DGRAD: `w` is the forward-layout weight (Cout, Cin, kvol), not the packed transpose
    // batched epilogue (store_acc): byte extents of the output (and of the split-K slabs) when they fit 32-bit offsets, else 0
    unsigned out_bytes, slab_bytes;
    const void* pre;       // persistent prologue region filled earlier (otal_conv_prologue[_batch]); null: build it in the workspace
    int half;              // the large activation operand is STORED as bf16 (fwd: y, dgrad / wgrad: dy); selected kernels only
    // PAIR launches (otal_conv_*_pair): a second problem of the SAME geometry, strides and options rides in the launch of
    // the first -- the 1-D pyramid's sibling layers (loc / conf towers, the two ProposalBranches) are latency-bound at
    // twice the launch floor, so two of them in one grid cost what one does.  pair == 0: the fields below are unused.
    int pair;
    const float* x2;       // as x / dy / out / scale / shift / wp / slab / pre, second problem
    const float* dy2;
    float* out2;
    const float* scale2;
    const float* shift2;
    const unsigned short* wp2;
    float* slab2;
    const void* pre2;
    const float* w2;
    // bf16 STORAGE of the tensors around a backbone layer (ops.HALF_STORAGE; part 1 of this file).  Strides stay in ELEMENTS.
    //   half  : the tensor on the layer's OUTPUT side is bf16 -- fwd: y; dgrad / wgrad: dy
    //   xhalf : the tensor on its INPUT side is bf16          -- fwd: x; dgrad: dx; wgrad: x
    //   mhalf : dgrad: `emask` is bf16 (it has dx's layout)
    int xhalf, mhalf;
};
namespace {
using otal_conv::ConvArgs;

// ---- operand element fetch -------------------------------------------------------------------
// Every gather loads UNCONDITIONALLY: an out-of-range element reads a device zero word instead of
// being guarded.  A guarded load (`ok ? p[off] : 0`) makes hipcc branch around each load and wait
// vmcnt(0) inside the branch (the 16 loads of a K step serialise), and even a select AFTER the load
// pins the wait in front of the MFMA block.  Selecting the POINTER leaves the loaded registers
// untouched until the LDS store behind the MFMAs, so HBM/L2 latency hides under the matrix work.
__device__ __attribute__((aligned(16))) float g_zero4[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ float ld_or_zero(const float* base, int64_t off, bool ok) {
    const float* p = ok ? base + off : g_zero4;
    return *p;
}

// Per-thread gather anchor.  Everything that depends only on the thread's own position is folded,
// ONCE, into (a) a base pointer and (b) a 24-bit validity mask: bit dt says tap dt is inside the
// (level-aware) temporal range, bit 8+dh / 16+dw the same for h / w (kernels up to 8 per axis).  A tap
// is then checked with three shifts and two ANDs and addressed with `base + uniform offset`, where the
// uniform offset (channel stride, tap displacement) is computed on the scalar unit.
struct Anchor {
    const float* base;
    unsigned mask;
};
// an OUTPUT position o reads input (o*s - p + tap): FWD and WGRAD
__device__ __forceinline__ Anchor anchor_of_output(const ConvGeom& g, const float* x, const PosDec& o, bool live) {
    const int t0 = o.t * g.st - g.pt, h0 = o.h * g.sh - g.ph, w0 = o.w * g.sw - g.pw;
    int lo, up;
    level_bounds(g, o.t, g.Ti, lo, up);
    unsigned m = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        m |= (unsigned)(d < g.kt && t0 + d >= lo && t0 + d < up) << d;
        m |= (unsigned)(d < g.kh && (unsigned)(h0 + d) < (unsigned)g.Hi) << (8 + d);
        m |= (unsigned)(d < g.kw && (unsigned)(w0 + d) < (unsigned)g.Wi) << (16 + d);
    }
    Anchor r;
    r.base = x + ((int64_t)o.b * g.x_bs + ((int64_t)t0 * g.Hi + h0) * g.Wi + w0);
    r.mask = live ? m : 0u;
    return r;
}
// an INPUT position i is reached from output ((i + p - tap) / s) when that division is exact: DGRAD.
// Strides are 1 or 2, so (i + p - tap) / s == ((i + p) >> (s-1)) - (tap >> (s-1)) whenever it is exact.
__device__ __forceinline__ Anchor anchor_of_input(const ConvGeom& g, const float* dy, const PosDec& i, bool live) {
    const int tn = i.t + g.pt, hn = i.h + g.ph, wn = i.w + g.pw;
    const int st1 = g.st - 1, sh1 = g.sh - 1, sw1 = g.sw - 1;
    int lo, up;
    level_bounds(g, i.t, g.Ti, lo, up);
    if (g.nlev <= 1) { lo = 0; up = g.To; }
    unsigned m = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int vt = tn - d, vh = hn - d, vw = wn - d;
        m |= (unsigned)(d < g.kt && vt >= 0 && (vt & st1) == 0 && (vt >> st1) >= lo && (vt >> st1) < up) << d;
        m |= (unsigned)(d < g.kh && vh >= 0 && (vh & sh1) == 0 && (vh >> sh1) < g.Ho) << (8 + d);
        m |= (unsigned)(d < g.kw && vw >= 0 && (vw & sw1) == 0 && (vw >> sw1) < g.Wo) << (16 + d);
    }
    Anchor r;
    r.base = dy + ((int64_t)i.b * g.y_bs + ((int64_t)(tn >> st1) * g.Ho + (hn >> sh1)) * g.Wo + (wn >> sw1));
    r.mask = live ? m : 0u;
    return r;
}
__device__ __forceinline__ bool tap_ok(unsigned mask, int dt, int dh, int dw) {
    return ((mask >> dt) & (mask >> (8 + dh)) & (mask >> (16 + dw)) & 1u) != 0;
}
__device__ __forceinline__ float ld_sel(const float* p, bool ok, const float* zero) {
    const float* q = ok ? p : zero;
    return *q;
}

typedef short bf16x8 __attribute__((ext_vector_type(8)));
struct Words4 { unsigned a, b, c, d; };
struct Words2 { unsigned a, b; };
// a + b with 32-bit wrap-around, hidden from the compiler: hipcc folds `voffset + constant` into the buffer
// instruction's immediate offset, and the hardware adds that immediate WITHOUT wrapping (0xfffffffc + 4 is out of range)
__device__ __forceinline__ unsigned wrap_add(unsigned a, unsigned b) {
    unsigned r = a + b;
    asm volatile("" : "+v"(r));
    return r;
}
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {     // v_cvt_pk_bf16_f32: RNE, lo in bits 0..15
    const hw_f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, hw_bf16x2));
}

// ---- helpers of the bf16-STORED tensors (template flag H).  Two words that each hold the bf16 values of two neighbouring
// positions (lo | hi << 16), one word per channel: the pair (channel a, channel b) of the LOW / HIGH position -- one v_perm_b32
// each, where the fp32 tensors need one v_cvt_pk_bf16_f32: the transposition "positions along the load, channels along the
// MFMA operand" costs the same VALU work as the conversion it replaces.
__device__ __forceinline__ unsigned pair_lo(unsigned ch_a, unsigned ch_b) { return __builtin_amdgcn_perm(ch_b, ch_a, 0x05040100u); }
__device__ __forceinline__ unsigned pair_hi(unsigned ch_a, unsigned ch_b) { return __builtin_amdgcn_perm(ch_b, ch_a, 0x07060302u); }
// (value > 0) of a bf16 bit pattern, as all-ones / zero for the half word it sits in: positive, not zero, not NaN
__device__ __forceinline__ bool bf16_pos(unsigned bits16) { return bits16 - 1u < 0x7f80u; }
__device__ __forceinline__ unsigned bf16_relu_mask2(unsigned w) {            // per half word of w: 0xffff where its bf16 is > 0
    return (bf16_pos(w & 0xffffu) ? 0x0000ffffu : 0u) | (bf16_pos(w >> 16) ? 0xffff0000u : 0u);
}
__device__ __forceinline__ float bf16lo_f32(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_f32(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {     // round to nearest even, lo in bits 0..15
    unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
    a += 0x7fffu + ((a >> 16) & 1u);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xffff0000u);
}

// ---- epilogue.  C/D map of the 32x32 MFMAs: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
// Epilogue.  The first version walked the tile element by element -- load mask, wait, load scale, wait, load old value,
// wait, store, with a branch per element -- so every output cost three serialised memory latencies: the 1x1 layers (one
// K step per block) ran at 1.6 TB/s and even Conv3d_2c spent >10 % of its time here.  Now:
//   * per-ROW values (folded BN scale / shift, the producer's BN scale for the fused ReLU/BN backward) are staged ONCE
//     per workgroup in LDS (the A tile's buffer is free after the main loop);
//   * uniform options (mask, accumulate, ReLU, split-K) are hoisted out of the element loops;
//   * the 16 mask / accumulate loads of a 32x32 MFMA tile are issued back to back, then the 16 stores;
//   * rows >= M / columns >= N select an out-of-range buffer offset (loads return 0, stores are dropped): no branches.
// `rows` = LDS scratch of at least 2*BM floats.  Needs the output (and slabs) to fit 32-bit byte offsets; otherwise the
// element-wise fallback below runs.
template <int MODE, int WM, int WN>
__device__ __forceinline__ void store_acc_slow(const ConvArgs& a, const f32x16 (&acc)[WM][WN], int m0, int n0, int wm0, int wn0,
                                               int lane, int split) {
    const ConvGeom& g = a.g;
    const ConvFastDiv& fd = a.fd;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        if (n >= a.N) continue;
        int64_t nbase = 0;
        if (a.splits == 1) {
            if constexpr (MODE == MODE_FWD) nbase = conv_out_offset(g, dec_pos_fd(n, fd.To, fd.Ho, fd.Wo), 0);
            else if constexpr (MODE == MODE_DGRAD) nbase = conv_in_offset(g, dec_pos_fd(n, fd.Ti, fd.Hi, fd.Wi), 0);
            else nbase = n;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= a.M) continue;
                float v = acc[i][j][r];
                if (a.splits > 1) {
                    a.slab[((int64_t)split * a.M + m) * a.N + n] = v;
                    continue;
                }
                int64_t off;
                if constexpr (MODE == MODE_FWD) {
                    if (a.scale) v *= a.scale[m];
                    if (a.shift) v += a.shift[m];
                    if (a.flags & EPI_RELU) v = fmaxf(v, 0.f);
                    off = nbase + (int64_t)m * g.y_cs;
                } else if constexpr (MODE == MODE_DGRAD) {
                    off = nbase + (int64_t)m * g.x_cs;
                    if (a.emask) v = a.emask[off] > 0.f ? v * a.escale[m] : 0.f;
                } else if (a.flags & EPI_NPAD8) {          // columns are (row, dw padded to 8): drop the padding, compact to kw
                    if ((n & 7) >= g.kw) continue;
                    off = (int64_t)m * ((a.N >> 3) * g.kw) + (n >> 3) * g.kw + (n & 7);
                } else {
                    off = (int64_t)m * a.N + nbase;
                }
                if (a.flags & EPI_ACCUM) v += a.out[off];
                a.out[off] = v;
            }
    }
}

template <int MODE, int WM, int WN, int BM>
__device__ __forceinline__ void store_acc(const ConvArgs& a, const f32x16 (&acc)[WM][WN], int m0, int n0, int wm0, int wn0,
                                          int lane, int split, float* rows) {
    const bool slabs = a.splits > 1;
    if ((slabs ? a.slab_bytes : a.out_bytes) == 0u) {
        store_acc_slow<MODE, WM, WN>(a, acc, m0, n0, wm0, wn0, lane, split);
        return;
    }
    const ConvGeom& g = a.g;
    const ConvFastDiv& fd = a.fd;
    constexpr unsigned OOB = 0xffffffffu;
    if (slabs) {            // raw partial sums: slab[split][m][n]
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.slab, 0, (int)a.slab_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + wn0 + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const unsigned vo = (n < a.N && m < a.M) ? (unsigned)((((unsigned)split * a.M + m) * (unsigned)a.N + n) * 4u) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][r]), rs, vo, 0, 0);
                }
        }
        return;
    }
    // ---- per-row values -> LDS
    __syncthreads();        // every wave is done reading the operand tiles
    for (int r = threadIdx.x; r < BM; r += (int)blockDim.x) {
        const int m = m0 + r;
        float s0 = 1.f, s1 = 0.f;
        if (m < a.M) {
            if constexpr (MODE == MODE_FWD) { if (a.scale) s0 = a.scale[m]; if (a.shift) s1 = a.shift[m]; }
            else if constexpr (MODE == MODE_DGRAD) { if (a.escale) s0 = a.escale[m]; }
        }
        rows[2 * r] = s0; rows[2 * r + 1] = s1;
    }
    __syncthreads();
    const auto ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const auto rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.emask ? a.emask : a.out), 0, (int)a.out_bytes, 0x00020000);
    const bool relu = (a.flags & EPI_RELU) != 0, accum = (a.flags & EPI_ACCUM) != 0, masked = MODE == MODE_DGRAD && a.emask != nullptr;
    const unsigned row_stride = MODE == MODE_FWD ? (unsigned)g.y_cs : (MODE == MODE_DGRAD ? (unsigned)g.x_cs : 0u);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const bool okn = n < a.N;
        unsigned nbase = 0;
        if constexpr (MODE == MODE_FWD) nbase = (unsigned)conv_out_offset(g, dec_pos_fd(okn ? n : 0, fd.To, fd.Ho, fd.Wo), 0);
        else if constexpr (MODE == MODE_DGRAD) nbase = (unsigned)conv_in_offset(g, dec_pos_fd(okn ? n : 0, fd.Ti, fd.Hi, fd.Wi), 0);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            unsigned vo[16];
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int m = m0 + lr;
                bool ok = okn && m < a.M;
                unsigned off;
                if constexpr (MODE == MODE_WGRAD) {
                    if (a.flags & EPI_NPAD8) {     // columns are (row, dw padded to 8): drop the padding, compact to kw
                        ok = ok && (n & 7) < g.kw;
                        off = (unsigned)m * (unsigned)((a.N >> 3) * g.kw) + (unsigned)((n >> 3) * g.kw + (n & 7));
                    } else {
                        off = (unsigned)m * (unsigned)a.N + (unsigned)n;
                    }
                } else {
                    off = nbase + (unsigned)m * row_stride;
                }
                vo[r] = ok ? off * 4u : OOB;
                float x = acc[i][j][r];
                if constexpr (MODE == MODE_FWD) {
                    x = x * rows[2 * lr] + rows[2 * lr + 1];
                    if (relu) x = fmaxf(x, 0.f);
                } else if constexpr (MODE == MODE_DGRAD) {
                    x *= rows[2 * lr];              // escale (1 when there is no mask)
                }
                v[r] = x;
            }
            if (masked) {
                float mk[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) mk[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rm, vo[r], 0, 0));
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = mk[r] > 0.f ? v[r] : 0.f;
            }
            if (accum) {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) old[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, vo[r], 0, 0));
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += old[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), ro, vo[r], 0, 0);
        }
    }
}

// Epilogue for a bf16-STORED output (H kernels; FWD: y, DGRAD: dx).  The fp32 epilogue above issues sixteen 4-byte stores
// per MFMA tile and lane -- and, for a masked data gradient, sixteen 4-byte mask loads in front of them: on the 1x1x1
// layers that was 45 - 65 % of the launch.  Here the tile is finished in registers (folded BN / ReLU, or the
// producer's BN scale), rounded to nearest even ONCE, transposed through LDS (the operand tiles are dead) and written as
// 16-byte runs of eight positions; the ReLU mask of a data gradient is the bf16 activation itself, read with the same
// 16-byte pieces.  Positions: n = b * P + p with P % 8 == 0 (checked on the host), so a run never leaves its sample and is
// contiguous whatever the convolution's stride.  No split-K here (slabs stay fp32: store_acc), no accumulate.
// `lds`: at least BM * (BN * 2 + 16) + BM * 8 bytes.
template <int MODE, int WM, int WN, int BM, int BN>
__device__ __forceinline__ void store_acc_h(const ConvArgs& a, const f32x16 (&acc)[WM][WN], int m0, int n0, int wm0, int wn0,
                                            int lane, unsigned char* lds, int nthreads) {
    static_assert(MODE != MODE_WGRAD, "weight gradients are fp32");
    constexpr int PT = BN * 2 + 16;
    const ConvGeom& g = a.g;
    float* rows = reinterpret_cast<float*>(lds + BM * PT);
    __syncthreads();        // every wave is done reading the operand tiles
    for (int r = threadIdx.x; r < BM; r += nthreads) {
        const int m = m0 + r;
        float s0 = 1.f, s1 = 0.f;
        if (m < a.M) {
            if constexpr (MODE == MODE_FWD) { if (a.scale) s0 = a.scale[m]; if (a.shift) s1 = a.shift[m]; }
            else { if (a.escale) s0 = a.escale[m]; }
        }
        rows[2 * r] = s0; rows[2 * r + 1] = s1;
    }
    __syncthreads();
    const bool relu = (a.flags & EPI_RELU) != 0;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int nl = wn0 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r];
                if constexpr (MODE == MODE_FWD) {
                    v = v * rows[2 * lr] + rows[2 * lr + 1];
                    if (relu) v = fmaxf(v, 0.f);
                } else {
                    v *= rows[2 * lr];
                }
                *reinterpret_cast<unsigned short*>(lds + lr * PT + nl * 2) = (unsigned short)(cvt_pk_bf16(v, 0.f) & 0xffffu);
            }
        }
    __syncthreads();
    unsigned short* const out = reinterpret_cast<unsigned short*>(a.out);
    const unsigned short* const mk = (MODE == MODE_DGRAD) ? reinterpret_cast<const unsigned short*>(a.emask) : nullptr;
    const int64_t bs = MODE == MODE_FWD ? g.y_bs : g.x_bs, cs = MODE == MODE_FWD ? g.y_cs : g.x_cs;
    const FastDiv fP = a.fd.P;                                          // DGRAD: the host admits stride-1 SAME layers only (in = out positions)
    constexpr int PIECES = BM * (BN / 8);
    constexpr int U = BM >= 192 ? 3 : 6;                    // pieces per trip: all their mask loads are in flight together (one memory
                                                            // round trip per trip -- the first version took two pieces per trip, three trips)
    for (int p0 = threadIdx.x; p0 < PIECES; p0 += U * nthreads) {
        int64_t off[U];
        bool ok[U];
        Words4 m4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u * nthreads;
            const int row = p / (BN / 8), q = p - row * (BN / 8);
            const int m = m0 + row, n = n0 + q * 8;
            ok[u] = p < PIECES && m < a.M && n < a.N;
            const uint32_t b = fd_div(fP, ok[u] ? (uint32_t)n : 0u);
            off[u] = ok[u] ? (int64_t)b * bs + (int64_t)m * cs + (int64_t)((uint32_t)n - b * fP.d) : 0;
            if (mk) m4[u] = *reinterpret_cast<const Words4*>(mk + off[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const int p = p0 + u * nthreads;
            const int row = p / (BN / 8), q = p - row * (BN / 8);
            Words4 v = *reinterpret_cast<const Words4*>(lds + row * PT + q * 16);
            if (mk) {
                v.a &= bf16_relu_mask2(m4[u].a); v.b &= bf16_relu_mask2(m4[u].b);
                v.c &= bf16_relu_mask2(m4[u].c); v.d &= bf16_relu_mask2(m4[u].d);
            }
            *reinterpret_cast<Words4*>(out + off[u]) = v;
        }
    }
}

// XCD-aware tile order.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  Give every
// XCD a CONTIGUOUS range of logical tiles instead, ordered so that the tiles which read the same gathered data are
// neighbours: FWD / DGRAD -- the M tiles of one position tile, then the next position tile (shared halo rows);
// WGRAD -- all (N, M) tiles of one split (they read the same positions of x and dy), then the next split.
// A bijection for any grid; placement is an optimisation only (the id -> XCD map is not architecturally guaranteed).
struct TileId { int x, y, z; };
__device__ __forceinline__ TileId xcd_tile(bool y_fastest) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int total = gx * gy * gz;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int xcd = lin & 7, idx = lin >> 3;
    const int full = total >> 3, rem = total & 7;
    int l = xcd * full + min(xcd, rem) + idx;
    TileId t;
    if (y_fastest) { t.y = l % gy; l /= gy; t.x = l % gx; t.z = l / gx; }
    else { t.x = l % gx; l /= gx; t.y = l % gy; t.z = l / gy; }
    return t;
}

// PREC 0: fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32; BK = 16, LDS tiles k-major [k][m]).
// PREC 1: operands rounded to bf16 while they are staged into LDS, v_mfma_f32_32x32x16_bf16 with fp32
//         accumulation (16x the matrix rate; BK = 32, LDS tiles k-contiguous [m][k] with an 80-byte row
//         pitch so that every ds_read_b128 operand fetch is bank-conflict free).  Tensors stay fp32 in HBM.
template <int BM, int BN, int WM, int WN, int MODE, bool AVEC, int PREC>
__global__ __launch_bounds__(NT) void conv_gemm_kernel(const ConvArgs a) {
    constexpr int BK = PREC ? 32 : 16;
    constexpr int KP = 40;                                  // bf16 row pitch (elements): 32 + 8 pad = 80 bytes
    constexpr int LDA = BM + 2, LDB = BN + 2;               // fp32 k-major pitches
    constexpr int A_BYTES = PREC ? BM * KP * 2 : BK * LDA * 4;
    constexpr int B_BYTES = PREC ? BN * KP * 2 : BK * LDB * 4;
    __shared__ __attribute__((aligned(16))) char smemA[2][A_BYTES];
    __shared__ __attribute__((aligned(16))) char smemB[2][B_BYTES];
    static_assert(BN == 128, "BN");

    const ConvGeom& g = a.g;
    const ConvFastDiv& fd = a.fd;
    const float* const zp = a.zero;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileId tile = xcd_tile(MODE != MODE_WGRAD);
    const int m0 = tile.y * BM;
    const int n0 = tile.x * BN;
    const int split = tile.z;
    const int k_begin = split * a.k_per_split;
    const int k_end = min(a.K, k_begin + a.k_per_split);
    const int HWi = g.Hi * g.Wi;

    // ---- thread -> tile element maps
    // B, n-fast (FWD/DGRAD): n = tid & 127; this thread owns BK/2 CONSECUTIVE k rows: k = (tid >> 7) * BK/2 + j
    //                         (consecutive so that the bf16 path can pack them into ds_write_b128; wave-uniform)
    // A, weights AVEC       : float4 along k: k = (tid % (BK/4)) * 4, m = tid / (BK/4) + (NT*4/BK) * j
    // A, weights scalar     : k = tid % BK, m = tid / BK + (NT/BK) * j
    // WGRAD (k-fast, both operands): k = (tid & 15) + 16 * p (p < BK/16), rows/cols = (tid >> 4) + 16 * j
    constexpr int B_PER = BK / 2;                           // B elements per thread per K step (n-fast)
    constexpr int AV_TPR = BK / 4;                          // threads per A row (float4 path)
    constexpr int AV_ROWS = NT / AV_TPR;                    // rows per pass: 64 (fp32) / 32 (bf16)
    constexpr int AV_PASS = (BM + AV_ROWS - 1) / AV_ROWS;
    constexpr int AS_ROWS = NT / BK;                        // scalar path rows per pass: 16 / 8
    constexpr int AS_PER = BM / AS_ROWS;
    constexpr int KSUB = BK / 16;                           // WGRAD k sub-steps per K step
    constexpr int WA_PER = BM / 16, WB_PER = BN / 16;       // WGRAD rows / cols per thread
    const int b_n = tid & (BN - 1);
    const int b_kq = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int v_k = (tid % AV_TPR) * 4, v_m = tid / AV_TPR;
    const int s_k = tid % BK, s_m = tid / BK;
    const int w_k = tid & 15, w_r = tid >> 4;

    Anchor anchor = {zp, 0u};           // FWD: this thread's output column; DGRAD: its input position
    int wcoff[MODE == MODE_WGRAD ? WB_PER : 1];             // WGRAD: fixed (ci, tap) columns: element offset
    int wtap[MODE == MODE_WGRAD ? WB_PER : 1];              //        dt | dh << 8 | dw << 16, or -1
    if constexpr (MODE == MODE_FWD) {
        const int n = n0 + b_n;
        anchor = anchor_of_output(g, a.x, dec_pos_fd(n < a.N ? n : 0, fd.To, fd.Ho, fd.Wo), n < a.N);
    } else if constexpr (MODE == MODE_DGRAD) {
        const int n = n0 + b_n;
        anchor = anchor_of_input(g, a.dy, dec_pos_fd(n < a.N ? n : 0, fd.Ti, fd.Hi, fd.Wi), n < a.N);
    } else {
#pragma unroll
        for (int j = 0; j < WB_PER; ++j) {
            const int n = n0 + w_r + 16 * j;
            const TapDec t = dec_tap_fd(fd, n < a.N ? n : 0);
            wcoff[j] = t.c * (int)g.x_cs + t.dt * HWi + t.dh * g.Wi + t.dw;
            wtap[j] = n < a.N ? (t.dt | (t.dh << 8) | (t.dw << 16)) : -1;
        }
    }
    // A rows of this thread as 32-bit element offsets (-1 = out of range)
    // (WGRAD with AVEC: dy rows, 4 consecutive positions per float4 -- legal because P % 4 == 0 keeps a
    //  group of 4 k inside one sample and 16-byte aligned)
    constexpr int A_ROWSN = AVEC ? AV_PASS : (MODE == MODE_WGRAD ? WA_PER : AS_PER);
    int arow[A_ROWSN];
#pragma unroll
    for (int j = 0; j < A_ROWSN; ++j) {
        if constexpr (MODE == MODE_WGRAD && AVEC) {
            const int m = m0 + v_m + AV_ROWS * j;
            arow[j] = (v_m + AV_ROWS * j < BM && m < a.M) ? m * (int)g.y_cs : -1;
        } else if constexpr (MODE == MODE_WGRAD) {
            const int m = m0 + w_r + 16 * j;
            arow[j] = m < a.M ? m * (int)g.y_cs : -1;
        } else if constexpr (AVEC) {
            const int m = m0 + v_m + AV_ROWS * j;
            arow[j] = (v_m + AV_ROWS * j < BM && m < a.M) ? m * a.K + v_k : -1;
        } else {
            const int m = m0 + s_m + AS_ROWS * j;
            arow[j] = m < a.M ? m * a.K + s_k : -1;
        }
    }

    // ---- staging registers
    constexpr int A_REGS = AVEC ? 4 * AV_PASS : (MODE == MODE_WGRAD ? WA_PER * KSUB : AS_PER);
    constexpr int B_REGS = MODE == MODE_WGRAD ? WB_PER * KSUB : B_PER;
    constexpr int A_LOADS = AVEC ? AV_PASS : (MODE == MODE_WGRAD ? WA_PER * KSUB : AS_PER);
    constexpr int B_LOADS = B_REGS;
    float ra[A_REGS], rb[B_REGS];
    // WGRAD per-(K step, sub-step) position state
    const float* wxp[KSUB];
    const float* wdyp[KSUB];
    int wt0[KSUB], wh0[KSUB], ww0[KSUB];
    unsigned wtr[KSUB];
#pragma unroll
    for (int p = 0; p < KSUB; ++p) { wxp[p] = zp; wdyp[p] = zp; wt0[p] = wh0[p] = ww0[p] = 0; wtr[p] = 0u; }

    int2 te[MODE == MODE_WGRAD ? 1 : B_PER];                // this K step's tap-table entries (scalar registers)
    const float* wdy4 = zp;                            // WGRAD+AVEC: dy + b*y_bs + p of this thread's 4 k
    bool wok4 = false;
    auto prep = [&](int k0, bool live) {
        if constexpr (MODE == MODE_WGRAD && AVEC) {
            const int k = k0 + v_k;
            wok4 = live & (k < k_end);
            const uint32_t kb = fd_div(fd.P, wok4 ? (uint32_t)k : 0u);          // sample index
            wdy4 = a.dy + ((int64_t)kb * g.y_bs + (int64_t)((wok4 ? k : 0) - (int)(kb * fd.P.d)));
        }
        if constexpr (MODE != MODE_WGRAD) {
            const int2* tp = a.tab + (k0 + b_kq * B_PER);       // wave-uniform -> wide s_load; table is padded past K
#pragma unroll
            for (int q = 0; q < B_PER; ++q) te[q] = tp[q];
        }
        if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
            for (int p = 0; p < KSUB; ++p) {
                const int k = k0 + 16 * p + w_k;
                const bool kok = live & (k < k_end);
                const PosDec o = dec_pos_fd(kok ? k : 0, fd.To, fd.Ho, fd.Wo);
                int lo, up;
                level_bounds(g, o.t, g.Ti, lo, up);
                const int t0 = o.t * g.st - g.pt;
                wh0[p] = o.h * g.sh - g.ph;
                ww0[p] = o.w * g.sw - g.pw;
                wt0[p] = t0 - lo;
                wtr[p] = kok ? (unsigned)(up - lo) : 0u;
                wdyp[p] = a.dy + ((int64_t)o.b * g.y_bs + ((int64_t)o.t * g.Ho + o.h) * g.Wo + o.w);
                wxp[p] = a.x + ((int64_t)o.b * g.x_bs + ((int64_t)t0 * g.Hi + wh0[p]) * g.Wi + ww0[p]);
            }
        }
    };
    auto loadA = [&](int q, int k0, bool live) {         // q-th A load of the K step
        if constexpr (MODE == MODE_WGRAD && AVEC) {
            const float* ap = (wok4 & (arow[q] >= 0)) ? wdy4 + arow[q] : zp;
            const float4 v = *reinterpret_cast<const float4*>(ap);
            ra[4 * q] = v.x; ra[4 * q + 1] = v.y; ra[4 * q + 2] = v.z; ra[4 * q + 3] = v.w;
        } else if constexpr (MODE == MODE_WGRAD) {
            const int p = q / WA_PER, j = q % WA_PER;
            ra[q] = ld_sel(wdyp[p] + arow[j], (wtr[p] != 0u) & (arow[j] >= 0), zp);
        } else if constexpr (AVEC) {
            const bool ok = (live & (arow[q] >= 0)) & (k0 + v_k < k_end);
            const float* ap = ok ? a.w + arow[q] + k0 : zp;
            const float4 v = *reinterpret_cast<const float4*>(ap);
            ra[4 * q] = v.x; ra[4 * q + 1] = v.y; ra[4 * q + 2] = v.z; ra[4 * q + 3] = v.w;
        } else {
            ra[q] = ld_sel(a.w + arow[q] + k0, (live & (arow[q] >= 0)) & (k0 + s_k < k_end), zp);
        }
    };
    auto loadB = [&](int q, int k0, bool live) {         // q-th B load of the K step
        if constexpr (MODE == MODE_WGRAD) {
            const int p = q / WB_PER, j = q % WB_PER;
            const int tp = wtap[j];
            const bool ok = (tp >= 0) & ((unsigned)(wt0[p] + (tp & 255)) < wtr[p]) &
                            ((unsigned)(wh0[p] + ((tp >> 8) & 255)) < (unsigned)g.Hi) &
                            ((unsigned)(ww0[p] + ((tp >> 16) & 255)) < (unsigned)g.Wi);
            rb[q] = ld_sel(wxp[p] + wcoff[j], ok, zp);
        } else {
            // k is wave-uniform; its tap was decoded once per launch into the table: e.x = element offset
            // relative to this thread's anchor, e.y = (1<<dt | 1<<(8+dh) | 1<<(16+dw)).  The tap is inside the
            // tensor for this thread iff all three bits are set in the thread's validity mask.
            const int kk = k0 + b_kq * B_PER + q;
            const int2 e = te[q];
            const bool ok = (live & (kk < k_end)) & ((anchor.mask & (unsigned)e.y) == (unsigned)e.y);
            rb[q] = ld_sel(anchor.base + e.x, ok, zp);
        }
    };
    auto store_tiles = [&](int buf) {
        if constexpr (PREC == 0) {
            float* As = reinterpret_cast<float*>(smemA[buf]);
            float* Bs = reinterpret_cast<float*>(smemB[buf]);
            // A
            if constexpr (AVEC) {
#pragma unroll
                for (int j = 0; j < AV_PASS; ++j)
                    if (v_m + AV_ROWS * j < BM) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) As[(v_k + i) * LDA + v_m + AV_ROWS * j] = ra[4 * j + i];
                    }
            } else if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
                for (int j = 0; j < WA_PER; ++j) As[w_k * LDA + w_r + 16 * j] = ra[j];
            } else {
#pragma unroll
                for (int j = 0; j < AS_PER; ++j) As[s_k * LDA + s_m + AS_ROWS * j] = ra[j];
            }
            // B
            if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
                for (int j = 0; j < WB_PER; ++j) Bs[w_k * LDB + w_r + 16 * j] = rb[j];
            } else {
#pragma unroll
                for (int j = 0; j < B_PER; ++j) Bs[(b_kq * B_PER + j) * LDB + b_n] = rb[j];
            }
        } else {
            unsigned short* As = reinterpret_cast<unsigned short*>(smemA[buf]);
            unsigned short* Bs = reinterpret_cast<unsigned short*>(smemB[buf]);
            // A
            if constexpr (AVEC) {
#pragma unroll
                for (int j = 0; j < AV_PASS; ++j)
                    if (v_m + AV_ROWS * j < BM) {
                        uint2 pk;
                        pk.x = pack_bf16x2(ra[4 * j], ra[4 * j + 1]);
                        pk.y = pack_bf16x2(ra[4 * j + 2], ra[4 * j + 3]);
                        *reinterpret_cast<uint2*>(As + (v_m + AV_ROWS * j) * KP + v_k) = pk;
                    }
            } else if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
                for (int q = 0; q < WA_PER * KSUB; ++q)
                    As[(w_r + 16 * (q % WA_PER)) * KP + 16 * (q / WA_PER) + w_k] = (unsigned short)(pack_bf16x2(ra[q], 0.f) & 0xffffu);
            } else {
#pragma unroll
                for (int j = 0; j < AS_PER; ++j)
                    As[(s_m + AS_ROWS * j) * KP + s_k] = (unsigned short)(pack_bf16x2(ra[j], 0.f) & 0xffffu);
            }
            // B
            if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
                for (int q = 0; q < WB_PER * KSUB; ++q)
                    Bs[(w_r + 16 * (q % WB_PER)) * KP + 16 * (q / WB_PER) + w_k] = (unsigned short)(pack_bf16x2(rb[q], 0.f) & 0xffffu);
            } else {
#pragma unroll
                for (int h = 0; h < B_PER / 8; ++h) {
                    uint4 pk;
                    pk.x = pack_bf16x2(rb[8 * h], rb[8 * h + 1]);
                    pk.y = pack_bf16x2(rb[8 * h + 2], rb[8 * h + 3]);
                    pk.z = pack_bf16x2(rb[8 * h + 4], rb[8 * h + 5]);
                    pk.w = pack_bf16x2(rb[8 * h + 6], rb[8 * h + 7]);
                    *reinterpret_cast<uint4*>(Bs + b_n * KP + b_kq * B_PER + 8 * h) = pk;
                }
            }
        }
    };

    // ---- wave tile placement
    constexpr int WAVES_N = BN / (32 * WN);
    const int wm0 = (wave / WAVES_N) * (32 * WM);
    const int wn0 = (wave % WAVES_N) * (32 * WN);
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (k_end - k_begin + BK - 1) / BK;
    const bool dbg_load = !(a.flags & DBG_NOLOAD);
    prep(k_begin, nk > 0);
#pragma unroll
    for (int q = 0; q < A_LOADS; ++q) loadA(q, k_begin, nk > 0);
#pragma unroll
    for (int q = 0; q < B_LOADS; ++q) loadB(q, k_begin, nk > 0);
    store_tiles(0);
    __syncthreads();
    // Main loop, software-pipelined INSIDE the wave: the address math + global loads of tile it+1 are
    // issued in NCH slices between the MFMA groups of tile it, the loaded registers are only touched by
    // the LDS store after the last MFMA, and one barrier per K step flips the LDS buffers.
    constexpr int NCH = PREC ? 2 : 8;                       // MFMA groups per K step
    constexpr int A_PER_CH = (A_LOADS + NCH - 1) / NCH, B_PER_CH = (B_LOADS + NCH - 1) / NCH;
    for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        const bool has_next = (it + 1 < nk) && dbg_load;
        const int kn = k_begin + (it + 1) * BK;
        prep(kn, has_next);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int q = 0; q < A_PER_CH; ++q)
                if (c * A_PER_CH + q < A_LOADS) loadA(c * A_PER_CH + q, kn, has_next);
#pragma unroll
            for (int q = 0; q < B_PER_CH; ++q)
                if (c * B_PER_CH + q < B_LOADS) loadB(c * B_PER_CH + q, kn, has_next);
            if constexpr (PREC == 0) {
                const float* as = reinterpret_cast<const float*>(smemA[buf]);
                const float* bs = reinterpret_cast<const float*>(smemB[buf]);
                float av[WM], bv[WN];
                const int kr = 2 * c + (lane >> 5);
#pragma unroll
                for (int i = 0; i < WM; ++i) av[i] = as[kr * LDA + wm0 + i * 32 + (lane & 31)];
#pragma unroll
                for (int j = 0; j < WN; ++j) bv[j] = bs[kr * LDB + wn0 + j * 32 + (lane & 31)];
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            } else {
                const unsigned short* as = reinterpret_cast<const unsigned short*>(smemA[buf]);
                const unsigned short* bs = reinterpret_cast<const unsigned short*>(smemB[buf]);
                bf16x8 av[WM], bv[WN];
                const int ko = 16 * c + (lane >> 5) * 8;        // lanes 0-31: k 0..7, lanes 32-63: k 8..15 of the group
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    av[i] = *reinterpret_cast<const bf16x8*>(as + (wm0 + i * 32 + (lane & 31)) * KP + ko);
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    bv[j] = *reinterpret_cast<const bf16x8*>(bs + (wn0 + j * 32 + (lane & 31)) * KP + ko);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the slices where they are
        }
        if (!(a.flags & DBG_NOSTORE)) store_tiles(buf ^ 1);   // harmless on the last step (buffer is never read)
        if (!(a.flags & DBG_NOBARRIER)) __syncthreads();
    }

    store_acc<MODE, WM, WN, BM>(a, acc, m0, n0, wm0, wn0, lane, split, reinterpret_cast<float*>(smemA[0]));
}

// =================================================================================================
// Chunked bf16 kernel (FWD / DGRAD, channel count % 8 == 0): the vector ALU, not the matrix core, bounded
// the generic kernel above (rocprofv3: ~40 VALU + 21 SALU instructions per MFMA, profiles/r01_pmc_sq_*):
// per gathered element it paid a validity test, a pointer select, 64-bit address math and a software
// bf16 rounding.  Here
//   * K is re-ordered in chunks of 8 channels x one tap, k = ((c/8) * kvol + tap) * 8 + c%8 (padded to a multiple
//     of 32), so the 8 consecutive k a thread fetches share one tap: ONE validity test + ONE offset select per 8
//     loads; the 27 taps of a channel block are adjacent in K, so neighbouring taps re-hit the same lines in L1;
//   * loads are buffer_load_dword with the per-thread byte offset in a VGPR and the channel stride i*cs in an
//     SGPR; an invalid tap selects an offset == num_records and the hardware bounds check returns 0.0;
//   * the weights are packed ONCE per launch by the prologue kernel into bf16 [Mpad][Kp] (zero padded: no row /
//     tail predicates, 8-byte loads, no conversion in the loop);
//   * activations are rounded with v_cvt_pk_bf16_f32 (RNE) while they are staged into LDS.
// Result: ~5 VALU per MFMA.  MFMA loop, LDS layout (80-byte pitch) and epilogue are those of the generic kernel.
// CW = consecutive positions one thread fetches per load (the texture addresser is the next bound after the VALU:
// a wave64 buffer_load_dword costs ~14 CU cycles whatever it returns, a dwordx4 ~23 -- tools/ubench/bufcheck.hip):
//   CW 1: thread = 1 position x 16 k (two chunks), 16 dword loads          -- any geometry
//   CW 4: thread = 4 positions x 4 k of the wave's chunk, 4 dwordx4 loads  -- stride 1, Wo == Wi, W % 4 == 0, kw in {1,3}
//   CW 2: thread = 2 positions x 8 k of the wave's chunk, 8 dwordx2 loads  -- same with W % 2 == 0
// With CW > 1 the w-axis validity is not a per-thread bit any more: the vector is contiguous in w, a tap shifted by
// -1 / +1 invalidates only its FIRST / LAST element and only for the thread whose vector touches the row start / end,
// so two conditional zeroings per vector replace the per-element test.  LDS rows are permuted
// (row(n) = (n % CW) * PB + n / CW) so that both the staged writes and the ds_read_b128 operand reads stay
// conflict free (PB = 36 / 72, searched with the bank model of MI355X_MICROARCH.md).
//   KWV (with CW 1, forward only): for layers whose channel count is not a multiple of 8 (Conv3d_1a: Cin = 3) a chunk
//   is the kw-run of ONE (ci, dt, dh) row padded to 8 taps, k = ((ci*kt + dt)*kh + dh)*8 + dw.  The 8 taps are
//   contiguous in memory whatever the stride: two dwordx4 per chunk; t / h validity per chunk as before, w validity
//   is a per-THREAD constant (the anchor's w bits) applied as 8 masks.
// Occupancy: left alone hipcc gives the 96-row variant 86 VGPRs + 48 AGPRs = 134 registers -- three workgroups per CU where
// LDS (37 KB) allows four -- and the 192-row one 230 (two where 53 KB allows three).  Asked for the occupancy it keeps the
// accumulators in VGPRs (106 / 164 registers, no spills).  The 1x1x1 layers these kernels mostly serve are latency-bound
// per workgroup (8 .. 26 K steps between a cold first load and a 16-store epilogue): Mixed_3c fused 1x1 forward
// 101 -> 94 us.  (The 128-row variant spills at three per CU: left at two.)
// H: the gathered tensor (FWD: x, DGRAD: dy) and the output are STORED as bf16 (ConvArgs::half / xhalf).  A load of CW positions
// is CW * 2 bytes (b64 / b32 / b16 for CW 4 / 2 / 1 -- raw-buffer loads are legal at 2-byte aligned addresses on gfx950 and
// return correct data, tools/ubench/align2probe.hip; plain global loads are NOT: they drop address bit 1); the values go to
// LDS as they are (pair_lo / pair_hi transpose "positions along the load" into "channels along the operand" for the price
// of the conversion they replace), so the MFMA operands -- and with them every accumulator -- are bit for bit those of the
// fp32-tensor kernel fed with the same bf16 values; the epilogue is store_acc_h.  Split-K slabs stay fp32.
template <int BM, int WM, int WN, int MODE, int CW, bool KWV = false, bool H = false>
__global__ __launch_bounds__(NT, KWV ? 1 : (BM == 96 ? 4 : (BM == 192 ? 3 : 1))) void conv_gemm_bf16c_kernel(const ConvArgs a) {
    static_assert(!KWV || (CW == 1 && MODE == MODE_FWD && !H), "kw-vector mode");
    constexpr int BN = 128, BK = 32, KP = 40;
    constexpr int ESZ = H ? 2 : 4;                           // bytes per element of the gathered tensor
    constexpr int PB = CW == 4 ? 36 : (CW == 2 ? 72 : 0);   // LDS row-block pitch of the position permutation
    constexpr int B_ROWS = CW == 1 ? BN : (CW - 1) * PB + BN / CW;
    constexpr int A_PIECES = (BM * 4 + NT - 1) / NT;        // 16-byte weight pieces per thread per K step
    constexpr int OPER_BYTES = 2 * (BM + B_ROWS) * KP * 2, EPI_BYTES = H ? BM * (BN * 2 + 16) + BM * 8 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem_[OPER_BYTES > EPI_BYTES ? OPER_BYTES : EPI_BYTES];
    unsigned short (*const smA)[BM * KP] = reinterpret_cast<unsigned short (*)[BM * KP]>(smem_);
    unsigned short (*const smB)[B_ROWS * KP] = reinterpret_cast<unsigned short (*)[B_ROWS * KP]>(smem_ + 2 * BM * KP * 2);

    const ConvGeom& g = a.g;
    const ConvFastDiv& fd = a.fd;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileId tile = xcd_tile(true);
    const int m0 = tile.y * BM;
    const int n0 = tile.x * BN;
    const int split = tile.z;
    const int k_begin = split * a.k_per_split;
    const int k_end = min(a.Kp, k_begin + a.k_per_split);
    const int nk = (k_end - k_begin) / BK;                  // Kp and k_per_split are multiples of 32

    const float* src = MODE == MODE_FWD ? a.x : a.dy;       // (H: bf16 data behind the float pointer; offsets below are in elements)
    const int cs_bytes = (int)((MODE == MODE_FWD ? g.x_cs : g.y_cs) * ESZ);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)a.src_bytes, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wp), 0, (int)a.wp_bytes, 0x00020000);

    // ---- thread -> B tile map
    //  CW 1: column b_n = tid & 127, chunks (tid >> 7) * 2 + {0, 1} of the K step
    //  CW 4: columns 4q..4q+3 (q = tid & 31), chunk = wave, k = (lane >> 5) * 4 + i inside it
    //  CW 2: columns 2q, 2q+1 (q = lane),     chunk = wave, k = i
    const int b_q = CW == 1 ? (tid & 127) : (CW == 4 ? (tid & 31) : lane);
    const int b_kq = __builtin_amdgcn_readfirstlane(tid >> 7);
    unsigned voff0, vmask, keepL_thr = 0u, keepR_thr = 0u;      // keep*_thr: all-ones when this thread's vector touches the row start / end
    {
        const int n = n0 + CW * b_q;
        Anchor an;
        PosDec pd;
        if constexpr (MODE == MODE_FWD) {
            pd = dec_pos_fd(n < a.N ? n : 0, fd.To, fd.Ho, fd.Wo);
            an = anchor_of_output(g, src, pd, n < a.N);
        } else {
            pd = dec_pos_fd(n < a.N ? n : 0, fd.Ti, fd.Hi, fd.Wi);
            an = anchor_of_input(g, src, pd, n < a.N);
        }
        voff0 = (unsigned)((an.base - src) * ESZ);          // may be "negative": the sum with a valid tap offset is not
        vmask = an.mask;
        if constexpr (CW > 1) {
            voff0 += (unsigned)((CW == 4 ? (lane >> 5) * 4 : 0) * cs_bytes);
            keepL_thr = pd.w == 0 ? 0xffffffffu : 0u;
            keepR_thr = pd.w + CW == g.Wi ? 0xffffffffu : 0u;
        }
    }
    unsigned keepw[KWV ? 8 : 1];                            // KWV: all-ones where tap dw of this thread's window is inside the row
    if constexpr (KWV) {
#pragma unroll
        for (int j = 0; j < 8; ++j) keepw[j] = (vmask >> (16 + j)) & 1u ? 0xffffffffu : 0u;
    }
    // weight pieces: piece p = tid + 256 j -> row p >> 2, 8 bf16 at k = (p & 3) * 8
    unsigned voffA[A_PIECES];
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
        const int p = tid + NT * j;
        voffA[j] = (unsigned)(((m0 + (p >> 2)) * a.Kp + (p & 3) * 8) * 2);
    }

    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[A_PIECES];
    float rb[H ? 1 : 16];                                   // CW 1: [chunk h][k i]; CW 4: [k i][pos j]; CW 2: [k i][pos j]
    unsigned rh[H ? 16 : 1];                                // H: the loaded words -- CW 4: [k i][2]; CW 2: [k i]; CW 1: [chunk h][k i], one bf16 each
    // byte offset of a chunk-table entry: the table is built for 4-byte elements
    auto tab_off = [](unsigned long long e) { return H ? (unsigned)((int)(unsigned)e >> 1) : (unsigned)e; };
    auto loadA = [&](int k0) {
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j)
            if ((BM * 4) % NT == 0 || tid + NT * j < BM * 4)
                ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, voffA[j], k0 * 2, 0);
    };
    // chunk-table entries of the NEXT K step, fetched one step ahead as 64-bit scalar loads
    constexpr int NE = CW == 1 ? 2 : 1;
    const unsigned long long* ctab64 = reinterpret_cast<const unsigned long long*>(a.ctab) + (CW == 1 ? b_kq * 2 : wave);
    unsigned long long ce[NE], ce_next[NE];
    auto loadT = [&](int k0) {
#pragma unroll
        for (int h = 0; h < NE; ++h) ce_next[h] = ctab64[(k0 >> 3) + h];
    };
    // loadB only ISSUES the loads.  Everything that touches the loaded values -- the row-edge masks and the rare
    // "vector starts in front of the tensor" re-fetch -- runs in fixB() at LDS-store time, behind the MFMAs: applied
    // right after the loads it made hipcc wait for the gather (s_waitcnt vmcnt) BEFORE the matrix work of the step.
    // The table entries ce[] of the tile being stored are still resident then, so offsets / masks are recomputed.
    auto loadB = [&](int h) {
        if constexpr (KWV) {                               // h-th chunk = 8 consecutive w taps of one (ci, dt, dh) row
            const unsigned ex = (unsigned)ce[h], ey = (unsigned)(ce[h] >> 32);
            const unsigned sel = (vmask & ey) == ey ? 0xffffffffu : 0u;
            const unsigned vo = ((voff0 + ex) & sel) | (a.src_bytes & ~sel);
            const Words4 v0 = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
            const Words4 v1 = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16u, 0, 0));
            float* r = rb + 8 * h;
            r[0] = __builtin_bit_cast(float, v0.a); r[1] = __builtin_bit_cast(float, v0.b);
            r[2] = __builtin_bit_cast(float, v0.c); r[3] = __builtin_bit_cast(float, v0.d);
            r[4] = __builtin_bit_cast(float, v1.a); r[5] = __builtin_bit_cast(float, v1.b);
            r[6] = __builtin_bit_cast(float, v1.c); r[7] = __builtin_bit_cast(float, v1.d);
        } else if constexpr (CW == 1) {                    // h-th 8-k chunk of this thread's 16
            const unsigned ex = tab_off(ce[h]), ey = (unsigned)(ce[h] >> 32);
            const unsigned sel = (vmask & ey) == ey ? 0xffffffffu : 0u;
            const unsigned vo = ((voff0 + ex) & sel) | (a.src_bytes & ~sel);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (H) rh[8 * h + i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, vo, i * cs_bytes, 0);
                else rb[8 * h + i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, i * cs_bytes, 0));
            }
        } else if (h == 0) {
            const unsigned ex = tab_off(ce[0]), ey = (unsigned)(ce[0] >> 32);
            const unsigned eth = ey & 0x8000ffffu;                 // t / h bits (+ the never-valid bit of padding entries)
            const unsigned sel = (vmask & eth) == eth ? 0xffffffffu : 0u;
            const unsigned vo = ((voff0 + ex) & sel) | (a.src_bytes & ~sel);
            constexpr int NL = 16 / CW;                            // loads per thread per K step
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                // (the loaded vector is bit-cast to a plain struct: hipcc 7.2 miscompiles element access combined with
                //  bit operations on the builtin's vector result -- elements 1 and 2 come back as element 0)
                if constexpr (H && CW == 4) {
                    const Words2 v = __builtin_bit_cast(Words2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, i * cs_bytes, 0));
                    rh[2 * i] = v.a; rh[2 * i + 1] = v.b;
                } else if constexpr (H) {
                    rh[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, i * cs_bytes, 0);
                } else if constexpr (CW == 4) {
                    const Words4 v = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, i * cs_bytes, 0));
                    rb[4 * i] = __builtin_bit_cast(float, v.a); rb[4 * i + 1] = __builtin_bit_cast(float, v.b);
                    rb[4 * i + 2] = __builtin_bit_cast(float, v.c); rb[4 * i + 3] = __builtin_bit_cast(float, v.d);
                } else {
                    const Words2 v = __builtin_bit_cast(Words2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, i * cs_bytes, 0));
                    rb[2 * i] = __builtin_bit_cast(float, v.a); rb[2 * i + 1] = __builtin_bit_cast(float, v.b);
                }
            }
        }
    };
    auto fixB = [&]() {
        if constexpr (KWV) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned ex = (unsigned)ce[h], ey = (unsigned)(ce[h] >> 32);
                const unsigned sel = (vmask & ey) == ey ? 0xffffffffu : 0u;
                const unsigned vo = ((voff0 + ex) & sel) | (a.src_bytes & ~sel);
                float* r = rb + 8 * h;
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = __builtin_bit_cast(float, __float_as_uint(r[j]) & keepw[j]);
                // first load starting in front of the tensor (first row, left padding): rejected as a whole -> refetch
                const bool neg = vo >= 0xfffffff0u;
                if (__builtin_amdgcn_ballot_w64(neg) != 0ull) {
#pragma unroll
                    for (int j = 1; j < 8; ++j) {          // (the second vector's immediate offset does not wrap either)
                        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, neg ? wrap_add(vo, 4u * j) : a.src_bytes, 0, 0);
                        r[j] = neg ? __builtin_bit_cast(float, v & keepw[j]) : r[j];
                    }
                }
            }
        } else if constexpr (CW > 1 && H) {
            const unsigned ex = tab_off(ce[0]), ey = (unsigned)(ce[0] >> 32);
            const unsigned eth = ey & 0x8000ffffu;
            const unsigned dwb = (ey >> 16) & 0xffu;
            const bool below = (dwb & ((1u << g.pw) - 1u)) != 0u, above = (dwb >> (g.pw + 1)) != 0u;
            const unsigned sneg = (MODE == MODE_FWD ? below : above) ? 0xffffffffu : 0u;
            const unsigned spos = (MODE == MODE_FWD ? above : below) ? 0xffffffffu : 0u;
            const unsigned keepL = ~(keepL_thr & sneg) | 0xffff0000u;      // first position = low half of a load's first word
            const unsigned keepR = ~(keepR_thr & spos) | 0x0000ffffu;      // last position = high half of its last word
            const unsigned sel = (vmask & eth) == eth ? 0xffffffffu : 0u;
            const unsigned vo = ((voff0 + ex) & sel) | (a.src_bytes & ~sel);
            constexpr int NL = 16 / CW, WPL = CW / 2;                     // loads per K step, words per load
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                rh[WPL * i] &= keepL;
                rh[WPL * i + WPL - 1] &= keepR;
            }
            // The bounds check works on DWORDS of the access: a vector that starts 2 bytes in front of the tensor is rejected
            // as a whole, and one whose last dword straddles the tensor's end loses that dword's in-range half (a tap shifted
            // by one element makes the vector 2-byte aligned).  Both happen for a handful of lanes per launch: wave-uniform
            // branch, the elements re-fetched one by one (an element outside the tensor reads 0 by itself).
            // (the end case concerns ONE of the thread's channels: the check adds the load's channel offset i * cs_bytes)
            const bool neg = vo >= 0xfffffff0u;
            const unsigned gap = a.src_bytes + 2u - 2u * CW - vo;          // the load of channel i straddles the end iff gap == i * cs_bytes
            bool any = neg;
#pragma unroll
            for (int i = 0; i < NL; ++i) any = any || (vo < a.src_bytes && gap == (unsigned)(i * cs_bytes));
            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const bool fix = neg || (vo < a.src_bytes && gap == (unsigned)(i * cs_bytes));
#pragma unroll
                    for (int w = 0; w < WPL; ++w) {
                        const unsigned lo = __builtin_amdgcn_raw_buffer_load_b16(rs, fix ? wrap_add(vo, 4u * w) : a.src_bytes, i * cs_bytes, 0);
                        const unsigned hi = __builtin_amdgcn_raw_buffer_load_b16(rs, fix ? wrap_add(vo, 4u * w + 2u) : a.src_bytes, i * cs_bytes, 0);
                        unsigned v = lo | (hi << 16);
                        if (w == 0) v &= keepL;
                        if (w == WPL - 1) v &= keepR;
                        rh[WPL * i + w] = fix ? v : rh[WPL * i + w];
                    }
                }
            }
        } else if constexpr (CW > 1) {
            const unsigned ex = (unsigned)ce[0], ey = (unsigned)(ce[0] >> 32);
            const unsigned eth = ey & 0x8000ffffu;
            const unsigned dwb = (ey >> 16) & 0xffu;               // 1 << dw
            // w shift of this tap relative to the centre: FWD dw - pw, DGRAD pw - dw
            const bool below = (dwb & ((1u << g.pw) - 1u)) != 0u, above = (dwb >> (g.pw + 1)) != 0u;
            const unsigned sneg = (MODE == MODE_FWD ? below : above) ? 0xffffffffu : 0u;
            const unsigned spos = (MODE == MODE_FWD ? above : below) ? 0xffffffffu : 0u;
            const unsigned keepL = ~(keepL_thr & sneg), keepR = ~(keepR_thr & spos);
            const unsigned sel = (vmask & eth) == eth ? 0xffffffffu : 0u;
            const unsigned vo = ((voff0 + ex) & sel) | (a.src_bytes & ~sel);
            constexpr int NL = 16 / CW;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                rb[CW * i] = __builtin_bit_cast(float, __float_as_uint(rb[CW * i]) & keepL);
                rb[CW * i + CW - 1] = __builtin_bit_cast(float, __float_as_uint(rb[CW * i + CW - 1]) & keepR);
            }
            // a vector whose first element lies 4 bytes in front of the tensor (very first row, tap shifted by -1)
            // is rejected as a whole by the bounds check: re-fetch its other elements one by one.  The branch is
            // wave-uniform (ballot) and taken by one wave of the grid; lanes select their own result.
            const bool neg = vo >= 0xfffffff0u;
            if (__builtin_amdgcn_ballot_w64(neg) != 0ull) {
#pragma unroll
                for (int i = 0; i < NL; ++i)
#pragma unroll
                    for (int j = 1; j < CW; ++j) {
                        unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, neg ? wrap_add(vo, 4u * j) : a.src_bytes, i * cs_bytes, 0);
                        if (j == CW - 1) v &= keepR;
                        rb[CW * i + j] = neg ? __builtin_bit_cast(float, v) : rb[CW * i + j];
                    }
            }
        }
    };
    auto prow = [&](int n) { return CW == 1 ? n : (n % CW) * PB + n / CW; };     // LDS row of tile column n
    auto store_tiles = [&](int buf) {
        fixB();
        unsigned short* As = smA[buf];
        unsigned short* Bs = smB[buf];
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j) {
            const int p = tid + NT * j;
            if ((BM * 4) % NT == 0 || p < BM * 4) *reinterpret_cast<u32x4*>(As + (p >> 2) * KP + (p & 3) * 8) = ra[j];
        }
        if constexpr (H && CW == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4 pk;
                pk.x = pair_lo(rh[8 * h], rh[8 * h + 1]); pk.y = pair_lo(rh[8 * h + 2], rh[8 * h + 3]);
                pk.z = pair_lo(rh[8 * h + 4], rh[8 * h + 5]); pk.w = pair_lo(rh[8 * h + 6], rh[8 * h + 7]);
                *reinterpret_cast<uint4*>(Bs + b_q * KP + b_kq * 16 + 8 * h) = pk;
            }
        } else if constexpr (H && CW == 4) {               // rh[k i][word]: word j >> 1 holds position j in its low / high half
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x2 pk;
                pk[0] = (j & 1) ? pair_hi(rh[j >> 1], rh[2 + (j >> 1)]) : pair_lo(rh[j >> 1], rh[2 + (j >> 1)]);
                pk[1] = (j & 1) ? pair_hi(rh[4 + (j >> 1)], rh[6 + (j >> 1)]) : pair_lo(rh[4 + (j >> 1)], rh[6 + (j >> 1)]);
                *reinterpret_cast<u32x2*>(Bs + (j * PB + b_q) * KP + wave * 8 + (lane >> 5) * 4) = pk;
            }
        } else if constexpr (H) {                          // CW 2: rh[k i] = positions (0 | 1 << 16) of channel i
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 pk;
                pk.x = j ? pair_hi(rh[0], rh[1]) : pair_lo(rh[0], rh[1]);
                pk.y = j ? pair_hi(rh[2], rh[3]) : pair_lo(rh[2], rh[3]);
                pk.z = j ? pair_hi(rh[4], rh[5]) : pair_lo(rh[4], rh[5]);
                pk.w = j ? pair_hi(rh[6], rh[7]) : pair_lo(rh[6], rh[7]);
                *reinterpret_cast<uint4*>(Bs + (j * PB + b_q) * KP + wave * 8) = pk;
            }
        } else if constexpr (CW == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4 pk;
                pk.x = cvt_pk_bf16(rb[8 * h], rb[8 * h + 1]);
                pk.y = cvt_pk_bf16(rb[8 * h + 2], rb[8 * h + 3]);
                pk.z = cvt_pk_bf16(rb[8 * h + 4], rb[8 * h + 5]);
                pk.w = cvt_pk_bf16(rb[8 * h + 6], rb[8 * h + 7]);
                *reinterpret_cast<uint4*>(Bs + b_q * KP + b_kq * 16 + 8 * h) = pk;
            }
        } else if constexpr (CW == 4) {                    // rb[k i][pos j]: per position 4 bf16 (k = ksub .. ksub+3)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x2 pk;
                pk[0] = cvt_pk_bf16(rb[j], rb[4 + j]);
                pk[1] = cvt_pk_bf16(rb[8 + j], rb[12 + j]);
                *reinterpret_cast<u32x2*>(Bs + (j * PB + b_q) * KP + wave * 8 + (lane >> 5) * 4) = pk;
            }
        } else {                                           // rb[k i][pos j]: per position 8 bf16
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 pk;
                pk.x = cvt_pk_bf16(rb[j], rb[2 + j]);
                pk.y = cvt_pk_bf16(rb[4 + j], rb[6 + j]);
                pk.z = cvt_pk_bf16(rb[8 + j], rb[10 + j]);
                pk.w = cvt_pk_bf16(rb[12 + j], rb[14 + j]);
                *reinterpret_cast<uint4*>(Bs + (j * PB + b_q) * KP + wave * 8) = pk;
            }
        }
    };

    constexpr int WAVES_N = BN / (32 * WN);
    const int wm0 = (wave / WAVES_N) * (32 * WM);
    const int wn0 = (wave % WAVES_N) * (32 * WN);
    int brow[WN];                                           // LDS element offset of this lane's B rows
#pragma unroll
    for (int j = 0; j < WN; ++j) brow[j] = prow(wn0 + j * 32 + (lane & 31)) * KP + (lane >> 5) * 8;
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    loadT(k_begin);
#pragma unroll
    for (int h = 0; h < NE; ++h) ce[h] = ce_next[h];
    loadT(k_begin + BK);
    if (nk > 0) {
        loadA(k_begin);
        loadB(0);
        loadB(1);
        store_tiles(0);
    }
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        // the table and the packed weights are padded by two K steps, so the prefetch past the last step of a
        // launch is harmless (it reads real or padding entries and its LDS buffer is never consumed)
        const int kn = k_begin + (it + 1) * BK;
#pragma unroll
        for (int h = 0; h < NE; ++h) ce[h] = ce_next[h];   // entries of step it+1 (loaded during step it-1)
        loadT(k_begin + (it + 2) * BK);
        const unsigned short* as = smA[buf];
        const unsigned short* bs = smB[buf];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#ifdef OTAL_DIRECT_ABLATE
            if (c == 0) { if (!(a.flags & 128)) loadA(kn); if (!(a.flags & DBG_NOLOAD)) loadB(0); } else if (!(a.flags & DBG_NOLOAD)) loadB(1);
#else
            if (c == 0) { loadA(kn); loadB(0); } else loadB(1);
#endif
            bf16x8 av[WM], bv[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i)
                av[i] = *reinterpret_cast<const bf16x8*>(as + (wm0 + i * 32 + (lane & 31)) * KP + 16 * c + (lane >> 5) * 8);
#pragma unroll
            for (int j = 0; j < WN; ++j)
                bv[j] = *reinterpret_cast<const bf16x8*>(bs + brow[j] + 16 * c);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        store_tiles(buf ^ 1);
        __syncthreads();
    }
#ifdef OTAL_DIRECT_ABLATE
    if (a.flags & 64) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 1.2345678e30f) a.out[0] = t;
        return;
    }
#endif
    if constexpr (H) {
        if (a.splits == 1) {
            store_acc_h<MODE, WM, WN, BM, BN>(a, acc, m0, n0, wm0, wn0, lane, smem_, NT);
            return;
        }
    }
    store_acc<MODE, WM, WN, BM>(a, acc, m0, n0, wm0, wn0, lane, split, reinterpret_cast<float*>(smA[0]));
}

// =================================================================================================
// Vector WGRAD (bf16 operands, stride-1 SAME convolutions with kw in {1,3}, W even, To*Ho*Wo % 32 == 0):
//   dW[co][(ci,tap)] = sum over positions  dy[co][pos] * x[ci][pos + tap - pad]
// Both MFMA operands want 8 CONSECUTIVE POSITIONS per lane, and positions are contiguous in memory for a fixed
// channel, so each thread owns ONE column (ci, tap) of the tile for the whole launch and walks along the positions
// with wide loads: per K step (32 positions) 16 positions = 2 operand vectors, fetched as CW-position groups
// (CW = 8 / 4 / 2 by the divisibility of W; a group never leaves its row).  Validity is decided per GROUP:
//   t / h : one test of the group's bit mask (from the per-launch position table, wave-uniform -> scalar loads)
//           against the thread's constant tap bits; an invalid group selects the out-of-range offset (reads 0.0)
//   w     : a tap shifted by -1 / +1 invalidates only the FIRST / LAST element of a group that touches the row
//           start / end: two conditional zeroings.
// The generic kernel spent ~65 VALU instructions per MFMA on per-element decode + predicates and stored bf16
// values to LDS one by one; here it is ~7 per MFMA, all LDS stores are 8 / 16 bytes wide, and every global load
// moves 16 bytes per lane (the texture addresser is the limiter: tools/ubench/bufcheck.hip).
// dy rows are fetched as 16-byte pieces (4 positions) with a scalar sample/position offset.
// S2 (w stride 2, kw <= 7 -- Conv3d_1a): 8 output positions read the 16-float window x[2*wo0 - pw + dw + 2j], so a
// thread owns the column PAIR (dw = 2i, 2i+1) of one (ci, dt, dh) row: the window's even elements are the operand
// vector of column 2i, the odd ones of column 2i+1 (columns are padded to 8 per row; EPI_NPAD8 drops the padding).
// Thread = (pair = tid & 63, position group = wave): 4 dwordx4 per K step, validity of the window's ends by compare.
// H: x and dy are STORED as bf16.  A group of CW positions is CW * 2 bytes (b128 / b64 / b32; 2-byte aligned for the taps
// shifted by one element -- legal for raw-buffer loads, tools/ubench/align2probe.hip) and goes to LDS as loaded: positions ARE
// the K axis of both operands here, so there is neither a conversion nor a transposition left.  dy pieces carry 8 positions.
template <int BM, int WM, int WN, int CW, bool S2 = false, bool H = false>
__global__ __launch_bounds__(NT) void conv_wgrad_bf16v_kernel(const ConvArgs a) {
    static_assert(!S2 || (CW == 8 && !H), "pair mode works on groups of 8 output positions of fp32 tensors");
    constexpr int BN = 128, BK = 32, KP = 40;
    constexpr int ESZ = H ? 2 : 4;
    constexpr int A_PIECES = H ? (BM * 4 + NT - 1) / NT : BM / 32;     // 16-byte dy pieces (4 [H: 8] positions) per thread per K step
    constexpr int NG = S2 ? 1 : 16 / CW;                    // position groups per thread per K step
    __shared__ __attribute__((aligned(16))) unsigned short smA[2][BM * KP];
    __shared__ __attribute__((aligned(16))) unsigned short smB[2][BN * KP];

    const ConvGeom& g = a.g;
    const ConvFastDiv& fd = a.fd;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileId tile = xcd_tile(false);
    const int m0 = tile.y * BM;
    const int n0 = tile.x * BN;
    const int split = tile.z;
    const int k_begin = split * a.k_per_split;
    const int k_end = min(a.K, k_begin + a.k_per_split);
    const int nk = (k_end - k_begin) / BK;                  // K and k_per_split are multiples of 32

    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.src_bytes, 0x00020000);
    const auto rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);

    // ---- this thread's B column: n = (ci, dt, dh, dw); columns past N are clamped (their results are never stored)
    const int b_n = S2 ? 2 * (tid & 63) : (tid & (BN - 1));          // S2: first column of this thread's pair
    const int b_kq = __builtin_amdgcn_readfirstlane(S2 ? (tid >> 6) : (tid >> 7));   // position group(s) of the K step
    unsigned coloff, tbits, thrL = 0u, thrR = 0u;
    int dw0 = 0;                                            // S2: dw of the even column
    if constexpr (S2) {
        const int n = min(n0 + b_n, a.N - 2);
        const int row = n >> 3;                             // (ci*kt + dt)*kh + dh
        const uint32_t q1 = fd_div(fd.kh, (uint32_t)row);
        const int dh = row - (int)(q1 * fd.kh.d);
        const uint32_t ci = fd_div(fd.kt, q1);
        const int dt = (int)(q1 - ci * fd.kt.d);
        dw0 = n & 6;
        coloff = (unsigned)(((int64_t)ci * g.x_cs + ((int64_t)dt * g.Hi + dh) * g.Wi + dw0) * 4);
        tbits = (1u << dt) | (1u << (8 + dh));
    } else {
        const int n = min(n0 + b_n, a.N - 1);
        const TapDec t = dec_tap_fd(fd, (uint32_t)n);
        coloff = (unsigned)(((int64_t)t.c * g.x_cs + ((int64_t)t.dt * g.Hi + t.dh) * g.Wi + t.dw) * ESZ);
        tbits = (1u << t.dt) | (1u << (8 + t.dh));
        thrL = t.dw < g.pw ? 0xffffffffu : 0u;              // this tap reads one element to the left / right of the centre
        thrR = t.dw > g.pw ? 0xffffffffu : 0u;
    }
    // ---- dy pieces: piece p = tid + 256 j -> row p >> 3, positions (p & 7) * 4 .. + 3 of the K step (H: row p >> 2, 8 positions)
    unsigned voffA[A_PIECES];
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
        const int p = tid + NT * j;
        if constexpr (H) {
            const int m = min(m0 + (p >> 2), a.M - 1);
            voffA[j] = (unsigned)(((int64_t)m * g.y_cs + (p & 3) * 8) * 2);
        } else {
            const int m = min(m0 + (p >> 3), a.M - 1);
            voffA[j] = (unsigned)(((int64_t)m * g.y_cs + (p & 7) * 4) * 4);
        }
    }

    Words4 ra[A_PIECES];
    float rb[16];                                           // (unused, and removed by the compiler, when H)
    unsigned rh[8];                                         // H: the 16 positions of this thread's column as loaded (8 words)
    // byte offset of a position-table entry: the table is built for 4-byte elements
    auto tab_off = [](unsigned long long e) { return H ? (unsigned)((int)(unsigned)e >> 1) : (unsigned)e; };     // (offsets may be negative)
    // position -> (sample, position in sample): a K step never leaves its sample (P % 32 == 0)
    auto dy_soff = [&](int k0) {
        const int kc = min(k0, a.K - BK);                   // the prefetch past the end re-reads the last step
        const uint32_t b = fd_div(fd.P, (uint32_t)kc);
        return (int)(((int64_t)b * g.y_bs + (kc - (int)(b * fd.P.d))) * ESZ);
    };
    auto loadA = [&](int k0) {
        const int so = dy_soff(k0);
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j)
            if (!H || (BM * 4) % NT == 0 || tid + NT * j < BM * 4)
                ra[j] = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rdy, voffA[j], so, 0));
    };
    // position-table entries of the NEXT K step (64-bit scalar loads, one step ahead)
    const unsigned long long* ptab64 = reinterpret_cast<const unsigned long long*>(a.ptab) + b_kq * NG;
    unsigned long long pe[NG], pe_next[NG];
    // The loaders only ISSUE loads; masks, the rare re-fetch and (S2) the even/odd split are applied by fixB() at
    // LDS-store time, behind the MFMAs -- done right after the loads they made the wave wait for its gather first.
    auto loadB_s2 = [&]() {                                 // 16-float window of this thread's column pair -> rb[0..15] in window order
        const unsigned ex = (unsigned)pe[0], ey = (unsigned)(pe[0] >> 32);
        const unsigned sel = (ey & tbits) == tbits ? 0xffffffffu : 0u;
        const unsigned vo = ((ex + coloff) & sel) | (a.src_bytes & ~sel);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const Words4 v = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 16 * q, 0));
            rb[4 * q] = __builtin_bit_cast(float, v.a); rb[4 * q + 1] = __builtin_bit_cast(float, v.b);
            rb[4 * q + 2] = __builtin_bit_cast(float, v.c); rb[4 * q + 3] = __builtin_bit_cast(float, v.d);
        }
    };
    auto fixB_s2 = [&]() {
        const unsigned ex = (unsigned)pe[0], ey = (unsigned)(pe[0] >> 32);
        const unsigned sel = (ey & tbits) == tbits ? 0xffffffffu : 0u;
        const unsigned vo = ((ex + coloff) & sel) | (a.src_bytes & ~sel);
        const int first = (int)((ey >> 16) & 0xffu) - 16 + dw0;          // x index (w) of window element 0
        const int lo = -first, hi = g.Wi - first;                         // element e is inside the row iff lo <= e < hi
        const bool neg = vo >= 0xffffffc0u;                 // window starting in front of the tensor: refetch element-wise
        if (__builtin_amdgcn_ballot_w64(neg) != 0ull) {
#pragma unroll
            for (int j = 1; j < 16; ++j) {
                const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rx, neg ? wrap_add(vo, 4u * j) : a.src_bytes, 0, 0);
                rb[j] = neg ? __builtin_bit_cast(float, w) : rb[j];
            }
        }
        // only the first / last four elements can fall outside the row (pw <= 3, checked on the host)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rb[j] = j >= lo ? rb[j] : 0.f;
            rb[12 + j] = 12 + j < hi ? rb[12 + j] : 0.f;
        }
    };
    auto loadT = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NG; ++q) pe_next[q] = ptab64[k0 / CW + q];
    };
    auto loadB = [&](int q) {                              // q-th position group of this thread's 16 positions (issue only)
        const unsigned ex = tab_off(pe[q]), ey = (unsigned)(pe[q] >> 32);
        const unsigned sel = (ey & tbits) == tbits ? 0xffffffffu : 0u;
        const unsigned vo = ((ex + coloff) & sel) | (a.src_bytes & ~sel);
        if constexpr (H) {
            unsigned* r = rh + (CW / 2) * q;
            if constexpr (CW == 8) {
                const Words4 v = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0));
                r[0] = v.a; r[1] = v.b; r[2] = v.c; r[3] = v.d;
            } else if constexpr (CW == 4) {
                const Words2 v = __builtin_bit_cast(Words2, __builtin_amdgcn_raw_buffer_load_b64(rx, vo, 0, 0));
                r[0] = v.a; r[1] = v.b;
            } else {
                r[0] = __builtin_amdgcn_raw_buffer_load_b32(rx, vo, 0, 0);
            }
            return;
        }
        float* r = rb + CW * q;
        if constexpr (CW == 8) {
            const Words4 v0 = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0));
            const Words4 v1 = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 16, 0));
            r[0] = __builtin_bit_cast(float, v0.a); r[1] = __builtin_bit_cast(float, v0.b);
            r[2] = __builtin_bit_cast(float, v0.c); r[3] = __builtin_bit_cast(float, v0.d);
            r[4] = __builtin_bit_cast(float, v1.a); r[5] = __builtin_bit_cast(float, v1.b);
            r[6] = __builtin_bit_cast(float, v1.c); r[7] = __builtin_bit_cast(float, v1.d);
        } else if constexpr (CW == 4) {
            const Words4 v0 = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0));
            r[0] = __builtin_bit_cast(float, v0.a); r[1] = __builtin_bit_cast(float, v0.b);
            r[2] = __builtin_bit_cast(float, v0.c); r[3] = __builtin_bit_cast(float, v0.d);
        } else {
            const Words2 v0 = __builtin_bit_cast(Words2, __builtin_amdgcn_raw_buffer_load_b64(rx, vo, 0, 0));
            r[0] = __builtin_bit_cast(float, v0.a); r[1] = __builtin_bit_cast(float, v0.b);
        }
    };
    auto fixB = [&]() {
        if constexpr (S2) { fixB_s2(); return; }
        if constexpr (H) {
            constexpr int WPG = CW / 2;                    // words per group
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const unsigned ex = tab_off(pe[q]), ey = (unsigned)(pe[q] >> 32);
                const unsigned sel = (ey & tbits) == tbits ? 0xffffffffu : 0u;
                const unsigned vo = ((ex + coloff) & sel) | (a.src_bytes & ~sel);
                const unsigned eL = (ey & 0x10000u) ? 0xffffffffu : 0u, eR = (ey & 0x20000u) ? 0xffffffffu : 0u;
                const unsigned keepL = ~(thrL & eL) | 0xffff0000u, keepR = ~(thrR & eR) | 0x0000ffffu;
                unsigned* r = rh + WPG * q;
                r[0] &= keepL;
                r[WPG - 1] &= keepR;
                // dword-granular bounds check (see conv_gemm_bf16c_kernel): a group that starts in front of the tensor is
                // rejected as a whole, a 2-byte aligned one whose last dword straddles the end loses that dword's in-range half
                const bool fix = vo >= 0xfffffff0u || (vo < a.src_bytes && vo + 2u * CW == a.src_bytes + 2u);
                if (__builtin_amdgcn_ballot_w64(fix) != 0ull) {
#pragma unroll
                    for (int w = 0; w < WPG; ++w) {
                        const unsigned lo = __builtin_amdgcn_raw_buffer_load_b16(rx, fix ? wrap_add(vo, 4u * w) : a.src_bytes, 0, 0);
                        const unsigned hi = __builtin_amdgcn_raw_buffer_load_b16(rx, fix ? wrap_add(vo, 4u * w + 2u) : a.src_bytes, 0, 0);
                        unsigned v = lo | (hi << 16);
                        if (w == 0) v &= keepL;
                        if (w == WPG - 1) v &= keepR;
                        r[w] = fix ? v : r[w];
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const unsigned ex = (unsigned)pe[q], ey = (unsigned)(pe[q] >> 32);
            const unsigned sel = (ey & tbits) == tbits ? 0xffffffffu : 0u;
            const unsigned vo = ((ex + coloff) & sel) | (a.src_bytes & ~sel);
            const unsigned eL = (ey & 0x10000u) ? 0xffffffffu : 0u, eR = (ey & 0x20000u) ? 0xffffffffu : 0u;
            const unsigned keepL = ~(thrL & eL), keepR = ~(thrR & eR);
            float* r = rb + CW * q;
            r[0] = __builtin_bit_cast(float, __float_as_uint(r[0]) & keepL);
            r[CW - 1] = __builtin_bit_cast(float, __float_as_uint(r[CW - 1]) & keepR);
            // a group that starts 4 bytes in front of the tensor (channel 0, first row, tap shifted by -1) is rejected as a
            // whole by the bounds check: re-fetch its other elements one by one (wave-uniform branch, a few waves per launch)
            const bool neg = vo >= 0xfffffff0u;
            if (__builtin_amdgcn_ballot_w64(neg) != 0ull) {
#pragma unroll
                for (int j = 1; j < CW; ++j) {             // (the check does not wrap: voffset 0xfffffffc + 16 is out of range too)
                    unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rx, neg ? wrap_add(vo, 4u * j) : a.src_bytes, 0, 0);
                    if (j == CW - 1) v &= keepR;
                    r[j] = neg ? __builtin_bit_cast(float, v) : r[j];
                }
            }
        }
    };
    auto store_tiles = [&](int buf) {
        fixB();
        unsigned short* As = smA[buf];
        unsigned short* Bs = smB[buf];
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j) {
            const int p = tid + NT * j;
            if constexpr (H) {
                if ((BM * 4) % NT == 0 || p < BM * 4) *reinterpret_cast<Words4*>(As + (p >> 2) * KP + (p & 3) * 8) = ra[j];
            } else {
                Words2 pk;
                pk.a = cvt_pk_bf16(__builtin_bit_cast(float, ra[j].a), __builtin_bit_cast(float, ra[j].b));
                pk.b = cvt_pk_bf16(__builtin_bit_cast(float, ra[j].c), __builtin_bit_cast(float, ra[j].d));
                *reinterpret_cast<Words2*>(As + (p >> 3) * KP + (p & 7) * 4) = pk;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 pk;
            if constexpr (H) {
                pk.x = rh[4 * h]; pk.y = rh[4 * h + 1]; pk.z = rh[4 * h + 2]; pk.w = rh[4 * h + 3];
                *reinterpret_cast<uint4*>(Bs + b_n * KP + b_kq * 16 + 8 * h) = pk;
            } else if constexpr (S2) {     // window order in rb: even elements are column dw0's vector, odd ones column dw0+1's
                pk.x = cvt_pk_bf16(rb[h], rb[2 + h]);
                pk.y = cvt_pk_bf16(rb[4 + h], rb[6 + h]);
                pk.z = cvt_pk_bf16(rb[8 + h], rb[10 + h]);
                pk.w = cvt_pk_bf16(rb[12 + h], rb[14 + h]);
                *reinterpret_cast<uint4*>(Bs + (b_n + h) * KP + b_kq * 8) = pk;     // rows = the two columns
            } else {
                pk.x = cvt_pk_bf16(rb[8 * h], rb[8 * h + 1]);
                pk.y = cvt_pk_bf16(rb[8 * h + 2], rb[8 * h + 3]);
                pk.z = cvt_pk_bf16(rb[8 * h + 4], rb[8 * h + 5]);
                pk.w = cvt_pk_bf16(rb[8 * h + 6], rb[8 * h + 7]);
                *reinterpret_cast<uint4*>(Bs + b_n * KP + b_kq * 16 + 8 * h) = pk;
            }
        }
    };

    constexpr int WAVES_N = BN / (32 * WN);
    const int wm0 = (wave / WAVES_N) * (32 * WM);
    const int wn0 = (wave % WAVES_N) * (32 * WN);
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    loadT(k_begin);
#pragma unroll
    for (int q = 0; q < NG; ++q) pe[q] = pe_next[q];
    loadT(k_begin + BK);
    if (nk > 0) {
        loadA(k_begin);
        if constexpr (S2) loadB_s2();
        else {
#pragma unroll
            for (int q = 0; q < NG; ++q) loadB(q);
        }
        store_tiles(0);
    }
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        const int kn = k_begin + (it + 1) * BK;
#pragma unroll
        for (int q = 0; q < NG; ++q) pe[q] = pe_next[q];   // entries of step it+1 (loaded during step it-1; table is padded)
        loadT(k_begin + (it + 2) * BK);
        const unsigned short* as = smA[buf];
        const unsigned short* bs = smB[buf];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c == 0) loadA(kn);
            if constexpr (S2) {
                if (c == 0) loadB_s2();
            } else {
#pragma unroll
                for (int q = 0; q < NG / 2; ++q) loadB(c * (NG / 2) + q);
            }
            bf16x8 av[WM], bv[WN];
            const int ko = 16 * c + (lane >> 5) * 8;
#pragma unroll
            for (int i = 0; i < WM; ++i)
                av[i] = *reinterpret_cast<const bf16x8*>(as + (wm0 + i * 32 + (lane & 31)) * KP + ko);
#pragma unroll
            for (int j = 0; j < WN; ++j)
                bv[j] = *reinterpret_cast<const bf16x8*>(bs + (wn0 + j * 32 + (lane & 31)) * KP + ko);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        store_tiles(buf ^ 1);
        __syncthreads();
    }
    store_acc<MODE_WGRAD, WM, WN, BM>(a, acc, m0, n0, wm0, wn0, lane, split, reinterpret_cast<float*>(smA[0]));
}

// position table of the vector WGRAD: one entry per group of CW consecutive output positions
__global__ __launch_bounds__(256) void build_pos_table_kernel(int2* __restrict__ tab, ConvGeom g, ConvFastDiv fd, int CW,
                                                              int ngroups, int npad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    int2 e;
    e.x = 0; e.y = 0;                                        // padding: no tap bit set -> never valid
    if (i < ngroups) {
        const PosDec o = dec_pos_fd((uint32_t)i * CW, fd.To, fd.Ho, fd.Wo);
        const int t0 = o.t * g.st - g.pt, h0 = o.h * g.sh - g.ph, w0 = o.w * g.sw - g.pw;
        unsigned m = 0;
        for (int d = 0; d < 8; ++d) {
            m |= (unsigned)(d < g.kt && t0 + d >= 0 && t0 + d < g.Ti) << d;
            m |= (unsigned)(d < g.kh && h0 + d >= 0 && h0 + d < g.Hi) << (8 + d);
        }
        if (g.sw == 2) {
            m |= ((unsigned)(w0 + 16) & 0xffu) << 16;       // pair mode: x index of the window origin (+16)
        } else {
            if (o.w == 0) m |= 0x10000u;
            if (o.w + CW == g.Wo) m |= 0x20000u;
        }
        e.x = (int)(unsigned)(((int64_t)o.b * g.x_bs + ((int64_t)t0 * g.Hi + h0) * g.Wi + w0) * 4);
        e.y = (int)m;
    }
    tab[i] = e;
}

// prologue of the chunked path: chunk table + bf16 weight pack in one launch.
//   wsrc: FWD W (M=Cout, C=Cin, kvol);  DGRAD packed W^T (M=Cin, C=Cout, kvol)  -> wp[m][tap * C + c]
// One prologue = one PrepDesc; the batch kernel (blockIdx.y = layer) refreshes the persistent regions of every known
// layer in ONE launch at the start of a training step (otal_conv_prologue_batch).
struct PrepDesc {
    int2* ctab;
    unsigned* wp;
    const float* wsrc;
    ConvGeom g;
    int M, Mpad, C, kvol, K, Kp, nchunk, kwv, natural, mode;
    int direct;            // 1: the weight pack of conv3_direct_kernel ([Mpad][C/16][3][9][16] bf16) instead of a chunk-path prologue
    FastDiv fh, fk;        // host-built exact division by the bf16 pairs per packed row and by kvol (a runtime division is
                           // ~40 VALU instructions per packed pair: most of the pack kernels' time)
};

typedef __attribute__((address_space(1))) float GlobalF32;
typedef __attribute__((address_space(1))) unsigned GlobalU32;
typedef __attribute__((address_space(1))) unsigned long long GlobalU64;
constexpr int PREP_U = 4;       // packed pairs per thread and trip of the pack loops
constexpr int PREP_T_KVOL = 3;  // data-gradient packs from the natural W with kvol <= 3: transposed through LDS (prep_chunks_body)
constexpr int PREP_T_PITCH = 256 * 2 + 4;   // bytes of a tile row in LDS: 129 words, the 32 rows of a 2-byte column write on 32 banks

template <int MODE>
__device__ __forceinline__ void prep_chunks_body(const PrepDesc& d, unsigned bid, unsigned nblk) {
    // (the descriptor is read from memory in the batch kernel: without the explicit address space these become flat accesses)
    GlobalU64* __restrict__ ctab = (GlobalU64*)d.ctab;              // int2 entries {x, y} written as one 8-byte word
    GlobalU32* __restrict__ wp = (GlobalU32*)d.wp;
    const GlobalF32* __restrict__ wsrc = (const GlobalF32*)d.wsrc;
    const ConvGeom& g = d.g;
    const int M = d.M, Mpad = d.Mpad, C = d.C, kvol = d.kvol, K = d.K, Kp = d.Kp, nchunk = d.nchunk, kwv = d.kwv, natural = d.natural;
    const int64_t gid = (int64_t)bid * 256 + threadIdx.x;
    if (kwv) {      // forward only: chunk j = (ci, dt, dh) row, k = j * 8 + dw (dw >= kw: zero weight)
        if (gid < nchunk) {
            const int j = (int)gid;
            int2 e;
            e.x = 0; e.y = -1;
            if (j * 8 < K) {
                const int ci = j / (g.kt * g.kh), r = j - ci * (g.kt * g.kh), dt = r / g.kh, dh = r - dt * g.kh;
                e.x = (int)(unsigned)(((int64_t)ci * g.x_cs + ((int64_t)dt * g.Hi + dh) * g.Wi) * 4);
                e.y = (1 << dt) | (1 << (8 + dh));
            }
            ctab[gid] = (unsigned long long)(unsigned)e.x | ((unsigned long long)(unsigned)e.y << 32);
        }
        const unsigned half = (unsigned)Kp / 2;
        const unsigned pairs = (unsigned)Mpad * half;           // < 2^31 (checked by the launcher): 32-bit index math --
        const FastDiv fh = d.fh;
        for (unsigned p = (unsigned)gid; p < pairs; p += nblk * 256u) {
            const int m = (int)fd_div(fh, p), k = (int)(p - (unsigned)m * half) * 2;
            float v[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kk = k + i, j = kk >> 3, dw = kk & 7;
                v[i] = (m < M && kk < K && dw < g.kw) ? wsrc[((int64_t)m * C * g.kt * g.kh + j) * g.kw + dw] : 0.f;
            }
            wp[p] = cvt_pk_bf16(v[0], v[1]);
        }
        return;
    }
    if (gid < nchunk) {
        const int k0 = (int)gid * 8;
        int2 e;
        e.x = 0; e.y = -1;
        if (k0 < K) {
            const int j = k0 >> 3, cb = j / kvol, tap = j - cb * kvol, c0 = cb * 8;      // k = ((c/8) * kvol + tap) * 8 + c % 8
            const int khw = g.kh * g.kw;
            const int dt = tap / khw, r = tap - dt * khw, dh = r / g.kw, dw = r - dh * g.kw;
            int64_t off;
            if (MODE == MODE_FWD) off = (int64_t)c0 * g.x_cs + ((int64_t)dt * g.Hi + dh) * g.Wi + dw;
            else off = (int64_t)c0 * g.y_cs -
                       (((int64_t)(dt >> (g.st - 1)) * g.Ho + (dh >> (g.sh - 1))) * g.Wo + (dw >> (g.sw - 1)));
            e.x = (int)(unsigned)(off * 4);
            e.y = (1 << dt) | (1 << (8 + dh)) | (1 << (16 + dw));
        }
        ctab[gid] = (unsigned long long)(unsigned)e.x | ((unsigned long long)(unsigned)e.y << 32);
    }
    if (MODE == MODE_DGRAD && natural && kvol <= PREP_T_KVOL) {
        // The data gradient's operand rows are the INPUT channels of W (Cout, Cin, kvol) taken as it lies: A[m = ci][k <-> (co, tap)].
        // Walking k along a packed row, as the loop below does, reads W with a stride of Cin * kvol floats -- one 64-byte sector
        // per 4-byte element for the 1x1x1 and 1-D layers (16 M parameters per step: the pack launch fetched 774 MB for 179 MB of
        // weights).  Here a workgroup takes a 32-row x 256-k tile: 32 consecutive m of one (co, tap) are 32 * kvol contiguous
        // floats (lanes along m), the tile is transposed in LDS and leaves as 512-byte runs of the packed rows.
        __shared__ __attribute__((aligned(16))) unsigned char tl[32 * PREP_T_PITCH];
        const int tk = (Kp + 255) / 256, tiles = (Mpad / 32) * tk;
        const int l = threadIdx.x & 31, kq = threadIdx.x >> 5;
        const FastDiv fk = d.fk;
        for (int t = (int)bid; t < tiles; t += (int)nblk) {
            const int m0 = (t / tk) * 32, k0 = (t - (t / tk) * tk) * 256;
            const int m = m0 + l;
#pragma unroll
            for (int i0 = 0; i0 < 32; i0 += 8) {
                unsigned raw[8];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = k0 + kq + 8 * (i0 + u);
                    const int j = kk >> 3, cb = (int)fd_div(fk, (unsigned)j), tap = j - cb * kvol, c = cb * 8 + (kk & 7);
                    ok[u] = m < M && kk < K;
                    raw[u] = __float_as_uint(wsrc[ok[u] ? ((int64_t)c * M + m) * kvol + tap : 0]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned pk = cvt_pk_bf16(__uint_as_float(ok[u] ? raw[u] : 0u), 0.f);
                    *reinterpret_cast<unsigned short*>(tl + l * PREP_T_PITCH + (kq + 8 * (i0 + u)) * 2) = (unsigned short)pk;
                }
            }
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int r = ps * 8 + (threadIdx.x >> 5), seg = threadIdx.x & 31;
                const unsigned* src = reinterpret_cast<const unsigned*>(tl + r * PREP_T_PITCH + seg * 16);
                const unsigned w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
                const int kk = k0 + seg * 8;
                if (kk < Kp) {
                    GlobalU32* dst = wp + ((int64_t)(m0 + r) * Kp + kk) / 2;
                    dst[0] = w0; dst[1] = w1; dst[2] = w2; dst[3] = w3;
                }
            }
            __syncthreads();
        }
        return;
    }
    const unsigned half = (unsigned)Kp / 2;
    const unsigned pairs = (unsigned)Mpad * half;
    const FastDiv fh = d.fh, fk = d.fk;
    // PREP_U pairs per thread and trip (prep_blocks sizes the grid for it), their 2 * PREP_U loads issued together from clamped
    // indices and masked afterwards: one pair per thread made ~50 k workgroups per step whose whole life was a chain of
    // dependent fetches (layer search, descriptor, two conditional loads each behind its own branch and wait) -- 153 us for 150 MB
    const unsigned stride = nblk * 256u;
    for (unsigned p0 = (unsigned)gid; p0 < pairs; p0 += PREP_U * stride) {
        unsigned raw[PREP_U][2], keep[PREP_U][2];
#pragma unroll
        for (int u = 0; u < PREP_U; ++u) {
            const unsigned p = p0 + u * stride;
            const bool live = p < pairs;
            const int m = (int)fd_div(fh, live ? p : 0u), k = (int)((live ? p : 0u) - (unsigned)m * half) * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kk = k + i;
                const int j = kk >> 3, cb = (int)fd_div(fk, (unsigned)j), tap = j - cb * kvol, c = cb * 8 + (kk & 7);
                // source layout: [m][c][tap] (forward W, or the packed W^T of DGRAD) / natural W seen from DGRAD: [c][m][tap]
                const int64_t si = natural ? ((int64_t)c * M + m) : ((int64_t)m * C + c);
                const bool ok = live && m < M && kk < K;
                keep[u][i] = ok ? ~0u : 0u;
                raw[u][i] = __float_as_uint(wsrc[ok ? si * kvol + tap : 0]);
            }
        }
#pragma unroll
        for (int u = 0; u < PREP_U; ++u) {
            const unsigned p = p0 + u * stride;
            if (p < pairs) wp[p] = cvt_pk_bf16(__uint_as_float(raw[u][0] & keep[u][0]), __uint_as_float(raw[u][1] & keep[u][1]));
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void prep_chunks_kernel(const PrepDesc d) { prep_chunks_body<MODE>(d, blockIdx.x, gridDim.x); }

// weights -> [Mpad][cb][dt][dh*3+dw][16] bf16 for conv3_direct_kernel.  FWD: A[m][..] = w[m][cb*16+c][dt][dh][dw];
// DGRAD (natural layout W (Cout, Cin, 27)): A[m = ci][..] = w[co = cb*16+c][ci][2-dt][2-dh][2-dw]
template <int MODE>
__device__ __forceinline__ void pack_direct_body(unsigned* __restrict__ wp_, const float* __restrict__ w_, int M, int Mpad, int C, int natural,
                                                 unsigned bid, unsigned nblk, const FastDiv fh) {
    const int Ktot = C * 27;
    const unsigned hk = (unsigned)Ktot / 2, pairs = (unsigned)Mpad * hk;       // < 2^31: 32-bit index math
    const GlobalF32* __restrict__ w = (const GlobalF32*)w_;
    GlobalU32* __restrict__ wp = (GlobalU32*)wp_;
    const unsigned stride = nblk * 256u;
    for (unsigned p0 = bid * 256u + threadIdx.x; p0 < pairs; p0 += PREP_U * stride) {      // (see prep_chunks_body: loads first, masks after)
        unsigned raw[PREP_U][2], keep[PREP_U];
#pragma unroll
        for (int u = 0; u < PREP_U; ++u) {
            const unsigned p = p0 + u * stride;
            const bool live = p < pairs;
            const int m = (int)fd_div(fh, live ? p : 0u), k = (int)((live ? p : 0u) - (unsigned)m * hk) * 2;
            const bool ok = live && m < M;
            keep[u] = ok ? ~0u : 0u;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kk = k + i;
                const int c16 = kk & 15, g9 = (kk >> 4) % 9, sdt = (kk >> 4) / 9, dt = sdt % 3, cb = sdt / 3;
                const int c = cb * 16 + c16;
                int64_t si;
                if (MODE == MODE_FWD) si = ((int64_t)m * C + c) * 27 + dt * 9 + g9;
#ifdef OTAL_BREAK_DGRAD_TAP     // tools/break_dgrad_tap.sh: a deliberately mis-routed data gradient (temporal taps not flipped), to show
                                // that tests/test_bf16_parity_gpu.py's backward pin fails on it.  Never defined in the product build.
                else if (natural) si = ((int64_t)c * M + m) * 27 + dt * 9 + (8 - g9);
#else
                else if (natural) si = ((int64_t)c * M + m) * 27 + (2 - dt) * 9 + (8 - g9);
#endif
                else si = ((int64_t)m * C + c) * 27 + (2 - dt) * 9 + (8 - g9);      // packed W^T (Cin, Cout, 27)
                raw[u][i] = __float_as_uint(w[ok ? si : 0]);
            }
        }
#pragma unroll
        for (int u = 0; u < PREP_U; ++u) {
            const unsigned p = p0 + u * stride;
            if (p < pairs) wp[p] = cvt_pk_bf16(__uint_as_float(raw[u][0] & keep[u]), __uint_as_float(raw[u][1] & keep[u]));
        }
    }
}

// 1-D grid; starts[l] .. starts[l+1] are the workgroups of layer l (each layer gets exactly what its own launch would use)
__global__ __launch_bounds__(256) void prep_chunks_batch_kernel(const PrepDesc* __restrict__ descs, const int* __restrict__ starts, int n) {
    const int b = blockIdx.x;
    int lo = 0, hi = n;                     // largest l with starts[l] <= b
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (starts[mid] <= b) lo = mid; else hi = mid;
    }
    const PrepDesc& d = descs[lo];
    const unsigned bid = (unsigned)(b - starts[lo]), nblk = (unsigned)(starts[lo + 1] - starts[lo]);
    if (d.direct) {
        if (d.mode == MODE_FWD) pack_direct_body<MODE_FWD>(d.wp, d.wsrc, d.M, d.Mpad, d.C, d.natural, bid, nblk, d.fh);
        else pack_direct_body<MODE_DGRAD>(d.wp, d.wsrc, d.M, d.Mpad, d.C, d.natural, bid, nblk, d.fh);
        return;
    }
    if (d.mode == MODE_FWD) prep_chunks_body<MODE_FWD>(d, bid, nblk);
    else prep_chunks_body<MODE_DGRAD>(d, bid, nblk);
}

// tap table: one entry per GEMM k of a FWD / DGRAD launch (plus padding entries that never match)
template <int MODE>
__global__ __launch_bounds__(256) void build_tap_table_kernel(int2* __restrict__ tab, ConvGeom g, int K, int Kpad) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= Kpad) return;
    int2 e;
    if (k >= K) { e.x = 0; e.y = -1; tab[k] = e; return; }
    const TapDec t = dec_tap(g, k);
    if (MODE == MODE_FWD) {
        e.x = (int)((int64_t)t.c * g.x_cs + ((int64_t)t.dt * g.Hi + t.dh) * g.Wi + t.dw);
    } else {
        e.x = (int)((int64_t)t.c * g.y_cs -
                    (((int64_t)(t.dt >> (g.st - 1)) * g.Ho + (t.dh >> (g.sh - 1))) * g.Wo + (t.dw >> (g.sw - 1))));
    }
    e.y = (1 << t.dt) | (1 << (8 + t.dh)) | (1 << (16 + t.dw));
    tab[k] = e;
}

constexpr size_t TAB_PAD = 64;      // entries readable past K (a K step may run up to BK-1 rows over)
static inline size_t tab_bytes(int K) { return (((size_t)K + TAB_PAD) * sizeof(int2) + 255) & ~(size_t)255; }

// fixed-order reduction of the split-K slabs + the same epilogue.  V consecutive elements per thread (float4 when the
// slab size allows); the loads of 8 splits are issued together and THEN added in split order -- the sum order (and so the
// result) is the one of a plain loop, without its one-load-in-flight dependency chain.
// H (FWD / DGRAD): the output, and the data gradient's ReLU mask, are bf16 tensors (see store_acc_h).
template <bool H>
__device__ __forceinline__ bool relu_mask_at(const ConvArgs& a, int64_t off) {
    if constexpr (H) return bf16_pos(reinterpret_cast<const unsigned short*>(a.emask)[off]);
    else return a.emask[off] > 0.f;
}
template <bool H>
__device__ __forceinline__ void reduce_store(const ConvArgs& a, int64_t off, float v) {
    if constexpr (H) reinterpret_cast<unsigned short*>(a.out)[off] = (unsigned short)(cvt_pk_bf16(v, 0.f) & 0xffffu);
    else {
        if (a.flags & EPI_ACCUM) v += a.out[off];
        a.out[off] = v;
    }
}

template <int MODE, int V, bool H = false>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvArgs a) {
    const ConvGeom& g = a.g;
    const int64_t total = (int64_t)a.M * a.N;
    struct alignas(4 * V) vec_t { float f[V]; };
    for (int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; base < total; base += (int64_t)gridDim.x * 256 * V) {
        // ---- where the V results go and the epilogue operands they need -- issued BEFORE the slab loads and not touched until
        // the sums exist: these launches are a few microseconds long, and an operand fetched where it is used (scale, shift,
        // ReLU mask: up to three dependent loads per element behind the slab loads) is a memory round trip each
        int mm[V], nn[V];
        int64_t off[V];
        float sc[V], sh[V];
        unsigned mraw[V];
        {
            const int m0 = (int)(base / a.N);
            int m = m0, n = (int)(base - (int64_t)m0 * a.N);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                mm[e] = m; nn[e] = n;
                if (++n == a.N) { n = 0; ++m; }
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int m = mm[e], n = nn[e];
            sc[e] = 1.f; sh[e] = 0.f; mraw[e] = 0u;
            if (MODE == MODE_FWD) {
                off[e] = conv_out_offset(g, dec_pos_fd(n, a.fd.To, a.fd.Ho, a.fd.Wo), m);
                if (a.scale) sc[e] = a.scale[m];
                if (a.shift) sh[e] = a.shift[m];
            } else if (MODE == MODE_DGRAD) {
                off[e] = conv_in_offset(g, dec_pos_fd(n, a.fd.Ti, a.fd.Hi, a.fd.Wi), m);
                if (a.emask) {
                    sc[e] = a.escale[m];
                    if constexpr (H) mraw[e] = reinterpret_cast<const unsigned short*>(a.emask)[off[e]];
                    else mraw[e] = __float_as_uint(a.emask[off[e]]);
                }
            } else if (a.flags & EPI_NPAD8) {
                off[e] = (int64_t)m * ((a.N >> 3) * g.kw) + (n >> 3) * g.kw + (n & 7);
            } else {
                off[e] = base + e;
            }
        }
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        auto add8 = [&](int s0) __attribute__((always_inline)) {
            vec_t tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sidx = min(s0 + u, a.splits - 1);          // clamped: the surplus loads are not added
                tmp[u] = *reinterpret_cast<const vec_t*>(a.slab + (int64_t)sidx * total + base);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < a.splits) {
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += tmp[u].f[e];
                }
        };
        add8(0);                                                     // peeled: no loop header (= full wait) between the operand loads and these
        for (int s0 = 8; s0 < a.splits; s0 += 8) add8(s0);
#pragma unroll
        for (int e = 0; e < V; ++e) asm volatile("" : "+v"(sc[e]), "+v"(sh[e]), "+v"(mraw[e]));   // keeps the compiler from moving the operands' first use (a compare) up to the loads
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = acc[e];
            if (MODE == MODE_FWD) {
                if (a.scale) v *= sc[e];
                if (a.shift) v += sh[e];
                if (a.flags & EPI_RELU) v = fmaxf(v, 0.f);
            } else if (MODE == MODE_DGRAD) {
                if (a.emask) {
                    const bool pos = H ? bf16_pos((unsigned short)mraw[e]) : __uint_as_float(mraw[e]) > 0.f;
                    v = pos ? v * sc[e] : 0.f;
                }
            } else if (a.flags & EPI_NPAD8) {
                if ((nn[e] & 7) >= g.kw) continue;
            }
            reduce_store<H>(a, off[e], v);
        }
    }
}

// The same reduction for MANY splits of a SMALL result (weight gradients: 4 k .. 600 k outputs, up to 2048 slabs): one
// thread walking 500 slabs is a 10 us latency chain.  Here a workgroup owns 64 float4 outputs; its four waves sum a quarter
// of the slabs each (8 loads in flight, in slab order) and the quarters meet in LDS in a fixed order ((q0 + q1) + (q2 + q3)),
// so the result is deterministic -- a different, equally valid summation tree than the sequential kernel's.
template <int MODE, bool H = false>
__global__ __launch_bounds__(256) void splitk_reduce_tall_kernel(const ConvArgs a) {
    __shared__ float4 part[3][64];
    const ConvGeom& g = a.g;
    const int64_t total = (int64_t)a.M * a.N;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t base = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const bool live = base < total;
    const int per = (a.splits + 3) / 4;
    const int s_lo = q * per, s_hi = min(a.splits, s_lo + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int s0 = s_lo; s0 < s_hi; s0 += 8) {
            float4 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sidx = min(s0 + u, s_hi - 1);
                tmp[u] = *reinterpret_cast<const float4*>(a.slab + (int64_t)sidx * total + base);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < s_hi) { acc.x += tmp[u].x; acc.y += tmp[u].y; acc.z += tmp[u].z; acc.w += tmp[u].w; }
        }
    }
    if (q > 0) part[q - 1][lane] = acc;
    __syncthreads();
    if (q > 0 || !live) return;
    const float4 p1 = part[0][lane], p2 = part[1][lane], p3 = part[2][lane];
    float r[4] = {(acc.x + p1.x) + (p2.x + p3.x), (acc.y + p1.y) + (p2.y + p3.y), (acc.z + p1.z) + (p2.z + p3.z), (acc.w + p1.w) + (p2.w + p3.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t idx = base + e;
        const int m = (int)(idx / a.N), n = (int)(idx - (int64_t)m * a.N);
        float v = r[e];
        int64_t off;
        if (MODE == MODE_FWD) {
            if (a.scale) v *= a.scale[m];
            if (a.shift) v += a.shift[m];
            if (a.flags & EPI_RELU) v = fmaxf(v, 0.f);
            off = conv_out_offset(g, dec_pos_fd(n, a.fd.To, a.fd.Ho, a.fd.Wo), m);
        } else if (MODE == MODE_DGRAD) {
            off = conv_in_offset(g, dec_pos_fd(n, a.fd.Ti, a.fd.Hi, a.fd.Wi), m);
            if (a.emask) v = relu_mask_at<H>(a, off) ? v * a.escale[m] : 0.f;
        } else if (a.flags & EPI_NPAD8) {
            if ((n & 7) >= g.kw) continue;
            off = (int64_t)m * ((a.N >> 3) * g.kw) + (n >> 3) * g.kw + (n & 7);
        } else {
            off = idx;
        }
        reduce_store<H>(a, off, v);
    }
}

// ---- deferred weight-gradient reductions.  A weight gradient is consumed by nobody before the optimizer (or the gradient
// all-reduce), so its split-K reduce need not follow its GEMM: between otal_conv_defer_reduces(1) and
// otal_conv_flush_reduces() the weight-gradient launches leave their slabs in the caller's workspace (the caller hands every
// later launch the workspace BEHIND otal_conv_deferred_end()) and up to DEFER_MAX reductions run as ONE launch -- the six
// weight gradients of an Inception module, or the GroupNorm blocks between two bucket flushes of the trainer.  Same
// summation tree as splitk_reduce_tall_kernel (four quarter sums in slab order, combined (q0+q1)+(q2+q3)): deterministic.
// Process-global state, used from the one thread that issues the training step.
}  // namespace
namespace otal_conv {
constexpr int DEFER_MAX = 24;
struct DeferItem { const float* slab; float* out; int64_t total; int splits, flags, N, kw; };
struct DeferBatch { DeferItem it[DEFER_MAX]; int start[DEFER_MAX + 1]; };    // start[i]: first workgroup of item i (1-D grid: no empty workgroups)
// (the record is process-wide: calls that touch it are serialised by `mu`, so a binding that issues from several host threads
//  cannot corrupt it -- but it stays ONE list: defer / flush belong to one logical issuer at a time, see the header)
struct DeferState { bool on = false; int n = 0; DeferBatch batch; uintptr_t last_end = 0; std::recursive_mutex mu; };
__attribute__((visibility("hidden"))) extern DeferState g_defer;       // ONE list for both parts of this file (defined in part 0)
}  // namespace otal_conv
#if OTAL_CONV_PART == 0
otal_conv::DeferState otal_conv::g_defer;
#endif
namespace {
using otal_conv::DEFER_MAX;
using otal_conv::DeferItem;
using otal_conv::DeferBatch;
using otal_conv::g_defer;

__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(const DeferBatch b) {
    __shared__ float4 part[3][64];
    int item = 0;
#pragma unroll 1
    while (item + 1 < DEFER_MAX && b.start[item + 1] <= (int)blockIdx.x) ++item;     // (uniform; start[] is non-decreasing, unused tail = grid size)
    const DeferItem& d = b.it[item];
    const int wg = (int)blockIdx.x - b.start[item];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t base = ((int64_t)wg * 64 + lane) * 4;
    const bool live = base < d.total;
    const int per = (d.splits + 3) / 4;
    const int s_lo = q * per, s_hi = min(d.splits, s_lo + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int s0 = s_lo; s0 < s_hi; s0 += 8) {
            float4 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sidx = min(s0 + u, s_hi - 1);
                tmp[u] = *reinterpret_cast<const float4*>(d.slab + (int64_t)sidx * d.total + base);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < s_hi) { acc.x += tmp[u].x; acc.y += tmp[u].y; acc.z += tmp[u].z; acc.w += tmp[u].w; }
        }
    }
    if (q > 0) part[q - 1][lane] = acc;
    __syncthreads();
    if (q > 0 || !live) return;
    const float4 p1 = part[0][lane], p2 = part[1][lane], p3 = part[2][lane];
    float r[4] = {(acc.x + p1.x) + (p2.x + p3.x), (acc.y + p1.y) + (p2.y + p3.y), (acc.z + p1.z) + (p2.z + p3.z), (acc.w + p1.w) + (p2.w + p3.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t idx = base + e;
        float v = r[e];
        int64_t off = idx;
        if (d.flags & EPI_NPAD8) {
            const int m = (int)(idx / d.N), n = (int)(idx - (int64_t)m * d.N);
            if ((n & 7) >= d.kw) continue;
            off = (int64_t)m * ((d.N >> 3) * d.kw) + (n >> 3) * d.kw + (n & 7);
        }
        if (d.flags & EPI_ACCUM) v += d.out[off];
        d.out[off] = v;
    }
}

static int flush_deferred(hipStream_t st) {
    std::lock_guard<std::recursive_mutex> lock(g_defer.mu);
    if (g_defer.n == 0) return 0;
    int blocks = 0;
    for (int i = 0; i <= DEFER_MAX; ++i) {
        g_defer.batch.start[i] = blocks;
        if (i < g_defer.n) blocks += (int)((g_defer.batch.it[i].total + 255) / 256);
    }
    hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g_defer.batch);
    g_defer.n = 0;
    return otal_launch_status();
}

template <int MODE, bool H = false>
static int launch_splitk_reduce(const ConvArgs& a, hipStream_t st) {
    static_assert(!(H && MODE == MODE_WGRAD), "weight gradients are fp32");
    const int64_t total = (int64_t)a.M * a.N;
    const bool v4 = (total % 4 == 0) && (((uintptr_t)a.slab & 15) == 0);
    if (MODE == MODE_WGRAD) {
        std::lock_guard<std::recursive_mutex> lock(g_defer.mu);
        g_defer.last_end = 0;
        if (g_defer.on && v4 && total <= (1 << 23)) {
            if (g_defer.n == DEFER_MAX)
                if (int e = flush_deferred(st)) return e;
            DeferItem& d = g_defer.batch.it[g_defer.n++];
            d.slab = a.slab; d.out = a.out; d.total = total; d.splits = a.splits; d.flags = a.flags; d.N = a.N; d.kw = a.g.kw;
            g_defer.last_end = reinterpret_cast<uintptr_t>(a.slab + (int64_t)a.splits * total);
            return 0;
        }
    }
    if (v4 && a.splits >= 16 && total <= (1 << 21) && !OTAL_OPT("OTAL_CONV_NOTALLREDUCE", 0)) {
        const int blocks = (int)((total / 4 + 63) / 64);
        hipLaunchKernelGGL((splitk_reduce_tall_kernel<MODE, H>), dim3(blocks), dim3(256), 0, st, a);
        return otal_launch_status();
    }
    const int64_t work = v4 ? total / 4 : total;
    const int blocks = (int)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
    if (v4) hipLaunchKernelGGL((splitk_reduce_kernel<MODE, 4, H>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((splitk_reduce_kernel<MODE, 1, H>), dim3(blocks), dim3(256), 0, st, a);
    return otal_launch_status();
}

// W (Cout, Cin, kvol) -> W^T packed (Cin, Cout, kvol): the A operand of DGRAD
__global__ __launch_bounds__(256) void pack_wt_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                      int Cout, int Cin, int kvol) {
    const int64_t total = (int64_t)Cout * Cin * kvol;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int r = (int)(idx % kvol);
        const int64_t q = idx / kvol;
        const int co = (int)(q % Cout), ci = (int)(q / Cout);
        wt[idx] = w[((int64_t)co * Cin + ci) * kvol + r];
    }
}

int fill_geom(ConvGeom& g, const int* d) {
    // d: B,Cin,Cout, Ti,Hi,Wi, To,Ho,Wo, kt,kh,kw, st,sh,sw, pt,ph,pw, nlev, lev[0..8]
    g.B = d[0]; g.Cin = d[1]; g.Cout = d[2];
    g.Ti = d[3]; g.Hi = d[4]; g.Wi = d[5];
    g.To = d[6]; g.Ho = d[7]; g.Wo = d[8];
    g.kt = d[9]; g.kh = d[10]; g.kw = d[11];
    g.st = d[12]; g.sh = d[13]; g.sw = d[14];
    g.pt = d[15]; g.ph = d[16]; g.pw = d[17];
    g.nlev = d[18];
    for (int i = 0; i <= OTAL_CONV_MAX_LEVELS; ++i) g.lev[i] = d[19 + i];
    const int v[] = {g.B, g.Cin, g.Cout, g.Ti, g.Hi, g.Wi, g.To, g.Ho, g.Wo, g.kt, g.kh, g.kw, g.st, g.sh, g.sw};
    for (int x : v) if (x <= 0) return OTAL_E_SHAPE;
    if (g.pt < 0 || g.ph < 0 || g.pw < 0) return OTAL_E_SHAPE;
    if (g.kt > 8 || g.kh > 8 || g.kw > 8) return OTAL_E_UNSUPPORTED;      // 8-bit validity masks per axis
    if (g.st > 2 || g.sh > 2 || g.sw > 2) return OTAL_E_UNSUPPORTED;      // gathers are specialised for strides 1, 2
    if (g.nlev > 1) {
        if (g.nlev > OTAL_CONV_MAX_LEVELS || g.Hi != 1 || g.Wi != 1 || g.st != 1 || g.Ti != g.To) return OTAL_E_LEVELS;
        if (g.lev[0] != 0 || g.lev[g.nlev] != g.Ti) return OTAL_E_LEVELS;
        for (int i = 0; i < g.nlev; ++i) if (g.lev[i + 1] <= g.lev[i]) return OTAL_E_LEVELS;
    }
    if ((int64_t)g.Cin * g.Ti * g.Hi * g.Wi >= (1LL << 31) || (int64_t)g.Cout * g.To * g.Ho * g.Wo >= (1LL << 31) ||
        (int64_t)g.Cout * g.Cin * g.kt * g.kh * g.kw >= (1LL << 31)) return OTAL_E_SHAPE;   // 32-bit row offsets
    if ((int64_t)g.B * g.To * g.Ho * g.Wo >= (1LL << 31) || (int64_t)g.B * g.Ti * g.Hi * g.Wi >= (1LL << 31) ||
        (int64_t)g.Cin * g.kt * g.kh * g.kw >= (1LL << 31) || (int64_t)g.Cout * g.kt * g.kh * g.kw >= (1LL << 31)) return OTAL_E_SHAPE;
    return 0;
}

// tile height: least padded M, with a small penalty for the lower arithmetic intensity of short tiles
int choose_bm(int M, int tall = 0) {
    const int tall_env = OTAL_OPT("OTAL_CONV_TALL", 1);    // bit0: fwd/dgrad (on: 2c fwd +16 %), bit1: wgrad (off: -9 %)
    if ((tall & tall_env) && M % 192 == 0) return 192;     // one 192-row tile re-fetches the gathered operand half as often
    const int cand[4] = {128, 96, 64, 32};
    const double pen[4] = {1.00, 1.03, 1.10, 1.30};
    int best = 128;
    double bc = 1e30;
    for (int i = 0; i < 4; ++i) {
        const double c = (double)((M + cand[i] - 1) / cand[i] * cand[i]) * pen[i];
        if (c < bc) { bc = c; best = cand[i]; }
    }
    return best;
}

// choose split-K so that the grid fills the chip (256 CUs) without shredding K
int choose_splits(int tiles, int K, int prec = 1, bool wgrad = false) {
    const int target_env = OTAL_OPT("OTAL_CONV_SPLIT_BLOCKS", 0);
    // bf16: >= 8 K steps of 32 per split (fewer, larger slabs: measured +4 % step throughput over 4);
    // fp32 parity path: 128 k per split as in the version the gradient-parity fixtures were validated with
    const int minsteps_env = OTAL_OPT("OTAL_CONV_SPLIT_MINSTEPS", 0);
    const int cap_env = OTAL_OPT("OTAL_CONV_MAXSPLIT", 0);
    // The vector weight-gradient kernel keeps 4 workgroups per CU resident and its K is huge (all positions): it wants two
    // full waves of workgroups (2048; 512 left it at 2 waves per SIMD, 61 % of wave time parked).  Splits of >= 16 K steps:
    // a K step is latency-bound (~1 us) when few workgroups are resident, so the small 1x1 / 1-D layers (18 k or 1 k
    // positions, a handful of tiles) finish sooner as many short splits than as a few long ones (measured per step:
    // 48 steps 407.6 clips/s, 24: 419.0, 12: 420.1, 6: 416.2).  Forward / data gradient keep the 512-workgroup target.
    const bool wv = wgrad && prec;
    const int target = target_env ? target_env : (wv ? 2048 : 512);
    const int wg_minsteps_env = OTAL_OPT("OTAL_WGRAD_MINSTEPS", 0);
    const int minsteps = (wv && wg_minsteps_env) ? wg_minsteps_env : minsteps_env ? minsteps_env : (wv ? 16 : (prec ? 8 : 4));
    const int cap = cap_env ? cap_env : (wv ? 1024 : 384);
    if (tiles >= target * 3 / 4) return 1;
    int want = (target + tiles - 1) / tiles;
    int maxs = K / (minsteps * 32);
    if (maxs < 1) maxs = 1;
    int s = want < maxs ? want : maxs;
    return s < 1 ? 1 : (s > cap ? cap : s);
}

static const float* zero_word_address() {
    static const float* z = nullptr;         // address of a module global: constant for the process lifetime
    if (!z) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_zero4)) == hipSuccess) z = static_cast<const float*>(p);
    }
    return z;
}

// byte extents for the batched epilogue (0 = does not fit 32-bit offsets -> element-wise fallback)
template <int MODE>
static void set_epilogue_extents(ConvArgs& a) {
    const ConvGeom& g = a.g;
    int64_t out;
    if (MODE == MODE_FWD) out = 4 * ((int64_t)(g.B - 1) * g.y_bs + (int64_t)(g.Cout - 1) * g.y_cs + conv_out_positions(g));
    else if (MODE == MODE_DGRAD) out = 4 * ((int64_t)(g.B - 1) * g.x_bs + (int64_t)(g.Cin - 1) * g.x_cs + conv_in_positions(g));
    else out = 4 * (int64_t)g.Cout * g.Cin * conv_kvol(g);
    a.out_bytes = (out > 0 && out < (int64_t)0xfffffff0u && !OTAL_OPT("OTAL_CONV_SLOW_EPILOGUE", 0)) ? (unsigned)out : 0u;
    const int64_t slab = 4 * (int64_t)a.splits * a.M * a.N;
    a.slab_bytes = (a.splits > 1 && slab < (int64_t)0xfffffff0u && !OTAL_OPT("OTAL_CONV_SLOW_EPILOGUE", 0)) ? (unsigned)slab : 0u;
}

// =================================================================================================
// Direct 3x3x3 convolution (bf16 operands; stride 1, SAME, P % 128 == 0, channel count % 16 == 0, W <= 24).
// The gather kernels above re-fetch every input element 27 times through L1 and are bound by the texture addresser /
// L2->L1 rate.  Here a K step is (temporal tap dt, 16 input channels): the workgroup stages ONCE the span of input
// positions [n0 - W - 1, n0 + 127 + W + 1] of plane offset dt (128 + 2W + 2 positions x 16 channels, bf16, 48-byte
// position pitch) and the 144-wide slice of the packed weights, and feeds all 9 in-plane taps from LDS: the tap only
// shifts the lane's LDS address by a uniform amount, out-of-range taps are replaced by zero operands with the lane's
// validity bits.  Per K step and wave: 27 MFMAs (BM = 96) against ~10 sixteen-byte loads per thread -- the matrix core,
// not the load path, is the long pole.  DGRAD is the same kernel on dy with the taps flipped in the weight pack.
struct DirectArgs {
    ConvArgs c;                 // geometry, tensors, epilogue options (M, N, out, scale, ... as for the other kernels)
    const unsigned short* wp;   // packed weights [Mpad][C/16][3 dt][9 taps][16] bf16
    int C, Ktot;                // source channels, C * 27
    unsigned wp_bytes;
};

// WN = position tiles per wave.  With one (WN = 1) every wave re-reads the whole 96 x 144 weight slice for its 32 positions:
// 36 sixteen-byte LDS reads per 27 MFMAs, and the eight waves' reads (2300 LDS cycles per K step) outlast their MFMAs (1730).
// WN = 2 (512 positions per workgroup, still 8 waves) shares each weight fragment between two position tiles: 45 reads
// per 54 MFMAs -- the kernel becomes MFMA-bound.
// PX = byte pitch of a position's 16 channels in LDS.  48: padded, conflict-free as it is.  32: dense, with the two 16-byte
// halves of a position swapped where bit 3 of the position index is set -- slot(p, h) = 2 (p mod 8) + (h ^ ((p >> 3) & 1)) is a
// bijection of (p mod 16, h) onto the sixteen 16-byte slots of a bank row, so the sixteen lanes of every ds_read_b128 group
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... : sixteen distinct positions mod 16 whatever the tap shift) never collide.
// The dense pitch brings a 96 x 256 tile of four waves under 80 KB: TWO workgroups per CU that are not barrier-coupled, so
// one's staging / store / epilogue phases run under the other's MFMAs (the 8-wave 512-position tile leaves the matrix
// pipes idle in those phases: ~12 us of epilogue per 50 us tile on Conv3d_2c).
// XPF2: the gathered positions travel TWO K steps ahead of their use (two register sets, K loop unrolled by two) instead of
// one; the weights stay one step ahead (they are L2 hits).  Costs 8 * X_ITERS registers (the 96 x 512 forward variant lands
// on exactly 256, no spills), so only the one-workgroup-per-CU variants take it.  Worth 1 .. 5 % per launch, not the ~19 %
// a per-tile budget of Conv3d_2c forward had attributed to load stalls (48 us per tile = 24 MFMA at the sustained clock +
// 12 epilogue + 3 first loads + "9 waiting for positions"): one step of cover was nearly enough.
// H: the gathered tensor and the output are STORED as bf16.  A load item is (EIGHT positions, channel pair): two 16-byte loads,
// transposed into the [position][16 channels] tile by pair_lo / pair_hi -- half the load instructions of the fp32 form for
// the same LDS stores, no conversion; the span starts at an even position (W + 2 in front of the tile instead of W + 1) so
// that every load is 4-byte aligned.  Operands, accumulators and results are those of the fp32-tensor kernel on the same values.
template <int BM, int MODE, int BNP, int WN, int PX = 48, int MINW = 1, bool XPF2 = false, bool H = false>
__global__ __launch_bounds__(BNP * 2 / WN, MINW) void conv3_direct_kernel(const DirectArgs d) {
    constexpr int DNT = BNP * 2 / WN;                       // one wave per 32 * WN positions
    constexpr int WM = BM / 32, PA = 304, SPAN_MAX = BNP + (H ? 56 : 52);
    constexpr int ESZ = H ? 2 : 4, XV = H ? 8 : 4;           // bytes per element, positions per load
    static_assert(PX == 48 || PX == 32, "position pitch");
    constexpr int A_PIECES = (BM * 18 + DNT - 1) / DNT;       // 16-byte weight pieces per thread per K step
    constexpr int X_ITEMS = (SPAN_MAX / XV + 1) * 8;        // (quad [octet] of positions, channel pair) items per K step, upper bound
    constexpr int X_ITERS = (X_ITEMS + DNT - 1) / DNT;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];     // [2][BM * PA] weights, then [2][SPAN_MAX * PX] positions
    unsigned char* const smA0 = dsm;
    unsigned char* const smX0 = dsm + 2 * BM * PA;
    const unsigned char* const smZero = dsm + 2 * BM * PA + 2 * SPAN_MAX * PX;      // 16 zero bytes (written below)
    if (threadIdx.x < 4) reinterpret_cast<unsigned*>(dsm + 2 * BM * PA + 2 * SPAN_MAX * PX)[threadIdx.x] = 0u;
    auto smA = [&](int buf) { return smA0 + buf * (BM * PA); };
    auto smX = [&](int buf) { return smX0 + buf * (SPAN_MAX * PX); };
    const ConvArgs& a = d.c;
    const ConvGeom& g = a.g;
    const ConvFastDiv& fd = a.fd;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileId tile = xcd_tile(true);
    const int m0 = tile.y * BM, n0 = tile.x * BNP;
    const int W = g.Wi, HW = g.Hi * g.Wi;
    const int XO = H ? W + 2 : W + 1;                       // positions staged in front of the tile
    const int span = BNP + XO + W + 1, nq = (span + XV - 1) / XV;
    const float* src = MODE == MODE_FWD ? a.x : a.dy;
    const int64_t sbs = MODE == MODE_FWD ? g.x_bs : g.y_bs, scs = MODE == MODE_FWD ? g.x_cs : g.y_cs;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)a.src_bytes, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(d.wp), 0, (int)d.wp_bytes, 0x00020000);
    // the tile lies inside one sample (P % 128 == 0): sample b, first position p0
    const uint32_t bsm = fd_div(fd.P, (uint32_t)n0);
    const int p0 = n0 - (int)(bsm * fd.P.d);
    const unsigned tile_off = (unsigned)(((int64_t)bsm * sbs + p0 - XO) * ESZ);         // byte offset of span element 0 at dt = 1, channel 0

    // this lane's output positions: validity of the 3 + 3 + 3 taps (FWD form; DGRAD uses flipped taps in the pack)
    unsigned tmask[WN], hwmask[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int nl = n0 + (wave * WN + j) * 32 + (lane & 31);
        const PosDec o = dec_pos_fd(nl < a.N ? nl : 0, fd.To, fd.Ho, fd.Wo);
        const bool live = nl < a.N;
        tmask[j] = 0; hwmask[j] = 0;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) tmask[j] |= (unsigned)(live && (unsigned)(o.t + dt - 1) < (unsigned)g.Ti) << dt;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw)
                hwmask[j] |= (unsigned)((unsigned)(o.h + dh - 1) < (unsigned)g.Hi && (unsigned)(o.w + dw - 1) < (unsigned)g.Wi) << (dh * 3 + dw);
    }
    // ---- per-thread load items
    unsigned xvo[X_ITERS];          // byte offset of (quad, channel pair) relative to (dt = 1, chunk 0), or 0xffffffff if unused
    int xlds[X_ITERS];
#pragma unroll
    for (int i = 0; i < X_ITERS; ++i) {
        const int item = tid + DNT * i;
        const int cp = item & 7, q = item >> 3;            // channel pair fastest: consecutive lanes write consecutive LDS dwords
        const bool used = q < nq;
        xvo[i] = used ? tile_off + (unsigned)((2 * cp * scs + XV * q) * ESZ) : 0xffffffffu;
        if constexpr (H)                                           // positions 8q .. 8q+7 share bit 3 = bit 0 of q
            xlds[i] = PX == 48 ? (8 * q) * PX + cp * 4 : (8 * q) * PX + ((((cp >> 2) ^ q) & 1) << 4) + (cp & 3) * 4;
        else
            xlds[i] = PX == 48 ? (4 * q) * PX + cp * 4             // positions 4q .. 4q+3 share bit 3 = bit 1 of q
                               : (4 * q) * PX + ((((cp >> 2) ^ (q >> 1)) & 1) << 4) + (cp & 3) * 4;
    }
    unsigned avo[A_PIECES];
#pragma unroll
    for (int j = 0; j < A_PIECES; ++j) {
        const int p = tid + DNT * j;
        avo[j] = (unsigned)(((m0 + p / 18) * d.Ktot * 2) + (p % 18) * 16);
    }
    unsigned rxs[XPF2 ? 2 : 1][X_ITERS][2][4];
    Words4 ra[A_PIECES];
    const int nsteps = (d.C >> 4) * 3;
    auto load_x = [&](int s, auto SET) {
        unsigned (&rx)[X_ITERS][2][4] = rxs[decltype(SET)::value];
        const int cb = s / 3, dt = s - cb * 3;
        const unsigned so = (unsigned)(((int64_t)cb * 16 * scs + (int64_t)(dt - 1) * HW) * ESZ);
#pragma unroll
        for (int i = 0; i < X_ITERS; ++i) {
            const unsigned vo = xvo[i] == 0xffffffffu ? a.src_bytes : xvo[i] + so;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const Words4 v = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, h ? (int)(scs * ESZ) : 0, 0));
                rx[i][h][0] = v.a; rx[i][h][1] = v.b; rx[i][h][2] = v.c; rx[i][h][3] = v.d;
            }
        }
    };
    auto load_a = [&](int s) {
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j)
            if ((BM * 18) % DNT == 0 || tid + DNT * j < BM * 18)
                ra[j] = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rw, avo[j], s * 288, 0));
    };
    auto store = [&](int buf, int s, auto SET) {
        unsigned (&rx)[X_ITERS][2][4] = rxs[decltype(SET)::value];
        const int cb = s / 3, dt = s - cb * 3;
        const unsigned so = (unsigned)(((int64_t)cb * 16 * scs + (int64_t)(dt - 1) * HW) * ESZ);
#pragma unroll
        for (int i = 0; i < X_ITERS; ++i) {
            if (xvo[i] == 0xffffffffu) continue;
            // a quad that starts in front of the tensor is rejected as a whole: re-fetch its in-range elements (rare lanes)
            const unsigned vo = xvo[i] + so;
            const bool neg = vo >= 0xfffffff0u;
            if (__builtin_amdgcn_ballot_w64(neg) != 0ull) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if constexpr (H) {                  // (loads are 4-byte aligned here: whole words, the first ones out of range)
#pragma unroll
                        for (int e = 1; e < 4; ++e) {
                            const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, neg ? wrap_add(vo, 4u * e) : a.src_bytes,
                                                                                     h ? (int)(scs * ESZ) : 0, 0);
                            rx[i][h][e] = neg ? v : rx[i][h][e];
                        }
                    } else {
#pragma unroll
                        for (int e = 1; e < 4; ++e) {
                            const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, neg ? wrap_add(vo, 4u * e) : a.src_bytes,
                                                                                     h ? (int)(scs * 4) : 0, 0);
                            rx[i][h][e] = neg ? v : rx[i][h][e];
                        }
                    }
                }
            }
            if constexpr (H) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *reinterpret_cast<unsigned*>(smX(buf) + xlds[i] + e * PX) =
                        (e & 1) ? pair_hi(rx[i][0][e >> 1], rx[i][1][e >> 1]) : pair_lo(rx[i][0][e >> 1], rx[i][1][e >> 1]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    *reinterpret_cast<unsigned*>(smX(buf) + xlds[i] + e * PX) =
                        cvt_pk_bf16(__uint_as_float(rx[i][0][e]), __uint_as_float(rx[i][1][e]));
            }
        }
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j) {
            const int p = tid + DNT * j;
            if ((BM * 18) % DNT == 0 || p < BM * 18) *reinterpret_cast<Words4*>(smA(buf) + (p / 18) * PA + (p % 18) * 16) = ra[j];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, XPF2 ? 1 : 0>;
    load_x(0, Set0{});
    load_a(0);
    store(0, 0, Set0{});
    if constexpr (XPF2) load_x(nsteps > 1 ? 1 : 0, Set1{});     // step 1's positions: in flight while step 0 computes
    __syncthreads();
    const int xpos = wave * WN * 32 + (lane & 31) + XO;                                  // this lane's first position in the span
    const int xrow = xpos * PX + (lane >> 5) * 16;
    // Fragment reads of tap group g + 1 are issued ONE BY ONE BETWEEN the MFMAs of group g (second register set): the reads'
    // issue slots and their LDS latency then sit inside the matrix pipe's 32-cycle issue intervals instead of in front of a
    // burst of MFMAs.  tools/ubench/ldsmfma.hip (this loop without global traffic, random operands): "reads, then MFMAs"
    // 1515 -> interleaved 1858 TFLOP/s at 8 waves x (96 x 32) tiles, 1541 -> 1923 at 16 waves x (64 x 32); the MFMA-only
    // ceiling of the same loops is 1564 - 1834 TFLOP/s on this chip with random data (clock under matrix load), not 2500.
    // A tap outside the tensor reads sixteen zero bytes kept behind the tiles: the ADDRESS is selected before the read, so
    // nothing touches a fragment between its read and its MFMAs.
    auto frag_a = [&](int buf, int g9, int i) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(smA(buf) + (i * 32 + (lane & 31)) * PA + g9 * 32 + (lane >> 5) * 16);
    };
    auto frag_b = [&](int buf, int dt, int g9, int j) -> bf16x8 {
        const int dh = g9 / 3, dw = g9 - dh * 3;
        const unsigned char* src;
        if (PX == 48) {
            src = smX(buf) + xrow + (j * 32 + (dh - 1) * W + (dw - 1)) * PX;
        } else {
            const int p = xpos + j * 32 + (dh - 1) * W + (dw - 1);
            src = smX(buf) + p * PX + ((((lane >> 5) ^ (p >> 3)) & 1) << 4);
        }
        const bool ok = ((tmask[j] >> dt) & 1u) && ((hwmask[j] >> g9) & 1u);
        return *reinterpret_cast<const bf16x8*>(ok ? src : smZero);
    };
    // one K step; CUR = the register set that holds the positions of step s + 1 when XPF2 (loaded during step s - 1: stored
    // to LDS at the end of this step), while the loads of step s + 2 go to the other set; without XPF2 there is one set,
    // loaded and stored within the step
    auto kstep = [&](int s, auto CUR, auto OTHER) {
        const int buf = s & 1;
        const int dt = s % 3;
        const int sn = s + 1 < nsteps ? s + 1 : s;          // the prefetch past the end re-reads the last step
        const int sx = XPF2 ? (s + 2 < nsteps ? s + 2 : nsteps - 1) : sn;
        bf16x8 avA[WM], bvA[WN], avB[WM], bvB[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) avA[i] = frag_a(buf, 0, i);
#pragma unroll
        for (int j = 0; j < WN; ++j) bvA[j] = frag_b(buf, dt, 0, j);
#pragma unroll
        for (int g9 = 0; g9 < 9; ++g9) {
#ifdef OTAL_DIRECT_ABLATE       // timing experiments only (tools/ablate_direct.sh): results are wrong under these flags
            if (g9 == 0 && !(a.flags & DBG_NOLOAD)) load_x(sx, OTHER);
            if (g9 == 1 && !(a.flags & 128)) load_a(sn);
#else
            if (g9 == 0) load_x(sx, OTHER);
            if (g9 == 1) load_a(sn);
#endif
            bf16x8 (&av)[WM] = (g9 & 1) ? avB : avA;
            bf16x8 (&bv)[WN] = (g9 & 1) ? bvB : bvA;
            bf16x8 (&avn)[WM] = (g9 & 1) ? avA : avB;
            bf16x8 (&bvn)[WN] = (g9 & 1) ? bvA : bvB;
            int n = 0;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                    if (g9 < 8) {
                        if (n < WM) avn[n] = frag_a(buf, g9 + 1, n);
                        else if (n < WM + WN) bvn[n - WM] = frag_b(buf, dt, g9 + 1, n - WM);
                    }
                    ++n;
                }
            if (WM * WN < WM + WN && g9 < 8) {              // one position tile per wave (WN = 1): WM MFMAs, WM + 1 fragments
#pragma unroll
                for (int r = WM * WN; r < WM + WN; ++r) {
                    if (r < WM) avn[r] = frag_a(buf, g9 + 1, r);
                    else bvn[r - WM] = frag_b(buf, dt, g9 + 1, r - WM);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef OTAL_DIRECT_ABLATE
        if (!(a.flags & DBG_NOSTORE)) store(buf ^ 1, sn, CUR);
        if (!(a.flags & DBG_NOBARRIER)) __syncthreads();
#else
        store(buf ^ 1, sn, CUR);
        __syncthreads();
#endif
    };
    if constexpr (XPF2) {
        for (int s = 0; s < nsteps; s += 2) {
            kstep(s, Set1{}, Set0{});
            if (s + 1 < nsteps) kstep(s + 1, Set0{}, Set1{});
        }
    } else {
        for (int s = 0; s < nsteps; ++s) kstep(s, Set0{}, Set0{});
    }
    if ((MODE == MODE_FWD && a.half) || (MODE == MODE_DGRAD && a.xhalf)) {
        // bf16-STORED output (FWD: y -- H kernels, and the fp32-input layer that only feeds a strided max-pool, which commutes
        // with the monotonic rounding; DGRAD: dx of the H kernels): scale / shift / ReLU or the producer's BN scale, rounded
        // once, transposed through LDS (the operand tiles are dead: BM rows x BNP positions of bf16 + 16 bytes of pitch fit in
        // them), 16-byte runs along the positions; the data gradient's ReLU mask is the bf16 activation, read in the same pieces.
        constexpr int PT = BNP * 2 + 16;
        static_assert(BM * PT + BM * 8 <= 2 * BM * PA + 2 * SPAN_MAX * PX, "staging tile fits in the operand buffers");
        unsigned char* tile = dsm;
        float* rows = reinterpret_cast<float*>(dsm + BM * PT);
        __syncthreads();            // (every wave is done with the operand tiles: the K loop ends with a barrier, kept explicit here)
        for (int r = tid; r < BM; r += DNT) {
            const int m = m0 + r;
            if constexpr (MODE == MODE_FWD) {
                rows[2 * r] = (m < a.M && a.scale) ? a.scale[m] : 1.f;
                rows[2 * r + 1] = (m < a.M && a.shift) ? a.shift[m] : 0.f;
            } else {
                rows[2 * r] = (m < a.M && a.escale) ? a.escale[m] : 1.f;
                rows[2 * r + 1] = 0.f;
            }
        }
        __syncthreads();
        const bool relu = MODE == MODE_FWD && (a.flags & EPI_RELU) != 0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int nl = (wave * WN + j) * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[i][j][r] * rows[2 * lr] + rows[2 * lr + 1];
                    if (relu) v = fmaxf(v, 0.f);
                    *reinterpret_cast<unsigned short*>(tile + lr * PT + nl * 2) = (unsigned short)(cvt_pk_bf16(v, 0.f) & 0xffffu);
                }
            }
        __syncthreads();
        unsigned short* yh = reinterpret_cast<unsigned short*>(a.out);
        const unsigned short* mk = MODE == MODE_DGRAD ? reinterpret_cast<const unsigned short*>(a.emask) : nullptr;
        const int64_t ocs = MODE == MODE_FWD ? g.y_cs : g.x_cs;
        const int64_t ybase = (int64_t)bsm * (MODE == MODE_FWD ? g.y_bs : g.x_bs) + p0;
        constexpr int U = 4;        // pieces per trip: their mask loads travel together (one memory round trip, not four)
        for (int p0 = tid; p0 < BM * (BNP / 8); p0 += U * DNT) {
            int64_t off[U];
            bool ok[U];
            Words4 m4[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * DNT;
                const int row = p / (BNP / 8), q = p - row * (BNP / 8);
                ok[u] = p < BM * (BNP / 8) && m0 + row < a.M && n0 + q * 8 < a.N;
                off[u] = ok[u] ? ybase + (int64_t)(m0 + row) * ocs + q * 8 : 0;
                if (mk) m4[u] = *reinterpret_cast<const Words4*>(mk + off[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const int p = p0 + u * DNT;
                const int row = p / (BNP / 8), q = p - row * (BNP / 8);
                Words4 v = *reinterpret_cast<const Words4*>(tile + row * PT + q * 16);
                if (mk) {
                    v.a &= bf16_relu_mask2(m4[u].a); v.b &= bf16_relu_mask2(m4[u].b);
                    v.c &= bf16_relu_mask2(m4[u].c); v.d &= bf16_relu_mask2(m4[u].d);
                }
                *reinterpret_cast<Words4*>(yh + off[u]) = v;
            }
        }
        return;
    }
#ifdef OTAL_DIRECT_ABLATE
    if (a.flags & 64) {         // no epilogue (the accumulators stay live through an impossible store)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 1.2345678e30f) a.out[0] = t;
        return;
    }
#endif
    store_acc<MODE, WM, WN, BM>(a, acc, m0, n0, 0, wave * WN * 32, lane, 0, reinterpret_cast<float*>(smA(0)));
}

template <int MODE>
__global__ __launch_bounds__(256) void pack_direct_kernel(unsigned* __restrict__ wp, const float* __restrict__ w, int M, int Mpad,
                                                          int C, int natural, FastDiv fh) {
    pack_direct_body<MODE>(wp, w, M, Mpad, C, natural, blockIdx.x, gridDim.x, fh);
}

// ---- Conv3d_1a_7x7 forward, direct: 7x7x7 taps, stride 2, THREE input channels on a 96-wide plane.
// The gather kernels are staging-bound on this layer (18 VALU instructions per MFMA, matrix pipes 17 % busy: with 64 output
// channels each gathered element feeds only 128 FLOPs).  Here a workgroup owns a 2 (t) x 2 (h) x 48 (w) block of output
// positions and stages its whole receptive field -- 9 planes x 9 rows x 104 columns -- ONCE into LDS as channel-last
// bf16 pixels padded to four channels {c0, c1, c2, 0} (8 bytes).  A K step is one (dt, dh) kernel row: its 7 dw taps x 4
// channels (+ one all-zero eighth tap) are 32 consecutive k, and for an output position they are 64 CONTIGUOUS bytes of
// the patch starting at pixel 2 wo -- so the im2col operand of every MFMA is one aligned ds_read_b128, with no address
// tables, no masks (the zero padding is in the patch) and no per-element VALU work.  Weights: [Mpad][49 rows][32 k] bf16,
// zero where ci = 3 or dw = 7, one 4 KB slice per K step through two LDS slots.
// Measured (b = 8, 310 GFLOP): 0.84 ms against 0.95 ms of the gather kernel (tools/micro_conv.py 1a fwd).  Ablation: the
// epilogue + 49 barriers alone 0.29 ms (604 MB of output in 384-byte runs), staging 0.13, weight ring 0.07, LDS operand
// reads + MFMA 0.12 -- the phases of a workgroup run back to back and only two workgroups fit a CU (77 KB of LDS), so they
// add up instead of overlapping.  Round 4: geometries with To % 4 == 0 and Ho % 4 == 0 (the training shapes) run on
// conv1a_tile_fwd_kernel (conv1a_tile.hip: 4 x 4 x 48 tiles, persistent workgroups, weights in registers); this kernel
// serves the rest and is the bit-exact reference of that one (tests/test_ops_gpu.py).
constexpr int C1_TT = 2, C1_TR = 2, C1_WO = 48, C1_NPL = 9, C1_NR = 9, C1_NC = 104, C1_PITCH = C1_NC * 8;   // bytes per patch row
constexpr int C1_BNP = C1_TT * C1_TR * C1_WO, C1_NT = C1_BNP * 2, C1_PA = 80, C1_STEPS = 49;

struct Conv1aArgs {
    ConvArgs c;
    const unsigned short* wp;   // [Mpad][49][32] bf16
};

__global__ __launch_bounds__(256) void pack_conv1a_kernel(unsigned* __restrict__ wp, const float* __restrict__ w, int M, int Mpad) {
    const int pairs = Mpad * C1_STEPS * 16;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < pairs; p += gridDim.x * 256) {
        const int m = p / (C1_STEPS * 16), r = p - m * (C1_STEPS * 16), s = r >> 4, k = (r & 15) * 2;
        float v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kk = k + i, dw = kk >> 2, ci = kk & 3;
            v[i] = (m < M && dw < 7 && ci < 3) ? w[((int64_t)m * 3 + ci) * 343 + s * 7 + dw] : 0.f;    // s = dt * 7 + dh
        }
        wp[p] = cvt_pk_bf16(v[0], v[1]);
    }
}

__global__ __launch_bounds__(C1_NT) void conv1a_direct_fwd_kernel(const Conv1aArgs d) {
    constexpr int BM = 64, WM = 2;
    __shared__ __attribute__((aligned(16))) unsigned char patch[C1_NPL * C1_NR * C1_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char smA[2][BM * C1_PA];      // two K-step slices (+ patch = 77.6 KB: two workgroups per CU)
    const ConvArgs& a = d.c;
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile -> (sample, to0, ho0); neighbouring workgroups share input rows / planes
    const int tiles_h = g.Ho / C1_TR, tiles_t = g.To / C1_TT;
    // workgroups are dealt round-robin to the 8 XCDs, each with its own L2: give every XCD a CONTIGUOUS range of tiles, so
    // that the tiles resident on it at any time are neighbours in (t, h) and share their input planes in that L2 (a tile
    // reads 9 planes x 9 rows for 2 x 2 new ones: without the remap the 8x re-reads all go to the fabric)
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int th = bid % tiles_h; bid /= tiles_h;
    const int tt0 = bid % tiles_t;
    const int b = bid / tiles_t;
    const int to0 = tt0 * C1_TT, ho0 = th * C1_TR;
    const int m0 = blockIdx.y * BM;
    const float* xb = a.x + (int64_t)b * g.x_bs;

    // ---- stage the receptive field: item = (plane, row, quad of 4 pixels); out-of-range planes / rows are zero rows
    constexpr int ITEMS = C1_NPL * C1_NR * 24, ITERS = (ITEMS + C1_NT - 1) / C1_NT;
#pragma unroll
    for (int it0 = 0; it0 < ITERS; it0 += 3) {
        float4 v[3][3];
        int lds_off[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int item = tid + C1_NT * (it0 + u);
            const int pl = item / (C1_NR * 24), rem = item - pl * (C1_NR * 24), rr = rem / 24, q = rem - rr * 24;
            const int ti = 2 * to0 - g.pt + pl, hi = 2 * ho0 - g.ph + rr;
#ifdef OTAL_DIRECT_ABLATE
            const bool ok = !(a.flags & 4) && it0 + u < ITERS && item < ITEMS && (unsigned)ti < (unsigned)g.Ti && (unsigned)hi < (unsigned)g.Hi;
            lds_off[u] = !(a.flags & 8) && item < ITEMS && it0 + u < ITERS ? (pl * C1_NR + rr) * C1_PITCH + (4 * q + 2) * 8 : -1;
#else
            const bool ok = it0 + u < ITERS && item < ITEMS && (unsigned)ti < (unsigned)g.Ti && (unsigned)hi < (unsigned)g.Hi;
            lds_off[u] = item < ITEMS && it0 + u < ITERS ? (pl * C1_NR + rr) * C1_PITCH + (4 * q + 2) * 8 : -1;
#endif
            const float* src = xb + ((int64_t)(ok ? ti : 0) * g.Hi + (ok ? hi : 0)) * g.Wi + 4 * q;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
                v[u][ci] = ok ? *reinterpret_cast<const float4*>(src + ci * g.x_cs) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (lds_off[u] < 0) continue;
            uint2* dst = reinterpret_cast<uint2*>(patch + lds_off[u]);
            dst[0] = make_uint2(cvt_pk_bf16(v[u][0].x, v[u][1].x), cvt_pk_bf16(v[u][2].x, 0.f));
            dst[1] = make_uint2(cvt_pk_bf16(v[u][0].y, v[u][1].y), cvt_pk_bf16(v[u][2].y, 0.f));
            dst[2] = make_uint2(cvt_pk_bf16(v[u][0].z, v[u][1].z), cvt_pk_bf16(v[u][2].z, 0.f));
            dst[3] = make_uint2(cvt_pk_bf16(v[u][0].w, v[u][1].w), cvt_pk_bf16(v[u][2].w, 0.f));
        }
    }
    // the zero columns left and right of the plane: pixels 0, 1 (w = -2, -1) and 98 .. 103 (w = 96 .. 101)
    for (int i = tid; i < C1_NPL * C1_NR * 8; i += C1_NT) {
        const int row = i >> 3, e = i & 7;
        *reinterpret_cast<uint2*>(patch + row * C1_PITCH + (e < 2 ? e : 96 + e) * 8) = make_uint2(0u, 0u);
    }
    // ---- weights of K step s: 64 rows x 64 bytes = 256 sixteen-byte pieces
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(d.wp), 0,
                                                      (int)((int64_t)gridDim.y * BM * C1_STEPS * 64), 0x00020000);
    const unsigned avo = (unsigned)(((m0 + (tid >> 2)) * C1_STEPS) * 64 + (tid & 3) * 16);
    // The slices travel global -> registers -> LDS.  A slice is consumed one barrier after it is stored, and a K step is
    // only ~130 MFMA cycles long, so a load issued two steps ahead (first version) stalled EVERY step on the L2 latency:
    // the registers form a FIFO of seven slices (slice s lives in rq[s % 7]) -- loads run nine steps ahead of their use.
    Words4 rq[7];
#ifdef OTAL_DIRECT_ABLATE
    const bool abl_w = (a.flags & 128) != 0;
    auto load_a = [&](int s) { if (tid < 256 && !abl_w) rq[s % 7] = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rw, avo, s * 64, 0)); };
    auto store_a = [&](int s) { if (tid < 256 && !abl_w) *reinterpret_cast<Words4*>(smA[s & 1] + (tid >> 2) * C1_PA + (tid & 3) * 16) = rq[s % 7]; };
#else
    auto load_a = [&](int s) { if (tid < 256) rq[s % 7] = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rw, avo, s * 64, 0)); };
    auto store_a = [&](int s) { if (tid < 256) *reinterpret_cast<Words4*>(smA[s & 1] + (tid >> 2) * C1_PA + (tid & 3) * 16) = rq[s % 7]; };
#endif
    load_a(0);
    load_a(1);
#pragma unroll
    for (int s = 2; s < 7; ++s) load_a(s);
    store_a(0);
    store_a(1);
    load_a(7);                                              // into the registers slices 0 and 1 just left
    load_a(8);
    __syncthreads();

    // this lane's output position inside the tile and its pixel 2 wo of patch row (dt = 0, dh = 0)
    const int nl = wave * 32 + (lane & 31);
    const int lt = nl / (C1_TR * C1_WO), lrem = nl - lt * (C1_TR * C1_WO), lr = lrem / C1_WO, wo = lrem - lr * C1_WO;
    const int xbase = ((2 * lt) * C1_NR + 2 * lr) * C1_PITCH + (2 * wo) * 8 + (lane >> 5) * 16;
    const int abase = (lane & 31) * C1_PA + (lane >> 5) * 16;
    f32x16 acc[WM][1];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    // Software pipeline: the operands of step s+1 are read from LDS while the MFMAs of step s run (the patch is static and
    // weight slice s+1 became visible at the previous barrier).  Slice s+2 travels global -> registers -> the slot that held
    // slice s: its operands were read into registers during step s-1 and those reads retired before the last barrier.
    bf16x8 av[2][2][WM], bv[2][2];
    auto read_ops = [&](int set, int s) {
        const int dt = s / 7, dh = s - dt * 7;
        const unsigned char* xrow = patch + xbase + (dt * C1_NR + dh) * C1_PITCH;
        const unsigned char* arow = smA[s & 1] + abase;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < WM; ++i) av[set][kk][i] = *reinterpret_cast<const bf16x8*>(arow + i * 32 * C1_PA + kk * 32);
            bv[set][kk] = *reinterpret_cast<const bf16x8*>(xrow + kk * 32);
        }
    };
    read_ops(0, 0);
    __syncthreads();    // step 0 stores slice 2 into the slot of slice 0: every wave must have read slice 0 first (with the
                        // register FIFO the store no longer waits for a global load, so a late wave used to read slice 2)
#pragma unroll
    for (int s = 0; s < C1_STEPS; ++s) {    // fully unrolled: `set` and the FIFO slot index register arrays
        const int set = s & 1;
#ifdef OTAL_DIRECT_ABLATE       // timing experiments only (tools/ablate_direct.sh): results are wrong under these flags
        if (s + 1 < C1_STEPS && !(a.flags & 256)) read_ops(set ^ 1, s + 1);
        if (!(a.flags & 512))
#else
        if (s + 1 < C1_STEPS) read_ops(set ^ 1, s + 1);
#endif
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < WM; ++i)
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[set][kk][i], bv[set][kk], acc[i][0], 0, 0, 0);
        if (s + 2 < C1_STEPS) store_a(s + 2);              // into the slot of slice s (its operands were read during step s-1)
        if (s + 9 < C1_STEPS) load_a(s + 9);                // refills the register just stored
#ifdef OTAL_DIRECT_ABLATE
        if (!(a.flags & DBG_NOBARRIER))
#endif
        __syncthreads();
    }
#ifdef OTAL_DIRECT_ABLATE
    if (a.flags & 64) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][0][r];
        if (t == 1.2345678e30f) a.out[0] = t;
        return;
    }
#endif
    if (a.half) {
        // bf16 output (the layer's 604 MB of fp32 activations are only ever read back through bf16 roundings: MaxPool3d_2a
        // commutes with the monotonic rounding and Conv3d_2b rounds its operand anyway -- the forward values are unchanged):
        // scale / shift / ReLU, transpose through LDS (the patch is dead), 16-byte runs along w.
        constexpr int PT = 192 * 2 + 16;                    // bytes per output-channel row of the staging tile
        float* rows = reinterpret_cast<float*>(smA[0]);
        if (tid < BM) {
            const int m = m0 + tid;
            rows[2 * tid] = (m < a.M && a.scale) ? a.scale[m] : 1.f;
            rows[2 * tid + 1] = (m < a.M && a.shift) ? a.shift[m] : 0.f;
        }
        __syncthreads();
        const bool relu = (a.flags & EPI_RELU) != 0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][0][r] * rows[2 * lr] + rows[2 * lr + 1];
                if (relu) v = fmaxf(v, 0.f);
                *reinterpret_cast<unsigned short*>(patch + lr * PT + nl * 2) = (unsigned short)(cvt_pk_bf16(v, 0.f) & 0xffffu);
            }
        __syncthreads();
        unsigned short* yh = reinterpret_cast<unsigned short*>(a.out);
        for (int p = tid; p < BM * 24; p += C1_NT) {
            const int co = p / 24, rem = p - co * 24, lt2 = rem / 12, q = rem - lt2 * 12;
            if (m0 + co >= a.M) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(patch + co * PT + (lt2 * 96 + q * 8) * 2);
            *reinterpret_cast<uint4*>(yh + (int64_t)b * g.y_bs + (int64_t)(m0 + co) * g.y_cs + ((int64_t)(to0 + lt2) * g.Ho + ho0) * g.Wo + q * 8) = v;
        }
        return;
    }
    // this wave's 32 positions are consecutive in the output: rows ho0, ho0+1 of plane to0 + lt
    const int n_wave = ((b * g.To + to0 + (wave * 32) / (C1_TR * C1_WO)) * g.Ho + ho0) * g.Wo + (wave * 32) % (C1_TR * C1_WO);
    store_acc<MODE_FWD, WM, 1, BM>(a, acc, m0, n_wave, 0, 0, lane, 0, reinterpret_cast<float*>(smA[0]));
}

static inline bool conv1a_half_out_ok(const ConvGeom& g, const void* y) {
    return g.y_bs % 8 == 0 && g.y_cs % 8 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
}
static inline bool conv1a_direct_eligible(const ConvGeom& g, int mode, int prec, const void* x) {
    if (!prec || mode != MODE_FWD || g.nlev > 1 || OTAL_OPT("OTAL_CONV_NO1A", 0)) return false;
    if (g.Cin != 3 || g.kt != 7 || g.kh != 7 || g.kw != 7 || g.st != 2 || g.sh != 2 || g.sw != 2) return false;
    if (g.pt != 2 || g.ph != 2 || g.pw != 2 || g.Wi != 96 || g.Wo != C1_WO || g.Hi != 2 * g.Ho || g.Ti != 2 * g.To) return false;
    if (g.To % C1_TT || g.Ho % C1_TR || g.x_bs % 4 || g.x_cs % 4 || (reinterpret_cast<uintptr_t>(x) & 15)) return false;
    return (int64_t)g.B * g.To * g.Ho * g.Wo < (1LL << 31);
}
static inline size_t conv1a_wp_bytes(int M) { return (((size_t)((M + 63) / 64 * 64) * C1_STEPS * 64) + 255) & ~(size_t)255; }

int launch_conv1a_direct(ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
#ifdef OTAL_DIRECT_ABLATE
    a.flags |= (OTAL_OPT("OTAL_CONV_DEBUG", 0) & (4 | 8 | 16 | 64 | 128 | 256 | 512));
#endif
    if (otal_conv::conv1a_tile_eligible(a.g.To, a.g.Ho) && !(a.flags & EPI_ACCUM)) {      // 4 x 4 x 48 tiles (conv1a_tile.hip)
        otal_conv::Conv1aTileArgs t;
        t.x = a.x; t.w = a.w; t.wp = nullptr; t.out = a.out; t.scale = a.scale; t.shift = a.shift;
        t.x_bs = a.g.x_bs; t.x_cs = a.g.x_cs; t.y_bs = a.g.y_bs; t.y_cs = a.g.y_cs;
        t.B = a.g.B; t.Ti = a.g.Ti; t.Hi = a.g.Hi; t.To = a.g.To; t.Ho = a.g.Ho; t.M = a.M;
        t.relu = (a.flags & EPI_RELU) ? 1 : 0; t.half = a.half; t.flags = a.flags;
        const int e = otal_conv::launch_conv1a_tile(t, ws, ws_bytes, st);
        if (e != OTAL_E_UNSUPPORTED) return e;
    }
    const int tm = (a.M + 63) / 64, Mpad = tm * 64;
    const size_t wb = conv1a_wp_bytes(a.M);
    if (!ws || ws_bytes < wb) return OTAL_E_UNSUPPORTED;
    const int pairs = Mpad * C1_STEPS * 16;
    hipLaunchKernelGGL(pack_conv1a_kernel, dim3((pairs + 255) / 256), dim3(256), 0, st, reinterpret_cast<unsigned*>(ws), a.w, a.M, Mpad);
    if (int e = otal_launch_status()) return e;
    Conv1aArgs d;
    a.fd = make_conv_fastdiv(a.g);
    a.splits = 1; a.k_per_split = 0; a.slab = nullptr;
    set_epilogue_extents<MODE_FWD>(a);
    d.c = a;
    d.wp = reinterpret_cast<const unsigned short*>(ws);
    const dim3 grid(a.g.B * (a.g.To / C1_TT) * (a.g.Ho / C1_TR), tm, 1);
    hipLaunchKernelGGL(conv1a_direct_fwd_kernel, grid, dim3(C1_NT), 0, st, d);
    return otal_launch_status();
}

// ---- weight gradient of the 1-D pyramid / head layers (H = W = 1, stride 1, k = 1 or 3, level-packed or not).
// These 36 launches per step have 126 .. 256 positions per sample -- no vector path (P % 32 != 0, W = 1) -- and ran on the
// generic tap-table kernel at ~30 us each for 0.5 .. 1.6 GFLOP.  But with H = W = 1 both operands are K-CONTIGUOUS rows:
// dW[co][ci][dt] = sum_t dy[co][t] * x[ci][t + dt - pt].  A workgroup owns a 64 x 64 (co, ci) tile of one (sample,
// 128-position chunk): both row blocks are staged once in LDS as bf16 (x with one halo element per side), every MFMA
// operand is an aligned 16-byte LDS read, and the two shifted taps are the aligned neighbours funnel-shifted by one
// element, with the taps that would cross a level boundary masked per position.  Partial sums go to split-K slabs
// (split = sample x chunk), reduced in order by splitk_reduce_kernel.
constexpr int W1_TC = 128, W1_PITCH = 304;      // positions per chunk; LDS row pitch in bytes (152 bf16: 2-way conflicts at most)

template <int KT>
__global__ __launch_bounds__(256) void conv_wgrad1d_kernel(const ConvArgs a, int nchunks, int units, int upw) {
    __shared__ __attribute__((aligned(16))) unsigned char sdy[64 * W1_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char sx[64 * W1_PITCH];      // element 8 + tl holds x[t0 + tl]
    __shared__ __attribute__((aligned(16))) unsigned mstart[W1_TC / 2], mend[W1_TC / 2];   // bf16-pair masks: 0 where the tap is cut
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_co = (g.Cout + 63) / 64;
    // grid = (splits, input-channel tiles, output-channel tiles [x 2 for a pair launch]): the SPLIT (sample, chunk) is the fastest
    // index, so the workgroups of one sample's chunk -- which all read the same 64 x 128 blocks of x and dy rows -- share an
    // XCD's L2 (workgroups go to the XCDs round-robin by linear index; with the channel tile fastest every XCD fetched every sample)
    const bool second = (int)blockIdx.z >= tiles_co;        // pair launch: grid.z = 2 x the output-channel tiles
    const int co0 = ((int)blockIdx.z - (second ? tiles_co : 0)) * 64, ci0 = blockIdx.y * 64, split = blockIdx.x;
    const float* const pdy = second ? a.dy2 : a.dy;
    const float* const px = second ? a.x2 : a.x;
    const int T = g.Ti;
    const int mi = wave & 1, ni = wave >> 1, h8 = (lane >> 5) * 8;
    const unsigned char* arow = sdy + (mi * 32 + (lane & 31)) * W1_PITCH + h8 * 2;
    const unsigned char* brow = sx + (ni * 32 + (lane & 31)) * W1_PITCH + (8 + h8) * 2;
    f32x16 acc[KT];
#pragma unroll
    for (int d = 0; d < KT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    // a workgroup walks `upw` (sample, chunk) units: fewer, fatter split-K slabs (one epilogue and one slab per upw units)
    for (int unit = split * upw; unit < min(units, (split + 1) * upw); ++unit) {
    const int b = unit / nchunks, t0 = (unit - b * nchunks) * W1_TC;
    const float* dyb = pdy + (int64_t)b * g.y_bs + t0;
    const float* xb = px + (int64_t)b * g.x_bs + t0;
    __syncthreads();            // the previous unit's operand reads are done
    // halo: x[t0 - 1] at element 7, x[t0 + 128] at element 136 (pairs written whole: 6|7 and 136|137) -- fetched first (clamped
    // address, masked at the store below) so that it travels with the tile loads instead of one more round trip behind them
    unsigned halo_raw = 0u, halo_keep = 0u;
    if (tid < 128) {
        const int r = tid & 63, u = (tid >> 6) ? t0 + W1_TC : t0 - 1;
        const bool ok = ci0 + r < g.Cin && u >= 0 && u < T;
        halo_keep = ok ? ~0u : 0u;
        halo_raw = __float_as_uint(px[(int64_t)b * g.x_bs + (int64_t)min(ci0 + r, g.Cin - 1) * g.x_cs + (ok ? u : 0)]);
    }
    // ---- stage dy (64 rows x 128 positions) and x (+ 8 elements of left pad, of which the last is the halo x[t0 - 1])
    if ((T & 1) == 0) {
        // even T (every map of the models): a position pair is inside or outside as a whole.  All 32 loads of a thread are
        // issued from CLAMPED addresses before the first value is touched and rows / pairs outside the tensors are masked
        // afterwards -- with `if (inside) load` the compiler emitted a branch and a full s_waitcnt per load: 32 dependent
        // memory round trips per workgroup, most of this kernel's ~28 us (tools/isa_loads.sh)
        const int tl = (tid & 63) * 2, r0 = tid >> 6;
        const bool in = t0 + tl < T;
        const int tlc = in ? tl : 0;                         // (t0 < T: position t0 exists)
        uint2 dv[16], xv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = r0 + 4 * it;
            dv[it] = *reinterpret_cast<const uint2*>(dyb + (int64_t)min(co0 + r, g.Cout - 1) * g.y_cs + tlc);
            xv[it] = *reinterpret_cast<const uint2*>(xb + (int64_t)min(ci0 + r, g.Cin - 1) * g.x_cs + tlc);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = r0 + 4 * it;
            const unsigned kd = (in && co0 + r < g.Cout) ? ~0u : 0u, kx = (in && ci0 + r < g.Cin) ? ~0u : 0u;
            *reinterpret_cast<unsigned*>(sdy + r * W1_PITCH + tl * 2) = cvt_pk_bf16(__uint_as_float(dv[it].x & kd), __uint_as_float(dv[it].y & kd));
            *reinterpret_cast<unsigned*>(sx + r * W1_PITCH + (8 + tl) * 2) = cvt_pk_bf16(__uint_as_float(xv[it].x & kx), __uint_as_float(xv[it].y & kx));
        }
    } else
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int r = idx >> 6, tl = (idx & 63) * 2;
        const bool rowd = co0 + r < g.Cout, rowx = ci0 + r < g.Cin;
        const bool in0 = t0 + tl < T, in1 = t0 + tl + 1 < T;
        float2 d = make_float2(0.f, 0.f), v = make_float2(0.f, 0.f);
        if (rowd && in1) d = *reinterpret_cast<const float2*>(dyb + (int64_t)(co0 + r) * g.y_cs + tl);
        else if (rowd && in0) d.x = dyb[(int64_t)(co0 + r) * g.y_cs + tl];
        if (rowx && in1) v = *reinterpret_cast<const float2*>(xb + (int64_t)(ci0 + r) * g.x_cs + tl);
        else if (rowx && in0) v.x = xb[(int64_t)(ci0 + r) * g.x_cs + tl];
        *reinterpret_cast<unsigned*>(sdy + r * W1_PITCH + tl * 2) = cvt_pk_bf16(d.x, d.y);
        *reinterpret_cast<unsigned*>(sx + r * W1_PITCH + (8 + tl) * 2) = cvt_pk_bf16(v.x, v.y);
    }
    if (tid < 128) {
        const int r = tid & 63, right = tid >> 6;
        const float v = __uint_as_float(halo_raw & halo_keep);
        *reinterpret_cast<unsigned*>(sx + r * W1_PITCH + (right ? 136 : 6) * 2) = right ? cvt_pk_bf16(v, 0.f) : cvt_pk_bf16(0.f, v);
    }
    if (tid < W1_TC / 2) {  // a tap shifted by -1 is cut where t starts a level, one shifted by +1 where t ends one
        unsigned ms = 0xffffffffu, me = 0xffffffffu;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = t0 + 2 * tid + e;
            int lo, up;
            level_bounds(g, t < T ? t : T - 1, T, lo, up);
            if (t == lo) ms &= e ? 0x0000ffffu : 0xffff0000u;
            if (t + 1 == up) me &= e ? 0x0000ffffu : 0xffff0000u;
        }
        mstart[tid] = ms; mend[tid] = me;
    }
    __syncthreads();
#pragma unroll 2
    for (int k = 0; k < W1_TC / 16; ++k) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + k * 32);
        const Words4 c = *reinterpret_cast<const Words4*>(brow + k * 32);                   // x[t .. t+7]
        if constexpr (KT == 1) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, c), acc[0], 0, 0, 0);
        } else {
            const unsigned left = *reinterpret_cast<const unsigned*>(brow + k * 32 - 4);    // x[t-2], x[t-1]
            const unsigned right = *reinterpret_cast<const unsigned*>(brow + k * 32 + 16);   // x[t+8], x[t+9]
            const Words4 ms = *reinterpret_cast<const Words4*>(reinterpret_cast<const unsigned char*>(mstart) + k * 32 + h8 * 2);
            const Words4 me = *reinterpret_cast<const Words4*>(reinterpret_cast<const unsigned char*>(mend) + k * 32 + h8 * 2);
            Words4 m1, p1;                                                                   // x[t-1 .. t+6], x[t+1 .. t+8]
            m1.a = __builtin_amdgcn_alignbit(c.a, left, 16) & ms.a; m1.b = __builtin_amdgcn_alignbit(c.b, c.a, 16) & ms.b;
            m1.c = __builtin_amdgcn_alignbit(c.c, c.b, 16) & ms.c;  m1.d = __builtin_amdgcn_alignbit(c.d, c.c, 16) & ms.d;
            p1.a = __builtin_amdgcn_alignbit(c.b, c.a, 16) & me.a;  p1.b = __builtin_amdgcn_alignbit(c.c, c.b, 16) & me.b;
            p1.c = __builtin_amdgcn_alignbit(c.d, c.c, 16) & me.c;  p1.d = __builtin_amdgcn_alignbit(right, c.d, 16) & me.d;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, m1), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, c), acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, p1), acc[2], 0, 0, 0);
        }
    }
    }
    // slab[split][co][ci * KT + dt]
    const int ci = ci0 + ni * 32 + (lane & 31);
    float* slab = (second ? a.slab2 : a.slab) + (int64_t)split * a.M * a.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < g.Cout && ci < g.Cin) {
#pragma unroll
            for (int d = 0; d < KT; ++d) slab[(int64_t)co * a.N + ci * KT + d] = acc[d][r];
        }
    }
}

static inline int wgrad1d_chunks(const ConvGeom& g) { return (g.Ti + W1_TC - 1) / W1_TC; }
static inline bool wgrad1d_eligible(const ConvGeom& g, int prec, const void* x, const void* dy) {
    if (!prec || OTAL_OPT("OTAL_CONV_NOW1D", 0)) return false;
    if (g.Hi != 1 || g.Wi != 1 || g.Ho != 1 || g.Wo != 1 || g.kh != 1 || g.kw != 1 || g.st != 1 || g.To != g.Ti) return false;
    if (!((g.kt == 1 && g.pt == 0) || (g.kt == 3 && g.pt == 1))) return false;
    if (g.Cin % 64 || (g.x_bs | g.x_cs | g.y_bs | g.y_cs) & 1) return false;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 7) return false;
    return (int64_t)g.B * wgrad1d_chunks(g) <= 1024;
}

int launch_wgrad1d(ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    const int upw_env = OTAL_OPT("OTAL_W1D_UPW", 0);
    const int nchunks = wgrad1d_chunks(a.g), units = a.g.B * nchunks;
    const int tiles = ((a.g.Cout + 63) / 64) * (a.g.Cin / 64);
    // units per workgroup = (sample, chunk) units folded into one split-K slab.  Round 2 (every reduction its own launch behind
    // its GEMM, one lane): 1 -> 478.4, 2 -> 475.8, 4 -> 465.7 clips/s.  Round 6 (reductions batched on the weight-gradient lane, whose
    // slab traffic is what these eight launches cost -- 328 MB written, 328 MB read back per step): 1 / 2 / 4 / 8 -> 8.09 / 8.03 /
    // 8.02 / 8.02 ms.  Four where there are eight units or more (two slabs at b = 8), two from four units on.
    const int upw = upw_env > 0 ? upw_env : (units >= 8 ? 4 : units >= 4 ? 2 : 1);
    (void)tiles;
    const int splits = (units + upw - 1) / upw;
    const size_t need = ((size_t)splits * a.M * a.N * sizeof(float) + 255) & ~(size_t)255;
    if (!ws || ws_bytes < need * (a.pair ? 2 : 1)) return OTAL_E_UNSUPPORTED;
    a.splits = splits; a.k_per_split = 0; a.slab = reinterpret_cast<float*>(ws);
    if (a.pair) a.slab2 = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + need);
    const dim3 grid(splits, a.g.Cin / 64, ((a.g.Cout + 63) / 64) * (a.pair ? 2 : 1));
    if (a.g.kt == 1) hipLaunchKernelGGL(conv_wgrad1d_kernel<1>, grid, dim3(256), 0, st, a, nchunks, units, upw);
    else hipLaunchKernelGGL(conv_wgrad1d_kernel<3>, grid, dim3(256), 0, st, a, nchunks, units, upw);
    if (int e = otal_launch_status()) return e;
    if (int e = launch_splitk_reduce<MODE_WGRAD>(a, st)) return e;
    if (a.pair) {               // the second problem's slabs: its own reduction (recorded with the first's when deferred)
        ConvArgs b = a;
        b.slab = a.slab2; b.out = a.out2;
        return launch_splitk_reduce<MODE_WGRAD>(b, st);
    }
    return 0;
}

// ---- chunked bf16 path: eligibility, workspace layout [chunk table][packed bf16 weights][split-K slabs]
constexpr int CHUNK_PAD = 16;       // table entries readable past Kp/8 (two K steps of prefetch)
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int chunk_kp(int K) { return (K + 31) / 32 * 32; }
static inline size_t chunk_tab_bytes(int K) { return align256(((size_t)chunk_kp(K) / 8 + CHUNK_PAD) * sizeof(int2)); }
static inline size_t chunk_wp_bytes(int M, int BM, int K) {      // + one K step so the prefetch past Kp stays inside
    return align256(((size_t)((M + BM - 1) / BM * BM) * chunk_kp(K) + 64) * sizeof(unsigned short));
}
// extent in bytes of the tensor the gather reads (channel-sliced views: strides come from the caller)
static inline int64_t gather_extent_bytes(const ConvGeom& g, int mode, int esz = 4) {
    if (mode == MODE_FWD) return esz * ((int64_t)(g.B - 1) * g.x_bs + (int64_t)(g.Cin - 1) * g.x_cs + conv_in_positions(g));
    return esz * ((int64_t)(g.B - 1) * g.y_bs + (int64_t)(g.Cout - 1) * g.y_cs + conv_out_positions(g));
}
// bf16-stored tensors around a layer (H kernels): 16-byte runs of eight positions must stay inside a sample and be aligned
static inline bool half_layout_ok(const ConvGeom& g, const void* x, const void* y) {
    return conv_out_positions(g) % 8 == 0 && conv_in_positions(g) % 8 == 0 && g.x_bs % 8 == 0 && g.x_cs % 8 == 0 &&
           g.y_bs % 8 == 0 && g.y_cs % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}
static inline bool chunk_eligible(const ConvGeom& g, int mode, int prec) {
    if (!prec || mode == MODE_WGRAD) return false;
    if (OTAL_OPT("OTAL_CONV_NOCHUNK", 0)) return false;
    const int C = mode == MODE_FWD ? g.Cin : g.Cout;
    if (C % 8 && !(mode == MODE_FWD && g.kw >= 3 && !OTAL_OPT("OTAL_CONV_NOKWV", 0))) return false;    // forward has the kw-vector mode
    const int64_t ext = gather_extent_bytes(g, mode);
    return ext > 0 && ext < (int64_t)0xfffffff0u;       // 32-bit buffer offsets
}

// positions one thread may fetch with a single vector load (see conv_gemm_bf16c_kernel)
static inline int chunk_vector_width(const ConvGeom& g) {
    if (OTAL_OPT("OTAL_CONV_CW", 0) == 1) return 1;
    if (g.st != 1 || g.sh != 1 || g.sw != 1 || g.nlev > 1) return 1;
    if (g.Wo != g.Wi || (g.kw != 1 && g.kw != 3) || g.pw != (g.kw - 1) / 2) return 1;
    // 1x1x1: no shifted tap, so 4 consecutive positions of a sample are contiguous across row ends as well
    if (g.kt == 1 && g.kh == 1 && g.kw == 1 && g.To == g.Ti && g.Ho == g.Hi && conv_out_positions(g) % 4 == 0 &&
        !OTAL_OPT("OTAL_CONV_NO1X1V4", 0)) return 4;
    if (g.Wi % 4 == 0) return 4;
    if (g.Wi % 2 == 0) return 2;
    return 1;
}

// descriptor of a chunk-path prologue (a.K must already be the path's K: kw-vector mode pads it)
template <int MODE>
static void fill_prep_desc(PrepDesc& d, const ConvArgs& a, int2* ctab, unsigned short* wp) {
    const int C = MODE == MODE_FWD ? a.g.Cin : a.g.Cout;
    const bool kwv = MODE == MODE_FWD && (C % 8) != 0;
    const int BMsel = choose_bm(a.M, kwv ? 0 : 1);
    d.ctab = ctab; d.wp = reinterpret_cast<unsigned*>(wp); d.wsrc = a.w; d.g = a.g;
    d.M = a.M; d.Mpad = (a.M + BMsel - 1) / BMsel * BMsel; d.C = C; d.kvol = conv_kvol(a.g);
    d.K = a.K; d.Kp = chunk_kp(a.K); d.nchunk = d.Kp / 8 + CHUNK_PAD; d.kwv = kwv ? 1 : 0; d.natural = a.w_natural; d.mode = MODE;
    d.direct = 0;
    d.fh = make_fastdiv((uint32_t)(d.Kp / 2));
    d.fk = make_fastdiv((uint32_t)d.kvol);
}
static inline unsigned prep_blocks(const PrepDesc& d) {
    // the pack kernels index with 32 bits: a weight matrix of 2^31 bf16 pairs (8 GB) is refused (0 blocks = launch error)
    if (d.direct) {
        const int64_t pairs = (int64_t)d.Mpad * d.C * 27 / 2;
        if (pairs >= (1LL << 31)) return 0;
        const int64_t blocks = (pairs + 256 * PREP_U - 1) / (256 * PREP_U);
        return (unsigned)(blocks < 2048 ? blocks : 2048);
    }
    const int64_t pairs = (int64_t)d.Mpad * (d.Kp / 2);
    if (pairs >= (1LL << 31)) return 0;
    int64_t blocks = (pairs + 256 * PREP_U - 1) / (256 * PREP_U);
    if (d.mode == MODE_DGRAD && d.natural && d.kvol <= PREP_T_KVOL && !d.kwv) blocks = (int64_t)(d.Mpad / 32) * ((d.Kp + 255) / 256);   // tiles
    if (blocks < (d.nchunk + 255) / 256) blocks = (d.nchunk + 255) / 256;
    return (unsigned)(blocks > 2048 ? 2048 : blocks);
}
static void launch_prep(const PrepDesc& d, hipStream_t st) {
    if (d.direct) {
        if (d.mode == MODE_FWD) hipLaunchKernelGGL((pack_direct_kernel<MODE_FWD>), dim3(prep_blocks(d)), dim3(256), 0, st, d.wp, d.wsrc, d.M, d.Mpad, d.C, d.natural, d.fh);
        else hipLaunchKernelGGL((pack_direct_kernel<MODE_DGRAD>), dim3(prep_blocks(d)), dim3(256), 0, st, d.wp, d.wsrc, d.M, d.Mpad, d.C, d.natural, d.fh);
        return;
    }
    if (d.mode == MODE_FWD) hipLaunchKernelGGL((prep_chunks_kernel<MODE_FWD>), dim3(prep_blocks(d)), dim3(256), 0, st, d);
    else hipLaunchKernelGGL((prep_chunks_kernel<MODE_DGRAD>), dim3(prep_blocks(d)), dim3(256), 0, st, d);
}

#if OTAL_CONV_PART == 1
typedef short v4s16 __attribute__((ext_vector_type(4)));
#include "conv1x1_stream.inc"
#endif

template <int MODE, bool H = false>
int launch_chunked(ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    const int C = MODE == MODE_FWD ? a.g.Cin : a.g.Cout;
    const bool kwv = MODE == MODE_FWD && (C % 8) != 0;
    if (H && (kwv || (a.flags & EPI_ACCUM) || conv_in_positions(a.g) != conv_out_positions(a.g))) return OTAL_E_UNSUPPORTED;
    if (kwv) a.K = a.g.Cin * a.g.kt * a.g.kh * 8;          // kw padded to 8 taps per (ci, dt, dh) row
    const int BMpack = choose_bm(a.M, kwv ? 0 : 1);           // the weight pack's row padding (persistent regions are sized by it)
    int BMsel = BMpack;
    // bf16 tensors: the 128-row variant needs 247 registers (two workgroups per CU), the 96-row one 105 (four); rows past the
    // pack's padding read zeros through the buffer bounds check, so a different tile height may run on the same pack
    if (H && BMsel == 128 && OTAL_OPT("OTAL_CHUNK_H_NO128", 1)) BMsel = 96;
    const int tm = (a.M + BMsel - 1) / BMsel, tn = (a.N + 127) / 128;
    a.Kp = chunk_kp(a.K);
    const size_t tb = chunk_tab_bytes(a.K), wb = chunk_wp_bytes(a.M, BMpack, a.K);
    int2* ctab;
    unsigned short* wp;
    if (a.pre) {            // tables + packed weights were prepared into a caller-owned region
        ctab = reinterpret_cast<int2*>(const_cast<void*>(a.pre));
        wp = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(const_cast<void*>(a.pre)) + tb);
    } else {
        if (!ws || ws_bytes < tb + wb) return OTAL_E_UNSUPPORTED;
        ctab = reinterpret_cast<int2*>(ws);
        wp = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(ws) + tb);
        PrepDesc d;
        fill_prep_desc<MODE>(d, a, ctab, wp);
        launch_prep(d, st);
        if (int e = otal_launch_status()) return e;
        ws = reinterpret_cast<char*>(ws) + tb + wb;
        ws_bytes -= tb + wb;
    }
    a.ctab = ctab; a.wp = wp;
#if OTAL_CONV_PART == 1
    if constexpr (H) {
        // the 1x1x1 layers on bf16 tensors: the streaming kernel (conv1x1_stream.inc) reads the same weight pack
        if (conv1x1_stream_eligible(a.g, MODE)) {
            const int e = launch_conv1x1_stream<MODE>(a, wp, wb, (a.M + BMpack - 1) / BMpack * BMpack, st);
            if (e != OTAL_E_UNSUPPORTED) return e;
        }
    }
#endif
    a.src_bytes = (unsigned)gather_extent_bytes(a.g, MODE, H ? 2 : 4);
    a.wp_bytes = (unsigned)wb;
    a.fd = make_conv_fastdiv(a.g);
    int splits = choose_splits(tm * tn, a.Kp);
    if (splits > 1) {
        const size_t need = (size_t)splits * a.M * a.N * sizeof(float);
        if (ws_bytes < need) {
            splits = (int)(ws_bytes / ((size_t)a.M * a.N * sizeof(float)));
            if (splits < 2) splits = 1;
        }
    }
    const int kps = ((a.Kp + splits - 1) / splits + 31) / 32 * 32;
    splits = (a.Kp + kps - 1) / kps;
    a.splits = splits;
    a.k_per_split = kps;
    a.slab = splits > 1 ? (float*)ws : nullptr;
    set_epilogue_extents<MODE>(a);
#ifdef OTAL_DIRECT_ABLATE
    a.flags |= (OTAL_OPT("OTAL_CONV_DEBUG", 0) & (DBG_NOLOAD | 64 | 128));
#endif
    const dim3 grid(tn, tm, splits);
    const int cw = chunk_vector_width(a.g);
#define OTAL_LAUNCH_C(BM_, WM_, WN_)                                                                                   \
    do {                                                                                                               \
        if (cw == 4) hipLaunchKernelGGL((conv_gemm_bf16c_kernel<BM_, WM_, WN_, MODE, 4, false, H>), grid, dim3(NT), 0, st, a);   \
        else if (cw == 2) hipLaunchKernelGGL((conv_gemm_bf16c_kernel<BM_, WM_, WN_, MODE, 2, false, H>), grid, dim3(NT), 0, st, a); \
        else hipLaunchKernelGGL((conv_gemm_bf16c_kernel<BM_, WM_, WN_, MODE, 1, false, H>), grid, dim3(NT), 0, st, a);           \
    } while (0)
    if constexpr (MODE == MODE_FWD && !H) {
        if (kwv) {
            if (BMsel == 128) hipLaunchKernelGGL((conv_gemm_bf16c_kernel<128, 2, 2, MODE_FWD, 1, true>), grid, dim3(NT), 0, st, a);
            else if (BMsel == 96) hipLaunchKernelGGL((conv_gemm_bf16c_kernel<96, 3, 1, MODE_FWD, 1, true>), grid, dim3(NT), 0, st, a);
            else if (BMsel == 64) hipLaunchKernelGGL((conv_gemm_bf16c_kernel<64, 2, 1, MODE_FWD, 1, true>), grid, dim3(NT), 0, st, a);
            else hipLaunchKernelGGL((conv_gemm_bf16c_kernel<32, 1, 1, MODE_FWD, 1, true>), grid, dim3(NT), 0, st, a);
        }
    }
    if (kwv) {
    } else if (BMsel == 192) OTAL_LAUNCH_C(192, 6, 1);
    else if (BMsel == 128) OTAL_LAUNCH_C(128, 2, 2);
    else if (BMsel == 96) OTAL_LAUNCH_C(96, 3, 1);
    else if (BMsel == 64) OTAL_LAUNCH_C(64, 2, 1);
    else OTAL_LAUNCH_C(32, 1, 1);
#undef OTAL_LAUNCH_C
    if (int e = otal_launch_status()) return e;
    if (splits > 1) {
        return launch_splitk_reduce<MODE, H>(a, st);
    }
    return 0;
}

// ---- vector WGRAD: eligibility, workspace layout [position table][split-K slabs]
constexpr int PTAB_PAD = 64;        // entries readable past the last group (two K steps of prefetch at CW = 2 -> 32)
// stride-2 pair mode of the vector WGRAD (Conv3d_1a): window ends must stay within 4 elements of the row
static inline bool wgrad_pair_mode(const ConvGeom& g, int prec) {
    if (!prec || OTAL_OPT("OTAL_CONV_NOVEC_WGRAD", 0) || g.nlev > 1) return false;
    if (g.sw != 2 || g.st > 2 || g.sh > 2 || g.kw > 7 || g.pw > 3 || g.Wo % 8 || conv_out_positions(g) % 32) return false;
    if (g.Wi - (2 * (g.Wo - 8) - g.pw + 6) < 12) return false;      // last window: elements 0..11 inside the row
    const int64_t ex = 4 * ((int64_t)(g.B - 1) * g.x_bs + (int64_t)(g.Cin - 1) * g.x_cs + conv_in_positions(g));
    const int64_t ey = 4 * ((int64_t)(g.B - 1) * g.y_bs + (int64_t)(g.Cout - 1) * g.y_cs + conv_out_positions(g));
    return ex > 0 && ey > 0 && ex < (int64_t)0xffffff00u && ey < (int64_t)0xfffffff0u;
}
static inline int wgrad_vector_width(const ConvGeom& g, int prec) {
    if (!prec || OTAL_OPT("OTAL_CONV_NOVEC_WGRAD", 0)) return 0;
    if (g.st != 1 || g.sh != 1 || g.sw != 1 || g.nlev > 1) return 0;
    if (g.To != g.Ti || g.Ho != g.Hi || g.Wo != g.Wi) return 0;
    if ((g.kw != 1 && g.kw != 3) || g.pw != (g.kw - 1) / 2) return 0;
    if (conv_out_positions(g) % 32) return 0;
    const int64_t ex = 4 * ((int64_t)(g.B - 1) * g.x_bs + (int64_t)(g.Cin - 1) * g.x_cs + conv_in_positions(g));
    const int64_t ey = 4 * ((int64_t)(g.B - 1) * g.y_bs + (int64_t)(g.Cout - 1) * g.y_cs + conv_out_positions(g));
    if (ex <= 0 || ey <= 0 || ex >= (int64_t)0xfffffff0u || ey >= (int64_t)0xfffffff0u) return 0;
    // a 1x1x1 kernel has no shifted tap: any 8 consecutive positions of a sample are one contiguous vector, whatever the row
    // length (6x6 planes were on 2-element vectors, 3x3 planes on the generic kernel)
    if (g.kt == 1 && g.kh == 1 && g.kw == 1 && !OTAL_OPT("OTAL_CONV_NO1X1V8", 0)) return 8;
    if (g.Wi % 8 == 0) return 8;
    if (g.Wi % 4 == 0) return 4;
    if (g.Wi % 2 == 0) return 2;
    return 0;
}
static inline size_t ptab_bytes(const ConvGeom& g, int cw) {
    return align256(((size_t)g.B * conv_out_positions(g) / cw + PTAB_PAD) * sizeof(int2));
}

template <bool H = false>
int launch_wgrad_vector(ConvArgs& a, int cw, void* ws, size_t ws_bytes, hipStream_t st) {
    const bool pair = a.g.sw == 2;
    if (H && pair) return OTAL_E_UNSUPPORTED;
    if (pair) {                                             // columns padded to 8 per (ci, dt, dh) row
        a.N = a.g.Cin * a.g.kt * a.g.kh * 8;
        a.flags |= EPI_NPAD8;
    }
    int BMsel = choose_bm(a.M, pair ? 0 : 2);
    {   // (experiment knob: a smaller row block = a smaller register / LDS footprint per workgroup beside the main lane's kernels)
        const int cap = OTAL_OPT("OTAL_WGRADV_MAXBM", 192);
        while (BMsel > cap && BMsel > 32) BMsel = BMsel == 192 ? 128 : (BMsel == 128 ? 96 : (BMsel == 96 ? 64 : 32));
    }
    const int tm = (a.M + BMsel - 1) / BMsel, tn = (a.N + 127) / 128;
    const size_t tb = ptab_bytes(a.g, cw);
    a.fd = make_conv_fastdiv(a.g);
    a.P = conv_out_positions(a.g);
    a.src_bytes = (unsigned)gather_extent_bytes(a.g, MODE_FWD, H ? 2 : 4);
    a.dy_bytes = (unsigned)gather_extent_bytes(a.g, MODE_DGRAD, H ? 2 : 4);
    int2* ptab;
    if (a.pre) {
        ptab = reinterpret_cast<int2*>(const_cast<void*>(a.pre));      // geometry-only table, built once by the caller
    } else {
        if (!ws || ws_bytes < tb) return OTAL_E_UNSUPPORTED;
        ptab = reinterpret_cast<int2*>(ws);
        const int ngroups = a.K / cw, npad = ngroups + PTAB_PAD;
        hipLaunchKernelGGL(build_pos_table_kernel, dim3((npad + 255) / 256), dim3(256), 0, st, ptab, a.g, a.fd, cw, ngroups, npad);
        if (int e = otal_launch_status()) return e;
        ws = reinterpret_cast<char*>(ws) + tb;
        ws_bytes -= tb;
    }
    a.ptab = ptab;
    int splits = choose_splits(tm * tn, a.K, 1, true);
    if (splits > 1) {
        const size_t need = (size_t)splits * a.M * a.N * sizeof(float);
        if (ws_bytes < need) {
            splits = (int)(ws_bytes / ((size_t)a.M * a.N * sizeof(float)));
            if (splits < 2) splits = 1;
        }
    }
    const int kps = ((a.K + splits - 1) / splits + 31) / 32 * 32;
    splits = (a.K + kps - 1) / kps;
    a.splits = splits;
    a.k_per_split = kps;
    a.slab = splits > 1 ? (float*)ws : nullptr;
    set_epilogue_extents<MODE_WGRAD>(a);
    const dim3 grid(tn, tm, splits);
#define OTAL_LAUNCH_W(BM_, WM_, WN_)                                                                                   \
    do {                                                                                                               \
        if (cw == 8) hipLaunchKernelGGL((conv_wgrad_bf16v_kernel<BM_, WM_, WN_, 8, false, H>), grid, dim3(NT), 0, st, a);        \
        else if (cw == 4) hipLaunchKernelGGL((conv_wgrad_bf16v_kernel<BM_, WM_, WN_, 4, false, H>), grid, dim3(NT), 0, st, a);   \
        else hipLaunchKernelGGL((conv_wgrad_bf16v_kernel<BM_, WM_, WN_, 2, false, H>), grid, dim3(NT), 0, st, a);                \
    } while (0)
    if constexpr (!H) {
        if (pair) {
            if (BMsel == 128) hipLaunchKernelGGL((conv_wgrad_bf16v_kernel<128, 2, 2, 8, true>), grid, dim3(NT), 0, st, a);
            else if (BMsel == 96) hipLaunchKernelGGL((conv_wgrad_bf16v_kernel<96, 3, 1, 8, true>), grid, dim3(NT), 0, st, a);
            else if (BMsel == 64) hipLaunchKernelGGL((conv_wgrad_bf16v_kernel<64, 2, 1, 8, true>), grid, dim3(NT), 0, st, a);
            else hipLaunchKernelGGL((conv_wgrad_bf16v_kernel<32, 1, 1, 8, true>), grid, dim3(NT), 0, st, a);
        }
    }
    if (pair) {
    } else if (BMsel == 192) OTAL_LAUNCH_W(192, 6, 1);
    else if (BMsel == 128) OTAL_LAUNCH_W(128, 2, 2);
    else if (BMsel == 96) OTAL_LAUNCH_W(96, 3, 1);
    else if (BMsel == 64) OTAL_LAUNCH_W(64, 2, 1);
    else OTAL_LAUNCH_W(32, 1, 1);
#undef OTAL_LAUNCH_W
    if (int e = otal_launch_status()) return e;
    if (splits > 1) {
        return launch_splitk_reduce<MODE_WGRAD>(a, st);
    }
    return 0;
}

// ---- direct 3x3x3 path: eligibility and launch
static inline int direct_bm(const ConvGeom& g, int M) {
    if (M % 96 == 0) return 96;
    // one workgroup per CU and launch round: where 64-row tiles of 256 positions need a second, nearly empty round (the 6x6
    // planes of Mixed_4b..4d b1b forward: 4 x 72 = 288 workgroups) and 96-row tiles do not (3 x 72 = 216), the padded rows
    // are cheaper than the round -- forward 43 / 68 / 76 us on 64-row tiles against 56 us for Mixed_4e's 216 tiles of 96
    if (M > 96 && !OTAL_OPT("OTAL_CONV_DIRECT_NOROUNDS", 0)) {
        const int64_t nt = (int64_t)g.B * conv_out_positions(g) / 256;
        const int64_t w64 = (M + 63) / 64 * nt, w96 = (M + 95) / 96 * nt;
        if (w64 <= 512 && (w96 + 255) / 256 * 96 < (w64 + 255) / 256 * 64) return 96;
    }
    if (M % 64 == 0) return 64;
    // 16 .. 32 rows (data gradient of the Inception b2b layers: M = Cin = 16 / 24 / 32; forward of Mixed_3b.b2b): one 32-row
    // MFMA tile per wave.  LDS-read-bound (9 weight + 9 position fragments per 9 MFMAs), but these layers are tiny and ran
    // on the gather kernel at 25 .. 90 us for 0.1 .. 1 GFLOP of work per sample
    if (M <= 32 && !OTAL_OPT("OTAL_CONV_DIRECT_NO32", 0)) return 32;
    const int pad64 = (M + 63) / 64 * 64;
    return (pad64 - M) * 100 <= M * OTAL_OPT("OTAL_CONV_DIRECT_PAD", 34) ? 64 : 0;     // accept <= 34 % padded rows (Mixed_4e: 144 -> 192)
}
// positions per workgroup: 256 (8 waves), or 128 (4 waves) when 256 would leave the chip half empty (the 6x6 planes of
// Mixed_4x: 72 position tiles); 0 = too few tiles either way (no split-K on this path)
static inline int direct_bnp(const ConvGeom& g, int M) {
    const int BM = direct_bm(g, M);
    if (!BM) return 0;
    const int64_t tm = (M + BM - 1) / BM, NP = (int64_t)g.B * conv_out_positions(g);
    // (140: the 144 tiles of a one-M-tile layer on the 6x6 planes still take the 128-position form -- Mixed_4b / 4e b2b forward
    //  17.5 -> 10.1 us, 27.6 -> 12.9 us, Mixed_4b b1b data gradient 71 -> 54 us against the gather kernel; tools/micro_planes6.py)
    const int min_tiles = OTAL_OPT("OTAL_CONV_DIRECT_MINTILES", 140);
    // 512 positions (two tiles per wave: the weight fragments are shared, the kernel turns MFMA-bound) when that still gives
    // every CU two rounds of workgroups and the tile stays inside one sample
    // (96-row tiles only: a 64-row tile of 256 positions fits TWICE per CU -- 16 waves -- and measured faster than one 512 tile)
    if (BM == 96 && !OTAL_OPT("OTAL_CONV_DIRECT_NO512", 0) && conv_out_positions(g) % 512 == 0 &&
        tm * (NP / 512) >= OTAL_OPT("OTAL_CONV_DIRECT_MINTILES512", 512)) return 512;
    if (tm * (NP / 256) >= min_tiles) return 256;
    if (!OTAL_OPT("OTAL_CONV_DIRECT_NO128", 0) && tm * (NP / 128) >= min_tiles) return 128;
    return 0;
}
static inline bool direct_eligible(const ConvGeom& g, int mode, int prec, int M) {
    const bool off = OTAL_OPT("OTAL_CONV_NODIRECT", 0) != 0;
    if (off || !prec || mode == MODE_WGRAD || g.nlev > 1) return false;
    if (g.kt != 3 || g.kh != 3 || g.kw != 3 || g.st != 1 || g.sh != 1 || g.sw != 1 || g.pt != 1 || g.ph != 1 || g.pw != 1) return false;
    if (g.To != g.Ti || g.Ho != g.Hi || g.Wo != g.Wi || g.Wi > 24) return false;
    const int P = conv_out_positions(g);
    const int C = mode == MODE_FWD ? g.Cin : g.Cout;
    if (P % 256 || C % 16 || !direct_bnp(g, M)) return false;
    const int64_t ext = gather_extent_bytes(g, mode);
    return ext > 0 && ext < (1LL << 31);
}
static inline size_t direct_wp_bytes(const ConvGeom& g, int M, int C) {
    const int BM = direct_bm(g, M);
    return align256((size_t)((M + BM - 1) / BM * BM) * C * 27 * 2 + 1024);
}

template <int BM, int BNP, int PX = 48, bool H = false>
constexpr int direct_lds_bytes() { return 2 * BM * 304 + 2 * (BNP + (H ? 56 : 52)) * PX + 16; }
// four waves x two position tiles, dense LDS pitch: 78 KB (BM = 96) / 59 KB (BM = 64) -> two workgroups per CU
template <int BM, int MODE>
static int launch_direct256x2(const DirectArgs& d, dim3 grid, hipStream_t st) {
    constexpr int lds = direct_lds_bytes<BM, 256, 32>();
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_direct_kernel<BM, MODE, 256, 2, 32, 2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    hipLaunchKernelGGL((conv3_direct_kernel<BM, MODE, 256, 2, 32, 2>), grid, dim3(256), lds, st, d);
    return otal_launch_status();
}
// eight waves x one position tile, dense pitch, <= 128 registers: two 8-wave workgroups (16 waves) per CU
template <int BM, int MODE>
static int launch_direct256d(const DirectArgs& d, dim3 grid, hipStream_t st) {
    constexpr int lds = direct_lds_bytes<BM, 256, 32>();
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_direct_kernel<BM, MODE, 256, 1, 32, 4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    hipLaunchKernelGGL((conv3_direct_kernel<BM, MODE, 256, 1, 32, 4>), grid, dim3(512), lds, st, d);
    return otal_launch_status();
}
template <int BM, int MODE, bool XPF2 = false, bool H = false>
static int launch_direct512(const DirectArgs& d, dim3 grid, hipStream_t st) {
    constexpr int lds = direct_lds_bytes<BM, 512, 48, H>();
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_direct_kernel<BM, MODE, 512, 2, 48, 1, XPF2, H>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    hipLaunchKernelGGL((conv3_direct_kernel<BM, MODE, 512, 2, 48, 1, XPF2, H>), grid, dim3(512), lds, st, d);
    return otal_launch_status();
}

template <int MODE, bool H = false>
int launch_direct(ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    if (H && ((a.g.Wi & 1) || (a.flags & EPI_ACCUM))) return OTAL_E_UNSUPPORTED;       // (4-byte aligned loads need an even row length)
    const int C = MODE == MODE_FWD ? a.g.Cin : a.g.Cout;
    const int BM = direct_bm(a.g, a.M);
    const int tm = (a.M + BM - 1) / BM, Mpad = tm * BM;
    const size_t wb = direct_wp_bytes(a.g, a.M, C);
    if (a.pre) {            // packed earlier into a caller-owned region (otal_conv_prologue[_batch]): nothing to do per launch
        ws = const_cast<void*>(a.pre);
    } else {
        if (!ws || ws_bytes < wb) return OTAL_E_UNSUPPORTED;
        const int64_t pairs = (int64_t)Mpad * C * 27 / 2;
        const int blocks = (int)((pairs + 255) / 256 < 2048 ? (pairs + 255) / 256 : 2048);
        hipLaunchKernelGGL((pack_direct_kernel<MODE>), dim3(blocks), dim3(256), 0, st, reinterpret_cast<unsigned*>(ws), a.w,
                           a.M, Mpad, C, a.w_natural, make_fastdiv((uint32_t)(C * 27 / 2)));
        if (int e = otal_launch_status()) return e;
    }
    DirectArgs d;
    a.fd = make_conv_fastdiv(a.g);
    a.src_bytes = (unsigned)gather_extent_bytes(a.g, MODE, H ? 2 : 4);
    a.splits = 1; a.k_per_split = 0; a.slab = nullptr;
    set_epilogue_extents<MODE>(a);
#ifdef OTAL_DIRECT_ABLATE
    a.flags |= (OTAL_OPT("OTAL_CONV_DEBUG", 0) & (DBG_NOLOAD | DBG_NOSTORE | DBG_NOBARRIER | 64 | 128));
#endif
    d.c = a;
    d.wp = reinterpret_cast<const unsigned short*>(ws);
    d.C = C; d.Ktot = C * 27; d.wp_bytes = (unsigned)wb;
    const int bnp = direct_bnp(a.g, a.M);
    if (bnp == 128) {
        const dim3 grid(a.N / 128, tm, 1);
        if (BM == 96) hipLaunchKernelGGL((conv3_direct_kernel<96, MODE, 128, 1, 48, 1, false, H>), grid, dim3(256), (direct_lds_bytes<96, 128, 48, H>()), st, d);
        else if (BM == 32) hipLaunchKernelGGL((conv3_direct_kernel<32, MODE, 128, 1, 48, 1, false, H>), grid, dim3(256), (direct_lds_bytes<32, 128, 48, H>()), st, d);
        else hipLaunchKernelGGL((conv3_direct_kernel<64, MODE, 128, 1, 48, 1, false, H>), grid, dim3(256), (direct_lds_bytes<64, 128, 48, H>()), st, d);
        return otal_launch_status();
    }
    // positions two K steps ahead (XPF2): bit 0 -- 96 x 256 tiles, and 64 x 256 tiles when the grid has at most one workgroup
    // per CU (the variant is one workgroup per CU instead of two: it only pays where nobody would share the CU anyway);
    // bit 1 -- 96 x 512 (forward); bit 2 -- every 64 x 256 launch.  Measured (tools/xpf_sweep.sh): Conv3d_2c fwd 505 -> 499 us,
    // Mixed_3c.b1b fwd 239 -> 232, Mixed_3b.b1b dgrad 158 -> 150, Mixed_4c .. 4f b1b dgrad 309 -> 285 (sum); with bit 2
    // Conv3d_2c dgrad 476 -> 558 and the 64-row forward launches +15 %: the load latency was NOT what these tiles wait for
    const int xpf2 = OTAL_OPT("OTAL_CONV_DIRECT_XPF2", 3);
    if (bnp == 512) {           // two position tiles per wave (dynamic LDS: 112 KB)
        const dim3 grid(a.N / 512, tm, 1);
        if constexpr (MODE == MODE_FWD) {
            if (BM == 96 && (xpf2 & 2)) return launch_direct512<96, MODE, true, H>(d, grid, st);
        }
        if (BM == 96) return launch_direct512<96, MODE, false, H>(d, grid, st);
        return launch_direct512<64, MODE, false, H>(d, grid, st);
    }
    const dim3 grid(a.N / 256, tm, 1);
    if constexpr (!H) {         // experiments: fp32 tensors only
        if (OTAL_OPT("OTAL_CONV_DIRECT_256X2", 0) == 1 && BM != 32) {
            if (BM == 96) return launch_direct256x2<96, MODE>(d, grid, st);
            return launch_direct256x2<64, MODE>(d, grid, st);
        }
        if (OTAL_OPT("OTAL_CONV_DIRECT_256X2", 0) == 2 && BM == 96) return launch_direct256d<96, MODE>(d, grid, st);
    }
    if (BM == 96 && (xpf2 & 1)) hipLaunchKernelGGL((conv3_direct_kernel<96, MODE, 256, 1, 48, 1, true, H>), grid, dim3(512), (direct_lds_bytes<96, 256, 48, H>()), st, d);
    else if (BM == 64 && ((xpf2 & 4) || ((xpf2 & 1) && (int64_t)grid.x * grid.y <= 256))) hipLaunchKernelGGL((conv3_direct_kernel<64, MODE, 256, 1, 48, 1, true, H>), grid, dim3(512), (direct_lds_bytes<64, 256, 48, H>()), st, d);
    else if (BM == 96) hipLaunchKernelGGL((conv3_direct_kernel<96, MODE, 256, 1, 48, 1, false, H>), grid, dim3(512), (direct_lds_bytes<96, 256, 48, H>()), st, d);
    else if (BM == 32) hipLaunchKernelGGL((conv3_direct_kernel<32, MODE, 256, 1, 48, 1, false, H>), grid, dim3(512), (direct_lds_bytes<32, 256, 48, H>()), st, d);
    else hipLaunchKernelGGL((conv3_direct_kernel<64, MODE, 256, 1, 48, 1, false, H>), grid, dim3(512), (direct_lds_bytes<64, 256, 48, H>()), st, d);
    return otal_launch_status();
}

#include "conv_wgrad_direct.inc"
#include "conv1d_tile.inc"
#include "conv1a_wgrad.inc"
#include "proj_gemm.inc"
#include "wgrad1x1.inc"

// Which H kernel serves a launch with bf16-stored tensors on both sides (0: none).  fwd / dgrad: 1 direct 3x3x3, 2 chunked;
// wgrad: 3 direct, 4 wide 1x1, 5 vector (the number is the vector width + 16).  Shared by part 1's dispatcher and part 0's
// otal_conv_half_storage() query; x / dy may be null (the query has no tensors: alignment is then the caller's contract).
static inline int half_kernel_kind(const ConvGeom& g, int mode, const void* x, const void* dy) {
    if (g.nlev > 1 || OTAL_OPT("OTAL_CONV_NOHALF", 0)) return 0;
    if (conv_out_positions(g) % 8 || conv_in_positions(g) % 8 || g.x_bs % 8 || g.x_cs % 8 || g.y_bs % 8 || g.y_cs % 8) return 0;
    if (mode == MODE_FWD || mode == MODE_DGRAD) {
        const int M = mode == MODE_FWD ? g.Cout : g.Cin, C = mode == MODE_FWD ? g.Cin : g.Cout;
        if (direct_eligible(g, mode, 1, M) && !(g.Wi & 1)) return 1;
        if (chunk_eligible(g, mode, 1) && C % 8 == 0 && conv_in_positions(g) == conv_out_positions(g)) return 2;
        return 0;
    }
    if (wgrad_direct_eligible(g, 1, x, dy)) return 3;
    if (wgrad1x1_wide_eligible(g, 1, x, dy)) return 4;
    if (const int cw = wgrad_vector_width(g, 1)) return 16 + cw;
    return 0;
}

#if OTAL_CONV_PART == 1
// bf16-stored tensors on BOTH sides of a backbone layer (fwd: x and y; dgrad: dy, dx and the ReLU mask; wgrad: x and dy): the H
// instantiations of the direct 3x3x3, chunked, direct / wide / vector weight-gradient kernels.  Anything else: unsupported.
}  // namespace
int otal_conv::launch_half(int mode, ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    const ConvGeom& g = a.g;
    if (!a.prec || !a.xhalf || !a.half || (a.flags & EPI_ACCUM)) return OTAL_E_UNSUPPORTED;
    const void* px = mode == MODE_DGRAD ? (const void*)a.out : (const void*)a.x;
    const void* py = mode == MODE_FWD ? (const void*)a.out : (const void*)a.dy;
    if ((reinterpret_cast<uintptr_t>(px) | reinterpret_cast<uintptr_t>(py)) & 15) return OTAL_E_UNSUPPORTED;
    if (mode == MODE_DGRAD && a.emask && (!a.mhalf || (reinterpret_cast<uintptr_t>(a.emask) & 15))) return OTAL_E_UNSUPPORTED;
    const int kind = half_kernel_kind(g, mode, px, py);
    // a persistent prologue region holds what the fp32-tensor launch of this geometry would use (prologue_kind): the direct
    // kernel's weight pack wherever that kernel is eligible -- not what the chunked kernel reads
    if (kind == 2 && direct_eligible(g, mode, 1, mode == MODE_FWD ? g.Cout : g.Cin)) a.pre = nullptr;
    if (mode == MODE_FWD) {
        if (kind == 1) return launch_direct<MODE_FWD, true>(a, ws, ws_bytes, st);
        if (kind == 2) return launch_chunked<MODE_FWD, true>(a, ws, ws_bytes, st);
        return OTAL_E_UNSUPPORTED;
    }
    if (mode == MODE_DGRAD) {
        if (kind == 1) return launch_direct<MODE_DGRAD, true>(a, ws, ws_bytes, st);
        if (kind == 2) return launch_chunked<MODE_DGRAD, true>(a, ws, ws_bytes, st);
        return OTAL_E_UNSUPPORTED;
    }
    if (kind == 3) {
        const int e = launch_wgrad_direct<true>(a, ws, ws_bytes, st);
        if (e != OTAL_E_UNSUPPORTED) return e;              // slabs do not fit: the vector kernel
        if (const int cw = wgrad_vector_width(g, 1)) return launch_wgrad_vector<true>(a, cw, ws, ws_bytes, st);
        return e;
    }
    if (kind == 4) {
        const int e = launch_wgrad1x1_wide<true>(a, ws, ws_bytes, st);
        if (e != OTAL_E_UNSUPPORTED) return e;
        if (const int cw = wgrad_vector_width(g, 1)) return launch_wgrad_vector<true>(a, cw, ws, ws_bytes, st);
        return e;
    }
    if (kind > 16) return launch_wgrad_vector<true>(a, kind - 16, ws, ws_bytes, st);
    return OTAL_E_UNSUPPORTED;
}
namespace {
#endif

#if OTAL_CONV_PART == 0
template <int MODE>
int launch_mode(ConvArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    if (a.xhalf) return otal_conv::launch_half(MODE, a, ws, ws_bytes, st);      // bf16-stored tensors on both sides: part 1
    if constexpr (MODE == MODE_WGRAD) {
        if (conv1a_wgrad_eligible(a.g, a.prec, a.x, a.dy) && (!a.half || a.g.y_bs % 8 + a.g.y_cs % 8 == 0)) {
            const int e = launch_conv1a_wgrad(a, ws, ws_bytes, st);
            if (e != OTAL_E_UNSUPPORTED || a.half) return e;
        }
        if (a.half) return OTAL_E_UNSUPPORTED;              // bf16-stored dy: the kernels above only
        if (proj_wgrad_eligible(a.g, a.prec, a.x, a.dy)) {  // the pyramid projections: short K, no split (proj_gemm.inc)
            const int e = launch_proj_wgrad(a, st);
            if (e != OTAL_E_UNSUPPORTED) return e;
        }
        if (wgrad_direct_eligible(a.g, a.prec, a.x, a.dy)) {
            const int e = launch_wgrad_direct(a, ws, ws_bytes, st);
            if (e != OTAL_E_UNSUPPORTED) return e;          // slabs do not fit: the vector kernel below
        }
        if (wgrad1x1_wide_eligible(a.g, a.prec, a.x, a.dy)) {
            const int e = launch_wgrad1x1_wide(a, ws, ws_bytes, st);
            if (e != OTAL_E_UNSUPPORTED) return e;
        }
        if (wgrad_pair_mode(a.g, a.prec)) return launch_wgrad_vector(a, 8, ws, ws_bytes, st);
        if (const int cw = wgrad_vector_width(a.g, a.prec)) return launch_wgrad_vector(a, cw, ws, ws_bytes, st);
        if (wgrad1d_eligible(a.g, a.prec, a.x, a.dy)) {
            const int e = launch_wgrad1d(a, ws, ws_bytes, st);
            if (e != OTAL_E_UNSUPPORTED) return e;          // workspace too small for the slabs: the generic kernel below
        }
    }
    if constexpr (MODE == MODE_FWD) {
        if (proj_fwd_eligible(a.g, MODE, a.prec, a.x, a.w)) {
            const int e = launch_proj_fwd(a, ws, ws_bytes, st);
            if (e != OTAL_E_UNSUPPORTED) return e;
        }
        if (a.half) {       // bf16-stored y: Conv3d_1a's direct kernel and the direct 3x3x3 kernel
            if (!OTAL_OPT("OTAL_CONV_NO1A", 0) && conv1a_direct_eligible(a.g, MODE, a.prec, a.x) && conv1a_half_out_ok(a.g, a.out))
                return launch_conv1a_direct(a, ws, ws_bytes, st);
            if (direct_eligible(a.g, MODE, a.prec, a.M) && conv1a_half_out_ok(a.g, a.out)) return launch_direct<MODE>(a, ws, ws_bytes, st);
            return OTAL_E_UNSUPPORTED;
        }
        if (conv1a_direct_eligible(a.g, MODE, a.prec, a.x)) return launch_conv1a_direct(a, ws, ws_bytes, st);
    }
    if (a.half) return OTAL_E_UNSUPPORTED;
    if constexpr (MODE != MODE_WGRAD) {
        if (conv1d_tile_eligible(a.g, MODE, a.prec, MODE == MODE_FWD ? (const void*)a.x : (const void*)a.dy, a)) {
            const int e = launch_conv1d_tile<MODE>(a, ws, ws_bytes, st);
            if (e != OTAL_E_UNSUPPORTED) return e;
        }
        if (direct_eligible(a.g, MODE, a.prec, a.M)) return launch_direct<MODE>(a, ws, ws_bytes, st);
        if (chunk_eligible(a.g, MODE, a.prec)) return launch_chunked<MODE>(a, ws, ws_bytes, st);
    }
    if constexpr (MODE == MODE_DGRAD) {
        if (a.w_natural) {          // the generic kernel wants W^T packed (Cin, Cout, kvol): build it at the front of the workspace
            const int kvol = conv_kvol(a.g);
            const size_t wb = align256((size_t)a.g.Cin * a.g.Cout * kvol * sizeof(float));
            if (!ws || ws_bytes < wb) return OTAL_E_UNSUPPORTED;
            const int64_t total = (int64_t)a.g.Cout * a.g.Cin * kvol;
            const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
            hipLaunchKernelGGL(pack_wt_kernel, dim3(blocks), dim3(256), 0, st, a.w, reinterpret_cast<float*>(ws), a.g.Cout, a.g.Cin, kvol);
            if (int e = otal_launch_status()) return e;
            a.w = reinterpret_cast<const float*>(ws);
            a.w_natural = 0;
            ws = reinterpret_cast<char*>(ws) + wb;
            ws_bytes -= wb;
        }
    }
    a.zero = zero_word_address();
    if (!a.zero) return OTAL_E_UNSUPPORTED;
    a.flags |= (OTAL_OPT("OTAL_CONV_DEBUG", 0) & (DBG_NOLOAD | DBG_NOSTORE | DBG_NOBARRIER));
    if (MODE != MODE_WGRAD) {       // carve the tap table off the front of the workspace and build it
        const size_t tb = tab_bytes(a.K);
        if (!ws || ws_bytes < tb) return OTAL_E_UNSUPPORTED;
        int2* tab = reinterpret_cast<int2*>(ws);
        const int Kpad = a.K + (int)TAB_PAD;
        hipLaunchKernelGGL((build_tap_table_kernel<MODE>), dim3((Kpad + 255) / 256), dim3(256), 0, st, tab, a.g, a.K, Kpad);
        if (int e = otal_launch_status()) return e;
        a.tab = tab;
        ws = reinterpret_cast<char*>(ws) + tb;
        ws_bytes -= tb;
    }
    const int BMsel = choose_bm(a.M);
    const int BN = 128;
    const int tm = (a.M + BMsel - 1) / BMsel, tn = (a.N + BN - 1) / BN;
    a.fd = make_conv_fastdiv(a.g);
    if (MODE == MODE_WGRAD) {   // dy rows vectorise along positions when groups of 4 stay inside a sample and aligned
        const int64_t P = conv_out_positions(a.g);
        a.a_vec4 = (P % 4 == 0) && (a.g.y_bs % 4 == 0) && (a.g.y_cs % 4 == 0) && (((uintptr_t)a.dy & 15) == 0) && BMsel >= 64;
    } else {
        a.a_vec4 = (a.K % 4 == 0) && (((uintptr_t)a.w & 15) == 0) && BMsel >= 64;
    }
    int splits = choose_splits(tm * tn, a.K, a.prec);
    if (MODE == MODE_WGRAD && a.prec) {
        // the generic kernel gets the weight gradients of the 1-D pyramid / head layers (126 positions per sample: no vector
        // path): K = B * 126 positions, a few dozen tiles -- latency-bound K steps, so split down to OTAL_GWGRAD_MINSTEPS
        const int ms = OTAL_OPT("OTAL_GWGRAD_MINSTEPS", 4);      // measured: 8 -> 474.6, 4 -> 478.3, 2 -> 478.0, 1 -> 476.3 clips/s
        const int tg = OTAL_OPT("OTAL_GWGRAD_TARGET", 512);
        const int tiles = tm * tn;
        int want = (tg + tiles - 1) / tiles, maxs = a.K / (ms * 32);
        if (maxs < 1) maxs = 1;
        splits = tiles >= tg * 3 / 4 ? 1 : (want < maxs ? want : maxs);
        if (splits > 384) splits = 384;
    }
    if (splits > 1) {
        const size_t need = (size_t)splits * a.M * a.N * sizeof(float);
        if (!ws || ws_bytes < need) {       // shrink to what the workspace allows
            splits = ws ? (int)(ws_bytes / ((size_t)a.M * a.N * sizeof(float))) : 1;
            if (splits < 2) splits = 1;
        }
    }
    int kps = ((a.K + splits - 1) / splits + 31) / 32 * 32;      // multiple of both BK values
    splits = (a.K + kps - 1) / kps;
    a.splits = splits;
    a.k_per_split = kps;
    a.slab = splits > 1 ? (float*)ws : nullptr;
    set_epilogue_extents<MODE>(a);
    const dim3 grid(tn, tm, splits);
#define OTAL_LAUNCH(BM_, WM_, WN_, AV_)                                                                       \
    do {                                                                                                       \
        if (a.prec) hipLaunchKernelGGL((conv_gemm_kernel<BM_, 128, WM_, WN_, MODE, AV_, 1>), grid, dim3(NT), 0, st, a); \
        else hipLaunchKernelGGL((conv_gemm_kernel<BM_, 128, WM_, WN_, MODE, AV_, 0>), grid, dim3(NT), 0, st, a);        \
    } while (0)
    const bool av = a.a_vec4 != 0;
    if (false) {
    } else {
        if (BMsel == 128) { if (av) OTAL_LAUNCH(128, 2, 2, true); else OTAL_LAUNCH(128, 2, 2, false); }
        else if (BMsel == 96) { if (av) OTAL_LAUNCH(96, 3, 1, true); else OTAL_LAUNCH(96, 3, 1, false); }
        else if (BMsel == 64) { if (av) OTAL_LAUNCH(64, 2, 1, true); else OTAL_LAUNCH(64, 2, 1, false); }
        else OTAL_LAUNCH(32, 1, 1, false);
    }
#undef OTAL_LAUNCH
    if (int e = otal_launch_status()) return e;
    if (splits > 1) {
        return launch_splitk_reduce<MODE>(a, st);
    }
    return 0;
}
#endif      // OTAL_CONV_PART == 0

}  // namespace

#if OTAL_CONV_PART == 0
extern "C" size_t otal_conv_workspace_bytes(const int* geom, int mode) {
    ConvGeom g;
    if (!geom || fill_geom(g, geom)) return 0;
    const int kvol = conv_kvol(g);
    int64_t M, N, K;
    if (mode == MODE_FWD) { M = g.Cout; N = (int64_t)g.B * conv_out_positions(g); K = (int64_t)g.Cin * kvol; }
    else if (mode == MODE_DGRAD) { M = g.Cin; N = (int64_t)g.B * conv_in_positions(g); K = (int64_t)g.Cout * kvol; }
    else {
        M = g.Cout; N = (int64_t)g.Cin * kvol; K = (int64_t)g.B * conv_out_positions(g);
        if (g.sw == 2 && g.kw <= 7) N = (int64_t)g.Cin * g.kt * g.kh * 8;      // pair mode pads each kw row to 8 columns
    }
    const int BMsel = choose_bm((int)M);
    const int BN = 128;
    const int tiles = (int)(((M + BMsel - 1) / BMsel) * ((N + BN - 1) / BN));
    int s = choose_splits(tiles, (int)K, 0);            // the fp32 rule splits finer: size for it
    if (mode == MODE_WGRAD) { const int sw = choose_splits(tiles, (int)K, 1, true); if (sw > s) s = sw; }
    // precision is not an argument here: size for whichever path needs more (generic tap table vs chunk table + packed weights)
    size_t front = mode == MODE_WGRAD ? ptab_bytes(g, 2) : tab_bytes((int)K);
    if (mode != MODE_WGRAD) {
        const int Kc = mode == MODE_FWD && g.Cin * g.kt * g.kh * 8 > K ? g.Cin * g.kt * g.kh * 8 : (int)K;   // kw-vector mode pads kw to 8
        size_t cf = chunk_tab_bytes(Kc) + chunk_wp_bytes((int)M, BMsel, Kc);
        if (M % 192 == 0) { const size_t ct = chunk_tab_bytes(Kc) + chunk_wp_bytes((int)M, 192, Kc); if (ct > cf) cf = ct; }
        if (direct_eligible(g, mode, 1, (int)M)) { const size_t dd = direct_wp_bytes(g, (int)M, mode == MODE_FWD ? g.Cin : g.Cout); if (dd > cf) cf = dd; }
        if (cf > front) front = cf;
    }
    if (mode == MODE_DGRAD) front += align256((size_t)g.Cin * g.Cout * kvol * sizeof(float));    // natural-layout weights on the generic path
    return front + (s > 1 ? (size_t)(s + 1) * M * N * sizeof(float) : 0);
}

extern "C" int otal_conv_fwd(const int* geom, const int64_t* strides, const float* x, const float* w,
                             const float* scale, const float* shift, float* y, int relu, int precision,
                             const void* prologue, void* ws, size_t ws_bytes, void* stream) {
    if (!geom || !strides || !x || !w || !y) return OTAL_E_NULL;
    ConvArgs a = {};
    if (int e = fill_geom(a.g, geom)) return e;
    a.g.x_bs = strides[0]; a.g.x_cs = strides[1]; a.g.y_bs = strides[2]; a.g.y_cs = strides[3];
    a.x = x; a.w = w; a.out = y; a.scale = scale; a.shift = shift;
    a.M = a.g.Cout; a.N = a.g.B * conv_out_positions(a.g); a.K = a.g.Cin * conv_kvol(a.g);
    a.flags = relu ? EPI_RELU : 0;
    a.prec = (precision & 1) ? 1 : 0;
    a.half = (precision & 4) ? 1 : 0;
    a.xhalf = (precision & 8) ? 1 : 0;
    if ((a.half || a.xhalf) && !a.prec) return OTAL_E_UNSUPPORTED;
    if (a.xhalf && !a.half) return OTAL_E_UNSUPPORTED;      // a bf16 x is only served together with a bf16 y
    a.pre = prologue;
    return launch_mode<MODE_FWD>(a, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int otal_conv_dgrad(const int* geom, const int64_t* strides, const float* dy, const float* wt_packed,
                               float* dx, int accumulate, const float* out_mask, const float* out_scale,
                               int precision, const void* prologue, void* ws, size_t ws_bytes, void* stream) {
    if (!geom || !strides || !dy || !wt_packed || !dx) return OTAL_E_NULL;
    if ((out_mask == nullptr) != (out_scale == nullptr)) return OTAL_E_NULL;
    ConvArgs a = {};
    if (int e = fill_geom(a.g, geom)) return e;
    a.g.x_bs = strides[0]; a.g.x_cs = strides[1]; a.g.y_bs = strides[2]; a.g.y_cs = strides[3];
    a.dy = dy; a.w = wt_packed; a.out = dx;
    if (a.g.st > 2 || a.g.sh > 2 || a.g.sw > 2) return OTAL_E_UNSUPPORTED;   // gather is specialised for strides 1, 2
    a.emask = out_mask; a.escale = out_scale;
    a.M = a.g.Cin; a.N = a.g.B * conv_in_positions(a.g); a.K = a.g.Cout * conv_kvol(a.g);
    a.flags = accumulate ? EPI_ACCUM : 0;
    a.prec = (precision & 1) ? 1 : 0;
    a.w_natural = (precision & 2) ? 1 : 0;
    a.half = (precision & 4) ? 1 : 0;
    a.xhalf = (precision & 8) ? 1 : 0;
    a.mhalf = (precision & 16) ? 1 : 0;
    if ((a.half || a.xhalf || a.mhalf) && !a.prec) return OTAL_E_UNSUPPORTED;
    if (a.half != a.xhalf || (a.mhalf && !a.xhalf) || (out_mask && a.xhalf && !a.mhalf)) return OTAL_E_UNSUPPORTED;   // all three bf16, or none
    a.pre = prologue;
    return launch_mode<MODE_DGRAD>(a, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int otal_conv_wgrad(const int* geom, const int64_t* strides, const float* x, const float* dy,
                               float* dw, int accumulate, int precision,
                               const void* prologue, void* ws, size_t ws_bytes, void* stream) {
    if (!geom || !strides || !x || !dy || !dw) return OTAL_E_NULL;
    ConvArgs a = {};
    if (int e = fill_geom(a.g, geom)) return e;
    a.g.x_bs = strides[0]; a.g.x_cs = strides[1]; a.g.y_bs = strides[2]; a.g.y_cs = strides[3];
    a.x = x; a.dy = dy; a.out = dw;
    a.M = a.g.Cout; a.N = a.g.Cin * conv_kvol(a.g); a.K = a.g.B * conv_out_positions(a.g);
    a.flags = accumulate ? EPI_ACCUM : 0;
    a.prec = (precision & 1) ? 1 : 0;
    a.half = (precision & 4) ? 1 : 0;
    a.xhalf = (precision & 8) ? 1 : 0;
    if ((a.half || a.xhalf) && !a.prec) return OTAL_E_UNSUPPORTED;
    if (a.xhalf && !a.half) return OTAL_E_UNSUPPORTED;      // a bf16 x is only served together with a bf16 dy
    a.pre = prologue;
    return launch_mode<MODE_WGRAD>(a, ws, ws_bytes, (hipStream_t)stream);
}

// ---- pair launches: two problems of one geometry in one grid (the 1-D temporal layers only: anything else is
// OTAL_E_UNSUPPORTED and the caller launches the two problems one after the other)
namespace {
int pair_geom(ConvArgs& a, const int* geom, const int64_t* strides) {
    if (int e = fill_geom(a.g, geom)) return e;
    a.g.x_bs = strides[0]; a.g.x_cs = strides[1]; a.g.y_bs = strides[2]; a.g.y_cs = strides[3];
    a.pair = 1;
    return OTAL_OPT("OTAL_CONV_NOPAIR", 0) ? OTAL_E_UNSUPPORTED : 0;
}
}  // namespace

extern "C" int otal_conv_fwd_pair(const int* geom, const int64_t* strides, const float* const* x, const float* const* w,
                                  const float* const* scale, const float* const* shift, float* const* y, int relu, int precision,
                                  const void* const* prologue, void* ws, size_t ws_bytes, void* stream) {
    if (!geom || !strides || !x || !w || !y || !x[0] || !x[1] || !w[0] || !w[1] || !y[0] || !y[1]) return OTAL_E_NULL;
    ConvArgs a = {};
    if (int e = pair_geom(a, geom, strides)) return e;
    a.x = x[0]; a.x2 = x[1]; a.w = w[0]; a.w2 = w[1]; a.out = y[0]; a.out2 = y[1];
    a.scale = scale ? scale[0] : nullptr; a.scale2 = scale ? scale[1] : nullptr;
    a.shift = shift ? shift[0] : nullptr; a.shift2 = shift ? shift[1] : nullptr;
    if ((a.scale == nullptr) != (a.scale2 == nullptr) || (a.shift == nullptr) != (a.shift2 == nullptr)) return OTAL_E_NULL;
    a.M = a.g.Cout; a.N = a.g.B * conv_out_positions(a.g); a.K = a.g.Cin * conv_kvol(a.g);
    a.flags = relu ? EPI_RELU : 0;
    a.prec = (precision & 1) ? 1 : 0;
    a.pre = prologue ? prologue[0] : nullptr; a.pre2 = prologue ? prologue[1] : nullptr;
    if (!conv1d_tile_eligible(a.g, MODE_FWD, a.prec, a.x, a) || ((uintptr_t)a.x2 & 3)) return OTAL_E_UNSUPPORTED;
    return launch_conv1d_tile<MODE_FWD>(a, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int otal_conv_dgrad_pair(const int* geom, const int64_t* strides, const float* const* dy, const float* const* w,
                                    float* const* dx, int precision, const void* const* prologue, void* ws, size_t ws_bytes,
                                    void* stream) {
    if (!geom || !strides || !dy || !w || !dx || !dy[0] || !dy[1] || !w[0] || !w[1] || !dx[0] || !dx[1]) return OTAL_E_NULL;
    ConvArgs a = {};
    if (int e = pair_geom(a, geom, strides)) return e;
    a.dy = dy[0]; a.dy2 = dy[1]; a.w = w[0]; a.w2 = w[1]; a.out = dx[0]; a.out2 = dx[1];
    a.M = a.g.Cin; a.N = a.g.B * conv_in_positions(a.g); a.K = a.g.Cout * conv_kvol(a.g);
    a.prec = (precision & 1) ? 1 : 0;
    a.w_natural = (precision & 2) ? 1 : 0;
    a.pre = prologue ? prologue[0] : nullptr; a.pre2 = prologue ? prologue[1] : nullptr;
    if (!a.w_natural) return OTAL_E_UNSUPPORTED;            // forward-layout weights only (the prologue re-orders them)
    if (!conv1d_tile_eligible(a.g, MODE_DGRAD, a.prec, a.dy, a) || ((uintptr_t)a.dy2 & 3)) return OTAL_E_UNSUPPORTED;
    return launch_conv1d_tile<MODE_DGRAD>(a, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int otal_conv_wgrad_pair(const int* geom, const int64_t* strides, const float* const* x, const float* const* dy,
                                    float* const* dw, int precision, void* ws, size_t ws_bytes, void* stream) {
    if (!geom || !strides || !x || !dy || !dw || !x[0] || !x[1] || !dy[0] || !dy[1] || !dw[0] || !dw[1]) return OTAL_E_NULL;
    ConvArgs a = {};
    if (int e = pair_geom(a, geom, strides)) return e;
    a.x = x[0]; a.x2 = x[1]; a.dy = dy[0]; a.dy2 = dy[1]; a.out = dw[0]; a.out2 = dw[1];
    a.M = a.g.Cout; a.N = a.g.Cin * conv_kvol(a.g); a.K = a.g.B * conv_out_positions(a.g);
    a.prec = (precision & 1) ? 1 : 0;
    if (!wgrad1d_eligible(a.g, a.prec, a.x, a.dy) || !wgrad1d_eligible(a.g, a.prec, a.x2, a.dy2)) return OTAL_E_UNSUPPORTED;
    return launch_wgrad1d(a, ws, ws_bytes, (hipStream_t)stream);
}

// 1 when this geometry has a kernel for the bf16-STORED large operand (precision bit 2): fwd -> y, wgrad / dgrad -> dy.
// Pointer alignment (16 bytes) and strides that are multiples of 8 elements are the caller's side of the contract.
extern "C" int otal_conv_half_storage(const int* geom, const int64_t* strides, int mode, int precision) {
    ConvArgs a = {};
    if (!geom || !strides || fill_geom(a.g, geom) || !(precision & 1)) return 0;
    a.g.x_bs = strides[0]; a.g.x_cs = strides[1]; a.g.y_bs = strides[2]; a.g.y_cs = strides[3];
    if (a.g.y_bs % 8 || a.g.y_cs % 8) return 0;
    if (precision & 8) return half_kernel_kind(a.g, mode, nullptr, nullptr) ? 1 : 0;      // bf16 on both sides of the layer
    if (mode == MODE_FWD) {
        if (conv1a_direct_eligible(a.g, MODE_FWD, 1, nullptr) && !OTAL_OPT("OTAL_CONV_NO1A", 0)) return 1;
        return direct_eligible(a.g, MODE_FWD, 1, a.g.Cout) && !OTAL_OPT("OTAL_CONV_NODIRECT_HALF", 0) ? 1 : 0;
    }
    if (mode == MODE_WGRAD) return conv1a_wgrad_eligible(a.g, 1, nullptr, nullptr) ? 1 : 0;
    return 0;
}

extern "C" int otal_conv_pack_wt(const float* w, float* wt, int Cout, int Cin, int kvol, void* stream) {
    if (!w || !wt) return OTAL_E_NULL;
    if (Cout <= 0 || Cin <= 0 || kvol <= 0) return OTAL_E_SHAPE;
    const int64_t total = (int64_t)Cout * Cin * kvol;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_wt_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wt, Cout, Cin, kvol);
    return otal_launch_status();
}

// ---- persistent prologues (tables + packed bf16 weights in a caller-owned region) ------------------------------------
namespace {
// which prologue a launch with this geometry would run: 0 none (generic kernel), 1 chunk path (fwd / dgrad), 2 position table
int prologue_kind(const ConvGeom& g, int mode, int precision) {
    const int prec = precision & 1;
    if (mode == MODE_WGRAD) {
        if (wgrad_pair_mode(g, prec)) return 2;
        return wgrad_vector_width(g, prec) ? 2 : 0;
    }
    const int M = mode == MODE_FWD ? g.Cout : g.Cin;
    if (mode == MODE_FWD && g.kt == 1 && g.kh == g.Hi && g.kw == g.Wi && g.Hi * g.Wi == 36 && g.Ho == 1 && g.Wo == 1 &&
        g.Cin % 4 == 0 && !OTAL_OPT("OTAL_CONV_NOPROJ", 0)) return 0;       // the projection GEMM reads the fp32 weights in place
    if (conv1a_direct_eligible(g, mode, prec, nullptr)) return 0;
    if (direct_eligible(g, mode, prec, M)) return OTAL_OPT("OTAL_CONV_NODIRECTPRE", 0) ? 0 : 3;      // the direct kernel's weight pack
    return chunk_eligible(g, mode, prec) ? 1 : 0;
}
int fill_args_for_prologue(ConvArgs& a, const int* geom, const int64_t* strides, int mode, const float* w, int precision) {
    if (int e = fill_geom(a.g, geom)) return e;
    a.g.x_bs = strides[0]; a.g.x_cs = strides[1]; a.g.y_bs = strides[2]; a.g.y_cs = strides[3];
    a.w = w;
    a.prec = precision & 1;
    a.w_natural = (mode == MODE_DGRAD && (precision & 2)) ? 1 : 0;
    const int kvol = conv_kvol(a.g);
    if (mode == MODE_FWD) { a.M = a.g.Cout; a.K = a.g.Cin * kvol; if (a.g.Cin % 8) a.K = a.g.Cin * a.g.kt * a.g.kh * 8; }
    else if (mode == MODE_DGRAD) { a.M = a.g.Cin; a.K = a.g.Cout * kvol; }
    else { a.M = a.g.Cout; a.K = a.g.B * conv_out_positions(a.g); }
    return 0;
}
}  // namespace

extern "C" size_t otal_conv_prologue_bytes(const int* geom, const int64_t* strides, int mode, int precision) {
    ConvArgs a = {};
    if (!geom || !strides || fill_args_for_prologue(a, geom, strides, mode, nullptr, precision)) return 0;
    const int kind = prologue_kind(a.g, mode, precision);
    if (kind == 1) {
        const bool kwv = mode == MODE_FWD && (a.g.Cin % 8) != 0;
        return chunk_tab_bytes(a.K) + chunk_wp_bytes(a.M, choose_bm(a.M, kwv ? 0 : 1), a.K);
    }
    if (kind == 2) return ptab_bytes(a.g, a.g.sw == 2 ? 8 : wgrad_vector_width(a.g, a.prec));
    if (kind == 3) return direct_wp_bytes(a.g, a.M, mode == MODE_FWD ? a.g.Cin : a.g.Cout);
    return 0;
}

extern "C" size_t otal_conv_prologue_desc_bytes(void) { return sizeof(PrepDesc); }

// Fills `region` now (one launch).  For fwd / dgrad also writes the descriptor otal_conv_prologue_batch consumes into
// host_desc (nullable) and returns the number of workgroups that descriptor needs (> 0); wgrad returns 0; errors < 0.
extern "C" int otal_conv_prologue(const int* geom, const int64_t* strides, int mode, const float* w, int precision,
                                  void* region, size_t region_bytes, void* host_desc, void* stream) {
    if (!geom || !strides || !region) return OTAL_E_NULL;
    ConvArgs a = {};
    if (int e = fill_args_for_prologue(a, geom, strides, mode, w, precision)) return e;
    const int kind = prologue_kind(a.g, mode, precision);
    if (kind == 0 || region_bytes < otal_conv_prologue_bytes(geom, strides, mode, precision)) return OTAL_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (kind == 2) {
        const int cw = a.g.sw == 2 ? 8 : wgrad_vector_width(a.g, a.prec);
        a.fd = make_conv_fastdiv(a.g);
        const int ngroups = a.K / cw, npad = ngroups + PTAB_PAD;
        hipLaunchKernelGGL(build_pos_table_kernel, dim3((npad + 255) / 256), dim3(256), 0, st, reinterpret_cast<int2*>(region),
                           a.g, a.fd, cw, ngroups, npad);
        if (int e = otal_launch_status()) return e > 0 ? -100 - e : e;
        return 0;
    }
    if (!w) return OTAL_E_NULL;
    PrepDesc d;
    if (kind == 3) {
        d = PrepDesc{};
        const int BM = direct_bm(a.g, a.M);
        d.wp = reinterpret_cast<unsigned*>(region); d.wsrc = w; d.g = a.g; d.M = a.M; d.Mpad = (a.M + BM - 1) / BM * BM;
        d.C = mode == MODE_FWD ? a.g.Cin : a.g.Cout; d.natural = a.w_natural; d.mode = mode; d.direct = 1;
        d.fh = make_fastdiv((uint32_t)(d.C * 27 / 2));
        launch_prep(d, st);
        if (int e = otal_launch_status()) return e > 0 ? -100 - e : e;
        if (host_desc) memcpy(host_desc, &d, sizeof(d));
        return (int)prep_blocks(d);
    }
    int2* ctab = reinterpret_cast<int2*>(region);
    unsigned short* wp = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(region) + chunk_tab_bytes(a.K));
    if (mode == MODE_FWD) fill_prep_desc<MODE_FWD>(d, a, ctab, wp); else fill_prep_desc<MODE_DGRAD>(d, a, ctab, wp);
    launch_prep(d, st);
    if (int e = otal_launch_status()) return e > 0 ? -100 - e : e;
    if (host_desc) memcpy(host_desc, &d, sizeof(d));
    return (int)prep_blocks(d);
}

// Re-runs n fwd / dgrad prologues (descriptors in DEVICE memory, as written by otal_conv_prologue) in ONE launch;
// device_starts[n+1] = prefix sums of the workgroup counts otal_conv_prologue returned, total_blocks = device_starts[n].
extern "C" int otal_conv_prologue_batch(int n, const void* device_descs, const int* device_starts, int total_blocks,
                                        void* stream) {
    if (n <= 0) return 0;
    if (!device_descs || !device_starts) return OTAL_E_NULL;
    if (total_blocks <= 0) return OTAL_E_SHAPE;
    hipLaunchKernelGGL(prep_chunks_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const PrepDesc*>(device_descs), device_starts, n);
    return otal_launch_status();
}


extern "C" int otal_conv_defer_reduces(int on) {
    std::lock_guard<std::recursive_mutex> lock(g_defer.mu);
    if (!on && g_defer.n) return OTAL_E_SHAPE;          // flush first
    g_defer.on = on != 0;
    g_defer.last_end = 0;
    return 0;
}
extern "C" size_t otal_conv_deferred_end(void) { return (size_t)g_defer.last_end; }
extern "C" int otal_conv_deferred_count(void) { return g_defer.n; }
extern "C" int otal_conv_flush_reduces(void* stream) { return flush_deferred((hipStream_t)stream); }
#endif      // OTAL_CONV_PART == 0
