// opental_amd/csrc/misc.hip -- small fused kernels of the detection path: proposal-window index
// math (bit-exact), flat Adam, the masked / scaled strided copy that hands the pyramid's gradients to the backbone.
#include "common.h"

namespace {

struct Levels { int nlev; int lev[OTAL_MAX_LEVELS + 1]; };

// Restates the no_grad block of CoarsePyramid.forward (AFSD/thumos14/BDNet.py:355-384) in ONE launch
// for all levels (the reference issues ~25 tiny elementwise kernels per level).  The operation
// order of the reference is kept literally and the file is compiled with -ffp-contract=off, so
// the rounded window indices are bit-identical: torch.round == rintf (half to even), `/` is the
// correctly rounded fp32 division, the prior centre (c + 0.5) / t is formed in double and
// rounded once to float exactly as torch.Tensor([...]) does.
__global__ __launch_bounds__(256) void proposal_windows_kernel(const float* __restrict__ loc, float* __restrict__ seg,
                                                               float* __restrict__ fseg, int B, int Ntot,
                                                               float frame_num, Levels L) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * Ntot) return;
    const int k = idx % Ntot;
    int lo = L.lev[0], hi = L.lev[1];
#pragma unroll
    for (int j = 1; j < OTAL_MAX_LEVELS; ++j)
        if (j < L.nlev && k >= L.lev[j]) { lo = L.lev[j]; hi = L.lev[j + 1]; }
    const int t = hi - lo, c = k - lo;
    const float tf = (float)t;
    const float prior = (float)(((double)c + 0.5) / (double)t);
    const float l0 = loc[(size_t)idx * 2], l1 = loc[(size_t)idx * 2 + 1];
    // level space
    const float s0 = l0 / frame_num * tf, s1 = l1 / frame_num * tf;
    const float centre = rintf(prior * tf - 0.5f);
    float plen = s0 + s1;
    float inner = fmaxf(plen / 4.0f, 1.0f), outer = fmaxf(plen / 10.0f, 1.0f);
    const float left = centre - s0, right = centre + s1;
    float4 w;
    w.x = rintf(left - outer); w.y = rintf(left + inner); w.z = rintf(right - inner); w.w = rintf(right + outer);
    *reinterpret_cast<float4*>(seg + (size_t)idx * 4) = w;
    // frame space
    const float d0 = prior * frame_num - l0, d1 = prior * frame_num + l1;
    plen = d1 - d0 + 1.0f;
    inner = fmaxf(plen / 4.0f, 1.0f); outer = fmaxf(plen / 10.0f, 1.0f);
    w.x = rintf(d0 - outer); w.y = rintf(d0 + inner); w.z = rintf(d1 - inner); w.w = rintf(d1 + outer);
    *reinterpret_cast<float4*>(fseg + (size_t)idx * 4) = w;
}

// torch.optim.Adam (L2 weight decay folded into the gradient, reference AFSD/thumos14/train.py:321-323)
// over ONE flat parameter arena: p, g, m, v are parallel fp32 arrays of n elements.
// one element of the update (the arithmetic of torch.optim.Adam's single-tensor path, see the header comment above)
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float wd,
                                         float bc1, float bc2_sqrt, float grad_scale) {
    float gi = g * grad_scale;
    gi = gi + wd * p;
    const float mi = m + (gi - m) * (1.0f - b1);             // lerp form used by torch
    const float vi = v * b2 + (1.0f - b2) * gi * gi;
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p = p - (lr / bc1) * (mi / denom);
}
// The flat update streams 7 floats per parameter (4 read, 3 written).  One dword per lane and trip kept the kernel at
// 3.3 TB/s (26 dependent trips of four 4-byte loads per thread); 16-byte accesses: the same values from a quarter of the
// memory instructions and trips.  `n4` vectors when the four arrays are 16-byte aligned, the tail element-wise.
__device__ __forceinline__ void adam_sweep(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                           float bc1, float bc2_sqrt, float grad_scale) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int64_t n4 = vec ? n >> 2 : 0;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = tid; i < n4; i += nth) {
        float4 pa = p4[i], ma = m4[i], va = v4[i];
        const float4 ga = g4[i];
        adam_one(pa.x, ga.x, ma.x, va.x, lr, b1, b2, eps, wd, bc1, bc2_sqrt, grad_scale);
        adam_one(pa.y, ga.y, ma.y, va.y, lr, b1, b2, eps, wd, bc1, bc2_sqrt, grad_scale);
        adam_one(pa.z, ga.z, ma.z, va.z, lr, b1, b2, eps, wd, bc1, bc2_sqrt, grad_scale);
        adam_one(pa.w, ga.w, ma.w, va.w, lr, b1, b2, eps, wd, bc1, bc2_sqrt, grad_scale);
        p4[i] = pa; m4[i] = ma; v4[i] = va;
    }
    for (int64_t i = 4 * n4 + tid; i < n; i += nth) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_one(pi, g[i], mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt, grad_scale);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        float lr, float b1, float b2, float eps, float wd,
                                                        float bc1, float bc2_sqrt, float grad_scale) {
    adam_sweep(p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, grad_scale);
}

// same update with the bias corrections read from device memory: the launch arguments do not change from step
// to step, so the whole training step can be replayed from a captured HIP graph
__global__ __launch_bounds__(256) void adam_flat_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                            float lr, float b1, float b2, float eps, float wd,
                                                            const float* __restrict__ bias_corr, float grad_scale) {
    adam_sweep(p, g, m, v, n, lr, b1, b2, eps, wd, bias_corr[0], bias_corr[1], grad_scale);
}


// dst[b][c][t][s] (+)= (z[b][c][t][s] > 0 ? scale[c] : 0) * src[b][c][t][s], every tensor with its own (batch, channel, frame)
// strides and unit stride along s (the H*W plane): the ReLU + frozen-BN backward that I3DFeaturesFunction.backward applies
// to the gradients arriving at Mixed_4f / Mixed_5c (aten threshold_backward + mul + copy_ = 3 launches over the map) fused
// with the un-permute of the projection's data gradient, which the swapped-role GEMM leaves as [(b, t)][c][s]
// (ops.conv_dgrad_collapse) -- one pass, 12 bytes per element.  V = 4: S % 4 == 0 and 16-byte aligned rows.
struct Strides3 { int64_t b, c, t; };
template <int V>
__global__ __launch_bounds__(256) void masked_scale_copy_kernel(const float* __restrict__ src, Strides3 ss, const float* __restrict__ z,
                                                                Strides3 zs, const float* __restrict__ scale, float* __restrict__ dst,
                                                                Strides3 ds, int C, int T, int S, int64_t total, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int SV = S / V;
    const int sv = (int)(i % SV);
    int64_t r = i / SV;
    const int t = (int)(r % T); r /= T;
    const int c = (int)(r % C);
    const int64_t b = r / C;
    const int64_t so = b * ss.b + c * ss.c + t * ss.t + sv * V, zo = b * zs.b + c * zs.c + t * zs.t + sv * V,
                  dz = b * ds.b + c * ds.c + t * ds.t + sv * V;
    const float k = scale ? scale[c] : 1.f;
    if (V == 4) {
        const float4 g = *reinterpret_cast<const float4*>(src + so);
        const float4 m = *reinterpret_cast<const float4*>(z + zo);
        float4 o;
        o.x = m.x > 0.f ? g.x * k : 0.f; o.y = m.y > 0.f ? g.y * k : 0.f;
        o.z = m.z > 0.f ? g.z * k : 0.f; o.w = m.w > 0.f ? g.w * k : 0.f;
        if (accumulate) {
            const float4 d = *reinterpret_cast<const float4*>(dst + dz);
            o.x = d.x + o.x; o.y = d.y + o.y; o.z = d.z + o.z; o.w = d.w + o.w;
        }
        *reinterpret_cast<float4*>(dst + dz) = o;
    } else {
        const float o = z[zo] > 0.f ? src[so] * k : 0.f;
        dst[dz] = accumulate ? dst[dz] + o : o;
    }
}

// fp32 <-> bf16 STORAGE conversion of a (B, C, P) map with dense positions and arbitrary batch / channel strides (channel
// slices of a concat buffer): the boundary between the backbone's bf16-stored tensors and the fp32 tensors around them.
// TO_H: fp32 -> bf16, round to nearest even; else bf16 -> fp32 (exact).  Eight positions per thread.
template <bool TO_H>
__global__ __launch_bounds__(256) void convert_storage_kernel(const void* __restrict__ src, int64_t sbs, int64_t scs, void* __restrict__ dst,
                                                              int64_t dbs, int64_t dcs, int C, int P8, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % P8) * 8;
    int64_t r = i / P8;
    const int c = (int)(r % C);
    const int64_t b = r / C;
    const int64_t so = b * sbs + c * scs + p, dofs = b * dbs + c * dcs + p;
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    if constexpr (TO_H) {
        const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(src) + so);
        const float4 d = *reinterpret_cast<const float4*>(static_cast<const float*>(src) + so + 4);
        auto pk = [](float lo, float hi) { const f2 f = {lo, hi}; return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf2)); };
        *reinterpret_cast<uint4*>(static_cast<unsigned short*>(dst) + dofs) = make_uint4(pk(a.x, a.y), pk(a.z, a.w), pk(d.x, d.y), pk(d.z, d.w));
    } else {
        const uint4 w = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(src) + so);
        auto lo = [](unsigned v) { return __uint_as_float(v << 16); };
        auto hi = [](unsigned v) { return __uint_as_float(v & 0xffff0000u); };
        float* o = static_cast<float*>(dst) + dofs;
        *reinterpret_cast<float4*>(o) = make_float4(lo(w.x), hi(w.x), lo(w.y), hi(w.y));
        *reinterpret_cast<float4*>(o + 4) = make_float4(lo(w.z), hi(w.z), lo(w.w), hi(w.w));
    }
}

}  // namespace

extern "C" int otal_convert_storage(const void* src, int64_t src_bs, int64_t src_cs, void* dst, int64_t dst_bs, int64_t dst_cs,
                                    int to_bf16, int B, int C, int P, void* stream) {
    if (!src || !dst) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || P <= 0) return OTAL_E_SHAPE;
    if (P % 8 || src_bs % 8 || src_cs % 8 || dst_bs % 8 || dst_cs % 8 ||
        ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15)) return OTAL_E_UNSUPPORTED;
    const int64_t total = (int64_t)B * C * (P / 8);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (to_bf16) hipLaunchKernelGGL(convert_storage_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, src_bs, src_cs, dst, dst_bs, dst_cs, C, P / 8, total);
    else hipLaunchKernelGGL(convert_storage_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, src_bs, src_cs, dst, dst_bs, dst_cs, C, P / 8, total);
    return otal_launch_status();
}

extern "C" int otal_proposal_windows(const float* loc, float* seg, float* frame_seg, int B, int nlev,
                                     const int* lev, float frame_num, void* stream) {
    if (!loc || !seg || !frame_seg || !lev) return OTAL_E_NULL;
    if (B <= 0 || nlev < 1 || nlev > OTAL_MAX_LEVELS || lev[0] != 0) return OTAL_E_LEVELS;
    Levels L;
    L.nlev = nlev;
    for (int i = 0; i <= OTAL_MAX_LEVELS; ++i) L.lev[i] = lev[i <= nlev ? i : nlev];
    for (int i = 0; i < nlev; ++i) if (L.lev[i + 1] <= L.lev[i]) return OTAL_E_LEVELS;
    const int Ntot = L.lev[nlev];
    const int total = B * Ntot;
    hipLaunchKernelGGL(proposal_windows_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       loc, seg, frame_seg, B, Ntot, frame_num, L);
    return otal_launch_status();
}

extern "C" int otal_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v) return OTAL_E_NULL;
    if (n <= 0 || step < 1) return OTAL_E_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const int64_t blocks64 = (n + 255) / 256;
    const int blocks = (int)(blocks64 < 4096 ? blocks64 : 4096);
    hipLaunchKernelGGL(adam_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    return otal_launch_status();
}

extern "C" int otal_adam_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, const float* bias_corr,
                                  float grad_scale, void* stream) {
    if (!p || !g || !m || !v || !bias_corr) return OTAL_E_NULL;
    if (n <= 0) return OTAL_E_SHAPE;
    const int64_t blocks64 = (n + 255) / 256;
    const int blocks = (int)(blocks64 < 4096 ? blocks64 : 4096);
    hipLaunchKernelGGL(adam_flat_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, bias_corr, grad_scale);
    return otal_launch_status();
}

extern "C" int otal_masked_scale_copy(const float* src, const int64_t* src_strides, const float* z, const int64_t* z_strides,
                                      const float* scale, float* dst, const int64_t* dst_strides, int accumulate, int B, int C,
                                      int T, int S, void* stream) {
    if (!src || !z || !dst || !src_strides || !z_strides || !dst_strides) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || S <= 0) return OTAL_E_SHAPE;
    const Strides3 ss{src_strides[0], src_strides[1], src_strides[2]}, zs{z_strides[0], z_strides[1], z_strides[2]},
                   ds{dst_strides[0], dst_strides[1], dst_strides[2]};
    auto vec_ok = [&](const void* p, const Strides3& q) {
        return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && q.b % 4 == 0 && q.c % 4 == 0 && q.t % 4 == 0;
    };
    const bool v4 = S % 4 == 0 && vec_ok(src, ss) && vec_ok(z, zs) && vec_ok(dst, ds);
    const int64_t total = (int64_t)B * C * T * (v4 ? S / 4 : S);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (v4) hipLaunchKernelGGL(masked_scale_copy_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, src, ss, z, zs, scale, dst, ds, C, T, S, total, accumulate);
    else hipLaunchKernelGGL(masked_scale_copy_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, src, ss, z, zs, scale, dst, ds, C, T, S, total, accumulate);
    return otal_launch_status();
}

// ---- pyramid merge (AFSD/thumos14/BDNet.py:310-326): the two projected maps p0 (B,C,t0) and p1 (B,C,t0/2) -> the first two
// levels of the level-packed buffer, level 0 = p0 + nearest-upsampled p1 (F.interpolate(x, x0.size()[2:], mode='nearest'):
// source index t / 2), level 1 = p1, and the frame-level input F.interpolate(level 0, [frame_num, 1]) (index t / up) --
// one launch for what were upsample_nearest1d + add + the level copies of torch.cat + upsample_nearest2d.
namespace {
__global__ __launch_bounds__(256) void pyramid_merge_fwd_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                                float* __restrict__ packed, float* __restrict__ frame,
                                                                int rows, int t0, int T, int up) {
    const int t1 = t0 >> 1, F = t0 * up;
    const int row = blockIdx.x;                              // (b, c)
    if (row >= rows) return;
    for (int t = threadIdx.x; t < F; t += 256) {
        const int s = t / up;
        const float v = p0[(size_t)row * t0 + s] + p1[(size_t)row * t1 + (s >> 1)];
        frame[(size_t)row * F + t] = v;
        if (t < t0) packed[(size_t)row * T + t] = p0[(size_t)row * t0 + t] + p1[(size_t)row * t1 + (t >> 1)];
        if (t < t1) packed[(size_t)row * T + t0 + t] = p1[(size_t)row * t1 + t];
    }
}
// backward: da, db (B,C,T) the two gradients w.r.t. the packed buffer (only their first two levels are read), dframe
// (B,C,t0*up), dnext (B,C,t0/2) the data gradient of the stride-2 layer that reads level 1 (may be null):
//   d level0[t] = da[t] + db[t] + sum_{j<up} dframe[up t + j];  dp0 = d level0;
//   dp1[s] = da[t0+s] + db[t0+s] + dnext[s] + d level0[2s] + d level0[2s+1]        (fixed summation order)
__global__ __launch_bounds__(256) void pyramid_merge_bwd_kernel(const float* __restrict__ da, const float* __restrict__ db,
                                                                const float* __restrict__ dframe, const float* __restrict__ dnext,
                                                                float* __restrict__ dp0, float* __restrict__ dp1,
                                                                int rows, int t0, int T, int up) {
    __shared__ float l0[1024];
    const int t1 = t0 >> 1, F = t0 * up;
    const int row = blockIdx.x;
    if (row >= rows) return;
    for (int t = threadIdx.x; t < t0; t += 256) {
        float v = da[(size_t)row * T + t] + (db ? db[(size_t)row * T + t] : 0.f);
        float f = 0.f;
        for (int j = 0; j < up; ++j) f += dframe[(size_t)row * F + t * up + j];
        v += f;
        l0[t] = v;
        dp0[(size_t)row * t0 + t] = v;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < t1; s += 256) {
        float v = da[(size_t)row * T + t0 + s] + (db ? db[(size_t)row * T + t0 + s] : 0.f);
        if (dnext) v += dnext[(size_t)row * t1 + s];
        v += l0[2 * s] + l0[2 * s + 1];
        dp1[(size_t)row * t1 + s] = v;
    }
}
}  // namespace

extern "C" int otal_pyramid_merge_fwd(const float* p0, const float* p1, float* packed, float* frame, int B, int C, int t0,
                                      int T, int up, void* stream) {
    if (!p0 || !p1 || !packed || !frame) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || t0 <= 0 || (t0 & 1) || up <= 0 || T < t0 + t0 / 2) return OTAL_E_SHAPE;
    hipLaunchKernelGGL(pyramid_merge_fwd_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, p0, p1, packed, frame, B * C, t0, T, up);
    return otal_launch_status();
}
extern "C" int otal_pyramid_merge_bwd(const float* da, const float* db, const float* dframe, const float* dnext, float* dp0,
                                      float* dp1, int B, int C, int t0, int T, int up, void* stream) {
    if (!da || !dframe || !dp0 || !dp1) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || t0 <= 0 || (t0 & 1) || t0 > 1024 || up <= 0 || T < t0 + t0 / 2) return OTAL_E_SHAPE;
    hipLaunchKernelGGL(pyramid_merge_bwd_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, da, db, dframe, dnext, dp0, dp1,
                       B * C, t0, T, up);
    return otal_launch_status();
}
