// opental_amd/csrc/bmp.hip -- BoundaryMaxPooling forward/backward for gfx950 (MI355X).
//
// Replaces AFSD/prop_pooling/boundary_max_pooling_kernel.cu:17-145 (1 thread per output,
// uncoalesced window scans from global memory, atomicAdd backward).  Written for CDNA4 instead:
//   * a workgroup owns ROWS consecutive (n,c) feature rows -- contiguous in the (B,C,T) layout --
//     and stages them ONCE into LDS with coalesced loads (algorithmic traffic: every input
//     element is read from HBM exactly once, every output written once);
//   * the proposal windows of sample n are decoded (trunc + clamp) once per workgroup into LDS;
//   * lanes run along k (proposals), so neighbouring lanes scan neighbouring LDS addresses and
//     the output store is a coalesced row;
//   * backward is deterministic and atomic-free: arg-max per (row, proposal) in parallel, then one
//     lane per row adds grad_out into an LDS grad_in tile in ascending k, then a coalesced write-out.
//   * a level table lets ONE launch pool all pyramid levels (packed along T / N).
// HBM-bound: bytes = 4*(C*T + C*N) + 16*N per sample forward (kernels/bmp).
#include "common.h"

namespace {

// Feature dtypes (the reference dispatches float / double / half, boundary_max_pooling_kernel.cu:99,:131; bf16 is this
// library's own storage type).  S = the type rows are staged and gradients are summed in: float for f32 / bf16 / f16
// (loads are exact, the selected maximum is an input value, so the stored output is exact too), double for f64.
typedef _Float16 f16_t;
template <typename S> __device__ __forceinline__ S ld_as(const float* p, size_t i) { return (S)p[i]; }
template <typename S> __device__ __forceinline__ S ld_as(const bf16_t* p, size_t i) { return (S)bf16_to_f32(p[i]); }
template <typename S> __device__ __forceinline__ S ld_as(const f16_t* p, size_t i) { return (S)(float)p[i]; }
template <typename S> __device__ __forceinline__ S ld_as(const double* p, size_t i) { return (S)p[i]; }
template <typename S> __device__ __forceinline__ void st_as(float* p, size_t i, S v) { p[i] = (float)v; }
template <typename S> __device__ __forceinline__ void st_as(bf16_t* p, size_t i, S v) { p[i] = f32_to_bf16((float)v); }
template <typename S> __device__ __forceinline__ void st_as(f16_t* p, size_t i, S v) { p[i] = (f16_t)(float)v; }
template <typename S> __device__ __forceinline__ void st_as(double* p, size_t i, S v) { p[i] = (double)v; }
template <typename T> struct StageOf { typedef float type; };
template <> struct StageOf<double> { typedef double type; };

// level lookup with compile-time indices so the table stays in SGPRs
__device__ __forceinline__ void level_of_n(const LevelTab& lt, int k, int& tb, int& te) {
    tb = lt.ts[0]; te = lt.ts[1];
#pragma unroll
    for (int j = 1; j < OTAL_MAX_LEVELS; ++j)
        if (j < lt.nlev && k >= lt.ns[j]) { tb = lt.ts[j]; te = lt.ts[j + 1]; }
}
__device__ __forceinline__ void level_of_t(const LevelTab& lt, int i, int& kb, int& ke) {
    kb = lt.ns[0]; ke = lt.ns[1];
#pragma unroll
    for (int j = 1; j < OTAL_MAX_LEVELS; ++j)
        if (j < lt.nlev && i >= lt.ts[j]) { kb = lt.ns[j]; ke = lt.ns[j + 1]; }
}

// LDS carve (bytes, all 16-aligned)
struct Carve { int rows, win, g, arg, total; };
static Carve carve(int ROWS, int T, int N, bool bwd, int esz) {       // esz = sizeof(staging type)
    auto up = [](int v) { return (v + 15) & ~15; };
    Carve c;
    const int Tp = T | 1, Np = N | 1;
    c.rows = 0;
    c.win = up(ROWS * Tp * esz);
    c.g = c.win + up(N * 16);
    c.arg = c.g + (bwd ? up(ROWS * Np * esz) : 0);
    c.total = c.arg + (bwd ? up(ROWS * Np * 4) : 0);
    return c;
}

template <typename T, typename S>
__device__ __forceinline__ void stage_rows(S* rows, const T* src, int nrows, int len, int lenp,
                                           int lx, int tid) {
    // 2-D thread map without integer division: (1 << lx) lanes along the row, the rest across rows.  Eight rows are
    // loaded before the first LDS store: with one load in flight per lane (a plain loop: load, wait, store) a 16-row
    // tile paid 16 HBM latencies back to back, which WAS the kernel's run time.
    const int tx = tid & ((1 << lx) - 1), ty = tid >> lx, ny = 256 >> lx;
    for (int i = tx; i < len; i += (1 << lx)) {
        int r = ty;
        for (; r + 7 * ny < nrows; r += 8 * ny) {
            S v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld_as<S>(src, (size_t)(r + u * ny) * len + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) rows[(r + u * ny) * lenp + i] = v[u];
        }
        for (; r < nrows; r += ny) rows[r * lenp + i] = ld_as<S>(src, (size_t)r * len + i);
    }
}

// First maximum of row[l..rr] (strict >, as boundary_max_pooling_kernel.cu:33-40), four elements per trip: indices past
// rr are clamped to rr -- a repeated element can never beat the running maximum, so value and arg-max are unchanged --
// which lets the four LDS reads of a trip issue together instead of one dependent read per element.
template <typename S>
__device__ __forceinline__ S window_max(const S* row, int l, int rr, int& arg) {
    S best = row[l];
    int a = l;
    for (int i = l + 1; i <= rr; i += 4) {
        const int i1 = i + 1 < rr ? i + 1 : rr, i2 = i + 2 < rr ? i + 2 : rr, i3 = i + 3 < rr ? i + 3 : rr;
        const S v0 = row[i], v1 = row[i1], v2 = row[i2], v3 = row[i3];
        if (v0 > best) { best = v0; a = i; }
        if (v1 > best) { best = v1; a = i1; }
        if (v2 > best) { best = v2; a = i2; }
        if (v3 > best) { best = v3; a = i3; }
    }
    arg = a;
    return best;
}

__device__ __forceinline__ void stage_windows(int* win, const float* seg, int N, const LevelTab& lt, int tid) {
    for (int k = tid; k < N; k += 256) {
        int tb, te;
        level_of_n(lt, k, tb, te);
        const int hi = te - tb - 1;
        const float4 s = *reinterpret_cast<const float4*>(seg + (size_t)k * 4);
        int4 w;
        w.x = clampi((int)s.x, 0, hi) + tb;
        w.y = clampi((int)s.y, 0, hi) + tb;
        w.z = clampi((int)s.z, 0, hi) + tb;
        w.w = clampi((int)s.w, 0, hi) + tb;
        *reinterpret_cast<int4*>(win + k * 4) = w;
    }
}

template <typename T, int ROWS>
__global__ __launch_bounds__(256) void bmp_fwd_kernel(const T* __restrict__ in, const float* __restrict__ seg,
                                                      T* __restrict__ out, int C, int Tt, int Nt,
                                                      LevelTab lt, int lxT, int lxN, int off_win, int64_t out_bs) {
    typedef typename StageOf<T>::type S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    S* rows = reinterpret_cast<S*>(smem);
    int* win = reinterpret_cast<int*>(smem + off_win);
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * ROWS;          // global (n*C + c) row; C % ROWS == 0
    const int n = row0 / C, c0 = row0 - n * C;
    const int Tp = Tt | 1;
    stage_rows(rows, in + (size_t)row0 * Tt, ROWS, Tt, Tp, lxT, tid);
    stage_windows(win, seg + (size_t)n * Nt * 4, Nt, lt, tid);
    __syncthreads();
    const int kx = tid & ((1 << lxN) - 1), ky = tid >> lxN, nky = 256 >> lxN;
    const int half = C >> 1;
    for (int r = ky; r < ROWS; r += nky) {
        const int which = (c0 + r) >= half ? 2 : 0;
        const S* row = rows + r * Tp;
        for (int k = kx; k < Nt; k += (1 << lxN)) {
            const int l = win[k * 4 + which], rr = win[k * 4 + which + 1];
            int a;
            st_as<S>(out, (size_t)n * out_bs + (size_t)(c0 + r) * Nt + k, window_max(row, l, rr, a));     // (out may be a channel slice)
        }
    }
}

template <typename T, int ROWS>
__global__ __launch_bounds__(256) void bmp_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ in,
                                                      const float* __restrict__ seg, T* __restrict__ gin,
                                                      int C, int Tt, int Nt, LevelTab lt, int lxT, int lxN,
                                                      int off_win, int off_g, int off_arg, int64_t gout_bs) {
    typedef typename StageOf<T>::type S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    S* rows = reinterpret_cast<S*>(smem);
    int* win = reinterpret_cast<int*>(smem + off_win);
    S* g = reinterpret_cast<S*>(smem + off_g);
    int* arg = reinterpret_cast<int*>(smem + off_arg);
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * ROWS;
    const int n = row0 / C, c0 = row0 - n * C;
    const int Tp = Tt | 1, Np = Nt | 1;
    stage_rows(rows, in + (size_t)row0 * Tt, ROWS, Tt, Tp, lxT, tid);
    stage_rows(g, gout + (size_t)n * gout_bs + (size_t)c0 * Nt, ROWS, Nt, Np, lxN, tid);     // (grad_out may be a channel slice)
    stage_windows(win, seg + (size_t)n * Nt * 4, Nt, lt, tid);
    __syncthreads();
    {   // phase 1: arg-max per (row, proposal)
        const int kx = tid & ((1 << lxN) - 1), ky = tid >> lxN, nky = 256 >> lxN;
        const int half = C >> 1;
        for (int r = ky; r < ROWS; r += nky) {
            const int which = (c0 + r) >= half ? 2 : 0;
            const S* row = rows + r * Tp;
            for (int k = kx; k < Nt; k += (1 << lxN)) {
                const int l = win[k * 4 + which], rr = win[k * 4 + which + 1];
                int a;
                window_max(row, l, rr, a);
                arg[r * Np + k] = a;
            }
        }
    }
    __syncthreads();
    // phase 2: scatter in LDS.  The input tile is no longer needed, its LDS space becomes the grad_in tile.
    for (int i = tid; i < ROWS * Tp; i += 256) rows[i] = (S)0;
    __syncthreads();
    // One lane per row walks the proposals in ascending k and adds grad_out into grad_in[arg-max] -- a
    // dependent chain per row (rows are independent), entirely in LDS, so every position receives its
    // contributions in ascending k (deterministic) at O(N) per row instead of O(T*N).
    // Measured alternatives, all slower on MI355X (level call / frame call, us): this chain 13.7 / 21.3; sixteen lanes
    // per row, lane j owning the positions = j mod 16 and walking all proposals, 19.6 / 26.3 (LDS issue-bound: every lane
    // reads every index); fire-and-forget ds_add_f32 from the one lane 20.8 / 32.1; per-position bitmasks of 32 proposals
    // gathered by (row, position) lanes 29.8 / 30.1.  What did pay was in front of this phase: row staging with eight
    // loads in flight and the four-wide window scan (17.6 / 31.0 before).
    if (tid < ROWS) {
        S* gi = rows + tid * Tp;
        const int* ar = arg + tid * Np;
        const S* gr = g + tid * Np;
        for (int k = 0; k < Nt; ++k) gi[ar[k]] += gr[k];
    }
    __syncthreads();
    {
        const int tx = tid & ((1 << lxT) - 1), ty = tid >> lxT, nty = 256 >> lxT;
        for (int r = ty; r < ROWS; r += nty)
            for (int i = tx; i < Tt; i += (1 << lxT))
                st_as<S>(gin, (size_t)(row0 + r) * Tt + i, rows[r * Tp + i]);
    }
}

template <typename T>
int launch_fwd(const T* in, const float* seg, T* out, int B, int C, int Tt, int Nt, const LevelTab& lt, hipStream_t s,
               int64_t out_bs = 0) {
    if (out_bs == 0) out_bs = (int64_t)C * Nt;
    const int lxT = ilog2_ceil(Tt < 256 ? Tt : 256), lxN = ilog2_ceil(Nt < 256 ? Nt : 256);
    const int cand[4] = {16, 8, 4, 2};
    for (int ci = 0; ci < 4; ++ci) {
        const int R = cand[ci];
        if (C % R) continue;
        const Carve cv = carve(R, Tt, Nt, false, (int)sizeof(typename StageOf<T>::type));
        if (cv.total > 48 * 1024 && R > 2) continue;
        if (cv.total > 64 * 1024) return OTAL_E_UNSUPPORTED;
        const dim3 grid((unsigned)((size_t)B * C / R));
#define OTAL_FWD(RR) hipLaunchKernelGGL((bmp_fwd_kernel<T, RR>), grid, dim3(256), cv.total, s, in, seg, out, C, Tt, Nt, lt, lxT, lxN, cv.win, out_bs)
        if (R == 16) OTAL_FWD(16); else if (R == 8) OTAL_FWD(8); else if (R == 4) OTAL_FWD(4); else OTAL_FWD(2);
#undef OTAL_FWD
        return otal_launch_status();
    }
    return OTAL_E_UNSUPPORTED;
}

template <typename T>
int launch_bwd(const T* gout, const T* in, const float* seg, T* gin, int B, int C, int Tt, int Nt,
               const LevelTab& lt, hipStream_t s, int64_t gout_bs = 0) {
    if (gout_bs == 0) gout_bs = (int64_t)C * Nt;
    const int lxT = ilog2_ceil(Tt < 256 ? Tt : 256), lxN = ilog2_ceil(Nt < 256 ? Nt : 256);
    const int cand[4] = {16, 8, 4, 2};
    for (int ci = 0; ci < 4; ++ci) {
        const int R = cand[ci];
        if (C % R) continue;
        const Carve cv = carve(R, Tt, Nt, true, (int)sizeof(typename StageOf<T>::type));
        if (cv.total > 48 * 1024 && R > 2) continue;
        if (cv.total > 64 * 1024) return OTAL_E_UNSUPPORTED;
        const dim3 grid((unsigned)((size_t)B * C / R));
#define OTAL_BWD(RR) hipLaunchKernelGGL((bmp_bwd_kernel<T, RR>), grid, dim3(256), cv.total, s, gout, in, seg, gin, C, Tt, Nt, lt, lxT, lxN, cv.win, cv.g, cv.arg, gout_bs)
        if (R == 16) OTAL_BWD(16); else if (R == 8) OTAL_BWD(8); else if (R == 4) OTAL_BWD(4); else OTAL_BWD(2);
#undef OTAL_BWD
        return otal_launch_status();
    }
    return OTAL_E_UNSUPPORTED;
}

int check_levels(int nlev, const int* t_start, const int* n_start, LevelTab& lt) {
    if (nlev < 1 || nlev > OTAL_MAX_LEVELS || !t_start || !n_start) return OTAL_E_LEVELS;
    if (t_start[0] != 0 || n_start[0] != 0) return OTAL_E_LEVELS;
    lt.nlev = nlev;
    for (int i = 0; i <= OTAL_MAX_LEVELS; ++i) {
        lt.ts[i] = t_start[i <= nlev ? i : nlev];
        lt.ns[i] = n_start[i <= nlev ? i : nlev];
        if (i > 0 && i <= nlev && (lt.ts[i] <= lt.ts[i - 1] || lt.ns[i] <= lt.ns[i - 1])) return OTAL_E_LEVELS;
    }
    return 0;
}

}  // namespace

extern "C" int otal_bmp_fwd_levels(const void* in, const float* seg, void* out, int B, int C, int nlev,
                                   const int* t_start, const int* n_start, int dtype, void* stream) {
    if (!in || !seg || !out) return OTAL_E_NULL;
    if (B <= 0 || C <= 0) return OTAL_E_SHAPE;
    if (C & 1) return OTAL_E_ODD_C;
    LevelTab lt;
    if (int e = check_levels(nlev, t_start, n_start, lt)) return e;
    const int Tt = lt.ts[nlev], Nt = lt.ns[nlev];
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OTAL_F32) return launch_fwd<float>((const float*)in, seg, (float*)out, B, C, Tt, Nt, lt, s);
    if (dtype == OTAL_BF16) return launch_fwd<bf16_t>((const bf16_t*)in, seg, (bf16_t*)out, B, C, Tt, Nt, lt, s);
    if (dtype == OTAL_F16) return launch_fwd<f16_t>((const f16_t*)in, seg, (f16_t*)out, B, C, Tt, Nt, lt, s);
    if (dtype == OTAL_F64) return launch_fwd<double>((const double*)in, seg, (double*)out, B, C, Tt, Nt, lt, s);
    return OTAL_E_DTYPE;
}

extern "C" int otal_bmp_bwd_levels(const void* gout, const void* in, const float* seg, void* gin, int B, int C,
                                   int nlev, const int* t_start, const int* n_start, int dtype, void* stream) {
    if (!gout || !in || !seg || !gin) return OTAL_E_NULL;
    if (B <= 0 || C <= 0) return OTAL_E_SHAPE;
    if (C & 1) return OTAL_E_ODD_C;
    LevelTab lt;
    if (int e = check_levels(nlev, t_start, n_start, lt)) return e;
    const int Tt = lt.ts[nlev], Nt = lt.ns[nlev];
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OTAL_F32) return launch_bwd<float>((const float*)gout, (const float*)in, seg, (float*)gin, B, C, Tt, Nt, lt, s);
    if (dtype == OTAL_BF16) return launch_bwd<bf16_t>((const bf16_t*)gout, (const bf16_t*)in, seg, (bf16_t*)gin, B, C, Tt, Nt, lt, s);
    if (dtype == OTAL_F16) return launch_bwd<f16_t>((const f16_t*)gout, (const f16_t*)in, seg, (f16_t*)gin, B, C, Tt, Nt, lt, s);
    if (dtype == OTAL_F64) return launch_bwd<double>((const double*)gout, (const double*)in, seg, (double*)gin, B, C, Tt, Nt, lt, s);
    return OTAL_E_DTYPE;
}

// The level-batched calls with the POOLED tensor as a channel slice of a wider (B,Ctot,N) buffer (pooled_bs = Ctot * N
// elements between samples): the ProposalBranch concatenation torch.cat([roi, pooled, short], 1) (AFSD/thumos14/BDNet.py:111)
// is written in place going forward and its gradient read in place going backward -- no copy either way.  fp32.
extern "C" int otal_bmp_fwd_levels_to(const float* in, const float* seg, float* out, int64_t pooled_bs, int B, int C, int nlev,
                                      const int* t_start, const int* n_start, void* stream) {
    if (!in || !seg || !out) return OTAL_E_NULL;
    if (B <= 0 || C <= 0) return OTAL_E_SHAPE;
    if (C & 1) return OTAL_E_ODD_C;
    LevelTab lt;
    if (int e = check_levels(nlev, t_start, n_start, lt)) return e;
    const int Tt = lt.ts[nlev], Nt = lt.ns[nlev];
    if (pooled_bs < (int64_t)C * Nt) return OTAL_E_SHAPE;
    return launch_fwd<float>(in, seg, out, B, C, Tt, Nt, lt, (hipStream_t)stream, pooled_bs);
}
extern "C" int otal_bmp_bwd_levels_from(const float* gout, int64_t pooled_bs, const float* in, const float* seg, float* gin, int B,
                                        int C, int nlev, const int* t_start, const int* n_start, void* stream) {
    if (!gout || !in || !seg || !gin) return OTAL_E_NULL;
    if (B <= 0 || C <= 0) return OTAL_E_SHAPE;
    if (C & 1) return OTAL_E_ODD_C;
    LevelTab lt;
    if (int e = check_levels(nlev, t_start, n_start, lt)) return e;
    const int Tt = lt.ts[nlev], Nt = lt.ns[nlev];
    if (pooled_bs < (int64_t)C * Nt) return OTAL_E_SHAPE;
    return launch_bwd<float>(gout, in, seg, gin, B, C, Tt, Nt, lt, (hipStream_t)stream, pooled_bs);
}

extern "C" int otal_bmp_fwd(const void* in, const float* seg, void* out, int B, int C, int T, int N,
                            int seg_batch, int dtype, void* stream) {
    if (T <= 0 || N <= 0) return OTAL_E_SHAPE;
    if (seg_batch != B) return OTAL_E_BATCH;
    const int ts[2] = {0, T}, ns[2] = {0, N};
    return otal_bmp_fwd_levels(in, seg, out, B, C, 1, ts, ns, dtype, stream);
}

extern "C" int otal_bmp_bwd(const void* gout, const void* in, const float* seg, void* gin, int B, int C, int T,
                            int N, int seg_batch, int compat_ref_stride, int dtype, void* stream) {
    if (T <= 0 || N <= 0) return OTAL_E_SHAPE;
    if (seg_batch != B) return OTAL_E_BATCH;
    if (!gin) return OTAL_E_NULL;
    int Teff = T;
    if (compat_ref_stride && N != T) {
        // reference launcher: tscale = grad_output.size(2) = N (boundary_max_pooling_kernel.cu:121).
        // Same kernel over the same buffers viewed as (B*C) rows of N; the tail stays zero.
        if (N > T) return OTAL_E_SHAPE;   // the reference would write out of bounds here
        if (dtype < OTAL_F32 || dtype > OTAL_F64) return OTAL_E_DTYPE;
        const size_t esz = dtype == OTAL_F32 ? 4 : (dtype == OTAL_F64 ? 8 : 2);
        hipError_t e = hipMemsetAsync(gin, 0, (size_t)B * C * T * esz, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
        Teff = N;
    }
    const int ts[2] = {0, Teff}, ns[2] = {0, N};
    return otal_bmp_bwd_levels(gout, in, seg, gin, B, C, 1, ts, ns, dtype, stream);
}
