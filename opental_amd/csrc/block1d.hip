// opental_amd/csrc/block1d.hip -- the 1-D blocks of the pyramid as ONE launch each (ABI 24, include/opental_hip.h).
//
// Replaces, per Unit1D + GroupNorm(32, C) + ReLU block of AFSD/thumos14/BDNet.py:67-103,:129-203,:274-284 (Unit1D =
// AFSD/common/layers.py:178-214): the convolution launch (+ its split-K reduce) and the GroupNorm launch of the forward pass,
// and the GroupNorm-backward launch + the consumers' data-gradient launches (+ autograd's gradient adds) of the backward
// pass.  Why one workgroup per (sample, group): GroupNorm needs the statistics of a whole (sample, group, level) before any
// element can be normalised.  With 32-row x 64-position conv tiles those statistics were a grid-wide dependency (a second
// launch); here the workgroup that computes the group's 16 (32) rows over ALL positions of its levels holds them in LDS and
// normalises on the spot.  The GEMM is tiny per workgroup (16 x <=256 x K<=6144): what bounds the kernel is staging the
// operands, so the activations are read ONCE per workgroup (the tiled kernel read every 64-position slab once per 32-row
// block: 16 times) and the weights once per (sample, group).
//
// Main loop (conv1d_tile.inc's scheme at M = 16): K runs over segments (independent source tensors with their own weight
// packs -- the pieces of a torch.cat, the consumers whose data gradients meet in one tensor) and, inside a segment, over
// chunks of kc channels, double buffered in LDS:
//   A = rows of the bf16 operand pack (k = ((c/8)*kt + tap)*8 + c%8): a chunk is a contiguous run of every row;
//   B = the chunk's source rows, transposed while staged to [position][channel] bf16 cells, so the MFMA fragment of tap dt
//       at position n is 16 contiguous bytes of cell (n*mul + off + sgn*dt) >> shr -- strides, upsampling and the data
//       gradient's reversed taps are index arithmetic of the read, level boundaries a per-lane zero select.
// v_mfma_f32_16x16x32_bf16: A row = lane & 15, B column = lane & 15, both k group = lane >> 4; C/D column = lane & 15,
// row = 4 * (lane >> 4) + r.
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Words4 { unsigned a, b, c, d; };
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {     // v_cvt_pk_bf16_f32: RNE, lo in bits 0..15
    const hw_f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, hw_bf16x2));
}

constexpr int MAXB = 9;         // staging trips of the source cells: (cells / 4) * (kc / 2) <= 2304 items
constexpr int MAXA = 6;         // staging trips of the weight rows: cpg * kt * kc / 8 <= 1536 pieces
constexpr int MAXTW = 4;        // position tiles (16 wide) per wave: ranges of <= 256 positions

struct Launch {
    otal_b1d_problem p[OTAL_B1D_MAX_PROB];
    int n;
    int wg_start[OTAL_B1D_MAX_PROB + 1];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
// every (channel c, position t) of a level of `len` positions with `nlanes` lanes, without a division per element (gn.hip)
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

__device__ __forceinline__ int seg_cell0(const otal_b1d_seg& s, int n0) {
    const int qlo = n0 * s.mul + s.off + (s.sgn < 0 ? -(s.kt - 1) : 0);
    return ((qlo >> s.shr) >> 2) << 2;          // (arithmetic shifts: a multiple of four at or below the first source position)
}
__device__ __forceinline__ int seg_cells(const otal_b1d_seg& s, int n0, int n1) {
    const int qhi = (n1 - 1) * s.mul + s.off + (s.sgn > 0 ? s.kt - 1 : 0);
    return ((((qhi >> s.shr) - seg_cell0(s, n0) + 1) + 3) >> 2) << 2;
}

__global__ __launch_bounds__(256) void block1d_kernel(const Launch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < OTAL_B1D_MAX_PROB; ++i)
        if (i < L.n && (int)blockIdx.x >= L.wg_start[i]) pi = i;
    const otal_b1d_problem& P = L.p[pi];
    // workgroup -> (sample fastest: the eight XCDs receive workgroups round-robin, so with B = 8 an XCD's L2 holds ONE sample's
    // activations next to the weight packs), range, group
    const int local = (int)blockIdx.x - L.wg_start[pi];
    const int b = local % P.B, rest = local / P.B;
    const int r = rest % P.nrange, g = rest / P.nrange;
    const int cpg = P.cpg, kc = P.kc;
    const int m0 = g * cpg;
    const int lvA = P.range_lev[r], lvB = P.range_lev[r + 1];
    const int n0 = P.lev[lvA], n1 = P.lev[lvB];
    const int N = n1 - n0;
    const int ntiles = (N + 15) >> 4;
    const int mt = cpg >> 4;                                 // 16-row MFMA tiles: 1 or 2
    const int CP = kc * 2 + 16;                              // cell pitch, bytes

    // LDS: [2][XB] cells, [2][AB] weight rows; sized by the host for the launch's largest problem
    int XB = 0, AB = 0;
#pragma unroll
    for (int s = 0; s < OTAL_B1D_MAX_SEG; ++s)
        if (s < P.nseg) {
            XB = max(XB, seg_cells(P.seg[s], n0, n1) * CP);
            AB = max(AB, cpg * (P.seg[s].kt * kc * 2 + 16));
        }

    // ---- this lane's output positions (one per tile it owns) and their level bounds
    const int col = lane & 15, kb = lane >> 4;
    int npos[MAXTW], lo_[MAXTW], up_[MAXTW];
    bool live[MAXTW];
#pragma unroll
    for (int j = 0; j < MAXTW; ++j) {
        const int tile = wave + 4 * j;
        const int n = n0 + tile * 16 + col;
        live[j] = tile < ntiles && n < n1;
        npos[j] = n;
        int lo = 0, up = P.T;
        if (P.nlev > 1) {
#pragma unroll
            for (int l = 0; l < OTAL_MAX_LEVELS; ++l)
                if (l < P.nlev && n >= P.lev[l]) { lo = P.lev[l]; up = P.lev[l + 1]; }
        }
        lo_[j] = lo; up_[j] = up;
    }
    const int ntw = (ntiles - wave + 3) >> 2;                // tiles of this wave (wave-uniform)

    f32x4 acc[2][MAXTW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MAXTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging state of the segment being loaded
    unsigned bvo[MAXB];
    int blds[MAXB];
    unsigned avo[MAXA];
    int alds[MAXA];
    int b_items = 0, a_pieces = 0;
    unsigned cs4 = 0;                                        // one channel of the source, bytes
    __amdgpu_buffer_rsrc_t rs, rw;
    auto setup_stage = [&](const otal_b1d_seg& S) {
        const int cell0 = seg_cell0(S, n0), ncell = seg_cells(S, n0, n1);
        const int quads = ncell >> 2;
        b_items = quads * (kc >> 1);
        cs4 = (unsigned)(S.src_cs * 4);
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            const int item = tid + 256 * i;
            const int q = item % quads, cp = item / quads;
            bvo[i] = item < b_items ? (unsigned)(((int64_t)b * S.src_bs + (int64_t)(2 * cp) * S.src_cs + (cell0 + 4 * q)) * 4) : 0xffffffffu;
            blds[i] = (4 * q) * CP + cp * 4;
        }
        const int ppr = S.kt * kc >> 3;                      // 16-byte pieces per row and chunk
        const int AP = S.kt * kc * 2 + 16;
        a_pieces = cpg * ppr;
#pragma unroll
        for (int j = 0; j < MAXA; ++j) {
            const int p = tid + 256 * j;
            const int m = p / ppr, q = p - m * ppr;
            avo[j] = p < a_pieces ? (unsigned)(((int64_t)(m0 + m) * S.wp_pitch) * 2 + q * 16) : 0xffffffffu;
            alds[j] = m * AP + q * 16;
        }
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.src), 0, (int)(S.src_elems * 4), 0x00020000);
        rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(S.wp), 0, (int)(S.wp_elems * 2), 0x00020000);
    };
    Words4 rb[MAXB][2];
    Words4 ra[MAXA];
    auto load = [&](const otal_b1d_seg& S, int chunk) {
        const int so = (int)((int64_t)chunk * kc * S.src_cs * 4);
#pragma unroll
        for (int i = 0; i < MAXB; ++i)
            if (i * 256 < b_items) {
                rb[i][0] = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rs, bvo[i], so, 0));
                rb[i][1] = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rs, bvo[i], so + (int)cs4, 0));
            }
        const int ao = chunk * (kc * S.kt * 2);
#pragma unroll
        for (int j = 0; j < MAXA; ++j)
            if (j * 256 < a_pieces) ra[j] = __builtin_bit_cast(Words4, __builtin_amdgcn_raw_buffer_load_b128(rw, avo[j], ao, 0));
    };
    auto store = [&](int buf) {
        unsigned char* X = smem + buf * XB;
        unsigned char* A = smem + 2 * XB + buf * AB;
#pragma unroll
        for (int i = 0; i < MAXB; ++i)
            if (tid + 256 * i < b_items) {
                *reinterpret_cast<unsigned*>(X + blds[i]) = cvt_pk_bf16(__uint_as_float(rb[i][0].a), __uint_as_float(rb[i][1].a));
                *reinterpret_cast<unsigned*>(X + blds[i] + CP) = cvt_pk_bf16(__uint_as_float(rb[i][0].b), __uint_as_float(rb[i][1].b));
                *reinterpret_cast<unsigned*>(X + blds[i] + 2 * CP) = cvt_pk_bf16(__uint_as_float(rb[i][0].c), __uint_as_float(rb[i][1].c));
                *reinterpret_cast<unsigned*>(X + blds[i] + 3 * CP) = cvt_pk_bf16(__uint_as_float(rb[i][0].d), __uint_as_float(rb[i][1].d));
            }
#pragma unroll
        for (int j = 0; j < MAXA; ++j)
            if (tid + 256 * j < a_pieces) *reinterpret_cast<Words4*>(A + alds[j]) = ra[j];
    };

    // ---- fragment state of the segment being multiplied
    int nq[MAXTW];
    unsigned okm[MAXTW];
    int f_cell0 = 0, f_ncell = 0, f_kt = 1, f_sgn = 1, f_shr = 0, f_AP = 0;
    auto setup_frag = [&](const otal_b1d_seg& S) {
        f_cell0 = seg_cell0(S, n0); f_ncell = seg_cells(S, n0, n1);
        f_kt = S.kt; f_sgn = S.sgn; f_shr = S.shr; f_AP = S.kt * kc * 2 + 16;
#pragma unroll
        for (int j = 0; j < MAXTW; ++j) {
            const int q0 = npos[j] * S.mul + S.off;
            const int lo = S.use_levels ? lo_[j] : 0, up = S.use_levels ? up_[j] : S.Tv;
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int q = q0 + S.sgn * t;
                m |= (unsigned)(live[j] && t < S.kt && q >= lo && q < up && (q & S.par) == 0) << t;
            }
            nq[j] = q0; okm[j] = m;
        }
    };

    if (P.nseg > 0) {
        int seg = 0, chunk = 0;
        setup_stage(P.seg[0]);
        load(P.seg[0], 0);
        store(0);
        setup_frag(P.seg[0]);
        __syncthreads();
        int buf = 0;
        while (true) {
            int nseg_ = seg, nchunk = chunk + 1;
            if (nchunk * kc >= P.seg[seg].C) { nseg_ = seg + 1; nchunk = 0; }
            const bool more = nseg_ < P.nseg;
            if (more) {
                if (nseg_ != seg) setup_stage(P.seg[nseg_]);
                load(P.seg[nseg_], nchunk);
            }
            const unsigned char* X = smem + buf * XB;
            const unsigned char* A = smem + 2 * XB + buf * AB + col * f_AP;
            const int steps = f_kt * kc >> 5;                // k32 steps of the chunk
            for (int s = 0; s < steps; ++s) {
                const int q8 = 4 * s + kb;                   // this lane's k8 group: (channel block q8 / kt, tap q8 % kt)
                const int cb = f_kt == 3 ? q8 / 3 : q8, tap = q8 - cb * f_kt;
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(A + q8 * 16);
                bf16x8 a1 = a0;
                if (mt == 2) a1 = *reinterpret_cast<const bf16x8*>(A + 16 * f_AP + q8 * 16);
#pragma unroll
                for (int j = 0; j < MAXTW; ++j)
                    if (j < ntw) {
                        int cell = ((nq[j] + f_sgn * tap) >> f_shr) - f_cell0;
                        cell = min(max(cell, 0), f_ncell - 1);
                        bf16x8 bv = *reinterpret_cast<const bf16x8*>(X + cell * CP + cb * 16);
                        const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                        bv = ((okm[j] >> tap) & 1u) ? bv : zero;
                        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bv, acc[0][j], 0, 0, 0);
                        if (mt == 2) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bv, acc[1][j], 0, 0, 0);
                    }
            }
            if (more) store(buf ^ 1);
            __syncthreads();
            if (!more) break;
            if (nseg_ != seg) setup_frag(P.seg[nseg_]);
            seg = nseg_; chunk = nchunk; buf ^= 1;
        }
    }

    // ================================================================ epilogues (the operand buffers are free)
    float* const t0 = reinterpret_cast<float*>(smem);        // [cpg][N]: c (FWD), dy -> dyh (BWD), acc + adds (PLAIN)
    float* const t1 = t0 + cpg * N;                          // BWD: c -> xhat
    float* const t2 = t1 + cpg * N;                          // BWD: dc (for the channel sums)
    float* const red = t2 + cpg * N;                         // 8 floats
    float* const gam = red + 8;
    float* const bet = gam + cpg;
    const int G = P.M / cpg;
    // accumulators (+ bias) -> t0
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (i < mt) {
            float bs[4] = {0.f, 0.f, 0.f, 0.f};
            if (P.epilogue == OTAL_B1D_FWD && P.bias) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bs[q] = P.bias[m0 + 16 * i + 4 * kb + q];
            }
#pragma unroll
            for (int j = 0; j < MAXTW; ++j)
                if (j < ntw) {
                    const int c = (wave + 4 * j) * 16 + col;
                    if (c < N) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) t0[(16 * i + 4 * kb + q) * N + c] = acc[i][j][q] + bs[q];
                    }
                }
        }
    if (tid < cpg && P.epilogue != OTAL_B1D_PLAIN) { gam[tid] = P.gamma[m0 + tid]; bet[tid] = P.beta[m0 + tid]; }
    __syncthreads();
    // rows of the group: wave w takes rows w, w + 4, ...; lanes along the positions (coalesced global rows)
    if (P.epilogue != OTAL_B1D_FWD) {
        for (int c = wave; c < cpg; c += 4)
            for (int t = lane; t < N; t += 64) {
                float v = t0[c * N + t];
#pragma unroll
                for (int k = 0; k < OTAL_B1D_MAX_ADD; ++k)
                    if (k < P.nadd && n0 + t < P.add[k].Ta)
                        v += P.add[k].p[(int64_t)b * P.add[k].bs + (int64_t)(m0 + c) * P.add[k].cs + n0 + t];
                if (P.epilogue == OTAL_B1D_PLAIN) {
                    P.y[(int64_t)b * P.y_bs + (int64_t)(m0 + c) * P.y_cs + n0 + t] = v;
                } else {
                    t0[c * N + t] = v;
                    t1[c * N + t] = P.c[(int64_t)b * P.c_bs + (int64_t)(m0 + c) * P.c_cs + n0 + t];
                }
            }
        if (P.epilogue == OTAL_B1D_PLAIN) return;
        __syncthreads();
    }
    const bool solo = lvB - lvA == 1;                        // one level: the whole workgroup reduces it; else a wave per level
    const int first = solo ? tid : lane, stride = solo ? 256 : 64;
    if (P.epilogue == OTAL_B1D_FWD) {
        for (int c = wave; c < cpg; c += 4)                  // the convolution output, kept for the backward pass
            for (int t = lane; t < N; t += 64) P.c[(int64_t)b * P.c_bs + (int64_t)(m0 + c) * P.c_cs + n0 + t] = t0[c * N + t];
#pragma unroll 1
        for (int l = solo ? lvA : lvA + wave; l < lvB; l += solo ? 1 : 4) {
            const int lo = P.lev[l] - n0, len = P.lev[l + 1] - P.lev[l];
            const int cnt = cpg * len;
            float s = 0.f;
            for_level(first, stride, cpg, len, [&](int c, int t) { s += t0[c * N + lo + t]; });
            const float mean = (solo ? block_sum(s, red, tid) : __shfl(wave_sum(s), 0, 64)) / (float)cnt;
            float q = 0.f;
            for_level(first, stride, cpg, len, [&](int c, int t) { const float d = t0[c * N + lo + t] - mean; q += d * d; });
            const float var = (solo ? block_sum(q, red, tid) : __shfl(wave_sum(q), 0, 64)) / (float)cnt;
            const float rstd = 1.0f / sqrtf(var + P.eps);
            if (first == 0) {
                P.stats[(((int64_t)b * G + g) * P.nlev + l) * 2 + 0] = mean;
                P.stats[(((int64_t)b * G + g) * P.nlev + l) * 2 + 1] = rstd;
            }
            for_level(first, stride, cpg, len, [&](int c, int t) {
                float v = (t0[c * N + lo + t] - mean) * rstd * gam[c] + bet[c];
                if (P.relu) v = fmaxf(v, 0.f);
                P.y[(int64_t)b * P.y_bs + (int64_t)(m0 + c) * P.y_cs + n0 + lo + t] = v;
            });
        }
        return;
    }
    // ---- GroupNorm + ReLU backward (gn.hip's arithmetic): t0 = dy, t1 = c
#pragma unroll 1
    for (int l = solo ? lvA : lvA + wave; l < lvB; l += solo ? 1 : 4) {
        const int lo = P.lev[l] - n0, len = P.lev[l + 1] - P.lev[l];
        const int cnt = cpg * len;
        const float mean = P.stats[(((int64_t)b * G + g) * P.nlev + l) * 2 + 0];
        const float rstd = P.stats[(((int64_t)b * G + g) * P.nlev + l) * 2 + 1];
        float s1 = 0.f, s2 = 0.f;
        for_level(first, stride, cpg, len, [&](int c, int t) {
            const int p = c * N + lo + t;
            const float ga = gam[c];
            const float xh = (t1[p] - mean) * rstd;
            float d = t0[p];
            if (P.relu && !(xh * ga + bet[c] > 0.f)) d = 0.f;
            t1[p] = xh;
            t0[p] = d;
            const float dg = d * ga;
            s1 += dg;
            s2 += dg * xh;
        });
        float m1, m2;
        if (solo) {
            m1 = block_sum(s1, red, tid) / (float)cnt;
            m2 = block_sum(s2, red, tid) / (float)cnt;
        } else {
            m1 = __shfl(wave_sum(s1), 0, 64) / (float)cnt;
            m2 = __shfl(wave_sum(s2), 0, 64) / (float)cnt;
        }
        for_level(first, stride, cpg, len, [&](int c, int t) {
            const int p = c * N + lo + t;
            const float v = rstd * (t0[p] * gam[c] - m1 - t1[p] * m2);
            P.y[(int64_t)b * P.y_bs + (int64_t)(m0 + c) * P.y_cs + n0 + lo + t] = v;
            t2[p] = v;
        });
    }
    __syncthreads();
    // per-channel sums over this range's positions (fixed order): one wave per channel
    for (int c = wave; c < cpg; c += 4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int t = lane; t < N; t += 64) {
            const float d = t0[c * N + t], xh = t1[c * N + t];
            a0 += d * xh;
            a1 += d;
            a2 += t2[c * N + t];
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) {
            float* p = P.partial + ((int64_t)b * P.nrange + r) * 3 * P.M + m0 + c;
            p[0] = a0; p[P.M] = a1; p[2 * P.M] = a2;
        }
    }
}

// ---- operand packs
struct PackItem { const float* w; unsigned short* fwd; unsigned short* dgrad; int Cout, Cin, kt, first_block; };
static_assert(sizeof(PackItem) == 40, "PackItem layout (include/opental_hip.h)");

__global__ __launch_bounds__(256) void b1d_pack_kernel(const PackItem* __restrict__ items, int n) {
    int i = 0;
    for (int k = 1; k < n; ++k)
        if ((int)blockIdx.x >= items[k].first_block) i = k;
    const PackItem it = items[i];
    const int grp = ((int)blockIdx.x - it.first_block) * 256 + threadIdx.x;     // one 8-element group of a pack row
    const int kt = it.kt, Cout = it.Cout, Cin = it.Cin;
    const int groups = Cout * Cin * kt / 8;
    if (grp >= groups) return;
    if (it.fwd) {           // row co, group (cb, tap): w[co][8 cb .. 8 cb + 7][tap]
        const int gpr = Cin / 8 * kt;
        const int co = grp / gpr, q = grp - co * gpr;
        const int cb = q / kt, tap = q - cb * kt;
        const float* src = it.w + ((int64_t)co * Cin + cb * 8) * kt + tap;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e * kt];
        Words4 o = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
        *reinterpret_cast<Words4*>(it.fwd + (int64_t)grp * 8) = o;
    }
    if (it.dgrad) {         // row ci, group (cob, tap): w[8 cob .. 8 cob + 7][ci][tap]
        const int gpr = Cout / 8 * kt;
        const int ci = grp / gpr, q = grp - ci * gpr;
        const int cob = q / kt, tap = q - cob * kt;
        const float* src = it.w + ((int64_t)(cob * 8) * Cin + ci) * kt + tap;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(int64_t)e * Cin * kt];
        Words4 o = {cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
        *reinterpret_cast<Words4*>(it.dgrad + (int64_t)grp * 8) = o;
    }
}

inline int host_cell0(const otal_b1d_seg& s, int n0) {
    const int qlo = n0 * s.mul + s.off + (s.sgn < 0 ? -(s.kt - 1) : 0);
    return ((qlo >> s.shr) >> 2) * 4;
}
inline int host_cells(const otal_b1d_seg& s, int n0, int n1) {
    const int qhi = (n1 - 1) * s.mul + s.off + (s.sgn > 0 ? s.kt - 1 : 0);
    return (((qhi >> s.shr) - host_cell0(s, n0) + 1) + 3) / 4 * 4;
}

}  // namespace

extern "C" size_t otal_b1d_problem_bytes(void) { return sizeof(otal_b1d_problem); }

extern "C" int otal_b1d_launch(const otal_b1d_problem* problems, int n, void* stream) {
    if (!problems) return OTAL_E_NULL;
    if (n <= 0 || n > OTAL_B1D_MAX_PROB) return OTAL_E_SHAPE;
    static_assert(sizeof(Launch) <= 4096, "kernel arguments: 4 KB");
    Launch L;
    L.n = n;
    size_t lds = 0;
    int wgs = 0;
    for (int i = 0; i < n; ++i) {
        const otal_b1d_problem& P = problems[i];
        if (P.epilogue < 0 || P.epilogue > 2) return OTAL_E_SHAPE;
        if (P.B <= 0 || P.M <= 0 || P.T <= 0 || P.nseg < 0 || P.nseg > OTAL_B1D_MAX_SEG || P.nadd < 0 || P.nadd > OTAL_B1D_MAX_ADD)
            return OTAL_E_SHAPE;
        if ((P.cpg != 16 && P.cpg != 32) || P.M % P.cpg || (P.kc != 64 && P.kc != 128)) return OTAL_E_UNSUPPORTED;
        if (P.nlev < 1 || P.nlev > OTAL_MAX_LEVELS || P.lev[0] != 0 || P.lev[P.nlev] != P.T) return OTAL_E_LEVELS;
        for (int l = 0; l < P.nlev; ++l) if (P.lev[l + 1] <= P.lev[l]) return OTAL_E_LEVELS;
        if (P.nrange < 1 || P.nrange > OTAL_B1D_MAX_RANGE || P.range_lev[0] != 0 || P.range_lev[P.nrange] != P.nlev) return OTAL_E_LEVELS;
        if (!P.y) return OTAL_E_NULL;
        if (P.epilogue != OTAL_B1D_PLAIN && (!P.c || !P.stats || !P.gamma || !P.beta)) return OTAL_E_NULL;
        if (P.epilogue == OTAL_B1D_BWD && !P.partial) return OTAL_E_NULL;
        if (P.nseg == 0 && P.epilogue == OTAL_B1D_FWD) return OTAL_E_SHAPE;
        for (int k = 0; k < P.nadd; ++k) if (!P.add[k].p) return OTAL_E_NULL;
        int nmax = 0;
        size_t stage = 0;
        for (int r = 0; r < P.nrange; ++r) {
            if (P.range_lev[r + 1] <= P.range_lev[r]) return OTAL_E_LEVELS;
            const int n0 = P.lev[P.range_lev[r]], n1 = P.lev[P.range_lev[r + 1]];
            if (n1 - n0 > 256) return OTAL_E_UNSUPPORTED;
            nmax = n1 - n0 > nmax ? n1 - n0 : nmax;
            size_t xb = 0, ab = 0;
            for (int s = 0; s < P.nseg; ++s) {
                const otal_b1d_seg& S = P.seg[s];
                if (!S.src || !S.wp) return OTAL_E_NULL;
                if ((S.kt != 1 && S.kt != 3) || S.C <= 0 || S.C % P.kc || S.shr < 0 || S.shr > 2 || (S.sgn != 1 && S.sgn != -1))
                    return OTAL_E_UNSUPPORTED;
                if (((uintptr_t)S.src & 7) || (S.src_cs & 1) || (S.src_bs & 1) || ((uintptr_t)S.wp & 15) || (S.wp_pitch & 7))
                    return OTAL_E_UNSUPPORTED;
                if (S.src_elems <= 0 || S.src_elems >= (1LL << 29) || S.wp_elems <= 0 || S.wp_elems >= (1LL << 30)) return OTAL_E_UNSUPPORTED;
                const int cells = host_cells(S, n0, n1);
                if ((cells / 4) * (P.kc / 2) > MAXB * 256 || P.cpg * S.kt * P.kc / 8 > MAXA * 256) return OTAL_E_UNSUPPORTED;
                const size_t x = (size_t)cells * (P.kc * 2 + 16), a = (size_t)P.cpg * (S.kt * P.kc * 2 + 16);
                xb = x > xb ? x : xb; ab = a > ab ? a : ab;
            }
            stage = 2 * (xb + ab) > stage ? 2 * (xb + ab) : stage;
        }
        // (the epilogue's arrays t0, t1, t2, red, gam, bet are laid out one behind the other whatever the mode: reserve all)
        const size_t epi = 3 * (size_t)P.cpg * nmax * 4 + 32 + (size_t)P.cpg * 8;
        const size_t want = stage > epi ? stage : epi;
        lds = want > lds ? want : lds;
        L.p[i] = P;
        L.wg_start[i] = wgs;
        wgs += P.B * (P.M / P.cpg) * P.nrange;
    }
    for (int i = n; i <= OTAL_B1D_MAX_PROB; ++i) L.wg_start[i] = wgs;
    if (lds > 160 * 1024) return OTAL_E_UNSUPPORTED;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(block1d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) { (void)hipGetLastError(); return OTAL_E_UNSUPPORTED; }
        configured = true;
    }
    hipLaunchKernelGGL(block1d_kernel, dim3(wgs), dim3(256), lds, (hipStream_t)stream, L);
    return otal_launch_status();
}

extern "C" int otal_b1d_pack(const void* items, int n_items, int total_blocks, void* stream) {
    if (!items) return OTAL_E_NULL;
    if (n_items <= 0 || total_blocks <= 0) return OTAL_E_SHAPE;
    hipLaunchKernelGGL(b1d_pack_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const PackItem*>(items), n_items);
    return otal_launch_status();
}
