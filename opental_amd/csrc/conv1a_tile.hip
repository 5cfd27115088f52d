// opental_amd/csrc/conv1a_tile.hip -- Conv3d_1a_7x7 forward (7x7x7 taps, stride 2, THREE input channels, 96-wide planes;
// reference AFSD/common/i3d_backbone.py:194-199, Unit3D :7-87), the 4 x 4 x 48 tile kernel.
//
// conv1a_direct_fwd_kernel (conv_gemm.hip) gave a workgroup a 2 (t) x 2 (h) x 48 (w) block of outputs: 12 288 workgroups
// at b = 8, each staging 9 planes x 9 rows for 2 x 2 new ones (20x halo), two per CU, their phases (staging, 49 K steps,
// epilogue) back to back -- tools/ablate_1a.sh: every phase switched off still left 0.24 of 0.74 ms (workgroup launches
// and fixed per-workgroup work), the MFMA + LDS-read loop was 0.25, and the parts added up instead of overlapping.
//
// Here a workgroup of EIGHT waves owns 4 (t) x 4 (h) x 48 (w) = 768 output positions x 64 channels at a time:
//   * 3 072 tiles at b = 8, walked by PERSISTENT workgroups (one per CU: 135 KB of LDS; contiguous tile ranges), 13 planes x
//     13 rows staged for 4 x 4 new ones (10.6 input row-planes per output row instead of 20);
//   * the patch is the same channel-last bf16 {c0, c1, c2, 0} pixel layout -- a K step is one (dt, dh) kernel row, its
//     8 dw x 4 ci are 64 contiguous bytes from pixel 2 wo on, every patch operand one aligned ds_read_b128 -- but only the
//     planes the first kernel plane (dt = 0) reads are staged before the K loop starts (4 of 13; for every tile but a
//     workgroup's first their loads run under the previous tile's epilogue); the other nine are loaded while the loop runs
//     (plane p is first read at dt = 1 for odd p < 8, at dt = p - 6 for p >= 8), their global loads four K steps in front
//     of the LDS stores;
//   * the weights do not go through LDS: packed in MFMA-operand order they are one 16-byte load per lane and operand, two K
//     steps ahead in registers -- the K loop has three workgroup barriers (behind the staged planes), not 49;
//   * a wave computes 64 channels x 96 positions (2 x 3 MFMA tiles): 6 LDS operand reads per 12 MFMAs instead of 6 per 4;
//   * bf16 output: after the K loop each wave transposes its 64 channels x 96 positions through its own 13 KB of the dead
//     patch (no per-round waits) and stores 16-byte pieces, 192 contiguous bytes per channel row.
// The accumulation order of an output is that of conv1a_direct_fwd_kernel (K steps in (dt, dh) order, two MFMAs per step),
// so the two kernels agree bit for bit.  Measured (b = 8): 0.64 -> 0.41 ms per op, matrix pipe 33 -> 58 % busy.
#include "common.h"
#include "conv1a_tile.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
struct Words4 { unsigned a, b, c, d; };
struct Words2 { unsigned a, b; };
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {     // v_cvt_pk_bf16_f32: RNE, lo in bits 0..15
    const hw_f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, hw_bf16x2));
}

// compile-time loop: the K steps index register arrays and carry a static staging schedule (a `#pragma unroll` of this body
// is refused by the optimizer)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int TT = 4, TR = 4, WO = 48;                      // output tile: planes x rows x columns
constexpr int NPL = 2 * TT + 5, NR = 2 * TR + 5;            // input planes / rows under it
constexpr int NC = 102, PITCH = NC * 8, PLANE = NR * PITCH; // pixels per patch row (w = -2 .. 99), bytes
constexpr int NT = 512, NWAVE = NT / 64, WPOS = TT * TR * WO / NWAVE;      // 96 positions per wave
constexpr int BM = 64, WM = 2, WN = WPOS / 32;
constexpr int STEPS = 49;
constexpr int EPI_PITCH = WPOS * 2 + 16;                  // bytes per channel row of a wave's transposition tile (in the dead patch)
constexpr int QPR = 24, IPP = NR * QPR;                     // 4-pixel quads per row; staging items per plane
static_assert(TT * TR * WO == NWAVE * WPOS && WPOS % 32 == 0 && (TR * WO) % WPOS == 0, "a wave's positions lie in one output plane");
static_assert(NPL * PLANE + 2 * BM * 4 <= 160 * 1024 && NWAVE * BM * EPI_PITCH <= NPL * PLANE, "LDS");

struct TileAt {                 // one work item: a 4 x 4 x 48 output tile of one sample and one 64-channel block
    int b, to0, ho0, mblk;
};

__global__ __launch_bounds__(NT) void conv1a_tile_fwd_kernel(const otal_conv::Conv1aTileArgs a, int nwork, int per_wg, int tm) {
    __shared__ __attribute__((aligned(16))) unsigned char patch[NPL * PLANE];
    __shared__ __attribute__((aligned(16))) float rows[2 * BM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_h = a.Ho / TR, tiles_t = a.To / TT;
    // PERSISTENT workgroups (one per CU: 135 KB of LDS), each with a contiguous range of tiles -- neighbours in (h, t), so
    // a workgroup's consecutive tiles share halo rows / planes in its XCD's L2.  Between two tiles nothing drains: the
    // output stores of tile i retire while tile i + 1 is staged and multiplied, and the loads of the first four planes of
    // tile i + 1 are issued BEFORE the epilogue of tile i (one workgroup per tile, first version: the store acknowledgements
    // and the first staging round trip were exposed twelve times per CU, ~0.1 of 0.45 ms).
    // (workgroups are dealt round-robin to the 8 XCDs, each with its own L2: XCD x gets a CONTIGUOUS run of tile ranges, so the
    //  t-neighbours that share 9 of their 13 input planes run on one XCD at the same time)
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (wg & 7) * (gridDim.x >> 3) + (wg >> 3);
    const int w_begin = wg * per_wg, w_end = min(nwork, w_begin + per_wg);
    if (w_begin >= w_end) return;
    auto tile_at = [&](int w) {
        TileAt t;
        t.mblk = w % tm; w /= tm;
        const int th = w % tiles_h; w /= tiles_h;
        t.to0 = (w % tiles_t) * TT;
        t.b = w / tiles_t;
        t.ho0 = th * TR;
        return t;
    };
    const unsigned cs_bytes = (unsigned)(a.x_cs * 4);
    const int x_extent = (int)((2 * a.x_cs + (int64_t)a.Ti * a.Hi * 96) * 4);

    // ---- staging item e of a plane group {pl0, pl0 + plstep, ...} of tile t: (plane, row, quad of 4 pixels) x 3 channels.
    // One sample's three channels sit behind a buffer descriptor: a plane / row outside the input reads zeros through the
    // bounds check (offset 0xffffffff), no branch around the load.
    auto item_issue = [&](const TileAt& t, int e, int pl0, int plstep, int nitems, f32x4 (&v)[3], int& off) {
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (int64_t)t.b * a.x_bs), 0, x_extent, 0x00020000);
        const bool live = e < nitems;
        const int gi = e / IPP, rem = e - gi * IPP, rr = rem / QPR, q = rem - rr * QPR;
        const int pl = pl0 + plstep * gi;
        const int ti = 2 * t.to0 - 2 + pl, hi = 2 * t.ho0 - 2 + rr;     // front pad 2
        const bool ok = live && (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi;
#ifdef OTAL_DIRECT_ABLATE
        const bool ld = ok && !(a.flags & 4);
#else
        const bool ld = ok;
#endif
        off = live ? pl * PLANE + rr * PITCH + (4 * q + 2) * 8 : -1;
        const unsigned vo = ld ? (unsigned)((((int64_t)ti * a.Hi + hi) * 96 + 4 * q) * 4) : 0xffffffffu;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
            v[ci] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, ci * cs_bytes, 0));
    };
    auto item_commit = [&](const f32x4 (&v)[3], int off) {
        if (off < 0) return;
        u32x4 lo, hi;
        lo[0] = cvt_pk_bf16(v[0][0], v[1][0]); lo[1] = cvt_pk_bf16(v[2][0], 0.f);
        lo[2] = cvt_pk_bf16(v[0][1], v[1][1]); lo[3] = cvt_pk_bf16(v[2][1], 0.f);
        hi[0] = cvt_pk_bf16(v[0][2], v[1][2]); hi[1] = cvt_pk_bf16(v[2][2], 0.f);
        hi[2] = cvt_pk_bf16(v[0][3], v[1][3]); hi[3] = cvt_pk_bf16(v[2][3], 0.f);
        *reinterpret_cast<u32x4*>(patch + off) = lo;
        *reinterpret_cast<u32x4*>(patch + off + 16) = hi;
    };

    // ---- weights: NO LDS.  pack_conv1a_operand_order_kernel lays them out in MFMA-operand order -- [64-row block][K step][kk][i]
    // [lane][8 bf16] -- so an A operand is ONE 16-byte load per lane, 1 KB contiguous per wave, the same for all eight
    // waves (L1 hits).  A FIFO of WD K steps (4 operands each) in registers; with no weight ring in LDS the K loop needs
    // no per-step barrier and the waves drift apart instead of draining the MFMA pipe 49 times in lockstep.
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.wp), 0, (int)((int64_t)tm * BM * STEPS * 64), 0x00020000);
    constexpr int WD = 2;

    // this lane's output positions (one per 32-column MFMA tile j) and the byte offset of pixel 2 wo in patch row (dt = 0, dh = 0)
    const int lt = wave >> 1;                               // output plane of the tile: a wave's 96 positions lie in one
    int xbase[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int o = (wave & 1) * WPOS + j * 32 + (lane & 31);
        const int lr = o / WO, wo = o - lr * WO;
        xbase[j] = ((2 * lt) * NR + 2 * lr) * PITCH + (2 * wo) * 8 + (lane >> 5) * 16;
    }

    // planes 0, 2, 4, 6 (read by dt = 0) of the first tile: three items per thread
    f32x4 pv[3][3];
    int poff[3];
    TileAt cur = tile_at(w_begin);
#pragma unroll
    for (int u = 0; u < 3; ++u) item_issue(cur, tid + NT * u, 0, 2, 4 * IPP, pv[u], poff[u]);

    // ---- start of a tile: scale / shift rows, the zero columns of every patch row (the last epilogue wrote over them) and the
    // stores of planes 0, 2, 4, 6, whose loads were issued under the previous tile's epilogue
    auto begin_tile = [&](const TileAt& t, int tid_l) {
        if (tid_l < BM) {
            const int m = t.mblk * BM + tid_l;
            rows[2 * tid_l] = (m < a.M && a.scale) ? a.scale[m] : 1.f;
            rows[2 * tid_l + 1] = (m < a.M && a.shift) ? a.shift[m] : 0.f;
        }
        for (int i = tid_l; i < NPL * NR * 6; i += NT) {    // pixels 0, 1 (w = -2, -1) and 98 .. 101 (w = 96 .. 99)
            const int row = i / 6, e = i - row * 6;
            *reinterpret_cast<Words2*>(patch + row * PITCH + (e < 2 ? e : 96 + e) * 8) = Words2{0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) item_commit(pv[u], poff[u]);
    };
    begin_tile(cur, tid);

    for (int work = w_begin; work < w_end; ++work) {
        const int m0 = cur.mblk * BM;
        // (per-lane addresses of the staging items and of the epilogue are re-derived per tile from an opaque copy of the
        //  thread index: hoisted out of this loop they are ~100 live registers across the K loop, i.e. spills)
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));
        const int lane_l = tid_l & 63;
#ifdef OTAL_DIRECT_ABLATE
        const unsigned wvo = (a.flags & 128) ? 0xffffffffu : (unsigned)(cur.mblk * (BM * STEPS * 64) + lane * 16);
#else
        const unsigned wvo = (unsigned)(cur.mblk * (BM * STEPS * 64) + lane * 16);
#endif
        bf16x8 aw[WD][2][WM];
        auto load_w = [&](int s) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    aw[s % WD][kk][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvo, ((s * 2 + kk) * WM + i) * 1024, 0));
        };
#pragma unroll
        for (int s = 0; s < WD; ++s) load_w(s);
        // planes 1, 3, 5, 7 (first read at dt = 1) and 8 .. 12 travel under the K loop: two items per thread in flight
        f32x4 sva[3], svb[3];
        int soffa, soffb;
        item_issue(cur, tid_l, 1, 2, 4 * IPP, sva, soffa);
        item_issue(cur, tid_l + NT, 1, 2, 4 * IPP, svb, soffb);
        __syncthreads();

        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // Software pipeline: the patch operands of step s+1 are read from LDS while the MFMAs of step s run.
        bf16x8 bv[2][2][WN];
        auto read_ops = [&](int set, int s) {
            const int dt = s / 7, dh = s - dt * 7;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    bv[set][kk][j] = *reinterpret_cast<const bf16x8*>(patch + xbase[j] + (dt * NR + dh) * PITCH + kk * 32);
        };
        read_ops(0, 0);
        static_for<0, STEPS>([&](auto step) {   // `set`, the FIFO slot and the staging schedule are compile-time
            constexpr int s = decltype(step)::value;
            constexpr int set = s & 1;
#ifdef OTAL_DIRECT_ABLATE
            if (s + 1 < STEPS && !(a.flags & 256)) read_ops(set ^ 1, s + 1);
            if (!(a.flags & 512))
#else
            if (s + 1 < STEPS) read_ops(set ^ 1, s + 1);
#endif
            __builtin_amdgcn_sched_barrier(0);  // the LDS reads of the next step first: their latency runs under these MFMAs
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[s % WD][kk][i], bv[set][kk][j], acc[i][j], 0, 0, 0);
            if (s + WD < STEPS) load_w(s + WD);
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks these loads to just in front of their use)
            // The planes read from dt = 1 on: 1, 3, 5, 7 (first read during step 6, for step 7), then 8 .. 12 (plane 8 + k
            // first read during step 13 + 7 k).  An item's loads are FOUR steps in front of its LDS stores (HBM / L2; vmcnt
            // retires in order, so a late one also holds up the weight loads queued behind it).  A workgroup barrier after
            // the stores of steps 5, 9 and 13: the only three of the K loop.
            if (s == 1) { item_commit(sva, soffa); item_issue(cur, tid_l + 2 * NT, 1, 2, 4 * IPP, sva, soffa); }
            if (s == 3) { item_commit(svb, soffb); item_issue(cur, tid_l, 8, 1, 5 * IPP, svb, soffb); }
            if (s == 5) { item_commit(sva, soffa); item_issue(cur, tid_l + NT, 8, 1, 5 * IPP, sva, soffa); }
            if (s == 7) { item_commit(svb, soffb); item_issue(cur, tid_l + 2 * NT, 8, 1, 5 * IPP, svb, soffb); }
            if (s == 9) { item_commit(sva, soffa); item_issue(cur, tid_l + 3 * NT, 8, 1, 5 * IPP, sva, soffa); }
            if (s == 11) item_commit(svb, soffb);
            if (s == 13) item_commit(sva, soffa);
            if (s == 5 || s == 9 || s == 13) {
#ifdef OTAL_DIRECT_ABLATE
                if (!(a.flags & 16))
#endif
                __syncthreads();
            }
        });
        const TileAt done = cur;
        // the first four planes of the NEXT tile: issued as soon as the accumulators have left their registers, in flight
        // under the rest of this tile's epilogue
        auto prefetch_next = [&]() {    // (unconditional: behind the last tile it re-reads that tile's planes and drops them)
            cur = tile_at(min(work + 1, w_end - 1));
#pragma unroll
            for (int u = 0; u < 3; ++u) item_issue(cur, tid_l + NT * u, 0, 2, 4 * IPP, pv[u], poff[u]);
        };
        __syncthreads();        // every wave has left the K loop: the patch is dead
#ifdef OTAL_DIRECT_ABLATE
        if (a.flags & 64) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[i][j][r];
            if (t == 1.2345678e30f) reinterpret_cast<float*>(a.out)[0] = t;
            prefetch_next();
            __syncthreads();
            if (work + 1 < w_end) begin_tile(cur, tid_l);
            continue;
        }
#endif
        // ---- epilogue.  The wave's 96 positions are CONTIGUOUS in the output: rows ho0 .. ho0+3 of plane to0 + lt are
        // whole 48-wide rows, the wave holds the first or second 96 of those 192 elements.
        const int64_t pbase = (int64_t)done.b * a.y_bs + ((int64_t)(done.to0 + lt) * a.Ho + done.ho0) * WO + (wave & 1) * WPOS;
        const bool relu = a.relu != 0;
        if (a.half) {
            // bf16 output through ONE transposition per wave: each wave owns 64 channel rows x 96 positions of the dead
            // patch.  All 96 values of a lane go to LDS, then 12 sixteen-byte pieces per lane come back and leave as 192
            // contiguous bytes per channel row.
            unsigned short* yh = reinterpret_cast<unsigned short*>(a.out) + pbase;
            unsigned char* tile = patch + wave * (BM * EPI_PITCH);
            f32x4 ss[WM][4][2];                             // {scale, shift} x 4 rows of this lane's half wave, per (i, q)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        ss[i][q][u] = *reinterpret_cast<const f32x4*>(rows + 2 * (i * 32 + 8 * q + 4 * (lane_l >> 5)) + 4 * u);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float sc = ss[i][q][rr >> 1][2 * (rr & 1)], sh = ss[i][q][rr >> 1][2 * (rr & 1) + 1];
#pragma unroll
                        for (int j = 0; j < WN; ++j) {
                            float v = acc[i][j][4 * q + rr] * sc + sh;
                            if (relu) v = fmaxf(v, 0.f);
                            *reinterpret_cast<unsigned short*>(tile + (i * 32 + 8 * q + 4 * (lane_l >> 5) + rr) * EPI_PITCH + (j * 32 + (lane_l & 31)) * 2) =
                                (unsigned short)(cvt_pk_bf16(v, 0.f) & 0xffffu);
                        }
                    }
            __builtin_amdgcn_sched_barrier(0);
            prefetch_next();
            __builtin_amdgcn_sched_barrier(0);              // (LDS executes a wave's instructions in order: no wait needed)
#pragma unroll
            for (int u = 0; u < BM * 12 / 64; ++u) {        // 64 rows x 12 sixteen-byte pieces
                const int p = lane_l + 64 * u;
                const int row = p / 12, c = p - row * 12;
                if (m0 + row < a.M) {
                    const Words4 v = *reinterpret_cast<const Words4*>(tile + row * EPI_PITCH + c * 16);
                    *reinterpret_cast<Words4*>(yh + (int64_t)(m0 + row) * a.y_cs + c * 8) = v;
                }
            }
        } else {
            float* yf = reinterpret_cast<float*>(a.out) + pbase;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane_l >> 5);
                    const float sc = rows[2 * lr], sh = rows[2 * lr + 1];
                    if (m0 + lr >= a.M) continue;
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        float v = acc[i][j][r] * sc + sh;
                        if (relu) v = fmaxf(v, 0.f);
                        yf[(int64_t)(m0 + lr) * a.y_cs + j * 32 + (lane_l & 31)] = v;
                    }
                }
            prefetch_next();
        }
        __syncthreads();        // the transposition tiles and `rows` have been read: the next tile may overwrite them
        if (work + 1 < w_end) begin_tile(cur, tid_l);
    }
}

// weights (M, 3, 7, 7, 7) fp32 -> bf16 in MFMA-operand order: [64-row block][s = dt * 7 + dh][kk][i][lane = h * 32 + n][e],
// row = 64 block + 32 i + n, k = 16 kk + 8 h + e = 4 dw + ci; zero where ci = 3, dw = 7 or row >= M
__global__ __launch_bounds__(256) void pack_conv1a_operand_order_kernel(unsigned* __restrict__ wp, const float* __restrict__ w, int M, int pairs) {
    for (int p = blockIdx.x * 256 + threadIdx.x; p < pairs; p += gridDim.x * 256) {
        const int e2 = p & 3, ln = (p >> 2) & 63, i = (p >> 8) & 1, kk = (p >> 9) & 1, rest = p >> 10;
        const int s = rest % STEPS, blk = rest / STEPS;
        const int m = blk * BM + i * 32 + (ln & 31);
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = kk * 16 + (ln >> 5) * 8 + e2 * 2 + u, dw = k >> 2, ci = k & 3;
            v[u] = (m < M && dw < 7 && ci < 3) ? w[((int64_t)m * 3 + ci) * 343 + s * 7 + dw] : 0.f;
        }
        wp[p] = cvt_pk_bf16(v[0], v[1]);
    }
}

}  // namespace

int otal_conv::conv1a_tile_eligible(int To, int Ho) {
    return To % TT == 0 && Ho % TR == 0 && !OTAL_OPT("OTAL_CONV_1A_NOTILE", 0);
}

int otal_conv::launch_conv1a_tile(const Conv1aTileArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    if ((2 * a.x_cs + (int64_t)a.Ti * a.Hi * 96) * 4 >= (1LL << 31)) return OTAL_E_UNSUPPORTED;
    const int tm = (a.M + BM - 1) / BM;
    const size_t wb = (size_t)tm * BM * STEPS * 64;
    if (!ws || ws_bytes < wb) return OTAL_E_UNSUPPORTED;
    const int pairs = tm * BM * STEPS * 16;
    hipLaunchKernelGGL(pack_conv1a_operand_order_kernel, dim3((pairs + 255) / 256), dim3(256), 0, st, reinterpret_cast<unsigned*>(ws), a.w, a.M, pairs);
    if (int e = otal_launch_status()) return e;
    Conv1aTileArgs t = a;
    t.wp = reinterpret_cast<const unsigned short*>(ws);
    // persistent workgroups, one per CU; every workgroup gets the same number of tiles where that is possible
    const int nwork = a.B * (a.To / TT) * (a.Ho / TR) * tm;
    int ncu = OTAL_OPT("OTAL_CONV_1A_WGS", 0);
    if (ncu <= 0) {
        static int cus = 0;                 // compute units of the current device (256 on MI355X), asked once
        if (!cus) {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            cus = n;
        }
        ncu = cus;
    }
    const int per_wg = (nwork + ncu - 1) / ncu;
    const int nwg = (nwork + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(conv1a_tile_fwd_kernel, dim3(nwg), dim3(NT), 0, st, t, nwork, per_wg, tm);
    return otal_launch_status();
}
