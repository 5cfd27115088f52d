// tests/cpu_conv_index.cpp -- CPU harness: naive GEMM loops over the SAME index functions the
// gfx950 kernels use (opental_amd/csrc/conv_index.h).  Built by tests/test_conv_index_cpu.py with g++.
#include <vector>
#include <cstring>
#include "conv_index.h"

static void fill(ConvGeom& g, const int* d, const int64_t* s) {
    g.B = d[0]; g.Cin = d[1]; g.Cout = d[2]; g.Ti = d[3]; g.Hi = d[4]; g.Wi = d[5];
    g.To = d[6]; g.Ho = d[7]; g.Wo = d[8]; g.kt = d[9]; g.kh = d[10]; g.kw = d[11];
    g.st = d[12]; g.sh = d[13]; g.sw = d[14]; g.pt = d[15]; g.ph = d[16]; g.pw = d[17];
    g.nlev = d[18];
    for (int i = 0; i <= OTAL_CONV_MAX_LEVELS; ++i) g.lev[i] = d[19 + i];
    g.x_bs = s[0]; g.x_cs = s[1]; g.y_bs = s[2]; g.y_cs = s[3];
}

extern "C" void cpu_conv_fwd(const int* d, const int64_t* s, const float* x, const float* w, float* y) {
    ConvGeom g; fill(g, d, s);
    const int K = g.Cin * conv_kvol(g), N = g.B * conv_out_positions(g);
    for (int m = 0; m < g.Cout; ++m)
        for (int n = 0; n < N; ++n) {
            PosDec o = dec_pos(n, g.To, g.Ho, g.Wo);
            double acc = 0;
            for (int k = 0; k < K; ++k) {
                int64_t off;
                if (conv_src_of_output(g, o, dec_tap(g, k), off)) acc += (double)w[(int64_t)m * K + k] * x[off];
            }
            y[conv_out_offset(g, o, m)] = (float)acc;
        }
}

extern "C" void cpu_conv_dgrad(const int* d, const int64_t* s, const float* dy, const float* w, float* dx) {
    ConvGeom g; fill(g, d, s);
    const int kvol = conv_kvol(g), K = g.Cout * kvol, N = g.B * conv_in_positions(g);
    std::vector<float> wt((size_t)g.Cin * K);   // packed W^T (Cin, Cout, kvol)
    for (int co = 0; co < g.Cout; ++co)
        for (int ci = 0; ci < g.Cin; ++ci)
            for (int r = 0; r < kvol; ++r) wt[((size_t)ci * g.Cout + co) * kvol + r] = w[((size_t)co * g.Cin + ci) * kvol + r];
    for (int m = 0; m < g.Cin; ++m)
        for (int n = 0; n < N; ++n) {
            PosDec i = dec_pos(n, g.Ti, g.Hi, g.Wi);
            double acc = 0;
            for (int k = 0; k < K; ++k) {
                int64_t off;
                if (conv_src_of_input(g, i, dec_tap(g, k), off)) acc += (double)wt[(size_t)m * K + k] * dy[off];
            }
            dx[conv_in_offset(g, i, m)] = (float)acc;
        }
}

extern "C" void cpu_conv_wgrad(const int* d, const int64_t* s, const float* x, const float* dy, float* dw) {
    ConvGeom g; fill(g, d, s);
    const int N = g.Cin * conv_kvol(g), K = g.B * conv_out_positions(g);
    for (int m = 0; m < g.Cout; ++m)
        for (int n = 0; n < N; ++n) {
            TapDec t = dec_tap(g, n);
            double acc = 0;
            for (int k = 0; k < K; ++k) {
                PosDec o = dec_pos(k, g.To, g.Ho, g.Wo);
                int64_t off;
                if (conv_src_of_output(g, o, t, off)) acc += (double)dy[conv_out_offset(g, o, m)] * x[off];
            }
            dw[(int64_t)m * N + n] = (float)acc;
        }
}

extern "C" void cpu_same_pad(int size, int k, int s, int* front, int* out) { same_pad(size, k, s, *front, *out); }

// fast division == true division (exhaustive over small divisors, sampled numerators up to 2^32-1)
extern "C" int cpu_fastdiv_check(void) {
    const uint32_t ds[] = {1, 2, 3, 4, 5, 6, 7, 9, 12, 24, 27, 36, 48, 64, 96, 126, 128, 343, 1029, 18432, 65537, 1u << 20, 2147483647u};
    for (uint32_t d : ds) {
        FastDiv f = make_fastdiv(d);
        for (uint64_t n = 0; n < (1ull << 32); n += (n < 100000 ? 1 : 65521)) {
            if (fd_div(f, (uint32_t)n) != (uint32_t)n / d) return (int)d;
        }
        const uint32_t edge[] = {0xffffffffu, 0xfffffffeu, 0x80000000u, 0x7fffffffu, d - 1, d, d + 1, 2 * d - 1, 2 * d};
        for (uint32_t n : edge) if (fd_div(f, n) != n / d) return -(int)d;
    }
    return 0;
}
extern "C" int cpu_dec_fd_check(const int* d, const int64_t* s) {
    ConvGeom g; fill(g, d, s);
    ConvFastDiv f = make_conv_fastdiv(g);
    const int K = g.Cin * conv_kvol(g), N = g.B * conv_out_positions(g), Ni = g.B * conv_in_positions(g);
    for (int k = 0; k < K; ++k) { TapDec a = dec_tap(g, k), b = dec_tap_fd(f, k); if (a.c != b.c || a.dt != b.dt || a.dh != b.dh || a.dw != b.dw) return 1; }
    for (int n = 0; n < N; ++n) { PosDec a = dec_pos(n, g.To, g.Ho, g.Wo), b = dec_pos_fd(n, f.To, f.Ho, f.Wo); if (a.b != b.b || a.t != b.t || a.h != b.h || a.w != b.w) return 2; }
    for (int n = 0; n < Ni; ++n) { PosDec a = dec_pos(n, g.Ti, g.Hi, g.Wi), b = dec_pos_fd(n, f.Ti, f.Hi, f.Wi); if (a.b != b.b || a.t != b.t || a.h != b.h || a.w != b.w) return 3; }
    return 0;
}
