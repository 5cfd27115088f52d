// Hardware facts the conv gather design depends on (gfx950):
//  (1) raw-buffer bounds check: per dword or per access?  is soffset part of the check?
//  (2) dword-aligned (not 16-byte aligned) buffer_load_dwordx4
//  (3) L1/TA throughput of per-lane-scattered dwordx4 loads vs coalesced dword loads
// build: hipcc --offload-arch=gfx950 -O3 -o bufcheck bufcheck.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void semantics(const unsigned* buf, unsigned nrec, unsigned* out) {
    auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf) + 64, 0, (int)nrec, 0x00020000);   // descriptor starts 256 B into the allocation
    const int l = threadIdx.x;
    u32x4 v;
    unsigned s;
    v = __builtin_amdgcn_raw_buffer_load_b128(r, 0xfffffffcu, 0, 0);  if (l == 0) { out[0]=v[0]; out[1]=v[1]; out[2]=v[2]; out[3]=v[3]; }   // starts 4 B before
    v = __builtin_amdgcn_raw_buffer_load_b128(r, nrec - 8, 0, 0);     if (l == 0) { out[4]=v[0]; out[5]=v[1]; out[6]=v[2]; out[7]=v[3]; }   // last 2 dwords beyond
    v = __builtin_amdgcn_raw_buffer_load_b128(r, 4, 0, 0);            if (l == 0) { out[8]=v[0]; out[9]=v[1]; out[10]=v[2]; out[11]=v[3]; } // dword aligned
    v = __builtin_amdgcn_raw_buffer_load_b128(r, 12, 0, 0);           if (l == 0) { out[12]=v[0]; out[13]=v[1]; out[14]=v[2]; out[15]=v[3]; }
    s = __builtin_amdgcn_raw_buffer_load_b32(r, 0, nrec, 0);          if (l == 0) out[16] = s;        // OOB only through soffset
    s = __builtin_amdgcn_raw_buffer_load_b32(r, nrec, 0, 0);          if (l == 0) out[17] = s;        // OOB through voffset
    s = __builtin_amdgcn_raw_buffer_load_b32(r, nrec - 4, 4, 0);      if (l == 0) out[18] = s;        // voffset in range, +soffset beyond
    s = __builtin_amdgcn_raw_buffer_load_b32(r, 8, 16, 0);            if (l == 0) out[19] = s;        // plain: dword (8+16)/4 = 6
}

// throughput: every wave issues `iters` loads; pattern selects the per-lane address map inside a window that stays L2/L1 resident
template <int PAT>
__global__ __launch_bounds__(256) void tput(const unsigned* buf, unsigned bytes, int iters, unsigned* sink) {
    auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, (int)bytes, 0x00020000);
    const int l = threadIdx.x & 63, w = (blockIdx.x * 4 + (threadIdx.x >> 6));
    unsigned acc = 0;
    unsigned base = (unsigned)(w * 8192) % (bytes / 2);
    for (int i = 0; i < iters; ++i) {
        const unsigned step = (unsigned)(i & 63);
        if (PAT == 0) {            // coalesced dword: 256 B per wave-instruction
            acc += __builtin_amdgcn_raw_buffer_load_b32(r, base + l * 4, step * 256, 0);
        } else if (PAT == 1) {     // coalesced dwordx4: 1 KB per wave-instruction
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, base + l * 16, step * 1024, 0); acc += v[0] + v[1] + v[2] + v[3];
        } else if (PAT == 2) {     // per-lane rows 2304 B apart (one channel plane row each), 16 B per lane, walking along the row
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, base + l * 2304, step * 16, 0); acc += v[0] + v[1] + v[2] + v[3];
        } else if (PAT == 3) {     // tap-like: groups of 3 lanes 4 B apart (dw), 9 groups 96 B apart (dh/dt rows), ~2.4 channels per wave
            const unsigned tap = l % 27, ci = l / 27;
            const unsigned off = ci * 294912u % 65536u + (tap / 9) * 2304 + ((tap / 3) % 3) * 96 + (tap % 3) * 4;
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, base + off, step * 32, 0); acc += v[0] + v[1] + v[2] + v[3];
        } else if (PAT == 4) {     // per-lane 32 B apart dwordx4 (half-overlapping sectors)
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, base + l * 32, step * 2048 % 65536, 0); acc += v[0] + v[1] + v[2] + v[3];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PAT>
void run(const char* name, const unsigned* d, unsigned bytes, unsigned* sink, double bytes_per_instr) {
    const int blocks = 256 * 8, iters = 2048;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(tput<PAT>, dim3(blocks), dim3(256), 0, 0, d, bytes, iters, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL(tput<PAT>, dim3(blocks), dim3(256), 0, 0, d, bytes, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 4 * iters;
    const double cyc_per_instr_cu = ms * 1e-3 * 2.4e9 * 256 / instr;
    printf("%-34s %.3f ms  %.1f cycles/wave-instr/CU  %.1f B/clk/CU  %.2f TB/s\n", name, ms, cyc_per_instr_cu,
           bytes_per_instr / cyc_per_instr_cu, instr * bytes_per_instr / (ms * 1e-3) / 1e12);
}

int main() {
    const unsigned N = 1 << 22;   // 16 MB
    std::vector<unsigned> h(N);
    for (unsigned i = 0; i < N; ++i) h[i] = 0xA0000000u + i;
    unsigned *d, *o;
    hipMalloc(&d, N * 4); hipMalloc(&o, 256);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(o, 0xEE, 256);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d, 256u, o);
    unsigned r[20]; hipMemcpy(r, o, 80, hipMemcpyDeviceToHost);
    printf("descriptor covers dwords [64,128) of the allocation, values 0xA0000040..0xA000007F\n");
    printf("b128 @-4      : %08x %08x %08x %08x\n", r[0], r[1], r[2], r[3]);
    printf("b128 @nrec-8  : %08x %08x %08x %08x\n", r[4], r[5], r[6], r[7]);
    printf("b128 @4       : %08x %08x %08x %08x\n", r[8], r[9], r[10], r[11]);
    printf("b128 @12      : %08x %08x %08x %08x\n", r[12], r[13], r[14], r[15]);
    printf("b32 v=0 s=nrec: %08x   b32 v=nrec s=0: %08x   b32 v=nrec-4 s=4: %08x   b32 v=8 s=16: %08x\n", r[16], r[17], r[18], r[19]);
    run<0>("coalesced dword", d, N * 4, o, 256);
    run<1>("coalesced dwordx4", d, N * 4, o, 1024);
    run<2>("scattered rows dwordx4", d, N * 4, o, 1024);
    run<3>("tap-like dwordx4", d, N * 4, o, 1024);
    run<4>("32B-stride dwordx4", d, N * 4, o, 1024);
    return 0;
}
