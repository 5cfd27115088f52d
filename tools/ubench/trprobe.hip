// Probe of ds_read_b64_tr_b16 (gfx950): which element lands in which lane.
// LDS holds e[i] = i (16-bit).  Lane l supplies byte address addr[l]; we print the 4 values each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int a = addr[threadIdx.x];
    auto p = reinterpret_cast<__attribute__((address_space(3))) v4s*>(
        (__attribute__((address_space(3))) char*)lds + a);
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int h[64]; short o[256];
    int *d; short* od;
    hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
    // case 1: lane i of each 16-lane group -> row i/4 (pitch 64 B), chunk i%4 (8 B); groups offset by 1024 B
    for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; h[l] = g * 1024 + (i / 4) * 64 + (i % 4) * 8; }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, od);
    hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
    printf("case1: row pitch 32 elements (64 B); lane: values (element index = row*32 + col within group base g*512)\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h[l], o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    // case 2: scattered rows: row r of group at arbitrary cells
    int cells[4] = {3, 17, 40, 9};
    for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; h[l] = cells[i / 4] * 64 + g * 2048 + (i % 4) * 8; }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, od);
    hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
    printf("case2: scattered rows (cells 3,17,40,9; 64 B each)\n");
    for (int l = 0; l < 20; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h[l], o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    return 0;
}
