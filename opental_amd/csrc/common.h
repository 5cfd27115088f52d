// opental_amd/csrc/common.h -- shared device/host helpers for libopental_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "opental_hip.h"

#define OTAL_WAVE 64

struct LevelTab {          // passed by value as a kernel argument (lives in SGPRs)
    int nlev;
    int ts[OTAL_MAX_LEVELS + 1];   // column starts of each level along T
    int ns[OTAL_MAX_LEVELS + 1];   // column starts of each level along N (proposals)
};

typedef unsigned short bf16_t;     // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {          // round to nearest even
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float ld_f32(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ld_f32(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
__device__ __forceinline__ void st_f32(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st_f32(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16(v); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

static inline int otal_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
static inline int ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Named run-time switch (core.hip): first lookup reads the environment variable `name` (absent: dflt; present but not a
// number: 1), later lookups are one load; otal_set_option() changes it.  Never getenv() on a launch path.
int* otal_option_slot(const char* name, int dflt);
#define OTAL_OPT(name, dflt) ([]() -> int { static int* const slot_ = otal_option_slot(name, dflt); return *slot_; }())
