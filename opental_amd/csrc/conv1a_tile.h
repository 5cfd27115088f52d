// opental_amd/csrc/conv1a_tile.h -- the launcher of conv1a_tile.hip (its own translation unit), called by conv_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace otal_conv {

struct Conv1aTileArgs {
    const float* x;             // (B, 3, Ti, Hi, 96) fp32, strides x_bs / x_cs in elements
    const float* w;             // (M, 3, 7, 7, 7) fp32
    const unsigned short* wp;   // set by the launcher: the weights packed in MFMA-operand order (workspace)
    void* out;                  // (B, M, To, Ho, 48): fp32, or bf16 when `half`
    const float* scale;         // per output channel (nullable -> 1)
    const float* shift;         // per output channel (nullable -> 0)
    int64_t x_bs, x_cs, y_bs, y_cs;
    int B, Ti, Hi, To, Ho, M;
    int relu, half;
    int flags;                  // ablation builds only
};

// 1 when the geometry is the tiled kernel's (To % 4 == 0, Ho % 4 == 0; everything else is checked by the caller)
__attribute__((visibility("hidden"))) int conv1a_tile_eligible(int To, int Ho);
__attribute__((visibility("hidden"))) int launch_conv1a_tile(const Conv1aTileArgs& a, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace otal_conv
