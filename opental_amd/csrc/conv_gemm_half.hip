// opental_amd/csrc/conv_gemm_half.hip -- part 1 of conv_gemm.hip: the instantiations of its kernel templates for bf16-STORED
// activations and gradients (template flag H) and their dispatcher otal_conv::launch_half().  A translation unit of its own so
// that the two halves of the convolution family compile side by side (csrc/build.py); there is no code in this file.
#define OTAL_CONV_PART 1
#include "conv_gemm.hip"
