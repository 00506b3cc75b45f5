// The narrow-tile (128 x 256, 256 x 128) instantiations of gemm_bt8_kernel: the kernel template and launch8 come from gemm8.hip,
// compiled here with its launchers switched off, so that the two halves build side by side (see launch8_narrow there).
#define LR_GEMM8_NARROW_TU 1
#include "gemm8.hip"
