// 16-bit storage (bf16 / f16) instantiations of the implicit GEMM: the kernel body and launchers of kpf_conv.hip compiled once more
// with KPF_CONV_H16, which selects kpf_conv2d_h16 and the ARITH_BF16 / ARITH_F16 instantiations (a separate translation unit so
// that the two compile in parallel).
#define KPF_CONV_H16 1
#include "kpf_conv.hip"
