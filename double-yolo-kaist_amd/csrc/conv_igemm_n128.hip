// Instantiations of the implicit-GEMM convolution for the 128-pixel tile (one translation unit per tile width, element
// type and epilogue family so that make -j builds them in parallel).
#include "conv_igemm_kernel.h"

int dyk_conv_launch_n128b(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_n128f(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_n128n(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_n128(const DykConvDesc* d, hipStream_t s) {
    if (d->flags & DYK_EPI_BNFWD) return dyk_conv_launch_n128n(d, s);
    if (d->dtype == DYK_F32) return dyk_conv_launch_n128f(d, s);
    if (d->flags & DYK_EPI_BNBWD) return dyk_conv_launch_n128b(d, s);
    if (d->dtype == DYK_BF16) return dispatch_conv_bn<bf16_t, 128>(d, s);
    return DYK_ERR_UNSUPPORTED;
}
