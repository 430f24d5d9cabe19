// Instantiations of the implicit-GEMM convolution for the 80-pixel tile (one translation unit per tile width and
// epilogue family so that make -j builds them in parallel).
#include "conv_igemm_kernel.h"

int dyk_conv_launch_n80b(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_n80n(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_n80(const DykConvDesc* d, hipStream_t s) {
    if (d->flags & DYK_EPI_BNFWD) return dyk_conv_launch_n80n(d, s);
    if (d->flags & DYK_EPI_BNBWD) return dyk_conv_launch_n80b(d, s);
    if (d->dtype == DYK_BF16) return dispatch_conv_bn<bf16_t, 80>(d, s);
    return DYK_ERR_UNSUPPORTED;
}
