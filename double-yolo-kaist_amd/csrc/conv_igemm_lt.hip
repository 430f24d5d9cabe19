// Instantiations of the large-tile 3x3 convolution (conv_lt_kernel.h): forward / plain data-gradient epilogues.
#include "conv_lt_kernel.h"

int dyk_conv_launch_ltb(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_lt(const DykConvDesc* d, hipStream_t s) {
    if (d->dtype != DYK_BF16) return DYK_ERR_UNSUPPORTED;
    if (d->flags & DYK_EPI_BNBWD) return dyk_conv_launch_ltb(d, s);
    return dispatch_conv_lt<0>(d, s);
}
