// Instantiations of the 3x3 halo-tile convolution (kernel in conv_igemm_kernel.h).
#include "conv_igemm_kernel.h"

int dyk_conv_launch_halob(const DykConvDesc* d, hipStream_t s, int th);
int dyk_conv_launch_halo(const DykConvDesc* d, hipStream_t s, int th) {
    if (d->dtype != DYK_BF16) return DYK_ERR_UNSUPPORTED;
    if (d->flags & DYK_EPI_BNBWD) return dyk_conv_launch_halob(d, s, th);
    if (th == 4) return dispatch_conv_halo<bf16_t, 4>(d, s);
    if (th == 8) return dispatch_conv_halo<bf16_t, 8>(d, s);
    return DYK_ERR_UNSUPPORTED;
}
