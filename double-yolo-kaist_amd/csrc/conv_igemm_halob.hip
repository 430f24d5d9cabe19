// 3x3 halo-tile convolution, fused BatchNorm-backward epilogues (DYK_EPI_BNBWD)
#include "conv_igemm_kernel.h"

int dyk_conv_launch_halob(const DykConvDesc* d, hipStream_t s, int th) {
    if (d->dtype != DYK_BF16) return DYK_ERR_UNSUPPORTED;
    if (th == 4) return dispatch_conv_halo<bf16_t, 4, 1>(d, s);
    if (th == 8) return dispatch_conv_halo<bf16_t, 8, 1>(d, s);
    return DYK_ERR_UNSUPPORTED;
}
