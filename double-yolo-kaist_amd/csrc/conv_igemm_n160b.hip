// 160-pixel tile, fused BatchNorm-backward epilogues (DYK_EPI_BNBWD)
#include "conv_igemm_kernel.h"

int dyk_conv_launch_n160b(const DykConvDesc* d, hipStream_t s) {
    if (d->dtype == DYK_BF16) return dispatch_conv_bn<bf16_t, 160, 1>(d, s);
    return DYK_ERR_UNSUPPORTED;
}
