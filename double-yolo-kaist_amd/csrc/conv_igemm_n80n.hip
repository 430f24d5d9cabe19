// 80-pixel tile, bf16, conv + train-mode BatchNorm + activation in one launch (DYK_EPI_BNFWD, epilogue family 3)
#include "conv_igemm_kernel.h"

int dyk_conv_launch_n80n(const DykConvDesc* d, hipStream_t s) {
    if (d->dtype == DYK_BF16) return dispatch_conv_bn<bf16_t, 80, 3>(d, s);
    return DYK_ERR_UNSUPPORTED;
}
