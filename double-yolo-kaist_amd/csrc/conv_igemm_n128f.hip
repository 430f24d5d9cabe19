// 128-pixel tile, fp32 (exact-fp32 MFMA: tight parity and the fp32 config), both epilogue families
#include "conv_igemm_kernel.h"

int dyk_conv_launch_n128f(const DykConvDesc* d, hipStream_t s) {
    if (d->dtype != DYK_F32) return DYK_ERR_UNSUPPORTED;
    if (d->flags & DYK_EPI_BNBWD) return dispatch_conv_bn<float, 128, 1>(d, s);
    return dispatch_conv_bn<float, 128>(d, s);
}
