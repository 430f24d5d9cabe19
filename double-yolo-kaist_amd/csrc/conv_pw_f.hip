// Instantiations of the persistent pointwise convolution (conv_pw_kernel.h): plain store and BatchNorm-statistics epilogues.
#include "conv_pw_kernel.h"

int dyk_conv_launch_pwb(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_pwc(const DykConvDesc* d, hipStream_t s);
// pixel-tile code 7 of the conv tune word; DYK_ERR_UNSUPPORTED = the caller falls back to the generic tiles
int dyk_conv_launch_pw(const DykConvDesc* d, hipStream_t s) {
    if (!conv_pw_eligible(d)) return DYK_ERR_UNSUPPORTED;
    if (d->flags & DYK_EPI_ADDEND) return dyk_conv_launch_pwc(d, s);
    if (d->flags & DYK_EPI_BNBWD) return dyk_conv_launch_pwb(d, s);
    if (d->flags & DYK_EPI_STATS) return dispatch_conv_pw<1, 0>(d, s);
    return dispatch_conv_pw<0, 0>(d, s);
}
