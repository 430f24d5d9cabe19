// Instantiations of the persistent pointwise convolution (conv_pw_kernel.h): fused BatchNorm-backward reduce (DYK_EPI_BNBWD).
#include "conv_pw_kernel.h"

int dyk_conv_launch_pwb(const DykConvDesc* d, hipStream_t s) {
    switch (d->act) {
    case DYK_ACT_MISH: return dispatch_conv_pw<2, DYK_ACT_MISH>(d, s);
    case DYK_ACT_LEAKY: return dispatch_conv_pw<2, DYK_ACT_LEAKY>(d, s);
    default: return dispatch_conv_pw<2, -1>(d, s);
    }
}
