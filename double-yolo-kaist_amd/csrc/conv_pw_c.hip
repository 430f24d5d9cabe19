// Instantiations of the persistent pointwise convolution (conv_pw_kernel.h): BatchNorm-backward reduce in residual-chain form
// (DYK_EPI_BNBWD | DYK_EPI_ADDEND).
#include "conv_pw_kernel.h"

int dyk_conv_launch_pwc(const DykConvDesc* d, hipStream_t s) {
    switch (d->act) {
    case DYK_ACT_MISH: return dispatch_conv_pw<3, DYK_ACT_MISH>(d, s);
    case DYK_ACT_LEAKY: return dispatch_conv_pw<3, DYK_ACT_LEAKY>(d, s);
    default: return dispatch_conv_pw<3, -1>(d, s);
    }
}
