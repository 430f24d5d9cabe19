// Instantiations of the large-tile 3x3 convolution (conv_lt_kernel.h): fused BatchNorm-backward epilogues.
#include "conv_lt_kernel.h"

int dyk_conv_launch_ltb(const DykConvDesc* d, hipStream_t s) { return dispatch_conv_lt<1>(d, s); }
