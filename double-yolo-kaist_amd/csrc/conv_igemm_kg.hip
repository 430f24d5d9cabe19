// K-grouped instantiations of the implicit-GEMM convolution (two 4-wave groups per workgroup walk the two halves of the
// input channels; conv_igemm_kernel.h, template parameter KG): bf16, 128-byte K step, 2-stage rings, 80 / 160 pixel
// tiles.  For the deep layers whose ~256 tiles leave one 4-wave workgroup per CU.  Both epilogue families (8 kernels).
#include "conv_igemm_kernel.h"

template <int EPIK>
static int launch_kg(const DykConvDesc* d, hipStream_t s) {
    const int tile = (d->tune >> 12) & 0xf;
    int bm = d->Cout > 64 ? 128 : 64;
    const int bm_code = (d->tune >> 24) & 0xf;
    if (bm_code == 2) bm = 64;
    if (bm_code == 3) bm = 128;
    if (tile == 1) return bm == 128 ? launch_conv_impl<bf16_t, 128, 80, 128, 2, 2, EPIK>(d, s) : launch_conv_impl<bf16_t, 64, 80, 128, 2, 2, EPIK>(d, s);
    if (tile == 2) return bm == 128 ? launch_conv_impl<bf16_t, 128, 160, 128, 2, 2, EPIK>(d, s) : launch_conv_impl<bf16_t, 64, 160, 128, 2, 2, EPIK>(d, s);
    return DYK_ERR_UNSUPPORTED;
}

int dyk_conv_launch_kg(const DykConvDesc* d, hipStream_t s) {
    if (d->dtype != DYK_BF16 || (d->Cin * 2) % 128 || d->Cin / 64 < 2 * conv_splitk_of(d)) return DYK_ERR_UNSUPPORTED;
    return (d->flags & DYK_EPI_BNBWD) ? launch_kg<1>(d, s) : launch_kg<0>(d, s);
}
