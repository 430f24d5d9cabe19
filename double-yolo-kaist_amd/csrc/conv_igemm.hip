// C-ABI front end of the implicit-GEMM convolution: argument validation and the choice of the pixel-tile
// instantiation (conv_igemm_n{80,128,160}.hip; kernel in conv_igemm_kernel.h).
#include <stddef.h>
#include <string.h>
#include "dyk_common.h"

int dyk_conv_launch_n128(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_n80(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_n160(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_halo(const DykConvDesc* d, hipStream_t s, int th);
int dyk_conv_launch_kg(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_lt(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_sc(const DykConvDesc* d, hipStream_t s);
int dyk_conv_launch_pw(const DykConvDesc* d, hipStream_t s);

static int conv_validate(const DykConvDesc* d) {
    if (!d || !d->x || !d->w || !d->y) return DYK_ERR_ARG;
    if (d->ntaps < 0 || d->ntaps > DYK_MAX_TAPS) return DYK_ERR_ARG;
    if (d->B <= 0 || d->Hg <= 0 || d->Wg <= 0 || d->Cout <= 0 || d->Cin <= 0) return DYK_ERR_ARG;
    if (d->ncls < 0 || d->ncls > 4) return DYK_ERR_ARG;
    for (int c = 0; c < (d->ncls > 1 ? d->ncls : 0); ++c)
        if (d->cls_first[c] < 0 || d->cls_ntaps[c] < 0 || d->cls_first[c] + d->cls_ntaps[c] > d->ntaps || d->cls_ooy[c] < 0 ||
            d->cls_ooy[c] >= d->osy || d->cls_oox[c] < 0 || d->cls_oox[c] >= d->osx)
            return DYK_ERR_ARG;
    if ((d->flags & DYK_EPI_STATS) && !d->stats) return DYK_ERR_ARG;
    if ((d->flags & DYK_EPI_RESIDUAL) && !d->res) return DYK_ERR_ARG;
    if ((d->flags & DYK_EPI_ADDEND) && (!(d->flags & DYK_EPI_BNBWD) || ((uintptr_t)d->add % 16))) return DYK_ERR_ARG;     // (add == NULL: zero addend)
    if (d->flags & DYK_EPI_BNFWD) {
        // conv + BatchNorm + activation in one launch: statistics on, bf16, one problem, whole 16-byte channel chunks
        if (!(d->flags & DYK_EPI_STATS) || (d->flags & (DYK_EPI_AFFINE | DYK_EPI_ACCUM | DYK_EPI_OUT_F32 | DYK_EPI_BNBWD | DYK_EPI_ADDEND)))
            return DYK_ERR_ARG;
        if (d->dtype != DYK_BF16 || d->twin || d->ncls > 1 || d->bn_count <= 0) return DYK_ERR_ARG;
        if (!d->y2 || !d->bn_counter || !d->scale || !d->shift || ((uintptr_t)d->y2 % 16) || ((uintptr_t)d->bn_counter % 8)) return DYK_ERR_ARG;
        if (d->Cout % 8 || d->ldy % 8 || d->ldy2 % 8 || d->ldy2 < d->Cout) return DYK_ERR_ARG;
        if ((d->flags & DYK_EPI_RESIDUAL) && (d->ldr % 4 || ((uintptr_t)d->res % 8))) return DYK_ERR_ARG;
        if ((long)d->B * d->Ho * d->Wo * d->ldy2 >= (1L << 31)) return DYK_ERR_ARG;
    }
    if (d->flags & DYK_EPI_BNBWD) {
        if (d->flags & (DYK_EPI_AFFINE | DYK_EPI_RESIDUAL | DYK_EPI_STATS | DYK_EPI_ACCUM | DYK_EPI_OUT_F32)) return DYK_ERR_ARG;
        if (!d->res || !d->stats || !d->scale || !d->shift || !d->aux0 || !d->aux1) return DYK_ERR_ARG;
        const int es = d->dtype == DYK_BF16 ? 2 : 4;
        // needs the staged (vector) epilogue: 16-byte aligned rows of y and res, whole 16-byte channel chunks
        if (d->Cout % (16 / es) || ((size_t)d->ldy * es) % 16 || ((size_t)d->ldr * es) % 16 || ((uintptr_t)d->y % 16) ||
            ((uintptr_t)d->res % 16))
            return DYK_ERR_ARG;
    }
    if (d->splitk > 1) {
        // split-K across workgroups: private scratch + zeroed counters; one problem, one parity class, no in-launch BatchNorm
        if (d->splitk > 16 || !d->sk_ws || !d->sk_cnt || ((uintptr_t)d->sk_ws % 16) || ((uintptr_t)d->sk_cnt % 4) || d->sk_cnt_n <= 0 ||
            d->sk_ws_bytes <= 0)
            return DYK_ERR_ARG;
        if (d->twin || d->ncls > 1 || (d->flags & DYK_EPI_BNFWD)) return DYK_ERR_ARG;
    }
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    if (d->dtype != DYK_BF16 && d->dtype != DYK_F32) return DYK_ERR_ARG;
    if (d->ldx % epv || ((uintptr_t)d->x % 16) || ((uintptr_t)d->w % 16)) return DYK_ERR_ARG;
    // 32-bit element offsets inside the kernel
    if ((long)d->B * d->Hi * d->Wi * d->ldx >= (1L << 31)) return DYK_ERR_ARG;
    if ((long)d->B * d->Ho * d->Wo * d->ldy >= (1L << 31)) return DYK_ERR_ARG;
    if (d->Hi > 16000 || d->Wi > 16000) return DYK_ERR_ARG;
    return DYK_OK;
}

extern "C" int dyk_conv_igemm(const DykConvDesc* d, void* stream) {
    int rc0 = conv_validate(d);
    if (rc0 != DYK_OK) return rc0;
    if (d->twin) {
        // two-problem launch: the twin must be a valid problem of its own and agree in every non-pointer field
        if ((rc0 = conv_validate(d->twin)) != DYK_OK) return rc0;
        const size_t lo = offsetof(DykConvDesc, dtype), hi = offsetof(DykConvDesc, twin);
        if (memcmp((const char*)d + lo, (const char*)d->twin + lo, hi - lo) != 0) return DYK_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int tile = d->dtype == DYK_BF16 ? (d->tune >> 12) & 0xf : 0;      // 80 / 160 pixel tiles are built for bf16 only
    if (d->flags & DYK_EPI_BNFWD) {        // generic tiles only (the halo / K-grouped kernels do not carry this epilogue)
        if (tile == 1 || tile == 3) return dyk_conv_launch_n80(d, s);
        if (tile == 2 || tile == 4) return dyk_conv_launch_n160(d, s);
        return dyk_conv_launch_n128(d, s);
    }
    if (tile == 5) {                       // large-tile 3x3 kernels (conv_lt_kernel.h); generic 160-pixel tiles where they do not apply
        const int rc = dyk_conv_launch_lt(d, s);
        if (rc != DYK_ERR_UNSUPPORTED || ((d->tune >> 23) & 1)) return rc;       // (bit 23, analysis: no fallback)
        return dyk_conv_launch_n160(d, s);
    }
    if (tile == 6) {                       // resident-weight 3x3 data gradient into 32-channel tensors (conv_sc.hip)
        const int rc = d->splitk > 1 ? DYK_ERR_UNSUPPORTED : dyk_conv_launch_sc(d, s);
        if (rc != DYK_ERR_UNSUPPORTED || ((d->tune >> 23) & 1)) return rc;       // (bit 23, analysis: no fallback)
        return dyk_conv_launch_n128(d, s);
    }
    if (tile == 7) {                       // persistent resident-weight pointwise kernels (conv_pw_kernel.h)
        const int rc = dyk_conv_launch_pw(d, s);
        if (rc != DYK_ERR_UNSUPPORTED || ((d->tune >> 23) & 1)) return rc;       // (bit 23, analysis: no fallback)
        return dyk_conv_launch_n128(d, s);
    }
    if (((d->tune >> 28) & 7) == 1) {      // K-grouped workgroups (conv_igemm_kg.hip); generic tiles where they do not apply
        const int rc = dyk_conv_launch_kg(d, s);
        if (rc != DYK_ERR_UNSUPPORTED) return rc;
    }
    if (tile == 3 || tile == 4) {          // 3x3 halo kernel; falls back to the generic tiles when the problem does not fit it
        const int rc = dyk_conv_launch_halo(d, s, tile == 3 ? 4 : 8);
        if (rc != DYK_ERR_UNSUPPORTED) return rc;
        return tile == 3 ? dyk_conv_launch_n80(d, s) : dyk_conv_launch_n160(d, s);
    }
    if (tile == 1) return dyk_conv_launch_n80(d, s);
    if (tile == 2) return dyk_conv_launch_n160(d, s);
    return dyk_conv_launch_n128(d, s);
}

extern "C" int dyk_conv_bnfwd_max_grid(void) { return 256; }

// workgroups of the generic-tile launch of `d` (the tile selection of dispatch_conv_bn; halo / K-grouped launches differ)
extern "C" int dyk_conv_grid(const DykConvDesc* d) {
    if (!d || d->B <= 0 || d->Hg <= 0 || d->Wg <= 0 || d->Cout <= 0) return DYK_ERR_ARG;
    const int tile = d->dtype == DYK_BF16 ? (d->tune >> 12) & 0xf : 0;
    const int bn = (tile == 1 || tile == 3) ? 80 : ((tile == 2 || tile == 4) ? 160 : 128);
    int bm = d->Cout > 64 ? 128 : (d->Cout > 32 ? 64 : 32);
    const int bm_code = (d->tune >> 24) & 0xf;
    if (bm_code >= 1 && bm_code <= 3) bm = 16 << bm_code;
    if (bn == 80 && bm == 32) bm = 64;
    const long n = (long)d->B * d->Hg * d->Wg;
    return (int)((n + bn - 1) / bn) * ((d->Cout + bm - 1) / bm) * (d->ncls > 1 ? d->ncls : 1);
}

// split-K scratch: bytes of sk_ws and words of sk_cnt that hold for every kernel the tune word of `d` can select (the chosen
// kernel AND the generic tile it falls back to): S * 4 bytes * (Cout rounded up to the channel tile) * (positions rounded up
// to the pixel tile), tiles = their product counted in tiles
extern "C" int64_t dyk_conv_splitk_ws_bytes(const DykConvDesc* d, int32_t* tiles) {
    if (tiles) *tiles = 0;
    if (!d || d->B <= 0 || d->Hg <= 0 || d->Wg <= 0 || d->Cout <= 0) return DYK_ERR_ARG;
    if (d->splitk <= 1) return 0;
    const long n = (long)d->B * d->Hg * d->Wg;
    const int tile = d->dtype == DYK_BF16 ? (d->tune >> 12) & 0xf : 0;
    int64_t bytes = 0;
    int32_t nt = 0;
    auto cover = [&](int bm, int bn) {
        const int64_t tm = (d->Cout + bm - 1) / bm, tn = (n + bn - 1) / bn;
        const int64_t b = tm * tn * (int64_t)d->splitk * bm * bn * 4;
        if (b > bytes) bytes = b;
        if (tm * tn > nt) nt = (int32_t)(tm * tn);
    };
    // generic tile (also the fallback of the large-tile / halo / resident-weight codes)
    {
        const int bn = (tile == 1 || tile == 3) ? 80 : ((tile == 2 || tile == 4 || tile == 5) ? 160 : 128);
        int bm = d->Cout > 64 ? 128 : (d->Cout > 32 ? 64 : 32);
        const int bm_code = (d->tune >> 24) & 0xf;
        if (tile != 5 && bm_code >= 1 && bm_code <= 3) bm = 16 << bm_code;
        if (bn == 80 && bm == 32) bm = 64;
        if (((d->tune >> 28) & 7) == 1 && bm < 64) bm = 64;
        cover(bm, bn);
    }
    if (tile == 5) {
        const int shape = (d->tune >> 8) & 0xf;
        if (shape == 1) cover(128, 320);
        else if (shape == 2) cover(256, 160);
        else cover(128, 160);
    }
    if (tiles) *tiles = nt;
    return bytes;
}
