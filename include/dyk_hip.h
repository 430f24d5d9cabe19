/*
 * dyk_hip.h -- C ABI of libdyk_hip.so, the MI355X (gfx950) native layer under the
 * Double-YOLO-Kaist Python operator surface.
 *
 * The reference (Ye-zixiao/Double-YOLO-Kaist) has no FFI of its own: its hot path is
 * Python calling torch.nn / torchvision ops.  Every entry point below therefore
 * replaces one *torch call site* of the reference; the call site is cited next to
 * each declaration (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless the
 *     name ends in _host.  No allocation, no synchronisation, no ownership transfer:
 *     buffers are borrowed for the duration of the enqueued work.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the call
 *     returns immediately.
 *   - return value: 0 on success, negative DYK_ERR_* otherwise.  Nothing throws,
 *     nothing calls exit().
 *   - activations are channels-last: element (b, y, x, c) of a tensor lives at
 *     base + ((b*H + y)*W + x)*ld + c with ld >= C (ld > C when the tensor is a
 *     channel slice of a concat buffer).  dtype is DYK_BF16 (raw uint16) or DYK_F32.
 *     Channel counts, ld and channel offsets must be multiples of 8 (bf16) / 4 (f32)
 *     so that every pixel row starts 16-byte aligned.
 */
#ifndef DYK_HIP_H
#define DYK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYK_ABI_VERSION 1

enum {
    DYK_OK = 0,
    DYK_ERR_ARG = -1,      /* invalid argument / unsupported shape */
    DYK_ERR_HIP = -2,      /* a HIP runtime call or kernel launch failed */
    DYK_ERR_UNSUPPORTED = -3,
    DYK_ERR_STATE = -4     /* plan used before bind, etc. */
};

enum { DYK_F32 = 0, DYK_BF16 = 1 };

/* activation codes; the strings are the cfg `activation=` values (models.py:51-64) */
enum {
    DYK_ACT_LINEAR = 0,
    DYK_ACT_LEAKY = 1,     /* nn.LeakyReLU(0.1) */
    DYK_ACT_MISH = 2,      /* nn.Mish */
    DYK_ACT_RELU = 3,
    DYK_ACT_RELU6 = 4,
    DYK_ACT_HSIGMOID = 5,  /* nn.Hardsigmoid */
    DYK_ACT_HSWISH = 6     /* nn.Hardswish */
};

int dyk_abi_version(void);
const char* dyk_error_string(int code);

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the matrix cores (MFMA), forward and data-gradient.
 * Replaces nn.Conv2d forward (models.py:34-42, layers.py:181-182,223-231) and, with a
 * transposed weight pack and the tap table of the gradient, torch autograd's
 * conv backward-data.  One launch computes, for every position (b, yo, xo) of a launch
 * grid [B, Hg, Wg] and every output channel co:
 *     acc = sum_t sum_ci  x[b, yo*isy + tdy[t], xo*isx + tdx[t], ci] * w[twt[t]][co][ci]
 * (taps falling outside [0,Hi)x[0,Wi) contribute zero) and stores epilogue(acc) at
 *     y[b, yo*osy + ooy, xo*osx + oox, co].
 * A stride-1/2 forward conv uses (isy,isx)=stride, (osy,osx)=1; the data gradient of a
 * stride-s conv is s*s launches, one per output parity class, with (osy,osx)=s.
 * ---------------------------------------------------------------------------------- */
#define DYK_MAX_TAPS 25

enum {
    DYK_EPI_AFFINE = 1,   /* v = acc*scale[co] + shift[co]  (scale==NULL -> 1, shift==NULL -> 0) */
    DYK_EPI_RESIDUAL = 2, /* v += res[b, y, x, co]   (after the activation) */
    DYK_EPI_STATS = 4,    /* atomically add sum(acc), sum(acc^2) per channel into stats[0..Cout), stats[Cout..2Cout) */
    DYK_EPI_ACCUM = 8,    /* y = y_old + v (gradient accumulation) */
    DYK_EPI_OUT_F32 = 16  /* y is float regardless of dtype */
};

typedef struct DykConvDesc {
    const void* x;        /* input activations, dtype */
    const void* w;        /* packed weights [ntaps_total][Cout][Cin], dtype (dyk_pack_conv_weight) */
    void* y;              /* output */
    const float* scale;   /* [Cout] or NULL */
    const float* shift;   /* [Cout] or NULL */
    const void* res;      /* residual, dtype, or NULL */
    double* stats;        /* [2*Cout] or NULL */
    int32_t dtype;
    int32_t ldx, ldy, ldr;          /* pixel strides in elements */
    int32_t B, Hi, Wi, Cin, Cout;
    int32_t Hg, Wg;                 /* launch grid */
    int32_t Ho, Wo;                 /* full output extent */
    int32_t isy, isx, osy, osx, ooy, oox;
    int32_t ntaps;
    int8_t tdy[DYK_MAX_TAPS];
    int8_t tdx[DYK_MAX_TAPS];
    int8_t twt[DYK_MAX_TAPS];
    int8_t _pad;
    int32_t act;                    /* DYK_ACT_* applied after the affine */
    int32_t flags;                  /* DYK_EPI_* */
} DykConvDesc;

int dyk_conv_igemm(const DykConvDesc* desc, void* stream);

/* ------------------------------------------------------------------------------------
 * Convolution weight gradient (split-K MFMA GEMM over output pixels, fp32 atomics):
 *     dw[twt[t]][co][ci] += sum_{b,yo,xo} dy[b,yo,xo,co] * x[b, yo*isy + tdy[t], xo*isx + tdx[t], ci]
 * dw is the fp32 gradient in the packed [tap][Cout][Cin] order (the order the master weights
 * are stored in); it is ACCUMULATED into, so the caller zeroes it when a fresh
 * gradient is wanted.  Replaces autograd's convolution_backward(weight) for models.py:34-42.
 * Rows of x / dy must be readable up to round_up(C, 16 bytes) (ld >= that).
 * ---------------------------------------------------------------------------------- */
typedef struct DykWgradDesc {
    const void* x;
    const void* dy;
    float* dw;
    int32_t dtype;
    int32_t ldx, lddy;
    int32_t B, Hi, Wi, Cin, Ho, Wo, Cout;
    int32_t isy, isx;
    int32_t ntaps;
    int8_t tdy[DYK_MAX_TAPS];
    int8_t tdx[DYK_MAX_TAPS];
    int8_t twt[DYK_MAX_TAPS];
    int8_t _pad;
    int32_t splits;                 /* K splits; <= 0 selects automatically */
} DykWgradDesc;

int dyk_conv_wgrad(const DykWgradDesc* desc, void* stream);

/* Weight pack: torch OIHW float32 [Cout][Cin][kh][kw] (nn.Conv2d.weight, models.py:34)
 *   transposed == 0:  out[t][co][ci] = w[co][ci][t]     (forward)
 *   transposed == 1:  out[t][ci][co] = w[co][ci][t]     (data gradient: roles of Cin/Cout swap)
 * with t = kh_index*kw + kw_index; rows are padded with zeros up to Cin_pad / Cout_pad. */
int dyk_pack_conv_weight(const float* w_oihw, void* out, int32_t Cout, int32_t Cin, int32_t kh,
                         int32_t kw, int32_t Cout_pad, int32_t Cin_pad, int32_t transposed,
                         int32_t dtype, void* stream);

/* Layout/precision conversion at the drop-in boundary (YOLO.forward takes torch NCHW float32,
 * models.py:279): out[b,y,x,c] = in[b,c,y,x] * mul for c < C, zero for C <= c < Cpad. */
int dyk_nchw_to_nhwc(const float* in, void* out, int32_t B, int32_t C, int32_t H, int32_t W,
                     int32_t Cpad, int32_t ldo, float mul, int32_t dtype, void* stream);
int dyk_nhwc_to_nchw(const void* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                     int32_t ldi, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYK_HIP_H */
