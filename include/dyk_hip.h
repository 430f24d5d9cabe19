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

#define DYK_ABI_VERSION 5   /* 5: round 6, DykWgradDesc in-launch fold fields (sk_ws .. sk_cnt_n), dyk_conv_wgrad_fold_ws_bytes, second region of DYK_OP_MEMSET; 4: round 5, DykConvDesc split-K fields (sk_ws .. splitk), dyk_build_sha; 3: round 4, DykStemDesc fused BatchNorm-backward apply (bn_* fields); 2: round-3 descriptor layouts */

enum {
    DYK_OK = 0,
    DYK_ERR_ARG = -1,      /* invalid argument / unsupported shape */
    DYK_ERR_HIP = -2,      /* a HIP runtime call or kernel launch failed */
    DYK_ERR_UNSUPPORTED = -3,
    DYK_ERR_STATE = -4     /* plan used before bind, etc. */
};

enum { DYK_F32 = 0, DYK_BF16 = 1, DYK_U8 = 2 /* image input of dyk_image_prep only */ };

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
/* digest (16 hex digits) of the kernel sources, this header and the Makefile the library was built from
 * (double-yolo-kaist_amd/dyk/buildinfo.py); the Python loader refuses a library whose digest differs from the tree's */
const char* dyk_build_sha(void);
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
    DYK_EPI_STATS = 4,    /* atomically add sum(acc), sum(acc^2) per channel into one of `stats_slots` replicas of
                             stats[0..Cout) / stats[Cout..2Cout): replica (workgroup id % stats_slots), 2*Cout doubles each
                             (replication bounds the atomic contention per address; dyk_bn_finalize sums the replicas) */
    DYK_EPI_ACCUM = 8,    /* y = y_old + v (gradient accumulation) */
    DYK_EPI_OUT_F32 = 16, /* y is float regardless of dtype */
    DYK_EPI_BNBWD = 32,   /* data-gradient launches only: the tensor being produced is the gradient wrt the output z of a
                             train-mode BatchNorm + activation whose raw conv output is `res` (same shape as y).  The
                             epilogue stores  da = v * act'(res*scale + shift)  instead of v and adds sum(da),
                             sum(da * (res - aux0) * aux1) per channel into a `stats` replica -- the reduce pass of that
                             BatchNorm's backward, fused (`act` = its activation; scale/shift/aux0/aux1 = its
                             scale, shift, saved mean, saved rstd).  Excludes AFFINE, RESIDUAL, STATS, ACCUM, OUT_F32. */
    DYK_EPI_BNFWD = 128,  /* with DYK_EPI_STATS, forward launches of a train-mode Conv2d + BatchNorm2d + activation block
                             (models.py:34-62) whose workgroups are ALL resident at once: conv, batch statistics, finalize and
                             normalise + activation in ONE launch.  Every workgroup stores its raw tile to y and adds its sums
                             to the statistics replicas, then arrives at the device-wide counter `bn_counter` (zero on entry;
                             the workgroup that leaves last zeroes it again, so successive launches share it without
                             re-arming) and waits until all gridDim.x workgroups have; it folds the replicas
                             of its own channels (same order and arithmetic as dyk_bn_finalize_act_fwd: scale = gamma * rstd,
                             shift = beta - mean * scale), normalises the tile it still holds in its accumulators -- rounded
                             to dtype first, i.e. exactly the values a separate pass would read back from y -- applies
                             `act` (and the RESIDUAL, if flagged) and stores it to y2 (pixel stride ldy2).  The workgroups of
                             the first pixel tile publish scale / shift / saved mean / saved rstd and update the running
                             statistics.  bf16 only; launches larger than dyk_conv_bnfwd_max_grid() are refused
                             (DYK_ERR_UNSUPPORTED): the caller may run at most TWO such launches concurrently.  A wait that
                             does not complete within 50 ms sets bit 0 of bn_counter[1] and abandons the normalise step
                             instead of hanging. */
    DYK_EPI_ADDEND = 64   /* with DYK_EPI_BNBWD only: dz = v + add[b, y, x, co] (`add`: the gradient arriving over a plain
                             [shortcut], same geometry and pixel stride as y).  The epilogue stores dz ITSELF (the next
                             link of the residual chain needs it), rounded to dtype, and reduces the sums of
                             da = dz * act'(...) computed from the rounded value; the BatchNorm-backward apply pass
                             then applies act' itself.  Removes the gradient copy of the [shortcut] and the separate
                             reduce pass for residual blocks.  add == NULL: no addend (dz = v) -- the keep-dz form alone,
                             for the LAST [shortcut] of a residual chain, whose gradient the skip branch still needs. */
};

typedef struct DykConvDesc {
    const void* x;        /* input activations, dtype */
    const void* w;        /* packed weights [ntaps_total][Cout][Cin], dtype (dyk_pack_conv_weight) */
    void* y;              /* output */
    const float* scale;   /* [Cout] or NULL */
    const float* shift;   /* [Cout] or NULL */
    const void* res;      /* residual, dtype, or NULL */
    double* stats;        /* [2*Cout] or NULL */
    const float* aux0;    /* DYK_EPI_BNBWD: saved mean [Cout] */
    const float* aux1;    /* DYK_EPI_BNBWD: saved rstd [Cout] */
    const void* add;      /* DYK_EPI_ADDEND: gradient addend, dtype, laid out like y */
    /* DYK_EPI_BNFWD (else ignored): */
    void* y2;             /* normalised + activated output, dtype */
    const float* bn_gamma;      /* [Cout] or NULL (1) */
    const float* bn_beta;       /* [Cout] or NULL (0) */
    float* bn_running_mean;     /* [Cout] or NULL: updated with bn_momentum (unbiased variance), as nn.BatchNorm2d does */
    float* bn_running_var;
    float* bn_save_mean;        /* [Cout] or NULL: batch mean / rstd for the backward pass; scale / shift above are OUTPUTS here */
    float* bn_save_rstd;
    uint32_t* bn_counter;       /* [4]: arrivals, error word, departures, unused; zero before the first launch */
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
    /* Output-parity classes in ONE launch (data gradient of a stride-s conv): ncls > 1 splits the tap table into
     * ncls runs, class c = taps [cls_first[c], +cls_ntaps[c]) written at output offset (cls_ooy[c], cls_oox[c]);
     * ntaps is the total, ooy/oox are ignored, every class walks the same Hg x Wg grid.  Workgroups of the classes of
     * one pixel tile get consecutive ids (one XCD): the gradient tile is fetched once and the interleaved output
     * lines meet in that L2.  ncls <= 1: a single class (ntaps, ooy, oox). */
    int8_t ncls;
    int8_t cls_first[4], cls_ntaps[4], cls_ooy[4], cls_oox[4];
    int8_t _pad2[3];
    int32_t act;                    /* DYK_ACT_* applied after the affine */
    int32_t flags;                  /* DYK_EPI_* */
    int32_t stats_slots;            /* number of stats replicas (>= 1; 0 is read as 1) */
    int32_t ldy2;                   /* DYK_EPI_BNFWD: pixel stride of y2 in elements */
    int32_t bn_count;               /* DYK_EPI_BNFWD: values per channel (B * Ho * Wo) */
    float bn_momentum, bn_eps;      /* DYK_EPI_BNFWD */
    int32_t tune;                   /* 0 = built-in heuristic; else tile configuration chosen by the plan compiler's
                                       per-shape measurement: bits 0..7 K-step bytes (64|128), 8..11 LDS ring stages
                                       (2|3|4|6), 12..15 pixel tile (0 = 128, 1 = 80, 2 = 160, 3 / 4 = halo kernel, 5 = large-tile kernels,
                                       6 = resident-weight 3x3 data gradient into 32-channel tensors; bf16), 24..27 channel
                                       tile (0 = by Cout, 1 = 32, 2 = 64, 3 = 128); bits 16..23 analysis switches; 1 << 28 = two K-groups per
                                       workgroup (512 threads, halves of Cin, bf16 / 128-byte K step / 80|160-pixel tiles) */
    const struct DykConvDesc* twin; /* HOST pointer or NULL.  Two-problem launch: a second problem of IDENTICAL geometry, flags and
                                       tile configuration (every non-pointer field equal) whose pointer fields x, w, y, scale,
                                       shift, res, stats, aux0, aux1, add are used -- one launch covers both, the workgroups of
                                       the second problem follow those of the first.  For the shape-identical RGB / LWIR twin
                                       backbones of a dual-stream net (models.py:288,299-303): half the launches, twice the
                                       workgroups per launch on the deep stages.  The twin's own `twin` field is ignored. */
    /* Split-K ACROSS workgroups (ABI 4).  The deep stages of the network (32x40 / 16x20 maps: 20 480 / 5 120 pixels at batch
     * 16) have 64..256 output tiles of a K loop 1 000..18 000 long: tiles large enough to feed the matrix cores leave most
     * CUs idle.  splitk = S >= 2 launches S workgroups per output tile, slice s walking the s-th share of the input-channel
     * chunks; every slice parks its fp32 accumulators in its slab of `sk_ws` (write-through stores), takes a ticket on
     * sk_cnt[tile], and the workgroup that draws the LAST ticket adds the S slabs in slice order -- the sum does not depend on
     * which slice arrives last: bit-reproducible -- and runs the epilogue of `flags` on the folded tile (statistics,
     * BatchNorm-backward reduce, ... unchanged).  No spinning: nothing depends on residency or dispatch order.
     * sk_ws: 16-byte aligned scratch private to this descriptor, sk_ws_bytes >= dyk_conv_splitk_ws_bytes(desc);
     * sk_cnt: sk_cnt_n >= number of output tiles 32-bit words, ZERO before the first launch (the last arriver re-arms its
     * word).  splitk <= 1: off.  Not combinable with twin, ncls > 1 or DYK_EPI_BNFWD; generic tiles (incl. K-grouped) and
     * the large-tile kernels carry it, the halo / resident-weight kernels fall back to their generic tile. */
    void* sk_ws;
    uint32_t* sk_cnt;
    int64_t sk_ws_bytes;
    int32_t sk_cnt_n;
    int32_t splitk;
} DykConvDesc;

int dyk_conv_igemm(const DykConvDesc* desc, void* stream);
/* largest launch (workgroups) DYK_EPI_BNFWD accepts, and the workgroups the launch of `desc` (with its tune word) would take */
int dyk_conv_bnfwd_max_grid(void);
int dyk_conv_grid(const DykConvDesc* desc);
/* split-K scratch of `desc` (its tune word and splitk): bytes of sk_ws, and (via *tiles, may be NULL) the words of sk_cnt;
 * upper bounds that hold for every kernel the tune word can select.  0 when splitk <= 1; negative error code on bad input */
int64_t dyk_conv_splitk_ws_bytes(const DykConvDesc* desc, int32_t* tiles);

/* ------------------------------------------------------------------------------------
 * Convolution weight gradient (split-K MFMA GEMM over output pixels, fp32 atomics):
 *     dw[twt[t]][co][ci] += sum_{b,yo,xo} dy[b,yo,xo,co] * x[b, yo*isy + tdy[t], xo*isx + tdx[t], ci]
 * dw is the fp32 gradient in the packed [tap][Cout][Cin] order (the order the master weights
 * are stored in); it is ACCUMULATED into, so the caller zeroes it when a fresh
 * gradient is wanted.  Replaces autograd's convolution_backward(weight) for models.py:34-42.
 * Rows of x / dy must be readable up to round_up(C, 16 bytes) (ld >= that).
 * ---------------------------------------------------------------------------------- */
/* one problem of a GROUPED weight-gradient launch (DykWgradDesc.group): identical geometry, own tensors */
typedef struct DykWgradGroupEntry {
    const void* x;
    const void* dy;
    float* dw;
    float* part;
} DykWgradGroupEntry;

typedef struct DykWgradDesc {
    const void* x;
    const void* dy;
    float* dw;
    float* part;                    /* NULL: partial tiles are added to dw with fp32 atomics.  Else: K split s stores its
                                       tiles (plain stores, same [tap][Cout][Cin] layout as dw) into the plane
                                       part + s * part_stride; dyk_grad_reduce later adds the planes to dw --
                                       no atomics, bit-reproducible.  Needs the split count of dyk_conv_wgrad_splits. */
    int64_t part_stride;            /* floats between the planes of consecutive splits */
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
    int32_t lddw;                   /* row stride of dw in floats; <= 0 means Cin */
    int32_t tune;                   /* 0 = default; else LDS ring stages (2 | 3) | K-groups per workgroup (1 | 2) << 8 | tile cap << 24
                                       (1 = tiles of at most 64 x 64: more tiles, fewer K splits for small GEMMs) | 1 << 28: multi-tap
                                       kernel (3x3 / pad 1, bf16: dy and the x halo tile of a row segment staged once for all nine
                                       taps; ignored where it does not apply) | 2 << 28: row-block kernel (3x3 / pad 1, bf16:
                                       K steps of 128 pixels -- 256 with K-groups field = 2 -- shaped nimg x rows x columns to
                                       fit the map, 64 x 32 x 9-tap tiles; ignored where it does not apply) | 1 << 20: the caller vouches
                                       that nothing else adds to dw while the launch runs -- a launch with ONE K split may then
                                       read-add-write its tiles instead of issuing atomics (row-block kernel)
                                       | 3 << 28: pixel-streaming kernel (1x1 / stride 1, bf16: 8-wave workgroups, LDS-DMA ring of up
                                       to 8 stages; low byte = ring stages, bits 8..11 = 1 caps the tile at 64 x 64, bits 12..15 = 1:
                                       64-pixel stages for 64 x 64 tiles; ignored where it does not apply) */
    const struct DykWgradDesc* twin;/* HOST pointer or NULL: second problem of identical geometry / splits / tune whose x, dy, dw, part
                                       are used (two-problem launch, see DykConvDesc.twin) */
    /* In-launch fold of the K splits (ABI 5; part == NULL, splits >= 2; kernels: pixel-streaming 1x1): the S slices of an
     * output tile hand their accumulators over through private write-through slabs in sk_ws, take a ticket on sk_cnt[tile],
     * and the workgroup that draws the last ticket adds the S slabs IN SLICE ORDER (bit-reproducible) and adds the folded tile
     * to dw -- read-add-write when tune bit 20 vouches for a single writer, fp32 atomics otherwise.  No partial planes, no
     * dyk_grad_reduce entry.  sk_ws: 16-byte aligned, sk_ws_bytes >= dyk_conv_wgrad_fold_ws_bytes(desc); sk_cnt: sk_cnt_n >=
     * tiles 32-bit words, ZERO before the first launch (the last arriver re-arms its word).  NULL: off. */
    void* sk_ws;
    uint32_t* sk_cnt;
    int64_t sk_ws_bytes;
    int32_t sk_cnt_n;
    int32_t _pad2;
    /* Grouped launch (round 6; kernels: pixel-streaming 1x1, row-block 3x3): group_n >= 2 problems of THIS geometry / tune /
     * splits / part_stride in one launch -- the weight gradients of the repeated residual units of a stage, which become
     * ready one after the other and are nobody's input before the optimizer.  `group` is a DEVICE array of group_n entries
     * (x, dy, dw, part per problem; this descriptor's own x / dy / dw / part are ignored); the workgroups of problem p are
     * blocks [p * tiles * splits, (p + 1) * tiles * splits).  Every problem's result is what its own launch would give, bit for
     * bit.  Not combinable with twin or the in-launch fold. */
    const DykWgradGroupEntry* group;
    int32_t group_n;
    int32_t _pad3;
} DykWgradDesc;

int dyk_conv_wgrad(const DykWgradDesc* desc, void* stream);
/* Number of K splits (> 0) dyk_conv_wgrad will use for this descriptor (its `splits` and `tune` included): the
 * number of planes a `part` buffer must hold.  Negative = error code. */
int dyk_conv_wgrad_splits(const DykWgradDesc* desc);
/* Which kernel dyk_conv_wgrad runs for this descriptor (its `tune` included): 0 = per-tap split-K kernel, 1 = multi-tap
 * 3x3 kernel (row segments), 2 = row-block 3x3 kernel (conv_wgrad_rb.hip), 3 = pixel-streaming 1x1 kernel
 * (conv_wgrad_ps.hip).  A tune word that asks for a variant the problem is not eligible for falls back to 0.
 * Negative = error code. */
int dyk_conv_wgrad_variant(const DykWgradDesc* desc);
/* in-launch fold scratch of `desc` (its tune word and `splits` >= 2, taken literally): bytes of sk_ws, and (via *tiles, may be
 * NULL) the words of sk_cnt.  0 when the kernel the tune word selects has no in-launch fold or splits < 2; negative = error */
int64_t dyk_conv_wgrad_fold_ws_bytes(const DykWgradDesc* desc, int32_t* tiles);

/* G[g_off + i] += sum_s part[part_off + s * plane + i], i < n, for every entry of a device table: folds the per-split
 * planes written by dyk_conv_wgrad into the flat gradient buffer, one launch for many layers.  n is a multiple of 4;
 * chunk_begin = exclusive prefix sum of ceil(n / 1024) over the entries, total_chunks its total. */
typedef struct DykGradReduceEntry {
    int64_t g_off, part_off, plane;
    int32_t n, splits, chunk_begin, _pad;
} DykGradReduceEntry;
int dyk_grad_reduce(float* G, const float* part, const DykGradReduceEntry* table_dev, int32_t n_entries,
                    int32_t total_chunks, void* stream);

/* ------------------------------------------------------------------------------------
 * Channels-last elementwise / per-channel kernels share one descriptor; each entry point
 * documents the fields it reads.  Tensors a, b, out have `dtype`; npix = B*H*W pixels of C
 * channels with pixel strides lda/ldb/ldo.  p0..p3 are per-channel fp32 vectors, red a
 * fp64 reduction buffer.
 * ---------------------------------------------------------------------------------- */
enum {
    DYK_EW_ACCUM = 1,     /* out += result instead of out = result */
    DYK_EW_SKIP = 2       /* dyk_bn_act_bwd_apply: do nothing (the pass is done inside the consumer: DykStemDesc.bn_fused) */
};

typedef struct DykEwDesc {
    const void* a;
    const void* b;
    void* out;
    const float* p0;
    const float* p1;
    const float* p2;
    const float* p3;
    double* red;
    void* aux;                      /* op-specific extra pointer (max-pool argmax map, SE pooled vector, BN dgamma) */
    void* aux2;                     /* second op-specific pointer (BN dbeta) */
    int32_t dtype;
    int32_t npix, C;
    int32_t lda, ldb, ldo;
    int32_t act, flags;
    int32_t B, H, W, k;             /* spatial ops (pool / upsample) */
    float alpha, beta;
    int32_t slots;                  /* replicas of `red` (reductions; 0 is read as 1) */
    const struct DykEwDesc* twin;   /* HOST pointer or NULL: second problem with equal non-pointer fields (two-problem launch, see
                                       DykConvDesc.twin).  Honoured by dyk_bn_act_fwd, dyk_bn_finalize_act_fwd, dyk_bn_act_bwd_reduce,
                                       dyk_bn_act_bwd_apply and dyk_axpby; the other entry points return DYK_ERR_UNSUPPORTED for it */
} DykEwDesc;

/* BatchNorm2d, training mode (nn.BatchNorm2d at models.py:47, torch defaults eps=1e-5,
 * momentum=0.1): from the per-channel sum / sum-of-squares the conv epilogue accumulated in
 * stats[2C] over `count` pixels, produce scale = gamma*rstd, shift = beta - mean*scale, the
 * saved mean / rstd for backward, update running_mean / running_var (unbiased variance) and
 * reset stats to zero. */
typedef struct DykBnFinalizeDesc {
    double* stats;              /* [2C] in/out (zeroed on return) */
    const float* gamma;         /* [C] or NULL */
    const float* beta;          /* [C] or NULL */
    float* running_mean;        /* [C] or NULL (no update) */
    float* running_var;
    float* scale;               /* [C] out */
    float* shift;               /* [C] out */
    float* save_mean;           /* [C] out or NULL */
    float* save_rstd;           /* [C] out or NULL */
    int32_t C;
    int32_t count;
    float momentum, eps;
    int32_t slots;              /* replicas of stats written by the conv epilogue (0 is read as 1) */
    const struct DykBnFinalizeDesc* twin;   /* HOST pointer or NULL: second problem, equal C / count / momentum / eps / slots */
} DykBnFinalizeDesc;

int dyk_bn_finalize(const DykBnFinalizeDesc* desc, void* stream);
/* dyk_bn_finalize + dyk_bn_act_fwd in one launch: out = act(a * scale + shift) (+ b) with scale / shift derived from
 * the statistics replicas inside the kernel; scale, shift, save_mean, save_rstd and the running statistics are written
 * as by dyk_bn_finalize, but the replicas are left untouched (the caller re-arms them, e.g. one memset per pass). */
int dyk_bn_finalize_act_fwd(const DykBnFinalizeDesc* fin, const DykEwDesc* act_desc, void* stream);

/* eval-mode BatchNorm folded to scale/shift from the running statistics */
int dyk_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                const float* running_var, float eps, float* scale, float* shift, int32_t C,
                void* stream);

/* out = act(a*p0[c] + p1[c]) (+ b)           -- BN apply + activation (+ residual add of a
 * following [shortcut], layers.py:79).  p0/p1 NULL -> 1/0, b NULL -> no residual. */
int dyk_bn_act_fwd(const DykEwDesc* desc, void* stream);

/* backward of the above w.r.t. the raw conv output y (b):  a = dz, p0 = scale, p1 = shift,
 * p2 = mean, p3 = rstd.  reduce: red[c] += sum dact, red[C+c] += sum dact*xhat with
 * dact = dz*act'(y*scale+shift), spread over `slots` replicas of red (2C doubles each);
 * params: folds the replicas into replica 0, then dbeta += red[c], dgamma += red[C+c];
 * apply: out = scale*(dact - red[c]/npix - xhat*red[C+c]/npix).  The fused form (slots > 0) may read its sums as COLUMNS of
 * a wider set of replicas -- the joint reduction a convolution's DYK_EPI_BNBWD epilogue leaves for all the conv + BatchNorm
 * sections a [route] concatenates (reference models.py:116-124 / layers.py:40-48): desc->H = replica stride in doubles
 * (0 = 2C), desc->W = offset of sum(dact*xhat) from sum(dact) (0 = C), `red` points at this layer's first column. */
int dyk_bn_act_bwd_reduce(const DykEwDesc* desc, void* stream);
int dyk_bn_bwd_params(double* red, float* dgamma, float* dbeta, int32_t C, int32_t slots, void* stream);
int dyk_bn_act_bwd_apply(const DykEwDesc* desc, void* stream);

/* out = alpha*s0*a (+ beta*s1*b), s0 = p0 ? p0[0] : 1, s1 = p1 ? p1[0] : 1 (device scalars).
 * Covers channel-slice copy for [route] concat (layers.py:44), the plain [shortcut] add
 * (layers.py:79), the weighted fusion x*w0 + a*w1 (layers.py:66-73) and every gradient
 * accumulation (flags & DYK_EW_ACCUM). */
int dyk_axpby(const DykEwDesc* desc, void* stream);
/* red[0] += sum over pixels/channels of a*b   (gradient of a fusion weight).  With desc->out the same pass also stores
 * out = alpha*s0*a (+ out with DYK_EW_ACCUM), s0 = p0 ? p0[0] : 1 -- the scaled gradient copy of that source, exactly as
 * dyk_axpby would make it (autograd of layers.py:66-73 for one source in ONE pass over the gradient) */
int dyk_dot(const DykEwDesc* desc, void* stream);
/* WeightedFeatureFusion weights (layers.py:66): weff[i] = sigmoid(w[i]) * 2/n, and its backward
 * dw[i] += red[i] * 2/n * sigmoid'(w[i]) */
int dyk_wfuse_weights(const float* w, float* weff, int32_t n, void* stream);
int dyk_wfuse_bwd_params(const float* w, const double* red, float* dw, int32_t n, void* stream);

/* nn.Upsample(scale_factor=2) nearest (models.py:100-101): a [B,H,W,C] -> out [B,2H,2W,C];
 * backward: a = dout [B,2H,2W,C] -> out = din [B,H,W,C] (sum of the 2x2 block). */
int dyk_upsample2x_fwd(const DykEwDesc* desc, void* stream);
int dyk_upsample2x_bwd(const DykEwDesc* desc, void* stream);

/* nn.MaxPool2d(k, stride, padding=(k-1)//2) (models.py:91-94), k <= 15; the stride travels in `slots` (0 / 1 = stride 1,
 * the SPP pools of the shipped cfgs).  H, W are the INPUT extents; the output is [(H+2p-k)/stride+1, (W+2p-k)/stride+1].
 * argmax is a uint8 [B*Ho*Wo][C] map of the window position (dy*k+dx) of the first maximum in scan order
 * (torch CPU tie rule); backward (a = dout, out = din) gathers dout through it (deterministic, no atomics). */
int dyk_maxpool_fwd(const DykEwDesc* desc, uint8_t* argmax, void* stream);
int dyk_maxpool_bwd(const DykEwDesc* desc, const uint8_t* argmax, void* stream);

/* SqueezeExcitation (layers.py:175-190).
 *   dyk_se_pool : pooled[b][c] = alpha * sum_hw a[b,hw,c] * (b ? b[b,hw,c] : 1)
 *                 (alpha = 1/HW gives adaptive_avg_pool2d; with b = dz it is the gradient of the
 *                 per-channel scale).  desc->aux2 (optional): scratch of DYK_SE_POOL_SPLITS * B * C floats;
 *                 with it the pixels of an image are reduced by up to DYK_SE_POOL_SPLITS workgroups whose
 *                 partial sums are folded in a fixed order (same result from run to run)
 *   dyk_se_fc_fwd : scale = hardsigmoid(W2 relu(W1 pooled + b1) + b2); two launches that spread the rows of W1 / W2 over
 *                   the chip.  Needs `ws` (B*(C + 2*Cs) floats): h = relu(..) and t2 = W2 h + b2 are parked there
 *   dyk_se_scale  : out[b,hw,c] = a[b,hw,c]*p0[b*C+c] (+ alpha*p1[b*C+c])
 *                   With desc->red set (backward of a block whose input is the activated output of a conv + BatchNorm layer,
 *                   autograd of models.py:47-56 behind layers.py:188): b = that layer's raw conv output, p2 = its
 *                   scale | shift | mean | rstd vectors [4][C], act = its activation; sum(da), sum(da * xhat) with
 *                   da = out * act'(scale * b + shift) go to the `slots` fp64 replicas red[slots][2][C] exactly as
 *                   dyk_bn_act_bwd_reduce over `out` would leave them; `out` itself stays the gradient
 *   dyk_se_fc_bwd : from dscale = d(loss)/d(scale) produce dpooled and accumulate dW1,db1,dW2,db2.  MUST follow
 *                   dyk_se_fc_fwd on the same `ws` (it reads the h and t2 the forward call parked; nothing is
 *                   recomputed).  Three launches: dt1 = relu'(h) * W2^T (dscale * hardsigmoid'(t2)) -> ws,
 *                   dpooled = W1^T dt1, then one thread per weight element sums its outer products over the
 *                   batch -- no atomics on the weight gradients, every sum in a fixed order.
 *                   dyk_se_fc_bwd may be called in two halves: with dw1 = db1 = dw2 = db2 = NULL it computes dpooled
 *                   (and dt1 into ws) only; with dpooled = NULL only the parameter gradients, from what the first
 *                   half left in ws -- the parameter half is not on the path to dx (dyk/plan.py issues it as its own
 *                   command so that the dependency scheduler can move it off the backward chain). */
typedef struct DykSeFcDesc {
    const float* pooled;   /* [B][C] */
    const float* w1;       /* [Cs][C]  fc1.weight */
    const float* b1;       /* [Cs] */
    const float* w2;       /* [C][Cs]  fc2.weight */
    const float* b2;       /* [C] */
    float* scale;          /* [B][C] out (fwd) */
    const float* dscale;   /* [B][C] (bwd) */
    float* dpooled;        /* [B][C] out (bwd) */
    float* dw1; float* db1; float* dw2; float* db2;   /* accumulated (bwd) */
    float* ws;             /* B*(C + 2*Cs) floats: h [B][Cs] (fwd) | dt1 [B][Cs] (bwd) | t2 [B][C] (fwd) */
    int32_t B, C, Cs;
} DykSeFcDesc;
#define DYK_SE_POOL_SPLITS 16
int dyk_se_pool(const DykEwDesc* desc, float* pooled, void* stream);
int dyk_se_fc_fwd(const DykSeFcDesc* desc, void* stream);
int dyk_se_fc_bwd(const DykSeFcDesc* desc, void* stream);
int dyk_se_scale(const DykEwDesc* desc, void* stream);

/* YOLOLayer training-mode reshape (models.py:229): head conv output y [B,ny,nx,ld] fp32 with
 * channel = a*no + o  ->  p [B,na,ny,nx,no] fp32.  Backward scatters dp back into a zero-padded
 * dy [B,ny,nx,ld] of `dtype` and, if dbias != NULL, accumulates the head bias gradient. */
int dyk_head_permute_fwd(const float* y, float* p, int32_t B, int32_t ny, int32_t nx, int32_t na,
                         int32_t no, int32_t ld, void* stream);
int dyk_head_permute_bwd(const float* dp, void* dy, float* dbias, int32_t B, int32_t ny, int32_t nx,
                         int32_t na, int32_t no, int32_t ld, int32_t dtype, void* stream);

/* Input path of the harness (train_utils/kaist_train_eval_utils.py:54-55 `imgs.to(device).float() / 255.0`, and under
 * multi-scale training :59-71 `F.interpolate(imgs, size=ns, mode='bilinear', align_corners=False)`) as one pass:
 * src [planes = B*C][Hi][Wi] of src_dtype (DYK_U8 loader output, kaist_dataset.py:385-386, or DYK_F32)
 *   -> dst f32 [planes][Ho][Wo] = bilinear(src / div).  Hi == Ho && Wi == Wo is the plain conversion.
 * Source index / lambda arithmetic is ATen's (scale = in/out, src = scale*(dst+0.5)-0.5 clamped at 0). */
int dyk_image_prep(const void* src, float* dst, int32_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                   int32_t src_dtype, float div, void* stream);

/* First-layer patch gather: torch NCHW fp32 image batch -> channels-last patches
 * out[b,yo,xo,(kh*k+kw)*Cin + c] = in[b,c,yo*stride+kh-pad,xo*stride+kw-pad]*mul (zero padded up
 * to ld channels), which turns the Cin=3 stem conv (models.py:35-36) into a 1x1 MFMA conv. */
int dyk_patch_gather(const float* in, void* out, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t k,
                     int32_t stride, int32_t pad, int32_t ld, float mul, int32_t dtype, void* stream);

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

/* ------------------------------------------------------------------------------------
 * Depthwise convolution (nn.Conv2d(groups=C), reference models.py:41 for `groups=` sections and
 * layers.py:223-224 for DepthwiseSeparableConv2d), square kernel k, zero padding `pad`, stride 1 or 2.
 * w is the fp32 master weight in tap-major order [k*k][C] (the parameter store layout).
 *   dyk_dwconv_fwd   : y[b,yo,xo,c] = sum_t x[b, yo*s+kh-pad, xo*s+kw-pad, c] * w[t][c]
 *                      stats (optional): replicated fp64 [slots][2][C] sum / sum of squares of y
 *   dyk_dwconv_dgrad : x := gradient wrt the conv input [B,Hi,Wi,C] (written, or accumulated with
 *                      DYK_EW_ACCUM), y := gradient wrt the conv output [B,Ho,Wo,C] (read)
 *   dyk_dwconv_wgrad : dw[t][c] += sum_p y[p][c] * x[src(p,t)][c]     (y = output gradient)
 * In the two gradient entry points the roles follow the tensor, not the data direction:
 * x/ldx/Hi/Wi always describe the conv-input-shaped tensor, y/ldy/Ho/Wo the conv-output-shaped one.
 * ---------------------------------------------------------------------------------- */
typedef struct DykDwDesc {
    void* x;
    void* y;
    const float* w;
    float* dw;
    double* stats;
    float* part;                    /* dyk_dwconv_wgrad only.  NULL: workgroups add their sums to dw with fp32 atomics.  Else
                                       workgroup row r (of dyk_dwconv_wgrad_rows) stores them into the plane
                                       part + r * k*k*C; dyk_grad_reduce folds the planes (reproducible). */
    int32_t dtype, ldx, ldy;
    int32_t B, Hi, Wi, Ho, Wo, C;
    int32_t k, stride, pad;
    int32_t flags, stats_slots;
    /* dyk_dwconv_dgrad only, res != NULL: the BatchNorm-backward reduce of the layer that produced the conv input is
     * folded into this data gradient (as DYK_EPI_BNBWD does for dyk_conv_igemm): with u = res (the producer's raw conv
     * output, row length ldr), da = dx * act'(scale*u + shift) is stored instead of dx and sum(da), sum(da * (u-mean)*rstd)
     * are added to the fp64 replicas stats[stats_slots][2][C].  bn = scale | shift | mean | rstd, C floats each.
     * Stride 1, k in {3, 5}, bf16, no DYK_EW_ACCUM. */
    const void* res;
    const float* bn;
    int32_t ldr, act;
    /* dyk_dwconv_fwd / dyk_dwconv_wgrad, pre != NULL (round 5): x is the RAW output u of a train-mode Conv2d + BatchNorm2d +
     * activation block (the 1x1 expansion conv of a MobileNet block, reference models.py:34-62 in front of :41 groups=C) whose
     * normalise + activation pass was NOT run: the kernel forms z = dtype(act(scale * u + shift)) on load -- exactly the
     * values that pass would have stored -- and convolves / correlates z.  pre = scale | shift, C floats each (the first half
     * of the `bn` layout), pre_act = that block's DYK_ACT_*.  Padding taps stay zero (the affine is not applied outside the
     * image).  Forward: the LDS-tiled stride-1 kernel only (dyk_dwconv_tile_ok), else DYK_ERR_UNSUPPORTED. */
    const float* pre;
    int32_t pre_act, _pad;
} DykDwDesc;
int dyk_dwconv_fwd(const DykDwDesc* desc, void* stream);
/* 1 when dyk_dwconv_fwd runs `desc` on the LDS-tiled kernel (stride 1, k in {3, 5}, bf16, tile fits): the precondition of
 * `pre` in the forward and of `res` in the data gradient; 0 otherwise; negative = error code */
int dyk_dwconv_tile_ok(const DykDwDesc* desc);
int dyk_dwconv_dgrad(const DykDwDesc* desc, void* stream);
int dyk_dwconv_wgrad(const DykDwDesc* desc, void* stream);
/* number of workgroup rows (= planes of `part`) dyk_dwconv_wgrad uses for this descriptor; negative = error code.  Stride-1
 * 3x3 / 5x5 bf16 problems run on the LDS-tiled persistent kernel (round 5: one pass over dy and x, the next tile's loads in flight
 * while a tile is computed; planes = its workgroups per channel group), everything else on the row kernel (planes = workgroup rows
 * per kernel row); the count depends on the descriptor's shape fields only, never on `part` / `dw`. */
int dyk_dwconv_wgrad_rows(const DykDwDesc* desc);

/* ------------------------------------------------------------------------------------
 * Parameter staging.  Master parameters and gradients are fp32, conv weights stored
 * tap-major [kh*kw][Cout][Cin] (DESIGN.md "parameter store").
 *   dyk_cast_f32       : dst[i] = (dtype) src[i]                      (whole flat buffer, one launch)
 *   dyk_cast_pad_rows  : dst[r][c] = src[r][c] (c < C), 0 (C <= c < Cpad)   (stem weights, K padded)
 *   dyk_transpose_taps : for each entry e of a device table: dst_e[t][ci][co] = src_e[t][co][ci]
 *                        (weights for the data-gradient GEMM), one launch for all layers; dst rows may be
 *                        padded to dst_ld (channel counts that are not a multiple of the GEMM K step).
 * ---------------------------------------------------------------------------------- */
typedef struct DykTransposeEntry {
    int64_t src_off;      /* element offset into src (fp32) */
    int64_t dst_off;      /* element offset into dst (dtype) */
    int32_t taps, rows, cols;   /* src is [taps][rows][cols] */
    int32_t tile_begin;   /* first 32x32 tile index of this entry (exclusive prefix sum) */
    int32_t dst_ld;       /* row length of dst (>= rows, <= rows rounded up to 32; 0 = rows); the tail is zero-filled */
    int32_t _pad;
} DykTransposeEntry;
/* dyk_cast_pad_table: every K-padded weight pack of a model in ONE launch (the MobileNet cfgs have 68 of them: 68 launches of
 * 3 us on the caller's stream at every step boundary, round 5).  Entry e: dst_e[r][c] = (dtype) src_e[r][c] for c < cols,
 * 0 for cols <= c < cpad; with transpose_f32 set dst_e is fp32 [cols][rows] = the transpose of src_e (no padding).  blk_begin =
 * exclusive prefix sum of ceil(elements_e / 2048) over the entries; total_blocks = its end. */
typedef struct DykPadEntry {
    const float* src;
    void* dst;
    int32_t rows, cols, cpad;
    int32_t blk_begin;
    int32_t transpose_f32;
    int32_t _pad;
} DykPadEntry;
int dyk_cast_pad_table(const DykPadEntry* table_dev, int32_t n_entries, int32_t total_blocks, int32_t dtype, void* stream);
int dyk_cast_f32(const float* src, void* dst, int64_t n, int32_t dtype, void* stream);
int dyk_cast_pad_rows(const float* src, void* dst, int32_t R, int32_t C, int32_t Cpad, int32_t dtype, void* stream);
int dyk_transpose_taps(const float* src, void* dst, const DykTransposeEntry* table_dev, int32_t n_entries,
                       int32_t total_tiles, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Command lists.  A compiled plan (one per cfg x batch shape x dtype x mode) is a flat array of
 * commands whose descriptors already hold resolved device pointers; dyk_run_commands enqueues
 * them in order on `stream` from native code (no per-kernel Python / ctypes overhead).
 * desc points at the struct named next to each op code; DykMiscDesc carries the arguments of the
 * entry points that take scalars (slot assignment listed per op).
 * ---------------------------------------------------------------------------------- */
typedef struct DykMiscDesc {
    void* p[6];
    int64_t n;
    int32_t i[12];
    float f[4];
} DykMiscDesc;

enum {
    DYK_OP_CONV = 1,            /* DykConvDesc        -> dyk_conv_igemm */
    DYK_OP_WGRAD = 2,           /* DykWgradDesc       -> dyk_conv_wgrad */
    DYK_OP_BN_FINALIZE = 3,     /* DykBnFinalizeDesc  -> dyk_bn_finalize */
    DYK_OP_BN_ACT_FWD = 4,      /* DykEwDesc */
    DYK_OP_BN_BWD_REDUCE = 5,   /* DykEwDesc */
    DYK_OP_BN_BWD_APPLY = 6,    /* DykEwDesc */
    DYK_OP_AXPBY = 7,           /* DykEwDesc */
    DYK_OP_DOT = 8,             /* DykEwDesc */
    DYK_OP_UPSAMPLE_FWD = 9,    /* DykEwDesc */
    DYK_OP_UPSAMPLE_BWD = 10,   /* DykEwDesc */
    DYK_OP_MAXPOOL_FWD = 11,    /* DykEwDesc, aux = argmax */
    DYK_OP_MAXPOOL_BWD = 12,    /* DykEwDesc, aux = argmax */
    DYK_OP_SE_POOL = 13,        /* DykEwDesc, aux = pooled */
    DYK_OP_SE_FC_FWD = 14,      /* DykSeFcDesc */
    DYK_OP_SE_FC_BWD = 15,      /* DykSeFcDesc */
    DYK_OP_SE_SCALE = 16,       /* DykEwDesc */
    DYK_OP_BN_BWD_PARAMS = 17,  /* Misc: p0=red p1=dgamma p2=dbeta i0=C i1=slots */
    DYK_OP_BN_FOLD = 18,        /* Misc: p0=gamma p1=beta p2=rmean p3=rvar p4=scale p5=shift i0=C f0=eps */
    DYK_OP_WFUSE_WEIGHTS = 19,  /* Misc: p0=w p1=weff i0=n */
    DYK_OP_WFUSE_BWD_PARAMS = 20, /* Misc: p0=w p1=red p2=dw i0=n */
    DYK_OP_HEAD_PERMUTE_FWD = 21, /* Misc: p0=y p1=p i0=B i1=ny i2=nx i3=na i4=no i5=ld */
    DYK_OP_HEAD_PERMUTE_BWD = 22, /* Misc: p0=dp p1=dy p2=dbias i0..i5 as fwd, i6=dtype */
    DYK_OP_PATCH_GATHER = 23,   /* Misc: p0=in p1=out i0=B i1=Cin i2=H i3=W i4=k i5=stride i6=pad i7=ld i8=dtype f0=mul */
    DYK_OP_MEMSET = 24,         /* Misc: p0=ptr n=bytes i0=value; optional second region p1=ptr i1=bytes (zeros) */
    DYK_OP_YOLO_DECODE = 25,    /* DykDecodeDesc */
    DYK_OP_DW_FWD = 26,         /* DykDwDesc -> dyk_dwconv_fwd */
    DYK_OP_DW_DGRAD = 27,       /* DykDwDesc -> dyk_dwconv_dgrad */
    DYK_OP_DW_WGRAD = 28,       /* DykDwDesc -> dyk_dwconv_wgrad */
    DYK_OP_CAST_PAD_ROWS = 29,  /* Misc: p0=src p1=dst i0=R i1=C i2=Cpad i3=dtype */
    DYK_OP_BN_FWD_FUSED = 30,   /* Misc: p0=DykBnFinalizeDesc* p1=DykEwDesc* -> dyk_bn_finalize_act_fwd */
    DYK_OP_GRAD_REDUCE = 31,    /* Misc: p0=G p1=part p2=table i0=n_entries i1=total_chunks -> dyk_grad_reduce */
    DYK_OP_STEM_FWD = 32,       /* DykStemDesc -> dyk_stem_conv_fwd */
    DYK_OP_STEM_WGRAD = 33,     /* DykStemDesc -> dyk_stem_conv_wgrad */
    DYK_OP_COUNT_
};

typedef struct DykCommand {
    int32_t op;
    int32_t lane;         /* scheduling hints for dyk_run_commands_overlap (ignored by dyk_run_commands):
                             bit 0 = belongs to the second, independent branch (the LWIR backbone of a dual-stream
                             net); bit 1 = fork point (the branch may start once everything before this command is
                             done); bit 2 = join (this command needs both branches); bit 3 = a weight-gradient command that
                             stays on its own stream (the tail of a backward list: balances the side stream) */
    const void* desc;
} DykCommand;

/* One two-problem launch of commands a and b (same op, descriptors equal in every non-pointer field; see DykConvDesc.twin).
 * Ops: DYK_OP_CONV, DYK_OP_WGRAD, DYK_OP_BN_FINALIZE, DYK_OP_BN_ACT_FWD, DYK_OP_BN_FWD_FUSED, DYK_OP_BN_BWD_REDUCE,
 * DYK_OP_BN_BWD_APPLY, DYK_OP_AXPBY; DYK_ERR_UNSUPPORTED otherwise, DYK_ERR_ARG when the two descriptors differ in shape. */
int dyk_run_command_pair(const DykCommand* a, const DykCommand* b, void* stream);

/* Enqueue cmds[0..n) in order.  Stops at the first failure and returns its code; *failed_index
 * (may be NULL) receives the index of the failing command. */
int dyk_run_commands(const DykCommand* cmds, int32_t n, void* stream, int32_t* failed_index);
/* Same result, more concurrency: commands tagged as the second branch (DykCommand.lane) run on a library-owned
 * stream between their fork and join points, and weight-gradient commands (DYK_OP_WGRAD, DYK_OP_DW_WGRAD; nothing
 * in a backward pass reads dW) on another one, each behind an event that covers everything enqueued before it on
 * the issuing stream.  `stream` waits for both side streams before the call returns (no host synchronisation). */
int dyk_run_commands_overlap(const DykCommand* cmds, int32_t n, void* stream, int32_t* failed_index);

/* ------------------------------------------------------------------------------------
 * YOLOLayer inference decode (models.py:234-258): p [B,na,ny,nx,no] fp32 raw logits ->
 * io rows [row_offset + (a*ny + y)*nx + x] of a [B, rows_total, no] fp32 buffer:
 *   v3: xy = (sigmoid(t) + grid)*stride, wh = exp(t)*anchor_vec*stride, rest = sigmoid
 *   v4: s = sigmoid(t); xy = (s*2 - 0.5 + grid)*stride, wh = (s*2)^2*anchor_vec*stride, rest = s
 * ---------------------------------------------------------------------------------- */
typedef struct DykDecodeDesc {
    const float* p;
    float* io;
    int32_t B, na, ny, nx, no;
    int32_t rows_total, row_offset;
    int32_t v4;
    float stride;
    float anchor_vec[2 * 8];        /* (w, h) per anchor, already divided by stride (models.py:182) */
} DykDecodeDesc;
int dyk_yolo_decode(const DykDecodeDesc* desc, void* stream);

/* ------------------------------------------------------------------------------------
 * Target assignment and loss (build_utils/utils.py:209-384).
 * dyk_build_targets: for every head h and every (anchor a, target t) pair in anchor-major order,
 *   keep the pair iff wh_iou(anchor_vec[h][a], (w,h)*grid) > iou_t and emit
 *   indices[h][0..3][m] = (image, anchor, gj, gi) (int64, truncation, no clamping), tbox[h][m] =
 *   (gx-gi, gy-gj, gw, gh), anch[h][m], tcls[h][m]; counts[h] = number of matches.  Arrays are
 *   sized for cap = na*nt matches per head: indices [nheads][4][cap], tbox [nheads][cap][4],
 *   anch [nheads][cap][2], tcls [nheads][cap].  targets is [nt][6] = (image, class, xc, yc, w, h).
 * dyk_yolo_loss: runs the assignment, then the CIoU/GIoU box loss, the objectness BCE over all
 *   cells (targets scattered with last-match-wins) and the class BCE (nc > 1), writes
 *   out[0..2] = (box, obj, cls) * hyp gains and the gradient of each term w.r.t. the logits into
 *   dp[h] (box -> channels 0..3, obj -> 4, cls -> 5..): the three terms touch disjoint channels, so
 *   dyk_loss_scale_grads can apply the three upstream gradients afterwards.
 *   *flag gets bit 0 set if a target indexed outside the grid (the reference raises IndexError).
 * ---------------------------------------------------------------------------------- */
typedef struct DykTargetsDesc {
    const float* targets;
    int32_t nt, nheads, na;
    int32_t ny[3], nx[3];
    float anchor_vec[3][16];
    float iou_t;
    int32_t* counts;       /* [nheads] */
    int64_t* indices;
    float* tbox;
    float* anch;
    int64_t* tcls;
} DykTargetsDesc;

typedef struct DykLossDesc {
    const float* p[3];     /* [B][na][ny][nx][no] raw logits */
    float* dp[3];          /* same shape, out */
    float* tobj[3];        /* [B][na][ny][nx] scratch */
    int32_t nheads, B, no, nc;
    int32_t v4;            /* box parameterisation: 'yolov4' in cfg (utils.py:252) */
    int32_t ciou;          /* 'ciou' in hyp (utils.py:264), else GIoU */
    float hyp_box, hyp_obj, hyp_cls, cls_pw, obj_pw, gr;
    float fl_gamma;        /* hyp['fl_gamma'] > 0: both BCE terms wrapped in FocalLoss (utils.py:174-201, :236-238); 0 = plain BCE */
    float fl_alpha;        /* FocalLoss alpha (utils.py:176 default 0.25; read only when fl_gamma > 0) */
    double* acc;           /* [12] scratch */
    float* out;            /* [3] */
    int32_t* flag;         /* bit 0 is SET when a target falls outside the grid; dyk_yolo_loss clears it first, like dp / tobj /
                              acc.  Two back-to-back layouts get ONE fill for everything: acc | flag | 4 spare bytes | dp[0] |
                              dp[1] | .. | tobj[0] | ..  (acc leads: 8-byte aligned for any grid) and dp[0] | .. | tobj[..] | acc
                              | flag; any other layout takes one fill per buffer */
} DykLossDesc;

int dyk_build_targets(const DykTargetsDesc* desc, void* stream);
int dyk_yolo_loss(const DykLossDesc* desc, const DykTargetsDesc* targets, void* stream);
int dyk_loss_scale_grads(float* dp, int64_t n, int32_t no, const float* g3, void* stream);
/* The same with the three upstream gradients as separate device scalars (NULL = 0: the term is not in the differentiated sum) --
 * autograd hands them over separately (reference utils.py:287-293 returns three tensors), no concatenation launch; dp may be the
 * back-to-back gradient buffers of all heads (one launch). */
int dyk_loss_scale_grads3(float* dp, int64_t n, int32_t no, const float* gbox, const float* gobj, const float* gcls, void* stream);

/* ------------------------------------------------------------------------------------
 * non_max_suppression (build_utils/utils.py:387-464) incl. torchvision.ops.nms (:448), one
 * workgroup per image.  pred [B][N][no] decoded boxes (cx,cy,w,h,obj,cls...).  Per image b:
 *   counts[b] = k <= max_num, out[b][0..k) = (x1,y1,x2,y2,conf,cls) in descending-score order,
 *   out_rows[b][0..k) = row of pred each detection came from.
 * Semantics: strict `>` thresholds, 2 < w,h < 4096, conf = obj*cls, best class unless
 * multi_label (and nc > 1), optional class filter, per-class offset 4096 unless agnostic, greedy
 * suppression of IoU > iou_thres in stable descending score order, first max_num kept.  The
 * reference's 10 s wall-clock bail-out (:461-462) is deliberately not reproduced.
 * ws: B * ws_per_image bytes, ws_per_image >= dyk_nms_workspace_bytes(N, no, multi_label).
 * ---------------------------------------------------------------------------------- */
typedef struct DykNmsDesc {
    const float* pred;
    float* out;            /* [B][max_num][6] */
    int32_t* out_rows;     /* [B][max_num] */
    int32_t* counts;       /* [B] */
    void* ws;
    int64_t ws_per_image;
    int32_t B, N, no;
    float conf_thres, iou_thres;
    int32_t multi_label, agnostic, max_num;
    int32_t n_classes;     /* 0 = no class filter */
    int32_t classes[16];
} DykNmsDesc;
int64_t dyk_nms_workspace_bytes(int32_t N, int32_t no, int32_t multi_label);
int dyk_nms(const DykNmsDesc* desc, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused optimizer step over the flat parameter store (replaces the per-tensor torch.optim.Adam /
 * optim.SGD(nesterov=True) updates of reference train.py:85-91; same update rule incl. L2
 * weight_decay added to the gradient).  n must be a multiple of 4, pointers 16-byte aligned.
 *   grad_scale : multiplies the gradient first (1/world_size after an all-reduce SUM, 1/loss_scale)
 *   wc         : if not NULL, also writes the bf16 copy of the updated parameters (same offsets)
 *   zero_grad  : if non-zero, clears g after use (next backward accumulates into zeros)
 *   mask       : if not NULL, one byte per element, 0 = frozen: no update, no decay, moments untouched (the
 *                reference builds its optimizer from `p.requires_grad` parameters only, train.py:84); a parameter's
 *                elements start at multiples of 64, so four consecutive bytes always agree
 * dyk_adam_step : m = exp_avg, v = exp_avg_sq, beta1/beta2/eps, step = 1-based step count
 * dyk_sgd_step  : m = momentum buffer, beta1 = momentum, Nesterov, dampening 0 (v, beta2, eps unused)
 * ---------------------------------------------------------------------------------- */
typedef struct DykOptimDesc {
    float* p;
    float* g;
    float* m;
    float* v;
    void* wc;
    const uint8_t* mask;
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay, grad_scale;
    int32_t step;
    int32_t zero_grad;
} DykOptimDesc;
int dyk_adam_step(const DykOptimDesc* desc, void* stream);
int dyk_sgd_step(const DykOptimDesc* desc, void* stream);

/* ------------------------------------------------------------------------------------
 * Stem convolution (Cin = 3, 3x3, pad 1, stride 1 | 2, Cout 16 | 32) straight from the image batch, no im2col.
 * Replaces `imgs.float() / 255.0` (train_utils/kaist_train_eval_utils.py:54-55, evaluate.py:67-68) + nn.Conv2d(3, C, 3)
 * of module_list[0] / module_list[second_index] (models.py:34-42) and that layer's weight gradient.
 *   img  : [B][3][H][W] NCHW, float32 in 0..1, or uint8 (in_u8 != 0: each value is divided by 255.0f on the fly, the
 *          same fp32 quotient the reference's `.float() / 255.0` produces)
 *   fwd  : y[b, yo, xo, co] = sum_{ky,kx,c} img[b, c, yo*stride-1+ky, xo*stride-1+kx] * w[co][ky][kx][c]  (fp32 FMAs),
 *          wt = the same weights transposed to [27][Cout] fp32; raw output + per-channel sum / sum of squares into a
 *          `stats` replica (training), or act(y*scale + shift) when scale != NULL (eval: folded BatchNorm)
 *   wgrad: dw[co][(ky*3+kx)*3 + c] += sum over pixels dy * img, on the fp32-input MFMA; `part` = workspace of
 *          dyk_stem_wgrad_planes(desc) * Cout * 27 floats (per-wave partial tiles, folded in a fixed order)
 * ---------------------------------------------------------------------------------- */
typedef struct DykStemDesc {
    const void* img;
    const float* wt;       /* [27][Cout] (forward) */
    void* y;               /* forward output, dtype, rows of ldy elements */
    double* stats;         /* [stats_slots][2*Cout] or NULL */
    const float* scale;    /* [Cout] or NULL */
    const float* shift;    /* [Cout] or NULL */
    const void* dy;        /* wgrad: gradient wrt y, dtype, rows of lddy elements */
    float* dw;             /* wgrad: [Cout][27], accumulated */
    float* part;           /* wgrad workspace */
    /* wgrad with bn_fused != 0 (uint8 images, bf16): the BatchNorm-backward APPLY pass of the stem's own BatchNorm is done on
     * the fly instead of by a separate dyk_bn_act_bwd_apply launch -- `dy` is not read; the kernel reads da (the gradient wrt the
     * normalised output with act' already applied, as a fused data-gradient epilogue leaves it) and the raw conv output, folds
     * the two reduction sums from the replicas and feeds  dz = scale * (da - S1/N - xhat * S2/N),  xhat = (yraw - mean) * rstd,
     * rounded to bf16 as the apply pass would store it, to the weight gradient; workgroup 0 adds S2 / S1 to dgamma / dbeta.
     * Saves one pass over the largest activation of the net (512 x 640 x 32 per image): written once less, read once less. */
    const void* bn_da;     /* [B,Ho,Wo,Cout] dtype, rows of Cout elements */
    const void* bn_yraw;   /* raw conv output, same layout */
    const float* bn_vecs;  /* scale | shift | mean | rstd, Cout floats each */
    const double* bn_red;  /* [bn_slots][2][Cout]: sum(da), sum(da * xhat) replicas */
    float* bn_dgamma;      /* += S2 (or NULL) */
    float* bn_dbeta;       /* += S1 (or NULL) */
    int32_t dtype, in_u8;
    int32_t B, H, W, Cout, k, stride, pad, Ho, Wo;
    int32_t ldy, lddy, act, stats_slots;
    int32_t bn_fused, bn_slots;
} DykStemDesc;
int dyk_stem_conv_fwd(const DykStemDesc* desc, void* stream);
int dyk_stem_conv_wgrad(const DykStemDesc* desc, void* stream);
int dyk_stem_wgrad_planes(const DykStemDesc* desc);
/* 1 when dyk_stem_conv_wgrad would accept `desc` with bn_fused set (its bn_* fields and in_u8 / img filled in): the caller then
 * sets bn_fused and DYK_EW_SKIP on the BatchNorm's own dyk_bn_act_bwd_apply descriptor; 0: run the two passes separately. */
int dyk_stem_wgrad_bn_fusable(const DykStemDesc* desc);

/* Dependency-scheduled execution on several HIP streams.  The plan compiler derives the read / write sets of every
 * command from its descriptor, builds the dependency graph and list-schedules it onto n_streams in-order streams
 * (dyk/sched.py); this entry point replays the result.  Entries are in ISSUE order.  Entry k launches command
 * cmds[sched[k].cmd] on stream sched[k].stream (0 = the caller's `stream`, 1.. = library-owned streams) after making
 * that stream wait for the completion events of the entries sched[k].wait[0..nwait) (positions < k, always on other
 * streams), and records a completion event afterwards when `record` is set.  On entry every library stream waits for
 * what the caller's stream holds so far; on return the caller's stream waits for all of them.  No host synchronisation.
 * low_priority_last != 0: the last stream is created with the lowest priority (filler work such as weight gradients). */
typedef struct DykSchedEntry {
    int32_t cmd;
    int16_t stream;
    int8_t nwait;
    int8_t record;
    int32_t wait[7];
    int32_t cmd2;         /* >= 0: cmds[cmd] and cmds[cmd2] are a shape-identical, mutually independent pair (the twin backbones):
                             enqueued as ONE two-problem launch (dyk_run_command_pair); < 0: none */
} DykSchedEntry;
int dyk_run_schedule(const DykCommand* cmds, const DykSchedEntry* sched, int32_t n_entries, int32_t n_streams,
                     int32_t low_priority_last, void* stream, int32_t* failed_index);
/* A command range as a hipGraph built from dependency lists: command i depends on commands dep_idx[dep_off[i] ..
 * dep_off[i+1]) (all < i).  Each command is captured alone (single-stream capture) into a child graph node of the master
 * graph; the instantiated graph is launched with one host call per pass.  Kernel arguments -- every pointer inside the
 * descriptors -- are frozen at capture; capture again when one changes.  *graph_out is an opaque handle. */
int dyk_dag_graph_create(const DykCommand* cmds, int32_t n, const int32_t* dep_off, const int32_t* dep_idx,
                         void** graph_out, int32_t* failed_index);
int dyk_schedule_graph_launch(void* graph, void* stream);
int dyk_schedule_graph_destroy(void* graph);

/* ------------------------------------------------------------------------------------
 * Box-coordinate helpers of the evaluation chain (build_utils/utils.py:40-92; callers evaluate.py:82,
 * detect.py:114).  Rows of `ld` floats, the first four columns are the box.  fp32, rounded exactly like the
 * reference's torch-CPU expressions.
 *   dyk_box_convert : to_xyxy != 0: (xc,yc,w,h) -> (x1,y1,x2,y2) [xywh2xyxy :50-57]; else the inverse
 *                     [xyxy2xywh :40-47].  in may equal out.
 *   dyk_scale_coords: in place.  do_scale != 0: x = (x - pad_x) / gain, y = (y - pad_y) / gain [scale_coords
 *                     :60-81] and then, always, clamp x to [0, w0], y to [0, h0] [clip_coords :84-92].
 * ---------------------------------------------------------------------------------- */
int dyk_box_convert(const float* in, float* out, int32_t n, int32_t ld_in, int32_t ld_out, int32_t to_xyxy, void* stream);
int dyk_scale_coords(float* boxes, int32_t n, int32_t ld, float pad_x, float pad_y, float gain, float w0, float h0,
                     int32_t do_scale, void* stream);

/* Profiling variant of dyk_run_commands: brackets every command with HIP events on `stream`
 * and, after synchronising the stream, writes each command's duration in milliseconds to
 * ms_out[0..n).  Used by bench.py's roofline pass, never in a timed throughput region. */
int dyk_run_commands_timed(const DykCommand* cmds, int32_t n, void* stream, float* ms_out_host);
/* The same for the entries of a schedule, in issue order on ONE stream (two-problem entries as the single launch they
 * are in the step): ms_out[k] = duration of entry k. */
int dyk_run_schedule_timed(const DykCommand* cmds, const DykSchedEntry* sched, int32_t n_entries, void* stream,
                           float* ms_out_host);
/* Library stream idx (1..7) of the current device's schedule runtime (the streams dyk_run_schedule replays on; created on
 * first use).  For host code with side-stream work of its own: the HIP runtime maps all streams of a process onto four
 * hardware queues, so such work goes onto one of these rather than onto a fifth stream. */
int dyk_sched_stream(int32_t idx, void** stream_out);

#ifdef __cplusplus
}
#endif
#endif /* DYK_HIP_H */
