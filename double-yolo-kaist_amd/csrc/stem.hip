// The Cin = 3 stem convolutions (first layer of each backbone: 3x3, pad 1, stride 1 | 2, 16 | 32 filters, + BatchNorm)
// straight from the image batch -- no im2col.
//
// Replaces: `imgs.float() / 255.0` (reference train_utils/kaist_train_eval_utils.py:54-55, evaluate.py:67-68) followed
// by nn.Conv2d(3, C, 3) of module_list[0] / module_list[second_index] (models.py:34-42), and the weight gradient of
// that layer in autograd's backward.  (The stem has no data gradient: nothing upstream is trainable.)
//
// Why not the MFMA implicit GEMM: K = 27.  Round 1 gathered 27-element patches into 32-channel rows (887 MB written and
// read back per launch at the BASELINE size) to feed a 1x1 MFMA GEMM; the layer is bound by its 335 MB output, not by
// 9 GFLOP.  Here:
//   forward   one thread = one output pixel: its 27 inputs come from the NCHW image (fp32, or uint8 divided by 255
//             exactly as the reference does), the 27 x C weights stream through SGPRs (transposed copy [27][C], scalar
//             loads), C fp32 accumulators, one 2*C-byte channels-last store; BatchNorm statistics wave -> block ->
//             one fp64 atomic per channel into a replica.  HBM-bound: image once (L1/L2 serve the 3x3 overlap),
//             output once.
//   wgrad     D[co][tc] += dy[p][co] * patch[p][tc] over all pixels on the matrix cores with the fp32-input MFMA
//             v_mfma_f32_32x32x2_f32 (exact fp32, A = one dy value per lane, B = one image value per lane: both in
//             their natural memory order -- no LDS, no transposition); every wave owns a fixed pixel range and a plane
//             of partial sums, a second tiny launch folds the planes in a fixed order (bit-reproducible, no atomics).
#include "dyk_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <bool U8> __device__ inline float load_px(const void* img, long idx) {
    if (U8) return (float)((const uint8_t*)img)[idx] / 255.0f;      // == uint8 -> float -> / 255.0 of the reference, bit for bit
    return ((const float*)img)[idx];
}

// -------------------------------------------------------------------------------------------------- forward
template <typename T, int COUT, bool U8>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const DykStemDesc d, const float* __restrict__ wt) {
    // wt: [27][COUT] as a `const __restrict__` kernel argument -- uniform addresses of provably read-only memory become
    // scalar loads (s_load_dwordx16 into SGPRs, one SGPR operand per FMA); read through the by-value descriptor they
    // were 216 vector loads of one address per pixel and the kernel ran at 1.1 ms instead of ~0.1
    __shared__ float s_red[4][2 * COUT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int HoWo = d.Ho * d.Wo, npix = d.B * HoWo;
    const long plane = (long)d.H * d.W;
    const bool stats = d.stats != nullptr;
    const bool affine = d.scale != nullptr;
    float s1[COUT], s2[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) s1[c] = s2[c] = 0.f;
    for (int p = blockIdx.x * 256 + tid; p < npix; p += gridDim.x * 256) {
        const int b = p / HoWo;
        const int r = p - b * HoWo;
        const int yo = r / d.Wo, xo = r - yo * d.Wo;
        const int y0 = yo * d.stride - 1, x0 = xo * d.stride - 1;
        float x[27];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yi = y0 + ky, xi = x0 + kx;
                const bool in = (unsigned)yi < (unsigned)d.H && (unsigned)xi < (unsigned)d.W;
                const long off = (long)b * 3 * plane + (long)(in ? yi : 0) * d.W + (in ? xi : 0);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = load_px<U8>(d.img, off + c * plane);
                    x[(ky * 3 + kx) * 3 + c] = in ? v : 0.f;
                }
            }
        float acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
        // the weights are re-read (scalar cache) for every pixel: hoisted out of the pixel loop all 27*COUT of them would
        // live in SGPRs and spill into VGPR lanes (1600 v_readlane per pixel)
        uintptr_t wa = (uintptr_t)wt;
        asm volatile("" : "+s"(wa));
        const __attribute__((address_space(4))) float* w = (const __attribute__((address_space(4))) float*)wa;   // constant address space: s_load
#pragma unroll
        for (int t = 0; t < 27; ++t)
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[c] = fmaf(x[t], w[t * COUT + c], acc[c]);
        if (stats) {
#pragma unroll
            for (int c = 0; c < COUT; ++c) { s1[c] += acc[c]; s2[c] += acc[c] * acc[c]; }
        }
        if (affine) {                            // eval: folded BatchNorm + activation (one switch, not one per channel)
            switch (d.act) {
#define STEM_ACT_CASE(A) case A: _Pragma("unroll") for (int c = 0; c < COUT; ++c) acc[c] = act_fwd_c<A>(acc[c] * d.scale[c] + (d.shift ? d.shift[c] : 0.f), A); break;
                STEM_ACT_CASE(DYK_ACT_LINEAR)
                STEM_ACT_CASE(DYK_ACT_LEAKY)
                STEM_ACT_CASE(DYK_ACT_MISH)
#undef STEM_ACT_CASE
            default:
                for (int c = 0; c < COUT; ++c) acc[c] = act_fwd(d.act, acc[c] * d.scale[c] + (d.shift ? d.shift[c] : 0.f));
            }
        }
        if (sizeof(T) == 2) {
            bf16_t* yp = (bf16_t*)d.y + (long)p * d.ldy;
#pragma unroll
            for (int c = 0; c < COUT; c += 8) {
                uint4 pk;
                pk.x = f32x2_to_bf16x2(acc[c + 0], acc[c + 1]); pk.y = f32x2_to_bf16x2(acc[c + 2], acc[c + 3]);
                pk.z = f32x2_to_bf16x2(acc[c + 4], acc[c + 5]); pk.w = f32x2_to_bf16x2(acc[c + 6], acc[c + 7]);
                *(uint4*)(yp + c) = pk;
            }
        } else {
            float* yp = (float*)d.y + (long)p * d.ldy;
#pragma unroll
            for (int c = 0; c < COUT; c += 4) *(float4*)(yp + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
        }
    }
    if (stats) {
        // thread -> wave (xor shuffles, fixed tree) -> block (wave order) -> ONE fp64 atomic per channel and block
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            float a = s1[c], q = s2[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
            if (lane == 0) { s_red[wid][c] = a; s_red[wid][COUT + c] = q; }
        }
        __syncthreads();
        if (tid < 2 * COUT) {
            const float tot = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
            const int slots = d.stats_slots > 0 ? d.stats_slots : 1;
            double* st = d.stats + (size_t)(blockIdx.x % (unsigned)slots) * 2 * COUT;
            atomicAdd(st + tid, (double)tot);          // [0, COUT): sums, [COUT, 2 COUT): sums of squares
        }
    }
}

// -------------------------------------------------------------------------------------------------- weight gradient
// One wave = one 32 x 32 fp32 accumulator tile D[co][tc] (16 registers per lane) over its own pixel range.
// v_mfma_f32_32x32x2_f32: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]; here i = output channel,
// j = patch element tc = (ky*3 + kx)*3 + c (27 used), k = pixel (two per instruction).
template <typename T, int COUT, bool U8>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const DykStemDesc d, int pix_per_wave) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave = blockIdx.x * 4 + wid;
    const int HoWo = d.Ho * d.Wo, npix = d.B * HoWo;
    const long plane = (long)d.H * d.W;
    const int i = lane & 31, kk = lane >> 5;
    // B operand: which image element this lane fetches relative to the pixel
    const int tc = i;
    const int c = tc % 3, kx = (tc / 3) % 3, ky = tc / 9;
    const bool tc_ok = tc < 27;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int p_begin = wave * pix_per_wave;
    int p_end = p_begin + pix_per_wave;
    if (p_end > npix) p_end = npix;
    const T* __restrict__ dy = (const T*)d.dy;
    // this lane's pixel walks p_begin + kk, +2, +2, ...: coordinates are carried along (one division per wave, not per MFMA)
    int p = p_begin + kk;
    int b = p / HoWo, r0 = p - b * HoWo;
    int yo = r0 / d.Wo, xo = r0 - yo * d.Wo;
    const T* dyp = dy + (long)p * d.lddy + i;
    const long dy_step = 2L * d.lddy;
    for (int p0 = p_begin; p0 < p_end; p0 += 2, p += 2) {         // wave-uniform trip count: the MFMA needs all 64 lanes
        const bool live = p < p_end;
        float a = 0.f, bv = 0.f;
        if (live && i < COUT) a = ElemTraits<T>::to_f32(*dyp);
        const int yi = yo * d.stride - 1 + ky, xi = xo * d.stride - 1 + kx;
        if (live && tc_ok && (unsigned)yi < (unsigned)d.H && (unsigned)xi < (unsigned)d.W)
            bv = load_px<U8>(d.img, ((long)b * 3 + c) * plane + (long)yi * d.W + xi);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
        dyp += dy_step;
        xo += 2;
        if (xo >= d.Wo) {
            xo -= d.Wo; ++yo;
            if (xo >= d.Wo) { xo -= d.Wo; ++yo; }          // Wo == 1
            if (yo >= d.Ho) { yo -= d.Ho; ++b; }
        }
    }
    // C/D: col = lane & 31 (tc), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (co)
    float* out = d.part + (size_t)wave * COUT * 27;
    if (tc_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (co < COUT) out[co * 27 + tc] = acc[r];
        }
    }
}

// dw[e] += sum over planes, fixed order: 8 lanes per element, 8 strided partial sums folded by xor shuffles
__global__ __launch_bounds__(256) void stem_wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int n, int planes) {
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    float s = 0.f;
    if (g < n)
        for (int q = sub; q < planes; q += 8) s += part[(size_t)q * n + g];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (g < n && sub == 0) dw[g] += s;
}

int check(const DykStemDesc* d) {
    if (!d || !d->img || d->B <= 0 || d->H <= 0 || d->W <= 0) return DYK_ERR_ARG;
    if (d->k != 3 || d->pad != 1 || (d->stride != 1 && d->stride != 2)) return DYK_ERR_UNSUPPORTED;
    if (d->Cout != 16 && d->Cout != 32) return DYK_ERR_UNSUPPORTED;
    if (d->dtype != DYK_BF16 && d->dtype != DYK_F32) return DYK_ERR_ARG;
    if ((long)d->B * d->H * d->W * 3 >= (1L << 31)) return DYK_ERR_ARG;       // 32-bit pixel indices
    if (d->Ho != (d->H + 2 - 3) / d->stride + 1 || d->Wo != (d->W + 2 - 3) / d->stride + 1) return DYK_ERR_ARG;
    return DYK_OK;
}

}  // namespace

#define STEM_DISPATCH(KERNEL, grid, ...)                                                                             \
    do {                                                                                                              \
        const bool u8 = d->in_u8 != 0;                                                                                \
        if (d->dtype == DYK_BF16) {                                                                                   \
            if (d->Cout == 32) { if (u8) hipLaunchKernelGGL((KERNEL<bf16_t, 32, true>), grid, dim3(256), 0, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<bf16_t, 32, false>), grid, dim3(256), 0, s, __VA_ARGS__); } \
            else               { if (u8) hipLaunchKernelGGL((KERNEL<bf16_t, 16, true>), grid, dim3(256), 0, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<bf16_t, 16, false>), grid, dim3(256), 0, s, __VA_ARGS__); } \
        } else {                                                                                                      \
            if (d->Cout == 32) { if (u8) hipLaunchKernelGGL((KERNEL<float, 32, true>), grid, dim3(256), 0, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<float, 32, false>), grid, dim3(256), 0, s, __VA_ARGS__); }   \
            else               { if (u8) hipLaunchKernelGGL((KERNEL<float, 16, true>), grid, dim3(256), 0, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<float, 16, false>), grid, dim3(256), 0, s, __VA_ARGS__); }   \
        }                                                                                                             \
    } while (0)

extern "C" int dyk_stem_conv_fwd(const DykStemDesc* d, void* stream) {
    const int rc = check(d);
    if (rc) return rc;
    if (!d->wt || !d->y || d->ldy < d->Cout || (d->ldy * (d->dtype == DYK_BF16 ? 2 : 4)) % 16 || ((uintptr_t)d->y % 16)) return DYK_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long npix = (long)d->B * d->Ho * d->Wo;
    long blocks = (npix + 255) / 256;
    // a fixed, bounded grid: every block folds its statistics once (2 * Cout atomics per block)
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks);
    STEM_DISPATCH(stem_fwd_kernel, grid, *d, d->wt);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

// number of partial planes (= waves) of the weight-gradient launch: the workspace `part` holds that many [Cout][27] tiles
extern "C" int dyk_stem_wgrad_planes(const DykStemDesc* d) {
    if (check(d)) return 0;
    const long npix = (long)d->B * d->Ho * d->Wo;
    long waves = (npix + 2047) / 2048;               // >= 2048 pixels (1024 MFMAs) per wave
    if (waves > 2048) waves = 2048;
    waves = (waves + 3) / 4 * 4;
    return (int)waves;
}

extern "C" int dyk_stem_conv_wgrad(const DykStemDesc* d, void* stream) {
    const int rc = check(d);
    if (rc) return rc;
    if (!d->dy || !d->dw || !d->part || d->lddy < d->Cout) return DYK_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long npix = (long)d->B * d->Ho * d->Wo;
    const int waves = dyk_stem_wgrad_planes(d);
    long ppw = (npix + waves - 1) / waves;
    ppw = (ppw + 1) / 2 * 2;
    const dim3 grid((unsigned)(waves / 4));
    STEM_DISPATCH(stem_wgrad_kernel, grid, *d, (int)ppw);
    DYK_LAUNCH_CHECK();
    const int n = d->Cout * 27;
    hipLaunchKernelGGL(stem_wgrad_fold_kernel, dim3((n * 8 + 255) / 256), dim3(256), 0, s, (const float*)d->part, d->dw, n, waves);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
