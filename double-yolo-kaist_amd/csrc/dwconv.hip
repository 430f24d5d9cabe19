// Depthwise convolution (groups == channels), forward / data gradient / weight gradient, channels-last.
// Replaces nn.Conv2d(groups=C) at reference models.py:41 (MobileNet [convolutional] sections with `groups=`)
// and the first conv of DepthwiseSeparableConv2d (layers.py:223-224).  4.5-12 flop/byte: HBM-bound, so these
// are streaming kernels (16-byte vectors of 8 channels, the k*k re-reads of a pixel neighbourhood come from
// L1/L2), not MFMA work.  Weights are read directly from the fp32 master copy in tap-major order [k*k][C].
// The forward optionally accumulates the per-channel sum / sum of squares for a following train-mode
// BatchNorm into the same replicated fp64 buffers the MFMA conv epilogue uses.
#include "dyk_common.h"

namespace {

// thread mapping: tx = channel vector in a group of CVB, ty = pixel lane; grid.x over channel-vector
// groups, grid.y strides over output pixels
template <typename T, bool GRAD>
__global__ __launch_bounds__(256) void dwconv_kernel(DykDwDesc d, int CVB) {
    constexpr int EPV = ElemTraits<T>::EPV;
    __shared__ float red[256 * 2 * 8];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int cv = blockIdx.x * CVB + tx;
    const int c = cv * EPV;
    const bool active = c < d.C;
    const T* __restrict__ x = GRAD ? (const T*)d.y : (const T*)d.x;     // tensor read
    T* __restrict__ y = GRAD ? (T*)d.x : (T*)d.y;                       // tensor written
    const int ld_src = GRAD ? d.ldy : d.ldx, ld_dst = GRAD ? d.ldx : d.ldy;
    const int k = d.k, s = d.stride, pad = d.pad;
    // forward: output grid (Ho, Wo), input (Hi, Wi).  GRAD: output grid is the *input* of the conv (Hi, Wi),
    // source is the output gradient (Ho, Wo): dx[yi] += dy[(yi + pad - kh)/s] * w[kh] when divisible.
    const int Hout = GRAD ? d.Hi : d.Ho, Wout = GRAD ? d.Wi : d.Wo;
    const int Hsrc = GRAD ? d.Ho : d.Hi, Wsrc = GRAD ? d.Wo : d.Wi;
    const long npix = (long)d.B * Hout * Wout;
    const bool accum = d.flags & DYK_EW_ACCUM;
    const bool stats = (!GRAD) && d.stats != nullptr;
    float s1[EPV], s2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s1[j] = s2[j] = 0.f;
    if (active) {
        for (long p = (long)blockIdx.y * PY + ty; p < npix; p += (long)gridDim.y * PY) {
            const int xo = (int)(p % Wout);
            const long q = p / Wout;
            const int yo = (int)(q % Hout);
            const int b = (int)(q / Hout);
            float acc[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
            for (int kh = 0; kh < k; ++kh) {
                int ys;
                if (GRAD) {
                    const int t = yo + pad - kh;
                    if (t < 0 || t % s) continue;
                    ys = t / s;
                } else {
                    ys = yo * s + kh - pad;
                }
                if (ys < 0 || ys >= Hsrc) continue;
                for (int kw = 0; kw < k; ++kw) {
                    int xs;
                    if (GRAD) {
                        const int t = xo + pad - kw;
                        if (t < 0 || t % s) continue;
                        xs = t / s;
                    } else {
                        xs = xo * s + kw - pad;
                    }
                    if (xs < 0 || xs >= Wsrc) continue;
                    float xv[EPV];
                    vec_unpack<T>(*(const uint4*)(x + (((long)b * Hsrc + ys) * Wsrc + xs) * ld_src + c), xv);
                    const float* wp = d.w + (long)(kh * k + kw) * d.C + c;
                    const float4 w0 = *(const float4*)wp;
                    float wv[8] = {w0.x, w0.y, w0.z, w0.w, 0.f, 0.f, 0.f, 0.f};
                    if (EPV == 8) {
                        const float4 w1 = *(const float4*)(wp + 4);
                        wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
                    }
#pragma unroll
                    for (int j = 0; j < EPV; ++j) acc[j] += xv[j] * wv[j];
                }
            }
            if (stats) {
#pragma unroll
                for (int j = 0; j < EPV; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
            }
            T* yp = y + p * ld_dst + c;
            if (accum) {
                float old[EPV];
                vec_unpack<T>(*(const uint4*)yp, old);
#pragma unroll
                for (int j = 0; j < EPV; ++j) acc[j] += old[j];
            }
            *(uint4*)yp = vec_pack<T>(acc);
        }
    }
    if (stats) {
        float* mine = red + threadIdx.x * 16;
#pragma unroll
        for (int j = 0; j < EPV; ++j) { mine[j] = s1[j]; mine[8 + j] = s2[j]; }
        __syncthreads();
        if (ty == 0 && active) {
            for (int q = 1; q < PY; ++q) {
                const float* o = red + (q * CVB + tx) * 16;
#pragma unroll
                for (int j = 0; j < EPV; ++j) { s1[j] += o[j]; s2[j] += o[8 + j]; }
            }
            double* st = d.stats + (size_t)(blockIdx.y % (unsigned)(d.stats_slots > 0 ? d.stats_slots : 1)) * 2 * d.C;
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                atomicAdd(st + c + j, (double)s1[j]);
                atomicAdd(st + d.C + c + j, (double)s2[j]);
            }
        }
    }
}

// dw[t][c] += sum_p dy[p][c] * x[src(p, t)][c];  grid.z = tap
template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DykDwDesc d, int CVB) {
    constexpr int EPV = ElemTraits<T>::EPV;
    __shared__ float red[256 * 8];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int cv = blockIdx.x * CVB + tx;
    const int c = cv * EPV;
    const bool active = c < d.C;
    const int tap = blockIdx.z;
    const int kh = tap / d.k, kw = tap - kh * d.k;
    const T* __restrict__ x = (const T*)d.x;
    const T* __restrict__ dy = (const T*)d.y;       // output gradient [B,Ho,Wo,C]
    const long npix = (long)d.B * d.Ho * d.Wo;
    float acc[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
    if (active) {
        for (long p = (long)blockIdx.y * PY + ty; p < npix; p += (long)gridDim.y * PY) {
            const int xo = (int)(p % d.Wo);
            const long q = p / d.Wo;
            const int yo = (int)(q % d.Ho);
            const int b = (int)(q / d.Ho);
            const int yi = yo * d.stride + kh - d.pad, xi = xo * d.stride + kw - d.pad;
            if (yi < 0 || yi >= d.Hi || xi < 0 || xi >= d.Wi) continue;
            float g[EPV], xv[EPV];
            vec_unpack<T>(*(const uint4*)(dy + p * d.ldy + c), g);
            vec_unpack<T>(*(const uint4*)(x + (((long)b * d.Hi + yi) * d.Wi + xi) * d.ldx + c), xv);
#pragma unroll
            for (int j = 0; j < EPV; ++j) acc[j] += g[j] * xv[j];
        }
    }
    float* mine = red + threadIdx.x * 8;
#pragma unroll
    for (int j = 0; j < EPV; ++j) mine[j] = acc[j];
    __syncthreads();
    if (ty == 0 && active) {
        for (int q = 1; q < PY; ++q) {
            const float* o = red + (q * CVB + tx) * 8;
#pragma unroll
            for (int j = 0; j < EPV; ++j) acc[j] += o[j];
        }
#pragma unroll
        for (int j = 0; j < EPV; ++j) unsafeAtomicAdd(d.dw + (long)tap * d.C + c + j, acc[j]);
    }
}

int check_dw(const DykDwDesc* d) {
    if (!d || !d->x || !d->y || d->B <= 0 || d->C <= 0 || d->k <= 0 || d->k > 7 || d->stride <= 0) return DYK_ERR_ARG;
    if (d->dtype != DYK_BF16 && d->dtype != DYK_F32) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    if (d->C % epv || d->ldx % epv || d->ldy % epv) return DYK_ERR_ARG;
    if (d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0) return DYK_ERR_ARG;
    return DYK_OK;
}

inline int grid2d(int CV, long npix, int* gx, int* gy, int per_thread, int cap_blocks) {
    int CVB = 1;
    while (CVB < CV && CVB < 32) CVB <<= 1;
    const int PY = 256 / CVB;
    *gx = (CV + CVB - 1) / CVB;
    long g = (npix + (long)PY * per_thread - 1) / ((long)PY * per_thread);
    const long cap = cap_blocks / *gx > 0 ? cap_blocks / *gx : 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    *gy = (int)g;
    return CVB;
}

}  // namespace

extern "C" int dyk_dwconv_fwd(const DykDwDesc* d, void* stream) {
    const int rc = check_dw(d);
    if (rc) return rc;
    if (!d->w) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    int gx, gy;
    const int CVB = grid2d(d->C / epv, (long)d->B * d->Ho * d->Wo, &gx, &gy, 2, 4096);
    if (d->dtype == DYK_BF16)
        hipLaunchKernelGGL((dwconv_kernel<bf16_t, false>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    else
        hipLaunchKernelGGL((dwconv_kernel<float, false>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_dwconv_dgrad(const DykDwDesc* d, void* stream) {
    const int rc = check_dw(d);
    if (rc) return rc;
    if (!d->w) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    int gx, gy;
    const int CVB = grid2d(d->C / epv, (long)d->B * d->Hi * d->Wi, &gx, &gy, 2, 4096);
    if (d->dtype == DYK_BF16)
        hipLaunchKernelGGL((dwconv_kernel<bf16_t, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    else
        hipLaunchKernelGGL((dwconv_kernel<float, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_dwconv_wgrad(const DykDwDesc* d, void* stream) {
    const int rc = check_dw(d);
    if (rc) return rc;
    if (!d->dw) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    int gx, gy;
    const int CVB = grid2d(d->C / epv, (long)d->B * d->Ho * d->Wo, &gx, &gy, 16, 512);
    if (d->dtype == DYK_BF16)
        hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16_t>, dim3(gx, gy, d->k * d->k), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    else
        hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, dim3(gx, gy, d->k * d->k), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
