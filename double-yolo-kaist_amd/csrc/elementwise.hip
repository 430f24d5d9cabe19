// HBM-bound channels-last kernels around the convolutions: axpby (copy / add / weighted
// feature fusion), nearest 2x upsample, max-pool with arg-max, squeeze-excitation, YOLO head
// permute, first-layer patch gather.  Forward and backward of each.
#include "dyk_common.h"

namespace {

inline int ew_grid(long total_vec) {
    long g = (total_vec + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
inline int epv_of(int dtype) { return dtype == DYK_BF16 ? 8 : 4; }

// ------------------------------------------------------------------ axpby
// out = sa*a (+ sb*b), sa = alpha * (p0 ? p0[0] : 1), sb = beta * (p1 ? p1[0] : 1).  alpha == 0: `a` is NOT read (the term is an
// exact zero whatever the buffer holds: how a gradient slice nobody has written yet is cleared)
template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(DykEwPair pr) {
    const DykEwDesc d = pr.d[blockIdx.z];          // a COPY, not a reference (see DykEwPair in dyk_common.h)
    constexpr int EPV = ElemTraits<T>::EPV;
    const int CV = d.C / EPV;
    const long total = (long)d.npix * CV;
    const float sa = d.alpha * (d.p0 ? d.p0[0] : 1.f);
    const float sb = d.beta * (d.p1 ? d.p1[0] : 1.f);
    const T* __restrict__ a = (const T*)d.a;
    const T* __restrict__ b = (const T*)d.b;
    T* __restrict__ o = (T*)d.out;
    const bool accum = d.flags & DYK_EW_ACCUM;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
        const long p = v / CV;
        const int c = (int)(v - p * CV) * EPV;
        float x[EPV];
        if (d.alpha != 0.f) {
            vec_unpack<T>(*(const uint4*)(a + p * d.lda + c), x);
#pragma unroll
            for (int j = 0; j < EPV; ++j) x[j] *= sa;
        } else {
#pragma unroll
            for (int j = 0; j < EPV; ++j) x[j] = 0.f;
        }
        if (b) {
            float y[EPV];
            vec_unpack<T>(*(const uint4*)(b + p * d.ldb + c), y);
#pragma unroll
            for (int j = 0; j < EPV; ++j) x[j] += sb * y[j];
        }
        if (accum) {
            float y[EPV];
            vec_unpack<T>(*(const uint4*)(o + p * d.ldo + c), y);
#pragma unroll
            for (int j = 0; j < EPV; ++j) x[j] += y[j];
        }
        *(uint4*)(o + p * d.ldo + c) = vec_pack<T>(x);
    }
}

// red[0] += sum a*b  over all pixels/channels (weighted-fusion weight gradient).  With `out` (round 5): the same pass also
// leaves out = sa * a (+ out with DYK_EW_ACCUM), sa = alpha * (p0 ? p0[0] : 1) -- the scaled gradient copy dyk_axpby would make
// (same arithmetic, same rounding): the backward of one source of a weighted fusion in ONE pass over the gradient
template <typename T>
__global__ __launch_bounds__(256) void dot_kernel(DykEwDesc d) {
    constexpr int EPV = ElemTraits<T>::EPV;
    const int CV = d.C / EPV;
    const long total = (long)d.npix * CV;
    const T* __restrict__ a = (const T*)d.a;
    const T* __restrict__ b = (const T*)d.b;
    T* __restrict__ o = (T*)d.out;
    const float sa = d.alpha * (d.p0 ? d.p0[0] : 1.f);
    const bool accum = d.flags & DYK_EW_ACCUM;
    float s = 0.f;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
        const long p = v / CV;
        const int c = (int)(v - p * CV) * EPV;
        float x[EPV], y[EPV];
        vec_unpack<T>(*(const uint4*)(a + p * d.lda + c), x);
        vec_unpack<T>(*(const uint4*)(b + p * d.ldb + c), y);
#pragma unroll
        for (int j = 0; j < EPV; ++j) s += x[j] * y[j];
        if (o) {
#pragma unroll
            for (int j = 0; j < EPV; ++j) x[j] *= sa;
            if (accum) {
                float z[EPV];
                vec_unpack<T>(*(const uint4*)(o + p * d.ldo + c), z);
#pragma unroll
                for (int j = 0; j < EPV; ++j) x[j] += z[j];
            }
            *(uint4*)(o + p * d.ldo + c) = vec_pack<T>(x);
        }
    }
    __shared__ float ws[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(d.red, (double)(ws[0] + ws[1] + ws[2] + ws[3]));
}

// weighted feature fusion weights (layers.py:66): weff[i] = sigmoid(w[i]) * 2/n
__global__ void wfuse_weights_kernel(const float* w, float* weff, int n) {
    const int i = threadIdx.x;
    if (i < n) weff[i] = (1.f / (1.f + __expf(-w[i]))) * (2.f / n);
}
// dw[i] += red[i] * 2/n * s(1-s)
__global__ void wfuse_bwd_params_kernel(const float* w, const double* red, float* dw, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        const float s = 1.f / (1.f + __expf(-w[i]));
        dw[i] += (float)red[i] * (2.f / n) * s * (1.f - s);
    }
}

// ------------------------------------------------------------------ nearest 2x upsample
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(DykEwDesc d) {   // a: [B,H,W,C] -> out [B,2H,2W,C]
    constexpr int EPV = ElemTraits<T>::EPV;
    const int CV = d.C / EPV;
    const int Ho = 2 * d.H, Wo = 2 * d.W;
    const long total = (long)d.B * Ho * Wo * CV;
    const T* __restrict__ a = (const T*)d.a;
    T* __restrict__ o = (T*)d.out;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
        const long p = v / CV;
        const int c = (int)(v - p * CV) * EPV;
        const int xo = (int)(p % Wo);
        const long q = p / Wo;
        const int yo = (int)(q % Ho);
        const int b = (int)(q / Ho);
        const long pi = ((long)b * d.H + (yo >> 1)) * d.W + (xo >> 1);
        *(uint4*)(o + p * d.ldo + c) = *(const uint4*)(a + pi * d.lda + c);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(DykEwDesc d) {   // a: dout [B,2H,2W,C] -> out din [B,H,W,C]
    constexpr int EPV = ElemTraits<T>::EPV;
    const int CV = d.C / EPV;
    const int Wo = 2 * d.W;
    const long total = (long)d.B * d.H * d.W * CV;
    const T* __restrict__ a = (const T*)d.a;
    T* __restrict__ o = (T*)d.out;
    const bool accum = d.flags & DYK_EW_ACCUM;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
        const long p = v / CV;
        const int c = (int)(v - p * CV) * EPV;
        const int x = (int)(p % d.W);
        const long q = p / d.W;
        const int y = (int)(q % d.H);
        const int b = (int)(q / d.H);
        const long p00 = ((long)b * 2 * d.H + 2 * y) * Wo + 2 * x;
        float s[EPV], t[EPV];
        vec_unpack<T>(*(const uint4*)(a + p00 * d.lda + c), s);
        vec_unpack<T>(*(const uint4*)(a + (p00 + 1) * d.lda + c), t);
#pragma unroll
        for (int j = 0; j < EPV; ++j) s[j] += t[j];
        vec_unpack<T>(*(const uint4*)(a + (p00 + Wo) * d.lda + c), t);
#pragma unroll
        for (int j = 0; j < EPV; ++j) s[j] += t[j];
        vec_unpack<T>(*(const uint4*)(a + (p00 + Wo + 1) * d.lda + c), t);
#pragma unroll
        for (int j = 0; j < EPV; ++j) s[j] += t[j];
        if (accum) {
            vec_unpack<T>(*(const uint4*)(o + p * d.ldo + c), t);
#pragma unroll
            for (int j = 0; j < EPV; ++j) s[j] += t[j];
        }
        *(uint4*)(o + p * d.ldo + c) = vec_pack<T>(s);
    }
}

// ------------------------------------------------------------------ max pool k x k, stride 1, pad (k-1)/2
// idx (uint8, [npix][C]) holds the window position dy*k+dx of the first maximum in scan order
// (torch CPU max_pool2d keeps the first element for which val > max, so ties go to the earliest).
// K = compile-time window (0: runtime d.k).  The taps of one window row are loaded in one batch of K unconditional
// loads (clamped addresses, a validity bit per tap): a load behind a bounds branch is waited for on the spot, which made
// the 13 x 13 pool of the SPP block a chain of 169 dependent memory latencies (90 us for a 10 MB tensor).
template <typename T, int K>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(DykEwDesc d, uint8_t* __restrict__ idx) {
    // nn.MaxPool2d(k, stride, padding=(k-1)//2) (reference models.py:91-94); stride = d.slots (0 / 1: stride 1, the SPP pools)
    constexpr int EPV = ElemTraits<T>::EPV;
    constexpr int KB = K > 0 ? K : 1;
    const int CV = d.C / EPV;
    const int k = K > 0 ? K : d.k, pad = (k - 1) / 2, st = d.slots > 1 ? d.slots : 1;
    const int Ho = (d.H + 2 * pad - k) / st + 1, Wo = (d.W + 2 * pad - k) / st + 1;
    const long total = (long)d.B * Ho * Wo * CV;
    const T* __restrict__ a = (const T*)d.a;
    T* __restrict__ o = (T*)d.out;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
        const long p = v / CV;
        const int c = (int)(v - p * CV) * EPV;
        const int x = (int)(p % Wo);
        const long q = p / Wo;
        const int y = (int)(q % Ho);
        const int b = (int)(q / Ho);
        float m[EPV]; int mi[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) { m[j] = -INFINITY; mi[j] = 0; }
        bool first = true;
        for (int dy = 0; dy < k; ++dy) {
            const int yy = y * st + dy - pad;
            if (yy < 0 || yy >= d.H) continue;
            const T* row = a + ((long)b * d.H + yy) * d.W * d.lda + c;
            for (int dx0 = 0; dx0 < k; dx0 += KB) {
                uint4 raw[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int xx = x * st + dx0 + u - pad;
                    raw[u] = *(const uint4*)(row + (long)(xx < 0 ? 0 : (xx >= d.W ? d.W - 1 : xx)) * d.lda);
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int xx = x * st + dx0 + u - pad;
                    if (xx < 0 || xx >= d.W) continue;
                    float t[EPV];
                    vec_unpack<T>(raw[u], t);
#pragma unroll
                    for (int j = 0; j < EPV; ++j)
                        if (first || t[j] > m[j] || t[j] != t[j]) { m[j] = t[j]; mi[j] = dy * k + dx0 + u; }
                    first = false;
                }
            }
        }
        *(uint4*)(o + p * d.ldo + c) = vec_pack<T>(m);
        if (idx) {
#pragma unroll
            for (int j = 0; j < EPV; ++j) idx[p * d.C + c + j] = (uint8_t)mi[j];
        }
    }
}

template <typename T, int K>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(DykEwDesc d, const uint8_t* __restrict__ idx) {
    // a = dout [B,Ho,Wo,C], out = din [B,H,W,C]; every input pixel gathers from the outputs whose window holds it
    // (window rows in batches of K gradient + K argmax loads, as in the forward kernel)
    constexpr int EPV = ElemTraits<T>::EPV;
    constexpr int KB = K > 0 ? K : 1;
    const int CV = d.C / EPV;
    const int k = K > 0 ? K : d.k, pad = (k - 1) / 2, st = d.slots > 1 ? d.slots : 1;
    const int Ho = (d.H + 2 * pad - k) / st + 1, Wo = (d.W + 2 * pad - k) / st + 1;
    const long total = (long)d.B * d.H * d.W * CV;
    const T* __restrict__ a = (const T*)d.a;
    T* __restrict__ o = (T*)d.out;
    const bool accum = d.flags & DYK_EW_ACCUM;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
        const long p = v / CV;
        const int c = (int)(v - p * CV) * EPV;
        const int x = (int)(p % d.W);
        const long q = p / d.W;
        const int y = (int)(q % d.H);
        const int b = (int)(q / d.H);
        float s[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) s[j] = 0.f;
        for (int dy = 0; dy < k; ++dy) {
            const int ty = y - dy + pad;          // = yo * stride for the output row whose window position dy is this row
            if (ty < 0 || ty % st) continue;
            const int yo = ty / st;
            if (yo >= Ho) continue;
            const long prow = ((long)b * Ho + yo) * Wo;
            for (int dx0 = 0; dx0 < k; dx0 += KB) {
                uint4 graw[KB];
                uint32_t iraw[KB][EPV / 4];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int tx = x - (dx0 + u) + pad;
                    const int xo = tx / st;
                    const bool ok = tx >= 0 && tx % st == 0 && xo < Wo;
                    const long po = prow + (ok ? xo : 0);
                    graw[u] = *(const uint4*)(a + po * d.lda + c);
                    const uint32_t* ip = (const uint32_t*)(idx + po * d.C + c);
#pragma unroll
                    for (int w = 0; w < EPV / 4; ++w) iraw[u][w] = ip[w];
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int tx = x - (dx0 + u) + pad;
                    const int xo = tx / st;
                    if (!(tx >= 0 && tx % st == 0 && xo < Wo)) continue;
                    const uint32_t code = (uint32_t)(dy * k + dx0 + u);
                    float g[EPV];
                    vec_unpack<T>(graw[u], g);
#pragma unroll
                    for (int j = 0; j < EPV; ++j)
                        if (((iraw[u][j / 4] >> (8 * (j % 4))) & 0xffu) == code) s[j] += g[j];
                }
            }
        }
        if (accum) {
            float t[EPV];
            vec_unpack<T>(*(const uint4*)(o + p * d.ldo + c), t);
#pragma unroll
            for (int j = 0; j < EPV; ++j) s[j] += t[j];
        }
        *(uint4*)(o + p * d.ldo + c) = vec_pack<T>(s);
    }
}

// Stride-1 pools on maps that fit LDS (the SPP block: 5 / 9 / 13 windows on the 16 x 20 map, models.py:91-94): one workgroup
// per (image, 16-byte channel vector) holds its plane of the map in LDS.
//   forward   separable: row pass (first maximum of each window row and its dx), column pass over the row results.  The
//             first row that holds the window maximum, and the first dx inside it, IS the first maximum in scan order (and
//             with NaNs the last NaN, as torch's `val > max || isnan(val)` keeps it): values and argmax codes are those of
//             the k*k scan above, for 2k instead of k*k loads per output (13 x 13: 43 -> 8 us per launch).
//   backward  the k*k gather of the kernel above, same (dy, dx) order -- same bits -- from an LDS copy of the gradient and
//             argmax planes.
template <typename T>
__global__ __launch_bounds__(1024) void maxpool_tile_fwd_kernel(DykEwDesc d, uint8_t* __restrict__ idx) {
    constexpr int EPV = ElemTraits<T>::EPV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HW = d.H * d.W, k = d.k, pad = (k - 1) / 2;
    uint4* tile = (uint4*)smem;                  // [HW] raw values
    uint4* rmax = tile + HW;                     // [HW] row-window maxima (as T)
    uint2* radx = (uint2*)(rmax + HW);           // [HW] dx of the row maximum, one byte per element
    const int CV = d.C / EPV;
    // (an image's channel vectors on ONE XCD: the 8 workgroups that share a 128-byte line of a pixel row then share an L2)
    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int b = bid / CV, c = (bid % CV) * EPV;
    const T* __restrict__ a = (const T*)d.a + (long)b * HW * d.lda + c;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) tile[p] = *(const uint4*)(a + (long)p * d.lda);
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const int y = p / d.W, x = p - y * d.W;
        float m[EPV]; uint32_t mi[EPV];
        bool first = true;
        for (int dx = 0; dx < k; ++dx) {
            const int xx = x + dx - pad;
            if (xx < 0 || xx >= d.W) continue;
            float t[EPV];
            vec_unpack<T>(tile[y * d.W + xx], t);
#pragma unroll
            for (int j = 0; j < EPV; ++j)
                if (first || t[j] > m[j] || t[j] != t[j]) { m[j] = t[j]; mi[j] = (uint32_t)dx; }
            first = false;
        }
        rmax[p] = vec_pack<T>(m);                // (exact: the maxima are elements of the tile)
        uint2 pk = make_uint2(0u, 0u);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            if (j < 4) pk.x |= mi[j] << (8 * j); else pk.y |= mi[j] << (8 * (j - 4));
        }
        radx[p] = pk;
    }
    __syncthreads();
    T* __restrict__ o = (T*)d.out + (long)b * HW * d.ldo + c;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const int y = p / d.W, x = p - y * d.W;
        float m[EPV]; uint32_t mi[EPV];
        bool first = true;
        for (int dy = 0; dy < k; ++dy) {
            const int yy = y + dy - pad;
            if (yy < 0 || yy >= d.H) continue;
            float t[EPV];
            vec_unpack<T>(rmax[yy * d.W + x], t);
            const uint2 ax = radx[yy * d.W + x];
#pragma unroll
            for (int j = 0; j < EPV; ++j)
                if (first || t[j] > m[j] || t[j] != t[j]) {
                    m[j] = t[j];
                    mi[j] = (uint32_t)(dy * k) + (((j < 4 ? ax.x : ax.y) >> (8 * (j & 3))) & 0xffu);
                }
            first = false;
        }
        *(uint4*)(o + (long)p * d.ldo) = vec_pack<T>(m);
        if (idx) {
            uint8_t* ip = idx + ((long)b * HW + p) * d.C + c;
            uint32_t lo = 0u, hi = 0u;
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                if (j < 4) lo |= mi[j] << (8 * j); else hi |= mi[j] << (8 * (j - 4));
            }
            *(uint32_t*)ip = lo;                 // (C % 4 == 0: aligned)
            if (EPV > 4) *(uint32_t*)(ip + 4) = hi;
        }
    }
}

// (Tried in round 3 and dropped: the backward as a SCATTER -- one lane per (image, channel) walking its outputs in scan order
// with ds_add_f32 into a [pixel][64] fp32 plane, deterministic because one lane owns an address.  One add per output instead
// of k*k compare-selects per input, but a 64-channel slab needs 143 KB of LDS, i.e. one workgroup per CU and 128 workgroups
// per launch: 51 us per launch whatever the window, against 12 / 40 / 80 us for the 5 / 9 / 13 gathers below.)
// Round 4: the gather is SEPARABLE too.  The forward's code of output (y, x) is dy * k + radx[y + dy - pad][x]: the dx part
// belongs to the (row, column) of the row-window result, not to the output, so every output that selects row yy at column x
// carries the same dx.  Stage 1 walks the k outputs above / below (yy, x) and sums those whose code selects row yy -- the
// gradient of the row-window result R[yy][x] -- keeping their common dx; stage 2 walks the k row-window results left / right
// of (yy, xx) and sums those whose dx points at xx.  2k compare-selects per input instead of k * k (13 x 13 on the 16 x 20
// map, batch 16: 68 -> 15 us; the three SPP pools sit back to back on the backward critical path).  fp32 sums, the same
// terms as the k * k gather in a different association.
template <typename T>
__global__ __launch_bounds__(1024) void maxpool_tile_bwd_kernel(DykEwDesc d, const uint8_t* __restrict__ idx) {
    constexpr int EPV = ElemTraits<T>::EPV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HW = d.H * d.W, k = d.k, pad = (k - 1) / 2;
    uint4* gt = (uint4*)smem;                    // [HW] output gradients
    float4* rg = (float4*)(gt + HW);             // [HW][2] gradients of the row-window results, fp32
    uint2* it = (uint2*)(rg + 2 * HW);           // [HW] argmax codes, one byte per element
    uint2* rd = it + HW;                         // [HW] dx of the row-window result (0xff: nobody selected it)
    const int CV = d.C / EPV;
    // (an image's channel vectors on ONE XCD: the 8 workgroups that share a 128-byte line of a pixel row then share an L2)
    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int b = bid / CV, c = (bid % CV) * EPV;
    const T* __restrict__ a = (const T*)d.a + (long)b * HW * d.lda + c;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        gt[p] = *(const uint4*)(a + (long)p * d.lda);
        const uint8_t* ip = idx + ((long)b * HW + p) * d.C + c;
        uint2 pk;
        pk.x = *(const uint32_t*)ip;
        pk.y = EPV > 4 ? *(const uint32_t*)(ip + 4) : 0u;
        it[p] = pk;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const int yy = p / d.W, x = p - yy * d.W;
        float r[EPV]; uint32_t dxs[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) { r[j] = 0.f; dxs[j] = 0xffu; }
        for (int dy = 0; dy < k; ++dy) {
            const int yo = yy - dy + pad;
            if (yo < 0 || yo >= d.H) continue;
            const uint2 ix = it[yo * d.W + x];
            float g[EPV];
            vec_unpack<T>(gt[yo * d.W + x], g);
            const uint32_t lo = (uint32_t)(dy * k);
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                const uint32_t code = (((j < 4 ? ix.x : ix.y) >> (8 * (j & 3))) & 0xffu) - lo;     // dx if this output selects row yy
                if (code < (uint32_t)k) { r[j] += g[j]; dxs[j] = code; }
            }
        }
        rg[2 * p] = make_float4(r[0], r[1], r[2], r[3]);
        if (EPV > 4) rg[2 * p + 1] = make_float4(r[4 % EPV], r[5 % EPV], r[6 % EPV], r[7 % EPV]);
        uint2 pk = make_uint2(0u, 0u);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            if (j < 4) pk.x |= dxs[j] << (8 * j); else pk.y |= dxs[j] << (8 * (j - 4));
        }
        rd[p] = pk;
    }
    __syncthreads();
    T* __restrict__ o = (T*)d.out + (long)b * HW * d.ldo + c;
    const bool accum = d.flags & DYK_EW_ACCUM;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const int y = p / d.W, x = p - y * d.W;
        float s[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) s[j] = 0.f;
        for (int dx = 0; dx < k; ++dx) {
            const int xo = x - dx + pad;
            if (xo < 0 || xo >= d.W) continue;
            const uint2 ix = rd[y * d.W + xo];
            const float4 r0 = rg[2 * (y * d.W + xo)];
            float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPV > 4) r1 = rg[2 * (y * d.W + xo) + 1];
            const float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int j = 0; j < EPV; ++j)
                if ((((j < 4 ? ix.x : ix.y) >> (8 * (j & 3))) & 0xffu) == (uint32_t)dx) s[j] += r[j];
        }
        if (accum) {
            float t[EPV];
            vec_unpack<T>(*(const uint4*)(o + (long)p * d.ldo), t);
#pragma unroll
            for (int j = 0; j < EPV; ++j) s[j] += t[j];
        }
        *(uint4*)(o + (long)p * d.ldo) = vec_pack<T>(s);
    }
}

// ------------------------------------------------------------------ squeeze-excitation
// pooled[b][c] = alpha * sum_hw a[b,p,c] * (b ? b[b,p,c] : 1)        grid (CV groups, B, HS)
// HS = gridDim.z > 1: the pixels of an image are split over HS workgroups which store unscaled partial sums
// part[z][b][c]; se_pool_fold_kernel adds them in z order (bit-reproducible, no atomics).  A (C = 120, 64 x 80, B = 32)
// tensor is otherwise reduced by 32 workgroups -- 0.7 TB/s.
template <typename T>
__global__ __launch_bounds__(256) void se_pool_kernel(DykEwDesc d, float* __restrict__ pooled, int CVB) {
    constexpr int EPV = ElemTraits<T>::EPV;
    __shared__ float red[256 * 8];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int CV = d.C / EPV;
    const int cv = blockIdx.x * CVB + tx;
    const int c = cv * EPV;
    const int b = blockIdx.y;
    const int HW = d.H * d.W;
    const bool active = cv < CV;
    float s[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s[j] = 0.f;
    if (active) {
        const T* __restrict__ a = (const T*)d.a;
        const T* __restrict__ bb = (const T*)d.b;
        // four pixels per thread in flight: with one dependent load (pair) per iteration the loop ran at HBM latency
        // (32..192 workgroups per launch: nothing else hides it)
        constexpr int U = 4;
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
        const int chunk = (HW + (int)gridDim.z - 1) / (int)gridDim.z;
        const int pbeg = (int)blockIdx.z * chunk, pend = min(HW, pbeg + chunk);
        for (int p0 = pbeg + ty; p0 < pend; p0 += PY * U) {
            uint4 va[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * PY;
                const long pp = (long)b * HW + (p < pend ? p : pbeg);
                va[u] = *(const uint4*)(a + pp * d.lda + c);
                if (p >= pend) va[u] = z4;                                          // (zero contributes nothing)
                if (bb) vb[u] = *(const uint4*)(bb + pp * d.ldb + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float x[EPV];
                vec_unpack<T>(va[u], x);
                if (bb) {
                    float y[EPV];
                    vec_unpack<T>(vb[u], y);
#pragma unroll
                    for (int j = 0; j < EPV; ++j) s[j] += x[j] * y[j];
                } else {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) s[j] += x[j];
                }
            }
        }
    }
    float* mine = red + threadIdx.x * 8;
#pragma unroll
    for (int j = 0; j < EPV; ++j) mine[j] = s[j];
    __syncthreads();
    if (ty == 0 && active) {
        for (int q = 1; q < PY; ++q) {
            const float* o = red + (q * CVB + tx) * 8;
#pragma unroll
            for (int j = 0; j < EPV; ++j) s[j] += o[j];
        }
        if (gridDim.z > 1) {
            float* part = (float*)d.aux2 + ((long)blockIdx.z * d.B + b) * d.C + c;
#pragma unroll
            for (int j = 0; j < EPV; ++j) part[j] = s[j];
        } else {
#pragma unroll
            for (int j = 0; j < EPV; ++j) pooled[(long)b * d.C + c + j] = s[j] * d.alpha;
        }
    }
}
__global__ __launch_bounds__(256) void se_pool_fold_kernel(const float* __restrict__ part, float* __restrict__ pooled, int n,
                                                           int HS, float alpha) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v[DYK_SE_POOL_SPLITS];
#pragma unroll
    for (int z = 0; z < DYK_SE_POOL_SPLITS; ++z) v[z] = part[(long)(z < HS ? z : 0) * n + i];
    float t = 0.f;
#pragma unroll
    for (int z = 0; z < DYK_SE_POOL_SPLITS; ++z) t += z < HS ? v[z] : 0.f;
    pooled[i] = t * alpha;
}

// dot(W[row], v) for the SE matrix-vector products: a 16-lane group per output row (4 rows per wave in flight),
// float4 loads, DPP row reduction -- the wave-per-row version spent its time in 6-step ds_bpermute reductions.
// n is a multiple of 4 (channel counts are multiples of 8); every lane of the group returns the sum.
__device__ inline float dot16(const float* __restrict__ wrow, const float* v, int n, int l16) {
    // four 16-byte weight loads in flight per lane and trip (clamped address, zero weight past the end): the one-load
    // loop was a chain of n/64 dependent L2 latencies per output row
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = l16 * 4; i < n; i += 256) {
        const int i1 = i + 64, i2 = i + 128, i3 = i + 192;
        const int j1 = i1 < n ? i1 : i, j2 = i2 < n ? i2 : i, j3 = i3 < n ? i3 : i;
        const float4 w0 = *(const float4*)(wrow + i), w1 = *(const float4*)(wrow + j1), w2 = *(const float4*)(wrow + j2),
                     w3 = *(const float4*)(wrow + j3);
        const float m1 = i1 < n ? 1.f : 0.f, m2 = i2 < n ? 1.f : 0.f, m3 = i3 < n ? 1.f : 0.f;
        a0 += w0.x * v[i] + w0.y * v[i + 1] + w0.z * v[i + 2] + w0.w * v[i + 3];
        a1 += m1 * (w1.x * v[j1] + w1.y * v[j1 + 1] + w1.z * v[j1 + 2] + w1.w * v[j1 + 3]);
        a2 += m2 * (w2.x * v[j2] + w2.y * v[j2 + 1] + w2.z * v[j2 + 2] + w2.w * v[j2 + 3]);
        a3 += m3 * (w3.x * v[j3] + w3.y * v[j3 + 1] + w3.z * v[j3 + 2] + w3.w * v[j3 + 3]);
    }
    return row16_sum((a0 + a1) + (a2 + a3));
}

// The two FC layers of a squeeze-excitation block, h = relu(W1 pooled + b1), s = hardsigmoid(W2 h + b2) (layers.py:185-189),
// and their backward, as small launches that spread the WEIGHT MATRICES over the chip.  (Rounds 1-2 ran one 1024-thread
// block per image through the whole chain: every block streamed both matrices -- 2 x 1 MB fp32 at C = 1024 -- through one
// CU, 24 us forward and 57 us backward per block for 4 MFLOP; 19 such blocks per MobileNetV3 step.)
//   forward    se_rows_kernel<0>: h  [B][Cs] -> ws        se_rows_kernel<1>: s -> scale, t2 = W2 h + b2 -> ws
//   backward   se_cols_kernel<0>: dt2 = dscale * hardsigmoid'(t2);  dt1 = (h > 0) * (W2^T dt2) -> ws
//              se_cols_kernel<1>: dpooled = W1^T dt1             se_fc_wgrad_kernel: parameter gradients
// Every sum has a fixed order (16-lane DPP rows / 16 partial sums folded as a tree): bit-reproducible.
// ws: h [B][Cs] | dt1 [B][Cs] | t2 [B][C]  -- h and t2 are written by the forward call and read by the backward call.
template <int ACT>
__global__ __launch_bounds__(256) void se_rows_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                      const float* __restrict__ in, float* __restrict__ out, float* __restrict__ pre,
                                                      int R, int K, int B, int bper) {
    // out[b][r] = act(dot(W[r], in[b]) + bias[r]): a 16-lane group per row, the images of this block's slice in turn
    const int l16 = threadIdx.x & 15, g16 = threadIdx.x >> 4;
    const int r = blockIdx.x * 16 + g16;
    if (r >= R) return;
    const int b0 = blockIdx.y * bper, b1 = min(B, b0 + bper);
    const float bs = bias[r];
    for (int b = b0; b < b1; ++b) {
        const float acc = dot16(W + (long)r * K, in + (long)b * K, K, l16);
        if (l16 == 0) {
            const float t = acc + bs;
            if (ACT == 0) out[(long)b * R + r] = fmaxf(t, 0.f);
            else {
                out[(long)b * R + r] = fminf(fmaxf(t + 3.f, 0.f), 6.f) * (1.f / 6.f);
                if (pre) pre[(long)b * R + r] = t;
            }
        }
    }
}

// out[b][k] = sum_r W[r][k] v[b][r]: 16 columns x 16 row groups per block and image, partial sums folded in a fixed order
//   MODE 0: W = W2 [C][Cs], v = dt2 (built here from t2 and dscale), out = dt1 = (h > 0) ? sum : 0
//   MODE 1: W = W1 [Cs][C], v = dt1,                                 out = dpooled
template <int MODE>
__global__ __launch_bounds__(256) void se_cols_kernel(DykSeFcDesc d) {
    extern __shared__ float sm[];           // v[R] | part[16][16]
    const int R = MODE == 0 ? d.C : d.Cs, K = MODE == 0 ? d.Cs : d.C;
    const float* __restrict__ W = MODE == 0 ? d.w2 : d.w1;
    const int b = blockIdx.y;
    float* v = sm;
    float* part = sm + R;
    float* h = d.ws + (long)b * d.Cs;
    float* dt1 = d.ws + (long)d.B * d.Cs + (long)b * d.Cs;
    const float* t2 = d.ws + 2L * d.B * d.Cs + (long)b * d.C;
    for (int r = threadIdx.x; r < R; r += 256) {
        if (MODE == 0) {
            const float t = t2[r];
            v[r] = (t > -3.f && t < 3.f) ? d.dscale[(long)b * d.C + r] * (1.f / 6.f) : 0.f;
        } else {
            v[r] = dt1[r];
        }
    }
    __syncthreads();
    const int kt = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + kt;
    float acc = 0.f;
    if (k < K) {
#pragma unroll 16
        for (int r = rg; r < R; r += 16) acc += W[(long)r * K + k] * v[r];
    }
    part[rg * 16 + kt] = acc;
    __syncthreads();
    if (threadIdx.x < 16 && k < K) {
        float q[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) q[i] = part[i * 16 + kt];
#pragma unroll
        for (int w = 8; w > 0; w >>= 1)
#pragma unroll
            for (int i = 0; i < w; ++i) q[i] += q[i + w];
        if (MODE == 0) dt1[k] = h[k] > 0.f ? q[0] : 0.f;
        else d.dpooled[(long)b * d.C + k] = q[0];
    }
}

// dW2[c][j] += sum_b dt2[b][c] h[b][j] ;  dW1[j][c] += sum_b dt1[b][j] pooled[b][c] ;  db2[c] += sum_b dt2[b][c] ;
// db1[j] += sum_b dt1[b][j].  One workgroup per ROW of a weight matrix: the row's batch vector (dt2[:, c] or dt1[:, j]) goes
// to LDS once, the threads walk the row's columns with coalesced loads of the other factor -- sums over the batch in image
// order (reproducible, no atomics).  (One thread per weight element, each re-reading its row's batch vector from memory,
// was 3 vector loads per image and element: 18 us per launch on the C = 960 blocks of MobileNetV3.)  dt2 is rebuilt from
// t2 and dscale: the backward launches never overwrite what the forward call parked.
__global__ __launch_bounds__(256) void se_fc_wgrad_kernel(DykSeFcDesc d) {
    __shared__ float vb[256];                              // the row's factor for every image (B <= 256, checked by the host)
    const float* h = d.ws;
    const float* dt1 = d.ws + (long)d.B * d.Cs;
    const float* t2 = d.ws + 2L * d.B * d.Cs;
    const int row = blockIdx.x;
    const bool second = row < d.C;                         // rows [0, C): dW2 / db2, rows [C, C + Cs): dW1 / db1
    const int r = second ? row : row - d.C;
    if ((int)threadIdx.x < d.B) {
        const int b = threadIdx.x;
        float v;
        if (second) {
            const float t = t2[(long)b * d.C + r];
            v = (t > -3.f && t < 3.f) ? d.dscale[(long)b * d.C + r] * (1.f / 6.f) : 0.f;
        } else {
            v = dt1[(long)b * d.Cs + r];
        }
        vb[b] = v;
    }
    __syncthreads();
    const int ncol = second ? d.Cs : d.C;
    const float* __restrict__ other = second ? h : d.pooled;       // [B][ncol]
    float* __restrict__ dw = second ? d.dw2 + (long)r * d.Cs : d.dw1 + (long)r * d.C;
    for (int col = threadIdx.x; col < ncol; col += 256) {
        float acc = 0.f;
#pragma unroll 8
        for (int b = 0; b < d.B; ++b) acc += vb[b] * other[(long)b * ncol + col];
        dw[col] += acc;
    }
    if (threadIdx.x == 0) {
        float acc = 0.f;
        for (int b = 0; b < d.B; ++b) acc += vb[b];
        if (second) d.db2[r] += acc; else d.db1[r] += acc;
    }
}

// out[b,p,c] = a[b,p,c]*p0[b*C+c] (+ p1[b*C+c]*alpha)      (SE scale; backward apply with p1 = dpooled, alpha = 1/HW)
// block = (CVB channel vectors) x (PY pixel lanes), grid (channel groups, pixel groups, B): the per-(image, channel)
// factors sit in registers, no integer division per element, four pixels in flight per thread (clamped addresses).
// ACT >= -1 (round 5): `out` is the gradient w.r.t. the activated output of a conv + BatchNorm layer whose raw output is `b` and
// whose scale | shift | mean | rstd vectors are p2[4][C]: the BatchNorm-backward reduce of that layer rides along -- sum(da),
// sum(da * xhat) with da = out * act'(scale * b + shift), out rounded to the storage type first (what a separate
// dyk_bn_act_bwd_reduce pass over `out` would read), into the fp64 replicas `red`; `out` itself stays the gradient (the apply
// pass forms act' again: the keep-dz form of DykConvDesc's DYK_EPI_BNBWD | DYK_EPI_ADDEND).  ACT = -2: no reduce.
template <typename T, int ACT>
__global__ __launch_bounds__(256) void se_scale_kernel(DykEwDesc d, int CVB) {
    constexpr int EPV = ElemTraits<T>::EPV;
    constexpr bool RED = ACT >= -1;
    __shared__ float red[RED ? 256 * 2 * 8 : 8];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int cv = blockIdx.x * CVB + tx;
    const bool active = cv * EPV < d.C;
    if (!RED && !active) return;
    const int c = active ? cv * EPV : 0;
    const int HW = d.H * d.W;
    float sc[EPV], sh[EPV], mu[EPV], rs[EPV], s1[EPV], s2[EPV];
    if constexpr (RED) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            sc[j] = d.p2[c + j]; sh[j] = d.p2[d.C + c + j]; mu[j] = d.p2[2 * d.C + c + j]; rs[j] = d.p2[3 * d.C + c + j];
            s1[j] = s2[j] = 0.f;
        }
    }
    const bool accum = d.flags & DYK_EW_ACCUM;
    constexpr int U = 4;
    const int pstep = (int)gridDim.y * PY;
    // (with the reduce a workgroup walks several images: on the small maps a (channel group, pixel group, image) cell is 64 pixels,
    // less work than the workgroup's own fold + 2 x 256 atomics -- 36 us per launch on the MobileNetV3 cfg's 16 x 20 blocks)
    for (int b = blockIdx.z; b < d.B; b += gridDim.z) {
        float f0[EPV], f1[EPV];
#pragma unroll
        for (int j = 0; j < EPV; j += 4) {
            const float4 q = *(const float4*)(d.p0 + (long)b * d.C + c + j);
            f0[j] = q.x; f0[j + 1] = q.y; f0[j + 2] = q.z; f0[j + 3] = q.w;
            if (d.p1) {
                const float4 r = *(const float4*)(d.p1 + (long)b * d.C + c + j);
                f1[j] = r.x * d.alpha; f1[j + 1] = r.y * d.alpha; f1[j + 2] = r.z * d.alpha; f1[j + 3] = r.w * d.alpha;
            } else {
                f1[j] = f1[j + 1] = f1[j + 2] = f1[j + 3] = 0.f;
            }
        }
        const T* __restrict__ a = (const T*)d.a + (long)b * HW * d.lda + c;
        const T* __restrict__ yr = (const T*)d.b + (long)b * HW * d.ldb + c;
        T* __restrict__ o = (T*)d.out + (long)b * HW * d.ldo + c;
        if (!active) continue;
        for (int p0 = (int)blockIdx.y * PY + ty; p0 < HW; p0 += pstep * U) {
            uint4 va[U], vo[U], vy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * pstep;
                const long pp = p < HW ? p : p0;
                va[u] = *(const uint4*)(a + pp * d.lda);
                if (accum) vo[u] = *(const uint4*)(o + pp * d.ldo);
                if constexpr (RED) vy[u] = *(const uint4*)(yr + pp * d.ldb);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * pstep;
                if (p >= HW) break;
                float x[EPV];
                vec_unpack<T>(va[u], x);
#pragma unroll
                for (int j = 0; j < EPV; ++j) x[j] = x[j] * f0[j] + f1[j];
                if (accum) {
                    float y[EPV];
                    vec_unpack<T>(vo[u], y);
#pragma unroll
                    for (int j = 0; j < EPV; ++j) x[j] += y[j];
                }
                const uint4 pk = vec_pack<T>(x);
                *(uint4*)(o + (long)p * d.ldo) = pk;
                if constexpr (RED) {
                    float g[EPV], yy[EPV];
                    vec_unpack<T>(pk, g);
                    vec_unpack<T>(vy[u], yy);
#pragma unroll
                    for (int j = 0; j < EPV; ++j) {
                        const float da = g[j] * act_bwd_c<ACT>(yy[j] * sc[j] + sh[j], d.act);
                        s1[j] += da;
                        s2[j] += da * ((yy[j] - mu[j]) * rs[j]);
                    }
                }
            }
        }
    }
    if constexpr (RED) {
        // pixel lanes through LDS in lane order, then one fp64 atomic per channel and sum into this workgroup's replica
        float* mine = red + threadIdx.x * 2 * 8;
#pragma unroll
        for (int j = 0; j < EPV; ++j) { mine[j] = s1[j]; mine[8 + j] = s2[j]; }
        __syncthreads();
        if (ty == 0 && active) {
            for (int q = 1; q < PY; ++q) {
                const float* oo = red + (q * CVB + tx) * 2 * 8;
#pragma unroll
                for (int j = 0; j < EPV; ++j) { s1[j] += oo[j]; s2[j] += oo[8 + j]; }
            }
            const unsigned wgi = blockIdx.z * gridDim.y + blockIdx.y;
            double* rd = d.red + (size_t)(wgi % (unsigned)(d.slots > 0 ? d.slots : 1)) * 2 * d.C;
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                atomicAdd(rd + c + j, (double)s1[j]);
                atomicAdd(rd + d.C + c + j, (double)s2[j]);
            }
        }
    }
}

// ------------------------------------------------------------------ YOLO head permute
// fwd: y [B,ny,nx,ld] fp32 (channel = a*no + o)  ->  p [B,na,ny,nx,no] fp32   (models.py:229)
__global__ void head_permute_fwd_kernel(const float* __restrict__ y, float* __restrict__ p, int B, int ny, int nx,
                                        int na, int no, int ld) {
    const long total = (long)B * na * ny * nx * no;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i % no);
        long q = i / no;
        const int x = (int)(q % nx); q /= nx;
        const int yy = (int)(q % ny); q /= ny;
        const int a = (int)(q % na);
        const int b = (int)(q / na);
        p[i] = y[(((long)b * ny + yy) * nx + x) * ld + a * no + o];
    }
}
// bwd: dp [B,na,ny,nx,no] fp32 -> dy [B,ny,nx,ld] T (channels >= na*no zero)
template <typename T>
__global__ void head_permute_bwd_kernel(const float* __restrict__ dp, T* __restrict__ dy, int B, int ny, int nx,
                                        int na, int no, int ld) {
    const long total = (long)B * ny * nx * ld;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld);
        long q = i / ld;
        const int x = (int)(q % nx); q /= nx;
        const int yy = (int)(q % ny);
        const int b = (int)(q / ny);
        float v = 0.f;
        if (c < na * no) {
            const int a = c / no, o = c - a * no;
            v = dp[((((long)b * na + a) * ny + yy) * nx + x) * no + o];
        }
        dy[i] = ElemTraits<T>::from_f32(v);
    }
}
// bias gradient of the head conv: db[c] += sum_{b,y,x} dp[b, c/no, y, x, c%no]
// One 1024-thread workgroup per channel (fixed summation order: bit-reproducible), four independent loads per thread and
// trip -- the 256-thread form with a 64-bit division per element took 140 us for the 64 x 80 head at batch 32.
__global__ __launch_bounds__(1024) void head_bias_grad_kernel(const float* __restrict__ dp, float* __restrict__ db, int B,
                                                              int ny, int nx, int na, int no) {
    const int c = blockIdx.x;
    const int a = c / no, o = c - a * no;
    const int cells = ny * nx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* src = dp + ((long)b * na + a) * cells * no + o;
        int r = threadIdx.x;
        for (; r + 3072 < cells; r += 4096) {
            const float v0 = src[(long)r * no], v1 = src[(long)(r + 1024) * no], v2 = src[(long)(r + 2048) * no],
                        v3 = src[(long)(r + 3072) * no];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; r < cells; r += 1024) s0 += src[(long)r * no];
    }
    __shared__ float ws[16];
    const float s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += ws[w];
        db[c] += t;
    }
}

// ------------------------------------------------------------------ first-layer patch gather
// in NCHW fp32 [B,Cin,H,W] -> out [B,Ho,Wo,ld] T with out[.., (kh*k+kw)*Cin + c] = in[b,c,yo*s+kh-pad,xo*s+kw-pad]*mul
// One thread per output pixel: the k*k*Cin gathered values (consecutive threads read consecutive x: coalesced NCHW
// reads) are packed in registers and the whole zero-padded row of `ld` channels leaves as contiguous 16-byte stores.
template <typename T>
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ in, T* __restrict__ out, int B,
                                                           int Cin, int H, int W, int k, int stride, int pad, int Ho,
                                                           int Wo, int ld, float mul) {
    constexpr int EPV = ElemTraits<T>::EPV;
    const int npix = B * Ho * Wo;
    const int K = k * k * Cin;
    const int CV = ld / EPV;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const int xo = p % Wo;
        const int q = p / Wo;
        const int yo = q % Ho;
        const int b = q / Ho;
        const float* img = in + (long)b * Cin * H * W;
        T* row = out + (long)p * ld;
        int j = 0, tap = 0, c = 0;                 // j = tap * Cin + c walks the gathered row
        for (int v = 0; v < CV; ++v) {
            float val[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float t = 0.f;
                if (j < K) {
                    const int kh = tap / k, kw = tap - kh * k;
                    const int yy = yo * stride + kh - pad, xx = xo * stride + kw - pad;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) t = img[((long)c * H + yy) * W + xx] * mul;
                    if (++c == Cin) { c = 0; ++tap; }
                    ++j;
                }
                val[e] = t;
            }
            *(uint4*)(row + v * EPV) = vec_pack<T>(val);
        }
    }
}

int check_ew(const DykEwDesc* d, bool need_b, bool need_out = true, bool twin_ok = false) {
    if (!d || !d->a || (need_out && !d->out) || (need_b && !d->b)) return DYK_ERR_ARG;
    if (d->twin && !twin_ok) return DYK_ERR_UNSUPPORTED;      // two-problem launches: BatchNorm passes and axpby only
    if (d->dtype != DYK_BF16 && d->dtype != DYK_F32) return DYK_ERR_ARG;
    const int epv = epv_of(d->dtype);
    if (d->C <= 0 || d->C % epv || d->lda % epv || (need_out && d->ldo % epv) || (d->b && d->ldb % epv)) return DYK_ERR_ARG;
    return DYK_OK;
}

}  // namespace

#define DISPATCH_T(kern, grid, block, lds, stream, ...)                                              \
    do {                                                                                             \
        if (d->dtype == DYK_BF16) hipLaunchKernelGGL(kern<bf16_t>, grid, block, lds, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern<float>, grid, block, lds, stream, __VA_ARGS__);                 \
        DYK_LAUNCH_CHECK();                                                                          \
    } while (0)

extern "C" int dyk_axpby(const DykEwDesc* d, void* stream) {
    const int rc = check_ew(d, false, true, true);
    if (rc) return rc;
    if (d->npix <= 0) return DYK_ERR_ARG;
    const int grid = ew_grid((long)d->npix * (d->C / epv_of(d->dtype)));
    DykEwPair pr;
    const int nz = dyk_fill_ew_pair(pr, d);
    if (!nz) return DYK_ERR_ARG;
    DISPATCH_T(axpby_kernel, dim3(grid, 1, nz), dim3(256), 0, (hipStream_t)stream, pr);
    return DYK_OK;
}

extern "C" int dyk_dot(const DykEwDesc* d, void* stream) {
    const int rc = check_ew(d, true, false);
    if (rc) return rc;
    if (d->npix <= 0 || !d->red || (d->out && d->ldo % epv_of(d->dtype))) return DYK_ERR_ARG;
    long g = ((long)d->npix * (d->C / epv_of(d->dtype)) + 256 * 8 - 1) / (256 * 8);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    DISPATCH_T(dot_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, *d);
    return DYK_OK;
}

extern "C" int dyk_wfuse_weights(const float* w, float* weff, int32_t n, void* stream) {
    if (!w || !weff || n <= 0 || n > 64) return DYK_ERR_ARG;
    hipLaunchKernelGGL(wfuse_weights_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, weff, n);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_wfuse_bwd_params(const float* w, const double* red, float* dw, int32_t n, void* stream) {
    if (!w || !red || !dw || n <= 0 || n > 64) return DYK_ERR_ARG;
    hipLaunchKernelGGL(wfuse_bwd_params_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, red, dw, n);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_upsample2x_fwd(const DykEwDesc* d, void* stream) {
    const int rc = check_ew(d, false);
    if (rc) return rc;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0) return DYK_ERR_ARG;
    const int grid = ew_grid((long)d->B * d->H * d->W * 4 * (d->C / epv_of(d->dtype)));
    DISPATCH_T(upsample2x_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d);
    return DYK_OK;
}

extern "C" int dyk_upsample2x_bwd(const DykEwDesc* d, void* stream) {
    const int rc = check_ew(d, false);
    if (rc) return rc;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0) return DYK_ERR_ARG;
    const int grid = ew_grid((long)d->B * d->H * d->W * (d->C / epv_of(d->dtype)));
    DISPATCH_T(upsample2x_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d);
    return DYK_OK;
}

#define MAXPOOL_K(kern, KK, arg)                                                                                  \
    do {                                                                                                         \
        if (d->dtype == DYK_BF16) hipLaunchKernelGGL((kern<bf16_t, KK>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d, arg); \
        else hipLaunchKernelGGL((kern<float, KK>), dim3(grid), dim3(256), 0, (hipStream_t)stream, *d, arg);      \
    } while (0)
#define MAXPOOL_DISPATCH(kern, arg)                                                                               \
    do {                                                                                                         \
        switch (d->k) {                                                                                          \
        case 2: MAXPOOL_K(kern, 2, arg); break;                                                                  \
        case 3: MAXPOOL_K(kern, 3, arg); break;                                                                  \
        case 5: MAXPOOL_K(kern, 5, arg); break;                                                                  \
        case 9: MAXPOOL_K(kern, 9, arg); break;                                                                  \
        case 13: MAXPOOL_K(kern, 13, arg); break;                                                                \
        default: MAXPOOL_K(kern, 0, arg); break;                                                                 \
        }                                                                                                        \
        DYK_LAUNCH_CHECK();                                                                                      \
    } while (0)

// the LDS-plane kernels: stride 1, odd window (output map = input map), map of at most 1024 pixels, 4-byte aligned argmax rows
static bool maxpool_tile_ok(const DykEwDesc* d) {
    return d->slots <= 1 && (d->k & 1) && d->H * d->W <= 1024 && d->C % 4 == 0;
}
static int maxpool_tile_threads(int HW) { const int t = (HW + 63) / 64 * 64; return t > 1024 ? 1024 : t; }

extern "C" int dyk_maxpool_fwd(const DykEwDesc* d, uint8_t* argmax, void* stream) {
    const int rc = check_ew(d, false);
    if (rc) return rc;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->k <= 0 || d->k > 15 || d->slots < 0 || d->slots > 8) return DYK_ERR_ARG;
    if (maxpool_tile_ok(d)) {
        const int HW = d->H * d->W, threads = maxpool_tile_threads(HW);
        const unsigned grid = (unsigned)(d->B * (d->C / epv_of(d->dtype)));
        if (d->dtype == DYK_BF16) hipLaunchKernelGGL(maxpool_tile_fwd_kernel<bf16_t>, dim3(grid), dim3(threads), (size_t)HW * 40, (hipStream_t)stream, *d, argmax);
        else hipLaunchKernelGGL(maxpool_tile_fwd_kernel<float>, dim3(grid), dim3(threads), (size_t)HW * 40, (hipStream_t)stream, *d, argmax);
        DYK_LAUNCH_CHECK();
        return DYK_OK;
    }
    const int grid = ew_grid((long)d->B * d->H * d->W * (d->C / epv_of(d->dtype)));
    MAXPOOL_DISPATCH(maxpool_fwd_kernel, argmax);
    return DYK_OK;
}

extern "C" int dyk_maxpool_bwd(const DykEwDesc* d, const uint8_t* argmax, void* stream) {
    const int rc = check_ew(d, false);
    if (rc) return rc;
    if (!argmax || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->k <= 0 || d->k > 15 || d->slots < 0 || d->slots > 8) return DYK_ERR_ARG;
    if (maxpool_tile_ok(d)) {
        const int HW = d->H * d->W, threads = maxpool_tile_threads(HW);
        const unsigned grid = (unsigned)(d->B * (d->C / epv_of(d->dtype)));
        if (d->dtype == DYK_BF16) hipLaunchKernelGGL(maxpool_tile_bwd_kernel<bf16_t>, dim3(grid), dim3(threads), (size_t)HW * 64, (hipStream_t)stream, *d, argmax);
        else hipLaunchKernelGGL(maxpool_tile_bwd_kernel<float>, dim3(grid), dim3(threads), (size_t)HW * 64, (hipStream_t)stream, *d, argmax);
        DYK_LAUNCH_CHECK();
        return DYK_OK;
    }
    const int grid = ew_grid((long)d->B * d->H * d->W * (d->C / epv_of(d->dtype)));
    MAXPOOL_DISPATCH(maxpool_bwd_kernel, argmax);
    return DYK_OK;
}

extern "C" int dyk_se_pool(const DykEwDesc* d, float* pooled, void* stream) {
    const int rc = check_ew(d, false, false);
    if (rc) return rc;
    if (!pooled || d->B <= 0 || d->H <= 0 || d->W <= 0) return DYK_ERR_ARG;
    const int CV = d->C / epv_of(d->dtype);
    int CVB = 1;
    while (CVB < CV && CVB < 16) CVB <<= 1;
    const int gx = (CV + CVB - 1) / CVB, PY = 256 / CVB;
    // split the pixels of an image when the (channel groups x images) grid leaves the chip idle and a scratch buffer
    // (aux2: DYK_SE_POOL_SPLITS * B * C floats) is there; >= 8 pixel rows per thread stay in every part
    int HS = 1;
    if (d->aux2) {
        const long hw = (long)d->H * d->W;
        while (HS < DYK_SE_POOL_SPLITS && (long)gx * d->B * HS < 1024 && hw / (HS * 2) >= (long)PY * 8) HS <<= 1;
    }
    DISPATCH_T(se_pool_kernel, dim3(gx, d->B, HS), dim3(256), 0, (hipStream_t)stream, *d, pooled, CVB);
    if (HS > 1) {
        const int n = d->B * d->C;
        hipLaunchKernelGGL(se_pool_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)d->aux2,
                           pooled, n, HS, d->alpha);
        DYK_LAUNCH_CHECK();
    }
    return DYK_OK;
}

static void se_rows_grid(int R, int B, dim3* grid, int* bper) {
    const int gx = (R + 15) / 16;
    int gy = 512 / gx;                      // about two workgroups per CU over (row blocks) x (image slices)
    if (gy < 1) gy = 1;
    if (gy > B) gy = B;
    *bper = (B + gy - 1) / gy;
    *grid = dim3((unsigned)gx, (unsigned)((B + *bper - 1) / *bper));
}

extern "C" int dyk_se_fc_fwd(const DykSeFcDesc* d, void* stream) {
    if (!d || !d->pooled || !d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->scale || !d->ws || d->B <= 0 || d->C <= 0 || d->Cs <= 0 ||
        d->C % 4 || d->Cs % 4)
        return DYK_ERR_ARG;
    float* h = d->ws;
    float* t2 = d->ws + 2L * d->B * d->Cs;
    dim3 grid;
    int bper;
    se_rows_grid(d->Cs, d->B, &grid, &bper);
    hipLaunchKernelGGL(se_rows_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, d->w1, d->b1, d->pooled, h, (float*)nullptr, d->Cs,
                       d->C, d->B, bper);
    se_rows_grid(d->C, d->B, &grid, &bper);
    hipLaunchKernelGGL(se_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, d->w2, d->b2, (const float*)h, d->scale, t2, d->C,
                       d->Cs, d->B, bper);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_se_fc_bwd(const DykSeFcDesc* d, void* stream) {
    // Two halves that a caller may issue as separate calls (round 4: the parameter gradients are not on the path to dx, and
    // one call held the backward chain for all three launches):
    //   dpooled != NULL                the data half: dt1 -> ws, dpooled
    //   dw1, db1, dw2, db2 != NULL     the parameter half (reads h, dt1, t2 from ws: after the data half)
    // all five set = both, in that order; the four gradient pointers come all or none.
    const bool data = d && d->dpooled, params = d && (d->dw1 || d->db1 || d->dw2 || d->db2);
    if (!d || !d->pooled || !d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->dscale || (!data && !params) ||
        (params && (!d->dw1 || !d->db1 || !d->dw2 || !d->db2)) || !d->ws || d->B <= 0 || d->B > 256 || d->C <= 0 || d->Cs <= 0 ||
        d->C > 8192 || d->Cs > 8192)
        return DYK_ERR_ARG;
    if (data) {
        hipLaunchKernelGGL(se_cols_kernel<0>, dim3((unsigned)((d->Cs + 15) / 16), (unsigned)d->B), dim3(256), (size_t)(d->C + 256) * 4,
                           (hipStream_t)stream, *d);
        hipLaunchKernelGGL(se_cols_kernel<1>, dim3((unsigned)((d->C + 15) / 16), (unsigned)d->B), dim3(256), (size_t)(d->Cs + 256) * 4,
                           (hipStream_t)stream, *d);
    }
    if (params) hipLaunchKernelGGL(se_fc_wgrad_kernel, dim3((unsigned)(d->C + d->Cs)), dim3(256), 0, (hipStream_t)stream, *d);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_se_scale(const DykEwDesc* d, void* stream) {
    const int rc = check_ew(d, false);
    if (rc) return rc;
    if (!d->p0 || d->B <= 0 || d->H <= 0 || d->W <= 0) return DYK_ERR_ARG;
    const int CV = d->C / epv_of(d->dtype);
    int CVB = 1;
    while (CVB < CV && CVB < 32) CVB <<= 1;
    const int PY = 256 / CVB, gx = (CV + CVB - 1) / CVB;
    const long hw = (long)d->H * d->W;
    long gy = (hw + (long)PY * 8 - 1) / ((long)PY * 8);          // >= 8 pixels per thread
    const long cap = 4096 / ((long)gx * d->B) > 0 ? 4096 / ((long)gx * d->B) : 1;
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    dim3 grid(gx, (int)gy, d->B);
    if (d->red) {          // with the BatchNorm-backward reduce of the producer of `out`'s tensor (see the kernel)
        if (!d->b || !d->p2 || d->ldb % epv_of(d->dtype)) return DYK_ERR_ARG;
        long nz = 768 / ((long)gx * gy);                 // ~3 workgroups per CU, each over B / nz images
        if (nz < 1) nz = 1;
        if (nz > d->B) nz = d->B;
        grid.z = (unsigned)nz;
#define DYK_SE_RED(A)                                                                                                     \
        do {                                                                                                              \
            if (d->dtype == DYK_BF16) hipLaunchKernelGGL((se_scale_kernel<bf16_t, A>), grid, dim3(256), 0, (hipStream_t)stream, *d, CVB); \
            else hipLaunchKernelGGL((se_scale_kernel<float, A>), grid, dim3(256), 0, (hipStream_t)stream, *d, CVB);        \
        } while (0)
        switch (d->act) {
        case DYK_ACT_MISH: DYK_SE_RED(DYK_ACT_MISH); break;
        case DYK_ACT_RELU: DYK_SE_RED(DYK_ACT_RELU); break;
        case DYK_ACT_HSWISH: DYK_SE_RED(DYK_ACT_HSWISH); break;
        case DYK_ACT_RELU6: DYK_SE_RED(DYK_ACT_RELU6); break;
        default: DYK_SE_RED(-1); break;
        }
#undef DYK_SE_RED
        DYK_LAUNCH_CHECK();
        return DYK_OK;
    }
    if (d->dtype == DYK_BF16) hipLaunchKernelGGL((se_scale_kernel<bf16_t, -2>), grid, dim3(256), 0, (hipStream_t)stream, *d, CVB);
    else hipLaunchKernelGGL((se_scale_kernel<float, -2>), grid, dim3(256), 0, (hipStream_t)stream, *d, CVB);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_head_permute_fwd(const float* y, float* p, int32_t B, int32_t ny, int32_t nx, int32_t na,
                                    int32_t no, int32_t ld, void* stream) {
    if (!y || !p || B <= 0 || ny <= 0 || nx <= 0 || na <= 0 || no <= 0 || ld < na * no) return DYK_ERR_ARG;
    const int grid = ew_grid((long)B * na * ny * nx * no);
    hipLaunchKernelGGL(head_permute_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, p, B, ny, nx, na, no, ld);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_head_permute_bwd(const float* dp, void* dy, float* dbias, int32_t B, int32_t ny, int32_t nx,
                                    int32_t na, int32_t no, int32_t ld, int32_t dtype, void* stream) {
    if (!dp || !dy || B <= 0 || ny <= 0 || nx <= 0 || na <= 0 || no <= 0 || ld < na * no) return DYK_ERR_ARG;
    const int grid = ew_grid((long)B * ny * nx * ld);
    if (dtype == DYK_BF16)
        hipLaunchKernelGGL(head_permute_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dp, (bf16_t*)dy, B, ny, nx, na, no, ld);
    else if (dtype == DYK_F32)
        hipLaunchKernelGGL(head_permute_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dp, (float*)dy, B, ny, nx, na, no, ld);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    if (dbias) {
        hipLaunchKernelGGL(head_bias_grad_kernel, dim3(na * no), dim3(1024), 0, (hipStream_t)stream, dp, dbias, B, ny, nx, na, no);
        DYK_LAUNCH_CHECK();
    }
    return DYK_OK;
}

// Input path of the training harness: `imgs.float() / 255.0` (+ bilinear resize, align_corners=False) in one pass.
// Index arithmetic follows ATen's area_pixel_compute_source_index / guard_index_and_lambda so that the four taps and
// both lambdas are the ones F.interpolate uses; one thread per output pixel, x fastest (coalesced stores).
template <typename S>
__global__ __launch_bounds__(256) void image_prep_kernel(const S* __restrict__ src, float* __restrict__ dst, int planes,
                                                         int Hi, int Wi, int Ho, int Wo, float div) {
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    const long total = (long)planes * Ho * Wo;
    const bool same = Hi == Ho && Wi == Wo;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (same) {
            dst[i] = (float)src[i] / div;
            continue;
        }
        const int xo = (int)(i % Wo);
        const long q = i / Wo;
        const int yo = (int)(q % Ho);
        const S* pl = src + (q / Ho) * (long)Hi * Wi;
        float fy = fmaf(sh, (float)yo + 0.5f, -0.5f), fx = fmaf(sw, (float)xo + 0.5f, -0.5f);   // fused like torch's build
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
        const int y1 = y0 + (y0 < Hi - 1), x1 = x0 + (x0 < Wi - 1);
        const float ly = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
        const float v00 = (float)pl[(long)y0 * Wi + x0] / div, v01 = (float)pl[(long)y0 * Wi + x1] / div;
        const float v10 = (float)pl[(long)y1 * Wi + x0] / div, v11 = (float)pl[(long)y1 * Wi + x1] / div;
        const float top = __fadd_rn(__fmul_rn(1.f - lx, v00), __fmul_rn(lx, v01));
        const float bot = __fadd_rn(__fmul_rn(1.f - lx, v10), __fmul_rn(lx, v11));
        dst[i] = __fadd_rn(__fmul_rn(1.f - ly, top), __fmul_rn(ly, bot));
    }
}

extern "C" int dyk_image_prep(const void* src, float* dst, int32_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                              int32_t src_dtype, float div, void* stream) {
    if (!src || !dst || planes <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || !(div > 0.f)) return DYK_ERR_ARG;
    const int grid = ew_grid((long)planes * Ho * Wo);
    if (src_dtype == DYK_U8)
        hipLaunchKernelGGL(image_prep_kernel<uint8_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, dst, planes, Hi, Wi, Ho, Wo, div);
    else if (src_dtype == DYK_F32)
        hipLaunchKernelGGL(image_prep_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, planes, Hi, Wi, Ho, Wo, div);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_patch_gather(const float* in, void* out, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t k,
                                int32_t stride, int32_t pad, int32_t ld, float mul, int32_t dtype, void* stream) {
    if (!in || !out || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0) return DYK_ERR_ARG;
    const int epv = epv_of(dtype);
    if (ld % epv || ld < k * k * Cin) return DYK_ERR_ARG;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if ((long)B * Ho * Wo >= (1L << 31)) return DYK_ERR_ARG;
    const int grid = ew_grid((long)B * Ho * Wo);
    if (dtype == DYK_BF16)
        hipLaunchKernelGGL(patch_gather_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, B, Cin, H, W, k, stride, pad, Ho, Wo, ld, mul);
    else if (dtype == DYK_F32)
        hipLaunchKernelGGL(patch_gather_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (float*)out, B, Cin, H, W, k, stride, pad, Ho, Wo, ld, mul);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
