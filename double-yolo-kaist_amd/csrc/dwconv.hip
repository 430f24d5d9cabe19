// Depthwise convolution (groups == channels), forward / data gradient / weight gradient, channels-last.
// Replaces nn.Conv2d(groups=C) at reference models.py:41 (MobileNet [convolutional] sections with `groups=`)
// and the first conv of DepthwiseSeparableConv2d (layers.py:223-224).  4.5-12 flop/byte: HBM-bound, so these
// are streaming kernels (16-byte vectors of 8 channels, the k*k re-reads of a pixel neighbourhood come from
// L1/L2), not MFMA work.  Weights are read directly from the fp32 master copy in tap-major order [k*k][C].
// The forward optionally accumulates the per-channel sum / sum of squares for a following train-mode
// BatchNorm into the same replicated fp64 buffers the MFMA conv epilogue uses.
#include <type_traits>
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
        // image rows over grid.y, pixels of a row over the pixel lanes: no per-pixel integer division
        const int nrows = d.B * Hout;
        for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
          const int b = row / Hout, yo = row - b * Hout;
          for (int xo = ty; xo < Wout; xo += PY) {
            const long p = (long)row * Wout + xo;
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

// Stride-1 fast path of the kernel above: a thread produces XT = 4 neighbouring pixels of a row for its 8 channels and
// keeps the (4 + K - 1) source vectors of each kernel row in registers: (K + 3) / 4 loads per output and row instead of
// K (2 x fewer for 3x3, 2.5 x fewer for 5x5), weights of a kernel row read once per strip.
template <typename T, int K, bool GRAD>
__global__ __launch_bounds__(256) void dwconv_strip_kernel(DykDwDesc d, int CVB) {
    constexpr int EPV = ElemTraits<T>::EPV;
    constexpr int XT = 4, NS = XT + K - 1;
    __shared__ float red[256 * 2 * 8];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int cv = blockIdx.x * CVB + tx;
    const int c = cv * EPV;
    const bool active = c < d.C;
    const T* __restrict__ x = GRAD ? (const T*)d.y : (const T*)d.x;     // tensor read
    T* __restrict__ y = GRAD ? (T*)d.x : (T*)d.y;                       // tensor written
    const int ld_src = GRAD ? d.ldy : d.ldx, ld_dst = GRAD ? d.ldx : d.ldy;
    const int pad = d.pad;
    const int H = d.Hi, W = d.Wi;                  // stride 1: source and destination extents differ only by k - 1 - 2 pad
    const int Hout = GRAD ? d.Hi : d.Ho, Wout = GRAD ? d.Wi : d.Wo;
    const int Hsrc = GRAD ? d.Ho : d.Hi, Wsrc = GRAD ? d.Wo : d.Wi;
    (void)H; (void)W;
    const bool accum = d.flags & DYK_EW_ACCUM;
    const bool stats = (!GRAD) && d.stats != nullptr;
    float s1[EPV], s2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s1[j] = s2[j] = 0.f;
    if (active) {
        const int nrows = d.B * Hout;
        for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
            const int b = row / Hout, yo = row - b * Hout;
            for (int xo0 = ty * XT; xo0 < Wout; xo0 += PY * XT) {
                float acc[XT][EPV];
#pragma unroll
                for (int o = 0; o < XT; ++o)
#pragma unroll
                    for (int j = 0; j < EPV; ++j) acc[o][j] = 0.f;
                // source column of strip slot 0:  forward xs = xo + kw - pad ; gradient xs = xo + pad - kw
                const int xbase = GRAD ? xo0 + pad - (K - 1) : xo0 - pad;
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    const int ys = GRAD ? yo + pad - kh : yo + kh - pad;
                    if (ys < 0 || ys >= Hsrc) continue;
                    float wv[K][EPV];
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        const float* wp = d.w + (long)(kh * K + kw) * d.C + c;
                        const float4 w0 = *(const float4*)wp;
                        wv[kw][0] = w0.x; wv[kw][1] = w0.y; wv[kw][2] = w0.z; wv[kw][3] = w0.w;
                        if (EPV == 8) {
                            const float4 w1 = *(const float4*)(wp + 4);
                            wv[kw][4] = w1.x; wv[kw][5] = w1.y; wv[kw][6] = w1.z; wv[kw][7] = w1.w;
                        }
                    }
                    const T* srow = x + ((long)b * Hsrc + ys) * Wsrc * ld_src + c;
                    // the NS source vectors of the row: unconditional loads from clamped columns, zeroed afterwards -- a
                    // branch around each load made hipcc wait for every one of them on the spot (5x5: 600 GB/s)
                    uint4 raw[NS];
                    // (5x5: keep the batches of different kernel rows apart -- hoisted together they need 256 VGPRs, one
                    // wave per SIMD; one row's eight loads in flight at 2-3 waves per SIMD is the better trade)
                    if (K >= 5) asm volatile("" ::: "memory");
#pragma unroll
                    for (int sl = 0; sl < NS; ++sl) {
                        const int xs = xbase + sl;
                        const int xc = xs < 0 ? 0 : (xs >= Wsrc ? Wsrc - 1 : xs);
                        raw[sl] = *(const uint4*)(srow + (long)xc * ld_src);
                    }
#pragma unroll
                    for (int sl = 0; sl < NS; ++sl) {
                        const int xs = xbase + sl;
                        const bool in = xs >= 0 && xs < Wsrc;
                        float xv[EPV];
                        vec_unpack<T>(raw[sl], xv);
#pragma unroll
                        for (int j = 0; j < EPV; ++j) xv[j] = in ? xv[j] : 0.f;
#pragma unroll
                        for (int o = 0; o < XT; ++o) {
                            const int kw = GRAD ? o + (K - 1) - sl : sl - o;       // compile-time after unrolling
                            if (kw < 0 || kw >= K) continue;
#pragma unroll
                            for (int j = 0; j < EPV; ++j) acc[o][j] += xv[j] * wv[kw][j];
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < XT; ++o) {
                    const int xo = xo0 + o;
                    if (xo >= Wout) break;
                    if (stats) {
#pragma unroll
                        for (int j = 0; j < EPV; ++j) { s1[j] += acc[o][j]; s2[j] += acc[o][j] * acc[o][j]; }
                    }
                    T* yp = y + ((long)row * Wout + xo) * ld_dst + c;
                    if (accum) {
                        float old[EPV];
                        vec_unpack<T>(*(const uint4*)yp, old);
#pragma unroll
                        for (int j = 0; j < EPV; ++j) acc[o][j] += old[j];
                    }
                    *(uint4*)yp = vec_pack<T>(acc[o]);
                }
            }
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

// LDS-tiled stride-1 depthwise conv (bf16; forward, and the data gradient as the same correlation with the flipped kernel
// and pad' = K - 1 - pad).  A workgroup stages the input patch of a TH x 16 pixel output tile for CT <= 8 channel vectors
// in LDS -- every thread issues all of its (<= 8) 16-byte loads before anything waits, ONE memory round trip per tile
// -- and the k x k taps are then ds_read_b128s.  The register-strip kernel above re-reads the patch through L1 in K
// dependent batches per strip (one round trip per kernel row): 1.1-1.9 TB/s on the MobileNetV3 layers, latency-bound.
// Thread = (channel vector ct, strip of XT = 4 pixels, tile row); tile [rows][cols][S = CT|1] 16-byte slots (odd slot
// stride: <= 2-way bank conflicts, irrelevant next to the VALU work); weights [K*K][CT*8] fp32 behind the tile.
template <int K, bool GRAD>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(DykDwDesc d, int CT, int groups, unsigned m_ct, int tiles_x,
                                                          int tiles_y) {
    using T = bf16_t;
    constexpr int EPV = 8, TW = 16, XT = 4, SPR = TW / XT, NS = XT + K - 1, IC = TW + K - 1;
    constexpr int NLD = K == 5 ? 8 : 6;                       // staging loads per thread (host checks the tile fits)
    extern __shared__ uint4 smem[];
    const int S = CT | 1;
    const int RT = (256 / CT) / SPR, TH = RT, IR = TH + K - 1;
    float* wl = (float*)(smem + IR * IC * S);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = bid % groups; bid /= groups;
    const int txi = bid % tiles_x; bid /= tiles_x;
    const int tyi = bid % tiles_y;
    const int b = bid / tiles_y;
    const int tile_id = (b * tiles_y + tyi) * tiles_x + txi;
    const int cv0 = grp * CT, CV = d.C / EPV;
    const T* __restrict__ x = GRAD ? (const T*)d.y : (const T*)d.x;     // tensor read
    T* __restrict__ y = GRAD ? (T*)d.x : (T*)d.y;                       // tensor written
    const int ld_src = GRAD ? d.ldy : d.ldx, ld_dst = GRAD ? d.ldx : d.ldy;
    const int padp = GRAD ? K - 1 - d.pad : d.pad;
    const int Hout = GRAD ? d.Hi : d.Ho, Wout = GRAD ? d.Wi : d.Wo;
    const int Hsrc = GRAD ? d.Ho : d.Hi, Wsrc = GRAD ? d.Wo : d.Wi;
    const int y0 = tyi * TH, x0 = txi * TW;
    const int tid = threadIdx.x;
    {   // ---- stage the patch: all loads first (clamped addresses), then the LDS stores (zeros outside the image)
        const int nvec = IR * IC * CT;
        uint4 v[NLD];
        int dst[NLD], cts[NLD];
        unsigned okm = 0;
        const T* img = x + (long)b * Hsrc * Wsrc * ld_src + (long)cv0 * EPV;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const unsigned idx = tid + i * 256;
            const unsigned q = CT == 1 ? idx : __umulhi(idx, m_ct);     // idx / CT
            const int ct = (int)(idx - q * CT);
            const int row = (int)(q / IC), col = (int)(q - (unsigned)row * IC);
            const int ys = y0 + row - padp, xs = x0 + col - padp;
            const bool ok = idx < (unsigned)nvec && ys >= 0 && ys < Hsrc && xs >= 0 && xs < Wsrc && cv0 + ct < CV;
            const int yc = ys < 0 ? 0 : (ys >= Hsrc ? Hsrc - 1 : ys), xc = xs < 0 ? 0 : (xs >= Wsrc ? Wsrc - 1 : xs);
            const int cc = cv0 + ct < CV ? ct : 0;
            v[i] = *(const uint4*)(img + ((long)yc * Wsrc + xc) * ld_src + cc * EPV);
            okm |= ok ? (1u << i) : 0u;
            cts[i] = cc;
            dst[i] = idx < (unsigned)nvec ? (row * IC + col) * S + ct : -1;
        }
        // weights of this channel group, tap order flipped for the data gradient
        for (int i = tid; i < K * K * CT * 2; i += 256) {
            const int tap = i / (CT * 2), rem = i - tap * (CT * 2);
            const int cch = cv0 * EPV + rem * 4;
            const int tsrc = GRAD ? K * K - 1 - tap : tap;
            float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cch < d.C) w4 = *(const float4*)(d.w + (long)tsrc * d.C + cch);
            *(float4*)(wl + (tap * CT * 2 + rem) * 4) = w4;
        }
        if (!GRAD && d.pre != nullptr) {
            // DykDwDesc.pre: the patch holds the RAW output of the expansion conv; z = dtype(act(scale * u + shift)) is formed
            // here, between the loads (all in flight) and the LDS stores.  scale | shift of this workgroup's channel vectors
            // go through LDS (behind the weights): a staged vector belongs to any of the CT vectors
            float* pl = wl + K * K * CT * 8;
            for (int i = tid; i < CT * 4; i += 256) {                 // i = (which, ct, half)
                const int which = i / (CT * 2), rem = i - which * (CT * 2);
                const int cch = cv0 * EPV + rem * 4;
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cch < d.C) q = *(const float4*)(d.pre + (long)which * d.C + cch);
                *(float4*)(pl + i * 4) = q;
            }
            __syncthreads();
            // (ONE switch around the whole pass: with the activation switch inside the unrolled loops the transform is all branches)
            auto xform = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    float u[EPV];
                    vec_unpack<T>(v[i], u);
                    const float4 s0 = *(const float4*)(pl + cts[i] * 8), s1 = *(const float4*)(pl + cts[i] * 8 + 4);
                    const float4 h0 = *(const float4*)(pl + CT * 8 + cts[i] * 8), h1 = *(const float4*)(pl + CT * 8 + cts[i] * 8 + 4);
                    const float sc[EPV] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    const float sh[EPV] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                    for (int j = 0; j < EPV; ++j) u[j] = act_fwd_c<ACT>(u[j] * sc[j] + sh[j], d.pre_act);
                    v[i] = vec_pack<T>(u);
                }
            };
            switch (d.pre_act) {
            case DYK_ACT_RELU: xform(std::integral_constant<int, DYK_ACT_RELU>{}); break;
            case DYK_ACT_RELU6: xform(std::integral_constant<int, DYK_ACT_RELU6>{}); break;
            case DYK_ACT_HSWISH: xform(std::integral_constant<int, DYK_ACT_HSWISH>{}); break;
            case DYK_ACT_LEAKY: xform(std::integral_constant<int, DYK_ACT_LEAKY>{}); break;
            default: xform(std::integral_constant<int, -1>{}); break;
            }
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (!((okm >> i) & 1u)) v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (dst[i] >= 0) smem[dst[i]] = v[i];
        }
    }
    __syncthreads();
    const unsigned sl = CT == 1 ? (unsigned)tid : __umulhi((unsigned)tid, m_ct);
    const int ct = tid - (int)sl * CT;
    const int strip = sl % SPR, r = sl / SPR;
    const bool active = r < RT && cv0 + ct < CV;
    const bool accum = d.flags & DYK_EW_ACCUM;
    const bool bnbwd = GRAD && d.res != nullptr;               // fused BatchNorm-backward reduce of the producer of x
    const bool stats = ((!GRAD) && d.stats != nullptr) || bnbwd;
    float s1[EPV], s2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s1[j] = s2[j] = 0.f;
    if (active) {
        float acc[XT][EPV];
#pragma unroll
        for (int o = 0; o < XT; ++o)
#pragma unroll
            for (int j = 0; j < EPV; ++j) acc[o][j] = 0.f;
        // the producer's raw conv output at this thread's four pixels: requested now, needed after the tap loop
        uint4 uraw[XT];
        float bsc[EPV], bsh[EPV], bmu[EPV], brs[EPV];
        if (GRAD && bnbwd) {
            const int c = (cv0 + ct) * EPV;
#pragma unroll
            for (int j = 0; j < EPV; j += 4) {
                const float4 q0 = *(const float4*)(d.bn + c + j), q1 = *(const float4*)(d.bn + d.C + c + j);
                const float4 q2 = *(const float4*)(d.bn + 2 * d.C + c + j), q3 = *(const float4*)(d.bn + 3 * d.C + c + j);
                bsc[j] = q0.x; bsc[j + 1] = q0.y; bsc[j + 2] = q0.z; bsc[j + 3] = q0.w;
                bsh[j] = q1.x; bsh[j + 1] = q1.y; bsh[j + 2] = q1.z; bsh[j + 3] = q1.w;
                bmu[j] = q2.x; bmu[j + 1] = q2.y; bmu[j + 2] = q2.z; bmu[j + 3] = q2.w;
                brs[j] = q3.x; brs[j + 1] = q3.y; brs[j + 2] = q3.z; brs[j + 3] = q3.w;
            }
            const int yo = y0 + r < Hout ? y0 + r : Hout - 1;
#pragma unroll
            for (int o = 0; o < XT; ++o) {
                const int xo = x0 + strip * XT + o;
                uraw[o] = *(const uint4*)((const T*)d.res + (((long)b * Hout + yo) * Wout + (xo < Wout ? xo : Wout - 1)) * d.ldr +
                                          (long)(cv0 + ct) * EPV);
            }
        }
        // (kernel rows NOT unrolled: hoisted together their LDS reads need 500 VGPRs -- one wave per SIMD)
#pragma unroll 1
        for (int kh = 0; kh < K; ++kh) {
            float wv[K][EPV];
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const float4 w0 = *(const float4*)(wl + ((kh * K + kw) * CT + ct) * 8);
                const float4 w1 = *(const float4*)(wl + ((kh * K + kw) * CT + ct) * 8 + 4);
                wv[kw][0] = w0.x; wv[kw][1] = w0.y; wv[kw][2] = w0.z; wv[kw][3] = w0.w;
                wv[kw][4] = w1.x; wv[kw][5] = w1.y; wv[kw][6] = w1.z; wv[kw][7] = w1.w;
            }
            const uint4* prow = smem + ((r + kh) * IC + strip * XT) * S + ct;
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                float xv[EPV];
                vec_unpack<T>(prow[q * S], xv);
#pragma unroll
                for (int o = 0; o < XT; ++o) {
                    const int kw = q - o;                          // compile-time after unrolling
                    if (kw < 0 || kw >= K) continue;
#pragma unroll
                    for (int j = 0; j < EPV; ++j) acc[o][j] += xv[j] * wv[kw][j];
                }
            }
        }
        const int yo = y0 + r;
        if (yo < Hout) {
#pragma unroll
            for (int o = 0; o < XT; ++o) {
                const int xo = x0 + strip * XT + o;
                if (xo >= Wout) break;
                if (GRAD && bnbwd) {
                    float u[EPV];
                    vec_unpack<T>(uraw[o], u);
                    // the gradient is rounded to the storage dtype first (what an unfused reduce pass would read back)
                    const uint4 pk = vec_pack<T>(acc[o]);
                    vec_unpack<T>(pk, acc[o]);
#pragma unroll
                    for (int j = 0; j < EPV; ++j) {
                        const float da = acc[o][j] * act_bwd(d.act, u[j] * bsc[j] + bsh[j]);
                        s1[j] += da;
                        s2[j] += da * ((u[j] - bmu[j]) * brs[j]);
                        acc[o][j] = da;
                    }
                } else if (stats) {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) { s1[j] += acc[o][j]; s2[j] += acc[o][j] * acc[o][j]; }
                }
                T* yp = y + (((long)b * Hout + yo) * Wout + xo) * ld_dst + (long)(cv0 + ct) * EPV;
                if (accum) {
                    float old[EPV];
                    vec_unpack<T>(*(const uint4*)yp, old);
#pragma unroll
                    for (int j = 0; j < EPV; ++j) acc[o][j] += old[j];
                }
                *(uint4*)yp = vec_pack<T>(acc[o]);
            }
        }
    }
    if (stats) {
        // workgroup reduction in a fixed order (same sums from run to run), spread over all threads: thread (part, ct, j)
        // adds every parts-th strip lane of output (ct, j), the first 16 * CT threads fold the parts and issue ONE
        // fp64 atomic each -- eight threads walking 32 lanes and issuing 16 atomics apiece cost ~25 us per launch
        __syncthreads();                                       // everyone is done with the patch: reuse it
        float* red = (float*)smem;
        float* red2 = red + 256 * 16;
        float* mine = red + tid * 16;
#pragma unroll
        for (int j = 0; j < EPV; ++j) { mine[j] = s1[j]; mine[8 + j] = s2[j]; }
        __syncthreads();
        const int nout = 16 * CT, parts = 256 / nout, nsl = 256 / CT;
        const int part = tid / nout, o = tid - part * nout;
        if (part < parts) {
            float t = 0.f;
            for (int q = part; q < nsl; q += parts) t += red[q * nout + o];
            red2[part * nout + o] = t;
        }
        __syncthreads();
        if (tid < nout) {
            float t = 0.f;
            for (int p = 0; p < parts; ++p) t += red2[p * nout + tid];
            const int cto = tid >> 4, j = tid & 15;
            if (cv0 + cto < CV) {
                double* st = d.stats + (size_t)((unsigned)tile_id % (unsigned)(d.stats_slots > 0 ? d.stats_slots : 1)) * 2 * d.C;
                atomicAdd(st + (j < 8 ? 0 : d.C) + (cv0 + cto) * EPV + (j & 7), (double)t);
            }
        }
    }
}

template <int K, bool GRAD>
int launch_dw_tile(const DykDwDesc* d, hipStream_t stream) {
    constexpr int TW = 16, IC = TW + K - 1, NLD = K == 5 ? 8 : 6;
    const int CV = d->C / 8;
    const int groups = (CV + 7) / 8, CT = (CV + groups - 1) / groups;
    const int RT = (256 / CT) / 4, IR = RT + K - 1, S = CT | 1;
    if (IR * IC * CT > NLD * 256) return DYK_ERR_UNSUPPORTED;
    const int Hout = GRAD ? d->Hi : d->Ho, Wout = GRAD ? d->Wi : d->Wo;
    const int tiles_x = (Wout + TW - 1) / TW, tiles_y = (Hout + RT - 1) / RT;
    size_t lds = (size_t)IR * IC * S * 16 + (size_t)K * K * CT * 8 * 4 + (size_t)CT * 16 * 4;   // patch | weights | pre scale, shift
    if (lds < 256 * 16 * 4 + 2048) lds = 256 * 16 * 4 + 2048;            // the statistics reduction reuses the patch
    const unsigned m_ct = (unsigned)(0xFFFFFFFFu / (unsigned)CT) + 1u;   // idx / CT == umulhi(idx, m_ct) for idx < 2^32 / CT
    const long nblk = (long)groups * tiles_x * tiles_y * d->B;
    if (nblk >= (1L << 31) || lds > 64 * 1024) return DYK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((dwconv_tile_kernel<K, GRAD>), dim3((unsigned)nblk), dim3(256), lds, stream, *d, CT, groups,
                       m_ct, tiles_x, tiles_y);
    return DYK_OK;
}
inline bool dw_tile_on() { return true; }

// Stride-2 fast path (MobileNet down-sampling layers): K and the stride are compile-time, so the taps of a pixel are a
// fixed, unrolled set whose loads are issued together -- clamped coordinates, values zeroed afterwards (the generic kernel
// above walks runtime loops with a branch around every load: one memory round trip per tap).
//   forward : y[yo, xo] = sum_{kh,kw} x[2 yo + kh - pad, 2 xo + kw - pad] w[kh, kw]              (K*K loads)
//   gradient: dx[yi, xi] = sum over the taps of yi's / xi's parity, kh = (yi + pad) % 2 + 2 a:
//             dy[(yi + pad - kh) / 2, (xi + pad - kw) / 2] w[kh, kw]                              (ceil(K/2)^2 loads)
template <typename T, int K, bool GRAD>
__global__ __launch_bounds__(256) void dwconv_s2_kernel(DykDwDesc d, int CVB) {
    constexpr int EPV = ElemTraits<T>::EPV;
    constexpr int KT = GRAD ? (K + 1) / 2 : K;                 // taps per dimension a pixel touches
    __shared__ float red[256 * 2 * 8];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int cv = blockIdx.x * CVB + tx;
    const int c = cv * EPV;
    const bool active = c < d.C;
    const T* __restrict__ x = GRAD ? (const T*)d.y : (const T*)d.x;     // tensor read
    T* __restrict__ y = GRAD ? (T*)d.x : (T*)d.y;                       // tensor written
    const int ld_src = GRAD ? d.ldy : d.ldx, ld_dst = GRAD ? d.ldx : d.ldy;
    const int pad = d.pad;
    const int Hout = GRAD ? d.Hi : d.Ho, Wout = GRAD ? d.Wi : d.Wo;
    const int Hsrc = GRAD ? d.Ho : d.Hi, Wsrc = GRAD ? d.Wo : d.Wi;
    const bool accum = d.flags & DYK_EW_ACCUM;
    const bool stats = (!GRAD) && d.stats != nullptr;
    float s1[EPV], s2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s1[j] = s2[j] = 0.f;
    if (active) {
        const int nrows = d.B * Hout;
        for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
            const int b = row / Hout, yo = row - b * Hout;
            const int kh0 = GRAD ? (yo + pad) & 1 : 0;
            for (int xo = ty; xo < Wout; xo += PY) {
                const int kw0 = GRAD ? (xo + pad) & 1 : 0;
                uint4 raw[KT][KT];
                bool ok[KT][KT];
#pragma unroll
                for (int a = 0; a < KT; ++a) {
                    const int kh = GRAD ? kh0 + 2 * a : a;
                    const int ys = GRAD ? (yo + pad - kh) >> 1 : yo * 2 + kh - pad;
                    const bool yok = kh < K && ys >= 0 && ys < Hsrc && (!GRAD || yo + pad - kh >= 0);
                    const int yc = ys < 0 ? 0 : (ys >= Hsrc ? Hsrc - 1 : ys);
#pragma unroll
                    for (int e = 0; e < KT; ++e) {
                        const int kw = GRAD ? kw0 + 2 * e : e;
                        const int xs = GRAD ? (xo + pad - kw) >> 1 : xo * 2 + kw - pad;
                        ok[a][e] = yok && kw < K && xs >= 0 && xs < Wsrc && (!GRAD || xo + pad - kw >= 0);
                        const int xc = xs < 0 ? 0 : (xs >= Wsrc ? Wsrc - 1 : xs);
                        raw[a][e] = *(const uint4*)(x + (((long)b * Hsrc + yc) * Wsrc + xc) * ld_src + c);
                    }
                }
                float acc[EPV];
#pragma unroll
                for (int j = 0; j < EPV; ++j) acc[j] = 0.f;
#pragma unroll
                for (int a = 0; a < KT; ++a)
#pragma unroll
                    for (int e = 0; e < KT; ++e) {
                        const int kh = GRAD ? kh0 + 2 * a : a, kw = GRAD ? kw0 + 2 * e : e;
                        const int t = (kh < K && kw < K) ? kh * K + kw : 0;
                        const float* wp = d.w + (long)t * d.C + c;
                        const float4 w0 = *(const float4*)wp;
                        float wv[8] = {w0.x, w0.y, w0.z, w0.w, 0.f, 0.f, 0.f, 0.f};
                        if (EPV == 8) {
                            const float4 w1 = *(const float4*)(wp + 4);
                            wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
                        }
                        float xv[EPV];
                        vec_unpack<T>(raw[a][e], xv);
                        const bool in = ok[a][e];
#pragma unroll
                        for (int j = 0; j < EPV; ++j) acc[j] += in ? xv[j] * wv[j] : 0.f;
                    }
                if (stats) {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
                }
                T* yp = y + ((long)row * Wout + xo) * ld_dst + c;
                if (accum) {
                    float old[EPV];
                    vec_unpack<T>(*(const uint4*)yp, old);
#pragma unroll
                    for (int j = 0; j < EPV; ++j) acc[j] += old[j];
                }
                *(uint4*)yp = vec_pack<T>(acc);
            }
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

// dw[t][c] += sum_p dy[p][c] * x[src(p, t)][c].  grid.z = kernel row kh; a thread keeps the K taps of that row for its
// 8 channels in registers (K*8 accumulators), reads the output-gradient vector of a pixel once and the K input vectors of
// the row from L1 -- one pass over dy per kernel row instead of one per tap, K x fewer workgroup reductions and atomics.
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DykDwDesc d, int CVB) {
    constexpr int EPV = ElemTraits<T>::EPV;
    __shared__ float red[256 * 8];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int cv = blockIdx.x * CVB + tx;
    const int c = cv * EPV;
    const bool active = c < d.C;
    const int kh = blockIdx.z;
    const T* __restrict__ x = (const T*)d.x;
    const T* __restrict__ dy = (const T*)d.y;       // output gradient [B,Ho,Wo,C]
    const long npix = (long)d.B * d.Ho * d.Wo;
    float acc[K][EPV];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int j = 0; j < EPV; ++j) acc[t][j] = 0.f;
    // DykDwDesc.pre: x holds the raw output of the producing conv; z = dtype(act(scale * u + shift)) is formed on load
    const bool pre = d.pre != nullptr;
    float psc[EPV], psh[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) { psc[j] = 1.f; psh[j] = 0.f; }
    if (pre && active) {
#pragma unroll
        for (int j = 0; j < EPV; ++j) { psc[j] = d.pre[c + j]; psh[j] = d.pre[d.C + c + j]; }
    }
    // branch-free for the activations the depthwise blocks use (ReLU, ReLU6, h-swish, leaky, linear):
    //   act(t) = clamp(t, lo, hi) * (mulx ? clamp(t + 3, 0, 6) / 6 : 1)   with leaky as max(t, 0.1 t)
    const int pact = d.pre_act;
    const bool p_generic = pre && !(pact == DYK_ACT_LINEAR || pact == DYK_ACT_RELU || pact == DYK_ACT_RELU6 || pact == DYK_ACT_HSWISH ||
                                    pact == DYK_ACT_LEAKY);
    const float p_lo = (pact == DYK_ACT_RELU || pact == DYK_ACT_RELU6) ? 0.f : -__builtin_inff();
    const float p_hi = pact == DYK_ACT_RELU6 ? 6.f : __builtin_inff();
    const float p_leak = pact == DYK_ACT_LEAKY ? 0.1f : 1.f;
    const bool p_hsw = pact == DYK_ACT_HSWISH;
    auto pre_apply = [&](float (&xv)[EPV]) {
        if (pre) {
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                const float t = xv[j] * psc[j] + psh[j];
                float r = fminf(fmaxf(fmaxf(t, p_leak * t), p_lo), p_hi);
                if (p_hsw) r = t * fminf(fmaxf(t + 3.f, 0.f), 6.f) * (1.f / 6.f);
                if (p_generic) r = act_fwd(pact, t);
                xv[j] = r;
            }
            const uint4 pk = vec_pack<T>(xv);                  // rounded to the storage type: what the separate pass would have stored
            vec_unpack<T>(pk, xv);
        }
    };
    if (active) {
        const int nrows = d.B * d.Ho;
        for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
          const int b = row / d.Ho, yo = row - b * d.Ho;
            const int yi = yo * d.stride + kh - d.pad;
            if (yi < 0 || yi >= d.Hi) continue;
            const T* xrow = x + ((long)b * d.Hi + yi) * d.Wi * d.ldx + c;
            if (d.stride == 1) {
                // strips of 4 output pixels: 4 gradient vectors + (4 + K - 1) input vectors per strip and kernel row
                constexpr int XT = 4;
                for (int xo0 = ty * XT; xo0 < d.Wo; xo0 += PY * XT) {
                    float g[XT][EPV];
                    // all loads of the strip first (clamped columns, values zeroed afterwards): one memory round trip
                    uint4 graw[XT], xraw[XT + K - 1];
#pragma unroll
                    for (int o = 0; o < XT; ++o) {
                        const int xo = xo0 + o < d.Wo ? xo0 + o : d.Wo - 1;
                        graw[o] = *(const uint4*)(dy + ((long)row * d.Wo + xo) * d.ldy + c);
                    }
#pragma unroll
                    for (int sl = 0; sl < XT + K - 1; ++sl) {
                        const int xi = xo0 - d.pad + sl;
                        const int xc = xi < 0 ? 0 : (xi >= d.Wi ? d.Wi - 1 : xi);
                        xraw[sl] = *(const uint4*)(xrow + (long)xc * d.ldx);
                    }
#pragma unroll
                    for (int o = 0; o < XT; ++o) {
                        vec_unpack<T>(graw[o], g[o]);
                        const bool in = xo0 + o < d.Wo;
#pragma unroll
                        for (int j = 0; j < EPV; ++j) g[o][j] = in ? g[o][j] : 0.f;
                    }
#pragma unroll
                    for (int sl = 0; sl < XT + K - 1; ++sl) {
                        const int xi = xo0 - d.pad + sl;
                        const bool in = xi >= 0 && xi < d.Wi;
                        float xv[EPV];
                        vec_unpack<T>(xraw[sl], xv);
                        pre_apply(xv);
#pragma unroll
                        for (int j = 0; j < EPV; ++j) xv[j] = in ? xv[j] : 0.f;
#pragma unroll
                        for (int o = 0; o < XT; ++o) {
                            const int t = sl - o;                  // tap within the kernel row (compile-time after unrolling)
                            if (t < 0 || t >= K) continue;
#pragma unroll
                            for (int j = 0; j < EPV; ++j) acc[t][j] += g[o][j] * xv[j];
                        }
                    }
                }
                continue;
            }
          for (int xo = ty; xo < d.Wo; xo += PY) {
            const long p = (long)row * d.Wo + xo;
            float g[EPV];
            vec_unpack<T>(*(const uint4*)(dy + p * d.ldy + c), g);
            const int xi0 = xo * d.stride - d.pad;
            uint4 xraw[K];
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int xi = xi0 + t;
                xraw[t] = *(const uint4*)(xrow + (long)(xi < 0 ? 0 : (xi >= d.Wi ? d.Wi - 1 : xi)) * d.ldx);
            }
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int xi = xi0 + t;
                const bool in = xi >= 0 && xi < d.Wi;
                float xv[EPV];
                vec_unpack<T>(xraw[t], xv);
                pre_apply(xv);
#pragma unroll
                for (int j = 0; j < EPV; ++j) acc[t][j] += in ? g[j] * xv[j] : 0.f;
            }
          }
        }
    }
    // fold the pixel lanes through LDS, one tap at a time (8 floats per thread)
#pragma unroll
    for (int t = 0; t < K; ++t) {
        float* mine = red + threadIdx.x * 8;
#pragma unroll
        for (int j = 0; j < EPV; ++j) mine[j] = acc[t][j];
        __syncthreads();
        if (ty == 0 && active) {
            float s[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) s[j] = acc[t][j];
            for (int q = 1; q < PY; ++q) {
                const float* o = red + (q * CVB + tx) * 8;
#pragma unroll
                for (int j = 0; j < EPV; ++j) s[j] += o[j];
            }
            if (d.part) {          // plane mode: this workgroup row's own plane, plain stores
                float* pp = d.part + ((long)blockIdx.y * K * K + (kh * K + t)) * d.C + c;
#pragma unroll
                for (int j = 0; j < EPV; ++j) pp[j] = s[j];
            } else {
#pragma unroll
                for (int j = 0; j < EPV; ++j) unsafeAtomicAdd(d.dw + (long)(kh * K + t) * d.C + c + j, s[j]);
            }
        }
        __syncthreads();
    }
}

// LDS-tiled, PERSISTENT stride-1 depthwise weight gradient (bf16, k = 3 | 5; round 5).  The kernel above makes K passes over dy
// and x (grid.z = kernel row) with 12 sixteen-byte loads in flight per thread and three waves per SIMD: ~0.6 MB in flight over the
// chip, 0.5-2 TB/s, waiting 0.64-0.74 of its cycles (tools/dw_pmc.sh).  Here a workgroup walks output tiles of TH x 16 pixels for
// CT <= 8 channel vectors (the geometry of dwconv_tile_kernel): the x patch and the dy tile of the NEXT tile are requested into
// registers (12 loads per thread, every one issued before anything waits) and stay in flight while the current tile is computed
// from LDS -- one pass over both tensors, halo re-reads only.  Thread = (channel vector, kernel row kh, pixel lane): K * 8
// accumulators, a strip of 4 pixels costs 4 + (4 + K - 1) ds_read_b128 for 4 * K * 8 multiply-adds.  Out-of-image patch pixels,
// pixels beyond the output extent and ragged tiles are zeros in LDS, so no tap needs a bounds test.  Workgroup wg of its channel
// group owns plane wg (DykDwDesc.part: plain stores, folded in plane order by dyk_grad_reduce -- same sums from run to run);
// without planes, one fp32 atomic per (tap, channel) and workgroup.
typedef __bf16 dw_bf16x2_t __attribute__((ext_vector_type(2)));
template <int K, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dwconv_wgrad_tile_kernel(DykDwDesc d, int CT, int groups, unsigned m_ct, int tiles_x,
                                                                int tiles_y, int P) {
    using T = bf16_t;
    constexpr int EPV = 8, TW = 16, XT = 4, SPR = TW / XT, NS = XT + K - 1, IC = TW + K - 1;
    constexpr int NLX = K == 5 ? 8 : 6, NLG = 4, NLD = NLX + NLG;
    extern __shared__ uint4 smem[];
    const int S = CT | 1;
    const int TH = (256 / CT) / SPR, IR = TH + K - 1;
    uint4* patch = smem;                                       // [IR][IC][S]  x
    uint4* gtile = smem + IR * IC * S;                         // [TH][TW][S]  dy
    const int tid = threadIdx.x;
    const int grp = blockIdx.x % groups, wg = blockIdx.x / groups;
    const int cv0 = grp * CT, CV = d.C / EPV;
    const T* __restrict__ x = (const T*)d.x;
    const T* __restrict__ dy = (const T*)d.y;
    const int ntiles = d.B * tiles_y * tiles_x;
    // ---- staging roles (the same for every tile): load i of this thread = vector (row, col, ct) of the patch / of the dy tile
    int role[NLD];                                             // row | col << 8 | ct << 16, -1 = none
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const bool isx = i < NLX;
        const unsigned idx = tid + (isx ? i : i - NLX) * 256;
        const unsigned q = CT == 1 ? idx : __umulhi(idx, m_ct);
        const int ct = (int)(idx - q * CT);
        const int w = isx ? IC : TW;
        const int row = (int)(q / w), col = (int)(q - (unsigned)row * w);
        const bool valid = idx < (unsigned)((isx ? IR * IC : TH * TW) * CT) && cv0 + ct < CV;
        role[i] = valid ? (row | (col << 8) | (ct << 16)) : -1;
    }
    uint4 v[NLD];
    unsigned okm = 0;
    auto request = [&](int t) __attribute__((always_inline)) {
        const int txi = t % tiles_x, r1 = t / tiles_x;
        const int tyi = r1 % tiles_y, b = r1 / tiles_y;
        const int y0 = tyi * TH, x0 = txi * TW;
        const T* ximg = x + (long)b * d.Hi * d.Wi * d.ldx + (long)cv0 * EPV;
        const T* gimg = dy + (long)b * d.Ho * d.Wo * d.ldy + (long)cv0 * EPV;
        okm = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ro = role[i] < 0 ? 0 : role[i];
            const int row = ro & 0xff, col = (ro >> 8) & 0xff, ct = ro >> 16;
            if (i < NLX) {
                const int ys = y0 + row - d.pad, xs = x0 + col - d.pad;
                const bool ok = role[i] >= 0 && ys >= 0 && ys < d.Hi && xs >= 0 && xs < d.Wi;
                const int yc = ys < 0 ? 0 : (ys >= d.Hi ? d.Hi - 1 : ys), xc = xs < 0 ? 0 : (xs >= d.Wi ? d.Wi - 1 : xs);
                v[i] = *(const uint4*)(ximg + ((long)yc * d.Wi + xc) * d.ldx + ct * EPV);
                okm |= ok ? (1u << i) : 0u;
            } else {
                const int ys = y0 + row, xs = x0 + col;
                const bool ok = role[i] >= 0 && ys < d.Ho && xs < d.Wo;
                const int yc = ys >= d.Ho ? d.Ho - 1 : ys, xc = xs >= d.Wo ? d.Wo - 1 : xs;
                v[i] = *(const uint4*)(gimg + ((long)yc * d.Wo + xc) * d.ldy + ct * EPV);
                okm |= ok ? (1u << i) : 0u;
            }
        }
    };
    // ---- compute role
    const int cct = tid % CT, rest = tid / CT;
    const int kh = rest % K, lane = rest / K;
    const int NL = (256 / CT) / K;
    const bool active = lane < NL && cv0 + cct < CV;
    float acc[K][EPV];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int j = 0; j < EPV; ++j) acc[t][j] = 0.f;

    // DykDwDesc.pre (PRE): x holds the RAW output of the producing conv; z = dtype(act(scale * u + shift)) is formed between the
    // loads and the LDS stores -- the patch then holds exactly what the separate normalise pass would have stored.  Branch-free
    // for ReLU / ReLU6 / h-swish / leaky / linear: act(t) = clamp(max(t, leak * t), lo, hi), h-swish t * clamp(t + 3, 0, 6) / 6
    float* pl = (float*)(smem + (IR * IC + TH * TW) * S);      // [2][CT * 8] scale | shift of this channel group
    const int pact = d.pre_act;
    const bool p_generic = !(pact == DYK_ACT_LINEAR || pact == DYK_ACT_RELU || pact == DYK_ACT_RELU6 || pact == DYK_ACT_HSWISH ||
                             pact == DYK_ACT_LEAKY);
    const float p_lo = (pact == DYK_ACT_RELU || pact == DYK_ACT_RELU6) ? 0.f : -__builtin_inff();
    const float p_hi = pact == DYK_ACT_RELU6 ? 6.f : __builtin_inff();
    const float p_leak = pact == DYK_ACT_LEAKY ? 0.1f : 1.f;
    const bool p_hsw = pact == DYK_ACT_HSWISH;
    if constexpr (PRE) {
        for (int i = tid; i < 2 * CT * 8; i += 256) {
            const int which = i / (CT * 8), rem = i - which * (CT * 8);
            const int cch = cv0 * EPV + rem;
            pl[i] = cch < d.C ? d.pre[(long)which * d.C + cch] : 0.f;
        }
    }

    int t = wg;
    if (t < ntiles) request(t);
    for (; t < ntiles; t += P) {
        __syncthreads();                                       // the previous tile's LDS reads are done (and, first trip, pl is there)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (role[i] < 0) continue;
            const int row = role[i] & 0xff, col = (role[i] >> 8) & 0xff, ct = role[i] >> 16;
            if constexpr (PRE) {
                if (i < NLX) {
                    float u[EPV];
                    vec_unpack<T>(v[i], u);
#pragma unroll
                    for (int j = 0; j < EPV; ++j) {
                        const float tt = u[j] * pl[ct * 8 + j] + pl[CT * 8 + ct * 8 + j];
                        float r = fminf(fmaxf(fmaxf(tt, p_leak * tt), p_lo), p_hi);
                        if (p_hsw) r = tt * fminf(fmaxf(tt + 3.f, 0.f), 6.f) * (1.f / 6.f);
                        if (p_generic) r = act_fwd(pact, tt);
                        u[j] = r;
                    }
                    v[i] = vec_pack<T>(u);
                }
            }
            const uint4 z = ((okm >> i) & 1u) ? v[i] : make_uint4(0u, 0u, 0u, 0u);
            if (i < NLX) patch[(row * IC + col) * S + ct] = z;
            else gtile[(row * TW + col) * S + ct] = z;
        }
        __syncthreads();
        if (t + P < ntiles) request(t + P);                    // in flight while this tile is computed
        if (active) {
#pragma unroll 1
            for (int s = lane; s < TH * SPR; s += NL) {
                const int r = s / SPR, strip = s - r * SPR;
                const uint4* grow = gtile + (r * TW + strip * XT) * S + cct;
                const uint4* prow = patch + ((r + kh) * IC + strip * XT) * S + cct;
                // v_dot2c_f32_bf16: acc += x.lo * g.lo + x.hi * g.hi on the PACKED pairs -- with one half of g masked to zero
                // it is the fp32 multiply-add of one channel (a bf16 x bf16 product is exact in fp32: one rounding, as the
                // fma), and the x vectors need no unpacking: 8 VALU ops fewer per input vector.
                // PROPERTY: the unmasked neighbour half of x still meets the zero half of g, so a NON-FINITE x in
                // channel c ^ 1 turns the gradient of channel c into NaN (0 * Inf) where the row kernel (fp32 fma per channel,
                // stride 2 / fp32 shapes) keeps the channels apart; finite inputs give the same sums as that kernel up to the
                // summation order.  Masking x as well would put the 8 VALU ops per vector back that this form saves; a
                // non-finite activation already poisons the step's loss (GradScaler skips it), so the forms are not unified.
                unsigned gm[XT][EPV];
#pragma unroll
                for (int o = 0; o < XT; ++o) {
                    const uint4 gr = grow[o * S];
                    const unsigned gw[4] = {gr.x, gr.y, gr.z, gr.w};
#pragma unroll
                    for (int m = 0; m < 4; ++m) { gm[o][2 * m] = gw[m] & 0xffffu; gm[o][2 * m + 1] = gw[m] & 0xffff0000u; }
                }
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    const uint4 xr = prow[q * S];
                    const unsigned xw[4] = {xr.x, xr.y, xr.z, xr.w};
#pragma unroll
                    for (int o = 0; o < XT; ++o) {
                        const int tp = q - o;
                        if (tp < 0 || tp >= K) continue;
#pragma unroll
                        for (int j = 0; j < EPV; ++j)
                            acc[tp][j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dw_bf16x2_t, xw[j >> 1]),
                                                                         __builtin_bit_cast(dw_bf16x2_t, gm[o][j]), acc[tp][j], false);
                    }
                }
            }
        }
    }
    // ---- fold the pixel lanes through LDS in lane order, one tap at a time
    __syncthreads();
    float* red = (float*)smem;
#pragma unroll
    for (int tp = 0; tp < K; ++tp) {
        float* mine = red + tid * 8;
#pragma unroll
        for (int j = 0; j < EPV; ++j) mine[j] = acc[tp][j];
        __syncthreads();
        if (active && lane == 0) {
            float sum[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) sum[j] = acc[tp][j];
            for (int q = 1; q < NL; ++q) {
                const float* o = red + ((q * K + kh) * CT + cct) * 8;
#pragma unroll
                for (int j = 0; j < EPV; ++j) sum[j] += o[j];
            }
            const int c = (cv0 + cct) * EPV;
            if (d.part) {
                float* pp = d.part + ((long)wg * K * K + (kh * K + tp)) * d.C + c;
                if (((size_t)pp & 15) == 0) {
                    *(float4*)pp = make_float4(sum[0], sum[1], sum[2], sum[3]);
                    *(float4*)(pp + 4) = make_float4(sum[4], sum[5], sum[6], sum[7]);
                } else {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) pp[j] = sum[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < EPV; ++j) unsafeAtomicAdd(d.dw + (long)(kh * K + tp) * d.C + c + j, sum[j]);
            }
        }
        __syncthreads();
    }
}

struct DwWgTile { int CT, groups, tiles_x, tiles_y, P; unsigned m_ct; size_t lds; };
inline int dw_wgrad_tile_target() {
    return 512;                                                // workgroups per launch (two resident per CU)
}
// geometry of the tiled weight gradient for this problem; false = not eligible (the row kernel runs)
inline bool dw_wgrad_tile_cfg(const DykDwDesc* d, DwWgTile* c) {
    const int K = d->k;
    if (d->dtype != DYK_BF16 || d->stride != 1 || (K != 3 && K != 5) || dw_wgrad_tile_target() <= 0) return false;
    const int CV = d->C / 8, NLX = K == 5 ? 8 : 6;
    c->groups = (CV + 7) / 8;
    c->CT = (CV + c->groups - 1) / c->groups;
    const int TH = (256 / c->CT) / 4, IR = TH + K - 1, IC = 16 + K - 1, S = c->CT | 1;
    if (IR * IC * c->CT > NLX * 256 || (256 / c->CT) / K < 1 || IR > 255) return false;
    c->tiles_x = (d->Wo + 15) / 16;
    c->tiles_y = (d->Ho + TH - 1) / TH;
    const long ntiles = (long)d->B * c->tiles_x * c->tiles_y;
    if (ntiles >= (1L << 30)) return false;
    // (two workgroups per CU; three for the 3x3 kernel at 164 VGPRs measured equal: 70 / 87 / 20 / 29 us against 65 / 88 / 20 / 29)
    long p0 = dw_wgrad_tile_target() / c->groups;
    if (p0 < 1) p0 = 1;
    if (p0 > ntiles) p0 = ntiles;
    const long rounds = (ntiles + p0 - 1) / p0;
    c->P = (int)((ntiles + rounds - 1) / rounds);
    c->m_ct = (unsigned)(0xFFFFFFFFu / (unsigned)c->CT) + 1u;
    c->lds = (size_t)(IR * IC + TH * 16) * S * 16 + (size_t)2 * c->CT * 8 * 4;             // patch | dy tile | pre scale, shift
    if (c->lds < 256 * 8 * 4) c->lds = 256 * 8 * 4;
    return c->lds <= 64 * 1024;
}
inline int launch_dw_wgrad_tile(const DykDwDesc* d, const DwWgTile& c, hipStream_t s) {
    const dim3 grid((unsigned)(c.groups * c.P));
#define DYK_DWW(KK, PP) hipLaunchKernelGGL((dwconv_wgrad_tile_kernel<KK, PP>), grid, dim3(256), c.lds, s, *d, c.CT, c.groups, c.m_ct, c.tiles_x, c.tiles_y, c.P)
    if (d->k == 3) { if (d->pre) DYK_DWW(3, true); else DYK_DWW(3, false); }
    else { if (d->pre) DYK_DWW(5, true); else DYK_DWW(5, false); }
#undef DYK_DWW
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

int check_dw(const DykDwDesc* d) {
    if (!d || !d->x || !d->y || d->B <= 0 || d->C <= 0 || d->k <= 0 || d->k > 7 || d->stride <= 0) return DYK_ERR_ARG;
    if (d->dtype != DYK_BF16 && d->dtype != DYK_F32) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    if (d->C % epv || d->ldx % epv || d->ldy % epv) return DYK_ERR_ARG;
    if (d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0) return DYK_ERR_ARG;
    return DYK_OK;
}

// grid.x = channel-vector groups, grid.y = image rows (strided), capped at cap_blocks workgroups
inline int grid2d(int CV, long nrows, int* gx, int* gy, int cap_blocks) {
    int CVB = 1;
    while (CVB < CV && CVB < 32) CVB <<= 1;
    *gx = (CV + CVB - 1) / CVB;
    long g = nrows;
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
    const int CVB = grid2d(d->C / epv, (long)d->B * d->Ho, &gx, &gy, 4096);
    if (d->stride == 1 && (d->k == 3 || d->k == 5) && d->dtype == DYK_BF16 && dw_tile_on()) {
        const int rt = d->k == 3 ? launch_dw_tile<3, false>(d, (hipStream_t)stream) : launch_dw_tile<5, false>(d, (hipStream_t)stream);
        if (rt != DYK_ERR_UNSUPPORTED) { DYK_LAUNCH_CHECK(); return rt; }
    }
    if (d->pre) return DYK_ERR_UNSUPPORTED;        // normalise + activation on load: the LDS-tiled kernel only (dyk_dwconv_tile_ok)
    if (d->stride == 1 && (d->k == 3 || d->k == 5) && d->dtype == DYK_BF16) {
        if (d->k == 3) hipLaunchKernelGGL((dwconv_strip_kernel<bf16_t, 3, false>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
        else hipLaunchKernelGGL((dwconv_strip_kernel<bf16_t, 5, false>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    } else if (d->dtype == DYK_BF16)      // (stride 2 forward: the generic kernel measured faster than dwconv_s2_kernel<.., false>)
        hipLaunchKernelGGL((dwconv_kernel<bf16_t, false>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    else
        hipLaunchKernelGGL((dwconv_kernel<float, false>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_dwconv_tile_ok(const DykDwDesc* d) {
    const int rc = check_dw(d);
    if (rc) return rc;
    if (!(d->stride == 1 && (d->k == 3 || d->k == 5) && d->dtype == DYK_BF16 && dw_tile_on())) return 0;
    const int K = d->k, TW = 16, IC = TW + K - 1, NLD = K == 5 ? 8 : 6;
    const int CV = d->C / 8;
    const int groups = (CV + 7) / 8, CT = (CV + groups - 1) / groups;
    const int RT = (256 / CT) / 4, IR = RT + K - 1, S = CT | 1;
    if (IR * IC * CT > NLD * 256) return 0;
    const size_t lds = (size_t)IR * IC * S * 16 + (size_t)K * K * CT * 8 * 4 + (size_t)CT * 16 * 4;
    const long nblk = (long)groups * ((d->Wo + TW - 1) / TW) * ((d->Ho + RT - 1) / RT) * d->B;
    return (nblk < (1L << 31) && lds <= 64 * 1024) ? 1 : 0;
}

extern "C" int dyk_dwconv_dgrad(const DykDwDesc* d, void* stream) {
    const int rc = check_dw(d);
    if (rc) return rc;
    if (!d->w) return DYK_ERR_ARG;
    if (d->res) {                    // fused BatchNorm-backward reduce: LDS-tiled kernel only
        if (!d->bn || !d->stats || d->stats_slots <= 0 || d->ldr < d->C || d->ldr % 8 || ((uintptr_t)d->res % 16)) return DYK_ERR_ARG;
        if (d->stride != 1 || (d->k != 3 && d->k != 5) || d->dtype != DYK_BF16 || (d->flags & DYK_EW_ACCUM)) return DYK_ERR_UNSUPPORTED;
        const int rt = d->k == 3 ? launch_dw_tile<3, true>(d, (hipStream_t)stream) : launch_dw_tile<5, true>(d, (hipStream_t)stream);
        if (rt == DYK_OK) DYK_LAUNCH_CHECK();
        return rt;
    }
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    int gx, gy;
    const int CVB = grid2d(d->C / epv, (long)d->B * d->Hi, &gx, &gy, 4096);
    if (d->stride == 1 && (d->k == 3 || d->k == 5) && d->dtype == DYK_BF16 && dw_tile_on()) {
        const int rt = d->k == 3 ? launch_dw_tile<3, true>(d, (hipStream_t)stream) : launch_dw_tile<5, true>(d, (hipStream_t)stream);
        if (rt != DYK_ERR_UNSUPPORTED) { DYK_LAUNCH_CHECK(); return rt; }
    }
    if (d->stride == 1 && (d->k == 3 || d->k == 5) && d->dtype == DYK_BF16) {
        if (d->k == 3) hipLaunchKernelGGL((dwconv_strip_kernel<bf16_t, 3, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
        else hipLaunchKernelGGL((dwconv_strip_kernel<bf16_t, 5, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    } else if (d->stride == 2 && (d->k == 3 || d->k == 5) && d->dtype == DYK_BF16) {
        if (d->k == 3) hipLaunchKernelGGL((dwconv_s2_kernel<bf16_t, 3, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
        else hipLaunchKernelGGL((dwconv_s2_kernel<bf16_t, 5, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    } else if (d->dtype == DYK_BF16)
        hipLaunchKernelGGL((dwconv_kernel<bf16_t, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    else
        hipLaunchKernelGGL((dwconv_kernel<float, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *d, CVB);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

template <typename T>
int launch_dw_wgrad(const DykDwDesc* d, hipStream_t s, int gx, int gy, int CVB) {
    const dim3 grid(gx, gy, d->k);
    switch (d->k) {
    case 1: hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 1>), grid, dim3(256), 0, s, *d, CVB); break;
    case 3: hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 3>), grid, dim3(256), 0, s, *d, CVB); break;
    case 5: hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 5>), grid, dim3(256), 0, s, *d, CVB); break;
    case 7: hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 7>), grid, dim3(256), 0, s, *d, CVB); break;
    default: return DYK_ERR_UNSUPPORTED;
    }
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_dwconv_wgrad_rows(const DykDwDesc* d) {
    const int rc = check_dw(d);
    if (rc) return rc;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    DwWgTile c;
    if (dw_wgrad_tile_cfg(d, &c)) return c.P;
    int gx, gy;
    grid2d(d->C / epv, (long)d->B * d->Ho, &gx, &gy, 768 / d->k);
    return gy;
}

extern "C" int dyk_dwconv_wgrad(const DykDwDesc* d, void* stream) {
    const int rc = check_dw(d);
    if (rc) return rc;
    if (!d->dw) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    DwWgTile c;
    if (dw_wgrad_tile_cfg(d, &c)) return launch_dw_wgrad_tile(d, c, (hipStream_t)stream);
    int gx, gy;
    // few, long-running workgroups: every workgroup ends with one atomic per (tap, channel) on only k*k*C addresses
    const int CVB = grid2d(d->C / epv, (long)d->B * d->Ho, &gx, &gy, 768 / d->k);
    if (d->dtype == DYK_BF16) return launch_dw_wgrad<bf16_t>(d, (hipStream_t)stream, gx, gy, CVB);
    return launch_dw_wgrad<float>(d, (hipStream_t)stream, gx, gy, CVB);
}
