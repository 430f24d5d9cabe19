// BatchNorm2d (train-mode batch statistics) + activation, forward and backward.  All HBM-bound
// channels-last kernels, 16-byte vectors, per-channel parameters in fp32, cross-block
// reductions in fp64 atomics.
//
// Forward of nn.Sequential(Conv2d, BatchNorm2d, act) (reference models.py:34-62) in training:
//   conv kernel (raw output + per-channel sum / sum^2)  ->  dyk_bn_finalize  ->  dyk_bn_act_fwd
// Backward:
//   dyk_bn_act_bwd_reduce (sum dact, sum dact*xhat) -> dyk_bn_bwd_params (dgamma, dbeta)
//   -> dyk_bn_act_bwd_apply (gradient w.r.t. the raw conv output) -> conv dgrad / wgrad.
#include <stdlib.h>
#include "dyk_common.h"

namespace {

// 32 lanes per channel: lane r sums replicas r, r+32, ... , a 5-step xor-shuffle folds them, lane 0 finalises
__global__ __launch_bounds__(256) void bn_finalize_kernel(DykFinPair pr) {
    const DykBnFinalizeDesc d = pr.d[blockIdx.z];
    const int sub = threadIdx.x & 31;
    const int c = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (c >= d.C) return;
    const double n = (double)d.count;
    const int slots = d.slots > 0 ? d.slots : 1;
    double s1 = 0.0, s2 = 0.0;
    for (int r = sub; r < slots; r += 32) {
        double* st = d.stats + (size_t)r * 2 * d.C;
        s1 += st[c]; s2 += st[d.C + c];
        st[c] = 0.0; st[d.C + c] = 0.0;          // re-arm the accumulators for the next step
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (sub != 0) return;
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)d.eps));
    const float g = d.gamma ? d.gamma[c] : 1.f, b = d.beta ? d.beta[c] : 0.f;
    const float sc = g * rstd;
    d.scale[c] = sc;
    d.shift[c] = b - (float)mean * sc;
    if (d.save_mean) d.save_mean[c] = (float)mean;
    if (d.save_rstd) d.save_rstd[c] = rstd;
    if (d.running_mean) {
        const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
        d.running_mean[c] = (1.f - d.momentum) * d.running_mean[c] + d.momentum * (float)mean;
        d.running_var[c] = (1.f - d.momentum) * d.running_var[c] + d.momentum * (float)unb;
    }
}

// eval-mode BN folded to scale/shift: scale = gamma / sqrt(running_var + eps)
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = (gamma ? gamma[c] : 1.f) / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

// Elementwise kernels use a 2-D thread mapping: tx = channel vector inside a group of CVB, ty = pixel
// lane; grid.x walks channel-vector groups, grid.y splits the pixels into one CONTIGUOUS run per workgroup (round 3: with the
// workgroups striding over the whole tensor -- pixel p0 + u * gridDim.y * PY -- the 168 MB passes ran 3-6 % slower: 58.0 -> 54.7 us
// normalise, 102.0 -> 98.7 us apply on 256 x 320 x 64; C3 step -0.15 ms over four interleaved runs).  No integer division in the
// loop, per-channel parameters live in registers, and a wave touches CVB*16 contiguous bytes per pixel
// row (whole rows for C <= 256 bf16), i.e. fully coalesced when ld == C.
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(DykEwPair pr, int CVB) {
    const DykEwDesc d = pr.d[blockIdx.z];          // a COPY, not a reference (see DykEwPair in dyk_common.h)
    constexpr int EPV = ElemTraits<T>::EPV;
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int cv = blockIdx.x * CVB + tx;
    if (cv * EPV >= d.C) return;
    const int c = cv * EPV;
    float sc[EPV], sh[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) { sc[j] = d.p0 ? d.p0[c + j] : 1.f; sh[j] = d.p1 ? d.p1[c + j] : 0.f; }
    const T* __restrict__ a = (const T*)d.a;
    const T* __restrict__ r = (const T*)d.b;
    T* __restrict__ o = (T*)d.out;
    constexpr int U = 4;                      // pixels in flight per thread (memory-level parallelism)
    const long chunk = ((d.npix + gridDim.y - 1) / gridDim.y + PY - 1) / PY * PY;
    const long pend = min((long)d.npix, ((long)blockIdx.y + 1) * chunk);
    const long pstep = PY;
    for (long p0 = (long)blockIdx.y * chunk + ty; p0 < pend; p0 += pstep * U) {
        uint4 vx[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u * pstep;
            if (p < pend) {
                vx[u] = ld_stream16(a + p * d.lda + c);
                if (r) vr[u] = *(const uint4*)(r + p * d.ldb + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u * pstep;
            if (p >= pend) break;
            float x[EPV], y[EPV];
            vec_unpack<T>(vx[u], x);
#pragma unroll
            for (int j = 0; j < EPV; ++j) y[j] = act_fwd_c<ACT>(x[j] * sc[j] + sh[j]);
            if (r) {
                float rr[EPV];
                vec_unpack<T>(vr[u], rr);
#pragma unroll
                for (int j = 0; j < EPV; ++j) y[j] += rr[j];
            }
            *(uint4*)(o + p * d.ldo + c) = vec_pack<T>(y);
        }
    }
}

// bn_finalize + bn_act_fwd in one launch (the 7 us finalize launch sat on the critical chain of the forward pass):
// every workgroup folds the statistics replicas of its own <= 256 channels (8 independent fp64 loads in flight per
// thread), derives scale / shift into LDS, and the workgroups of grid row 0 also publish scale / shift / mean / rstd
// for the backward pass and update the running statistics.  The replicas are NOT re-armed here (other workgroups
// may still be reading them): the caller zeroes the statistics arena once per forward pass.
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_fused_fwd_kernel(DykFinPair fp, DykEwPair pr, int CVB) {
    const DykBnFinalizeDesc f = fp.d[blockIdx.z];
    const DykEwDesc d = pr.d[blockIdx.z];          // a COPY, not a reference (see DykEwPair in dyk_common.h)
    constexpr int EPV = ElemTraits<T>::EPV;
    __shared__ float s_aff[2][256];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int cv = blockIdx.x * CVB + tx;
    {
        const int c_base = blockIdx.x * CVB * EPV;
        const int nch = min(CVB * EPV, d.C - c_base);
        const int slots = f.slots > 0 ? f.slots : 1;
        const size_t rs = (size_t)2 * f.C;
        for (int cl = threadIdx.x; cl < nch; cl += 256) {
            const int c = c_base + cl;
            // FB replicas x 2 sums requested back to back: the fold is a chain of slots / FB dependent L2 round trips in front of
            // the first pixel load of a launch-bound pass (round 3, tools/gpu_probe.py bnfold: 32 replicas in batches of 4 cost
            // 3.5 us of a 10 us launch on the 32 x 40 x 256 tensors, 1 replica 0 us)
            constexpr int FB = 16;
            double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
            int r = 0;
            for (; r + FB <= slots; r += FB) {
                double v1[FB], v2[FB];
#pragma unroll
                for (int u = 0; u < FB; ++u) { v1[u] = f.stats[(size_t)(r + u) * rs + c]; v2[u] = f.stats[(size_t)(r + u) * rs + f.C + c]; }
#pragma unroll
                for (int u = 0; u < FB; ++u) { a1[u & 3] += v1[u]; a2[u & 3] += v2[u]; }
            }
            for (; r + 4 <= slots; r += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { a1[u] += f.stats[(size_t)(r + u) * rs + c]; a2[u] += f.stats[(size_t)(r + u) * rs + f.C + c]; }
            }
            for (; r < slots; ++r) { a1[0] += f.stats[(size_t)r * rs + c]; a2[0] += f.stats[(size_t)r * rs + f.C + c]; }
            const double n = (double)f.count;
            const double mean = ((a1[0] + a1[1]) + (a1[2] + a1[3])) / n;
            double var = ((a2[0] + a2[1]) + (a2[2] + a2[3])) / n - mean * mean;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
            const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
            const float sc = g * rstd, sh = b - (float)mean * sc;
            s_aff[0][cl] = sc;
            s_aff[1][cl] = sh;
            if (blockIdx.y == 0) {
                f.scale[c] = sc;
                f.shift[c] = sh;
                if (f.save_mean) f.save_mean[c] = (float)mean;
                if (f.save_rstd) f.save_rstd[c] = rstd;
                if (f.running_mean) {
                    const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                    f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
                    f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
                }
            }
        }
        __syncthreads();
    }
    if (cv * EPV >= d.C) return;
    const int c = cv * EPV;
    float sc[EPV], sh[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) { sc[j] = s_aff[0][tx * EPV + j]; sh[j] = s_aff[1][tx * EPV + j]; }
    const T* __restrict__ a = (const T*)d.a;
    const T* __restrict__ r = (const T*)d.b;
    T* __restrict__ o = (T*)d.out;
    constexpr int U = 4;
    const long chunk = ((d.npix + gridDim.y - 1) / gridDim.y + PY - 1) / PY * PY;
    const long pend = min((long)d.npix, ((long)blockIdx.y + 1) * chunk);
    const long pstep = PY;
    for (long p0 = (long)blockIdx.y * chunk + ty; p0 < pend; p0 += pstep * U) {
        uint4 vx[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u * pstep;
            if (p < pend) {
                vx[u] = ld_stream16(a + p * d.lda + c);
                if (r) vr[u] = *(const uint4*)(r + p * d.ldb + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u * pstep;
            if (p >= pend) break;
            float x[EPV], y[EPV];
            vec_unpack<T>(vx[u], x);
#pragma unroll
            for (int j = 0; j < EPV; ++j) y[j] = act_fwd_c<ACT>(x[j] * sc[j] + sh[j]);
            if (r) {
                float rr[EPV];
                vec_unpack<T>(vr[u], rr);
#pragma unroll
                for (int j = 0; j < EPV; ++j) y[j] += rr[j];
            }
            *(uint4*)(o + p * d.ldo + c) = vec_pack<T>(y);
        }
    }
}

// block = (CVB channel vectors) x (PY pixel lanes); grid.x over channel-vector groups, grid.y over pixels
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(DykEwPair pr, int CVB) {
    const DykEwDesc d = pr.d[blockIdx.z];          // a COPY, not a reference (see DykEwPair in dyk_common.h)
    constexpr int EPV = ElemTraits<T>::EPV;
    __shared__ float red[256 * 2 * 8];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int CV = d.C / EPV;
    const int cv = blockIdx.x * CVB + tx;
    const int c = cv * EPV;
    const bool active = cv < CV;
    float s1[EPV], s2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s1[j] = s2[j] = 0.f;
    if (active) {
        float sc[EPV], sh[EPV], mu[EPV], rs[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            sc[j] = d.p0[c + j]; sh[j] = d.p1[c + j]; mu[j] = d.p2[c + j]; rs[j] = d.p3[c + j];
        }
        const T* __restrict__ dz = (const T*)d.a;
        const T* __restrict__ y = (const T*)d.b;
        constexpr int U = 4;
        const long chunk = ((d.npix + gridDim.y - 1) / gridDim.y + PY - 1) / PY * PY;
            const long pend = min((long)d.npix, ((long)blockIdx.y + 1) * chunk);
            const long pstep = PY;
            for (long p0 = (long)blockIdx.y * chunk + ty; p0 < pend; p0 += pstep * U) {
            uint4 vg[U], vy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long p = p0 + u * pstep;
                if (p < pend) { vg[u] = ld_stream16(dz + p * d.lda + c); vy[u] = ld_stream16(y + p * d.ldb + c); }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (p0 + u * pstep >= pend) break;
                float g[EPV], yy[EPV];
                vec_unpack<T>(vg[u], g);
                vec_unpack<T>(vy[u], yy);
#pragma unroll
                for (int j = 0; j < EPV; ++j) {
                    const float da = g[j] * act_bwd_c<ACT>(yy[j] * sc[j] + sh[j]);
                    s1[j] += da;
                    s2[j] += da * ((yy[j] - mu[j]) * rs[j]);
                }
            }
        }
    }
    // reduce over ty through LDS
    float* mine = red + threadIdx.x * 2 * 8;
#pragma unroll
    for (int j = 0; j < EPV; ++j) { mine[j] = s1[j]; mine[8 + j] = s2[j]; }
    __syncthreads();
    if (ty == 0 && active) {
        for (int q = 1; q < PY; ++q) {
            const float* o = red + (q * CVB + tx) * 2 * 8;
#pragma unroll
            for (int j = 0; j < EPV; ++j) { s1[j] += o[j]; s2[j] += o[8 + j]; }
        }
        double* rd = d.red + (size_t)(blockIdx.y % (unsigned)(d.slots > 0 ? d.slots : 1)) * 2 * d.C;
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            atomicAdd(rd + c + j, (double)s1[j]);
            atomicAdd(rd + d.C + c + j, (double)s2[j]);
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_params_kernel(double* red, float* dgamma, float* dbeta, int C, int slots) {
    const int sub = threadIdx.x & 31;
    const int c = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int r = sub; r < slots; r += 32) { s1 += red[(size_t)r * 2 * C + c]; s2 += red[(size_t)r * 2 * C + C + c]; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (sub != 0) return;
    red[c] = s1; red[C + c] = s2;            // replica 0 now holds the totals (read by the apply kernel)
    if (dbeta) dbeta[c] += (float)s1;
    if (dgamma) dgamma[c] += (float)s2;
}

// The totals of the reduce pass are folded here (every workgroup sums the `slots` replicas of its own <= 256
// channels through LDS -- 2 x 32 coalesced fp64 loads per thread), and the workgroups of grid row 0 add them to
// dgamma / dbeta (aux / aux2): no separate parameter-gradient launch on the critical chain of the backward pass.
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(DykEwPair pr, int CVB) {
    const DykEwDesc d = pr.d[blockIdx.z];          // a COPY, not a reference (see DykEwPair in dyk_common.h)
    constexpr int EPV = ElemTraits<T>::EPV;
    __shared__ float s_tot[2][256];
    const int PY = 256 / CVB;
    const int tx = threadIdx.x % CVB, ty = threadIdx.x / CVB;
    const int cv = blockIdx.x * CVB + tx;
    {
        const int c_base = blockIdx.x * CVB * EPV;
        const int nch = min(CVB * EPV, d.C - c_base);
        const int slots = d.slots > 0 ? d.slots : 1;
        // replicas [slots][2][C] of this layer's own reduction -- or columns of a wider one (the joint reduction over the
        // sections a [route] concatenates, DYK_EPI_BNBWD of the route's reader): H = replica stride, W = offset of the second sum
        const size_t rs = d.H > 0 ? (size_t)d.H : (size_t)2 * d.C;
        const size_t so = d.W > 0 ? (size_t)d.W : (size_t)d.C;
        for (int cl = threadIdx.x; cl < nch; cl += 256) {
            const double* p = d.red + c_base + cl;
            // both sums of a channel, 16 replicas each, requested back to back: ONE L2 round trip in front of the pixel loop for the
            // backward's 16 replicas (a dependent chain of `slots` round trips cost ~10 us here; batches of 8 per sum, the two
            // sums of a 256-channel group one after the other, still cost four)
            double a8[2][8] = {{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
            int r = 0;
            for (; r + 16 <= slots; r += 16) {
                double v[2][16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { v[0][u] = p[(size_t)(r + u) * rs]; v[1][u] = p[(size_t)(r + u) * rs + so]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) { a8[0][u & 7] += v[0][u]; a8[1][u & 7] += v[1][u]; }
            }
            for (; r + 8 <= slots; r += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { a8[0][u] += p[(size_t)(r + u) * rs]; a8[1][u] += p[(size_t)(r + u) * rs + so]; }
            }
            for (; r < slots; ++r) { a8[0][0] += p[(size_t)r * rs]; a8[1][0] += p[(size_t)r * rs + so]; }
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const double acc = ((a8[which][0] + a8[which][1]) + (a8[which][2] + a8[which][3])) + ((a8[which][4] + a8[which][5]) + (a8[which][6] + a8[which][7]));
                s_tot[which][cl] = (float)acc;
                if (blockIdx.y == 0) {
                    float* g = which ? (float*)d.aux : (float*)d.aux2;      // sum(dact * xhat) -> dgamma, sum(dact) -> dbeta
                    if (g) g[c_base + cl] += (float)acc;
                }
            }
        }
        __syncthreads();
    }
    if (cv * EPV >= d.C) return;
    const int c = cv * EPV;
    const float invn = 1.f / (float)d.npix;
    float sc[EPV], sh[EPV], mu[EPV], rs[EPV], m1[EPV], m2[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        sc[j] = d.p0[c + j]; sh[j] = d.p1[c + j]; mu[j] = d.p2[c + j]; rs[j] = d.p3[c + j];
        m1[j] = s_tot[0][tx * EPV + j] * invn; m2[j] = s_tot[1][tx * EPV + j] * invn;
    }
    const T* __restrict__ dz = (const T*)d.a;
    const T* __restrict__ y = (const T*)d.b;
    T* __restrict__ o = (T*)d.out;
    const bool accum = d.flags & DYK_EW_ACCUM;
    constexpr int U = 4;
    const long chunk = ((d.npix + gridDim.y - 1) / gridDim.y + PY - 1) / PY * PY;
    const long pend = min((long)d.npix, ((long)blockIdx.y + 1) * chunk);
    const long pstep = PY;
    for (long p0 = (long)blockIdx.y * chunk + ty; p0 < pend; p0 += pstep * U) {
        uint4 vg[U], vy[U], vo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u * pstep;
            if (p < pend) {
                vg[u] = ld_stream16(dz + p * d.lda + c);
                vy[u] = ld_stream16(y + p * d.ldb + c);
                if (accum) vo[u] = *(const uint4*)(o + p * d.ldo + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u * pstep;
            if (p >= pend) break;
            float g[EPV], yy[EPV], r[EPV];
            vec_unpack<T>(vg[u], g);
            vec_unpack<T>(vy[u], yy);
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                const float da = g[j] * act_bwd_c<ACT>(yy[j] * sc[j] + sh[j]);
                const float xh = (yy[j] - mu[j]) * rs[j];
                r[j] = sc[j] * (da - m1[j] - xh * m2[j]);      // sc = gamma * rstd
            }
            if (accum) {
                float old[EPV];
                vec_unpack<T>(vo[u], old);
#pragma unroll
                for (int j = 0; j < EPV; ++j) r[j] += old[j];
            }
            *(uint4*)(o + p * d.ldo + c) = vec_pack<T>(r);
        }
    }
}

// grid for the 2-D elementwise mapping: returns CVB, fills gx / gy
inline int ew_grid2d(int CV, long npix, int* gx, int* gy) {
    int CVB = 1;
    while (CVB < CV && CVB < 32) CVB <<= 1;
    const int PY = 256 / CVB;
    *gx = (CV + CVB - 1) / CVB;
    constexpr int ppt = 8, capb = 2048;                         // (4 / 16 pixels per thread, grid caps 1 024 / 4 096: +-0.2 ms, round 2)
    long g = (npix + (long)PY * ppt - 1) / ((long)PY * ppt);    // >= 8 pixels per thread (2 unrolled iterations)
    const long cap = capb / *gx > 0 ? capb / *gx : 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    *gy = (int)g;
    return CVB;
}

// one instantiation per (element type, activation): the activation switch is resolved on the host, the kernels'
// inner loops are branch-free
#define DYK_BN_LAUNCH_ACT(KERNEL, T, ...)                                                                         \
    switch (d->act) {                                                                                              \
    case DYK_ACT_LINEAR: hipLaunchKernelGGL((KERNEL<T, DYK_ACT_LINEAR>), __VA_ARGS__); break;                      \
    case DYK_ACT_LEAKY: hipLaunchKernelGGL((KERNEL<T, DYK_ACT_LEAKY>), __VA_ARGS__); break;                        \
    case DYK_ACT_MISH: hipLaunchKernelGGL((KERNEL<T, DYK_ACT_MISH>), __VA_ARGS__); break;                          \
    case DYK_ACT_RELU: hipLaunchKernelGGL((KERNEL<T, DYK_ACT_RELU>), __VA_ARGS__); break;                          \
    case DYK_ACT_RELU6: hipLaunchKernelGGL((KERNEL<T, DYK_ACT_RELU6>), __VA_ARGS__); break;                        \
    case DYK_ACT_HSIGMOID: hipLaunchKernelGGL((KERNEL<T, DYK_ACT_HSIGMOID>), __VA_ARGS__); break;                  \
    case DYK_ACT_HSWISH: hipLaunchKernelGGL((KERNEL<T, DYK_ACT_HSWISH>), __VA_ARGS__); break;                      \
    default: return DYK_ERR_ARG;                                                                                   \
    }
#define DYK_BN_LAUNCH(KERNEL, ...)                                        \
    if (d->dtype == DYK_BF16) { DYK_BN_LAUNCH_ACT(KERNEL, bf16_t, __VA_ARGS__) } \
    else { DYK_BN_LAUNCH_ACT(KERNEL, float, __VA_ARGS__) }

inline int ew_check(const DykEwDesc* d, bool need_b) {
    if (!d || !d->a || !d->out || d->npix <= 0 || d->C <= 0) return DYK_ERR_ARG;
    if (need_b && !d->b) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    if (d->dtype != DYK_BF16 && d->dtype != DYK_F32) return DYK_ERR_ARG;
    if (d->C % epv || d->lda % epv || d->ldo % epv || (d->b && d->ldb % epv)) return DYK_ERR_ARG;
    return DYK_OK;
}

}  // namespace

extern "C" int dyk_bn_finalize(const DykBnFinalizeDesc* d, void* stream) {
    if (!d || !d->stats || !d->scale || !d->shift || d->C <= 0 || d->count <= 0) return DYK_ERR_ARG;
    DykFinPair fp;
    const int nz = dyk_fill_fin_pair(fp, d);
    if (!nz) return DYK_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((d->C + 7) / 8, 1, nz), dim3(256), 0, (hipStream_t)stream, fp);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, float* scale, float* shift, int32_t C,
                           void* stream) {
    if (!running_mean || !running_var || !scale || !shift || C <= 0) return DYK_ERR_ARG;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, scale, shift, C);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_bn_act_fwd(const DykEwDesc* d, void* stream) {
    const int rc = ew_check(d, false);
    if (rc) return rc;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    int gx, gy;
    const int CVB = ew_grid2d(d->C / epv, d->npix, &gx, &gy);
    DykEwPair pr;
    const int nz = dyk_fill_ew_pair(pr, d);
    if (!nz) return DYK_ERR_ARG;
    DYK_BN_LAUNCH(bn_act_fwd_kernel, dim3(gx, gy, nz), dim3(256), 0, (hipStream_t)stream, pr, CVB)
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_bn_finalize_act_fwd(const DykBnFinalizeDesc* f, const DykEwDesc* d, void* stream) {
    if (!f || !f->stats || !f->scale || !f->shift || f->C <= 0 || f->count <= 0) return DYK_ERR_ARG;
    const int rc = ew_check(d, false);
    if (rc) return rc;
    if (d->C != f->C) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    int gx, gy;
    const int CVB = ew_grid2d(d->C / epv, d->npix, &gx, &gy);
    DykEwPair pr;
    DykFinPair fp;
    const int nz = dyk_fill_ew_pair(pr, d);
    if (!nz || dyk_fill_fin_pair(fp, f) != nz) return DYK_ERR_ARG;       // both descriptors paired, or neither
    DYK_BN_LAUNCH(bn_fused_fwd_kernel, dim3(gx, gy, nz), dim3(256), 0, (hipStream_t)stream, fp, pr, CVB)
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_bn_act_bwd_reduce(const DykEwDesc* d, void* stream) {
    if (!d || !d->a || !d->b || !d->red || !d->p0 || !d->p1 || !d->p2 || !d->p3) return DYK_ERR_ARG;
    if (d->npix <= 0 || d->C <= 0 || (d->dtype != DYK_BF16 && d->dtype != DYK_F32)) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    if (d->C % epv || d->lda % epv || d->ldb % epv) return DYK_ERR_ARG;
    const int CV = d->C / epv;
    int CVB = 1;
    while (CVB < CV && CVB < 32) CVB <<= 1;
    const int PY = 256 / CVB;
    const int gx = (CV + CVB - 1) / CVB;
    long gy = ((long)d->npix + PY * 8 - 1) / (PY * 8);      // >= 8 pixels per thread
    const long cap = 2048 / gx > 0 ? 2048 / gx : 1;
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    DykEwPair pr;
    const int nz = dyk_fill_ew_pair(pr, d);
    if (!nz) return DYK_ERR_ARG;
    DYK_BN_LAUNCH(bn_act_bwd_reduce_kernel, dim3(gx, (int)gy, nz), dim3(256), 0, (hipStream_t)stream, pr, CVB)
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_bn_bwd_params(double* red, float* dgamma, float* dbeta, int32_t C, int32_t slots, void* stream) {
    if (!red || C <= 0) return DYK_ERR_ARG;
    hipLaunchKernelGGL(bn_bwd_params_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, red, dgamma, dbeta, C,
                       slots > 0 ? slots : 1);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_bn_act_bwd_apply(const DykEwDesc* d, void* stream) {
    const int rc = ew_check(d, true);
    if (rc) return rc;
    if (d->flags & DYK_EW_SKIP) return DYK_OK;          // done inside the consumer (DykStemDesc.bn_fused)
    if (!d->red || !d->p0 || !d->p1 || !d->p2 || !d->p3) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    int gx, gy;
    const int CVB = ew_grid2d(d->C / epv, d->npix, &gx, &gy);
    DykEwPair pr;
    const int nz = dyk_fill_ew_pair(pr, d);
    if (!nz) return DYK_ERR_ARG;
    DYK_BN_LAUNCH(bn_act_bwd_apply_kernel, dim3(gx, gy, nz), dim3(256), 0, (hipStream_t)stream, pr, CVB)
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
