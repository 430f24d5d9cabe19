// Shared device helpers for the dyk HIP kernels (gfx950 / CDNA4 only).
//
// Data layout everywhere below the C ABI: activations are channels-last
// ("NHWC"), element type T in {bf16 (raw uint16), f32}; a tensor is addressed
// as base + pixel * ld + channel, with ld >= C so that a tensor can be a
// channel slice of a wider (concat) buffer.  16-byte vectors (8 bf16 / 4 f32)
// are the unit of every global and LDS access.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/dyk_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define DYK_WAVE 64

#define DYK_HIP_TRY(expr)                                   \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) return DYK_ERR_HIP;           \
    } while (0)

#define DYK_LAUNCH_CHECK()                                  \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return DYK_ERR_HIP; \
    } while (0)

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __host__ inline float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}
// round-to-nearest-even, NaN preserved (same rule as torch.bfloat16 casts); on the device this is the
// gfx950 hardware conversion v_cvt_pk_bf16_f32 (no integer-rounding sequence, no NaN branch)
typedef __attribute__((ext_vector_type(2))) __bf16 dyk_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float dyk_f32x2_t;
__device__ inline uint32_t f32x2_to_bf16x2(float lo, float hi) {
    const dyk_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dyk_bf16x2_t));
}
__device__ __host__ inline bf16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (bf16_t)(f32x2_to_bf16x2(f, 0.f) & 0xffffu);
#endif
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<bf16_t> {
    static constexpr int EPV = 8;  // elements per 16-byte vector
    static __device__ inline float to_f32(bf16_t v) { return bf16_to_f32(v); }
    static __device__ inline bf16_t from_f32(float f) { return f32_to_bf16(f); }
};
template <> struct ElemTraits<float> {
    static constexpr int EPV = 4;
    static __device__ inline float to_f32(float v) { return v; }
    static __device__ inline float from_f32(float f) { return f; }
};

// 16-byte vector <-> EPV floats
// 16-byte load of a READ-ONCE stream (the raw conv outputs and gradients the BatchNorm passes walk): nontemporal, so that the
// lines do not displace what other kernels re-read through the L2 (weights, activation tiles under nine taps).  Round 3,
// in-call A/B of the whole C3 step, six interleaved runs each: plain 32.21 ms, BatchNorm passes nontemporal 31.93 ms; the same
// hint on the optimizer state, the partial planes, axpby and the raw-output loads of the fused BatchNorm-backward epilogue
// 32.03 ms (no further gain: not adopted); nontemporal STORES changed nothing.  -DDYK_NO_NT builds the plain form.
typedef unsigned dyk_v4u_t __attribute__((ext_vector_type(4)));
__device__ inline uint4 ld_stream16(const void* p) {
#ifdef DYK_NO_NT
    return *(const uint4*)p;
#else
    return __builtin_bit_cast(uint4, __builtin_nontemporal_load((const dyk_v4u_t*)p));
#endif
}
__device__ inline float4 ld_stream_f4(const void* p) { return __builtin_bit_cast(float4, ld_stream16(p)); }

template <typename T> __device__ inline void vec_unpack(const uint4& v, float* out);
template <> __device__ inline void vec_unpack<bf16_t>(const uint4& v, float* out) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i]     = __uint_as_float(w[i] << 16);
        out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <> __device__ inline void vec_unpack<float>(const uint4& v, float* out) {
    out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y);
    out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
}
template <typename T> __device__ inline uint4 vec_pack(const float* in);
template <> __device__ inline uint4 vec_pack<bf16_t>(const float* in) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = f32x2_to_bf16x2(in[2 * i], in[2 * i + 1]);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
template <> __device__ inline uint4 vec_pack<float>(const float* in) {
    return make_uint4(__float_as_uint(in[0]), __float_as_uint(in[1]),
                      __float_as_uint(in[2]), __float_as_uint(in[3]));
}

// ---------------------------------------------------------------- activations
// Forward value and derivative w.r.t. the pre-activation, matching the
// torch.nn modules the reference instantiates at models.py:51-62
// (Mish, ReLU, LeakyReLU(0.1), ReLU6, Hardsigmoid, Hardswish).
__device__ inline float act_fwd(int act, float x) {
    switch (act) {
    case DYK_ACT_LEAKY: return x > 0.f ? x : 0.1f * x;
    case DYK_ACT_MISH: {
        // x * tanh(softplus(x)) with tanh(log(1+e^x)) = n / (n + 2), n = e^x (e^x + 2).  torch switches softplus to the
        // identity above 20; clamping the exponent there gives n/(n+2) == 1.0f exactly, i.e. the same value, branch-free
        const float e = __expf(fminf(x, 20.f));
        const float n = e * (e + 2.f);
        return x * (n * __builtin_amdgcn_rcpf(n + 2.f));      // v_rcp_f32: 1 ulp, no IEEE-division expansion
    }
    case DYK_ACT_RELU: return x > 0.f ? x : 0.f;
    case DYK_ACT_RELU6: return fminf(fmaxf(x, 0.f), 6.f);
    case DYK_ACT_HSIGMOID: return fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    case DYK_ACT_HSWISH: return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    default: return x;
    }
}
// compile-time activation (hoists the switch out of unrolled epilogues)
__device__ inline float act_bwd(int act, float x);
template <int ACT> __device__ inline float act_fwd_c(float x, int runtime_act = 0) {
    if constexpr (ACT == DYK_ACT_LINEAR) return x;
    else if constexpr (ACT < 0) return act_fwd(runtime_act, x);      // ACT = -1: runtime switch
    else return act_fwd(ACT, x);
}
template <int ACT> __device__ inline float act_bwd_c(float x, int runtime_act = 0) {
    if constexpr (ACT < 0) return act_bwd(runtime_act, x);
    else return act_bwd(ACT, x);
}
__device__ inline float act_bwd(int act, float x) {
    switch (act) {
    case DYK_ACT_LEAKY: return x > 0.f ? 1.f : 0.1f;
    case DYK_ACT_MISH: {
#ifdef DYK_MISH_OLD
        const float e = __expf(fminf(x, 20.f));               // x > 20: t == 1, derivative == 1 (see act_fwd)
        const float n = e * (e + 2.f);
        const float t = n * __builtin_amdgcn_rcpf(n + 2.f);   // tanh(softplus(x))
        const float sg = e * __builtin_amdgcn_rcpf(1.f + e);  // sigmoid(x)
        return t + x * (1.f - t * t) * sg;
#else
        // mish(x) = x n / (n + 2), n = e (e + 2), e = exp(x); n' = 2 e (e + 1), so
        //   mish'(x) = [n (n + 2) + 4 x e (e + 1)] / (n + 2)^2
        // ONE exponential and ONE reciprocal (the tanh / sigmoid form above takes two reciprocals; the transcendental unit
        // runs at a quarter of the VALU rate and the step evaluates this ~8 G times: every fused BatchNorm-backward
        // epilogue and apply pass).  Exponent clamped at 20 as in act_fwd: (n + 2)^2 <= e^80 stays inside fp32, the value is
        // 1 + O(x e^-40) there.
        const float e = __expf(fminf(x, 20.f));
        const float n = e * (e + 2.f);
        const float d = n + 2.f;
        const float num = n * d + (4.f * x) * (e * e + e);
        return num * __builtin_amdgcn_rcpf(d * d);
#endif
    }
    case DYK_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case DYK_ACT_RELU6: return (x > 0.f && x < 6.f) ? 1.f : 0.f;
    case DYK_ACT_HSIGMOID: return (x > -3.f && x < 3.f) ? (1.f / 6.f) : 0.f;
    case DYK_ACT_HSWISH: return x < -3.f ? 0.f : (x <= 3.f ? (x * (1.f / 3.f) + 0.5f) : 1.f);
    default: return 1.f;
    }
}

// Two values at once (packed fp32 VALU: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process two floats per lane and
// instruction).  Mish: the same formula as act_bwd above, the exponential and the reciprocal per element (the transcendental
// unit has no packed form); other activations fall back to the scalar function.
template <int ACT> __device__ inline dyk_f32x2_t act_bwd2_c(dyk_f32x2_t x, int runtime_act = 0) {
    if constexpr (ACT == DYK_ACT_MISH) {
        const dyk_f32x2_t xc = {fminf(x[0], 20.f), fminf(x[1], 20.f)};
        const dyk_f32x2_t e = {__expf(xc[0]), __expf(xc[1])};
        const dyk_f32x2_t two = {2.f, 2.f}, four = {4.f, 4.f};
        const dyk_f32x2_t n = e * (e + two);
        const dyk_f32x2_t d = n + two;
        const dyk_f32x2_t num = n * d + (four * x) * (e * e + e);
        const dyk_f32x2_t dd = d * d;
        const dyk_f32x2_t r = {__builtin_amdgcn_rcpf(dd[0]), __builtin_amdgcn_rcpf(dd[1])};
        return num * r;
    } else {
        return (dyk_f32x2_t){act_bwd_c<ACT>(x[0], runtime_act), act_bwd_c<ACT>(x[1], runtime_act)};
    }
}

// ---------------------------------------------------------------- wave helpers
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the 16 lanes of a DPP row (lanes 16k..16k+15); every lane of the row receives the total.
// Four v_add_f32 with row_ror DPP modifiers -- no LDS crossbar traffic (ds_bpermute) as __shfl_xor would emit.
__device__ inline float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));
    return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Bijective XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8):
// consecutive remapped ids land on the same XCD so neighbouring tiles share L2.
__device__ inline int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

static inline int dyk_div_up(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE attribute and exec.hip serves several devices in one
// process: a function-local `static DykDeviceOnce` answers "first launch of this instantiation on the CURRENT device?"
// (a per-process bool left the second device at the default 64 KB cap)
struct DykDeviceOnce {
    unsigned long long seen[4] = {0, 0, 0, 0};           // 256 device ids
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return true;
        // (atomic: two host threads launching on different devices share the word)
        const unsigned long long bit = 1ull << (dev & 63);
        return (__atomic_fetch_or(&seen[(dev >> 6) & 3], bit, __ATOMIC_ACQ_REL) & bit) == 0;
    }
};

// Two-problem launches of the elementwise / BatchNorm kernels (DykEwDesc.twin, DykBnFinalizeDesc.twin): the kernel takes
// both descriptors and blockIdx.z selects the problem.  fill_* return the number of problems (1 | 2), 0 when the twin
// differs in a non-pointer field.
// Kernels COPY their descriptor out of the pair (`const DykEwDesc d = pr.d[blockIdx.z];`): through a reference into the
// kernel arguments every field was re-loaded (s_load + s_waitcnt lgkmcnt(0)) in front of each load and store of the pixel
// loop -- the BatchNorm passes had become 5 % slower when the two-problem form was introduced (ISA, round 3).
struct DykEwPair { DykEwDesc d[2]; };
struct DykFinPair { DykBnFinalizeDesc d[2]; };
static inline int dyk_fill_ew_pair(DykEwPair& p, const DykEwDesc* d) {
    p.d[0] = *d;
    p.d[0].twin = nullptr;
    if (!d->twin) return 1;
    const size_t lo = offsetof(DykEwDesc, dtype), hi = offsetof(DykEwDesc, twin);
    const DykEwDesc* t = d->twin;
    if (__builtin_memcmp((const char*)d + lo, (const char*)t + lo, hi - lo) != 0) return 0;
    // the same optional operands on both sides
    if ((!d->b) != (!t->b) || (!d->out) != (!t->out) || (!d->p0) != (!t->p0) || (!d->p1) != (!t->p1) || (!d->p2) != (!t->p2) ||
        (!d->p3) != (!t->p3) || (!d->red) != (!t->red) || (!d->aux) != (!t->aux) || (!d->aux2) != (!t->aux2) || !t->a)
        return 0;
    p.d[1] = *t;
    p.d[1].twin = nullptr;
    return 2;
}
static inline int dyk_fill_fin_pair(DykFinPair& p, const DykBnFinalizeDesc* d) {
    p.d[0] = *d;
    p.d[0].twin = nullptr;
    if (!d->twin) return 1;
    const size_t lo = offsetof(DykBnFinalizeDesc, C), hi = offsetof(DykBnFinalizeDesc, twin);
    const DykBnFinalizeDesc* t = d->twin;
    if (__builtin_memcmp((const char*)d + lo, (const char*)t + lo, hi - lo) != 0) return 0;
    if (!t->stats || !t->scale || !t->shift || (!d->gamma) != (!t->gamma) || (!d->beta) != (!t->beta) ||
        (!d->running_mean) != (!t->running_mean) || (!d->save_mean) != (!t->save_mean) || (!d->save_rstd) != (!t->save_rstd))
        return 0;
    p.d[1] = *t;
    p.d[1].twin = nullptr;
    return 2;
}
