// Fused optimizer steps over the flat parameter / gradient buffers (one launch for all 116 M
// parameters): Adam and SGD-Nesterov with L2 weight decay exactly as torch.optim applies them
// (reference train.py:85-91 builds optim.Adam / optim.SGD over one parameter group), optional
// gradient scaling (1/world_size, loss-scale removal), optional emission of the compute-dtype weight
// copy and optional zeroing of the gradient for the next accumulation.  HBM-bound: 16-byte vectors.
#include "dyk_common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(DykOptimDesc d, float bc1, float bc2_sqrt) {
    const long n4 = d.n >> 2;
    const float lr = d.lr, b1 = d.beta1, b2 = d.beta2, eps = d.eps, wd = d.weight_decay, gs = d.grad_scale;
    const float step_size = lr / bc1;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        if (d.mask && ((const uint32_t*)d.mask)[i] == 0) {      // frozen parameters (entries are 64-element aligned)
            if (d.zero_grad) ((float4*)d.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        float4 p = ((float4*)d.p)[i];
        float4 g = ((const float4*)d.g)[i];
        float4 m = ((float4*)d.m)[i];
        float4 v = ((float4*)d.v)[i];
        float* pp = (float*)&p; float* gg = (float*)&g; float* mm = (float*)&m; float* vv = (float*)&v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = gg[j] * gs;
            if (wd != 0.f) gr += wd * pp[j];
            mm[j] = b1 * mm[j] + (1.f - b1) * gr;                 // exp_avg.lerp_(grad, 1 - beta1)
            vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;            // exp_avg_sq
            const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
            pp[j] -= step_size * (mm[j] / denom);
        }
        ((float4*)d.p)[i] = p;
        ((float4*)d.m)[i] = m;
        ((float4*)d.v)[i] = v;
        if (d.zero_grad) ((float4*)d.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.wc) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(pp[0]) | ((uint32_t)f32_to_bf16(pp[1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(pp[2]) | ((uint32_t)f32_to_bf16(pp[3]) << 16);
            ((uint2*)d.wc)[i] = pk;
        }
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(DykOptimDesc d, int first_step) {
    const long n4 = d.n >> 2;
    const float lr = d.lr, mom = d.beta1, wd = d.weight_decay, gs = d.grad_scale;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        if (d.mask && ((const uint32_t*)d.mask)[i] == 0) {
            if (d.zero_grad) ((float4*)d.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        float4 p = ((float4*)d.p)[i];
        float4 g = ((const float4*)d.g)[i];
        float4 m = ((float4*)d.m)[i];
        float* pp = (float*)&p; float* gg = (float*)&g; float* mm = (float*)&m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = gg[j] * gs;
            if (wd != 0.f) gr += wd * pp[j];
            mm[j] = first_step ? gr : mom * mm[j] + gr;           // momentum buffer (dampening 0)
            gr = gr + mom * mm[j];                                // nesterov
            pp[j] -= lr * gr;
        }
        ((float4*)d.p)[i] = p;
        ((float4*)d.m)[i] = m;
        if (d.zero_grad) ((float4*)d.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.wc) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(pp[0]) | ((uint32_t)f32_to_bf16(pp[1]) << 16);
            pk.y = (uint32_t)f32_to_bf16(pp[2]) | ((uint32_t)f32_to_bf16(pp[3]) << 16);
            ((uint2*)d.wc)[i] = pk;
        }
    }
}

int check(const DykOptimDesc* d, bool need_v) {
    if (!d || !d->p || !d->g || !d->m || (need_v && !d->v) || d->n <= 0 || (d->n & 3)) return DYK_ERR_ARG;
    if (((uintptr_t)d->p | (uintptr_t)d->g | (uintptr_t)d->m | (uintptr_t)d->v) & 15) return DYK_ERR_ARG;
    if (d->wc && ((uintptr_t)d->wc & 7)) return DYK_ERR_ARG;
    if (d->mask && ((uintptr_t)d->mask & 3)) return DYK_ERR_ARG;
    return DYK_OK;
}

}  // namespace

extern "C" int dyk_adam_step(const DykOptimDesc* d, void* stream) {
    const int rc = check(d, true);
    if (rc) return rc;
    if (d->step <= 0) return DYK_ERR_ARG;
    const double bc1 = 1.0 - pow((double)d->beta1, (double)d->step);
    const double bc2 = 1.0 - pow((double)d->beta2, (double)d->step);
    long g = ((d->n >> 2) + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, *d, (float)bc1, (float)sqrt(bc2));
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_sgd_step(const DykOptimDesc* d, void* stream) {
    const int rc = check(d, false);
    if (rc) return rc;
    if (d->step <= 0) return DYK_ERR_ARG;
    long g = ((d->n >> 2) + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(sgd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, *d, d->step == 1 ? 1 : 0);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
