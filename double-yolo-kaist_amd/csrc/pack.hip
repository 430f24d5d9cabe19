// Weight packing and boundary layout conversion kernels (HBM-bound, elementwise).
#include "dyk_common.h"

namespace {

// out[t][r][c] (r < R_pad rows, c < C_pad cols); forward: r = co, c = ci ; transposed: r = ci, c = co
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin,
                                   int taps, int R_pad, int C_pad, int transposed) {
    const long total = (long)taps * R_pad * C_pad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C_pad);
        const long q = i / C_pad;
        const int r = (int)(q % R_pad);
        const int t = (int)(q / R_pad);
        const int co = transposed ? c : r;
        const int ci = transposed ? r : c;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((long)co * Cin + ci) * taps + t];
        out[i] = ElemTraits<T>::from_f32(v);
    }
}

// one block handles a [32 pixels][32 channels] transpose through LDS so that both the
// NCHW side (pixel-contiguous) and the NHWC side (channel-contiguous) are coalesced.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int C, int HW,
                                    int Cpad, int ldo, float mul) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        float v = 0.f;
        if (c < C && p < HW) v = in[((long)b * C + c) * HW + p] * mul;
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < HW && c < Cpad) out[((long)b * HW + p) * ldo + c] = ElemTraits<T>::from_f32(tile[tx][j]);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int C, int HW, int ldi) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        float v = 0.f;
        if (p < HW && c < C) v = ElemTraits<T>::to_f32(in[((long)b * HW + p) * ldi + c]);
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        if (c < C && p < HW) out[((long)b * C + c) * HW + p] = tile[tx][j];
    }
}

}  // namespace

extern "C" int dyk_pack_conv_weight(const float* w, void* out, int32_t Cout, int32_t Cin, int32_t kh, int32_t kw,
                                    int32_t Cout_pad, int32_t Cin_pad, int32_t transposed, int32_t dtype,
                                    void* stream) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || Cout_pad < Cout || Cin_pad < Cin)
        return DYK_ERR_ARG;
    const int taps = kh * kw;
    const int R = transposed ? Cin_pad : Cout_pad, C = transposed ? Cout_pad : Cin_pad;
    const long total = (long)taps * R * C;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DYK_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w, (bf16_t*)out, Cout, Cin, taps, R, C, transposed);
    else if (dtype == DYK_F32)
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)out, Cout, Cin, taps, R, C, transposed);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_nchw_to_nhwc(const float* in, void* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cpad,
                                int32_t ldo, float mul, int32_t dtype, void* stream) {
    if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C || ldo < Cpad) return DYK_ERR_ARG;
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (Cpad + 31) / 32, B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DYK_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, s, in, (bf16_t*)out, C, HW, Cpad, ldo, mul);
    else if (dtype == DYK_F32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, s, in, (float*)out, C, HW, Cpad, ldo, mul);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_nhwc_to_nchw(const void* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t ldi,
                                int32_t dtype, void* stream) {
    if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || ldi < C) return DYK_ERR_ARG;
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DYK_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, out, C, HW, ldi);
    else if (dtype == DYK_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, s, (const float*)in, out, C, HW, ldi);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
