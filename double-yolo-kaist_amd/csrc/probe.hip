// Hardware probes (test infrastructure, not part of the product library):
//  - dyk_probe_tr16: dumps what ds_read_b64_tr_b16 returns for a given per-lane LDS address
//    pattern, to pin the transpose-read semantics the weight-gradient kernel relies on.
//  - dyk_probe_mfma_layout: D = A*B for the 16x16x32 bf16 MFMA with one-hot operands.
#include "dyk_common.h"

typedef short v4i16 __attribute__((__vector_size__(4 * sizeof(short))));

__global__ void tr16_kernel(const int* lane_byte_addr, unsigned short* out, int n_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int addr = lane_byte_addr[threadIdx.x];
        v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4i16*)((__attribute__((address_space(3))) char*)lds + addr));
        for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
    }
}

extern "C" int dyk_probe_tr16(const int* lane_byte_addr, unsigned short* out, void* stream) {
    hipLaunchKernelGGL(tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lane_byte_addr, out, 4096);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// A[m][k] (16x32) and B[k][n] (32x16) given as float matrices in global memory; each lane
// gathers its fragment under the ASSUMED layout a[j] = A[lane&15][(lane>>4)*8+j],
// b[j] = B[(lane>>4)*8+j][lane&15]; D written as D[(lane>>4)*4+r][lane&15] = acc[r].
__global__ void mfma_layout_kernel(const float* A, const float* B, float* D) {
    const int lane = threadIdx.x;
    float av[8], bv[8];
    for (int j = 0; j < 8; ++j) {
        av[j] = A[(lane & 15) * 32 + (lane >> 4) * 8 + j];
        bv[j] = B[((lane >> 4) * 8 + j) * 16 + (lane & 15)];
    }
    uint4 a = vec_pack<bf16_t>(av), b = vec_pack<bf16_t>(bv);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
}

extern "C" int dyk_probe_mfma_layout(const float* A, const float* B, float* D, void* stream) {
    hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
