// Box-coordinate helpers of the evaluation path on the device: xywh <-> xyxy, clip to the image, and the
// un-letterbox transform that maps detections from the network input back to the original frame.
//
// Replaces (reference build_utils/utils.py): xyxy2xywh :40-47, xywh2xyxy :50-57, scale_coords :60-81,
// clip_coords :84-92 -- as called by evaluate.py:82 / detect.py:114 on the NMS output [n,6].
// Plain fp32, no contraction: every value must round exactly like the torch-CPU expression of the reference
// (x - pad) / gain -> clamp, x -/+ w / 2, (x1 + x2) / 2.  Rows are `ld` floats apart (a [n,6] detection
// tensor is converted in place on its first four columns).
#pragma clang fp contract(off)
#include "dyk_common.h"

namespace {

__global__ void box_convert_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int ldi, int ldo, int to_xyxy) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* s = in + (long)i * ldi;
        float* d = out + (long)i * ldo;
        const float a = s[0], b = s[1], c = s[2], e = s[3];
        if (to_xyxy) {                       // (xc, yc, w, h) -> (x1, y1, x2, y2)
            d[0] = a - c / 2.f; d[1] = b - e / 2.f; d[2] = a + c / 2.f; d[3] = b + e / 2.f;
        } else {                             // (x1, y1, x2, y2) -> (xc, yc, w, h)
            d[0] = (a + c) / 2.f; d[1] = (b + e) / 2.f; d[2] = c - a; d[3] = e - b;
        }
    }
}

__device__ inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }   // NaN passes through like torch.clamp_

__global__ void scale_coords_kernel(float* __restrict__ boxes, int n, int ld, float pad_x, float pad_y, float gain,
                                    float w0, float h0, int do_scale) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float* b = boxes + (long)i * ld;
        float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
        if (do_scale) {
            x1 = (x1 - pad_x) / gain; x2 = (x2 - pad_x) / gain;
            y1 = (y1 - pad_y) / gain; y2 = (y2 - pad_y) / gain;
        }
        b[0] = clampf(x1, 0.f, w0); b[1] = clampf(y1, 0.f, h0);
        b[2] = clampf(x2, 0.f, w0); b[3] = clampf(y2, 0.f, h0);
    }
}

}  // namespace

extern "C" int dyk_box_convert(const float* in, float* out, int32_t n, int32_t ld_in, int32_t ld_out, int32_t to_xyxy, void* stream) {
    if (n == 0) return DYK_OK;
    if (!in || !out || n < 0 || ld_in < 4 || ld_out < 4) return DYK_ERR_ARG;
    const int grid = (n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256;
    hipLaunchKernelGGL(box_convert_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, n, ld_in, ld_out, to_xyxy);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_scale_coords(float* boxes, int32_t n, int32_t ld, float pad_x, float pad_y, float gain, float w0, float h0,
                                int32_t do_scale, void* stream) {
    if (n == 0) return DYK_OK;
    if (!boxes || n < 0 || ld < 4 || (do_scale && gain == 0.f)) return DYK_ERR_ARG;
    const int grid = (n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256;
    hipLaunchKernelGGL(scale_coords_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, boxes, n, ld, pad_x, pad_y, gain, w0, h0, do_scale);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
