// Per-image detection filtering + greedy non-maximum suppression on the device.
// Replaces non_max_suppression (reference build_utils/utils.py:387-464) together with the
// torchvision.ops.nms call inside it (:448).  One 1024-thread workgroup per image:
//   1. candidate generation in row order (ordered compaction by wave ballot + prefix):
//      obj > conf_thres, 2 < w,h < 4096, conf = obj*cls, best class (or every class when
//      multi_label), conf > conf_thres, optional class filter; boxes converted centre->corner;
//   2. stable descending sort by score: bitonic sort of 64-bit keys (inverted score bits | index);
//   3. greedy suppression in score order, 64 candidates per step on one wavefront: each lane tests
//      its candidate against the kept list (LDS), then the wave resolves the chunk internally with
//      cross-lane broadcasts; stops at max_num kept boxes (the reference truncates [:max_num]).
// IoU arithmetic follows torchvision's CPU kernel operation for operation in fp32 (no FMA
// contraction) so that the keep-set is bit-identical for identical inputs.
#include "dyk_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int NT = 1024;
constexpr int MAX_KEEP = 512;

struct Cand {
    float x1, y1, x2, y2, conf, cls;
    int row;
};

__device__ inline float iou_tv(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2,
                               float by2, float barea) {
    const float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1);
    const float xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    return inter / ((aarea + barea) - inter);
}

__global__ __launch_bounds__(NT) void nms_kernel(DykNmsDesc d) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nc = d.no - 5;
    const bool multi = d.multi_label && nc > 1;
    const long cap = (long)d.N * (multi ? nc : 1);
    long P = 1;
    while (P < cap) P <<= 1;
    char* wsb = (char*)d.ws + (size_t)b * d.ws_per_image;
    unsigned long long* keys = (unsigned long long*)wsb;               // [P]
    Cand* cand = (Cand*)(wsb + P * sizeof(unsigned long long));        // [cap]
    const float* pred = d.pred + (long)b * d.N * d.no;

    __shared__ int wave_cnt[NT / 64];
    __shared__ int base_s;
    __shared__ float kx1[MAX_KEEP], ky1[MAX_KEEP], kx2[MAX_KEEP], ky2[MAX_KEEP], kar[MAX_KEEP];
    __shared__ int nkept_s;
    if (tid == 0) { base_s = 0; nkept_s = 0; }
    __syncthreads();

    // ---------------------------------------------------------------- 1. candidates
    for (long start = 0; start < cap; start += NT) {
        const long idx = start + tid;
        bool ok = false;
        Cand c;
        if (idx < cap) {
            const int row = (int)(multi ? idx / nc : idx);
            const float* x = pred + (long)row * d.no;
            const float obj = x[4];
            if (obj > d.conf_thres && x[2] > 2.f && x[2] < 4096.f && x[3] > 2.f && x[3] < 4096.f) {   // :408-409
                float conf;
                int cls;
                if (multi) {
                    cls = (int)(idx - (long)row * nc);
                    conf = x[5 + cls] * obj;                                  // :416
                } else {
                    conf = x[5] * obj;
                    cls = 0;
                    for (int k = 1; k < nc; ++k) {                            // first maximum wins
                        const float v = x[5 + k] * obj;
                        if (v > conf) { conf = v; cls = k; }
                    }
                }
                ok = conf > d.conf_thres;                                     // :423 / :427
                if (ok && d.n_classes > 0) {                                  // :430-431
                    bool in = false;
                    for (int k = 0; k < d.n_classes; ++k) in |= (d.classes[k] == cls);
                    ok = in;
                }
                if (ok) {
                    c.x1 = x[0] - x[2] / 2; c.y1 = x[1] - x[3] / 2;           // xywh2xyxy :50-57
                    c.x2 = x[0] + x[2] / 2; c.y2 = x[1] + x[3] / 2;
                    c.conf = conf; c.cls = (float)cls; c.row = row;
                }
            }
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int q = 0; q < wv; ++q) off += wave_cnt[q];
        if (ok) {
            const int o = off + __popcll(m & ((1ull << lane) - 1ull));
            cand[o] = c;
            const unsigned int sb = __float_as_uint(c.conf);                  // conf > 0: bits are monotonic
            keys[o] = ((unsigned long long)(0xFFFFFFFFu - sb) << 32) | (unsigned int)o;
        }
        __syncthreads();
        if (tid == 0) { int s = 0; for (int q = 0; q < NT / 64; ++q) s += wave_cnt[q]; base_s += s; }
        __syncthreads();
    }
    const int n = base_s;
    if (n == 0) {
        if (tid == 0) d.counts[b] = 0;
        return;
    }
    // ---------------------------------------------------------------- 2. sort (bitonic, padded to a power of two)
    long Pn = 1;
    while (Pn < n) Pn <<= 1;
    for (long i = n + tid; i < Pn; i += NT) keys[i] = ~0ull;
    __syncthreads();
    for (long k = 2; k <= Pn; k <<= 1) {
        for (long j = k >> 1; j > 0; j >>= 1) {
            for (long i = tid; i < Pn; i += NT) {
                const long ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], bb = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > bb) == up) { keys[i] = bb; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    // ---------------------------------------------------------------- 3. greedy suppression (wave 0)
    const int max_keep = d.max_num < MAX_KEEP ? d.max_num : MAX_KEEP;
    if (wv == 0) {
        int nkept = 0;
        const float off_mul = d.agnostic ? 0.f : 4096.f;                       // :446-447 class offset
        for (int start = 0; start < n && nkept < max_keep; start += 64) {
            const int i = start + lane;
            bool alive = i < n;
            int ci = 0;
            float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, area = 0.f;
            if (alive) {
                ci = (int)(keys[i] & 0xFFFFFFFFu);
                const Cand c = cand[ci];
                const float o = c.cls * off_mul;
                x1 = c.x1 + o; y1 = c.y1 + o; x2 = c.x2 + o; y2 = c.y2 + o;
                area = (x2 - x1) * (y2 - y1);
                for (int k = 0; k < nkept && alive; ++k)
                    if (iou_tv(kx1[k], ky1[k], kx2[k], ky2[k], kar[k], x1, y1, x2, y2, area) > d.iou_thres) alive = false;
            }
            for (int s = 0; s < 64 && nkept < max_keep; ++s) {
                const bool sa = __shfl((int)alive, s, 64) != 0;
                if (!sa) continue;
                const float sx1 = __shfl(x1, s, 64), sy1 = __shfl(y1, s, 64), sx2 = __shfl(x2, s, 64), sy2 = __shfl(y2, s, 64);
                const float sar = __shfl(area, s, 64);
                if (lane == s) {
                    kx1[nkept] = x1; ky1[nkept] = y1; kx2[nkept] = x2; ky2[nkept] = y2; kar[nkept] = area;
                    const Cand c = cand[ci];
                    float* o = d.out + ((long)b * d.max_num + nkept) * 6;
                    o[0] = c.x1; o[1] = c.y1; o[2] = c.x2; o[3] = c.y2; o[4] = c.conf; o[5] = c.cls;
                    d.out_rows[(long)b * d.max_num + nkept] = c.row;
                }
                if (lane > s && alive && iou_tv(sx1, sy1, sx2, sy2, sar, x1, y1, x2, y2, area) > d.iou_thres) alive = false;
                ++nkept;
            }
            __builtin_amdgcn_wave_barrier();       // kept-list LDS writes precede the next chunk's reads
        }
        if (lane == 0) d.counts[b] = nkept;
    }
}

}  // namespace

extern "C" int64_t dyk_nms_workspace_bytes(int32_t N, int32_t no, int32_t multi_label) {
    const int nc = no - 5;
    const long cap = (long)N * ((multi_label && nc > 1) ? nc : 1);
    long P = 1;
    while (P < cap) P <<= 1;
    const long bytes = P * 8 + cap * (long)sizeof(Cand);
    return (bytes + 255) / 256 * 256;
}

extern "C" int dyk_nms(const DykNmsDesc* d, void* stream) {
    if (!d || !d->pred || !d->out || !d->out_rows || !d->counts || !d->ws) return DYK_ERR_ARG;
    if (d->B <= 0 || d->N <= 0 || d->no < 6 || d->max_num <= 0 || d->max_num > MAX_KEEP) return DYK_ERR_ARG;
    if (d->n_classes < 0 || d->n_classes > 16) return DYK_ERR_ARG;
    if (d->ws_per_image < dyk_nms_workspace_bytes(d->N, d->no, d->multi_label)) return DYK_ERR_ARG;
    if (!(d->conf_thres >= 0.f)) return DYK_ERR_ARG;     // score bits must be monotonic (positive floats)
    hipLaunchKernelGGL(nms_kernel, dim3(d->B), dim3(NT), 0, (hipStream_t)stream, *d);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
