// YOLO target assignment and loss (forward value + analytic gradient w.r.t. the head logits) on the
// device, fp32 values / int64 indices.  Replaces build_targets (reference build_utils/utils.py:296-384)
// and compute_loss (:209-293) including bbox_iou (:95-138) and wh_iou (:166-171).
//
// Index arithmetic (anchor matching threshold, image / anchor / cell indices, match order) is
// bit-exact w.r.t. the torch CPU evaluation: every fp32 operation below is written in the same order
// as the reference expression and floating-point contraction is disabled for this file.
//
// Structure per call (all on the caller's stream, no host sync):
//   memset(dp, tobj, acc) -> build_targets_kernel (1 block / head, ordered compaction)
//   -> match_loss_kernel (1 block / head: IoU loss + class BCE + objectness targets + their gradients)
//   -> obj_loss_kernel (dense objectness BCE + gradient) -> finalize_kernel (3 scalars)
// Gradients of the IoU term are obtained with forward-mode dual numbers over (x, y, w, h).
#include "dyk_common.h"

#pragma clang fp contract(off)

namespace {

// ---------------------------------------------------------------- dual numbers (value + d/d{x,y,w,h})
struct D4 {
    float v;
    float d[4];
};
__device__ inline D4 dconst(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ inline D4 dvar(float v, int k) { D4 r = dconst(v); r.d[k] = 1.f; return r; }
__device__ inline D4 operator+(const D4& a, const D4& b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ inline D4 operator-(const D4& a, const D4& b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ inline D4 operator*(const D4& a, const D4& b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ inline D4 operator/(const D4& a, const D4& b) {
    D4 r; r.v = a.v / b.v;
    const float inv = 1.f / b.v;
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
__device__ inline D4 operator+(const D4& a, float c) { D4 r = a; r.v = a.v + c; return r; }
__device__ inline D4 operator*(const D4& a, float c) { D4 r; r.v = a.v * c; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * c; return r; }
__device__ inline D4 dmin(const D4& a, const D4& b) { return a.v <= b.v ? a : b; }
__device__ inline D4 dmax(const D4& a, const D4& b) { return a.v >= b.v ? a : b; }
__device__ inline D4 dclamp0(const D4& a) { return a.v >= 0.f ? a : dconst(0.f); }
__device__ inline D4 datan(const D4& a) {
    D4 r; r.v = atanf(a.v);
    const float g = 1.f / (1.f + a.v * a.v);
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * g;
    return r;
}
__device__ inline D4 dnograd(const D4& a) { return dconst(a.v); }

// bbox_iou(pbox.T, tbox, x1y1x2y2=False, GIoU / CIoU)   utils.py:95-138
__device__ inline D4 box_iou_dual(const D4 p[4], const float t[4], bool ciou) {
    const D4 half_w = p[2] * 0.5f, half_h = p[3] * 0.5f;          // box1[2] / 2
    const D4 b1x1 = p[0] - half_w, b1x2 = p[0] + half_w;
    const D4 b1y1 = p[1] - half_h, b1y2 = p[1] + half_h;
    const D4 b2x1 = dconst(t[0] - t[2] / 2), b2x2 = dconst(t[0] + t[2] / 2);
    const D4 b2y1 = dconst(t[1] - t[3] / 2), b2y2 = dconst(t[1] + t[3] / 2);
    const D4 inter = dclamp0(dmin(b1x2, b2x2) - dmax(b1x1, b2x1)) * dclamp0(dmin(b1y2, b2y2) - dmax(b1y1, b2y1));
    const D4 w1 = b1x2 - b1x1, h1 = b1y2 - b1y1;
    const D4 w2 = b2x2 - b2x1, h2 = b2y2 - b2y1;
    const D4 uni = ((w1 * h1 + 1e-16f) + w2 * h2) - inter;
    const D4 iou = inter / uni;
    const D4 cw = dmax(b1x2, b2x2) - dmin(b1x1, b2x1);
    const D4 ch = dmax(b1y2, b2y2) - dmin(b1y1, b2y1);
    if (!ciou) {
        const D4 c_area = cw * ch + 1e-16f;
        return iou - (c_area - uni) / c_area;
    }
    const D4 c2 = (cw * cw + ch * ch) + 1e-16f;
    const D4 dx = (b2x1 + b2x2) - (b1x1 + b1x2), dy = (b2y1 + b2y2) - (b1y1 + b1y2);
    const D4 rho2 = (dx * dx) * 0.25f + (dy * dy) * 0.25f;
    const D4 da = datan(w2 / h2) - datan(w1 / h1);
    const D4 v = (da * da) * (float)(4.0 / (M_PI * M_PI));
    const D4 alpha = dnograd(v / ((dconst(1.f) - iou) + v));
    return iou - (rho2 / c2 + v * alpha);
}

__device__ inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------- build_targets
// One block per head.  Pairs (a, t) are visited in anchor-major order and compacted in that order
// (utils.py:361: `at[j]`, `t.repeat(na,1,1)[j]` with j of shape [na, nt]).
__global__ __launch_bounds__(256) void build_targets_kernel(DykTargetsDesc d) {
    const int h = blockIdx.x;
    const int nt = d.nt, na = d.na;
    const float nxf = (float)d.nx[h], nyf = (float)d.ny[h];
    const long cap = (long)na * nt;
    long* ib = d.indices + (long)h * 4 * cap;          // b | a | gj | gi
    float* tb = d.tbox + (long)h * 4 * cap;
    float* an = d.anch + (long)h * 2 * cap;
    long* tc = d.tcls + (long)h * cap;
    __shared__ int wave_cnt[4];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long start = 0; start < cap; start += 256) {
        const long idx = start + threadIdx.x;
        bool match = false;
        int a = 0, t = 0;
        float gw = 0.f, gh = 0.f;
        if (idx < cap) {
            a = (int)(idx / nt);
            t = (int)(idx - (long)a * nt);
            gw = d.targets[t * 6 + 4] * nxf;           // targets * gain, gain = (1,1,nx,ny,nx,ny)  :328,:339
            gh = d.targets[t * 6 + 5] * nyf;
            const float aw = d.anchor_vec[h][2 * a], ah = d.anchor_vec[h][2 * a + 1];
            const float inter = fminf(aw, gw) * fminf(ah, gh);            // wh_iou :166-171
            const float iou = inter / ((aw * ah + gw * gh) - inter);
            match = iou > d.iou_t;                                          // :352
        }
        const unsigned long long m = __ballot(match);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[w] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int q = 0; q < w; ++q) off += wave_cnt[q];
        if (match) {
            const long o = off + before;
            const float gx = d.targets[t * 6 + 2] * nxf, gy = d.targets[t * 6 + 3] * nyf;
            const long gi = (long)gx, gj = (long)gy;                        // .long() truncation :370
            ib[0 * cap + o] = (long)d.targets[t * 6 + 0];                    // image index  :367
            ib[1 * cap + o] = a;
            ib[2 * cap + o] = gj;
            ib[3 * cap + o] = gi;
            tb[o * 4 + 0] = gx - (float)gi;
            tb[o * 4 + 1] = gy - (float)gj;
            tb[o * 4 + 2] = gw;
            tb[o * 4 + 3] = gh;
            an[o * 2 + 0] = d.anchor_vec[h][2 * a];
            an[o * 2 + 1] = d.anchor_vec[h][2 * a + 1];
            tc[o] = (long)d.targets[t * 6 + 1];
        }
        __syncthreads();
        if (threadIdx.x == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) d.counts[h] = base;
}

// One element of nn.BCEWithLogitsLoss(pos_weight=pw, reduction='none') and its derivative, optionally wrapped in the
// reference's FocalLoss (utils.py:184-194): loss *= alpha_factor * (1 - p_t)^gamma with p = sigmoid(x),
// p_t = z p + (1 - z)(1 - p), alpha_factor = z alpha + (1 - z)(1 - alpha).  z may be a soft target (objectness: the
// detached IoU ratio), exactly as the reference feeds it.  d/dx: af * (bce' * m + bce * m'),
// m' = -gamma (1 - p_t)^(gamma - 1) (2z - 1) p (1 - p).
__device__ inline void bce_elem(float x, float z, float pw, float gamma, float alpha, float& loss, float& grad) {
    const float lw = 1.f + (pw - 1.f) * z;
    const float l = (1.f - z) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
    const float s = sigmoidf_(x);
    const float dl = s * (1.f - z + pw * z) - pw * z;
    if (!(gamma > 0.f)) { loss = l; grad = dl; return; }
    const float pt = z * s + (1.f - z) * (1.f - s);
    const float af = z * alpha + (1.f - z) * (1.f - alpha);
    const float q = 1.f - pt;
    const float m = powf(q, gamma);
    const float dm = -gamma * powf(q, gamma - 1.f) * (2.f * z - 1.f) * s * (1.f - s);
    loss = l * af * m;
    grad = af * (dl * m + l * dm);
}

// ---------------------------------------------------------------- matched-cell terms
// acc layout per head: [0] sum(1-iou)  [1] valid matches  [2] sum obj BCE  [3] sum cls BCE
__global__ __launch_bounds__(256) void match_loss_kernel(DykLossDesc d, DykTargetsDesc td) {
    const int h = blockIdx.x;
    const int n = td.counts[h];
    if (n <= 0) return;
    const long cap = (long)td.na * td.nt;
    const long* ib = td.indices + (long)h * 4 * cap;
    const float* tb = td.tbox + (long)h * 4 * cap;
    const float* an = td.anch + (long)h * 2 * cap;
    const long* tc = td.tcls + (long)h * cap;
    const int ny = td.ny[h], nx = td.nx[h], na = td.na, no = d.no, nc = d.nc;
    const float* p = d.p[h];
    float* dp = d.dp[h];
    float* tobj = d.tobj[h];
    // number of in-range matches (the reference would raise on an out-of-range index)
    __shared__ int nvalid;
    if (threadIdx.x == 0) nvalid = 0;
    __syncthreads();
    int mine = 0;
    for (int m = threadIdx.x; m < n; m += blockDim.x) {
        const long b = ib[m], gj = ib[2 * cap + m], gi = ib[3 * cap + m];
        const long cls = tc[m];
        const bool ok = b >= 0 && b < d.B && gj >= 0 && gj < ny && gi >= 0 && gi < nx && (nc <= 1 || (cls >= 0 && cls < nc));
        if (ok) ++mine; else atomicOr(d.flag, 1);
    }
    atomicAdd(&nvalid, mine);
    __syncthreads();
    const int nv = nvalid;
    if (nv == 0) return;
    const float inv_n = 1.f / (float)nv;
    // cell of every match in LDS (when they fit): the "last match of a cell wins" scan below then compares one LDS word per
    // candidate instead of four 8-byte global loads (400 matches on a head: 75 -> 25 us, on the path from the forward to the
    // backward pass); -1 = out of range, never equal to a valid cell
    constexpr int KEYS = 4096;
    __shared__ long keys[KEYS];
    const bool keyed = n <= KEYS;
    if (keyed) {
        for (int m = threadIdx.x; m < n; m += blockDim.x) {
            const long b = ib[m], a = ib[cap + m], gj = ib[2 * cap + m], gi = ib[3 * cap + m];
            keys[m] = (b >= 0 && b < d.B && gj >= 0 && gj < ny && gi >= 0 && gi < nx) ? ((b * na + a) * ny + gj) * nx + gi : -1;
        }
        __syncthreads();
    }
    float sbox = 0.f, scls = 0.f;
    for (int m = threadIdx.x; m < n; m += blockDim.x) {
        const long b = ib[m], a = ib[cap + m], gj = ib[2 * cap + m], gi = ib[3 * cap + m];
        if (!(b >= 0 && b < d.B && gj >= 0 && gj < ny && gi >= 0 && gi < nx)) continue;
        const long cell = ((b * na + a) * ny + gj) * nx + gi;
        const float* ps = p + cell * no;
        float* g = dp + cell * no;
        const float aw = an[m * 2], ah = an[m * 2 + 1];
        // predicted box and its derivative w.r.t. the logits  (utils.py:258-262)
        float pv[4], dpdt[4];
        if (d.v4) {
            const float sx = sigmoidf_(ps[0]), sy = sigmoidf_(ps[1]), sw = sigmoidf_(ps[2]), sh = sigmoidf_(ps[3]);
            pv[0] = sx * 2.f - 0.5f; dpdt[0] = 2.f * sx * (1.f - sx);
            pv[1] = sy * 2.f - 0.5f; dpdt[1] = 2.f * sy * (1.f - sy);
            const float w2 = sw * 2.f, h2 = sh * 2.f;
            pv[2] = (w2 * w2) * aw; dpdt[2] = 8.f * sw * sw * (1.f - sw) * aw;
            pv[3] = (h2 * h2) * ah; dpdt[3] = 8.f * sh * sh * (1.f - sh) * ah;
        } else {
            const float sx = sigmoidf_(ps[0]), sy = sigmoidf_(ps[1]);
            pv[0] = sx; dpdt[0] = sx * (1.f - sx);
            pv[1] = sy; dpdt[1] = sy * (1.f - sy);
            const float ew = expf(ps[2]), eh = expf(ps[3]);
            pv[2] = fminf(ew, 1e3f) * aw; dpdt[2] = ew < 1e3f ? ew * aw : 0.f;
            pv[3] = fminf(eh, 1e3f) * ah; dpdt[3] = eh < 1e3f ? eh * ah : 0.f;
        }
        D4 pb[4];
        for (int k = 0; k < 4; ++k) pb[k] = dvar(pv[k], k);
        const D4 iou = box_iou_dual(pb, tb + m * 4, d.ciou != 0);
        sbox += 1.f - iou.v;
        // d(hyp_box * mean(1 - iou)) / d logits
        for (int k = 0; k < 4; ++k) atomicAdd(g + k, -d.hyp_box * inv_n * iou.d[k] * dpdt[k]);
        // objectness target: last match in (anchor-major, target) order wins for a shared cell (:271)
        bool last = true;
        if (keyed) {
            for (int q = m + 1; q < n; ++q)
                if (keys[q] == cell) { last = false; break; }
        } else {
            for (int q = m + 1; q < n; ++q)
                if (ib[q] == b && ib[cap + q] == a && ib[2 * cap + q] == gj && ib[3 * cap + q] == gi) { last = false; break; }
        }
        if (last) tobj[cell] = (1.f - d.gr) + d.gr * fmaxf(iou.v, 0.f);
        if (nc > 1) {                                                        // class BCE  :274-277
            const long cls = tc[m];
            const float inv_cls = inv_n / (float)nc;
            for (int c = 0; c < nc; ++c) {
                const float x = ps[5 + c], z = (c == cls) ? 1.f : 0.f;
                float l, dl;
                bce_elem(x, z, d.cls_pw, d.fl_gamma, d.fl_alpha, l, dl);
                scls += l;
                atomicAdd(g + 5 + c, d.hyp_cls * inv_cls * dl);
            }
        }
    }
    __shared__ float ws[2][4];
    sbox = wave_sum(sbox); scls = wave_sum(scls);
    if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = sbox; ws[1][threadIdx.x >> 6] = scls; }
    __syncthreads();
    if (threadIdx.x == 0) {
        d.acc[h * 4 + 0] = (double)(ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3]);
        d.acc[h * 4 + 1] = (double)nv;
        d.acc[h * 4 + 3] = (double)(ws[1][0] + ws[1][1] + ws[1][2] + ws[1][3]);
    }
}

// ---------------------------------------------------------------- dense objectness BCE (:283)
__global__ __launch_bounds__(256) void obj_loss_kernel(DykLossDesc d, DykTargetsDesc td) {
    const int h = blockIdx.y;                                   // one launch for the heads (was one per head)
    const long ncell = (long)d.B * td.na * td.ny[h] * td.nx[h];
    const float* p = d.p[h];
    float* dp = d.dp[h];
    const float* tobj = d.tobj[h];
    const int no = d.no;
    const float pw = d.obj_pw;
    const float gscale = d.hyp_obj / (float)ncell;
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < ncell; i += (long)gridDim.x * blockDim.x) {
        const float x = p[i * no + 4], z = tobj[i];
        float l, dl;
        bce_elem(x, z, pw, d.fl_gamma, d.fl_alpha, l, dl);
        s += l;
        dp[i * no + 4] = gscale * dl;
    }
    __shared__ float ws[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(d.acc + h * 4 + 2, (double)(ws[0] + ws[1] + ws[2] + ws[3]));
}

__global__ void loss_finalize_kernel(DykLossDesc d, DykTargetsDesc td) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double lbox = 0.0, lobj = 0.0, lcls = 0.0;
    for (int h = 0; h < d.nheads; ++h) {
        const double nv = d.acc[h * 4 + 1];
        if (nv > 0.0) {
            lbox += d.acc[h * 4 + 0] / nv;
            if (d.nc > 1) lcls += d.acc[h * 4 + 3] / (nv * d.nc);
        }
        const double ncell = (double)d.B * td.na * td.ny[h] * td.nx[h];
        lobj += d.acc[h * 4 + 2] / ncell;
    }
    d.out[0] = (float)(lbox * d.hyp_box);
    d.out[1] = (float)(lobj * d.hyp_obj);
    d.out[2] = (float)(lcls * d.hyp_cls);
}

// dp[..., 0:4] *= g[0]; dp[..., 4] *= g[1]; dp[..., 5:] *= g[2]; the three factors from separate device scalars (NULL = 0:
// that loss term was not part of the differentiated sum)
__global__ void loss_scale_grads3_kernel(float* dp, long n, int no, const float* gb, const float* go, const float* gc) {
    const float g0 = gb ? gb[0] : 0.f, g1 = go ? go[0] : 0.f, g2 = gc ? gc[0] : 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % no);
        dp[i] *= (c < 4) ? g0 : (c == 4 ? g1 : g2);
    }
}
__global__ void loss_scale_grads_kernel(float* dp, long n, int no, const float* g) {
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % no);
        dp[i] *= (c < 4) ? g0 : (c == 4 ? g1 : g2);
    }
}

int check_targets(const DykTargetsDesc* t) {
    if (!t || t->nheads <= 0 || t->nheads > 3 || t->na <= 0 || t->na > 8 || t->nt < 0) return DYK_ERR_ARG;
    if (!t->counts || !t->indices || !t->tbox || !t->anch || !t->tcls) return DYK_ERR_ARG;
    if (t->nt > 0 && !t->targets) return DYK_ERR_ARG;
    return DYK_OK;
}

}  // namespace

extern "C" int dyk_build_targets(const DykTargetsDesc* t, void* stream) {
    const int rc = check_targets(t);
    if (rc) return rc;
    hipLaunchKernelGGL(build_targets_kernel, dim3(t->nheads), dim3(256), 0, (hipStream_t)stream, *t);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_yolo_loss(const DykLossDesc* d, const DykTargetsDesc* t, void* stream) {
    int rc = check_targets(t);
    if (rc) return rc;
    if (!d || d->nheads != t->nheads || !d->acc || !d->out || !d->flag || d->no < 5 || d->nc != d->no - 5 || d->B <= 0)
        return DYK_ERR_ARG;
    if (!(d->fl_gamma >= 0.f) || (d->fl_gamma > 0.f && !(d->fl_alpha >= 0.f && d->fl_alpha <= 1.f))) return DYK_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    // dp / tobj / acc / flag start from zero.  A caller that lays them out back to back gets ONE fill instead of
    // 2 * nheads + 2: either  acc (12 doubles) | flag | 4 spare bytes | dp[0] | dp[1] | .. | tobj[0] | ..  (what dyk/detect.py
    // allocates: acc sits at the start of the allocation and is 8-byte aligned whatever the grid sizes are -- with B, ny and
    // nx all odd the float count in front of a trailing acc is odd), or  dp[0] | .. | tobj[..] | acc | flag  when that
    // offset happens to be 8-byte aligned.  Any other layout takes one fill per buffer, the flag word included.
    size_t ncells[3] = {0, 0, 0};
    bool contiguous = true;
    char* expect = (char*)d->dp[0];
    for (int h = 0; h < d->nheads; ++h) {
        if (!d->p[h] || !d->dp[h] || !d->tobj[h]) return DYK_ERR_ARG;
        ncells[h] = (size_t)d->B * t->na * t->ny[h] * t->nx[h];
        contiguous = contiguous && (char*)d->dp[h] == expect;
        expect += ncells[h] * d->no * sizeof(float);
    }
    for (int h = 0; h < d->nheads; ++h) {
        contiguous = contiguous && (char*)d->tobj[h] == expect;
        expect += ncells[h] * sizeof(float);
    }
    const size_t acc_bytes = 12 * sizeof(double);
    const bool head = contiguous && (char*)d->flag == (char*)d->acc + acc_bytes && (char*)d->dp[0] == (char*)d->flag + 8;
    const bool tail = contiguous && (char*)d->acc == expect;
    if (head) {
        DYK_HIP_TRY(hipMemsetAsync(d->acc, 0, (size_t)(expect - (char*)d->acc), s));
    } else if (tail) {
        const bool flag_behind = (char*)d->flag == expect + acc_bytes;
        DYK_HIP_TRY(hipMemsetAsync(d->dp[0], 0, (size_t)(expect - (char*)d->dp[0]) + acc_bytes + (flag_behind ? sizeof(int32_t) : 0), s));
        if (!flag_behind) DYK_HIP_TRY(hipMemsetAsync(d->flag, 0, sizeof(int32_t), s));
    } else {
        for (int h = 0; h < d->nheads; ++h) {
            DYK_HIP_TRY(hipMemsetAsync(d->dp[h], 0, ncells[h] * d->no * sizeof(float), s));
            DYK_HIP_TRY(hipMemsetAsync(d->tobj[h], 0, ncells[h] * sizeof(float), s));
        }
        DYK_HIP_TRY(hipMemsetAsync(d->acc, 0, acc_bytes, s));
        DYK_HIP_TRY(hipMemsetAsync(d->flag, 0, sizeof(int32_t), s));
    }
    hipLaunchKernelGGL(build_targets_kernel, dim3(t->nheads), dim3(256), 0, s, *t);
    DYK_LAUNCH_CHECK();
    hipLaunchKernelGGL(match_loss_kernel, dim3(d->nheads), dim3(256), 0, s, *d, *t);
    DYK_LAUNCH_CHECK();
    {
        size_t big = ncells[0] > ncells[1] ? ncells[0] : ncells[1];
        if (ncells[2] > big) big = ncells[2];
        long g = ((long)big + 1023) / 1024;
        if (g > 512) g = 512;
        if (g < 1) g = 1;
        hipLaunchKernelGGL(obj_loss_kernel, dim3((int)g, d->nheads), dim3(256), 0, s, *d, *t);
        DYK_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, *d, *t);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_loss_scale_grads3(float* dp, int64_t n, int32_t no, const float* gbox, const float* gobj, const float* gcls, void* stream) {
    if (!dp || n <= 0 || no < 5) return DYK_ERR_ARG;
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(loss_scale_grads3_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, dp, (long)n, no, gbox, gobj, gcls);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_loss_scale_grads(float* dp, int64_t n, int32_t no, const float* g3, void* stream) {
    if (!dp || !g3 || n <= 0 || no < 5) return DYK_ERR_ARG;
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(loss_scale_grads_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, dp, (long)n, no, g3);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
