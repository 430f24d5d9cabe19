// 3x3 data gradients into 32-CHANNEL tensors with the fused BatchNorm-backward epilogue (DYK_EPI_BNBWD), round 4:
// resident weights, one patch per tile for every tap and every output-parity class, epilogue straight from the accumulators.
//
// The data gradient of the first 3x3 stride-2 conv of each backbone (32 -> 64 at 512x640 -> 256x320) writes a 336 MB tensor from
// 24 GMAC of work: an HBM / VALU job (floor 105 us).  The generic kernel ran it as 4 parity classes x 10 240 tiles = 40 960
// workgroups of 128 pixels x 32 channels, each re-staging its gradient tile per tap and writing half cache lines: 410 us, and it
// sits at the very end of the backward pass where nothing overlaps it.  Its stride-1 sibling (64 -> 32 at 256x320): 167 us for a
// 52 us floor.  Here
//   * a workgroup is persistent: it loads the whole packed weight [9][32][K] into LDS once (37 KB for K = 64) and walks tiles
//     of 8 x 16 launch-grid positions; the gradient patch of a tile (with the taps' halo: 9 x 17 or 10 x 18 pixels) is staged
//     ONCE by LDS-DMA, double buffered (the next tile's patch lands while this one is computed);
//   * all parity classes of a tile are computed from that one patch (16 accumulator tiles per wave for stride 2), so the
//     workgroup owns a contiguous 16 x 32 block of output pixels: whole lines of y / res, every line touched by one workgroup;
//   * the epilogue runs from the accumulators: lane (pixel, channel quad) loads its 8 bytes of the raw conv output, computes
//     da = g * act'(u), adds the two BatchNorm sums and stores 8 bytes; the sums stay in registers across ALL tiles of the
//     workgroup and are folded once (DPP row sums -> LDS -> one fp64 atomic per channel and workgroup).
// Same arithmetic as the generic kernel's staged epilogue: the gradient is rounded to bf16 before act' is applied.
//
// Replaces torch autograd's conv backward-data for nn.Conv2d(32, 64, 3, 2) / nn.Conv2d(32, 64, 3, 1) behind reference
// models.py:34-42 (with the BatchNorm2d + activation backward of the producing block, models.py:43-62, fused in).
#include <stddef.h>
#include <string.h>
#include <type_traits>
#include "dyk_common.h"

namespace {

#define LDS_AS __attribute__((address_space(3)))
__device__ uint4 dyk_sc_zero_page[8];

struct ScArgs {
    DykConvDesc d;
    int PH, PW, npatch;          // patch rows / columns / pixels
    int miny, minx;              // smallest tap offset: patch pixel (0, 0) = grid position (y0 + miny, x0 + minx)
    int tiles_y, tiles_x, ntiles;
    int tap_shift[9];            // patch-pixel offset of tap t: (tdy - miny) * PW + (tdx - minx)
    int tap_w[9];                // weight tap index
    int cls_first[4], cls_ntaps[4], cls_ooy[4], cls_oox[4];
};

__device__ inline void sc_glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}
__device__ inline unsigned sc_lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS const char*)p);
}

constexpr int SC_M = 32, SC_TH = 8, SC_TW = 16;

template <int NCLS, int K> struct ScCfg {
    static constexpr int PITCH = K * 2;                         // bytes per pixel / weight row
    static constexpr int NCH = PITCH / 16;                      // 16-byte chunks per row
    static constexpr int RPI = 1024 / PITCH;                    // rows per DMA instruction
    static constexpr int R256 = PITCH >= 256 ? 1 : 256 / PITCH; // rows per 256 bytes of LDS (one pass over the banks)
    static constexpr int PROWS = NCLS == 4 ? 9 * 17 : 10 * 18;  // patch pixels: taps in {0,1} (stride 2) | {-1,0,1}
    static constexpr int NPI = ((PROWS + RPI - 1) / RPI + 3) / 4;   // patch DMA instructions per wave
    static constexpr int P_BYTES = NPI * 4 * 1024;
    static constexpr int W_BYTES = 9 * SC_M * PITCH;
    static constexpr int NWI = W_BYTES / 1024 / 4;              // weight DMA instructions per wave
    static constexpr int LDS = W_BYTES + 2 * P_BYTES + 4 * 2 * SC_M * 4 + 4 * SC_M * 4;
    static_assert(W_BYTES % 4096 == 0, "weights in whole instructions per wave");
};
// chunk swizzle: sixteen lanes read the same logical chunk of sixteen consecutive rows -> sixteen different bank groups
template <int K> __device__ inline int sc_key(int row) { return (row / ScCfg<1, K>::R256) & (ScCfg<1, K>::NCH - 1); }

template <int NCLS, int K, int ACTB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_sc_kernel(const ScArgs g) {
    using C = ScCfg<NCLS, K>;
    using T = bf16_t;
    constexpr int PITCH = C::PITCH, NPI = C::NPI, NWI = C::NWI, RPI = C::RPI, VPR = C::NCH;
    const DykConvDesc& a = g.d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sP = smem + C::W_BYTES;
    float* s_stat = (float*)(smem + C::W_BYTES + 2 * C::P_BYTES);      // [4 waves][2][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const T* zero = (const T*)dyk_sc_zero_page;
    const int Hi_s = __builtin_amdgcn_readfirstlane(a.Hi), Wi_s = __builtin_amdgcn_readfirstlane(a.Wi);
    const int ldx_s = __builtin_amdgcn_readfirstlane(a.ldx), ldy_s = __builtin_amdgcn_readfirstlane(a.ldy), ldr_s = __builtin_amdgcn_readfirstlane(a.ldr);
    const int Ho_s = __builtin_amdgcn_readfirstlane(a.Ho), Wo_s = __builtin_amdgcn_readfirstlane(a.Wo);
    const int osy = __builtin_amdgcn_readfirstlane(a.osy), osx = __builtin_amdgcn_readfirstlane(a.osx);
    const int PW = g.PW;

    // ---- weights: [9][32][K] -> LDS once, chunk-swizzled on the source side
    {
        const T* wg = (const T*)a.w;
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
            const int q = j * 4 + wv;
            const int row = q * RPI + lane / VPR, ps = lane % VPR;       // row = tap * 32 + m
            const int lc = ps ^ sc_key<K>(row);
            sc_glds16(wg + (long)row * K + lc * 8, sc_lds_addr(sW + q * 1024));
        }
    }
    // ---- patch DMA slots (as conv_wgrad_rb.hip): element offset relative to the patch corner + packed coordinates
    int p_eoff[NPI], p_yx[NPI];
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
        const int row = (j * 4 + wv) * RPI + lane / VPR, ps = lane % VPR;
        const int lc = ps ^ sc_key<K>(row);
        const int py = row / PW, px = row - py * PW;
        p_yx[j] = row < g.npatch ? (py << 16) | px : (0x7000 << 16);
        p_eoff[j] = (py * Wi_s + px) * ldx_s + lc * 8;
    }
    auto stage_patch = [&](int tile, int buf) {
        const int tx = tile % g.tiles_x, r = tile / g.tiles_x;
        const int ty = r % g.tiles_y, b = r / g.tiles_y;
        const int ys = ty * SC_TH + g.miny, xs = tx * SC_TW + g.minx;      // grid position of the patch corner
        const T* x00 = (const T*)a.x + (long)((b * Hi_s + ys) * Wi_s + xs) * ldx_s;
        char* dst = sP + buf * C::P_BYTES;
#pragma unroll
        for (int j = 0; j < NPI; ++j) {
            const int hy = p_yx[j] >> 16, hx = p_yx[j] & 0xffff;
            const bool ok = ((unsigned)(ys + hy) < (unsigned)Hi_s) & ((unsigned)(xs + hx) < (unsigned)Wi_s);
            const T* cand = x00 + p_eoff[j];
            sc_glds16(ok ? cand : zero, sc_lds_addr(dst + (j * 4 + wv) * 1024));
        }
    };

    // ---- BatchNorm vectors [scale | shift | mean | rstd][32] in LDS (read per 16-channel group in the epilogue: 16 live registers
    //      instead of 32), the two sums of this lane's eight channels (mi * 16 + 4 kq + r) in registers across all tiles
    float* s_bn = s_stat + 4 * 2 * SC_M;
    if (tid < 4 * SC_M) {
        const int which = tid / SC_M, ch = tid % SC_M;
        const float* src = which == 0 ? a.scale : which == 1 ? a.shift : which == 2 ? a.aux0 : a.aux1;
        s_bn[tid] = src[ch];
    }
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    int pb[2];                               // patch pixel of this lane's B-fragment pixel under shift 0: tile row 2 wv + ni, column i16
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) pb[ni] = (2 * wv + ni) * PW + i16;
    int wa[2];                               // byte offset of this lane's A-fragment row inside a tap's [32][K] block, chunk kq
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) wa[mi] = (mi * 16 + i16) * PITCH;
    const int wkey[2] = {sc_key<K>(i16), sc_key<K>(16 + i16)};

    int buf = 0;
    int tile = blockIdx.x;
    if (tile < g.ntiles) stage_patch(tile, 0);
    for (; tile < g.ntiles; tile += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tile + (int)gridDim.x < g.ntiles) stage_patch(tile + gridDim.x, buf ^ 1);
        const char* P = sP + buf * C::P_BYTES;

        // tile coordinates, output pixels of this lane and its raw conv output -- requested BEFORE the K loop (the loads do not
        // depend on the accumulators: their latency hides behind the fragment reads and MFMAs; first version: loads after the
        // loop, 92 of 366 us)
        const int tx = tile % g.tiles_x, rr = tile / g.tiles_x;
        const int ty = rr % g.tiles_y, b = rr / g.tiles_y;
        const int xo = (tx * SC_TW + i16) * osx;
        uint2 raw[NCLS][2][2];
        int opix[NCLS][2];
#pragma unroll
        for (int c = 0; c < NCLS; ++c)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int yo = (ty * SC_TH + 2 * wv + ni) * osy + (NCLS > 1 ? g.cls_ooy[c] : a.ooy);
                opix[c][ni] = (b * Ho_s + yo) * Wo_s + xo + (NCLS > 1 ? g.cls_oox[c] : a.oox);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    raw[c][mi][ni] = *(const uint2*)((const T*)a.res + (long)opix[c][ni] * ldr_s + mi * 16 + kq * 4);
            }

        f32x4_t acc[NCLS][2][2];
#pragma unroll
        for (int c = 0; c < NCLS; ++c)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[c][mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // taps per class are compile-time (1, 2, 2, 4 for the parity classes of a stride-2 3x3 conv in (py, px) order | 9): the
        // loop nest unrolls completely and the scheduler overlaps the fragment reads of a tap with the MFMAs of the one before
        {
            int t = 0;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                constexpr int NT[4] = {NCLS == 4 ? 1 : 9, 2, 2, 4};
#pragma unroll
                for (int q = 0; q < NT[c]; ++q, ++t) {
                    const int shift = g.tap_shift[t];
                    const char* Wt = sW + g.tap_w[t] * (SC_M * PITCH);
                    int prow[2], pkey[2];
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) { prow[ni] = pb[ni] + shift; pkey[ni] = sc_key<K>(prow[ni]); }
#pragma unroll
                    for (int kb = 0; kb < K / 32; ++kb) {
                        const int chunk = kq + 4 * kb;
                        bf16x8_t fa[2], fb[2];
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi) fa[mi] = *(const bf16x8_t*)(Wt + wa[mi] + ((chunk ^ wkey[mi]) << 4));
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) fb[ni] = *(const bf16x8_t*)(P + prow[ni] * PITCH + ((chunk ^ pkey[ni]) << 4));
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni)
                                acc[c][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mi], fb[ni], acc[c][mi][ni], 0, 0, 0);
                    }
                }
            }
        }

        // ---- epilogue from the accumulators: acc[c][mi][ni][r] = gradient of channel mi*16 + 4 kq + r at grid position
        //      (y0 + 2 wv + ni, x0 + i16), stored at output pixel (yo * osy + ooy_c, xo * osx + oox_c).  Packed fp32 math
        //      (v_pk_fma_f32 / v_pk_mul_f32: two channels per instruction) -- this epilogue is VALU-bound
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const f32x4_t scv = *(const f32x4_t*)(s_bn + 0 * SC_M + mi * 16 + kq * 4), shv = *(const f32x4_t*)(s_bn + 1 * SC_M + mi * 16 + kq * 4);
            const f32x4_t muv = *(const f32x4_t*)(s_bn + 2 * SC_M + mi * 16 + kq * 4), rsv = *(const f32x4_t*)(s_bn + 3 * SC_M + mi * 16 + kq * 4);
#pragma unroll
            for (int c = 0; c < NCLS; ++c)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const f32x4_t v = acc[c][mi][ni];
                    // the generic kernel stages the gradient as bf16 before the epilogue reads it back: same rounding here
                    const uint32_t g01 = f32x2_to_bf16x2(v[0], v[1]), g23 = f32x2_to_bf16x2(v[2], v[3]);
                    const uint2 y2 = raw[c][mi][ni];
                    uint32_t ow[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t gw = h ? g23 : g01, yw = h ? y2.y : y2.x;
                        const dyk_f32x2_t gv = {__uint_as_float(gw << 16), __uint_as_float(gw & 0xffff0000u)};
                        const dyk_f32x2_t yv = {__uint_as_float(yw << 16), __uint_as_float(yw & 0xffff0000u)};
                        const int j = mi * 4 + 2 * h;
                        const dyk_f32x2_t sc2 = {scv[2 * h], scv[2 * h + 1]}, sh2 = {shv[2 * h], shv[2 * h + 1]};
                        const dyk_f32x2_t mu2 = {muv[2 * h], muv[2 * h + 1]}, rs2 = {rsv[2 * h], rsv[2 * h + 1]};
                        const dyk_f32x2_t u = yv * sc2 + sh2;
                        const dyk_f32x2_t da = gv * act_bwd2_c<ACTB>(u, a.act);
                        const dyk_f32x2_t xh = (yv - mu2) * rs2;
                        const dyk_f32x2_t p = da * xh;
                        s1[j] += da[0]; s1[j + 1] += da[1];
                        s2[j] += p[0]; s2[j + 1] += p[1];
                        ow[h] = f32x2_to_bf16x2(da[0], da[1]);
                    }
                    *(uint2*)((T*)a.y + (long)opix[c][ni] * ldy_s + mi * 16 + kq * 4) = make_uint2(ow[0], ow[1]);
                }
        }
        buf ^= 1;
    }

    // ---- the two BatchNorm sums: 16 pixel lanes (DPP row sum) -> wave slot in LDS -> waves in order -> one fp64 atomic per channel
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = row16_sum(s1[j]); s2[j] = row16_sum(s2[j]); }
    if (i16 == 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s_stat[(wv * 2 + 0) * SC_M + mi * 16 + kq * 4 + r] = s1[mi * 4 + r];
                s_stat[(wv * 2 + 1) * SC_M + mi * 16 + kq * 4 + r] = s2[mi * 4 + r];
            }
    }
    __syncthreads();
    if (tid < 2 * SC_M) {
        const int ml = tid % SC_M, which = tid / SC_M;
        const float tot = (s_stat[(0 * 2 + which) * SC_M + ml] + s_stat[(1 * 2 + which) * SC_M + ml]) +
                          (s_stat[(2 * 2 + which) * SC_M + ml] + s_stat[(3 * 2 + which) * SC_M + ml]);
        double* st = a.stats + (size_t)((unsigned)blockIdx.x % (unsigned)(a.stats_slots > 0 ? a.stats_slots : 1)) * 2 * a.Cout;
        atomicAdd(st + which * a.Cout + ml, (double)tot);
    }
}

bool sc_eligible(const DykConvDesc* d) {
    if (d->dtype != DYK_BF16 || d->ntaps != 9 || d->twin) return false;
    if (d->flags != DYK_EPI_BNBWD) return false;                       // the fused BatchNorm-backward epilogue, no residual chain
    if (d->Cout != SC_M || (d->Cin != 64)) return false;
    if (d->isy != 1 || d->isx != 1 || d->osy != d->osx) return false;
    if (d->Hg % SC_TH || d->Wg % SC_TW) return false;
    if (d->ldx % 8 || d->ldy % 4 || d->ldr % 4 || ((uintptr_t)d->y % 8) || ((uintptr_t)d->res % 8)) return false;
    if (d->ncls > 1) {
        if (d->ncls != 4 || d->osy != 2) return false;
        if (d->Ho != 2 * d->Hg || d->Wo != 2 * d->Wg) return false;
        static const int nt[4] = {1, 2, 2, 4};                           // parity classes in (py, px) order, taps back to back
        for (int c = 0, q = 0; c < 4; q += nt[c], ++c)
            if (d->cls_ntaps[c] != nt[c] || d->cls_first[c] != q) return false;
    } else {
        if (d->osy != 1 || d->Ho != d->Hg || d->Wo != d->Wg) return false;
    }
    for (int t = 0; t < 9; ++t) {
        if (d->ncls > 1 ? (d->tdy[t] < 0 || d->tdy[t] > 1 || d->tdx[t] < 0 || d->tdx[t] > 1)
                        : (d->tdy[t] < -1 || d->tdy[t] > 1 || d->tdx[t] < -1 || d->tdx[t] > 1)) return false;
        if (d->twt[t] < 0 || d->twt[t] > 8) return false;
    }
    return true;
}

template <int NCLS, int K, int ACTB>
int sc_launch(const ScArgs& g, hipStream_t stream) {
    using C = ScCfg<NCLS, K>;
    static DykDeviceOnce attr_set;   // (the attribute is per DEVICE: one flag per device id)
    auto kfn = conv_sc_kernel<NCLS, K, ACTB>;
    if (attr_set.first()) {
        DYK_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
    }
    int grid = 256 * (int)((160 * 1024) / C::LDS);   // as many workgroups per CU as the LDS holds (two for stride 2), persistent over the tiles
    if (grid > g.ntiles) grid = g.ntiles;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), C::LDS, stream, g);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

template <int NCLS, int K>
int sc_dispatch_act(const ScArgs& g, hipStream_t s) {
    switch (g.d.act) {
    case DYK_ACT_MISH: return sc_launch<NCLS, K, DYK_ACT_MISH>(g, s);
    case DYK_ACT_LEAKY: return sc_launch<NCLS, K, DYK_ACT_LEAKY>(g, s);
    default: return sc_launch<NCLS, K, -1>(g, s);
    }
}

}  // namespace

// pixel-tile code 6 of the conv tune word; DYK_ERR_UNSUPPORTED = the caller falls back to the generic tiles
int dyk_conv_launch_sc(const DykConvDesc* d, hipStream_t s) {
    if (!sc_eligible(d)) return DYK_ERR_UNSUPPORTED;
    ScArgs g;
    g.d = *d;
    g.d.twin = nullptr;
    const int ncls = d->ncls > 1 ? 4 : 1;
    g.miny = ncls == 4 ? 0 : -1;
    g.minx = g.miny;
    g.PH = SC_TH + (ncls == 4 ? 1 : 2);
    g.PW = SC_TW + (ncls == 4 ? 1 : 2);
    g.npatch = g.PH * g.PW;
    g.tiles_y = d->Hg / SC_TH;
    g.tiles_x = d->Wg / SC_TW;
    g.ntiles = d->B * g.tiles_y * g.tiles_x;
    for (int t = 0; t < 9; ++t) {
        g.tap_shift[t] = (d->tdy[t] - g.miny) * g.PW + (d->tdx[t] - g.minx);
        g.tap_w[t] = d->twt[t];
    }
    for (int c = 0; c < 4; ++c) {
        g.cls_first[c] = ncls == 4 ? d->cls_first[c] : 0;
        g.cls_ntaps[c] = ncls == 4 ? d->cls_ntaps[c] : (c == 0 ? 9 : 0);
        g.cls_ooy[c] = ncls == 4 ? d->cls_ooy[c] : 0;
        g.cls_oox[c] = ncls == 4 ? d->cls_oox[c] : 0;
    }
    return ncls == 4 ? sc_dispatch_act<4, 64>(g, s) : sc_dispatch_act<1, 64>(g, s);
}
