// Convolution weight gradient on the matrix cores:
//     dw[t][co][ci] += sum_n dy[n][co] * x[src(n, t)][ci]        (fp32, packed [tap][Cout][Cin])
// GEMM view per tap: M = Cout, N = Cin, K = output pixels n = (b, yo, xo); split-K over pixel
// ranges, partial tiles combined with fp32 atomics straight into the gradient buffer (which
// is therefore also where gradient accumulation over micro-batches happens).
//
// Both operands are channels-last, i.e. K (pixels) is the *strided* dimension, so tiles are
// staged pixel-major in LDS ([k][channel], exactly as they sit in HBM, coalesced 16-byte
// loads) and transposed on the way to the MFMA operand registers:
//   bf16: ds_read_b64_tr_b16 -- within a 16-lane group, lane o receives element e of
//         M[4e + (o>>2)][o&3] where M[i] are the 8 bytes at lane i's address (pinned by
//         tools/gpu_probe.py on gfx950).  Lane i = 4e+q points at pixel row e, channels
//         m0+4q.. so lane o ends up with 4 consecutive pixels of channel m0+o.
//   f32:  one ds_read_b32 per lane (v_mfma_f32_16x16x4_f32 takes one k per lane).
//
// Replaces autograd's convolution_backward (weight gradient) for nn.Conv2d at reference
// models.py:34-42.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "dyk_common.h"
#include "conv_wgrad_tile.h"

namespace {

// d[1] = the twin problem of a two-problem launch (DykWgradDesc.twin: x, dy, dw, part differ only); pair_blocks = workgroups
// per problem rounded up to a multiple of 8 (both halves see the same block -> XCD relation), 0 = single problem
struct WgArgs {
    DykWgradDesc d[2];
    int pair_blocks;
};
__device__ inline int wg_pick_problem(const WgArgs& args, int& blk, int& nblk) {
    blk = blockIdx.x; nblk = gridDim.x;
    if (args.pair_blocks == 0) return 0;
    const int sel = blk >= args.pair_blocks ? 1 : 0;
    blk -= sel * args.pair_blocks;
    nblk = args.pair_blocks;
    return sel;
}
inline unsigned wg_fill_args(WgArgs& args, const DykWgradDesc* d, int blocks) {
    args.d[0] = *d;
    args.d[0].twin = nullptr;
    args.pair_blocks = 0;
    if (!d->twin) return (unsigned)blocks;
    args.d[1] = *d->twin;
    args.d[1].twin = nullptr;
    args.pair_blocks = (blocks + 7) & ~7;
    return 2u * (unsigned)args.pair_blocks;
}

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_t> { static constexpr int ROWS = 64; };  // pixels per K step
template <> struct WgTraits<float>  { static constexpr int ROWS = 32; };

// PIPE: LDS-DMA ring stages (2 | 3)
// KG:   K-groups per workgroup.  The fp32 atomics of the epilogue run at ~1 element/clk/L2 channel, so their
//       count (= workgroups x tile area) bounds the kernel; with KG = 2 a 512-thread workgroup holds two 4-wave
//       groups that walk the two halves of its pixel range with private LDS rings, group 1 hands its
//       accumulators to group 0 through LDS and only group 0 issues atomics: half the workgroups (and atomics)
//       at the same number of waves per CU.
template <typename T, int BM, int BN, int PIPE, int KG>
__global__ __launch_bounds__(256 * KG) void conv_wgrad_kernel(const WgArgs args, const int splits, const int chunk) {
    int blk, nblk;
    const DykWgradDesc& a = args.d[wg_pick_problem(args, blk, nblk)];
    constexpr int ROWS = WgTraits<T>::ROWS;
    constexpr int EPV = 16 / (int)sizeof(T);
    constexpr int VPR_A = BM / EPV, VPR_B = BN / EPV;          // 16-byte vectors per tile row
    constexpr int NV_A = ROWS * VPR_A, NV_B = ROWS * VPR_B;
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int A_BYTES = ROWS * BM * (int)sizeof(T), B_BYTES = ROWS * BN * (int)sizeof(T);
    constexpr int NSTAGE = PIPE;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    char* sA = smem + grp * (NSTAGE * (A_BYTES + B_BYTES));     // [NSTAGE][A_BYTES]  dy tile (per K-group)
    char* sB = sA + NSTAGE * A_BYTES;                           // [NSTAGE][B_BYTES]  x tile

    const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    const int tiles_m = (a.Cout + BM - 1) / BM;
    const int tiles_n = (a.Cin + BN - 1) / BN;
    // consecutive remapped ids run on one XCD: the tiles and taps of one pixel range share that XCD's L2, so dy / x
    // of the range are fetched from HBM once instead of once per tap and tile (measured 3.2x over-fetch without)
    int bid = xcd_remap(blk, nblk);
    if (bid >= tiles_m * tiles_n * a.ntaps * splits) return;     // padding blocks of a two-problem launch
    const int tm = bid % tiles_m; bid /= tiles_m;
    const int tn = bid % tiles_n; bid /= tiles_n;
    const int tap = __builtin_amdgcn_readfirstlane(bid % a.ntaps);   // (provably uniform: the tap offsets below come by scalar load)
    const int sp = bid / a.ntaps;
    const int m0 = tm * BM, n0 = tn * BN;
    const int HWo = a.Ho * a.Wo;
    const int Ntot = a.B * HWo;
    // the workgroup owns pixels [sp*chunk, +chunk); K-group g the g-th slice of chunk/KG pixels (a multiple of ROWS)
    const int gchunk = chunk / KG;
    const int p_begin = sp * chunk + grp * gchunk;
    const int p_end = min(Ntot, p_begin + gchunk);
    if (KG == 1 && p_begin >= p_end && !a.part) return;      // (with `part` every launched split owns a plane and writes it)
    // (readfirstlane: the tap offsets come out of a dynamically indexed kernel-argument array, i.e. a VECTOR load; consumed
    // for the first time inside the K loop, that load made the compiler wait vmcnt(0) -- for every LDS-DMA in flight -- in
    // the middle of each step's staging block: the dy tile's DMAs were drained before the x tile's were issued)
    const int tdy = __builtin_amdgcn_readfirstlane((int)a.tdy[tap]), tdx = __builtin_amdgcn_readfirstlane((int)a.tdx[tap]);
    const T* __restrict__ dyg = wg_sgpr_ptr((const T*)a.dy);
    const T* __restrict__ xg = wg_sgpr_ptr((const T*)a.x);

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const char* pa, const char* pb) {
        const int i16 = lane & 15, kq = lane >> 4;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < ROWS / 32; ++kk) {
                uint4 fa[MI], fb[NI];
                const int r0 = kk * 32 + kq * 8 + (i16 >> 2);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int ch = wm * WTM + mi * 16 + 4 * (i16 & 3);
                    v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)(pa + wg_off<T, BM>(r0, ch)));
                    v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)(pa + wg_off<T, BM>(r0 + 4, ch)));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fa[mi] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int ch = wn * WTN + ni * 16 + 4 * (i16 & 3);
                    v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)(pb + wg_off<T, BN>(r0, ch)));
                    v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)(pb + wg_off<T, BN>(r0 + 4, ch)));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fb[ni] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, fa[mi]), __builtin_bit_cast(bf16x8_t, fb[ni]), acc[mi][ni], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < ROWS / 4; ++kk) {
                float fa[MI], fb[NI];
                const int r0 = kk * 4 + kq;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    fa[mi] = *(const float*)(pa + wg_off<T, BM>(r0, wm * WTM + mi * 16 + i16));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    fb[ni] = *(const float*)(pb + wg_off<T, BN>(r0, wn * WTN + ni * 16 + i16));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
            }
        }
        // the MFMA block must not sink below the loop's `s_waitcnt vmcnt(..)`: it would wait for the NEXT step's DMA before
        // computing this one (found in the ISA of the convolution kernels, round 3)
        __builtin_amdgcn_sched_barrier(0);
    };

    // bits 16.. of `tune`: ablation switches for kernel analysis (tools/gpu_probe.py wgablate), never set by the plan
    const bool abl_noatomic = (a.tune >> 16) & 1, abl_noloop = (a.tune >> 17) & 1;
    const int S = abl_noloop ? 0 : (KG > 1 ? gchunk / ROWS : (p_end - p_begin + ROWS - 1) / ROWS);   // uniform over the K-groups
    {
        // ---- LDS-DMA ring; each wave instruction fills RPI consecutive pixel rows of a tile
        constexpr int RPI_A = 64 / VPR_A, RPI_B = 64 / VPR_B;
        constexpr int NI_A = A_BYTES / 1024, NI_B = B_BYTES / 1024;         // both multiples of 4
        constexpr int NIA_W = NI_A / 4, NIB_W = NI_B / 4;
        constexpr int NPW = NIA_W + NIB_W;
        const int wv = __builtin_amdgcn_readfirstlane(wid);
        const T* zero = (const T*)dyk_wg_zero_page;
        int a_row[NIA_W], a_ch[NIA_W];
#pragma unroll
        for (int j = 0; j < NIA_W; ++j) {
            a_row[j] = (j * 4 + wv) * RPI_A + lane / VPR_A;
            a_ch[j] = wg_logical_ch<T, BM>(a_row[j], lane % VPR_A);
        }
        int b_row[NIB_W], b_ch[NIB_W], b_img[NIB_W], b_yo[NIB_W], b_xo[NIB_W];
#pragma unroll
        for (int j = 0; j < NIB_W; ++j) {
            b_row[j] = (j * 4 + wv) * RPI_B + lane / VPR_B;
            b_ch[j] = wg_logical_ch<T, BN>(b_row[j], lane % VPR_B);
            const int n = p_begin + b_row[j];
            const int b = n / HWo, r = n - b * HWo;
            b_img[j] = b; b_yo[j] = r / a.Wo; b_xo[j] = r - b_yo[j] * a.Wo;
        }
        int sp0 = p_begin;                       // first pixel of the next step to stage
        // descriptor scalars in SGPRs for the whole loop: read through `a` they are re-loaded from the kernel arguments behind
        // every LDS-DMA statement (its "memory" clobber), with a wait each, and the address selects turn into exec-mask branches
        const int Cout_s = __builtin_amdgcn_readfirstlane(a.Cout), Cin_s = __builtin_amdgcn_readfirstlane(a.Cin);
        const int lddy_s = __builtin_amdgcn_readfirstlane(a.lddy), ldx_s = __builtin_amdgcn_readfirstlane(a.ldx);
        const int isy_s = __builtin_amdgcn_readfirstlane(a.isy), isx_s = __builtin_amdgcn_readfirstlane(a.isx);
        const int Hi_s = __builtin_amdgcn_readfirstlane(a.Hi), Wi_s = __builtin_amdgcn_readfirstlane(a.Wi);
        const int Ho_s = __builtin_amdgcn_readfirstlane(a.Ho), Wo_s = __builtin_amdgcn_readfirstlane(a.Wo);
        const int adv_q = ROWS / Wo_s, adv_r = ROWS - adv_q * Wo_s;
        const bool adv_fast = adv_q + 1 <= 2 * Ho_s;
        auto stage_next = [&](int buf) {
            char* da = sA + buf * A_BYTES;
            char* db = sB + buf * B_BYTES;
#pragma unroll
            for (int j = 0; j < NIA_W; ++j) {
                const int n = sp0 + a_row[j];
                const int c = m0 + a_ch[j];
                const bool ok = (n < p_end) & (c < Cout_s);          // (& not &&: straight-line address selects, no exec-mask branches)
                const T* cand = dyg + (long)n * lddy_s + c;
                const T* src = ok ? cand : zero;
                wg_glds16(src, wg_lds_addr(da + (j * 4 + wv) * 1024));
            }
#pragma unroll
            for (int j = 0; j < NIB_W; ++j) {
                const int n = sp0 + b_row[j];
                const int c = n0 + b_ch[j];
                const int yi = b_yo[j] * isy_s + tdy, xi = b_xo[j] * isx_s + tdx;
                const bool ok = (n < p_end) & (c < Cin_s) & ((unsigned)yi < (unsigned)Hi_s) & ((unsigned)xi < (unsigned)Wi_s);
                const T* cand = xg + ((long)(b_img[j] * Hi_s + yi) * Wi_s + xi) * ldx_s + c;
                const T* src = ok ? cand : zero;
                wg_glds16(src, wg_lds_addr(db + (j * 4 + wv) * 1024));
            }
            // advance the per-lane (image, row, column) by ROWS pixels.  Straight-line code on every real map (two
            // conditional subtractions cover adv_q + 1 <= 2 * Ho); per-lane wrap loops -- exec-mask branches that cost
            // more than the MFMA block of a K step -- only on maps of a few pixels
            if (adv_fast) {
#pragma unroll
                for (int j = 0; j < NIB_W; ++j) {
                    int xo = b_xo[j] + adv_r, yo = b_yo[j] + adv_q, bb = b_img[j];
                    const bool cx = xo >= Wo_s;
                    xo -= cx ? Wo_s : 0; yo += cx ? 1 : 0;
                    const bool c1 = yo >= Ho_s;
                    yo -= c1 ? Ho_s : 0; bb += c1 ? 1 : 0;
                    const bool c2 = yo >= Ho_s;
                    yo -= c2 ? Ho_s : 0; bb += c2 ? 1 : 0;
                    b_xo[j] = xo; b_yo[j] = yo; b_img[j] = bb;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NIB_W; ++j) {
                    int xo = b_xo[j] + ROWS, yo = b_yo[j], bb = b_img[j];
                    while (xo >= Wo_s) { xo -= Wo_s; ++yo; }
                    while (yo >= Ho_s) { yo -= Ho_s; ++bb; }
                    b_xo[j] = xo; b_yo[j] = yo; b_img[j] = bb;
                }
            }
            sp0 += ROWS;
        };
        if constexpr (PIPE >= 3) {
            // N-stage ring: AHEAD = N-1 steps in flight while one is computed.  (A 4-stage instantiation -- 3 x 32 KB in flight
            // per workgroup, meant for the HBM-bound 1x1 gradients that run on a few dozen workgroups -- was offered to the tuner
            // in round 3 and won for NO problem of the target cfg: 1.02-1.15x the 2- / 3-stage time everywhere.  Not instantiated.)
            constexpr int AHEAD = PIPE - 1;
            constexpr int KEEP = (AHEAD - 1) * NPW;        // DMA instructions that may still be in flight per wave
            static_assert(KEEP <= 63, "vmcnt range");
#pragma unroll
            for (int i = 0; i < AHEAD; ++i)
                if (S > i) stage_next(i);
            if (S >= AHEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int cur = 0, nxt = AHEAD;
            for (int s = 0; s < S; ++s) {
                const bool more = (s + AHEAD < S);
                if (more) stage_next(nxt);
                compute(sA + cur * A_BYTES, sB + cur * B_BYTES);
                if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                cur = (cur == PIPE - 1) ? 0 : cur + 1;
                nxt = (nxt == PIPE - 1) ? 0 : nxt + 1;
            }
        } else {
            if (S > 0) stage_next(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) stage_next((s + 1) & 1);
                compute(sA + (s & 1) * A_BYTES, sB + (s & 1) * B_BYTES);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }

    if constexpr (KG > 1) {
        // fold the K-groups: group g > 0 parks its accumulators in LDS (lane-linear float4, conflict free), group 0 adds
        float4* park = (float4*)smem;            // overlays the rings (all waves are behind the loop's last barrier)
        for (int g = 1; g < KG; ++g) {
            if (grp == g) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        park[((mi * NI + ni) * 4 + wid) * 64 + lane] = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const float4 v = park[((mi * NI + ni) * 4 + wid) * 64 + lane];
                        acc[mi][ni][0] += v.x; acc[mi][ni][1] += v.y; acc[mi][ni][2] += v.z; acc[mi][ni][3] += v.w;
                    }
            }
            if (g + 1 < KG) __syncthreads();
        }
        if (grp != 0) return;
    }
    // ---- epilogue: acc[r] = D[co = (lane>>4)*4 + r][ci = lane&15]
    const int lddw = a.lddw > 0 ? a.lddw : a.Cin;
    float* dw = a.dw + (long)a.twt[tap] * a.Cout * lddw;
    if (abl_noatomic) {
        float sum = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) sum += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
        if (sum == 123.456f) dw[0] = sum;
        return;
    }
    if (a.part) {
        // atomic-free mode: this split's plane gets the tile with plain stores (16 lanes = one 64-byte row segment);
        // dyk_grad_reduce folds the planes.  The fp32 atomics retire at ~1 lane per clock per L2 channel and were
        // 37 % of this kernel's time; the stores run at HBM speed and the result is bit-reproducible.
        float* pw = a.part + (long)sp * a.part_stride + (long)a.twt[tap] * a.Cout * lddw;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + wm * WTM + mi * 16 + (lane >> 4) * 4 + r;
                if (co >= a.Cout) continue;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int ci = n0 + wn * WTN + ni * 16 + (lane & 15);
                    if (ci < a.Cin) pw[(long)co * lddw + ci] = acc[mi][ni][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m0 + wm * WTM + mi * 16 + (lane >> 4) * 4 + r;
            if (co >= a.Cout) continue;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int ci = n0 + wn * WTN + ni * 16 + (lane & 15);
                if (ci < a.Cin) unsafeAtomicAdd(dw + (long)co * lddw + ci, acc[mi][ni][r]);
            }
        }
    }
}

// query != NULL: only report the split count this launch would use
template <typename T, int BM, int BN, int PIPE, int KG>
int launch_wgrad_impl(const DykWgradDesc* d, hipStream_t stream, int* query) {
    constexpr int ROWS = WgTraits<T>::ROWS;
    constexpr size_t ring = KG * PIPE * (size_t)ROWS * (BM + BN) * sizeof(T);
    constexpr size_t park = KG > 1 ? (size_t)BM * BN * 4 : 0;
    constexpr size_t lds = ring > park ? ring : park;
    static_assert(lds <= 160 * 1024, "LDS budget");
    const long Ntot = (long)d->B * d->Ho * d->Wo;
    const int tiles = dyk_div_up(d->Cout, BM) * dyk_div_up(d->Cin, BN) * d->ntaps;
    const int ksteps = dyk_div_up(Ntot, ROWS);
    int splits = d->splits;
    if (splits <= 0) {
        // 4-wave workgroups: ~3 per CU; K-grouped (8-wave) workgroups occupy a CU alone (LDS) unless the tile is small
        const int target = KG > 1 ? (lds > 80 * 1024 ? 256 : 512) : 768;
        splits = dyk_div_up(target, tiles);
        const int max_splits = ksteps / (8 * KG) > 0 ? ksteps / (8 * KG) : 1;   // at least 8 K steps per K-group (amortise the atomics)
        if (splits > max_splits) splits = max_splits;
    }
    if (splits * KG > ksteps) splits = ksteps / KG > 0 ? ksteps / KG : 1;
    const int chunk = dyk_div_up(ksteps, splits * KG) * ROWS * KG;      // pixels per workgroup: KG slices of whole K steps
    if (!(d->part && d->splits > 0)) splits = dyk_div_up(Ntot, chunk);  // (plane mode with a given count: exactly that many
                                                                         //  planes are written, trailing empty ones with zeros)
    if (query) { *query = splits; return DYK_OK; }
    static DykDeviceOnce attr_set;   // (the attribute is per DEVICE: one flag per device id; after the query path: no device there)
    auto kfn = conv_wgrad_kernel<T, BM, BN, PIPE, KG>;
    if (attr_set.first()) {
        DYK_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    WgArgs args;
    const unsigned grid = wg_fill_args(args, d, tiles * splits);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256 * KG), lds, stream, args, splits, chunk);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

template <typename T, int BM, int BN>
int launch_wgrad(const DykWgradDesc* d, hipStream_t stream, int* query) {
    // tune: low byte = LDS ring stages (2 | 3), bits 8..15 = K-groups per workgroup (1 | 2)
    const int kg = (d->tune >> 8) & 0xff;
    if ((d->tune & 0xff) == 3) return launch_wgrad_impl<T, BM, BN, 3, 1>(d, stream, query);
    if (kg == 2) return launch_wgrad_impl<T, BM, BN, 2, 2>(d, stream, query);
    return launch_wgrad_impl<T, BM, BN, 2, 1>(d, stream, query);
}

template <typename T, int BM>
int dispatch_wgrad_n(const DykWgradDesc* d, hipStream_t s, int* query) {
    const bool cap64 = ((d->tune >> 24) & 0xf) == 1;        // bits 24..27 = 1: tiles capped at 64 x 64 (small GEMMs: more
                                                            // tiles, so fewer K splits -- partial planes -- fill the chip)
    if (d->Cin > 64 && !cap64) return launch_wgrad<T, BM, 128>(d, s, query);
    if (d->Cin > 32) return launch_wgrad<T, BM, 64>(d, s, query);
    return launch_wgrad<T, BM, 32>(d, s, query);
}
template <typename T>
int dispatch_wgrad(const DykWgradDesc* d, hipStream_t s, int* query) {
    const bool cap64 = ((d->tune >> 24) & 0xf) == 1;
    if (d->Cout > 64 && !cap64) return dispatch_wgrad_n<T, 128>(d, s, query);
    if (d->Cout > 32) return dispatch_wgrad_n<T, 64>(d, s, query);
    return dispatch_wgrad_n<T, 32>(d, s, query);
}

// ======================================================================================
// Multi-tap weight gradient for 3x3 convs with few channels (the early layers: 32..128 channels on 256x320 / 128x160
// maps).  The per-tap kernel above re-stages dy and x for each of the nine taps, which makes those layers L2-bandwidth
// bound (9 x 250 MB per launch through L2 at 256x320).  Here one K step is a segment of KW consecutive output pixels
// of one output row: its dy tile [KW][64 co] and the x tile WITH HALO [3 input rows][(KW-1)*SI + 3 pixels][BN ci] are
// staged once, and the nine taps are row-shifted (stride-SI) views of the halo tile -- each lane of the transposing
// LDS read supplies its own address, so a shifted / strided pixel run costs nothing extra.  The dy fragments are read
// once per step and feed all nine taps; a workgroup holds the nine [64 x BN] accumulator tiles (72 / 144 VGPRs).
// bf16 only; taps must be the standard 3x3 / pad 1 table (tdy = t/3 - 1, tdx = t%3 - 1); Wo % KW == 0.
template <int BN, int SI, int KW>
__global__ __launch_bounds__(256) void conv_wgrad_mt_kernel(const WgArgs args, const int splits, const int chunk) {
    int blk, nblk;
    const DykWgradDesc& a = args.d[wg_pick_problem(args, blk, nblk)];
    using T = bf16_t;
    constexpr int BM = 64;
    constexpr int XW = (KW - 1) * SI + 3;                      // halo pixels per input row
    constexpr int HROWS = 3 * XW;
    constexpr int VPR_A = BM / 8, VPR_B = BN / 8;              // 16-byte vectors per tile row
    constexpr int RPI_A = 64 / VPR_A, RPI_B = 64 / VPR_B;      // tile rows per DMA wave instruction
    constexpr int NI_A = KW / RPI_A;                           // 8 (KW 64) | 4 (KW 32): multiples of 4
    constexpr int NI_B = ((HROWS + RPI_B - 1) / RPI_B + 3) / 4 * 4;
    constexpr int NIA_W = NI_A / 4, NIB_W = NI_B / 4;
    constexpr int A_BYTES = NI_A * 1024, B_BYTES = NI_B * 1024;
    constexpr int WTN = BN / 2;                                // 2 x 2 waves over (64 co, BN ci)
    constexpr int MI = 2, NI = WTN / 16;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                           // [2][A_BYTES]  dy tile  [KW][64]
    char* sB = smem + 2 * A_BYTES;                             // [2][B_BYTES]  x halo tile [3 * XW (+pad)][BN]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_m = (a.Cout + BM - 1) / BM;
    const int tiles_n = (a.Cin + BN - 1) / BN;
    int bid = xcd_remap(blk, nblk);
    if (bid >= tiles_m * tiles_n * splits) return;               // padding blocks of a two-problem launch
    const int tm = bid % tiles_m; bid /= tiles_m;
    const int tn = bid % tiles_n;
    const int sp = bid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int SPR = (a.Wo + KW - 1) / KW;                      // segments per output row (the last one may be ragged: its
                                                               // missing dy rows come from the zero page and contribute 0)
    const int nseg = a.B * a.Ho * SPR;
    const int g_begin = sp * chunk;
    const int g_end = min(nseg, g_begin + chunk);
    const int S = g_end > g_begin ? g_end - g_begin : 0;
    const T* __restrict__ dyg = wg_sgpr_ptr((const T*)a.dy);
    const T* __restrict__ xg = wg_sgpr_ptr((const T*)a.x);

    f32x4_t acc[9][MI][NI];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[t][mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    auto tr_read = [&](const char* p0, const char* p1) -> uint4 {
        v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)p0);
        v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)p1);
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    auto compute = [&](const char* pa, const char* pb) {
        const int i16 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < KW / 32; ++kk) {
            const int r0 = kk * 32 + kq * 8 + (i16 >> 2);     // this lane's K pixel (and r0 + 4)
            uint4 fa[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int ch = wm * 32 + mi * 16 + 4 * (i16 & 3);
                fa[mi] = tr_read(pa + wg_off<T, BM>(r0, ch), pa + wg_off<T, BM>(r0 + 4, ch));
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int h0 = (t / 3) * XW + r0 * SI + (t % 3);      // halo row of K pixel r0 under tap t
                uint4 fb[NI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int ch = wn * WTN + ni * 16 + 4 * (i16 & 3);
                    fb[ni] = tr_read(pb + wg_off<T, BN>(h0, ch), pb + wg_off<T, BN>(h0 + 4 * SI, ch));
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[t][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8_t, fa[mi]), __builtin_bit_cast(bf16x8_t, fb[ni]), acc[t][mi][ni], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    {
        const int wv = __builtin_amdgcn_readfirstlane(wid);
        const T* zero = (const T*)dyk_wg_zero_page;
        int a_row[NIA_W], a_ch[NIA_W];
#pragma unroll
        for (int j = 0; j < NIA_W; ++j) {
            a_row[j] = (j * 4 + wv) * RPI_A + lane / VPR_A;
            a_ch[j] = m0 + wg_logical_ch<T, BM>(a_row[j], lane % VPR_A);
        }
        int b_rr[NIB_W], b_jj[NIB_W], b_ch[NIB_W];
#pragma unroll
        for (int j = 0; j < NIB_W; ++j) {
            const int h = (j * 4 + wv) * RPI_B + lane / VPR_B;
            b_rr[j] = h < HROWS ? h / XW : -100000;            // beyond the halo tile: always out of bounds -> zero page
            b_jj[j] = h % XW;
            b_ch[j] = n0 + wg_logical_ch<T, BN>(h, lane % VPR_B);
        }
        int seg = g_begin;
        const int Cout_s = __builtin_amdgcn_readfirstlane(a.Cout), Cin_s = __builtin_amdgcn_readfirstlane(a.Cin);
        const int lddy_s = __builtin_amdgcn_readfirstlane(a.lddy), ldx_s = __builtin_amdgcn_readfirstlane(a.ldx);
        const int Hi_s = __builtin_amdgcn_readfirstlane(a.Hi), Wi_s = __builtin_amdgcn_readfirstlane(a.Wi);
        const int Ho_s = __builtin_amdgcn_readfirstlane(a.Ho), Wo_s = __builtin_amdgcn_readfirstlane(a.Wo);
        const int SPR_s = __builtin_amdgcn_readfirstlane(SPR);
        auto stage_next = [&](int buf) {
            const int b = seg / (Ho_s * SPR_s);
            const int r = seg - b * (Ho_s * SPR_s);
            const int yo = r / SPR_s, xo0 = (r - yo * SPR_s) * KW;
            char* da = sA + buf * A_BYTES;
            char* db = sB + buf * B_BYTES;
            const long nbase = ((long)b * Ho_s + yo) * Wo_s + xo0;
#pragma unroll
            for (int j = 0; j < NIA_W; ++j) {
                const bool ok = (a_ch[j] < Cout_s) & (xo0 + a_row[j] < Wo_s);
                const T* cand = dyg + (nbase + a_row[j]) * lddy_s + a_ch[j];
                const T* src = ok ? cand : zero;
                wg_glds16(src, wg_lds_addr(da + (j * 4 + wv) * 1024));
            }
#pragma unroll
            for (int j = 0; j < NIB_W; ++j) {
                const int yi = yo * SI + b_rr[j] - 1, xi = xo0 * SI + b_jj[j] - 1;
                const bool ok = (b_ch[j] < Cin_s) & ((unsigned)yi < (unsigned)Hi_s) & ((unsigned)xi < (unsigned)Wi_s);
                const T* cand = xg + ((long)(b * Hi_s + yi) * Wi_s + xi) * ldx_s + b_ch[j];
                const T* src = ok ? cand : zero;
                wg_glds16(src, wg_lds_addr(db + (j * 4 + wv) * 1024));
            }
            ++seg;
        };
        if (S > 0) stage_next(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < S; ++s) {
            if (s + 1 < S) stage_next((s + 1) & 1);
            compute(sA + (s & 1) * A_BYTES, sB + (s & 1) * B_BYTES);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- epilogue: acc[t][mi][ni][r] = D_t[co = m0 + wm*32 + mi*16 + (lane>>4)*4 + r][ci = n0 + wn*WTN + ni*16 + (lane&15)]
    const int lddw = a.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const long toff = (long)a.twt[t] * a.Cout * lddw;
        float* dw = a.part ? a.part + (long)sp * a.part_stride + toff : a.dw + toff;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + wm * 32 + mi * 16 + (lane >> 4) * 4 + r;
                if (co >= a.Cout) continue;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int ci = n0 + wn * WTN + ni * 16 + (lane & 15);
                    if (ci >= a.Cin) continue;
                    if (a.part) dw[(long)co * lddw + ci] = acc[t][mi][ni][r];
                    else unsafeAtomicAdd(dw + (long)co * lddw + ci, acc[t][mi][ni][r]);
                }
            }
        }
    }
}

// the multi-tap kernel covers: bf16, the standard 3x3 / pad 1 tap table, stride 1 | 2, Wo a multiple of 32, whole 8-channel
// vectors, plain [tap][Cout][Cin] gradient rows
bool mt_eligible(const DykWgradDesc* d) {
    if (d->dtype != DYK_BF16 || d->ntaps != 9 || d->isy != d->isx || (d->isy != 1 && d->isy != 2)) return false;
    // (Wo need not be a multiple of the 32-pixel segment: ragged last segments are zero filled -- 80-, 40- and 20-pixel rows
    // of the deep stages waste 17 / 37 / 37 % of the MFMA work of a row but stage dy and x once for all nine taps)
    constexpr bool ragged = true;
    if (d->Wo < 16 || (!ragged && d->Wo % 32) || d->Cin % 8 || d->Cout % 8 || (d->lddw > 0 && d->lddw != d->Cin)) return false;
    for (int t = 0; t < 9; ++t)
        if (d->tdy[t] != t / 3 - 1 || d->tdx[t] != t % 3 - 1) return false;
    return true;
}

template <int BN, int SI, int KW>
int launch_wgrad_mt(const DykWgradDesc* d, hipStream_t stream, int* query) {
    constexpr int XW = (KW - 1) * SI + 3, HROWS = 3 * XW, RPI_B = 64 / (BN / 8);
    constexpr int NI_B = ((HROWS + RPI_B - 1) / RPI_B + 3) / 4 * 4;
    constexpr size_t lds = 2 * ((size_t)(KW / 8) * 1024 + (size_t)NI_B * 1024);
    static_assert(lds <= 160 * 1024, "LDS budget");
    const int tiles = dyk_div_up(d->Cout, 64) * dyk_div_up(d->Cin, BN);
    const int nseg = d->B * d->Ho * ((d->Wo + KW - 1) / KW);
    int splits = d->splits;
    if (splits <= 0) {
        splits = dyk_div_up(512, tiles);
        const int max_splits = nseg / 16 > 0 ? nseg / 16 : 1;          // at least 16 segments per workgroup
        if (splits > max_splits) splits = max_splits;
    }
    if (splits > nseg) splits = nseg;
    const int chunk = dyk_div_up(nseg, splits);
    if (!(d->part && d->splits > 0)) splits = dyk_div_up(nseg, chunk);
    if (query) { *query = splits; return DYK_OK; }
    static DykDeviceOnce attr_set;   // (the attribute is per DEVICE: one flag per device id; after the query path: no device there)
    auto kfn = conv_wgrad_mt_kernel<BN, SI, KW>;
    if (attr_set.first()) {
        DYK_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    WgArgs args;
    const unsigned grid = wg_fill_args(args, d, tiles * splits);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, args, splits, chunk);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

int dispatch_wgrad_mt(const DykWgradDesc* d, hipStream_t s, int* query) {
    const bool wide = d->Cin > 32;
    if (d->Wo % 64 == 0 && d->isy == 1) return wide ? launch_wgrad_mt<64, 1, 64>(d, s, query) : launch_wgrad_mt<32, 1, 64>(d, s, query);
    if (d->isy == 1) return wide ? launch_wgrad_mt<64, 1, 32>(d, s, query) : launch_wgrad_mt<32, 1, 32>(d, s, query);
    if (d->Wo % 64 == 0 && !wide) return launch_wgrad_mt<32, 2, 64>(d, s, query);
    return wide ? launch_wgrad_mt<64, 2, 32>(d, s, query) : launch_wgrad_mt<32, 2, 32>(d, s, query);
}

// one block per 1024-element chunk of one table entry (binary search over chunk_begin, as dyk_transpose_taps)
__global__ __launch_bounds__(256) void grad_reduce_kernel(float* __restrict__ G, const float* __restrict__ part,
                                                          const DykGradReduceEntry* __restrict__ tab, int n_entries) {
    const int cidx = blockIdx.x;
    int lo = 0, hi = n_entries - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].chunk_begin <= cidx) lo = mid; else hi = mid - 1;
    }
    const DykGradReduceEntry e = tab[lo];
    const long i = ((long)(cidx - e.chunk_begin) * 256 + threadIdx.x) * 4;
    if (i >= e.n) return;
    const float* p = part + e.part_off + i;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= e.splits; s += 8) {                 // eight planes in flight (early layers fold up to 512 planes: the
        float4 v[8];                                    //  four-at-a-time loop was a chain of 128 dependent round trips)
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(p + (long)(s + u) * e.plane);
        acc.x += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
        acc.y += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
        acc.z += ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
        acc.w += ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
    }
    for (; s + 4 <= e.splits; s += 4) {                 // four planes in flight
        const float4 v0 = *(const float4*)(p + (long)s * e.plane), v1 = *(const float4*)(p + (long)(s + 1) * e.plane);
        const float4 v2 = *(const float4*)(p + (long)(s + 2) * e.plane), v3 = *(const float4*)(p + (long)(s + 3) * e.plane);
        acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
        acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < e.splits; ++s) {
        const float4 v = *(const float4*)(p + (long)s * e.plane);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4* g = (float4*)(G + e.g_off + i);
    float4 o = *g;
    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
    *g = o;
}

}  // namespace

// row-block 3x3 kernel (conv_wgrad_rb.hip): tune bits 28..30 == 2; bits 8..15 == 2 selects 256-pixel K steps
bool dyk_wgrad_rb_eligible(const DykWgradDesc* d);
int dyk_wgrad_rb_dispatch(const DykWgradDesc* d, hipStream_t s, int* query);
// pixel-streaming 1x1 kernel (conv_wgrad_ps.hip): tune bits 28..30 == 3
bool dyk_wgrad_ps_eligible(const DykWgradDesc* d);
int dyk_wgrad_ps_dispatch(const DykWgradDesc* d, hipStream_t s, int* query_splits, int* query_tiles, int64_t* query_slab);

static int wgrad_validate(const DykWgradDesc* d) {
    if (!d || !d->x || !d->dy || !d->dw) return DYK_ERR_ARG;
    if (d->ntaps <= 0 || d->ntaps > DYK_MAX_TAPS) return DYK_ERR_ARG;
    if (d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0 || d->Cin <= 0) return DYK_ERR_ARG;
    const int epv = d->dtype == DYK_BF16 ? 8 : 4;
    // every 16-byte load must stay inside its pixel row: ld >= round_up(C, epv)
    if (d->ldx % epv || d->lddy % epv) return DYK_ERR_ARG;
    if (d->ldx < (d->Cin + epv - 1) / epv * epv || d->lddy < (d->Cout + epv - 1) / epv * epv) return DYK_ERR_ARG;
    if (((uintptr_t)d->x % 16) || ((uintptr_t)d->dy % 16)) return DYK_ERR_ARG;
    if ((long)d->B * d->Ho * d->Wo >= (1L << 31)) return DYK_ERR_ARG;
    if (d->part && (d->part_stride < (int64_t)d->Cout * (d->lddw > 0 ? d->lddw : d->Cin))) return DYK_ERR_ARG;
    return DYK_OK;
}

extern "C" int dyk_conv_wgrad(const DykWgradDesc* d, void* stream) {
    int rc0 = wgrad_validate(d);
    if (rc0 != DYK_OK) return rc0;
    if (d->twin) {
        // two-problem launch: equal in every non-pointer field (part_stride included), both with or both without planes
        if ((rc0 = wgrad_validate(d->twin)) != DYK_OK) return rc0;
        const size_t lo = offsetof(DykWgradDesc, part_stride), hi = offsetof(DykWgradDesc, twin);
        if (memcmp((const char*)d + lo, (const char*)d->twin + lo, hi - lo) != 0 || (!d->part) != (!d->twin->part)) return DYK_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (((d->tune >> 28) & 7) == 1 && mt_eligible(d)) return dispatch_wgrad_mt(d, s, nullptr);    // multi-tap 3x3 variant
    if (((d->tune >> 28) & 7) == 2 && dyk_wgrad_rb_eligible(d)) return dyk_wgrad_rb_dispatch(d, s, nullptr);   // row-block 3x3 variant
    if (((d->tune >> 28) & 7) == 3 && dyk_wgrad_ps_eligible(d)) return dyk_wgrad_ps_dispatch(d, s, nullptr, nullptr, nullptr);   // pixel-streaming 1x1 variant
    if (d->sk_cnt && !d->part) return DYK_ERR_UNSUPPORTED;      // the in-launch fold exists in the pixel-streaming kernel only
    if (d->group_n != 0) return DYK_ERR_UNSUPPORTED;            // grouped launches: pixel-streaming and row-block kernels only
    if (d->dtype == DYK_BF16) return dispatch_wgrad<bf16_t>(d, s, nullptr);
    if (d->dtype == DYK_F32) return dispatch_wgrad<float>(d, s, nullptr);
    return DYK_ERR_ARG;
}

extern "C" int dyk_conv_wgrad_splits(const DykWgradDesc* d) {
    if (!d || d->ntaps <= 0 || d->ntaps > DYK_MAX_TAPS || d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0 || d->Cin <= 0)
        return DYK_ERR_ARG;
    int q = 0;
    int rc = DYK_ERR_ARG;
    if (((d->tune >> 28) & 7) == 1 && mt_eligible(d)) rc = dispatch_wgrad_mt(d, nullptr, &q);
    else if (((d->tune >> 28) & 7) == 2 && dyk_wgrad_rb_eligible(d)) rc = dyk_wgrad_rb_dispatch(d, nullptr, &q);
    else if (((d->tune >> 28) & 7) == 3 && dyk_wgrad_ps_eligible(d)) rc = dyk_wgrad_ps_dispatch(d, nullptr, &q, nullptr, nullptr);
    else if (d->dtype == DYK_BF16) rc = dispatch_wgrad<bf16_t>(d, nullptr, &q);
    else if (d->dtype == DYK_F32) rc = dispatch_wgrad<float>(d, nullptr, &q);
    return rc == DYK_OK ? q : rc;
}

extern "C" int dyk_conv_wgrad_variant(const DykWgradDesc* d) {
    if (!d || d->ntaps <= 0 || d->ntaps > DYK_MAX_TAPS) return DYK_ERR_ARG;
    if (((d->tune >> 28) & 7) == 1 && mt_eligible(d)) return 1;
    if (((d->tune >> 28) & 7) == 2 && dyk_wgrad_rb_eligible(d)) return 2;
    if (((d->tune >> 28) & 7) == 3 && dyk_wgrad_ps_eligible(d)) return 3;
    return 0;
}

extern "C" int64_t dyk_conv_wgrad_fold_ws_bytes(const DykWgradDesc* d, int32_t* tiles) {
    if (!d || d->ntaps <= 0 || d->ntaps > DYK_MAX_TAPS || d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0 || d->Cin <= 0)
        return DYK_ERR_ARG;
    if (tiles) *tiles = 0;
    if (d->splits < 2) return 0;
    int sp = 0, nt = 0;
    int64_t slab = 0;
    if (((d->tune >> 28) & 7) == 3 && dyk_wgrad_ps_eligible(d)) {
        DykWgradDesc q = *d;                      // (the count as the fold mode will see it: `splits` taken literally)
        q.part = nullptr;
        uint32_t dummy = 0;
        q.sk_cnt = &dummy;
        if (dyk_wgrad_ps_dispatch(&q, nullptr, &sp, &nt, &slab) != DYK_OK) return DYK_ERR_ARG;
    } else {
        return 0;                                 // kernels without the in-launch fold
    }
    if (sp < 2) return 0;
    if (tiles) *tiles = nt;
    return (int64_t)nt * sp * slab;
}

extern "C" int dyk_grad_reduce(float* G, const float* part, const DykGradReduceEntry* tab, int32_t n_entries,
                               int32_t total_chunks, void* stream) {
    if (!G || !part || !tab || n_entries <= 0 || total_chunks <= 0) return DYK_ERR_ARG;
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream, G, part, tab, n_entries);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
