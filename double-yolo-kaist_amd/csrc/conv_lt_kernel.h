// Large-tile 3x3 convolution (forward and stride-1 data gradient) for the CDNA4 matrix cores: ONE 512-thread workgroup per CU.
//
// Why a second kernel family: the generic implicit-GEMM tiles (conv_igemm_kernel.h) top out at 128 x 160 with four
// waves, re-stage the activation tile for each of the nine taps and expose every K step's LDS / DMA latency when a CU holds a
// single workgroup (the 32x40 / 16x20 stages).  Here:
//   * eight waves, every wave owns a 64-channel x 80-pixel accumulator tile (4 x 5 MFMA tiles of v_mfma_f32_16x16x32_bf16);
//     the eight waves form KG K-groups of WMn x WNn waves:  <2,4,1> = 128 x 320 output tile, <4,2,1> = 256 x 160,
//     <2,2,2> = 128 x 160 with the input channels split over two groups (the 32x40 stage: 256 workgroups of full-size wave
//     tiles need the K split; the groups fold their accumulators through LDS in a fixed order -- reproducible, no split-K
//     pass through memory).  A four-group 128 x 80 form for the 16x20 stage does not fit: 4 x (weight ring + halo pair) > 160 KB;
//   * the activation operand is a HALO patch: TH x TW pixels of one image plus a one-pixel border, staged ONCE per 32-channel
//     chunk by LDS-DMA (double buffered: the next chunk arrives piecewise during tap steps 0..6) -- the nine taps are row-shifted
//     views of it.  Only the weight tile (BM rows x 64 B) streams per step: 1-4 DMA instructions per wave and step instead of 9;
//   * K step = (tap, 32 channels): 20 MFMAs per wave.  Weight ring of five stages, DMA four steps ahead with a COUNTED
//     s_waitcnt vmcnt (never 0 in the steady state), raw s_barrier, and the fragments of step s+1 are read from LDS while the
//     MFMAs of step s run (register double buffer); every load / DMA instruction of a step sits between two groups of five
//     MFMAs (sched_barrier-pinned), so the matrix pipe always has work queued while a wave issues them.
// Measured (tools/lt_probe.py, B = 16, warm caches, every launch alone): the K loop of 128 -> 128 @64x80 runs at ~1.3 PF in
// BOTH kernel families (18.6 us here, 19.8 us in the generic 128 x 160 tile; launch + prologue 3.7 us, statistics epilogue
// 5.6 us, BatchNorm-backward epilogue 15 us) -- the loop is at the chip's power-limited rate (cdna_hip_programming.md: 1.32-1.47 PF for
// the 256^2 8-phase GEMM on random data); what the large tiles buy is 5-8 % on the layers with Cin != Cout.
// Same descriptor (DykConvDesc) and the same epilogues (conv_epilogue: statistics, affine / activation / residual, fused
// BatchNorm-backward) as the generic kernels.  bf16, ntaps == 9 with offsets in [-1, 1]^2, unit strides, Cin % 32 == 0,
// H % TH == 0, W % TW == 0.
//
// Replaces: nn.Conv2d(k=3, s=1, p=1) forward at reference models.py:34-42 and autograd's input gradient of the same layers.
#pragma once
#include "conv_igemm_kernel.h"

namespace {

constexpr int LT_ROWB = 64;                 // bytes per LDS row = 32 bf16 channels per K step
// LAG = steps whose DMA may still be in flight when a step ends (counted vmcnt).  Weight tiles are issued LAG + 2 steps ahead
// into a ring of LAG + 3 stages; the halo of the next chunk is issued during tap steps 0 .. 7 - LAG of the current one.
constexpr int LT_LAG = 2;
constexpr int LT_NA = LT_LAG + 3, LT_AHEAD = LT_LAG + 2, LT_HSMAX = 8 - LT_LAG;

struct LtGeom { int TH, TW, HW, HR, NB; };  // patch rows / cols, halo row width, halo rows, DMA instructions per halo chunk
inline LtGeom lt_geom(int TH, int TW) {
    LtGeom g;
    g.TH = TH; g.TW = TW; g.HW = TW + 2; g.HR = (TH + 2) * (TW + 2); g.NB = (g.HR + 15) / 16;
    return g;
}
// LDS layout: [t_out BN][t_res BN] ints | sink 1024 | s_stat 8 x 2 x BM floats | (1 KiB aligned) per K-group: weight ring
//             NA x BM x 64 B, halo 2 x NB x 1024 B
template <int BM, int BN> constexpr int lt_table_bytes() { return BN * 8 + 1024 + 8 * 2 * BM * 4; }
template <int BM, int BN> constexpr int lt_rings_off() { return (lt_table_bytes<BM, BN>() + 1023) & ~1023; }

template <int WMn, int WNn, int KG, int EPIK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_lt_kernel(const ConvArgs args, const int TH, const int TW) {
    static_assert(WMn * WNn * KG == 8, "eight waves");
    using T = bf16_t;
    constexpr int BM = 64 * WMn, BN = 80 * WNn;
    constexpr int GW = WMn * WNn;                  // waves per K-group
    constexpr int GT = 64 * GW;                    // threads per K-group (= the threads that run the epilogue)
    constexpr int MI = 4, NI = 5;
    constexpr int A_BYTES = BM * LT_ROWB;
    constexpr int NI_A = A_BYTES / 1024;           // weight DMA instructions per step and group
    constexpr int NAW = NI_A / GW;                 // ... per wave
    static_assert(NI_A % GW == 0, "uniform weight DMA count per wave");
    constexpr int NPW = NAW + 1;                   // DMA instructions per wave and step: weights + one halo piece (or a dummy)
    static_assert(LT_LAG * NPW <= 63, "vmcnt range");
    constexpr int TABLE_BYTES = lt_table_bytes<BM, BN>();
    const DykConvDesc& a = args.d[0];

    const int HW = TW + 2, HR = (TH + 2) * HW, NB = (HR + 15) >> 4;
    const int HB_BYTES = NB * 1024;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* t_out = (int*)smem;                       // [BN]
    int* t_res = t_out + BN;                       // [BN]
    char* sink = (char*)(t_res + BN);              // [1024] target of dummy DMA
    float* s_stat = (float*)(sink + 1024);         // [8][2][BM]
    char* rings = smem + lt_rings_off<BM, BN>();
    const int grp_bytes = LT_NA * A_BYTES + 2 * HB_BYTES;
    char* sC = smem + TABLE_BYTES;                 // epilogue staging tile (overlays the rings)

    const int tid_all = threadIdx.x;
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane(tid_all / GT) : 0;
    const int tid = KG > 1 ? tid_all - grp * GT : tid_all;
    const int lane = tid & 63, wid = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const int wm = wid / WNn, wn = wid % WNn;
    char* sA = rings + grp * grp_bytes;            // [NA][A_BYTES]
    char* sH = sA + LT_NA * A_BYTES;               // [2][HB_BYTES]

    const int Hi = __builtin_amdgcn_readfirstlane(a.Hi), Wi = __builtin_amdgcn_readfirstlane(a.Wi);
    const int ldx = __builtin_amdgcn_readfirstlane(a.ldx), Cin = __builtin_amdgcn_readfirstlane(a.Cin);
    const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
    const int tiles_x = Wi / TW, tiles_y = Hi / TH;
    const int tiles_n = a.B * tiles_y * tiles_x;
    const int blk = blockIdx.x;
    int bid = xcd_remap(blk, gridDim.x);
    // split-K across workgroups (DykConvDesc.splitk): the S slices of a tile have consecutive remapped ids
    const int SK = a.splitk > 1 ? __builtin_amdgcn_readfirstlane(a.splitk) : 1;
    int slice = 0;
    if (SK > 1) { slice = bid % SK; bid /= SK; }
    const int tile_id = bid;
    const int m0 = (bid / tiles_n) * BM;
    int nt = bid % tiles_n;
    const int bimg = nt / (tiles_y * tiles_x);
    nt -= bimg * (tiles_y * tiles_x);
    const int ty0 = (nt / tiles_x) * TH, tx0 = (nt % tiles_x) * TW;

    for (int p = tid_all; p < BN; p += 512) {
        const int py = p / TW, px = p - py * TW;
        const int pix = (bimg * a.Ho + ty0 + py) * a.Wo + tx0 + px;
        t_out[p] = pix * a.ldy;
        t_res[p] = pix * a.ldr;
    }
    // tap tables in SGPRs (every use below has a compile-time tap index): halo-row offset and weight element offset of a tap
    int s_tap_h[9], s_tap_w[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        s_tap_h[q] = __builtin_amdgcn_readfirstlane((int)a.tdy[q] * HW + (int)a.tdx[q]);
        s_tap_w[q] = __builtin_amdgcn_readfirstlane((int)a.twt[q] * Cout * Cin);
    }
    // halo DMA pieces of this wave: piece q = t * GW + wave of tap step t covers halo rows q*16 .. +15; lane -> (row, physical
    // slot); source element offset of the lane's 16 bytes, -1 = zero page (outside the image / behind the patch)
    int hsrc[LT_HSMAX];
#pragma unroll
    for (int t = 0; t < LT_HSMAX; ++t) {
        const int q = t * GW + wv;
        const int row = q * 16 + (lane >> 2), pslot = lane & 3;
        int src = -1;
        if (row < HR) {
            const int hy = row / HW, hx = row - hy * HW;
            const int y = ty0 + hy - 1, x = tx0 + hx - 1;
            if ((unsigned)y < (unsigned)Hi && (unsigned)x < (unsigned)Wi) {
                const int ls = (lds_off<LT_ROWB>(row, pslot) - row * LT_ROWB) >> 4;     // logical slot stored at this physical slot
                src = ((bimg * Hi + y) * Wi + x) * ldx + ls * 8;
            }
        }
        hsrc[t] = src;
    }
    __syncthreads();

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const T* __restrict__ xg = sgpr_ptr((const T*)a.x);
    const T* __restrict__ wg = sgpr_ptr((const T*)a.w);
    const T* zero = (const T*)dyk_zero_page;
    // K-groups split the 32-channel chunks; the chunk count of group 0 (the largest share) drives the common barriers
    const int nchunks_all = Cin >> 5;
    const int k_lo = SK > 1 ? (slice * nchunks_all) / SK : 0;                 // this workgroup's share of the 32-channel chunks
    const int nchunks = SK > 1 ? ((slice + 1) * nchunks_all) / SK - k_lo : nchunks_all;
    const int per = (nchunks + KG - 1) / KG;
    const int c_begin = k_lo + (grp * per < nchunks ? grp * per : nchunks);
    const int c_end = c_begin + per < k_lo + nchunks ? c_begin + per : k_lo + nchunks;
    const int nch = ((a.tune >> 17) & 1) ? 0 : c_end - c_begin;      // chunks of this group (tune bit 17, analysis: no K loop)
    const int frow = lane & 15, fslot = lane >> 4;

    // fragment addressing: weight rows are wave-constant (stage base + immediate), halo rows move with the tap
    const unsigned sA_u = lds_addr_of(sA), sH_u = lds_addr_of(sH), sink_u = lds_addr_of(sink);
    const unsigned a_fbase = (unsigned)lds_off<LT_ROWB>(wm * 64 + frow, fslot);     // + mi * 1024 (row + 16: same swizzle key)
    int hrow[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int p = wn * 80 + ni * 16 + frow;
        const int py = p / TW;
        hrow[ni] = (py + 1) * HW + (p - py * TW) + 1;
    }
    // weight DMA: instruction `inst` of a step covers tile rows inst*16 .. +15; lane -> (row, physical slot)
    int a_off[NAW];
#pragma unroll
    for (int j = 0; j < NAW; ++j) {
        const int inst = j * GW + wv;
        const int row = inst * 16 + (lane >> 2);
        const int co = m0 + row;
        const int ls = (lds_off<LT_ROWB>(row, lane & 3) - row * LT_ROWB) >> 4;
        a_off[j] = co < Cout ? co * Cin + ls * 8 : -1;
    }
    auto stage_a = [&](int buf, int c0, int tapw) {
        const long wbase = (long)tapw + c0;
#pragma unroll
        for (int j = 0; j < NAW; ++j) {
            const int inst = j * GW + wv;
            const T* src = a_off[j] >= 0 ? wg + wbase + a_off[j] : zero;
            glds16(src, sA_u + buf * A_BYTES + inst * 1024);
        }
    };
    // one halo DMA instruction of tap step t: piece q = t * GW + wave of the patch of channel offset c0 into buffer hb; a dummy
    // into the sink when the piece does not exist (every wave issues the same number of DMA instructions per step: ONE counted
    // vmcnt fits all)
    auto stage_h = [&](int hb, int c0, int t, int so, bool real) {
        const int q = t * GW + wv;
        const bool ok = real && q < NB;
        const T* src = (ok && so >= 0) ? xg + (long)so + c0 : zero;
        glds16(src, ok ? sH_u + hb * HB_BYTES + q * 1024 : sink_u);
    };
    // fragment reads: byte offsets from the start of the dynamic LDS block (the compiler folds them into ds_read_b128)
    const unsigned sA_o = (unsigned)(sA - smem), sH_o = (unsigned)(sH - smem);
    uint4 fa[2][MI], fb[2][NI];
    auto read_a = [&](int slot, unsigned stage_o) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[slot][mi] = *(const uint4*)(smem + (stage_o + a_fbase + mi * 1024));
    };
    auto read_b = [&](int slot, unsigned halo_o, int toff) {
        // (opaque: otherwise the 9 x 5 swizzled halo addresses are hoisted out of the chunk loop as loop invariants -- 90 VGPRs,
        // spilled to scratch, whose reloads wait vmcnt(0) and drain the DMA ring; five VALU ops per fragment in the MFMA shadow)
        asm volatile("" : "+s"(toff));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[slot][ni] = *(const uint4*)(smem + (halo_o + (unsigned)lds_off<LT_ROWB>(hrow[ni] + toff, fslot)));
    };

    // ---- prologue: the whole halo of the first chunk, weight tiles of steps 0 .. AHEAD-1; fragments of step 0
    if (nch > 0) {
#pragma unroll
        for (int t = 0; t < LT_HSMAX; ++t) stage_h(0, c_begin * 32, t, hsrc[t], true);
#pragma unroll
        for (int i = 0; i < LT_AHEAD; ++i) {
            const int cc = (i / 9 < nch) ? i / 9 : nch - 1;
            stage_a(i, (c_begin + cc) * 32, s_tap_w[i % 9]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nch > 0) { read_a(0, sA_o); read_b(0, sH_o, s_tap_h[0]); }

    // ---- main loop, nine unrolled tap steps per 32-channel chunk.  Step s = (chunk c, tap t), fragments of slot t & 1:
    //   5 MFMAs | read the weight fragments of step s+1 | 5 MFMAs | read its halo fragments | 5 MFMAs | DMA: weights of step
    //   s + AHEAD | 5 MFMAs | DMA: a halo piece of chunk c+1 (or a dummy) | counted vmcnt: everything issued LAG steps ago has
    //   landed | barrier.  Every non-MFMA instruction sits between two MFMA groups of the same wave; with two waves per SIMD
    //   the matrix pipe always has work queued while a wave issues its loads.
    int wr = LT_AHEAD % LT_NA;                     // ring stage the next weight DMA goes to
    int rd = 1 % LT_NA;                            // ring stage of step s+1
    auto mma_row = [&](int slot, int mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], fa[slot][mi], fb[slot][ni]);
    };
    auto step = [&](int c, auto t_tag) {
        constexpr int t = decltype(t_tag)::value;
        constexpr int CUR = t & 1, NXT = CUR ^ 1;
        constexpr int tn = (t + 1) % 9;                            // tap of step s+1
        constexpr int ta = (t + LT_AHEAD) % 9, da = (t + LT_AHEAD) / 9;    // tap / chunk advance of the step whose weights go out now
        mma_row(CUR, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(NXT, sA_o + rd * A_BYTES);
        mma_row(CUR, 1);
        __builtin_amdgcn_sched_barrier(0);
        read_b(NXT, sH_o + (((c + (t + 1) / 9) & 1) ? HB_BYTES : 0), s_tap_h[tn]);
        mma_row(CUR, 2);
        __builtin_amdgcn_sched_barrier(0);
        {
            // (behind the last step of the group: re-stage its last tile into a stage nobody reads any more -- uniform DMA count)
            const int cc = c + da < nch ? c + da : nch - 1;
            stage_a(wr, (c_begin + cc) * 32, s_tap_w[ta]);
        }
        mma_row(CUR, 3);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t < LT_HSMAX) stage_h((c + 1) & 1, (c_begin + c + 1) * 32, t, hsrc[t], c + 1 < nch);
        else glds16(zero, sink_u);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LT_LAG * NPW) : "memory");
        __builtin_amdgcn_s_barrier();
        wr = wr + 1 == LT_NA ? 0 : wr + 1;
        rd = rd + 1 == LT_NA ? 0 : rd + 1;
    };
    for (int c = 0; c < (((a.tune >> 17) & 1) ? 0 : per); ++c) {
        if (c < nch) {
            step(c, std::integral_constant<int, 0>{}); step(c, std::integral_constant<int, 1>{}); step(c, std::integral_constant<int, 2>{});
            step(c, std::integral_constant<int, 3>{}); step(c, std::integral_constant<int, 4>{}); step(c, std::integral_constant<int, 5>{});
            step(c, std::integral_constant<int, 6>{}); step(c, std::integral_constant<int, 7>{}); step(c, std::integral_constant<int, 8>{});
            // nine steps = an odd number: the fragments of the next chunk's first step sit in slot 1 -- move them to slot 0
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[0][mi] = fa[1][mi];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fb[0][ni] = fb[1][ni];
        } else {
            // a K-group with fewer chunks keeps the common barriers company
            for (int q = 0; q < 9; ++q) __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the re-staged tail tiles: nothing may land in LDS behind the epilogue's back)
    __builtin_amdgcn_s_barrier();

    if constexpr (KG > 1) {
        // fold the K-groups in group order: group g parks its accumulators in LDS (lane-linear float4), group 0 adds them
        float4* park = (float4*)(smem + TABLE_BYTES);      // overlays the rings: every wave is behind the loop's last barrier
#pragma unroll 1
        for (int g = 1; g < KG; ++g) {
            if (grp == g) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        park[((mi * NI + ni) * GW + wid) * 64 + lane] = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const float4 v = park[((mi * NI + ni) * GW + wid) * 64 + lane];
                        acc[mi][ni][0] += v.x; acc[mi][ni][1] += v.y; acc[mi][ni][2] += v.z; acc[mi][ni][3] += v.w;
                    }
            }
            __syncthreads();
        }
        if (grp != 0) return;                              // (ended waves no longer count at the barriers of the epilogue)
    }
    if (SK > 1) {
        if (!splitk_exchange<GW, MI, NI>(acc, a, tile_id, slice, SK, tid, (int*)s_stat)) return;
    }
    conv_epilogue<T, BM, BN, EPIK, GT, WMn, WNn>(a, acc, sC, s_stat, t_out, t_res, m0, SK > 1 ? tile_id : blk);
}

// patch of BN pixels for an H x W map: TW | W, TH = BN / TW | H, halo loadable in LT_HSMAX steps by GW waves; `code` > 0 forces
// TW = 10 << code, 0 = the patch with the fewest halo rows.  Returns false when no patch fits.
inline bool lt_pick_patch(int H, int W, int BN, int GW, int code, LtGeom& g) {
    int best = 0;
    for (int tw = 20; tw <= 320; tw <<= 1) {
        if (code > 0 && tw != (10 << code)) continue;
        if (W % tw || BN % tw) continue;
        const int th = BN / tw;
        if (H % th) continue;
        const LtGeom q = lt_geom(th, tw);
        if (q.NB > LT_HSMAX * GW) continue;
        if (!best || q.HR < best) { best = q.HR; g = q; }
    }
    return best != 0;
}

inline bool conv_lt_eligible(const DykConvDesc* d) {
    if (d->dtype != DYK_BF16 || d->ntaps != 9 || d->ncls > 1 || d->twin) return false;
    if (d->isy != 1 || d->isx != 1 || d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0) return false;
    if (d->Hg != d->Hi || d->Wg != d->Wi || d->Ho != d->Hi || d->Wo != d->Wi) return false;
    if (d->Cin % 32 || d->Cout % 8) return false;
    if (d->flags & (DYK_EPI_OUT_F32 | DYK_EPI_BNFWD)) return false;
    for (int q = 0; q < 9; ++q)
        if (d->tdy[q] < -1 || d->tdy[q] > 1 || d->tdx[q] < -1 || d->tdx[q] > 1) return false;
    return true;
}

template <int WMn, int WNn, int KG, int EPIK>
int launch_conv_lt(const DykConvDesc* d, hipStream_t stream) {
    constexpr int BM = 64 * WMn, BN = 80 * WNn, GW = WMn * WNn;
    if (!conv_lt_eligible(d)) return DYK_ERR_UNSUPPORTED;
    if (!conv_vec_ok(d, 2, 2)) return DYK_ERR_UNSUPPORTED;           // staged 16-byte epilogue only
    const int sk = conv_splitk_of(d);
    if ((d->Cin / 32) < KG * sk) return DYK_ERR_UNSUPPORTED;
    LtGeom g;
    if (!lt_pick_patch(d->Hi, d->Wi, BN, GW, (d->tune >> 24) & 0xf, g)) return DYK_ERR_UNSUPPORTED;
    const size_t ring = (size_t)lt_rings_off<BM, BN>() + (size_t)KG * (LT_NA * (size_t)BM * LT_ROWB + 2 * (size_t)g.NB * 1024);
    size_t stage_c = lt_table_bytes<BM, BN>() + (size_t)BN * (BM * 2 + 16);
    if constexpr (EPIK == 1) {
        const size_t need = lt_table_bytes<BM, BN>() + bnbwd_sliced_bytes<bf16_t, BM, WNn>((d->flags & DYK_EPI_ADDEND) != 0);
        if (need > stage_c) stage_c = need;
    }
    const size_t park = KG > 1 ? lt_table_bytes<BM, BN>() + (size_t)GW * 64 * 20 * 16 : 0;
    size_t lds = ring > stage_c ? ring : stage_c;
    if (park > lds) lds = park;
    if (lds > 160 * 1024) return DYK_ERR_UNSUPPORTED;
    static DykDeviceOnce attr_set;   // (the attribute is per DEVICE: one flag per device id)
    auto kfn = conv_lt_kernel<WMn, WNn, KG, EPIK>;
    if (attr_set.first()) {
        DYK_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int tiles_n = d->B * (d->Hi / g.TH) * (d->Wi / g.TW);
    const int tiles_m = dyk_div_up(d->Cout, BM);
    ConvArgs args;
    if (sk > 1 && (tiles_n * tiles_m > d->sk_cnt_n || (int64_t)tiles_n * tiles_m * sk * BM * BN * 4 > d->sk_ws_bytes)) return DYK_ERR_ARG;
    const unsigned grid = conv_fill_args(args, d, tiles_n * tiles_m * sk, true);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, stream, args, g.TH, g.TW);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

// tune word of the large-tile kernels: pixel-tile code 5 (bits 12..15), shape in bits 8..11 (1 = 128 x 320, 2 = 256 x 160,
// 3 = 128 x 160 with two K-groups), patch width code in bits 24..27 (0 = fewest halo rows, k: TW = 10 << k)
template <int EPIK>
int dispatch_conv_lt(const DykConvDesc* d, hipStream_t stream) {
    switch ((d->tune >> 8) & 0xf) {
    case 1: return launch_conv_lt<2, 4, 1, EPIK>(d, stream);
    case 2: return launch_conv_lt<4, 2, 1, EPIK>(d, stream);
    case 3: return launch_conv_lt<2, 2, 2, EPIK>(d, stream);
    default: return DYK_ERR_UNSUPPORTED;
    }
}

}  // namespace
