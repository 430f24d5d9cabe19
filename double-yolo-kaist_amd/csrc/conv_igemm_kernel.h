// Implicit-GEMM convolution (forward and data gradient) on the CDNA4 matrix cores.
//
// GEMM view: M = output channels (A operand = packed weights [tap][Cout][Cin]),
//            N = launch-grid positions (B operand = channels-last activations),
//            K = taps x Cin, walked as (Cin chunk outer, tap inner) so that the nine
//            shifted re-reads of one activation chunk hit L1/L2 instead of HBM.
// Block = 256 threads = 4 waves, tile BM x BN (BN = 80 | 128 | 160 pixels), K step = BKB bytes of channels
// (64 or 128 B per row).  Global -> LDS by LDS-DMA (XOR-swizzled 16-byte slots, zero fill for
// padding taps / ragged tiles from a zero page), 2- or 3-stage ring, one barrier per K step.
// MFMA: v_mfma_f32_16x16x32_bf16 (bf16) or 4 x v_mfma_f32_16x16x4_f32 (f32) per 16-byte
// fragment pair; both operands are read from LDS with the same (row = lane&15,
// slot = lane>>4) pattern so the K permutation inside a fragment cancels.
// D layout: acc[r] = D[m = (lane>>4)*4 + r][n = lane&15]  -> four consecutive output
// channels of one pixel per lane = one 8/16-byte channels-last store.
//
// Replaces: nn.Conv2d forward at reference models.py:34-42 (+ the BatchNorm2d/activation
// that follow it at :46-62 when run with the AFFINE epilogue), and autograd's
// convolution_backward (input gradient) for the same layers.
#pragma once
#pragma once
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "dyk_common.h"

namespace {

// d[1] = the twin problem of a two-problem launch (DykConvDesc.twin: pointer fields differ only); pair_tiles = workgroups
// per problem rounded up to a multiple of 8 (so that both halves see the same block -> XCD relation), 0 = single problem
struct ConvArgs {
    DykConvDesc d[2];
    int pair_tiles;
    // Tap table in closed form, detected by the launcher: tdy[q] = a0 + sy * (q / kw), tdx[q] = b0 + sx * (q % kw),
    // twt[q] = w0 + q -- every forward k x k table (ops.fwd_taps) and the data gradient of a stride-1 conv.  The kernel
    // then BUILDS its LDS tap tables instead of loading the byte arrays of the descriptor: indexed by the lane those came
    // through vector memory, a ~1 us round trip in front of every workgroup's first barrier (kw = 0: generic table, loaded).
    int aff_kw, aff_a0, aff_sy, aff_b0, aff_sx, aff_w0;
};

// workgroup -> (problem, block id within the problem, blocks per problem)
__device__ inline int conv_pick_problem(const ConvArgs& args, int& blk, int& nblk) {
    blk = blockIdx.x; nblk = gridDim.x;
    if (args.pair_tiles == 0) return 0;
    const int sel = blk >= args.pair_tiles ? 1 : 0;
    blk -= sel * args.pair_tiles;
    nblk = args.pair_tiles;
    return sel;
}

// largest DYK_EPI_BNFWD launch: with <= 80 KB of LDS and a two-wave register budget two such launches are resident together
constexpr int DYK_BNFWD_MAX_GRID = 256;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a compile-time constant in the body
template <typename F, int... I> __device__ __forceinline__ void dyk_static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void dyk_static_for(F& f) { dyk_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// LDS the sliced BatchNorm-backward epilogue needs in the staging region: BatchNorm vectors + three slice buffers per streamed
// tensor (raw output; chain mode: + the addend); WN = wave columns of the tile
template <typename T, int BM, int WN> constexpr size_t bnbwd_sliced_bytes(bool chain) {
    return (size_t)4 * BM * 4 + (size_t)(chain ? 6 : 3) * 16 * WN * BM * sizeof(T);
}

// set by the launcher when y / ldy allow 8/16-byte vector stores
constexpr int EPI_INTERNAL_VEC = 1 << 30;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ inline void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ inline void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// byte offset of 16-byte slot `slot` of row `row` in a [rows][BKB] tile.
// BKB=128: 2 rows per 256-B bank row, slot ^= row & 6 ; BKB=64: 4 rows per bank row, slot ^= (row >> 1) & 2.
// Every ds_read_b128 lane group (MI355X_MICROARCH section LDS: rows {0-3, 12-15} of one slot + rows {4-11} of the next) then
// hits 16 distinct slots for the (row = base + (lane & 15), slot = lane >> 4) fragment pattern at ANY base row -- found by
// exhaustive search over the linear keys (round 4).  The round-1 keys ((row >> 1) & 7 and 3 * ((row >> 3) & 1)) were
// conflict-free for fragment-aligned bases only: the tap-shifted reads of the halo kernels were 2-way conflicts on 3 of 4
// (128-B rows) / 7 of 8 (64-B rows) shifts.
template <int BKB> __device__ inline int lds_off(int row, int slot) {
    if (BKB == 128) return row * 128 + ((slot ^ (row & 6)) << 4);
    return row * 64 + ((slot ^ ((row >> 1) & 2)) << 4);
}

// a wave-uniform pointer pinned in SGPRs (two v_readfirstlane): the compiler cannot re-materialise it by re-loading the kernel argument
template <typename P> __device__ inline P* sgpr_ptr(P* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (P*)(((unsigned long long)hi << 32) | lo);
}

// all-zero source for padding taps / ragged rows of the LDS-DMA path
__device__ uint4 dyk_zero_page[8];

#define DYK_AS3 __attribute__((address_space(3)))

// One LDS-DMA wave instruction: 64 lanes x 16 B from per-lane global addresses to the wave-uniform
// LDS byte address `lds_addr` + lane*16.  Issued through inline asm on purpose: hipcc drains
// vmcnt(0) in front of every ds_read while a *builtin* LDS-DMA is in flight (it cannot prove the
// ring slots disjoint), which serialises the pipeline; hidden in asm, the DMA is ordered solely by
// the counted s_waitcnt vmcnt(N) + s_barrier below (cdna_hip_programming.md §5.7).
__device__ inline void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}
__device__ inline unsigned lds_addr_of(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(DYK_AS3 const char*)p);
}

// how the four waves of a workgroup tile BM (channels) x BN (pixels): BN = 128 -> 2x2 / 1x4, BN = 160 -> 2x2,
// BN = 80 -> 4x1 (80 and 160 pixels = 5 and 10 MFMA columns: the feature maps of this path have 5*2^k pixels, so
// these tile widths give whole waves of workgroups where 128 leaves 1.25 or 0.6)
template <int BM, int BN> struct WaveGrid {
    static constexpr int WM = (BN == 80) ? 4 : ((BN == 160) ? 2 : ((BM >= 128) ? 2 : 1));
    static constexpr int WN = 4 / WM;
    static_assert(BM / WM >= 16 && (BN / WN) % 16 == 0, "unsupported tile");
};
template <int BN> constexpr int table_bytes() { return BN * (3 * 4 + 2 * 2) + 1024 + 4 * 32 * 4 + 4 * 2 * 128 * 4; }

// Shared epilogue of the convolution kernels: BN statistics, affine / activation / residual / accumulate, staged
// coalesced stores.  Called by every thread after the K loop's last barrier (the operand ring is free: sC overlays it).
// EPIK = 0: forward / plain data-gradient epilogues (statistics, affine, activation, residual, accumulate);
// EPIK = 1: the fused BatchNorm-backward epilogues (DYK_EPI_BNBWD);
// EPIK = 3: conv + train-mode BatchNorm + activation in one launch (DYK_EPI_BNFWD: statistics, device-wide arrival counter, fold,
//           normalise from the accumulators; bf16; `pub` = this workgroup publishes the BatchNorm vectors of its channels).
// Separate kernel instantiations: with both families in
// one kernel the 128 x 160 tile spilled 320-350 VGPRs (272-332 bytes of scratch per lane, also paid by the forward launches:
// +0.4 ms per step over all forward convolutions when the LDS-DMA form of the BatchNorm-backward epilogue was added).
// NT / WMP / WNP: threads that run the epilogue (threadIdx.x < NT; 256 = the four waves of the generic kernels) and their
// wave grid (0 = WaveGrid<BM, BN>); the 8-wave large-tile kernels (conv_lt_kernel.h) pass their own.
template <int BM, int BN, int WMP, int WNP> struct EpiGrid {
    static constexpr int WM = WMP > 0 ? WMP : WaveGrid<BM, BN>::WM;
    static constexpr int WN = WNP > 0 ? WNP : WaveGrid<BM, BN>::WN;
};
template <typename T, int BM, int BN, int EPIK = 0, int NT = 256, int WMP = 0, int WNP = 0>
__device__ __forceinline__ void conv_epilogue(const DykConvDesc& desc, f32x4_t (&acc)[(BM / EpiGrid<BM, BN, WMP, WNP>::WM) / 16][(BN / EpiGrid<BM, BN, WMP, WNP>::WN) / 16],
                                              char* sC, float* s_stat, const int* t_out, const int* t_res, int m0, int blk,
                                              bool pub = false, unsigned nblk = 0) {
    // The descriptor fields this epilogue uses, in SGPRs: read through the kernel-argument reference they are re-loaded
    // (s_load + s_waitcnt lgkmcnt(0)) behind every barrier / LDS-DMA statement -- 34 scalar loads in the staged store loop of
    // the 128 x 160 tile (ISA, round 3).
    struct {
        void* y; const void* res; const void* add; double* stats; const float* scale; const float* shift; const float* aux0; const float* aux1;
        int Cout, act, flags, stats_slots, tune;
    } a;
    a.y = sgpr_ptr(desc.y); a.res = sgpr_ptr(desc.res); a.add = sgpr_ptr(desc.add); a.stats = sgpr_ptr(desc.stats);
    a.scale = sgpr_ptr(desc.scale); a.shift = sgpr_ptr(desc.shift); a.aux0 = sgpr_ptr(desc.aux0); a.aux1 = sgpr_ptr(desc.aux1);
    a.Cout = __builtin_amdgcn_readfirstlane(desc.Cout); a.act = __builtin_amdgcn_readfirstlane(desc.act);
    a.flags = __builtin_amdgcn_readfirstlane(desc.flags); a.stats_slots = __builtin_amdgcn_readfirstlane(desc.stats_slots);
    a.tune = __builtin_amdgcn_readfirstlane(desc.tune);
    constexpr int WM = EpiGrid<BM, BN, WMP, WNP>::WM, WN = EpiGrid<BM, BN, WMP, WNP>::WN;
    constexpr int NWV = NT / 64;
    static_assert(WM * WN == NWV, "wave grid must cover the epilogue threads");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const bool abl_nostore = (a.tune >> 16) & 1;
    if ((a.tune >> 20) & 1) {                  // ablation: no epilogue (keep the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) sum += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
        if (sum == 123.456f) ((float*)a.y)[0] = sum;
        return;
    }
    const int flags = a.flags;
    const int mlane = (lane >> 4) * 4;
    if ((EPIK == 0 || EPIK == 3) && (flags & DYK_EPI_STATS)) {
        // per-channel sum / sum of squares of the raw accumulators: in-lane over ni, DPP row rotate-adds over the
        // 16 pixel lanes, one LDS slot per wave (summed in wave order: the forward pass is reproducible -- with LDS
        // float atomics the order of the adds, and through the chaotic random-weight nets the outputs, changed from
        // run to run), then ONE fp64 atomic per channel and workgroup into a replica of the statistics buffer.
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float v = acc[mi][ni][r];
                    s1 += v; s2 += v * v;
                }
                s1 = row16_sum(s1);
                s2 = row16_sum(s2);
                if ((lane & 15) == 0) {
                    const int ml = wm * WTM + mi * 16 + mlane + r;
                    s_stat[(wn * 2 + 0) * BM + ml] = s1;
                    s_stat[(wn * 2 + 1) * BM + ml] = s2;
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < 2 * BM; e += NT) {
            const int ml = e % BM, which = e / BM;
            if (m0 + ml < a.Cout) {
                float tot = 0.f;
#pragma unroll
                for (int q = 0; q < WN; ++q) tot += s_stat[(q * 2 + which) * BM + ml];
                double* st = a.stats + (size_t)((unsigned)blk % (unsigned)(a.stats_slots > 0 ? a.stats_slots : 1)) * 2 * a.Cout;
                atomicAdd(st + which * a.Cout + m0 + ml, (double)tot);
            }
        }
    }
    const bool affine = flags & DYK_EPI_AFFINE;
    const bool has_res = (flags & DYK_EPI_RESIDUAL) && EPIK != 3;     // (EPIK 3 adds the residual in its own second pass)
    const bool accum = flags & DYK_EPI_ACCUM;
    const bool out_f32 = (flags & DYK_EPI_OUT_F32) || sizeof(T) == 4;

    if (flags & EPI_INTERNAL_VEC) {
        // ---- staged, coalesced store: accumulators -> LDS tile [pixel][channel] -> 16-byte global stores
        // in which 16 consecutive lanes cover one contiguous channel row of a pixel (the per-lane 8-byte
        // scatter of the MFMA layout wrote 32-byte fragments and cost 2-3x the store time).
        // The code is instantiated per (output type, activation) and selected by ONE switch: with the
        // activation switch inside the 64-value unrolled loop the epilogue was 20k instructions of
        // branches and cost 9 us per launch (tools/gpu_probe.py ablate).
        auto staged = [&](auto of32_tag, auto act_tag, auto affine_tag) {
            constexpr bool OF32 = decltype(of32_tag)::value;
            constexpr int ACT = decltype(act_tag)::value;
            constexpr bool AFF = decltype(affine_tag)::value;
            constexpr int eso = OF32 ? 4 : 2;                     // output element size
            constexpr int rstride = BM * eso + 16;                // padded row stride (bytes)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int ml = wm * WTM + mi * 16 + mlane;
                const int m = m0 + ml;
                float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (AFF) {
                    if (m + 3 < a.Cout) {
                        if (a.scale) { const float4 t = *(const float4*)(a.scale + m); sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; }
                        if (a.shift) { const float4 t = *(const float4*)(a.shift + m); sh[0] = t.x; sh[1] = t.y; sh[2] = t.z; sh[3] = t.w; }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (m + r < a.Cout) {
                                if (a.scale) sc[r] = a.scale[m + r];
                                if (a.shift) sh[r] = a.shift[m + r];
                            }
                    }
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int nl = wn * WTN + ni * 16 + (lane & 15);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float u = acc[mi][ni][r];
                        if constexpr (AFF) u = u * sc[r] + sh[r];
                        v[r] = act_fwd_c<ACT>(u, a.act);
                    }
                    char* dst = sC + nl * rstride + ml * eso;
                    if constexpr (OF32) {
                        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        uint2 pk;
                        pk.x = f32x2_to_bf16x2(v[0], v[1]);
                        pk.y = f32x2_to_bf16x2(v[2], v[3]);
                        *(uint2*)dst = pk;
                    }
                }
            }
            __syncthreads();
            constexpr int epv_o = 16 / eso;                       // output elements per 16-byte chunk
            constexpr int cpr = BM / epv_o;                       // chunks per tile row
            constexpr int nchunk = BN * cpr;
            for (int q = tid; q < nchunk; q += NT) {
                const int row = q / cpr, cc = q % cpr;
                const int po = t_out[row];
                const int mc = m0 + cc * epv_o;
                if (po < 0 || mc >= a.Cout) continue;
                if (abl_nostore && t_out[0] != -12345) continue;
                uint4 val = *(const uint4*)(sC + row * rstride + cc * 16);
                const bool whole = (mc + epv_o <= a.Cout);
                if constexpr (OF32) {
                    float* yp = (float*)a.y + (long)po + mc;
                    float f[4] = {__uint_as_float(val.x), __uint_as_float(val.y), __uint_as_float(val.z), __uint_as_float(val.w)};
                    if (has_res) {
                        const T* rp = (const T*)a.res + (long)t_res[row] + mc;
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (mc + j < a.Cout) f[j] += ElemTraits<T>::to_f32(rp[j]);
                    }
                    if (accum) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (mc + j < a.Cout) f[j] += yp[j];
                    }
                    if (whole) *(float4*)yp = make_float4(f[0], f[1], f[2], f[3]);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (mc + j < a.Cout) yp[j] = f[j];
                    }
                } else {
                    bf16_t* yp = (bf16_t*)a.y + (long)po + mc;
                    if (whole && !has_res && !accum) {
                        *(uint4*)yp = val;
                    } else if (whole) {
                        float f[8];
                        vec_unpack<bf16_t>(val, f);
                        if (has_res) {
                            float g[8];
                            vec_unpack<bf16_t>(*(const uint4*)((const bf16_t*)a.res + (long)t_res[row] + mc), g);
#pragma unroll
                            for (int j = 0; j < 8; ++j) f[j] += g[j];
                        }
                        if (accum) {
                            float g[8];
                            vec_unpack<bf16_t>(*(const uint4*)yp, g);
#pragma unroll
                            for (int j = 0; j < 8; ++j) f[j] += g[j];
                        }
                        *(uint4*)yp = vec_pack<bf16_t>(f);
                    } else {
                        float f[8];
                        vec_unpack<bf16_t>(val, f);
                        for (int j = 0; j < 8; ++j) {
                            if (mc + j >= a.Cout) break;
                            float u = f[j];
                            if (has_res) u += bf16_to_f32(((const bf16_t*)a.res + (long)t_res[row] + mc)[j]);
                            if (accum) u += bf16_to_f32(yp[j]);
                            yp[j] = f32_to_bf16(u);
                        }
                    }
                }
            }
        };
        // ---- BatchNorm-backward reduce fused into the data gradient that produces dz (DYK_EPI_BNBWD): the staged
        // gradient tile is read back chunk by chunk together with the matching chunk of the raw conv output, turned
        // into da = dz * act'(u) and stored; sum(da), sum(da * xhat) go thread -> wave (shuffle) -> workgroup (LDS
        // atomics) -> one fp64 atomic per channel into a replica of the reduction buffer.
        auto staged_bnbwd = [&](auto actb_tag) {
            constexpr int ACTB = decltype(actb_tag)::value;
            constexpr int eso = (int)sizeof(T);
            constexpr int rstride = BM * eso + 16;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int ml = wm * WTM + mi * 16 + mlane;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int nl = wn * WTN + ni * 16 + (lane & 15);
                    char* dst = sC + nl * rstride + ml * eso;
                    if constexpr (sizeof(T) == 4) {
                        *(float4*)dst = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
                    } else {
                        uint2 pk;
                        pk.x = f32x2_to_bf16x2(acc[mi][ni][0], acc[mi][ni][1]);
                        pk.y = f32x2_to_bf16x2(acc[mi][ni][2], acc[mi][ni][3]);
                        *(uint2*)dst = pk;
                    }
                }
            }
            __syncthreads();
            constexpr int EPVT = 16 / eso;                        // elements per 16-byte chunk
            constexpr int cpr = BM / EPVT;                        // chunks per tile row (divides 256: one chunk column per thread)
            constexpr int nchunk = BN * cpr;
            const int cc = tid % cpr;
            const int mc = m0 + cc * EPVT;
            const bool live = mc + EPVT <= a.Cout;
            float sc[EPVT], sh[EPVT], mu[EPVT], rs[EPVT], s1[EPVT], s2[EPVT];
#pragma unroll
            for (int j = 0; j < EPVT; ++j) {
                sc[j] = live ? a.scale[mc + j] : 0.f; sh[j] = live ? a.shift[mc + j] : 0.f;
                mu[j] = live ? a.aux0[mc + j] : 0.f; rs[j] = live ? a.aux1[mc + j] : 0.f;
                s1[j] = s2[j] = 0.f;
            }
            const bool chain = (flags & DYK_EPI_ADDEND) != 0;     // residual chain: store dz itself, reduce da
            if (live) {
                // The raw conv output (and the chain addend) of this thread's chunks are fetched in batches of UB loads
                // issued back to back -- one memory round trip per batch.  One load per loop trip, consumed at once, made
                // this epilogue a chain of nchunk/256 dependent HBM latencies (dgrad 128->128 @64x80: 60 us against 39 us
                // for the forward conv of the same GEMM).  Addresses of dead chunks are clamped, their values ignored.
                constexpr int NIT = (nchunk + NT - 1) / NT;
                static_assert(NT % cpr == 0, "one chunk column per thread");
                // (round 3: batches of 5 / 4 / 3 loads, possible without spills now that this epilogue has its own kernels, measured
                // WORSE than two: 1x1 256->256 @32x40 chain mode 340 -> 371 us per 16 launches, +0.2 ms over all data gradients)
                constexpr int UB = (NIT % 2 == 0) ? 2 : 1;
                auto batched = [&](auto chain_tag) {
                    constexpr bool CH = decltype(chain_tag)::value;
                    for (int j0 = 0; j0 < NIT; j0 += UB) {
                        uint4 yraw[UB], adv[CH ? UB : 1];
                        int pos[UB], rows[UB];
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int q = tid + (j0 + u) * NT;
                            const int row = q < nchunk ? q / cpr : 0;
                            const int po = q < nchunk ? t_out[row] : -1;
                            rows[u] = row; pos[u] = po;
                            yraw[u] = *(const uint4*)((const T*)a.res + (po >= 0 ? (long)t_res[row] + mc : 0L));
                            if constexpr (CH) adv[u] = a.add ? *(const uint4*)((const T*)a.add + (po >= 0 ? (long)po + mc : 0L)) : make_uint4(0u, 0u, 0u, 0u);   // (no addend: zero)
                        }
#pragma unroll
                        for (int u = 0; u < UB; ++u) {
                            const int po = pos[u];
                            if (po < 0) continue;
                            float g[EPVT], yv[EPVT];
                            vec_unpack<T>(*(const uint4*)(sC + rows[u] * rstride + cc * 16), g);
                            vec_unpack<T>(yraw[u], yv);
                            if constexpr (CH) {
                                float ad[EPVT];
                                vec_unpack<T>(adv[u], ad);
#pragma unroll
                                for (int j = 0; j < EPVT; ++j) g[j] += ad[j];
                                const uint4 pk = vec_pack<T>(g);
                                *(uint4*)((T*)a.y + (long)po + mc) = pk;
                                vec_unpack<T>(pk, g);              // the apply pass will see the rounded dz: reduce the same values
                            }
#pragma unroll
                            for (int j = 0; j < EPVT; ++j) {
                                const float da = g[j] * act_bwd_c<ACTB>(yv[j] * sc[j] + sh[j], a.act);
                                s1[j] += da;
                                s2[j] += da * ((yv[j] - mu[j]) * rs[j]);
                                g[j] = da;
                            }
                            if constexpr (!CH) *(uint4*)((T*)a.y + (long)po + mc) = vec_pack<T>(g);
                        }
                    }
                };
                if (chain) batched(std::true_type{}); else batched(std::false_type{});
            }
#pragma unroll
            for (int j = 0; j < EPVT; ++j) {
#pragma unroll
                for (int o = cpr; o < 64; o <<= 1) {
                    s1[j] += __shfl_xor(s1[j], o, 64);
                    s2[j] += __shfl_xor(s2[j], o, 64);
                }
            }
            if (lane < cpr) {                                     // one slot per wave, summed in wave order below
#pragma unroll
                for (int j = 0; j < EPVT; ++j) {
                    s_stat[(wid * 2 + 0) * BM + cc * EPVT + j] = live ? s1[j] : 0.f;
                    s_stat[(wid * 2 + 1) * BM + cc * EPVT + j] = live ? s2[j] : 0.f;
                }
            }
            __syncthreads();
            for (int e = tid; e < 2 * BM; e += NT) {
                const int ml = e % BM, which = e / BM;
                if (m0 + ml < a.Cout) {
                    // (four waves: (0 + 1) + (2 + 3), the order the generic kernels have always used; eight: two such sums)
                    float tot = (s_stat[(0 * 2 + which) * BM + ml] + s_stat[(1 * 2 + which) * BM + ml]) +
                                (s_stat[(2 * 2 + which) * BM + ml] + s_stat[(3 * 2 + which) * BM + ml]);
                    if constexpr (NWV == 8)
                        tot += (s_stat[(4 * 2 + which) * BM + ml] + s_stat[(5 * 2 + which) * BM + ml]) +
                               (s_stat[(6 * 2 + which) * BM + ml] + s_stat[(7 * 2 + which) * BM + ml]);
                    double* st = a.stats + (size_t)((unsigned)blk % (unsigned)(a.stats_slots > 0 ? a.stats_slots : 1)) * 2 * a.Cout;
                    atomicAdd(st + which * a.Cout + m0 + ml, (double)tot);
                }
            }
        };
        // ---- BatchNorm-backward epilogue, round-4 form: SLICED and software-pipelined.  The round-3 form fetched the raw conv
        // output of the whole tile by LDS-DMA, waited, computed, then stored: three serial phases (tools/lt_probe.py: 15 us behind
        // an 18.6 us K loop on 3x3 128->128 @64x80), and its per-lane cell reads (16 pixel lanes x 256-byte dense rows) were 16-way
        // bank conflicts.  Here the tile is walked in NI slices of 16 pixels per wave column (slice ni = accumulator column ni of
        // every wave): the raw output (and the chain addend) of slice ni+2 is in flight by LDS-DMA while slice ni is computed and
        // the coalesced stores of slices ni-1, ni-2 drain -- one counted s_waitcnt vmcnt per slice (stores count too: every
        // thread issues the same number per slice, so the immediates are exact), raw s_barrier (a __syncthreads() would drain
        // vmcnt(0)).  Rows are dense (a DMA instruction fills whole rows) with the 16-byte chunk XOR-swizzled by the row index on
        // the SOURCE side: the 16 pixel lanes of a cell read hit 16 different chunks.  BatchNorm vectors sit in LDS; sums are kept
        // in registers across the slices and folded once (DPP row sums, waves in order, one fp64 atomic per channel).
        // Full tiles only (every pixel and channel of the tile valid -- the caller checks); ragged tiles take staged_bnbwd.
        auto sliced_bnbwd = [&](auto actb_tag, auto chain_tag) __attribute__((always_inline)) {
            constexpr int ACTB = decltype(actb_tag)::value;
            constexpr bool CH = decltype(chain_tag)::value;
            constexpr int eso = (int)sizeof(T);
            constexpr int EPVT = 16 / eso;
            constexpr int RB = BM * eso;                          // dense row (bytes)
            constexpr int CPRW = RB / 16;                         // 16-byte chunks per row
            constexpr int RPI = 64 / CPRW;                        // rows per DMA instruction
            constexpr int SLP = 16 * WN;                          // pixels per slice
            constexpr int SB = SLP * RB;                          // bytes per slice buffer
            constexpr int NIS = SB / 1024;                        // DMA instructions per slice and tensor
            constexpr int NDU = NIS / NWV;                        // ... per wave (= 16-byte stores per thread and slice)
            constexpr int D = NDU * (CH ? 2 : 1);                 // DMA instructions per wave and slice
            static_assert(NIS % NWV == 0 && NDU >= 1 && 64 % CPRW == 0, "sliced epilogue: uniform DMA / store counts");
            const int wv = __builtin_amdgcn_readfirstlane(wid);
            float* s_par = (float*)sC;                            // [4][BM]: scale, shift, mean, rstd of the tile's channels
            char* ubuf = sC + 4 * BM * 4;                         // [3][SB] raw output slices; chain mode: + [3][SB] addend slices
            const unsigned ubuf_u = lds_addr_of(ubuf);
            for (int e = tid; e < 4 * BM; e += NT) {
                const int k = e / BM, c = e - k * BM;
                const float* src = k == 0 ? a.scale : (k == 1 ? a.shift : (k == 2 ? a.aux0 : a.aux1));
                s_par[e] = src[m0 + c];
            }
            // slice-local row r -> tile pixel: wave column r / 16, accumulator column ni, lane pixel r % 16
            auto dma_slice = [&](int ni, int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NDU; ++j) {
                    const int inst = j * NWV + wv;
                    const int r = inst * RPI + lane / CPRW, pc = lane % CPRW;
                    const int nl = (r >> 4) * WTN + ni * 16 + (r & 15);
                    const int ch = (pc ^ (r & (CPRW - 1))) * EPVT;          // logical chunk stored at physical chunk pc
                    glds16((const T*)a.res + (long)t_res[nl] + m0 + ch, ubuf_u + buf * SB + inst * 1024);
                    if constexpr (CH) glds16(a.add ? (const T*)a.add + (long)t_out[nl] + m0 + ch : (const T*)dyk_zero_page, ubuf_u + (3 + buf) * SB + inst * 1024);   // (no addend: the zero page)
                }
            };
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // (the vector loads above)
            __builtin_amdgcn_s_barrier();                          // tables / parameters visible; the rings are free
            dma_slice(0, 0);
            if constexpr (NI > 1) dma_slice(1, 1);
            float s1[MI][4], s2[MI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) s1[mi][r] = s2[mi][r] = 0.f;
            const int prow = wn * 16 + (lane & 15);               // this lane's row of a slice buffer
            auto slice = [&](auto ni_tag) __attribute__((always_inline)) {
                constexpr int ni = decltype(ni_tag)::value;
                constexpr int buf = ni % 3;
                // everything older than the DMA of slice ni has completed: younger are the stores of slices ni-2, ni-1 and the
                // DMA of slice ni+1
                constexpr int young = (ni == 0 ? (NI > 1 ? D : 0)
                                               : (ni == 1 ? NDU + (NI > 2 ? D : 0) : 2 * NDU + (ni + 1 < NI ? D : 0)));
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(young) : "memory");
                __builtin_amdgcn_s_barrier();                      // the whole slice has landed; slice ni-1's buffer reads are done
                if constexpr (ni + 2 < NI) dma_slice(ni + 2, (ni + 2) % 3);
                char* ub = ubuf + buf * SB + prow * RB;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int ml = wm * WTM + mi * 16 + mlane;
                    const int off = (((ml * eso) >> 4) ^ (prow & (CPRW - 1))) * 16 + ((ml * eso) & 15);
                    // (opaque per slice: the compiler otherwise reads the 16 x MI BatchNorm values once and keeps them live across
                    // all slices -- spills, and a scratch reload waits vmcnt(0): it would drain the DMA pipeline)
                    int po = ml;
                    asm volatile("" : "+v"(po));
                    const float4 sc = *(const float4*)(s_par + po), sh = *(const float4*)(s_par + BM + po);
                    const float4 mu = *(const float4*)(s_par + 2 * BM + po), rs = *(const float4*)(s_par + 3 * BM + po);
                    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
                    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
                    float yv[4], g[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[r] = acc[mi][ni][r];
                    if constexpr (sizeof(T) == 4) {
                        const float4 v = *(const float4*)(ub + off);
                        yv[0] = v.x; yv[1] = v.y; yv[2] = v.z; yv[3] = v.w;
                        if constexpr (CH) {
                            const float4 w = *(const float4*)(ub + 3 * SB + off);
                            g[0] += w.x; g[1] += w.y; g[2] += w.z; g[3] += w.w;
                        }
                    } else {
                        const uint2 v = *(const uint2*)(ub + off);
                        yv[0] = __uint_as_float(v.x << 16); yv[1] = __uint_as_float(v.x & 0xffff0000u);
                        yv[2] = __uint_as_float(v.y << 16); yv[3] = __uint_as_float(v.y & 0xffff0000u);
                        if constexpr (CH) {
                            // dz = acc + addend, rounded to the storage type: the apply pass will see the rounded value
                            const uint2 w = *(const uint2*)(ub + 3 * SB + off);
                            g[0] += __uint_as_float(w.x << 16); g[1] += __uint_as_float(w.x & 0xffff0000u);
                            g[2] += __uint_as_float(w.y << 16); g[3] += __uint_as_float(w.y & 0xffff0000u);
                            const uint32_t p0 = f32x2_to_bf16x2(g[0], g[1]), p1 = f32x2_to_bf16x2(g[2], g[3]);
                            g[0] = __uint_as_float(p0 << 16); g[1] = __uint_as_float(p0 & 0xffff0000u);
                            g[2] = __uint_as_float(p1 << 16); g[3] = __uint_as_float(p1 & 0xffff0000u);
                        }
                    }
                    float outv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float da = g[r] * act_bwd_c<ACTB>(yv[r] * scv[r] + shv[r], a.act);
                        s1[mi][r] += da;
                        s2[mi][r] += da * ((yv[r] - muv[r]) * rsv[r]);
                        outv[r] = CH ? g[r] : da;
                    }
                    if constexpr (sizeof(T) == 4) {
                        *(float4*)(ub + off) = make_float4(outv[0], outv[1], outv[2], outv[3]);
                    } else {
                        uint2 pk;
                        pk.x = f32x2_to_bf16x2(outv[0], outv[1]);
                        pk.y = f32x2_to_bf16x2(outv[2], outv[3]);
                        *(uint2*)(ub + off) = pk;
                    }
                    // (the running sums are made opaque here: the optimizer otherwise sinks the whole chain of adds to the end of the
                    // epilogue and keeps every slice's da / da * xhat products live until then -- 35 more registers per slice)
#pragma unroll
                    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(s1[mi][r]), "+v"(s2[mi][r]));
                }
                // (pin the slice: register-only arithmetic is free to move across barriers and asm statements, and the scheduler
                // otherwise defers the derivative / sum arithmetic of ALL slices to the end of the block, keeping every slice's
                // operands live -- 1 450 spilled registers in the 128 x 160 kernels)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // the slice's results are in its buffer
#pragma unroll
                for (int j = 0; j < NDU; ++j) {
                    const int q = tid + j * NT;
                    const int r = q / CPRW, pc = q % CPRW;
                    const int nl = (r >> 4) * WTN + ni * 16 + (r & 15);
                    const int mc = m0 + (pc ^ (r & (CPRW - 1))) * EPVT;
                    *(uint4*)((T*)a.y + (long)t_out[nl] + mc) = *(const uint4*)(ubuf + buf * SB + r * RB + pc * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            dyk_static_for<NI>(slice);
            // fold the sums: DPP row sums over the 16 pixel lanes, one LDS slot per wave column, waves in order
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int ml = wm * WTM + mi * 16 + mlane;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t1 = row16_sum(s1[mi][r]), t2 = row16_sum(s2[mi][r]);
                    if ((lane & 15) == 0) {
                        s_stat[(wn * 2 + 0) * BM + ml + r] = t1;
                        s_stat[(wn * 2 + 1) * BM + ml + r] = t2;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int e = tid; e < 2 * BM; e += NT) {
                const int ml = e % BM, which = e / BM;
                float tot = 0.f;
#pragma unroll
                for (int q = 0; q < WN; ++q) tot += s_stat[(q * 2 + which) * BM + ml];
                double* st = a.stats + (size_t)((unsigned)blk % (unsigned)(a.stats_slots > 0 ? a.stats_slots : 1)) * 2 * a.Cout;
                atomicAdd(st + which * a.Cout + m0 + ml, (double)tot);
            }
        };

        // ---- DYK_EPI_BNFWD (EPIK 3): the raw tile has its statistics in the replicas (above) and goes to y through the staged
        // store; then arrival counter -> wait for the whole launch -> fold the replicas of this workgroup's channels -> normalise
        // the tile from the accumulators (rounded to the storage type first: the values a separate pass would read back from y),
        // activation, residual, staged store to y2.  Requires every workgroup of the launch to be resident (front end: grid limit).
        auto bn_forward = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            static_assert(sizeof(T) == 2, "DYK_EPI_BNFWD is built for bf16");
            constexpr int eso = 2;
            constexpr int rstride = BM * eso + 16;
            // Ordering without fences: an agent-scope release / acquire on this 8-XCD part writes back and invalidates the whole L2 of
            // the XCD (measured: the launch took 73 us instead of 27 for conv + normalise).  Everything the workgroups exchange goes
            // through agent-scope ATOMICS, which are performed at the device's coherence point: the statistics sums (fp64 atomic
            // adds here, atomic loads below) and the counters.  s_waitcnt vmcnt(0) = this thread's atomic adds have been performed;
            // the workgroup barrier then orders them before thread 0's arrival.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* s_flag = (int*)(s_stat + 2 * BM);
            uint32_t* cnt = (uint32_t*)sgpr_ptr(desc.bn_counter);
            if (tid == 0) {
                // cnt[0] arrivals, cnt[1] error word, cnt[2] departures.  All three are zero on entry; the workgroup that leaves
                // last puts them back to zero, so the next launch on this counter (any grid size) needs no re-arming
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int ok = 1;
                const unsigned long long t0 = wall_clock64();
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nblk) {
                    __builtin_amdgcn_s_sleep(8);
                    if (wall_clock64() - t0 > 5000000ull) {     // 50 ms of the 100 MHz clock: not every workgroup is resident --
                        ok = 0;                                  // flag it instead of hanging (the counters stay dirty: error state)
                        __hip_atomic_fetch_or(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                if (ok && __hip_atomic_fetch_add(cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1u) {
                    // every workgroup has passed the wait: nobody reads the counters any more
                    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(cnt + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                *s_flag = ok;
            }
            __syncthreads();
            if (*s_flag == 0) return;
            float* s_sc = s_stat;
            float* s_sh = s_stat + BM;
            if (tid < BM) {
                const int c = m0 + tid;
                float sc = 0.f, sh = 0.f;
                if (c < a.Cout) {
                    const int slots = a.stats_slots > 0 ? a.stats_slots : 1;
                    const size_t rs = (size_t)2 * a.Cout;
                    // four accumulators over r & 3, folded (0 + 1) + (2 + 3): the order of bn_fused_fwd_kernel
                    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
                    int r = 0;
                    for (; r + 4 <= slots; r += 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            a1[u] += __hip_atomic_load(a.stats + (size_t)(r + u) * rs + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            a2[u] += __hip_atomic_load(a.stats + (size_t)(r + u) * rs + a.Cout + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    for (; r < slots; ++r) {
                        a1[0] += __hip_atomic_load(a.stats + (size_t)r * rs + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a2[0] += __hip_atomic_load(a.stats + (size_t)r * rs + a.Cout + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const double n = (double)desc.bn_count;
                    const double mean = ((a1[0] + a1[1]) + (a1[2] + a1[3])) / n;
                    double var = ((a2[0] + a2[1]) + (a2[2] + a2[3])) / n - mean * mean;
                    if (var < 0.0) var = 0.0;
                    const float rstd = (float)(1.0 / sqrt(var + (double)desc.bn_eps));
                    const float g = desc.bn_gamma ? desc.bn_gamma[c] : 1.f, b = desc.bn_beta ? desc.bn_beta[c] : 0.f;
                    sc = g * rstd;
                    sh = b - (float)mean * sc;
                    if (pub) {
                        ((float*)a.scale)[c] = sc;
                        ((float*)a.shift)[c] = sh;
                        if (desc.bn_save_mean) desc.bn_save_mean[c] = (float)mean;
                        if (desc.bn_save_rstd) desc.bn_save_rstd[c] = rstd;
                        if (desc.bn_running_mean) {
                            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                            desc.bn_running_mean[c] = (1.f - desc.bn_momentum) * desc.bn_running_mean[c] + desc.bn_momentum * (float)mean;
                            desc.bn_running_var[c] = (1.f - desc.bn_momentum) * desc.bn_running_var[c] + desc.bn_momentum * (float)unb;
                        }
                    }
                }
                s_sc[tid] = sc;
                s_sh[tid] = sh;
            }
            __syncthreads();                                       // (also: every thread is done with the raw tile in sC)
            const bool res2 = (flags & DYK_EPI_RESIDUAL) != 0;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int ml = wm * WTM + mi * 16 + mlane;
                float sc[4], sh[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { sc[r] = s_sc[ml + r]; sh[r] = s_sh[ml + r]; }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int nl = wn * WTN + ni * 16 + (lane & 15);
                    const uint32_t p0 = f32x2_to_bf16x2(acc[mi][ni][0], acc[mi][ni][1]), p1 = f32x2_to_bf16x2(acc[mi][ni][2], acc[mi][ni][3]);
                    const float u[4] = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u),
                                        __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_fwd_c<ACT>(u[r] * sc[r] + sh[r], a.act);
                    if (res2 && t_out[nl] >= 0 && m0 + ml < a.Cout) {
                        // the [shortcut] source in fp32 BEFORE the one rounding, as the separate normalise pass adds it
                        const uint2 rv = *(const uint2*)((const T*)a.res + (long)t_res[nl] + m0 + ml);
                        v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
                        v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
                    }
                    uint2 pk;
                    pk.x = f32x2_to_bf16x2(v[0], v[1]);
                    pk.y = f32x2_to_bf16x2(v[2], v[3]);
                    *(uint2*)(sC + nl * rstride + ml * eso) = pk;
                }
            }
            __syncthreads();
            constexpr int cpr = BM / 8;
            constexpr int nchunk = BN * cpr;
            const int ldy_s = __builtin_amdgcn_readfirstlane(desc.ldy), ldy2_s = __builtin_amdgcn_readfirstlane(desc.ldy2);
            bf16_t* y2 = (bf16_t*)sgpr_ptr(desc.y2);
            for (int q = tid; q < nchunk; q += NT) {
                const int row = q / cpr, cc = q % cpr;
                const int po = t_out[row];
                const int mc = m0 + cc * 8;
                if (po < 0 || mc + 8 > a.Cout) continue;
                *(uint4*)(y2 + (long)(po / ldy_s) * ldy2_s + mc) = *(const uint4*)(sC + row * rstride + cc * 16);
            }
        };
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        if constexpr (EPIK == 3) {
            if constexpr (sizeof(T) == 2) {
                staged(false_type{}, integral_constant<int, DYK_ACT_LINEAR>{}, false_type{});       // raw tile -> y
                switch (a.act) {
                case DYK_ACT_LINEAR: bn_forward(integral_constant<int, DYK_ACT_LINEAR>{}); break;
                case DYK_ACT_LEAKY: bn_forward(integral_constant<int, DYK_ACT_LEAKY>{}); break;
                case DYK_ACT_MISH: bn_forward(integral_constant<int, DYK_ACT_MISH>{}); break;
                default: bn_forward(integral_constant<int, -1>{}); break;
                }
            }
            return;
        } else if constexpr (EPIK == 1) {
            // full tiles: the sliced, pipelined form (plain and residual-chain); ragged tiles (a pixel or channel of the tile
            // outside the problem), tile shapes whose slices do not split evenly over the waves, and tune bit 21: the batched
            // register form
            constexpr bool SLICED_OK = ((16 * WN * BM * (int)sizeof(T)) / 1024) % NWV == 0 && (16 * WN * BM * (int)sizeof(T)) / 1024 >= NWV
                                       && 64 % (BM * (int)sizeof(T) / 16) == 0;
            const bool full = m0 + BM <= a.Cout && t_out[BN - 1] >= 0;
#ifndef DYK_X_NOSLICED
            if constexpr (SLICED_OK) {
                if (full && !((a.tune >> 21) & 1)) {
                    const bool chain = (flags & DYK_EPI_ADDEND) != 0;
#ifdef DYK_X_ONLY1
                    sliced_bnbwd(integral_constant<int, DYK_ACT_MISH>{}, false_type{}); return;
#endif
                    switch (a.act) {
                    case DYK_ACT_LINEAR: if (chain) sliced_bnbwd(integral_constant<int, DYK_ACT_LINEAR>{}, true_type{}); else sliced_bnbwd(integral_constant<int, DYK_ACT_LINEAR>{}, false_type{}); break;
                    case DYK_ACT_LEAKY: if (chain) sliced_bnbwd(integral_constant<int, DYK_ACT_LEAKY>{}, true_type{}); else sliced_bnbwd(integral_constant<int, DYK_ACT_LEAKY>{}, false_type{}); break;
                    case DYK_ACT_MISH: if (chain) sliced_bnbwd(integral_constant<int, DYK_ACT_MISH>{}, true_type{}); else sliced_bnbwd(integral_constant<int, DYK_ACT_MISH>{}, false_type{}); break;
                    default: if (chain) sliced_bnbwd(integral_constant<int, -1>{}, true_type{}); else sliced_bnbwd(integral_constant<int, -1>{}, false_type{}); break;
                    }
                    return;
                }
            }
#endif
#ifndef DYK_X_NOSTAGED
            switch (a.act) {
            case DYK_ACT_LINEAR: staged_bnbwd(integral_constant<int, DYK_ACT_LINEAR>{}); break;
            case DYK_ACT_LEAKY: staged_bnbwd(integral_constant<int, DYK_ACT_LEAKY>{}); break;
            case DYK_ACT_MISH: staged_bnbwd(integral_constant<int, DYK_ACT_MISH>{}); break;
            default: staged_bnbwd(integral_constant<int, -1>{}); break;
            }
#endif
            return;
        } else {
            if (out_f32) {
                // heads (bias, linear) and the fp32 dtype: runtime activation, few launches
                if (!affine && a.act == 0) staged(true_type{}, integral_constant<int, 0>{}, false_type{});
                else if (a.act == 0) staged(true_type{}, integral_constant<int, 0>{}, true_type{});
                else staged(true_type{}, integral_constant<int, -1>{}, true_type{});
                return;
            }
            if (!affine && a.act == 0) { staged(false_type{}, integral_constant<int, 0>{}, false_type{}); return; }
            switch (a.act) {
            case DYK_ACT_LINEAR: staged(false_type{}, integral_constant<int, DYK_ACT_LINEAR>{}, true_type{}); break;
            case DYK_ACT_LEAKY: staged(false_type{}, integral_constant<int, DYK_ACT_LEAKY>{}, true_type{}); break;
            case DYK_ACT_MISH: staged(false_type{}, integral_constant<int, DYK_ACT_MISH>{}, true_type{}); break;
            default: staged(false_type{}, integral_constant<int, -1>{}, true_type{}); break;
            }
            return;
        }
    }
    if constexpr (EPIK == 1 || EPIK == 3) return;   // (these epilogues exist in the staged form only: validated by the front end)
    // ---- fallback: per-lane stores straight from the MFMA layout (unaligned / odd-stride outputs)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * WTM + mi * 16 + mlane;
        if (m >= a.Cout) continue;
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (affine) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (m + r < a.Cout) {
                    if (a.scale) sc[r] = a.scale[m + r];
                    if (a.shift) sh[r] = a.shift[m + r];
                }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int nl = wn * WTN + ni * 16 + (lane & 15);
            const int po = t_out[nl];
            if (po < 0) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = acc[mi][ni][r];
                if (affine) u = u * sc[r] + sh[r];
                v[r] = act_fwd(a.act, u);
            }
            if (has_res) {
                const T* rp = (const T*)a.res + (long)t_res[nl] + m;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < a.Cout) v[r] += ElemTraits<T>::to_f32(rp[r]);
            }
            if (out_f32) {
                float* yp = (float*)a.y + (long)po + m;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < a.Cout) yp[r] = accum ? yp[r] + v[r] : v[r];
            } else {
                bf16_t* yp = (bf16_t*)a.y + (long)po + m;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < a.Cout) yp[r] = f32_to_bf16(accum ? bf16_to_f32(yp[r]) + v[r] : v[r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-K across workgroups (DykConvDesc.splitk): the hand-off between the S workgroups of one output tile.
// Protocol = the counter form of cdna_hip_programming.md Guideline 16 R1 / section 5 "in-launch split-K reduction":
//   every slice stores its accumulators into its slab WRITE-THROUGH (global_store ... sc1: no release fence, no L2 write-back),
//   16 bytes per lane in accumulator order (a wave instruction writes 1 KiB contiguous) -> every wave drains vmcnt(0) ->
//   workgroup barrier -> ONE lane takes a relaxed agent-scope ticket on the tile's counter -> the workgroup that draws S - 1
//   is the reducer: it re-arms the counter (all S have arrived; the next launch is stream-ordered behind this one), ONE
//   agent-scope acquire (drops this CU's stale L1 lines), barrier, then plain 16-byte loads of all S slabs IN SLICE ORDER --
//   its own included, so the fp32 sum is the same whichever slice arrives last (bit-reproducible).  The others exit.
// No workgroup waits for another one: correct for any dispatch order, residency and workgroup -> XCD placement.
// Returns true in the reducer (acc = folded tile), false in the workgroups that are done.  NWV = waves holding accumulators
// (all threads with tid < 64 * NWV call it; other waves of the workgroup must have exited), s_flag = one free LDS word.
__device__ inline void st_wt16(void* p, const f32x4_t& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int NWV, int MI, int NI>
__device__ __forceinline__ bool splitk_exchange(f32x4_t (&acc)[MI][NI], const DykConvDesc& a, int tile, int slice, int S, int tid, int* s_flag) {
    constexpr int PER = MI * NI * NWV * 64;                // 16-byte cells per slab (= BM * BN * 4 bytes)
    const int lane = tid & 63, wid = tid >> 6;
    f32x4_t* slab = (f32x4_t*)sgpr_ptr((char*)a.sk_ws) + (size_t)tile * S * PER;
    f32x4_t* mine = slab + (size_t)slice * PER + wid * 64 + lane;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) st_wt16(mine + (mi * NI + ni) * NWV * 64, acc[mi][ni]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave (the asm stores are invisible to hipcc's counters)
    __syncthreads();
    if (tid == 0) {
        uint32_t* cnt = (uint32_t*)sgpr_ptr((char*)a.sk_cnt) + tile;
        const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned)(S - 1) ? 1 : 0;
        if (last) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *s_flag = last;
    }
    __syncthreads();
    const int last = *s_flag;
    __syncthreads();                                       // (the flag word goes back to the epilogue's scratch)
    if (!last) return false;
    const f32x4_t* src = slab + wid * 64 + lane;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < S; ++sl) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            f32x4_t v[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) v[ni] = src[(size_t)sl * PER + (mi * NI + ni) * NWV * 64];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += v[ni];
        }
    }
    return true;
}

// PIPE: LDS-DMA ring stages (2 | 3 | 4 | 6)
// KG:   K-groups per workgroup (1 | 2).  The deep layers (16x20 / 32x40 maps) have ~256 tiles of a long K loop: one
//       4-wave workgroup per CU, every step's barrier / LDS / DMA latency exposed.  With KG = 2 a 512-thread workgroup
//       holds two 4-wave groups that walk the two halves of the input channels with private LDS rings (same barriers);
//       group 1 hands its accumulators to group 0 through LDS before the epilogue -- twice the waves per CU without a
//       split-K pass through memory.  2-stage ring only.
// Waves per SIMD the kernel is compiled for (= its VGPR budget: 168 at three, 256 at two).  Three wherever the LDS footprint
// lets three 4-wave workgroups share a CU; two where LDS allows two workgroups anyway -- there the 168-register cap only
// cost: the ISA of the 128 x 160 / 128-byte-K-step tile re-used ONE register quad for the four weight fragments of the
// second half step (ds_read -> s_waitcnt lgkmcnt(0) -> 5 MFMAs, four times per K step: four exposed LDS latencies).  The
// 128 x 160 tile with the BatchNorm-backward epilogue needs the larger budget in any case (72-209 spilled registers at 168).
template <typename T, int BM, int BN, int BKB, int PIPE, int KG, int EPIK>
constexpr int conv_waves_per_eu() {
    constexpr size_t ring = (size_t)KG * PIPE * (BM + BN) * BKB;
    constexpr size_t sc = (size_t)BN * (BM * sizeof(T) + 16);
    constexpr size_t lds = table_bytes<BN>() + (ring > sc ? ring : sc);
    return (EPIK == 1 && BM * BN >= 128 * 160) || 3 * lds > 160 * 1024 ? 2 : 3;
}
template <typename T, int BM, int BN, int BKB, int PIPE, int KG = 1, int EPIK = 0>
__global__ __launch_bounds__(256 * KG) __attribute__((amdgpu_waves_per_eu(conv_waves_per_eu<T, BM, BN, BKB, PIPE, KG, EPIK>(), conv_waves_per_eu<T, BM, BN, BKB, PIPE, KG, EPIK>())))
void conv_igemm_kernel(const ConvArgs args) {
    static_assert(KG == 1 || PIPE == 2, "K-groups use the 2-stage ring");
    int blk, nblk;
    const DykConvDesc& a = args.d[conv_pick_problem(args, blk, nblk)];
    constexpr int TABLE_BYTES = table_bytes<BN>();
    constexpr int EPV = 16 / (int)sizeof(T);
    constexpr int BK = BKB / (int)sizeof(T);
    constexpr int WM = WaveGrid<BM, BN>::WM;  // waves along M
    constexpr int WN = WaveGrid<BM, BN>::WN;
    constexpr int WTM = BM / WM;
    constexpr int WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int KK = BKB / 64;
    constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB;
    constexpr int NSTAGE = PIPE;

    // LDS: [pixel / tap tables | dummy-DMA sink | stats scratch] then the operand ring, which the
    // epilogue re-uses as the output staging tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* t_in = (int*)smem;                   // [BN] input base offset
    int* t_out = t_in + BN;                   // [BN] output offset or -1
    int* t_res = t_out + BN;                  // [BN] residual offset
    short* t_y = (short*)(t_res + BN);        // [BN] input row of tap (0,0)
    short* t_x = t_y + BN;                    // [BN]
    char* sink = (char*)(t_x + BN);           // [1024] target of dummy LDS-DMA writes (DMA path only)
    int* tap_x = (int*)(sink + 1024);         // [32] activation element offset of a tap: (dy*Wi + dx)*ldx
    int* tap_w = tap_x + 32;                  // [32] weight element offset of a tap: twt*Cout*Cin
    int* tap_dy = tap_w + 32;                 // [32]
    int* tap_dx = tap_dy + 32;                // [32]
    float* s_stat = (float*)(tap_dx + 32);    // [4][2][BM] per-wave channel sums (STATS / BNBWD epilogues)
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    char* sA = smem + TABLE_BYTES + grp * (NSTAGE * (A_BYTES + B_BYTES));   // [NSTAGE][A_BYTES]  (per K-group)
    char* sB = sA + NSTAGE * A_BYTES;              // [NSTAGE][B_BYTES]
    char* sC = smem + TABLE_BYTES;                 // epilogue staging tile (overlays the rings)

    const int tid_all = threadIdx.x;               // tables are built by the first 256 threads
    const int tid = tid_all & 255, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    if ((a.tune >> 18) & 1) return;            // ablation: empty kernel

    // Consecutive remapped ids share an XCD (and its 4 MB L2).  Pixel tiles vary fastest so that an XCD sees few
    // channel tiles: one 128-row slab of 3x3 weights is 9 x heavier than one 128-pixel slab of activations, and
    // the full weight tensor of the deep layers (4.7-18 MB) does not fit one L2.
    const int HWg = a.Hg * a.Wg;
    const int Ntot = a.B * HWg;
    const int tiles_n = (Ntot + BN - 1) / BN;
    int bid = xcd_remap(blk, nblk);
    // split-K across workgroups: the S slices of an output tile have consecutive remapped ids (one XCD, dispatched together)
    const int SK = a.splitk > 1 ? __builtin_amdgcn_readfirstlane(a.splitk) : 1;
    int slice = 0;
    if (SK > 1) { slice = bid % SK; bid /= SK; }
    const int tile_id = bid;
    if (bid >= tiles_n * ((a.Cout + BM - 1) / BM) * (a.ncls > 1 ? a.ncls : 1)) return;    // padding blocks of a two-problem launch
    // output-parity classes of a strided data gradient in one launch: class fastest, so the classes of a pixel tile
    // run side by side on one XCD (shared gradient tile, interleaved output lines merge in its L2)
    int ntaps = a.ntaps, tap0 = 0, ooy = a.ooy, oox = a.oox;
    if (a.ncls > 1) {
        const int c = bid % a.ncls;
        bid /= a.ncls;
        tap0 = a.cls_first[c]; ntaps = a.cls_ntaps[c]; ooy = a.cls_ooy[c]; oox = a.cls_oox[c];
    }
    const int m0 = (bid / tiles_n) * BM;
    const int n0 = (bid % tiles_n) * BN;

    if (tid_all < BN) {
        const int n = n0 + tid;
        if (n < Ntot) {
            const int b = n / HWg;
            const int r = n - b * HWg;
            const int yo = r / a.Wg;
            const int xo = r - yo * a.Wg;
            const int yi = yo * a.isy, xi = xo * a.isx;
            t_in[tid] = ((b * a.Hi + yi) * a.Wi + xi) * a.ldx;
            t_y[tid] = (short)yi;
            t_x[tid] = (short)xi;
            const int py = yo * a.osy + ooy, px = xo * a.osx + oox;
            t_out[tid] = ((b * a.Ho + py) * a.Wo + px) * a.ldy;
            t_res[tid] = ((b * a.Ho + py) * a.Wo + px) * a.ldr;
        } else {
            t_in[tid] = 0; t_y[tid] = -20000; t_x[tid] = -20000; t_out[tid] = -1; t_res[tid] = 0;
        }
    }
    if (tid_all >= 256 - 32 && tid_all < 256 - 32 + ntaps) {   // tap tables in LDS: no vector-memory loads inside the K loop
        const int q = tid - (256 - 32);
        int dy, dx, wt;
        if (args.aff_kw > 0 && a.ncls <= 1) {       // closed form (see ConvArgs): no memory round trip before the first barrier
            const int r = q / args.aff_kw, c = q - r * args.aff_kw;
            dy = args.aff_a0 + args.aff_sy * r; dx = args.aff_b0 + args.aff_sx * c; wt = args.aff_w0 + q;
        } else {
            dy = a.tdy[tap0 + q]; dx = a.tdx[tap0 + q]; wt = a.twt[tap0 + q];
        }
        tap_dy[q] = dy; tap_dx[q] = dx;
        tap_x[q] = (dy * a.Wi + dx) * a.ldx;
        tap_w[q] = wt * a.Cout * a.Cin;
    }
    __syncthreads();
    if ((a.tune >> 19) & 1) return;            // ablation: tables only

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const T* __restrict__ xg = sgpr_ptr((const T*)a.x);
    const T* __restrict__ wg = sgpr_ptr((const T*)a.w);
    // bits 16.. of `tune` are ablation switches for kernel analysis (tools/gpu_probe.py ablate): never set by the plan
    const bool abl_nostore = (a.tune >> 16) & 1, abl_noloop = (a.tune >> 17) & 1;
    const bool abl_nobar = (a.tune >> 22) & 1;      // analysis: K loop without its workgroup barrier (WRONG results: timing only)
    // K-groups split the input-channel chunks; the step count of group 0 (the larger half) drives the common barriers
    const int nchunks_all = a.Cin / BK;
    // this workgroup's share of the input-channel chunks (split-K slice), then the K-groups' halves of it
    const int k_lo = SK > 1 ? (slice * nchunks_all) / SK : 0;
    const int nchunks = SK > 1 ? ((slice + 1) * nchunks_all) / SK - k_lo : nchunks_all;
    const int c_begin = k_lo + ((KG > 1 && grp) ? (nchunks + 1) / 2 : 0);
    const int c_end = k_lo + ((KG > 1 && !grp) ? (nchunks + 1) / 2 : nchunks);
    const int S = abl_noloop ? 0 : (c_end - c_begin) * ntaps;
    const int Smax = abl_noloop ? 0 : (KG > 1 ? ((nchunks + 1) / 2) * ntaps : S);
    const int frow = lane & 15, fslot = lane >> 4;

    // `mid` (the DMA issue of a later step) runs between the first fragment reads and their MFMAs: the ~100 cycles per
    // global_load_lds instruction are then spent while the LDS reads are in flight / the matrix pipe drains, instead
    // of in front of the step with nothing else going on in the wave.
    auto compute = [&](const char* pa, const char* pb, auto&& mid) {
        if constexpr (KK == 2 && conv_waves_per_eu<T, BM, BN, BKB, PIPE, KG, EPIK>() == 2) {
            // 128-byte K step with the 256-register budget: the fragments of BOTH half steps are requested before the first
            // MFMA (round 3: the compiler had re-used one register quad for the weight fragments of the second half step --
            // ds_read, full wait, five MFMAs, four times per K step)
            uint4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa0[mi] = *(const uint4*)(pa + lds_off<BKB>(wm * WTM + mi * 16 + frow, fslot));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fb0[ni] = *(const uint4*)(pb + lds_off<BKB>(wn * WTN + ni * 16 + frow, fslot));
            mid();
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa1[mi] = *(const uint4*)(pa + lds_off<BKB>(wm * WTM + mi * 16 + frow, 4 + fslot));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fb1[ni] = *(const uint4*)(pb + lds_off<BKB>(wn * WTN + ni * 16 + frow, 4 + fslot));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], fa0[mi], fb0[ni]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], fa1[mi], fb1[ni]);
            // nothing may sink below this point: the loop's `s_waitcnt vmcnt(..)` follows, and an MFMA block scheduled behind it
            // would wait for the NEXT step's DMA before computing this one (seen in the ISA without the fence)
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            uint4 fa[MI], fb[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                fa[mi] = *(const uint4*)(pa + lds_off<BKB>(wm * WTM + mi * 16 + frow, kk * 4 + fslot));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                fb[ni] = *(const uint4*)(pb + lds_off<BKB>(wn * WTN + ni * 16 + frow, kk * 4 + fslot));
            if (kk == 0) mid();
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], fa[mi], fb[ni]);
        }
        // (see above: without the fence the compiler sinks the MFMA block below the loop's `s_waitcnt vmcnt(0)` -- ISA of the
        // 64-byte-K-step tiles: 19 of the 20 MFMAs of a step waited for the NEXT step's DMA)
        __builtin_amdgcn_sched_barrier(0);
    };

    {
        // ---- LDS-DMA pipeline: global_load_lds (16 B per lane, 1 KiB per wave instruction) straight
        // into a 3-deep ring of swizzled tiles; step s+2 is in flight while step s feeds the MFMAs.
        // A wave instruction fills RPI consecutive tile rows; lane -> (row, physical slot); the XOR
        // swizzle is applied on the SOURCE side (the LDS image of an LDS-DMA is lane-linear).
        constexpr int SPR = BKB / 16;                  // 16-byte slots per tile row
        constexpr int RPI = 64 / SPR;                  // tile rows per wave instruction
        constexpr int NI_A = BM * BKB / 1024, NI_B = BN * BKB / 1024;
        constexpr int NIA_W = (NI_A + 3) / 4, NIB_W = (NI_B + 3) / 4;
        constexpr int NPW = NIA_W + NIB_W;             // DMA instructions per wave per step (uniform count)
        const int wv = __builtin_amdgcn_readfirstlane(wid);
        const int lrow = lane / SPR, pslot = lane % SPR;
        int a_off[NIA_W]; bool a_ok[NIA_W];
#pragma unroll
        for (int j = 0; j < NIA_W; ++j) {
            const int inst = j * 4 + wv;
            const int row = inst * RPI + lrow;
            const int co = m0 + row;
            const int ls = (lds_off<BKB>(row, pslot) - row * BKB) >> 4;    // logical slot stored at this physical slot
            a_ok[j] = (inst < NI_A) && (co < a.Cout);
            a_off[j] = co * a.Cin + ls * EPV;
        }
        int b_off[NIB_W];
        int b_y0[NIB_W], b_x0[NIB_W];
#pragma unroll
        for (int j = 0; j < NIB_W; ++j) {
            const int inst = j * 4 + wv;
            const int row = inst * RPI + lrow;
            const int ls = (lds_off<BKB>(row, pslot) - row * BKB) >> 4;
            const bool live = (NI_B % 4 == 0) || inst < NI_B;
            b_off[j] = live ? t_in[row] + ls * EPV : 0;
            b_y0[j] = live ? t_y[row] : -20000; b_x0[j] = live ? t_x[row] : -20000;
        }
        const T* zero = (const T*)dyk_zero_page;
        // Image extents in SGPRs for the whole loop.  Read through `a` inside the loop they were RE-LOADED from the kernel
        // arguments (s_load_dword + s_waitcnt lgkmcnt(0)) in front of every activation-row DMA -- the LDS-DMA asm statements
        // clobber "memory", so the compiler does not keep descriptor fields across them -- and the short-circuit && of the
        // in-image test became two exec-mask branches per DMA (ISA of the 128 x 160 tile: 6 scalar loads, 5 branches and 17
        // waits around the 9 DMA instructions of a K step).
        const unsigned Hi_u = (unsigned)__builtin_amdgcn_readfirstlane(a.Hi), Wi_u = (unsigned)__builtin_amdgcn_readfirstlane(a.Wi);
        const unsigned sA_u = lds_addr_of(sA), sB_u = lds_addr_of(sB), sink_u = lds_addr_of(sink);
        auto stage = [&](int buf, int c0, int t) {
            const int toff = tap_x[t] + c0;
            const int tdy_t = tap_dy[t], tdx_t = tap_dx[t];
            const long wbase = (long)tap_w[t] + c0;
            // (LDS destinations as plain integers: a generic -> LDS pointer cast per DMA costs a null check, 4 SALU instructions)
            const unsigned da = sA_u + buf * A_BYTES, db = sB_u + buf * B_BYTES;
#pragma unroll
            for (int j = 0; j < NIA_W; ++j) {
                const int inst = j * 4 + wv;
                if (NI_A % 4 == 0 || inst < NI_A) {
                    const T* src = a_ok[j] ? wg + wbase + a_off[j] : zero;
                    glds16(src, da + inst * 1024);
                } else {
                    // keep the per-wave DMA count uniform so that one counted vmcnt fits all waves
                    glds16(zero, sink_u);
                }
            }
#pragma unroll
            for (int j = 0; j < NIB_W; ++j) {
                const int inst = j * 4 + wv;
                // in-image test of this row under tap t: five VALU ops in the shadow of the step's MFMAs (a per-row tap
                // bitmask built in the prologue cost ~400 exposed instructions per workgroup on a 3x3 conv)
                const bool ok = ((unsigned)(b_y0[j] + tdy_t) < Hi_u) & ((unsigned)(b_x0[j] + tdx_t) < Wi_u);
                const T* src = ok ? xg + (long)b_off[j] + toff : zero;
                glds16(src, (NI_B % 4 == 0 || inst < NI_B) ? db + inst * 1024 : sink_u);
            }
        };
        // staging iterator (runs two steps ahead of the compute iterator)
        int sc0 = c_begin * BK, st = 0;
        auto stage_next = [&](int buf) {
            stage(buf, sc0, st);
            if (++st == ntaps) { st = 0; sc0 += BK; }
        };
        if constexpr (PIPE >= 3) {
            // N-stage ring: AHEAD = N-1 steps are in flight while one is computed.  Workgroups that sit alone on a
            // CU (deep layers: few, long-K tiles) need the depth: with two stages a K step costs one L2 round trip.
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
                compute(sA + cur * A_BYTES, sB + cur * B_BYTES, [&]() { if (more) stage_next(nxt); });
                if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!abl_nobar) __builtin_amdgcn_s_barrier();
                cur = (cur == PIPE - 1) ? 0 : cur + 1;
                nxt = (nxt == PIPE - 1) ? 0 : nxt + 1;
            }
        } else {
            // 2-stage ring for short K loops (1x1 convs, small Cin): half the LDS, 2-3x the resident
            // workgroups per CU -- latency is hidden across workgroups instead of inside one
            if (S > 0) stage_next(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int s = 0; s < Smax; ++s) {
                if (KG == 1 || s < S)
                    compute(sA + (s & 1) * A_BYTES, sB + (s & 1) * B_BYTES, [&]() { if (s + 1 < S) stage_next((s + 1) & 1); });
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!abl_nobar) __builtin_amdgcn_s_barrier();
            }
        }
    }

    if constexpr (KG > 1) {
        // fold the K-groups: group 1 parks its accumulators in LDS (lane-linear float4, conflict free), group 0 adds them
        float4* park = (float4*)(smem + TABLE_BYTES);      // overlays the rings: every wave is behind the loop's last barrier
        if (grp == 1) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    park[((mi * NI + ni) * 4 + wid) * 64 + lane] = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
        }
        __syncthreads();
        if (grp != 0) return;                              // (ended waves no longer count at the barriers below)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const float4 v = park[((mi * NI + ni) * 4 + wid) * 64 + lane];
                acc[mi][ni][0] += v.x; acc[mi][ni][1] += v.y; acc[mi][ni][2] += v.z; acc[mi][ni][3] += v.w;
            }
        __syncthreads();                                   // the epilogue's staging tile overlays the park area
    }

    if constexpr (EPIK != 3) {
        if (SK > 1) {
            if (KG == 1) __syncthreads();                  // every wave is done with the operand ring (s_stat's first word is the flag)
            if (!splitk_exchange<4, MI, NI>(acc, a, tile_id, slice, SK, tid, (int*)s_stat)) return;
        }
    }
    // ------------------------------------------------------------------ epilogue
    // (split-K: the statistics replica is chosen by the TILE, not by whichever slice's workgroup happened to arrive last)
    conv_epilogue<T, BM, BN, EPIK>(a, acc, sC, s_stat, t_out, t_res, m0, SK > 1 ? tile_id : blk, n0 == 0, (unsigned)nblk);
}

// ======================================================================================
// 3x3 stride-1 convolution with a HALO tile.  The generic kernel above re-stages the activation tile for each of the
// nine taps, which makes L2->LDS bytes (64 flop per byte for a 128x128 tile) its limiter.  Here a workgroup owns a
// TH x 20 pixel patch of one image (all feature maps of this path are multiples of 20 wide), stages the patch with a
// one-pixel halo ONCE per 32/64-channel chunk (double buffered, loaded piecewise during the nine tap steps of the
// previous chunk) and reads the nine taps as row-shifted views of it; only the weight tiles stream per step
// (3-stage ring).  Bytes per step drop from (BM + BN) to (BM + 1.4 BN / 9) rows.
// Same descriptor, same epilogue; requires ntaps == 9 with tap offsets in [-1,1]^2, unit strides, bf16.
template <int TH> struct HaloGeom {
    static constexpr int TW = 20, HW = TW + 2, BN = TH * TW, HR = (TH + 2) * HW;
};
template <int BM, int TH> constexpr int halo_table_bytes() {
    return HaloGeom<TH>::BN * 8 + 1024 + 2 * 32 * 4 + 4 * 2 * 128 * 4;
}

template <typename T, int BM, int TH, int BKB, int EPIK = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EPIK == 1 && BM * HaloGeom<TH>::BN >= 128 * 160 ? 2 : 3)))
void conv_halo_kernel(const ConvArgs args) {
    int blk, nblk;
    const DykConvDesc& a = args.d[conv_pick_problem(args, blk, nblk)];
    using G = HaloGeom<TH>;
    constexpr int TW = G::TW, HW = G::HW, BN = G::BN, HR = G::HR;
    constexpr int EPV = 16 / (int)sizeof(T);
    constexpr int BK = BKB / (int)sizeof(T);
    constexpr int WM = WaveGrid<BM, BN>::WM, WN = WaveGrid<BM, BN>::WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int KK = BKB / 64;
    constexpr int A_BYTES = BM * BKB;
    constexpr int SPR = BKB / 16, RPI = 64 / SPR;          // slots per row, rows per DMA instruction
    constexpr int NB = (HR + RPI - 1) / RPI;               // DMA instructions per halo tile
    constexpr int HB_BYTES = NB * 1024;
    constexpr int NA = 3;                                  // weight ring stages
    constexpr int NI_A = BM * BKB / 1024;
    constexpr int NIA_W = (NI_A + 3) / 4;
    constexpr int NPW = NIA_W + 1;                         // DMA instructions per wave per step: weights + one halo piece
    static_assert(NB <= 32, "halo tile must be loadable in 8 steps x 4 waves");
    constexpr int TABLE_BYTES = halo_table_bytes<BM, TH>();

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* t_out = (int*)smem;                  // [BN]
    int* t_res = t_out + BN;                  // [BN]
    char* sink = (char*)(t_res + BN);         // [1024]
    int* tap_w = (int*)(sink + 1024);         // [32] weight element offset of a tap
    int* tap_h = tap_w + 32;                  // [32] halo-row offset of a tap: dy*HW + dx
    float* s_stat = (float*)(tap_h + 32);     // [4][2][BM]
    char* sA = smem + TABLE_BYTES;            // [NA][A_BYTES]
    char* sH = sA + NA * A_BYTES;             // [2][HB_BYTES]
    int* t_hsrc = (int*)(sH + 2 * HB_BYTES);  // [NB*64] per (instruction, lane): source element offset or -1 (zero page)
    char* sC = sA;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tiles_x = a.Wi / TW, tiles_y = a.Hi / TH;
    const int tiles_n = a.B * tiles_y * tiles_x;
    const int bid = xcd_remap(blk, nblk);
    if (bid >= tiles_n * ((a.Cout + BM - 1) / BM)) return;     // padding blocks of a two-problem launch
    const int m0 = (bid / tiles_n) * BM;
    int nt = bid % tiles_n;
    const int bimg = nt / (tiles_y * tiles_x);
    nt -= bimg * (tiles_y * tiles_x);
    const int ty0 = (nt / tiles_x) * TH, tx0 = (nt % tiles_x) * TW;

    if (tid < BN) {
        const int py = tid / TW, px = tid - py * TW;
        const int pix = (bimg * a.Ho + ty0 + py) * a.Wo + tx0 + px;
        t_out[tid] = pix * a.ldy;
        t_res[tid] = pix * a.ldr;
    }
    if (tid >= 256 - 32 && tid < 256 - 32 + a.ntaps) {
        const int q = tid - (256 - 32);
        int dy, dx, wt;
        if (args.aff_kw > 0) {
            const int r = q / args.aff_kw, c = q - r * args.aff_kw;
            dy = args.aff_a0 + args.aff_sy * r; dx = args.aff_b0 + args.aff_sx * c; wt = args.aff_w0 + q;
        } else {
            dy = a.tdy[q]; dx = a.tdx[q]; wt = a.twt[q];
        }
        tap_h[q] = dy * HW + dx;
        tap_w[q] = wt * a.Cout * a.Cin;
    }
    for (int e = tid; e < NB * 64; e += 256) {
        const int q = e >> 6, l = e & 63;
        const int row = q * RPI + l / SPR, pslot = l % SPR;
        int src = -1;
        if (row < HR) {
            const int hy = row / HW, hx = row - hy * HW;
            const int y = ty0 + hy - 1, x = tx0 + hx - 1;
            if ((unsigned)y < (unsigned)a.Hi && (unsigned)x < (unsigned)a.Wi) {
                const int ls = (lds_off<BKB>(row, pslot) - row * BKB) >> 4;
                src = ((bimg * a.Hi + y) * a.Wi + x) * a.ldx + ls * EPV;
            }
        }
        t_hsrc[e] = src;
    }
    __syncthreads();

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const T* __restrict__ xg = (const T*)a.x;
    const T* __restrict__ wg = (const T*)a.w;
    const T* zero = (const T*)dyk_zero_page;
    const int nchunk = a.Cin / BK;
    const int S = nchunk * 9;
    const int frow = lane & 15, fslot = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const int lrow = lane / SPR, pslot = lane % SPR;

    int hrow[NI];                              // halo row of this lane's pixel of fragment ni, tap (0,0)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int p = wn * WTN + ni * 16 + frow;
        const int py = p / TW;
        hrow[ni] = (py + 1) * HW + (p - py * TW) + 1;
    }
    int a_off[NIA_W]; bool a_ok[NIA_W];
#pragma unroll
    for (int j = 0; j < NIA_W; ++j) {
        const int inst = j * 4 + wv;
        const int row = inst * RPI + lrow;
        const int co = m0 + row;
        const int ls = (lds_off<BKB>(row, pslot) - row * BKB) >> 4;
        a_ok[j] = (inst < NI_A) && (co < a.Cout);
        a_off[j] = co * a.Cin + ls * EPV;
    }
    auto stage_a = [&](int buf, int c0, int t) {
        const long wbase = (long)tap_w[t] + c0;
        char* da = sA + buf * A_BYTES;
#pragma unroll
        for (int j = 0; j < NIA_W; ++j) {
            const int inst = j * 4 + wv;
            if (NI_A % 4 == 0 || inst < NI_A) {
                const T* src = a_ok[j] ? wg + wbase + a_off[j] : zero;
                glds16(src, lds_addr_of(da + inst * 1024));
            } else {
                glds16(zero, lds_addr_of(sink));
            }
        }
    };
    // one halo DMA instruction: piece q of the tile of channel chunk c0 into buffer hb (q >= NB: dummy into the sink)
    auto stage_h = [&](int hb, int c0, int q) {
        if (q < NB) {
            const int so = t_hsrc[q * 64 + lane];
            const T* src = so >= 0 ? xg + (long)so + c0 : zero;
            glds16(src, lds_addr_of(sH + hb * HB_BYTES + q * 1024));
        } else {
            glds16(zero, lds_addr_of(sink));
        }
    };
    auto compute = [&](const char* pa, const char* ph, int toff) {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            uint4 fa[MI], fb[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                fa[mi] = *(const uint4*)(pa + lds_off<BKB>(wm * WTM + mi * 16 + frow, kk * 4 + fslot));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                fb[ni] = *(const uint4*)(ph + lds_off<BKB>(hrow[ni] + toff, kk * 4 + fslot));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], fa[mi], fb[ni]);
        }
        __builtin_amdgcn_sched_barrier(0);     // (the MFMA block must not sink below the loop's s_waitcnt vmcnt: see conv_igemm_kernel)
    };

    // prologue: whole halo tile of chunk 0, weight tiles of steps 0 and 1
    for (int q = wv; q < ((NB + 3) / 4) * 4; q += 4) stage_h(0, 0, q);
    int sc0 = 0, st = 0;                       // weight staging iterator (two steps ahead)
    auto stage_a_next = [&](int buf) {
        stage_a(buf, sc0, st);
        if (++st == 9) { st = 0; sc0 += BK; }
    };
    if (S > 0) stage_a_next(0);
    if (S > 1) stage_a_next(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0, nxt = 2, c = 0, t = 0;
    for (int s = 0; s < S; ++s) {
        const bool more = (s + 2 < S);
        if (more) {
            stage_a_next(nxt);
            // halo of the next chunk: one piece per wave per step during tap steps 0..7 of this chunk
            const bool piece = (c + 1 < nchunk) && t < 8;
            stage_h((c + 1) & 1, (c + 1) * BK, piece ? t * 4 + wv : NB);
        }
        compute(sA + cur * A_BYTES, sH + (c & 1) * HB_BYTES, tap_h[t]);
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = (cur == NA - 1) ? 0 : cur + 1;
        nxt = (nxt == NA - 1) ? 0 : nxt + 1;
        if (++t == 9) { t = 0; ++c; }
    }
    conv_epilogue<T, BM, BN, EPIK>(a, acc, sC, s_stat, t_out, t_res, m0, blk);
}

// fills the kernel arguments of a single- or two-problem launch; returns the grid size.  `vec`: the staged (16-byte
// vector) epilogue is possible for every problem of the launch
inline unsigned conv_fill_args(ConvArgs& args, const DykConvDesc* d, int tiles, bool vec) {
    args.aff_kw = 0;
    for (int kw = 1; kw <= 7 && !args.aff_kw && d->ntaps > 0 && d->ncls <= 1; kw += 2) {
        if (d->ntaps % kw) continue;
        const int a0 = d->tdy[0], b0 = d->tdx[0], w0 = d->twt[0];
        const int sy = d->ntaps > kw ? d->tdy[kw] - a0 : 0, sx = kw > 1 ? d->tdx[1] - b0 : 0;
        bool ok = true;
        for (int q = 0; q < d->ntaps && ok; ++q)
            ok = d->tdy[q] == a0 + sy * (q / kw) && d->tdx[q] == b0 + sx * (q % kw) && d->twt[q] == w0 + q;
        if (ok) { args.aff_kw = kw; args.aff_a0 = a0; args.aff_sy = sy; args.aff_b0 = b0; args.aff_sx = sx; args.aff_w0 = w0; }
    }
    args.d[0] = *d;
    args.d[0].twin = nullptr;
    args.pair_tiles = 0;
    unsigned grid = (unsigned)tiles;
    if (d->twin) {
        args.d[1] = *d->twin;
        args.d[1].twin = nullptr;
        args.pair_tiles = (tiles + 7) & ~7;
        grid = 2u * (unsigned)args.pair_tiles;
    }
    for (int q = 0; q < (d->twin ? 2 : 1); ++q) {
        if (vec) args.d[q].flags |= EPI_INTERNAL_VEC;
        else args.d[q].flags &= ~EPI_INTERNAL_VEC;
    }
    return grid;
}
// split count of a launch (validated by the front end: no twin, one parity class, scratch present)
inline int conv_splitk_of(const DykConvDesc* d) { return d->splitk > 1 ? d->splitk : 1; }
// can the staged epilogue store 16-byte vectors for this problem?
inline bool conv_vec_ok(const DykConvDesc* d, int eso, size_t elem) {
    bool vec = ((size_t)d->ldy * eso) % 16 == 0 && ((uintptr_t)d->y % 16) == 0;
    if ((d->flags & DYK_EPI_RESIDUAL) && (((size_t)d->ldr * elem) % 16 != 0 || ((uintptr_t)d->res % 16) != 0)) vec = false;
    return vec;
}

inline bool conv_halo_eligible(const DykConvDesc* d, int TH) {
    if (d->dtype != DYK_BF16 || d->ntaps != 9 || d->splitk > 1) return false;      // (split-K: generic / large-tile kernels only)
    if (d->isy != 1 || d->isx != 1 || d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0) return false;
    if (d->Hg != d->Hi || d->Wg != d->Wi || d->Ho != d->Hi || d->Wo != d->Wi) return false;
    if (d->Wi % 20 || d->Hi % TH) return false;
    for (int q = 0; q < 9; ++q)
        if (d->tdy[q] < -1 || d->tdy[q] > 1 || d->tdx[q] < -1 || d->tdx[q] > 1) return false;
    return true;
}

template <typename T, int BM, int TH, int BKB, int EPIK = 0>
int launch_conv_halo(const DykConvDesc* d, hipStream_t stream) {
    using G = HaloGeom<TH>;
    constexpr int BN = G::BN;
    constexpr int RPI = 64 / (BKB / 16);
    constexpr int NB = (G::HR + RPI - 1) / RPI;
    constexpr size_t ring = 3 * (size_t)BM * BKB + 2 * (size_t)NB * 1024;
    constexpr size_t hsrc = (size_t)NB * 64 * 4;
    const bool of32 = (d->flags & DYK_EPI_OUT_F32) != 0;
    size_t stage_c = (size_t)BN * (BM * (of32 ? 4 : 2) + 16);
    if constexpr (EPIK == 1) {
        const size_t need = bnbwd_sliced_bytes<T, BM, WaveGrid<BM, BN>::WN>((d->flags & DYK_EPI_ADDEND) != 0);
        if (need > stage_c) stage_c = need;
    }
    const size_t body = ring + hsrc > stage_c ? ring + hsrc : stage_c;
    const size_t lds = halo_table_bytes<BM, TH>() + body;
    static DykDeviceOnce attr_set;   // (the attribute is per DEVICE: one flag per device id)
    auto kfn = conv_halo_kernel<T, BM, TH, BKB, EPIK>;
    if (attr_set.first()) {
        constexpr size_t sc_max = (size_t)BN * (BM * 4 + 16) > bnbwd_sliced_bytes<T, BM, WaveGrid<BM, BN>::WN>(true)
                                      ? (size_t)BN * (BM * 4 + 16) : bnbwd_sliced_bytes<T, BM, WaveGrid<BM, BN>::WN>(true);
        constexpr size_t lds_max = halo_table_bytes<BM, TH>() + (ring + hsrc > sc_max ? ring + hsrc : sc_max);
        DYK_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    }
    const int tiles_n = d->B * (d->Hi / TH) * (d->Wi / 20);
    const int tiles_m = dyk_div_up(d->Cout, BM);
    ConvArgs args;
    const int eso = of32 ? 4 : 2;
    const bool vec = conv_vec_ok(d, eso, sizeof(T)) && (!d->twin || conv_vec_ok(d->twin, eso, sizeof(T)));
    const unsigned grid = conv_fill_args(args, d, tiles_n * tiles_m, vec);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, args);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

// pixel-tile codes 3 / 4 of the tune word: halo kernel with 4x20 / 8x20 pixel patches
template <typename T, int TH, int EPIK = 0>
int dispatch_conv_halo(const DykConvDesc* d, hipStream_t stream) {
    if (!conv_halo_eligible(d, TH)) return DYK_ERR_UNSUPPORTED;
    const int row_bytes = d->Cin * (int)sizeof(T);
    if (row_bytes % 64) return DYK_ERR_ARG;
    const bool k128 = (d->tune & 0xff) == 128 && row_bytes % 128 == 0;
    int bm = d->Cout > 64 ? 128 : 64;
    const int bm_code = (d->tune >> 24) & 0xf;
    if (bm_code == 2) bm = 64;
    if (bm_code == 3) bm = 128;
    if constexpr (TH == 8) {
        // 32-channel tiles (8 x 20 patches only: a wave needs 16 channel rows): the data gradients into 32-channel tensors
        // at 256 x 320 (3x3 32 -> 64 of the first CSP block) spend a 64-row tile half on zero rows
        if (d->Cout <= 32 && bm_code != 2 && bm_code != 3)
            return k128 ? launch_conv_halo<T, 32, TH, 128, EPIK>(d, stream) : launch_conv_halo<T, 32, TH, 64, EPIK>(d, stream);
    }
    if (k128) return bm == 128 ? launch_conv_halo<T, 128, TH, 128, EPIK>(d, stream) : launch_conv_halo<T, 64, TH, 128, EPIK>(d, stream);
    return bm == 128 ? launch_conv_halo<T, 128, TH, 64, EPIK>(d, stream) : launch_conv_halo<T, 64, TH, 64, EPIK>(d, stream);
}

template <typename T, int BM, int BN, int BKB, int PIPE, int KG = 1, int EPIK = 0>
int launch_conv_impl(const DykConvDesc* d, hipStream_t stream) {
    constexpr int TABLE_BYTES = table_bytes<BN>();
    constexpr size_t ring = KG > 1 ? (size_t)KG * PIPE * (BM + BN) * BKB + 0 : PIPE * (size_t)(BM + BN) * BKB;   // K-groups: private rings; the 16 KiB-per-wave park area (BM*BN*4) fits inside
    static_assert(KG == 1 || ring >= (size_t)BM * BN * 4, "park area");
    static_assert(table_bytes<BN>() + ring <= 160 * 1024 || KG == 1, "LDS budget");
    const bool of32 = (d->flags & DYK_EPI_OUT_F32) || sizeof(T) == 4;
    size_t stage_c = (size_t)BN * (BM * (of32 ? 4 : 2) + 16);
    if constexpr (EPIK == 1) {
        const size_t need = bnbwd_sliced_bytes<T, BM, WaveGrid<BM, BN>::WN>((d->flags & DYK_EPI_ADDEND) != 0);
        if (need > stage_c) stage_c = need;
    }
    const size_t lds = TABLE_BYTES + (ring > stage_c ? ring : stage_c);
    static DykDeviceOnce attr_set;   // (the attribute is per DEVICE: one flag per device id)
    auto kfn = conv_igemm_kernel<T, BM, BN, BKB, PIPE, KG, EPIK>;
    if (attr_set.first()) {
        constexpr size_t sc_max = (size_t)BN * (BM * 4 + 16) > bnbwd_sliced_bytes<T, BM, WaveGrid<BM, BN>::WN>(true)
                                      ? (size_t)BN * (BM * 4 + 16) : bnbwd_sliced_bytes<T, BM, WaveGrid<BM, BN>::WN>(true);
        constexpr size_t lds_max = TABLE_BYTES + (ring > sc_max ? ring : sc_max);
        DYK_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    }
    const long Ntot = (long)d->B * d->Hg * d->Wg;
    const int tiles_n = dyk_div_up(Ntot, BN);
    const int tiles_m = dyk_div_up(d->Cout, BM);
    ConvArgs args;
    // the staged epilogue needs 16-byte aligned pixel rows of the output (and residual)
    const int eso = of32 ? 4 : 2;
    bool vec = conv_vec_ok(d, eso, sizeof(T)) && (!d->twin || conv_vec_ok(d->twin, eso, sizeof(T)));
    const int sk = conv_splitk_of(d);
    if (sk > 1) {
        if constexpr (EPIK == 3) return DYK_ERR_UNSUPPORTED;
        if (tiles_n * tiles_m > d->sk_cnt_n || (int64_t)tiles_n * tiles_m * sk * BM * BN * 4 > d->sk_ws_bytes) return DYK_ERR_ARG;
    }
    const unsigned grid = conv_fill_args(args, d, tiles_n * tiles_m * (d->ncls > 1 ? d->ncls : 1) * sk, vec);
    if constexpr (EPIK == 3) {
        // every workgroup must be resident while the launch waits on its arrival counter: bounded grid, vector epilogue, one problem
        if (grid > (unsigned)DYK_BNFWD_MAX_GRID || !vec || d->twin || d->ncls > 1 || lds > 80 * 1024) return DYK_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256 * KG), lds, stream, args);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

template <typename T, int BM, int BN, int BKB, int EPIK = 0>
int launch_conv(const DykConvDesc* d, hipStream_t stream) {
    // 2-stage ring by default: 2-4 resident workgroups per CU beat a deeper ring unless the autotuner says otherwise
    switch ((d->tune >> 8) & 0xf) {
    case 3: return launch_conv_impl<T, BM, BN, BKB, 3, 1, EPIK>(d, stream);
    case 4: return launch_conv_impl<T, BM, BN, BKB, 4, 1, EPIK>(d, stream);
    case 6: if constexpr ((size_t)6 * (BM + BN) * BKB + table_bytes<BN>() <= 160 * 1024) return launch_conv_impl<T, BM, BN, BKB, 6, 1, EPIK>(d, stream);
            else return launch_conv_impl<T, BM, BN, BKB, 4, 1, EPIK>(d, stream);
    default: return launch_conv_impl<T, BM, BN, BKB, 2, 1, EPIK>(d, stream);
    }
}

// tune word: bits 0..7 K-step bytes (64 | 128), 8..11 ring stages (2 | 3 | 4 | 6), 12..15 pixel tile (0 = 128, 1 = 80, 2 = 160),
// 24..27 channel tile (0 = by Cout, 1 = 32, 2 = 64, 3 = 128); bits 16..23 are analysis switches
template <typename T, int BN, int EPIK = 0>
int dispatch_conv_bn(const DykConvDesc* d, hipStream_t stream) {
    const int row_bytes = d->Cin * (int)sizeof(T);
    if ((row_bytes % 64) != 0) return DYK_ERR_ARG;
    // default K step: 128 bytes for long loops over wide inputs, else 64 (more resident workgroups); see DESIGN.md
    bool k128 = (row_bytes % 128) == 0 && (row_bytes / 128) * d->ntaps > 4 && d->Cin >= 512;
    if ((d->tune & 0xff) == 64) k128 = false;
    if ((d->tune & 0xff) == 128 && (row_bytes % 128) == 0) k128 = true;
    int bm = d->Cout > 64 ? 128 : (d->Cout > 32 ? 64 : 32);
    const int bm_code = (d->tune >> 24) & 0xf;
    if (bm_code >= 1 && bm_code <= 3) bm = 16 << bm_code;
    if constexpr (BN == 80) { if (bm == 32) bm = 64; }          // a wave needs 16 channel rows
    if (k128) {
        if (bm == 128) return launch_conv<T, 128, BN, 128, EPIK>(d, stream);
        if (bm == 64) return launch_conv<T, 64, BN, 128, EPIK>(d, stream);
        if constexpr (BN != 80) return launch_conv<T, 32, BN, 128, EPIK>(d, stream);
    }
    if (bm == 128) return launch_conv<T, 128, BN, 64, EPIK>(d, stream);
    if (bm == 64) return launch_conv<T, 64, BN, 64, EPIK>(d, stream);
    if constexpr (BN != 80) return launch_conv<T, 32, BN, 64, EPIK>(d, stream);
    return DYK_ERR_UNSUPPORTED;
}

}  // namespace
