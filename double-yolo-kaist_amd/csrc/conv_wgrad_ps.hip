// Pixel-streaming weight gradient of the 1x1 convolutions on the matrix cores (bf16), round 6:
//     dw[co][ci] += sum_p dy[p][co] * x[p][ci]                 (p = (image, row, column): ONE linear pixel index)
// A 1x1 / stride-1 weight gradient is a GEMM whose K dimension is the pixel index of BOTH operands and whose operands are
// 20-500 MB against 16 KB - 2 MB of result: HBM / L2 streaming work (64-340 flop per byte, the chip balances at 312).  The
// per-tap kernel (conv_wgrad.hip) ran these layers at 1.1-1.7 TB/s on 40-80 workgroups (r05_cmd_roofline_c3_full.txt: 64x80
// 128>128 24 us for a 5 us problem, 16x20 1024>512 28 us for 2 us): a 4-wave workgroup with a 2-stage ring has ONE 32 KB
// stage in flight and drains it (vmcnt(0)) every step -- 32 KB per memory round trip = ~43 GB/s per workgroup, and more
// workgroups mean more partial planes to fold.  This kernel raises the bytes a workgroup keeps in flight instead:
//   * a workgroup is 8 waves = 2 K-halves x (2 x 2) waves over a BM x BN tile of dw (BM, BN in {64, 128}); ALL eight waves
//     issue the LDS-DMA of a stage (KR = 64 or 128 pixels of dy[.., BM] and x[.., BN], 16-32 KB) into a ring of up to 8
//     stages (<= 128 KB) with counted `s_waitcnt vmcnt(N)`: up to 96 KB per workgroup in flight, one barrier per stage;
//   * 1x1 geometry: the pixel index is linear, a lane's source address advances by KR rows per stage (one 64-bit add per DMA
//     instruction; no tap table, no division, no in-image test); ragged ends read the zero page;
//   * K-half g computes pixel rows [g KR/2, (g+1) KR/2) of every stage (same fragment reads per MFMA as a 4-wave workgroup,
//     twice the waves to hide the LDS latency); at the end the halves are exchanged through LDS SYMMETRICALLY -- each half
//     hands over the fragments the other one finishes -- so all eight waves add and all eight store (a + b == b + a: the sum
//     does not depend on which half finishes a fragment);
//   * results: plane mode (one plane per K split, folded by dyk_grad_reduce), in-launch fold (DykWgradDesc.sk_cnt: the S
//     slices of a tile hand their fragments over through write-through slabs and a ticket, the last arriver adds them IN SLICE
//     ORDER and read-add-writes dw: bit-reproducible, no planes in the gradient-reduce pass), single split / atomics.
// Replaces autograd's convolution_backward (weight gradient) for the 1x1 nn.Conv2d of reference models.py:34-42.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "conv_wgrad_tile.h"

namespace {

using T = bf16_t;

template <int BM, int BN, int KR, int NS> struct PsCfg {
    static constexpr int A_BYTES = KR * BM * 2, B_BYTES = KR * BN * 2, STAGE = A_BYTES + B_BYTES;
    static constexpr int NI_A = A_BYTES / 1024, NI_B = B_BYTES / 1024;      // DMA wave instructions per stage and tile
    static constexpr int NPW = (NI_A + NI_B) / 8;                           // per wave
    static constexpr int NPW_A = NI_A / 8;
    static constexpr int MI = BM / 32, NI = BN / 32, MH = MI / 2;           // 16 x 16 fragments per wave; MH: finished per K-half
    static constexpr int PARK = 2 * MH * NI * 4 * 64 * 16;                  // exchange buffer of the two K-halves
    static constexpr int RING = NS * STAGE;
    static constexpr int LDS = RING > PARK ? RING : PARK;
    static_assert(NI_A % 8 == 0 && NI_B % 8 == 0, "every wave issues whole instructions of one tile");
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert((NS - 2) * NPW <= 63, "vmcnt range");
};

__device__ inline void ps_st_wt16(void* p, const f32x4_t& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int BM, int BN, int KR, int NS>
__global__ __launch_bounds__(512) void conv_wgrad_ps_kernel(const DykWgradDesc a, const int splits, const int chunk) {
    using C = PsCfg<BM, BN, KR, NS>;
    constexpr int VPR_A = BM / 8, VPR_B = BN / 8;                 // 16-byte vectors per tile row
    constexpr int RPI_A = 64 / VPR_A, RPI_B = 64 / VPR_B;         // tile rows per DMA wave instruction
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int MI = C::MI, NI = C::NI, MH = C::MH;
    constexpr int HALF = KR / 2;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wid >> 2, wm = (wid >> 1) & 1, wn = wid & 1, w4 = wid & 3;

    const int Cout_s = __builtin_amdgcn_readfirstlane(a.Cout), Cin_s = __builtin_amdgcn_readfirstlane(a.Cin);
    const int lddy_s = __builtin_amdgcn_readfirstlane(a.lddy), ldx_s = __builtin_amdgcn_readfirstlane(a.ldx);
    const int tiles_m = (Cout_s + BM - 1) / BM;
    const int tiles_n = (Cin_s + BN - 1) / BN;
    // consecutive remapped ids run on one XCD: the tiles of one pixel range share that XCD's L2
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    // grouped launch: problem p owns blocks [p per, (p + 1) per); its tensors come out of the device table (scalar loads)
    const void *x_p = a.x, *dy_p = a.dy;
    float *dw_p = a.dw, *part_p = a.part;
    if (a.group_n > 0) {
        const int per = tiles_m * tiles_n * splits;
        const int prob = __builtin_amdgcn_readfirstlane(bid / per);
        bid -= prob * per;
        const DykWgradGroupEntry* e = wg_sgpr_ptr(a.group) + prob;
        x_p = e->x; dy_p = e->dy; dw_p = e->dw; part_p = e->part;
    }
    const int tile = bid % (tiles_m * tiles_n);
    const int tm = bid % tiles_m; bid /= tiles_m;
    const int tn = bid % tiles_n;
    const int sp = bid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Ntot = a.B * a.Ho * a.Wo;
    const int p_begin = sp * chunk;
    const int p_end = min(Ntot, p_begin + chunk);
    const int S = p_end > p_begin ? (p_end - p_begin + KR - 1) / KR : 0;
    const T* __restrict__ dyg = wg_sgpr_ptr((const T*)dy_p);
    const T* __restrict__ xg = wg_sgpr_ptr((const T*)x_p);

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    auto tr_read = [&](const char* p0, const char* p1) -> uint4 {
        v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)p0);
        v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)p1);
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    auto compute = [&](const char* pa) {
        const char* pb = pa + C::A_BYTES;
        const int i16 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < HALF / 32; ++kk) {
            uint4 fa[MI], fb[NI];
            const int r0 = g * HALF + kk * 32 + kq * 8 + (i16 >> 2);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int ch = wm * WTM + mi * 16 + 4 * (i16 & 3);
                fa[mi] = tr_read(pa + wg_off<T, BM>(r0, ch), pa + wg_off<T, BM>(r0 + 4, ch));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int ch = wn * WTN + ni * 16 + 4 * (i16 & 3);
                fb[ni] = tr_read(pb + wg_off<T, BN>(r0, ch), pb + wg_off<T, BN>(r0 + 4, ch));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, fa[mi]), __builtin_bit_cast(bf16x8_t, fb[ni]), acc[mi][ni], 0, 0, 0);
        }
        // (the MFMA block must not sink below the loop's `s_waitcnt vmcnt(..)`, see conv_wgrad.hip)
        __builtin_amdgcn_sched_barrier(0);
    };

    {
        // ---- LDS-DMA ring: wave w issues instructions w, w + 8, ... of the stage's list (dy tile first, then the x tile)
        const T* zero = (const T*)dyk_wg_zero_page;
        const T* src[C::NPW];
        int row[C::NPW], lim[C::NPW];
        unsigned off[C::NPW];
        long adv[C::NPW];
#pragma unroll
        for (int i = 0; i < C::NPW; ++i) {
            const bool isA = i < C::NPW_A;
            const int idx = (isA ? i : i - C::NPW_A) * 8 + wid;
            if (isA) {
                row[i] = idx * RPI_A + lane / VPR_A;
                const int c = m0 + wg_logical_ch<T, BM>(row[i], lane % VPR_A);
                lim[i] = c < Cout_s ? p_end : -0x7fffffff;
                src[i] = dyg + (long)(p_begin + row[i]) * lddy_s + c;
                off[i] = (unsigned)(idx * 1024);
                adv[i] = (long)KR * lddy_s;
            } else {
                row[i] = idx * RPI_B + lane / VPR_B;
                const int c = n0 + wg_logical_ch<T, BN>(row[i], lane % VPR_B);
                lim[i] = c < Cin_s ? p_end : -0x7fffffff;
                src[i] = xg + (long)(p_begin + row[i]) * ldx_s + c;
                off[i] = (unsigned)(C::A_BYTES + idx * 1024);
                adv[i] = (long)KR * ldx_s;
            }
        }
        const unsigned lds0 = wg_lds_addr(smem);
        int sp0 = p_begin;
        auto stage_next = [&](int buf) {
            const unsigned base = lds0 + (unsigned)buf * C::STAGE;
#pragma unroll
            for (int i = 0; i < C::NPW; ++i) {
                const bool ok = sp0 + row[i] < lim[i];
                const T* s = ok ? src[i] : zero;
                wg_glds16(s, __builtin_amdgcn_readfirstlane(base + off[i]));
                src[i] += adv[i];
            }
            sp0 += KR;
        };
        constexpr int AHEAD = NS - 1;
        constexpr int KEEP = (AHEAD - 1) * C::NPW;        // DMA instructions that may still be in flight per wave
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
            compute(smem + cur * C::STAGE);
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur = (cur == NS - 1) ? 0 : cur + 1;
            nxt = (nxt == NS - 1) ? 0 : nxt + 1;
        }
    }

    // ---- exchange of the K-halves: half g finishes the fragments mi in [g MH, (g + 1) MH); it parks the OTHER fragments
    // (lane-linear float4, conflict free) and adds what the other half parked.  fin[ml][ni] = finished fragment mi = g MH + ml
    f32x4_t fin[MH][NI];
    {
        float4* park = (float4*)smem;            // overlays the ring (all waves are behind the loop's last barrier)
        constexpr int CELLS = MH * NI * 4 * 64;  // per K-half
        if (g == 0) {
#pragma unroll
            for (int ml = 0; ml < MH; ++ml)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const f32x4_t v = acc[MH + ml][ni];
                    park[((ml * NI + ni) * 4 + w4) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
                }
        } else {
#pragma unroll
            for (int ml = 0; ml < MH; ++ml)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const f32x4_t v = acc[ml][ni];
                    park[CELLS + ((ml * NI + ni) * 4 + w4) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
                }
        }
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int ml = 0; ml < MH; ++ml)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float4 v = park[CELLS + ((ml * NI + ni) * 4 + w4) * 64 + lane];
                    fin[ml][ni] = acc[ml][ni] + (f32x4_t){v.x, v.y, v.z, v.w};
                }
        } else {
#pragma unroll
            for (int ml = 0; ml < MH; ++ml)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float4 v = park[((ml * NI + ni) * 4 + w4) * 64 + lane];
                    fin[ml][ni] = (f32x4_t){v.x, v.y, v.z, v.w} + acc[MH + ml][ni];
                }
        }
    }

    // ---- results: fin[ml][ni][r] = D[co = m0 + wm WTM + (g MH + ml) 16 + (lane >> 4) 4 + r][ci = n0 + wn WTN + ni 16 + (lane & 15)]
    const int lddw = a.lddw > 0 ? a.lddw : Cin_s;
    const long toff = (long)a.twt[0] * Cout_s * lddw;
    const bool exclusive = (a.tune >> 20) & 1;
    bool rmw = exclusive && splits == 1;          // single writer, single split: read-add-write instead of atomics
    float* dst = dw_p + toff;
    bool plain = false;
    if (part_p) {
        dst = part_p + (long)sp * a.part_stride + toff;
        plain = true;
    } else if (a.sk_cnt && splits > 1) {
        // in-launch fold of the S = splits slices of this tile (protocol of conv_igemm_kernel.h splitk_exchange): write-through
        // slab -> vmcnt(0) -> barrier -> one relaxed agent-scope ticket -> the last arriver re-arms the counter, takes ONE agent
        // acquire and adds the slabs in slice order, its own included (bit-reproducible); nobody waits for anybody
        constexpr int PER = MH * NI * 8 * 64;            // 16-byte cells per slab (= BM BN 4 bytes)
        f32x4_t* slab = (f32x4_t*)wg_sgpr_ptr((char*)a.sk_ws) + (size_t)tile * splits * PER;
        f32x4_t* mine = slab + (size_t)sp * PER + wid * 64 + lane;
#pragma unroll
        for (int ml = 0; ml < MH; ++ml)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) ps_st_wt16(mine + (ml * NI + ni) * 8 * 64, fin[ml][ni]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* s_flag = (int*)(smem + C::PARK - 16);       // (a parked cell nobody reads any more: all waves are behind the barrier)
        if (tid == 0) {
            uint32_t* cnt = (uint32_t*)wg_sgpr_ptr((char*)a.sk_cnt) + tile;
            const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = t == (unsigned)(splits - 1) ? 1 : 0;
            if (last) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            *s_flag = last;
        }
        __syncthreads();
        if (!*s_flag) return;
        const f32x4_t* sl0 = slab + wid * 64 + lane;
#pragma unroll
        for (int ml = 0; ml < MH; ++ml)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fin[ml][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) {
            f32x4_t v[MH][NI];
#pragma unroll
            for (int ml = 0; ml < MH; ++ml)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) v[ml][ni] = sl0[(size_t)s * PER + (ml * NI + ni) * 8 * 64];
#pragma unroll
            for (int ml = 0; ml < MH; ++ml)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) fin[ml][ni] += v[ml][ni];
        }
        rmw = exclusive;
    }
#pragma unroll
    for (int ml = 0; ml < MH; ++ml) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m0 + wm * WTM + (g * MH + ml) * 16 + (lane >> 4) * 4 + r;
            if (co >= Cout_s) continue;
            float* rowp = dst + (long)co * lddw;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int ci = n0 + wn * WTN + ni * 16 + (lane & 15);
                if (ci >= Cin_s) continue;
                if (plain) rowp[ci] = fin[ml][ni][r];
                else if (rmw) rowp[ci] += fin[ml][ni][r];
                else unsafeAtomicAdd(rowp + ci, fin[ml][ni][r]);
            }
        }
    }
}

struct PsPlan {
    int tiles, splits, chunk, kr;
};

template <int BM, int BN, int KR>
PsPlan ps_plan(const DykWgradDesc* d) {
    PsPlan p;
    const long Ntot = (long)d->B * d->Ho * d->Wo;
    p.kr = KR;
    p.tiles = dyk_div_up(d->Cout, BM) * dyk_div_up(d->Cin, BN);
    const int ksteps = dyk_div_up(Ntot, KR);
    int splits = d->splits;
    if (splits <= 0) {
        // one 8-wave workgroup per CU, every workgroup at least four stages
        splits = dyk_div_up(256, p.tiles);
        const int max_splits = ksteps / 4 > 0 ? ksteps / 4 : 1;
        if (splits > max_splits) splits = max_splits;
    }
    if (splits > ksteps) splits = ksteps;
    p.chunk = dyk_div_up(ksteps, splits) * KR;
    // (plane / fold mode with a given count: exactly that many slices are written, trailing empty ones with zeros -- also when
    // the count exceeds the number of stages)
    if ((d->part || d->sk_cnt) && d->splits > 0) splits = d->splits;
    else splits = dyk_div_up(Ntot, p.chunk);
    p.splits = splits;
    return p;
}

template <int BM, int BN, int KR, int NS>
int launch_ps(const DykWgradDesc* d, hipStream_t stream, PsPlan* query) {
    using C = PsCfg<BM, BN, KR, NS>;
    const PsPlan p = ps_plan<BM, BN, KR>(d);
    if (query) { *query = p; return DYK_OK; }
    if (d->group_n < 0 || d->group_n == 1 || (d->group_n > 0 && (!d->group || (d->sk_cnt && !d->part)))) return DYK_ERR_ARG;
    if (d->sk_cnt && !d->part && p.splits > 1) {
        if (p.tiles > d->sk_cnt_n || (int64_t)p.tiles * p.splits * BM * BN * 4 > d->sk_ws_bytes || ((uintptr_t)d->sk_ws % 16)) return DYK_ERR_ARG;
    }
    static DykDeviceOnce attr_set;   // (the attribute is per DEVICE: one flag per device id)
    auto kfn = conv_wgrad_ps_kernel<BM, BN, KR, NS>;
    if (attr_set.first()) {
        DYK_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
    }
    const unsigned nprob = d->group_n > 0 ? (unsigned)d->group_n : 1u;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(p.tiles * p.splits) * nprob), dim3(512), C::LDS, stream, *d, p.splits, p.chunk);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

// tune word (bits 28..30 == 3 selects this kernel): bits 0..7 ring stages (0 = default), bits 8..11 tile cap (1 = 64 x 64),
// bits 12..15 pixels per stage (0 = by tile, 1 = 64, 2 = 128)
template <int BM, int BN, int KR>
int dispatch_ps_ns(const DykWgradDesc* d, hipStream_t s, PsPlan* q) {
    constexpr int STAGE = KR * (BM + BN) * 2;
    int ns = d->tune & 0xff;
    if (ns <= 0) ns = 4;
    if constexpr (STAGE == 16 * 1024) {
        if (ns >= 8) return launch_ps<BM, BN, KR, 8>(d, s, q);
        if (ns >= 4) return launch_ps<BM, BN, KR, 4>(d, s, q);
        return launch_ps<BM, BN, KR, 2>(d, s, q);
    } else if constexpr (STAGE == 24 * 1024) {
        if (ns >= 6) return launch_ps<BM, BN, KR, 6>(d, s, q);
        if (ns >= 4) return launch_ps<BM, BN, KR, 4>(d, s, q);
        if (ns == 3) return launch_ps<BM, BN, KR, 3>(d, s, q);
        return launch_ps<BM, BN, KR, 2>(d, s, q);
    } else {
        if (ns >= 4) return launch_ps<BM, BN, KR, 4>(d, s, q);
        if (ns == 3) return launch_ps<BM, BN, KR, 3>(d, s, q);
        return launch_ps<BM, BN, KR, 2>(d, s, q);
    }
}

int dispatch_ps(const DykWgradDesc* d, hipStream_t s, PsPlan* q) {
    const bool cap64 = ((d->tune >> 8) & 0xf) == 1;
    const int krc = (d->tune >> 12) & 0xf;
    const bool m128 = d->Cout > 64 && !cap64, n128 = d->Cin > 64 && !cap64;
    if (m128 && n128) return dispatch_ps_ns<128, 128, 64>(d, s, q);
    if (m128) return dispatch_ps_ns<128, 64, 64>(d, s, q);
    if (n128) return dispatch_ps_ns<64, 128, 64>(d, s, q);
    if (krc == 1) return dispatch_ps_ns<64, 64, 64>(d, s, q);
    return dispatch_ps_ns<64, 64, 128>(d, s, q);
}

}  // namespace

// the pixel-streaming kernel covers: bf16, ONE tap at offset (0, 0), stride 1 (input pixel == output pixel), single problem
bool dyk_wgrad_ps_eligible(const DykWgradDesc* d) {
    if (d->dtype != DYK_BF16 || d->ntaps != 1 || d->tdy[0] || d->tdx[0] || d->isy != 1 || d->isx != 1) return false;
    if (d->Hi != d->Ho || d->Wi != d->Wo || d->twin) return false;
    return true;
}

int dyk_wgrad_ps_dispatch(const DykWgradDesc* d, hipStream_t s, int* query_splits, int* query_tiles, int64_t* query_slab) {
    if (query_splits || query_tiles || query_slab) {
        PsPlan p;
        const int rc = dispatch_ps(d, s, &p);
        if (rc != DYK_OK) return rc;
        if (query_splits) *query_splits = p.splits;
        if (query_tiles) *query_tiles = p.tiles;
        if (query_slab) {
            const bool cap64 = ((d->tune >> 8) & 0xf) == 1;
            const int bm = d->Cout > 64 && !cap64 ? 128 : 64, bn = d->Cin > 64 && !cap64 ? 128 : 64;
            *query_slab = (int64_t)bm * bn * 4;
        }
        return DYK_OK;
    }
    return dispatch_ps(d, s, nullptr);
}
