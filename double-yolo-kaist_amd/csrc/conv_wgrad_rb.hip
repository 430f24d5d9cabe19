// Row-block weight gradient for 3x3 / pad 1 convolutions on the matrix cores (bf16), round 4:
//     dw[t][co][ci] += sum_n dy[n][co] * x[src(n, t)][ci]
// The per-tap kernel (conv_wgrad.hip) stages a dy tile and an x tile of 64 pixels per tap and K step: 32 KB through the
// LDS-DMA path for 1 M multiply-adds, nine times per pixel range.  Measured on 3x3 128->128 @64x80 (tools/gpu_probe.py
// wgablate): 1.65 us per K step of a workgroup whose MFMA work is 0.43 us -- the LDS port (DMA writes + fragment reads) and
// the L2 -> LDS fill path carry 10 TB/s chip-wide and bound the kernel at 26 % of the MFMA rate inside its loop.
//
// Here a K step is a BLOCK of output pixels (nimg images x rh rows x wseg columns = 128 or 256 pixels); its dy tile
// [KP][64 co] and the x tile WITH HALO [nimg][(rh-1)*si+3][(wseg-1)*si+3][32 ci] are staged once and feed all nine taps:
// 9 x 64 x 32 x 128 = 2.4 M multiply-adds per 31-34 KB staged (2.7x the per-tap kernel's, 2.2x the 32-pixel-segment multi-tap
// kernel's with its ragged rows on the 80- / 40- / 20-pixel maps), and the block shape is chosen per map so that no pixel
// of a step is padding.  A workgroup is 8 waves = 2 halves of the 32 input channels x 4 K-quarters of the step: each wave
// holds the nine [64 co x 16 ci] accumulator tiles (144 registers) and reads 8 dy + 18 x fragments per 36 MFMAs (0.72 LDS
// reads per MFMA; the 2 x 2 wave split of the older kernels: 1.0-1.2); the four K-quarters are summed through LDS at the
// end (all eight waves store), so a workgroup writes ONE 64 x 32 x 9 tile (73 KB: planes cost what the per-tap kernel's 128 x 128 tiles cost).
//
// The transposing fragment read (ds_read_b64_tr_b16) takes an address per lane, so a tap is nothing but a different address
// into the halo tile: the 18 addresses of a wave's (k-block, half, tap) combinations are computed once and live in
// registers -- no address arithmetic inside the loop.  K order inside a 32-pixel block is permuted (lane group kq reads
// pixels 4 kq .. 4 kq + 3 and 16 + 4 kq ..) so that a 32-lane access group touches EIGHT CONSECUTIVE tile rows, which the
// row-keyed chunk swizzle makes bank-conflict free for any tap shift (x: 64-byte rows, key = row bit 2; dy: 128-byte rows,
// key = row bits 1-2).
//
// Replaces autograd's convolution_backward (weight gradient) for the 3x3 nn.Conv2d of reference models.py:34-42.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "dyk_common.h"

namespace {

typedef short v4i16_t __attribute__((__vector_size__(4 * sizeof(short))));
#define LDS_AS __attribute__((address_space(3)))

__device__ uint4 dyk_rb_zero_page[8];

struct RbGeom {
    int nimg, rh, wseg;      // output pixels of a K step: nimg x rh x wseg  (= KP)
    int hr, xw, hp;          // halo tile per image: hr rows x xw columns; hp = nimg * hr * xw halo pixels
    int nbx, nby, nsteps;    // blocks per row / per column of blocks, K steps of the whole problem
    int si;
};

__device__ inline void rb_glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}
__device__ inline unsigned rb_lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS const char*)p);
}
template <typename P> __device__ inline P* rb_sgpr_ptr(P* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (P*)(((unsigned long long)hi << 32) | lo);
}

// byte offset of (row, 32-byte chunk) in the dy tile [KP][64 co] (128-byte rows) and in the x tile [rows][32 ci] (64-byte rows)
__device__ inline int rb_a_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 3)) << 5); }
__device__ inline int rb_b_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 1)) << 5); }

constexpr int RB_BM = 64, RB_BN = 32;
constexpr int RB_RED_BYTES = 4 * 36 * 1024;          // four waves' accumulators in the end-of-kernel fold

template <int KKW, int NXW> struct RbCfg {
    static constexpr int KP = 128 * KKW;                   // pixels per K step
    static constexpr int A_BYTES = KP * 128;
    static constexpr int NAW = 2 * KKW;                    // dy DMA instructions per wave and step (1 KB = 8 tile rows each)
    static constexpr int B_BYTES = NXW * 8 * 1024;         // x tile: NXW DMA instructions per wave (1 KB = 16 tile rows each)
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int NS_FIT = (160 * 1024) / STAGE;
    static constexpr int NS = NS_FIT > 4 ? 4 : NS_FIT;
    static constexpr int LDS = NS * STAGE > RB_RED_BYTES ? NS * STAGE : RB_RED_BYTES;
    static_assert(NS >= 2, "ring");
};

template <int KKW, int NXW>
__global__ __launch_bounds__(512) void conv_wgrad_rb_kernel(const DykWgradDesc a, const RbGeom g, const int splits, const int chunk) {
    using C = RbCfg<KKW, NXW>;
    using T = bf16_t;
    constexpr int KP = C::KP, A_BYTES = C::A_BYTES, NAW = C::NAW, STAGE = C::STAGE, NS = C::NS;
    constexpr int NPW = NAW + NXW;
    constexpr int AHEAD = NS - 1;
    constexpr int KEEP = (AHEAD - 1) * NPW;
    static_assert(KEEP <= 63, "vmcnt range");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nh = wv & 1, kg = wv >> 1;                  // half of the 32 input channels | K-quarter of a step
    const int i16 = lane & 15, kq = lane >> 4;

    const int tiles_m = (a.Cout + RB_BM - 1) / RB_BM;
    const int tiles_n = (a.Cin + RB_BN - 1) / RB_BN;
    // consecutive remapped ids share an XCD: the channel tiles of one pixel range read the same dy / x lines through one L2
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    // grouped launch (DykWgradDesc.group): problem p owns blocks [p per, (p + 1) per); its tensors come out of the device table
    const void *x_p = a.x, *dy_p = a.dy;
    float *dw_p = a.dw, *part_p = a.part;
    if (a.group_n > 0) {
        const int per = tiles_m * tiles_n * splits;
        const int prob = __builtin_amdgcn_readfirstlane(bid / per);
        bid -= prob * per;
        const DykWgradGroupEntry* e = rb_sgpr_ptr(a.group) + prob;
        x_p = e->x; dy_p = e->dy; dw_p = e->dw; part_p = e->part;
    }
    const int tm = bid % tiles_m; bid /= tiles_m;
    const int tn = bid % tiles_n;
    const int sp = bid / tiles_n;
    const int m0 = tm * RB_BM, n0 = tn * RB_BN;
    const int g_begin = sp * chunk;
    const int g_end = min(g.nsteps, g_begin + chunk);
    const int S = g_end > g_begin ? g_end - g_begin : 0;
    const T* __restrict__ dyg = rb_sgpr_ptr((const T*)dy_p);
    const T* __restrict__ xg = rb_sgpr_ptr((const T*)x_p);

    const int Cout_s = __builtin_amdgcn_readfirstlane(a.Cout), Cin_s = __builtin_amdgcn_readfirstlane(a.Cin);
    const int lddy_s = __builtin_amdgcn_readfirstlane(a.lddy), ldx_s = __builtin_amdgcn_readfirstlane(a.ldx);
    const int Hi_s = __builtin_amdgcn_readfirstlane(a.Hi), Wi_s = __builtin_amdgcn_readfirstlane(a.Wi);
    const int Ho_s = __builtin_amdgcn_readfirstlane(a.Ho), Wo_s = __builtin_amdgcn_readfirstlane(a.Wo);
    const int si = g.si, wseg = g.wseg, rh = g.rh, xw = g.xw, hr = g.hr;

    // pixel r of a step (K order) -> (image, row, column) inside the block
    auto pix_of = [&](int r, int& im, int& py, int& px) {
        im = r / (rh * wseg);
        const int q = r - im * (rh * wseg);
        py = q / wseg;
        px = q - py * wseg;
    };

    // ---- fragment addresses (byte offsets inside a stage), fixed for the whole kernel.  dy: one address per (k-block, half);
    //      the four 16-channel groups are that address XOR (mi << 5) (the chunk swizzle is an XOR on address bits 5-6)
    int aoff[KKW][2], boff[KKW][2][9];
#pragma unroll
    for (int w = 0; w < KKW; ++w) {
        const int kk = kg + 4 * w;                             // this wave's 32-pixel K block of the step
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = kk * 32 + 16 * h + kq * 4 + (i16 >> 2);
            aoff[w][h] = rb_a_off(r, 0) + (i16 & 3) * 8;
            int im, py, px;
            pix_of(r, im, py, px);
            const int h0 = (im * hr + py * si) * xw + px * si;     // halo pixel of K pixel r under tap (0, 0)
#pragma unroll
            for (int t = 0; t < 9; ++t) boff[w][h][t] = A_BYTES + rb_b_off(h0 + (t / 3) * xw + (t % 3), nh) + (i16 & 3) * 8;
        }
    }

    // ---- LDS-DMA sources: one wave instruction fills 1 KB = 8 dy tile rows | 16 x tile rows; lane i carries the 16-byte
    //      slot i of that kilobyte, the swizzle is applied on the source side (logical channel of the physical slot).
    //      Per slot ONE element offset relative to the step's origin pixel (dy: negative = channel beyond Cout -> zero page)
    //      and, for x, the halo coordinates packed into one register for the border test.
    const T* zero = (const T*)dyk_rb_zero_page;
    int a_eoff[NAW];
#pragma unroll
    for (int j = 0; j < NAW; ++j) {
        const int row = (j * 8 + wv) * 8 + (lane >> 3), ps = lane & 7;
        const int lc = (ps >> 1) ^ ((row >> 1) & 3);
        int im, py, px;
        pix_of(row, im, py, px);
        const int c = m0 + lc * 16 + (ps & 1) * 8;
        a_eoff[j] = c < Cout_s ? ((im * Ho_s + py) * Wo_s + px) * lddy_s + c : -1;
    }
    int b_eoff[NXW], b_yx[NXW];
#pragma unroll
    for (int j = 0; j < NXW; ++j) {
        const int row = (j * 8 + wv) * 16 + (lane >> 2), ps = lane & 3;
        const int lc = (ps >> 1) ^ ((row >> 2) & 1);
        const int im = row / (hr * xw);
        const int q = row - im * (hr * xw);
        const int hy = q / xw, hx = q - hy * xw;
        const int c = n0 + lc * 16 + (ps & 1) * 8;
        const bool live = row < g.hp && c < Cin_s;
        b_yx[j] = live ? (hy << 16) | hx : (0x7000 << 16);     // dead slots fail the row test below -> zero page
        b_eoff[j] = ((im * Hi_s + hy - 1) * Wi_s + hx - 1) * ldx_s + c;
    }

    // block counters of the next step to stage
    int bx = g_begin % g.nbx, by = (g_begin / g.nbx) % g.nby, bb = g_begin / (g.nbx * g.nby);
    auto stage_next = [&](int buf) {
        char* da = smem + buf * STAGE;
        char* db = da + A_BYTES;
        const int b0 = bb * g.nimg, y0 = by * rh, x0 = bx * wseg;
        const T* dy0 = dyg + (long)((b0 * Ho_s + y0) * Wo_s + x0) * lddy_s;
        const int ys = y0 * si - 1, xs = x0 * si - 1;          // input coordinates of the halo tile's corner
        const T* x00 = xg + (long)((b0 * Hi_s + ys + 1) * Wi_s + xs + 1) * ldx_s;
#pragma unroll
        for (int j = 0; j < NAW; ++j) {
            const T* cand = dy0 + a_eoff[j];
            const T* src = a_eoff[j] >= 0 ? cand : zero;
            rb_glds16(src, rb_lds_addr(da + (j * 8 + wv) * 1024));
        }
#pragma unroll
        for (int j = 0; j < NXW; ++j) {
            const int hy = b_yx[j] >> 16, hx = b_yx[j] & 0xffff;
            const bool ok = ((unsigned)(ys + hy) < (unsigned)Hi_s) & ((unsigned)(xs + hx) < (unsigned)Wi_s);
            const T* cand = x00 + b_eoff[j];
            const T* src = ok ? cand : zero;
            rb_glds16(src, rb_lds_addr(db + (j * 8 + wv) * 1024));
        }
        if (++bx == g.nbx) { bx = 0; if (++by == g.nby) { by = 0; ++bb; } }
    };

    f32x4_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[t][mi] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    auto tr_read = [&](const char* p0, const char* p1) -> bf16x8_t {
        v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)p0);
        v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS v4i16_t*)(LDS_AS char*)p1);
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return __builtin_bit_cast(bf16x8_t, make_uint4(l2.x, l2.y, h2.x, h2.y));
    };
    auto compute = [&](const char* ps) {
#pragma unroll
        for (int w = 0; w < KKW; ++w) {
            bf16x8_t fa[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fa[mi] = tr_read(ps + (aoff[w][0] ^ (mi << 5)), ps + (aoff[w][1] ^ (mi << 5)));
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bf16x8_t fb = tr_read(ps + boff[w][0][t], ps + boff[w][1][t]);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[t][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mi], fb, acc[t][mi], 0, 0, 0);
            }
            // issue order: the dy fragments and the first two taps' x fragments, then per tap the reads of tap t + 2 between
            // the MFMAs of tap t (two taps of fragments in flight; the scheduler otherwise hoists all 26 reads: +36 registers)
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (t + 2 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        // the MFMA block must not sink below the next step's `s_waitcnt vmcnt(..)` (see conv_wgrad.hip)
        __builtin_amdgcn_sched_barrier(0);
    };

    const bool abl_noloop = (a.tune >> 17) & 1;                // analysis switch (tools/gpu_probe.py), never set by the plan
    const int SL = abl_noloop ? 0 : S;
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (SL > i) stage_next(i);
    int cur = 0, nxt = AHEAD;
    for (int s = 0; s < SL; ++s) {
        // stage s has landed once at most the DMAs of the AHEAD - 1 younger stages are outstanding -- if those were issued
        if (s + AHEAD - 1 < SL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // everybody's part of stage s is there; everybody is done with s - 1
        if (s + AHEAD < SL) stage_next(nxt);                   // ... whose buffer is refilled
        compute(smem + cur * STAGE);
        cur = (cur == NS - 1) ? 0 : cur + 1;
        nxt = (nxt == NS - 1) ? 0 : nxt + 1;
    }

    // ---- fold the four K-quarters through LDS and store the tile with ALL waves.  Round 1: quarters 2, 3 hand their
    //      accumulators (as they sit in registers) to quarters 0, 1.  Round 2: quarters 0, 1 write their sums as row-major
    //      tiles [t][64 co][32 ci]; then every thread adds the two tiles for one float4 of a 128-byte gradient row per tap and
    //      stores it (the first version stored from two waves, 4 bytes per lane and instruction: 9 us of fixed cost per launch)
    f32x4_t* red = (f32x4_t*)smem;
    __syncthreads();
    if (kg >= 2) {
        f32x4_t* dst = red + ((kg - 2) * 2 + nh) * (36 * 64) + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) dst[(t * 4 + mi) * 64] = acc[t][mi];
    }
    __syncthreads();
    if (kg < 2) {
        const f32x4_t* src = red + (kg * 2 + nh) * (36 * 64) + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[t][mi] += src[(t * 4 + mi) * 64];
            __builtin_amdgcn_sched_barrier(0);                 // (four loads in flight, not all 36: they would spill the accumulators)
        }
    }
    __syncthreads();
    float* tile = (float*)smem;                                // [2 quarters][9][64][32]
    if (kg < 2) {
        // acc[t][mi][r] = D_t[co = mi*16 + (lane>>4)*4 + r][ci = nh*16 + (lane&15)]
        float* dst = tile + kg * (9 * 64 * 32) + ((lane >> 4) * 4) * 32 + nh * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(t * 64 + mi * 16 + r) * 32] = acc[t][mi][r];
    }
    __syncthreads();
    if (S == 0 && !part_p) return;
    const int lddw = a.Cin;
    const bool plane = part_p != nullptr;
    float* out = plane ? part_p + (long)sp * a.part_stride : dw_p;
    // one K split and a caller that vouches for it (tune bit 20: nobody else adds to dw while this launch runs): every
    // gradient element has exactly one writer, read-add-write replaces 4.7 M atomics on the 16x20 layers (53 -> 8 us)
    const bool rmw = !plane && splits == 1 && ((a.tune >> 20) & 1);
    const bool vec = (plane || rmw) && ((((size_t)out) | ((size_t)lddw * 4) | ((size_t)a.Cout * lddw * 4)) & 15) == 0;
    if (vec) {
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const int co = m0 + row, ci = n0 + c4;
        if (co >= a.Cout || ci >= a.Cin) return;
        float* base = out + (long)co * lddw + ci;
        const float* t0 = tile + row * 32 + c4;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4_t u = *(const f32x4_t*)(t0 + t * (64 * 32)), v = *(const f32x4_t*)(t0 + (9 + t) * (64 * 32));
            f32x4_t* q = (f32x4_t*)(base + (long)a.twt[t] * a.Cout * lddw);
            *q = rmw ? *q + (u + v) : u + v;
        }
        return;
    }
    // atomic mode (or an unaligned plane): one float per lane, a wave instruction covers two whole 128-byte gradient rows
    const int col = tid & 31, row0 = tid >> 5;
    if (n0 + col >= a.Cin) return;
    float* base = out + (long)m0 * lddw + n0 + col;
    const float* t0 = tile + col;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* q = base + (long)a.twt[t] * a.Cout * lddw;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = row0 + 16 * p;
            if (m0 + row >= a.Cout) continue;
            const float sum = t0[(t * 64 + row) * 32] + t0[((9 + t) * 64 + row) * 32];
            if (plane) q[(long)row * lddw] = sum;
            else if (rmw) q[(long)row * lddw] += sum;
            else unsafeAtomicAdd(q + (long)row * lddw, sum);
        }
    }
}

// Block shape of a K step: nimg x rh x wseg = KP output pixels, each factor dividing its extent, with the smallest halo tile.
// Columns in whole multiples of 4 (the four pixels of a lane group's fragment word are consecutive tile rows).
bool rb_geometry(const DykWgradDesc* d, int KP, RbGeom* out) {
    const int si = d->isy;
    long best = -1;
    for (int wseg = 4; wseg <= d->Wo && wseg <= KP; wseg += 4) {
        if (d->Wo % wseg || KP % wseg) continue;
        const int rest = KP / wseg;
        for (int rh = 1; rh <= d->Ho && rh <= rest; ++rh) {
            if (d->Ho % rh || rest % rh) continue;
            const int nimg = rest / rh;
            if (d->B % nimg) continue;
            if (nimg > 1 && (rh != d->Ho)) continue;           // (several images per step only as whole columns of blocks)
            const int hr = (rh - 1) * si + 3, xw = (wseg - 1) * si + 3;
            const long hp = (long)nimg * hr * xw;
            // prefer column counts in multiples of 8 (conflict-free fragment reads), then the smaller halo
            const long cost = hp * 2 + ((wseg % 8) ? hp / 2 : 0);
            if (best < 0 || cost < best) {
                best = cost;
                out->nimg = nimg; out->rh = rh; out->wseg = wseg; out->hr = hr; out->xw = xw; out->hp = (int)hp;
            }
        }
    }
    if (best < 0) return false;
    out->si = si;
    out->nbx = d->Wo / out->wseg;
    out->nby = d->Ho / out->rh;
    out->nsteps = (d->B / out->nimg) * out->nby * out->nbx;
    return true;
}

template <int KKW, int NXW>
int rb_launch(const DykWgradDesc* d, const RbGeom& g, hipStream_t stream, int* query) {
    using C = RbCfg<KKW, NXW>;
    const int tiles = dyk_div_up(d->Cout, RB_BM) * dyk_div_up(d->Cin, RB_BN);
    int splits = d->splits;
    if (splits <= 0) {
        constexpr int target = 256;                            // (128 ... 384 workgroups: +-0.1 ms in the step, round 4)
        splits = dyk_div_up(target, tiles);                    // one workgroup (8 waves, its LDS) per CU
        const int max_splits = g.nsteps / 4 > 0 ? g.nsteps / 4 : 1;
        if (splits > max_splits) splits = max_splits;
    }
    if (splits > g.nsteps) splits = g.nsteps;
    const int chunk = dyk_div_up(g.nsteps, splits);
    if (!(d->part && d->splits > 0)) splits = dyk_div_up(g.nsteps, chunk);
    if (query) { *query = splits; return DYK_OK; }
    if (d->group_n < 0 || d->group_n == 1 || (d->group_n > 0 && !d->group)) return DYK_ERR_ARG;
    static DykDeviceOnce attr_set;   // (the attribute is per DEVICE: one flag per device id; after the query path: no device there)
    auto kfn = conv_wgrad_rb_kernel<KKW, NXW>;
    if (attr_set.first()) {
        DYK_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
    }
    DykWgradDesc a = *d;
    a.twin = nullptr;
    const unsigned nprob = d->group_n > 0 ? (unsigned)d->group_n : 1u;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(tiles * splits) * nprob), dim3(512), C::LDS, stream, a, g, splits, chunk);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

template <int KKW>
int rb_dispatch_n(const DykWgradDesc* d, const RbGeom& g, hipStream_t s, int* query) {
    const int nxw = dyk_div_up(g.hp, 128);                     // 8 waves x 16 tile rows per DMA instruction
    switch (nxw) {
    case 1: case 2: return rb_launch<KKW, 2>(d, g, s, query);
    case 3: return rb_launch<KKW, 3>(d, g, s, query);
    case 4: return rb_launch<KKW, 4>(d, g, s, query);
    case 5: return rb_launch<KKW, 5>(d, g, s, query);
    default: break;
    }
    return DYK_ERR_UNSUPPORTED;
}

}  // namespace

// 0 = the row-block kernel does not cover this problem (the caller falls back to the per-tap kernel)
bool dyk_wgrad_rb_eligible(const DykWgradDesc* d) {
    if (d->dtype != DYK_BF16 || d->ntaps != 9 || d->isy != d->isx || (d->isy != 1 && d->isy != 2) || d->twin) return false;
    if (d->Cin % 8 || d->Cout % 8 || (d->lddw > 0 && d->lddw != d->Cin)) return false;
    for (int t = 0; t < 9; ++t)
        if (d->tdy[t] != t / 3 - 1 || d->tdx[t] != t % 3 - 1) return false;
    const int kkw = ((d->tune >> 8) & 0xff) == 2 ? 2 : 1;
    RbGeom g;
    if (!rb_geometry(d, 128 * kkw, &g)) return false;
    const int nxw = dyk_div_up(g.hp, 128);
    return nxw <= 5;
}

int dyk_wgrad_rb_dispatch(const DykWgradDesc* d, hipStream_t s, int* query) {
    const int kkw = ((d->tune >> 8) & 0xff) == 2 ? 2 : 1;
    RbGeom g;
    if (!rb_geometry(d, 128 * kkw, &g)) return DYK_ERR_UNSUPPORTED;
    return kkw == 2 ? rb_dispatch_n<2>(d, g, s, query) : rb_dispatch_n<1>(d, g, s, query);
}
