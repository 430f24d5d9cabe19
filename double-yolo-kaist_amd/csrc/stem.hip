// The Cin = 3 stem convolutions (first layer of each backbone: 3x3, pad 1, stride 1 | 2, 16 | 32 filters, + BatchNorm)
// straight from the image batch -- no im2col.
//
// Replaces: `imgs.float() / 255.0` (reference train_utils/kaist_train_eval_utils.py:54-55, evaluate.py:67-68) followed
// by nn.Conv2d(3, C, 3) of module_list[0] / module_list[second_index] (models.py:34-42), and the weight gradient of
// that layer in autograd's backward.  (The stem has no data gradient: nothing upstream is trainable.)
//
// Why not the MFMA implicit GEMM: K = 27.  Round 1 gathered 27-element patches into 32-channel rows (887 MB written and
// read back per launch at the BASELINE size) to feed a 1x1 MFMA GEMM; the layer is bound by its 335 MB output, not by
// 9 GFLOP.  Here:
//   forward   one thread = one output pixel: its 27 inputs come from the NCHW image (fp32, or uint8 divided by 255
//             exactly as the reference does), the 27 x C weights stream through SGPRs (transposed copy [27][C], scalar
//             loads), C fp32 accumulators, one 2*C-byte channels-last store; BatchNorm statistics wave -> block ->
//             one fp64 atomic per channel into a replica.  HBM-bound: image once (L1/L2 serve the 3x3 overlap),
//             output once.
//   wgrad     D[co][tc] += dy[p][co] * patch[p][tc] over all pixels on the matrix cores with the fp32-input MFMA
//             v_mfma_f32_32x32x2_f32 (exact fp32, A = one dy value per lane, B = one image value per lane: both in
//             their natural memory order -- no LDS, no transposition); every wave owns a fixed pixel range and a plane
//             of partial sums, a second tiny launch folds the planes in a fixed order (bit-reproducible, no atomics).
#include "dyk_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <bool U8> __device__ inline float load_px(const void* img, long idx) {
    if (U8) return (float)((const uint8_t*)img)[idx] / 255.0f;      // == uint8 -> float -> / 255.0 of the reference, bit for bit
    return ((const float*)img)[idx];
}

// -------------------------------------------------------------------------------------------------- forward
template <typename T, int COUT, bool U8>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const DykStemDesc d, const float* __restrict__ wt) {
    // wt: [27][COUT] as a `const __restrict__` kernel argument -- uniform addresses of provably read-only memory become
    // scalar loads (s_load_dwordx16 into SGPRs, one SGPR operand per FMA); read through the by-value descriptor they
    // were 216 vector loads of one address per pixel and the kernel ran at 1.1 ms instead of ~0.1
    __shared__ float s_red[4][2 * COUT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int HoWo = d.Ho * d.Wo, npix = d.B * HoWo;
    const long plane = (long)d.H * d.W;
    const bool stats = d.stats != nullptr;
    const bool affine = d.scale != nullptr;
    float s1[COUT], s2[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) s1[c] = s2[c] = 0.f;
    for (int p = blockIdx.x * 256 + tid; p < npix; p += gridDim.x * 256) {
        const int b = p / HoWo;
        const int r = p - b * HoWo;
        const int yo = r / d.Wo, xo = r - yo * d.Wo;
        const int y0 = yo * d.stride - 1, x0 = xo * d.stride - 1;
        float x[27];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yi = y0 + ky, xi = x0 + kx;
                const bool in = (unsigned)yi < (unsigned)d.H && (unsigned)xi < (unsigned)d.W;
                const long off = (long)b * 3 * plane + (long)(in ? yi : 0) * d.W + (in ? xi : 0);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = load_px<U8>(d.img, off + c * plane);
                    x[(ky * 3 + kx) * 3 + c] = in ? v : 0.f;
                }
            }
        float acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
        // the weights are re-read (scalar cache) for every pixel: hoisted out of the pixel loop all 27*COUT of them would
        // live in SGPRs and spill into VGPR lanes (1600 v_readlane per pixel).  (Two pixels per thread and weight fetch
        // were tried: the second patch spills to scratch and the kernel runs 17x slower.)
        uintptr_t wa = (uintptr_t)wt;
        asm volatile("" : "+s"(wa));
        const __attribute__((address_space(4))) float* w = (const __attribute__((address_space(4))) float*)wa;   // constant address space: s_load
#pragma unroll
        for (int t = 0; t < 27; ++t)
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[c] = fmaf(x[t], w[t * COUT + c], acc[c]);
        if (stats) {
#pragma unroll
            for (int c = 0; c < COUT; ++c) { s1[c] += acc[c]; s2[c] += acc[c] * acc[c]; }
        }
        if (affine) {                            // eval: folded BatchNorm + activation (one switch, not one per channel)
            switch (d.act) {
#define STEM_ACT_CASE(A) case A: _Pragma("unroll") for (int c = 0; c < COUT; ++c) acc[c] = act_fwd_c<A>(acc[c] * d.scale[c] + (d.shift ? d.shift[c] : 0.f), A); break;
                STEM_ACT_CASE(DYK_ACT_LINEAR)
                STEM_ACT_CASE(DYK_ACT_LEAKY)
                STEM_ACT_CASE(DYK_ACT_MISH)
#undef STEM_ACT_CASE
            default:
                for (int c = 0; c < COUT; ++c) acc[c] = act_fwd(d.act, acc[c] * d.scale[c] + (d.shift ? d.shift[c] : 0.f));
            }
        }
        if (sizeof(T) == 2) {
            bf16_t* yp = (bf16_t*)d.y + (long)p * d.ldy;
#pragma unroll
            for (int c = 0; c < COUT; c += 8) {
                uint4 pk;
                pk.x = f32x2_to_bf16x2(acc[c + 0], acc[c + 1]); pk.y = f32x2_to_bf16x2(acc[c + 2], acc[c + 3]);
                pk.z = f32x2_to_bf16x2(acc[c + 4], acc[c + 5]); pk.w = f32x2_to_bf16x2(acc[c + 6], acc[c + 7]);
                *(uint4*)(yp + c) = pk;
            }
        } else {
            float* yp = (float*)d.y + (long)p * d.ldy;
#pragma unroll
            for (int c = 0; c < COUT; c += 4) *(float4*)(yp + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
        }
    }
    if (stats) {
        // thread -> wave (xor shuffles, fixed tree) -> block (wave order) -> ONE fp64 atomic per channel and block
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            float a = s1[c], q = s2[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
            if (lane == 0) { s_red[wid][c] = a; s_red[wid][COUT + c] = q; }
        }
        __syncthreads();
        if (tid < 2 * COUT) {
            const float tot = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
            const int slots = d.stats_slots > 0 ? d.stats_slots : 1;
            double* st = d.stats + (size_t)(blockIdx.x % (unsigned)slots) * 2 * COUT;
            atomicAdd(st + tid, (double)tot);          // [0, COUT): sums, [COUT, 2 COUT): sums of squares
        }
    }
}

constexpr int STEM_SEG = 128;      // output pixels of one row per staged segment (MFMA kernels below)

// Forward for the loader's uint8 batches with bf16 output and 32 filters, on the bf16 MFMA: out[p][co] = sum_tc patch[p][tc]
// w[tc][co] with the patch as bf16 (pixel values are integers <= 255: exact) and the fp32 weights split into THREE bf16
// terms (8 + 8 + 8 mantissa bits: hi + mid + lo == w exactly), three MFMAs per 16 patch elements -- fp32-exact products and
// fp32 accumulation, 1/255 applied to the sums.  One wave per workgroup walks segments of 128 output pixels of a row: the
// 3 x 3 image rows under a segment are staged in LDS (aligned 32-bit loads, next segment prefetched), a 32-pixel tile is
// 6 MFMAs, the output tile goes through LDS to 16-byte channels-last stores.
// The round-2 form of this kernel (in the history: stem_fwd_u8_kernel) issued 269 VALU instructions per 32-pixel tile and wave
// -- uint8 -> bf16 conversion while parking the rows, 16 address adds for the patch gather, 16 conversions + 16 two-byte LDS
// stores + their addresses in the epilogue -- VALU active 0.28 of every wave's cycles at two to three waves per SIMD
// (tools/stem_pmc.sh): instruction-bound at 194 us for a 336 MB output.  This form (round 5), same arithmetic: 76.8 us, 4.6 TB/s.
//   * the fetched image words are parked RAW (nine ds_write_b32 per segment); a patch byte is read by ds_read_u8 at a
//     per-lane base + a compile-time tile offset (the four tiles of a segment are unrolled) and converted on the way to the
//     operand (v_cvt_f32_ubyte0 + one v_perm per pair);
//   * operands swapped: A = weights (rows = filters), B = patch (columns = pixels).  A lane then holds 16 filters of ONE pixel
//     as four runs of four: packed conversion, four ds_write_b64 into an 80-byte-pitch pixel row, two ds_read_b128 + two
//     16-byte global stores per lane (whole 64-byte pixels, consecutive pixels contiguous);
//   * BatchNorm statistics per lane and filter in registers across all tiles, folded over the pixels (lanes) once per wave.
template <int STRIDE, bool AFF, bool STATS>
__global__ __launch_bounds__(64) void stem_fwd_u8t_kernel(const DykStemDesc d, const float* __restrict__ wt, int segs_per_wg,
                                                          int segs_per_row, int nsegs) {
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;
    constexpr int COUT = 32;
    constexpr int WSEG = STEM_SEG * STRIDE + 2;                   // image columns under a segment (+ halo)
    constexpr int PW = (WSEG + 3 + 3) / 4;                        // 32-bit words per parked row (the first starts 4 bytes left of the halo: aligned)
    constexpr int NH = (PW + 63) / 64;                            // words per lane and row
    constexpr int OP = 80;                                        // bytes per pixel of the output tile in LDS (64 + 16: bank spread)
    __shared__ __attribute__((aligned(16))) char s_out[32 * OP];
    __shared__ uint32_t s_raw[9 * PW];
    __shared__ float s_fold[2 * COUT];
    const int lane = threadIdx.x;
    const int i = lane & 31, g = lane >> 5;
    const long plane = (long)d.H * d.W;
    // A fragments (constant over the launch): row co = i, k = tc = 16 ks + 8 g + j; fp32 weight = hi + mid + lo in bf16, exactly
    uint4 wf[2][3];
    int addrB[2][8];                                              // byte of s_raw under (pixel i of a tile, patch element k)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint32_t h[8], m[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tc = ks * 16 + 8 * g + j;
            const float w = tc < 27 ? wt[tc * COUT + i] : 0.f;
            const uint32_t hb = __float_as_uint(w) & 0xffff0000u;
            const float r1 = w - __uint_as_float(hb);
            const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mb);
            h[j] = hb >> 16; m[j] = mb >> 16; l[j] = __float_as_uint(r2) >> 16;
            const int tcc = tc < 27 ? tc : 0;                     // (k >= 27: any valid byte, its weight is zero)
            const int c = tcc % 3, kx = (tcc / 3) % 3, ky = tcc / 9;
            addrB[ks][j] = (ky * 3 + c) * PW * 4 + 3 + kx + i * STRIDE;
        }
        wf[ks][0] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        wf[ks][1] = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
        wf[ks][2] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    }
    // this lane's 16 filters: register r <-> filter (r & 3) + 8 (r >> 2) + 4 g
    float sc[AFF ? 16 : 1], sh[AFF ? 16 : 1];
    if constexpr (AFF) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * g;
            sc[r] = d.scale[co];
            sh[r] = d.shift ? d.shift[co] : 0.f;
        }
    }
    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;
    uint32_t iw[9][NH];
    const uint8_t* img8 = (const uint8_t*)d.img;
    auto fetch = [&](int sg) __attribute__((always_inline)) {
        const int row = sg / segs_per_row, x_begin = (sg - row * segs_per_row) * STEM_SEG;
        const int b = row / d.Ho, yo = row - b * d.Ho;
        const int a0 = x_begin * STRIDE - 4;
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
            const int kyy = rr / 3, cc = rr - kyy * 3;
            const int yi = yo * STRIDE - 1 + kyy;
            const bool rok = (unsigned)yi < (unsigned)d.H;
            const uint8_t* rp = img8 + ((long)b * 3 + cc) * plane + (long)(rok ? yi : 0) * d.W;
#pragma unroll
            for (int h2 = 0; h2 < NH; ++h2) {
                const int w = lane + h2 * 64;
                const int xw = a0 + 4 * w;
                const bool ok = rok && w < PW && xw >= 0 && xw < d.W;
                const uint32_t v = *(const uint32_t*)(rp + (ok ? xw : 0));
                iw[rr][h2] = ok ? v : 0u;
            }
        }
    };
    const uint8_t* raw8 = (const uint8_t*)s_raw;
    char* const wptr = s_out + i * OP + 8 * g;                                // this lane's pixel row of the tile, its first run of four filters
    const char* const rptr0 = s_out + (lane >> 2) * OP + (lane & 3) * 16;     // the two 16-byte vectors this lane stores: pixels lane / 4, 16 + lane / 4
    const char* const rptr1 = rptr0 + 16 * OP;
    const long goff0 = (long)(lane >> 2) * d.ldy + (lane & 3) * 8, goff1 = goff0 + 16L * d.ldy;
    const int seg0 = blockIdx.x * segs_per_wg;
    const int seg1 = min(nsegs, seg0 + segs_per_wg);
    if (seg0 < seg1) fetch(seg0);
    for (int sg = seg0; sg < seg1; ++sg) {
        const int row = sg / segs_per_row, x_begin = (sg - row * segs_per_row) * STEM_SEG;
        const int npx = min(STEM_SEG, d.Wo - x_begin);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 9; ++rr)
#pragma unroll
            for (int h2 = 0; h2 < NH; ++h2)
                if (lane + h2 * 64 < PW) s_raw[rr * PW + lane + h2 * 64] = iw[rr][h2];
        __syncthreads();
        if (sg + 1 < seg1) fetch(sg + 1);
        bf16_t* yrow = (bf16_t*)d.y + ((long)row * d.Wo + x_begin) * d.ldy;
#pragma unroll
        for (int tl = 0; tl < STEM_SEG / 32; ++tl) {
            const int px0 = tl * 32;
            if (px0 >= npx) break;
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint32_t bw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float f0 = (float)raw8[addrB[ks][2 * j] + px0 * STRIDE];
                    const float f1 = (float)raw8[addrB[ks][2 * j + 1] + px0 * STRIDE];
                    bw[j] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);   // {bf16(f0), bf16(f1)}: integers <= 255 are exact
                }
                const bf16x8_v bv = __builtin_bit_cast(bf16x8_v, make_uint4(bw[0], bw[1], bw[2], bw[3]));
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, wf[ks][q]), bv, acc, 0, 0, 0);
            }
            // C/D: column = lane & 31 (pixel of the tile), row = (r & 3) + 8 (r >> 2) + 4 g (filter)
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[r] * (1.0f / 255.0f);
            if constexpr (STATS) {
                if (px0 + 32 <= npx) {                            // (whole tile: no lane mask)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
                } else if (px0 + i < npx) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
                }
            }
            if constexpr (AFF) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = act_fwd(d.act, v[r] * sc[r] + sh[r]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 pk;
                pk.x = f32x2_to_bf16x2(v[4 * q], v[4 * q + 1]);
                pk.y = f32x2_to_bf16x2(v[4 * q + 2], v[4 * q + 3]);
                *(uint2*)(wptr + 16 * q) = pk;
            }
            __syncthreads();
            if (px0 + 32 <= npx) {
                *(uint4*)(yrow + (long)px0 * d.ldy + goff0) = *(const uint4*)rptr0;
                *(uint4*)(yrow + (long)px0 * d.ldy + goff1) = *(const uint4*)rptr1;
            } else {
                if (px0 + (lane >> 2) < npx) *(uint4*)(yrow + (long)px0 * d.ldy + goff0) = *(const uint4*)rptr0;
                if (px0 + 16 + (lane >> 2) < npx) *(uint4*)(yrow + (long)px0 * d.ldy + goff1) = *(const uint4*)rptr1;
            }
            __syncthreads();
        }
    }
    if constexpr (STATS) {
        // fold the pixels (the 32 lanes of a half-wave), then one fp64 atomic per filter and sum
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { s1[r] += __shfl_xor(s1[r], o, 64); s2[r] += __shfl_xor(s2[r], o, 64); }
        }
        if (i == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * g;
                s_fold[co] = s1[r];
                s_fold[COUT + co] = s2[r];
            }
        }
        __syncthreads();
        const int slots = d.stats_slots > 0 ? d.stats_slots : 1;
        double* st = d.stats + (size_t)(blockIdx.x % (unsigned)slots) * 2 * COUT;
        atomicAdd(st + lane, (double)s_fold[lane]);              // lanes 0..31: sum, 32..63: sum of squares ([2][COUT] contiguous)
    }
}

// -------------------------------------------------------------------------------------------------- weight gradient
// v_mfma_f32_32x32x2_f32: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]; here i = output channel,
// j = patch element tc = (ky*3 + kx)*3 + c (27 used), k = pixel (two per instruction), D[co][tc] 16 registers per lane.
// ONE WAVE per workgroup walks segments of SEG output pixels of one row: the dy segment (SEG x 32 channels, coalesced
// 16-byte loads, one batch = one memory round trip) and the 3 rows x 3 channels of image under it (fp32, converted
// once, zero padded) are staged in its private LDS slice, the MFMA operands are single LDS reads per lane.  Many small
// independent workgroups (12 per CU) hide the staging latency of one behind the MFMAs of the others; operands gathered
// straight from global memory cost an L1 transaction per distinct line (27 per instruction on the image side: 370 us
// per launch), and a 256-thread workgroup staging whole rows behind barriers was latency bound as well (380-480 us).
template <typename T, int COUT, bool U8>
__global__ __launch_bounds__(64) void stem_wgrad_kernel(const DykStemDesc d, int segs_per_wg, int segs_per_row, int nsegs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int Wseg = STEM_SEG * d.stride + 2;                     // image columns under a segment (+ halo)
    T* s_dy = (T*)smem;                                           // [SEG][32]
    float* s_img = (float*)(smem + STEM_SEG * 32 * sizeof(T));    // [3 ky][3 c][Wseg]
    const int i = lane & 31, kk = lane >> 5;
    const int tc = i;
    const int c = tc % 3, kx = (tc / 3) % 3, ky = tc / 9;
    const bool tc_ok = tc < 27;
    const long plane = (long)d.H * d.W;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* s_b = s_img + (ky * 3 + c) * Wseg + kx;          // + (pixel in segment) * stride
    constexpr int VEC = 16 / (int)sizeof(T);                     // dy elements per 16-byte load
    constexpr int NV = STEM_SEG * 32 / VEC / 64;                  // 16-byte loads per lane per segment (8 | 16)
    const int nimg = 9 * Wseg;
    const int seg0 = blockIdx.x * segs_per_wg;
    for (int sg = seg0; sg < seg0 + segs_per_wg && sg < nsegs; ++sg) {
        const int row = sg / segs_per_row, x_begin = (sg - row * segs_per_row) * STEM_SEG;
        const int b = row / d.Ho, yo = row - b * d.Ho;
        const int npx = min(STEM_SEG, d.Wo - x_begin);
        __syncthreads();                                          // previous segment consumed (one wave: a cheap barrier)
        // ---- dy segment (clamped index, unconditional load, select: one batch = one memory round trip)
        const T* dyseg = (const T*)d.dy + ((long)row * d.Wo + x_begin) * d.lddy;
        if (d.lddy == 32) {
            const int nvec = npx * 32 / VEC;
            uint4 t[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) { const int v = lane + u * 64; t[u] = ((const uint4*)dyseg)[v < nvec ? v : 0]; }
#pragma unroll
            for (int u = 0; u < NV; ++u) { const int v = lane + u * 64; if (v < nvec) ((uint4*)s_dy)[v] = t[u]; }
        } else if (d.lddy == 16 && ((uintptr_t)dyseg & 15) == 0) {
            // tight 16-channel rows: the segment is contiguous, two (bf16) / four (fp32) vectors per pixel go to the first
            // half of the pixel's 32-channel LDS row (channels >= COUT are masked at the MFMA)
            constexpr int VPP = 16 / VEC;
            const int nvec = npx * VPP;
            uint4 t[NV / 2];
#pragma unroll
            for (int u = 0; u < NV / 2; ++u) { const int v = lane + u * 64; t[u] = ((const uint4*)dyseg)[v < nvec ? v : 0]; }
#pragma unroll
            for (int u = 0; u < NV / 2; ++u) {
                const int v = lane + u * 64;
                if (v < nvec) ((uint4*)s_dy)[(v / VPP) * (32 / VEC) + (v % VPP)] = t[u];
            }
        } else {
            for (int e = lane; e < npx * 32; e += 64) {
                const int px = e >> 5, ch = e & 31;
                s_dy[e] = ch < d.lddy ? dyseg[(long)px * d.lddy + ch] : (T)0;
            }
        }
        // ---- image: 3 rows x 3 channels x Wseg columns starting at x_begin*stride - 1
        const int xi0 = x_begin * d.stride - 1;
        for (int e0 = lane; e0 < nimg; e0 += 64 * 8) {
            float t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 64;
                const int ec = e < nimg ? e : 0;
                const int rc = ec / Wseg, xp = ec - rc * Wseg;
                const int kyy = rc / 3, cc = rc - kyy * 3;
                const int yi = yo * d.stride - 1 + kyy, xi = xi0 + xp;
                const bool in = (unsigned)yi < (unsigned)d.H && (unsigned)xi < (unsigned)d.W;
                const float v = load_px<U8>(d.img, in ? ((long)b * 3 + cc) * plane + (long)yi * d.W + xi : 0L);
                t8[u] = in ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + u * 64; if (e < nimg) s_img[e] = t8[u]; }
        }
        __syncthreads();
        // ---- MFMAs: pixel pairs, four pairs per trip (their LDS reads are in flight together)
        for (int x0 = 0; x0 < npx; x0 += 8) {
            float a[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int xo = x0 + u * 2 + kk;
                const bool live = xo < npx;
                const float av = ElemTraits<T>::to_f32(s_dy[(live ? xo : 0) * 32 + i]);
                const float xv = s_b[(live ? xo : 0) * d.stride];
                a[u] = (live && i < COUT) ? av : 0.f;
                bv[u] = (live && tc_ok) ? xv : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u], acc, 0, 0, 0);
        }
    }
    // C/D: col = lane & 31 (tc), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (co)
    float* out = d.part + (size_t)blockIdx.x * COUT * 27;
    if (tc_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (co < COUT) out[co * 27 + tc] = acc[r];
        }
    }
}

// The loader's uint8 batches with bf16 gradients (the path a training step takes): same MFMA scheme as above, but
//  * the image patch is fetched as aligned 32-bit words -- 9 rows x <= 2 words per lane and segment instead of 19 / 37
//    single-byte loads, each behind two integer divisions -- (W % 4 == 0: a word is inside or outside the image as a whole);
//  * dy and the patch of the NEXT segment are requested before the MFMAs of the current one (registers as the second
//    buffer), so a segment costs one exposed memory round trip per workgroup instead of ~5 per segment.
template <int COUT>
__global__ __launch_bounds__(64) void stem_wgrad_u8_kernel(const DykStemDesc d, int segs_per_wg, int segs_per_row, int nsegs) {
    // bf16 MFMA (v_mfma_f32_32x32x16_bf16, 16 pixels per instruction instead of 2): a uint8 pixel value is an integer
    // <= 255 and EXACT in bf16, dy is bf16 already, products and sums are fp32 -- sum(dy * x) / 255 in place of
    // sum(dy * (x / 255)) differs by one rounding of the final scale.  A[i = co][k], B[k][j = tc]: lane (i | j = lane & 31,
    // g = lane >> 5) supplies k = 8 g .. 8 g + 7 of both operands = pixels x0 + 8 g + (0..7) (the same pixels on both
    // sides, so the sum over k is right whatever order the hardware walks them in).
    using T = bf16_t;
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int Wseg = STEM_SEG * d.stride + 2;
    constexpr int RS = STEM_SEG + 8;                              // dy^T row: 128 pixels + 16 bytes (bank spread)
    uint16_t* s_dyT = (uint16_t*)smem;                            // [32 co][RS]
    uint16_t* s_img = (uint16_t*)(smem + 32 * RS * 2);            // [3 ky][3 c][Wseg] pixel values as bf16 integers
    const int i = lane & 31, kk = lane >> 5;
    const int tc = i;
    const int c = tc % 3, kx = (tc / 3) % 3, ky = tc / 9;
    const bool tc_ok = tc < 27;
    const long plane = (long)d.H * d.W;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const uint16_t* s_b = s_img + (ky * 3 + c) * Wseg + kx;
    constexpr int VPP = COUT / 8;                                 // 16-byte vectors per pixel (tight rows: lddy == COUT)
    constexpr int NV = STEM_SEG * VPP / 64;                       // per lane and segment: 4 (16 channels) | 8 (32)
    const int wpr = (Wseg + 3 + 3) / 4;                           // words per patch row, from the aligned start x0 - 4
    uint4 t[NV], ty[NV];
    uint32_t iw[9][2];
    const uint8_t* img8 = (const uint8_t*)d.img;
    // ---- fused BatchNorm-backward apply (DykStemDesc.bn_fused): fold the two reduction sums of the replicas (lane = (sum, channel):
    //      one batch of loads, in flight together with the first segment's), workgroup 0 adds them to dgamma / dbeta, every lane
    //      keeps the constants of ITS eight channels (a lane always parks the same channel group: 64 % VPP == 0)
    const bool fused = d.bn_fused != 0;
    float* s_tot = (float*)(smem + 32 * RS * 2 + 9 * (STEM_SEG * 2 + 2 + 8) * 2 + 64);       // [2][32], behind the largest patch
    float f_sc[8], f_mu[8], f_rs[8], f_m1[8], f_m2[8];
    auto fold = [&]() {
        const int which = lane >> 5, ch = lane & 31;
        double acc2 = 0.0;
        if (ch < COUT) {
            const int slots = d.bn_slots > 0 ? d.bn_slots : 1;
            double a4[4] = {0.0, 0.0, 0.0, 0.0};
            int r = 0;
            for (; r + 4 <= slots; r += 4) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = d.bn_red[((size_t)(r + u) * 2 + which) * COUT + ch];
#pragma unroll
                for (int u = 0; u < 4; ++u) a4[u] += v[u];
            }
            for (; r < slots; ++r) a4[0] += d.bn_red[((size_t)r * 2 + which) * COUT + ch];
            acc2 = (a4[0] + a4[1]) + (a4[2] + a4[3]);
            if (blockIdx.x == 0) {
                float* gp = which ? d.bn_dgamma : d.bn_dbeta;          // sum(da * xhat) -> dgamma, sum(da) -> dbeta
                if (gp) gp[ch] += (float)acc2;
            }
        }
        s_tot[lane] = (float)acc2;
        __syncthreads();
        const float invn = 1.f / (float)((long)d.B * d.Ho * d.Wo);
        const int c0 = (lane % VPP) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f_sc[j] = d.bn_vecs[c0 + j]; f_mu[j] = d.bn_vecs[2 * COUT + c0 + j]; f_rs[j] = d.bn_vecs[3 * COUT + c0 + j];
            f_m1[j] = s_tot[c0 + j] * invn; f_m2[j] = s_tot[32 + c0 + j] * invn;
        }
    };
    auto fetch = [&](int sg) {
        const int row = sg / segs_per_row, x_begin = (sg - row * segs_per_row) * STEM_SEG;
        const int b = row / d.Ho, yo = row - b * d.Ho;
        const int npx = min(STEM_SEG, d.Wo - x_begin);
        const T* dyseg = (const T*)(fused ? d.bn_da : d.dy) + ((long)row * d.Wo + x_begin) * d.lddy;
        const int nvec = npx * VPP;
#pragma unroll
        for (int u = 0; u < NV; ++u) { const int v = lane + u * 64; t[u] = ((const uint4*)dyseg)[v < nvec ? v : 0]; }
        if (fused) {
            const T* yseg = (const T*)d.bn_yraw + ((long)row * d.Wo + x_begin) * d.lddy;
#pragma unroll
            for (int u = 0; u < NV; ++u) { const int v = lane + u * 64; ty[u] = ((const uint4*)yseg)[v < nvec ? v : 0]; }
        }
        const int a0 = x_begin * d.stride - 4;                   // aligned start: the patch begins at a0 + 3
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
            const int kyy = rr / 3, cc = rr - kyy * 3;
            const int yi = yo * d.stride - 1 + kyy;
            const bool rok = (unsigned)yi < (unsigned)d.H;
            const uint8_t* rp = img8 + ((long)b * 3 + cc) * plane + (long)(rok ? yi : 0) * d.W;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int w = lane + h * 64;
                const int xw = a0 + 4 * w;
                const bool ok = rok && w < wpr && xw >= 0 && xw < d.W;
                const uint32_t v = *(const uint32_t*)(rp + (ok ? xw : 0));
                iw[rr][h] = ok ? v : 0u;
            }
        }
    };
    auto park = [&](int sg) {
        const int row = sg / segs_per_row, x_begin = (sg - row * segs_per_row) * STEM_SEG;
        const int npx = min(STEM_SEG, d.Wo - x_begin);
        const int nvec = npx * VPP;
#pragma unroll
        for (int u = 0; u < NV; ++u) {                            // transpose on the way in: [pixel][co] -> [co][pixel]
            const int v = lane + u * 64;
            if (v < nvec) {
                const int px = v / VPP, c0 = (v % VPP) * 8;
                uint32_t w4[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
                if (fused) {
                    // dz = scale * (da - S1/N - xhat * S2/N), xhat = (yraw - mean) * rstd: bn_act_bwd_apply_kernel's expression with
                    // act' already in da, rounded to bf16 as that pass stores it
                    const uint32_t y4[4] = {ty[u].x, ty[u].y, ty[u].z, ty[u].w};
                    float dzv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float da = __uint_as_float((j & 1) ? (w4[j >> 1] & 0xffff0000u) : (w4[j >> 1] << 16));
                        const float yy = __uint_as_float((j & 1) ? (y4[j >> 1] & 0xffff0000u) : (y4[j >> 1] << 16));
                        const float xh = (yy - f_mu[j]) * f_rs[j];
                        dzv[j] = f_sc[j] * (da - f_m1[j] - xh * f_m2[j]);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) w4[q] = f32x2_to_bf16x2(dzv[2 * q], dzv[2 * q + 1]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s_dyT[(c0 + j) * RS + px] = (uint16_t)(w4[j >> 1] >> (16 * (j & 1)));
            }
        }
        // pixels [npx, next multiple of 16) take part in the last MFMA: zero gradient
        const int ntail = ((npx + 15) & ~15) - npx;
        for (int e = lane; e < COUT * ntail; e += 64) s_dyT[(e / ntail) * RS + npx + e % ntail] = 0;
#pragma unroll
        for (int rr = 0; rr < 9; ++rr)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int w = lane + h * 64;
                if (w >= wpr) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xp = 4 * w + j - 3;
                    if (xp >= 0 && xp < Wseg)
                        s_img[rr * Wseg + xp] = (uint16_t)(__float_as_uint((float)((iw[rr][h] >> (8 * j)) & 0xffu)) >> 16);
                }
            }
    };
    const int seg0 = blockIdx.x * segs_per_wg;
    const int seg1 = min(nsegs, seg0 + segs_per_wg);
    if (seg0 < seg1) fetch(seg0);
    if (fused) fold();
    for (int sg = seg0; sg < seg1; ++sg) {
        const int row = sg / segs_per_row, x_begin = (sg - row * segs_per_row) * STEM_SEG;
        const int npx = min(STEM_SEG, d.Wo - x_begin);
        __syncthreads();                                          // previous segment consumed
        park(sg);
        __syncthreads();
        if (sg + 1 < seg1) fetch(sg + 1);                         // in flight during the MFMAs below
        for (int x0 = 0; x0 < npx; x0 += 16) {
            uint4 av = *(const uint4*)(s_dyT + i * RS + x0 + 8 * kk);
            if (i >= COUT) av = make_uint4(0u, 0u, 0u, 0u);
            uint32_t bw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = s_b[(x0 + 8 * kk + 2 * j) * d.stride], hi = s_b[(x0 + 8 * kk + 2 * j + 1) * d.stride];
                bw[j] = tc_ok ? (lo | (hi << 16)) : 0u;
            }
            const uint4 bv = make_uint4(bw[0], bw[1], bw[2], bw[3]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, av), __builtin_bit_cast(bf16x8_v, bv), acc, 0, 0, 0);
        }
    }
    float* out = d.part + (size_t)blockIdx.x * COUT * 27;
    if (tc_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (co < COUT) out[co * 27 + tc] = acc[r] * (1.0f / 255.0f);
        }
    }
}

// dw[e] += sum over planes, fixed order: one wave per element, lane l takes planes l, l + 64, ... (eight independent
// loads in flight per lane), then a fixed xor-shuffle tree.  (With 8 lanes per element every lane walked hundreds of
// planes through dependent loads: 250-500 us, several times the MFMA kernel it follows.)
__global__ __launch_bounds__(256) void stem_wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ dw, int n, int planes) {
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if (g < n) {
        for (int q0 = lane; q0 < planes; q0 += 64 * 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int q = q0 + u * 64; t[u] = part[(size_t)(q < planes ? q : 0) * n + g]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int q = q0 + u * 64; s += q < planes ? t[u] : 0.f; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (g < n && lane == 0) dw[g] += s;
}

int check(const DykStemDesc* d) {
    if (!d || !d->img || d->B <= 0 || d->H <= 0 || d->W <= 0) return DYK_ERR_ARG;
    if (d->k != 3 || d->pad != 1 || (d->stride != 1 && d->stride != 2)) return DYK_ERR_UNSUPPORTED;
    if (d->Cout != 16 && d->Cout != 32) return DYK_ERR_UNSUPPORTED;
    if (d->dtype != DYK_BF16 && d->dtype != DYK_F32) return DYK_ERR_ARG;
    if ((long)d->B * d->H * d->W * 3 >= (1L << 31)) return DYK_ERR_ARG;       // 32-bit pixel indices
    if (d->Ho != (d->H + 2 - 3) / d->stride + 1 || d->Wo != (d->W + 2 - 3) / d->stride + 1) return DYK_ERR_ARG;
    return DYK_OK;
}

}  // namespace

template <typename K> static inline void stem_allow_lds(K kfn, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
#define STEM_DISPATCH(KERNEL, grid, ...) STEM_DISPATCH_LDS(KERNEL, grid, 0, __VA_ARGS__)
#define STEM_DISPATCH_LDS(KERNEL, grid, LDSB, ...)                                                                   \
    do {                                                                                                              \
        const bool u8 = d->in_u8 != 0;                                                                                \
        if (d->dtype == DYK_BF16) {                                                                                   \
            if (d->Cout == 32) { if (u8) { stem_allow_lds(KERNEL<bf16_t, 32, true>, LDSB); hipLaunchKernelGGL((KERNEL<bf16_t, 32, true>), grid, dim3(256), LDSB, s, __VA_ARGS__); } else { stem_allow_lds(KERNEL<bf16_t, 32, false>, LDSB); hipLaunchKernelGGL((KERNEL<bf16_t, 32, false>), grid, dim3(256), LDSB, s, __VA_ARGS__); }; } \
            else               { if (u8) { stem_allow_lds(KERNEL<bf16_t, 16, true>, LDSB); hipLaunchKernelGGL((KERNEL<bf16_t, 16, true>), grid, dim3(256), LDSB, s, __VA_ARGS__); } else { stem_allow_lds(KERNEL<bf16_t, 16, false>, LDSB); hipLaunchKernelGGL((KERNEL<bf16_t, 16, false>), grid, dim3(256), LDSB, s, __VA_ARGS__); }; } \
        } else {                                                                                                      \
            if (d->Cout == 32) { if (u8) { stem_allow_lds(KERNEL<float, 32, true>, LDSB); hipLaunchKernelGGL((KERNEL<float, 32, true>), grid, dim3(256), LDSB, s, __VA_ARGS__); } else { stem_allow_lds(KERNEL<float, 32, false>, LDSB); hipLaunchKernelGGL((KERNEL<float, 32, false>), grid, dim3(256), LDSB, s, __VA_ARGS__); }; }   \
            else               { if (u8) { stem_allow_lds(KERNEL<float, 16, true>, LDSB); hipLaunchKernelGGL((KERNEL<float, 16, true>), grid, dim3(256), LDSB, s, __VA_ARGS__); } else { stem_allow_lds(KERNEL<float, 16, false>, LDSB); hipLaunchKernelGGL((KERNEL<float, 16, false>), grid, dim3(256), LDSB, s, __VA_ARGS__); }; }   \
        }                                                                                                             \
    } while (0)

#define STEM_DISPATCH_W(KERNEL, grid, LDSB, ...)                                                                     \
    do {                                                                                                              \
        const bool u8 = d->in_u8 != 0;                                                                                \
        if (d->dtype == DYK_BF16) {                                                                                   \
            if (d->Cout == 32) { if (u8) hipLaunchKernelGGL((KERNEL<bf16_t, 32, true>), grid, dim3(64), LDSB, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<bf16_t, 32, false>), grid, dim3(64), LDSB, s, __VA_ARGS__); } \
            else               { if (u8) hipLaunchKernelGGL((KERNEL<bf16_t, 16, true>), grid, dim3(64), LDSB, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<bf16_t, 16, false>), grid, dim3(64), LDSB, s, __VA_ARGS__); } \
        } else {                                                                                                      \
            if (d->Cout == 32) { if (u8) hipLaunchKernelGGL((KERNEL<float, 32, true>), grid, dim3(64), LDSB, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<float, 32, false>), grid, dim3(64), LDSB, s, __VA_ARGS__); }   \
            else               { if (u8) hipLaunchKernelGGL((KERNEL<float, 16, true>), grid, dim3(64), LDSB, s, __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<float, 16, false>), grid, dim3(64), LDSB, s, __VA_ARGS__); }   \
        }                                                                                                             \
    } while (0)

extern "C" int dyk_stem_conv_fwd(const DykStemDesc* d, void* stream) {
    const int rc = check(d);
    if (rc) return rc;
    if (!d->wt || !d->y || d->ldy < d->Cout || (d->ldy * (d->dtype == DYK_BF16 ? 2 : 4)) % 16 || ((uintptr_t)d->y % 16)) return DYK_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long npix = (long)d->B * d->Ho * d->Wo;
    constexpr bool fast = true;
    // (16 filters / stride 2 -- the MobileNet stem -- measured faster in the scalar kernel: half of the 32-wide tile idles)
    if (fast && d->in_u8 && d->dtype == DYK_BF16 && d->Cout == 32 && d->W % 4 == 0 && ((uintptr_t)d->img % 4) == 0 && (d->stride == 1 || d->stride == 2)) {
        const int segs_per_row = (d->Wo + STEM_SEG - 1) / STEM_SEG;
        const long nsegs = (long)d->B * d->Ho * segs_per_row;
        const long want = nsegs < 3072 ? nsegs : 3072;                  // one-wave workgroups, ~12 per CU
        const int spw = (int)((nsegs + want - 1) / want);
        const dim3 grid((unsigned)((nsegs + spw - 1) / spw));
        {
#define DYK_STEM_T(S_, A_, T_) hipLaunchKernelGGL((stem_fwd_u8t_kernel<S_, A_, T_>), grid, dim3(64), 0, s, *d, d->wt, spw, segs_per_row, (int)nsegs)
#define DYK_STEM_S(S_)                                                                                     \
            do {                                                                                           \
                if (d->scale) { if (d->stats) DYK_STEM_T(S_, true, true); else DYK_STEM_T(S_, true, false); } \
                else { if (d->stats) DYK_STEM_T(S_, false, true); else DYK_STEM_T(S_, false, false); }     \
            } while (0)
            if (d->stride == 1) DYK_STEM_S(1); else DYK_STEM_S(2);
#undef DYK_STEM_S
#undef DYK_STEM_T
            DYK_LAUNCH_CHECK();
            return DYK_OK;
        }
    }
    long blocks = (npix + 255) / 256;
    // a fixed, bounded grid: every block folds its statistics once (2 * Cout atomics per block)
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks);
    STEM_DISPATCH(stem_fwd_kernel, grid, *d, d->wt);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

// number of partial planes (= one-wave workgroups) of the weight-gradient launch: `part` holds that many [Cout][27] tiles
extern "C" int dyk_stem_wgrad_planes(const DykStemDesc* d) {
    if (check(d)) return 0;
    const long nsegs = (long)d->B * d->Ho * ((d->Wo + STEM_SEG - 1) / STEM_SEG);
    long spw = (nsegs + 4095) / 4096;                // <= 4096 workgroups
    if (spw < 1) spw = 1;
    return (int)((nsegs + spw - 1) / spw);
}

// the conditions of the uint8 / bf16 MFMA kernel (the only one that carries the fused BatchNorm-backward apply)
static bool stem_wgrad_u8_ok(const DykStemDesc* d, const void* grad) {
    constexpr bool fast = true;
    return fast && d->in_u8 && d->dtype == DYK_BF16 && d->lddy == d->Cout && d->W % 4 == 0 && ((uintptr_t)d->img % 4) == 0 &&
           ((uintptr_t)grad % 16) == 0 && STEM_SEG * d->stride + 2 + 6 <= 4 * 128 && (d->Cout == 32 || d->Cout == 16);
}

// 1 = dyk_stem_conv_wgrad would run `d` with bn_fused set (the caller may then skip the separate apply pass: DYK_EW_SKIP)
extern "C" int dyk_stem_wgrad_bn_fusable(const DykStemDesc* d) {
    if (!d || check(d) != DYK_OK || !d->bn_da || !d->bn_yraw || !d->bn_vecs || !d->bn_red) return 0;
    return stem_wgrad_u8_ok(d, d->bn_da) && ((uintptr_t)d->bn_yraw % 16) == 0 ? 1 : 0;
}

extern "C" int dyk_stem_conv_wgrad(const DykStemDesc* d, void* stream) {
    const int rc = check(d);
    if (rc) return rc;
    if (d->bn_fused && !dyk_stem_wgrad_bn_fusable(d)) return DYK_ERR_UNSUPPORTED;
    if ((!d->dy && !d->bn_fused) || !d->dw || !d->part || d->lddy < d->Cout) return DYK_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long npix = (long)d->B * d->Ho * d->Wo;
    const int planes = dyk_stem_wgrad_planes(d);
    const int segs_per_row = (d->Wo + STEM_SEG - 1) / STEM_SEG;
    const long nsegs = (long)d->B * d->Ho * segs_per_row;
    const int spw = (int)((nsegs + planes - 1) / planes);
    const dim3 grid((unsigned)planes);
    const size_t es = d->dtype == DYK_BF16 ? 2 : 4;
    const size_t lds = (size_t)STEM_SEG * 32 * es + (size_t)9 * (STEM_SEG * d->stride + 2) * 4 + 1024;
    (void)npix;
    if (stem_wgrad_u8_ok(d, d->bn_fused ? d->bn_da : d->dy)) {
        if (d->Cout == 32) hipLaunchKernelGGL((stem_wgrad_u8_kernel<32>), grid, dim3(64), lds, s, *d, spw, segs_per_row, (int)nsegs);
        else hipLaunchKernelGGL((stem_wgrad_u8_kernel<16>), grid, dim3(64), lds, s, *d, spw, segs_per_row, (int)nsegs);
    } else
    STEM_DISPATCH_W(stem_wgrad_kernel, grid, lds, *d, spw, segs_per_row, (int)nsegs);
    DYK_LAUNCH_CHECK();
    const int n = d->Cout * 27;
    hipLaunchKernelGGL(stem_wgrad_fold_kernel, dim3((n * 64 + 255) / 256), dim3(256), 0, s, (const float*)d->part, d->dw, n, planes);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}
