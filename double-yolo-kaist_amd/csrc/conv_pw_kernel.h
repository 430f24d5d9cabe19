// Persistent resident-weight POINTWISE convolution (1x1, stride 1) for the CDNA4 matrix cores: forward and data gradient.
//
// Why a third convolution family: the 1x1 layers of this path (reference models.py:34-42 with
// size=1: 115 of the target cfg's 191 Conv2d, every expand / project conv of the MobileNet cfgs, layers.py:223-231) are
// HBM-bound GEMMs  y[pixel][co] = sum_ci x[pixel][ci] * w[co][ci]  with K = Cin <= 256: 2-4 K steps.  In the generic
// implicit-GEMM kernel such a launch is all fixed cost -- tables, one L2 round trip per K step that nothing overlaps,
// statistics + staging epilogue, and all workgroups of a launch load, compute and store in lockstep (13.5 us for 42 MB:
// K loop 4 us, epilogue 7 us, launch 2.3 us).  Here:
//   * PERSISTENT workgroups (at most two per CU), each keeping the weights of its channel tile resident in LDS
//     ([K chunks][BM rows][BKB bytes], loaded once) and walking pixel tiles b, b + grid, ...;
//   * the activation tile of a pixel tile ([K chunks][BN pixels][BKB bytes]: the WHOLE K) arrives by LDS-DMA into a ring of
//     2-4 stages, one to three tiles ahead, with counted s_waitcnt vmcnt and ONE workgroup barrier per tile -- the loads of
//     tile t+1.. and the stores of tile t-1 are in flight while tile t runs its MFMAs (v_mfma_f32_16x16x32_bf16, no barrier
//     inside the K walk);
//   * the epilogue is WAVE-LOCAL: every wave stages its own BM/WM x BN/WN sub-tile through its own LDS region and issues its
//     own coalesced 16-byte stores (whole 128-byte channel rows) -- no workgroup barrier, no tables (the positions of a 1x1
//     stride-1 conv are linear: pixel n reads x + n * ldx and writes y + n * ldy);
//   * BatchNorm statistics (forward) and the fused BatchNorm-backward sums (data gradient: DYK_EPI_BNBWD, plain and
//     residual-chain form) stay in registers across ALL tiles of the workgroup and cost one fold + one fp64 atomic per channel
//     and workgroup; the raw conv output (and the chain addend) of the next tile are prefetched by LDS-DMA into the wave's
//     own buffers while the current tile computes;
//   * stores are buffer stores whose dead lanes (pixels behind the tensor, channels behind Cout) carry an out-of-range
//     offset: the hardware drops them, the instruction still issues, so the vmcnt immediates are exact for ragged tiles too.
// Same descriptor (DykConvDesc), same arithmetic as the generic kernel: the K walk is chunk-ascending with the same MFMA, so
// raw outputs are bit-identical to the generic tiles; sums are grouped per workgroup instead of per tile.
// bf16; ntaps == 1, unit strides, dense launch grid; Cin (padded) % 32 == 0, Cout % 8 == 0; flags in {0, STATS, BNBWD,
// BNBWD | ADDEND}.  Anything else: DYK_ERR_UNSUPPORTED (the front end falls back to the generic tiles).
//
// Replaces: nn.Conv2d(k=1) forward at reference models.py:34-42 / layers.py:229 and autograd's input gradient of those layers.
#pragma once
#include "conv_igemm_kernel.h"

namespace {

struct PwArgs {
    DykConvDesc d;
    int ntiles;          // pixel tiles of BN positions
    int mt;              // channel tiles of BM rows; workgroup b keeps tile b % mt resident
    int kc;              // K chunks of BKB bytes (Cin * 2 / BKB)
    int nxs;             // stages of the activation ring (2..4)
    int young;           // vmcnt immediate of the steady-state wait (see the kernel)
};

__device__ __forceinline__ void pw_wait_vmcnt(int n) {      // n wave-uniform, 0..63
    switch (n) {
#define DYK_PW_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        DYK_PW_W(0) DYK_PW_W(1) DYK_PW_W(2) DYK_PW_W(3) DYK_PW_W(4) DYK_PW_W(5) DYK_PW_W(6) DYK_PW_W(7)
        DYK_PW_W(8) DYK_PW_W(9) DYK_PW_W(10) DYK_PW_W(11) DYK_PW_W(12) DYK_PW_W(13) DYK_PW_W(14) DYK_PW_W(15)
        DYK_PW_W(16) DYK_PW_W(17) DYK_PW_W(18) DYK_PW_W(19) DYK_PW_W(20) DYK_PW_W(21) DYK_PW_W(22) DYK_PW_W(23)
        DYK_PW_W(24) DYK_PW_W(25) DYK_PW_W(26) DYK_PW_W(27) DYK_PW_W(28) DYK_PW_W(29) DYK_PW_W(30) DYK_PW_W(31)
        DYK_PW_W(32) DYK_PW_W(33) DYK_PW_W(34) DYK_PW_W(35) DYK_PW_W(36) DYK_PW_W(37) DYK_PW_W(38) DYK_PW_W(39)
        DYK_PW_W(40) DYK_PW_W(41) DYK_PW_W(42) DYK_PW_W(43) DYK_PW_W(44) DYK_PW_W(45) DYK_PW_W(46) DYK_PW_W(47)
        DYK_PW_W(48) DYK_PW_W(49) DYK_PW_W(50) DYK_PW_W(51) DYK_PW_W(52) DYK_PW_W(53) DYK_PW_W(54) DYK_PW_W(55)
        DYK_PW_W(56) DYK_PW_W(57) DYK_PW_W(58) DYK_PW_W(59) DYK_PW_W(60) DYK_PW_W(61) DYK_PW_W(62)
#undef DYK_PW_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// geometry of one instantiation: BM x BN tile, WM waves along the channels (4 / WM along the pixels), BKB-byte K chunks
template <int BM, int WM, int BN, int BKB> struct PwCfg {
    static constexpr int WN = 4 / WM, WTM = BM / WM, WTN = BN / WN;
    static constexpr int MI = WTM / 16, NI = WTN / 16, KK = BKB / 64, BK = BKB / 2;
    static constexpr int SPR = BKB / 16, RPI = 64 / SPR;                 // 16-byte slots per row, rows per DMA instruction
    static constexpr int IPCX = BN * BKB / 1024, IPCW = BM * BKB / 1024;  // DMA instructions per K chunk: activations / weights
    static constexpr int RB = WTM * 2, CPR = RB / 16, RPIU = 64 / CPR;    // wave sub-tile rows: bytes, 16-byte chunks, rows per DMA instruction
    static constexpr int NDU = WTN / RPIU;                                 // DMA instructions per wave and epilogue tensor
    static constexpr int NSTW = WTN * CPR / 64;                            // 16-byte stores per lane and tile
    static constexpr int STAT_BYTES = WN * 2 * BM * 4, PAR_BYTES = 4 * BM * 4;
    static constexpr int STG_BYTES = WTN * (RB + 16);                      // staging of a wave (plain / statistics epilogues)
    static constexpr int UB_BYTES = WTN * RB;                              // one epilogue-operand buffer of a wave
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && MI >= 1 && NI >= 1, "wave tile");
    static_assert(IPCX % 4 == 0, "uniform activation DMA count per wave");
    static_assert(WTN % RPIU == 0 && (WTN * CPR) % 64 == 0, "uniform epilogue DMA / store counts");
    static_assert(STAT_BYTES % 1024 == 0 && PAR_BYTES % 512 == 0, "1 KiB aligned DMA destinations");
};
// bytes of LDS of a launch (kc K chunks, nxs ring stages); EPI as in the kernel
template <int BM, int WM, int BN, int BKB> constexpr size_t pw_lds_bytes(int kc, int nxs, int epi) {
    using C = PwCfg<BM, WM, BN, BKB>;
    size_t b = C::STAT_BYTES + (epi >= 2 ? 4096 : 0) /* BatchNorm vectors (<= 4 x 256 floats) */ + (size_t)kc * BM * BKB + (size_t)nxs * kc * BN * BKB;
    b += epi <= 1 ? 4 * ((size_t)C::STG_BYTES + 15 & ~(size_t)15) : (size_t)4 * (epi == 3 ? 4 : 2) * C::UB_BYTES;
    return b;
}

// EPI: 0 = plain store, 1 = store + BatchNorm statistics (DYK_EPI_STATS), 2 = fused BatchNorm-backward reduce (DYK_EPI_BNBWD),
//      3 = the same in residual-chain form (| DYK_EPI_ADDEND).  ACTB: activation of the BatchNorm being differentiated (-1: runtime)
template <int BM, int WM, int BN, int BKB, int EPI, int ACTB>
__global__ __launch_bounds__(256) void conv_pw_kernel(const PwArgs g) {
    using C = PwCfg<BM, WM, BN, BKB>;
    using T = bf16_t;
    constexpr int WN = C::WN, WTM = C::WTM, WTN = C::WTN, MI = C::MI, NI = C::NI, KK = C::KK, BK = C::BK;
    constexpr int SPR = C::SPR, RPI = C::RPI, IPCX = C::IPCX, IPCW = C::IPCW;
    constexpr int RB = C::RB, CPR = C::CPR, RPIU = C::RPIU, NDU = C::NDU, NSTW = C::NSTW;
    const DykConvDesc& a = g.d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_stat = (float*)smem;                               // [WN][2][BM]
    float* s_par = (float*)(smem + C::STAT_BYTES);              // [4][BM]: scale, shift, mean, rstd (EPI 2 / 3)
    const int kc = __builtin_amdgcn_readfirstlane(g.kc), nxs = __builtin_amdgcn_readfirstlane(g.nxs);
    const int WBYTES = kc * BM * BKB, XBYTES = kc * BN * BKB;
    char* sW = smem + C::STAT_BYTES + (EPI >= 2 ? 4096 : 0);
    char* sX = sW + WBYTES;
    char* sE = sX + nxs * XBYTES;                               // per-wave epilogue regions

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const int wm = wv / WN, wn = wv % WN;
    const int Ntot = __builtin_amdgcn_readfirstlane(a.B * a.Hg * a.Wg);
    const int Cout = __builtin_amdgcn_readfirstlane(a.Cout), Kp = __builtin_amdgcn_readfirstlane(a.Cin);
    const int ldx = __builtin_amdgcn_readfirstlane(a.ldx), ldy = __builtin_amdgcn_readfirstlane(a.ldy), ldr = __builtin_amdgcn_readfirstlane(a.ldr);
    const int mt = __builtin_amdgcn_readfirstlane(g.mt);
    const int m0 = ((int)blockIdx.x % mt) * BM;
    const int t_first = (int)blockIdx.x / mt, t_step = (int)gridDim.x / mt;
    const int ntiles = __builtin_amdgcn_readfirstlane(g.ntiles);
    const int my_tiles = t_first < ntiles ? (ntiles - t_first + t_step - 1) / t_step : 0;

    const T* __restrict__ xg = sgpr_ptr((const T*)a.x);
    const T* __restrict__ wg = sgpr_ptr((const T*)a.w);
    const T* zero = (const T*)dyk_zero_page;
    const unsigned sW_u = lds_addr_of(sW), sX_u = lds_addr_of(sX);

    // lane -> (row within a DMA instruction, logical 16-byte slot): the swizzle key of lds_off depends on row bits below RPI only
    const int lrow = lane / SPR;
    const int ls8 = ((lds_off<BKB>(lrow, lane % SPR) - lrow * BKB) >> 4) * 8;
    // ---- weights of this channel tile: resident for the life of the workgroup
    for (int inst = wv; inst < kc * IPCW; inst += 4) {
        const int c = inst / IPCW, row = (inst - c * IPCW) * RPI + lrow;
        const int co = m0 + row;
        const T* src = co < Cout ? wg + (long)co * Kp + c * BK + ls8 : zero;
        glds16(src, sW_u + inst * 1024);
    }
    if constexpr (EPI >= 2) {
        for (int e = tid; e < 4 * BM; e += 256) {
            const int k = e / BM, c = e - k * BM;
            const float* src = k == 0 ? a.scale : (k == 1 ? a.shift : (k == 2 ? a.aux0 : a.aux1));
            s_par[e] = m0 + c < Cout ? src[m0 + c] : 0.f;
        }
    }
    const int ndxw = kc * IPCX / 4;                            // activation DMA instructions per wave and tile
    auto dma_x = [&](int t, int buf) {                         // tile index within this workgroup's sequence (may lie behind it: zeros)
        const int n0 = (t_first + t * t_step) * BN;
        const bool live = t < my_tiles;
        const unsigned dst = sX_u + buf * XBYTES;
        for (int j = 0; j < ndxw; ++j) {
            const int inst = j * 4 + wv;
            const int c = inst / IPCX, row = (inst - c * IPCX) * RPI + lrow;
            const int n = n0 + row;
            const T* src = (live && n < Ntot) ? xg + (long)n * ldx + c * BK + ls8 : zero;
            glds16(src, dst + inst * 1024);
        }
    };
    // wave-local epilogue operands (EPI 2 / 3): raw conv output u (and the chain addend) of the wave's sub-tile, dense rows of
    // RB bytes with the 16-byte chunk XOR-swizzled by the row on the source side
    char* sEw = sE + wv * (EPI <= 1 ? ((C::STG_BYTES + 15) & ~15) : (EPI == 3 ? 4 : 2) * C::UB_BYTES);
    const unsigned sEw_u = lds_addr_of(sEw);
    const T* __restrict__ ug = sgpr_ptr((const T*)a.res);
    const T* __restrict__ ag = sgpr_ptr((const T*)a.add);
    auto dma_u = [&](int t, int buf) {
        if constexpr (EPI >= 2) {
            const int n0 = (t_first + t * t_step) * BN + wn * WTN;
            const bool live = t < my_tiles;
#pragma unroll
            for (int j = 0; j < NDU; ++j) {
                const int row = j * RPIU + lane / CPR, pc = lane % CPR;
                const int n = n0 + row, ch = m0 + wm * WTM + (pc ^ (row & (CPR - 1))) * 8;
                const bool ok = live && n < Ntot && ch < Cout;
                glds16(ok ? ug + (long)n * ldr + ch : zero, sEw_u + buf * C::UB_BYTES + j * 1024);
                if constexpr (EPI == 3) glds16(ok && ag ? ag + (long)n * ldy + ch : zero, sEw_u + (2 + buf) * C::UB_BYTES + j * 1024);   // (chain form without an addend: zeros)
            }
        }
    };
    // ---- prologue: tiles 0 .. nxs-2 of the ring, epilogue operands of tile 0
    for (int i = 0; i < nxs - 1; ++i) dma_x(i, i);
    dma_u(0, 0);

    float s1[MI][4], s2[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[mi][r] = s2[mi][r] = 0.f;
    const int frow = lane & 15, fslot = lane >> 4, mlane = (lane >> 4) * 4;
    // output addressing through a buffer descriptor: dead lanes carry an offset behind num_records and are dropped by the hardware
    const unsigned ybytes = (unsigned)__builtin_amdgcn_readfirstlane(a.B * a.Ho * a.Wo * a.ldy) * 2u;
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sgpr_ptr((char*)a.y), 0, (int)ybytes, 0x00020000);
    const int young = __builtin_amdgcn_readfirstlane(g.young);
    const int act = __builtin_amdgcn_readfirstlane(a.act);

    int xb = 0, nb = nxs - 1;                                  // ring stage of tile t / of the tile whose DMA goes out in iteration t
    for (int t = 0; t < my_tiles; ++t) {
        // everything tile t needs has landed: its activations (issued nxs-1 iterations ago) and, EPI 2 / 3, its epilogue
        // operands (issued in iteration t-1, followed by that iteration's activation DMA and stores).  Warm-up: full drain.
        if (EPI >= 2 ? t >= 1 : t >= nxs - 1) pw_wait_vmcnt(young);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (first iteration: the BatchNorm vectors written above)
        __builtin_amdgcn_s_barrier();                          // all waves' DMA parts are in; stage `nb` (tile t-1) is free
        // (operands of the NEXT tile's epilogue first: the wait at the top of iteration t+1 then leaves the activation DMA
        // issued here in flight -- ring depth 3: two iterations to land)
        dma_u(t + 1, (t + 1) & 1);
        dma_x(t + nxs - 1, nb);

        f32x4_t acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const char* pb0 = sX + xb * XBYTES;
        for (int c = 0; c < kc; ++c) {
            const char* pa = sW + c * (BM * BKB);
            const char* pb = pb0 + c * (BN * BKB);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                uint4 fa[MI], fb[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fa[mi] = *(const uint4*)(pa + lds_off<BKB>(wm * WTM + mi * 16 + frow, kk * 4 + fslot));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) fb[ni] = *(const uint4*)(pb + lds_off<BKB>(wn * WTN + ni * 16 + frow, kk * 4 + fslot));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) Mma<T>::run(acc[mi][ni], fa[mi], fb[ni]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int n0w = (t_first + t * t_step) * BN + wn * WTN;     // first pixel of this wave's sub-tile
        if constexpr (EPI <= 1) {
            if constexpr (EPI == 1) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p1 = 0.f, p2 = 0.f;
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) { const float v = acc[mi][ni][r]; p1 += v; p2 += v * v; }
                        s1[mi][r] += p1; s2[mi][r] += p2;
                    }
            }
            constexpr int RS = RB + 16;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    uint2 pk;
                    pk.x = f32x2_to_bf16x2(acc[mi][ni][0], acc[mi][ni][1]);
                    pk.y = f32x2_to_bf16x2(acc[mi][ni][2], acc[mi][ni][3]);
                    *(uint2*)(sEw + (ni * 16 + frow) * RS + (mi * 16 + mlane) * 2) = pk;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < NSTW; ++j) {
                const int q = lane + 64 * j;
                const int row = q / CPR, cc = q % CPR;
                const int n = n0w + row, ch = m0 + wm * WTM + cc * 8;
                const dyk_v4u_t v = *(const dyk_v4u_t*)(sEw + row * RS + cc * 16);
                const unsigned off = (n < Ntot && ch < Cout) ? (unsigned)(n * ldy + ch) * 2u : 0xfffffff0u;
                __builtin_amdgcn_raw_buffer_store_b128(v, yrsrc, off, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the staging reads are done before the next tile's writes
            __builtin_amdgcn_wave_barrier();
        } else {
            char* ub = sEw + (t & 1) * C::UB_BYTES;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int po = wm * WTM + mi * 16 + mlane;
                asm volatile("" : "+v"(po));                       // (keep the BatchNorm vectors out of long-lived registers)
                const float4 sc = *(const float4*)(s_par + po), sh = *(const float4*)(s_par + BM + po);
                const float4 mu = *(const float4*)(s_par + 2 * BM + po), rs = *(const float4*)(s_par + 3 * BM + po);
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
                const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
                const int cb = (mi * 16 + mlane) * 2;              // byte column of the lane's four channels in a sub-tile row
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int row = ni * 16 + frow;
                    char* cell = ub + row * RB + (((cb >> 4) ^ (row & (CPR - 1))) << 4) + (cb & 15);
                    const uint2 v = *(const uint2*)cell;
                    const float yv[4] = {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                         __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
                    float gq[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
                    if constexpr (EPI == 3) {
                        // dz = acc + addend, rounded to the storage type: the apply pass will see the rounded value
                        const uint2 w = *(const uint2*)(cell + 2 * C::UB_BYTES);
                        gq[0] += __uint_as_float(w.x << 16); gq[1] += __uint_as_float(w.x & 0xffff0000u);
                        gq[2] += __uint_as_float(w.y << 16); gq[3] += __uint_as_float(w.y & 0xffff0000u);
                        const uint32_t p0 = f32x2_to_bf16x2(gq[0], gq[1]), p1 = f32x2_to_bf16x2(gq[2], gq[3]);
                        gq[0] = __uint_as_float(p0 << 16); gq[1] = __uint_as_float(p0 & 0xffff0000u);
                        gq[2] = __uint_as_float(p1 << 16); gq[3] = __uint_as_float(p1 & 0xffff0000u);
                    }
                    float outv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float da = gq[r] * act_bwd_c<ACTB>(yv[r] * scv[r] + shv[r], act);
                        s1[mi][r] += da;
                        s2[mi][r] += da * ((yv[r] - muv[r]) * rsv[r]);
                        outv[r] = EPI == 3 ? gq[r] : da;
                    }
                    uint2 pk;
                    pk.x = f32x2_to_bf16x2(outv[0], outv[1]);
                    pk.y = f32x2_to_bf16x2(outv[2], outv[3]);
                    *(uint2*)cell = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < NSTW; ++j) {
                const int q = lane + 64 * j;
                const int row = q / CPR, pc = q % CPR;
                const int n = n0w + row, ch = m0 + wm * WTM + (pc ^ (row & (CPR - 1))) * 8;
                const dyk_v4u_t v = *(const dyk_v4u_t*)(ub + row * RB + pc * 16);
                const unsigned off = (n < Ntot && ch < Cout) ? (unsigned)(n * ldy + ch) * 2u : 0xfffffff0u;
                __builtin_amdgcn_raw_buffer_store_b128(v, yrsrc, off, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        xb = xb + 1 == nxs ? 0 : xb + 1;
        nb = nb + 1 == nxs ? 0 : nb + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (run-ahead DMA of tiles behind the sequence: nothing may land after the exit)
    if constexpr (EPI >= 1) {
        // fold the sums: DPP row sums over the 16 pixel lanes, one LDS slot per wave column, columns in order, ONE fp64
        // atomic per channel and workgroup into the replica blockIdx.x % stats_slots
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
        __syncthreads();
        double* stats = sgpr_ptr(a.stats);
        const int slots = a.stats_slots > 0 ? a.stats_slots : 1;
        for (int e = tid; e < 2 * BM; e += 256) {
            const int ml = e % BM, which = e / BM;
            if (m0 + ml < Cout && my_tiles > 0) {
                float tot = 0.f;
#pragma unroll
                for (int q = 0; q < WN; ++q) tot += s_stat[(q * 2 + which) * BM + ml];
                atomicAdd(stats + (size_t)((unsigned)blockIdx.x % (unsigned)slots) * 2 * Cout + which * Cout + m0 + ml, (double)tot);
            }
        }
    }
}

inline bool conv_pw_eligible(const DykConvDesc* d) {
    if (d->dtype != DYK_BF16 || d->ntaps != 1 || d->tdy[0] || d->tdx[0] || d->ncls > 1 || d->twin || d->splitk > 1) return false;
    if (d->isy != 1 || d->isx != 1 || d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0) return false;
    if (d->Hg != d->Hi || d->Wg != d->Wi || d->Ho != d->Hi || d->Wo != d->Wi) return false;
    const int f = d->flags;
    if (f != 0 && f != DYK_EPI_STATS && f != DYK_EPI_BNBWD && f != (DYK_EPI_BNBWD | DYK_EPI_ADDEND)) return false;
    // the plain / statistics epilogues (EPI 0 / 1) store the raw sums: an activation without DYK_EPI_AFFINE -- which the generic
    // tiles do apply -- is not theirs to take
    if (!(f & DYK_EPI_BNBWD) && d->act != DYK_ACT_LINEAR) return false;
    if (d->Cin % 32 || d->Cin > 256 || d->Cout % 8 || d->ldx % 8 || d->ldy % 8 || ((uintptr_t)d->y % 16)) return false;
    if ((f & DYK_EPI_BNBWD) && (d->ldr % 8 || ((uintptr_t)d->res % 16))) return false;
    if ((f & DYK_EPI_ADDEND) && ((uintptr_t)d->add % 16)) return false;
    if ((long)d->B * d->Ho * d->Wo * d->ldy * 2 >= (1L << 31)) return false;      // buffer descriptor: 32-bit byte offsets
    return true;
}

template <int BM, int WM, int BN, int BKB, int EPI, int ACTB>
int launch_conv_pw(const DykConvDesc* d, hipStream_t stream) {
    using C = PwCfg<BM, WM, BN, BKB>;
    const int kc = d->Cin * 2 / BKB;
    const long Ntot = (long)d->B * d->Hg * d->Wg;
    const int ntiles = dyk_div_up(Ntot, BN);
    const int mt = dyk_div_up(d->Cout, BM);
    if ((size_t)kc * BM * BKB > 64 * 1024) return DYK_ERR_UNSUPPORTED;            // resident weights of one channel tile
    int nxs = (d->tune >> 8) & 0xf;
    if (nxs < 2) nxs = 3;
    if (nxs > 4) nxs = 4;
    const int ndxw = kc * C::IPCX / 4;
    // operations younger than what the wait at the top of an iteration needs (see the kernel): plain / statistics epilogues
    // (issue order per iteration: activation DMA, stores): (nxs-1) store groups + (nxs-2) DMA groups; BatchNorm-backward
    // epilogues (epilogue-operand DMA, activation DMA, stores): the stores and, from ring depth 3, one activation DMA group
    if (EPI >= 2 && nxs > 3) nxs = 3;
    auto young_of = [&](int n) { return EPI >= 2 ? C::NSTW + (n >= 3 ? ndxw : 0) : (n - 1) * C::NSTW + (n - 2) * ndxw; };
    while (nxs > 2 && (pw_lds_bytes<BM, WM, BN, BKB>(kc, nxs, EPI) > 160 * 1024 || young_of(nxs) > 60)) --nxs;
    const size_t lds = pw_lds_bytes<BM, WM, BN, BKB>(kc, nxs, EPI);
    if (lds > 160 * 1024 || young_of(nxs) > 60) return DYK_ERR_UNSUPPORTED;
    static DykDeviceOnce attr_set;   // (the attribute is per DEVICE: one flag per device id)
    auto kfn = conv_pw_kernel<BM, WM, BN, BKB, EPI, ACTB>;
    if (attr_set.first()) {
        DYK_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    // persistent: as many workgroups per CU as the LDS holds (at most two), a multiple of the channel tiles
    constexpr int cap = 2;                                     // (1 / 2 / 3 / 4 per CU: 18.85 / 18.31 / 18.31 / 18.21 ms on C5, round 5)
    int per_cu = (160 * 1024) / (int)lds;
    if (per_cu > cap) per_cu = cap;
    if (per_cu < 1) per_cu = 1;
    long grid = 256L * per_cu;
    if (grid > (long)ntiles * mt) grid = (long)ntiles * mt;
    grid -= grid % mt;
    if (grid < mt) grid = mt;
    PwArgs g;
    g.d = *d;
    g.d.twin = nullptr;
    g.ntiles = ntiles; g.mt = mt; g.kc = kc; g.nxs = nxs; g.young = young_of(nxs);
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(256), lds, stream, g);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

// tune word of the pointwise kernels: pixel-tile code 7 (bits 12..15), ring stages in bits 8..11 (0 = 3), bit 24 = 128-pixel tiles
// (channel tiles <= 64 rows only), bit 25 = 32-pixel tiles (128-row channel tiles only)
template <int EPI, int ACTB>
int dispatch_conv_pw(const DykConvDesc* d, hipStream_t s) {
    const bool bn128 = (d->tune >> 24) & 1;
    const bool k128 = (d->Cin % 64) == 0;
    if (d->Cout <= 32) {
        if (bn128) return k128 ? launch_conv_pw<32, 1, 128, 128, EPI, ACTB>(d, s) : launch_conv_pw<32, 1, 128, 64, EPI, ACTB>(d, s);
        return k128 ? launch_conv_pw<32, 1, 64, 128, EPI, ACTB>(d, s) : launch_conv_pw<32, 1, 64, 64, EPI, ACTB>(d, s);
    }
    if (d->Cout <= 64) {
        if (bn128) return k128 ? launch_conv_pw<64, 1, 128, 128, EPI, ACTB>(d, s) : launch_conv_pw<64, 1, 128, 64, EPI, ACTB>(d, s);
        return k128 ? launch_conv_pw<64, 1, 64, 128, EPI, ACTB>(d, s) : launch_conv_pw<64, 1, 64, 64, EPI, ACTB>(d, s);
    }
    if (bn128) return DYK_ERR_UNSUPPORTED;
    if ((d->tune >> 25) & 1)               // 32-pixel tiles: smaller ring stages and staging, two workgroups per CU at K = 128
        return k128 ? launch_conv_pw<128, 2, 32, 128, EPI, ACTB>(d, s) : DYK_ERR_UNSUPPORTED;
    // 128-row channel tiles (two for 136..256 output channels: the activation tile is then read by two workgroups, through L2);
    // 32-pixel tiles where K = 256 and the epilogue operands of the BatchNorm-backward forms do not fit beside 64-pixel tiles
    const int rc = k128 ? launch_conv_pw<128, 2, 64, 128, EPI, ACTB>(d, s) : launch_conv_pw<128, 2, 64, 64, EPI, ACTB>(d, s);
    if (rc != DYK_ERR_UNSUPPORTED) return rc;
    if constexpr (EPI >= 2) return k128 ? launch_conv_pw<128, 2, 32, 128, EPI, ACTB>(d, s) : DYK_ERR_UNSUPPORTED;
    return rc;
}

}  // namespace
