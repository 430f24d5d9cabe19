// Shared pieces of the weight-gradient kernels (conv_wgrad.hip, conv_wgrad_ps.hip): the pixel-major LDS tile with its XOR
// chunk swizzle, the LDS-DMA wave instruction, SGPR-pinned pointers.  Each translation unit gets its own zero page.
#pragma once
#include "dyk_common.h"

namespace {

// a wave-uniform pointer pinned in SGPRs (the compiler cannot re-materialise it by re-loading the kernel argument in the loop)
template <typename P> __device__ inline P* wg_sgpr_ptr(P* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (P*)(((unsigned long long)hi << 32) | lo);
}

typedef short v4i16_t __attribute__((__vector_size__(4 * sizeof(short))));
#define LDS_AS __attribute__((address_space(3)))

// byte offset of (row, channel) in a [ROWS][C] tile, row bytes RB = C*sizeof(T)
template <typename T, int C> __device__ inline int wg_off(int row, int ch) {
    constexpr int RB = C * (int)sizeof(T);
    if (sizeof(T) == 2) {
        constexpr int NCH = RB / 32;                     // 32-byte chunks per row
        // A 32-lane group of the transposing fragment read touches ONE 32-byte chunk column of the eight rows
        // b + {0,1,2,3, 8,9,10,11} (b = K offset of the step, plus the tap shift in the multi-tap kernel: any value).  Rows
        // that are congruent modulo 256 / RB share their banks, so the XOR key must tell exactly those rows apart -- from row
        // bits that differ for ANY b: 256-byte rows (all eight collide) bits 0,1,3; 128-byte rows (the four of equal parity
        // collide) bits 1,3; 64-byte rows (b + i and b + 8 + i collide) bit 3.  (Rounds 1-3 keyed every width with
        // `row & 3 | bit 3 << 2` masked to the chunk count: right for 256-byte rows only -- 64-wide tiles, i.e. every dy tile
        // of the multi-tap kernel, read with 2-way conflicts: 57 % of its LDS cycles in the round-3 counters.)
        const int f = NCH >= 8 ? ((row & 3) | (((row >> 3) & 1) << 2)) & (NCH - 1)
                    : NCH == 4 ? (((row >> 1) & 1) | (((row >> 3) & 1) << 1))
                    : NCH == 2 ? ((row >> 3) & 1) : 0;
        const int chunk = (ch >> 4) ^ f;
        return row * RB + chunk * 32 + (ch & 15) * 2;
    } else {
        constexpr int NCH = RB / 64;                     // 64-byte chunks per row
        const int chunk = (ch >> 4) ^ (row & 1 & (NCH - 1));
        return row * RB + chunk * 64 + (ch & 15) * 4;
    }
}

// one LDS-DMA wave instruction (see conv_igemm.hip: issued via inline asm so that hipcc does not
// drain vmcnt(0) in front of every LDS read while the ring is in flight)
static __device__ uint4 dyk_wg_zero_page[8];
__device__ inline void wg_glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}
__device__ inline unsigned wg_lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS const char*)p);
}

// logical channel stored at physical 16-byte slot `ps` of tile row `row` (inverse of wg_off's swizzle,
// which is an XOR on the 32-byte (bf16) / 64-byte (f32) chunk index and therefore its own inverse)
template <typename T, int C> __device__ inline int wg_logical_ch(int row, int ps) {
    constexpr int EPV = 16 / (int)sizeof(T);
    return (wg_off<T, C>(row, ps * EPV) - row * C * (int)sizeof(T)) / (int)sizeof(T);
}

}  // namespace
