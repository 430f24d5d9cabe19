// Native command-list executor, parameter staging kernels and the YOLO box decode.
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "dyk_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    // 4 elements per thread, 16-byte loads
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)src)[i];
        if (sizeof(T) == 2) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
            pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
            ((uint2*)dst)[i] = pk;
        } else {
            ((float4*)dst)[i] = v;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        dst[i] = ElemTraits<T>::from_f32(src[i]);
    }
}

template <typename T>
__global__ void cast_pad_rows_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int C, int Cpad) {
    const long total = (long)R * Cpad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long r = i / Cpad;
        dst[i] = ElemTraits<T>::from_f32(c < C ? src[r * C + c] : 0.f);
    }
}

// dyk_cast_pad_table: one block = 2048 consecutive elements of one entry's destination (binary search over blk_begin).
// Plain entries: dst[r][c] = (T) src[r][c] for c < cols, 0 for cols <= c < cpad.  transpose_f32 entries: dst is fp32
// [cols][rows], dst[c][r] = src[r][c] (the [27][Cout] transpose the direct stem kernel streams through scalar loads).
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_table_kernel(const DykPadEntry* __restrict__ tab, int n_entries) {
    const int bidx = blockIdx.x;
    int lo = 0, hi = n_entries - 1;           // last entry with blk_begin <= bidx
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].blk_begin <= bidx) lo = mid; else hi = mid - 1;
    }
    const DykPadEntry e = tab[lo];
    const long total = e.transpose_f32 ? (long)e.rows * e.cols : (long)e.rows * e.cpad;
    const long i0 = (long)(bidx - e.blk_begin) * 2048;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long i = i0 + u * 256 + threadIdx.x;
        if (i >= total) break;
        if (e.transpose_f32) {
            const int r = (int)(i % e.rows);
            const long c = i / e.rows;
            ((float*)e.dst)[i] = e.src[(long)r * e.cols + c];
        } else {
            const int c = (int)(i % e.cpad);
            const long r = i / e.cpad;
            ((T*)e.dst)[i] = ElemTraits<T>::from_f32(c < e.cols ? e.src[r * e.cols + c] : 0.f);
        }
    }
}

// dst[t][col][row] = src[t][row][col] in 32x32 tiles of (entry, tap).  A workgroup walks TRANSPOSE_RUN consecutive tiles: the
// table lookup (a dependent chain of ~8 global loads) is paid once per run, not once per 6 KB tile (round 5: 324 -> us per
// launch of the target cfg's 58 M weights was mostly that chain); the next tile's loads are issued before the current tile
// is stored.
constexpr int TRANSPOSE_RUN = 8;
template <typename T>
__global__ __launch_bounds__(256) void transpose_taps_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                             const DykTransposeEntry* __restrict__ tab, int n_entries, int total_tiles) {
    __shared__ float tile[32][33];
    int tidx = blockIdx.x * TRANSPOSE_RUN;
    const int tend = min(total_tiles, tidx + TRANSPOSE_RUN);
    int lo = 0, hi = n_entries - 1;           // last entry with tile_begin <= tidx
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile_begin <= tidx) lo = mid; else hi = mid - 1;
    }
    DykTransposeEntry e = tab[lo];
    int e_end = lo + 1 < n_entries ? tab[lo + 1].tile_begin : total_tiles;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (; tidx < tend; ++tidx) {
        while (tidx >= e_end) {               // (entries without tiles are skipped)
            ++lo;
            e = tab[lo];
            e_end = lo + 1 < n_entries ? tab[lo + 1].tile_begin : total_tiles;
        }
        int local = tidx - e.tile_begin;
        const int tr = (e.rows + 31) / 32, tc = (e.cols + 31) / 32;
        const int t = local / (tr * tc);
        local -= t * tr * tc;
        const int r0 = (local / tc) * 32, c0 = (local % tc) * 32;
        const float* s = src + e.src_off + (long)t * e.rows * e.cols;
        const int dld = e.dst_ld > 0 ? e.dst_ld : e.rows;
        T* d = dst + e.dst_off + (long)t * dld * e.cols;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + ty + 8 * q, c = c0 + tx;
            v[q] = (r < e.rows && c < e.cols) ? s[(long)r * e.cols + c] : 0.f;
        }
        __syncthreads();                      // the previous tile has been read out
#pragma unroll
        for (int q = 0; q < 4; ++q) tile[ty + 8 * q][tx] = v[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + ty + 8 * q, r = r0 + tx;
            if (c < e.cols && r < dld) d[(long)c * dld + r] = ElemTraits<T>::from_f32(tile[tx][ty + 8 * q]);
        }
    }
}

__global__ void yolo_decode_kernel(DykDecodeDesc d) {
    const long cells = (long)d.na * d.ny * d.nx;
    const long total = (long)d.B * cells;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / cells);
        const long r = i - (long)b * cells;
        const int x = (int)(r % d.nx);
        const int y = (int)((r / d.nx) % d.ny);
        const int a = (int)(r / ((long)d.nx * d.ny));
        const float* t = d.p + i * d.no;
        float* o = d.io + ((long)b * d.rows_total + d.row_offset + r) * d.no;
        const float aw = d.anchor_vec[2 * a], ah = d.anchor_vec[2 * a + 1];
        // exact-rounded expf / division: decode values feed the NMS comparisons
        if (d.v4) {
            const float sx = 1.f / (1.f + expf(-t[0])), sy = 1.f / (1.f + expf(-t[1]));
            const float sw = 1.f / (1.f + expf(-t[2])), sh = 1.f / (1.f + expf(-t[3]));
            o[0] = (sx * 2.f - 0.5f + (float)x) * d.stride;
            o[1] = (sy * 2.f - 0.5f + (float)y) * d.stride;
            const float w2 = sw * 2.f, h2 = sh * 2.f;
            o[2] = (w2 * w2) * aw * d.stride;
            o[3] = (h2 * h2) * ah * d.stride;
        } else {
            o[0] = (1.f / (1.f + expf(-t[0])) + (float)x) * d.stride;
            o[1] = (1.f / (1.f + expf(-t[1])) + (float)y) * d.stride;
            o[2] = expf(t[2]) * aw * d.stride;
            o[3] = expf(t[3]) * ah * d.stride;
        }
        for (int k = 4; k < d.no; ++k) o[k] = 1.f / (1.f + expf(-t[k]));
    }
}

inline int grid_for(long n, int per_block) {
    long g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int dyk_cast_f32(const float* src, void* dst, int64_t n, int32_t dtype, void* stream) {
    if (!src || !dst || n <= 0) return DYK_ERR_ARG;
    if (((uintptr_t)src % 16) || ((uintptr_t)dst % 16)) return DYK_ERR_ARG;
    const int grid = grid_for(n / 4 + 1, 256);
    if (dtype == DYK_BF16)
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
    else if (dtype == DYK_F32)
        hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, (long)n);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_cast_pad_rows(const float* src, void* dst, int32_t R, int32_t C, int32_t Cpad, int32_t dtype,
                                 void* stream) {
    if (!src || !dst || R <= 0 || C <= 0 || Cpad < C) return DYK_ERR_ARG;
    const int grid = grid_for((long)R * Cpad, 256);
    if (dtype == DYK_BF16)
        hipLaunchKernelGGL(cast_pad_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R, C, Cpad);
    else if (dtype == DYK_F32)
        hipLaunchKernelGGL(cast_pad_rows_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, R, C, Cpad);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_cast_pad_table(const DykPadEntry* tab, int32_t n_entries, int32_t total_blocks, int32_t dtype, void* stream) {
    if (!tab || n_entries <= 0 || total_blocks <= 0) return DYK_ERR_ARG;
    if (dtype == DYK_BF16)
        hipLaunchKernelGGL(cast_pad_table_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, tab, n_entries);
    else if (dtype == DYK_F32)
        hipLaunchKernelGGL(cast_pad_table_kernel<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, tab, n_entries);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_transpose_taps(const float* src, void* dst, const DykTransposeEntry* tab, int32_t n_entries,
                                  int32_t total_tiles, int32_t dtype, void* stream) {
    if (!src || !dst || !tab || n_entries <= 0 || total_tiles <= 0) return DYK_ERR_ARG;
    if (dtype == DYK_BF16)
        hipLaunchKernelGGL(transpose_taps_kernel<bf16_t>, dim3((total_tiles + TRANSPOSE_RUN - 1) / TRANSPOSE_RUN), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, tab, n_entries, total_tiles);
    else if (dtype == DYK_F32)
        hipLaunchKernelGGL(transpose_taps_kernel<float>, dim3((total_tiles + TRANSPOSE_RUN - 1) / TRANSPOSE_RUN), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, tab, n_entries, total_tiles);
    else
        return DYK_ERR_ARG;
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_yolo_decode(const DykDecodeDesc* d, void* stream) {
    if (!d || !d->p || !d->io || d->B <= 0 || d->na <= 0 || d->na > 8 || d->ny <= 0 || d->nx <= 0 || d->no < 5)
        return DYK_ERR_ARG;
    if (d->row_offset < 0 || d->row_offset + d->na * d->ny * d->nx > d->rows_total) return DYK_ERR_ARG;
    const int grid = grid_for((long)d->B * d->na * d->ny * d->nx, 256);
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *d);
    DYK_LAUNCH_CHECK();
    return DYK_OK;
}

extern "C" int dyk_run_commands(const DykCommand* cmds, int32_t n, void* stream, int32_t* failed_index) {
    if (!cmds || n < 0) return DYK_ERR_ARG;
    // analysis (timing only, results are garbage): DYK_SKIP_OPS=<op code>,<op code>,... leaves those commands out of every
    // pass -- what the step would cost if a kernel family were free (tools/r6_ablate.sh)
    static const unsigned long long skip_ops = [] {
        unsigned long long m = 0;
        if (const char* s = getenv("DYK_SKIP_OPS"))
            for (const char* p = s; *p;) {
                const long v = strtol(p, const_cast<char**>(&p), 10);
                if (v > 0 && v < 64) m |= 1ull << v;
                while (*p == ',' || *p == ' ') ++p;
            }
        return m;
    }();
    for (int32_t k = 0; k < n; ++k) {
        const void* dp = cmds[k].desc;
        int rc = DYK_ERR_ARG;
        if (dp && cmds[k].op > 0 && cmds[k].op < 64 && ((skip_ops >> cmds[k].op) & 1)) {
            rc = DYK_OK;
        } else if (dp) {
            const DykEwDesc* e = (const DykEwDesc*)dp;
            const DykMiscDesc* m = (const DykMiscDesc*)dp;
            switch (cmds[k].op) {
            case DYK_OP_CONV: rc = dyk_conv_igemm((const DykConvDesc*)dp, stream); break;
            case DYK_OP_WGRAD: rc = dyk_conv_wgrad((const DykWgradDesc*)dp, stream); break;
            case DYK_OP_BN_FINALIZE: rc = dyk_bn_finalize((const DykBnFinalizeDesc*)dp, stream); break;
            case DYK_OP_BN_ACT_FWD: rc = dyk_bn_act_fwd(e, stream); break;
            case DYK_OP_BN_BWD_REDUCE: rc = dyk_bn_act_bwd_reduce(e, stream); break;
            case DYK_OP_BN_BWD_APPLY: rc = dyk_bn_act_bwd_apply(e, stream); break;
            case DYK_OP_AXPBY: rc = dyk_axpby(e, stream); break;
            case DYK_OP_DOT: rc = dyk_dot(e, stream); break;
            case DYK_OP_UPSAMPLE_FWD: rc = dyk_upsample2x_fwd(e, stream); break;
            case DYK_OP_UPSAMPLE_BWD: rc = dyk_upsample2x_bwd(e, stream); break;
            case DYK_OP_MAXPOOL_FWD: rc = dyk_maxpool_fwd(e, (uint8_t*)e->aux, stream); break;
            case DYK_OP_MAXPOOL_BWD: rc = dyk_maxpool_bwd(e, (const uint8_t*)e->aux, stream); break;
            case DYK_OP_SE_POOL: rc = dyk_se_pool(e, (float*)e->aux, stream); break;
            case DYK_OP_SE_FC_FWD: rc = dyk_se_fc_fwd((const DykSeFcDesc*)dp, stream); break;
            case DYK_OP_SE_FC_BWD: rc = dyk_se_fc_bwd((const DykSeFcDesc*)dp, stream); break;
            case DYK_OP_SE_SCALE: rc = dyk_se_scale(e, stream); break;
            case DYK_OP_BN_BWD_PARAMS:
                rc = dyk_bn_bwd_params((double*)m->p[0], (float*)m->p[1], (float*)m->p[2], m->i[0], m->i[1], stream); break;
            case DYK_OP_BN_FOLD:
                rc = dyk_bn_fold((const float*)m->p[0], (const float*)m->p[1], (const float*)m->p[2], (const float*)m->p[3],
                                 m->f[0], (float*)m->p[4], (float*)m->p[5], m->i[0], stream); break;
            case DYK_OP_WFUSE_WEIGHTS: rc = dyk_wfuse_weights((const float*)m->p[0], (float*)m->p[1], m->i[0], stream); break;
            case DYK_OP_WFUSE_BWD_PARAMS:
                rc = dyk_wfuse_bwd_params((const float*)m->p[0], (const double*)m->p[1], (float*)m->p[2], m->i[0], stream); break;
            case DYK_OP_HEAD_PERMUTE_FWD:
                rc = dyk_head_permute_fwd((const float*)m->p[0], (float*)m->p[1], m->i[0], m->i[1], m->i[2], m->i[3], m->i[4],
                                          m->i[5], stream); break;
            case DYK_OP_HEAD_PERMUTE_BWD:
                rc = dyk_head_permute_bwd((const float*)m->p[0], m->p[1], (float*)m->p[2], m->i[0], m->i[1], m->i[2], m->i[3],
                                          m->i[4], m->i[5], m->i[6], stream); break;
            case DYK_OP_PATCH_GATHER:
                rc = dyk_patch_gather((const float*)m->p[0], m->p[1], m->i[0], m->i[1], m->i[2], m->i[3], m->i[4], m->i[5],
                                      m->i[6], m->i[7], m->f[0], m->i[8], stream); break;
            case DYK_OP_MEMSET:
                rc = (m->p[0] && m->n > 0 && hipMemsetAsync(m->p[0], m->i[0], (size_t)m->n, (hipStream_t)stream) == hipSuccess)
                         ? DYK_OK : DYK_ERR_HIP;
                // second region (p[1], i[1] bytes, zeros): the split-K tile counters of the plan's convolutions, re-armed once
                // per pass so that a launch that faulted / was aborted cannot leave a ticket behind for every later step
                if (rc == DYK_OK && m->p[1] && m->i[1] > 0 &&
                    hipMemsetAsync(m->p[1], 0, (size_t)m->i[1], (hipStream_t)stream) != hipSuccess) rc = DYK_ERR_HIP;
                break;
            case DYK_OP_YOLO_DECODE: rc = dyk_yolo_decode((const DykDecodeDesc*)dp, stream); break;
            case DYK_OP_DW_FWD: rc = dyk_dwconv_fwd((const DykDwDesc*)dp, stream); break;
            case DYK_OP_DW_DGRAD: rc = dyk_dwconv_dgrad((const DykDwDesc*)dp, stream); break;
            case DYK_OP_DW_WGRAD: rc = dyk_dwconv_wgrad((const DykDwDesc*)dp, stream); break;
            case DYK_OP_GRAD_REDUCE:
                rc = dyk_grad_reduce((float*)m->p[0], (const float*)m->p[1], (const DykGradReduceEntry*)m->p[2], m->i[0], m->i[1], stream); break;
            case DYK_OP_BN_FWD_FUSED:
                rc = dyk_bn_finalize_act_fwd((const DykBnFinalizeDesc*)m->p[0], (const DykEwDesc*)m->p[1], stream); break;
            case DYK_OP_STEM_FWD: rc = dyk_stem_conv_fwd((const DykStemDesc*)dp, stream); break;
            case DYK_OP_STEM_WGRAD: rc = dyk_stem_conv_wgrad((const DykStemDesc*)dp, stream); break;
            case DYK_OP_CAST_PAD_ROWS:
                rc = dyk_cast_pad_rows((const float*)m->p[0], m->p[1], m->i[0], m->i[1], m->i[2], m->i[3], stream); break;
            default: rc = DYK_ERR_UNSUPPORTED; break;
            }
        }
        if (rc != DYK_OK) {
            if (failed_index) *failed_index = k;
            return rc;
        }
    }
    return DYK_OK;
}

// Two commands of the same op on shape-identical, mutually independent problems (the RGB / LWIR twin backbones of a
// dual-stream net, reference models.py:288,299-303) as ONE two-problem launch: a copy of a's descriptor gets `twin` = b's.
extern "C" int dyk_run_command_pair(const DykCommand* a, const DykCommand* b, void* stream) {
    if (!a || !b || !a->desc || !b->desc || a->op != b->op) return DYK_ERR_ARG;
    if (a->desc == b->desc) return DYK_ERR_ARG;          // two problems: one descriptor twice would write its outputs twice (in-place / accumulate forms: wrong) -- ADVICE r3
    switch (a->op) {
    case DYK_OP_CONV: {
        DykConvDesc t = *(const DykConvDesc*)a->desc;
        t.twin = (const DykConvDesc*)b->desc;
        return dyk_conv_igemm(&t, stream);
    }
    case DYK_OP_WGRAD: {
        DykWgradDesc t = *(const DykWgradDesc*)a->desc;
        t.twin = (const DykWgradDesc*)b->desc;
        return dyk_conv_wgrad(&t, stream);
    }
    case DYK_OP_BN_FINALIZE: {
        DykBnFinalizeDesc t = *(const DykBnFinalizeDesc*)a->desc;
        t.twin = (const DykBnFinalizeDesc*)b->desc;
        return dyk_bn_finalize(&t, stream);
    }
    case DYK_OP_BN_FWD_FUSED: {
        const DykMiscDesc* ma = (const DykMiscDesc*)a->desc;
        const DykMiscDesc* mb = (const DykMiscDesc*)b->desc;
        if (!ma->p[0] || !ma->p[1] || !mb->p[0] || !mb->p[1]) return DYK_ERR_ARG;
        DykBnFinalizeDesc f = *(const DykBnFinalizeDesc*)ma->p[0];
        DykEwDesc e = *(const DykEwDesc*)ma->p[1];
        f.twin = (const DykBnFinalizeDesc*)mb->p[0];
        e.twin = (const DykEwDesc*)mb->p[1];
        return dyk_bn_finalize_act_fwd(&f, &e, stream);
    }
    case DYK_OP_BN_ACT_FWD: case DYK_OP_BN_BWD_REDUCE: case DYK_OP_BN_BWD_APPLY: case DYK_OP_AXPBY: {
        DykEwDesc t = *(const DykEwDesc*)a->desc;
        t.twin = (const DykEwDesc*)b->desc;
        if (a->op == DYK_OP_BN_ACT_FWD) return dyk_bn_act_fwd(&t, stream);
        if (a->op == DYK_OP_BN_BWD_REDUCE) return dyk_bn_act_bwd_reduce(&t, stream);
        if (a->op == DYK_OP_BN_BWD_APPLY) return dyk_bn_act_bwd_apply(&t, stream);
        return dyk_axpby(&t, stream);
    }
    default: return DYK_ERR_UNSUPPORTED;
    }
}

namespace {
// one schedule entry: a single command, or a two-problem launch of cmds[e.cmd] and cmds[e.cmd2]
inline int run_entry(const DykCommand* cmds, const DykSchedEntry& e, void* stream) {
    if (e.cmd2 >= 0) return dyk_run_command_pair(cmds + e.cmd, cmds + e.cmd2, stream);
    return dyk_run_commands(cmds + e.cmd, 1, stream, nullptr);
}
}  // namespace

// Concurrency beyond one in-order stream (the lists are chains of short kernels whose ramp-up and tail leave CUs idle):
//  * the weight-gradient launches of a backward list are off the critical path (nothing in the pass reads dW): they go
//    to a side stream, each behind an event covering everything enqueued before it on the issuing stream;
//  * the second backbone of a dual-stream net is independent of the first between its fork and join points
//    (DykCommand.lane, set by the plan compiler): it runs on a branch stream.
// The caller's stream waits for both at the end; nothing synchronises with the host.
extern "C" int dyk_run_commands_overlap(const DykCommand* cmds, int32_t n, void* stream, int32_t* failed_index) {
    if (!cmds || n < 0) return DYK_ERR_ARG;
    static hipStream_t branch = nullptr, side = nullptr;
    static hipEvent_t ring[64];
    static hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_done = nullptr;
    if (!branch) {
        // the weight gradients are filler work: lowest priority, so that their workgroups do not delay the kernels of
        // the critical chain
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);             // lo = least, hi = greatest priority (numerically smaller)
        const bool prio = true;
        if (hipStreamCreateWithPriority(&branch, hipStreamNonBlocking, prio ? hi : 0) != hipSuccess) return DYK_ERR_HIP;
        if (hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio ? lo : 0) != hipSuccess) return DYK_ERR_HIP;
        for (auto& e : ring)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return DYK_ERR_HIP;
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) return DYK_ERR_HIP;
        if (hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) return DYK_ERR_HIP;
        if (hipEventCreateWithFlags(&ev_done, hipEventDisableTiming) != hipSuccess) return DYK_ERR_HIP;
    }
    hipStream_t main_s = (hipStream_t)stream;
    int ev = 0;
    bool forked = false, used_side = false;
    auto join_branch = [&]() -> bool {
        if (hipEventRecord(ev_join, branch) != hipSuccess || hipStreamWaitEvent(main_s, ev_join, 0) != hipSuccess) return false;
        forked = false;
        return true;
    };
    // issue one command on `target` (weight gradients: on the side stream behind an event of `target`)
    auto issue = [&](int32_t k, hipStream_t target) -> int {
        if ((cmds[k].op == DYK_OP_WGRAD || cmds[k].op == DYK_OP_DW_WGRAD || cmds[k].op == DYK_OP_GRAD_REDUCE) && !(cmds[k].lane & 8)) {
            hipEvent_t e = ring[ev++ & 63];
            if (hipEventRecord(e, target) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) return DYK_ERR_HIP;
            target = side;
            used_side = true;
        }
        const int rc = dyk_run_commands(cmds + k, 1, (void*)target, nullptr);
        if (rc != DYK_OK && failed_index) *failed_index = k;
        return rc;
    };
    int32_t k = 0;
    while (k < n) {
        const int lane = cmds[k].lane;
        if ((lane & 4) && forked && !join_branch()) return DYK_ERR_HIP;
        if (lane & 2) {
            if (forked && !join_branch()) return DYK_ERR_HIP;
            if (hipEventRecord(ev_fork, main_s) != hipSuccess || hipStreamWaitEvent(branch, ev_fork, 0) != hipSuccess) return DYK_ERR_HIP;
            forked = true;
            // The list holds the two independent runs back to back: [k, p) for this stream, [p, q) for the branch.
            // Enqueue them INTERLEAVED -- the host needs ~9 us per command, so issuing 200+ commands of one run
            // first would start the other stream milliseconds late and waste most of the overlap.
            int32_t p = k;
            while (p < n && !(cmds[p].lane & 1) && !(p > k && (cmds[p].lane & 6))) ++p;
            int32_t q = p;
            while (q < n && (cmds[q].lane & 1) && !(cmds[q].lane & 6)) ++q;
            int32_t i = k, j = p;
            while (i < p || j < q) {
                if (i < p) { const int rc = issue(i, main_s); if (rc != DYK_OK) return rc; ++i; }
                if (j < q) { const int rc = issue(j, branch); if (rc != DYK_OK) return rc; ++j; }
            }
            k = q;
            continue;
        }
        const int rc = issue(k, (forked && (lane & 1)) ? branch : main_s);
        if (rc != DYK_OK) return rc;
        ++k;
    }
    if (forked && !join_branch()) return DYK_ERR_HIP;
    if (used_side) {
        if (hipEventRecord(ev_done, side) != hipSuccess || hipStreamWaitEvent(main_s, ev_done, 0) != hipSuccess) return DYK_ERR_HIP;
    }
    return DYK_OK;
}

namespace {
// Replays an issue-ordered schedule: shared by the direct path (dyk_run_schedule) and by stream capture into a hipGraph
// (dyk_schedule_graph_create).  The event pools are per use (direct / capture): an event recorded inside a capture
// belongs to that capture.
struct SchedRuntime {
    hipStream_t aux[2][8] = {};
    hipEvent_t* events = nullptr;
    int n_events = 0;
    hipEvent_t ev_start = nullptr, ev_join[8] = {};
};

// streams and events of a runtime, created up front (never inside a stream capture)
int sched_prepare(SchedRuntime& rt, int32_t n, int32_t n_streams, int32_t low_priority_last) {
    if (!rt.ev_start) {
        if (hipEventCreateWithFlags(&rt.ev_start, hipEventDisableTiming) != hipSuccess) return DYK_ERR_HIP;
        for (auto& e : rt.ev_join)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return DYK_ERR_HIP;
    }
    if (n > rt.n_events) {
        hipEvent_t* grown = (hipEvent_t*)realloc(rt.events, sizeof(hipEvent_t) * (size_t)n);
        if (!grown) return DYK_ERR_HIP;
        rt.events = grown;
        for (int i = rt.n_events; i < n; ++i)
            if (hipEventCreateWithFlags(&rt.events[i], hipEventDisableTiming) != hipSuccess) return DYK_ERR_HIP;
        rt.n_events = n;
    }
    const int lp = low_priority_last ? 1 : 0;
    for (int s = 1; s < n_streams; ++s)
        if (!rt.aux[lp][s]) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            int prio = (lp && s == n_streams - 1) ? lo : 0;
            // (tiered priorities -- second chain highest, filler streams lowest -- cost 11 ms per step: ANY stream created with a
            // non-default priority gets a hardware queue of its own, r04_ab_sched_stream_priorities.log; removed in round 6)
            if (hipStreamCreateWithPriority(&rt.aux[lp][s], hipStreamNonBlocking, prio) != hipSuccess) return DYK_ERR_HIP;
        }
    return DYK_OK;
}

int sched_replay(SchedRuntime& rt, const DykCommand* cmds, const DykSchedEntry* sched, int32_t n, int32_t n_streams,
                 int32_t low_priority_last, hipStream_t main_s, int32_t* failed_index) {
    if (sched_prepare(rt, n, n_streams, low_priority_last) != DYK_OK) return DYK_ERR_HIP;
    const int lp = low_priority_last ? 1 : 0;
    bool used[8] = {};
    if (n_streams > 1 && hipEventRecord(rt.ev_start, main_s) != hipSuccess) return DYK_ERR_HIP;
    static const bool trace = getenv("DYK_SCHED_TRACE") != nullptr;
    for (int32_t k = 0; k < n; ++k) {
        const DykSchedEntry& e = sched[k];
        if (e.stream < 0 || e.stream >= n_streams || e.nwait < 0 || e.nwait > 7) return DYK_ERR_ARG;
        hipStream_t s = e.stream ? rt.aux[lp][e.stream] : main_s;
        if (e.stream && !used[e.stream]) {
            if (hipStreamWaitEvent(s, rt.ev_start, 0) != hipSuccess) return DYK_ERR_HIP;
            used[e.stream] = true;
        }
        for (int q = 0; q < e.nwait; ++q) {
            const int32_t w = e.wait[q];
            if (w < 0 || w >= k) return DYK_ERR_ARG;
            if (hipStreamWaitEvent(s, rt.events[w], 0) != hipSuccess) return DYK_ERR_HIP;
        }
        if (e.cmd >= 0) {
            if (trace) { fprintf(stderr, "sched k=%d cmd=%d op=%d stream=%d nwait=%d w0=%d rec=%d\n", k, e.cmd, cmds[e.cmd].op, e.stream, e.nwait, e.nwait ? e.wait[0] : -1, e.record); fflush(stderr); }
            const int rc = run_entry(cmds, e, (void*)s);
            if (rc != DYK_OK) {
                if (failed_index) *failed_index = e.cmd;
                return rc;
            }
        }
        if (e.record && hipEventRecord(rt.events[k], s) != hipSuccess) return DYK_ERR_HIP;
    }
    for (int s = 1; s < n_streams; ++s)
        if (used[s]) {
            if (hipEventRecord(rt.ev_join[s], rt.aux[lp][s]) != hipSuccess || hipStreamWaitEvent(main_s, rt.ev_join[s], 0) != hipSuccess)
                return DYK_ERR_HIP;
        }
    return DYK_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// One host thread per stream.  A launch costs the host ~9 us (runtime call + argument marshalling); a train step is
// ~1150 launches, so ONE issuing thread needs ~10 ms per step -- as long as the whole batch-1 step takes on the GPU, and a
// third of the batch-16 step during which later streams start late.  The schedule already says which stream every command
// belongs to, so each library stream gets its own issuing thread (the caller's thread serves the caller's stream); the
// only host-side ordering needed is that a stream's hipStreamWaitEvent must come after the hipEventRecord it refers to
// has been issued: a per-entry flag, spun on by the waiter.  Entries are numbered in a topological order and a thread
// never waits on a later entry, so the flags cannot deadlock.
struct IssuePool {
    struct Job {
        SchedRuntime* rt = nullptr;
        const DykCommand* cmds = nullptr;
        const DykSchedEntry* sched = nullptr;
        int32_t n = 0, n_streams = 0, lp = 0, device = 0;
    } job;
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    uint64_t generation = 0;
    int pending = 0;
    std::vector<std::atomic<uint8_t>> issued;       // entry k: its completion event has been recorded (host side)
    std::atomic<int> error{0};
    std::atomic<int> failed_cmd{-1};
    bool stop = false;

    int run_stream(int sidx, hipStream_t s) {
        const Job& j = job;
        bool first = true;
        for (int32_t k = 0; k < j.n; ++k) {
            const DykSchedEntry& e = j.sched[k];
            if (e.stream != sidx) continue;
            if (error.load(std::memory_order_relaxed)) return DYK_ERR_HIP;
            if (first && sidx) {
                if (hipStreamWaitEvent(s, j.rt->ev_start, 0) != hipSuccess) return DYK_ERR_HIP;
                first = false;
            }
            for (int q = 0; q < e.nwait; ++q) {
                const int32_t w = e.wait[q];
                if (w < 0 || w >= k) return DYK_ERR_ARG;
                int spins = 0;
                while (!issued[w].load(std::memory_order_acquire)) {
                    if (error.load(std::memory_order_relaxed)) return DYK_ERR_HIP;
                    if (++spins > 64) std::this_thread::yield();
                }
                if (hipStreamWaitEvent(s, j.rt->events[w], 0) != hipSuccess) return DYK_ERR_HIP;
            }
            if (e.cmd >= 0) {
                const int rc = run_entry(j.cmds, e, (void*)s);
                if (rc != DYK_OK) { failed_cmd.store(e.cmd); return rc; }
            }
            if (e.record) {
                if (hipEventRecord(j.rt->events[k], s) != hipSuccess) return DYK_ERR_HIP;
                issued[k].store(1, std::memory_order_release);
            }
        }
        if (sidx && !first && hipEventRecord(j.rt->ev_join[sidx], s) != hipSuccess) return DYK_ERR_HIP;
        return first && sidx ? 1 : DYK_OK;        // 1: this stream had no work (no join event recorded)
    }

    int worker_status[8] = {};

    void worker(int sidx) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv_go.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            const bool mine = sidx < job.n_streams;
            lk.unlock();
            int rc = 1;
            if (mine) {
                (void)hipSetDevice(job.device);
                rc = run_stream(sidx, job.rt->aux[job.lp][sidx]);
                if (rc < 0) error.store(rc);
            }
            lk.lock();
            worker_status[sidx] = rc;
            if (--pending == 0) cv_done.notify_one();
        }
    }

    int run(SchedRuntime& rt, const DykCommand* cmds, const DykSchedEntry* sched, int32_t n, int32_t n_streams, int32_t lp,
            hipStream_t main_s, int32_t* failed_index) {
        if (sched_prepare(rt, n, n_streams, lp) != DYK_OK) return DYK_ERR_HIP;
        if ((int)issued.size() < n) issued = std::vector<std::atomic<uint8_t>>((size_t)n + 256);
        for (int32_t k = 0; k < n; ++k) issued[k].store(0, std::memory_order_relaxed);
        error.store(0);
        failed_cmd.store(-1);
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (n_streams > 1 && hipEventRecord(rt.ev_start, main_s) != hipSuccess) return DYK_ERR_HIP;
        {
            std::lock_guard<std::mutex> lk(m);
            while ((int)threads.size() < 7) { const int idx = (int)threads.size() + 1; threads.emplace_back([this, idx] { worker(idx); }); }
            job.rt = &rt; job.cmds = cmds; job.sched = sched; job.n = n; job.n_streams = n_streams; job.lp = lp; job.device = dev;
            pending = (int)threads.size();
            ++generation;
        }
        cv_go.notify_all();
        int rc0 = run_stream(0, main_s);
        if (rc0 < 0) error.store(rc0);
        {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        const int err = error.load();
        if (err) {
            if (failed_index) *failed_index = failed_cmd.load();
            return err;
        }
        for (int s = 1; s < n_streams; ++s)
            if (worker_status[s] == DYK_OK && hipStreamWaitEvent(main_s, rt.ev_join[s], 0) != hipSuccess) return DYK_ERR_HIP;
        return DYK_OK;
    }
};

struct SchedGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};
}  // namespace

namespace {
// the table must be replayable before anything is enqueued: a wait on an entry that records no event would spin forever
// in the threaded path (and race silently in the single-threaded one)
int sched_validate(const DykSchedEntry* sched, int32_t n, int32_t n_streams) {
    for (int32_t k = 0; k < n; ++k) {
        const DykSchedEntry& e = sched[k];
        if (e.stream < 0 || e.stream >= n_streams || e.nwait < 0 || e.nwait > 7) return DYK_ERR_ARG;
        for (int q = 0; q < e.nwait; ++q) {
            const int32_t w = e.wait[q];
            if (w < 0 || w >= k || !sched[w].record || sched[w].stream == e.stream) return DYK_ERR_ARG;
        }
    }
    return DYK_OK;
}
constexpr int DYK_MAX_DEVICES = 16;
SchedRuntime& sched_runtime(int dev) {
    static SchedRuntime rts[DYK_MAX_DEVICES];
    return rts[dev];
}
}  // namespace

// Library stream `idx` (1 .. 7) of the current device's schedule runtime, created on first use: host code that has work of
// its own for a side stream (dyk/optim.py: the early part of the fused optimizer step, the rebuild of the transposed weight
// packs) runs it on one of THESE instead of creating another stream -- the HIP runtime multiplexes all streams of a process
// onto four hardware queues, and a fifth stream shares a queue with whichever stream the runtime picks.
extern "C" int dyk_sched_stream(int32_t idx, void** stream_out) {
    if (idx < 1 || idx > 7 || !stream_out) return DYK_ERR_ARG;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DYK_MAX_DEVICES) return DYK_ERR_HIP;
    SchedRuntime& rt = sched_runtime(dev);
    if (!rt.aux[0][idx] && hipStreamCreateWithPriority(&rt.aux[0][idx], hipStreamNonBlocking, 0) != hipSuccess) return DYK_ERR_HIP;
    *stream_out = (void*)rt.aux[0][idx];
    return DYK_OK;
}

extern "C" int dyk_run_schedule(const DykCommand* cmds, const DykSchedEntry* sched, int32_t n, int32_t n_streams,
                                int32_t low_priority_last, void* stream, int32_t* failed_index) {
    if (!cmds || !sched || n < 0 || n_streams < 1 || n_streams > 8) return DYK_ERR_ARG;
    if (sched_validate(sched, n, n_streams) != DYK_OK) return DYK_ERR_ARG;
    // streams, events and issuing threads belong to ONE device: a process that runs plans on several GPUs (or moves a
    // model from cuda:0 to cuda:1) gets a runtime per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DYK_MAX_DEVICES) return DYK_ERR_HIP;
    SchedRuntime& rt = sched_runtime(dev);
    // DYK_ISSUE_THREADS=0: everything is issued by the calling thread (one launch at a time)
    static const bool threaded = !(getenv("DYK_ISSUE_THREADS") && getenv("DYK_ISSUE_THREADS")[0] == '0');
    int rc;
    if (threaded && n_streams > 1) {
        static IssuePool* pools[DYK_MAX_DEVICES] = {};   // (leaked on purpose: worker threads outlive static destruction)
        if (!pools[dev]) pools[dev] = new IssuePool();
        rc = pools[dev]->run(rt, cmds, sched, n, n_streams, low_priority_last ? 1 : 0, (hipStream_t)stream, failed_index);
    } else {
        rc = sched_replay(rt, cmds, sched, n, n_streams, low_priority_last, (hipStream_t)stream, failed_index);
    }
    if (rc != DYK_OK) {
        // error path: whatever the library streams hold must not outlive the call (the caller will reuse the arenas)
        const int lp = low_priority_last ? 1 : 0;
        for (int s = 1; s < n_streams; ++s)
            if (rt.aux[lp][s]) (void)hipStreamSynchronize(rt.aux[lp][s]);
    }
    return rc;
}

// A command range as a hipGraph built from its DEPENDENCY lists (dyk/sched.py): every command is captured on its own
// into a small child graph (single-stream capture of one dyk_run_commands call) and added to the master graph as a child
// node behind the nodes of the commands it depends on.  No cross-stream events take part in a capture -- the multi-stream
// capture of the issue-ordered schedule is what one would write first, but the HIP runtime PyTorch-ROCm 7.0 ships recurses
// without end in hip::Stream::EndCapture as soon as two library streams have waited on each other's events (found with
// rocgdb; the stand-alone ROCm 7.2 runtime handles it).  Kernel arguments are frozen at capture: capture again when a
// pointer in a descriptor changes (dyk/plan.py keys its graphs on the per-call pointers).
extern "C" int dyk_dag_graph_create(const DykCommand* cmds, int32_t n, const int32_t* dep_off, const int32_t* dep_idx,
                                    void** graph_out, int32_t* failed_index) {
    if (!cmds || n <= 0 || !dep_off || !graph_out) return DYK_ERR_ARG;
    static hipStream_t origin = nullptr;
    constexpr bool dbg = false;
    if (!origin && hipStreamCreateWithFlags(&origin, hipStreamNonBlocking) != hipSuccess) return DYK_ERR_HIP;
    SchedGraph* g = new SchedGraph();
    if (hipGraphCreate(&g->graph, 0) != hipSuccess) { delete g; return DYK_ERR_HIP; }
    hipGraphNode_t* nodes = (hipGraphNode_t*)calloc((size_t)n, sizeof(hipGraphNode_t));
    hipGraphNode_t* deps = (hipGraphNode_t*)calloc((size_t)n, sizeof(hipGraphNode_t));
    int rc = DYK_OK;
    for (int32_t i = 0; i < n && rc == DYK_OK; ++i) {
        hipGraph_t child = nullptr;
        if (hipStreamBeginCapture(origin, hipStreamCaptureModeRelaxed) != hipSuccess) { rc = DYK_ERR_HIP; break; }
        const int rc1 = dyk_run_commands(cmds + i, 1, (void*)origin, nullptr);
        const hipError_t e = hipStreamEndCapture(origin, &child);
        if (rc1 != DYK_OK || e != hipSuccess || !child) {
            if (failed_index) *failed_index = i;
            rc = rc1 != DYK_OK ? rc1 : DYK_ERR_HIP;
            if (child) (void)hipGraphDestroy(child);
            break;
        }
        int nd = 0;
        for (int32_t q = dep_off[i]; q < dep_off[i + 1]; ++q) {
            const int32_t j = dep_idx[q];
            if (j < 0 || j >= i) { rc = DYK_ERR_ARG; break; }
            deps[nd++] = nodes[j];
        }
        if (rc == DYK_OK && hipGraphAddChildGraphNode(&nodes[i], g->graph, deps, (size_t)nd, child) != hipSuccess) {
            if (failed_index) *failed_index = i;
            rc = DYK_ERR_HIP;
        }
        (void)hipGraphDestroy(child);             // the child node holds its own clone
    }
    free(nodes);
    free(deps);
    if (rc == DYK_OK) {
        const hipError_t e2 = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
        if (dbg) fprintf(stderr, "dyk graph: %d commands, instantiate: %s\n", n, hipGetErrorString(e2));
        if (e2 != hipSuccess) rc = DYK_ERR_HIP;
    }
    if (rc != DYK_OK) {
        (void)hipGetLastError();
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        return rc;
    }
    *graph_out = g;
    return DYK_OK;
}

extern "C" int dyk_schedule_graph_launch(void* graph, void* stream) {
    SchedGraph* g = (SchedGraph*)graph;
    if (!g || !g->exec) return DYK_ERR_ARG;
    return hipGraphLaunch(g->exec, (hipStream_t)stream) == hipSuccess ? DYK_OK : DYK_ERR_HIP;
}

extern "C" int dyk_schedule_graph_destroy(void* graph) {
    SchedGraph* g = (SchedGraph*)graph;
    if (!g) return DYK_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return DYK_OK;
}

extern "C" int dyk_run_commands_timed(const DykCommand* cmds, int32_t n, void* stream, float* ms_out) {
    if (!cmds || n <= 0 || !ms_out) return DYK_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t* ev = new hipEvent_t[n + 1];
    for (int i = 0; i <= n; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) { delete[] ev; return DYK_ERR_HIP; }
    int rc = DYK_OK;
    hipEventRecord(ev[0], s);
    for (int k = 0; k < n && rc == DYK_OK; ++k) {
        rc = dyk_run_commands(cmds + k, 1, stream, nullptr);
        hipEventRecord(ev[k + 1], s);
    }
    if (hipStreamSynchronize(s) != hipSuccess) rc = DYK_ERR_HIP;
    if (rc == DYK_OK)
        for (int k = 0; k < n; ++k)
            if (hipEventElapsedTime(&ms_out[k], ev[k], ev[k + 1]) != hipSuccess) { rc = DYK_ERR_HIP; break; }
    for (int i = 0; i <= n; ++i) hipEventDestroy(ev[i]);
    delete[] ev;
    return rc;
}

extern "C" int dyk_run_schedule_timed(const DykCommand* cmds, const DykSchedEntry* sched, int32_t n, void* stream,
                                      float* ms_out) {
    if (!cmds || !sched || n <= 0 || !ms_out) return DYK_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t* ev = new hipEvent_t[n + 1];
    for (int i = 0; i <= n; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) { delete[] ev; return DYK_ERR_HIP; }
    int rc = DYK_OK;
    hipEventRecord(ev[0], s);
    for (int k = 0; k < n && rc == DYK_OK; ++k) {
        if (sched[k].cmd >= 0) rc = run_entry(cmds, sched[k], stream);
        hipEventRecord(ev[k + 1], s);
    }
    if (hipStreamSynchronize(s) != hipSuccess) rc = DYK_ERR_HIP;
    if (rc == DYK_OK)
        for (int k = 0; k < n; ++k)
            if (hipEventElapsedTime(&ms_out[k], ev[k], ev[k + 1]) != hipSuccess) { rc = DYK_ERR_HIP; break; }
    for (int i = 0; i <= n; ++i) hipEventDestroy(ev[i]);
    delete[] ev;
    return rc;
}
