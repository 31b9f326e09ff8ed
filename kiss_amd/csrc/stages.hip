// stages.hip -- stage-level entry points of the C ABI for the sharded (multi-GPU) suffix sort.
//
// One process per GPU (torch.distributed / RCCL owns the exchange; this library owns the arithmetic).
// Pipeline of SURVEY.md section 8(e), all pointers are device pointers on the ctx's device:
//   rank r : stage_classify(slice r of the text)        -> local ascending LMS list (key32, pos)
//            stage_key_hist                               -> histogram of the first 16 key bits  [all_reduce]
//            stage_partition(splitters)                   -> list grouped by destination rank    [all_to_all]
//            stage_sort(received list)                    -> k-ordered positions of its key range [gather]
//   rank 0 : stage_induce(all sorted pieces, near-end list, global counts) -> SA
// Stability across the exchange (receiver concatenates in source-rank order = ascending text position) keeps
// the reference's position tie-break (include/biovoltron/algo/sort/kiss1_core.hpp:131-133).
#include "kiss_internal.hpp"
#include <mutex>
#include <cstring>

namespace {

constexpr int ST_THREADS = 256;

struct Splitters {
    uint32_t s[63];
    int count;
};

__global__ __launch_bounds__(ST_THREADS) void k_key_hist(const uint64_t *__restrict__ keys, uint64_t count, int shift,
                                                        unsigned long long *__restrict__ hist)
{
    uint64_t i = (uint64_t)blockIdx.x * ST_THREADS + threadIdx.x;
    if (i >= count) return;
    atomicAdd(&hist[keys[i] >> shift], 1ull);
}

// the same with one private histogram per workgroup in LDS (bits <= 14: 64 KiB of 32-bit counters), flushed once: one
// global atomic per key on 2^16 bins took 49 ms for the 906 M keys of a chm13-size text (bench.py --sharded-timings)
__global__ __launch_bounds__(ST_THREADS) void k_key_hist_lds(const uint64_t *__restrict__ keys, uint64_t count, int shift,
                                                            uint32_t bins, unsigned long long *__restrict__ hist)
{
    extern __shared__ uint32_t lh[];
    for (uint32_t b = threadIdx.x; b < bins; b += ST_THREADS) lh[b] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * ST_THREADS + threadIdx.x; i < count; i += (uint64_t)gridDim.x * ST_THREADS)
        atomicAdd(&lh[keys[i] >> shift], 1u);
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins; b += ST_THREADS)
        if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
}

__global__ __launch_bounds__(ST_THREADS) void k_group_ids(const uint64_t *__restrict__ keys, uint64_t count, int shift,
                                                         Splitters sp, uint32_t *__restrict__ group)
{
    uint64_t i = (uint64_t)blockIdx.x * ST_THREADS + threadIdx.x;
    if (i >= count) return;
    const uint32_t top = (uint32_t)(keys[i] >> shift);
    uint32_t g = 0;
    for (int t = 0; t < sp.count; t++) g += (sp.s[t] <= top) ? 1u : 0u;
    group[i] = g;
}

} // namespace

uint64_t kiss_depth_of(uint64_t n, uint32_t k)
{
    if ((uint64_t)k >= n) return 0;
    return (uint64_t)KISS_STRIDE * ((uint64_t)k / KISS_STRIDE + 1);
}
static uint64_t depth_of(uint64_t n, uint32_t k) { return kiss_depth_of(n, k); }

// histogram (u64[2^bits]) of the first `bits` key bits; queued on ctx->stream, not synchronised
int kiss_key_hist(kiss_hip_ctx *ctx, const uint64_t *d_keys, uint64_t count, int bits, uint64_t *d_hist)
{
    KTRY(kiss_zero_u32(ctx, d_hist, 2ull << bits));
    if (count && bits <= 14) {
        const uint64_t blocks = div_up(count, (uint64_t)ST_THREADS * 64);
        hipLaunchKernelGGL(k_key_hist_lds, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(ST_THREADS),
                           (size_t)sizeof(uint32_t) << bits, ctx->stream, d_keys, count, 64 - bits, 1u << bits,
                           (unsigned long long *)d_hist);
        KCHECK(hipGetLastError());
    } else if (count) {
        hipLaunchKernelGGL(k_key_hist, dim3((unsigned)div_up(count, ST_THREADS)), dim3(ST_THREADS), 0, ctx->stream, d_keys,
                           count, 64 - bits, (unsigned long long *)d_hist);
        KCHECK(hipGetLastError());
    }
    return KISS_HIP_OK;
}

// stable partition of (key, pos) by destination group g = #{splitters <= first `bits` key bits}: one radix pass on the
// group id.  The ids live in lms_sorted_far / lms_ctx_far (m_cap entries each, free until the sort writes its output).
// Queued on ctx->stream; groups == 1 moves nothing (the caller keeps using the input buffers).
int kiss_partition_by_splitters(kiss_hip_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_pos, uint64_t count, int bits,
                                const uint32_t *splitters, int groups, uint64_t *d_keys_out, uint32_t *d_pos_out)
{
    if (count == 0 || groups <= 1) return KISS_HIP_OK;
    if (count > ctx->m_cap) return KINTERNAL();
    Splitters sp;
    sp.count = groups - 1;
    for (int t = 0; t < groups - 1; t++) sp.s[t] = splitters[t];
    uint32_t *gid0 = ctx->lms_sorted_far, *gid1 = ctx->lms_ctx_far;
    hipLaunchKernelGGL(k_group_ids, dim3((unsigned)div_up(count, ST_THREADS)), dim3(ST_THREADS), 0, ctx->stream, d_keys,
                       count, 64 - bits, sp, gid0);
    KCHECK(hipGetLastError());
    if (count == 1) {
        KCHECK(hipMemcpyAsync(d_keys_out, d_keys, 8, hipMemcpyDeviceToDevice, ctx->stream));
        KCHECK(hipMemcpyAsync(d_pos_out, d_pos, 4, hipMemcpyDeviceToDevice, ctx->stream));
        return KISS_HIP_OK;
    }
    RadixBufs rb;
    rb.key[0] = const_cast<uint64_t *>(d_keys);
    rb.key[1] = d_keys_out;
    rb.pos[0] = const_cast<uint32_t *>(d_pos);
    rb.pos[1] = d_pos_out;
    rb.seg[0] = gid0;
    rb.seg[1] = gid1;
    int res = 0;
    KTRY(kiss_radix_sort(ctx, rb, count, 64, 8, &res));
    if (res != 1) return KINTERNAL();
    return KISS_HIP_OK;
}

extern "C" {

int kiss_hip_stage_classify(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t k, uint64_t lo, uint64_t hi,
                            uint64_t counts13[13], void *stream)
{
    if (!ctx || !d_S || !counts13 || n == 0 || n > ctx->max_n || lo > hi || hi > n) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    kiss_opts_refresh(ctx);
    // one device phase at a time per device and process, like sort_dev (api.hip): a stage call returns
    // synchronised, so the lock is held for exactly its device work
    std::unique_lock<std::mutex> device_lock;
    if (!ctx->opts.no_serialize) device_lock = std::unique_lock<std::mutex>(kiss_device_mutex(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    ctx->hfar = ctx->hmerged = nullptr; // (tie flags of an exact-order sort_dev: never a stage call's, see api.hip)
    ctx->h_depth = 0;
    KTRY(kiss_workspace_ready(ctx));
    std::memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->stats.n = n;
    ctx->stats.k = k;
    ctx->n = n;
    KTRY(kiss_pack_text(ctx, d_S, n));
    KTRY(kiss_classify(ctx, n, depth_of(n, k), lo, hi));
    for (int i = 0; i < 12; i++) counts13[i] = ctx->counts[i];
    counts13[12] = ctx->m_far;
    KCHECK(hipStreamSynchronize(ctx->stream));
    return KISS_HIP_OK;
}

int kiss_hip_stage_local_lms(kiss_hip_ctx *ctx, uint64_t *d_keys_out, uint32_t *d_pos_out, uint64_t *m_local,
                             uint64_t *m_far_local)
{
    if (!ctx || !m_local || !m_far_local) return KISS_HIP_E_INVALID;
    *m_local = ctx->m;
    *m_far_local = ctx->m_far;
    if (d_keys_out && d_pos_out && ctx->m) { // second call: copy the list into caller buffers
        KCHECK(hipSetDevice(ctx->device));
        KCHECK(hipMemcpyAsync(d_keys_out, ctx->keyA, ctx->m * 8, hipMemcpyDeviceToDevice, ctx->stream));
        KCHECK(hipMemcpyAsync(d_pos_out, ctx->lms_pos, ctx->m * 4, hipMemcpyDeviceToDevice, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
    }
    return KISS_HIP_OK;
}

int kiss_hip_stage_view(kiss_hip_ctx *ctx, int which, void **d_ptr, uint64_t *capacity)
{
    if (!ctx || !d_ptr) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    KTRY(kiss_workspace_ready(ctx));
    void *p = nullptr;
    switch (which) {
    case KISS_HIP_VIEW_LOCAL_KEYS: p = ctx->keyA; break;
    case KISS_HIP_VIEW_LOCAL_POS: p = ctx->lms_pos; break;
    case KISS_HIP_VIEW_PART_KEYS: p = ctx->keyB; break;
    case KISS_HIP_VIEW_PART_POS: p = ctx->posB; break;
    case KISS_HIP_VIEW_SORTED: p = ctx->lms_sorted_far; break;
    case KISS_HIP_VIEW_SORTED_CTX: p = ctx->lms_ctx_far; break;
    default: return KISS_HIP_E_INVALID;
    }
    *d_ptr = p;
    if (capacity) *capacity = ctx->m_cap;
    return KISS_HIP_OK;
}

int kiss_hip_stage_reserve(kiss_hip_ctx *ctx, uint64_t lms_capacity)
{
    if (!ctx || lms_capacity > ctx->max_n / 2 + 2) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    KTRY(kiss_workspace_ready(ctx));
    if (lms_capacity <= ctx->m_cap) return KISS_HIP_OK;
    KCHECK(hipStreamSynchronize(ctx->stream));
    return kiss_lms_reserve(ctx, lms_capacity + lms_capacity / 64 + 1024);
}

int kiss_hip_stage_key_hist(kiss_hip_ctx *ctx, const uint64_t *d_keys, uint64_t count, int bits, uint64_t *d_hist,
                            void *stream)
{
    if (!ctx || !d_hist || bits < 1 || bits > 24 || (count && !d_keys)) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    kiss_opts_refresh(ctx);
    // one device phase at a time per device and process, like sort_dev (api.hip): a stage call returns
    // synchronised, so the lock is held for exactly its device work
    std::unique_lock<std::mutex> device_lock;
    if (!ctx->opts.no_serialize) device_lock = std::unique_lock<std::mutex>(kiss_device_mutex(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    ctx->hfar = ctx->hmerged = nullptr; // (tie flags of an exact-order sort_dev: never a stage call's, see api.hip)
    ctx->h_depth = 0;
    KTRY(kiss_workspace_ready(ctx));
    KTRY(kiss_key_hist(ctx, d_keys, count, bits, d_hist));
    KCHECK(hipStreamSynchronize(ctx->stream));
    return KISS_HIP_OK;
}

int kiss_hip_stage_partition(kiss_hip_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_pos, uint64_t count, int bits,
                             const uint32_t *splitters, int groups, uint64_t *d_keys_out, uint32_t *d_pos_out,
                             void *stream)
{
    if (!ctx || groups < 1 || groups > 64 || bits < 1 || bits > 24 || (groups > 1 && !splitters)) return KISS_HIP_E_INVALID;
    if (count && (!d_keys || !d_pos || !d_keys_out || !d_pos_out)) return KISS_HIP_E_INVALID;
    if (count > ctx->m_cap) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    kiss_opts_refresh(ctx);
    // one device phase at a time per device and process, like sort_dev (api.hip): a stage call returns
    // synchronised, so the lock is held for exactly its device work
    std::unique_lock<std::mutex> device_lock;
    if (!ctx->opts.no_serialize) device_lock = std::unique_lock<std::mutex>(kiss_device_mutex(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    ctx->hfar = ctx->hmerged = nullptr; // (tie flags of an exact-order sort_dev: never a stage call's, see api.hip)
    ctx->h_depth = 0;
    KTRY(kiss_workspace_ready(ctx));
    if (count == 0) return KISS_HIP_OK;
    if (groups == 1) { // nothing to move (callers that hold views of the ctx's own buffers skip this call altogether)
        if (d_keys_out != d_keys) KCHECK(hipMemcpyAsync(d_keys_out, d_keys, count * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_pos_out != d_pos) KCHECK(hipMemcpyAsync(d_pos_out, d_pos, count * 4, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        if (d_keys_out == d_keys || d_pos_out == d_pos) return KISS_HIP_E_INVALID;
        KTRY(kiss_partition_by_splitters(ctx, d_keys, d_pos, count, bits, splitters, groups, d_keys_out, d_pos_out));
    }
    return kiss_radix_check(ctx);
}

int kiss_hip_stage_sort(kiss_hip_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_pos, uint64_t count, uint64_t n,
                        uint32_t k, uint32_t *d_sorted_out, uint32_t *d_ctx_out, void *stream)
{
    if (!ctx || n == 0 || n != ctx->n) return KISS_HIP_E_INVALID; // stage_classify packed the text of this ctx
    if (count && (!d_keys || !d_pos || !d_sorted_out)) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    kiss_opts_refresh(ctx);
    // one device phase at a time per device and process, like sort_dev (api.hip): a stage call returns
    // synchronised, so the lock is held for exactly its device work
    std::unique_lock<std::mutex> device_lock;
    if (!ctx->opts.no_serialize) device_lock = std::unique_lock<std::mutex>(kiss_device_mutex(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    ctx->hfar = ctx->hmerged = nullptr; // (tie flags of an exact-order sort_dev: never a stage call's, see api.hip)
    ctx->h_depth = 0;
    KTRY(kiss_workspace_ready(ctx));
    if (count > ctx->m_cap) KTRY(kiss_lms_reserve(ctx, count + count / 64 + 1024));
    // the ctx's own buffers (kiss_hip_stage_view) are sorted where they are: no copy in, no copy out
    const bool in_place = count && d_keys == ctx->keyA && d_pos == ctx->lms_pos;
    if (count && !in_place) {
        KCHECK(hipMemcpyAsync(ctx->keyA, d_keys, count * 8, hipMemcpyDeviceToDevice, ctx->stream));
        KCHECK(hipMemcpyAsync(ctx->lms_pos, d_pos, count * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // the digit counts of the emit pass stay good only for the very list it emitted (one rank, nothing exchanged)
    if (!(in_place && ctx->rx_ghist_count == count && count == ctx->m_far)) ctx->rx_ghist_count = 0;
    ctx->m = ctx->m_far = count;
    {
        const int rc = kiss_lms_sort(ctx, n, k, depth_of(n, k));
        if (rc == KISS_INTERNAL_TOO_DEEP) { // exact order on very long repeats: the caller switches to k = 256 + doubling
            (void)hipStreamSynchronize(ctx->stream);
            return KISS_HIP_E_DEEP;
        }
        if (rc) return rc;
    }
    if (count) {
        if (d_sorted_out != ctx->lms_sorted_far)
            KCHECK(hipMemcpyAsync(d_sorted_out, ctx->lms_sorted_far, count * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_ctx_out && d_ctx_out != ctx->lms_ctx_far) // context words from the key payload (0 = to be gathered)
            KCHECK(hipMemcpyAsync(d_ctx_out, ctx->lms_ctx_far, count * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    KTRY(kiss_radix_check(ctx));
    ktimer_collect(ctx);
    return KISS_HIP_OK;
}

static int stage_induce_impl(kiss_hip_ctx *ctx, uint64_t n, uint32_t k, uint32_t exact_h0, const uint32_t *d_far_sorted,
                             const uint32_t *d_far_ctx, uint64_t m_far, const uint32_t *d_near_pos, uint64_t near_count,
                             const uint64_t counts12[12], uint32_t *d_SA, void *stream, int *exact_out)
{
    if (!ctx || !counts12 || !d_SA || n == 0 || n != ctx->n) return KISS_HIP_E_INVALID;
    if ((m_far && !d_far_sorted) || (near_count && !d_near_pos)) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    kiss_opts_refresh(ctx);
    // one device phase at a time per device and process, like sort_dev (api.hip): a stage call returns
    // synchronised, so the lock is held for exactly its device work
    std::unique_lock<std::mutex> device_lock;
    if (!ctx->opts.no_serialize) device_lock = std::unique_lock<std::mutex>(kiss_device_mutex(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    ctx->hfar = ctx->hmerged = nullptr; // (tie flags of an exact-order sort_dev: never a stage call's, see api.hip)
    ctx->h_depth = 0;
    KTRY(kiss_workspace_ready(ctx));
    const uint64_t m = m_far + near_count;
    const bool in_place = d_far_sorted == ctx->lms_sorted_far; // views of the ctx's own buffers: nothing to copy
    if (m > ctx->m_cap) {
        if (in_place) return KISS_HIP_E_INVALID; // (kiss_hip_stage_reserve before the pieces are gathered)
        KTRY(kiss_lms_reserve(ctx, m + m / 64 + 1024));
    }
    if (m_far) {
        if (!in_place)
            KCHECK(hipMemcpyAsync(ctx->lms_sorted_far, d_far_sorted, m_far * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (d_far_ctx && d_far_ctx != ctx->lms_ctx_far)
            KCHECK(hipMemcpyAsync(ctx->lms_ctx_far, d_far_ctx, m_far * 4, hipMemcpyDeviceToDevice, ctx->stream));
        else if (!d_far_ctx) // no context words came along: gather them all; nothing is known about ties either, so all are tainted
            KTRY(kiss_fill_u32(ctx, ctx->lms_ctx_far, KISS_CTX_TAINT, m_far)); // (a kernel: see kiss_fill_u32)
    }
    if (near_count && d_near_pos != ctx->lms_pos + m_far) // kiss_place_lms reads the near-end suffixes as the tail of the ascending list
        KCHECK(hipMemcpyAsync(ctx->lms_pos + m_far, d_near_pos, near_count * 4, hipMemcpyDeviceToDevice, ctx->stream));
    for (int i = 0; i < 12; i++) ctx->counts[i] = counts12[i];
    ctx->m = m;
    ctx->m_far = m_far;
    ctx->stats.m = m;
    const uint64_t depth = depth_of(n, k);
    KTRY(kiss_place_lms(ctx, n, k, depth));
    bool resolved = false;
    if (exact_h0 && !ctx->opts.no_lms_exact) {
        // the gathered list is h0-ordered: exact order of the LMS suffixes by rank doubling before the induction, as in the
        // one-device path (api.hip: sort_dev); tie flags by comparison, bin sizes of the rank array counted from the list
        ctx->lms_pos_complete = false;
        ctx->hfar = nullptr;
        KTRY(kiss_lms_exact_refine(ctx, n, exact_h0, d_SA, &resolved));
    }
    KTRY(kiss_induce(ctx, n, d_SA));
    KTRY(kiss_radix_check(ctx));
    if (resolved) ctx->stats.refine_form = 1;
    if (exact_out) *exact_out = resolved ? 1 : 0;
    ktimer_collect(ctx);
    return KISS_HIP_OK;
}

int kiss_hip_stage_induce(kiss_hip_ctx *ctx, uint64_t n, uint32_t k, const uint32_t *d_far_sorted,
                          const uint32_t *d_far_ctx, uint64_t m_far, const uint32_t *d_near_pos, uint64_t near_count,
                          const uint64_t counts12[12], uint32_t *d_SA, void *stream)
{
    return stage_induce_impl(ctx, n, k, 0, d_far_sorted, d_far_ctx, m_far, d_near_pos, near_count, counts12, d_SA, stream,
                             nullptr);
}

int kiss_hip_stage_induce_exact(kiss_hip_ctx *ctx, uint64_t n, uint32_t h0, const uint32_t *d_far_sorted,
                                const uint32_t *d_far_ctx, uint64_t m_far, const uint32_t *d_near_pos, uint64_t near_count,
                                const uint64_t counts12[12], uint32_t *d_SA, void *stream, int *exact_out)
{
    if (!exact_out) return KISS_HIP_E_INVALID;
    *exact_out = 0;
    if (h0 < 32 || n < 4ull * h0 + 1024) return KISS_HIP_E_UNSUPPORTED;
    return stage_induce_impl(ctx, n, h0, h0, d_far_sorted, d_far_ctx, m_far, d_near_pos, near_count, counts12, d_SA, stream,
                             exact_out);
}

} // extern "C"

int kiss_hip_stage_refine_exact(kiss_hip_ctx *ctx, uint64_t n, uint32_t h0, uint32_t *d_SA, void *stream)
{
    if (!ctx || !d_SA || n == 0 || n != ctx->n) return KISS_HIP_E_INVALID; // stage_classify packed the text of this ctx
    if (h0 < 32 || n < 4ull * h0 + 1024) return KISS_HIP_E_UNSUPPORTED;
    KCHECK(hipSetDevice(ctx->device));
    kiss_opts_refresh(ctx);
    // one device phase at a time per device and process, like sort_dev (api.hip): a stage call returns
    // synchronised, so the lock is held for exactly its device work
    std::unique_lock<std::mutex> device_lock;
    if (!ctx->opts.no_serialize) device_lock = std::unique_lock<std::mutex>(kiss_device_mutex(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    ctx->hfar = ctx->hmerged = nullptr; // (tie flags of an exact-order sort_dev: never a stage call's, see api.hip)
    ctx->h_depth = 0;
    KTRY(kiss_workspace_ready(ctx));
    KTRY(kiss_exact_refine(ctx, n, h0, d_SA));
    KCHECK(hipStreamSynchronize(ctx->stream));
    KTRY(kiss_radix_check(ctx));
    ktimer_collect(ctx);
    return KISS_HIP_OK;
}
