// isa.hip -- inverse suffix array (isa[SA[i]] = i) without random HBM writes.
//
// A direct scatter issues n random 4-byte writes (measured 140 ms for n = 3.1 G: every write is a partial line).
// SA is a permutation, so every window of W consecutive positions receives exactly W entries -- the bin sizes of
// a partition by position are known in advance, no histogram pass is needed and the order inside a bin is
// irrelevant.  Two levels:
//   level 1: (position, index) pairs partitioned on position >> 24 into bins of exactly 2^24 pairs
//            (one streaming pass over SA; a workgroup reserves room per bin with one global atomic);
//   level 2: per level-1 bin (128 MiB of pairs, processed while it is still in the last-level cache):
//            partition on the next 8 bits into a small scratch, then scatter into the bin's 64 MiB window of isa
//            through 256 KiB sub-windows that live in L2, so isa lines leave the cache complete.
// Used by the PREFIX_DOUBLING refinement (lms_sort.hip: kiss_exact_refine).
#include "kiss_internal.hpp"
#include <cstdlib>
#include <vector>

namespace {

constexpr int IB_THREADS = 1024;
constexpr int IB_ITEMS = 16;
constexpr int IB_TILE = IB_THREADS * IB_ITEMS; // 16384 pairs = 128 KiB of LDS staging
constexpr int L1_SHIFT = 24;                   // level-1 bin = 2^24 positions
constexpr int L2_SHIFT = 16;                   // level-2 bin = 2^16 positions (256 KiB of isa)
constexpr int CUR_STRIDE = 64;                 // one bin cursor per 256 B: the reservations spread over channels

// Partition one tile by an 8-bit digit.  FROM_SA: items are SA[base + i] (pair = position << 32 | index), else
// items are pairs.  digit = (position >> shift) & 255.  Bin d of the output starts at out + ((uint64_t)d <<
// bin_shift) and `cursor[d]` counts the pairs already placed there.
template <bool FROM_SA>
__device__ __forceinline__ void isa_partition_tile(const uint32_t *__restrict__ SA, const uint64_t *__restrict__ pairs_in,
                                                   uint64_t base, uint64_t count, int shift, int bin_shift,
                                                   uint32_t *__restrict__ cursor, uint64_t *__restrict__ out, int idx_shift,
                                                   const uint32_t *__restrict__ binbase, uint32_t bstride, uint32_t binsub,
                                                   uint32_t tile)
// FROM_SA: the pair's index is SA[i] >> idx_shift.
// binbase (optional): bins of unequal, known size -- bin d starts at out + binbase[d * bstride] - binsub
// (the sparse form, kiss_rank_build_lms: not every index has an entry) instead of at out + (d << bin_shift)
{
    __shared__ uint64_t stage[IB_TILE];
    __shared__ uint32_t lcnt[256];  // items of this tile per bin
    __shared__ uint32_t loff[256];  // exclusive prefix of lcnt
    __shared__ uint32_t gpos[256];  // where this tile's items of bin d start inside bin d
    __shared__ uint32_t wsum[IB_THREADS / 64];
    const uint64_t tile_base = (uint64_t)tile * IB_TILE;
    const uint32_t tile_count = (uint32_t)(count - tile_base < (uint64_t)IB_TILE ? count - tile_base : IB_TILE);
    if (threadIdx.x < 256) lcnt[threadIdx.x] = 0;
    __syncthreads();
    uint64_t pr[IB_ITEMS];
    uint32_t rk[IB_ITEMS];
    // the order of the items inside a bin is irrelevant: a thread takes IB_ITEMS consecutive ones (16-byte loads)
    const uint32_t l0 = threadIdx.x * IB_ITEMS;
    if (l0 + IB_ITEMS <= tile_count) {
        const uint64_t g0 = tile_base + l0;
        if (FROM_SA) {
#pragma unroll
            for (int q = 0; q < IB_ITEMS / 4; q++) {
                const uint4 t = *reinterpret_cast<const uint4 *>(SA + base + g0 + 4 * q);
                pr[4 * q] = ((uint64_t)(t.x >> idx_shift) << 32) | (base + g0 + 4 * q);
                pr[4 * q + 1] = ((uint64_t)(t.y >> idx_shift) << 32) | (base + g0 + 4 * q + 1);
                pr[4 * q + 2] = ((uint64_t)(t.z >> idx_shift) << 32) | (base + g0 + 4 * q + 2);
                pr[4 * q + 3] = ((uint64_t)(t.w >> idx_shift) << 32) | (base + g0 + 4 * q + 3);
            }
        } else {
#pragma unroll
            for (int q = 0; q < IB_ITEMS / 2; q++) {
                const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(pairs_in + g0 + 2 * q);
                pr[2 * q] = t.x;
                pr[2 * q + 1] = t.y;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < IB_ITEMS; j++) {
            const uint32_t li = l0 + (uint32_t)j;
            if (li < tile_count) {
                const uint64_t g = tile_base + li;
                pr[j] = FROM_SA ? (((uint64_t)(SA[base + g] >> idx_shift) << 32) | (base + g)) : pairs_in[g];
            } else
                pr[j] = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < IB_ITEMS; j++) {
        const uint32_t li = l0 + (uint32_t)j;
        if (li < tile_count) {
            const uint32_t d = (uint32_t)(pr[j] >> (32 + shift)) & 255u;
            // rank inside (tile, bin), < 16384.  A wave whose 64 items share one bin (sorted stretches of SA, e.g.
            // runs of one base) takes one LDS atomic instead of 64 colliding ones.
            const uint32_t d0 = __shfl(d, 0, 64);
            if (__all(d == d0) && __popcll(__ballot(1)) == 64) {
                uint32_t b0 = 0;
                if (lane_id() == 0) b0 = atomicAdd(&lcnt[d], 64u);
                rk[j] = (d << 16) | (__shfl(b0, 0, 64) + lane_id());
            } else
                rk[j] = (d << 16) | atomicAdd(&lcnt[d], 1u);
        } else
            rk[j] = 0xFFFFFFFFu;
    }
    __syncthreads();
    {
        // exclusive scan of lcnt over the 256 digits (threads 0..255 = 4 waves) + one global reservation per bin
        const bool dig = threadIdx.x < 256;
        const uint32_t c = dig ? lcnt[threadIdx.x] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = __shfl_up(inc, dd, 64);
            if ((int)lane_id() >= dd) inc += o;
        }
        if (dig && lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        if (dig) {
            uint32_t start = inc - c;
            for (int w = 0; w < (int)(threadIdx.x >> 6); w++) start += wsum[w];
            loff[threadIdx.x] = start;
            gpos[threadIdx.x] = c ? atomicAdd(&cursor[threadIdx.x * CUR_STRIDE], c) : 0u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IB_ITEMS; j++)
        if (rk[j] != 0xFFFFFFFFu) stage[loff[rk[j] >> 16] + (rk[j] & 0xFFFFu)] = pr[j];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < IB_ITEMS; r++) {
        const uint32_t idx = threadIdx.x + (uint32_t)r * IB_THREADS;
        if (idx < tile_count) {
            const uint64_t v = stage[idx];
            const uint32_t d = (uint32_t)(v >> (32 + shift)) & 255u;
            const uint64_t bin0 = binbase ? (uint64_t)(binbase[d * bstride] - binsub) : ((uint64_t)d << bin_shift);
            out[bin0 + gpos[d] + (idx - loff[d])] = v;
        }
    }
}

template <bool FROM_SA>
__global__ __launch_bounds__(IB_THREADS) void k_isa_partition(const uint32_t *__restrict__ SA,
                                                             const uint64_t *__restrict__ pairs_in, uint64_t base,
                                                             uint64_t count, int shift, int bin_shift,
                                                             uint32_t *__restrict__ cursor, uint64_t *__restrict__ out,
                                                             int idx_shift, const uint32_t *__restrict__ binbase,
                                                             uint32_t bstride)
{
    isa_partition_tile<FROM_SA>(SA, pairs_in, base, count, shift, bin_shift, cursor, out, idx_shift, binbase, bstride,
                                binbase ? binbase[0] : 0u, blockIdx.x);
}

// Level 2 of the sparse form for ALL level-1 bins in one launch (93 bins at chm13 size: three short launches per bin
// left the GPU idle between them for as long as they ran).  bt: the bounds table (bin b's pairs are [bt[256 b],
// bt[256 (b + 1)]) in pairs_in, its sub-bin s starts at bt[256 b + s] in out); tile0[b]: first workgroup of bin b.
__global__ __launch_bounds__(IB_THREADS) void k_rank_partition_all(const uint64_t *__restrict__ pairs_in,
                                                                  const uint32_t *__restrict__ bt,
                                                                  const uint32_t *__restrict__ tile0, uint32_t bins,
                                                                  uint32_t *__restrict__ cursor, uint64_t *__restrict__ out)
{
    uint32_t b = 0;
    while (b + 1 < bins && tile0[b + 1] <= blockIdx.x) b++;
    const uint32_t lo = bt[256 * b], cnt = bt[256 * (b + 1)] - lo;
    isa_partition_tile<false>(nullptr, pairs_in + lo, 0ull, cnt, L2_SHIFT, L2_SHIFT, cursor + (uint64_t)b * 256 * CUR_STRIDE,
                              out, 0, bt + 256 * b, 1u, 0u, blockIdx.x - tile0[b]);
}

__global__ __launch_bounds__(256) void k_isa_write(const uint64_t *__restrict__ pairs, uint64_t count,
                                                   uint32_t *__restrict__ isa)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) {
        const uint64_t v = pairs[i];
        isa[v >> 32] = (uint32_t)v;
    }
}

__global__ __launch_bounds__(256) void k_isa_direct(const uint32_t *__restrict__ SA, uint64_t count,
                                                    uint32_t *__restrict__ isa)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) isa[SA[i]] = (uint32_t)i;
}

} // namespace

// SA: a permutation of [0, total).  Scratch (ctx-owned, allocated on first use): pairs1 = round_up(total, 2^24)
// u64, pairs2 = 2^24 u64 + the bin cursors.
int kiss_isa_build(kiss_hip_ctx *ctx, const uint32_t *SA, uint64_t total, uint32_t *isa)
{
    if (total == 0) return KISS_HIP_OK;
    uint64_t direct_max = 1ull << 25; // 128 MiB of isa: the plain scatter stays in the last-level cache
    if (ctx->opts.isa_direct_max) direct_max = ctx->opts.isa_direct_max; // (hooks build)
    if (total <= direct_max) {
        KTimer t(ctx, KISS_HIP_K_ISA, total);
        hipLaunchKernelGGL(k_isa_direct, dim3((unsigned)div_up(total, 256)), dim3(256), 0, ctx->stream, SA, total, isa);
        KCHECK(hipGetLastError());
        return KISS_HIP_OK;
    }
    const uint64_t bins = div_up(total, 1ull << L1_SHIFT);
    if (bins > 256) return KINTERNAL(); // total < 2^32
    const uint64_t need1 = bins << L1_SHIFT;
    if (ctx->pairs_cap < need1) {
        if (ctx->pairs1) (void)hipFree(ctx->pairs1);
        ctx->pairs1 = nullptr;
        ctx->ws_bytes -= ctx->pairs_cap * sizeof(uint64_t);
        ctx->pairs_cap = 0;
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, need1 * sizeof(uint64_t));
        if (e != hipSuccess) {
            ctx->last_hip_error = (int)e;
            return KISS_HIP_E_NOMEM;
        }
        ctx->pairs1 = (uint64_t *)p;
        ctx->pairs_cap = need1;
        ctx->ws_bytes += need1 * sizeof(uint64_t);
    }
    if (!ctx->pairs2) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, (sizeof(uint64_t) << L1_SHIFT) + 512 * CUR_STRIDE * sizeof(uint32_t));
        if (e != hipSuccess) {
            ctx->last_hip_error = (int)e;
            return KISS_HIP_E_NOMEM;
        }
        ctx->pairs2 = (uint64_t *)p;
        ctx->ws_bytes += (sizeof(uint64_t) << L1_SHIFT) + 512 * CUR_STRIDE * sizeof(uint32_t);
    }
    uint32_t *cursor = reinterpret_cast<uint32_t *>(ctx->pairs2 + (1ull << L1_SHIFT)); // 256 strided cursors for level 1, 256 for level 2
    KTRY(kiss_zero_u32(ctx, cursor, 256 * CUR_STRIDE));
    {
        KTimer t(ctx, KISS_HIP_K_ISA, total);
        hipLaunchKernelGGL((k_isa_partition<true>), dim3((unsigned)div_up(total, IB_TILE)), dim3(IB_THREADS), 0, ctx->stream,
                           SA, nullptr, 0ull, total, L1_SHIFT, L1_SHIFT, cursor, ctx->pairs1, 0, nullptr, 0u);
        KCHECK(hipGetLastError());
    }
    KTimer t(ctx, KISS_HIP_K_ISA, total);
    for (uint64_t b = 0; b < bins; b++) {
        const uint64_t lo = b << L1_SHIFT;
        const uint64_t cnt = total - lo < (1ull << L1_SHIFT) ? total - lo : (1ull << L1_SHIFT);
        KTRY(kiss_zero_u32(ctx, cursor + 256 * CUR_STRIDE, 256 * CUR_STRIDE));
        hipLaunchKernelGGL((k_isa_partition<false>), dim3((unsigned)div_up(cnt, IB_TILE)), dim3(IB_THREADS), 0, ctx->stream,
                           nullptr, ctx->pairs1 + lo, 0ull, cnt, L2_SHIFT, L2_SHIFT, cursor + 256 * CUR_STRIDE, ctx->pairs2, 0,
                           nullptr, 0u);
        // pairs2 is bin-major with 2^16-pair bins; in a short last bin the sub-bins are not full: write bin by bin
        if (cnt == (1ull << L1_SHIFT)) {
            hipLaunchKernelGGL(k_isa_write, dim3((unsigned)div_up(cnt, 256)), dim3(256), 0, ctx->stream, ctx->pairs2, cnt,
                               isa);
        } else {
            for (uint64_t s = 0; s < 256 && (s << L2_SHIFT) < cnt; s++) {
                const uint64_t c2 = cnt - (s << L2_SHIFT) < (1ull << L2_SHIFT) ? cnt - (s << L2_SHIFT) : (1ull << L2_SHIFT);
                hipLaunchKernelGGL(k_isa_write, dim3((unsigned)div_up(c2, 256)), dim3(256), 0, ctx->stream,
                                   ctx->pairs2 + (s << L2_SHIFT), c2, isa);
            }
        }
        KCHECK(hipGetLastError());
    }
    return KISS_HIP_OK;
}

// ---- the sparse form: rank[L[i] >> 1] = i for the m LMS positions in L (no two LMS positions are neighbours, so
// position >> 1 is a collision-free index: half the array of a full inverse).  Not every index has an entry, so the bin
// sizes are not the bin widths -- but the ascending LMS list says how many positions fall below any bound: one binary
// search per 2^16-index sub-bin gives the table both partition levels take their bin starts from.
namespace {
__global__ __launch_bounds__(256) void k_lms_bounds(const uint32_t *__restrict__ lms_asc, uint64_t m, uint32_t entries,
                                                    uint32_t *__restrict__ bt)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= entries) return;
    const uint64_t bound = (uint64_t)j << (L2_SHIFT + 1); // positions below it have index < j * 2^16
    uint64_t lo = 0, hi = m;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)lms_asc[mid] < bound) lo = mid + 1;
        else hi = mid;
    }
    bt[j] = (uint32_t)lo;
}
// One level-2 sub-bin (2^16 indexes, its pairs contiguous in `pairs`) leaves as two complete 128 KiB windows of the rank
// array: a workgroup clears its window in LDS (0xFFFFFFFF = no LMS position here), drops the pairs of its half into it and
// writes it out whole.  The array needs no clearing pass and HBM sees full lines only -- six of ten indexes have no entry,
// and 4-byte writes into cleared lines made this step 7.3 ms at chm13 size where the dense inverse, three times the
// entries, takes 20 ms.
constexpr int RW_THREADS = 1024;
constexpr int RW_SHIFT = 15;
__global__ __launch_bounds__(RW_THREADS) void k_rank_window_write(const uint64_t *__restrict__ pairs,
                                                                 const uint32_t *__restrict__ bt, // bounds of all sub-bins
                                                                 uint64_t r_words, uint32_t *__restrict__ rank)
{
    __shared__ uint32_t win[1 << RW_SHIFT];
    const uint32_t sub = blockIdx.x; // sub-bin over all level-1 bins; its two windows one after the other: the pairs
                                     // (~300 KB) come out of L2 the second time
    const uint32_t lo = bt[sub], hi = bt[sub + 1];
    for (uint32_t half = 0; half < 2; half++) {
        const uint64_t idx0 = ((uint64_t)sub << L2_SHIFT) + ((uint64_t)half << RW_SHIFT);
        if (idx0 >= r_words) return;
        for (uint32_t i = threadIdx.x; i < (1u << RW_SHIFT) / 4; i += RW_THREADS)
            reinterpret_cast<uint4 *>(win)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        __syncthreads();
        for (uint32_t i = lo + threadIdx.x; i < hi; i += RW_THREADS) {
            const uint64_t v = pairs[i];
            const uint32_t idx = (uint32_t)(v >> 32);
            if (((idx >> RW_SHIFT) & 1u) == half) win[idx & ((1u << RW_SHIFT) - 1u)] = (uint32_t)v;
        }
        __syncthreads();
        const uint64_t left = r_words - idx0;
        const uint32_t cnt = left < (1ull << RW_SHIFT) ? (uint32_t)left : (1u << RW_SHIFT);
        for (uint32_t i = threadIdx.x * 4; i < cnt; i += RW_THREADS * 4) {
            if (i + 4 <= cnt) {
                *reinterpret_cast<uint4 *>(rank + idx0 + i) = *reinterpret_cast<const uint4 *>(win + i);
            } else {
                for (uint32_t e = i; e < cnt; e++) rank[idx0 + e] = win[e];
            }
        }
        __syncthreads(); // the window is cleared again next
    }
}

// The same table without the ascending list (the multi-device entry: device 0 holds the gathered sorted list, but only its
// own slice of the ascending one): count the list's indexes per sub-bin -- one LDS histogram per workgroup over all
// sub-bins (<= 32 768 of them: 128 KiB), its non-zero counts added to the table -- and scan.
constexpr int BH_THREADS = 1024;
__global__ __launch_bounds__(BH_THREADS) void k_lms_bounds_hist(const uint32_t *__restrict__ L, uint64_t m, uint32_t subs,
                                                               uint32_t *__restrict__ bt) // bt[1 + s] += #indexes in sub-bin s
{
    __shared__ uint32_t h[32768];
    for (uint32_t i = threadIdx.x; i < subs; i += BH_THREADS) h[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * BH_THREADS + threadIdx.x; i < m; i += (uint64_t)gridDim.x * BH_THREADS)
        atomicAdd(&h[(L[i] >> 1) >> L2_SHIFT], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < subs; i += BH_THREADS)
        if (h[i]) atomicAdd(&bt[1 + i], h[i]);
}

__global__ __launch_bounds__(256) void k_rank_direct(const uint32_t *__restrict__ L, uint64_t count,
                                                     uint32_t *__restrict__ rank)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) rank[L[i] >> 1] = (uint32_t)i;
}
} // namespace

// L: the m LMS positions in sorted order; lms_asc: the same positions ascending (or null: the table is counted from L).  rank: (n >> 1) + 1 words, every one of
// them written (0xFFFFFFFF where no LMS position maps to it).
// pairs1, pairs2: m u64 of scratch each; small: 256 KiB + 64 KiB per level-1 bin (bin cursors, bounds table, tile table:
// <= 8.3 MiB).
int kiss_rank_build_lms(kiss_hip_ctx *ctx, const uint32_t *L, const uint32_t *lms_asc, uint64_t m, uint64_t n,
                        uint32_t *rank, uint64_t *pairs1, uint64_t *pairs2, uint32_t *small, uint64_t small_words)
{
    if (m == 0) return KISS_HIP_OK;
    uint64_t direct_max = 1ull << 25;
    if (ctx->opts.isa_direct_max) direct_max = ctx->opts.isa_direct_max; // (hooks build)
    const uint64_t idx_total = (n >> 1) + 1;
    if (m <= direct_max || m < (1ull << 16)) {
        KTimer t(ctx, KISS_HIP_K_ISA, m);
        KTRY(kiss_fill_u32(ctx, rank, 0xFFFFFFFFu, idx_total)); // (a kernel, not hipMemsetAsync: see kiss_fill_u32)
        hipLaunchKernelGGL(k_rank_direct, dim3((unsigned)div_up(m, 256)), dim3(256), 0, ctx->stream, L, m, rank);
        KCHECK(hipGetLastError());
        return KISS_HIP_OK;
    }
    const uint64_t bins = div_up(idx_total, 1ull << L1_SHIFT);
    if (bins > 128) return KINTERNAL();
    const uint32_t sub_total = (uint32_t)(bins * 256);
    if (512ull * CUR_STRIDE + sub_total + 1 + bins + 1 + bins * 256 * CUR_STRIDE > small_words) return KINTERNAL();
    uint32_t *cursor = small;                     // 2 x 256 strided cursors
    uint32_t *bt = small + 512 * CUR_STRIDE;      // sub_total + 1 entries
    if (lms_asc) {
        KTimer t(ctx, KISS_HIP_K_ISA, m);
        hipLaunchKernelGGL(k_lms_bounds, dim3((unsigned)div_up(sub_total + 1, 256)), dim3(256), 0, ctx->stream, lms_asc, m,
                           sub_total + 1, bt);
        KCHECK(hipGetLastError());
    } else { // counts per sub-bin, then their running sums (bt[0] = 0)
        if (sub_total > 32768) return KINTERNAL();
        KTRY(kiss_zero_u32(ctx, bt, sub_total + 1));
        {
            KTimer t(ctx, KISS_HIP_K_ISA, m);
            const uint64_t wgs = div_up(m, (uint64_t)BH_THREADS * 64);
            hipLaunchKernelGGL(k_lms_bounds_hist, dim3((unsigned)(wgs < 1024 ? wgs : 1024)), dim3(BH_THREADS), 0, ctx->stream, L, m,
                               sub_total, bt);
            KCHECK(hipGetLastError());
        }
        // inclusive running sums in place: an exclusive scan of bt[1 ..] shifted by one entry would need a second array;
        // the table is small -- scan it on the host while it is being fetched anyway (below)
    }
    std::vector<uint32_t> h_bt(sub_total + 1);
    KCHECK(hipMemcpyAsync(h_bt.data(), bt, (sub_total + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));
    if (!lms_asc) { // counts -> bounds, and back to the device
        for (uint32_t i = 1; i <= sub_total; i++) h_bt[i] += h_bt[i - 1];
        KCHECK(hipMemcpyAsync(bt, h_bt.data(), (sub_total + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
    }
    if (h_bt[sub_total] != m) return KINTERNAL();
    KTRY(kiss_zero_u32(ctx, cursor, 256 * CUR_STRIDE));
    {
        KTimer t(ctx, KISS_HIP_K_ISA, m);
        hipLaunchKernelGGL((k_isa_partition<true>), dim3((unsigned)div_up(m, IB_TILE)), dim3(IB_THREADS), 0, ctx->stream, L,
                           nullptr, 0ull, m, L1_SHIFT, L1_SHIFT, cursor, pairs1, 1, bt, 256u);
        KCHECK(hipGetLastError());
    }
    // level 2 for all bins at once, then every sub-bin as two complete windows of the rank array
    std::vector<uint32_t> h_tile0(bins + 1);
    uint32_t tiles = 0;
    for (uint64_t b = 0; b < bins; b++) {
        const uint64_t cnt = h_bt[256 * (b + 1)] - h_bt[256 * b];
        if (cnt > (1ull << L1_SHIFT)) return KINTERNAL();
        h_tile0[b] = tiles;
        tiles += (uint32_t)div_up(cnt, IB_TILE);
    }
    h_tile0[bins] = tiles;
    uint32_t *d_tile0 = bt + sub_total + 1;
    uint32_t *cursor2 = d_tile0 + bins + 1; // bins x 256 strided cursors
    KCHECK(hipMemcpyAsync(d_tile0, h_tile0.data(), (bins + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    KTRY(kiss_zero_u32(ctx, cursor2, bins * 256 * CUR_STRIDE));
    KTimer t(ctx, KISS_HIP_K_ISA, m);
    if (tiles)
        hipLaunchKernelGGL(k_rank_partition_all, dim3(tiles), dim3(IB_THREADS), 0, ctx->stream, pairs1, bt, d_tile0, (uint32_t)bins,
                           cursor2, pairs2);
    hipLaunchKernelGGL(k_rank_window_write, dim3(sub_total), dim3(RW_THREADS), 0, ctx->stream, pairs2, bt, idx_total, rank);
    KCHECK(hipGetLastError());
    KCHECK(hipStreamSynchronize(ctx->stream)); // h_tile0 goes out of scope
    return KISS_HIP_OK;
}
