// lms_sort.hip -- k-ordered sort of the LMS suffixes (the "far" ones: p + D <= n).
//
// Contract (reference comparator, include/biovoltron/algo/sort/kiss1_core.hpp:94-135, applied after the
// stable 10-mer bucketing of :41-83): for two LMS suffixes that both have D = 125*(k/125+1) bases left
// the order is (first D bases, then text position).  k >= n means the exact suffix order (D unbounded;
// past-the-end bases read as 'A' exactly like the reference's zero padding, structs.hpp:94-96).
//
// GPU formulation: MSD refinement in rounds of 32 bases (one u64 key per suffix per round):
//   round r: every still-tied suffix fetches bases [32r, 32r+32) -> stable radix sort by
//   (segment id, key) -> neighbours with equal (segment, key) stay tied, singletons retire.
// Stability + ascending initial order gives the position tie-break for free.
#include "kiss_internal.hpp"
#include <cstring>
#include <utility>

namespace {

constexpr int LS_THREADS = 256;

// key of round r for every active item
__global__ __launch_bounds__(LS_THREADS) void k_gather_keys(const uint64_t *__restrict__ pk, uint64_t n,
                                                           const uint32_t *__restrict__ pos, uint64_t count,
                                                           uint64_t depth_off, uint64_t mask,
                                                           uint64_t *__restrict__ key)
{
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    uint64_t q = (uint64_t)pos[i] + depth_off;
    uint64_t k = (q < n) ? kiss_key32(pk, q) : 0ull;
    key[i] = k & mask;
}

// flags[i] = (survivor << 32) | surviving_head
template <bool HAS_SEG>
__global__ __launch_bounds__(LS_THREADS) void k_flag(const uint64_t *__restrict__ key,
                                                    const uint32_t *__restrict__ seg, uint64_t count, int last_round,
                                                    uint64_t *__restrict__ flags)
{
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    uint64_t k = key[i];
    uint32_t s = HAS_SEG ? seg[i] : 0u;
    bool head = (i == 0) || key[i - 1] != k || (HAS_SEG && seg[i - 1] != s);
    bool nhead = (i + 1 == count) || key[i + 1] != k || (HAS_SEG && seg[i + 1] != s);
    bool single = head && nhead;
    bool surv = !single && !last_round;
    flags[i] = ((uint64_t)(surv ? 1u : 0u) << 32) | (uint64_t)((surv && head) ? 1u : 0u);
}

// retire singletons (and everything in the last round) into out[slot]; compact survivors
template <bool HAS_SLOT>
__global__ __launch_bounds__(LS_THREADS) void k_compact(const uint64_t *__restrict__ key,
                                                       const uint32_t *__restrict__ seg_unused,
                                                       const uint32_t *__restrict__ pos,
                                                       const uint32_t *__restrict__ slot, uint64_t count,
                                                       const uint64_t *__restrict__ flags_in, // original flags
                                                       const uint64_t *__restrict__ ex,       // exclusive scan of flags
                                                       uint32_t *__restrict__ npos, uint32_t *__restrict__ nslot,
                                                       uint32_t *__restrict__ nseg, uint32_t *__restrict__ out)
{
    (void)key;
    (void)seg_unused;
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    uint64_t f = flags_in[i];
    uint64_t e = ex[i];
    uint32_t sl = HAS_SLOT ? slot[i] : (uint32_t)i;
    uint32_t p = pos[i];
    if (f >> 32) {
        uint32_t ni = (uint32_t)(e >> 32);
        npos[ni] = p;
        nslot[ni] = sl;
        nseg[ni] = (uint32_t)(e & 0xFFFFFFFFull) + (uint32_t)(f & 1ull) - 1u;
    } else {
        out[sl] = p;
    }
}

__global__ void k_last_total(const uint64_t *__restrict__ flags, const uint64_t *__restrict__ ex, uint64_t count,
                             uint64_t *__restrict__ total)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) total[0] = ex[count - 1] + flags[count - 1];
}

} // namespace

int kiss_lms_sort(kiss_hip_ctx *ctx, uint64_t n, uint32_t k, uint64_t depth)
{
    (void)k;
    const uint64_t m_far = ctx->m_far;
    ctx->stats.lms_rounds = 0;
    ctx->stats.sort_item_rounds = 0;
    if (m_far == 0) return KISS_HIP_OK;
    if (m_far == 1) {
        KCHECK(hipMemcpyAsync(ctx->lms_sorted_far, ctx->lms_pos, sizeof(uint32_t), hipMemcpyDeviceToDevice,
                              ctx->stream));
        return KISS_HIP_OK;
    }
    // flags needs 2 x count u64 (flags + scan); ctx->flags has 2*m_cap entries
    uint64_t *flags = ctx->flags;
    uint64_t *ex = ctx->flags + ctx->m_cap;

    RadixBufs rb;
    rb.key[0] = ctx->keyA;
    rb.key[1] = ctx->keyB;
    rb.pos[0] = ctx->posA;
    rb.pos[1] = ctx->posB;
    rb.seg[0] = ctx->segA;
    rb.seg[1] = ctx->segB;
    uint32_t *slot_cur = ctx->slotA, *slot_nxt = ctx->slotB;

    uint64_t count = m_far;
    uint64_t nseg = 1;
    bool has_slot = false;
    // round 0: keys were produced by the classify/emit kernel in keyA; positions are lms_pos
    KCHECK(hipMemcpyAsync(ctx->posA, ctx->lms_pos, count * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));

    for (uint64_t round = 0;; round++) {
        const uint64_t off = round * 32;
        // bases of this round that still count: depth - off (all 32 when unbounded)
        uint64_t rem = depth ? depth - off : 32;
        if (rem > 32) rem = 32;
        const bool last_round = depth ? (off + 32 >= depth) : false;
        const uint64_t mask = rem >= 32 ? ~0ull : (~0ull << (64 - 2 * rem));
        const int key_lo_bit = (int)(64 - 2 * rem);

        if (round > 0) {
            KTimer t(ctx, KISS_HIP_K_KEYGATHER, count);
            hipLaunchKernelGGL(k_gather_keys, dim3((unsigned)div_up(count, LS_THREADS)), dim3(LS_THREADS), 0,
                               ctx->stream, ctx->pk, n, rb.pos[0], count, off, mask, rb.key[0]);
            KCHECK(hipGetLastError());
        } else if (mask != ~0ull) {
            // depth < 32 can not happen (D >= 125), but keep round 0 honest
            KTimer t(ctx, KISS_HIP_K_KEYGATHER, count);
            hipLaunchKernelGGL(k_gather_keys, dim3((unsigned)div_up(count, LS_THREADS)), dim3(LS_THREADS), 0,
                               ctx->stream, ctx->pk, n, rb.pos[0], count, off, mask, rb.key[0]);
            KCHECK(hipGetLastError());
        }
        int seg_bits = 0;
        if (nseg > 1) {
            uint64_t v = nseg - 1;
            while (v) {
                seg_bits++;
                v >>= 1;
            }
        }
        int res = 0;
        KTRY(kiss_radix_sort(ctx, rb, count, key_lo_bit, seg_bits, &res));
        ctx->stats.lms_rounds++;
        ctx->stats.sort_item_rounds += count;
        const uint64_t *skey = rb.key[res];
        const uint32_t *sseg = rb.seg[res];
        const uint32_t *spos = rb.pos[res];
        const int other = res ^ 1;
        {
            KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
            if (seg_bits > 0)
                hipLaunchKernelGGL((k_flag<true>), dim3((unsigned)div_up(count, LS_THREADS)), dim3(LS_THREADS), 0,
                                   ctx->stream, skey, sseg, count, (int)last_round, flags);
            else
                hipLaunchKernelGGL((k_flag<false>), dim3((unsigned)div_up(count, LS_THREADS)), dim3(LS_THREADS), 0,
                                   ctx->stream, skey, sseg, count, (int)last_round, flags);
            KCHECK(hipGetLastError());
        }
        KTRY(kiss_scan_u64(ctx, flags, ex, count));
        {
            KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
            // survivors go to the buffers of index 0 for the next round: make sure we do not read and write
            // the same buffer: compact from `res` into `other`, then swap roles so that index 0 is current.
            if (has_slot)
                hipLaunchKernelGGL((k_compact<true>), dim3((unsigned)div_up(count, LS_THREADS)), dim3(LS_THREADS), 0,
                                   ctx->stream, skey, sseg, spos, slot_cur, count, flags, ex, rb.pos[other], slot_nxt,
                                   rb.seg[other], ctx->lms_sorted_far);
            else
                hipLaunchKernelGGL((k_compact<false>), dim3((unsigned)div_up(count, LS_THREADS)), dim3(LS_THREADS), 0,
                                   ctx->stream, skey, sseg, spos, slot_cur, count, flags, ex, rb.pos[other], slot_nxt,
                                   rb.seg[other], ctx->lms_sorted_far);
            hipLaunchKernelGGL(k_last_total, dim3(1), dim3(64), 0, ctx->stream, flags, ex, count,
                               (uint64_t *)ctx->d_small);
            KCHECK(hipGetLastError());
        }
        KCHECK(hipMemcpyAsync(ctx->h_pinned, ctx->d_small, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
        uint64_t tot;
        std::memcpy(&tot, ctx->h_pinned, sizeof tot);
        const uint64_t survivors = tot >> 32;
        const uint64_t new_segs = tot & 0xFFFFFFFFull;
        if (survivors == 0) break;
        if (last_round) return KISS_HIP_E_INTERNAL;
        // make buffer index 0 the one holding the survivors
        if (other != 0) {
            std::swap(rb.key[0], rb.key[1]);
            std::swap(rb.pos[0], rb.pos[1]);
            std::swap(rb.seg[0], rb.seg[1]);
        }
        std::swap(slot_cur, slot_nxt);
        has_slot = true;
        count = survivors;
        nseg = new_segs;
        if (!depth && off > n + 64) return KISS_HIP_E_INTERNAL; // unbounded mode must have terminated by now
    }
    return KISS_HIP_OK;
}
