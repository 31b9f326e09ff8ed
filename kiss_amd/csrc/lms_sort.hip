// lms_sort.hip -- k-ordered sort of the LMS suffixes (the "far" ones: p + D <= n).
//
// Contract (reference comparator, include/biovoltron/algo/sort/kiss1_core.hpp:94-135, applied after the
// stable 10-mer bucketing of :41-83): for two LMS suffixes that both have D = 125*(k/125+1) bases left
// the order is (first D bases, then text position).  k >= n means the exact suffix order (D unbounded;
// past-the-end bases read as 'A' exactly like the reference's zero padding, structs.hpp:94-96).
//
// GPU formulation: MSD refinement in rounds.
//   round 0 : stable LSD radix sort of all suffixes on their first 20 bases (5 passes of 8 bits);
//             suffixes that share those 20 bases form a segment, singletons retire to their final slot.
//   round r : every still-tied suffix fetches the next 32 bases (one u64 key);
//             segments of <= SMALL_SEG suffixes are ordered by brute-force counting inside the segment
//             (one thread per suffix, the segment's keys come from L1/L2), larger segments go through
//             the radix sort on (segment id, key); equal neighbours stay tied, singletons retire.
// Every step is stable and the initial order is ascending text position, which yields the reference's
// position tie-break (kiss1_core.hpp:131-133) once the depth D is exhausted.
#include "kiss_internal.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>

namespace {

constexpr int LS_THREADS = 256;
constexpr uint32_t SMALL_SEG = 64;
constexpr int ROUND0_BASES = 20;

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

// small segments: order inside the segment by counting; big segments: flag for the radix path.
// big[i] = (1 << 32) | (first item of a big segment)
__global__ __launch_bounds__(LS_THREADS) void k_seg_rank(const uint64_t *__restrict__ key,
                                                        const uint32_t *__restrict__ pos,
                                                        const uint32_t *__restrict__ seg,
                                                        const uint32_t *__restrict__ segstart, uint64_t count,
                                                        uint64_t *__restrict__ okey, uint32_t *__restrict__ opos,
                                                        uint64_t *__restrict__ big, uint32_t *__restrict__ nbig)
{
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    bool isbig = false;
    if (i < count) {
        uint32_t sg = seg[i];
        uint32_t a = segstart[sg], b = segstart[sg + 1];
        if (b - a <= SMALL_SEG) {
            uint64_t ki = key[i];
            uint32_t r = 0;
            for (uint32_t j = a; j < b; j++) {
                uint64_t kj = key[j];
                r += (kj < ki || (kj == ki && j < (uint32_t)i)) ? 1u : 0u;
            }
            okey[a + r] = ki;
            opos[a + r] = pos[i];
            big[i] = 0;
        } else {
            isbig = true;
            big[i] = (1ull << 32) | (uint64_t)((uint32_t)i == a ? 1u : 0u);
        }
    }
    uint64_t bm = __ballot(isbig);
    if (bm && lane_id() == 0) atomicAdd(nbig, (uint32_t)__popcll(bm));
}

__global__ __launch_bounds__(LS_THREADS) void k_big_compact(const uint64_t *__restrict__ key,
                                                           const uint32_t *__restrict__ pos, uint64_t count,
                                                           const uint64_t *__restrict__ big,
                                                           const uint64_t *__restrict__ ex,
                                                           uint64_t *__restrict__ bkey, uint32_t *__restrict__ bpos,
                                                           uint32_t *__restrict__ bseg, uint32_t *__restrict__ bslot)
{
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    uint64_t f = big[i];
    if (!(f >> 32)) return;
    uint64_t e = ex[i];
    uint32_t kx = (uint32_t)(e >> 32);
    bkey[kx] = key[i];
    bpos[kx] = pos[i];
    bslot[kx] = (uint32_t)i;
    bseg[kx] = (uint32_t)(e & 0xFFFFFFFFull) + (uint32_t)(f & 1ull) - 1u;
}

__global__ __launch_bounds__(LS_THREADS) void k_big_writeback(const uint64_t *__restrict__ bkey,
                                                             const uint32_t *__restrict__ bpos,
                                                             const uint32_t *__restrict__ bslot, uint64_t nbig,
                                                             uint64_t *__restrict__ okey, uint32_t *__restrict__ opos)
{
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= nbig) return;
    uint32_t s = bslot[i];
    okey[s] = bkey[i];
    opos[s] = bpos[i];
}

// flags[i] = (survivor << 32) | surviving_head ; keys are compared on bits [cmp_shift, 64)
template <bool HAS_SEG>
__global__ __launch_bounds__(LS_THREADS) void k_flag(const uint64_t *__restrict__ key,
                                                    const uint32_t *__restrict__ seg, uint64_t count, int cmp_shift,
                                                    int last_round, uint64_t *__restrict__ flags)
{
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    uint64_t k = key[i] >> cmp_shift;
    uint32_t s = HAS_SEG ? seg[i] : 0u;
    bool head = (i == 0) || (key[i - 1] >> cmp_shift) != k || (HAS_SEG && seg[i - 1] != s);
    bool nhead = (i + 1 == count) || (key[i + 1] >> cmp_shift) != k || (HAS_SEG && seg[i + 1] != s);
    bool single = head && nhead;
    bool surv = !single && !last_round;
    flags[i] = ((uint64_t)(surv ? 1u : 0u) << 32) | (uint64_t)((surv && head) ? 1u : 0u);
}

// retire singletons (and everything in the last round) into out[slot]; compact survivors and record
// where every surviving segment starts
template <bool HAS_SLOT>
__global__ __launch_bounds__(LS_THREADS) void k_compact(const uint32_t *__restrict__ pos,
                                                       const uint32_t *__restrict__ slot, uint64_t count,
                                                       const uint64_t *__restrict__ flags_in,
                                                       const uint64_t *__restrict__ ex, uint32_t *__restrict__ npos,
                                                       uint32_t *__restrict__ nslot, uint32_t *__restrict__ nseg,
                                                       uint32_t *__restrict__ nsegstart, uint32_t *__restrict__ out)
{
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    uint64_t f = flags_in[i];
    uint64_t e = ex[i];
    uint32_t sl = HAS_SLOT ? slot[i] : (uint32_t)i;
    uint32_t p = pos[i];
    if (f >> 32) {
        uint32_t ni = (uint32_t)(e >> 32);
        uint32_t sid = (uint32_t)(e & 0xFFFFFFFFull) + (uint32_t)(f & 1ull) - 1u;
        npos[ni] = p;
        nslot[ni] = sl;
        nseg[ni] = sid;
        if (f & 1ull) nsegstart[sid] = ni;
    } else {
        out[sl] = p;
    }
}

__global__ void k_last_total(const uint64_t *__restrict__ flags, const uint64_t *__restrict__ ex, uint64_t count,
                             uint64_t *__restrict__ total)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) total[0] = ex[count - 1] + flags[count - 1];
}

__global__ void k_set_u32(uint32_t *p, uint32_t v, uint32_t *zero_me)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        *p = v;
        if (zero_me) *zero_me = 0;
    }
}

int bits_for(uint64_t count)
{
    int b = 0;
    if (count > 1) {
        uint64_t v = count - 1;
        while (v) {
            b++;
            v >>= 1;
        }
    }
    return b;
}

int read_u64(kiss_hip_ctx *ctx, const void *dptr, uint64_t *out)
{
    KCHECK(hipMemcpyAsync(ctx->h_pinned, dptr, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));
    std::memcpy(out, ctx->h_pinned, sizeof(uint64_t));
    return KISS_HIP_OK;
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
    uint64_t *F1 = ctx->flags;
    uint64_t *F2 = ctx->flags + ctx->m_cap;
    uint32_t *d_nbig = ctx->d_small + 8;
    uint64_t *d_total = (uint64_t *)(ctx->d_small + 2);
    const unsigned T = LS_THREADS;

    // ------------------------------ round 0 ------------------------------------------------
    uint64_t count = m_far;
    KCHECK(hipMemcpyAsync(ctx->posA, ctx->lms_pos, count * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    RadixBufs rb;
    rb.key[0] = ctx->keyA;
    rb.key[1] = ctx->keyB;
    rb.pos[0] = ctx->posA;
    rb.pos[1] = ctx->posB;
    rb.seg[0] = rb.seg[1] = nullptr;
    // depth >= 125 whenever bounded, so the first 20 bases always count in full
    const int r0_shift = 64 - 2 * ROUND0_BASES;
    int res = 0;
    KTRY(kiss_radix_sort(ctx, rb, count, r0_shift, 0, &res));
    ctx->stats.lms_rounds++;
    ctx->stats.sort_item_rounds += count;
    uint32_t *Pc = rb.pos[res ^ 1]; // receives the survivors' positions
    uint32_t *Po = rb.pos[res];
    uint32_t *Sc = ctx->slotA, *Sn = ctx->slotB;
    uint32_t *Gc = ctx->segA, *Gn = ctx->segB;
    uint32_t *SSc = ctx->segstartA, *SSn = ctx->segstartB;
    {
        KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
        hipLaunchKernelGGL((k_flag<false>), dim3((unsigned)div_up(count, T)), dim3(T), 0, ctx->stream, rb.key[res],
                           (const uint32_t *)nullptr, count, r0_shift, 0, F1);
        KCHECK(hipGetLastError());
    }
    KTRY(kiss_scan_u64(ctx, F1, F2, count));
    {
        KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
        hipLaunchKernelGGL((k_compact<false>), dim3((unsigned)div_up(count, T)), dim3(T), 0, ctx->stream, Po,
                           (const uint32_t *)nullptr, count, F1, F2, Pc, Sc, Gc, SSc, ctx->lms_sorted_far);
        hipLaunchKernelGGL(k_last_total, dim3(1), dim3(64), 0, ctx->stream, F1, F2, count, d_total);
        KCHECK(hipGetLastError());
    }
    uint64_t tot;
    KTRY(read_u64(ctx, d_total, &tot));
    count = tot >> 32;
    uint64_t nseg = tot & 0xFFFFFFFFull;
    uint64_t *K1 = ctx->keyA, *K2 = ctx->keyB;
    const bool dbg = getenv("KISS_HIP_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[kiss_hip] round 0: items %llu -> survivors %llu in %llu segments\n", (unsigned long long)m_far, (unsigned long long)count, (unsigned long long)nseg);

    // ------------------------------ rounds >= 1 ---------------------------------------------
    uint64_t off = ROUND0_BASES;
    while (count > 0) {
        if (depth && off >= depth) return KISS_HIP_E_INTERNAL; // depth exhausted with ties left: flagged last round
        uint64_t rem = depth ? depth - off : 32;
        if (rem > 32) rem = 32;
        const bool last_round = depth ? (off + 32 >= depth) : false;
        const uint64_t mask = rem >= 32 ? ~0ull : (~0ull << (64 - 2 * rem));
        const int key_lo_bit = (int)(64 - 2 * rem);
        const unsigned grid = (unsigned)div_up(count, T);
        {
            KTimer t(ctx, KISS_HIP_K_KEYGATHER, count);
            hipLaunchKernelGGL(k_gather_keys, dim3(grid), dim3(T), 0, ctx->stream, ctx->pk, n, Pc, count, off, mask,
                               K1);
            KCHECK(hipGetLastError());
        }
        {
            KTimer t(ctx, KISS_HIP_K_SEGRANK, count);
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(64), 0, ctx->stream, SSc + nseg, (uint32_t)count, d_nbig);
            hipLaunchKernelGGL(k_seg_rank, dim3(grid), dim3(T), 0, ctx->stream, K1, Pc, Gc, SSc, count, K2, Po, F1,
                               d_nbig);
            KCHECK(hipGetLastError());
        }
        uint64_t nbig64 = 0;
        KTRY(read_u64(ctx, d_nbig, &nbig64));
        const uint64_t nbig = nbig64 & 0xFFFFFFFFull;
        if (nbig > 0) {
            KTRY(kiss_scan_u64(ctx, F1, F2, count));
            {
                KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
                hipLaunchKernelGGL(k_big_compact, dim3(grid), dim3(T), 0, ctx->stream, K1, Pc, count, F1, F2,
                                   ctx->bkeyA, ctx->bposA, ctx->bsegA, ctx->bslot);
                hipLaunchKernelGGL(k_last_total, dim3(1), dim3(64), 0, ctx->stream, F1, F2, count, d_total);
                KCHECK(hipGetLastError());
            }
            uint64_t bt;
            KTRY(read_u64(ctx, d_total, &bt));
            if ((bt >> 32) != nbig) return KISS_HIP_E_INTERNAL;
            const uint64_t nbigseg = bt & 0xFFFFFFFFull;
            RadixBufs bb;
            bb.key[0] = ctx->bkeyA;
            bb.key[1] = ctx->bkeyB;
            bb.pos[0] = ctx->bposA;
            bb.pos[1] = ctx->bposB;
            bb.seg[0] = ctx->bsegA;
            bb.seg[1] = ctx->bsegB;
            int bres = 0;
            KTRY(kiss_radix_sort(ctx, bb, nbig, key_lo_bit, bits_for(nbigseg), &bres));
            {
                KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, nbig);
                hipLaunchKernelGGL(k_big_writeback, dim3((unsigned)div_up(nbig, T)), dim3(T), 0, ctx->stream,
                                   bb.key[bres], bb.pos[bres], ctx->bslot, nbig, K2, Po);
                KCHECK(hipGetLastError());
            }
            ctx->stats.big_item_rounds += nbig;
        }
        ctx->stats.lms_rounds++;
        ctx->stats.sort_item_rounds += count;
        {
            KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
            hipLaunchKernelGGL((k_flag<true>), dim3(grid), dim3(T), 0, ctx->stream, K2, Gc, count, 0, (int)last_round,
                               F1);
            KCHECK(hipGetLastError());
        }
        KTRY(kiss_scan_u64(ctx, F1, F2, count));
        {
            KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
            hipLaunchKernelGGL((k_compact<true>), dim3(grid), dim3(T), 0, ctx->stream, Po, Sc, count, F1, F2, Pc, Sn,
                               Gn, SSn, ctx->lms_sorted_far);
            hipLaunchKernelGGL(k_last_total, dim3(1), dim3(64), 0, ctx->stream, F1, F2, count, d_total);
            KCHECK(hipGetLastError());
        }
        KTRY(read_u64(ctx, d_total, &tot));
        if (dbg) fprintf(stderr, "[kiss_hip] round off=%llu: items %llu (big %llu) -> survivors %llu in %llu segments\n", (unsigned long long)off, (unsigned long long)count, (unsigned long long)nbig, (unsigned long long)(tot >> 32), (unsigned long long)(tot & 0xFFFFFFFFull));
        std::swap(Sc, Sn);
        std::swap(Gc, Gn);
        std::swap(SSc, SSn);
        count = tot >> 32;
        nseg = tot & 0xFFFFFFFFull;
        off += 32;
        if (!depth && off > n + 64 && count > 0) return KISS_HIP_E_INTERNAL; // exact mode must have terminated
    }
    return KISS_HIP_OK;
}
