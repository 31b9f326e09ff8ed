// place.hip -- finishes the k-ordered LMS order and attaches context words.
//
// 1. Near-end rule.  LMS suffixes with fewer than D bases left are compared by the reference
//    comparator's scalar tail (include/biovoltron/algo/sort/kiss1_core.hpp:120-134): at most k bases, then
//    position; a suffix that runs off the text is smaller.  There are < D/2 of them.  Each is ranked
//    against the sorted far list by binary search with the full comparator (10-mer bucket with 'A'
//    padding, kiss1_core.hpp:41-83 + structs.hpp:175-184, then cmp), and among themselves pairwise
//    (a few hundred at k = 256) or, for the k in the hundred thousands and above that make them many, by a
//    merge sort with the same comparator.
//    Every far suffix precedes every near-end suffix in text position, so cmp(far, near) is monotone
//    over the far list and the insertion point is unique.
// 2. Merge far + near-end into the final list lmsP and attach the context word (the bases that precede each
//    suffix) the induction sweeps consume instead of random text reads: 11 bases from the payload of the
//    classification key for suffixes that round 0 of the sort made unique, a 15-base text gather for the rest.
#include "kiss_internal.hpp"

// ---- flight recorder of the near-end tie marks (hooks build only, KISS_HIP_TIE_TRACE=1) ----------------
// What k_near_tie_runs computed and saw, what k_near_tie_mark read, when each ran, and what a later kernel finds in the
// same places: compared on the host after the sort (kiss_tie_trace_report).
#ifdef KISS_HIP_HOOKS
#define KISS_TRACE(...) __VA_ARGS__
#define KISS_TRACE_ARG(ctx) , (ctx)->tie_dbg_on ? (ctx)->tie_dbg : (uint32_t *)nullptr
#ifndef KISS_TT_NO_RUNS // (-DKISS_TT_NO_RUNS: the hooks build with k_near_tie_runs compiled as shipped -- the kernel the fault follows)
#define KISS_TT_RUNS
#endif
#define KISS_TT_MARK
#define KISS_TT_TABLE
#else
#define KISS_TRACE(...)
#define KISS_TRACE_ARG(ctx) , (uint32_t *)nullptr
#endif
// (-DKISS_TT_RUNS / _MARK / _TABLE without the hooks: variant builds in which ONE of the three kernels is compiled as in
//  the hooks build, recorder code and parameter included but handed a null pointer -- the A-B of DESIGN.md 4.2 that tells
//  which kernel's generated code the fault lives in)
#ifdef KISS_TT_RUNS
#define KISS_TRACE_R(...) __VA_ARGS__
#else
#define KISS_TRACE_R(...)
#endif
#ifdef KISS_TT_MARK
#define KISS_TRACE_M(...) __VA_ARGS__
#else
#define KISS_TRACE_M(...)
#endif
#ifdef KISS_TT_TABLE
#define KISS_TRACE_T(...) __VA_ARGS__
#else
#define KISS_TRACE_T(...)
#endif

namespace {

constexpr int PL_THREADS = 256;
#if defined(KISS_TT_RUNS) || defined(KISS_TT_MARK) || defined(KISS_TT_TABLE)
// trace layout (32-bit words): header [0, 16), then per near-end suffix e < TT_MAX_E
constexpr uint32_t TT_MAX_E = 1024, TT_RUNS = 16, TT_MARK = TT_RUNS + 8 * TT_MAX_E, TT_POST = TT_MARK + 4 * TT_MAX_E,
                   TT_WORDS = TT_POST + 8 * TT_MAX_E;
#endif

// compare len bases at i and j (both ranges inside the text): <0, 0, >0
__device__ int cmp_bases(const uint64_t *__restrict__ pk, uint64_t i, uint64_t j, uint64_t len)
{
    while (len >= 32) {
        uint64_t a = kiss_key32(pk, i), b = kiss_key32(pk, j);
        if (a != b) return a < b ? -1 : 1;
        i += 32;
        j += 32;
        len -= 32;
    }
    if (len) {
        uint64_t mask = ~0ull << (64 - 2 * len);
        uint64_t a = kiss_key32(pk, i) & mask, b = kiss_key32(pk, j) & mask;
        if (a != b) return a < b ? -1 : 1;
    }
    return 0;
}

// the reference comparator (kiss1_core.hpp:94-135) on the packed text
__device__ bool lms_less(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k, uint64_t i, uint64_t j)
{
    uint64_t sl = 0;
    while (sl <= k && i + KISS_STRIDE <= n && j + KISS_STRIDE <= n) {
        int r = cmp_bases(pk, i, j, KISS_STRIDE);
        if (r) return r < 0;
        sl += KISS_STRIDE;
        i += KISS_STRIDE;
        j += KISS_STRIDE;
    }
    // scalar tail: at most k bases in total, stop at the end of either suffix
    uint64_t room = k > sl ? k - sl : 0;
    uint64_t li = n - i, lj = n - j;
    uint64_t len = room;
    if (li < len) len = li;
    if (lj < len) len = lj;
    int r = cmp_bases(pk, i, j, len);
    if (r) return r < 0;
    sl += len;
    i += len;
    j += len;
    if (sl >= k) return i < j;
    return i == n;
}

__device__ __forceinline__ uint32_t prefix10(const uint64_t *__restrict__ pk, uint64_t n, uint64_t p)
{
    if (p >= n) return 0;
    return (uint32_t)(kiss_key32(pk, p) >> 44); // 10 bases; zero padding past the end
}

__device__ bool lms_less_full(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k, uint64_t i, uint64_t j)
{
    uint32_t a = prefix10(pk, n, i), b = prefix10(pk, n, j);
    if (a != b) return a < b;
    return lms_less(pk, n, k, i, j);
}

// near_idx[e] = number of far suffixes that sort before near-end suffix e
__global__ __launch_bounds__(PL_THREADS) void k_near_rank(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k,
                                                         const uint32_t *__restrict__ far_sorted, uint64_t m_far,
                                                         const uint32_t *__restrict__ near_pos, uint32_t E,
                                                         uint32_t *__restrict__ near_idx)
{
    uint32_t e = blockIdx.x * PL_THREADS + threadIdx.x;
    if (e >= E) return;
    uint64_t pe = near_pos[e];
    uint64_t lo = 0, hi = m_far;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (lms_less_full(pk, n, k, far_sorted[mid], pe)) lo = mid + 1;
        else hi = mid;
    }
    near_idx[e] = (uint32_t)lo;
}

// near_fin[e] = final index of near-end suffix e in the merged list
__global__ __launch_bounds__(PL_THREADS) void k_near_order(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k,
                                                          const uint32_t *__restrict__ near_pos,
                                                          const uint32_t *__restrict__ near_idx, uint32_t E,
                                                          uint32_t *__restrict__ near_fin)
{
    uint32_t e = blockIdx.x * PL_THREADS + threadIdx.x;
    if (e >= E) return;
    uint64_t pe = near_pos[e];
    uint32_t ie = near_idx[e];
    uint32_t r = 0;
    for (uint32_t f = 0; f < E; f++) {
        if (f == e) continue;
        uint32_t jf = near_idx[f];
        if (jf < ie || (jf == ie && lms_less_full(pk, n, k, near_pos[f], pe))) r++;
    }
    near_fin[e] = ie + r;
}

// ---- large E (k in the hundred thousands and beyond, but < n): O(E log^2 E) instead of O(E^2) -------------------
// One round of a bottom-up merge sort of suffix positions by the reference comparator: `in` holds sorted runs of
// length `run`; every element finds its place in the merged run of length 2 * run by a binary search in the sibling
// run (left elements count the strictly smaller right ones, right elements the left ones that are not greater: the
// comparator is a strict total order -- ties go to the position --, so the places are distinct).
__global__ __launch_bounds__(PL_THREADS) void k_near_merge_round(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k,
                                                                const uint32_t *__restrict__ in, uint32_t E,
                                                                uint32_t run, uint32_t *__restrict__ out)
{
    const uint64_t e = (uint64_t)blockIdx.x * PL_THREADS + threadIdx.x;
    if (e >= E) return;
    const uint64_t base = e / (2ull * run) * (2ull * run);
    const uint64_t mid = base + run < E ? base + run : E, end = base + 2ull * run < E ? base + 2ull * run : E;
    const uint32_t x = in[e];
    if (e < mid) {
        uint64_t lo = mid, hi = end;
        while (lo < hi) {
            const uint64_t md = (lo + hi) >> 1;
            if (lms_less_full(pk, n, k, in[md], x)) lo = md + 1;
            else hi = md;
        }
        out[e + (lo - mid)] = x;
    } else {
        uint64_t lo = base, hi = mid;
        while (lo < hi) {
            const uint64_t md = (lo + hi) >> 1;
            if (!lms_less_full(pk, n, k, x, in[md])) lo = md + 1;
            else hi = md;
        }
        out[(e - mid) + lo] = x;
    }
}

// both lists are sorted by the same total order, so the insertion indexes ascend along the sorted near-end list and the
// final index of its j-th member is simply near_idx[j] + j
__global__ __launch_bounds__(PL_THREADS) void k_near_fin_sorted(const uint32_t *__restrict__ near_idx, uint32_t E,
                                                               uint32_t *__restrict__ near_fin)
{
    const uint32_t e = blockIdx.x * PL_THREADS + threadIdx.x;
    if (e < E) near_fin[e] = near_idx[e] + e;
}

// merged list + context words.  near_sidx: insertion indexes sorted ascending (E entries)
__device__ __forceinline__ uint32_t near_shift(const uint32_t *__restrict__ near_sidx, uint32_t E, uint64_t i)
{
    uint32_t lo = 0, hi = E; // #{e : near_idx[e] <= i}
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (near_sidx[mid] <= i) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// a thread moves four consecutive entries (16-byte loads; 16-byte stores when no near-end suffix lands among them)
__global__ __launch_bounds__(PL_THREADS) void k_merge_far(const uint64_t *__restrict__ pk,
                                                         const uint32_t *__restrict__ far_sorted,
                                                         const uint32_t *__restrict__ far_ctx, uint64_t m_far,
                                                         const uint32_t *__restrict__ near_sidx, uint32_t E,
                                                         uint32_t *__restrict__ lmsP, uint32_t *__restrict__ lmsC,
                                                         const uint8_t *__restrict__ hfar, // optional: tie flags of the far
                                                         uint8_t *__restrict__ hmerged)   // list, carried over to the merged one
{
    struct __attribute__((packed, aligned(4))) U4 {
        uint32_t v[4];
    };
    const uint64_t i0 = ((uint64_t)blockIdx.x * PL_THREADS + threadIdx.x) * 4;
    if (i0 >= m_far) return;
    if (i0 + 4 <= m_far) {
        const uint4 x = *reinterpret_cast<const uint4 *>(far_sorted + i0);
        const uint4 c = *reinterpret_cast<const uint4 *>(far_ctx + i0);
        uint32_t xs[4] = {x.x, x.y, x.z, x.w}, cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int e = 0; e < 4; e++)
            // no word from the sort (the suffix stayed tied past the first refinement round): gather it; the taint bit
            // (kiss_internal.hpp: KISS_CTX_TAINT), set where the suffix retired tied, stays
            if (KISS_CTX_WORD(cs[e]) == 0) cs[e] = kiss_load_ctx(pk, xs[e]) | (cs[e] & KISS_CTX_TAINT);
        const uint32_t lo0 = near_shift(near_sidx, E, i0), lo3 = E ? near_shift(near_sidx, E, i0 + 3) : 0u;
        if (lo0 == lo3) {
            U4 wp, wc;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                wp.v[e] = xs[e];
                wc.v[e] = cs[e];
            }
            *reinterpret_cast<U4 *>(lmsP + i0 + lo0) = wp;
            *reinterpret_cast<U4 *>(lmsC + i0 + lo0) = wc;
            if (hfar) {
                const uint32_t h4 = *reinterpret_cast<const uint32_t *>(hfar + i0);
                if ((lo0 & 3u) == 0) *reinterpret_cast<uint32_t *>(hmerged + i0 + lo0) = h4;
                else {
#pragma unroll
                    for (int e = 0; e < 4; e++) hmerged[i0 + lo0 + e] = (uint8_t)(h4 >> (8 * e));
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t lo = near_shift(near_sidx, E, i0 + e);
                lmsP[i0 + e + lo] = xs[e];
                lmsC[i0 + e + lo] = cs[e];
                if (hfar) hmerged[i0 + e + lo] = hfar[i0 + e];
            }
        }
    } else {
        for (uint64_t i = i0; i < m_far; i++) {
            const uint32_t lo = near_shift(near_sidx, E, i);
            const uint32_t x = far_sorted[i];
            uint32_t c = far_ctx[i];
            if (KISS_CTX_WORD(c) == 0) c = kiss_load_ctx(pk, x) | (c & KISS_CTX_TAINT);
            lmsP[i + lo] = x;
            lmsC[i + lo] = c;
            if (hfar) hmerged[i + lo] = hfar[i];
        }
    }
}

// tie flag of a near-end suffix in the merged list: it starts a group unless it and its predecessor there share h0 bases
// (a far suffix is never tied with a near-end predecessor: every far suffix that ties with a near-end one sorts before it).
// The predecessor is looked up in the arrays the merge is made FROM (the far list, the near-end suffixes' insertion indexes
// and final places), not in the merged list itself.  sorted_form: near_fin ascends with e (kiss_place_lms, form 2).
__global__ __launch_bounds__(PL_THREADS) void k_near_heads(const uint64_t *__restrict__ pk, uint64_t n, uint64_t h0,
                                                          const uint32_t *__restrict__ far_sorted,
                                                          const uint32_t *__restrict__ near_pos,
                                                          const uint32_t *__restrict__ near_idx,
                                                          const uint32_t *__restrict__ near_fin, uint32_t E, int sorted_form,
                                                          uint8_t *__restrict__ hmerged)
{
    const uint32_t e = blockIdx.x * PL_THREADS + threadIdx.x;
    if (e >= E) return;
    const uint32_t f = near_fin[e];
    uint8_t head = 1;
    if (f > 0) {
        const uint64_t p = near_pos[e];
        uint64_t q = ~0ull;
        if (sorted_form) {
            if (e > 0 && near_fin[e - 1] == f - 1) q = near_pos[e - 1];
        } else {
            for (uint32_t g = 0; g < E; g++) // (this form has a few hundred near-end suffixes, 4095 at most)
                if (near_fin[g] == f - 1) q = near_pos[g];
        }
        if (q == ~0ull && near_idx[e] > 0) q = far_sorted[near_idx[e] - 1];
        if (q != ~0ull && p + h0 <= n && q + h0 <= n && cmp_bases(pk, q, p, h0) == 0) head = 0;
    }
    hmerged[f] = head;
}

__global__ __launch_bounds__(PL_THREADS) void k_merge_near(const uint64_t *__restrict__ pk,
                                                          const uint32_t *__restrict__ near_pos,
                                                          const uint32_t *__restrict__ near_fin, uint32_t E,
                                                          uint32_t *__restrict__ lmsP, uint32_t *__restrict__ lmsC)
{
    uint32_t e = blockIdx.x * PL_THREADS + threadIdx.x;
    if (e >= E) return;
    uint32_t x = near_pos[e];
    lmsP[near_fin[e]] = x;
    lmsC[near_fin[e]] = kiss_load_ctx(pk, x) | KISS_CTX_TAINT; // ranked by the scalar tail of the comparator: tie rules apply
}

// ---- taint of the far suffixes that TIE with a near-end suffix ------------------------------------------------
// cmp(far f, near e) (kiss1_core.hpp:120-134) walks at most k bases; if f and e agree on all of them (e has between k
// and D bases left) the smaller position wins, which is f: every far suffix that ties with e sorts immediately before
// it.  Such an f may be unique among the far suffixes, so nothing else taints it -- but its place relative to e is a
// tie-rule place, not an exact-order one.  The common prefix with e can only grow along the sorted list towards e, so
// the run of far suffixes sharing >= k bases with e is found by a galloping + binary search from e's insertion index.
__device__ __forceinline__ bool shares_k(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k, uint64_t f, uint64_t e)
{
    if (f + k > n || e + k > n) return false; // fewer than k bases left: a comparison ends at the end of the text
    return cmp_bases(pk, f, e, k) == 0;
}

__global__ __launch_bounds__(PL_THREADS) void k_near_tie_runs(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k,
                                                             const uint32_t *__restrict__ far_sorted,
                                                             const uint32_t *__restrict__ near_pos,
                                                             const uint32_t *__restrict__ near_idx, uint32_t E,
                                                             uint32_t *__restrict__ run_start  KISS_TRACE_R(, uint32_t *dbg))
{
    const uint32_t e = blockIdx.x * PL_THREADS + threadIdx.x;
    KISS_TRACE_R(if (dbg && e == 0) { dbg[0] = E; dbg[1] = (uint32_t)k; dbg[8] = (uint32_t)wall_clock64(); })
    if (e >= E) return;
    const uint64_t pe = near_pos[e];
    const uint64_t hi = near_idx[e]; // far suffixes [0, hi) sort before e
    uint64_t lo = hi;                // first far index known to share k bases with e (hi: none yet)
    uint64_t step = 1;
    uint64_t bad = hi; // exclusive upper bound of indexes known NOT to share
    // gallop towards the front while the suffix at lo - step still shares k bases
    while (lo > 0) {
        const uint64_t j = lo >= step ? lo - step : 0;
        if (shares_k(pk, n, k, far_sorted[j], pe)) {
            lo = j;
            step *= 2;
            if (j == 0) break;
        } else {
            bad = j + 1; // indexes <= j do not share; the run starts in (j, lo]
            // binary search in [bad, lo)
            uint64_t a = bad, b = lo;
            while (a < b) {
                const uint64_t mid = (a + b) >> 1;
                if (shares_k(pk, n, k, far_sorted[mid], pe)) b = mid;
                else a = mid + 1;
            }
            lo = a;
            break;
        }
    }
    (void)bad;
    run_start[e] = (uint32_t)lo; // far suffixes [lo, hi) tie with e
    KISS_TRACE_R(if (dbg && e < TT_MAX_E) {
        uint32_t *r = dbg + TT_RUNS + 8 * e;
        const uint32_t f1 = hi >= 1 ? far_sorted[hi - 1] : 0xFFFFFFFFu, f2 = hi >= 2 ? far_sorted[hi - 2] : 0xFFFFFFFFu,
                       f3 = hi >= 3 ? far_sorted[hi - 3] : 0xFFFFFFFFu;
        r[0] = (uint32_t)lo;
        r[1] = (uint32_t)hi;
        r[2] = f1;
        r[3] = (hi >= 1 && shares_k(pk, n, k, f1, pe) ? 1u : 0u) | (hi >= 2 && shares_k(pk, n, k, f2, pe) ? 2u : 0u) |
               (hi >= 3 && shares_k(pk, n, k, f3, pe) ? 4u : 0u);
        r[4] = f2;
        r[5] = f3;
        r[6] = (uint32_t)pe;
        r[7] = (uint32_t)wall_clock64(); // (end of this thread; the host takes the latest)
    })
}

// The run of a near-end suffix e is [run_start[e], near_idx[e]): its first member shares k bases with e, the far suffix in
// front of it does not.  Two probes per suffix check what k_near_tie_runs found with ~30 -- in its own kernel, with its own
// code: round 4 traced the wrong exact-order results of two contexts sorting at once on one device to launches
// of k_near_tie_runs that left EMPTY runs for every lane of their wave although every input and every scalar of the wave
// reads correctly microseconds later; the tie marks of the exact order hang on these runs, so they are verified before they
// are used and the search is repeated when they do not hold.
__global__ __launch_bounds__(PL_THREADS) void k_near_tie_verify(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k,
                                                               const uint32_t *__restrict__ far_sorted,
                                                               const uint32_t *__restrict__ near_pos,
                                                               const uint32_t *__restrict__ near_idx, uint32_t E,
                                                               const uint32_t *__restrict__ run_start, uint32_t *__restrict__ bad)
{
    const uint32_t e = blockIdx.x * PL_THREADS + threadIdx.x;
    if (e >= E) return;
    const uint64_t pe = near_pos[e], hi = near_idx[e], lo = run_start[e];
    bool ok = lo <= hi;
    if (ok && lo < hi) ok = shares_k(pk, n, k, far_sorted[lo], pe);
    if (ok && lo > 0) ok = !shares_k(pk, n, k, far_sorted[lo - 1], pe);
    if (!ok) atomicAdd(bad, 1u);
}

// far_ctx[j] |= taint for j in [run_start[e], near_idx[e]): NT_BLOCKS workgroups per near suffix, grid-stride over its run
constexpr uint32_t NT_BLOCKS = 16;
// (the grid is capped: with E in the millions -- a bounded k of several million bases -- NT_BLOCKS * E workgroups of 256
//  threads would pass the 2^32 threads a HIP grid dimension holds; the workgroups stride over the (suffix, part) pairs)
constexpr uint32_t NT_MAX_GRID = 1u << 20;
__global__ __launch_bounds__(PL_THREADS) void k_near_tie_mark(const uint32_t *__restrict__ run_start,
                                                             const uint32_t *__restrict__ near_idx,
                                                             uint32_t *__restrict__ far_ctx, uint32_t E,
                                                             uint8_t *__restrict__ hfar // optional (ctx->hfar): the far suffixes
                                                             KISS_TRACE_M(, uint32_t *dbg)) // of a run share k bases with each other too -- one group
{
    KISS_TRACE_M(if (dbg && threadIdx.x == 0 && blockIdx.x == 0) {
        dbg[4] = E;
        dbg[5] = hfar ? 1u : 0u;
        dbg[6] = (uint32_t)wall_clock64(); // start of the first workgroup
    })
    for (uint64_t b = blockIdx.x; b < (uint64_t)NT_BLOCKS * E; b += gridDim.x) {
        const uint32_t e = (uint32_t)(b / NT_BLOCKS), sub = (uint32_t)(b % NT_BLOCKS);
        const uint64_t lo = run_start[e], hi = near_idx[e];
        KISS_TRACE_M(if (dbg && threadIdx.x == 0 && e < TT_MAX_E && sub == 0) {
            uint32_t *q = dbg + TT_MARK + 4 * e;
            q[0] = (uint32_t)lo;
            q[1] = (uint32_t)hi;
            q[2] = (uint32_t)wall_clock64();
        })
        for (uint64_t j = lo + (uint64_t)sub * PL_THREADS + threadIdx.x; j < hi; j += (uint64_t)NT_BLOCKS * PL_THREADS) {
            far_ctx[j] |= KISS_CTX_TAINT; // (a word of 0 = "gather me" becomes 0x80000000: still gathered, see k_merge_far)
            if (hfar && j > lo) hfar[j] = 0;
        }
    }
}

} // namespace

#include <algorithm>
#include <string>
#include <vector>

#include <cstdlib>

// near_idx / near_fin / near_pos / near_tmp hold one entry per near-end suffix: 65 536 to start with, regrown for the
// inputs that have more (a bounded k in the hundred thousands or above; the reference takes any k, kiss1_core.hpp:94-135)
static int near_reserve(kiss_hip_ctx *ctx, uint64_t E)
{
    if (E <= ctx->near_cap && ctx->near_tmp) return KISS_HIP_OK;
    uint64_t cap = ctx->near_cap ? ctx->near_cap : 65536;
    while (cap < E) cap *= 2;
    uint32_t **ptrs[] = {&ctx->near_idx, &ctx->near_fin, &ctx->near_pos, &ctx->near_tmp, &ctx->near_tmp2};
    for (uint32_t **p : ptrs) {
        if (*p) {
            (void)hipFree(*p);
            ctx->ws_bytes -= ctx->near_cap * sizeof(uint32_t);
            *p = nullptr;
        }
    }
    ctx->near_cap = 0;
    for (uint32_t **p : ptrs) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, cap * sizeof(uint32_t));
        if (e != hipSuccess) {
            ctx->last_hip_error = (int)e;
            return KISS_HIP_E_NOMEM;
        }
        *p = reinterpret_cast<uint32_t *>(q);
    }
    ctx->near_cap = cap;
    ctx->ws_bytes += 5 * cap * sizeof(uint32_t);
    return KISS_HIP_OK;
}

namespace {
// near-end suffixes by ascending merged index (E is small here: all pairs)
__global__ __launch_bounds__(PL_THREADS) void k_near_table(const uint32_t *__restrict__ near_pos,
                                                          const uint32_t *__restrict__ near_fin, uint32_t E,
                                                          uint32_t *__restrict__ tab_fin, uint32_t *tab_pos
                                                          KISS_TRACE_T(, uint32_t *dbg, const uint32_t *tt_far, const uint32_t *tt_ctx,
                                                                     const uint8_t *tt_hfar, const uint32_t *tt_idx))
{
    const uint32_t e = blockIdx.x * PL_THREADS + threadIdx.x;
    KISS_TRACE_T(
        if (dbg && e == 0) dbg[3] = (uint32_t)wall_clock64();
        if (dbg && e < E && e < TT_MAX_E) { // the same places again, as the kernel after the marks finds them (tab_pos[e] = run_start[e] still)
            const uint64_t hi = tt_idx[e];
            uint32_t *q = dbg + TT_POST + 8 * e;
            q[0] = (uint32_t)hi;
            q[1] = hi >= 1 ? tt_far[hi - 1] : 0xFFFFFFFFu;
            q[2] = hi >= 2 ? tt_far[hi - 2] : 0xFFFFFFFFu;
            q[3] = hi >= 3 ? tt_far[hi - 3] : 0xFFFFFFFFu;
            q[4] = tt_hfar ? ((hi >= 1 ? tt_hfar[hi - 1] : 9u) | (hi >= 2 ? tt_hfar[hi - 2] : 9u) << 8 | (hi >= 3 ? tt_hfar[hi - 3] : 9u) << 16) : 0xFFFFFFFFu;
            q[5] = (hi >= 1 ? tt_ctx[hi - 1] >> 31 : 9u) | (hi >= 2 ? tt_ctx[hi - 2] >> 31 : 9u) << 8 | (hi >= 3 ? tt_ctx[hi - 3] >> 31 : 9u) << 16;
            q[6] = tab_pos[e];
            q[7] = near_pos[e];
        }
        if (dbg) __syncthreads(); // (one workgroup, E <= 256: every thread has read its tab_pos[e] before any thread overwrites the array)
    )
    if (e >= E) return;
    const uint32_t fe = near_fin[e];
    uint32_t r = 0;
    for (uint32_t f = 0; f < E; f++) r += near_fin[f] < fe ? 1u : 0u;
    tab_fin[r] = fe;
    tab_pos[r] = near_pos[e];
}


} // namespace

// k_near_tie_runs, verified (k_near_tie_verify) and repeated if its runs do not hold: at most three searches
static int near_tie_runs_verified(kiss_hip_ctx *ctx, uint64_t n, uint32_t k, const uint32_t *near_pos, uint32_t E, unsigned egrid)
{
    uint32_t *bad = ctx->d_small + 62;
    for (int attempt = 0;; attempt++) {
        hipLaunchKernelGGL(k_near_tie_runs, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, n, (uint64_t)k, ctx->lms_sorted_far,
                           near_pos, ctx->near_idx, E, ctx->near_tmp2 KISS_TRACE_R(KISS_TRACE_ARG(ctx)));
        KTRY(kiss_zero_u32(ctx, bad, 1));
        hipLaunchKernelGGL(k_near_tie_verify, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, n, (uint64_t)k, ctx->lms_sorted_far,
                           near_pos, ctx->near_idx, E, ctx->near_tmp2, bad);
        KCHECK(hipGetLastError());
        KTRY(kiss_readback(ctx, bad, 1));
        if (ctx->h_pinned[0] == 0) return KISS_HIP_OK;
        ctx->stats.tie_run_retries++;
        if (ctx->opts.debug)
            fprintf(stderr, "[kiss_hip] near-end tie runs: %u of %u runs do not hold (search %d), searching again\n", ctx->h_pinned[0], E, attempt + 1);
        if (attempt == 2) return KINTERNAL();
    }
}

int kiss_place_lms(kiss_hip_ctx *ctx, uint64_t n, uint32_t k, uint64_t depth)
{
    (void)depth;
    const uint64_t m = ctx->m, m_far = ctx->m_far;
    const uint64_t E64 = m - m_far;
    ctx->stats.near_end = E64;
    ctx->lms_merged = false;
    ctx->near_form = 0;
    ctx->near_sorted = nullptr;
    ctx->rm_fin = ctx->rm_pos = nullptr;
    ctx->rm_E = 0;
    if (m == 0) return KISS_HIP_OK;
    KTRY(near_reserve(ctx, E64 ? E64 : 1));
    const uint32_t E = (uint32_t)E64;
    // pairwise ranking is the cheapest form for the handful of near-end suffixes of k = 32 / 256 (E ~ 0.3 D); beyond
    // this size the merge-sort form takes over (the hooks build can move the switch).
    const uint32_t merge_min = ctx->opts.near_merge_min;
    {
        KTimer t(ctx, KISS_HIP_K_PLACE, m);
        const unsigned egrid = (unsigned)div_up(E ? E : 1, PL_THREADS);
        const uint64_t tie_blocks = (uint64_t)NT_BLOCKS * E;
        const unsigned tie_grid = (unsigned)(tie_blocks < NT_MAX_GRID ? (tie_blocks ? tie_blocks : 1) : NT_MAX_GRID);
        if (E > 0 && E >= merge_min) {
            // sort the near-end suffixes among themselves, then rank them against the far list
            const uint32_t *cur = ctx->lms_pos + m_far; // ascending text positions = runs of length 1
            uint32_t *bufs[2] = {ctx->near_pos, ctx->near_tmp};
            int w = 0;
            for (uint64_t run = 1; run < E; run *= 2) {
                hipLaunchKernelGGL(k_near_merge_round, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, n, (uint64_t)k,
                                   cur, E, (uint32_t)run, bufs[w]);
                cur = bufs[w];
                w ^= 1;
            }
            const uint32_t *near_sorted = cur; // E == 1: the input itself
            hipLaunchKernelGGL(k_near_rank, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, n, (uint64_t)k,
                               ctx->lms_sorted_far, m_far, near_sorted, E, ctx->near_idx);
            hipLaunchKernelGGL(k_near_fin_sorted, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->near_idx, E, ctx->near_fin);
            if (m_far && (uint64_t)k < n) {
                KTRY(near_tie_runs_verified(ctx, n, k, near_sorted, E, egrid));
                hipLaunchKernelGGL(k_near_tie_mark, dim3(tie_grid), dim3(PL_THREADS), 0, ctx->stream, ctx->near_tmp2, ctx->near_idx,
                                   ctx->lms_ctx_far, E, (k == ctx->h_depth) ? ctx->hfar : (uint8_t *)nullptr KISS_TRACE_M(, (uint32_t *)nullptr));
            }
            ctx->near_form = 2;
            ctx->near_sorted = near_sorted;
            ctx->rm_fin = ctx->near_fin; // both lists are k-sorted: the merged indexes ascend already
            ctx->rm_pos = near_sorted;
        } else if (E > 0) {
            const uint32_t *near_pos = ctx->lms_pos + m_far; // ascending list: the near-end suffixes are its tail
#ifdef KISS_HIP_HOOKS
            ctx->tie_dbg_on = false;
            if (ctx->opts.tie_trace && m_far && (uint64_t)k < n) {
                if (!ctx->tie_dbg && hipMalloc((void **)&ctx->tie_dbg, TT_WORDS * sizeof(uint32_t)) != hipSuccess) ctx->tie_dbg = nullptr;
                if (ctx->tie_dbg) { // (every word read by the report is written by this sort's kernels: nothing to clear)
                    ctx->tie_dbg_on = true;
                    ctx->tie_dbg_E = E;
                    ctx->tie_dbg_k = k;
                }
            }
#endif
            hipLaunchKernelGGL(k_near_rank, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, n, (uint64_t)k,
                               ctx->lms_sorted_far, m_far, near_pos, E, ctx->near_idx);
            hipLaunchKernelGGL(k_near_order, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, n, (uint64_t)k, near_pos,
                               ctx->near_idx, E, ctx->near_fin);
            if (m_far && (uint64_t)k < n) {
                KTRY(near_tie_runs_verified(ctx, n, k, near_pos, E, egrid));
                hipLaunchKernelGGL(k_near_tie_mark, dim3(tie_grid), dim3(PL_THREADS), 0, ctx->stream, ctx->near_tmp2, ctx->near_idx,
                                   ctx->lms_ctx_far, E, (k == ctx->h_depth) ? ctx->hfar : (uint8_t *)nullptr KISS_TRACE_M(KISS_TRACE_ARG(ctx)));
            }
            // (stream order: k_near_tie_mark has read near_tmp2 before the table overwrites it)
            hipLaunchKernelGGL(k_near_table, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, near_pos, ctx->near_fin, E, ctx->near_tmp,
                               ctx->near_tmp2 KISS_TRACE_T(KISS_TRACE_ARG(ctx), ctx->lms_sorted_far, ctx->lms_ctx_far,
                                              (k == ctx->h_depth) ? ctx->hfar : (const uint8_t *)nullptr, ctx->near_idx));
            ctx->near_form = 1;
            ctx->rm_fin = ctx->near_tmp;
            ctx->rm_pos = ctx->near_tmp2;
        }
        ctx->rm_E = E;
        KCHECK(hipGetLastError());
    }
    const bool merge_now = ctx->opts.merge_lms; // (hooks build: the merged copy of round 1)
    if (merge_now) KTRY(kiss_merge_lms(ctx));
    return KISS_HIP_OK;
}

// The merged list as an array of its own (what `put_lms_suffix` writes into SA in the reference): only the stage
// outputs of the tests and the A-B hook ask for it.
int kiss_merge_lms(kiss_hip_ctx *ctx)
{
    if (ctx->lms_merged || ctx->m == 0) return KISS_HIP_OK;
    const uint64_t m_far = ctx->m_far;
    const uint32_t E = ctx->rm_E;
    KTimer t(ctx, KISS_HIP_K_PLACE, ctx->m);
    const unsigned egrid = (unsigned)div_up(E ? E : 1, PL_THREADS);
    const unsigned fgrid = (unsigned)div_up(div_up(m_far, 4), PL_THREADS);
    // exact order through the LMS-level doubling: the far list's tie flags go along (both pointers set, or neither)
    const uint8_t *hf = ctx->hmerged ? ctx->hfar : nullptr;
    uint8_t *hm = hf ? ctx->hmerged : nullptr;
    if (E == 0) {
        hipLaunchKernelGGL(k_merge_far, dim3(fgrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, ctx->lms_sorted_far, ctx->lms_ctx_far,
                           m_far, (const uint32_t *)nullptr, 0u, ctx->lmsP, ctx->lmsC, hf, hm);
    } else if (ctx->near_form == 2) {
        if (m_far)
            hipLaunchKernelGGL(k_merge_far, dim3(fgrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, ctx->lms_sorted_far,
                               ctx->lms_ctx_far, m_far, ctx->near_idx, E, ctx->lmsP, ctx->lmsC, hf, hm);
        hipLaunchKernelGGL(k_merge_near, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, ctx->near_sorted, ctx->near_fin, E,
                           ctx->lmsP, ctx->lmsC);
    } else {
        const uint32_t *near_pos = ctx->lms_pos + m_far;
        // sorted insertion indexes for the merge (E is tiny: sort on the host)
        std::vector<uint32_t> idx(E);
        KCHECK(hipMemcpyAsync(idx.data(), ctx->near_idx, E * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
        std::sort(idx.begin(), idx.end());
        KCHECK(hipMemcpyAsync(ctx->near_pos, idx.data(), E * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream)); // idx goes out of scope
        if (m_far)
            hipLaunchKernelGGL(k_merge_far, dim3(fgrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, ctx->lms_sorted_far,
                               ctx->lms_ctx_far, m_far, ctx->near_pos, E, ctx->lmsP, ctx->lmsC, hf, hm);
        hipLaunchKernelGGL(k_merge_near, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, near_pos, ctx->near_fin, E, ctx->lmsP,
                           ctx->lmsC);
    }
    if (hm && E)
        hipLaunchKernelGGL(k_near_heads, dim3(egrid), dim3(PL_THREADS), 0, ctx->stream, ctx->pk, ctx->n, (uint64_t)ctx->h_depth,
                           ctx->lms_sorted_far, ctx->near_form == 2 ? ctx->near_sorted : ctx->lms_pos + m_far, ctx->near_idx,
                           ctx->near_fin, E, ctx->near_form == 2 ? 1 : 0, hm);
    KCHECK(hipGetLastError());
    ctx->lms_merged = true;
    return KISS_HIP_OK;
}

#ifdef KISS_HIP_HOOKS
// after the sort has finished (stream synchronised): one line per traced sort with KISS_HIP_TIE_TRACE=2, and a dump of
// everything recorded when what the kernels saw does not fit together
void kiss_tie_trace_report(kiss_hip_ctx *ctx)
{
    if (!ctx->tie_dbg_on || !ctx->tie_dbg) return;
    ctx->tie_dbg_on = false;
    std::vector<uint32_t> h(TT_WORDS);
    // (on the context's own stream: a plain hipMemcpy would wait for every other stream of the device, the other thread's too)
    if (hipMemcpyAsync(h.data(), ctx->tie_dbg, TT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return;
    const uint32_t E = ctx->tie_dbg_E < TT_MAX_E ? ctx->tie_dbg_E : TT_MAX_E;
    const uint64_t n = ctx->n, k = ctx->tie_dbg_k;
#ifndef KISS_TT_RUNS
    // k_near_tie_runs compiled as shipped (no recorder in it): one line per sort with what the mark kernel read and what the
    // table kernel found afterwards, for the near-end suffixes that can tie; the stress harness compares the lines of one text
    {
        std::string line;
        char buf[1024];
        unsigned can = 0, stale = 0, nomark = 0;
        for (uint32_t e = 0; e < E; e++) {
            const uint32_t *q = &h[TT_MARK + 4 * e], *p = &h[TT_POST + 8 * e];
            if (n - p[7] < k) continue;
            can++;
            if (q[0] != p[6]) stale++;                         // the mark kernel read another run start than the table kernel
            if (p[4] != 0xFFFFFFFFu && (p[4] & 0xFF) != 0) nomark++; // far[hi - 1] still a group head
            if (can <= 2) {
                snprintf(buf, sizeof buf, " [e %u pos %u: mark read lo %u hi %u; afterwards run start %u hi %u far %u %u %u hfar %06x taint %06x]",
                         e, p[7], q[0], q[1], p[6], p[0], p[1], p[2], p[3], p[4], p[5]);
                line += buf;
            }
        }
#ifdef KISS_TT_RUNS_ASM_RECORD
        // the variant whose k_near_tie_runs was patched at the assembly level (tools/repro/patch_runs_asm.py): behind its last
        // store every lane leaves 16 words of its wave state at run_start[1024 + 16 e ..] -- the shipped instruction
        // sequence in front of it is untouched
        {
            std::vector<uint32_t> w(32 * (size_t)E);
            if (hipMemcpyAsync(w.data(), ctx->near_tmp2 + 1024, w.size() * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                hipStreamSynchronize(ctx->stream) == hipSuccess) {
                unsigned shown = 0;
                for (uint32_t e = 0; e < E && shown < 2; e++) {
                    const uint32_t *p = &h[TT_POST + 8 * e], *r = &w[32 * (size_t)e];
                    if (n - p[7] < k) continue;
                    shown++;
                    snprintf(buf, sizeof buf, " {asm e %u: mask %08x%08x pk %08x%08x n %08x%08x k %08x%08x far %08x%08x pe %u lo %u v16 %u exec %08x%08x tag %08x | "
                                              "far[lo-1] again: plain %u coherent %u | its pk word plain %08x%08x coherent %08x%08x | pk word of pe plain %08x%08x coherent %08x%08x}",
                             e, r[1], r[0], r[3], r[2], r[5], r[4], r[7], r[6], r[9], r[8], r[10], r[11], r[12], r[14], r[13], r[15], r[16], r[17],
                             r[19], r[18], r[21], r[20], r[23], r[22], r[25], r[24]);
                    line += buf;
                }
            }
        }
#endif
        fprintf(stderr, "[kiss_hip] tie_trace2 ctx %p n %llu k %llu E %u (mark kernel saw E %u hfar %u): can tie %u, mark/table disagree %u, "
                        "far[hi-1] unmarked %u; mark start tick %u table start tick %u%s\n",
                (void *)ctx, (unsigned long long)n, (unsigned long long)k, ctx->tie_dbg_E, h[4], h[5], can, stale, nomark, h[6], h[3], line.c_str());
        return;
    }
#endif
    unsigned bad_read = 0, bad_list = 0, bad_mark = 0, can_tie = 0, empty_runs = 0;
    uint32_t runs_end = 0;
    bool have_end = false;
    int first = -1;
    for (uint32_t e = 0; e < E; e++) {
        const uint32_t *r = &h[TT_RUNS + 8 * e], *q = &h[TT_MARK + 4 * e], *p = &h[TT_POST + 8 * e];
        if (n - r[6] < k) continue; // fewer than k bases left: ties with nothing
        if (first < 0) first = (int)e;
        can_tie++;
        if (r[1] - r[0] < 2) empty_runs++;
        if (q[0] != r[0] || q[1] != r[1] || p[6] != r[0]) bad_read++;
        if ((int32_t)(r[7] - runs_end) > 0 || !have_end) { runs_end = r[7]; have_end = true; }
        if (p[1] != r[2] || p[2] != r[4] || p[3] != r[5] || p[0] != r[1]) bad_list++;
        if (r[1] - r[0] >= 2 && ((p[5] & 0xFF) != 1 || (p[4] != 0xFFFFFFFFu && (p[4] & 0xFF) != 0))) bad_mark++;
    }
    const bool order_bad = (can_tie && ((int32_t)(h[6] - runs_end) < 0 || (int32_t)(h[3] - h[6]) < 0));
    const bool hdr_bad = h[0] != ctx->tie_dbg_E || h[4] != ctx->tie_dbg_E || h[1] != (uint32_t)k;
    const bool anomaly = bad_read || bad_list || bad_mark || order_bad || hdr_bad;
    if (ctx->opts.tie_trace >= 2 || anomaly) {
        fprintf(stderr, "[kiss_hip] tie_trace%s ctx %p n %llu k %llu E %u (kernels saw E %u / %u, k %u, hfar %u): %u near-end suffixes can tie, "
                        "%u with a run < 2; mark kernel read something else for %u, far list changed for %u, marks missing for %u; "
                        "ticks since the runs kernel started: runs end %u, mark start %u, table start %u\n",
                anomaly ? " ANOMALY" : "", (void *)ctx, (unsigned long long)n, (unsigned long long)k, ctx->tie_dbg_E, h[0], h[4], h[1], h[5],
                can_tie, empty_runs, bad_read, bad_list, bad_mark, runs_end - h[8], h[6] - h[8], h[3] - h[8]);
        if (first >= 0) {
            int shown = 0;
            for (uint32_t e = (uint32_t)first; e < E && shown < (anomaly ? 6 : 1); e++) {
                const uint32_t *r = &h[TT_RUNS + 8 * e], *q = &h[TT_MARK + 4 * e], *p = &h[TT_POST + 8 * e];
                if (n - r[6] < k) continue;
                shown++;
                fprintf(stderr, "[kiss_hip]   e %u pos %u: runs kernel lo %u hi %u far[hi-1..hi-3] %u %u %u shares %u | mark kernel read lo %u hi %u "
                                "| afterwards hi %u far %u %u %u hfar %06x taint %06x run start %u\n",
                        e, r[6], r[0], r[1], r[2], r[4], r[5], r[3], q[0], q[1], p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
            }
        }
    }
}
#endif
