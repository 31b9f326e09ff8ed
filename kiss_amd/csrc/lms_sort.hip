// lms_sort.hip -- k-ordered sort of the LMS suffixes (the "far" ones: p + D <= n).
//
// Contract (reference comparator, include/biovoltron/algo/sort/kiss1_core.hpp:94-135, applied after the
// stable 10-mer bucketing of :41-83): for two LMS suffixes that both have D = 125*(k/125+1) bases left
// the order is (first D bases, then text position).  k >= n means the exact suffix order (D unbounded;
// past-the-end bases read as 'A' exactly like the reference's zero padding, structs.hpp:94-96).
//
// GPU formulation: MSD refinement.
//   round 0 : stable LSD radix sort of all suffixes on their first 20 bases (5 passes of 8 bits);
//             suffixes that share those 20 bases form a segment, singletons retire to their final slot.
//   round r : every still-tied suffix fetches its next 32 bases (one u64 key).
//             - a segment of <= SMALL_SEG suffixes is FINISHED on the spot: one lane per suffix ranks it
//               against the others of its segment, first on the fetched key and, for pairs that tie on it,
//               by walking both suffixes to the full depth D (then position).  All of them retire.
//             - larger segments are compacted and sorted on (segment id, key): from the second refinement round on
//               by a stable three-way split around the key of the segment's middle item (what is left then is
//               mostly long tandem arrays, whose members carry one key per round) followed by a radix sort of the
//               < and > groups only; in the first round by the radix sort directly.  Then they are split where
//               neighbours differ and go round again.
// Every step is stable and the initial order is ascending text position, which yields the reference's
// position tie-break (kiss1_core.hpp:131-133) once the depth D is exhausted.
#include "kiss_internal.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <utility>
#include <vector>

namespace {

constexpr int LS_THREADS = 256;
constexpr uint32_t SMALL_SEG = 64;      // the doubling rounds of the exact-order finish
// the refinement rounds of the LMS sort: measured at chm13 size 8 / 16 / 24 / 64 / 256 -> 83.7 / 83.5 / 83.1 / 85.3 /
// 90.1 ms per sort (KISS_HIP_SMALL_SEG): the all-pairs finish is O(s^2) walks per segment, the pivot rounds are not
constexpr uint32_t LMS_SMALL_SEG = 24;
constexpr int ROUND0_BASES = 20;

// seg / segstart (optional): only members of segments of min_len .. max_len items get a key.  A segment of exactly two is
// compared once by k_seg_finish, walking both suffixes from the segment's depth on, and a segment that goes to a pivot
// round is keyed there against its reference string (k_pivot_lcp): for both the 32-base key would be one more random
// text read per member for nothing (pairs are more than half of what survives round 0 -- the two copies of a duplicated
// stretch --, the big segments most of the rest)
__global__ __launch_bounds__(LS_THREADS) void k_gather_keys(const uint64_t *__restrict__ pk, uint64_t n,
                                                           const uint32_t *__restrict__ pos, uint64_t count,
                                                           uint64_t depth_off, uint64_t mask,
                                                           uint64_t *__restrict__ key, const uint32_t *__restrict__ seg,
                                                           const uint32_t *__restrict__ segstart, uint32_t min_len,
                                                           uint32_t max_len)
{
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    if (seg) {
        const uint32_t sg = seg[i];
        const uint32_t len = segstart[sg + 1] - segstart[sg];
        if (len < min_len || len > max_len) return;
    }
    uint64_t q = (uint64_t)pos[i] + depth_off;
    uint64_t k = (q < n) ? kiss_key32(pk, q) : 0ull;
    key[i] = k & mask;
}

// true if suffix pi sorts before suffix pj, both already equal on [0, off + 32); bases from off + 32 on are
// compared up to depth (0 = unbounded), ties go to the smaller index (= smaller text position)
// *tied (optional) is set when the walk reached the depth without a difference: the order then is the tie rule's, and
// both suffixes are tainted for the exact-order finish (KISS_CTX_TAINT)
// (off + 32 = the first base not yet compared: callers that have compared the 32-base key of the round pass `off`,
//  the pair path of k_seg_finish, which has no key, passes off - 32 and lets the walk start at the segment's depth)
__device__ __forceinline__ bool deep_less(const uint64_t *__restrict__ pk, uint64_t n, uint64_t pi, uint64_t pj,
                                          uint64_t off, uint64_t depth, bool i_before_j, bool *tied = nullptr)
{
    uint64_t d = off + 32;
    for (;;) {
        if ((depth && d >= depth) || (pi + d >= n && pj + d >= n)) { // (both off the text: not for two LMS suffixes)
            if (tied) *tied = true;
            return i_before_j;
        }
        uint64_t qi = pi + d, qj = pj + d;
        if (qi + 96 < n && qj + 96 < n && (!depth || depth - d >= 128)) {
            // 128 bases per step: the ten word loads are independent, so one round trip to memory covers four
            // 32-base compares (the walk through a long repeat is a chain of dependent loads otherwise)
            const uint32_t si = 2u * (uint32_t)(qi & 31u), sj = 2u * (uint32_t)(qj & 31u);
            uint64_t a[5], b[5];
            kiss_words5(pk, qi >> 5, a); // (aligned loads: kiss_internal.hpp)
            kiss_words5(pk, qj >> 5, b);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint64_t ki = (a[t] << si) | ((a[t + 1] >> 1) >> (63u - si));
                const uint64_t kj = (b[t] << sj) | ((b[t + 1] >> 1) >> (63u - sj));
                if (ki != kj) return ki < kj;
            }
            d += 128;
            continue;
        }
        if (depth >= 128 && pi + depth <= n && pj + depth <= n) {
            // fewer than 128 bases to go: ONE more step over the last 128 bases of the depth instead of up to four
            // dependent 32-base steps.  It overlaps bases already found equal, which changes nothing: the first
            // differing word still holds the first differing base.
            const uint64_t ri = pi + depth - 128, rj = pj + depth - 128;
            const uint32_t si = 2u * (uint32_t)(ri & 31u), sj = 2u * (uint32_t)(rj & 31u);
            uint64_t a[5], b[5];
            kiss_words5(pk, ri >> 5, a);
            kiss_words5(pk, rj >> 5, b);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint64_t ki = (a[t] << si) | ((a[t + 1] >> 1) >> (63u - si));
                const uint64_t kj = (b[t] << sj) | ((b[t + 1] >> 1) >> (63u - sj));
                if (ki != kj) return ki < kj;
            }
            if (tied) *tied = true;
            return i_before_j;
        }
        uint64_t ki = qi < n ? kiss_key32(pk, qi) : 0ull;
        uint64_t kj = qj < n ? kiss_key32(pk, qj) : 0ull;
        if (depth && depth - d < 32) {
            uint64_t mask = ~0ull << (64 - 2 * (depth - d));
            ki &= mask;
            kj &= mask;
        }
        if (ki != kj) return ki < kj;
        d += 32;
    }
}

// phase A for small segments of >= 3 members: is my successor in the segment not smaller than me?
// (a segment whose adjacent pairs are all in order is already sorted: the common case for periodic repeats,
//  where all-pairs ranking would walk s^2 pairs to the full depth)
__global__ __launch_bounds__(LS_THREADS) void k_seg_adjacent(const uint64_t *__restrict__ pk, uint64_t n,
                                                            const uint64_t *__restrict__ key,
                                                            const uint32_t *__restrict__ pos,
                                                            const uint32_t *__restrict__ seg,
                                                            const uint32_t *__restrict__ segstart, uint64_t count,
                                                            uint64_t off, uint64_t depth, uint32_t small_seg,
                                                            uint8_t *__restrict__ inorder)
{
    uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    const uint32_t sg = seg[i];
    const uint32_t a = segstart[sg], b = segstart[sg + 1];
    const uint32_t len = b - a;
    if (len < 3 || len > small_seg) return;
    uint8_t ok = 1; // 0: my successor is smaller, 1: in order, 2: in order by the tie rule (equal through the depth)
    if ((uint32_t)i + 1 < b) {
        const uint64_t ki = key[i], kn = key[i + 1];
        if (kn != ki) ok = kn > ki;
        else {
            bool tied = false;
            ok = !deep_less(pk, n, pos[i + 1], pos[i], off, depth, false, &tied);
            if (tied) ok = 2;
        }
    }
    inorder[i] = ok;
}

// small segments: final rank of every member, written straight to its final slot;
// big segments: flagged for the radix path.  big[i] = 0: finished here, 1: member of a big segment, 3: its first member
// (one byte per item since round 4: k_bigb_count / k_bigb_compact below count and compact them tile by tile; the
//  8-byte flag words of rounds 1-3 went through a full scan, 5 GB of traffic for 35 M big-segment items)
__global__ __launch_bounds__(LS_THREADS) void k_seg_finish(const uint64_t *__restrict__ pk, uint64_t n,
                                                          const uint64_t *__restrict__ key,
                                                          const uint32_t *__restrict__ pos,
                                                          const uint32_t *__restrict__ slot,
                                                          const uint32_t *__restrict__ seg,
                                                          const uint32_t *__restrict__ segstart, uint64_t count,
                                                          uint64_t off, uint64_t depth, uint32_t small_seg,
                                                          const uint8_t *__restrict__ inorder,
                                                          uint32_t *__restrict__ out, uint8_t *__restrict__ big,
                                                          uint32_t *__restrict__ nbig,
                                                          const uint32_t *__restrict__ tctx, // first round only: the
                                                          uint32_t *__restrict__ octx,       // items' context words
                                                          uint8_t *__restrict__ hfar) // optional: 0 for an item that retires
                                                          // tied with the one before it in the final order (ctx->hfar)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    const bool valid = i < count;
    uint32_t a = 0, b = 0;
    if (valid) {
        const uint32_t sg = seg[i];
        a = segstart[sg];
        b = segstart[sg + 1];
    }
    const bool small = valid && (b - a <= small_seg);
    // a pair (by far the most common tied segment: two copies of a repeat) is compared once: its first member walks
    // the two suffixes, the second one -- the next lane -- takes the opposite rank
    const bool second_in_wave = small && (b - a == 2) && ((uint32_t)i == a + 1) && lane_id() > 0;
    uint32_t r = 0;
    uint64_t pi = 0;
    // taint (KISS_CTX_TAINT): this item is tied with another one through the full depth (a walk of deep_less, here or in
    // k_seg_adjacent, ended without a difference): its place among its mates is the tie rule's
    bool taint = false;
    bool tied_before = false; // tied with a mate that precedes it in the final order (position order among mates)
    if (small) {
        pi = pos[i];
        if (!second_in_wave && b - a == 2) {
            // a pair: no keys were gathered for it (k_gather_keys); one walk from the segment's depth decides
            const uint32_t j = (uint32_t)i == a ? a + 1 : a;
            r = deep_less(pk, n, pos[j], pi, off - 32, depth, j < (uint32_t)i, &taint) ? 1u : 0u;
        } else if (!second_in_wave) {
            const uint64_t ki = key[i];
            bool sorted = false;
            if (b - a >= 3) { // adjacent pairs all in order -> already sorted
                sorted = true;
                for (uint32_t j = a; j + 1 < b; j++) sorted = sorted && inorder[j];
            }
            if (sorted) {
                r = (uint32_t)i - a;
                tied_before = (uint32_t)i > a && inorder[i - 1] == 2;
                taint = tied_before || ((uint32_t)i + 1 < b && inorder[i] == 2);
            } else {
                for (uint32_t j = a; j < b; j++) {
                    if (j == (uint32_t)i) continue;
                    uint64_t kj = key[j];
                    bool jless;
                    if (kj != ki) jless = kj < ki;
                    else {
                        bool tj = false;
                        jless = deep_less(pk, n, pos[j], pi, off, depth, j < (uint32_t)i, &tj);
                        taint = taint || tj;
                        tied_before = tied_before || (tj && jless);
                    }
                    r += jless ? 1u : 0u;
                }
            }
        }
    }
    const uint32_t r_prev = __shfl_up(r, 1, 64);
    const bool t_prev = __shfl_up((int)taint, 1, 64) != 0;
    if (second_in_wave) {
        r = 1u - r_prev;
        taint = t_prev; // the pair's first member did the comparison
    }
    if (small && b - a == 2) tied_before = taint && r == 1;
    if (small) {
        const uint32_t dst = slot[a + r];
        out[dst] = (uint32_t)pi;
        if (hfar && tied_before) hfar[dst] = 0;
        // the context word a finished item brought along from round 0 (key payload) spares the placement step a
        // random text gather; items that stay tied past this round get theirs gathered there (and are tainted there)
        if (tctx) octx[dst] = tctx[i] | (taint ? KISS_CTX_TAINT : 0u);
        else if (taint) octx[dst] = KISS_CTX_TAINT; // no word yet (gathered at placement), but tainted
        big[i] = 0;
    } else if (valid) {
        big[i] = (uint32_t)i == a ? (uint8_t)3 : (uint8_t)1;
    }
    (void)nbig; // the number of big-segment items comes out of the flag scan (one address hit by every wave's
                // atomicAdd cost more than the rest of this kernel)
}

// ---- big segments: three-way split around a pivot key before any radix pass ---------------------------------
// The members of a long tandem array stay tied round after round: most of a big segment carries ONE key (the
// unmutated continuation of the repeat) and only the items whose 32-base window holds a mutation differ.  Sorting
// such a segment on (segment, key) costs 8 + 3 radix passes over all of it.  Instead: pivot = the key of the
// segment's middle item; a stable segmented partition into < pivot | == pivot | > pivot (one scan, one scatter);
// only the < and > groups -- now segments of their own -- go through the radix sort.  The result is the same
// (segment, key) order.  cls[i] = [key < pivot] | [key == pivot] << 32.
__device__ __forceinline__ uint32_t pivot_class(const uint64_t *__restrict__ bkey, uint32_t a, uint32_t b, uint64_t k)
{
    const uint64_t pv = bkey[(a + b) >> 1];
    return k < pv ? 0u : (k == pv ? 1u : 2u);
}

__global__ __launch_bounds__(LS_THREADS) void k_pivot_class(const uint64_t *__restrict__ bkey,
                                                           const uint32_t *__restrict__ bseg,
                                                           const uint32_t *__restrict__ bsegstart, uint64_t nbig,
                                                           uint64_t *__restrict__ cls)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= nbig) return;
    const uint32_t sid = bseg[i];
    const uint32_t c = pivot_class(bkey, bsegstart[sid], bsegstart[sid + 1], bkey[i]);
    cls[i] = c == 0 ? 1ull : (c == 1 ? (1ull << 32) : 0ull);
}

// ex = exclusive scan of cls, tot[0] = its grand total.  Items move inside their segment; ne[d] flags the members of
// the < and > groups at their new place: (1 << 32) | starts-a-group.
__global__ __launch_bounds__(LS_THREADS) void k_pivot_scatter(const uint64_t *__restrict__ bkey,
                                                             const uint32_t *__restrict__ bpos,
                                                             const uint32_t *__restrict__ bseg,
                                                             const uint32_t *__restrict__ bsegstart, uint64_t nbig,
                                                             const uint64_t *__restrict__ cls,
                                                             const uint64_t *__restrict__ ex,
                                                             const uint64_t *__restrict__ tot,
                                                             uint64_t *__restrict__ okey, uint32_t *__restrict__ opos,
                                                             uint64_t *__restrict__ ne)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= nbig) return;
    const uint32_t sid = bseg[i];
    const uint32_t a = bsegstart[sid], b = bsegstart[sid + 1];
    const uint64_t ea = ex[a], eb = (uint64_t)b < nbig ? ex[b] : tot[0], ei = ex[i], ci = cls[i];
    const uint32_t t0 = (uint32_t)eb - (uint32_t)ea, t1 = (uint32_t)(eb >> 32) - (uint32_t)(ea >> 32);
    const uint32_t lt_before = (uint32_t)ei - (uint32_t)ea, eq_before = (uint32_t)(ei >> 32) - (uint32_t)(ea >> 32);
    const uint32_t c = (uint32_t)ci ? 0u : ((ci >> 32) ? 1u : 2u);
    uint32_t d;
    if (c == 0) d = a + lt_before;
    else if (c == 1) d = a + t0 + eq_before;
    else d = a + t0 + t1 + ((uint32_t)i - a - lt_before - eq_before);
    okey[d] = bkey[i];
    opos[d] = bpos[i];
    uint64_t f = 0;
    if (c == 0) f = (1ull << 32) | (uint64_t)(d == a ? 1u : 0u);
    else if (c == 2) f = (1ull << 32) | (uint64_t)(d == a + t0 + t1 ? 1u : 0u);
    ne[d] = f;
}

// members of long groups -> compact arrays (dense group ids), remembering where they came from
__global__ __launch_bounds__(LS_THREADS) void k_big_extract(const uint64_t *__restrict__ key,
                                                           const uint32_t *__restrict__ pos, uint64_t count,
                                                           const uint64_t *__restrict__ big,
                                                           const uint64_t *__restrict__ ex, uint64_t *__restrict__ bkey,
                                                           uint32_t *__restrict__ bpos, uint32_t *__restrict__ bseg,
                                                           uint32_t *__restrict__ bidx)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    const uint64_t f = big[i];
    if (!(f >> 32)) return;
    const uint64_t e = ex[i];
    const uint32_t kx = (uint32_t)(e >> 32);
    bkey[kx] = key[i];
    bpos[kx] = pos[i];
    bseg[kx] = (uint32_t)(e & 0xFFFFFFFFull) + (uint32_t)(f & 1ull) - 1u;
    bidx[kx] = (uint32_t)i;
}

// sorted on (group, key): the c-th item goes back to where the c-th extracted item came from
__global__ __launch_bounds__(LS_THREADS) void k_big_writeback(const uint64_t *__restrict__ skey,
                                                             const uint32_t *__restrict__ spos,
                                                             const uint32_t *__restrict__ bidx, uint64_t nbig,
                                                             uint64_t *__restrict__ okey, uint32_t *__restrict__ opos)
{
    const uint64_t c = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (c >= nbig) return;
    const uint32_t i = bidx[c];
    okey[i] = skey[c];
    opos[i] = spos[c];
}

// ---- pivot rounds: the members of a big segment against one reference string, to the full depth -------------------------------------
// What is still tied in big segments after two rounds is long tandem arrays (and the odd high-copy repeat).  Their
// members stay tied round after round -- eleven more 32-base rounds of ~40 launches and an 8 + 2-pass radix sort each at
// D = 375 -- although one comparison says almost everything: every member is compared with ONE reference string (the
// segment's consensus, see k_pivot_lcp) from the current depth to the full depth D, giving
//     side = member < reference | equal through D | member > reference,   d = first differing base,   c = the member's base there.
// Two members on the < side order by (d ascending, c); on the > side by (d descending, c); members equal through D
// are final in their present (= text position) order.  One stable radix sort on (segment, side, d', c) -- 13 bits + the
// segment id at k = 256 -- replaces the rounds.  Members with the same (side, d, c) are tied through off + d + 1 bases:
// they form a new, much smaller segment that goes round again from the SAME depth `off` (so all survivors still share
// one depth; the few bases they walk twice cost less than a per-segment depth would).  Progress: if every member of a
// segment fell into the same (side, d, c) class, the three members the reference is made of would all carry base c at
// d, and so would the reference -- so a segment always splits into at least two classes or retires as a whole.
// Bounded depth only: in exact order (D unbounded) "equal through D" does not exist, the 32-base rounds run instead.
// four aligned 32-base keys starting at base q (five independent word loads)
__device__ __forceinline__ void keys128(const uint64_t *__restrict__ pk, uint64_t q, uint64_t k[4])
{
    const uint32_t sh = 2u * (uint32_t)(q & 31u);
    uint64_t x[5];
    kiss_words5(pk, q >> 5, x);
#pragma unroll
    for (int t = 0; t < 4; t++) k[t] = (x[t] << sh) | ((x[t + 1] >> 1) >> (63u - sh));
}

__device__ __forceinline__ uint64_t maj3(uint64_t a, uint64_t b, uint64_t c) { return (a & b) | (a & c) | (b & c); }

// Key of a member: up to `slots` deviations from the reference, most significant first, then a "complete" bit:
//     [side:2 | d':dbits | c:2] x slots ... [complete:1]          in the TOP bits of a 64-bit word
// side 0: the member's base is smaller than the reference's at d (d' = d: the earlier the deviation, the smaller the
// member), side 2: larger (d' = rem - 1 - d: the earlier, the larger), side 1: no further deviation through the depth
// (terminator; the walk is complete and the key says everything the comparator can see).  Members that are equal up to
// and including a deviation keep following the same argument from the next base on, so the slots compare
// lexicographically.  After a substitution a tandem array is back in phase with its consensus, so the next slot holds
// the member's next mutation: one round looks `slots` mutations deep.
__global__ __launch_bounds__(LS_THREADS) void k_pivot_lcp(const uint64_t *__restrict__ pk,
                                                         const uint32_t *__restrict__ bpos,
                                                         const uint32_t *__restrict__ bseg,
                                                         const uint32_t *__restrict__ bsegstart, uint64_t nbig,
                                                         uint64_t off, uint64_t depth, int dbits, int slots,
                                                         uint64_t *__restrict__ keyout, int frac_den, int with_ctx)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= nbig) return;
    const uint32_t sid = bseg[i];
    const uint32_t a = bsegstart[sid], b = bsegstart[sid + 1], len = b - a;
    // The reference the members are compared with may be ANY fixed string (the order argument only needs it to be the
    // same for the whole segment).  A real member carries its own mutations, and everything that follows the consensus
    // up to that member's first mutation would land in one class.  The bitwise majority of three members is the
    // consensus wherever at most one of them is mutated: members of a tandem array then deviate from the reference at
    // their OWN mutations.
    const uint64_t p = (uint64_t)bpos[i] + off;
    // Which three: the quarters of the segment in one round, 2/7 4/7 5/7 in the next, 3/11 5/11 8/11 in the third, and
    // round again -- arrays of equal length (telomeres) put the quarters of a segment exactly at array ends, where
    // members stop following the consensus; the next round then picks elsewhere.
    const uint64_t f1 = frac_den == 4 ? 1 : (frac_den == 7 ? 2 : 3), f2 = frac_den == 4 ? 2 : (frac_den == 7 ? 4 : 5),
                   f3 = frac_den == 4 ? 3 : (frac_den == 7 ? 5 : 8);
    const uint64_t q1 = (uint64_t)bpos[a + (uint32_t)(((uint64_t)len * f1) / (uint64_t)frac_den)] + off,
                   q2 = (uint64_t)bpos[a + (uint32_t)(((uint64_t)len * f2) / (uint64_t)frac_den)] + off,
                   q3 = (uint64_t)bpos[a + (uint32_t)(((uint64_t)len * f3) / (uint64_t)frac_den)] + off;
    const uint64_t rem = depth - off; // all of them are far suffixes: position + rem lies inside the text
    const int slotbits = dbits + 4;
    uint64_t key = 0;
    int used = 0;
    bool complete = false;
    auto take = [&](uint64_t kx, uint64_t kr, uint64_t base0) { // deviations inside one 32-base word, in text order
        uint64_t diff = kx ^ kr;
        diff = (diff | (diff >> 1)) & 0x5555555555555555ull; // one flag per base (its low bit)
        while (diff && used < slots) {
            const uint32_t lz = (uint32_t)__clzll((long long)diff) >> 1; // base index inside the word
            const uint32_t sh = 62u - 2u * lz;
            const uint64_t cx = (kx >> sh) & 3ull, cr = (kr >> sh) & 3ull;
            const uint64_t d = base0 + lz;
            const uint64_t side = cx < cr ? 0ull : 2ull;
            const uint64_t dk = side == 0 ? d : rem - 1 - d;
            key = (key << slotbits) | (side << (dbits + 2)) | (dk << 2) | cx;
            used++;
            diff &= ~(1ull << sh);
        }
    };
    uint64_t done = 0;
    while (done < rem && used < slots) {
        if (rem - done >= 128) { // four words per step: the loads of one step are independent (one round trip to memory)
            uint64_t kx[4], k1[4], k2[4], k3[4];
            keys128(pk, p + done, kx);
            keys128(pk, q1 + done, k1);
            keys128(pk, q2 + done, k2);
            keys128(pk, q3 + done, k3);
#pragma unroll
            for (int t = 0; t < 4; t++) take(kx[t], maj3(k1[t], k2[t], k3[t]), done + 32u * (uint32_t)t);
            done += 128;
            continue;
        }
        uint64_t ki = kiss_key32(pk, p + done);
        uint64_t kr = maj3(kiss_key32(pk, q1 + done), kiss_key32(pk, q2 + done), kiss_key32(pk, q3 + done));
        if (rem - done < 32) {
            const uint64_t mask = ~0ull << (64 - 2 * (rem - done));
            ki &= mask;
            kr &= mask;
        }
        take(ki, kr, done);
        done += 32;
    }
    if (used < slots) { // the walk reached the depth: terminator slot, the key is complete
        key = (key << slotbits) | (1ull << (dbits + 2));
        used++;
        complete = true;
    }
    key <<= (uint64_t)slotbits * (uint64_t)(slots - used); // unused slots: zeros (only behind a terminator)
    key = (key << 1) | (complete ? 1ull : 0ull);
    // The low 24 bits of the word are not sorted on (with_ctx: the key ends at bit 24 or above) and carry the member's
    // short context word, like the low bits of a round-0 key: a member that retires in this round gets its word without
    // the random text gather of the induction (the bases in front of the member sit in the sector this walk has read)
    keyout[i] = (key << (64 - (slots * slotbits + 1))) | (with_ctx ? (uint64_t)kiss_load_ctx_n(pk, bpos[i], KISS_KEY_CTX_BASES) : 0ull);
}

// boundaries of the new segments in the sorted list: one byte per item, 1 = (segment, key) differs from the predecessor's,
// or the item's key is complete (it says everything the comparator sees: the item is final where the stable sort left it;
// it and its successor both start a "segment", which makes it a singleton for the compaction)
__global__ __launch_bounds__(LS_THREADS) void k_pivot_heads(const uint64_t *__restrict__ key,
                                                           const uint32_t *__restrict__ seg, uint64_t nbig,
                                                           int complete_bit, uint8_t *__restrict__ heads,
                                                           uint64_t cmp_mask) // the key bits (a payload may sit below them)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= nbig) return;
    const uint64_t k = key[i] & cmp_mask;
    const bool complete = ((k >> complete_bit) & 1ull) != 0;
    const bool same_prev = i > 0 && seg[i] == seg[i - 1] && (key[i - 1] & cmp_mask) == k;
    uint8_t h = 1;
    if (same_prev && !complete) h = 0;
    // bit 1: a complete key shared with a neighbour = equal through the full depth: final here by the tie rule (taint)
    if (complete && (same_prev || (i + 1 < nbig && seg[i + 1] == seg[i] && (key[i + 1] & cmp_mask) == k))) h |= 2;
    if (complete && same_prev) h |= 4; // bit 2: ... and that neighbour is the predecessor (ctx->hfar)
    heads[i] = h;
}



// ---- fused flag + compaction over 2048-item tiles (no per-item flag / scan arrays) ------------------------
// pass 1: per-tile (survivors << 32 | surviving heads); pass 2 (after an exclusive scan over tiles): the same
// flags again, wave-ballot prefix inside the tile, survivors compacted, singletons retired.
constexpr int FC_THREADS = 256;
constexpr int FC_ITEMS = 8;
constexpr int FC_ITEMS_DEFAULT = FC_ITEMS;
constexpr int FC_TILE = FC_THREADS * FC_ITEMS;

// where a segment starts: FC_KEY = the key changes, FC_KEY_SEG = the key or the segment id changes,
// FC_HEADS = `key` points at one byte per item (1 = starts a segment)
constexpr int FC_KEY = 0, FC_KEY_SEG = 1, FC_HEADS = 2;

template <int SRC>
__device__ __forceinline__ void fc_flags(const uint64_t *__restrict__ key, const uint32_t *__restrict__ seg, uint64_t i,
                                         uint64_t count, int cmp_shift, int last_round, bool &surv, bool &shead,
                                         bool *tie = nullptr, // *tie: the item shares everything compared with a neighbour
                                         bool *tie_prev = nullptr) // ... with its predecessor
{
    surv = shead = false;
    if (tie) *tie = false;
    if (tie_prev) *tie_prev = false;
    if (i >= count) return;
    bool head, nhead;
    if constexpr (SRC == FC_HEADS) {
        const uint8_t *hb = reinterpret_cast<const uint8_t *>(key);
        head = (i == 0) || hb[i] != 0;
        nhead = (i + 1 == count) || hb[i + 1] != 0;
        if (tie) *tie = (hb[i] & 2) != 0; // the producer of the head bytes says so (k_pivot_heads)
        if (tie_prev) *tie_prev = (hb[i] & 4) != 0;
    } else {
        constexpr bool HAS_SEG = SRC == FC_KEY_SEG;
        uint64_t k = key[i] >> cmp_shift;
        uint32_t s = HAS_SEG ? seg[i] : 0u;
        head = (i == 0) || (key[i - 1] >> cmp_shift) != k || (HAS_SEG && seg[i - 1] != s);
        nhead = (i + 1 == count) || (key[i + 1] >> cmp_shift) != k || (HAS_SEG && seg[i + 1] != s);
    }
    surv = !(head && nhead) && !last_round;
    shead = surv && head;
    if constexpr (SRC != FC_HEADS) {
        if (tie) *tie = !(head && nhead);
        if (tie_prev) *tie_prev = !head;
    }
}

template <int SRC>
__global__ __launch_bounds__(FC_THREADS) void k_fc_count(const uint64_t *__restrict__ key,
                                                        const uint32_t *__restrict__ seg, uint64_t count,
                                                        int cmp_shift, int last_round, uint64_t *__restrict__ tcnt)
{
    __shared__ uint32_t ws[FC_THREADS / 64][2];
    const int wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * FC_TILE + (uint64_t)wave * (FC_ITEMS * 64) + lane_id();
    uint32_t ns = 0, nh = 0;
#pragma unroll
    for (int j = 0; j < FC_ITEMS; j++) {
        bool sv, sh;
        fc_flags<SRC>(key, seg, base + (uint64_t)j * 64, count, cmp_shift, last_round, sv, sh);
        ns += (uint32_t)__popcll(__ballot(sv));
        nh += (uint32_t)__popcll(__ballot(sh));
    }
    if (lane_id() == 0) {
        ws[wave][0] = ns;
        ws[wave][1] = nh;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t a = 0, b = 0;
        for (int w = 0; w < FC_THREADS / 64; w++) {
            a += ws[w][0];
            b += ws[w][1];
        }
        tcnt[blockIdx.x] = (a << 32) | b;
    }
}

// out: retired items land in out[slot] (nullptr: they already are where they belong);
// isa (optional): inverse array, isa[position] = slot for every retired item;
// octx (optional, round 0 only): context word per slot from the key payload, 0 for slots still tied
template <int SRC, bool HAS_SLOT>
__global__ __launch_bounds__(FC_THREADS) void k_fc_compact(const uint64_t *__restrict__ key,
                                                          const uint32_t *__restrict__ seg,
                                                          const uint32_t *__restrict__ pos,
                                                          const uint32_t *__restrict__ slot, uint64_t count,
                                                          int cmp_shift, int last_round,
                                                          const uint64_t *__restrict__ tex, // exclusive scan of tcnt
                                                          uint32_t *__restrict__ npos, uint32_t *__restrict__ nslot,
                                                          uint32_t *__restrict__ nseg, uint32_t *__restrict__ nsegstart,
                                                          uint32_t *__restrict__ out, uint32_t *__restrict__ isa,
                                                          uint32_t *__restrict__ octx,
                                                          uint32_t *__restrict__ tmark, // taint marks of retiring items
                                                          const uint64_t *__restrict__ payload, // optional, with tmark: words
                                                          // whose low KISS_KEY_CTX bits are the items' context words
                                                          int isa_shift, // isa is indexed by position >> isa_shift
                                                          const uint64_t *__restrict__ pk_fix, // optional, with cfix: the
                                                          uint32_t *__restrict__ cfix, // context word of a retiring item's own
                                                          // position is gathered from the packed text into cfix[slot]
                                                          uint8_t *__restrict__ hmark) // optional: hmark[slot] = 0 for an item
                                                          // that retires tied with its predecessor (ctx->hfar)
{
    __shared__ uint32_t ws[FC_THREADS / 64][2];
    const int wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * FC_TILE + (uint64_t)wave * (FC_ITEMS * 64) + lane_id();
    uint32_t rs[FC_ITEMS], rh[FC_ITEMS]; // rank among survivors / surviving heads inside the wave (inclusive for heads)
    uint32_t fl = 0;                      // bit 2j = survivor, bit 2j+1 = surviving head
    uint32_t tl = 0;                      // bit j = retires while tied with a neighbour (last round / complete pivot key)
    uint32_t tp = 0;                      // bit j = ... and that neighbour is its predecessor
    uint32_t ns = 0, nh = 0;
#pragma unroll
    for (int j = 0; j < FC_ITEMS; j++) {
        bool sv, sh, ti, tpv;
        fc_flags<SRC>(key, seg, base + (uint64_t)j * 64, count, cmp_shift, last_round, sv, sh, &ti, &tpv);
        tl |= (ti && !sv ? 1u : 0u) << j;
        tp |= (ti && tpv && !sv ? 1u : 0u) << j;
        const uint64_t ms = __ballot(sv), mh = __ballot(sh);
        rs[j] = ns + (uint32_t)__popcll(ms & lanemask_lt());
        rh[j] = nh + (uint32_t)__popcll(mh & lanemask_lt()) + (sh ? 1u : 0u);
        ns += (uint32_t)__popcll(ms);
        nh += (uint32_t)__popcll(mh);
        fl |= (sv ? 1u : 0u) << (2 * j) | (sh ? 2u : 0u) << (2 * j);
    }
    if (lane_id() == 0) {
        ws[wave][0] = ns;
        ws[wave][1] = nh;
    }
    __syncthreads();
    const uint64_t te = tex[blockIdx.x];
    uint32_t bs = (uint32_t)(te >> 32), bh = (uint32_t)(te & 0xFFFFFFFFull);
    for (int w = 0; w < wave; w++) {
        bs += ws[w][0];
        bh += ws[w][1];
    }
#pragma unroll
    for (int j = 0; j < FC_ITEMS; j++) {
        const uint64_t i = base + (uint64_t)j * 64;
        if (i >= count) continue;
        const uint32_t p = pos[i];
        const uint32_t sl = HAS_SLOT ? slot[i] : (uint32_t)i;
        if ((fl >> (2 * j)) & 1u) {
            const uint32_t ni = bs + rs[j];
            const uint32_t sid = bh + rh[j] - 1u; // heads up to and including this item's own segment head
            npos[ni] = p;
            nslot[ni] = sl;
            nseg[ni] = sid;
            if ((fl >> (2 * j)) & 2u) nsegstart[sid] = ni;
            if (octx) octx[sl] = 0; // tied so far: its context word is gathered at placement
        } else {
            if (out) out[sl] = p;
            if (tmark && payload) // the word came along with the sort key (k_pivot_lcp)
                tmark[sl] = (uint32_t)(payload[i] & KISS_KEY_CTX_MASK) | (((tl >> j) & 1u) ? KISS_CTX_TAINT : 0u);
            else if (tmark && ((tl >> j) & 1u)) tmark[sl] = KISS_CTX_TAINT; // no context word yet: gathered at placement
            if (isa) isa[p >> isa_shift] = sl;
            if (cfix) cfix[sl] = kiss_load_ctx(pk_fix, p);
            if (hmark && ((tp >> j) & 1u)) hmark[sl] = 0;
            if constexpr (SRC != FC_HEADS) {
                if (octx) octx[sl] = (uint32_t)(key[i] & KISS_KEY_CTX_MASK); // round 0: payload of the classification key
            }
        }
    }
}

// ---- round 0 (keys only, slot = index): the same flag + compaction with 16-byte accesses ----------------------
// A thread owns FC_ITEMS consecutive items (two keys per load, four positions per load); survivor ranks come from a
// wave prefix of the per-thread counts.  Every slot gets its position (a tied suffix's slot is rewritten when it
// retires later) and its context word (0 while tied), so the stores are plain 16-byte streams.
template <int FC_ITEMS = FC_ITEMS_DEFAULT>
__device__ __forceinline__ void fc0_load(const uint64_t *__restrict__ key, uint64_t i0, uint64_t count, int cmp_shift,
                                         uint64_t k[FC_ITEMS + 2], uint32_t &valid, uint32_t *low = nullptr)
{
    // k[0] = key before my first item, k[1..FC_ITEMS] = my items, k[FC_ITEMS + 1] = key after; all >> cmp_shift
    valid = i0 >= count ? 0u : (count - i0 >= FC_ITEMS ? (uint32_t)FC_ITEMS : (uint32_t)(count - i0));
    if (valid == FC_ITEMS) {
#pragma unroll
        for (int q = 0; q < FC_ITEMS / 2; q++) {
            const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(key + i0 + 2 * q);
            k[1 + 2 * q] = kk.x >> cmp_shift;
            k[2 + 2 * q] = kk.y >> cmp_shift;
            if (low) { // the payload bits below the compared ones
                low[2 * q] = (uint32_t)(kk.x & KISS_KEY_CTX_MASK);
                low[2 * q + 1] = (uint32_t)(kk.y & KISS_KEY_CTX_MASK);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < FC_ITEMS; e++) {
            const uint64_t kk = (uint32_t)e < valid ? key[i0 + e] : 0ull;
            k[1 + e] = kk >> cmp_shift;
            if (low) low[e] = (uint32_t)(kk & KISS_KEY_CTX_MASK);
        }
    }
    k[0] = (valid && i0 > 0) ? (key[i0 - 1] >> cmp_shift) : 0ull;
    k[FC_ITEMS + 1] = (valid && i0 + FC_ITEMS < count) ? (key[i0 + FC_ITEMS] >> cmp_shift) : 0ull;
}

template <int FC_ITEMS = FC_ITEMS_DEFAULT>
__device__ __forceinline__ void fc0_flags(const uint64_t k[FC_ITEMS + 2], uint64_t i0, uint64_t count, uint32_t valid,
                                          uint32_t &survmask, uint32_t &headmask)
{
    survmask = headmask = 0;
#pragma unroll
    for (int e = 0; e < FC_ITEMS; e++) {
        if ((uint32_t)e < valid) {
            const uint64_t i = i0 + (uint64_t)e;
            const bool head = i == 0 || k[e] != k[e + 1];
            const bool nhead = i + 1 == count || k[e + 2] != k[e + 1];
            const bool surv = !(head && nhead);
            survmask |= (surv ? 1u : 0u) << e;
            headmask |= ((surv && head) ? 1u : 0u) << e;
        }
    }
}

__global__ __launch_bounds__(FC_THREADS) void k_fc0_count(const uint64_t *__restrict__ key, uint64_t count, int cmp_shift,
                                                         uint64_t *__restrict__ tcnt)
{
    __shared__ uint32_t ws[FC_THREADS / 64][2];
    const uint64_t i0 = ((uint64_t)blockIdx.x * FC_THREADS + threadIdx.x) * FC_ITEMS;
    uint64_t k[FC_ITEMS + 2];
    uint32_t valid, sm, hm;
    fc0_load(key, i0, count, cmp_shift, k, valid);
    fc0_flags(k, i0, count, valid, sm, hm);
    uint32_t ns = (uint32_t)__popc(sm), nh = (uint32_t)__popc(hm);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ns += __shfl_xor(ns, d, 64);
        nh += __shfl_xor(nh, d, 64);
    }
    if (lane_id() == 0) {
        ws[threadIdx.x >> 6][0] = ns;
        ws[threadIdx.x >> 6][1] = nh;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t a = 0, b = 0;
        for (int w = 0; w < FC_THREADS / 64; w++) {
            a += ws[w][0];
            b += ws[w][1];
        }
        tcnt[blockIdx.x] = (a << 32) | b;
    }
}

__global__ __launch_bounds__(FC_THREADS) void k_fc0_compact(const uint64_t *__restrict__ key,
                                                           const uint32_t *__restrict__ pos, uint64_t count,
                                                           int cmp_shift, const uint64_t *__restrict__ tex,
                                                           uint32_t *__restrict__ npos, uint32_t *__restrict__ nslot,
                                                           uint32_t *__restrict__ nseg, uint32_t *__restrict__ nsegstart,
                                                           uint32_t *__restrict__ out, uint32_t *__restrict__ octx,
                                                           uint32_t *__restrict__ nctx)
{
    __shared__ uint32_t ws[FC_THREADS / 64][2];
    const int wave = threadIdx.x >> 6;
    const uint64_t i0 = ((uint64_t)blockIdx.x * FC_THREADS + threadIdx.x) * FC_ITEMS;
    uint64_t k[FC_ITEMS + 2];
    uint32_t valid, sm, hm;
    uint32_t p[FC_ITEMS], cw[FC_ITEMS];
    fc0_load(key, i0, count, cmp_shift, k, valid, cw);
    fc0_flags(k, i0, count, valid, sm, hm);
    const uint32_t ns = (uint32_t)__popc(sm), nh = (uint32_t)__popc(hm);
    uint32_t is = ns, ih = nh; // inclusive wave prefixes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t os = __shfl_up(is, d, 64), oh = __shfl_up(ih, d, 64);
        if ((int)lane_id() >= d) {
            is += os;
            ih += oh;
        }
    }
    if (lane_id() == 63) {
        ws[wave][0] = is;
        ws[wave][1] = ih;
    }
    __syncthreads();
    const uint64_t te = tex[blockIdx.x];
    uint32_t bs = (uint32_t)(te >> 32) + (is - ns), bh = (uint32_t)(te & 0xFFFFFFFFull) + (ih - nh);
    for (int w = 0; w < wave; w++) {
        bs += ws[w][0];
        bh += ws[w][1];
    }
    if (valid == 0) return;
    // cw = payload of my keys: the low KISS_KEY_CTX bits (cmp_shift >= 24 in round 0), kept from the one key load
    if (valid == FC_ITEMS) {
#pragma unroll
        for (int q = 0; q < FC_ITEMS / 4; q++) {
            const uint4 t = *reinterpret_cast<const uint4 *>(pos + i0 + 4 * q);
            p[4 * q] = t.x;
            p[4 * q + 1] = t.y;
            p[4 * q + 2] = t.z;
            p[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < FC_ITEMS; e++) p[e] = (uint32_t)e < valid ? pos[i0 + e] : 0u;
    }
#pragma unroll
    for (int e = 0; e < FC_ITEMS; e++) {
        if ((sm >> e) & 1u) {
            const uint32_t ni = bs++;
            if ((hm >> e) & 1u) bh++;
            const uint32_t sid = bh - 1u; // heads up to and including this item's own segment head
            npos[ni] = p[e];
            nslot[ni] = (uint32_t)(i0 + e);
            nseg[ni] = sid;
            if ((hm >> e) & 1u) nsegstart[sid] = ni;
            if (nctx) nctx[ni] = cw[e]; // travels with the tied item through the first refinement round (k_seg_finish)
            cw[e] = 0;                  // tied so far: written when the item is finished, else gathered at placement
        }
    }
    // out == nullptr: the positions already are where they belong (the last radix pass wrote them into the output list)
    if (valid == FC_ITEMS) {
#pragma unroll
        for (int q = 0; q < FC_ITEMS / 4; q++) {
            if (out) *reinterpret_cast<uint4 *>(out + i0 + 4 * q) = make_uint4(p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]);
            *reinterpret_cast<uint4 *>(octx + i0 + 4 * q) = make_uint4(cw[4 * q], cw[4 * q + 1], cw[4 * q + 2], cw[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < FC_ITEMS; e++)
            if ((uint32_t)e < valid) {
                if (out) out[i0 + e] = p[e];
                octx[i0 + e] = cw[e];
            }
    }
}

// ---- round 0 in ONE pass over the sorted keys ---------------------------------------------------------------------
// k_fc0_count + scan + k_fc0_compact read the 7 GB of sorted keys twice.  Here a tile of FC1_TILE items flags its items,
// publishes its (survivors, segment heads) pair, finds the sum over all earlier tiles by a decoupled look-back and
// compacts -- the keys are read once.  One 64-bit descriptor per tile: [status:2 | survivors:31 | heads:31] (status 1 =
// this tile's counts, 2 = counts of this and all earlier tiles); wave 0 inspects 64 predecessors per round trip.  Tiles
// are numbered by an atomic ticket, so every predecessor of a running tile is running or done, and the wait is bounded
// like the radix look-back (radix.hip) -- an error flag instead of a hung GPU.  The host learns the totals only
// afterwards, so stores beyond `cap` (the tied-segment arrays) are dropped; the caller regrows and runs the pass again.
// Layout (round 4): STRIPED.  A lane takes two neighbouring items per 16-byte key load and the 64 lanes of a wave take
// 128 neighbouring items per load instruction (1 KiB, every byte of every line used); a wave owns FC1_ITEMS / 2 such
// chunks.  The blocked form of round 3 (a thread owned 8 neighbouring items = 64 bytes, so ONE load instruction of a wave
// touched 64 different lines and used 16 bytes of each; the 16 waves' 64 KiB of lines did not fit the 32 KiB L1) spent
// its time re-fetching lines from L2: 7.4 ms for 17 GB, and 10.4 / 15.3 ms with 16 / 32 items per thread.  Flags come
// from one lane shift of the keys per chunk, ranks from ballots; a wave's stores of one chunk go to consecutive
// survivor slots.  The survivor order (rank = tied items in front, in list order) is the same as before.
#ifndef KISS_FC1_THREADS
#define KISS_FC1_THREADS 1024
#endif
#ifndef KISS_FC1_ITEMS
#define KISS_FC1_ITEMS 8
#endif
#ifndef KISS_FC1_MIN_WAVES
#define KISS_FC1_MIN_WAVES 1
#endif
#ifndef KISS_FC1_STORE_UNROLL
#define KISS_FC1_STORE_UNROLL 4 // (= all chunks; 1 saves 12 registers and buys no second workgroup per CU)
#endif
constexpr int FC1_THREADS = KISS_FC1_THREADS;
constexpr int FC1_ITEMS = KISS_FC1_ITEMS;
constexpr int FC1_CHUNKS = FC1_ITEMS / 2;         // chunks of 128 items per wave
constexpr int FC1_WAVE_ITEMS = 64 * FC1_ITEMS;    // 512
constexpr int FC1_TILE = FC1_THREADS * FC1_ITEMS; // 8192
constexpr uint32_t FC1_SPIN_LIMIT = 1u << 22;
static_assert(FC1_ITEMS % 2 == 0 && FC1_ITEMS <= 16, "two items per lane and chunk; four flag bits per chunk in one word");

#ifdef FC_PROF
// -DFC_PROF (tools/build_prof.sh): wall-clock ticks between the phase boundaries of a tile, summed over all tiles by thread 0
__device__ unsigned long long fc_prof[12];
#define FC_MARK(i)                                                                                                     \
    do {                                                                                                               \
        if (threadIdx.x == 0) {                                                                                        \
            const unsigned long long now_ = wall_clock64();                                                            \
            atomicAdd(&fc_prof[i], now_ - t_prev_);                                                                    \
            t_prev_ = now_;                                                                                            \
        }                                                                                                              \
    } while (0)
#else
#define FC_MARK(i)
#endif
__device__ __forceinline__ uint64_t lane_value_u64(uint64_t v, int src_lane) // (the same lane in all callers)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(FC1_THREADS, KISS_FC1_MIN_WAVES) void k_fc0_onepass(const uint64_t *__restrict__ key,
                                                            const uint32_t *__restrict__ pos, uint64_t count,
                                                            int cmp_shift, uint64_t tiles, uint64_t *__restrict__ desc,
                                                            uint32_t *__restrict__ ticket, uint32_t *__restrict__ err,
                                                            uint32_t *__restrict__ npos, uint32_t *__restrict__ nslot,
                                                            uint32_t *__restrict__ nseg, uint32_t *__restrict__ nsegstart,
                                                            uint32_t *__restrict__ octx, uint32_t *__restrict__ nctx,
                                                            uint64_t cap, uint64_t *__restrict__ total)
{
    __shared__ uint32_t ws[FC1_THREADS / 64][2];
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_excl[2];
#ifdef FC_PROF
    unsigned long long t_prev_ = wall_clock64();
#endif
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    FC_MARK(0); // ticket + barrier
    const uint64_t tile = s_tile;
    if (tile >= tiles) return;
    const int wave = threadIdx.x >> 6;
    const uint32_t lane = lane_id();
    const uint64_t wbase = tile * FC1_TILE + (uint64_t)wave * FC1_WAVE_ITEMS; // first item of my wave
    const bool full = wbase + FC1_WAVE_ITEMS <= count;                        // (wave-uniform)
    uint64_t k0[FC1_CHUNKS], k1[FC1_CHUNKS]; // compared bits of my two items of chunk j
    uint32_t c0[FC1_CHUNKS], c1[FC1_CHUNKS]; // their payload: the low KISS_KEY_CTX bits (cmp_shift >= 24 in round 0)
#pragma unroll
    for (int j = 0; j < FC1_CHUNKS; j++) {
        const uint64_t a = wbase + (uint64_t)j * 128 + 2 * lane;
        ulonglong2 kk;
        if (full) kk = *reinterpret_cast<const ulonglong2 *>(key + a);
        else {
            kk.x = a < count ? key[a] : 0ull;
            kk.y = a + 1 < count ? key[a + 1] : 0ull;
        }
        k0[j] = kk.x >> cmp_shift;
        k1[j] = kk.y >> cmp_shift;
        c0[j] = (uint32_t)(kk.x & KISS_KEY_CTX_MASK);
        c1[j] = (uint32_t)(kk.y & KISS_KEY_CTX_MASK);
    }
    // the items either side of my wave's stretch
    const bool has_prev = wbase > 0 && wbase < count, has_next = wbase + FC1_WAVE_ITEMS < count;
    const uint64_t kprev = has_prev ? key[wbase - 1] >> cmp_shift : 0ull;
    const uint64_t knext = has_next ? key[wbase + FC1_WAVE_ITEMS] >> cmp_shift : 0ull;
    // the positions are not needed before the offsets are known: their loads overlap the flags and the look-back
    uint32_t p0[FC1_CHUNKS], p1[FC1_CHUNKS];
#pragma unroll
    for (int j = 0; j < FC1_CHUNKS; j++) {
        const uint64_t a = wbase + (uint64_t)j * 128 + 2 * lane;
        if (full) {
            const uint2 t = *reinterpret_cast<const uint2 *>(pos + a);
            p0[j] = t.x;
            p1[j] = t.y;
        } else {
            p0[j] = a < count ? pos[a] : 0u;
            p1[j] = a + 1 < count ? pos[a + 1] : 0u;
        }
    }
#ifdef FC_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FC_MARK(1); // keys, neighbours and positions have arrived
#endif
    // head(i) = item i starts a segment (differs from item i - 1; items past the end count as heads)
    uint64_t H0[FC1_CHUNKS]; // ballot of head(first item of the lane), per chunk
    uint32_t h01 = 0;        // my own head bits: bit 2j = first item, bit 2j + 1 = second item of chunk j
#pragma unroll
    for (int j = 0; j < FC1_CHUNKS; j++) {
        const uint64_t a = wbase + (uint64_t)j * 128 + 2 * lane;
        uint64_t up = __shfl_up(k1[j], 1, 64); // the second item of the lane below
        const uint64_t before = j == 0 ? kprev : lane_value_u64(k1[j > 0 ? j - 1 : 0], 63);
        if (lane == 0) up = before;
        const bool h0 = a == 0 || a >= count || up != k0[j];
        const bool h1 = a + 1 >= count || k0[j] != k1[j];
        H0[j] = __ballot(h0);
        h01 |= (h0 ? 1u : 0u) << (2 * j) | (h1 ? 1u : 0u) << (2 * j + 1);
    }
    const bool next_head = !has_next || lane_value_u64(k1[FC1_CHUNKS - 1], 63) != knext; // head(first item after my wave)
    // survivors (tied with a neighbour) and the heads among them; ranks inside the wave
    uint32_t fl = 0;                              // bits 4j .. 4j+3: surv0, surv1, headsurv0, headsurv1
    uint32_t rs[FC1_CHUNKS], gs[FC1_CHUNKS];      // survivors / surviving heads of my wave in front of my first item of chunk j
    uint32_t run_s = 0, run_h = 0;                // (wave-uniform running totals)
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < FC1_CHUNKS; j++) {
        const uint64_t a = wbase + (uint64_t)j * 128 + 2 * lane;
        const bool h0 = (h01 >> (2 * j)) & 1u, h1 = (h01 >> (2 * j + 1)) & 1u;
        const uint64_t nxt = j + 1 < FC1_CHUNKS ? H0[j + 1 < FC1_CHUNKS ? j + 1 : j] : (next_head ? 1ull : 0ull);
        const bool hn = lane < 63u ? ((H0[j] >> (lane + 1u)) & 1ull) != 0 : (nxt & 1ull) != 0; // head(item after my second)
        const bool s0 = a < count && !(h0 && h1), s1 = a + 1 < count && !(h1 && hn);
        const bool m0 = s0 && h0, m1 = s1 && h1;
        const uint64_t S0 = __ballot(s0), S1 = __ballot(s1), M0 = __ballot(m0), M1 = __ballot(m1);
        rs[j] = run_s + (uint32_t)__popcll(S0 & below) + (uint32_t)__popcll(S1 & below);
        gs[j] = run_h + (uint32_t)__popcll(M0 & below) + (uint32_t)__popcll(M1 & below);
        run_s += (uint32_t)__popcll(S0) + (uint32_t)__popcll(S1);
        run_h += (uint32_t)__popcll(M0) + (uint32_t)__popcll(M1);
        fl |= ((s0 ? 1u : 0u) | (s1 ? 2u : 0u) | (m0 ? 4u : 0u) | (m1 ? 8u : 0u)) << (4 * j);
    }
    if (lane == 0) {
        ws[wave][0] = run_s;
        ws[wave][1] = run_h;
    }
    FC_MARK(2); // flags and ranks of my wave
    __syncthreads();
    FC_MARK(3); // ... of the slowest wave
    uint32_t ts = 0, th = 0, ws0 = 0, wh0 = 0; // tile totals; totals of the waves before mine
#pragma unroll
    for (int w = 0; w < FC1_THREADS / 64; w++) {
        if (w < wave) {
            ws0 += ws[w][0];
            wh0 += ws[w][1];
        }
        ts += ws[w][0];
        th += ws[w][1];
    }
    if (wave == 0) {
        constexpr uint64_t M31 = 0x7FFFFFFFull;
        const uint64_t mine = ((uint64_t)ts << 31) | (uint64_t)th;
        uint64_t es = 0, eh = 0; // survivors / heads in all earlier tiles
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&desc[0], (2ull << 62) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&desc[tile], (1ull << 62) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t base = (int64_t)tile;
            uint32_t spins = 0;
            for (;;) {
                const int64_t t = base - 1 - (int64_t)lane;
                const uint64_t v = t >= 0 ? __hip_atomic_load(&desc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : (2ull << 62); // in front of tile 0: nothing
                const uint32_t st = (uint32_t)(v >> 62);
                const uint64_t incl = __ballot(st == 2u), ready = __ballot(st != 0u);
                const uint32_t first = incl ? (uint32_t)__builtin_ctzll(incl) : 64u;      // nearest running total
                const uint64_t need = first >= 63u ? ~0ull : ((2ull << first) - 1ull); // lanes 0 .. first
                if ((ready & need) != need) { // a predecessor this side of it has not published yet
                    if (++spins > FC1_SPIN_LIMIT) {
                        if (lane == 0) *err = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                uint64_t a = lane <= first ? (v >> 31) & M31 : 0ull, b = lane <= first ? v & M31 : 0ull;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    a += __shfl_xor(a, d, 64);
                    b += __shfl_xor(b, d, 64);
                }
                es += a;
                eh += b;
                if (first < 64u) break;
                base -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&desc[tile], (2ull << 62) | ((es + ts) << 31) | (eh + th), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl[0] = (uint32_t)es;
            s_excl[1] = (uint32_t)eh;
            if (tile + 1 == tiles) total[0] = ((es + ts) << 32) | (eh + th);
        }
    }
    FC_MARK(4); // look-back (wave 0)
    __syncthreads();
    FC_MARK(5);
    if (wbase >= count) return;
    const uint32_t bs = s_excl[0] + ws0, bh = s_excl[1] + wh0;
#pragma unroll KISS_FC1_STORE_UNROLL
    for (int j = 0; j < FC1_CHUNKS; j++) {
        const uint64_t a = wbase + (uint64_t)j * 128 + 2 * lane;
        const uint32_t f = (fl >> (4 * j)) & 15u;
        const uint32_t ni0 = bs + rs[j], ni1 = ni0 + (f & 1u);
        const uint32_t sid0 = bh + gs[j] + ((f >> 2) & 1u) - 1u;   // heads up to and including the item's own, minus one
        const uint32_t sid1 = sid0 + ((f >> 3) & 1u);
        if (f & 1u) {
            if (ni0 < cap) {
                npos[ni0] = p0[j];
                nslot[ni0] = (uint32_t)a;
                nseg[ni0] = sid0;
                if (f & 4u) nsegstart[sid0] = ni0;
                if (nctx) nctx[ni0] = c0[j]; // travels with the tied item through the first refinement round
            }
            c0[j] = 0; // tied so far: written when the item is finished, else gathered at placement
        }
        if (f & 2u) {
            if (ni1 < cap) {
                npos[ni1] = p1[j];
                nslot[ni1] = (uint32_t)(a + 1);
                nseg[ni1] = sid1;
                if (f & 8u) nsegstart[sid1] = ni1;
                if (nctx) nctx[ni1] = c1[j];
            }
            c1[j] = 0;
        }
        if (full) *reinterpret_cast<uint2 *>(octx + a) = make_uint2(c0[j], c1[j]);
        else {
            if (a < count) octx[a] = c0[j];
            if (a + 1 < count) octx[a + 1] = c1[j];
        }
    }
    FC_MARK(6); // stores issued
#ifdef FC_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FC_MARK(7); // stores done
#endif
}

// ---- tie flags as bytes (the doubling phase's first compaction over all n + 1 suffixes): 8 flags per load, and the
// suffix array is read only for the few entries that are tied ------------------------------------------------
__device__ __forceinline__ void fch_flags(const uint8_t *__restrict__ hb, uint64_t i0, uint64_t count, uint32_t &valid,
                                          uint32_t &survmask, uint32_t &headmask)
{
    valid = i0 >= count ? 0u : (count - i0 >= FC_ITEMS ? (uint32_t)FC_ITEMS : (uint32_t)(count - i0));
    survmask = headmask = 0;
    if (!valid) return;
    uint8_t h[FC_ITEMS + 1];
    if (valid == FC_ITEMS) {
        const uint64_t w = *reinterpret_cast<const uint64_t *>(hb + i0);
#pragma unroll
        for (int e = 0; e < FC_ITEMS; e++) h[e] = (uint8_t)(w >> (8 * e));
    } else {
#pragma unroll
        for (int e = 0; e < FC_ITEMS; e++) h[e] = (uint32_t)e < valid ? hb[i0 + e] : (uint8_t)1;
    }
    h[FC_ITEMS] = i0 + FC_ITEMS < count ? hb[i0 + FC_ITEMS] : (uint8_t)1;
#pragma unroll
    for (int e = 0; e < FC_ITEMS; e++) {
        if ((uint32_t)e < valid) {
            const uint64_t i = i0 + (uint64_t)e;
            const bool head = i == 0 || h[e] != 0;
            const bool nhead = i + 1 == count || h[e + 1] != 0;
            const bool surv = !(head && nhead);
            survmask |= (surv ? 1u : 0u) << e;
            headmask |= ((surv && head) ? 1u : 0u) << e;
        }
    }
}

__global__ __launch_bounds__(FC_THREADS) void k_fch_count(const uint8_t *__restrict__ hb, uint64_t count,
                                                         uint64_t *__restrict__ tcnt)
{
    __shared__ uint32_t ws[FC_THREADS / 64][2];
    const uint64_t i0 = ((uint64_t)blockIdx.x * FC_THREADS + threadIdx.x) * FC_ITEMS;
    uint32_t valid, sm, hm;
    fch_flags(hb, i0, count, valid, sm, hm);
    uint32_t ns = (uint32_t)__popc(sm), nh = (uint32_t)__popc(hm);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ns += __shfl_xor(ns, d, 64);
        nh += __shfl_xor(nh, d, 64);
    }
    if (lane_id() == 0) {
        ws[threadIdx.x >> 6][0] = ns;
        ws[threadIdx.x >> 6][1] = nh;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t a = 0, b = 0;
        for (int w = 0; w < FC_THREADS / 64; w++) {
            a += ws[w][0];
            b += ws[w][1];
        }
        tcnt[blockIdx.x] = (a << 32) | b;
    }
}

__global__ __launch_bounds__(FC_THREADS) void k_fch_compact(const uint8_t *__restrict__ hb,
                                                           const uint32_t *__restrict__ pos, uint64_t count,
                                                           const uint64_t *__restrict__ tex, uint32_t *__restrict__ npos,
                                                           uint32_t *__restrict__ nslot, uint32_t *__restrict__ nseg,
                                                           uint32_t *__restrict__ nsegstart)
{
    __shared__ uint32_t ws[FC_THREADS / 64][2];
    const int wave = threadIdx.x >> 6;
    const uint64_t i0 = ((uint64_t)blockIdx.x * FC_THREADS + threadIdx.x) * FC_ITEMS;
    uint32_t valid, sm, hm;
    fch_flags(hb, i0, count, valid, sm, hm);
    const uint32_t ns = (uint32_t)__popc(sm), nh = (uint32_t)__popc(hm);
    uint32_t is = ns, ih = nh;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t os = __shfl_up(is, d, 64), oh = __shfl_up(ih, d, 64);
        if ((int)lane_id() >= d) {
            is += os;
            ih += oh;
        }
    }
    if (lane_id() == 63) {
        ws[wave][0] = is;
        ws[wave][1] = ih;
    }
    __syncthreads();
    const uint64_t te = tex[blockIdx.x];
    uint32_t bs = (uint32_t)(te >> 32) + (is - ns), bh = (uint32_t)(te & 0xFFFFFFFFull) + (ih - nh);
    for (int w = 0; w < wave; w++) {
        bs += ws[w][0];
        bh += ws[w][1];
    }
#pragma unroll
    for (int e = 0; e < FC_ITEMS; e++) {
        if ((sm >> e) & 1u) {
            const uint32_t ni = bs++;
            if ((hm >> e) & 1u) bh++;
            const uint32_t sid = bh - 1u;
            npos[ni] = pos[i0 + e];
            nslot[ni] = (uint32_t)(i0 + e);
            nseg[ni] = sid;
            if ((hm >> e) & 1u) nsegstart[sid] = ni;
        }
    }
}

__global__ void k_fc_total(const uint64_t *__restrict__ tcnt, const uint64_t *__restrict__ tex, uint64_t tiles,
                           uint64_t *__restrict__ total)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) total[0] = tex[tiles - 1] + tcnt[tiles - 1];
}

// ---- the items of big segments after a refinement round (k_seg_finish's byte flags): count per tile, scan over tiles,
// compact -- the tile-wise form of the flag + compaction above ---------------------------------------------------------
__device__ __forceinline__ void bigb_flags(const uint8_t *__restrict__ bb, uint64_t i0, uint64_t count, uint32_t &valid,
                                           uint32_t &bigmask, uint32_t &headmask)
{
    valid = i0 >= count ? 0u : (count - i0 >= FC_ITEMS ? (uint32_t)FC_ITEMS : (uint32_t)(count - i0));
    bigmask = headmask = 0;
    if (!valid) return;
    uint64_t w = 0;
    if (valid == FC_ITEMS) w = *reinterpret_cast<const uint64_t *>(bb + i0);
    else
        for (uint32_t e = 0; e < valid; e++) w |= (uint64_t)bb[i0 + e] << (8 * e);
#pragma unroll
    for (int e = 0; e < FC_ITEMS; e++) {
        const uint32_t f = (uint32_t)(w >> (8 * e)) & 3u;
        bigmask |= (f & 1u) << e;
        headmask |= (f >> 1) << e;
    }
}

__global__ __launch_bounds__(FC_THREADS) void k_bigb_count(const uint8_t *__restrict__ bb, uint64_t count,
                                                          uint64_t *__restrict__ tcnt)
{
    __shared__ uint32_t ws[FC_THREADS / 64][2];
    const uint64_t i0 = ((uint64_t)blockIdx.x * FC_THREADS + threadIdx.x) * FC_ITEMS;
    uint32_t valid, sm, hm;
    bigb_flags(bb, i0, count, valid, sm, hm);
    uint32_t ns = (uint32_t)__popc(sm), nh = (uint32_t)__popc(hm);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ns += __shfl_xor(ns, d, 64);
        nh += __shfl_xor(nh, d, 64);
    }
    if (lane_id() == 0) {
        ws[threadIdx.x >> 6][0] = ns;
        ws[threadIdx.x >> 6][1] = nh;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t a = 0, b = 0;
        for (int w = 0; w < FC_THREADS / 64; w++) {
            a += ws[w][0];
            b += ws[w][1];
        }
        tcnt[blockIdx.x] = (a << 32) | b;
    }
}

__global__ __launch_bounds__(FC_THREADS) void k_bigb_compact(const uint8_t *__restrict__ bb, const uint64_t *__restrict__ key,
                                                            const uint32_t *__restrict__ pos,
                                                            const uint32_t *__restrict__ slot, uint64_t count,
                                                            const uint64_t *__restrict__ tex, uint64_t *__restrict__ bkey,
                                                            uint32_t *__restrict__ bpos, uint32_t *__restrict__ bseg,
                                                            uint32_t *__restrict__ bslot, uint32_t *__restrict__ bsegstart)
{
    __shared__ uint32_t ws[FC_THREADS / 64][2];
    const int wave = threadIdx.x >> 6;
    const uint64_t i0 = ((uint64_t)blockIdx.x * FC_THREADS + threadIdx.x) * FC_ITEMS;
    uint32_t valid, sm, hm;
    bigb_flags(bb, i0, count, valid, sm, hm);
    const uint32_t ns = (uint32_t)__popc(sm), nh = (uint32_t)__popc(hm);
    uint32_t is = ns, ih = nh;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t os = __shfl_up(is, d, 64), oh = __shfl_up(ih, d, 64);
        if ((int)lane_id() >= d) {
            is += os;
            ih += oh;
        }
    }
    if (lane_id() == 63) {
        ws[wave][0] = is;
        ws[wave][1] = ih;
    }
    __syncthreads();
    const uint64_t te = tex[blockIdx.x];
    uint32_t bs = (uint32_t)(te >> 32) + (is - ns), bh = (uint32_t)(te & 0xFFFFFFFFull) + (ih - nh);
    for (int w = 0; w < wave; w++) {
        bs += ws[w][0];
        bh += ws[w][1];
    }
#pragma unroll
    for (int e = 0; e < FC_ITEMS; e++) {
        if ((sm >> e) & 1u) {
            const uint32_t kx = bs++;
            if ((hm >> e) & 1u) bh++;
            const uint32_t sid = bh - 1u; // big-segment heads up to and including this item's own
            if (key) bkey[kx] = key[i0 + e]; // (null: a pivot round follows, which keys the members against its reference string)
            bpos[kx] = pos[i0 + e];
            bslot[kx] = slot[i0 + e]; // slots stay in index order: the k-th item after the sort takes the k-th slot
            bseg[kx] = sid;
            if ((hm >> e) & 1u) bsegstart[sid] = kx;
        }
    }
}

// flag + compact of `count` sorted items, in two halves so a caller can size buffers in between:
// fc_count + fc_read_total: (survivors << 32) | surviving segments;  fc_compact: the data movement
template <int SRC>
int fc_count(kiss_hip_ctx *ctx, const uint64_t *key, const uint32_t *seg, uint64_t count, int cmp_shift, int last_round,
             uint64_t *d_total)
{
    const uint64_t tiles = div_up(count, FC_TILE);
    if (2 * tiles > ctx->flags_cap) return KINTERNAL();
    uint64_t *tcnt = ctx->flags, *tex = ctx->flags + tiles;
    {
        KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
        if (SRC == FC_HEADS)
            hipLaunchKernelGGL(k_fch_count, dim3((unsigned)tiles), dim3(FC_THREADS), 0, ctx->stream,
                               reinterpret_cast<const uint8_t *>(key), count, tcnt);
        else if (SRC == FC_KEY && !last_round)
            hipLaunchKernelGGL(k_fc0_count, dim3((unsigned)tiles), dim3(FC_THREADS), 0, ctx->stream, key, count, cmp_shift, tcnt);
        else
            hipLaunchKernelGGL((k_fc_count<SRC>), dim3((unsigned)tiles), dim3(FC_THREADS), 0, ctx->stream, key, seg, count,
                               cmp_shift, last_round, tcnt);
        KCHECK(hipGetLastError());
    }
    KTRY(kiss_scan_u64(ctx, tcnt, tex, tiles));
    hipLaunchKernelGGL(k_fc_total, dim3(1), dim3(64), 0, ctx->stream, tcnt, tex, tiles, d_total);
    KCHECK(hipGetLastError());
    return KISS_HIP_OK;
}

int fc_read_total(kiss_hip_ctx *ctx, const uint64_t *d_total, uint64_t *tot)
{
    KTRY(kiss_readback(ctx, d_total, 2));
    std::memcpy(tot, ctx->h_pinned, sizeof(uint64_t));
    return KISS_HIP_OK;
}

template <int SRC, bool HAS_SLOT>
int fc_compact(kiss_hip_ctx *ctx, const uint64_t *key, const uint32_t *seg, const uint32_t *pos, const uint32_t *slot,
               uint64_t count, int cmp_shift, int last_round, uint32_t *npos, uint32_t *nslot, uint32_t *nseg,
               uint32_t *nsegstart, uint32_t *out, uint32_t *isa, uint32_t *octx = nullptr, uint32_t *nctx = nullptr,
               bool *nctx_written = nullptr, uint32_t *tmark = nullptr, const uint64_t *payload = nullptr, int isa_shift = 0,
               uint32_t *cfix = nullptr)
{
    uint8_t *hmark = tmark ? ctx->hfar : nullptr; // (the calls that mark taints are the retirements of the LMS sort)
    if (nctx_written) *nctx_written = false;
    const uint64_t tiles = div_up(count, FC_TILE);
    const uint64_t *tex = ctx->flags + tiles; // left there by fc_count
    KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
    if (SRC == FC_HEADS && !HAS_SLOT && !out && !isa && !octx)
        hipLaunchKernelGGL(k_fch_compact, dim3((unsigned)tiles), dim3(FC_THREADS), 0, ctx->stream,
                           reinterpret_cast<const uint8_t *>(key), pos, count, tex, npos, nslot, nseg, nsegstart);
    else if (SRC == FC_KEY && !HAS_SLOT && !last_round && out && octx && !isa && cmp_shift >= 24)
    {
        hipLaunchKernelGGL(k_fc0_compact, dim3((unsigned)tiles), dim3(FC_THREADS), 0, ctx->stream, key, pos, count, cmp_shift, tex,
                           npos, nslot, nseg, nsegstart, out == pos ? (uint32_t *)nullptr : out, octx, nctx);
        if (nctx_written) *nctx_written = nctx != nullptr;
    }
    else // (without slots an item's slot is its index: out == pos means the list is in place already)
        hipLaunchKernelGGL((k_fc_compact<SRC, HAS_SLOT>), dim3((unsigned)tiles), dim3(FC_THREADS), 0, ctx->stream, key, seg,
                           pos, slot, count, cmp_shift, last_round, tex, npos, nslot, nseg, nsegstart,
                           (!HAS_SLOT && out == pos) ? (uint32_t *)nullptr : out, isa, octx, tmark, payload, isa_shift,
                           cfix ? ctx->pk : (const uint64_t *)nullptr, cfix, hmark);
    KCHECK(hipGetLastError());
    return KISS_HIP_OK;
}

template <bool HAS_SEG, bool HAS_SLOT>
int fused_compact(kiss_hip_ctx *ctx, const uint64_t *key, const uint32_t *seg, const uint32_t *pos, const uint32_t *slot,
                  uint64_t count, int cmp_shift, int last_round, uint32_t *npos, uint32_t *nslot, uint32_t *nseg,
                  uint32_t *nsegstart, uint64_t *d_total, uint64_t *tot)
{
    constexpr int SRC = HAS_SEG ? FC_KEY_SEG : FC_KEY;
    KTRY((fc_count<SRC>(ctx, key, seg, count, cmp_shift, last_round, d_total)));
    KTRY((fc_compact<SRC, HAS_SLOT>(ctx, key, seg, pos, slot, count, cmp_shift, last_round, npos, nslot, nseg, nsegstart,
                                    ctx->lms_sorted_far, nullptr, nullptr, nullptr, nullptr, ctx->lms_ctx_far)));
    return fc_read_total(ctx, d_total, tot);
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
    KTRY(kiss_readback(ctx, dptr, 2));
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
        KTRY(kiss_zero_u32(ctx, ctx->lms_ctx_far, 1));
        return KISS_HIP_OK;
    }
    uint32_t *d_nbig = ctx->d_small + 8;
    uint64_t *d_total = (uint64_t *)(ctx->d_small + 2);
    const unsigned T = LS_THREADS;
    const bool dbg = ctx->opts.debug;
    const bool pivot_on = !ctx->opts.no_pivot_rounds; // (hooks build: 32-base rounds only)
    // from the first refinement round on; pivot_from_round2 keeps the first one in the 32-base form (hooks build)
    const bool pivot_r1 = !ctx->opts.pivot_from_round2;
    static const int pivot_den[3] = {4, 7, 11};
    unsigned pivot_rounds_done = 0;
    // deviations recorded per member and round: 3 (the hooks build can sweep 1 .. 8)
    const int pivot_slots = ctx->opts.pivot_slots >= 1 && ctx->opts.pivot_slots <= 8 ? ctx->opts.pivot_slots : 3;
    const bool no_pair_keys = !ctx->opts.pair_keys;
    const uint32_t small_seg = ctx->opts.small_seg >= 2 && ctx->opts.small_seg <= 4096 ? ctx->opts.small_seg : LMS_SMALL_SEG;

    // ------------------------------ round 0 ------------------------------------------------
    uint64_t count = m_far;
    RadixBufs rb;
    rb.first_pos = ctx->lms_pos; // the ascending list stays intact: the first pass reads it, nothing writes it
    rb.key[0] = ctx->keyA;
    rb.key[1] = ctx->keyB;
    rb.pos[0] = ctx->posA;
    // five passes: the result lands in buffer 1.  The sorted positions ARE the k-ordered list for every suffix that round 0
    // makes unique, so the last pass writes them where they belong and the compaction below has nothing to copy.
    static_assert(KISS_R0_PASSES % 2 == 1, "round-0 result must land in buffer 1");
    rb.pos[1] = ctx->lms_sorted_far;
    rb.seg[0] = rb.seg[1] = nullptr;
    // depth >= 125 whenever bounded, so the first 20 bases always count in full
    const int r0_shift = 64 - 2 * ROUND0_BASES;
    static_assert(64 - 2 * ROUND0_BASES == KISS_R0_SHIFT && (2 * ROUND0_BASES + 7) / 8 == KISS_R0_PASSES, "classify.hip counts these digits");
    int res = 0;
    KTRY(kiss_radix_sort(ctx, rb, count, r0_shift, 0, &res));
    ctx->stats.lms_rounds++;
    ctx->stats.sort_item_rounds += count;
    uint32_t *Pc = rb.pos[res ^ 1]; // receives the survivors' positions
    uint64_t tot;
    bool have_tctx = false; // bslot is free until the big-segment path of the first round: the tied items' context words
    const bool no_onepass = ctx->opts.no_fc0_onepass;
    const uint64_t tiles1 = div_up(count, FC1_TILE);
    // (the result of the five passes is in buffer 1 = the output list itself, so there are no positions to copy)
    const bool onepass = !no_onepass && ctx->fc_desc && tiles1 >= 8 && tiles1 + 1 <= ctx->fc_desc_cap &&
                         rb.pos[res] == ctx->lms_sorted_far;
    if (onepass) {
        uint64_t *desc = ctx->fc_desc;
        uint32_t *ticket = reinterpret_cast<uint32_t *>(desc + tiles1);
        for (int attempt = 0; attempt < 2; attempt++) {
            KTRY(kiss_zero_u32(ctx, desc, 2 * tiles1 + 2));
            {
                KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
                hipLaunchKernelGGL(k_fc0_onepass, dim3((unsigned)tiles1), dim3(FC1_THREADS), 0, ctx->stream, rb.key[res], rb.pos[res],
                                   count, r0_shift, tiles1, desc, ticket, ctx->rx_ctl + 1, Pc, ctx->slotA, ctx->segA, ctx->segstartA,
                                   ctx->lms_ctx_far, ctx->bslot, ctx->t_cap, d_total);
                KCHECK(hipGetLastError());
            }
            KTRY(fc_read_total(ctx, d_total, &tot));
#ifdef FC_PROF
            {
                unsigned long long h[12], z[12] = {0};
                KCHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(fc_prof), sizeof h));
                KCHECK(hipMemcpyToSymbol(HIP_SYMBOL(fc_prof), z, sizeof z));
                unsigned long long sum = 0;
                for (int i = 0; i < 8; i++) sum += h[i];
                fprintf(stderr, "[fc_prof] %llu tiles, %.2f us per tile (100 MHz clock):", (unsigned long long)tiles1, (double)sum / 100.0 / (double)tiles1);
                for (int i = 0; i < 8; i++) fprintf(stderr, " %d:%.1f%%", i, sum ? 100.0 * (double)h[i] / (double)sum : 0.0);
                fprintf(stderr, "\n");
            }
#endif
            if ((tot >> 32) <= ctx->t_cap) break;
            if (attempt) return KINTERNAL();
            // more tied suffixes than the tied-segment arrays hold (their stores were dropped): regrow, once more
            const uint64_t surv = tot >> 32;
            KTRY(kiss_tied_reserve(ctx, surv + surv / 8 + 1024));
        }
        have_tctx = true;
    } else {
        KTRY((fc_count<FC_KEY>(ctx, rb.key[res], nullptr, count, r0_shift, 0, d_total)));
        KTRY(fc_read_total(ctx, d_total, &tot));
        if ((tot >> 32) > ctx->t_cap) { // more tied suffixes than the tied-segment arrays hold: regrow them (empty so far)
            const uint64_t surv = tot >> 32;
            KTRY(kiss_tied_reserve(ctx, surv + surv / 8 + 1024));
            KTRY((fc_count<FC_KEY>(ctx, rb.key[res], nullptr, count, r0_shift, 0, d_total)));
        }
    }
    uint32_t *Sc = ctx->slotA, *Gc = ctx->segA, *SSc = ctx->segstartA;
    uint64_t *F1 = ctx->flags;
    uint64_t *F2 = ctx->flags + ctx->t_cap;
    if (!onepass)
        KTRY((fc_compact<FC_KEY, false>(ctx, rb.key[res], nullptr, rb.pos[res], nullptr, count, r0_shift, 0, Pc, Sc, Gc, SSc,
                                       ctx->lms_sorted_far, nullptr, ctx->lms_ctx_far, ctx->bslot, &have_tctx)));
    count = tot >> 32;
    uint64_t nseg = tot & 0xFFFFFFFFull;
    if (dbg)
        fprintf(stderr, "[kiss_hip] round 0: items %llu -> survivors %llu in %llu segments\n",
                (unsigned long long)m_far, (unsigned long long)count, (unsigned long long)nseg);
    uint64_t *K1 = ctx->keyA;

    // ------------------------------ rounds >= 1 ---------------------------------------------
    uint64_t off = ROUND0_BASES;
    bool first_refine = true; // the tied items' context words (in bslot) are only good for the first pass over them
    for (;; first_refine = false) {
        if (count == 0) break;
        if (depth && off >= depth) return KINTERNAL(); // the last round retires everything
        uint64_t rem = depth ? depth - off : 32;
        if (rem > 32) rem = 32;
        const bool last_round = depth ? (off + 32 >= depth) : false;
        const uint64_t mask = rem >= 32 ? ~0ull : (~0ull << (64 - 2 * rem));
        const int key_lo_bit = (int)(64 - 2 * rem);
        const unsigned grid = (unsigned)div_up(count, T);
        // big segments need the round's key only where they are radix sorted on it (no pivot round ahead)
        const bool pivot_ahead = depth && (off > ROUND0_BASES || pivot_r1) && pivot_on;
        {
            KTimer t(ctx, KISS_HIP_K_KEYGATHER, count);
            // (segstart's end entry is needed by the pair test: set it first)
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(64), 0, ctx->stream, SSc + nseg, (uint32_t)count, (uint32_t *)nullptr);
            hipLaunchKernelGGL(k_gather_keys, dim3(grid), dim3(T), 0, ctx->stream, ctx->pk, n, Pc, count, off, mask,
                               K1, no_pair_keys ? Gc : (const uint32_t *)nullptr, SSc, 3u, pivot_ahead ? small_seg : 0xFFFFFFFFu);
            KCHECK(hipGetLastError());
        }
        uint8_t *bigb = nullptr;
        hipEvent_t dbg_e0 = nullptr, dbg_e1 = nullptr;
        if (dbg) {
            (void)hipEventCreate(&dbg_e0);
            (void)hipEventCreate(&dbg_e1);
            (void)hipEventRecord(dbg_e0, ctx->stream);
        }
        {
            KTimer t(ctx, KISS_HIP_K_SEGRANK, count);
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(64), 0, ctx->stream, SSc + nseg, (uint32_t)count, d_nbig);
            uint8_t *inorder = reinterpret_cast<uint8_t *>(F2); // F2 is free until the pivot / big-segment steps below
            bigb = inorder + ((count + 15) & ~15ull);           // the big-segment flags of k_seg_finish, behind them
            hipLaunchKernelGGL(k_seg_adjacent, dim3(grid), dim3(T), 0, ctx->stream, ctx->pk, n, K1, Pc, Gc, SSc, count, off,
                               depth, small_seg, inorder);
            hipLaunchKernelGGL(k_seg_finish, dim3(grid), dim3(T), 0, ctx->stream, ctx->pk, n, K1, Pc, Sc, Gc, SSc,
                               count, off, depth, small_seg, inorder, ctx->lms_sorted_far, bigb, d_nbig,
                               (have_tctx && first_refine) ? ctx->bslot : (const uint32_t *)nullptr, ctx->lms_ctx_far, ctx->hfar);
            KCHECK(hipGetLastError());
        }
        ctx->stats.lms_rounds++;
        ctx->stats.sort_item_rounds += count;
        // big-segment items and segments: counted tile by tile from the byte flags, scanned over the tiles
        const uint64_t btiles = div_up(count, FC_TILE);
        uint64_t *const btcnt = F1, *const btex = F1 + btiles;
        if (2 * btiles > ctx->t_cap) return KINTERNAL();
        {
            KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
            hipLaunchKernelGGL(k_bigb_count, dim3((unsigned)btiles), dim3(FC_THREADS), 0, ctx->stream, bigb, count, btcnt);
            KCHECK(hipGetLastError());
        }
        KTRY(kiss_scan_u64(ctx, btcnt, btex, btiles));
        hipLaunchKernelGGL(k_fc_total, dim3(1), dim3(64), 0, ctx->stream, btcnt, btex, btiles, d_total);
        uint64_t bt;
        KTRY(read_u64(ctx, d_total, &bt));
        const uint64_t nbig = bt >> 32;
        if (dbg) {
            float ms = 0;
            (void)hipEventRecord(dbg_e1, ctx->stream);
            (void)hipEventSynchronize(dbg_e1);
            (void)hipEventElapsedTime(&ms, dbg_e0, dbg_e1);
            fprintf(stderr, "[kiss_hip]   seg_finish + flag scan %.3f ms\n", ms);
            (void)hipEventDestroy(dbg_e0);
            (void)hipEventDestroy(dbg_e1);
            fprintf(stderr, "[kiss_hip] round off=%llu: items %llu in %llu segments, big-segment items %llu\n",
                    (unsigned long long)off, (unsigned long long)count, (unsigned long long)nseg,
                    (unsigned long long)nbig);
        }
        if (nbig == 0) break;
        // ---- big segments: compact, three-way split around a pivot key, radix sort of the < and > groups on
        //      (group, key), then split where neighbours differ; survivors go round again
        const uint64_t nbigseg = bt & 0xFFFFFFFFull;
        uint32_t *bss = ctx->segstartB; // start of every big segment in the compact arrays (+ end entry)
        const unsigned bgrid = (unsigned)div_up(nbig, T);
        {
            KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
            hipLaunchKernelGGL(k_bigb_compact, dim3((unsigned)btiles), dim3(FC_THREADS), 0, ctx->stream, bigb,
                               pivot_ahead ? (const uint64_t *)nullptr : K1, Pc, Sc, count, btex, ctx->bkeyA, ctx->bposA, ctx->bsegA,
                               ctx->bslot, bss);
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(64), 0, ctx->stream, bss + nbigseg, (uint32_t)nbig, (uint32_t *)nullptr);
            KCHECK(hipGetLastError());
        }
        uint64_t *d_ptot = (uint64_t *)(ctx->d_small + 24); // grand total of the class counts (segment ends at nbig)
        uint64_t *NE = ctx->flags + 2 * ctx->t_cap - nbig;  // flags of the < and > members, in the tail of `flags`
        if (2 * nbig > ctx->t_cap) NE = nullptr;            // it would overlap the scan words
        // the first refinement round still holds the diverged copies of interspersed repeats (78 % of the items differ
        // from their pivot at chm13 size): the split only pays from the second round on, when what is left is the
        // long tandem arrays
        if (off == ROUND0_BASES) NE = nullptr;
        // ---- pivot round (bounded depth): see k_pivot_lcp
        if (pivot_ahead) {
            const int dbits = bits_for(depth - off);          // d < depth - off
            int slots = 63 / (dbits + 4);                     // deviations per key: [side 2 | d | base 2] each + 1 bit
            if (slots > pivot_slots) slots = pivot_slots;
            const int kbits = slots * (dbits + 4) + 1;
            const int key_lo = 64 - 8 * ((kbits + 7) / 8);
            // room for the members' context words below the sorted bits (k = 256: 40 key bits, 24 bits free)
            const bool with_ctx = key_lo >= 24 && !ctx->opts.no_pivot_ctx;
            {
                KTimer t(ctx, KISS_HIP_K_SEGRANK, nbig);
                hipLaunchKernelGGL(k_pivot_lcp, dim3(bgrid), dim3(T), 0, ctx->stream, ctx->pk, ctx->bposA, ctx->bsegA, bss, nbig,
                                   off, depth, dbits, slots, ctx->bkeyB, pivot_den[pivot_rounds_done % 3], with_ctx ? 1 : 0);
                pivot_rounds_done++;
                KCHECK(hipGetLastError());
            }
#ifdef KISS_HIP_HOOKS
            if (const char *dump = ctx->opts.dump_pivot[0] ? ctx->opts.dump_pivot : nullptr) { // debugging aid: the inputs and keys of this pivot round
                std::vector<uint32_t> hp(nbig), hs(nbig), hss(nbigseg + 1);
                std::vector<uint64_t> hk(nbig);
                KCHECK(hipStreamSynchronize(ctx->stream));
                KCHECK(hipMemcpy(hp.data(), ctx->bposA, nbig * 4, hipMemcpyDeviceToHost));
                KCHECK(hipMemcpy(hs.data(), ctx->bsegA, nbig * 4, hipMemcpyDeviceToHost));
                KCHECK(hipMemcpy(hss.data(), bss, (nbigseg + 1) * 4, hipMemcpyDeviceToHost));
                KCHECK(hipMemcpy(hk.data(), ctx->bkeyB, nbig * 8, hipMemcpyDeviceToHost));
                char name[512];
                snprintf(name, sizeof name, "%s.%u", dump, ctx->stats.lms_rounds);
                if (FILE *f = fopen(name, "wb")) {
                    const uint64_t hdr[6] = {nbig, nbigseg, off, depth, (uint64_t)dbits, (uint64_t)slots};
                    fwrite(hdr, 8, 6, f);
                    fwrite(hp.data(), 4, nbig, f);
                    fwrite(hs.data(), 4, nbig, f);
                    fwrite(hss.data(), 4, nbigseg + 1, f);
                    fwrite(hk.data(), 8, nbig, f);
                    fclose(f);
                }
            }
#endif
            RadixBufs pb;
            pb.key[0] = ctx->bkeyB;
            pb.key[1] = ctx->keyB; // free since the round-0 compaction (K1 is keyA)
            pb.pos[0] = ctx->bposA;
            pb.pos[1] = ctx->bposB;
            pb.seg[0] = ctx->bsegA;
            pb.seg[1] = ctx->bsegB;
            int pres = 0;
            const int sbits = bits_for(nbigseg);
            KTRY(kiss_radix_sort(ctx, pb, nbig, key_lo, sbits, &pres));
            // a single segment has no segment digits: the sort then leaves the segment column alone (all the same id)
            const uint32_t *sseg = sbits ? pb.seg[pres] : ctx->bsegA;
            uint8_t *heads = reinterpret_cast<uint8_t *>(ctx->segB); // scratch of the 32-base form, free here
            {
                KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, nbig);
                hipLaunchKernelGGL(k_pivot_heads, dim3(bgrid), dim3(T), 0, ctx->stream, pb.key[pres], sseg, nbig, 64 - kbits,
                                   heads, ~0ull << (64 - kbits));
                KCHECK(hipGetLastError());
            }
            KTRY((fc_count<FC_HEADS>(ctx, reinterpret_cast<const uint64_t *>(heads), nullptr, nbig, 0, 0, d_total)));
            KTRY(fc_read_total(ctx, d_total, &tot));
            KTRY((fc_compact<FC_HEADS, true>(ctx, reinterpret_cast<const uint64_t *>(heads), nullptr, pb.pos[pres], ctx->bslot,
                                            nbig, 0, 0, Pc, Sc, Gc, SSc, ctx->lms_sorted_far, nullptr, nullptr, nullptr, nullptr,
                                            ctx->lms_ctx_far, with_ctx ? pb.key[pres] : (const uint64_t *)nullptr)));
            ctx->stats.big_item_rounds += nbig;
            count = tot >> 32;
            nseg = tot & 0xFFFFFFFFull;
            if (dbg)
                fprintf(stderr, "[kiss_hip]   pivot round: %llu big-segment items in %llu segments -> %llu still tied in %llu "
                                "segments (same depth)\n", (unsigned long long)nbig, (unsigned long long)nbigseg,
                        (unsigned long long)count, (unsigned long long)nseg);
            continue; // survivors are tied through off + d + 1 > off bases: the next round starts from the same depth
        }
        RadixBufs bb;
        uint64_t *H1, *H2;
        if (NE) {
            {
                KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, nbig);
                hipLaunchKernelGGL(k_pivot_class, dim3(bgrid), dim3(T), 0, ctx->stream, ctx->bkeyA, ctx->bsegA, bss, nbig, F1);
                KCHECK(hipGetLastError());
            }
            KTRY(kiss_scan_u64(ctx, F1, F2, nbig));
            KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, nbig);
            hipLaunchKernelGGL(k_last_total, dim3(1), dim3(64), 0, ctx->stream, F1, F2, nbig, d_ptot);
            hipLaunchKernelGGL(k_pivot_scatter, dim3(bgrid), dim3(T), 0, ctx->stream, ctx->bkeyA, ctx->bposA, ctx->bsegA, bss,
                               nbig, F1, F2, d_ptot, ctx->bkeyB, ctx->bposB, NE);
            KCHECK(hipGetLastError());
            H1 = NE;
            H2 = F1; // the class words are dead now
        }
        const uint64_t *skey = ctx->bkeyB;
        const uint32_t *spos = ctx->bposB, *sseg = ctx->bsegA;
        if (NE) {
            KTRY(kiss_scan_u64(ctx, H1, H2, nbig));
            hipLaunchKernelGGL(k_last_total, dim3(1), dim3(64), 0, ctx->stream, H1, H2, nbig, d_total);
            uint64_t nt;
            KTRY(read_u64(ctx, d_total, &nt));
            const uint64_t nne = nt >> 32, ngroups = nt & 0xFFFFFFFFull;
            if (dbg)
                fprintf(stderr, "[kiss_hip]   big segments %llu: %llu of %llu items differ from their pivot key (%llu groups)\n",
                        (unsigned long long)nbigseg, (unsigned long long)nne, (unsigned long long)nbig,
                        (unsigned long long)ngroups);
            if (nne) {
                bb.key[0] = ctx->keyB;  // free since the round-0 compaction (K1 is keyA)
                bb.key[1] = ctx->bkeyA; // free since the partition
                bb.pos[0] = ctx->lmsP;  // free until the placement step
                bb.pos[1] = ctx->lmsC;
                bb.seg[0] = ctx->bsegB;
                bb.seg[1] = ctx->segB;
                uint32_t *bidx = ctx->bposA;
                {
                    KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, nbig);
                    hipLaunchKernelGGL(k_big_extract, dim3(bgrid), dim3(T), 0, ctx->stream, ctx->bkeyB, ctx->bposB, nbig, H1, H2,
                                       bb.key[0], bb.pos[0], bb.seg[0], bidx);
                    KCHECK(hipGetLastError());
                }
                int bres = 0;
                KTRY(kiss_radix_sort(ctx, bb, nne, key_lo_bit, bits_for(ngroups), &bres));
                KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, nne);
                hipLaunchKernelGGL(k_big_writeback, dim3((unsigned)div_up(nne, T)), dim3(T), 0, ctx->stream, bb.key[bres],
                                   bb.pos[bres], bidx, nne, ctx->bkeyB, ctx->bposB);
                KCHECK(hipGetLastError());
            }
            ctx->stats.big_item_rounds += nne;
        } else { // no room for the flag words (the tied-segment arrays are nearly full): the plain radix sort
            bb.key[0] = ctx->bkeyA;
            bb.key[1] = ctx->bkeyB;
            bb.pos[0] = ctx->bposA;
            bb.pos[1] = ctx->bposB;
            bb.seg[0] = ctx->bsegA;
            bb.seg[1] = ctx->bsegB;
            int bres = 0;
            KTRY(kiss_radix_sort(ctx, bb, nbig, key_lo_bit, bits_for(nbigseg), &bres));
            ctx->stats.big_item_rounds += nbig;
            skey = bb.key[bres];
            spos = bb.pos[bres];
            sseg = bits_for(nbigseg) ? bb.seg[bres] : ctx->bsegA; // one segment: no segment digits, the column did not move
        }
        KTRY((fused_compact<true, true>(ctx, skey, sseg, spos, ctx->bslot, nbig, 0, (int)last_round, Pc, Sc, Gc, SSc, d_total,
                                       &tot)));
        count = tot >> 32;
        nseg = tot & 0xFFFFFFFFull;
        off += 32;
        if (!depth && off > n + 64 && count > 0) return KINTERNAL(); // exact mode must have terminated
        // exact order, repeats longer than this: 32 bases per round is the wrong tool, the caller switches to
        // rank doubling (same result)
        if (!depth && off >= KISS_EXACT_MSD_MAX_DEPTH && count > 0) return KISS_INTERNAL_TOO_DEEP;
    }
    return KISS_HIP_OK;
}

// =============================================================================================================
// Exact order by prefix doubling (the KISS2 / PREFIX_DOUBLING path, reference kiss2_core.hpp:728-797: sort,
// re-rank, compact, double).  The reference doubles over an encoded LMS string; here the doubling runs over
// the full suffix array, which 288 GB of HBM affords and which needs no sampled-rank bookkeeping:
//   1. the k-ordered pipeline has produced SA ordered by the first h0 bases (ties in arbitrary order);
//   2. group heads: SA[i] starts a group unless it shares h0 bases with SA[i-1]          (k_group_heads)
//   3. ISA[SA[i]] = i, then for tied suffixes the slot of their group head                 (k_isa_init/update)
//   4. rounds h = h0, 2*h0, ...: every tied suffix p fetches ISA[p + h], groups are radix sorted on
//      (group, rank), split where neighbours differ; singletons retire to SA, the rest get new ranks.
// Only the *result* (the unique suffix array) is comparable with the reference; the number of rounds is
// O(log(longest repeat / h0)) instead of the MSD path's O(longest repeat / 32).
// =============================================================================================================
namespace {

// cw (optional): the context words the induction left parallel to SA.  Only their taint bit is looked at: a suffix that
// does not descend from an LMS suffix the bounded-depth sort may have misplaced is where it belongs and starts a group of
// its own -- no text is read for it (at chm13 size 94 % of the 3.1 G entries; this kernel was 69 ms of random reads).
__global__ __launch_bounds__(LS_THREADS) void k_group_heads(const uint64_t *__restrict__ pk, uint64_t n,
                                                           const uint32_t *__restrict__ SA, uint64_t count, uint32_t h0,
                                                           const uint32_t *__restrict__ cw, uint8_t *__restrict__ heads,
                                                           uint32_t lo) // first entry that can be tied (SA: 1, SA[0] = n)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    const bool valid = i < count;
    const uint64_t p = valid ? SA[i] : 0;
    const bool tainted = valid && i >= lo && (!cw || (cw[i] & KISS_CTX_TAINT) != 0); // SA[0] = n has no context word
    const bool pfull = tainted && p + h0 <= n;
    const uint64_t kp = pfull ? kiss_key32(pk, p) : 0ull; // h0 >= 32
    // predecessor's first word: from the neighbouring lane, lane 0 loads it
    uint64_t q = __shfl_up(p, 1, 64);
    uint64_t kq = __shfl_up(kp, 1, 64);
    bool qfull = __shfl_up((int)pfull, 1, 64) != 0;
    if (lane_id() == 0 && pfull && i > lo) {
        q = SA[i - 1];
        qfull = q + h0 <= n && (!cw || (cw[i - 1] & KISS_CTX_TAINT) != 0);
        kq = qfull ? kiss_key32(pk, q) : 0ull;
    } else if (lane_id() == 0) {
        qfull = false;
    }
    if (!valid) return;
    uint8_t head = 1;
    if (i > 0 && pfull && qfull && kp == kq) {
        head = 0;
        uint32_t d = 32;
        // 128 bases per step while that much is left (five independent word loads per suffix and step: tainted
        // neighbours mostly ARE tied, so the walk usually runs the whole h0 bases)
        for (; d + 128 <= h0 && !head; d += 128) {
            uint64_t a[4], b[4];
            keys128(pk, p + d, a);
            keys128(pk, q + d, b);
            if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3]) head = 1;
        }
        if (d < h0 && !head && h0 >= 160) { // the rest as ONE step that ends at h0 (it overlaps what has been compared)
            uint64_t a[4], b[4];
            keys128(pk, p + h0 - 128, a);
            keys128(pk, q + h0 - 128, b);
            if (a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3]) head = 1;
            d = h0;
        }
        for (; d < h0 && !head; d += 32) {
            uint64_t a = kiss_key32(pk, p + d), b = kiss_key32(pk, q + d);
            if (h0 - d < 32) {
                const uint64_t mask = ~0ull << (64 - 2 * (h0 - d));
                a &= mask;
                b &= mask;
            }
            if (a != b) head = 1;
        }
    }
    heads[i] = head;
}

// ---- group heads in two steps (the form used when the context words with their taint bits are at hand) ---------------
// Only 5 % of the suffix-array entries of a genome are tainted; walked inside the one-lane-per-entry kernel above they
// leave 61 of 64 lanes of nearly every wave waiting for three dependent random reads (29 ms at chm13 size, all latency).
// Step 1 streams SA and the context words once: heads[i] = 1 for every entry that can not be tied with its
// predecessor, and the indexes of the candidates (both tainted, both with h0 bases left) are appended to a list, one
// atomic per workgroup -- spread over GH_REGIONS counters, each with its own slice of the list: three million adds to ONE
// address take 36 ms, which is more than the kernel they were meant to speed up.  Step 2 compares the candidates' h0
// bases with every lane busy.
constexpr int GH_THREADS = 1024;
constexpr int GH_ITEMS = 4;
constexpr int GH_REGIONS = 256;
__global__ __launch_bounds__(GH_THREADS) void k_heads_candidates(const uint32_t *__restrict__ SA, uint64_t count, uint64_t n,
                                                                uint32_t h0, const uint32_t *__restrict__ cw,
                                                                uint8_t *__restrict__ heads, uint32_t *__restrict__ list,
                                                                uint64_t region_cap, uint32_t *__restrict__ ncand,
                                                                uint32_t lo) // first entry that can be tied (SA: 1)
{
    __shared__ uint32_t wcount[GH_THREADS / 64];
    __shared__ uint32_t s_base;
    const uint32_t region = blockIdx.x % GH_REGIONS;
    const uint64_t i0 = ((uint64_t)blockIdx.x * GH_THREADS + threadIdx.x) * GH_ITEMS;
    uint32_t cm = 0; // bit e: entry i0 + e is a candidate
    if (i0 + GH_ITEMS <= count) { // four context words per load, four flag bytes per store
        const uint4 c4 = *reinterpret_cast<const uint4 *>(cw + i0);
        const uint32_t prev = i0 > 0 ? cw[i0 - 1] : 0u;
        const uint32_t t = (c4.x >> 31) | ((c4.y >> 31) << 1) | ((c4.z >> 31) << 2) | ((c4.w >> 31) << 3);
        cm = t & ((t << 1) | (prev >> 31)) & 0xFu;
#pragma unroll
        for (int e = 0; e < GH_ITEMS; e++)
            if (i0 + (uint64_t)e <= lo) cm &= ~(1u << e);
        if (cm) {
#pragma unroll
            for (int e = 0; e < GH_ITEMS; e++)
                if ((cm >> e) & 1u) {
                    const uint64_t p = SA[i0 + e], q = SA[i0 + e - 1];
                    if (!(p + h0 <= n && q + h0 <= n)) cm &= ~(1u << e);
                }
        }
        *reinterpret_cast<uint32_t *>(heads + i0) = 0x01010101u; // candidates that turn out tied are reset by k_heads_compare
    } else {
        for (int e = 0; e < GH_ITEMS; e++) {
            const uint64_t i = i0 + (uint64_t)e;
            if (i >= count) break;
            if (i > lo && (cw[i] & KISS_CTX_TAINT) && (cw[i - 1] & KISS_CTX_TAINT)) {
                const uint64_t p = SA[i], q = SA[i - 1];
                if (p + h0 <= n && q + h0 <= n) cm |= 1u << e;
            }
            heads[i] = 1;
        }
    }
    const uint32_t c = (uint32_t)__popc(cm);
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if ((int)lane_id() >= d) inc += o;
    }
    const int wave = threadIdx.x >> 6;
    if (lane_id() == 63) wcount[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < GH_THREADS / 64; w++) {
            const uint32_t cc = wcount[w];
            wcount[w] = t;
            t += cc;
        }
        s_base = t ? atomicAdd(&ncand[region], t) : 0u;
    }
    __syncthreads();
    if (cm) {
        uint64_t at = (uint64_t)s_base + wcount[wave] + (inc - c);
#pragma unroll
        for (int e = 0; e < GH_ITEMS; e++)
            if ((cm >> e) & 1u) {
                if (at < region_cap) list[(uint64_t)region * region_cap + at] = (uint32_t)(i0 + e); // (an overflow shows in the counter)
                at++;
            }
    }
}

__global__ __launch_bounds__(LS_THREADS) void k_heads_compare(const uint64_t *__restrict__ pk, const uint32_t *__restrict__ SA,
                                                             const uint32_t *__restrict__ list, uint64_t region_cap,
                                                             const uint32_t *__restrict__ ncand, uint32_t h0,
                                                             uint8_t *__restrict__ heads)
{
    const uint64_t j = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x; // blockIdx.y = region
    if (j >= ncand[blockIdx.y]) return;
    const uint32_t i = list[(uint64_t)blockIdx.y * region_cap + j];
    const uint64_t p = SA[i], q = SA[i - 1];
    bool differ = false;
    uint32_t d = 0;
    for (; d + 128 <= h0 && !differ; d += 128) {
        uint64_t a[4], b[4];
        keys128(pk, p + d, a);
        keys128(pk, q + d, b);
        differ = a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || a[3] != b[3];
    }
    for (; d < h0 && !differ; d += 32) {
        uint64_t a = kiss_key32(pk, p + d), b = kiss_key32(pk, q + d);
        if (h0 - d < 32) {
            const uint64_t mask = ~0ull << (64 - 2 * (h0 - d));
            a &= mask;
            b &= mask;
        }
        differ = a != b;
    }
    if (!differ) heads[i] = 0;
}

// debug: number of tainted context words
__global__ __launch_bounds__(LS_THREADS) void k_count_taint(const uint32_t *__restrict__ cw, uint64_t count,
                                                           unsigned long long *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    const uint64_t m = __ballot(i > 0 && i < count && (cw[i] & KISS_CTX_TAINT));
    if (lane_id() == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

__global__ __launch_bounds__(LS_THREADS) void k_isa_init(const uint32_t *__restrict__ SA, uint64_t count,
                                                        uint32_t *__restrict__ isa)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i < count) isa[SA[i]] = (uint32_t)i;
}

// rank of a tied suffix = slot of the first member of its group
// (also reports the longest group: *maxlen, zeroed by the caller; segstart[nseg] = count)
__global__ __launch_bounds__(LS_THREADS) void k_isa_update(const uint32_t *__restrict__ pos,
                                                          const uint32_t *__restrict__ slot,
                                                          const uint32_t *__restrict__ seg,
                                                          const uint32_t *__restrict__ segstart, uint64_t count,
                                                          uint32_t *__restrict__ isa, uint32_t *__restrict__ maxlen,
                                                          int shift) // isa is indexed by position >> shift
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    uint32_t len = 0;
    if (i < count) {
        const uint32_t sg = seg[i];
        const uint32_t a = segstart[sg];
        isa[pos[i] >> shift] = slot[a];
        if ((uint32_t)i == a) len = segstart[sg + 1] - a;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(len, d, 64);
        len = o > len ? o : len;
    }
    if (lane_id() == 0 && len > 64) atomicMax(maxlen, len); // only groups past the in-wave sort size matter
}

// groups of <= 64 members: every member ranks itself inside its group on (key, index) and moves there;
// the group ids stay where they are (all members of a group carry the same one)
__global__ __launch_bounds__(LS_THREADS) void k_group_sort_small(const uint64_t *__restrict__ key,
                                                                const uint32_t *__restrict__ pos,
                                                                const uint32_t *__restrict__ seg,
                                                                const uint32_t *__restrict__ segstart, uint64_t count,
                                                                uint64_t *__restrict__ okey, uint32_t *__restrict__ opos,
                                                                uint64_t *__restrict__ big)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    const uint32_t sg = seg[i];
    const uint32_t a = segstart[sg], b = segstart[sg + 1];
    if (big) { // longer groups exist: flag their members for the radix path, (1 << 32) | starts-a-group
        const bool isbig = b - a > SMALL_SEG;
        big[i] = isbig ? ((1ull << 32) | (uint64_t)((uint32_t)i == a ? 1u : 0u)) : 0ull;
        if (isbig) return;
    }
    const uint64_t ki = key[i];
    uint32_t r = 0;
    for (uint32_t j = a; j < b; j++) {
        const uint64_t kj = key[j];
        r += (kj < ki || (kj == ki && j < (uint32_t)i)) ? 1u : 0u;
    }
    okey[a + r] = ki;
    opos[a + r] = pos[i];
}

__global__ __launch_bounds__(LS_THREADS) void k_gather_ranks(const uint32_t *__restrict__ isa,
                                                            const uint32_t *__restrict__ pos, uint64_t count, uint64_t h,
                                                            uint64_t n, uint64_t *__restrict__ key)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    uint64_t q = (uint64_t)pos[i] + h;
    if (q > n) q = n; // cannot happen for a tied suffix (it shares h bases with another one); isa[n] = 0
    key[i] = (uint64_t)isa[q] << 32;
}

} // namespace

int kiss_exact_refine(kiss_hip_ctx *ctx, uint64_t n, uint32_t h0, uint32_t *d_SA, uint8_t *heads_in)
{
    // heads_in == nullptr: DNA, SA ordered by h0 >= 32 bases, ties found by comparing the packed text;
    // heads_in != nullptr: the caller knows the groups already (general alphabet: equal 7-character keys)
    if ((!heads_in && h0 < 32) || h0 == 0 || n < h0) return KINTERNAL();
    ctx->stats.refine_form = 2;
    const uint64_t total = n + 1;
    const unsigned T = LS_THREADS;
    const bool dbg = ctx->opts.debug;
    KTRY(kiss_need_ctx_words(ctx));
    uint32_t *isa = ctx->CTX; // the induction's context words are dead by now: (n + 2) u32
    uint64_t *d_total = (uint64_t *)(ctx->d_small + 2);
    uint8_t *heads = heads_in;
    if (!heads) { // one byte per suffix-array entry, kept by the ctx between calls (sized for max_n like pairs1 / pairs2)
        if (!ctx->refine_heads) {
            const uint64_t cap = ctx->max_n + 2;
            hipError_t e = hipMalloc((void **)&ctx->refine_heads, cap);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                ctx->refine_heads = nullptr;
                ctx->last_hip_error = (int)e;
                return KISS_HIP_E_NOMEM;
            }
            ctx->ws_bytes += cap;
        }
        heads = ctx->refine_heads;
    }
    int rc = KISS_HIP_OK;
    do {
        if (!heads_in) {
            KTimer t(ctx, KISS_HIP_K_GROUP_HEADS, total);
            if (dbg && ctx->ctx_words_valid) {
                unsigned long long *dc = (unsigned long long *)(ctx->d_small + 44), hc = 0;
                KTRY(kiss_zero_u32(ctx, dc, 2));
                hipLaunchKernelGGL(k_count_taint, dim3((unsigned)div_up(total, T)), dim3(T), 0, ctx->stream, ctx->CTX, total, dc);
                KCHECK(hipMemcpyAsync(&hc, dc, 8, hipMemcpyDeviceToHost, ctx->stream));
                KCHECK(hipStreamSynchronize(ctx->stream));
                fprintf(stderr, "[kiss_hip] refine: %llu of %llu suffix-array entries are tainted\n", hc, (unsigned long long)total);
            }
            // the context words are still in CTX (it becomes the inverse suffix array only after this kernel)
            const bool no_taint = ctx->opts.no_taint; // (hooks build: compare every neighbour pair)
            bool done = false;
            if (!no_taint && ctx->ctx_words_valid) {
                // candidates first, compared densely afterwards; the list lives in posA (dead here).  More candidates than
                // it holds (texts that are one repeat): the one-kernel form below does the whole job instead.
                uint32_t *d_nc = ctx->rx_ghist; // GH_REGIONS counters (digit-histogram scratch of the radix sort, 3072 words, dead here)
                const uint64_t region_cap = ctx->m_cap / GH_REGIONS;
                KTRY(kiss_zero_u32(ctx, d_nc, GH_REGIONS));
                hipLaunchKernelGGL(k_heads_candidates, dim3((unsigned)div_up(total, GH_THREADS * GH_ITEMS)), dim3(GH_THREADS), 0,
                                   ctx->stream, d_SA, total, n, h0, ctx->CTX, heads, ctx->posA, region_cap, d_nc, 1u);
                uint32_t h_nc[GH_REGIONS];
                KCHECK(hipMemcpyAsync(h_nc, d_nc, sizeof h_nc, hipMemcpyDeviceToHost, ctx->stream));
                KCHECK(hipStreamSynchronize(ctx->stream));
                uint32_t mx = 0;
                for (uint32_t v : h_nc) mx = v > mx ? v : mx;
                if (mx <= region_cap) {
                    if (mx)
                        hipLaunchKernelGGL(k_heads_compare, dim3((unsigned)div_up((uint64_t)mx, T), GH_REGIONS), dim3(T), 0,
                                           ctx->stream, ctx->pk, d_SA, ctx->posA, region_cap, d_nc, h0, heads);
                    done = true;
                }
            }
            if (!done)
                hipLaunchKernelGGL(k_group_heads, dim3((unsigned)div_up(total, T)), dim3(T), 0, ctx->stream, ctx->pk, n, d_SA,
                                   total, h0, (no_taint || !ctx->ctx_words_valid) ? (const uint32_t *)nullptr : ctx->CTX,
                                   heads, 1u);
        }
        uint64_t tot;
        if ((rc = fc_count<FC_HEADS>(ctx, reinterpret_cast<const uint64_t *>(heads), nullptr, total, 0, 0, d_total))) break;
        if ((rc = fc_read_total(ctx, d_total, &tot))) break;
        uint64_t count = tot >> 32, nseg = tot & 0xFFFFFFFFull;
        ctx->stats.refine_items = count;
        if (dbg)
            fprintf(stderr, "[kiss_hip] refine: %llu of %llu suffixes tied at depth %u in %llu groups\n",
                    (unsigned long long)count, (unsigned long long)total, h0, (unsigned long long)nseg);
        if (count == 0) break;
        ctx->ctx_words_valid = false; // CTX becomes the inverse suffix array from here on
        // the inverse suffix array is only needed when something is tied
        if (ctx->opts.isa_direct) { // (hooks build: the plain random scatter)
            KTimer t(ctx, KISS_HIP_K_ISA, total);
            hipLaunchKernelGGL(k_isa_init, dim3((unsigned)div_up(total, T)), dim3(T), 0, ctx->stream, d_SA, total, isa);
        } else if ((rc = kiss_isa_build(ctx, d_SA, total, isa)))
            break;
        if (count > ctx->m_cap || count > ctx->t_cap) { // regrow the work buffers (their contents are dead); keeps CTX, pk
            const uint64_t want = count + count / 64 + 1024;
            if (count > ctx->m_cap && (rc = kiss_lms_reserve(ctx, want, want))) break; // (both sets of arrays in one go)
            if (count > ctx->t_cap && (rc = kiss_tied_reserve(ctx, want))) break;
            if ((rc = fc_count<FC_HEADS>(ctx, reinterpret_cast<const uint64_t *>(heads), nullptr, total, 0, 0, d_total))) break;
        }
        uint32_t *P = ctx->posA, *P2 = ctx->posB;
        uint32_t *S = ctx->slotA, *S2 = ctx->slotB;
        uint32_t *G = ctx->segA, *G2 = ctx->segB;
        uint32_t *SS = ctx->segstartA, *SS2 = ctx->segstartB;
        if ((rc = fc_compact<FC_HEADS, false>(ctx, reinterpret_cast<const uint64_t *>(heads), nullptr, d_SA, nullptr, total, 0,
                                              0, P, S, G, SS, nullptr, nullptr)))
            break;
        uint32_t *d_maxlen = ctx->d_small + 8;
        {
            KTimer t(ctx, KISS_HIP_K_ISA, count);
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(64), 0, ctx->stream, SS + nseg, (uint32_t)count, d_maxlen);
            hipLaunchKernelGGL(k_isa_update, dim3((unsigned)div_up(count, T)), dim3(T), 0, ctx->stream, P, S, G, SS, count,
                               isa, d_maxlen, 0);
        }
        uint64_t h = h0;
        while (count > 0) {
            if (h > 2 * n + 64) {
                rc = KINTERNAL();
                break;
            }
            const unsigned grid = (unsigned)div_up(count, T);
            {
                KTimer t(ctx, KISS_HIP_K_KEYGATHER, count);
                hipLaunchKernelGGL(k_gather_ranks, dim3(grid), dim3(T), 0, ctx->stream, isa, P, count, h, n, ctx->bkeyA);
            }
            uint64_t ml;
            if ((rc = read_u64(ctx, d_maxlen, &ml))) break; // longest group (0: none longer than 64)
            const uint32_t maxlen = (uint32_t)ml;
            // groups of <= 64 sort themselves in place; members of longer groups are pulled out, radix sorted on
            // (group, rank) and put back -- sorted keys in bkeyB, positions in bposB, group ids unchanged in G
            uint64_t *F1 = ctx->flags, *F2 = ctx->flags + ctx->t_cap;
            {
                KTimer t(ctx, KISS_HIP_K_SEGRANK, count);
                hipLaunchKernelGGL(k_group_sort_small, dim3(grid), dim3(T), 0, ctx->stream, ctx->bkeyA, P, G, SS, count,
                                   ctx->bkeyB, ctx->bposB, maxlen > SMALL_SEG ? F1 : nullptr);
            }
            if (maxlen > SMALL_SEG) {
                if ((rc = kiss_scan_u64(ctx, F1, F2, count))) break;
                hipLaunchKernelGGL(k_last_total, dim3(1), dim3(64), 0, ctx->stream, F1, F2, count, d_total);
                uint64_t bt;
                if ((rc = read_u64(ctx, d_total, &bt))) break;
                const uint64_t nbig = bt >> 32, nbigseg = bt & 0xFFFFFFFFull;
                RadixBufs bb;
                bb.key[0] = ctx->keyA;
                bb.key[1] = ctx->keyB;
                bb.pos[0] = ctx->lmsP;
                bb.pos[1] = ctx->lmsC;
                bb.seg[0] = ctx->bsegA;
                bb.seg[1] = ctx->bsegB;
                uint32_t *bidx = ctx->bposA;
                {
                    KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
                    hipLaunchKernelGGL(k_big_extract, dim3(grid), dim3(T), 0, ctx->stream, ctx->bkeyA, P, count, F1, F2,
                                       bb.key[0], bb.pos[0], bb.seg[0], bidx);
                }
                int bres = 0;
                if ((rc = kiss_radix_sort(ctx, bb, nbig, 32, bits_for(nbigseg), &bres))) break;
                ctx->stats.big_item_rounds += nbig;
                KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, nbig);
                hipLaunchKernelGGL(k_big_writeback, dim3((unsigned)div_up(nbig, T)), dim3(T), 0, ctx->stream, bb.key[bres],
                                   bb.pos[bres], bidx, nbig, ctx->bkeyB, ctx->bposB);
            }
            const uint64_t *skey = ctx->bkeyB;
            const uint32_t *spos = ctx->bposB, *sseg = G;
            ctx->stats.doubling_rounds++;
            ctx->stats.sort_item_rounds += count;
            if ((rc = fc_count<FC_KEY_SEG>(ctx, skey, sseg, count, 32, 0, d_total))) break;
            // singletons retire into SA and ISA; survivors are compacted (slots stay in index order)
            if ((rc = fc_compact<FC_KEY_SEG, true>(ctx, skey, sseg, spos, S, count, 32, 0, P2, S2, G2, SS2, d_SA, isa))) break;
            if ((rc = fc_read_total(ctx, d_total, &tot))) break;
            const uint64_t ncount = tot >> 32;
            nseg = tot & 0xFFFFFFFFull;
            if (dbg)
                fprintf(stderr, "[kiss_hip] refine h=%llu: (host clock %.3f s) items %llu (longest group %s%u) -> %llu in %llu groups\n",
                        (unsigned long long)h, (double)clock() / CLOCKS_PER_SEC, (unsigned long long)count, maxlen ? "" : "<= ", maxlen ? maxlen : 64u,
                        (unsigned long long)ncount, (unsigned long long)nseg);
            count = ncount;
            std::swap(P, P2);
            std::swap(S, S2);
            std::swap(G, G2);
            std::swap(SS, SS2);
            if (count) {
                KTimer t(ctx, KISS_HIP_K_ISA, count);
                hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(64), 0, ctx->stream, SS + nseg, (uint32_t)count, d_maxlen);
                hipLaunchKernelGGL(k_isa_update, dim3((unsigned)div_up(count, T)), dim3(T), 0, ctx->stream, P, S, G, SS,
                                   count, isa, d_maxlen, 0);
            }
            h *= 2;
        }
    } while (0);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (rc == KISS_HIP_OK && e != hipSuccess) {
        ctx->last_hip_error = (int)e;
        rc = KISS_HIP_E_HIP;
    }

    return rc;
}

// =====================================================================================================================
// Exact order at the LMS level (round 3): the doubling runs BEFORE the induction, over the LMS suffixes alone.
//
// The reference's KISS2 does the same thing in its own way (`kiss2_core.hpp:835-886`: exact order of the LMS suffixes by
// prefix doubling over an encoded LMS string, then ONE induction).  Here the h0-ordered merged LMS list L (ctx->lmsP, all
// m LMS suffixes, ties in position order, the tied ones tainted) is refined in place:
//   * rank[p >> 1] = slot of p in L (LMS positions are never neighbours, so p >> 1 is collision-free; tied suffixes carry the
//     slot of their group's first member) -- a third of the entries of a full inverse suffix array, and the m-entry list is
//     what the tie detection, the compaction and the rounds stream instead of the n + 1 entries of SA;
//   * a round: all groups share >= D bases.  The members of a group share the LMS positions inside their common window
//     too, so a group picks the last position x = p + d, d < D, that the window alone proves to be an LMS position
//     (k_lms_stride: a type needs a differing base to its right, inside the window), and its members sort on
//     rank[(p + d) >> 1]: equal keys = equal through d + D bases.  The next round's D is D + the smallest stride.
//   * a group whose window ends in >= D/2 bases without such a position (a long run of one base, mostly) is "stuck": its
//     members (<= LX_STUCK_MAX) are ordered by comparing the text directly, which is final.
// Anything unexpected -- a stuck group of more members, comparisons that run past LX_STUCK_STEPS steps for one member,
// a rank that was never written -- aborts: the list is merged again as it was and kiss_exact_refine finishes the job on
// the suffix array as before (texts like A^1000 C repeated: their LMS suffixes are further apart than any window).
namespace {

constexpr uint32_t LX_STUCK_STEPS = 1u << 18; // all comparisons of one member of a stuck group together, <= 128 bases per step
constexpr uint32_t LX_STUCK_MAX = 8192;       // members of a stuck group (each is compared with all the others)
enum { LX_MAXLEN = 0, LX_MIN_D = 1, LX_ABORT = 2, LX_STUCK = 3 };

__global__ void k_lx_ctl_init(uint32_t *__restrict__ ctl)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctl[LX_MIN_D] = 0xFFFFFFFFu;
        ctl[LX_ABORT] = 0;
        ctl[LX_STUCK] = 0;
    }
}

// suffix pi < suffix pj, both known to be equal on their first d bases (pi + d, pj + d <= n); the suffix that runs
// off the text first is the smaller one (the sentinel)
__device__ __forceinline__ bool exact_less(const uint64_t *__restrict__ pk, uint64_t n, uint64_t pi, uint64_t pj, uint64_t d,
                                           uint32_t &steps_left, bool *gave_up)
{
    for (; steps_left; steps_left--) {
        const uint64_t qi = pi + d, qj = pj + d;
        const uint64_t ri = n - qi, rj = n - qj; // bases left
        if (ri >= 160 && rj >= 160) {
            const uint32_t si = 2u * (uint32_t)(qi & 31u), sj = 2u * (uint32_t)(qj & 31u);
            uint64_t a[5], b[5];
            kiss_words5(pk, qi >> 5, a); // (aligned loads: kiss_internal.hpp)
            kiss_words5(pk, qj >> 5, b);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint64_t ki = (a[t] << si) | ((a[t + 1] >> 1) >> (63u - si));
                const uint64_t kj = (b[t] << sj) | ((b[t + 1] >> 1) >> (63u - sj));
                if (ki != kj) return ki < kj;
            }
            d += 128;
            continue;
        }
        uint64_t len = ri < rj ? ri : rj;
        if (len > 32) len = 32;
        uint64_t ki = kiss_key32(pk, qi), kj = kiss_key32(pk, qj);
        if (len < 32) {
            const uint64_t mask = len ? ~0ull << (64 - 2 * len) : 0ull;
            ki &= mask;
            kj &= mask;
        }
        if (ki != kj) return ki < kj;
        if (len < 32) return ri < rj; // one of them ends here
        d += 32;
    }
    *gave_up = true;
    return false;
}

// One thread per group: the stride d (0 = stuck).  Scans the window [p0, p0 + D) of the group's first member from its
// right end, at most scan_cap bases: position y + 1 is an LMS position if base(y) > base(y + 1) and y + 1 is S-type, and
// a type is only known once a differing base has been seen to the right.
__global__ __launch_bounds__(LS_THREADS) void k_lms_stride(const uint64_t *__restrict__ pk, uint64_t n,
                                                          const uint32_t *__restrict__ pos,
                                                          const uint32_t *__restrict__ segstart, uint64_t nseg, uint64_t D,
                                                          uint64_t scan_cap, uint32_t *__restrict__ dG,
                                                          uint32_t *__restrict__ ctl)
{
    const uint64_t g = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    uint32_t d = 0;
    const bool valid = g < nseg;
    if (valid) {
        const uint64_t p0 = pos[segstart[g]];
        if (p0 + D <= n) {
            uint64_t e = p0 + D; // bases [ylo, e) are still to be looked at
            const uint64_t ylo = (D - 1 > scan_cap) ? (e - 1 - scan_cap) : p0;
            int t = 0; // type of the base right of the current one: 0 unknown, 1 S, 2 L
            uint32_t cn = 0;
            bool first = true;
            while (e > ylo && d == 0) {
                const uint64_t s = (e - ylo > 32) ? e - 32 : ylo;
                const uint64_t w = kiss_key32(pk, s);
                int j = (int)(e - s) - 1;
                if (first) {
                    cn = (uint32_t)(w >> (62 - 2 * j)) & 3u;
                    j--;
                    first = false;
                }
                for (; j >= 0; j--) {
                    const uint32_t c = (uint32_t)(w >> (62 - 2 * j)) & 3u;
                    if (c > cn) {
                        if (t == 1) {
                            d = (uint32_t)(s + (uint64_t)j + 1 - p0);
                            break;
                        }
                        t = 2;
                    } else if (c < cn) {
                        t = 1;
                    }
                    cn = c;
                }
                e = s;
            }
        } else {
            atomicOr(&ctl[LX_ABORT], 8u); // a tied suffix without D bases: cannot be
        }
        dG[g] = d;
    }
    // smallest stride of the round (one atomic per wave, and only while it still lowers the minimum)
    uint32_t md = (valid && d) ? d : 0xFFFFFFFFu;
#pragma unroll
    for (int x = 32; x >= 1; x >>= 1) {
        const uint32_t o = __shfl_xor(md, x, 64);
        md = o < md ? o : md;
    }
    const uint64_t stuck = __ballot(valid && d == 0);
    if (lane_id() == 0) {
        if (md != 0xFFFFFFFFu && md < __hip_atomic_load(&ctl[LX_MIN_D], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMin(&ctl[LX_MIN_D], md);
        if (stuck) atomicAdd(&ctl[LX_STUCK], (uint32_t)__popcll(stuck));
    }
}

__global__ __launch_bounds__(LS_THREADS) void k_gather_ranks_lms(const uint32_t *__restrict__ rank,
                                                                const uint32_t *__restrict__ pos,
                                                                const uint32_t *__restrict__ seg,
                                                                const uint32_t *__restrict__ dG, uint64_t count, uint64_t n,
                                                                uint64_t *__restrict__ key, uint32_t *__restrict__ ctl)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    const uint32_t d = dG[seg[i]];
    uint64_t k = 0;
    if (d) {
        const uint64_t q = (uint64_t)pos[i] + d;
        const uint32_t r = q < n ? rank[q >> 1] : 0xFFFFFFFFu;
        if (r == 0xFFFFFFFFu) atomicOr(&ctl[LX_ABORT], 1u); // not an LMS position after all: cannot be
        k = (uint64_t)r << 32;
    }
    key[i] = k;
}

// k_group_sort_small for the LMS rounds: a stuck group orders itself by comparing the text from base D on
__global__ __launch_bounds__(LS_THREADS) void k_group_sort_small_lms(const uint64_t *__restrict__ pk, uint64_t n,
                                                                    const uint64_t *__restrict__ key,
                                                                    const uint32_t *__restrict__ pos,
                                                                    const uint32_t *__restrict__ seg,
                                                                    const uint32_t *__restrict__ segstart,
                                                                    const uint32_t *__restrict__ dG, uint64_t count,
                                                                    uint64_t D, uint64_t *__restrict__ okey,
                                                                    uint32_t *__restrict__ opos, uint64_t *__restrict__ big,
                                                                    uint32_t *__restrict__ ctl)
{
    const uint64_t i = (uint64_t)blockIdx.x * LS_THREADS + threadIdx.x;
    if (i >= count) return;
    const uint32_t sg = seg[i];
    const uint32_t a = segstart[sg], b = segstart[sg + 1];
    const bool stuck = dG[sg] == 0;
    const bool isbig = !stuck && b - a > SMALL_SEG; // (a stuck group is not sorted on keys, whatever its size)
    if (big) big[i] = isbig ? ((1ull << 32) | (uint64_t)((uint32_t)i == a ? 1u : 0u)) : 0ull;
    if (isbig) return;
    if (stuck && b - a > LX_STUCK_MAX) { // left where it is (the abort undoes the round anyway)
        if ((uint32_t)i == a) atomicOr(&ctl[LX_ABORT], 2u);
        okey[i] = 0;
        opos[i] = pos[i];
        return;
    }
    uint32_t r = 0;
    uint64_t ko;
    if (!stuck) {
        const uint64_t ki = key[i];
        for (uint32_t j = a; j < b; j++) {
            const uint64_t kj = key[j];
            r += (kj < ki || (kj == ki && j < (uint32_t)i)) ? 1u : 0u;
        }
        ko = ki;
    } else {
        const uint64_t pi = pos[i];
        bool gave_up = false;
        uint32_t steps_left = LX_STUCK_STEPS;
        for (uint32_t j = a; j < b; j++)
            if (j != (uint32_t)i) r += exact_less(pk, n, pos[j], pi, D, steps_left, &gave_up) ? 1u : 0u;
        if (gave_up) atomicOr(&ctl[LX_ABORT], 4u);
        ko = (uint64_t)r << 32; // all different: every member retires
    }
    okey[a + r] = ko;
    opos[a + r] = pos[i];
}

} // namespace

int kiss_lms_exact_refine(kiss_hip_ctx *ctx, uint64_t n, uint32_t h0, uint32_t *scratch, bool *resolved)
{
    *resolved = false;
    if (h0 < 32 || n < h0 || !scratch) return KINTERNAL();
    const uint64_t m = ctx->m;
    const unsigned T = LS_THREADS;
    const bool dbg = ctx->opts.debug;
    // hooks build: bit i set = the host waits for the stream at sync point i of every round
    const unsigned sync_points = ctx->opts.lx_sync_points;
    auto sync_point = [&](unsigned i) {
        if ((sync_points >> i) & 1u) (void)hipStreamSynchronize(ctx->stream);
    };
    ctx->hmerged = nullptr;
    if (m < 2) {
        KTRY(kiss_merge_lms(ctx));
        *resolved = true;
        return KISS_HIP_OK;
    }
    KTRY(kiss_need_ctx_words(ctx));
    ctx->ctx_words_valid = false;
    uint32_t *L = ctx->lmsP, *C = ctx->lmsC;
    // CTX (n + 2 words, not in use before the induction): the rank array, then one tie flag byte per list entry
    // (and, at its far end, ctx->hfar: the tie flags the LMS sort wrote for the far list, api.hip)
    uint32_t *R = ctx->CTX;
    const uint64_t r_words = (n >> 1) + 1;
    uint8_t *heads = reinterpret_cast<uint8_t *>(ctx->CTX + ((r_words + 3) & ~3ull));
    if (((r_words + 3) & ~3ull) + (m + 16) / 4 + 1 > ctx->max_n + 2) return KINTERNAL();
    // the merged list; with the sort's own tie flags (ctx->hfar) the flags of the merged list come out of the same pass
    const bool sort_flags = ctx->hfar != nullptr && ctx->h_depth == h0;
    ctx->hmerged = sort_flags ? heads : nullptr;
    ctx->lms_merged = false;
    {
        const int mrc = kiss_merge_lms(ctx);
        ctx->hmerged = nullptr;
        if (mrc) return mrc;
    }
    uint64_t *d_total = (uint64_t *)(ctx->d_small + 2);
    uint32_t *ctl = ctx->d_small + 8;
    // the partition's pairs need 2 m words of the (n + 1)-word scratch, 8-byte aligned: a text with n/2 LMS suffixes in a
    // buffer that starts on an odd word has one word too few -- the suffix-array form takes that call
    if (((uintptr_t)scratch & 7) && 2 * m + 2 > n + 1) {
        ctx->lms_merged = false;
        return kiss_merge_lms(ctx);
    }
    uint64_t *pairs1 = reinterpret_cast<uint64_t *>(((uintptr_t)scratch + 7) & ~(uintptr_t)7);
    uint32_t *bigpos0 = scratch, *bigpos1 = scratch + m; // (after the rank array is built: pairs1 is dead then)

    int rc = KISS_HIP_OK;
    bool aborted = false;
    uint32_t why = 0;
    const uint64_t rounds_before = ctx->stats.doubling_rounds, item_rounds_before = ctx->stats.sort_item_rounds;
    do {
        if (!sort_flags || dbg) {
            // tie flags by comparison: both neighbours tainted and equal on their first h0 bases (the form without the sort's
            // flags -- KISS_HIP_LMS_HEADS_BY_COMPARE=1 --; with KISS_HIP_DEBUG also run beside them as a cross-check: every
            // pair the sort calls tied must be tied here too)
            uint8_t *hc = sort_flags ? heads + ((m + 64) & ~15ull) : heads;
            KTimer t(ctx, KISS_HIP_K_GROUP_HEADS, m);
            uint32_t *d_nc = ctx->rx_ghist;
            const uint64_t region_cap = ctx->m_cap / GH_REGIONS;
            if ((rc = kiss_zero_u32(ctx, d_nc, GH_REGIONS))) break;
            hipLaunchKernelGGL(k_heads_candidates, dim3((unsigned)div_up(m, GH_THREADS * GH_ITEMS)), dim3(GH_THREADS), 0, ctx->stream, L, m, n,
                               h0, C, hc, ctx->posA, region_cap, d_nc, 0u);
            uint32_t h_nc[GH_REGIONS];
            if (hipMemcpyAsync(h_nc, d_nc, sizeof h_nc, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                rc = KISS_HIP_E_HIP;
                break;
            }
            uint32_t mx = 0;
            for (uint32_t v : h_nc) mx = v > mx ? v : mx;
            if (mx <= region_cap) {
                if (mx)
                    hipLaunchKernelGGL(k_heads_compare, dim3((unsigned)div_up((uint64_t)mx, T), GH_REGIONS), dim3(T), 0, ctx->stream,
                                       ctx->pk, L, ctx->posA, region_cap, d_nc, h0, hc);
            } else {
                hipLaunchKernelGGL(k_group_heads, dim3((unsigned)div_up(m, T)), dim3(T), 0, ctx->stream, ctx->pk, n, L, m, h0, C,
                                   hc, 0u);
            }
            if (sort_flags) { // debug cross-check
                std::vector<uint8_t> a(m), b(m);
                if (hipMemcpyAsync(a.data(), heads, m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                    hipMemcpyAsync(b.data(), hc, m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                    hipStreamSynchronize(ctx->stream) != hipSuccess) {
                    rc = KISS_HIP_E_HIP;
                    break;
                }
                uint64_t tied_sort = 0, tied_cmp = 0, bad = 0;
                for (uint64_t i = 0; i < m; i++) {
                    tied_sort += a[i] == 0;
                    tied_cmp += b[i] == 0;
                    bad += (a[i] == 0 && b[i] != 0);
                }
                fprintf(stderr, "[kiss_hip] lms refine: tie flags from the sort: %llu entries tied with their predecessor; by comparison "
                                "of %u bases: %llu; tied for the sort but not by comparison: %llu\n",
                        (unsigned long long)tied_sort, h0, (unsigned long long)tied_cmp, (unsigned long long)bad);
                if (bad) {
                    std::vector<uint32_t> hl(m), hcw(m);
                    (void)hipMemcpy(hl.data(), L, m * 4, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(hcw.data(), C, m * 4, hipMemcpyDeviceToHost);
                    int shown = 0;
                    for (uint64_t i = 0; i < m && shown < 6; i++)
                        if (a[i] == 0 && b[i] != 0) {
                            shown++;
                            fprintf(stderr, "[kiss_hip]   entry %llu of %llu (m_far %llu, near %u): position %u (taint %u), predecessor %u "
                                            "(taint %u); flags around it, sort: %u %u %u  comparison: %u %u %u\n",
                                    (unsigned long long)i, (unsigned long long)m, (unsigned long long)ctx->m_far, ctx->rm_E, hl[i],
                                    hcw[i] >> 31, i ? hl[i - 1] : 0u, i ? hcw[i - 1] >> 31 : 0u, i ? a[i - 1] : 9u, a[i],
                                    i + 1 < m ? a[i + 1] : 9u, i ? b[i - 1] : 9u, b[i], i + 1 < m ? b[i + 1] : 9u);
                        }
                    rc = KINTERNAL();
                    break;
                }
            }
        }
        uint64_t tot;
        if ((rc = fc_count<FC_HEADS>(ctx, reinterpret_cast<const uint64_t *>(heads), nullptr, m, 0, 0, d_total))) break;
        if ((rc = fc_read_total(ctx, d_total, &tot))) break;
        uint64_t count = tot >> 32, nseg = tot & 0xFFFFFFFFull;
        ctx->stats.refine_items = count;
        if (dbg)
            fprintf(stderr, "[kiss_hip] lms refine: %llu of %llu LMS suffixes tied at depth %u in %llu groups\n",
                    (unsigned long long)count, (unsigned long long)m, h0, (unsigned long long)nseg);
        if (count == 0) break; // nothing is tied: the list is exact as it stands
        if (count > ctx->t_cap) { // (contents of the tied-segment arrays are dead)
            if ((rc = kiss_tied_reserve(ctx, count + count / 64 + 1024))) break;
            if ((rc = fc_count<FC_HEADS>(ctx, reinterpret_cast<const uint64_t *>(heads), nullptr, m, 0, 0, d_total))) break;
        }
        // ranks (entries without an LMS position read as 0xFFFFFFFF)
        if ((rc = kiss_rank_build_lms(ctx, L, ctx->lms_pos_complete ? ctx->lms_pos : (const uint32_t *)nullptr, m, n, R, pairs1, ctx->keyA,
                                      reinterpret_cast<uint32_t *>(ctx->keyB),
                                      2 * ctx->m_cap)))
            break;
        uint32_t *P = ctx->posA, *P2 = ctx->posB;
        uint32_t *S = ctx->slotA, *S2 = ctx->slotB;
        uint32_t *G = ctx->segA, *G2 = ctx->segB;
        uint32_t *SS = ctx->segstartA, *SS2 = ctx->segstartB;
        uint32_t *dG = ctx->bslot;
        if ((rc = fc_compact<FC_HEADS, false>(ctx, reinterpret_cast<const uint64_t *>(heads), nullptr, L, nullptr, m, 0, 0, P, S, G,
                                              SS, nullptr, nullptr)))
            break;
        {
            KTimer t(ctx, KISS_HIP_K_ISA, count);
            hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(64), 0, ctx->stream, SS + nseg, (uint32_t)count, ctl + LX_MAXLEN);
            hipLaunchKernelGGL(k_isa_update, dim3((unsigned)div_up(count, T)), dim3(T), 0, ctx->stream, P, S, G, SS, count, R,
                               ctl + LX_MAXLEN, 1);
        }
        uint64_t D = h0;
        int rounds = 0;
        while (count > 0) {
            if (++rounds > 64 || D + 1 > n) {
                aborted = true;
                why = 16;
                break;
            }
            const unsigned grid = (unsigned)div_up(count, T);
            {
                KTimer t(ctx, KISS_HIP_K_KEYGATHER, count);
                hipLaunchKernelGGL(k_lx_ctl_init, dim3(1), dim3(64), 0, ctx->stream, ctl);
                hipLaunchKernelGGL(k_lms_stride, dim3((unsigned)div_up(nseg, T)), dim3(T), 0, ctx->stream, ctx->pk, n, P, SS, nseg,
                                   D, D / 2, dG, ctl);
                hipLaunchKernelGGL(k_gather_ranks_lms, dim3(grid), dim3(T), 0, ctx->stream, R, P, G, dG, count, n, ctx->bkeyA,
                                   ctl);
            }
            if ((rc = kiss_readback(ctx, ctl, 4))) break;
            const uint32_t maxlen = ctx->h_pinned[LX_MAXLEN], min_d = ctx->h_pinned[LX_MIN_D];
            const uint32_t nstuck = ctx->h_pinned[LX_STUCK];
            if (ctx->h_pinned[LX_ABORT]) {
                aborted = true;
                why = ctx->h_pinned[LX_ABORT];
                break;
            }
            uint64_t *F1 = ctx->flags, *F2 = ctx->flags + ctx->t_cap;
            {
                KTimer t(ctx, KISS_HIP_K_SEGRANK, count);
                hipLaunchKernelGGL(k_group_sort_small_lms, dim3(grid), dim3(T), 0, ctx->stream, ctx->pk, n, ctx->bkeyA, P, G, SS,
                                   dG, count, D, ctx->bkeyB, ctx->bposB, maxlen > SMALL_SEG ? F1 : nullptr, ctl);
            }
            sync_point(0);
            if (maxlen > SMALL_SEG) {
                if ((rc = kiss_scan_u64(ctx, F1, F2, count))) break;
                sync_point(1);
                hipLaunchKernelGGL(k_last_total, dim3(1), dim3(64), 0, ctx->stream, F1, F2, count, d_total);
                uint64_t bt;
                if ((rc = read_u64(ctx, d_total, &bt))) break;
                const uint64_t nbig = bt >> 32, nbigseg = bt & 0xFFFFFFFFull;
                if (nbig) {
                    RadixBufs bb;
                    bb.key[0] = ctx->keyA;
                    bb.key[1] = ctx->keyB;
                    bb.pos[0] = bigpos0;
                    bb.pos[1] = bigpos1;
                    bb.seg[0] = ctx->bsegA;
                    bb.seg[1] = ctx->bsegB;
                    uint32_t *bidx = ctx->bposA;
                    {
                        KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, count);
                        hipLaunchKernelGGL(k_big_extract, dim3(grid), dim3(T), 0, ctx->stream, ctx->bkeyA, P, count, F1, F2, bb.key[0],
                                           bb.pos[0], bb.seg[0], bidx);
                    }
                    int bres = 0;
                    sync_point(2);
                    if ((rc = kiss_radix_sort(ctx, bb, nbig, 32, bits_for(nbigseg), &bres))) break;
                    sync_point(3);
                    ctx->stats.big_item_rounds += nbig;
                    KTimer t(ctx, KISS_HIP_K_FLAG_COMPACT, nbig);
                    hipLaunchKernelGGL(k_big_writeback, dim3((unsigned)div_up(nbig, T)), dim3(T), 0, ctx->stream, bb.key[bres],
                                       bb.pos[bres], bidx, nbig, ctx->bkeyB, ctx->bposB);
                    sync_point(4);
                }
            }
            const uint64_t *skey = ctx->bkeyB;
            const uint32_t *spos = ctx->bposB, *sseg = G;
            ctx->stats.doubling_rounds++;
            ctx->stats.sort_item_rounds += count;
            if ((rc = fc_count<FC_KEY_SEG>(ctx, skey, sseg, count, 32, 0, d_total))) break;
            sync_point(5);
            // singletons retire into the list and the rank array; survivors are compacted (slots stay in index order)
            // (and take the context word of their own position along: the list entry they replace was another suffix's)
            if ((rc = fc_compact<FC_KEY_SEG, true>(ctx, skey, sseg, spos, S, count, 32, 0, P2, S2, G2, SS2, L, R, nullptr, nullptr,
                                                   nullptr, nullptr, nullptr, 1, C)))
                break;
            if ((rc = fc_read_total(ctx, d_total, &tot))) break;
            const uint64_t ncount = tot >> 32;
            if ((rc = kiss_readback(ctx, ctl, 4))) break; // what the sort kernel had to say
            if (ctx->h_pinned[LX_ABORT]) {
                aborted = true;
                why = ctx->h_pinned[LX_ABORT];
                break;
            }
            if (dbg)
                fprintf(stderr, "[kiss_hip] lms refine D=%llu: items %llu in %llu groups (%u stuck, longest %s%u, stride >= %u) -> %llu\n",
                        (unsigned long long)D, (unsigned long long)count, (unsigned long long)nseg, nstuck, maxlen ? "" : "<= ",
                        maxlen ? maxlen : 64u, min_d, (unsigned long long)ncount);
            nseg = tot & 0xFFFFFFFFull;
            count = ncount;
            if (min_d != 0xFFFFFFFFu) D += min_d;
            std::swap(P, P2);
            std::swap(S, S2);
            std::swap(G, G2);
            std::swap(SS, SS2);
            if (count) {
                KTimer t(ctx, KISS_HIP_K_ISA, count);
                hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(64), 0, ctx->stream, SS + nseg, (uint32_t)count, ctl + LX_MAXLEN);
                hipLaunchKernelGGL(k_isa_update, dim3((unsigned)div_up(count, T)), dim3(T), 0, ctx->stream, P, S, G, SS, count, R,
                                   ctl + LX_MAXLEN, 1);
            }
        }
    } while (0);
    if (rc == KISS_HIP_OK && hipGetLastError() != hipSuccess) rc = KISS_HIP_E_HIP;
    if (rc) return rc;
    if (aborted) { // the list as the bounded phase left it: kiss_exact_refine takes over after the induction
        if (dbg) fprintf(stderr, "[kiss_hip] lms refine: gave up (reason bits %u), the suffix-array form takes over\n", why);
        ctx->stats.doubling_rounds = (uint32_t)rounds_before;
        ctx->stats.sort_item_rounds = item_rounds_before;
        ctx->lms_merged = false;
        ctx->hmerged = nullptr;
        return kiss_merge_lms(ctx);
    }
    *resolved = true;
    return KISS_HIP_OK;
}
