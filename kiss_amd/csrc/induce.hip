// induce.hip -- the L and S induction sweeps as a sequence of stable 4-way partition passes.
//
// Restates the *result* of put_lms_suffix + induced_sort (reference
// include/biovoltron/algo/sort/kiss_common.hpp:445-481, 372-403, 405-420, 192-224), not their loops.
//
// CPU form: one left-to-right scan of SA in which every visited suffix v appends v-1 to the head of
// bucket S[v-1] when v-1 is L-type, then the mirror image right-to-left for S-type.  With sigma = 4 the
// scan order is: sentinel, L-part(A), LMS(A), L-part(C), LMS(C), ... and every one of those source
// segments is homogeneous (same first character, same type).  So one segment is ONE data-parallel
// pass: read the segment in order, look at the preceding base, and append stably to at most four
// destination lists.  The only serial dependency left is a segment that appends to itself (runs of
// one character: L-part(c) feeding L-part(c)); it is processed in rounds, round r+1 being what round r
// appended, and the rounds shrink geometrically.  Small rounds are chased to exhaustion inside a
// single workgroup without returning to the host.
//
// No random text reads: every item carries a context word holding the <= 15 bases that precede it
// (gathered once per LMS suffix after the LMS sort); an induced item inherits its parent's word
// shifted by one base.  A word that runs empty (monotone runs longer than 15) is re-gathered.
//
// The LMS suffixes are never copied into SA (put_lms_suffix disappears): the L sweep reads them from
// the sorted LMS array, and the S sweep overwrites every S-type slot anyway (induced_clear disappears).
#include "kiss_internal.hpp"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int IN_THREADS = 256;
#ifndef KISS_IN_ITEMS
#define KISS_IN_ITEMS 8 // (items per thread of the partition passes; -DKISS_IN_ITEMS=4 / 16: the tile-size A-B of round 3)
#endif
constexpr int IN_ITEMS = KISS_IN_ITEMS;
constexpr int IN_WAVES = IN_THREADS / 64;
constexpr int IN_TILE = IN_THREADS * IN_ITEMS;   // 2048
constexpr int SM_THREADS = 1024;
constexpr uint64_t SMALL_MAX = 8192;             // rounds up to this size run in the single-workgroup kernel

// word an induced item inherits: the parent's without its nearest base; the taint bit stays where it is
__device__ __forceinline__ uint32_t child_ctx(uint32_t parent)
{
    return (KISS_CTX_WORD(parent) >> 2) | (parent & KISS_CTX_TAINT);
}

// One byte per context word written into the arrays parallel to SA (ctx->CLS, round 4): the word's class bits and a flag for
// "no bases left" -- all the count pass needs of a word, so it reads 1 byte per item instead of 4.
__device__ __forceinline__ uint32_t cls_byte(uint32_t c)
{
    return (c & 3u) | ((c & 0x7FFFFFFEu) == 0u ? 4u : 0u);
}
static_assert(KISS_EMPTY_CTX == 1u && KISS_CTX_TAINT == 0x80000000u, "no bases <=> all bits but 0 and 31 clear");

// The sorted LMS list is read where the LMS sort left it (ctx->lms_sorted_far / lms_ctx_far) instead of from a merged copy
// (place.hip): the handful of near-end suffixes, ranked by the scalar tail of the comparator, live in a small table
// sorted by their index in the merged list.  Merged index j holds near-end suffix t if fin[t] == j, else far suffix
// j - #{t : fin[t] < j}.
struct LmsRemap {
    const uint32_t *fin;  // merged index of the t-th near-end suffix, ascending
    const uint32_t *npos; // its text position
    uint32_t E;
};

__device__ __forceinline__ uint32_t remap_below(const LmsRemap &rm, uint64_t j) // #{t : fin[t] < j}
{
    uint32_t lo = 0, hi = rm.E;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)rm.fin[mid] < j) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// the same for an index that is the same in every lane of the calling wave (all lanes must call): the table is compared
// 64 entries at a time instead of by a chain of dependent loads per lane -- the count and scatter passes ask this twice per
// tile of 2048 items, and with the binary search those two questions cost more than the tile itself
__device__ __forceinline__ uint32_t remap_below_wave(const LmsRemap &rm, uint64_t j)
{
    if (rm.E > 256u) return remap_below(rm, j); // (a k in the hundred thousands: rare, and the search is uniform)
    uint32_t cnt = 0;
    for (uint32_t base = 0; base < rm.E; base += 64u) {
        const uint32_t i = base + lane_id();
        const bool lt = i < rm.E && (uint64_t)rm.fin[i] < j;
        const uint32_t c = (uint32_t)__popcll(__ballot(lt));
        cnt += c;
        if (c < 64u) break; // the table ascends
    }
    return cnt;
}

// class of a source item: 0..3 = append v-1 to bucket of that base, 4 = nothing to append.
// A word without bases (KISS_EMPTY_CTX: run empty; 0: a far LMS suffix that left the sort without one) is gathered
// from the text and kept.  REMAP: idx is an index into the merged LMS list (see LmsRemap), else a physical one.
template <bool REMAP>
__device__ __forceinline__ uint32_t item_class(const uint64_t *__restrict__ pk, const uint32_t *srcP, uint32_t *srcC,
                                               int64_t idx, uint32_t emitmask, uint32_t *v_out, uint32_t *ctx_out,
                                               const LmsRemap &rm, uint8_t *srcK = nullptr)
{
    int64_t phys = idx;
    if (REMAP) {
        const uint32_t t = remap_below(rm, (uint64_t)idx);
        if (t < rm.E && (int64_t)rm.fin[t] == idx) { // a near-end suffix: placed by the tie rules, so tainted
            const uint32_t v = rm.npos[t];
            const uint32_t c = kiss_load_ctx(pk, v) | KISS_CTX_TAINT;
            *v_out = v;
            *ctx_out = c;
            const uint32_t pc = c & 3u;
            return (v != 0 && ((emitmask >> pc) & 1u)) ? pc : 4u;
        }
        phys = idx - (int64_t)t;
    }
    uint32_t c = srcC[phys];
    uint32_t v = srcP[phys];
    *v_out = v;
    if (KISS_CTX_WORD(c) <= KISS_EMPTY_CTX) {
        if (v == 0) {
            *ctx_out = c;
            return 4u;
        }
        c = kiss_load_ctx(pk, v) | (c & KISS_CTX_TAINT);
        srcC[phys] = c; // keep the refreshed word: the scatter pass (and the other sweep) reuse it
        if (srcK) srcK[phys] = (uint8_t)cls_byte(c);
    }
    *ctx_out = c;
    uint32_t pc = c & 3u;
    return ((emitmask >> pc) & 1u) ? pc : 4u;
}

// the count pass only needs the class: the position is read only when the context word has no bases
template <bool REMAP>
__device__ __forceinline__ uint32_t item_class_only(const uint64_t *__restrict__ pk, const uint32_t *srcP, uint32_t *srcC,
                                                    int64_t idx, uint32_t emitmask, const LmsRemap &rm, uint8_t *srcK = nullptr)
{
    if (REMAP) {
        uint32_t v, c;
        return item_class<true>(pk, srcP, srcC, idx, emitmask, &v, &c, rm);
    }
    uint32_t c = srcC[idx];
    if (KISS_CTX_WORD(c) <= KISS_EMPTY_CTX) {
        const uint32_t v = srcP[idx];
        if (v == 0) return 4u;
        c = kiss_load_ctx(pk, v) | (c & KISS_CTX_TAINT);
        srcC[idx] = c;
        if (srcK) srcK[idx] = (uint8_t)cls_byte(c);
    }
    const uint32_t pc = c & 3u;
    return ((emitmask >> pc) & 1u) ? pc : 4u;
}

// inclusive prefix sum over the 64 lanes of the wave (all lanes must call): six v_add_u32_dpp
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true); // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true); // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true); // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true); // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true); // row_bcast:31 into rows 2 and 3
    return x;
}

// ---- pass 1: per-tile class counts ---------------------------------------------------
// One WAVE per tile (a workgroup = IN_WAVES tiles): counting does not care about order, so a lane takes four
// consecutive items per step (16-byte loads, all of the tile's steps in flight at once) and keeps the four class
// counts in 16-bit fields of one 64-bit word (a tile has 2048 items); no LDS, no barrier, no atomic.
template <bool REMAP>
__global__ __launch_bounds__(IN_THREADS) void k_induce_count(const uint64_t *__restrict__ pk, const uint32_t *srcP,
                                                            uint32_t *srcC, int64_t beg, uint64_t N, int dir,
                                                            uint32_t emitmask, uint32_t *__restrict__ counts,
                                                            uint64_t tiles, LmsRemap rm, uint8_t *srcK)
{
    if (blockIdx.x == 0 && threadIdx.x == 4) counts[4 * tiles] = 0; // the extra last entry of the scan input
    const uint64_t tile = (uint64_t)blockIdx.x * IN_WAVES + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    struct __attribute__((packed, aligned(4))) U4 {
        uint32_t v[4];
    };
    constexpr int STEPS = IN_TILE / 256; // 256 items per wave step
    const uint32_t lane = lane_id();
    const uint64_t t0 = tile * IN_TILE;
    uint64_t acc = 0;
    // REMAP (dir > 0 only): the tile is read as a block when no near-end suffix falls inside it -- then its items are
    // consecutive entries of the far list, `shift` entries below their merged index
    int64_t shift = 0;
    bool block = t0 + IN_TILE <= N;
    if (REMAP && block) { // (t0 is the same in all lanes of the wave)
        const uint32_t b0 = remap_below_wave(rm, (uint64_t)beg + t0);
        block = b0 == remap_below_wave(rm, (uint64_t)beg + t0 + IN_TILE);
        shift = (int64_t)b0;
    }
    // srcK (may be null): one class byte per word of srcC (cls_byte).  A full tile then reads 2 KiB instead of 8: the
    // byte stretch [A, A + 2048) as aligned 16-byte pieces -- two per lane, and a 129th for lane 0 when A is not aligned;
    // bytes outside the stretch are replaced by 8 ("skip").  A byte that says "no bases left" sends the whole tile to
    // the word path below, which refreshes the word (and its byte).
    bool counted = false;
    if (!REMAP && srcK != nullptr && block) { // (wave-uniform)
        const int64_t lo_index = dir > 0 ? beg + (int64_t)t0 : beg - (int64_t)(t0 + IN_TILE - 1);
        const uint8_t *const A = srcK + lo_index;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(A) & 15u);
        const uint4 *const ap = reinterpret_cast<const uint4 *>(A - mis);
        static_assert(IN_TILE == 2048, "64 lanes x 2 x 16 bytes");
        uint32_t w[12];
        {
            const uint4 q0 = ap[lane], q1 = ap[64 + lane];
            w[0] = q0.x, w[1] = q0.y, w[2] = q0.z, w[3] = q0.w;
            w[4] = q1.x, w[5] = q1.y, w[6] = q1.z, w[7] = q1.w;
            w[8] = w[9] = w[10] = w[11] = 0x08080808u;
        }
        if (mis != 0 && lane == 0) {
            const uint4 q2 = ap[128];
            const uint32_t tail[4] = {q2.x, q2.y, q2.z, q2.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int nb = (int)mis - 4 * i; // bytes of this word that lie below byte `mis` of the piece
                const uint32_t lowmask = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ~(0xFFFFFFFFu << (8 * nb)));
                w[i] = (w[i] & ~lowmask) | (0x08080808u & lowmask);         // first piece: in front of A
                w[8 + i] = (tail[i] & lowmask) | (0x08080808u & ~lowmask);  // 129th piece: the stretch's last bytes
            }
        }
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) any |= w[i];
        if (__ballot((any & 0x04040404u) != 0u) == 0ull) {
            uint32_t cnt8 = 0; // (a lane has at most 48 items)
#pragma unroll
            for (int i = 0; i < 12; i++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t b = (w[i] >> (8 * e)) & 0xFFu;
                    const uint32_t pc = b & 3u;
                    cnt8 += ((emitmask >> pc) & ~(b >> 3) & 1u) << (pc << 3);
                }
            }
            acc = (uint64_t)((cnt8 & 0xFFu) | ((cnt8 & 0xFF00u) << 8)) | ((uint64_t)(((cnt8 >> 16) & 0xFFu) | ((cnt8 >> 8) & 0xFF0000u)) << 32);
            counted = true;
        }
    }
    if (counted) {
    } else if (block) {
        U4 t[STEPS];
        int64_t p0[STEPS];
        uint32_t least = 0xFFFFFFFFu; // (a word without bases: refreshed from the text -- rare, one test per wave)
#pragma unroll
        for (int q = 0; q < STEPS; q++) {
            const uint64_t i0 = t0 + (uint64_t)q * 256 + 4 * lane;
            p0[q] = (dir > 0 ? beg + (int64_t)i0 : beg - (int64_t)(i0 + 3)) - shift; // lowest address of my four items
            t[q] = *reinterpret_cast<const U4 *>(srcC + p0[q]);
        }
#pragma unroll
        for (int q = 0; q < STEPS; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) least = min(least, t[q].v[e] & 0x7FFFFFFEu);
        uint32_t cnt8 = 0; // my items per class, 8-bit fields (a lane has 32 items of the tile)
        if (__ballot(least == 0u) == 0ull) {
#pragma unroll
            for (int q = 0; q < STEPS; q++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t pc = t[q].v[e] & 3u;
                    cnt8 += ((emitmask >> pc) & 1u) << (pc << 3);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < STEPS; q++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const uint32_t c = t[q].v[e];
                    uint32_t cls;
                    if (KISS_CTX_WORD(c) <= KISS_EMPTY_CTX) cls = item_class_only<false>(pk, srcP, srcC, p0[q] + e, emitmask, rm, REMAP ? nullptr : srcK); // refresh path
                    else {
                        const uint32_t pc = c & 3u;
                        cls = ((emitmask >> pc) & 1u) ? pc : 4u;
                    }
                    cnt8 += (cls < 4u ? 1u : 0u) << ((cls & 3u) << 3);
                }
            }
        }
        static_assert(IN_TILE / 64 <= 255, "8-bit counters per lane");
        acc = (uint64_t)((cnt8 & 0xFFu) | ((cnt8 & 0xFF00u) << 8)) | ((uint64_t)(((cnt8 >> 16) & 0xFFu) | ((cnt8 >> 8) & 0xFF0000u)) << 32);
    } else {
        for (uint64_t i = t0 + lane; i < N && i < t0 + IN_TILE; i += 64) {
            const uint32_t cls = item_class_only<REMAP>(pk, srcP, srcC, beg + (int64_t)dir * (int64_t)i, emitmask, rm, REMAP ? nullptr : srcK);
            acc += cls < 4u ? 1ull << (16 * cls) : 0ull;
        }
    }
    // totals of the wave: two DPP prefix sums over the 16-bit fields, the last lane holds the sums
    const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add((uint32_t)acc), 63);
    const uint32_t shi = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add((uint32_t)(acc >> 32)), 63);
    acc = ((uint64_t)shi << 32) | slo;
    if (lane < 4) counts[(uint64_t)lane * tiles + tile] = (uint32_t)(acc >> (16 * lane)) & 0xFFFFu;
}

struct DstPos {
    int64_t p[4];
};

// ---- pass 2: stable scatter ------------------------------------------------------------
// Round 4: the kernel was bound by its vector instructions, not by memory (SQ counters, profiles/r04_pmc_sq_*: 770 VALU
// and 380 SALU instructions per wave of 512 items = the kernel's whole duration on the four SIMDs of a CU).  Rewritten
// around packed arithmetic: the four class counters of a thread are 8-bit fields of one register, their wave prefixes
// 16-bit fields of two (two 6-instruction DPP scans instead of four shuffle scans), an item's place in the staged tile is
// one v_perm_b32 out of the packed bases plus its rank, and the direction of the sweep is a template parameter (the
// reversal of a thread's items is register naming, not selects).
template <bool REMAP, int DIR>
__global__ __launch_bounds__(IN_THREADS) void k_induce_scatter(const uint64_t *__restrict__ pk, const uint32_t *srcP,
                                                              uint32_t *srcC, int64_t beg, uint64_t N,
                                                              uint32_t emitmask, const uint32_t *__restrict__ ex,
                                                              uint64_t tiles, DstPos dst, uint32_t *SA, uint32_t *CTX,
                                                              uint32_t *__restrict__ totals, LmsRemap rm,
                                                              uint8_t *CLS,  // class bytes parallel to CTX (may be null)
                                                              uint8_t *srcK) // ... parallel to srcC (may be null)
{
    static_assert(IN_ITEMS == 8 && IN_WAVES == 4, "8-bit counters per thread, 4-bit ranks of a thread's items in one word");
    static_assert(!REMAP || DIR > 0, "the LMS list is read left to right only");
    __shared__ uint32_t wtot[IN_WAVES][2]; // class totals of a wave, packed: [0] = class 0 | class 1 << 16, [1] = 2 | 3 << 16
    __shared__ __attribute__((aligned(16))) uint32_t stP[IN_TILE];
    __shared__ __attribute__((aligned(16))) uint32_t stC[IN_TILE];
    // items appended per class = differences of the scanned counts (4 * tiles + 1 entries)
    if (blockIdx.x == 0 && threadIdx.x < 4) totals[threadIdx.x] = ex[(uint64_t)(threadIdx.x + 1) * tiles] - ex[(uint64_t)threadIdx.x * tiles];
    const int wave = threadIdx.x >> 6;
    // A thread owns IN_ITEMS consecutive items (item order = thread order, then order inside the thread), read with
    // 16-byte loads; its rank inside a class = items of that class in lower lanes (wave prefix of the per-lane
    // counts) + earlier ones of its own.
    struct __attribute__((packed, aligned(4))) U4 {
        uint32_t v[4];
    };
    const uint64_t i0 = (uint64_t)blockIdx.x * IN_TILE + (uint64_t)threadIdx.x * IN_ITEMS;
    uint32_t vv[IN_ITEMS], cc[IN_ITEMS]; // positions and context words, in the order of the pass
    uint32_t okm = 0;                    // bit e: item e appends
    uint32_t loc = 0;                    // bits 4e .. 4e+3: earlier items of my own with item e's class
    uint32_t cnt8 = 0;                   // my items per class, 8-bit fields
    auto take = [&](int e, uint32_t pc, uint32_t ok) { // class pc (meaningless when ok == 0)
        const uint32_t sh = pc << 3;
        loc |= ((cnt8 >> sh) & 15u) << (4 * e);
        cnt8 += ok << sh;
        okm |= ok << e;
    };
    int64_t shift = 0; // REMAP: see k_induce_count
    bool block = i0 + IN_ITEMS <= N;
    if (REMAP) {
        // first for the whole tile (one answer per wave, no dependent loads); only a tile that holds a near-end suffix
        // asks per thread
        const uint64_t tj = (uint64_t)beg + (uint64_t)blockIdx.x * IN_TILE;
        const uint32_t t0 = remap_below_wave(rm, tj), t1 = remap_below_wave(rm, tj + IN_TILE);
        if (t0 == t1) shift = (int64_t)t0;
        else if (block) {
            const uint32_t b0 = remap_below(rm, (uint64_t)beg + i0);
            block = b0 == remap_below(rm, (uint64_t)beg + i0 + IN_ITEMS);
            shift = (int64_t)b0;
        }
    }
    if (block) {
        const int64_t p0 = (DIR > 0 ? beg + (int64_t)i0 : beg - (int64_t)(i0 + IN_ITEMS - 1)) - shift; // lowest address of my items
        uint32_t bp[IN_ITEMS], bc[IN_ITEMS];
#pragma unroll
        for (int q = 0; q < IN_ITEMS / 4; q++) {
            const U4 tp = *reinterpret_cast<const U4 *>(srcP + p0 + 4 * q);
            const U4 tcw = *reinterpret_cast<const U4 *>(srcC + p0 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                bp[4 * q + e] = tp.v[e];
                bc[4 * q + e] = tcw.v[e];
            }
        }
        // a context word that has run out of bases is refreshed from the text: rare, so ONE test for the whole wave
        // decides between the straight-line form and the one with a branch per item
        uint32_t least = 0xFFFFFFFFu;
#pragma unroll
        for (int e = 0; e < IN_ITEMS; e++) least = min(least, bc[e] & 0x7FFFFFFEu);
        static_assert(KISS_EMPTY_CTX == 1u && KISS_CTX_TAINT == 0x80000000u, "no bases <=> all bits but 0 and 31 clear");
        constexpr int last = IN_ITEMS - 1;
        if (__ballot(least == 0u) == 0ull) {
#pragma unroll
            for (int e = 0; e < IN_ITEMS; e++) {
                const int a = DIR > 0 ? e : last - e; // place of item e inside the block
                const uint32_t pc = bc[a] & 3u;
                take(e, pc, (emitmask >> pc) & 1u);
                vv[e] = bp[a];
                cc[e] = bc[a];
            }
        } else {
#pragma unroll
            for (int e = 0; e < IN_ITEMS; e++) {
                const int a = DIR > 0 ? e : last - e;
                uint32_t v = bp[a], c = bc[a];
                if (KISS_CTX_WORD(c) <= KISS_EMPTY_CTX) { // refresh path
                    const uint32_t cls = item_class<false>(pk, srcP, srcC, p0 + a, emitmask, &v, &c, rm, REMAP ? nullptr : srcK);
                    take(e, cls & 3u, cls < 4u ? 1u : 0u);
                } else {
                    const uint32_t pc = c & 3u;
                    take(e, pc, (emitmask >> pc) & 1u);
                }
                vv[e] = v;
                cc[e] = c;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < IN_ITEMS; e++) {
            const uint64_t i = i0 + (uint64_t)e;
            uint32_t cls = 4u, v = 0, c = 0;
            if (i < N) cls = item_class<REMAP>(pk, srcP, srcC, beg + (int64_t)DIR * (int64_t)i, emitmask, &v, &c, rm, REMAP ? nullptr : srcK);
            take(e, cls & 3u, cls < 4u ? 1u : 0u);
            vv[e] = v;
            cc[e] = c;
        }
    }
    // A refreshed or remapped item's class is the low two bits of its (new) context word whenever it appends, so from here
    // on: class of item e = cc[e] & 3, valid where okm has bit e.
    // wave prefixes of the four counters, as 16-bit fields (a tile has 2048 items)
    const uint32_t mylo = (cnt8 & 0xFFu) | ((cnt8 & 0xFF00u) << 8), myhi = ((cnt8 >> 16) & 0xFFu) | ((cnt8 >> 8) & 0xFF0000u);
    const uint32_t inlo = wave_scan_add(mylo), inhi = wave_scan_add(myhi);
    if (lane_id() == 63) {
        wtot[wave][0] = inlo;
        wtot[wave][1] = inhi;
    }
    __syncthreads();
    // the tile staged class by class in LDS: class k starts at coff[k]; inside it the waves in order, inside a wave the lanes
    uint32_t tlo = 0, thi = 0, olo = 0, ohi = 0; // totals of the tile / of the waves before mine
#pragma unroll
    for (int w = 0; w < IN_WAVES; w++) {
        const uint32_t a = wtot[w][0], b = wtot[w][1];
        tlo += a;
        thi += b;
        if (w < wave) {
            olo += a;
            ohi += b;
        }
    }
    const uint32_t c1 = tlo & 0xFFFFu, c2 = c1 + (tlo >> 16), c3 = c2 + (thi & 0xFFFFu), total = c3 + (thi >> 16);
    // where my first item of class k goes, again as 16-bit fields (every sum stays below 2049)
    const uint32_t blo = (c1 << 16) + olo + (inlo - mylo), bhi = (c2 | (c3 << 16)) + ohi + (inhi - myhi);
#pragma unroll
    for (int e = 0; e < IN_ITEMS; e++) {
        if ((okm >> e) & 1u) {
            const uint32_t pc = cc[e] & 3u;
            const uint32_t base = __builtin_amdgcn_perm(bhi, blo, 0x0c0c0100u + pc * 0x0202u); // field pc of bhi:blo
            const uint32_t li = base + ((loc >> (4 * e)) & 15u);
            stP[li] = vv[e] - 1u;
            stC[li] = child_ctx(cc[e]);
        }
    }
    __syncthreads();
    // write-out: a thread takes four consecutive staged items of one class and stores them with one 16-byte store
    // per array (4-byte aligned); class boundaries and the tail fall back to single stores.
    // Staged item l of class k goes to db[k] + DIR * l  (db[k] = list head + DIR * (items of class k in earlier tiles - coff[k]))
    int64_t db[4];
    {
        const uint32_t cs[4] = {0u, c1, c2, c3};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t toff = ex[(uint64_t)k * tiles + blockIdx.x] - ex[(uint64_t)k * tiles];
            db[k] = dst.p[k] + (int64_t)DIR * ((int64_t)toff - (int64_t)cs[k]);
        }
    }
    for (uint32_t l0 = threadIdx.x * 4u; l0 < total; l0 += IN_THREADS * 4u) {
        const uint32_t cls = (l0 >= c1 ? 1u : 0u) + (l0 >= c2 ? 1u : 0u) + (l0 >= c3 ? 1u : 0u);
        const uint32_t cend = cls == 0 ? c1 : (cls == 1 ? c2 : (cls == 2 ? c3 : total));
        if (l0 + 4u <= cend) {
            const int64_t dbk = cls == 0 ? db[0] : (cls == 1 ? db[1] : (cls == 2 ? db[2] : db[3]));
            const int64_t d0 = dbk + (int64_t)DIR * (int64_t)l0; // place of item l0; l0 + e at d0 + DIR * e
            const uint4 sp = *reinterpret_cast<const uint4 *>(stP + l0), sc = *reinterpret_cast<const uint4 *>(stC + l0);
            U4 wp, wc;
            if (DIR > 0) {
                wp.v[0] = sp.x, wp.v[1] = sp.y, wp.v[2] = sp.z, wp.v[3] = sp.w;
                wc.v[0] = sc.x, wc.v[1] = sc.y, wc.v[2] = sc.z, wc.v[3] = sc.w;
            } else {
                wp.v[0] = sp.w, wp.v[1] = sp.z, wp.v[2] = sp.y, wp.v[3] = sp.x;
                wc.v[0] = sc.w, wc.v[1] = sc.z, wc.v[2] = sc.y, wc.v[3] = sc.x;
            }
            const int64_t lo = DIR > 0 ? d0 : d0 - 3;
            *reinterpret_cast<U4 *>(SA + lo) = wp;
            *reinterpret_cast<U4 *>(CTX + lo) = wc;
            if (CLS) { // the four words' class bytes: one 4-byte store (byte-aligned)
                struct __attribute__((packed, aligned(1))) B4 {
                    uint32_t v;
                };
                reinterpret_cast<B4 *>(CLS + lo)->v =
                    cls_byte(wc.v[0]) | (cls_byte(wc.v[1]) << 8) | (cls_byte(wc.v[2]) << 16) | (cls_byte(wc.v[3]) << 24);
            }
        } else {
            for (uint32_t li = l0; li < l0 + 4u && li < total; li++) {
                const uint32_t c2_ = (li >= c1 ? 1u : 0u) + (li >= c2 ? 1u : 0u) + (li >= c3 ? 1u : 0u);
                const int64_t dbk = c2_ == 0 ? db[0] : (c2_ == 1 ? db[1] : (c2_ == 2 ? db[2] : db[3]));
                const int64_t d = dbk + (int64_t)DIR * (int64_t)li;
                SA[d] = stP[li];
                CTX[d] = stC[li];
                if (CLS) CLS[d] = (uint8_t)cls_byte(stC[li]);
            }
        }
    }
}

// ---- count-free form: ONE pass per source segment -------------------------------------------------------------------
// k_induce_count read the segment once only to learn how many items of each class every tile holds; a scan turned the
// counts into offsets and k_induce_scatter read the segment again.  The scatter computes the same counts on its way (it
// ranks its items by class), so a tile can find the items of each class in all EARLIER tiles by a decoupled look-back
// instead: tiles take their number from an atomic ticket (every predecessor of a running tile is running or done),
// publish their four class counts, and wave c of the workgroup sums class c over the predecessors, 64 descriptors per
// round trip, until it meets a tile that has published its running total -- the scheme of k_fc0_onepass (lms_sort.hip)
// with the descriptor of the radix passes: [status:2 | pass epoch:30 | value:32], one 64-bit word per (class, tile), never
// cleared (the epoch keeps counting).  The count kernel, the three scan launches and their 4 bytes per item are gone
// (round 3: 5.1 ms of 21 per sort); the wait is bounded (error flag, not a hung GPU).
constexpr uint32_t IN_SPIN_LIMIT = 1u << 22;
static_assert(IN_WAVES == 4, "wave c of a partition tile walks class c");

__device__ __forceinline__ uint32_t in_lookback(uint64_t *__restrict__ d, uint32_t tile, uint32_t mine, uint64_t tag,
                                                uint32_t *__restrict__ err)
{
    const uint32_t lane = lane_id();
    constexpr uint64_t TAGMASK = 0x3FFFFFFFull << 32;
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(&d[0], (2ull << 62) | tag | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0u;
    }
    if (lane == 0) __hip_atomic_store(&d[tile], (1ull << 62) | tag | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0, spins = 0;
    int64_t base = (int64_t)tile;
    for (;;) {
        const int64_t t = base - 1 - (int64_t)lane;
        uint64_t v = t >= 0 ? __hip_atomic_load(&d[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((2ull << 62) | tag);
        if ((v & TAGMASK) != tag) v = 0; // a word of an earlier pass: not published yet
        const uint32_t st = (uint32_t)(v >> 62);
        const uint64_t incl = __ballot(st == 2u), ready = __ballot(st != 0u);
        const uint32_t first = incl ? (uint32_t)__builtin_ctzll(incl) : 64u;    // nearest running total
        const uint64_t need = first >= 63u ? ~0ull : ((2ull << first) - 1ull); // lanes 0 .. first
        if ((ready & need) != need) { // a predecessor this side of it has not published yet
            if (++spins > IN_SPIN_LIMIT) {
                if (lane == 0) *err = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        uint32_t a = lane <= first ? (uint32_t)v : 0u;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) a += __shfl_xor(a, dd, 64);
        excl += a;
        if (first < 64u) break;
        base -= 64;
    }
    if (lane == 0) __hip_atomic_store(&d[tile], (2ull << 62) | tag | (uint64_t)(uint32_t)(excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

template <bool REMAP>
__global__ __launch_bounds__(IN_THREADS) void k_induce_onepass(const uint64_t *__restrict__ pk, const uint32_t *srcP,
                                                              uint32_t *srcC, int64_t beg, uint64_t N, int dir,
                                                              uint32_t emitmask, uint64_t tiles, DstPos dst, uint32_t *SA,
                                                              uint32_t *CTX, uint32_t *__restrict__ totals, LmsRemap rm,
                                                              uint64_t *__restrict__ desc, uint64_t desc_stride,
                                                              uint32_t *__restrict__ ctl, uint32_t ticket_base, uint64_t epoch)
{
    __shared__ uint32_t wtot[IN_WAVES][4];
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_excl[4];
    __shared__ uint32_t stP[IN_TILE], stC[IN_TILE];
    if (threadIdx.x == 0) s_tile = atomicAdd(&ctl[2], 1u) - ticket_base;
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= tiles) return; // (never: the grid is `tiles` workgroups)
    const int wave = threadIdx.x >> 6;
    struct __attribute__((packed, aligned(4))) U4 {
        uint32_t v[4];
    };
    const uint64_t i0 = (uint64_t)tile * IN_TILE + (uint64_t)threadIdx.x * IN_ITEMS;
    uint32_t vv[IN_ITEMS], cc[IN_ITEMS], rr[IN_ITEMS]; // rr = (class << 28) | rank in wave
    uint32_t cnt[4] = {0, 0, 0, 0};
    int64_t shift = 0; // REMAP (dir > 0 only): see k_induce_count
    bool block = i0 + IN_ITEMS <= N;
    if (REMAP) {
        const uint64_t tj = (uint64_t)beg + (uint64_t)tile * IN_TILE;
        const uint32_t t0 = remap_below_wave(rm, tj), t1 = remap_below_wave(rm, tj + IN_TILE);
        if (t0 == t1) shift = (int64_t)t0;
        else if (block) {
            const uint32_t b0 = remap_below(rm, (uint64_t)beg + i0);
            block = b0 == remap_below(rm, (uint64_t)beg + i0 + IN_ITEMS);
            shift = (int64_t)b0;
        }
    }
    if (block) {
        const int64_t p0 = (dir > 0 ? beg + (int64_t)i0 : beg - (int64_t)(i0 + IN_ITEMS - 1)) - shift; // lowest address of my items
        uint32_t bp[IN_ITEMS], bc[IN_ITEMS];
#pragma unroll
        for (int q = 0; q < IN_ITEMS / 4; q++) {
            const U4 tp = *reinterpret_cast<const U4 *>(srcP + p0 + 4 * q);
            const U4 tcw = *reinterpret_cast<const U4 *>(srcC + p0 + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                bp[4 * q + e] = tp.v[e];
                bc[4 * q + e] = tcw.v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < IN_ITEMS; e++) {
            const int a = dir > 0 ? e : IN_ITEMS - 1 - e; // place of item e inside the block
            uint32_t v = bp[a], c = bc[a], cls;
            if (KISS_CTX_WORD(c) <= KISS_EMPTY_CTX) cls = item_class<false>(pk, srcP, srcC, p0 + a, emitmask, &v, &c, rm); // refresh path
            else {
                const uint32_t pc = c & 3u;
                cls = ((emitmask >> pc) & 1u) ? pc : 4u;
            }
            uint32_t local = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (cls == (uint32_t)k) local = cnt[k];
                cnt[k] += cls == (uint32_t)k ? 1u : 0u;
            }
            vv[e] = v;
            cc[e] = c;
            rr[e] = (cls << 28) | local;
        }
    } else {
#pragma unroll
        for (int e = 0; e < IN_ITEMS; e++) {
            const uint64_t i = i0 + (uint64_t)e;
            uint32_t cls = 4u, v = 0, c = 0;
            if (i < N) cls = item_class<REMAP>(pk, srcP, srcC, beg + (int64_t)dir * (int64_t)i, emitmask, &v, &c, rm);
            uint32_t local = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (cls == (uint32_t)k) local = cnt[k];
                cnt[k] += cls == (uint32_t)k ? 1u : 0u;
            }
            vv[e] = v;
            cc[e] = c;
            rr[e] = (cls << 28) | local;
        }
    }
    uint32_t run[4]; // wave totals
    uint32_t lex[4]; // items of class k in lower lanes
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t inc = cnt[k];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if ((int)lane_id() >= d) inc += o;
        }
        lex[k] = inc - cnt[k];
        run[k] = __shfl(inc, 63, 64);
    }
#pragma unroll
    for (int e = 0; e < IN_ITEMS; e++) {
        const uint32_t cls = rr[e] >> 28;
        if (cls < 4u) rr[e] += cls == 0 ? lex[0] : (cls == 1 ? lex[1] : (cls == 2 ? lex[2] : lex[3]));
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int c = 0; c < 4; c++) wtot[wave][c] = run[c];
    }
    __syncthreads();
    uint32_t coff[5]; // start of class c inside the staged tile
    uint32_t woff[4]; // this wave's start inside class c
    coff[0] = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint32_t t = 0, w0 = 0;
        for (int w = 0; w < IN_WAVES; w++) {
            if (w < wave) w0 += wtot[w][c];
            t += wtot[w][c];
        }
        woff[c] = w0;
        coff[c + 1] = coff[c] + t;
    }
    // wave c: items of class c in all earlier tiles (the other waves' round trips run at the same time; the staging
    // below does not need the answer)
    {
        const uint32_t mine = wave == 0 ? coff[1] - coff[0] : (wave == 1 ? coff[2] - coff[1] : (wave == 2 ? coff[3] - coff[2] : coff[4] - coff[3]));
        const uint64_t tag = (epoch & 0x3FFFFFFFull) << 32;
        const uint32_t ex = in_lookback(desc + (uint64_t)wave * desc_stride, tile, mine, tag, ctl + 1);
        if (lane_id() == 0) {
            s_excl[wave] = ex;
            if ((uint64_t)tile + 1 == tiles) totals[wave] = ex + mine;
        }
    }
    // stage the tile class by class in LDS, then write every class as one contiguous run
#pragma unroll
    for (int j = 0; j < IN_ITEMS; j++) {
        const uint32_t cls = rr[j] >> 28;
        if (cls < 4u) {
            const uint32_t li = coff[cls] + woff[cls] + (rr[j] & 0x0FFFFFFFu);
            stP[li] = vv[j] - 1u;
            stC[li] = child_ctx(cc[j]);
        }
    }
    __syncthreads();
    const uint32_t total = coff[4];
    for (uint32_t l0 = threadIdx.x * 4u; l0 < total; l0 += IN_THREADS * 4u) {
        const uint32_t cls = l0 < coff[1] ? 0u : (l0 < coff[2] ? 1u : (l0 < coff[3] ? 2u : 3u));
        const uint32_t cend = cls == 0 ? coff[1] : (cls == 1 ? coff[2] : (cls == 2 ? coff[3] : coff[4]));
        if (l0 + 4u <= cend) {
            const uint32_t toff = s_excl[cls];
            const uint32_t cstart = cls == 0 ? coff[0] : (cls == 1 ? coff[1] : (cls == 2 ? coff[2] : coff[3]));
            const int64_t dp = cls == 0 ? dst.p[0] : (cls == 1 ? dst.p[1] : (cls == 2 ? dst.p[2] : dst.p[3]));
            const int64_t d0 = dp + (int64_t)dir * (int64_t)((uint64_t)toff + (l0 - cstart)); // place of item l0; l0 + e at d0 + dir * e
            U4 wp, wc;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int a = dir > 0 ? e : 3 - e;
                wp.v[a] = stP[l0 + e];
                wc.v[a] = stC[l0 + e];
            }
            const int64_t lo = dir > 0 ? d0 : d0 - 3;
            *reinterpret_cast<U4 *>(SA + lo) = wp;
            *reinterpret_cast<U4 *>(CTX + lo) = wc;
        } else {
            for (uint32_t li = l0; li < l0 + 4u && li < total; li++) {
                const uint32_t c2 = li < coff[1] ? 0u : (li < coff[2] ? 1u : (li < coff[3] ? 2u : 3u));
                const uint32_t toff = s_excl[c2];
                const uint32_t cstart = c2 == 0 ? coff[0] : (c2 == 1 ? coff[1] : (c2 == 2 ? coff[2] : coff[3]));
                const int64_t dp = c2 == 0 ? dst.p[0] : (c2 == 1 ? dst.p[1] : (c2 == 2 ? dst.p[2] : dst.p[3]));
                const int64_t d = dp + (int64_t)dir * (int64_t)((uint64_t)toff + (li - cstart));
                SA[d] = stP[li];
                CTX[d] = stC[li];
            }
        }
    }
}

// ---- single-workgroup chain kernel -----------------------------------------------------
// Processes one source segment and, when selfclass >= 0, keeps processing what it appended to that
// class until nothing is appended any more.  out[0..3] = items appended per class, out[4] = rounds.
template <bool REMAP> // REMAP: one round over the LMS list (selfclass < 0)
__global__ __launch_bounds__(SM_THREADS) void k_induce_small(const uint64_t *__restrict__ pk, const uint32_t *srcP0,
                                                            uint32_t *srcC0, int64_t beg0, uint64_t N0, int dir,
                                                            uint32_t emitmask, int selfclass, DstPos dst, uint32_t *SA,
                                                            uint32_t *CTX, uint32_t *out, LmsRemap rm, uint8_t *CLS)
{
    __shared__ int64_t heads[4];
    __shared__ uint32_t wtot[SM_THREADS / 64][4];
    __shared__ uint32_t chunk_tot[4];
    __shared__ uint32_t round_self;
    if (threadIdx.x < 4) heads[threadIdx.x] = dst.p[threadIdx.x];
    if (threadIdx.x == 0) round_self = 0;
    __syncthreads();

    const uint32_t *srcP = srcP0;
    uint32_t *srcC = srcC0;
    int64_t beg = beg0;
    uint64_t N = N0;
    uint32_t rounds = 0;
    const int wave = threadIdx.x >> 6;

    while (N > 0) {
        rounds++;
        const int64_t self_head_at_start = selfclass >= 0 ? heads[selfclass] : 0;
        __syncthreads();
        for (uint64_t cb = 0; cb < N; cb += SM_THREADS) {
            uint64_t i = cb + threadIdx.x;
            uint32_t cls = 4u, v = 0, cw = 0;
            if (i < N) cls = item_class<REMAP>(pk, srcP, srcC, beg + (int64_t)dir * (int64_t)i, emitmask, &v, &cw, rm);
            uint32_t myrank = 0;
            uint32_t tot[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                uint64_t mk = __ballot(cls == (uint32_t)c);
                if (cls == (uint32_t)c) myrank = (uint32_t)__popcll(mk & lanemask_lt());
                tot[c] = (uint32_t)__popcll(mk);
            }
            if (lane_id() == 0) {
#pragma unroll
                for (int c = 0; c < 4; c++) wtot[wave][c] = tot[c];
            }
            __syncthreads();
            if (cls < 4u) {
                uint32_t o = myrank;
                for (int w = 0; w < wave; w++) o += wtot[w][cls];
                int64_t d = heads[cls] + (int64_t)dir * (int64_t)o;
                SA[d] = v - 1u;
                CTX[d] = child_ctx(cw);
                if (CLS) CLS[d] = (uint8_t)cls_byte(child_ctx(cw));
            }
            if (threadIdx.x < 4) {
                uint32_t t = 0;
                for (int w = 0; w < SM_THREADS / 64; w++) t += wtot[w][threadIdx.x];
                chunk_tot[threadIdx.x] = t;
            }
            __syncthreads();
            if (threadIdx.x < 4) heads[threadIdx.x] += (int64_t)dir * (int64_t)chunk_tot[threadIdx.x];
            if (threadIdx.x == 0 && selfclass >= 0) round_self += chunk_tot[selfclass];
            __syncthreads();
        }
        if (selfclass < 0) break;
        // next round = what this round appended to the self class
        uint64_t nextN = round_self;
        __syncthreads();
        if (threadIdx.x == 0) round_self = 0;
        srcP = SA;
        srcC = CTX;
        beg = self_head_at_start;
        N = nextN;
        __syncthreads();
    }
    if (threadIdx.x < 4) out[threadIdx.x] = (uint32_t)((heads[threadIdx.x] - dst.p[threadIdx.x]) * (int64_t)dir);
    if (threadIdx.x == 0) out[4] = rounds;
}

struct Sweep {
    kiss_hip_ctx *ctx;
    uint32_t *SA;
    int dir;
    int64_t pos[4]; // next write index per class
};

// one source segment; returns items appended per class in tot[4]
int run_pass(Sweep &sw, const uint32_t *srcP, uint32_t *srcC, int64_t beg, uint64_t N, uint32_t emitmask,
             int selfclass, uint64_t tot[4], bool *chain_done, const LmsRemap *remap = nullptr)
{
    kiss_hip_ctx *ctx = sw.ctx;
    const LmsRemap rm = remap ? *remap : LmsRemap{nullptr, nullptr, 0u};
    if (remap && (sw.dir < 0 || selfclass >= 0)) return KINTERNAL();
    for (int c = 0; c < 4; c++) tot[c] = 0;
    *chain_done = false;
    if (N == 0 || emitmask == 0) {
        *chain_done = true;
        return KISS_HIP_OK;
    }
    DstPos dp;
    for (int c = 0; c < 4; c++) dp.p[c] = sw.pos[c];
    ctx->stats.induce_passes++;
    // class bytes of the source words: only the arrays parallel to SA have them (every kernel that writes a context word
    // there writes its byte: k_induce_scatter, k_induce_small, k_chain_write, k_chain_term_write); the hooks build can
    // switch their use off, and does so with the one-pass form, whose kernel does not write them
    uint8_t *const srcK = (srcC == ctx->CTX && ctx->CLS && !ctx->opts.no_class_bytes && !ctx->opts.induce_one_pass) ? ctx->CLS : nullptr;
    const uint64_t small_max = ctx->opts.induce_small_max >= 64 && ctx->opts.induce_small_max <= (1u << 20) ? ctx->opts.induce_small_max
                                                                                                           : (uint64_t)SMALL_MAX;
    if (N <= small_max) {
        {
            KTimer t(ctx, KISS_HIP_K_INDUCE_SMALL, N);
            if (remap)
                hipLaunchKernelGGL(k_induce_small<true>, dim3(1), dim3(SM_THREADS), 0, ctx->stream, ctx->pk, srcP, srcC, beg, N,
                                   sw.dir, emitmask, selfclass, dp, sw.SA, ctx->CTX, ctx->d_small, rm, ctx->CLS);
            else
                hipLaunchKernelGGL(k_induce_small<false>, dim3(1), dim3(SM_THREADS), 0, ctx->stream, ctx->pk, srcP, srcC, beg, N,
                                   sw.dir, emitmask, selfclass, dp, sw.SA, ctx->CTX, ctx->d_small, rm, ctx->CLS);
            KCHECK(hipGetLastError());
        }
        KTRY(kiss_readback(ctx, ctx->d_small, 5));
        for (int c = 0; c < 4; c++) tot[c] = ctx->h_pinned[c];
        *chain_done = true;
    } else {
        const uint64_t tiles = div_up(N, IN_TILE);
        if (4 * tiles + 1 > ctx->ind_tiles_cap) return KINTERNAL();
#ifdef KISS_HIP_HOOKS
        if (ctx->opts.induce_one_pass && ctx->ind_desc && tiles <= ctx->ind_desc_stride && tiles < (1ull << 31)) {
            // (hooks build, measured and dropped in round 4: DESIGN.md 4) one pass, the tiles find their offsets by look-back
            KTimer t(ctx, KISS_HIP_K_INDUCE_SCATTER, N);
            ctx->ind_epoch++;
            if (remap)
                hipLaunchKernelGGL(k_induce_onepass<true>, dim3((unsigned)tiles), dim3(IN_THREADS), 0, ctx->stream, ctx->pk, srcP,
                                   srcC, beg, N, sw.dir, emitmask, tiles, dp, sw.SA, ctx->CTX, ctx->d_small, rm, ctx->ind_desc,
                                   ctx->ind_desc_stride, ctx->rx_ctl, ctx->ind_ticket_base, ctx->ind_epoch);
            else
                hipLaunchKernelGGL(k_induce_onepass<false>, dim3((unsigned)tiles), dim3(IN_THREADS), 0, ctx->stream, ctx->pk, srcP,
                                   srcC, beg, N, sw.dir, emitmask, tiles, dp, sw.SA, ctx->CTX, ctx->d_small, rm, ctx->ind_desc,
                                   ctx->ind_desc_stride, ctx->rx_ctl, ctx->ind_ticket_base, ctx->ind_epoch);
            KCHECK(hipGetLastError());
            ctx->ind_ticket_base += (uint32_t)tiles;
        } else
#endif
        {
        {
            KTimer t(ctx, KISS_HIP_K_INDUCE_COUNT, N);
            if (remap)
                hipLaunchKernelGGL(k_induce_count<true>, dim3((unsigned)div_up(tiles, IN_WAVES)), dim3(IN_THREADS), 0, ctx->stream,
                                   ctx->pk, srcP, srcC, beg, N, sw.dir, emitmask, ctx->ind_counts, tiles, rm, srcK);
            else
                hipLaunchKernelGGL(k_induce_count<false>, dim3((unsigned)div_up(tiles, IN_WAVES)), dim3(IN_THREADS), 0, ctx->stream,
                                   ctx->pk, srcP, srcC, beg, N, sw.dir, emitmask, ctx->ind_counts, tiles, rm, srcK);
            KCHECK(hipGetLastError());
        }
        KTRY(kiss_scan_u32(ctx, ctx->ind_counts, ctx->ind_counts, 4 * tiles + 1));
        {
            KTimer t(ctx, KISS_HIP_K_INDUCE_SCATTER, N);
            if (remap)
                hipLaunchKernelGGL((k_induce_scatter<true, 1>), dim3((unsigned)tiles), dim3(IN_THREADS), 0, ctx->stream, ctx->pk, srcP,
                                   srcC, beg, N, emitmask, ctx->ind_counts, tiles, dp, sw.SA, ctx->CTX, ctx->d_small, rm, ctx->CLS, srcK);
            else if (sw.dir > 0)
                hipLaunchKernelGGL((k_induce_scatter<false, 1>), dim3((unsigned)tiles), dim3(IN_THREADS), 0, ctx->stream, ctx->pk, srcP,
                                   srcC, beg, N, emitmask, ctx->ind_counts, tiles, dp, sw.SA, ctx->CTX, ctx->d_small, rm, ctx->CLS, srcK);
            else
                hipLaunchKernelGGL((k_induce_scatter<false, -1>), dim3((unsigned)tiles), dim3(IN_THREADS), 0, ctx->stream, ctx->pk, srcP,
                                   srcC, beg, N, emitmask, ctx->ind_counts, tiles, dp, sw.SA, ctx->CTX, ctx->d_small, rm, ctx->CLS, srcK);
            KCHECK(hipGetLastError());
        }
        }
        KTRY(kiss_readback(ctx, ctx->d_small, 4));
        for (int c = 0; c < 4; c++) tot[c] = ctx->h_pinned[c];
    }
    if (ctx->opts.debug)
        fprintf(stderr, "[kiss_hip] pass dir=%d beg=%lld N=%llu mask=%x self=%d -> %llu,%llu,%llu,%llu\n", sw.dir, (long long)beg,
                (unsigned long long)N, emitmask, selfclass, (unsigned long long)tot[0], (unsigned long long)tot[1],
                (unsigned long long)tot[2], (unsigned long long)tot[3]);
    for (int c = 0; c < 4; c++) sw.pos[c] += (int64_t)sw.dir * (int64_t)tot[c];
    return KISS_HIP_OK;
}

// ---- chain collapse ---------------------------------------------------------------------------------------
// A segment that appends to itself (L-part(c) feeding L-part(c): runs of one base) is a chain of rounds,
// round t+1 being the items of round t that are preceded by base c again.  Once a round is small the rest
// of the chain is computed in closed form instead of round by round:
//   every item v of the current round has a run length r(v) = number of c's right before v (capped at R);
//   it contributes v-1 .. v-r to rounds 1 .. r, and the round-t block holds the items with r >= t in their
//   original order  ->  a stable sort of the pairs (t, item) by t (two 8-bit radix passes);
//   the last element of a chain appends its predecessor to another bucket during round r(v), so those
//   terminal items are ordered by (r, item) inside their bucket  ->  one more small stable sort.
// Items whose run reaches the cap R continue from the last block in the next call.
// rounds of at most this many items end in the closed form.  Round 3 sweep at chm13 size (induction ms per sort): 16 K 23.7,
// 32 K 21.4, 64 K 21.2, 128 K 21.2, 256 K 21.2, 512 K 21.2, 1 M 22.2, 2 M (rounds 1-2) 23.2, 4 M 24.9, 8 M 27.1 -- the closed
// form pays a scan, an expansion and two small radix sorts per call, a plain round of that size a count and a scatter
constexpr uint64_t COLLAPSE_N = 1ull << 18;
constexpr uint32_t COLLAPSE_RCAP = 65535;
constexpr uint32_t COLLAPSE_RCAP_WIDE = (1u << 22) - 1; // runs of one base longer than 65535: wider steps
constexpr int CH_THREADS = 256;

// number of c's right before position v, capped.  Lane-serial for the first RUN_SERIAL bases; a lane whose run is
// still going after that is served by its whole wave (64 packed words = 2048 bases per coalesced step), so one
// multi-megabase run of a single base does not serialise on one lane.  Must be called by all lanes of a wave.
constexpr uint32_t RUN_SERIAL = 1024;

__device__ __forceinline__ uint32_t run_before(const uint64_t *__restrict__ pk, uint64_t v, uint32_t c, uint32_t cap,
                                               bool valid)
{
    const uint64_t pat = (uint64_t)c * 0x5555555555555555ull;
    const uint32_t scap = cap < RUN_SERIAL ? cap : RUN_SERIAL;
    uint32_t r = 0;
    uint64_t p = valid ? v : 0; // bases before v still to test: indices < p
    bool going = false;
    while (p > 0) {
        const uint64_t q = p - 1;
        const uint32_t in = (uint32_t)(q & 31u) + 1u; // bases of this word with index <= q
        const uint64_t y = (pk[q >> 5] ^ pat) >> (62u - 2u * (uint32_t)(q & 31u)); // base q in bits 1:0, q-1 in 3:2, ...
        uint32_t mt = y ? (uint32_t)(__ffsll((unsigned long long)y) - 1) >> 1 : 32u;
        if (mt > in) mt = in;
        r += mt;
        p -= mt;
        if (mt < in) break;
        if (r >= scap) {
            going = r < cap && p > 0; // p is a multiple of 32 here
            break;
        }
    }
    uint64_t todo = __ballot(going);
    while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        uint64_t lp = __shfl(p, leader, 64);
        uint32_t lr = __shfl(r, leader, 64);
        const uint32_t lcap = __shfl(cap, leader, 64);
        while (true) {
            const int64_t w = (int64_t)(lp >> 5) - 1 - (int64_t)lane_id();
            const uint64_t y = w >= 0 ? (pk[w] ^ pat) : 1ull; // before the text: stop, nothing matched
            const uint64_t stop = __ballot(y != 0);
            if (!stop) {
                lr += 2048u;
                lp -= 2048u;
                if (lr >= lcap) break;
                continue;
            }
            const int f = __ffsll((unsigned long long)stop) - 1;
            const uint64_t yf = __shfl(y, f, 64);
            const bool outside = ((int64_t)(lp >> 5) - 1 - f) < 0;
            lr += 32u * (uint32_t)f + (outside ? 0u : ((uint32_t)(__ffsll((unsigned long long)yf) - 1) >> 1));
            break;
        }
        if ((int)lane_id() == leader) r = lr;
    }
    return r > cap ? cap : r;
}

// counters: [0..3] terminal items per class, [4] items whose run hit the cap
__global__ __launch_bounds__(CH_THREADS) void k_chain_runs(const uint64_t *__restrict__ pk, const uint32_t *srcP,
                                                          const uint32_t *srcC, int64_t beg, uint64_t N, int dir, uint32_t c,
                                                          uint32_t cap,
                                                          int rshift, uint32_t termmask, uint32_t *__restrict__ run,
                                                          uint64_t *__restrict__ tkey, uint32_t *__restrict__ tpos,
                                                          uint32_t *__restrict__ counters)
{
    uint64_t i = (uint64_t)blockIdx.x * CH_THREADS + threadIdx.x;
    uint32_t cls = 4u;
    bool capped = false;
    const uint64_t v = i < N ? srcP[beg + (int64_t)dir * (int64_t)i] : 0;
    const uint32_t r = run_before(pk, v, c, cap, i < N);
    if (i < N) {
        run[i] = r;
        uint32_t u = 0;
        if (r >= cap) capped = true;
        else if (v - r >= 1) {
            u = (uint32_t)(v - r - 1);
            uint32_t x = kiss_base(pk, u);
            if ((termmask >> x) & 1u) cls = x;
        }
        // bit 0: the taint of the item the terminal descends from (payload, below the sorted bits)
        const uint64_t tn = srcC[beg + (int64_t)dir * (int64_t)i] >> 31;
        tkey[i] = cls < 4u ? (((uint64_t)cls << 62) | ((uint64_t)r << rshift) | tn) : ~0ull;
        tpos[i] = u;
    }
#pragma unroll
    for (uint32_t x = 0; x < 4; x++) {
        uint64_t mk = __ballot(cls == x);
        if (mk && lane_id() == 0) atomicAdd(&counters[x], (uint32_t)__popcll(mk));
    }
    uint64_t mk = __ballot(capped);
    if (mk && lane_id() == 0) atomicAdd(&counters[4], (uint32_t)__popcll(mk));
}

__global__ void k_chain_total(const uint32_t *__restrict__ run, const uint32_t *__restrict__ ex, uint64_t N,
                              uint32_t *__restrict__ counters)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) counters[5] = ex[N - 1] + run[N - 1];
}

__global__ __launch_bounds__(CH_THREADS) void k_chain_expand(const uint32_t *srcP, const uint32_t *srcC, int64_t beg, int dir,
                                                            uint64_t N,
                                                            const uint32_t *__restrict__ ex, uint64_t E, int tshift,
                                                            uint64_t *__restrict__ key, uint32_t *__restrict__ pos)
{
    uint64_t e = (uint64_t)blockIdx.x * CH_THREADS + threadIdx.x;
    if (e >= E) return;
    // last item i with ex[i] <= e
    uint64_t lo = 0, hi = N;
    while (hi - lo > 1) {
        uint64_t mid = (lo + hi) >> 1;
        if (ex[mid] <= e) lo = mid;
        else hi = mid;
    }
    const uint32_t t = (uint32_t)(e - ex[lo]) + 1u;
    const int64_t src = beg + (int64_t)dir * (int64_t)lo;
    key[e] = ((uint64_t)t << tshift) | (uint64_t)(srcC[src] >> 31); // bit 0: the source item's taint (payload)
    pos[e] = srcP[src] - t;
}

__global__ __launch_bounds__(CH_THREADS) void k_chain_write(const uint64_t *__restrict__ pk,
                                                           const uint64_t *__restrict__ sorted_key,
                                                           const uint32_t *__restrict__ sorted_pos, uint64_t E,
                                                           int64_t dst, int dir, uint32_t *SA, uint32_t *CTX, uint8_t *CLS)
{
    uint64_t e = (uint64_t)blockIdx.x * CH_THREADS + threadIdx.x;
    if (e >= E) return;
    const uint32_t u = sorted_pos[e];
    const uint32_t cw = kiss_load_ctx(pk, u) | ((uint32_t)(sorted_key[e] & 1ull) << 31);
    const int64_t d = dst + (int64_t)dir * (int64_t)e;
    SA[d] = u;
    CTX[d] = cw;
    if (CLS) CLS[d] = (uint8_t)cls_byte(cw);
}

struct TermDst {
    int64_t p[4];
    uint32_t off[5];
};

__global__ __launch_bounds__(CH_THREADS) void k_chain_term_write(const uint64_t *__restrict__ pk,
                                                                const uint64_t *__restrict__ sorted_tkey,
                                                                const uint32_t *__restrict__ sorted_tpos, uint64_t T,
                                                                TermDst td, int dir, uint32_t *SA, uint32_t *CTX, uint8_t *CLS)
{
    uint64_t j = (uint64_t)blockIdx.x * CH_THREADS + threadIdx.x;
    if (j >= T) return;
    uint32_t x = 0;
    while (x < 3 && j >= td.off[x + 1]) x++;
    const uint32_t u = sorted_tpos[j];
    const int64_t d = td.p[x] + (int64_t)dir * (int64_t)(j - td.off[x]);
    SA[d] = u;
    const uint32_t cw = kiss_load_ctx(pk, u) | ((uint32_t)(sorted_tkey[j] & 1ull) << 31);
    CTX[d] = cw;
    if (CLS) CLS[d] = (uint8_t)cls_byte(cw);
}

int verify_part(kiss_hip_ctx *ctx, const uint32_t *SA, int64_t lo, int64_t hi, uint32_t c, uint64_t n, const char *what);
int checksum_pk(kiss_hip_ctx *ctx, const char *what);

// collapses the chain that starts with the N items at [beg, beg + dir*N) of SA (self class c)
int run_collapse(Sweep &sw, uint32_t c, int64_t beg, uint64_t N, uint32_t termmask)
{
    kiss_hip_ctx *ctx = sw.ctx;
    uint32_t *run = ctx->slotA, *ex = ctx->segA, *tpos = ctx->bposA;
    uint64_t *tkey = ctx->bkeyA;
    uint32_t *counters = ctx->d_small + 16;
    bool wide = false; // after a call in which runs hit the 16-bit cap: 24-bit step field (one more radix pass)
    while (N > 0) {
        const uint64_t capq = ctx->m_cap / N;
        const uint32_t rcap = wide ? COLLAPSE_RCAP_WIDE : COLLAPSE_RCAP;
        const int tshift = wide ? 40 : 48; // step t in key bits [tshift, 64)
        const int rshift = tshift - 2;     // terminal key: class in bits 62..63, run in [rshift, 62)
        uint32_t cap = (uint32_t)(capq > rcap ? rcap : (capq < 1 ? 1 : capq));
        if (ctx->opts.collapse_cap >= 1 && ctx->opts.collapse_cap < cap) cap = ctx->opts.collapse_cap; // (hooks build: force short steps)
        const unsigned grid = (unsigned)div_up(N, CH_THREADS);
        ctx->stats.induce_passes++;
        {
            KTimer t(ctx, KISS_HIP_K_INDUCE_SMALL, N);
            KTRY(kiss_zero_u32(ctx, counters, 8));
            hipLaunchKernelGGL(k_chain_runs, dim3(grid), dim3(CH_THREADS), 0, ctx->stream, ctx->pk, sw.SA, ctx->CTX, beg, N,
                               sw.dir, c, cap, rshift, termmask & ~(1u << c), run, tkey, tpos, counters);
            KCHECK(hipGetLastError());
        }
        KTRY(kiss_scan_u32(ctx, run, ex, N));
        hipLaunchKernelGGL(k_chain_total, dim3(1), dim3(64), 0, ctx->stream, run, ex, N, counters);
        KTRY(kiss_readback(ctx, counters, 6));
        uint32_t cnt[4];
        for (int x = 0; x < 4; x++) cnt[x] = ctx->h_pinned[x];
        const uint64_t ncap = ctx->h_pinned[4];
        const uint64_t E = ctx->h_pinned[5];
        if (ctx->opts.debug)
            fprintf(stderr, "[kiss_hip] collapse dir=%d c=%u beg=%lld N=%llu cap=%u E=%llu ncap=%llu term=%u,%u,%u,%u\n", sw.dir, c,
                    (long long)beg, (unsigned long long)N, cap, (unsigned long long)E, (unsigned long long)ncap, cnt[0], cnt[1],
                    cnt[2], cnt[3]);
        if (E > ctx->m_cap) return KINTERNAL();
        const int64_t dst = sw.pos[c];
        if (E > 0) {
            KTimer t(ctx, KISS_HIP_K_INDUCE_SMALL, E);
            hipLaunchKernelGGL(k_chain_expand, dim3((unsigned)div_up(E, CH_THREADS)), dim3(CH_THREADS), 0, ctx->stream,
                               sw.SA, ctx->CTX, beg, sw.dir, N, ex, E, tshift, ctx->keyA, ctx->posA);
            KCHECK(hipGetLastError());
        }
        if (E > 0) {
            RadixBufs rb;
            rb.key[0] = ctx->keyA;
            rb.key[1] = ctx->keyB;
            rb.pos[0] = ctx->posA;
            rb.pos[1] = ctx->posB;
            rb.seg[0] = rb.seg[1] = nullptr;
            int res = 0;
            KTRY(kiss_radix_sort(ctx, rb, E, tshift, 0, &res)); // step index t sits in bits tshift..63
            KTimer t(ctx, KISS_HIP_K_INDUCE_SMALL, E);
            hipLaunchKernelGGL(k_chain_write, dim3((unsigned)div_up(E, CH_THREADS)), dim3(CH_THREADS), 0, ctx->stream,
                               ctx->pk, rb.key[res], rb.pos[res], E, dst, sw.dir, sw.SA, ctx->CTX, ctx->CLS);
            KCHECK(hipGetLastError());
            if (ctx->opts.verify) {
                int64_t lo = sw.dir > 0 ? dst : dst - (int64_t)E + 1, hi = sw.dir > 0 ? dst + (int64_t)E : dst + 1;
                KTRY(verify_part(ctx, sw.SA, lo, hi, c, ctx->n, "collapse block"));
                KTRY(checksum_pk(ctx, "after collapse block"));
            }
        }
        const uint64_t T = (uint64_t)cnt[0] + cnt[1] + cnt[2] + cnt[3];
        if (T > 0) {
            RadixBufs tb;
            tb.key[0] = ctx->bkeyA;
            tb.key[1] = ctx->bkeyB;
            tb.pos[0] = ctx->bposA;
            tb.pos[1] = ctx->bposB;
            tb.seg[0] = tb.seg[1] = nullptr;
            int res = 0;
            KTRY(kiss_radix_sort(ctx, tb, N, rshift, 0, &res)); // (class, run) ; items without a terminal sort last
            TermDst td;
            uint32_t o = 0;
            for (int x = 0; x < 4; x++) {
                td.p[x] = sw.pos[x];
                td.off[x] = o;
                o += cnt[x];
            }
            td.off[4] = o;
            KTimer t(ctx, KISS_HIP_K_INDUCE_SMALL, T);
            hipLaunchKernelGGL(k_chain_term_write, dim3((unsigned)div_up(T, CH_THREADS)), dim3(CH_THREADS), 0,
                               ctx->stream, ctx->pk, tb.key[res], tb.pos[res], T, td, sw.dir, sw.SA, ctx->CTX, ctx->CLS);
            KCHECK(hipGetLastError());
        }
        for (int x = 0; x < 4; x++) sw.pos[x] += (int64_t)sw.dir * (int64_t)cnt[x];
        sw.pos[c] += (int64_t)sw.dir * (int64_t)E;
        // items whose run reached the cap continue from the last block
        beg = dst + (int64_t)sw.dir * (int64_t)(E - ncap);
        N = ncap;
        wide = cap >= COLLAPSE_RCAP;
    }
    return KISS_HIP_OK;
}

__global__ void k_checksum(const uint64_t *__restrict__ pk, uint64_t words, unsigned long long *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    for (; i < words; i += (uint64_t)gridDim.x * blockDim.x) v += pk[i] * (2 * i + 1);
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane_id() == 0) atomicAdd(out, v);
}
int checksum_pk(kiss_hip_ctx *ctx, const char *what)
{
    unsigned long long *d = (unsigned long long *)(ctx->d_small + 60);
    KTRY(kiss_zero_u32(ctx, d, 2));
    hipLaunchKernelGGL(k_checksum, dim3(256), dim3(256), 0, ctx->stream, ctx->pk, ctx->n / 32 + 2, d);
    unsigned long long h = 0;
    KCHECK(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));
    fprintf(stderr, "[kiss_hip] pk checksum %016llx at %s\n", h, what);
    return KISS_HIP_OK;
}

// debug: every SA entry of [lo, hi) must start with base c and carry a context word consistent with the text
__global__ void k_verify_part(const uint64_t *__restrict__ pk, const uint32_t *SA, const uint32_t *CTX, int64_t lo,
                              int64_t hi, uint32_t c, uint64_t n, uint32_t *bad)
{
    int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    uint32_t v = SA[i], cw = KISS_CTX_WORD(CTX[i]);
    bool ok = v < n && kiss_base(pk, v) == c;
    if (ok && cw != KISS_EMPTY_CTX) {
        uint32_t full = kiss_load_ctx(pk, v);
        int mb = 31 - __clz(cw); // marker bit
        uint32_t mask = (1u << mb) - 1u;
        ok = (mb % 2 == 0) && mb <= 30 && ((full & mask) == (cw & mask)) && (31 - __clz(full)) >= mb;
    }
    if (!ok) {
        uint32_t k = atomicAdd(&bad[0], 1u);
        if (k < 6) {
            bad[1 + 3 * k] = (uint32_t)i;
            bad[2 + 3 * k] = v;
            bad[3 + 3 * k] = cw;
        }
    }
}

int verify_part(kiss_hip_ctx *ctx, const uint32_t *SA, int64_t lo, int64_t hi, uint32_t c, uint64_t n, const char *what)
{
    if (hi <= lo) return KISS_HIP_OK;
    uint32_t *bad = ctx->d_small + 40;
    KTRY(kiss_zero_u32(ctx, bad, 20));
    hipLaunchKernelGGL(k_verify_part, dim3((unsigned)div_up((uint64_t)(hi - lo), 256)), dim3(256), 0, ctx->stream, ctx->pk,
                       SA, ctx->CTX, lo, hi, c, n, bad);
    uint32_t h[20];
    KCHECK(hipMemcpyAsync(h, bad, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));
    fprintf(stderr, "[kiss_hip] verify %s c=%u [%lld,%lld): bad %u", what, c, (long long)lo, (long long)hi, h[0]);
    for (uint32_t k = 0; k < h[0] && k < 6; k++) fprintf(stderr, " (idx %u v %u ctx %08x)", h[1 + 3 * k], h[2 + 3 * k], h[3 + 3 * k]);
    fprintf(stderr, "\n");
    return KISS_HIP_OK;
}

} // namespace

int kiss_induce(kiss_hip_ctx *ctx, uint64_t n, uint32_t *d_SA)
{
    ctx->stats.induce_passes = 0;
    KTRY(kiss_need_ctx_words(ctx));
    ctx->ctx_words_valid = false;
    uint64_t cnt[4], cntS[4], cntL[4], cntLMS[4], start[5], lms_start[5];
    start[0] = 1;
    lms_start[0] = 0;
    for (int c = 0; c < 4; c++) {
        cnt[c] = ctx->counts[c];
        cntS[c] = ctx->counts[4 + c];
        cntLMS[c] = ctx->counts[8 + c];
        cntL[c] = cnt[c] - cntS[c];
        start[c + 1] = start[c] + cnt[c];
        lms_start[c + 1] = lms_start[c] + cntLMS[c];
    }
    if (start[4] != n + 1) return KINTERNAL();

    // SA[0] = n (the sentinel suffix); a one-item source {n} with an empty context word starts the L sweep
    uint32_t *seed = ctx->d_small + 32;
    ctx->h_pinned[32] = (uint32_t)n;
    ctx->h_pinned[33] = KISS_EMPTY_CTX;
    KCHECK(hipMemcpyAsync(seed, ctx->h_pinned + 32, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    KCHECK(hipMemcpyAsync(d_SA, ctx->h_pinned + 32, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));

    uint64_t tot[4];
    bool done;
    if (ctx->opts.verify) KTRY(checksum_pk(ctx, "induce start"));
    // rounds of at most this many items end in the closed form (hooks build: swept)
    const uint64_t collapse_n = ctx->opts.collapse_n >= 1024 && ctx->opts.collapse_n <= (1ull << 26) ? ctx->opts.collapse_n : (uint64_t)COLLAPSE_N;
    uint64_t collapse_max = collapse_n < ctx->m_cap ? collapse_n : ctx->m_cap; // chain scratch: m_cap- and t_cap-sized arrays
    if (collapse_max > ctx->t_cap) collapse_max = ctx->t_cap;

    const LmsRemap lms_rm{ctx->rm_fin, ctx->rm_pos, ctx->rm_E};
    // ---------------- L sweep: left to right ----------------
    Sweep L{ctx, d_SA, +1, {(int64_t)start[0], (int64_t)start[1], (int64_t)start[2], (int64_t)start[3]}};
    KTRY(run_pass(L, seed, seed + 1, 0, 1, 0xFu, -1, tot, &done));
    for (int c = 0; c < 4; c++) {
        int64_t a = (int64_t)start[c];
        const uint32_t mask_ge = (0xFu << c) & 0xFu; // L-type source of char c: v-1 is L-type iff S[v-1] >= c
        while (L.pos[c] > a) {
            uint64_t N = (uint64_t)(L.pos[c] - a);
            if (N <= collapse_max) { // the rest of the chain in closed form
                KTRY(run_collapse(L, (uint32_t)c, a, N, mask_ge));
                a = L.pos[c];
                break;
            }
            KTRY(run_pass(L, d_SA, ctx->CTX, a, N, mask_ge, c, tot, &done));
            a = done ? L.pos[c] : a + (int64_t)N;
        }
        if ((uint64_t)L.pos[c] != start[c] + cntL[c]) return KINTERNAL();
        // the L-type part of bucket c is final (later sources start with a larger base); slot 0 goes along with bucket A
        KTRY(kiss_early_out(ctx, d_SA, c == 0 ? 0 : start[c], start[c] + cntL[c]));
        if (cntLMS[c]) {
            const uint32_t mask_gt = (0xFu << (c + 1)) & 0xFu; // S[v-1] > c for every LMS suffix
            if (ctx->lms_merged) KTRY(run_pass(L, ctx->lmsP, ctx->lmsC, (int64_t)lms_start[c], cntLMS[c], mask_gt, -1, tot, &done));
            else KTRY(run_pass(L, ctx->lms_sorted_far, ctx->lms_ctx_far, (int64_t)lms_start[c], cntLMS[c], mask_gt, -1, tot, &done, &lms_rm));
        }
    }

    if (ctx->opts.verify) KTRY(checksum_pk(ctx, "L sweep end"));
    if (ctx->opts.verify)
        for (int c = 0; c < 4; c++) KTRY(verify_part(ctx, d_SA, (int64_t)start[c], (int64_t)(start[c] + cntL[c]), (uint32_t)c, n, "L-part"));
    // ---------------- S sweep: right to left ----------------
    Sweep S{ctx, d_SA, -1, {(int64_t)start[1] - 1, (int64_t)start[2] - 1, (int64_t)start[3] - 1, (int64_t)start[4] - 1}};
    for (int c = 3; c >= 0; c--) {
        int64_t hi = (int64_t)start[c + 1]; // exclusive upper end of the not yet processed S-part
        const uint32_t mask_le = (1u << (c + 1)) - 1u; // S-type source of char c: v-1 is S-type iff S[v-1] <= c
        while (S.pos[c] + 1 < hi) {
            uint64_t N = (uint64_t)(hi - (S.pos[c] + 1));
            if (N <= collapse_max) {
                KTRY(run_collapse(S, (uint32_t)c, hi - 1, N, mask_le));
                hi = S.pos[c] + 1;
                break;
            }
            KTRY(run_pass(S, d_SA, ctx->CTX, hi - 1, N, mask_le, c, tot, &done));
            hi = done ? S.pos[c] + 1 : hi - (int64_t)N;
        }
        if ((uint64_t)(S.pos[c] + 1) != start[c] + cntL[c]) {
            fprintf(stderr, "[kiss_hip] S sweep bucket %d: tail %lld, expected %llu\n", c, (long long)(S.pos[c] + 1),
                    (unsigned long long)(start[c] + cntL[c]));
            return KINTERNAL();
        }
        // the S-type part of bucket c is final: what is appended from here on starts with a smaller base
        KTRY(kiss_early_out(ctx, d_SA, start[c] + cntL[c], start[c + 1]));
        if (cntL[c] && c > 0) {
            const uint32_t mask_lt = (1u << c) - 1u; // L-type source of char c: v-1 is S-type iff S[v-1] < c
            KTRY(run_pass(S, d_SA, ctx->CTX, (int64_t)(start[c] + cntL[c]) - 1, cntL[c], mask_lt, -1, tot, &done));
        }
    }
    ctx->ctx_words_valid = true; // the taint bits of the words parallel to SA serve kiss_exact_refine
    return KISS_HIP_OK;
}
