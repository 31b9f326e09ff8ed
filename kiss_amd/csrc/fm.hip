// fm.hip -- FM-index kernels: batched backward search + locate, and index construction from (text, SA).
//
// Index = biovoltron FMIndex<SA_INTV = 4, uint32_t, KISS1Sorter<uint32_t>>{.LOOKUP_LEN = 0}
// (reference include/biovoltron/algo/align/exact_match/fm_index.hpp); array layout = the .fmi layout
// (fm_index.hpp:591-615) so a loaded .fmi can be handed over as raw pointers.
//
//   occ(c, i)  = occ1[i/256][c] + occ2[i/16][c] + #{c in bwt[16*(i/16) .. i)} - [c == 0 && 16*(i/16) <= pri < i]
//                                                                   (compute_occ, fm_index.hpp:166-182)
//   lf(c, i)   = cnt[c] + occ(c, i)                                 (:184-187)
//   range      : (0, N) then, right to left, beg = lf(c, beg), end = lf(c, end) while end - beg >= 1
//                                                                   (get_range/compute_range :553-584, 224-235)
//   locate     : breadth-first over <= SA_INTV - 1 LF levels, emitting sa_[j] + depth for the sampled rows
//                of every visited range, in the reference's FIFO order, stopping once end - beg offsets
//                have been collected (get_offsets :453-501).
// One lane owns one pattern: 2L dependent LF steps -- a latency-bound gather workload, so the grid keeps every CU's
// 2048 lanes busy rather than tiling.  The on-disk layout costs THREE sectors per occ (an occ1 row, an occ2 entry, a BWT
// word: three arrays); the device works on a derived, interleaved copy made at the start of every batch (k_fm_blocks,
// one streaming pass over the index, microseconds): per 64 rows ONE 32-byte block = the four cumulative counts up to
// the block (occ1 + occ2 of its first chunk) + its 64 BWT dibits, so an occ is one sector and the same arithmetic as
// compute_occ.  The .fmi arrays are only read by that pass and by the locate's sampled-SA / bit-vector look-ups.
// Locate: patterns with more than FM_HEAVY occurrences (tandem repeats reach 10^6) get a workgroup of their own,
// the rest stay one lane per pattern; both produce the reference's output order.
#include "kiss_internal.hpp"
#include <cstdlib>

namespace {

constexpr int FM_THREADS = 256;
// Patterns with more hits than this (tandem repeats: up to ~10^6 occurrences) are located by a whole workgroup;
// with one lane per pattern the longest one alone set the kernel time.
constexpr uint32_t FM_HEAVY = 256;
constexpr int FMH_THREADS = 512;
// (Measured and dropped, round 3: patterns above 16 384 hits shared by 32 workgroups each -- on the bench text 3 286
//  patterns hold 16 K .. 62 K hits, no single one dominates, and repeating the per-level bookkeeping 32 times tripled the
//  kernel time; profiles/r03_fm_*.)
// ... and patterns with more than FM_LIGHT occurrences a WAVE: with one lane per pattern a wave of the light kernel runs
// as long as its busiest lane (the breadth-first walk of a 200-hit pattern is some 85 ranges of dependent look-ups), and
// that tail was most of the locate time.  The wave form is the workgroup kernel with 64 threads.
// Measured (1 M x 32-base patterns, dm-size index; locate kernel time at FM_LIGHT = 4 / 8 / 16 / 32 / 64 / 256:
// 1.84 / 1.64 / 1.58 / 1.58 / 1.56 / 1.30 ms): the wave tier LOSES -- a wave per pattern pays four levels of dependent
// look-ups however few hits there are, and there are hundreds of thousands of such patterns.  So the default keeps it
// switched off (FM_LIGHT = FM_HEAVY: no pattern is "medium"); KISS_HIP_FM_LIGHT (1 .. 256) is the A-B hook.
// (default: FM_LIGHT = the workgroup threshold, i.e. no wave tier)
constexpr int FMM_THREADS = 64;

struct FmiD {
    uint64_t N;
    uint32_t cnt[4];
    uint32_t pri;
    uint64_t bwt_bytes;
    const uint8_t *bwt;
    const uint32_t *occ1;
    const uint8_t *occ2;
    const uint32_t *sa;
    const uint64_t *b;
    const uint32_t *b_occ;
    const uint4 *blk; // interleaved rank blocks: [2 * j] = counts of A, C, G, T in bwt[0, 64 j), [2 * j + 1] = the 64 dibits
};

// the 16 dibits of chunk i/16 as one u32 (dibit t at bits 2t); the index may end inside the word
__device__ __forceinline__ uint32_t bwt_word(const FmiD &f, uint64_t chunk)
{
    uint64_t byte = chunk * 4;
    if (byte + 4 <= f.bwt_bytes) {
        const uint8_t *p = f.bwt + byte;
        if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) return *reinterpret_cast<const uint32_t *>(p);
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    }
    uint32_t w = 0;
    for (uint64_t k = 0; byte + k < f.bwt_bytes && k < 4; k++) w |= (uint32_t)f.bwt[byte + k] << (8 * k);
    return w;
}

// compute_occ (fm_index.hpp:166-182) on the on-disk arrays: what k_fm_blocks condenses, kept as the statement of the rule
__device__ __forceinline__ uint32_t fm_occ_disk(const FmiD &f, uint32_t c, uint64_t i)
{
    const uint64_t o1 = i >> 8, o2 = i >> 4;
    const uint64_t beg = o2 << 4;
    const uint32_t rem = (uint32_t)(i - beg); // 0..15 dibits of the chunk count
    uint32_t cnt = 0;
    if (rem) {
        uint32_t x = bwt_word(f, o2) ^ (c * 0x55555555u);
        uint32_t m = ~(x | (x >> 1)) & 0x55555555u; // bit 2t set <=> dibit t == c
        m &= (1u << (2 * rem)) - 1u;
        cnt = (uint32_t)__popc(m);
    }
    const uint32_t pass_pri = (c == 0 && beg <= f.pri && f.pri < i) ? 1u : 0u;
    return f.occ1[o1 * 4 + c] + (uint32_t)f.occ2[o2 * 4 + c] + cnt - pass_pri;
}

// one 32-byte block: counts before the block + the block's dibits
struct FmBlock {
    uint4 cnt;
    uint4 bw; // dibit t of the block at bits 2 (t % 16) of word t / 16
};
__device__ __forceinline__ FmBlock fm_block(const FmiD &f, uint64_t j)
{
    FmBlock b;
    b.cnt = f.blk[2 * j];
    b.bw = f.blk[2 * j + 1];
    return b;
}
// occ(c, i) for a row i of block j = i / 64 (the block already loaded): the same sum as compute_occ with the block start in
// the place of the 16-row chunk start -- counts before the block + matching dibits in [64 j, i) - the primary row's
// placeholder 'A' when it lies in that stretch
__device__ __forceinline__ uint32_t fm_occ_in(const FmiD &f, const FmBlock &b, uint32_t c, uint64_t i)
{
    const uint32_t r = (uint32_t)(i & 63u);
    const uint32_t pat = c * 0x55555555u;
    const uint32_t w[4] = {b.bw.x, b.bw.y, b.bw.z, b.bw.w};
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
        const uint32_t x = w[t] ^ pat;
        uint32_t m = ~(x | (x >> 1)) & 0x55555555u; // bit 2 u set <=> dibit u of this word == c
        const int left = (int)r - (int)(16 * t);     // rows of this word below i
        m = left >= 16 ? m : (left <= 0 ? 0u : (m & ((1u << (2 * left)) - 1u)));
        cnt += (uint32_t)__popc(m);
    }
    const uint32_t base = c == 0 ? b.cnt.x : (c == 1 ? b.cnt.y : (c == 2 ? b.cnt.z : b.cnt.w));
    const uint64_t start = i & ~63ull;
    const uint32_t pass_pri = (c == 0 && start <= f.pri && f.pri < i) ? 1u : 0u;
    return base + cnt - pass_pri;
}
__device__ __forceinline__ uint32_t fm_occ(const FmiD &f, uint32_t c, uint64_t i)
{
    return fm_occ_in(f, fm_block(f, i >> 6), c, i);
}
__device__ __forceinline__ uint64_t fm_lf(const FmiD &f, uint32_t c, uint64_t i) { return (uint64_t)f.cnt[c] + fm_occ(f, c, i); }
// both ends of a range: one block load when they fall into the same 64 rows (the usual case once the range is narrow)
__device__ __forceinline__ void fm_lf2(const FmiD &f, uint32_t c, uint64_t &beg, uint64_t &end)
{
    const uint64_t jb = beg >> 6, je = end >> 6;
    const FmBlock bb = fm_block(f, jb);
    const uint32_t ob = fm_occ_in(f, bb, c, beg);
    const uint32_t oe = je == jb ? fm_occ_in(f, bb, c, end) : fm_occ_in(f, fm_block(f, je), c, end);
    beg = (uint64_t)f.cnt[c] + ob;
    end = (uint64_t)f.cnt[c] + oe;
}

__device__ __forceinline__ uint32_t fm_bwt(const FmiD &f, uint64_t i)
{
    return ((uint32_t)f.bwt[i >> 2] >> (2 * (uint32_t)(i & 3))) & 3u;
}

// ---- the interleaved rank blocks: one lane per 64 rows, from the on-disk arrays ------------------------------------
__global__ __launch_bounds__(FM_THREADS) void k_fm_blocks(FmiD f, uint64_t nblocks, uint4 *__restrict__ blk)
{
    const uint64_t j = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    if (j >= nblocks) return;
    const uint64_t row = j * 64; // <= N: occ1 has N / 256 + 1 rows, occ2 N / 16 + 1 entries
    const uint64_t o1 = row >> 8, o2 = row >> 4;
    uint4 c;
    c.x = f.occ1[o1 * 4 + 0] + (uint32_t)f.occ2[o2 * 4 + 0];
    c.y = f.occ1[o1 * 4 + 1] + (uint32_t)f.occ2[o2 * 4 + 1];
    c.z = f.occ1[o1 * 4 + 2] + (uint32_t)f.occ2[o2 * 4 + 2];
    c.w = f.occ1[o1 * 4 + 3] + (uint32_t)f.occ2[o2 * 4 + 3];
    uint4 w; // (words past the end of the BWT read as zero: rows >= N are never counted, fm_occ_in masks by i <= N)
    w.x = bwt_word(f, o2);
    w.y = bwt_word(f, o2 + 1);
    w.z = bwt_word(f, o2 + 2);
    w.w = bwt_word(f, o2 + 3);
    blk[2 * j] = c;
    blk[2 * j + 1] = w;
}
// compute_b_occ, fm_index.hpp:189-208
__device__ __forceinline__ uint32_t fm_b_occ(const FmiD &f, uint64_t i)
{
    const uint64_t w = i >> 6;
    const uint32_t r = (uint32_t)(i & 63);
    uint32_t c = f.b_occ[w];
    if (r) c += (uint32_t)__popcll(f.b[w] & ((1ull << r) - 1ull));
    return c;
}

// ---- backward search: one lane per pattern -------------------------------------------------------
__global__ __launch_bounds__(FM_THREADS) void k_fm_range(FmiD f, const uint8_t *__restrict__ pat, uint32_t L, uint64_t Q,
                                                        uint32_t *__restrict__ beg_out, uint32_t *__restrict__ end_out,
                                                        uint64_t *__restrict__ cap /* offset slots per pattern */,
                                                        uint64_t *__restrict__ fcap /* frontier slots per pattern */,
                                                        uint32_t *__restrict__ heavy_list, uint32_t *__restrict__ nheavy,
                                                        uint32_t *__restrict__ medium_list, uint32_t *__restrict__ nmedium,
                                                        uint32_t FM_LIGHT, uint32_t fm_heavy)
{
    uint64_t q = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    const bool live = q < Q; // (no early return: the list appends below are wave-wide)
    uint64_t beg = 0, end = live ? f.N : 0;
    const uint8_t *p = pat + (live ? q : 0) * L;
    uint32_t len = live ? L : 0;
    if (!(end == beg || len == 0)) {
        while (len > 0) {
            if (end - beg < 1) break;
            uint32_t c = p[len - 1] & 3u;
            fm_lf2(f, c, beg, end);
            len--;
        }
    }
    if (live) {
        beg_out[q] = (uint32_t)beg;
        end_out[q] = (uint32_t)end;
    }
    // get_offsets (fm_index.hpp:472-482) tests `offsets.size() < end - beg` only BEFORE a range is taken from the queue
    // and then emits every sampled row of that range: on an index whose k-ordered SA ties long repeats (telomere-like
    // arrays under the k = 32 build) the walk returns MORE than end - beg positions.  Fewer than end - beg were out
    // before the last range and a range never holds more rows than the first one: at most 2 (end - beg) - 1 positions.
    if (live) {
        cap[q] = 2 * (end - beg) + 4;
        fcap[q] = end - beg > FM_LIGHT ? 0 : (end - beg) + 4; // ranges of one level (rows of a level <= end - beg)
    }
    // located by a workgroup / by a wave (the order of the lists is irrelevant); the appends are aggregated per wave
    const bool hv = live && end - beg > fm_heavy, md = live && !hv && end - beg > FM_LIGHT;
    const uint64_t hm = __ballot(hv), mm = __ballot(md);
    uint32_t hb = 0, mb = 0;
    if (lane_id() == 0) {
        if (hm) hb = atomicAdd(nheavy, (uint32_t)__popcll(hm));
        if (mm) mb = atomicAdd(nmedium, (uint32_t)__popcll(mm));
    }
    hb = __shfl(hb, 0, 64);
    mb = __shfl(mb, 0, 64);
    if (hv) heavy_list[hb + (uint32_t)__popcll(hm & lanemask_lt())] = (uint32_t)q;
    if (md) medium_list[mb + (uint32_t)__popcll(mm & lanemask_lt())] = (uint32_t)q;
}

// ---- locate: the reference's FIFO breadth-first walk, one lane per pattern ------------------------------
__global__ __launch_bounds__(FM_THREADS) void k_fm_locate(FmiD f, const uint32_t *__restrict__ beg_in,
                                                         const uint32_t *__restrict__ end_in, uint64_t Q,
                                                         const uint64_t *__restrict__ cap_index,
                                                         const uint64_t *__restrict__ fcap_index,
                                                         uint2 *__restrict__ frontier0, uint2 *__restrict__ frontier1,
                                                         uint32_t *__restrict__ out, uint64_t *__restrict__ got_out,
                                                         unsigned long long *__restrict__ totals,
                                                         const uint32_t *__restrict__ overflow, uint32_t FM_LIGHT)
{
    uint64_t q = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    unsigned long long got = 0, sum = 0;
    if (*overflow) return; // see k_fm_check
    if (q < Q) {
        const uint64_t b0 = beg_in[q], e0 = end_in[q];
        const uint64_t want = e0 - b0 > FM_LIGHT ? 0 : e0 - b0; // the others: k_fm_locate_group (a wave or a workgroup each)
        const uint64_t base = cap_index[q], fbase = fcap_index[q];
        const uint64_t capq = 2 * want + 4, fcapq = want + 4;
        uint2 *cur = frontier0 + fbase, *nxt = frontier1 + fbase;
        uint64_t ncur = 1;
        if (e0 - b0 <= FM_LIGHT) cur[0] = make_uint2((uint32_t)b0, (uint32_t)e0); // (the others own no frontier slots)
        bool stop = false;
        for (int dep = 0; dep < 4 && !stop; dep++) {
            uint64_t nn = 0;
            for (uint64_t t = 0; t < ncur; t++) {
                if (got >= want) {
                    stop = true;
                    break;
                }
                const uint2 r = cur[t];
                const uint64_t cb = r.x, ce = r.y;
                const uint32_t ob = fm_b_occ(f, cb), oe = fm_b_occ(f, ce);
                for (uint32_t i = ob; i < oe; i++) {
                    uint32_t v = f.sa[i] + (uint32_t)dep;
                    if (got < capq) out[base + got] = v;
                    got++;
                    sum += v;
                }
                if (dep + 1 == 4) continue;
                if (cb + 1 == ce) {
                    uint64_t nb = fm_lf(f, fm_bwt(f, cb), cb);
                    if (nn < fcapq) nxt[nn] = make_uint2((uint32_t)nb, (uint32_t)(nb + 1));
                    nn++;
                } else {
                    const FmBlock bb = fm_block(f, cb >> 6);
                    const FmBlock be = (ce >> 6) == (cb >> 6) ? bb : fm_block(f, ce >> 6);
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++) {
                        const uint64_t nb = (uint64_t)f.cnt[c] + fm_occ_in(f, bb, c, cb), ne = (uint64_t)f.cnt[c] + fm_occ_in(f, be, c, ce);
                        if (nb != ne) {
                            if (nn < fcapq) nxt[nn] = make_uint2((uint32_t)nb, (uint32_t)ne);
                            nn++;
                        }
                    }
                }
            }
            uint2 *tmp = cur;
            cur = nxt;
            nxt = tmp;
            ncur = nn < fcapq ? nn : fcapq;
        }
        if (e0 - b0 <= FM_LIGHT) got_out[q] = got < capq ? got : capq;
        if (got > capq) got = capq; // can not happen (bound above); keeps totals consistent with the buffers
    }
    // wave-level reduction of (hits, checksum), one atomic pair per wave
    unsigned long long g = got, s = sum;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        g += __shfl_xor(g, d, 64);
        s += __shfl_xor(s, d, 64);
    }
    if (lane_id() == 0 && (g | s)) {
        atomicAdd(&totals[0], g);
        atomicAdd(&totals[1], s);
    }
}

// ---- locate for one heavy pattern per workgroup ------------------------------------------------------------
// Same breadth-first walk and the same output order; a level has at most 4^depth <= 64 ranges, so wave 0 handles
// the ranges (one lane each: sampled-row counts, the stop rule, the children) and then all threads write the
// level's offsets -- output o belongs to the range whose prefix count covers it.
template <int THREADS> // FMH_THREADS: one workgroup per pattern; FMM_THREADS: one wave per pattern
__global__ __launch_bounds__(THREADS) void k_fm_locate_group(FmiD f, const uint32_t *__restrict__ beg_in,
                                                                const uint32_t *__restrict__ end_in,
                                                                const uint32_t *__restrict__ heavy_list,
                                                                const uint64_t *__restrict__ cap_index,
                                                                uint32_t *__restrict__ out, uint64_t *__restrict__ got_out,
                                                                unsigned long long *__restrict__ totals,
                                                                const uint32_t *__restrict__ nheavy, // device-side count
                                                                const uint32_t *__restrict__ overflow)
{
    __shared__ uint2 fr[2][64];
    __shared__ uint32_t d_ob[64], d_pre[65];
    __shared__ uint32_t s_n, s_np, s_total, s_stop;
    __shared__ unsigned long long s_sum[THREADS / 64];
    if (*overflow) return; // the scratch of this batch does not fit what the ranges need: the host regrows and runs again
    // the number of heavy patterns is only known on the device: the grid is a fixed number of workgroups that walk the list
    for (uint32_t hq = blockIdx.x; hq < *nheavy; hq += gridDim.x) {
    __syncthreads(); // (the shared arrays of the previous pattern are dead)
    const uint32_t q = heavy_list[hq];
    const uint64_t b0 = beg_in[q], e0 = end_in[q];
    const uint64_t want = e0 - b0, base = cap_index[q], capq = 2 * want + 4;
    if (threadIdx.x == 0) {
        fr[0][0] = make_uint2((uint32_t)b0, (uint32_t)e0);
        s_n = 1;
    }
    __syncthreads();
    uint64_t got = 0; // uniform
    unsigned long long sum = 0;
    int cur = 0;
    for (int dep = 0; dep < 4; dep++) {
        const uint32_t ncur = s_n;
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t t = threadIdx.x;
            const bool valid = t < ncur;
            uint64_t cb = 0, ce = 0;
            uint32_t ob = 0, cnt = 0;
            if (valid) {
                const uint2 r = fr[cur][t];
                cb = r.x;
                ce = r.y;
                ob = fm_b_occ(f, cb);
                cnt = fm_b_occ(f, ce) - ob;
            }
            uint32_t inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 64);
                if ((int)t >= d) inc += o;
            }
            const uint32_t excl = inc - cnt;
            const bool proc = valid && (got + excl < want); // the reference checks `got >= want` before every range
            const uint64_t pm = __ballot(proc);
            const uint32_t np = (uint32_t)__popcll(pm);  // processed ranges form a prefix
            const uint32_t totp = np ? __shfl(inc, (int)np - 1, 64) : 0u;
            if (proc) {
                d_ob[t] = ob;
                d_pre[t] = excl;
            }
            // children
            uint2 ch[4];
            uint32_t nch = 0;
            if (proc && dep + 1 < 4) {
                if (cb + 1 == ce) {
                    const uint64_t nb = fm_lf(f, fm_bwt(f, cb), cb);
                    ch[nch++] = make_uint2((uint32_t)nb, (uint32_t)(nb + 1));
                } else {
                    const FmBlock bb = fm_block(f, cb >> 6);
                    const FmBlock be = (ce >> 6) == (cb >> 6) ? bb : fm_block(f, ce >> 6);
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++) {
                        const uint64_t nb = (uint64_t)f.cnt[c] + fm_occ_in(f, bb, c, cb), ne = (uint64_t)f.cnt[c] + fm_occ_in(f, be, c, ce);
                        if (nb != ne) ch[nch++] = make_uint2((uint32_t)nb, (uint32_t)ne);
                    }
                }
            }
            uint32_t cinc = nch;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(cinc, d, 64);
                if ((int)t >= d) cinc += o;
            }
            const uint32_t cex = cinc - nch;
            for (uint32_t j = 0; j < nch; j++)
                if (cex + j < 64) fr[cur ^ 1][cex + j] = ch[j]; // <= 4^(dep+1) <= 64 children
            const uint32_t ctot = __shfl(cinc, 63, 64);
            if (t == 0) {
                d_pre[np] = totp;
                s_np = np;
                s_n = ctot < 64 ? ctot : 64;
                s_total = totp;
                s_stop = (np < ncur) ? 1u : 0u;
            }
        }
        __syncthreads();
        const uint32_t total = s_total, np = s_np;
        // four outputs per thread and step: the four sampled-SA reads are independent (one round trip for four)
        uint32_t lo_hint = 0; // a thread's outputs ascend: the range of the next one is never before this one's
        for (uint32_t o0 = threadIdx.x; o0 < total; o0 += 4u * THREADS) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t o = o0 + (uint32_t)u * THREADS;
                v[u] = 0;
                if (o < total) {
                    // last processed range t with d_pre[t] <= o (ranges without sampled rows tie: the last one wins);
                    // mostly the range of the previous output or the next one: a short walk, then the binary search
                    uint32_t lo = lo_hint;
                    if (lo + 1 < np && d_pre[lo + 1] <= o) {
                        lo++;
                        if (lo + 1 < np && d_pre[lo + 1] <= o) {
                            uint32_t hi = np;
                            while (hi - lo > 1) {
                                const uint32_t mid = (lo + hi) >> 1;
                                if (d_pre[mid] <= o) lo = mid;
                                else hi = mid;
                            }
                        }
                    }
                    lo_hint = lo;
                    v[u] = f.sa[d_ob[lo] + (o - d_pre[lo])] + (uint32_t)dep;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t o = o0 + (uint32_t)u * THREADS;
                if (o < total) {
                    const uint64_t idx = got + o;
                    if (idx < capq) out[base + idx] = v[u];
                    sum += v[u];
                }
            }
        }
        got += total;
        const bool stop = s_stop != 0;
        cur ^= 1;
        __syncthreads();
        if (stop) break;
    }
    // totals
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane_id() == 0) s_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < THREADS / 64; w++) s += s_sum[w];
        const uint64_t g = got < capq ? got : capq;
        got_out[q] = g;
        atomicAdd(&totals[0], (unsigned long long)g);
        atomicAdd(&totals[1], s);
    }
    } // next heavy pattern of this workgroup
}

// the scratch of a batch is sized from the LAST batch of the ctx (no host read-back in the middle of a call): this
// says whether it holds what the ranges of THIS batch need.  ctl[0] = overflow flag, ctl[1] / ctl[2] = needed entries.
__global__ void k_fm_check(const uint64_t *__restrict__ cap_index, const uint64_t *__restrict__ fcap_index, uint64_t Q,
                           uint64_t scratch_entries, uint64_t frontier_entries, uint64_t *__restrict__ need,
                           uint32_t *__restrict__ overflow)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        need[0] = cap_index[Q];
        need[1] = fcap_index[Q];
        *overflow = (cap_index[Q] > scratch_entries || fcap_index[Q] + 1 > frontier_entries) ? 1u : 0u;
    }
}

// offsets of all patterns, compacted: one thread per output element
__global__ __launch_bounds__(FM_THREADS) void k_fm_gather_offsets_flat(const uint32_t *__restrict__ scratch,
                                                                      const uint64_t *__restrict__ cap_index,
                                                                      const uint64_t *__restrict__ off_index, uint64_t Q,
                                                                      uint32_t *__restrict__ offsets, uint64_t capacity)
{
    const uint64_t o = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    if (o >= off_index[Q] || o >= capacity) return;
    uint64_t lo = 0, hi = Q; // last q with off_index[q] <= o
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off_index[mid] <= o) lo = mid;
        else hi = mid;
    }
    offsets[o] = scratch[cap_index[lo] + (o - off_index[lo])];
}

// ---- construction --------------------------------------------------------------------------------------
// one lane per 16-row chunk: BWT dibits, primary row, per-chunk character counts (pri row not counted)
__global__ __launch_bounds__(FM_THREADS) void k_fm_bwt(const uint8_t *__restrict__ S, const uint32_t *__restrict__ SA,
                                                      uint64_t N, uint64_t chunks, uint8_t *__restrict__ bwt,
                                                      uint64_t bwt_bytes, uint32_t *__restrict__ chunk_cnt,
                                                      uint32_t *__restrict__ pri)
{
    uint64_t t = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    if (t >= chunks) return;
    uint32_t word = 0, cnt = 0;
    for (uint32_t k = 0; k < 16; k++) {
        uint64_t i = t * 16 + k;
        if (i >= N) break;
        uint32_t v = SA[i];
        uint32_t c = 0;
        if (v != 0) {
            c = S[v - 1] & 3u;
            cnt += 1u << (8 * c);
        } else {
            *pri = (uint32_t)i;
        }
        word |= c << (2 * k);
    }
    for (uint32_t k = 0; k < 4; k++)
        if (t * 4 + k < bwt_bytes) bwt[t * 4 + k] = (uint8_t)(word >> (8 * k));
    chunk_cnt[t] = cnt;
}

// one lane per 256-row block: occ2 (prefix inside the block) and the block totals
__global__ __launch_bounds__(FM_THREADS) void k_fm_occ2(const uint32_t *__restrict__ chunk_cnt, uint64_t chunks,
                                                       uint64_t blocks, uint8_t *__restrict__ occ2,
                                                       uint32_t *__restrict__ blk_tot /* 4 x blocks */)
{
    uint64_t bk = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    if (bk >= blocks) return;
    uint32_t run[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16; k++) {
        uint64_t t = bk * 16 + k;
        if (t >= chunks) break;
        for (int j = 0; j < 4; j++) occ2[t * 4 + j] = (uint8_t)run[j];
        uint32_t c = chunk_cnt[t];
        for (int j = 0; j < 4; j++) run[j] += (c >> (8 * j)) & 255u;
    }
    for (int j = 0; j < 4; j++) blk_tot[(uint64_t)j * blocks + bk] = run[j];
}

__global__ __launch_bounds__(FM_THREADS) void k_fm_occ1(const uint32_t *__restrict__ blk_ex, uint64_t blocks,
                                                       uint32_t *__restrict__ occ1)
{
    uint64_t bk = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    if (bk >= blocks) return;
    for (int j = 0; j < 4; j++) occ1[bk * 4 + j] = blk_ex[(uint64_t)j * blocks + bk];
}

// one lane per 64-row word of the sampling bit-vector
__global__ __launch_bounds__(FM_THREADS) void k_fm_bits(const uint32_t *__restrict__ SA, uint64_t N, uint64_t words,
                                                       uint64_t nbocc, uint32_t sa_mask, uint64_t *__restrict__ b,
                                                       uint32_t *__restrict__ wcnt)
{
    uint64_t w = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    if (w >= nbocc) return;
    uint64_t bits = 0;
    if (w < words) {
        for (uint32_t k = 0; k < 64; k++) {
            uint64_t i = w * 64 + k;
            if (i >= N) break;
            if ((SA[i] & sa_mask) == 0) bits |= 1ull << k;
        }
        b[w] = bits;
    }
    wcnt[w] = (uint32_t)__popcll(bits);
}

__global__ __launch_bounds__(FM_THREADS) void k_fm_sample(const uint32_t *__restrict__ SA, uint64_t N, uint64_t words,
                                                         const uint64_t *__restrict__ b,
                                                         const uint32_t *__restrict__ b_occ,
                                                         uint32_t *__restrict__ sa_out)
{
    uint64_t w = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    if (w >= words) return;
    uint64_t bits = b[w];
    uint32_t p = b_occ[w];
    while (bits) {
        int k = __ffsll((unsigned long long)bits) - 1;
        bits &= bits - 1;
        sa_out[p++] = SA[w * 64 + k];
    }
}

// column c of the scanned (char, block) matrix minus the total of the smaller chars
__global__ __launch_bounds__(FM_THREADS) void k_fm_rebase(uint32_t *a, uint64_t blocks, uint32_t b0, uint32_t b1,
                                                         uint32_t b2, uint32_t b3)
{
    uint64_t i = (uint64_t)blockIdx.x * FM_THREADS + threadIdx.x;
    if (i >= 4 * blocks) return;
    uint32_t j = (uint32_t)(i / blocks);
    a[i] -= j == 0 ? b0 : (j == 1 ? b1 : (j == 2 ? b2 : b3));
}

struct DevBuf {
    void *p = nullptr;
    bool pooled = false;
    ~DevBuf()
    {
        if (p && !pooled) (void)hipFree(p);
    }
    int alloc(kiss_hip_ctx *ctx, uint64_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) {
            ctx->last_hip_error = (int)e;
            p = nullptr;
            return KISS_HIP_E_NOMEM;
        }
        return KISS_HIP_OK;
    }
    // scratch of the batched query: kept in the ctx between calls (slot = fixed role), regrown when too small --
    // nine hipMalloc / hipFree pairs per batch cost as much as the kernels
    int take(kiss_hip_ctx *ctx, int slot, uint64_t bytes)
    {
        pooled = true;
        if (ctx->fm_pool_cap[slot] < bytes) {
            if (ctx->fm_pool[slot]) (void)hipFree(ctx->fm_pool[slot]);
            ctx->fm_pool[slot] = nullptr;
            ctx->fm_pool_cap[slot] = 0;
            const uint64_t want = bytes + bytes / 8 + 256;
            hipError_t e = hipMalloc(&ctx->fm_pool[slot], want);
            if (e != hipSuccess) {
                ctx->last_hip_error = (int)e;
                ctx->fm_pool[slot] = nullptr;
                return KISS_HIP_E_NOMEM;
            }
            ctx->fm_pool_cap[slot] = want;
        }
        p = ctx->fm_pool[slot];
        return KISS_HIP_OK;
    }
};

} // namespace

extern "C" {

int kiss_hip_fmi_query_batch_dev(kiss_hip_ctx *ctx, const kiss_hip_fmi_view *fmi, const uint8_t *patterns, uint32_t L,
                                 uint64_t Q, uint32_t *beg, uint32_t *end, uint64_t *hit_count_total,
                                 uint64_t *checksum, uint32_t *offsets, uint64_t *offsets_index,
                                 uint64_t offsets_capacity, void *stream)
{
    if (!ctx || !fmi || !beg || !end || (Q && !patterns)) return KISS_HIP_E_INVALID;
    if (fmi->sa_intv != 4) return KISS_HIP_E_UNSUPPORTED; // the CLI's FMIndex<4, ...> (fmindex_build.hpp:27)
    KCHECK(hipSetDevice(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    KTRY(kiss_workspace_ready(ctx));
    if (hit_count_total) *hit_count_total = 0;
    if (checksum) *checksum = 0;
    if (Q == 0) return KISS_HIP_OK;
    if (Q / 4096 + 16 > ctx->scan_tmp_cap) return KISS_HIP_E_UNSUPPORTED; // batch larger than the ctx can scan
    FmiD f;
    f.N = fmi->n_sa;
    for (int c = 0; c < 4; c++) f.cnt[c] = fmi->cnt[c];
    f.pri = fmi->pri;
    f.bwt_bytes = (fmi->n_sa + 3) / 4;
    f.bwt = fmi->bwt;
    f.occ1 = fmi->occ1;
    f.occ2 = fmi->occ2;
    f.sa = fmi->sa;
    f.b = fmi->b;
    f.b_occ = fmi->b_occ;

    DevBuf cap, capidx, fcap, fcapidx, got, gotidx, tot, fr0, fr1, scratch, heavy, medium, blocks;
    // the interleaved rank blocks, derived from the caller's arrays at the start of every batch (never kept: the arrays
    // may have changed between calls)
    const uint64_t nblocks = f.N / 64 + 1;
    KTRY(blocks.take(ctx, 11, nblocks * 32));
    f.blk = (const uint4 *)blocks.p;
    const bool split = ((ctx->profile_mask >> KISS_HIP_K_FM_QUERY) & 1ull) != 0;
    hipEvent_t sev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (split)
        for (auto &e : sev)
            if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    struct EvGuard {
        hipEvent_t *e;
        ~EvGuard()
        {
            for (int i = 0; i < 4; i++)
                if (e[i]) (void)hipEventDestroy(e[i]);
        }
    } ev_guard{sev};
    const bool timed = split && sev[0] && sev[1] && sev[2] && sev[3];
    {
        KTimer t(ctx, KISS_HIP_K_FM_BUILD, nblocks);
        hipLaunchKernelGGL(k_fm_blocks, dim3((unsigned)div_up(nblocks, FM_THREADS)), dim3(FM_THREADS), 0, ctx->stream, f, nblocks,
                           (uint4 *)blocks.p);
        KCHECK(hipGetLastError());
    }
    kiss_opts_refresh(ctx);
    // (hooks build: the two tier thresholds can be swept, 16 .. 65536 and 1 .. fm_heavy)
    const uint32_t fm_heavy = ctx->opts.fm_heavy >= 16 && ctx->opts.fm_heavy <= 65536 ? ctx->opts.fm_heavy : (uint32_t)FM_HEAVY;
    const uint32_t fm_light = ctx->opts.fm_light >= 1 && ctx->opts.fm_light <= fm_heavy ? ctx->opts.fm_light : fm_heavy;
    KTRY(heavy.take(ctx, 0, (Q + 2) * 4)); // [0] = count, [1..] = pattern numbers
    KTRY(kiss_zero_u32(ctx, heavy.p, 1));
    KTRY(medium.take(ctx, 12, (Q + 2) * 4));
    KTRY(kiss_zero_u32(ctx, medium.p, 1));

    KTRY(cap.take(ctx, 1, (Q + 1) * 8));
    KTRY(capidx.take(ctx, 2, (Q + 1) * 8));
    KTRY(fcap.take(ctx, 9, (Q + 1) * 8));
    KTRY(fcapidx.take(ctx, 10, (Q + 1) * 8));
    KTRY(got.take(ctx, 3, (Q + 1) * 8));
    KTRY(gotidx.take(ctx, 4, (Q + 1) * 8));
    KTRY(tot.take(ctx, 5, 64)); // [0] hits, [1] checksum, [2] offset-scratch entries needed, [3] frontier entries, [4] overflow
    const unsigned grid = (unsigned)div_up(Q, FM_THREADS);
    {
        KTimer t(ctx, KISS_HIP_K_FM_QUERY, Q);
        if (timed) (void)hipEventRecord(sev[0], ctx->stream);
        hipLaunchKernelGGL(k_fm_range, dim3(grid), dim3(FM_THREADS), 0, ctx->stream, f, patterns, L, Q, beg, end,
                           (uint64_t *)cap.p, (uint64_t *)fcap.p, (uint32_t *)heavy.p + 1, (uint32_t *)heavy.p,
                           (uint32_t *)medium.p + 1, (uint32_t *)medium.p, fm_light, fm_heavy);
        if (timed) (void)hipEventRecord(sev[1], ctx->stream);
        KCHECK(hipGetLastError());
    }
    KTRY(kiss_zero_u32(ctx, (uint8_t *)cap.p + Q * 8, 2));
    KTRY(kiss_scan_u64(ctx, (const uint64_t *)cap.p, (uint64_t *)capidx.p, Q + 1));
    KTRY(kiss_zero_u32(ctx, (uint8_t *)fcap.p + Q * 8, 2));
    KTRY(kiss_scan_u64(ctx, (const uint64_t *)fcap.p, (uint64_t *)fcapidx.p, Q + 1));
    // The offset scratch and the frontier arrays are sized by what the PREVIOUS batch of this ctx needed (the pool keeps
    // them): the locate kernels are queued right behind the scans, a one-thread kernel tells them on the device whether
    // the buffers hold this batch, and the host learns the sizes with the totals at the end -- one synchronisation per
    // call instead of two.  Only a batch that needs more than the pool holds (the first one, or a much heavier one) is
    // located a second time after the pool has grown.
    uint64_t *d_need = (uint64_t *)tot.p + 2;        // [2], [3] of the totals block
    uint32_t *d_over = (uint32_t *)((uint64_t *)tot.p + 4);
    uint64_t h[5] = {0, 0, 0, 0, 0};
    uint32_t nheavy = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        const uint64_t scratch_entries = ctx->fm_pool_cap[8] / sizeof(uint32_t);
        const uint64_t frontier_entries = (ctx->fm_pool_cap[6] < ctx->fm_pool_cap[7] ? ctx->fm_pool_cap[6] : ctx->fm_pool_cap[7]) / sizeof(uint2);
        KTRY(fr0.take(ctx, 6, ctx->fm_pool_cap[6]));
        KTRY(fr1.take(ctx, 7, ctx->fm_pool_cap[7]));
        KTRY(scratch.take(ctx, 8, ctx->fm_pool_cap[8]));
        KTRY(kiss_zero_u32(ctx, tot.p, 4));
        hipLaunchKernelGGL(k_fm_check, dim3(1), dim3(64), 0, ctx->stream, (const uint64_t *)capidx.p, (const uint64_t *)fcapidx.p, Q,
                           scratch_entries, frontier_entries, d_need, d_over);
        {
            KTimer t(ctx, KISS_HIP_K_FM_QUERY, Q);
            if (timed) (void)hipEventRecord(sev[2], ctx->stream);
            hipLaunchKernelGGL(k_fm_locate, dim3(grid), dim3(FM_THREADS), 0, ctx->stream, f, beg, end, Q,
                               (const uint64_t *)capidx.p, (const uint64_t *)fcapidx.p, (uint2 *)fr0.p, (uint2 *)fr1.p,
                               (uint32_t *)scratch.p, (uint64_t *)got.p, (unsigned long long *)tot.p, (const uint32_t *)d_over,
                               fm_light);
            const unsigned hgrid = (unsigned)(Q < 2048 ? Q : 2048);
            hipLaunchKernelGGL((k_fm_locate_group<FMH_THREADS>), dim3(hgrid), dim3(FMH_THREADS), 0, ctx->stream, f, beg, end,
                               (const uint32_t *)heavy.p + 1, (const uint64_t *)capidx.p, (uint32_t *)scratch.p,
                               (uint64_t *)got.p, (unsigned long long *)tot.p, (const uint32_t *)heavy.p, (const uint32_t *)d_over);
            const unsigned mgrid = (unsigned)(Q < 16384 ? Q : 16384); // one wave per pattern, walking the medium list
            hipLaunchKernelGGL((k_fm_locate_group<FMM_THREADS>), dim3(mgrid), dim3(FMM_THREADS), 0, ctx->stream, f, beg, end,
                               (const uint32_t *)medium.p + 1, (const uint64_t *)capidx.p, (uint32_t *)scratch.p,
                               (uint64_t *)got.p, (unsigned long long *)tot.p, (const uint32_t *)medium.p, (const uint32_t *)d_over);
            if (timed) (void)hipEventRecord(sev[3], ctx->stream);
            KCHECK(hipGetLastError());
        }
        KCHECK(hipMemcpyAsync(h, tot.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        KCHECK(hipMemcpyAsync(&nheavy, heavy.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
        if ((uint32_t)h[4] == 0) break; // the buffers held the batch
        if (attempt) return KINTERNAL();
        // grow the pool to what this batch needs and locate again (the ranges stand)
        KTRY(fr0.take(ctx, 6, (h[3] + 1) * sizeof(uint2)));
        KTRY(fr1.take(ctx, 7, (h[3] + 1) * sizeof(uint2)));
        KTRY(scratch.take(ctx, 8, h[2] * sizeof(uint32_t)));
    }
    const uint64_t total_cap = h[2];
    (void)nheavy;
    if (offsets && offsets_index) {
        KTRY(kiss_zero_u32(ctx, (uint8_t *)got.p + Q * 8, 2));
        KTRY(kiss_scan_u64(ctx, (const uint64_t *)got.p, offsets_index, Q + 1));
        const uint64_t bound = total_cap < offsets_capacity ? total_cap : offsets_capacity; // >= number of offsets
        if (bound)
            hipLaunchKernelGGL(k_fm_gather_offsets_flat, dim3((unsigned)div_up(bound, FM_THREADS)), dim3(FM_THREADS), 0,
                               ctx->stream, (const uint32_t *)scratch.p, (const uint64_t *)capidx.p, offsets_index, Q, offsets,
                               offsets_capacity);
        KCHECK(hipGetLastError());
    }
    if (offsets && offsets_index) KCHECK(hipStreamSynchronize(ctx->stream)); // (the gather above)
    if (hit_count_total) *hit_count_total = h[0];
    if (checksum) *checksum = h[1];
    if (timed) { // the two halves of the query apart (they accumulate like ms_kernel[] until the next sort resets the stats)
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, sev[0], sev[1]) == hipSuccess) ctx->stats.ms_fm_range += a;
        if (hipEventElapsedTime(&b, sev[2], sev[3]) == hipSuccess) ctx->stats.ms_fm_locate += b;
    }
    ktimer_collect(ctx);
    return KISS_HIP_OK;
}

int kiss_hip_fmi_build_dev(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, const uint32_t *d_SA, uint32_t sa_intv,
                           uint8_t *d_bwt, uint32_t *d_occ1, uint8_t *d_occ2, uint32_t *d_sa_sampled, uint64_t *d_b,
                           uint32_t *d_b_occ, uint32_t cnt_out[4], uint32_t *pri_out, void *stream)
{
    if (!ctx || !d_SA || (n && !d_S) || !d_bwt || !d_occ1 || !d_occ2 || !d_sa_sampled || !d_b || !d_b_occ || !cnt_out ||
        !pri_out)
        return KISS_HIP_E_INVALID;
    if (sa_intv != 4) return KISS_HIP_E_UNSUPPORTED;
    if (n > KISS_HIP_MAX_N) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    KTRY(kiss_workspace_ready(ctx));
    const uint64_t N = n + 1;
    const uint64_t chunks = N / 16 + 1, blocks = N / 256 + 1;
    const uint64_t words = (N + 63) / 64, nbocc = N / 64 + 1;
    const uint64_t bwt_bytes = (N + 3) / 4;
    if (blocks / 4096 + 16 > ctx->scan_tmp_cap || nbocc / 4096 + 16 > ctx->scan_tmp_cap) return KISS_HIP_E_UNSUPPORTED;
    DevBuf chunk_cnt, blk_tot, wcnt, pri;
    KTRY(chunk_cnt.alloc(ctx, chunks * 4));
    KTRY(blk_tot.alloc(ctx, (4 * blocks + 1) * 4));
    KTRY(wcnt.alloc(ctx, nbocc * 4));
    KTRY(pri.alloc(ctx, 4));
    KTimer t(ctx, KISS_HIP_K_FM_BUILD, N);
    hipLaunchKernelGGL(k_fm_bwt, dim3((unsigned)div_up(chunks, FM_THREADS)), dim3(FM_THREADS), 0, ctx->stream, d_S, d_SA,
                       N, chunks, d_bwt, bwt_bytes, (uint32_t *)chunk_cnt.p, (uint32_t *)pri.p);
    hipLaunchKernelGGL(k_fm_occ2, dim3((unsigned)div_up(blocks, FM_THREADS)), dim3(FM_THREADS), 0, ctx->stream,
                       (const uint32_t *)chunk_cnt.p, chunks, blocks, d_occ2, (uint32_t *)blk_tot.p);
    KCHECK(hipGetLastError());
    // digit-major exclusive scan over (char, block) gives, per char, prefix over blocks + totals of smaller chars;
    // subtract the per-char base afterwards on the host side of cnt[]
    KTRY(kiss_zero_u32(ctx, (uint32_t *)blk_tot.p + 4 * blocks, 1));
    KTRY(kiss_scan_u32(ctx, (const uint32_t *)blk_tot.p, (uint32_t *)blk_tot.p, 4 * blocks + 1));
    uint32_t base[5];
    for (int j = 0; j < 5; j++)
        KCHECK(hipMemcpyAsync(&base[j], (uint32_t *)blk_tot.p + (uint64_t)j * blocks, 4, hipMemcpyDeviceToHost,
                              ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));
    // occ1[block][c] = scan[c*blocks + block] - base[c]; do the subtraction in the interleave kernel by
    // temporarily rebasing: simplest is a tiny second pass per char
    hipLaunchKernelGGL(k_fm_rebase, dim3((unsigned)div_up(4 * blocks, FM_THREADS)), dim3(FM_THREADS), 0, ctx->stream,
                       (uint32_t *)blk_tot.p, blocks, base[0], base[1], base[2], base[3]);
    hipLaunchKernelGGL(k_fm_occ1, dim3((unsigned)div_up(blocks, FM_THREADS)), dim3(FM_THREADS), 0, ctx->stream,
                       (const uint32_t *)blk_tot.p, blocks, d_occ1);
    // sampling bit-vector, its rank directory and the sampled SA
    hipLaunchKernelGGL(k_fm_bits, dim3((unsigned)div_up(nbocc, FM_THREADS)), dim3(FM_THREADS), 0, ctx->stream, d_SA, N,
                       words, nbocc, sa_intv - 1, d_b, (uint32_t *)wcnt.p);
    KCHECK(hipGetLastError());
    KTRY(kiss_scan_u32(ctx, (const uint32_t *)wcnt.p, d_b_occ, nbocc));
    hipLaunchKernelGGL(k_fm_sample, dim3((unsigned)div_up(words, FM_THREADS)), dim3(FM_THREADS), 0, ctx->stream, d_SA, N,
                       words, d_b, d_b_occ, d_sa_sampled);
    KCHECK(hipGetLastError());
    uint32_t h_pri = 0;
    KCHECK(hipMemcpyAsync(&h_pri, pri.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));
    *pri_out = h_pri;
    // cnt_ = {1, 1+#A, 1+#A+#C, 1+#A+#C+#G} (fm_index.hpp:303-307); totals from the scan bases
    uint32_t sum = 1;
    for (int j = 0; j < 4; j++) {
        uint32_t tot = base[j + 1] - base[j];
        cnt_out[j] = sum;
        sum += tot;
    }
    return KISS_HIP_OK;
}

// ---- host-pointer forms (hosts that do not link HIP: the CLI, cgo/ctypes callers) -----------------------
int kiss_hip_fmi_sizes_for(uint64_t n, kiss_hip_fmi_sizes *out)
{
    if (!out || n > KISS_HIP_MAX_N) return KISS_HIP_E_INVALID;
    const uint64_t N = n + 1;
    out->n_sa = N;
    out->bwt_bytes = (N + 3) / 4;
    out->occ1_entries = (N / 256 + 1) * 4;
    out->occ2_bytes = (N / 16 + 1) * 4;
    out->sa_entries = (N + 3) / 4;
    out->b_words = (N + 63) / 64;
    out->b_occ_entries = N / 64 + 1;
    return KISS_HIP_OK;
}

int kiss_hip_fmi_build_host(const uint8_t *S, uint64_t n, const uint32_t *SA_or_null, uint8_t *bwt, uint32_t *occ1,
                            uint8_t *occ2, uint32_t *sa, uint64_t *b, uint32_t *b_occ, uint32_t cnt_out[4],
                            uint32_t *pri_out, int device)
{
    if (!S || !bwt || !occ1 || !occ2 || !sa || !b || !b_occ || !cnt_out || !pri_out || n == 0) return KISS_HIP_E_INVALID;
    kiss_hip_fmi_sizes z;
    KTRY(kiss_hip_fmi_sizes_for(n, &z));
    kiss_hip_ctx *ctx = nullptr;
    int rc = kiss_hip_ctx_create(&ctx, device, n);
    if (rc) return rc;
    DevBuf dS, dSA, dbwt, docc1, docc2, dsa, db, dbocc;
    do {
        if ((rc = dS.alloc(ctx, n)) || (rc = dSA.alloc(ctx, (n + 1) * 4)) || (rc = dbwt.alloc(ctx, z.bwt_bytes + 8)) ||
            (rc = docc1.alloc(ctx, z.occ1_entries * 4)) || (rc = docc2.alloc(ctx, z.occ2_bytes)) ||
            (rc = dsa.alloc(ctx, z.sa_entries * 4)) || (rc = db.alloc(ctx, z.b_words * 8 + 8)) ||
            (rc = dbocc.alloc(ctx, z.b_occ_entries * 4)))
            break;
        if (hipMemcpy(dS.p, S, n, hipMemcpyHostToDevice) != hipSuccess) { rc = KISS_HIP_E_HIP; break; }
        if (SA_or_null) {
            if (hipMemcpy(dSA.p, SA_or_null, (n + 1) * 4, hipMemcpyHostToDevice) != hipSuccess) { rc = KISS_HIP_E_HIP; break; }
        } else {
            // FMIndex::build sorts with k = 32 whatever the caller's flags say (fm_index.hpp:384-386)
            if ((rc = kiss_hip_ctx_suffix_sort_dna_u32_dev(ctx, (const uint8_t *)dS.p, n, 32u, KISS_HIP_ALGO_PARALLEL_SORTING,
                                                           (uint32_t *)dSA.p, nullptr)))
                break;
        }
        if ((rc = kiss_hip_fmi_build_dev(ctx, (const uint8_t *)dS.p, n, (const uint32_t *)dSA.p, 4, (uint8_t *)dbwt.p,
                                         (uint32_t *)docc1.p, (uint8_t *)docc2.p, (uint32_t *)dsa.p, (uint64_t *)db.p,
                                         (uint32_t *)dbocc.p, cnt_out, pri_out, nullptr)))
            break;
        hipError_t e = hipMemcpy(bwt, dbwt.p, z.bwt_bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(occ1, docc1.p, z.occ1_entries * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(occ2, docc2.p, z.occ2_bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(sa, dsa.p, z.sa_entries * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(b, db.p, z.b_words * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(b_occ, dbocc.p, z.b_occ_entries * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = KISS_HIP_E_HIP;
    } while (0);
    kiss_hip_ctx_destroy(ctx);
    return rc;
}

int kiss_hip_fmi_query_batch_host(const kiss_hip_fmi_view *fmi, const uint8_t *patterns, uint32_t L, uint64_t Q,
                                  uint32_t *beg, uint32_t *end, uint64_t *hit_count_total, uint64_t *checksum,
                                  uint32_t *offsets, uint64_t *offsets_index, uint64_t offsets_capacity, int device)
{
    if (!fmi || !beg || !end || (Q && !patterns) || fmi->n_sa == 0) return KISS_HIP_E_INVALID;
    kiss_hip_fmi_sizes z;
    KTRY(kiss_hip_fmi_sizes_for(fmi->n_sa - 1, &z));
    kiss_hip_ctx *ctx = nullptr;
    uint64_t max_n = fmi->n_sa > 4 * Q ? fmi->n_sa : 4 * Q;
    if (max_n < (1u << 20)) max_n = 1u << 20;
    int rc = kiss_hip_ctx_create(&ctx, device, max_n);
    if (rc) return rc;
    DevBuf dbwt, docc1, docc2, dsa, db, dbocc, dpat, dbeg, dend, doff, didx;
    do {
        if ((rc = dbwt.alloc(ctx, z.bwt_bytes + 8)) || (rc = docc1.alloc(ctx, z.occ1_entries * 4)) ||
            (rc = docc2.alloc(ctx, z.occ2_bytes)) || (rc = dsa.alloc(ctx, z.sa_entries * 4)) ||
            (rc = db.alloc(ctx, z.b_words * 8 + 8)) || (rc = dbocc.alloc(ctx, z.b_occ_entries * 4)) ||
            (rc = dpat.alloc(ctx, Q * L)) || (rc = dbeg.alloc(ctx, Q * 4)) || (rc = dend.alloc(ctx, Q * 4)))
            break;
        hipError_t e = hipMemcpy(dbwt.p, fmi->bwt, z.bwt_bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(docc1.p, fmi->occ1, z.occ1_entries * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(docc2.p, fmi->occ2, z.occ2_bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dsa.p, fmi->sa, z.sa_entries * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(db.p, fmi->b, z.b_words * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dbocc.p, fmi->b_occ, z.b_occ_entries * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && Q) e = hipMemcpy(dpat.p, patterns, Q * L, hipMemcpyHostToDevice);
        if (e != hipSuccess) { rc = KISS_HIP_E_HIP; break; }
        kiss_hip_fmi_view v = *fmi;
        v.bwt = (const uint8_t *)dbwt.p;
        v.occ1 = (const uint32_t *)docc1.p;
        v.occ2 = (const uint8_t *)docc2.p;
        v.sa = (const uint32_t *)dsa.p;
        v.b = (const uint64_t *)db.p;
        v.b_occ = (const uint32_t *)dbocc.p;
        const bool want = offsets && offsets_index && offsets_capacity;
        if (want && ((rc = doff.alloc(ctx, offsets_capacity * 4)) || (rc = didx.alloc(ctx, (Q + 1) * 8)))) break;
        if ((rc = kiss_hip_fmi_query_batch_dev(ctx, &v, (const uint8_t *)dpat.p, L, Q, (uint32_t *)dbeg.p, (uint32_t *)dend.p,
                                               hit_count_total, checksum, want ? (uint32_t *)doff.p : nullptr,
                                               want ? (uint64_t *)didx.p : nullptr, want ? offsets_capacity : 0, nullptr)))
            break;
        e = hipMemcpy(beg, dbeg.p, Q * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(end, dend.p, Q * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && want) e = hipMemcpy(offsets_index, didx.p, (Q + 1) * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess && want) e = hipMemcpy(offsets, doff.p, offsets_capacity * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = KISS_HIP_E_HIP;
    } while (0);
    kiss_hip_ctx_destroy(ctx);
    return rc;
}
}
