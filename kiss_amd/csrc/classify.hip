// classify.hip -- 2-bit packing and get_lms on the GPU.
//
// Restates the *result* of get_lms (reference include/biovoltron/algo/sort/kiss_common.hpp:483-579):
// the ascending list of LMS positions and the per-character histograms
// (count[c], S-type count, LMS count), not its thread-chunk structure.
//
// Types are computed bit-parallel on the packed text.  One thread owns one 64-bit
// word (32 bases).  For adjacent bases the word gives lt/eq masks; the suffix type
//     S(i) = lt(i) | (eq(i) & S(i+1))
// is a carry chain running from later bases (low bits) to earlier ones (high bits),
// evaluated with ONE 64-bit addition per word, one ballot+addition per wave, and a
// generate/propagate fold across waves and 256-word tiles (three launches:
// tile summaries -> serial-free scan of summaries -> count / emit).
#include "kiss_internal.hpp"

namespace {

constexpr uint64_t HI = 0xAAAAAAAAAAAAAAAAull;
constexpr uint64_t LO = 0x5555555555555555ull;
constexpr int CL_THREADS = 256; // words per tile

// ---- 2-bit packing --------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack(const uint8_t *__restrict__ S, uint64_t n, uint64_t *__restrict__ pk,
                                              uint64_t words)
{
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    uint64_t base = w * 32;
    uint64_t out = 0;
    if (base + 32 <= n && ((reinterpret_cast<uintptr_t>(S + base) & 15u) == 0)) {
        const uint4 *p = reinterpret_cast<const uint4 *>(S + base);
        uint4 v0 = p[0], v1 = p[1];
        uint32_t q[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int t = 0; t < 8; t++) {
            uint32_t u = q[t] & 0x03030303u; // bytes b0..b3 (b0 = lowest = earliest base)
            // -> 8 bits: b0 b1 b2 b3 from high to low
            uint32_t r = ((u & 0x3u) << 6) | (((u >> 8) & 0x3u) << 4) | (((u >> 16) & 0x3u) << 2) | ((u >> 24) & 0x3u);
            out |= (uint64_t)r << (56 - 8 * t);
        }
    } else {
        for (int j = 0; j < 32; j++) {
            uint64_t i = base + j;
            uint64_t c = (i < n) ? (uint64_t)(S[i] & 3u) : 0ull;
            out |= c << (62 - 2 * j);
        }
    }
    pk[w] = out;
}

// ---- per-word masks ---------------------------------------------------------------
struct WordMasks {
    uint64_t x;  // the packed word
    uint64_t A;  // adder operand (lt|eq at HI positions, all LO bits set)
    uint64_t B;  // adder operand (lt at HI positions)
};

// Consecutive lanes hold consecutive words (both callers: w = tile base + threadIdx.x), so a lane loads ITS word only -- one
// naturally aligned 8-byte load (never a 16-byte load at an 8-byte aligned address) -- and takes the next
// word from the next lane; lane 63 loads it.  ALL lanes of the wave must call (words >= `words` read as zero: the spare words
// behind the text are zero).
__device__ __forceinline__ WordMasks word_masks(const uint64_t *__restrict__ pk, uint64_t w, uint64_t n, uint64_t words)
{
    WordMasks m;
    const uint64_t x = w < words ? pk[w] : 0ull;
    uint64_t nx = __shfl_down(x, 1, 64);
    if (lane_id() == 63) nx = w + 1 < words ? pk[w + 1] : 0ull;
    if (w >= words) { // (beyond the text: the neutral element of the carry chain)
        m.x = 0;
        m.A = LO;
        m.B = 0;
        return m;
    }
    uint64_t y = (x << 2) | (nx >> 62); // field j of y = base 32w+j+1
    uint64_t d = x ^ y;
    uint64_t eqh = ~d & HI;
    uint64_t eql = ~d & LO;
    uint64_t eq = eqh & (eql << 1);
    uint64_t xy = ~x & y;
    uint64_t lt = (xy & HI) | (eqh & ((xy & LO) << 1));
    // only bases i < n-1 take part (base n-1 is L-type: the sentinel is smaller)
    int64_t rem = (int64_t)(n - 1) - (int64_t)(w * 32);
    uint64_t vmask = rem >= 32 ? ~0ull : (rem <= 0 ? 0ull : (~0ull << (64 - 2 * rem)));
    lt &= vmask;
    eq &= vmask;
    m.x = x;
    m.A = lt | eq | LO;
    m.B = lt;
    return m;
}

// carry-out of the word's chain for a given carry-in; *sum receives A+B+cin (low 64 bits)
__device__ __forceinline__ uint32_t word_carry(uint64_t A, uint64_t B, uint32_t cin, uint64_t *sum)
{
    uint64_t s = A + B;
    uint32_t c1 = s < A;
    uint64_t s2 = s + cin;
    uint32_t c2 = s2 < s;
    *sum = s2;
    return c1 | c2;
}

// wave-level fold.  G/P per lane (lane = word, higher lane = later text).  Returns this lane's carry-in
// given the carry-in of the whole wave; *wave_cout receives the wave's carry-out.
__device__ __forceinline__ uint32_t wave_carry(uint32_t G, uint32_t P, uint32_t wave_cin, uint32_t *wave_cout)
{
    uint64_t gm = __ballot(G != 0);
    uint64_t pm = __ballot(P != 0);
    uint64_t rg = __brevll(gm), rp = __brevll(pm); // bit b = lane 63-b : carries now run low -> high
    uint64_t a = rg | rp, b = rg;
    uint64_t s = a + b;
    uint32_t c1 = s < a;
    uint64_t s2 = s + wave_cin;
    uint32_t c2 = s2 < s;
    *wave_cout = c1 | c2;
    uint64_t cins = s2 ^ a ^ b; // bit b = carry into bit b
    return (uint32_t)(cins >> (63u - lane_id())) & 1u;
}

// Computes, for the calling thread's word, the S-type mask (bit 62-2j = type of base 32w+j is S)
// given the carry-in of the tile.  lds: 8 x u32.  All 256 threads must call.
__device__ __forceinline__ uint64_t tile_types(const WordMasks &m, uint32_t tile_cin, uint32_t *lds)
{
    const int wave = threadIdx.x >> 6;
    uint64_t sum;
    uint32_t c0 = word_carry(m.A, m.B, 0, &sum);
    uint32_t c1 = word_carry(m.A, m.B, 1, &sum);
    uint32_t G = c0, P = c1 & ~c0;
    uint32_t wc0, wc1;
    (void)wave_carry(G, P, 0, &wc0);
    (void)wave_carry(G, P, 1, &wc1);
    if (lane_id() == 0) {
        lds[wave * 2 + 0] = wc0;
        lds[wave * 2 + 1] = wc1 & ~wc0;
    }
    __syncthreads();
    uint32_t c = tile_cin;
    for (int w = CL_THREADS / 64 - 1; w > wave; w--) c = lds[w * 2] | (lds[w * 2 + 1] & c);
    uint32_t dummy;
    uint32_t cin = wave_carry(G, P, c, &dummy);
    uint32_t cout = word_carry(m.A, m.B, cin, &sum);
    __syncthreads();
    return ((~sum & LO) >> 2) | ((uint64_t)cout << 62);
}

// S[i-1] > S[i] for every field (at HI positions); px = previous word (0 for w == 0)
__device__ __forceinline__ uint64_t gt_prev_mask(uint64_t x, uint64_t px)
{
    uint64_t z = (x >> 2) | (px << 62);
    uint64_t dz = z ^ x;
    uint64_t eqh = ~dz & HI;
    uint64_t zx = z & ~x;
    return (zx & HI) | (eqh & ((zx & LO) << 1));
}

// ---- kernel 1: generate/propagate summary of each 256-word tile ----------------------
__global__ __launch_bounds__(CL_THREADS) void k_tile_gp(const uint64_t *__restrict__ pk, uint64_t n, uint64_t words,
                                                       uint32_t *__restrict__ tile_gp)
{
    __shared__ uint32_t lds[8];
    uint64_t w = (uint64_t)blockIdx.x * CL_THREADS + threadIdx.x;
    const WordMasks m = word_masks(pk, w, n, words);
    uint64_t sum;
    uint32_t c0 = word_carry(m.A, m.B, 0, &sum);
    uint32_t c1 = word_carry(m.A, m.B, 1, &sum);
    uint32_t wc0, wc1;
    (void)wave_carry(c0, c1 & ~c0, 0, &wc0);
    (void)wave_carry(c0, c1 & ~c0, 1, &wc1);
    const int wave = threadIdx.x >> 6;
    if (lane_id() == 0) {
        lds[wave * 2 + 0] = wc0;
        lds[wave * 2 + 1] = wc1 & ~wc0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 1;
        for (int v = CL_THREADS / 64 - 1; v >= 0; v--) {
            a = lds[v * 2] | (lds[v * 2 + 1] & a);
            b = lds[v * 2] | (lds[v * 2 + 1] & b);
        }
        tile_gp[blockIdx.x] = a | ((b & ~a) << 1);
    }
}

// ---- kernel 2: turn tile (G,P) into tile carry-ins (right-to-left fold), single workgroup -----
__global__ __launch_bounds__(1024) void k_tile_cin(uint32_t *__restrict__ tile_gp, uint64_t tiles)
{
    __shared__ uint32_t sg[1024], sp[1024], scin[1024];
    const uint64_t chunk = (tiles + 1023) / 1024;
    const uint64_t beg = (uint64_t)threadIdx.x * chunk;
    const uint64_t end = beg + chunk < tiles ? beg + chunk : tiles;
    uint32_t a = 0, b = 1; // chunk carry-out for cin = 0 / 1
    for (uint64_t t = end; t > beg; t--) {
        uint32_t gp = tile_gp[t - 1];
        uint32_t g = gp & 1u, p = (gp >> 1) & 1u;
        a = g | (p & a);
        b = g | (p & b);
    }
    if (beg >= end) { a = 0; b = 1; }
    sg[threadIdx.x] = a;
    sp[threadIdx.x] = b & ~a;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int i = 1023; i >= 0; i--) {
            scin[i] = c;
            c = sg[i] | (sp[i] & c);
        }
    }
    __syncthreads();
    uint32_t c = scin[threadIdx.x];
    for (uint64_t t = end; t > beg; t--) {
        uint32_t gp = tile_gp[t - 1];
        tile_gp[t - 1] = c;
        c = (gp & 1u) | (((gp >> 1) & 1u) & c);
    }
}

// The same fold over 380 k tiles (chm13 size) by ONE workgroup is 0.7 ms of dependent loads; with a level in front it
// is three launches of a few microseconds: chunks of TC_CHUNK tiles -> (G,P) per chunk -> k_tile_cin over the chunks ->
// carry-ins inside every chunk.
constexpr int TC_CHUNK = 16;
__global__ __launch_bounds__(256) void k_tile_cin_chunks(const uint32_t *__restrict__ tile_gp, uint64_t tiles, uint64_t chunks,
                                                         uint32_t *__restrict__ chunk_gp)
{
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= chunks) return;
    const uint64_t beg = c * TC_CHUNK, end = beg + TC_CHUNK < tiles ? beg + TC_CHUNK : tiles;
    uint32_t a = 0, b = 1; // chunk carry-out for carry-in 0 / 1
    for (uint64_t t = end; t > beg; t--) {
        const uint32_t gp = tile_gp[t - 1];
        const uint32_t g = gp & 1u, p = (gp >> 1) & 1u;
        a = g | (p & a);
        b = g | (p & b);
    }
    chunk_gp[c] = a | ((b & ~a) << 1);
}

__global__ __launch_bounds__(256) void k_tile_cin_apply(uint32_t *__restrict__ tile_gp, uint64_t tiles, uint64_t chunks,
                                                        const uint32_t *__restrict__ chunk_cin)
{
    const uint64_t ch = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (ch >= chunks) return;
    const uint64_t beg = ch * TC_CHUNK, end = beg + TC_CHUNK < tiles ? beg + TC_CHUNK : tiles;
    uint32_t gp[TC_CHUNK];
#pragma unroll
    for (int e = 0; e < TC_CHUNK; e++) gp[e] = beg + e < end ? tile_gp[beg + e] : 0u;
    uint32_t c = chunk_cin[ch];
#pragma unroll
    for (int e = TC_CHUNK - 1; e >= 0; e--) {
        if (beg + e < end) {
            tile_gp[beg + e] = c;
            c = (gp[e] & 1u) | (((gp[e] >> 1) & 1u) & c);
        }
    }
}

// ---- kernel 3/4: count (histograms + per-tile LMS count) and emit (positions + first keys) ------
template <bool EMIT>
__global__ __launch_bounds__(CL_THREADS) void k_classify(const uint64_t *__restrict__ pk, uint64_t n, uint64_t words,
                                                        uint64_t tiles, const uint32_t *__restrict__ tile_cin,
                                                        uint32_t *__restrict__ tile_cnt, // count: out; emit: offsets in
                                                        uint32_t *__restrict__ d_counts, uint64_t far_limit,
                                                        uint64_t win_lo, uint64_t win_hi, // only positions in [lo, hi)
                                                        uint64_t tile_lo, uint64_t tile_hi, // = the tiles that overlap it
                                                        uint32_t *__restrict__ lms_pos, uint64_t *__restrict__ lms_key,
                                                        uint32_t *__restrict__ ghist, // emit: round-0 digit histograms
                                                        uint32_t *__restrict__ part)  // per-workgroup partial sums (see below)
{
    // The emit pass also counts the five 8-bit digits round 0 of the LMS sort will scatter on (key bits 24..63 of the
    // far suffixes), so the radix sort does not read the 7 GB of keys once more just to count them (radix.hip).
    __shared__ uint32_t hh[EMIT ? KISS_R0_PASSES * 256 : 1];
    if (EMIT && ghist) {
        for (uint32_t i = threadIdx.x; i < KISS_R0_PASSES * 256; i += CL_THREADS) hh[i] = 0;
        __syncthreads();
    }
    __shared__ uint32_t lds[8];
    __shared__ uint32_t wsum[CL_THREADS / 64 + 1];
    __shared__ uint16_t stage[EMIT ? CL_THREADS * 16 : 1]; // offsets of the tile's LMS positions (<= 16 per word)
    // emit: the word in front of the tile, the tile's words, the word behind it -- a key and its context word are cut from
    // three neighbouring words, and a chain of L2 round trips per item (two 24-byte windows of the packed text) was what
    // the pass waited for (round 4: 64 % of its wave cycles parked, SQ_WAIT_ANY)
    __shared__ uint64_t xs[EMIT ? CL_THREADS + 2 : 1];
    uint32_t acc[13];
#pragma unroll
    for (int i = 0; i < 13; i++) acc[i] = 0;

    (void)tiles;
    for (uint64_t tile = tile_lo + blockIdx.x; tile < tile_hi; tile += gridDim.x) {
        uint64_t w = tile * CL_THREADS + threadIdx.x;
        const WordMasks m = word_masks(pk, w, n, words);
        uint64_t T = tile_types(m, tile_cin[tile], lds);
        uint64_t px = __shfl_up(m.x, 1, 64); // (the previous word: from the previous lane, lane 0 loads it)
        if (lane_id() == 0) px = (w > 0 && w <= words) ? pk[w - 1] : 0ull;
        if (EMIT) { // (read after the barriers below; the last reader of the previous tile is behind the loop's last barrier)
            xs[threadIdx.x + 1] = m.x;
            if (threadIdx.x == 0) xs[0] = px;
            if (threadIdx.x == CL_THREADS - 1) xs[CL_THREADS + 1] = w + 1 < words ? pk[w + 1] : 0ull;
        }
        // window mask at LO positions: fields j with win_lo <= 32w + j < win_hi
        uint64_t W = 0;
        if (w < words) {
            const int64_t a = (int64_t)win_lo - (int64_t)(w * 32), b = (int64_t)win_hi - (int64_t)(w * 32);
            const uint64_t below_b = b >= 32 ? LO : (b <= 0 ? 0ull : (LO & (~0ull << (64 - 2 * b)))); // fields j < b
            const uint64_t below_a = a >= 32 ? LO : (a <= 0 ? 0ull : (LO & (~0ull << (64 - 2 * a)))); // fields j < a
            W = below_b & ~below_a;
        }
        T &= W;
        uint64_t lmsmask = T & (gt_prev_mask(m.x, px) >> 1);
        uint32_t cnt = __popcll(lmsmask);
        if (!EMIT) {
            // histograms
            const uint64_t V = W; // win_hi <= n
            uint64_t hi = (m.x >> 1) & LO, lo = m.x & LO;
            uint64_t is[4] = {~hi & ~lo & LO, ~hi & lo, hi & ~lo, hi & lo};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                acc[c] += __popcll(is[c] & V);
                acc[4 + c] += __popcll(is[c] & T);
                acc[8 + c] += __popcll(is[c] & lmsmask);
            }
            // far LMS: position i = 32w + j with i <= far_limit  (far_limit = n - D, or ~0 for all)
            uint64_t farmask = lmsmask;
            if (far_limit != ~0ull) {
                int64_t fj = (int64_t)far_limit - (int64_t)(w * 32); // fields j <= fj are far
                farmask = fj >= 31 ? lmsmask : (fj < 0 ? 0ull : (lmsmask & (~0ull << (62 - 2 * fj))));
            }
            acc[12] += __popcll(farmask);
            // per-tile LMS count
            uint32_t v = cnt;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane_id() == 0) wsum[threadIdx.x >> 6] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t t = 0;
                for (int i = 0; i < CL_THREADS / 64; i++) t += wsum[i];
                tile_cnt[tile] = t;
            }
            __syncthreads();
        } else {
            // exclusive scan of cnt over the tile
            uint32_t inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t o = __shfl_up(inc, d, 64);
                if ((int)lane_id() >= d) inc += o;
            }
            if (lane_id() == 63) wsum[threadIdx.x >> 6] = inc;
            __syncthreads();
            // stage the tile's LMS offsets (13 bits each) in LDS, then write positions and keys coalesced
            uint32_t loff = inc - cnt;
            uint32_t tile_total = 0;
            for (int i = 0; i < CL_THREADS / 64; i++) {
                if (i < (int)(threadIdx.x >> 6)) loff += wsum[i];
                tile_total += wsum[i];
            }
            // highest set bit first = smallest position; the two halves of the word in turn (32-bit bit scans: the 64-bit
            // form was a third of the pass's vector instructions)
            const uint32_t tb = threadIdx.x * 32u;
            for (uint32_t mh = (uint32_t)(lmsmask >> 32); mh;) {
                const uint32_t lz = (uint32_t)__clz((int)mh); // bit 31 - lz of the high half
                mh &= ~(0x80000000u >> lz);
                stage[loff++] = (uint16_t)(tb + (lz >> 1)); // field = base index inside the word (two bits per base)
            }
            for (uint32_t ml = (uint32_t)lmsmask; ml;) {
                const uint32_t lz = (uint32_t)__clz((int)ml);
                ml &= ~(0x80000000u >> lz);
                stage[loff++] = (uint16_t)(tb + 16u + (lz >> 1));
            }
            __syncthreads();
            const uint64_t gbase = tile_cnt[tile];
            const uint64_t tbase = tile * (uint64_t)CL_THREADS * 32;
            // two items per step (idx and idx + CL_THREADS): their LDS reads and stores overlap -- one item per step was a
            // chain of three LDS round trips, and the waves of a workgroup all sat in it at once
            constexpr uint32_t CB = 2u * KISS_KEY_CTX_BASES;
            for (uint32_t idx = threadIdx.x; idx < tile_total; idx += 2 * CL_THREADS) {
                const uint32_t idx2 = idx + CL_THREADS;
                const bool two = idx2 < tile_total;
                uint32_t lb[2] = {stage[idx], two ? stage[idx2] : 0u}; // base offsets inside the tile
                uint64_t key[2];
                uint32_t cw[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    // first 32 bases; the 24 bits round 0 does not sort on carry the preceding 11 bases (kiss_internal.hpp)
                    const uint32_t lw = lb[u] >> 5, sh = (lb[u] & 31u) * 2u;
                    const uint64_t wp = xs[lw], wa = xs[lw + 1], wb = xs[lw + 2];
                    key[u] = (wa << sh) | ((wb >> 1) >> (63u - sh));                 // = kiss_key32(pk, pos)
                    const uint64_t before = (wp << sh) | ((wa >> 1) >> (63u - sh)); // the 32 bases in front of pos
                    cw[u] = ((uint32_t)before & ((1u << CB) - 1u)) | (1u << CB);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (u == 1 && !two) break;
                    const uint64_t pos = tbase + lb[u];
                    if (pos < KISS_KEY_CTX_BASES) cw[u] = kiss_load_ctx_n(pk, pos, KISS_KEY_CTX_BASES); // (the text's first bases)
                    const uint64_t o = gbase + idx + (uint32_t)u * CL_THREADS;
                    lms_pos[o] = (uint32_t)pos;
                    lms_key[o] = (key[u] & ~KISS_KEY_CTX_MASK) | cw[u];
                    if (ghist && pos <= far_limit) {
#pragma unroll
                        for (int p = 0; p < KISS_R0_PASSES; p++)
                            atomicAdd(&hh[p * 256 + ((uint32_t)(key[u] >> (KISS_R0_SHIFT + 8 * p)) & 255u)], 1u);
                    }
                }
            }
            __syncthreads();
        }
    }
    // Totals leave as one row of partial sums per workgroup (k_sum_rows adds the rows up): thousands of workgroups adding
    // to the same 13 / 1280 words with device-scope atomics queue up at the memory side.
    if (EMIT && ghist) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < KISS_R0_PASSES * 256; i += CL_THREADS)
            part[(uint64_t)blockIdx.x * (KISS_R0_PASSES * 256) + i] = hh[i];
    }
    if (!EMIT) {
        __shared__ uint32_t wacc[CL_THREADS / 64][16];
#pragma unroll
        for (int i = 0; i < 13; i++) {
            uint32_t v = acc[i];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane_id() == 0) wacc[threadIdx.x >> 6][i] = v;
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            uint32_t v = 0;
            if (threadIdx.x < 13)
                for (int wv = 0; wv < CL_THREADS / 64; wv++) v += wacc[wv][threadIdx.x];
            part[(uint64_t)blockIdx.x * 16 + threadIdx.x] = v;
        }
    }
}

// out[c] += sum over rows of part[row][c]  (rows x cols words, cols <= gridDim.x * 256; gridDim.y slabs of rows, one
// atomic per slab and column)
__global__ __launch_bounds__(256) void k_sum_rows(const uint32_t *__restrict__ part, uint32_t rows, uint32_t cols,
                                                  uint32_t *__restrict__ out)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const uint32_t per = (rows + gridDim.y - 1) / gridDim.y;
    const uint32_t r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    uint32_t v = 0;
#pragma unroll 8
    for (uint32_t r = r0; r < r1; r++) v += part[(uint64_t)r * cols + c];
    if (v) atomicAdd(&out[c], v);
}

} // namespace

int kiss_pack_text(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n)
{
    const uint64_t words = div_up(n, 32) + 4; // spare zero words: key loads may touch w+1 past the end
    if (words > ctx->pk_words) return KINTERNAL();
    KTimer t(ctx, KISS_HIP_K_PACK, n);
    hipLaunchKernelGGL(k_pack, dim3((unsigned)div_up(words, 256)), dim3(256), 0, ctx->stream, d_S, n, ctx->pk, words);
    KCHECK(hipGetLastError());
    return KISS_HIP_OK;
}

int kiss_classify(kiss_hip_ctx *ctx, uint64_t n, uint64_t depth, uint64_t win_lo, uint64_t win_hi)
{
    if (win_hi > n) win_hi = n;
    if (win_lo > win_hi) win_lo = win_hi;
    ctx->lms_pos_complete = win_lo == 0 && win_hi == n; // (the exchange of a sharded run overwrites the list anyway)
    const uint64_t words = div_up(n, 32);
    const uint64_t tiles = div_up(words, CL_THREADS);
    if (tiles > ctx->n_tiles_cap) return KINTERNAL();
    // far LMS suffixes: p + depth <= n  (all of them when unbounded)
    uint64_t far_limit = ~0ull;
    bool none_far = false;
    if (depth != 0) {
        if (depth > n) none_far = true;
        else far_limit = n - depth;
    }
    // a rank of the sharded form classifies its window only: the count and emit kernels visit the tiles that overlap
    // [win_lo, win_hi) (the carry fold in front of them still runs over the whole text: a type depends on what follows)
    constexpr uint64_t TILE_BASES = (uint64_t)CL_THREADS * 32;
    const uint64_t tile_lo = win_lo / TILE_BASES < tiles ? win_lo / TILE_BASES : tiles;
    uint64_t tile_hi = div_up(win_hi, TILE_BASES) < tiles ? div_up(win_hi, TILE_BASES) : tiles;
    if (tile_hi < tile_lo || win_hi <= win_lo) tile_hi = tile_lo;
    const uint64_t wtiles = tile_hi - tile_lo;
    KTRY(kiss_zero_u32(ctx, ctx->d_counts, 16));
    uint32_t *part = ctx->cl_part; // rows of per-workgroup partial sums (<= 8192 x 1280 words, api.hip)
    {
        KTimer t(ctx, KISS_HIP_K_CLASSIFY, n);
        hipLaunchKernelGGL(k_tile_gp, dim3((unsigned)tiles), dim3(CL_THREADS), 0, ctx->stream, ctx->pk, n, words,
                           ctx->tile_gp);
        if (tiles > 4096) { // two levels: per-chunk (G,P) in tile_cnt (free until the count pass writes it)
            const uint64_t chunks = div_up(tiles, TC_CHUNK);
            hipLaunchKernelGGL(k_tile_cin_chunks, dim3((unsigned)div_up(chunks, 256)), dim3(256), 0, ctx->stream, ctx->tile_gp,
                               tiles, chunks, ctx->tile_cnt);
            hipLaunchKernelGGL(k_tile_cin, dim3(1), dim3(1024), 0, ctx->stream, ctx->tile_cnt, chunks);
            hipLaunchKernelGGL(k_tile_cin_apply, dim3((unsigned)div_up(chunks, 256)), dim3(256), 0, ctx->stream, ctx->tile_gp,
                               tiles, chunks, ctx->tile_cnt);
        } else {
            hipLaunchKernelGGL(k_tile_cin, dim3(1), dim3(1024), 0, ctx->stream, ctx->tile_gp, tiles);
        }
        if (wtiles) {
            unsigned grid = (unsigned)(wtiles < 2048 ? wtiles : 2048);
            hipLaunchKernelGGL(k_classify<false>, dim3(grid), dim3(CL_THREADS), 0, ctx->stream, ctx->pk, n, words, tiles,
                               ctx->tile_gp, ctx->tile_cnt, ctx->d_counts, far_limit, win_lo, win_hi, tile_lo, tile_hi,
                               (uint32_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, part);
            hipLaunchKernelGGL(k_sum_rows, dim3(1, 32), dim3(256), 0, ctx->stream, part, grid, 16u, ctx->d_counts);
        }
        KCHECK(hipGetLastError());
    }
    // offsets of the window's tiles into the emitted list (the list holds the window's LMS suffixes only)
    if (wtiles) KTRY(kiss_scan_u32(ctx, ctx->tile_cnt + tile_lo, ctx->tile_cnt + tile_lo, wtiles));
    KTRY(kiss_readback(ctx, ctx->d_counts, 16));
    for (int i = 0; i < 12; i++) ctx->counts[i] = ctx->h_pinned[i];
    uint64_t m = (uint64_t)ctx->h_pinned[8] + ctx->h_pinned[9] + ctx->h_pinned[10] + ctx->h_pinned[11];
    ctx->m = m;
    ctx->m_far = none_far ? 0 : ctx->h_pinned[12];
    if (m > ctx->m_cap) KTRY(kiss_lms_reserve(ctx, m + m / 64 + 1024)); // more LMS suffixes than DNA-typical
    ctx->rx_ghist_count = 0;
    if (m > 0) {
        // digit histograms for round 0 ride along when the whole LMS list is emitted (not a window of it)
        const bool hist = ctx->rx_ghist && ctx->m_far > 1 && win_lo == 0 && win_hi == n;
        if (hist) KTRY(kiss_zero_u32(ctx, ctx->rx_ghist, 256ull * KISS_R0_PASSES));
        KTimer t(ctx, KISS_HIP_K_CLASSIFY, n);
        const unsigned grid = (unsigned)(wtiles < 8192 ? wtiles : 8192);
        hipLaunchKernelGGL(k_classify<true>, dim3(grid), dim3(CL_THREADS), 0, ctx->stream, ctx->pk, n,
                           words, tiles, ctx->tile_gp, ctx->tile_cnt, ctx->d_counts, far_limit, win_lo, win_hi, tile_lo, tile_hi,
                           ctx->lms_pos, ctx->keyA, hist ? ctx->rx_ghist : (uint32_t *)nullptr, part);
        if (hist)
            hipLaunchKernelGGL(k_sum_rows, dim3(KISS_R0_PASSES, 64), dim3(256), 0, ctx->stream, part, grid,
                               (uint32_t)(KISS_R0_PASSES * 256), ctx->rx_ghist);
        KCHECK(hipGetLastError());
        if (hist) ctx->rx_ghist_count = ctx->m_far; // consumed (or dropped) by the next kiss_radix_sort call
    }
    return KISS_HIP_OK;
}
