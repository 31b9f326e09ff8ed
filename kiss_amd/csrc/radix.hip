// radix.hip -- stable LSD radix sort of (key64 [, seg32], pos32) tuples, 8 bits per pass.
//
// Per pass: tile histogram -> device-wide exclusive scan of the (digit-major) tile
// histograms -> scatter.  The scatter ranks items inside a 16384-item tile with
// wave64 ballots (8 ballots give each lane the set of lanes holding its digit),
// reorders the tile in LDS so that every digit's items leave as one contiguous,
// coalesced run, and writes them out.  Item order inside a tile is
// (wave, round, lane) = memory order, so the sort is stable: equal keys keep the
// ascending text position order the k-ordered LMS contract needs
// (reference tie-break `i < j`, include/biovoltron/algo/sort/kiss1_core.hpp:131-133).
#include "kiss_internal.hpp"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int RX_THREADS = 1024;
constexpr int RX_ITEMS = 16;
constexpr int RX_WAVES = RX_THREADS / 64;
constexpr int RX_WAVE_TILE = RX_ITEMS * 64;        // 1024
constexpr int RX_TILE = RX_THREADS * RX_ITEMS;     // 8192

template <int SRC>
__device__ __forceinline__ uint32_t digit_of(uint64_t key, uint32_t seg, int shift)
{
    return SRC == 0 ? (uint32_t)(key >> shift) & 255u : (seg >> shift) & 255u;
}

template <int SRC>
__global__ __launch_bounds__(RX_THREADS) void k_radix_hist(const uint64_t *__restrict__ key,
                                                          const uint32_t *__restrict__ seg, uint64_t count, int shift,
                                                          uint32_t *__restrict__ tile_hist, uint64_t tiles)
{
    // one private 256-bin histogram per wave (LDS atomics only collide inside a wave), summed at the end
    __shared__ uint32_t h[RX_WAVES][256];
    (void)tiles;
    const int wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < RX_WAVES * 256; i += RX_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint64_t wbase = (uint64_t)blockIdx.x * RX_TILE + (uint64_t)wave * RX_WAVE_TILE;
    if (SRC == 0 && wbase + RX_WAVE_TILE <= count) { // full wave tile: all loads in flight before the first atomic
        ulonglong2 kk[RX_ITEMS / 2];
#pragma unroll
        for (int j = 0; j < RX_ITEMS / 2; j++)
            kk[j] = *reinterpret_cast<const ulonglong2 *>(key + wbase + (uint64_t)j * 128 + 2 * lane_id());
#pragma unroll
        for (int j = 0; j < RX_ITEMS / 2; j++) {
            atomicAdd(&h[wave][(uint32_t)(kk[j].x >> shift) & 255u], 1u);
            atomicAdd(&h[wave][(uint32_t)(kk[j].y >> shift) & 255u], 1u);
        }
    } else if (SRC == 0) { // 16-byte loads (two keys per lane): order is irrelevant for counting
#pragma unroll
        for (int j = 0; j < RX_ITEMS / 2; j++) {
            const uint64_t g = wbase + (uint64_t)j * 128 + 2 * lane_id();
            if (g + 1 < count) {
                const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(key + g);
                atomicAdd(&h[wave][(uint32_t)(kk.x >> shift) & 255u], 1u);
                atomicAdd(&h[wave][(uint32_t)(kk.y >> shift) & 255u], 1u);
            } else if (g < count) {
                atomicAdd(&h[wave][(uint32_t)(key[g] >> shift) & 255u], 1u);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < RX_ITEMS / 4; j++) {
            const uint64_t g = wbase + (uint64_t)j * 256 + 4 * lane_id();
            if (g + 3 < count) {
                const uint4 ss = *reinterpret_cast<const uint4 *>(seg + g);
                atomicAdd(&h[wave][(ss.x >> shift) & 255u], 1u);
                atomicAdd(&h[wave][(ss.y >> shift) & 255u], 1u);
                atomicAdd(&h[wave][(ss.z >> shift) & 255u], 1u);
                atomicAdd(&h[wave][(ss.w >> shift) & 255u], 1u);
            } else {
                for (int e = 0; e < 4; e++)
                    if (g + e < count) atomicAdd(&h[wave][(seg[g + e] >> shift) & 255u], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < RX_WAVES; w++) t += h[w][threadIdx.x];
        tile_hist[(uint64_t)blockIdx.x * 256 + threadIdx.x] = t; // tile-major row
    }
}

// ---- one-sweep form (Adinets & Merrill): the digit histograms of ALL passes come from one read of the input
// (they do not depend on the order of the items), and a scatter tile learns the number of items with its digit in
// earlier tiles by decoupled look-back instead of from a per-tile histogram pass + matrix scan.
// Descriptor word of (tile, digit): [63:62] state (1 = this tile's count, 2 = inclusive prefix over tiles 0..t),
// [61:32] epoch of the pass (stale words of earlier passes read as "not there yet", no clearing), [31:0] value.
// Forward progress: tiles take their number from an atomic ticket, so a tile only ever waits for tiles that started
// before it; the wait is bounded (RX_SPIN_LIMIT polls) and flags an error instead of hanging.
constexpr int RX_MAX_PASSES = 12;
constexpr uint32_t RX_SPIN_LIMIT = 1u << 24;
constexpr uint32_t RX_HIST_LDS_WORDS = 16384; // 64 KiB of histogram copies (the default dynamic-LDS limit)

template <bool HAS_SEG>
__global__ __launch_bounds__(RX_THREADS) void k_radix_hist_all(const uint64_t *__restrict__ key,
                                                              const uint32_t *__restrict__ seg, uint64_t count,
                                                              int key_shift0, int n_key, int n_seg,
                                                              uint32_t *__restrict__ ghist)
{
    // `copies` private histograms of np x 256 bins (dynamic LDS); waves w, w + copies, ... share one
    extern __shared__ uint32_t h[];
    const int np = n_key + n_seg;
    const uint32_t per = (uint32_t)np * 256u;
    const uint32_t copies = RX_HIST_LDS_WORDS / per < (uint32_t)RX_WAVES ? RX_HIST_LDS_WORDS / per : (uint32_t)RX_WAVES;
    for (uint32_t i = threadIdx.x; i < copies * per; i += RX_THREADS) h[i] = 0;
    __syncthreads();
    uint32_t *mine = h + ((threadIdx.x >> 6) % copies) * per;
    const uint64_t base = (uint64_t)blockIdx.x * RX_TILE + threadIdx.x;
    // Neighbouring items often carry the same digit (segment ids of a sorted list, keys inside a repeat): the first
    // lane of every run of equal digits adds the run length, instead of 64 lanes colliding on one LDS word.
    auto add_runs = [&](uint32_t *bins, uint32_t d, bool valid, uint32_t nvalid) {
        if (!HAS_SEG) { // unsorted input (round 0): digits are spread, plain adds are cheaper
            if (valid) atomicAdd(&bins[d], 1u);
            return;
        }
        const uint32_t prev = __shfl_up(d, 1, 64);
        const bool head = valid && (lane_id() == 0 || d != prev);
        const uint64_t hm = __ballot(head);
        if (head) {
            const uint64_t above = lane_id() == 63 ? 0ull : (hm >> (lane_id() + 1));
            const uint32_t next = above ? lane_id() + 1u + (uint32_t)(__ffsll((unsigned long long)above) - 1) : nvalid;
            atomicAdd(&bins[d], next - lane_id());
        }
    };
    if (!HAS_SEG) { // unsorted input: order is irrelevant, two keys per 16-byte load
        const uint64_t tb = (uint64_t)blockIdx.x * RX_TILE;
#pragma unroll 2
        for (int j = 0; j < RX_ITEMS / 2; j++) {
            const uint64_t g = tb + (uint64_t)j * (2 * RX_THREADS) + 2 * threadIdx.x;
            if (g + 1 < count) {
                const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(key + g);
                for (int p = 0; p < n_key; p++) {
                    atomicAdd(&mine[p * 256 + ((uint32_t)(kk.x >> (key_shift0 + 8 * p)) & 255u)], 1u);
                    atomicAdd(&mine[p * 256 + ((uint32_t)(kk.y >> (key_shift0 + 8 * p)) & 255u)], 1u);
                }
            } else if (g < count) {
                const uint64_t k = key[g];
                for (int p = 0; p < n_key; p++) atomicAdd(&mine[p * 256 + ((uint32_t)(k >> (key_shift0 + 8 * p)) & 255u)], 1u);
            }
        }
    } else
#pragma unroll 2
    for (int j = 0; j < RX_ITEMS; j++) {
        const uint64_t g = base + (uint64_t)j * RX_THREADS;
        const bool valid = g < count;
        const uint32_t nvalid = (uint32_t)__popcll(__ballot(valid)); // valid lanes are the low ones
        if (nvalid == 0) continue;
        const uint64_t k = valid ? key[g] : 0ull;
        for (int p = 0; p < n_key; p++) add_runs(mine + p * 256, (uint32_t)(k >> (key_shift0 + 8 * p)) & 255u, valid, nvalid);
        if (HAS_SEG) {
            const uint32_t sg = valid ? seg[g] : 0u;
            for (int q = 0; q < n_seg; q++) add_runs(mine + (n_key + q) * 256, (sg >> (8 * q)) & 255u, valid, nvalid);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < per; i += RX_THREADS) {
        uint32_t t = 0;
        for (uint32_t c = 0; c < copies; c++) t += h[c * per + i];
        if (t) atomicAdd(&ghist[i], t);
    }
}

// ghist[p][0..255] -> exclusive prefix in place (one workgroup of 256 per pass)
__global__ __launch_bounds__(256) void k_radix_digit_bases(uint32_t *__restrict__ ghist)
{
    __shared__ uint32_t ws[4];
    uint32_t *g = ghist + (uint64_t)blockIdx.x * 256;
    const uint32_t v = g[threadIdx.x];
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if ((int)lane_id() >= d) inc += o;
    }
    if (lane_id() == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t pre = inc - v;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pre += ws[w];
    g[threadIdx.x] = pre;
}

__device__ __forceinline__ uint64_t rx_desc_load(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rx_desc_store(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef RX_PROF
__device__ unsigned long long rx_prof[16]; // phase ticks [0..12], look-back hops [13], polls that found nothing [14]
#endif
// number of items with digit d in tiles 0 .. tile-1; publishes this tile's count / inclusive prefix
__device__ __forceinline__ uint32_t rx_lookback(uint64_t *__restrict__ desc, uint32_t tile, uint32_t d, uint32_t tot,
                                                uint64_t epoch, uint32_t *__restrict__ err)
{
    const uint64_t tag = (epoch & 0x3FFFFFFFull) << 32;
    uint64_t *mine = desc + (uint64_t)tile * 256 + d;
    if (tile == 0) {
        rx_desc_store(mine, (2ull << 62) | tag | tot);
        return 0;
    }
    rx_desc_store(mine, (1ull << 62) | tag | tot);
    // The walk is serial on purpose.  Measured with -DRX_PROF: a descriptor load (it has to bypass the per-XCD L2)
    // takes ~0.55 us and a tile walks ~11 predecessors; the tiles resident at one time move as a convoy (they publish
    // their counts at about the same time), so a walk that fetches 8 descriptors per round trip sees them all before
    // any has its prefix, goes three times as far back (32 descriptors) and ends up slower, as did a 16-lane walk.
    uint32_t excl = 0;
    uint32_t t = tile;
#ifdef RX_PROF
    uint32_t hops_ = 0, polls_ = 0;
#endif
    while (t > 0) {
        t--;
        const uint64_t *q = desc + (uint64_t)t * 256 + d;
        uint64_t v = rx_desc_load(q);
        uint32_t spins = 0;
#ifdef RX_PROF
        hops_++;
#endif
        while ((v >> 62) == 0 || (v & (0x3FFFFFFFull << 32)) != tag) {
            if (++spins > RX_SPIN_LIMIT) {
                *err = 1; // a predecessor never published: give up with what we have (writes stay in range)
                rx_desc_store(mine, (2ull << 62) | tag | (excl + tot));
                return excl;
            }
            __builtin_amdgcn_s_sleep(1);
            v = rx_desc_load(q);
#ifdef RX_PROF
            polls_++;
#endif
        }
        excl += (uint32_t)v;
        if ((v >> 62) == 2) break;
    }
#ifdef RX_PROF
    if (d == 0) {
        atomicAdd(&rx_prof[13], (unsigned long long)hops_);
        atomicAdd(&rx_prof[14], (unsigned long long)polls_);
    }
#endif
    rx_desc_store(mine, (2ull << 62) | tag | (excl + tot));
    return excl;
}

// Scatter.  LDS holds ONE 128 KiB staging buffer that the key, position and segment columns pass through in
// turn (each column is reordered locally and leaves as coalesced per-digit runs): one workgroup of 16 waves per CU.
// (Measured: 8192-item tiles with two workgroups per CU are 18 % slower -- run length per digit matters more
//  than overlap between workgroups.)
// ONE = false: tile_off holds the scanned per-(tile, digit) offsets (histogram pass + matrix scan ran before);
// ONE = true : tile_off holds the 256 global digit bases of this pass, tiles are numbered by ticket and find their
//              offsets by look-back over `desc`
#ifdef RX_PROF
#define RX_MARK(i)                                                                                                     \
    do {                                                                                                               \
        if (ONE && threadIdx.x == 0) {                                                                                 \
            const unsigned long long now_ = wall_clock64();                                                            \
            atomicAdd(&rx_prof[i], now_ - t_prev_);                                                                    \
            t_prev_ = now_;                                                                                            \
        }                                                                                                              \
    } while (0)
#else
#define RX_MARK(i)
#endif

template <int SRC, bool HAS_SEG, bool ONE>
__global__ __launch_bounds__(RX_THREADS, 4) void k_radix_scatter(const uint64_t *__restrict__ key_in,
                                                                const uint32_t *__restrict__ seg_in,
                                                                const uint32_t *__restrict__ pos_in,
                                                                uint64_t *__restrict__ key_out,
                                                                uint32_t *__restrict__ seg_out,
                                                                uint32_t *__restrict__ pos_out, uint64_t count,
                                                                int shift, const uint32_t *__restrict__ tile_off,
                                                                uint64_t tiles, uint64_t *__restrict__ desc,
                                                                uint32_t *__restrict__ ctl, uint32_t ticket_base,
                                                                uint64_t epoch)
{
    __shared__ uint64_t stage64[RX_TILE];
    __shared__ uint32_t wcnt[RX_WAVES][256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t wsum[RX_WAVES + 1];
    __shared__ uint32_t s_tile;
    uint32_t *stage32 = reinterpret_cast<uint32_t *>(stage64);
    (void)tiles;

    const int wave = threadIdx.x >> 6;
    const uint32_t lane = lane_id();
#ifdef RX_PROF
    unsigned long long t_prev_ = wall_clock64();
#endif
    if (ONE) {
        if (threadIdx.x == 0) s_tile = atomicAdd(&ctl[0], 1u) - ticket_base;
        __syncthreads();
    }
    RX_MARK(0);
    const uint32_t tile = ONE ? s_tile : blockIdx.x;
    const uint64_t tile_base = (uint64_t)tile * RX_TILE;
    const uint32_t tile_count = (uint32_t)((count - tile_base) < (uint64_t)RX_TILE ? (count - tile_base) : RX_TILE);

    if (threadIdx.x < 256) {
#pragma unroll
        for (int w = 0; w < RX_WAVES; w++) wcnt[w][threadIdx.x] = 0;
    }
    __syncthreads();

    uint64_t k[RX_ITEMS];
    uint32_t s[RX_ITEMS];
    uint32_t p[RX_ITEMS];
    uint32_t rk[RX_ITEMS]; // (digit << 16) | rank inside the wave among equal digits; 0xFFFFFFFF = invalid

    const uint32_t wbase = (uint32_t)wave * RX_WAVE_TILE + lane;
    // the positions are not needed before the last staging step: their loads are issued after the ranking (below), so
    // that the ranking waits for 8 bytes per item instead of 12 and the positions arrive under the look-back
    if (tile_count == RX_TILE) { // full tile: no per-item bounds checks around the loads
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) {
            const uint64_t g = tile_base + wbase + (uint32_t)j * 64;
            k[j] = key_in[g];
            s[j] = HAS_SEG ? seg_in[g] : 0u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) {
            const uint32_t li = wbase + (uint32_t)j * 64;
            const bool valid = li < tile_count;
            const uint64_t g = tile_base + li;
            k[j] = valid ? key_in[g] : 0ull;
            s[j] = (HAS_SEG && valid) ? seg_in[g] : 0u;
        }
    }
#ifdef RX_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RX_MARK(1);
#endif
    // Rank inside the wave: lanes holding the same digit = AND over the 8 digit bits of (bit ? ballot : ~ballot); with
    // sel = 0 / ~0 that is one three-input boolean op per half and bit (v_bitop3: acc & ~(ballot ^ sel)).  The rank
    // phase is VALU-bound (phase profile), so the loop is written instruction by instruction: the
    // running count of the digit is READ by all its lanes (one broadcast) and written back by the lowest one, instead
    // of an atomic with return in a leader plus a cross-lane broadcast.
    uint32_t *const wc = wcnt[wave];
    auto rank_items = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) {
            const bool valid = FULL || wbase + (uint32_t)j * 64 < tile_count;
            const uint32_t d = digit_of<SRC>(k[j], s[j], shift);
            uint32_t plo = ~0u, phi = ~0u;
            if (!FULL) {
                const uint64_t vm = __ballot(valid);
                plo = (uint32_t)vm;
                phi = (uint32_t)(vm >> 32);
            }
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const int sel = __builtin_amdgcn_sbfe((int)d, b, 1); // 0 or ~0
                const uint64_t bm = __ballot(sel < 0);
                plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bm, (uint32_t)sel, 0x90);
                phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bm >> 32), (uint32_t)sel, 0x90);
            }
            // equal digits in lower lanes; the lowest lane of the group moves the running count on
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const uint32_t old = wc[d];
            if (below == 0 && valid) wc[d] = old + (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
            rk[j] = valid ? ((d << 16) | (old + below)) : 0xFFFFFFFFu;
        }
    };
    if (tile_count == RX_TILE) rank_items(std::true_type{});
    else rank_items(std::false_type{});
    if (tile_count == RX_TILE) {
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) p[j] = pos_in[tile_base + wbase + (uint32_t)j * 64];
    } else {
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++) {
            const uint32_t li = wbase + (uint32_t)j * 64;
            p[j] = li < tile_count ? pos_in[tile_base + li] : 0u;
        }
    }
    RX_MARK(2);
    __syncthreads();
    RX_MARK(3);

    // per digit (threads 0..255): totals over waves, exclusive prefix over waves, exclusive scan over digits
    {
        const uint32_t d = threadIdx.x & 255u;
        const bool dig = threadIdx.x < 256;
        uint32_t c[RX_WAVES];
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < RX_WAVES; w++) {
            c[w] = dig ? wcnt[w][d] : 0u;
            tot += c[w];
        }
        uint32_t inc = tot;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            uint32_t o = __shfl_up(inc, dd, 64);
            if ((int)lane >= dd) inc += o;
        }
        if (lane == 63 && dig) wsum[wave] = inc;
        __syncthreads();
        RX_MARK(4);
        uint32_t start = inc - tot;
        if (dig) {
            for (int w = 0; w < wave; w++) start += wsum[w];
            uint32_t run = start;
#pragma unroll
            for (int w = 0; w < RX_WAVES; w++) {
                wcnt[w][d] = run;
                run += c[w];
            }
        }
        __syncthreads(); // local offsets are known: the other waves stage their items while waves 0..3 look back
        RX_MARK(5);
        if (dig) {
            if (ONE) gbase[d] = tile_off[d] + rx_lookback(desc, tile, d, tot, epoch, &ctl[1]) - start;
            else gbase[d] = tile_off[(uint64_t)tile * 256 + d] - start;
        }
    }
    RX_MARK(6);

    // local positions of this thread's items
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++)
        if (rk[j] != 0xFFFFFFFFu) rk[j] = wcnt[wave][rk[j] >> 16] + (rk[j] & 0xFFFFu);
    // global offsets of the output slots this thread will write: slot idx = threadIdx.x + 256 * r
    uint32_t gofs[RX_ITEMS];

    // ---- the column that carries the digit goes first (key for SRC == 0, segment for SRC == 1)
    if (SRC == 0) {
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++)
            if (rk[j] != 0xFFFFFFFFu) stage64[rk[j]] = k[j];
        lds_barrier();
        RX_MARK(7);
#pragma unroll
        for (int r = 0; r < RX_ITEMS; r++) {
            const uint32_t idx = threadIdx.x + (uint32_t)r * RX_THREADS;
            if (idx < tile_count) {
                const uint64_t kk = stage64[idx];
                gofs[r] = gbase[(uint32_t)(kk >> shift) & 255u] + idx;
                key_out[gofs[r]] = kk;
            }
        }
        RX_MARK(8);
        lds_barrier();
        RX_MARK(9);
        if (HAS_SEG) {
#pragma unroll
            for (int j = 0; j < RX_ITEMS; j++)
                if (rk[j] != 0xFFFFFFFFu) stage32[rk[j]] = s[j];
            lds_barrier();
#pragma unroll
            for (int r = 0; r < RX_ITEMS; r++) {
                const uint32_t idx = threadIdx.x + (uint32_t)r * RX_THREADS;
                if (idx < tile_count) seg_out[gofs[r]] = stage32[idx];
            }
            lds_barrier();
        }
    } else {
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++)
            if (rk[j] != 0xFFFFFFFFu) stage32[rk[j]] = s[j];
        lds_barrier();
#pragma unroll
        for (int r = 0; r < RX_ITEMS; r++) {
            const uint32_t idx = threadIdx.x + (uint32_t)r * RX_THREADS;
            if (idx < tile_count) {
                const uint32_t ss = stage32[idx];
                gofs[r] = gbase[(ss >> shift) & 255u] + idx;
                seg_out[gofs[r]] = ss;
            }
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < RX_ITEMS; j++)
            if (rk[j] != 0xFFFFFFFFu) stage64[rk[j]] = k[j];
        lds_barrier();
#pragma unroll
        for (int r = 0; r < RX_ITEMS; r++) {
            const uint32_t idx = threadIdx.x + (uint32_t)r * RX_THREADS;
            if (idx < tile_count) key_out[gofs[r]] = stage64[idx];
        }
        lds_barrier();
    }
    // ---- positions
#pragma unroll
    for (int j = 0; j < RX_ITEMS; j++)
        if (rk[j] != 0xFFFFFFFFu) stage32[rk[j]] = p[j];
    lds_barrier();
    RX_MARK(10);
#pragma unroll
    for (int r = 0; r < RX_ITEMS; r++) {
        const uint32_t idx = threadIdx.x + (uint32_t)r * RX_THREADS;
        if (idx < tile_count) pos_out[gofs[r]] = stage32[idx];
    }
#ifdef RX_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RX_MARK(11);
#endif
}

// ---- digit-major exclusive scan of the tile-major histogram matrix hist[tile][256] ------------------
// offset(tile, d) = sum over digits d' < d of total[d'] + sum over tiles t' < tile of hist[t'][d].
// k_col_sum: per chunk of RX_CHUNK tiles, column sums -> digit-major chunk matrix csum[d][chunk];
// generic exclusive scan over csum (digit-major linear order is exactly the order wanted);
// k_col_apply: per chunk, running column prefix written back over hist.  Rows are coalesced 1 KiB.
constexpr int RX_CHUNK = 64;

__global__ __launch_bounds__(256) void k_col_sum(const uint32_t *__restrict__ hist, uint64_t tiles, uint64_t chunks,
                                                 uint32_t *__restrict__ csum)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * RX_CHUNK;
    const uint64_t t1 = t0 + RX_CHUNK < tiles ? t0 + RX_CHUNK : tiles;
    uint32_t s = 0;
    uint64_t t = t0;
    for (; t + 8 <= t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = hist[(t + u) * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; t < t1; t++) s += hist[t * 256 + threadIdx.x];
    csum[(uint64_t)threadIdx.x * chunks + blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_col_apply(uint32_t *hist, uint64_t tiles, uint64_t chunks,
                                                   const uint32_t *__restrict__ coff)
{
    const uint64_t t0 = (uint64_t)blockIdx.x * RX_CHUNK;
    const uint64_t t1 = t0 + RX_CHUNK < tiles ? t0 + RX_CHUNK : tiles;
    uint32_t run = coff[(uint64_t)threadIdx.x * chunks + blockIdx.x];
    uint64_t t = t0;
    for (; t + 8 <= t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = hist[(t + u) * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            hist[(t + u) * 256 + threadIdx.x] = run;
            run += v[u];
        }
    }
    for (; t < t1; t++) {
        uint32_t v = hist[t * 256 + threadIdx.x];
        hist[t * 256 + threadIdx.x] = run;
        run += v;
    }
}

int radix_offsets(kiss_hip_ctx *ctx, uint64_t tiles)
{
    const uint64_t chunks = div_up(tiles, RX_CHUNK);
    uint32_t *csum = ctx->tile_hist + 256 * tiles; // room reserved behind the matrix
    if (256 * tiles + 256 * chunks > ctx->tile_hist_cap) return KINTERNAL();
    {
        KTimer t(ctx, KISS_HIP_K_SCAN, 256 * tiles);
        hipLaunchKernelGGL(k_col_sum, dim3((unsigned)chunks), dim3(256), 0, ctx->stream, ctx->tile_hist, tiles, chunks,
                           csum);
        KCHECK(hipGetLastError());
    }
    KTRY(kiss_scan_u32(ctx, csum, csum, 256 * chunks));
    {
        KTimer t(ctx, KISS_HIP_K_SCAN, 256 * tiles);
        hipLaunchKernelGGL(k_col_apply, dim3((unsigned)chunks), dim3(256), 0, ctx->stream, ctx->tile_hist, tiles,
                           chunks, csum);
        KCHECK(hipGetLastError());
    }
    return KISS_HIP_OK;
}

template <int SRC, bool HAS_SEG>
int radix_pass(kiss_hip_ctx *ctx, RadixBufs &b, int src, uint64_t count, int shift, const uint32_t *pos_in)
{
    const uint64_t tiles = div_up(count, RX_TILE);
    const int dst = src ^ 1;
    {
        KTimer t(ctx, KISS_HIP_K_RADIX_HIST, count);
        hipLaunchKernelGGL((k_radix_hist<SRC>), dim3((unsigned)tiles), dim3(RX_THREADS), 0, ctx->stream, b.key[src],
                           b.seg[src], count, shift, ctx->tile_hist, tiles);
        KCHECK(hipGetLastError());
    }
    KTRY(radix_offsets(ctx, tiles));
    {
        KTimer t(ctx, KISS_HIP_K_RADIX_SCATTER, count);
        hipLaunchKernelGGL((k_radix_scatter<SRC, HAS_SEG, false>), dim3((unsigned)tiles), dim3(RX_THREADS), 0, ctx->stream,
                           b.key[src], b.seg[src], pos_in, b.key[dst], b.seg[dst], b.pos[dst], count, shift,
                           ctx->tile_hist, tiles, nullptr, nullptr, 0u, 0ull);
        KCHECK(hipGetLastError());
    }
    return KISS_HIP_OK;
}

// one-sweep pass p (digit bases already in rx_ghist[p])
template <int SRC, bool HAS_SEG>
int radix_pass_one(kiss_hip_ctx *ctx, RadixBufs &b, int src, uint64_t count, int shift, int p, const uint32_t *pos_in)
{
    const uint64_t tiles = div_up(count, RX_TILE);
    const int dst = src ^ 1;
    KTimer t(ctx, KISS_HIP_K_RADIX_SCATTER, count);
    ctx->rx_epoch++;
    hipLaunchKernelGGL((k_radix_scatter<SRC, HAS_SEG, true>), dim3((unsigned)tiles), dim3(RX_THREADS), 0, ctx->stream,
                       b.key[src], b.seg[src], pos_in, b.key[dst], b.seg[dst], b.pos[dst], count, shift,
                       ctx->rx_ghist + 256 * p, tiles, ctx->rx_desc, ctx->rx_ctl, ctx->rx_ticket_base, ctx->rx_epoch);
    KCHECK(hipGetLastError());
    ctx->rx_ticket_base += (uint32_t)tiles;
    return KISS_HIP_OK;
}

} // namespace

// after a stream synchronisation: did any look-back give up?  (never observed; the wait is bounded so that a bug
// shows up as an error code instead of a hung GPU)
int kiss_radix_check(kiss_hip_ctx *ctx)
{
    if (!ctx->rx_ctl) return KISS_HIP_OK;
#ifdef RX_PROF
    {
        unsigned long long h[16], z[16] = {0};
        KCHECK(hipStreamSynchronize(ctx->stream));
        KCHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(rx_prof), sizeof h));
        KCHECK(hipMemcpyToSymbol(HIP_SYMBOL(rx_prof), z, sizeof z));
        unsigned long long tot = 0;
        for (int i = 0; i < 13; i++) tot += h[i];
        fprintf(stderr, "[rx_prof] phase ticks (thread 0 of every tile), total %llu:", tot);
        for (int i = 0; i < 13; i++) fprintf(stderr, " %d:%.1f%%", i, tot ? 100.0 * (double)h[i] / (double)tot : 0.0);
        fprintf(stderr, "  | look-back of digit 0: %llu hops, %llu empty polls\n", h[13], h[14]);
    }
#endif
    uint32_t err = 0;
    KCHECK(hipMemcpyAsync(&err, ctx->rx_ctl + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));
    if (err) {
        (void)kiss_zero_u32(ctx, ctx->rx_ctl + 1, 1);
        return KINTERNAL();
    }
    return KISS_HIP_OK;
}

int kiss_radix_sort(kiss_hip_ctx *ctx, RadixBufs &b, uint64_t count, int key_lo_bit, int seg_bits, int *result_idx)
{
    int cur = 0;
    if (count > 1) {
        const bool has_seg = seg_bits > 0;
        const int shift0 = key_lo_bit & ~7;
        const int n_key = (64 - shift0) / 8, n_seg = (seg_bits + 7) / 8;
        const uint64_t tiles = div_up(count, RX_TILE);
        const bool no_one = ctx->opts.no_onesweep; // (hooks build)
        const bool one = !no_one && ctx->rx_desc && tiles >= 2 && tiles <= ctx->rx_tiles_cap && n_key + n_seg <= RX_MAX_PASSES;
        // digit counts that came with the keys (classify.hip, round 0 only)
        const bool counted = one && ctx->rx_ghist_count == count && !has_seg && shift0 == KISS_R0_SHIFT && b.first_pos;
        ctx->rx_ghist_count = 0;
        if (counted) {
            KTimer t(ctx, KISS_HIP_K_RADIX_HIST, 0);
            hipLaunchKernelGGL(k_radix_digit_bases, dim3((unsigned)n_key), dim3(256), 0, ctx->stream, ctx->rx_ghist);
            KCHECK(hipGetLastError());
        } else if (one) {
            KTimer t(ctx, KISS_HIP_K_RADIX_HIST, count);
            KTRY(kiss_zero_u32(ctx, ctx->rx_ghist, 256ull * (uint64_t)(n_key + n_seg)));
            const uint32_t per = 256u * (uint32_t)(n_key + n_seg);
            const uint32_t copies = RX_HIST_LDS_WORDS / per < (uint32_t)RX_WAVES ? RX_HIST_LDS_WORDS / per : (uint32_t)RX_WAVES;
            const size_t hist_lds = (size_t)copies * per * sizeof(uint32_t);
            if (has_seg)
                hipLaunchKernelGGL((k_radix_hist_all<true>), dim3((unsigned)tiles), dim3(RX_THREADS), hist_lds, ctx->stream, b.key[0],
                                   b.seg[0], count, shift0, n_key, n_seg, ctx->rx_ghist);
            else
                hipLaunchKernelGGL((k_radix_hist_all<false>), dim3((unsigned)tiles), dim3(RX_THREADS), hist_lds, ctx->stream, b.key[0],
                                   nullptr, count, shift0, n_key, n_seg, ctx->rx_ghist);
            hipLaunchKernelGGL(k_radix_digit_bases, dim3((unsigned)(n_key + n_seg)), dim3(256), 0, ctx->stream, ctx->rx_ghist);
            KCHECK(hipGetLastError());
        }
        int p = 0;
        for (int shift = shift0; shift < 64; shift += 8, p++) {
            const uint32_t *pin = (p == 0 && b.first_pos) ? b.first_pos : b.pos[cur];
            if (one) {
                if (has_seg) KTRY((radix_pass_one<0, true>(ctx, b, cur, count, shift, p, pin)));
                else KTRY((radix_pass_one<0, false>(ctx, b, cur, count, shift, p, pin)));
            } else {
                if (has_seg) KTRY((radix_pass<0, true>(ctx, b, cur, count, shift, pin)));
                else KTRY((radix_pass<0, false>(ctx, b, cur, count, shift, pin)));
            }
            cur ^= 1;
        }
        for (int shift = 0; shift < seg_bits; shift += 8, p++) {
            const uint32_t *pin = (p == 0 && b.first_pos) ? b.first_pos : b.pos[cur];
            if (one) KTRY((radix_pass_one<1, true>(ctx, b, cur, count, shift, p, pin)));
            else KTRY((radix_pass<1, true>(ctx, b, cur, count, shift, pin)));
            cur ^= 1;
        }
    }
    if (count <= 1 && b.first_pos && count == 1) // nothing moved: the caller still expects the positions in pos[0]
        KCHECK(hipMemcpyAsync(b.pos[0], b.first_pos, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    *result_idx = cur;
    return KISS_HIP_OK;
}
