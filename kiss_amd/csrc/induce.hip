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

namespace {

constexpr int IN_THREADS = 256;
constexpr int IN_ITEMS = 8;
constexpr int IN_WAVES = IN_THREADS / 64;
constexpr int IN_WAVE_TILE = IN_ITEMS * 64;      // 512
constexpr int IN_TILE = IN_THREADS * IN_ITEMS;   // 2048
constexpr int SM_THREADS = 1024;
constexpr uint64_t SMALL_MAX = 8192;             // rounds up to this size run in the single-workgroup kernel

// class of a source item: 0..3 = append v-1 to bucket of that base, 4 = nothing to append
__device__ __forceinline__ uint32_t item_class(const uint64_t *__restrict__ pk, const uint32_t *srcP, uint32_t *srcC,
                                               int64_t phys, uint32_t emitmask, uint32_t *v_out, uint32_t *ctx_out)
{
    uint32_t c = srcC[phys];
    uint32_t v = srcP[phys];
    *v_out = v;
    if (c == KISS_EMPTY_CTX) {
        if (v == 0) {
            *ctx_out = c;
            return 4u;
        }
        c = kiss_load_ctx(pk, v);
        srcC[phys] = c; // keep the refreshed word: the scatter pass (and the other sweep) reuse it
    }
    *ctx_out = c;
    uint32_t pc = c & 3u;
    return ((emitmask >> pc) & 1u) ? pc : 4u;
}

// ---- pass 1: per-tile class counts ---------------------------------------------------
__global__ __launch_bounds__(IN_THREADS) void k_induce_count(const uint64_t *__restrict__ pk, const uint32_t *srcP,
                                                            uint32_t *srcC, int64_t beg, uint64_t N, int dir,
                                                            uint32_t emitmask, uint32_t *__restrict__ counts,
                                                            uint64_t tiles)
{
    __shared__ uint32_t tc[4];
    if (threadIdx.x < 4) tc[threadIdx.x] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * IN_TILE + (uint64_t)wave * IN_WAVE_TILE + lane_id();
    uint32_t run[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < IN_ITEMS; j++) {
        uint64_t i = base + (uint64_t)j * 64;
        uint32_t cls = 4u, v, cw;
        if (i < N) cls = item_class(pk, srcP, srcC, beg + (int64_t)dir * (int64_t)i, emitmask, &v, &cw);
#pragma unroll
        for (int c = 0; c < 4; c++) run[c] += (uint32_t)__popcll(__ballot(cls == (uint32_t)c));
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (run[c]) atomicAdd(&tc[c], run[c]);
    }
    __syncthreads();
    if (threadIdx.x < 4) counts[(uint64_t)threadIdx.x * tiles + blockIdx.x] = tc[threadIdx.x];
}

struct DstPos {
    int64_t p[4];
};

// ---- pass 2: stable scatter ------------------------------------------------------------
__global__ __launch_bounds__(IN_THREADS) void k_induce_scatter(const uint64_t *__restrict__ pk, const uint32_t *srcP,
                                                              uint32_t *srcC, int64_t beg, uint64_t N, int dir,
                                                              uint32_t emitmask, const uint32_t *__restrict__ ex,
                                                              uint64_t tiles, DstPos dst, uint32_t *SA, uint32_t *CTX)
{
    __shared__ uint32_t wtot[IN_WAVES][4];
    const int wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * IN_TILE + (uint64_t)wave * IN_WAVE_TILE + lane_id();
    uint32_t run[4] = {0, 0, 0, 0};
    uint32_t vv[IN_ITEMS], cc[IN_ITEMS], rr[IN_ITEMS]; // rr = (class << 28) | rank in wave
#pragma unroll
    for (int j = 0; j < IN_ITEMS; j++) {
        uint64_t i = base + (uint64_t)j * 64;
        uint32_t cls = 4u, v = 0, cw = 0;
        if (i < N) cls = item_class(pk, srcP, srcC, beg + (int64_t)dir * (int64_t)i, emitmask, &v, &cw);
        uint32_t myrank = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            uint64_t mk = __ballot(cls == (uint32_t)c);
            if (cls == (uint32_t)c) myrank = run[c] + (uint32_t)__popcll(mk & lanemask_lt());
            run[c] += (uint32_t)__popcll(mk);
        }
        vv[j] = v;
        cc[j] = cw;
        rr[j] = (cls << 28) | myrank;
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int c = 0; c < 4; c++) wtot[wave][c] = run[c];
    }
    __syncthreads();
    uint32_t off[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        uint32_t o = ex[(uint64_t)c * tiles + blockIdx.x] - ex[(uint64_t)c * tiles];
        for (int w = 0; w < wave; w++) o += wtot[w][c];
        off[c] = o;
    }
#pragma unroll
    for (int j = 0; j < IN_ITEMS; j++) {
        uint32_t cls = rr[j] >> 28;
        if (cls < 4u) {
            int64_t d = dst.p[cls] + (int64_t)dir * (int64_t)(off[cls] + (rr[j] & 0x0FFFFFFFu));
            SA[d] = vv[j] - 1u;
            CTX[d] = cc[j] >> 2;
        }
    }
}

// totals per class from the scanned counts (4*tiles+1 entries, the extra last input entry is 0)
__global__ void k_induce_totals(const uint32_t *__restrict__ ex, uint64_t tiles, uint32_t *__restrict__ out)
{
    if (threadIdx.x < 4) out[threadIdx.x] = ex[(uint64_t)(threadIdx.x + 1) * tiles] - ex[(uint64_t)threadIdx.x * tiles];
}

// ---- single-workgroup chain kernel -----------------------------------------------------
// Processes one source segment and, when selfclass >= 0, keeps processing what it appended to that
// class until nothing is appended any more.  out[0..3] = items appended per class, out[4] = rounds.
__global__ __launch_bounds__(SM_THREADS) void k_induce_small(const uint64_t *__restrict__ pk, const uint32_t *srcP0,
                                                            uint32_t *srcC0, int64_t beg0, uint64_t N0, int dir,
                                                            uint32_t emitmask, int selfclass, DstPos dst, uint32_t *SA,
                                                            uint32_t *CTX, uint32_t *out)
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
            if (i < N) cls = item_class(pk, srcP, srcC, beg + (int64_t)dir * (int64_t)i, emitmask, &v, &cw);
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
                CTX[d] = cw >> 2;
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
             int selfclass, uint64_t tot[4], bool *chain_done)
{
    kiss_hip_ctx *ctx = sw.ctx;
    for (int c = 0; c < 4; c++) tot[c] = 0;
    *chain_done = false;
    if (N == 0 || emitmask == 0) {
        *chain_done = true;
        return KISS_HIP_OK;
    }
    DstPos dp;
    for (int c = 0; c < 4; c++) dp.p[c] = sw.pos[c];
    ctx->stats.induce_passes++;
    if (N <= SMALL_MAX) {
        {
            KTimer t(ctx, KISS_HIP_K_INDUCE_SMALL, N);
            hipLaunchKernelGGL(k_induce_small, dim3(1), dim3(SM_THREADS), 0, ctx->stream, ctx->pk, srcP, srcC, beg, N,
                               sw.dir, emitmask, selfclass, dp, sw.SA, ctx->CTX, ctx->d_small);
            KCHECK(hipGetLastError());
        }
        KCHECK(hipMemcpyAsync(ctx->h_pinned, ctx->d_small, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
        for (int c = 0; c < 4; c++) tot[c] = ctx->h_pinned[c];
        *chain_done = true;
    } else {
        const uint64_t tiles = div_up(N, IN_TILE);
        if (4 * tiles + 1 > ctx->ind_tiles_cap) return KISS_HIP_E_INTERNAL;
        {
            KTimer t(ctx, KISS_HIP_K_INDUCE_COUNT, N);
            KCHECK(hipMemsetAsync(ctx->ind_counts + 4 * tiles, 0, sizeof(uint32_t), ctx->stream));
            hipLaunchKernelGGL(k_induce_count, dim3((unsigned)tiles), dim3(IN_THREADS), 0, ctx->stream, ctx->pk, srcP,
                               srcC, beg, N, sw.dir, emitmask, ctx->ind_counts, tiles);
            KCHECK(hipGetLastError());
        }
        KTRY(kiss_scan_u32(ctx, ctx->ind_counts, ctx->ind_counts, 4 * tiles + 1));
        {
            KTimer t(ctx, KISS_HIP_K_INDUCE_SCATTER, N);
            hipLaunchKernelGGL(k_induce_scatter, dim3((unsigned)tiles), dim3(IN_THREADS), 0, ctx->stream, ctx->pk, srcP,
                               srcC, beg, N, sw.dir, emitmask, ctx->ind_counts, tiles, dp, sw.SA, ctx->CTX);
            hipLaunchKernelGGL(k_induce_totals, dim3(1), dim3(64), 0, ctx->stream, ctx->ind_counts, tiles,
                               ctx->d_small);
            KCHECK(hipGetLastError());
        }
        KCHECK(hipMemcpyAsync(ctx->h_pinned, ctx->d_small, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
        for (int c = 0; c < 4; c++) tot[c] = ctx->h_pinned[c];
    }
    for (int c = 0; c < 4; c++) sw.pos[c] += (int64_t)sw.dir * (int64_t)tot[c];
    return KISS_HIP_OK;
}

} // namespace

int kiss_induce(kiss_hip_ctx *ctx, uint64_t n, uint32_t *d_SA)
{
    ctx->stats.induce_passes = 0;
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
    if (start[4] != n + 1) return KISS_HIP_E_INTERNAL;

    // SA[0] = n (the sentinel suffix); a one-item source {n} with an empty context word starts the L sweep
    uint32_t *seed = ctx->d_small + 32;
    ctx->h_pinned[32] = (uint32_t)n;
    ctx->h_pinned[33] = KISS_EMPTY_CTX;
    KCHECK(hipMemcpyAsync(seed, ctx->h_pinned + 32, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    KCHECK(hipMemcpyAsync(d_SA, ctx->h_pinned + 32, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    KCHECK(hipStreamSynchronize(ctx->stream));

    uint64_t tot[4];
    bool done;

    // ---------------- L sweep: left to right ----------------
    Sweep L{ctx, d_SA, +1, {(int64_t)start[0], (int64_t)start[1], (int64_t)start[2], (int64_t)start[3]}};
    KTRY(run_pass(L, seed, seed + 1, 0, 1, 0xFu, -1, tot, &done));
    for (int c = 0; c < 4; c++) {
        int64_t a = (int64_t)start[c];
        const uint32_t mask_ge = (0xFu << c) & 0xFu; // L-type source of char c: v-1 is L-type iff S[v-1] >= c
        while (L.pos[c] > a) {
            uint64_t N = (uint64_t)(L.pos[c] - a);
            KTRY(run_pass(L, d_SA, ctx->CTX, a, N, mask_ge, c, tot, &done));
            a = done ? L.pos[c] : a + (int64_t)N;
        }
        if ((uint64_t)L.pos[c] != start[c] + cntL[c]) return KISS_HIP_E_INTERNAL;
        if (cntLMS[c]) {
            const uint32_t mask_gt = (0xFu << (c + 1)) & 0xFu; // S[v-1] > c for every LMS suffix
            KTRY(run_pass(L, ctx->lmsP, ctx->lmsC, (int64_t)lms_start[c], cntLMS[c], mask_gt, -1, tot, &done));
        }
    }

    // ---------------- S sweep: right to left ----------------
    Sweep S{ctx, d_SA, -1, {(int64_t)start[1] - 1, (int64_t)start[2] - 1, (int64_t)start[3] - 1, (int64_t)start[4] - 1}};
    for (int c = 3; c >= 0; c--) {
        int64_t hi = (int64_t)start[c + 1]; // exclusive upper end of the not yet processed S-part
        const uint32_t mask_le = (1u << (c + 1)) - 1u; // S-type source of char c: v-1 is S-type iff S[v-1] <= c
        while (S.pos[c] + 1 < hi) {
            uint64_t N = (uint64_t)(hi - (S.pos[c] + 1));
            KTRY(run_pass(S, d_SA, ctx->CTX, hi - 1, N, mask_le, c, tot, &done));
            hi = done ? S.pos[c] + 1 : hi - (int64_t)N;
        }
        if ((uint64_t)(S.pos[c] + 1) != start[c] + cntL[c]) return KISS_HIP_E_INTERNAL;
        if (cntL[c] && c > 0) {
            const uint32_t mask_lt = (1u << c) - 1u; // L-type source of char c: v-1 is S-type iff S[v-1] < c
            KTRY(run_pass(S, d_SA, ctx->CTX, (int64_t)(start[c] + cntL[c]) - 1, cntL[c], mask_lt, -1, tot, &done));
        }
    }
    return KISS_HIP_OK;
}
