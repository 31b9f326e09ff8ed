// general.hip -- exact suffix array of a text over the byte alphabet.
//
// Replaces the general-alphabet entry of the reference, KISS1Sorter::get_suffix_array -> kiss1_suffix_array
// (include/biovoltron/algo/sort/kiss1_core.hpp:270-311), which is reachable only from the reference's tests and
// experiments (the command line always takes the DNA path).  For that entry only the k-order PROPERTY is defined
// (its small_cmp has no index tie-break, so its exact output depends on the C++ library's std::sort); the exact
// suffix array returned here satisfies the property for every k.
//
// No induction here (sigma = 256 would make every sweep a 256-way partition); the doubling machinery of the
// PREFIX_DOUBLING path does all the work:
//   key(p) = the first 7 characters of suffix p, each stored as c + 1 in 9 bits (0 = past the end, so a suffix that
//            is a proper prefix of another sorts first), stable LSD radix sort of (key, p) over 63 bits;
//   SA[0] = n, SA[1 + i] = sorted position i, group heads where the key changes;
//   rank doubling from h = 7 on the tied suffixes (lms_sort.hip: kiss_exact_refine with the heads given).
#include "kiss_internal.hpp"
#include <cstdlib>
#include <cstring>

namespace {

constexpr int GA_THREADS = 256;
constexpr uint32_t GA_CHARS = 7; // characters per 63-bit key

__global__ __launch_bounds__(GA_THREADS) void k_ga_keys(const uint8_t *__restrict__ S, uint64_t n, uint64_t *__restrict__ key,
                                                       uint32_t *__restrict__ pos)
{
    const uint64_t p = (uint64_t)blockIdx.x * GA_THREADS + threadIdx.x;
    if (p >= n) return;
    uint64_t k = 0;
#pragma unroll
    for (uint32_t j = 0; j < GA_CHARS; j++) {
        const uint64_t c = p + j < n ? (uint64_t)S[p + j] + 1ull : 0ull;
        k = (k << 9) | c;
    }
    key[p] = k;
    pos[p] = (uint32_t)p;
}

// SA[0] = n (the empty suffix), SA[1 + i] = pos[i]; heads[1 + i] = key[i] differs from key[i - 1]
__global__ __launch_bounds__(GA_THREADS) void k_ga_place(const uint64_t *__restrict__ key, const uint32_t *__restrict__ pos,
                                                        uint64_t n, uint32_t *__restrict__ SA, uint8_t *__restrict__ heads)
{
    const uint64_t i = (uint64_t)blockIdx.x * GA_THREADS + threadIdx.x;
    if (i == 0) {
        SA[0] = (uint32_t)n;
        heads[0] = 1;
    }
    if (i >= n) return;
    SA[1 + i] = pos[i];
    heads[1 + i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

// ---- texts over at most four distinct byte values (e.g. the reference's own 'A'..'D' test texts, tests/kiss.cpp) ----
// A 7-character key holds only 14 bits of information there, so every suffix would go into the doubling rounds
// (3.3 Gbytes/s); mapped to codes 0..3 in value order, the text is a DNA text and the induced-sorting path orders it
// (exact order through PREFIX_DOUBLING: the same suffix array, four times as fast).
// which byte values occur: out[0..7] = 256-bit set, out[8] = 1 as soon as a wave has seen more than four of them (every
// wave stops at its next step then: a text over a large alphabet costs a few microseconds)
__global__ __launch_bounds__(GA_THREADS) void k_ga_presence(const uint8_t *__restrict__ S, uint64_t n, uint32_t *__restrict__ out)
{
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // the same in every lane of the wave
    const uint64_t waves = (uint64_t)gridDim.x * (GA_THREADS / 64);
    const uint64_t wave = (uint64_t)blockIdx.x * (GA_THREADS / 64) + (threadIdx.x >> 6);
    volatile uint32_t *many = out + 8;
    for (uint64_t base = wave * 64; base < n; base += waves * 64) {
        if (*many) return;
        const uint64_t i = base + lane_id();
        const bool valid = i < n;
        const uint32_t b = valid ? S[i] : 0u;
        uint64_t todo = __ballot(valid);
        while (todo) {
            const uint32_t b0 = (uint32_t)__shfl((int)b, (int)__builtin_ctzll(todo), 64);
#pragma unroll
            for (int w = 0; w < 8; w++) m[w] |= (uint32_t)w == (b0 >> 5) ? 1u << (b0 & 31u) : 0u;
            todo &= ~__ballot(b == b0);
        }
        uint32_t distinct = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) distinct += (uint32_t)__popc(m[w]);
        if (distinct > 4u) {
            if (lane_id() == 0) *many = 1u;
            return;
        }
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int w = 0; w < 8; w++)
            if (m[w]) atomicOr(&out[w], m[w]);
    }
}

// code = rank of the byte among the (at most four) values that occur, v1 < v2 < v3 being the larger ones
__global__ __launch_bounds__(GA_THREADS) void k_ga_remap(const uint8_t *__restrict__ S, uint64_t n, uint32_t v1, uint32_t v2,
                                                        uint32_t v3, uint8_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * GA_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = S[i];
    out[i] = (uint8_t)((b >= v1 ? 1u : 0u) + (b >= v2 ? 1u : 0u) + (b >= v3 ? 1u : 0u));
}

} // namespace

extern "C" {

int kiss_hip_ctx_suffix_sort_u8_dev(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t *d_SA, void *stream)
{
    if (!ctx || !d_SA || (n && !d_S)) return KISS_HIP_E_INVALID;
    if (n > KISS_HIP_MAX_N || n > ctx->max_n) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    kiss_opts_refresh(ctx);
    ctx->hfar = ctx->hmerged = nullptr;
    ctx->h_depth = 0;
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    KTRY(kiss_workspace_ready(ctx));
    std::memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->stats.n = n;
    ctx->stats.k = 0xFFFFFFFFu;
    ctx->n = n;
    ctx->m = ctx->m_far = 0;
    if (n == 0) {
        KTRY(kiss_zero_u32(ctx, d_SA, 1));
        KCHECK(hipStreamSynchronize(ctx->stream));
        return KISS_HIP_OK;
    }
    if (n >= 4096 && !ctx->opts.no_small_alphabet) { // at most four distinct byte values: the DNA path
        uint32_t *d_set = ctx->d_small + 48; // 9 words
        KTRY(kiss_zero_u32(ctx, d_set, 9));
        const uint64_t blocks = div_up(n, 64ull * (GA_THREADS / 64) * 64); // ~64 steps per wave
        hipLaunchKernelGGL(k_ga_presence, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(GA_THREADS), 0, ctx->stream, d_S, n,
                           d_set);
        KCHECK(hipGetLastError());
        KTRY(kiss_readback(ctx, d_set, 9));
        uint32_t vals[4], nv = 0;
        bool small = ctx->h_pinned[8] == 0;
        for (uint32_t v = 0; v < 256 && small; v++)
            if ((ctx->h_pinned[v >> 5] >> (v & 31u)) & 1u) {
                if (nv == 4) small = false;
                else vals[nv++] = v;
            }
        if (small && nv >= 1) {
            if (!ctx->ga_codes || ctx->ga_codes_cap < n) {
                if (ctx->ga_codes) {
                    (void)hipFree(ctx->ga_codes);
                    ctx->ws_bytes -= ctx->ga_codes_cap;
                    ctx->ga_codes = nullptr;
                    ctx->ga_codes_cap = 0;
                }
                hipError_t e = hipMalloc((void **)&ctx->ga_codes, ctx->max_n);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    ctx->last_hip_error = (int)e;
                    return KISS_HIP_E_NOMEM;
                }
                ctx->ga_codes_cap = ctx->max_n;
                ctx->ws_bytes += ctx->max_n;
            }
            while (nv < 4) { // absent ranks: never reached by a byte of the text
                vals[nv] = 256;
                nv++;
            }
            hipLaunchKernelGGL(k_ga_remap, dim3((unsigned)div_up(n, GA_THREADS)), dim3(GA_THREADS), 0, ctx->stream, d_S, n, vals[1],
                               vals[2], vals[3], ctx->ga_codes);
            KCHECK(hipGetLastError());
            return kiss_hip_ctx_suffix_sort_dna_u32_dev(ctx, ctx->ga_codes, n, 0xFFFFFFFFu, KISS_HIP_ALGO_PREFIX_DOUBLING, d_SA,
                                                        (void *)ctx->stream);
        }
    }
    // every position is an item here: the per-LMS arrays have to hold n of them
    if (n + 2 > ctx->m_cap) KTRY(kiss_lms_reserve(ctx, n + n / 64 + 1024));
    hipEvent_t ev[2];
    for (auto &e : ev) KCHECK(hipEventCreate(&e));
    (void)hipEventRecord(ev[0], ctx->stream);
    uint8_t *heads = nullptr;
    int rc = KISS_HIP_OK;
    do {
        if (hipMalloc((void **)&heads, n + 1) != hipSuccess) {
            rc = KISS_HIP_E_NOMEM;
            break;
        }
        const unsigned grid = (unsigned)div_up(n, GA_THREADS);
        {
            KTimer t(ctx, KISS_HIP_K_KEYGATHER, n);
            hipLaunchKernelGGL(k_ga_keys, dim3(grid), dim3(GA_THREADS), 0, ctx->stream, d_S, n, ctx->keyA, ctx->posA);
        }
        RadixBufs rb;
        rb.key[0] = ctx->keyA;
        rb.key[1] = ctx->keyB;
        rb.pos[0] = ctx->posA;
        rb.pos[1] = ctx->posB;
        rb.seg[0] = rb.seg[1] = nullptr;
        int res = 0;
        if ((rc = kiss_radix_sort(ctx, rb, n, 0, 0, &res))) break;
        {
            KTimer t(ctx, KISS_HIP_K_PLACE, n);
            hipLaunchKernelGGL(k_ga_place, dim3((unsigned)div_up(n + 1, GA_THREADS)), dim3(GA_THREADS), 0, ctx->stream,
                               rb.key[res], rb.pos[res], n, d_SA, heads);
        }
        if (hipGetLastError() != hipSuccess) {
            rc = KISS_HIP_E_HIP;
            break;
        }
        ctx->stats.refine_depth = GA_CHARS;
        if (n >= GA_CHARS) {
            if ((rc = kiss_exact_refine(ctx, n, GA_CHARS, d_SA, heads))) break;
        }
        (void)hipEventRecord(ev[1], ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = KISS_HIP_E_HIP;
            break;
        }
        (void)hipEventElapsedTime(&ctx->stats.ms_total, ev[0], ev[1]);
        ctx->stats.ms_refine = ctx->stats.ms_total;
        rc = kiss_radix_check(ctx);
    } while (0);
    if (rc != KISS_HIP_OK) (void)hipStreamSynchronize(ctx->stream);
    if (heads) (void)hipFree(heads);
    for (auto &e : ev) (void)hipEventDestroy(e);
    ktimer_collect(ctx);
    return rc;
}

int kiss_hip_suffix_sort_u8(const uint8_t *S, uint64_t n, uint32_t *SA, int device)
{
    if (!SA || (n && !S)) return KISS_HIP_E_INVALID;
    if (n == 0) {
        SA[0] = 0;
        return KISS_HIP_OK;
    }
    kiss_hip_ctx *ctx = nullptr;
    int rc = kiss_hip_ctx_create(&ctx, device, n);
    if (rc) return rc;
    uint8_t *d_S = nullptr;
    uint32_t *d_SA = nullptr;
    do {
        if (hipMalloc((void **)&d_S, n) != hipSuccess || hipMalloc((void **)&d_SA, (n + 1) * sizeof(uint32_t)) != hipSuccess) {
            rc = KISS_HIP_E_NOMEM;
            break;
        }
        if (hipMemcpy(d_S, S, n, hipMemcpyHostToDevice) != hipSuccess) {
            rc = KISS_HIP_E_HIP;
            break;
        }
        if ((rc = kiss_hip_ctx_suffix_sort_u8_dev(ctx, d_S, n, d_SA, nullptr))) break;
        if (hipMemcpy(SA, d_SA, (n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = KISS_HIP_E_HIP;
    } while (0);
    if (d_S) (void)hipFree(d_S);
    if (d_SA) (void)hipFree(d_SA);
    kiss_hip_ctx_destroy(ctx);
    return rc;
}

} // extern "C"
