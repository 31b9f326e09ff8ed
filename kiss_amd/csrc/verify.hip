// verify.hip -- device-side check of a suffix array against the text it was built from.
//
// Independent of the sorter: reads the caller's byte text and SA only (no packed text, no LMS machinery), so that a
// fault shared by the sort kernels cannot hide here.  What is checked:
//   1. SA[0] == n (the sentinel, kiss1_core.hpp:237-238 / kiss_common.hpp:578), every entry <= n, no value twice
//      (bitmap with atomicOr)  ->  SA is a permutation of [0, n];
//   2. bounded k: the reference's own test property (tests/kiss.cpp:26-28)
//          S.substr(SA[i-1], k) <= S.substr(SA[i], k)   for all i >= 1
//      evaluated for every adjacent pair, 8 bytes per step;
//   3. k >= n: the linear-time exactness proof -- with rank = inverse SA, for all i >= 1
//          S[a] < S[b]  or  (S[a] == S[b] and rank[a+1] < rank[b+1])        a = SA[i-1], b = SA[i], rank[n] = 0
//      which together with 1. holds iff SA is THE suffix array (each suffix is compared with its predecessor by first
//      character and the already-verified order of the remaining suffixes);
//   4. an order-sensitive 64-bit digest of SA (sum of a 64-bit mix of (i, SA[i])), computable in parallel on any
//      device or host -- tools and tests compare it across paths.
// Bytes are compared as unsigned values, so the same code serves the DNA codes 0..3 and the general byte alphabet
// (kiss_hip_suffix_sort_u8).
#include "kiss_internal.hpp"
#include <cstring>

namespace {

// counters (u64): [0] out of range, [1] duplicates, [2] order violations, [3] first violating index (min),
// [4] adjacent pairs equal through k bases (bounded k only), [5] digest, [6] SA[0] != n
constexpr int VC = 8;

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_vfy_perm(const uint32_t *__restrict__ SA, uint64_t total /* n + 1 */,
                                                  uint32_t *__restrict__ bitmap, unsigned long long *__restrict__ c)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t bad_range = 0, dup = 0, dig = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint32_t v = SA[i];
        dig += mix64((i << 32) ^ (uint64_t)v ^ 0x5851F42D4C957F2Dull * i);
        if ((uint64_t)v >= total) {
            bad_range++;
            continue;
        }
        const uint32_t bit = 1u << (v & 31u);
        if (atomicOr(&bitmap[v >> 5], bit) & bit) dup++;
        if (i == 0 && (uint64_t)v != total - 1) atomicAdd(&c[6], 1ull);
    }
    bad_range = wave_sum64(bad_range);
    dup = wave_sum64(dup);
    dig = wave_sum64(dig);
    if (lane_id() == 0) {
        if (bad_range) atomicAdd(&c[0], (unsigned long long)bad_range);
        if (dup) atomicAdd(&c[1], (unsigned long long)dup);
        atomicAdd(&c[5], (unsigned long long)dig);
    }
}

// 8 text bytes starting at byte p as a big-endian integer (first byte most significant); the caller guarantees
// p + 8 <= n.  Two ALIGNED 8-byte loads + funnel shift.  Both words hold at least one byte of [S + p, S + p + 8), which
// lies inside the text, and an aligned 8-byte word never straddles a page: neither load can fault, and the bytes
// outside [p, p + 8) are shifted out.
__device__ __forceinline__ uint64_t load8_be(const uint8_t *__restrict__ S, uint64_t p)
{
    const uintptr_t addr = reinterpret_cast<uintptr_t>(S) + p;
    const uint32_t sh = (uint32_t)(addr & 7u) * 8u;
    const uint64_t *w = reinterpret_cast<const uint64_t *>(addr & ~(uintptr_t)7);
    uint64_t v = w[0];
    if (sh) v = (v >> sh) | (w[1] << (64u - sh));
    return __builtin_bswap64(v);
}

__device__ __forceinline__ void report_violation(unsigned long long *c, uint64_t i)
{
    atomicAdd(&c[2], 1ull);
    atomicMin(&c[3], (unsigned long long)i);
}

// tests/kiss.cpp:26-28 for every adjacent pair
__global__ __launch_bounds__(256) void k_vfy_order_k(const uint8_t *__restrict__ S, uint64_t n, uint64_t k,
                                                     const uint32_t *__restrict__ SA, unsigned long long *__restrict__ c)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint64_t tied = 0;
    if (i <= n) {
        const uint64_t a = SA[i - 1], b = SA[i];
        if (a <= n && b <= n) { // range errors are counted by k_vfy_perm
            const uint64_t la = (n - a < k) ? n - a : k, lb = (n - b < k) ? n - b : k; // lengths of the two substrings
            const uint64_t common = la < lb ? la : lb;
            uint64_t off = 0;
            int res = 0; // <0: a smaller, >0: b smaller
            while (off + 8 <= common) {
                const uint64_t x = load8_be(S, a + off), y = load8_be(S, b + off);
                if (x != y) {
                    res = x < y ? -1 : 1;
                    break;
                }
                off += 8;
            }
            if (!res)
                for (; off < common; off++) {
                    const uint8_t x = S[a + off], y = S[b + off];
                    if (x != y) {
                        res = x < y ? -1 : 1;
                        break;
                    }
                }
            if (res > 0) report_violation(c, i);
            else if (res == 0) {
                if (la > lb) report_violation(c, i); // the shorter substring is a proper prefix: it must come first
                else if (la == k && lb == k) tied = 1;
            }
        }
    }
    tied = wave_sum64(tied);
    if (lane_id() == 0 && tied) atomicAdd(&c[4], (unsigned long long)tied);
}

__global__ __launch_bounds__(256) void k_vfy_isa(const uint32_t *__restrict__ SA, uint64_t total, uint32_t *__restrict__ isa)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const uint32_t v = SA[i];
        if ((uint64_t)v < total) isa[v] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_vfy_order_exact(const uint8_t *__restrict__ S, uint64_t n,
                                                         const uint32_t *__restrict__ SA, const uint32_t *__restrict__ isa,
                                                         unsigned long long *__restrict__ c)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (i > n) return;
    const uint64_t a = SA[i - 1], b = SA[i];
    if (a > n || b > n) return;
    if (b == n) { // the empty suffix can only stand at index 0
        report_violation(c, i);
        return;
    }
    if (a == n) return; // empty suffix < everything
    const uint8_t x = S[a], y = S[b];
    if (x < y) return;
    if (x > y || !(isa[a + 1] < isa[b + 1])) report_violation(c, i);
}

} // namespace

extern "C" int kiss_hip_ctx_verify_sa_dev(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t k,
                                          const uint32_t *d_SA, kiss_hip_verify_report *out, void *stream)
{
    if (!ctx || !out || !d_SA || (n && !d_S) || n > KISS_HIP_MAX_N) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->own_stream;
    std::memset(out, 0, sizeof *out);
    out->n = n;
    out->k = k;
    const uint64_t total = n + 1;
    const bool exact = (uint64_t)k >= n;
    out->exact = exact ? 1u : 0u;
    const uint64_t bm_words = div_up(total, 32);
    uint32_t *bitmap = nullptr, *isa = nullptr;
    unsigned long long *c = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = KISS_HIP_OK;
    do {
        const uint64_t c_off_words = (bm_words + 1) & ~1ull; // counters 8-byte aligned behind the bitmap
        hipError_t e = hipMalloc((void **)&bitmap, c_off_words * 4 + VC * 8);
        if (e == hipSuccess && exact && n) e = hipMalloc((void **)&isa, total * 4);
        if (e != hipSuccess) {
            ctx->last_hip_error = (int)e;
            rc = KISS_HIP_E_NOMEM;
            break;
        }
        c = reinterpret_cast<unsigned long long *>(bitmap + c_off_words);
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
            rc = KISS_HIP_E_HIP;
            break;
        }
        (void)hipEventRecord(e0, st);
        ctx->stream = st;
        if ((rc = kiss_zero_u32(ctx, bitmap, c_off_words + VC * 2))) break; // (a kernel, not hipMemsetAsync: see kiss_fill_u32)
        const unsigned long long big = ~0ull;
        if (hipMemcpyAsync(&c[3], &big, 8, hipMemcpyHostToDevice, st) != hipSuccess) {
            rc = KISS_HIP_E_HIP;
            break;
        }
        const unsigned pgrid = (unsigned)(div_up(total, 256) < 256 * 64 ? div_up(total, 256) : 256 * 64);
        hipLaunchKernelGGL(k_vfy_perm, dim3(pgrid), dim3(256), 0, st, d_SA, total, bitmap, c);
        if (n) {
            const unsigned grid = (unsigned)div_up(n, 256);
            if (exact) {
                hipLaunchKernelGGL(k_vfy_isa, dim3((unsigned)div_up(total, 256)), dim3(256), 0, st, d_SA, total, isa);
                hipLaunchKernelGGL(k_vfy_order_exact, dim3(grid), dim3(256), 0, st, d_S, n, d_SA, isa, c);
            } else {
                hipLaunchKernelGGL(k_vfy_order_k, dim3(grid), dim3(256), 0, st, d_S, n, (uint64_t)k, d_SA, c);
            }
        }
        (void)hipEventRecord(e1, st);
        unsigned long long h[VC];
        hipError_t he = hipMemcpyAsync(h, c, sizeof h, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        if (he == hipSuccess) he = hipGetLastError();
        if (he != hipSuccess) {
            ctx->last_hip_error = (int)he;
            rc = KISS_HIP_E_HIP;
            break;
        }
        out->out_of_range = h[0];
        out->duplicates = h[1];
        out->order_violations = h[2];
        out->first_violation = h[2] ? h[3] : 0;
        out->tied_pairs = h[4];
        out->digest = h[5];
        out->sa0_ok = h[6] ? 0u : 1u;
        out->ok = (out->sa0_ok && !h[0] && !h[1] && !h[2]) ? 1u : 0u;
        (void)hipEventElapsedTime(&out->ms, e0, e1);
    } while (0);
    if (bitmap) (void)hipFree(bitmap);
    if (isa) (void)hipFree(isa);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

// The same digest on the host (tools / tests that hold an SA in host memory, e.g. the oracle's): sum over i of
// mix64 of (i, SA[i]) exactly as k_vfy_perm forms it.
extern "C" uint64_t kiss_hip_sa_digest_host(const uint32_t *SA, uint64_t count)
{
    uint64_t dig = 0;
    for (uint64_t i = 0; i < count; i++) {
        uint64_t x = (i << 32) ^ (uint64_t)SA[i] ^ 0x5851F42D4C957F2Dull * i;
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        dig += x ^ (x >> 31);
    }
    return dig;
}

// FNV-1a-64 over the bytes of a host buffer (the hash the round-1 full-size parity logs and tests/golden use);
// `seed` = 0xcbf29ce484222325 for a fresh hash, or the previous return value to continue over the next chunk.
extern "C" uint64_t kiss_hip_fnv1a64_host(const void *data, uint64_t bytes, uint64_t seed)
{
    const uint8_t *p = static_cast<const uint8_t *>(data);
    uint64_t h = seed;
    for (uint64_t i = 0; i < bytes; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
