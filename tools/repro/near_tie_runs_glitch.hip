// Stand-alone reproducer for DESIGN.md 4.2 (round 4): the device code of place.hip's k_near_tie_runs (copied verbatim with
// the helpers it inlines; hipcc 7.2 compiles it to the same instruction sequence as inside the library) run over and over on
// ONE stream over a small synthetic instance whose answer is known in closed form, while a second host thread keeps a second
// stream of the same GPU busy with short kernels -- no library, no sort, two streams, two kernels.
//   text      : (TTAGGG)^n, n = 400 000                      far list : M positions; the first B in another phase of the period
//   near-end  : E = 106 suffixes in the last 624 bases         (they share nothing with a near-end suffix), the rest in phase:
//               (19 of them have k = 512 bases left)            every one of those shares k bases -> run = [B, M) for the 19
// expected: run_start[e] = B for e < 19, M for the others.  The fault: a launch that leaves run_start[e] = M for ALL lanes.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 near_tie_runs_glitch.hip -o near_tie_runs_glitch -lpthread
// run:   ./near_tie_runs_glitch [launches = 200000] [noise threads = 1] [kernel mode: 0 as shipped, 1 two 8-byte loads (merged by the compiler), 2 two 8-byte loads kept apart, 3 aligned 16-byte load + 8-byte load] [noise kind: 1 memory, 2 ALU]
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

constexpr int PL_THREADS = 256;

// MODE 0: as in the library (one 16-byte load at an 8-byte aligned address); MODE 1: two 8-byte loads
template <int MODE>
__device__ __forceinline__ uint64_t kiss_key32(const uint64_t *__restrict__ pk, uint64_t p)
{
    uint64_t w = p >> 5;
    uint32_t s = (uint32_t)(p & 31u) * 2u;
    uint64_t a, b;
    if (MODE == 1) {
        a = pk[w];
        b = pk[w + 1];
    } else if (MODE == 2) { // two naturally aligned 8-byte loads the compiler cannot merge into one 16-byte load
        a = pk[w];
        asm volatile("" ::: "memory");
        b = pk[w + 1];
    } else if (MODE == 3) { // one 16-byte load at a 16-byte aligned address + the word behind it
        const ulonglong2 ab = *reinterpret_cast<const ulonglong2 *>(__builtin_assume_aligned(pk + (w & ~1ull), 16));
        const uint64_t c = pk[(w | 1ull) + 1];
        a = (w & 1ull) ? ab.y : ab.x;
        b = (w & 1ull) ? c : ab.y;
    } else {
        const ulonglong2 ab = *reinterpret_cast<const ulonglong2 *>(pk + w);
        a = ab.x;
        b = ab.y;
    }
    return (a << s) | ((b >> 1) >> (63u - s));
}
template <int MODE>
__device__ int cmp_bases(const uint64_t *__restrict__ pk, uint64_t i, uint64_t j, uint64_t len)
{
    while (len >= 32) {
        uint64_t a = kiss_key32<MODE>(pk, i), b = kiss_key32<MODE>(pk, j);
        if (a != b) return a < b ? -1 : 1;
        i += 32;
        j += 32;
        len -= 32;
    }
    if (len) {
        uint64_t mask = ~0ull << (64 - 2 * len);
        uint64_t a = kiss_key32<MODE>(pk, i) & mask, b = kiss_key32<MODE>(pk, j) & mask;
        if (a != b) return a < b ? -1 : 1;
    }
    return 0;
}
template <int MODE>
__device__ __forceinline__ bool shares_k(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k, uint64_t f, uint64_t e)
{
    if (f + k > n || e + k > n) return false;
    return cmp_bases<MODE>(pk, f, e, k) == 0;
}
template <int MODE>
__global__ __launch_bounds__(PL_THREADS) void k_near_tie_runs(const uint64_t *__restrict__ pk, uint64_t n, uint64_t k,
                                                             const uint32_t *__restrict__ far_sorted,
                                                             const uint32_t *__restrict__ near_pos,
                                                             const uint32_t *__restrict__ near_idx, uint32_t E,
                                                             uint32_t *__restrict__ run_start)
{
    const uint32_t e = blockIdx.x * PL_THREADS + threadIdx.x;
    if (e >= E) return;
    const uint64_t pe = near_pos[e];
    const uint64_t hi = near_idx[e];
    uint64_t lo = hi;
    uint64_t step = 1;
    uint64_t bad = hi;
    while (lo > 0) {
        const uint64_t j = lo >= step ? lo - step : 0;
        if (shares_k<MODE>(pk, n, k, far_sorted[j], pe)) {
            lo = j;
            step *= 2;
            if (j == 0) break;
        } else {
            bad = j + 1;
            uint64_t a = bad, b = lo;
            while (a < b) {
                const uint64_t mid = (a + b) >> 1;
                if (shares_k<MODE>(pk, n, k, far_sorted[mid], pe)) b = mid;
                else a = mid + 1;
            }
            lo = a;
            break;
        }
    }
    (void)bad;
    run_start[e] = (uint32_t)lo;
}

__global__ void k_check(const uint32_t *run_start, uint32_t E, uint32_t can_tie, uint32_t B, uint32_t M, unsigned long long *counts)
{
    // counts[0] launches with any wrong entry, [1] launches where EVERY suffix that can tie got an empty run (the fault's shape)
    uint32_t wrong = 0, empty = 0;
    for (uint32_t e = 0; e < E; e++) {
        const uint32_t want = e < can_tie ? B : M;
        wrong += run_start[e] != want;
        empty += e < can_tie && run_start[e] == M;
    }
    if (wrong) atomicAdd(&counts[0], 1ull);
    if (empty == can_tie) atomicAdd(&counts[1], 1ull);
}
__global__ void k_poison(uint32_t *run_start, uint32_t E)
{
    if (threadIdx.x < E) run_start[threadIdx.x] = 0xDEADBEEFu;
}
__global__ void k_noise(uint32_t *buf, uint32_t mask, uint32_t iters)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, a = 0;
    for (uint32_t it = 0; it < iters; it++) a += buf[(i * 2654435761u + it * 40503u) & mask];
    if (a == 0x12345678u) buf[0] = a;
}

__global__ void k_noise_alu(uint32_t *buf, uint32_t iters) // no memory traffic at all
{
    uint32_t a = threadIdx.x + blockIdx.x;
    for (uint32_t it = 0; it < iters * 64; it++) a = a * 1664525u + 1013904223u;
    if (a == 0x12345678u) buf[0] = a;
}
static int noise_kind = 1; // 1: memory reads, 2: ALU only
static std::atomic<bool> stop{false};
static void noise_thread(int t)
{
    hipStream_t st;
    (void)hipStreamCreate(&st);
    uint32_t *buf;
    (void)hipMalloc(&buf, 64u << 20);
    (void)hipMemset(buf, 1, 64u << 20);
    uint32_t r = 12345u + t;
    while (!stop.load()) {
        for (int q = 0; q < 64; q++) { // short kernels of very different sizes, like the launches of a sort
            r = r * 1664525u + 1013904223u;
            const unsigned grid = 1u << ((r >> 24) % 13);
            if (noise_kind == 2) hipLaunchKernelGGL(k_noise_alu, dim3(grid), dim3(256), 0, st, buf, 4u + ((r >> 8) & 63u));
            else hipLaunchKernelGGL(k_noise, dim3(grid), dim3(256), 0, st, buf, (16u << 20) - 1, 4u + ((r >> 8) & 63u));
        }
        (void)hipStreamSynchronize(st);
    }
}

int main(int argc, char **argv)
{
    const long launches = argc > 1 ? atol(argv[1]) : 200000;
    const int noisy = argc > 2 ? atoi(argv[2]) : 1;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    noise_kind = argc > 4 ? atoi(argv[4]) : 1;
    const uint64_t n = 400000, k = 512;
    const uint32_t E = 106, M = 52709, B = 42765, can_tie = 19;
    // text (TTAGGG)^n packed 2 bits per base, base i at bits 63 - 2 (i % 32) of word i / 32; A C G T = 0 1 2 3
    const uint8_t unit[6] = {3, 3, 0, 2, 2, 2};
    std::vector<uint64_t> pk(n / 32 + 8, 0);
    for (uint64_t i = 0; i < n; i++) pk[i >> 5] |= (uint64_t)unit[i % 6] << (62 - 2 * (i & 31));
    std::vector<uint32_t> far(M), npos(E), nidx(E, M);
    for (uint32_t j = 0; j < M; j++) far[j] = j < B ? 6 * j + 3 : 6 * (j - B) + 6; // phase 3: shares nothing; phase 0: shares all it has
    for (uint32_t e = 0; e < E; e++) npos[e] = (uint32_t)(n - 620 + 6 * e);       // 399380, 399386, ...: phase 2 -> see below
    // make the near-end suffixes in phase with the sharing far suffixes: shift them to a multiple of 6
    for (uint32_t e = 0; e < E; e++) npos[e] = (npos[e] / 6) * 6;
    uint32_t ct = 0;
    for (uint32_t e = 0; e < E; e++) ct += npos[e] + k <= n;
    if (ct != can_tie) {
        printf("instance: %u near-end suffixes can tie (expected %u)\n", ct, can_tie);
        return 2;
    }
    uint64_t *d_pk;
    uint32_t *d_far, *d_npos, *d_nidx, *d_run;
    unsigned long long *d_cnt;
    (void)hipMalloc(&d_pk, pk.size() * 8);
    (void)hipMalloc(&d_far, M * 4);
    (void)hipMalloc(&d_npos, E * 4);
    (void)hipMalloc(&d_nidx, E * 4);
    (void)hipMalloc(&d_run, 65536 * 4);
    (void)hipMalloc(&d_cnt, 16);
    (void)hipMemcpy(d_pk, pk.data(), pk.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_far, far.data(), M * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_npos, npos.data(), E * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_nidx, nidx.data(), E * 4, hipMemcpyHostToDevice);
    (void)hipMemset(d_cnt, 0, 16);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    std::vector<std::thread> th;
    for (int t = 0; t < noisy; t++) th.emplace_back(noise_thread, t);
    for (long it = 0; it < launches; it++) {
        hipLaunchKernelGGL(k_poison, dim3(1), dim3(256), 0, st, d_run, E);
        if (mode == 2) hipLaunchKernelGGL(k_near_tie_runs<2>, dim3(1), dim3(PL_THREADS), 0, st, d_pk, n, k, d_far, d_npos, d_nidx, E, d_run);
        else if (mode == 3) hipLaunchKernelGGL(k_near_tie_runs<3>, dim3(1), dim3(PL_THREADS), 0, st, d_pk, n, k, d_far, d_npos, d_nidx, E, d_run);
        else if (mode == 1) hipLaunchKernelGGL(k_near_tie_runs<1>, dim3(1), dim3(PL_THREADS), 0, st, d_pk, n, k, d_far, d_npos, d_nidx, E, d_run);
        else hipLaunchKernelGGL(k_near_tie_runs<0>, dim3(1), dim3(PL_THREADS), 0, st, d_pk, n, k, d_far, d_npos, d_nidx, E, d_run);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(1), 0, st, d_run, E, can_tie, B, M, d_cnt);
        if ((it & 255) == 255) (void)hipStreamSynchronize(st);
    }
    (void)hipStreamSynchronize(st);
    stop.store(true);
    for (auto &x : th) x.join();
    unsigned long long h[2];
    (void)hipMemcpy(h, d_cnt, 16, hipMemcpyDeviceToHost);
    printf("near_tie_runs_glitch: %llu of %ld launches wrong, %llu of them with an empty run for every suffix that can tie (noise threads: %d, kernel mode %d, noise kind %d)\n",
           h[0], launches, h[1], noisy, mode, noise_kind);
    return h[0] ? 1 : 0;
}
