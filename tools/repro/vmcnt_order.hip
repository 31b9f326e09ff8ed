// Minimal HIP probe (no kiss code) for DESIGN.md 4.2, round 4: is the data of the FIRST of two vector loads in its register once
// `s_waitcnt vmcnt(1)` lets the wave go on?  (The ISA says vector loads return in order, and hipcc relies on it: it waits with
// vmcnt(1) for the older of two outstanding loads.)  Every lane: A = table[random index] (a 256 MiB table: TLB and cache misses),
// B = hot[lane] (always cached); after vmcnt(1) the register of A is copied; after vmcnt(0) the copy is compared with what A
// finally delivered.  A second stream thrashes caches and TLBs, as a second sort on the same GPU does.
// build: hipcc -O2 --offload-arch=gfx950 vmcnt_order.hip -o vmcnt_order -lpthread ; run: ./vmcnt_order [iterations] [noise threads]
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_fill(uint32_t *t, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = i * 2654435761u + 12345u;
}
// the same question for ONE 16-byte load at an address that is only 8-byte aligned and crosses a 64-byte line (what
// kiss_key32 issues for one packed-text position in eight): are all four destination registers there after s_waitcnt vmcnt(0)?
// counts[2]: probes whose upper half was not; counts[3]: whose lower half was not
__global__ void k_probe_x4(const uint32_t *__restrict__ table, uint32_t mask, uint32_t iters, uint32_t byte_in_line,
                           unsigned long long *__restrict__ counts)
{
    uint32_t seed = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    unsigned long long up = 0, low = 0;
    for (uint32_t it = 0; it < iters; it++) {
        seed = seed * 1664525u + 1013904223u;
        const uint32_t ia = ((seed >> 4) & (mask >> 1) & ~15u) + byte_in_line / 4u; // dword index: line start + offset, well inside the table
        const uint32_t *pa = table + ia;
        uint32_t c0, c1, c2, c3;
        asm volatile("v_mov_b32 v40, -1\n\t"
                     "v_mov_b32 v41, -1\n\t"
                     "v_mov_b32 v42, -1\n\t"
                     "v_mov_b32 v43, -1\n\t"
                     "s_nop 4\n\t"
                     "global_load_dwordx4 v[40:43], %[pa], off\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "v_mov_b32 %[c0], v40\n\t"
                     "v_mov_b32 %[c1], v41\n\t"
                     "v_mov_b32 %[c2], v42\n\t"
                     "v_mov_b32 %[c3], v43\n\t"
                     : [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [c3] "=&v"(c3)
                     : [pa] "v"(pa)
                     : "memory", "v40", "v41", "v42", "v43");
        const uint32_t w0 = ia * 2654435761u + 12345u, w2 = (ia + 2) * 2654435761u + 12345u;
        if (c0 != w0) low++;
        if (c2 != w2) up++;
        seed += c1 + c3;
    }
    if (up) atomicAdd(&counts[2], up);
    if (low) atomicAdd(&counts[3], low);
}

// TWO 16-byte loads in flight at once, each at an 8-byte aligned address `byte_in_line` bytes into a 64-byte line (56: both
// cross their line), ALL waited for with s_waitcnt vmcnt(0): do the eight registers hold the eight words?
// counts[4 + q]: probes whose dword q (0-3: first load, 4-7: second) is not what the table holds there
__global__ void k_probe_2x4(const uint32_t *__restrict__ table, uint32_t mask, uint32_t iters, uint32_t byte_in_line,
                            unsigned long long *__restrict__ counts)
{
    uint32_t seed = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    unsigned long long bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t it = 0; it < iters; it++) {
        seed = seed * 1664525u + 1013904223u;
        const uint32_t ia = ((seed >> 4) & (mask >> 1) & ~15u) + byte_in_line / 4u;
        seed = seed * 1664525u + 1013904223u;
        // the second load near the first one half of the time (same pages), far away otherwise
        const uint32_t ib = (seed & 1u) ? (((seed >> 4) & (mask >> 1) & ~15u) + byte_in_line / 4u) : ((ia + 16u * (1u + ((seed >> 8) & 63u))) & (mask >> 1) & ~15u) + byte_in_line / 4u;
        const uint32_t *pa = table + ia, *pb = table + ib;
        uint32_t c[8];
        asm volatile("global_load_dwordx4 v[40:43], %[pa], off\n\t"
                     "global_load_dwordx4 v[44:47], %[pb], off\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "v_mov_b32 %[c0], v40\n\t"
                     "v_mov_b32 %[c1], v41\n\t"
                     "v_mov_b32 %[c2], v42\n\t"
                     "v_mov_b32 %[c3], v43\n\t"
                     "v_mov_b32 %[c4], v44\n\t"
                     "v_mov_b32 %[c5], v45\n\t"
                     "v_mov_b32 %[c6], v46\n\t"
                     "v_mov_b32 %[c7], v47\n\t"
                     : [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]), [c5] "=&v"(c[5]),
                       [c6] "=&v"(c[6]), [c7] "=&v"(c[7])
                     : [pa] "v"(pa), [pb] "v"(pb)
                     : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t idx = (q < 4 ? ia : ib) + (uint32_t)(q & 3);
            if (c[q] != idx * 2654435761u + 12345u) bad[q]++;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
        if (bad[q]) atomicAdd(&counts[4 + q], bad[q]);
}

// the shape hipcc gives the two loads of a key compare in k_near_tie_runs: the destination of each 16-byte load starts in its own
// address registers, the first is waited for with vmcnt(1) and used, then the second with vmcnt(0)
__global__ void k_probe_2x4_overlap(const uint32_t *__restrict__ table, uint32_t mask, uint32_t iters, uint32_t byte_in_line,
                                    unsigned long long *__restrict__ counts)
{
    uint32_t seed = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    unsigned long long bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t it = 0; it < iters; it++) {
        seed = seed * 1664525u + 1013904223u;
        const uint32_t ia = ((seed >> 4) & (mask >> 1) & ~15u) + byte_in_line / 4u;
        seed = seed * 1664525u + 1013904223u;
        const uint32_t ib = (seed & 1u) ? (((seed >> 4) & (mask >> 1) & ~15u) + byte_in_line / 4u) : ((ia + 16u * (1u + ((seed >> 8) & 63u))) & (mask >> 1) & ~15u) + byte_in_line / 4u;
        const uint64_t pa = (uint64_t)(uintptr_t)(table + ia), pb = (uint64_t)(uintptr_t)(table + ib);
        uint32_t c[8];
        asm volatile("v_mov_b32 v40, %[pal]\n\t"
                     "v_mov_b32 v41, %[pah]\n\t"
                     "v_mov_b32 v44, %[pbl]\n\t"
                     "v_mov_b32 v45, %[pbh]\n\t"
                     "s_nop 4\n\t"
                     "global_load_dwordx4 v[40:43], v[40:41], off\n\t"
                     "global_load_dwordx4 v[44:47], v[44:45], off\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %[c0], v40\n\t"
                     "v_mov_b32 %[c1], v41\n\t"
                     "v_mov_b32 %[c2], v42\n\t"
                     "v_mov_b32 %[c3], v43\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "v_mov_b32 %[c4], v44\n\t"
                     "v_mov_b32 %[c5], v45\n\t"
                     "v_mov_b32 %[c6], v46\n\t"
                     "v_mov_b32 %[c7], v47\n\t"
                     : [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]), [c5] "=&v"(c[5]),
                       [c6] "=&v"(c[6]), [c7] "=&v"(c[7])
                     : [pal] "v"((uint32_t)pa), [pah] "v"((uint32_t)(pa >> 32)), [pbl] "v"((uint32_t)pb), [pbh] "v"((uint32_t)(pb >> 32))
                     : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t idx = (q < 4 ? ia : ib) + (uint32_t)(q & 3);
            if (c[q] != idx * 2654435761u + 12345u) bad[q]++;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
        if (bad[q]) atomicAdd(&counts[4 + q], bad[q]);
}

__global__ void k_probe(const uint32_t *__restrict__ table, uint32_t mask, const uint32_t *__restrict__ hot, uint32_t iters,
                        unsigned long long *__restrict__ counts)
{
    uint32_t seed = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    unsigned long long early = 0, wrong = 0;
    for (uint32_t it = 0; it < iters; it++) {
        seed = seed * 1664525u + 1013904223u;
        const uint32_t ia = (seed >> 4) & mask;
        const uint32_t *pa = table + ia, *pb = hot + (threadIdx.x & 63u);
        uint32_t a, b, c;
        asm volatile("v_mov_b32 %[a], -1\n\t"
                     "s_nop 4\n\t"
                     "global_load_dword %[a], %[pa], off\n\t"
                     "global_load_dword %[b], %[pb], off\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %[c], %[a]\n\t" // the older load has returned -- says vmcnt(1)
                     "s_waitcnt vmcnt(0)\n\t"
                     : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c)
                     : [pa] "v"(pa), [pb] "v"(pb)
                     : "memory");
        const uint32_t want = ia * 2654435761u + 12345u;
        if (c != a) early++;   // the copy was taken before A's data arrived
        if (a != want) wrong++; // (never: the load itself is right)
        seed += b;
    }
    if (early) atomicAdd(&counts[0], early);
    if (wrong) atomicAdd(&counts[1], wrong);
}
__global__ void k_noise(uint32_t *buf, uint32_t mask, uint32_t iters)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, a = 0;
    for (uint32_t it = 0; it < iters; it++) a += buf[(i * 2654435761u + it * 40503u) & mask];
    if (a == 0x12345678u) buf[0] = a;
}
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
        for (int q = 0; q < 64; q++) {
            r = r * 1664525u + 1013904223u;
            hipLaunchKernelGGL(k_noise, dim3(1u << ((r >> 24) % 13)), dim3(256), 0, st, buf, (16u << 20) - 1, 4u + ((r >> 8) & 63u));
        }
        (void)hipStreamSynchronize(st);
    }
}
int main(int argc, char **argv)
{
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 20000;
    const int noisy = argc > 2 ? atoi(argv[2]) : 1;
    const uint32_t n = 1u << 26; // 256 MiB
    uint32_t *table, *hot;
    unsigned long long *cnt;
    (void)hipMalloc(&table, (size_t)n * 4);
    (void)hipMalloc(&hot, 256);
    (void)hipMalloc(&cnt, 128);
    (void)hipMemset(hot, 0, 256);
    (void)hipMemset(cnt, 0, 128);
    hipLaunchKernelGGL(k_fill, dim3(n / 256), dim3(256), 0, 0, table, n);
    (void)hipDeviceSynchronize();
    std::vector<std::thread> th;
    for (int t = 0; t < noisy; t++) th.emplace_back(noise_thread, t);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    unsigned long long probes = 0;
    for (int round = 0; round < 200; round++) { // single-wave launches, like the placement kernels, and a few larger ones
        const unsigned grid = round % 10 == 9 ? 512 : 1;
        hipLaunchKernelGGL(k_probe, dim3(grid), dim3(64), 0, st, table, n - 1, hot, iters, cnt);
        probes += (unsigned long long)grid * 64 * iters;
    }
    for (uint32_t off : {56u, 24u, 48u}) {
        (void)hipMemset(cnt + 2, 0, 16);
        for (int round = 0; round < 100; round++)
            hipLaunchKernelGGL(k_probe_x4, dim3(round % 10 == 9 ? 256 : 1), dim3(64), 0, st, table, n - 1, iters, off, cnt);
        (void)hipStreamSynchronize(st);
        unsigned long long x[2];
        (void)hipMemcpy(x, cnt + 2, 16, hipMemcpyDeviceToHost);
        printf("vmcnt_order: 16-byte load at byte %u of a 64-byte line: after s_waitcnt vmcnt(0) the upper half was not there in %llu probes, "
               "the lower half in %llu\n", off, x[0], x[1]);
    }
    for (uint32_t off : {56u, 24u, 48u}) {
        (void)hipMemset(cnt + 4, 0, 64);
        for (int round = 0; round < 100; round++)
            hipLaunchKernelGGL(k_probe_2x4, dim3(round % 10 == 9 ? 256 : 1), dim3(64), 0, st, table, n - 1, iters, off, cnt);
        (void)hipStreamSynchronize(st);
        unsigned long long x[8];
        (void)hipMemcpy(x, cnt + 4, 64, hipMemcpyDeviceToHost);
        printf("vmcnt_order: two 16-byte loads in flight, each at byte %u of its 64-byte line, after s_waitcnt vmcnt(0): wrong dwords per "
               "position %llu %llu %llu %llu | %llu %llu %llu %llu\n", off, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
    }
    for (uint32_t off : {56u, 24u, 48u}) {
        (void)hipMemset(cnt + 4, 0, 64);
        for (int round = 0; round < 100; round++)
            hipLaunchKernelGGL(k_probe_2x4_overlap, dim3(round % 10 == 9 ? 256 : 1), dim3(64), 0, st, table, n - 1, iters, off, cnt);
        (void)hipStreamSynchronize(st);
        unsigned long long x[8];
        (void)hipMemcpy(x, cnt + 4, 64, hipMemcpyDeviceToHost);
        printf("vmcnt_order: two 16-byte loads whose destinations start in their own address registers (vmcnt(1), then vmcnt(0)), each at byte %u "
               "of its line: wrong dwords per position %llu %llu %llu %llu | %llu %llu %llu %llu\n", off, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
    }
    (void)hipStreamSynchronize(st);
    stop.store(true);
    for (auto &x : th) x.join();
    unsigned long long h[2];
    (void)hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
    printf("vmcnt_order: %llu of %llu probes read the older load's register before its data was there after s_waitcnt vmcnt(1); "
           "%llu loads delivered a wrong value (noise threads: %d)\n", h[0], probes, h[1], noisy);
    return h[0] ? 1 : 0;
}
