// Minimal HIP reproducer (no kiss code) for the hypothesis of DESIGN.md 4.2:
// on gfx950, can an SALU instruction that reads VCC right behind the VALU compare that writes it see the OLD VCC
// when scalar loads of the same wave are returning / other waves share the SIMD?
//   v_cmp_lt_u64 vcc, s[n], v[x]      (the hoisted "pe + k <= n" test of k_near_tie_runs as hipcc 7.2 compiles it)
//   s_xor_b64    s[m], vcc, -1        (next instruction)
// Every wave repeats the pair with fresh operands, recomputes the mask the slow way and counts disagreements.
// build: hipcc -O2 --offload-arch=gfx950 vcc_salu_hazard.hip -o vcc_salu_hazard ; run: ./vcc_salu_hazard [iters] [mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <thread>
#include <vector>

__global__ void k_probe(const uint64_t *__restrict__ tab, uint32_t tab_mask, uint64_t n, uint32_t iters,
                        unsigned long long *__restrict__ bad, unsigned long long *__restrict__ stale_old)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t seed = (uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull + threadIdx.x;
    unsigned long long nbad = 0, nold = 0;
    for (uint32_t it = 0; it < iters; it++) {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        // x straddles n: about half of the lanes have n < x
        const uint64_t x = n - 512 + (seed >> 54);
        // a global load whose latency the pair waits for (as the kernel waits for near_pos[e]) and two scalar loads whose
        // return may fall on the pair (as run_start / pk pointer loads do)
        const uint64_t *p = tab + ((seed >> 20) & tab_mask);
        const uint64_t *q = tab + (((uint32_t)(seed >> 7) + 64u * it) & tab_mask & ~7u);
        uint64_t g, s1, s2, mask;
        const uint64_t qu = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)q & 0xFFFFFFFFu)) |
                            ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)q >> 32)) << 32);
        asm volatile("global_load_dwordx2 %[g], %[p], off\n\t"
                     "s_load_dwordx2 %[s1], %[q], 0x0\n\t"
                     "s_load_dwordx2 %[s2], %[q], 0x20\n\t"
                     "v_cmp_gt_u32_e32 vcc, 64, %[lane]\n\t" // old VCC: all ones
                     "s_waitcnt vmcnt(0)\n\t"
                     "v_cmp_lt_u64_e32 vcc, %[n], %[x]\n\t"
                     "s_xor_b64 %[m], vcc, -1\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     : [g] "=&v"(g), [s1] "=&s"(s1), [s2] "=&s"(s2), [m] "=&s"(mask)
                     : [p] "v"(p), [q] "s"(qu), [lane] "v"(lane), [n] "s"(n), [x] "v"(x)
                     : "vcc", "memory");
        // the same mask the slow way: the compare writes an SGPR pair of its own, and nothing reads it for 16 wait states
        uint64_t want;
        asm volatile("v_cmp_lt_u64_e64 %[w], %[n], %[x]\n\t"
                     "s_nop 7\n\t"
                     "s_nop 7\n\t"
                     "s_not_b64 %[w], %[w]\n\t"
                     : [w] "=&s"(want)
                     : [n] "s"(n), [x] "v"(x)
                     : "scc");
        if (lane == 0) {
            if (mask != want) {
                nbad++;
                if (mask == 0) nold++; // ~(all ones): the value the pair would give with the OLD vcc
            }
        }
        seed ^= g + s1 + s2; // keep the loads alive
    }
    if (lane == 0 && nbad) {
        atomicAdd(bad, nbad);
        atomicAdd(stale_old, nold);
    }
}

__global__ void k_noise(uint64_t *buf, uint32_t mask, uint32_t iters)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, a = 0;
    for (uint32_t it = 0; it < iters; it++) a += buf[(i * 2654435761u + it * 97u) & mask];
    if (a == 1) buf[0] = a;
}

int main(int argc, char **argv)
{
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 20000;
    const int noise = argc > 2 ? atoi(argv[2]) : 1;
    const uint32_t words = 1u << 24; // 128 MiB table
    uint64_t *tab = nullptr;
    unsigned long long *cnt = nullptr;
    if (hipMalloc(&tab, (size_t)words * 8) != hipSuccess || hipMalloc(&cnt, 16) != hipSuccess) return 2;
    (void)hipMemset(tab, 0x5A, (size_t)words * 8);
    (void)hipMemset(cnt, 0, 16);
    hipStream_t s0, s1;
    (void)hipStreamCreate(&s0);
    (void)hipStreamCreate(&s1);
    for (int round = 0; round < 8; round++) {
        if (noise) hipLaunchKernelGGL(k_noise, dim3(4096), dim3(256), 0, s1, tab, words - 1, 4000u);
        // few workgroups of one wave (the placement kernels are single-wave kernels) ... and a saturating grid
        hipLaunchKernelGGL(k_probe, dim3(round & 1 ? 8192 : 64), dim3(round & 2 ? 256 : 64), 0, s0, tab, words - 1, 0x100000000ull - 4096,
                           iters, cnt, cnt + 1);
        (void)hipDeviceSynchronize();
    }
    unsigned long long h[2] = {0, 0};
    (void)hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
    printf("vcc_salu_hazard: %llu wrong masks (%llu of them = the mask of the OLD vcc) in 8 rounds x %u iterations per wave\n", h[0], h[1], iters);
    return h[0] ? 1 : 0;
}
