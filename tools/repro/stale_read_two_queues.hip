// Minimal HIP reproducer attempt (no kiss code) for DESIGN.md 4.2: does a one-workgroup kernel ever read, from an array the
// kernels before it IN THE SAME STREAM wrote, something older than what they wrote -- while a second host thread drives a
// second stream of the same kind on the same GPU?  Each thread: loop { W: many workgroups write X[i] = epoch (scattered 4-byte
// stores, like the retirements of the LMS sort); P: a one-wave kernel publishes a word to coherent host memory and the host
// spins on it (the library's read-back); R: a one-wave kernel reads 64 entries of X, first a dependent chain as the binary
// searches do, and counts those that are not `epoch` }.
// build: hipcc -O2 --offload-arch=gfx950 stale_read_two_queues.hip -o stale_read_two_queues -lpthread ; run: ./stale_read_two_queues [seconds] [threads]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_write(uint32_t *X, uint32_t n, uint32_t epoch, uint32_t stride)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) X[(uint64_t)i * stride % n] = epoch; // a permutation of [0, n) when gcd(stride, n) = 1: scattered stores
}
__global__ void k_publish(const uint32_t *src, uint32_t *host_dst, uint32_t seq)
{
    if (threadIdx.x == 0) {
        __hip_atomic_store(&host_dst[0], src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&host_dst[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_read(const uint32_t *X, uint32_t n, uint32_t epoch, uint32_t seed, unsigned long long *bad, uint32_t *first_bad)
{
    uint32_t idx = (seed * 2654435761u + threadIdx.x * 40503u) % n;
    uint32_t wrong = 0, what = 0;
    for (int hop = 0; hop < 8; hop++) { // a dependent chain of loads, like a binary search over the sorted list
        const uint32_t v = X[idx];
        if (v != epoch) {
            wrong++;
            what = v;
        }
        idx = (idx * 1664525u + v + 1013904223u) % n;
    }
    if (wrong) {
        atomicAdd(bad, (unsigned long long)wrong);
        first_bad[0] = what;
        first_bad[1] = epoch;
    }
}

static std::atomic<bool> stop{false};

static void worker(int t, unsigned long long *total_bad, unsigned long long *epochs)
{
    hipStream_t st;
    (void)hipStreamCreate(&st);
    const uint32_t n = 106649 + 1000 * t; // the size of the sorted LMS list of the failing test text
    uint32_t *X, *fb, *h_pub, *d_pub;
    unsigned long long *bad;
    (void)hipMalloc(&X, n * 4);
    (void)hipMalloc(&bad, 8);
    (void)hipMalloc(&fb, 8);
    (void)hipMemset(bad, 0, 8);
    (void)hipMemset(X, 0, n * 4);
    (void)hipHostMalloc((void **)&h_pub, 64, hipHostMallocCoherent | hipHostMallocMapped);
    h_pub[0] = h_pub[1] = 0;
    (void)hipHostGetDevicePointer((void **)&d_pub, h_pub, 0);
    uint32_t epoch = 0;
    while (!stop.load()) {
        epoch++;
        hipLaunchKernelGGL(k_write, dim3((n + 255) / 256), dim3(256), 0, st, X, n, epoch, 7919u);
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, X, d_pub, epoch);
        while (__atomic_load_n(&h_pub[1], __ATOMIC_ACQUIRE) != epoch) {}
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_read, dim3(1), dim3(64), 0, st, X, n, epoch, epoch * 3u + r, bad, fb);
    }
    (void)hipStreamSynchronize(st);
    unsigned long long hb = 0;
    uint32_t hf[2] = {0, 0};
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hf, fb, 8, hipMemcpyDeviceToHost);
    total_bad[t] = hb;
    epochs[t] = epoch;
    if (hb) printf("thread %d: %llu stale reads in %u epochs (one of them: read %u, written %u)\n", t, hb, epoch, hf[0], hf[1]);
}

int main(int argc, char **argv)
{
    const int seconds = argc > 1 ? atoi(argv[1]) : 20, T = argc > 2 ? atoi(argv[2]) : 2;
    std::vector<unsigned long long> bad(T, 0), ep(T, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t, bad.data(), ep.data());
    std::this_thread::sleep_for(std::chrono::seconds(seconds));
    stop.store(true);
    for (auto &x : th) x.join();
    unsigned long long b = 0, e = 0;
    for (int t = 0; t < T; t++) {
        b += bad[t];
        e += ep[t];
    }
    printf("stale_read_two_queues: %llu stale reads in %llu epochs, %d host threads, %d s\n", b, e, T, seconds);
    return b ? 1 : 0;
}
