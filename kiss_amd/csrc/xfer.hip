// xfer.hip -- the host <-> device legs of the host-pointer entry points (kiss_hip_ctx_suffix_sort_dna_u32 and friends).
//
// The reference's timed region is host S -> host SA (include/command/suffix_sort.hpp:57-61), so for a drop-in caller
// the PCIe legs are part of the metric: 1 byte per base in, 4 bytes per base out (12.5 GB at chm13 size).
//   * host memory that is already page-locked (hipHostMalloc / hipHostRegister, torch pin_memory): ONE async copy on the
//     ctx stream at the full PCIe rate;
//   * pageable memory (what `kiss::vector` / std::vector / numpy hand over): XF_THREADS worker threads, each with two
//     page-locked bounce buffers owned by the ctx and its own stream -- the CPU copy of one chunk (and, for a freshly
//     allocated destination, its page faults) overlaps the DMA of the previous one and the other threads' chunks.
//     hipMemcpy on pageable memory does the same thing with ONE thread.
// Nothing here computes: bytes are moved as they are.
#include "kiss_internal.hpp"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr int XF_MAX_THREADS = 16;
// copy threads for pageable memory (each with two bounce buffers and a stream): 8, or what KISS_HIP_XFER_THREADS said when
// the context was created (KissOpts; the pool below holds buffers for XF_MAX_THREADS and creates them on demand)
static int xf_threads(const kiss_hip_ctx *ctx)
{
    const int v = ctx->opts.xfer_threads ? ctx->opts.xfer_threads : 8;
    return v < 1 ? 1 : (v > XF_MAX_THREADS ? XF_MAX_THREADS : v);
}
#define XF_THREADS xf_threads(ctx)
constexpr size_t XF_CHUNK = 16ull << 20;

bool host_is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError(); // plain malloc'ed memory: "invalid value", not an error for us
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int xfer_pool(kiss_hip_ctx *ctx)
{
    const int want = XF_THREADS;
    if (ctx->xf_ready >= want) return KISS_HIP_OK;
    for (int t = 0; t < want; t++) { // (a call that failed half-way is resumed, not repeated)
        if (!ctx->xf_stream[t] && hipStreamCreateWithFlags(&ctx->xf_stream[t], hipStreamNonBlocking) != hipSuccess)
            return KISS_HIP_E_HIP;
        for (int b = 0; b < 2; b++) {
            if (!ctx->xf_pin[t][b] && hipHostMalloc(&ctx->xf_pin[t][b], XF_CHUNK, hipHostMallocDefault) != hipSuccess)
                return KISS_HIP_E_NOMEM;
            if (!ctx->xf_done[t][b] && hipEventCreateWithFlags(&ctx->xf_done[t][b], hipEventDisableTiming) != hipSuccess)
                return KISS_HIP_E_HIP;
        }
    }
    ctx->xf_ready = want;
    return KISS_HIP_OK;
}

// to_device: h -> d, else d -> h
int xfer_staged(kiss_hip_ctx *ctx, void *d, void *h, uint64_t bytes, bool to_device)
{
    KTRY(xfer_pool(ctx));
    const uint64_t chunks = div_up(bytes, XF_CHUNK);
    const int threads = (int)(chunks < (uint64_t)XF_THREADS ? chunks : XF_THREADS);
    std::atomic<int> status{KISS_HIP_OK};
    const int device = ctx->device;
    auto work = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) {
            status = KISS_HIP_E_HIP;
            return;
        }
        hipStream_t st = ctx->xf_stream[t];
        if (to_device) {
            bool used[2] = {false, false};
            int b = 0;
            for (uint64_t c = (uint64_t)t; c < chunks && status == KISS_HIP_OK; c += (uint64_t)threads, b ^= 1) {
                const uint64_t off = c * XF_CHUNK;
                const size_t len = bytes - off < XF_CHUNK ? (size_t)(bytes - off) : XF_CHUNK;
                if (used[b]) (void)hipEventSynchronize(ctx->xf_done[t][b]);
                std::memcpy(ctx->xf_pin[t][b], (const char *)h + off, len);
                if (hipMemcpyAsync((char *)d + off, ctx->xf_pin[t][b], len, hipMemcpyHostToDevice, st) != hipSuccess) {
                    status = KISS_HIP_E_HIP;
                    return;
                }
                (void)hipEventRecord(ctx->xf_done[t][b], st);
                used[b] = true;
            }
        } else {
            // software pipeline: the DMA of chunk c+1 is in flight while chunk c is copied out of its bounce buffer
            uint64_t c = (uint64_t)t;
            int b = 0;
            auto issue = [&](uint64_t cc, int bb) -> bool {
                const uint64_t off = cc * XF_CHUNK;
                const size_t len = bytes - off < XF_CHUNK ? (size_t)(bytes - off) : XF_CHUNK;
                if (hipMemcpyAsync(ctx->xf_pin[t][bb], (const char *)d + off, len, hipMemcpyDeviceToHost, st) != hipSuccess)
                    return false;
                return hipEventRecord(ctx->xf_done[t][bb], st) == hipSuccess;
            };
            if (c < chunks && !issue(c, b)) {
                status = KISS_HIP_E_HIP;
                return;
            }
            for (; c < chunks && status == KISS_HIP_OK; c += (uint64_t)threads, b ^= 1) {
                const uint64_t next = c + (uint64_t)threads;
                if (next < chunks && !issue(next, b ^ 1)) {
                    status = KISS_HIP_E_HIP;
                    return;
                }
                if (hipEventSynchronize(ctx->xf_done[t][b]) != hipSuccess) {
                    status = KISS_HIP_E_HIP;
                    return;
                }
                const uint64_t off = c * XF_CHUNK;
                const size_t len = bytes - off < XF_CHUNK ? (size_t)(bytes - off) : XF_CHUNK;
                std::memcpy((char *)h + off, ctx->xf_pin[t][b], len);
            }
        }
        if (hipStreamSynchronize(st) != hipSuccess) status = KISS_HIP_E_HIP;
    };
    // no exception may cross the C ABI: a thread that can not be started (std::system_error) just leaves its chunks to
    // the calling thread
    std::vector<std::thread> th;
    std::vector<int> orphan;
    for (int t = 1; t < threads; t++) {
        try {
            th.emplace_back(work, t);
        } catch (...) {
            orphan.push_back(t);
        }
    }
    if (threads > 0) work(0);
    for (int t : orphan) work(t);
    for (auto &x : th) x.join();
    return status;
}

} // namespace

// ---- early download --------------------------------------------------------------------------------------------
// The reference's timed region ends with SA in host memory; 12.5 GB at 57 GB/s is 218 ms, more than twice the sort.  The
// L-type part of a bucket is final once the L sweep has passed it, the S-type part once the S sweep has: each such
// stretch is queued on a copy stream behind an event of the compute stream, so the PCIe transfer starts while the sweeps
// are still running instead of after them.  Armed only for page-locked destinations and the bounded orders (the
// exact-order finish rewrites SA afterwards).
int kiss_early_out(kiss_hip_ctx *ctx, const uint32_t *d_SA, uint64_t lo, uint64_t hi)
{
    if (!ctx->early_host_SA || hi <= lo) return KISS_HIP_OK;
    if (!ctx->early_stream) KCHECK(hipStreamCreateWithFlags(&ctx->early_stream, hipStreamNonBlocking));
    if (ctx->early_used == ctx->early_events.size()) {
        hipEvent_t e;
        KCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->early_events.push_back(e);
    }
    hipEvent_t ev = ctx->early_events[ctx->early_used++];
    KCHECK(hipEventRecord(ev, ctx->stream));
    KCHECK(hipStreamWaitEvent(ctx->early_stream, ev, 0));
    KCHECK(hipMemcpyAsync(ctx->early_host_SA + lo, d_SA + lo, (hi - lo) * sizeof(uint32_t), hipMemcpyDeviceToHost,
                          ctx->early_stream));
    ctx->early_bytes += (hi - lo) * sizeof(uint32_t);
    return KISS_HIP_OK;
}

bool kiss_host_is_pinned(const void *p) { return host_is_pinned(p); }

// ---- page faults of a fresh destination, taken while the device sorts -------------------------------------------
// A caller that hands over memory it has never touched (numpy.empty, new T[n], malloc) pays one page fault per 4 KiB of
// SA inside the download: 0.9-1.2 s for 12.5 GB against 0.25 s for the copy itself.  The host has nothing to do while
// the device sorts, so worker threads ask the kernel to populate the pages then (MADV_POPULATE_WRITE: maps them
// writable without changing a byte; where the call is not available nothing happens and the copy faults them in as
// before).
#include <sys/mman.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
struct kiss_prefault {
    std::vector<std::thread> th;
    std::atomic<bool> stop{false};
};

// has this range been touched before?  64 pages spread over it are asked for (mincore); memory that is already mapped
// needs no helper (populating 3 M present pages costs 0.15 s of page-table walks and slows the copy down)
static bool looks_untouched(uint64_t lo, uint64_t hi, uint64_t page)
{
    const uint64_t pages = (hi - lo) / page;
    unsigned missing = 0, asked = 0;
    for (uint64_t i = 0; i < 64; i++) {
        const uint64_t q = lo + (pages * i / 64) * page;
        unsigned char vec = 0;
        if (mincore((void *)(uintptr_t)q, (size_t)page, &vec) != 0) return false; // cannot tell: leave it alone
        asked++;
        missing += (vec & 1u) ? 0u : 1u;
    }
    return asked && missing * 2 > asked;
}

void *kiss_prefault_start(kiss_hip_ctx *ctx, void *p, uint64_t bytes)
{
    if (ctx->opts.no_prefault || bytes < (64ull << 20)) return nullptr;
    const uint64_t page = (uint64_t)sysconf(_SC_PAGESIZE);
    const uint64_t lo = ((uint64_t)(uintptr_t)p + page - 1) / page * page, hi = ((uint64_t)(uintptr_t)p + bytes) / page * page;
    if (hi <= lo || !looks_untouched(lo, hi, page)) return nullptr;
    kiss_prefault *h = new (std::nothrow) kiss_prefault;
    if (!h) return nullptr;
    int T = 4; // measured 4 / 8 / 14: more helpers take more from the thread that drives the sort than they give
    if (ctx->opts.prefault_threads >= 1 && ctx->opts.prefault_threads <= 64) T = ctx->opts.prefault_threads;
    const uint64_t pages = (hi - lo) / page, per = (pages + (uint64_t)T - 1) / (uint64_t)T;
    for (int t = 0; t < T; t++) {
        const uint64_t a = lo + (uint64_t)t * per * page, b = a + per * page < hi ? a + per * page : hi;
        if (a >= b) break;
        try {
            h->th.emplace_back([a, b, h] {
                // in slices, so that the pages the download reaches first are there first (and so that the helper can
                // be told to stop once the download is through)
                const uint64_t step = 64ull << 20;
                for (uint64_t q = a; q < b && !h->stop.load(std::memory_order_relaxed); q += step)
                    if (madvise((void *)(uintptr_t)q, (size_t)((b - q) < step ? (b - q) : step), MADV_POPULATE_WRITE) != 0) return;
            });
        } catch (...) {
            break; // fewer helpers: the copy takes the remaining faults itself
        }
    }
    return h;
}

void kiss_prefault_join(void *handle)
{
    kiss_prefault *h = static_cast<kiss_prefault *>(handle);
    if (!h) return;
    h->stop.store(true, std::memory_order_relaxed);
    for (auto &t : h->th) t.join();
    delete h;
}

void kiss_xfer_free(kiss_hip_ctx *ctx)
{
    for (hipEvent_t e : ctx->early_events) (void)hipEventDestroy(e);
    ctx->early_events.clear();
    if (ctx->early_stream) (void)hipStreamDestroy(ctx->early_stream);
    ctx->early_stream = nullptr;
    for (int t = 0; t < XF_MAX_THREADS; t++) {
        for (int b = 0; b < 2; b++) {
            if (ctx->xf_pin[t][b]) (void)hipHostFree(ctx->xf_pin[t][b]);
            if (ctx->xf_done[t][b]) (void)hipEventDestroy(ctx->xf_done[t][b]);
            ctx->xf_pin[t][b] = nullptr;
            ctx->xf_done[t][b] = nullptr;
        }
        if (ctx->xf_stream[t]) (void)hipStreamDestroy(ctx->xf_stream[t]);
        ctx->xf_stream[t] = nullptr;
    }
    ctx->xf_ready = false;
}

// Both return after the bytes have arrived.  The caller has synchronised the stream that produced / will consume the
// device buffer (the staged form uses its own streams).
int kiss_xfer_h2d(kiss_hip_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes)
{
    if (!bytes) return KISS_HIP_OK;
    if (host_is_pinned(h_src) || bytes < (1u << 20)) {
        KCHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
        return KISS_HIP_OK;
    }
    return xfer_staged(ctx, d_dst, const_cast<void *>(h_src), bytes, true);
}

int kiss_xfer_d2h(kiss_hip_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!bytes) return KISS_HIP_OK;
    if (host_is_pinned(h_dst) || bytes < (1u << 20)) {
        KCHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
        return KISS_HIP_OK;
    }
    return xfer_staged(ctx, const_cast<void *>(d_src), h_dst, bytes, false);
}
