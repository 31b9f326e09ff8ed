// multi.hip -- ONE process driving several GPUs: the sharded suffix sort behind the C ABI
// (kiss_hip_multi_*, kiss_hip_suffix_sort_dna_u32_multi; SURVEY.md section 8(b) `kiss_hip_opts{ngpus, device ids}` and
// section 8(e)).  What `kiss suffix_sort --gpus N` (reference include/command/suffix_sort.hpp:37-61) runs.
//
// Same pipeline as kiss_amd/multi_gpu.py (one process per GPU over RCCL), with host threads instead of ranks and peer
// copies over xGMI (hipMemcpyPeerAsync, pulled by the receiver) instead of collectives -- every pair of devices uses its
// own link, which is the direct all-to-all shape SURVEY 8(e) asks for:
//   device 0      : packs the text; the other devices pull the packed text (n/4 bytes)
//   every device r: classify text slice r -> local ascending LMS list (key, position); histogram of the first 14 key bits
//   host          : sum of the counters and histograms -> G key ranges balanced by LMS count -> count matrix
//   every device r: stable partition of its list by destination; receiver g pulls its piece from every r in rank
//                   order (= ascending text position: the reference's tie-break survives, kiss1_core.hpp:131-133)
//   every device g: k-ordered sort of its key range, in place
//   device 0      : pulls the sorted pieces (positions + context words) in key-range order and the near-end suffixes,
//                   runs placement + induction (one global dependency chain: it does not shard)
// No buffer is copied inside a device: the stages work on the ctx's own arrays, a device's own piece moves only because
// its offset in the receiving list differs (1/G of the data).  With one device nothing moves at all.
#include "kiss_internal.hpp"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr int MG_HIST_BITS = 14; // 7 bases: 16 384 bins split <= 64 key ranges finely enough; private LDS histograms
constexpr int MG_MAX_DEV = 64;
constexpr uint32_t MG_EXACT_H0 = KISS_EXACT_H0;

struct Barrier {
    std::mutex mu;
    std::condition_variable cv;
    int n = 1, waiting = 0;
    uint64_t gen = 0;
    bool aborted = false; // a rank thread could not be started: nobody waits for it any more
    void arrive()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return;
        const uint64_t g = gen;
        if (++waiting == n) {
            waiting = 0;
            gen++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g || aborted; });
        }
    }
    void abort()
    {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
    void reset(int parties)
    {
        std::lock_guard<std::mutex> lk(mu);
        n = parties;
        waiting = 0;
        aborted = false;
    }
};

} // namespace

struct kiss_hip_multi {
    int G = 0;
    int dev[MG_MAX_DEV] = {};
    kiss_hip_ctx *ctx[MG_MAX_DEV] = {};
    uint64_t max_n = 0;
    uint64_t *d_hist[MG_MAX_DEV] = {};  // 2^MG_HIST_BITS u64 on each device
    uint64_t *h_hist[MG_MAX_DEV] = {};  // page-locked host copies
    uint32_t *d_near[MG_MAX_DEV] = {};  // a rank's near-end positions, kept aside until device 0 pulls them
    uint64_t near_cap[MG_MAX_DEV] = {};
    kiss_hip_multi_stats stats{};
    // state shared by the rank threads of one call
    Barrier bar;
    std::atomic<int> status{KISS_HIP_OK};
    std::atomic<int> deep{0};
};

namespace {

using clk = std::chrono::steady_clock;

// The device work of a phase, when this rank's device is also another rank's (a test configuration: shares of one GPU):
// one rank at a time, for the reason given at api.hip: sort_dev.  Never held across a barrier.
struct SharedDeviceLock {
    std::unique_lock<std::mutex> lk;
    SharedDeviceLock(const kiss_hip_multi *mc, int r)
    {
        // (round 4: always, not only when this device is listed twice -- a plain sort of another context on the same device
        //  from another thread of the process must not run beside this phase either)
        if (!mc->ctx[r]->opts.no_serialize) lk = std::unique_lock<std::mutex>(kiss_device_mutex(mc->dev[r]));
    }
};

// key-range boundaries (on the first MG_HIST_BITS key bits) that balance the far LMS count over G ranks; group of a
// bin = #{s in sp : s <= bin} (the rule of kiss_amd/multi_gpu.py::choose_splitters, tests/test_multi_gpu.py)
void choose_splitters(const std::vector<uint64_t> &hist, int G, uint32_t *sp)
{
    uint64_t total = 0;
    for (uint64_t v : hist) total += v;
    size_t bin = 0;
    uint64_t cum = 0; // items in bins [0, bin)
    for (int g = 1; g < G; g++) {
        const uint64_t target = (total * (uint64_t)g + (uint64_t)G - 1) / (uint64_t)G;
        while (bin < hist.size() && cum + hist[bin] < target) cum += hist[bin++];
        // bin = first index whose inclusive prefix reaches the target; bins [0, bin + 1) hold >= target items
        size_t b = bin + 1;
        if (b > hist.size()) b = hist.size();
        sp[g - 1] = (uint32_t)b;
    }
}

void group_counts(const uint64_t *hist, size_t bins, const uint32_t *sp, int G, uint64_t *out)
{
    size_t lo = 0;
    for (int g = 0; g < G; g++) {
        const size_t hi = g + 1 < G ? sp[g] : bins;
        uint64_t c = 0;
        for (size_t b = lo; b < hi && b < bins; b++) c += hist[b];
        out[g] = c;
        if (hi > lo) lo = hi;
    }
}

int copy_between(kiss_hip_multi *mc, int dst_rank, void *dst, int src_rank, const void *src, uint64_t bytes, hipStream_t st)
{
    kiss_hip_ctx *ctx = mc->ctx[dst_rank];
    if (bytes == 0) return KISS_HIP_OK;
    if (mc->dev[dst_rank] == mc->dev[src_rank])
        KCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    else
        KCHECK(hipMemcpyPeerAsync(dst, mc->dev[dst_rank], src, mc->dev[src_rank], bytes, st));
    return KISS_HIP_OK;
}

int near_keep(kiss_hip_multi *mc, int r, const uint32_t *src, uint64_t count)
{
    kiss_hip_ctx *ctx = mc->ctx[r];
    if (count == 0) return KISS_HIP_OK;
    if (count > mc->near_cap[r]) {
        if (mc->d_near[r]) (void)hipFree(mc->d_near[r]);
        mc->d_near[r] = nullptr;
        mc->near_cap[r] = 0;
        uint64_t cap = 65536;
        while (cap < count) cap *= 2;
        KCHECK(hipMalloc((void **)&mc->d_near[r], cap * sizeof(uint32_t)));
        mc->near_cap[r] = cap;
    }
    KCHECK(hipMemcpyAsync(mc->d_near[r], src, count * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    return KISS_HIP_OK;
}

struct Shared {
    const uint8_t *d_S;
    uint32_t *d_SA;
    uint64_t n;
    uint32_t k;      // order of the current attempt
    uint64_t depth;
    uint32_t h0;     // != 0: device 0 finishes with rank doubling from this order (exact order asked for)
    // filled by the ranks, read after a barrier
    uint64_t counts[MG_MAX_DEV][13];
    uint64_t m_local[MG_MAX_DEV], m_far_local[MG_MAX_DEV];
};

// one rank of one attempt.  Every rank executes every barrier: after a failure (mc->status) the work between the
// barriers is skipped, the barriers are not.
void rank_attempt(kiss_hip_multi *mc, Shared *sh, int r, clk::time_point *marks)
{
    const int G = mc->G;
    kiss_hip_ctx *ctx = mc->ctx[r];
    const uint64_t n = sh->n;
    const size_t bins = (size_t)1 << MG_HIST_BITS;
    auto ok = [&] { return mc->status.load() == KISS_HIP_OK; };
    auto fail = [&](int rc) {
        int expect = KISS_HIP_OK;
        if (rc != KISS_HIP_OK) mc->status.compare_exchange_strong(expect, rc);
    };
    auto sync = [&]() -> int {
        KCHECK(hipStreamSynchronize(ctx->stream));
        return KISS_HIP_OK;
    };
    auto mark = [&](int i) {
        if (r == 0) marks[i] = clk::now();
    };
    const uint64_t lo = (n * (uint64_t)r) / (uint64_t)G, hi = (n * (uint64_t)(r + 1)) / (uint64_t)G;
    const uint64_t words = div_up(n, 32) + 4;
    kiss_opts_refresh(ctx);
    ctx->hfar = ctx->hmerged = nullptr; // (tie flags of an exact-order sort_dev on this ctx: never this path's, api.hip)
    ctx->h_depth = 0;

    // ---- packed text: device 0 packs, the others pull
    if (ok() && r == 0) {
        SharedDeviceLock device_lock(mc, r);
        int rc = kiss_pack_text(ctx, sh->d_S, n);
        if (!rc) rc = sync();
        fail(rc);
    }
    mc->bar.arrive();
    mark(1);
    // ---- classify slice r; histogram of the first key bits of its far suffixes
    // publish: the first classification of an attempt tells the other ranks its numbers (they read them after the next
    // barrier); a repeat after a regrow gives the same numbers and leaves the shared block alone (others may be reading)
    auto classify = [&](bool publish) -> int {
        KTRY(kiss_classify(ctx, n, sh->depth, lo, hi));
        if (!publish) return (ctx->m == sh->m_local[r] && ctx->m_far == sh->m_far_local[r]) ? KISS_HIP_OK : KINTERNAL();
        for (int i = 0; i < 12; i++) sh->counts[r][i] = ctx->counts[i];
        sh->counts[r][12] = ctx->m_far;
        sh->m_local[r] = ctx->m;
        sh->m_far_local[r] = ctx->m_far;
        return KISS_HIP_OK;
    };
    if (ok()) {
        SharedDeviceLock device_lock(mc, r);
        int rc = KISS_HIP_OK;
        if (r > 0) rc = copy_between(mc, r, ctx->pk, 0, mc->ctx[0]->pk, words * sizeof(uint64_t), ctx->stream);
        if (!rc) rc = classify(true);
        if (!rc && G > 1) {
            rc = kiss_key_hist(ctx, ctx->keyA, ctx->m_far, MG_HIST_BITS, mc->d_hist[r]);
            if (!rc && hipMemcpyAsync(mc->h_hist[r], mc->d_hist[r], bins * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                      ctx->stream) != hipSuccess)
                rc = KISS_HIP_E_HIP;
        }
        if (!rc) rc = sync();
        fail(rc);
    }
    mc->bar.arrive();
    mark(2);
    // ---- every rank derives the same global numbers from what all ranks published
    uint64_t gcounts[12] = {0}, m_far_total = 0, near_total = 0, R[MG_MAX_DEV] = {0};
    uint32_t sp[MG_MAX_DEV] = {0};
    std::vector<uint64_t> cnt_((size_t)G * (size_t)G, 0); // [src][dst] far LMS suffixes
    auto cnt = [&](int src, int dst) -> uint64_t & { return cnt_[(size_t)src * (size_t)G + (size_t)dst]; };
    if (ok()) {
        for (int q = 0; q < G; q++) {
            for (int i = 0; i < 12; i++) gcounts[i] += sh->counts[q][i];
            m_far_total += sh->m_far_local[q];
            near_total += sh->m_local[q] - sh->m_far_local[q];
        }
        if (G > 1) {
            std::vector<uint64_t> gh(bins, 0);
            for (int q = 0; q < G; q++)
                for (size_t b = 0; b < bins; b++) gh[b] += mc->h_hist[q][b];
            choose_splitters(gh, G, sp);
            for (int q = 0; q < G; q++) group_counts(mc->h_hist[q], bins, sp, G, &cnt(q, 0));
        } else {
            cnt(0, 0) = sh->m_far_local[0];
        }
        for (int g = 0; g < G; g++)
            for (int q = 0; q < G; q++) R[g] += cnt(q, g);
    }
    // ---- capacity: the receiving list of this rank, and on device 0 the whole sorted list + the near-end suffixes
    if (ok()) {
        SharedDeviceLock device_lock(mc, r);
        int rc = KISS_HIP_OK;
        uint64_t need = R[r];
        if (r == 0 && m_far_total + near_total > need) need = m_far_total + near_total;
        if (need > ctx->m_cap) { // regrow (contents lost) and classify the slice again: skewed key ranges, (AC)^n
            rc = kiss_lms_reserve(ctx, need + need / 64 + 1024);
            if (!rc) rc = classify(false);
        }
        // the near-end suffixes (only the rank(s) owning the end of the text have any) are set aside: the exchange
        // overwrites the tail of the ascending list they sit in
        const uint64_t near_r = sh->m_local[r] - sh->m_far_local[r];
        if (!rc && G > 1) rc = near_keep(mc, r, ctx->lms_pos + sh->m_far_local[r], near_r);
        // ---- stable partition by destination rank
        if (!rc && G > 1)
            rc = kiss_partition_by_splitters(ctx, ctx->keyA, ctx->lms_pos, sh->m_far_local[r], MG_HIST_BITS, sp, G, ctx->keyB,
                                             ctx->posB);
        if (!rc) rc = sync();
        fail(rc);
    }
    mc->bar.arrive();
    mark(3);
    // ---- the exchange: receiver r pulls its piece from every source in rank order
    if (ok() && G > 1) {
        SharedDeviceLock device_lock(mc, r);
        int rc = KISS_HIP_OK;
        uint64_t roff = 0;
        for (int q = 0; q < G && !rc; q++) {
            uint64_t soff = 0;
            for (int g = 0; g < r; g++) soff += cnt(q, g);
            const uint64_t c = cnt(q, r);
            rc = copy_between(mc, r, ctx->keyA + roff, q, mc->ctx[q]->keyB + soff, c * sizeof(uint64_t), ctx->stream);
            if (!rc) rc = copy_between(mc, r, ctx->lms_pos + roff, q, mc->ctx[q]->posB + soff, c * sizeof(uint32_t), ctx->stream);
            roff += c;
        }
        if (!rc) rc = sync();
        fail(rc);
    }
    mc->bar.arrive(); // every piece has left its source before a sort reuses the source buffers
    mark(4);
    // ---- k-ordered sort of the received key range, in place
    if (ok()) {
        SharedDeviceLock device_lock(mc, r);
        // the digit counts of the emit pass are good only for the very list it emitted (one device, nothing exchanged)
        if (G > 1) ctx->rx_ghist_count = 0;
        ctx->m = ctx->m_far = R[r];
        int rc = kiss_lms_sort(ctx, n, sh->k, sh->depth);
        if (rc == KISS_INTERNAL_TOO_DEEP) { // exact order, ties deeper than the 32-bases-per-round path handles
            mc->deep.store(1);
            rc = sync();
        } else if (!rc) {
            rc = kiss_radix_check(ctx); // (synchronises)
            // Round 4: the sorted piece LEAVES for device 0 as soon as this rank has it (positions + context words, behind
            // the pieces of the lower ranks), on this rank's own stream: the G - 1 pieces travel over G - 1 links at once and
            // under the sorts that are still running, instead of being pulled one after the other by device 0 after the
            // barrier.  Device 0's own sort works on [0, R[0]) of the same arrays; their capacity was settled before the
            // exchange (the sort regrows tied-segment arrays only).
            if (!rc && r > 0 && R[r]) {
                uint64_t off = 0;
                for (int g = 0; g < r; g++) off += R[g];
                kiss_hip_ctx *c0 = mc->ctx[0];
                rc = copy_between(mc, 0, c0->lms_sorted_far + off, r, ctx->lms_sorted_far, R[r] * sizeof(uint32_t), ctx->stream);
                if (!rc) rc = copy_between(mc, 0, c0->lms_ctx_far + off, r, ctx->lms_ctx_far, R[r] * sizeof(uint32_t), ctx->stream);
                if (!rc) rc = sync();
            }
        }
        fail(rc);
    }
    mc->bar.arrive();
    mark(5);
    if (mc->deep.load()) return; // the caller runs the attempt again for k = 256 and finishes with rank doubling
    // ---- device 0: pull the sorted pieces and the near-end suffixes, placement + induction
    if (ok() && r == 0) {
        SharedDeviceLock device_lock(mc, r);
        int rc = KISS_HIP_OK;
        // (the sorted pieces of the other ranks are here already: each rank pushed its own behind its sort, see above)
        if (G > 1) {
            uint64_t noff = m_far_total;
            for (int q = 0; q < G && !rc; q++) {
                const uint64_t c = sh->m_local[q] - sh->m_far_local[q];
                rc = copy_between(mc, 0, ctx->lms_pos + noff, q, mc->d_near[q], c * sizeof(uint32_t), ctx->stream);
                noff += c;
            }
        }
        if (!rc) rc = sync();
        mark(6);
        if (!rc) {
            for (int i = 0; i < 12; i++) ctx->counts[i] = gcounts[i];
            ctx->m = m_far_total + near_total;
            ctx->m_far = m_far_total;
            ctx->stats.m = ctx->m;
            for (int g = 0; g < G && g < 8; g++) mc->stats.piece[g] = R[g];
            mc->stats.m = ctx->m;
            rc = kiss_place_lms(ctx, n, sh->k, sh->depth);
        }
        // exact order: the doubling over the LMS suffixes before the induction, as in the one-device path (api.hip: sort_dev);
        // tie flags by comparison (the sort's own flags stayed on the devices that sorted), bin sizes of the rank array
        // counted from the gathered list (the ascending list is spread over the devices)
        bool lms_resolved = false;
        if (!rc && sh->h0 && !ctx->opts.no_lms_exact) {
            ctx->lms_pos_complete = false;
            ctx->hfar = nullptr;
            rc = kiss_lms_exact_refine(ctx, n, sh->h0, sh->d_SA, &lms_resolved);
        }
        if (!rc) rc = kiss_induce(ctx, n, sh->d_SA);
        if (!rc && sh->h0 && !lms_resolved) rc = kiss_exact_refine(ctx, n, sh->h0, sh->d_SA);
        if (!rc && sh->h0 && lms_resolved) ctx->stats.refine_form = 1;
        if (!rc) rc = sync();
        if (!rc) rc = kiss_radix_check(ctx);
        fail(rc);
    }
    mc->bar.arrive();
    mark(7);
}

int multi_sort_dev(kiss_hip_multi *mc, const uint8_t *d_S, uint64_t n, uint32_t k, int algo, uint32_t *d_SA)
{
    if (!mc || !d_SA || (n && !d_S)) return KISS_HIP_E_INVALID;
    if (n > mc->max_n || n > KISS_HIP_MAX_N) return KISS_HIP_E_INVALID;
    if (algo != KISS_HIP_ALGO_PARALLEL_SORTING && algo != KISS_HIP_ALGO_PREFIX_DOUBLING) return KISS_HIP_E_INVALID;
    const int G = mc->G;
    std::memset(&mc->stats, 0, sizeof mc->stats);
    mc->stats.n = n;
    mc->stats.ndev = (uint32_t)G;
    if (n == 0) { // kiss1_core.hpp:237-238
        kiss_hip_ctx *ctx = mc->ctx[0];
        KCHECK(hipSetDevice(ctx->device));
        ctx->stream = ctx->own_stream;
        KTRY(kiss_zero_u32(ctx, d_SA, 1));
        KCHECK(hipStreamSynchronize(ctx->stream));
        return KISS_HIP_OK;
    }
    Shared sh;
    std::memset(&sh, 0, sizeof sh);
    sh.d_S = d_S;
    sh.d_SA = d_SA;
    sh.n = n;
    sh.k = k;
    sh.depth = kiss_depth_of(n, k);
    sh.h0 = 0;
    if (algo == KISS_HIP_ALGO_PREFIX_DOUBLING && sh.depth == 0 && n >= 4ull * MG_EXACT_H0 + 1024) {
        // exact order by the bounded phase + rank doubling over the whole SA on device 0 (api.hip: sort_dev)
        sh.h0 = MG_EXACT_H0;
        sh.k = MG_EXACT_H0;
        sh.depth = kiss_depth_of(n, sh.k);
    }
    // (PREFIX_DOUBLING with a bounded k: the deterministic k-ordered array, like the single-device entry)
    mc->status.store(KISS_HIP_OK);
    mc->bar.reset(G);
    clk::time_point marks[8];
    const clk::time_point t_begin = clk::now();
    for (int attempt = 0; attempt < 2; attempt++) {
        mc->deep.store(0);
        marks[0] = clk::now();
        auto body = [&](int r) {
            kiss_hip_ctx *ctx = mc->ctx[r];
            int rc = KISS_HIP_OK;
            if (hipSetDevice(ctx->device) != hipSuccess) rc = KISS_HIP_E_HIP;
            ctx->stream = ctx->own_stream;
            if (!rc) rc = kiss_workspace_ready(ctx);
            std::memset(&ctx->stats, 0, sizeof ctx->stats);
            ctx->stats.n = n;
            ctx->stats.k = sh.k;
            ctx->n = n;
            ctx->m = ctx->m_far = 0;
            if (rc) {
                int expect = KISS_HIP_OK;
                mc->status.compare_exchange_strong(expect, rc);
            }
            rank_attempt(mc, &sh, r, marks);
            ktimer_collect(ctx);
        };
        if (G == 1) {
            body(0);
        } else {
            std::vector<std::thread> th;
            bool started_all = true;
            try {
                th.reserve((size_t)G);
                for (int r = 1; r < G; r++) th.emplace_back(body, r);
            } catch (...) { // no exception crosses the ABI: the ranks that did start are let through their barriers
                started_all = false;
                int expect = KISS_HIP_OK;
                mc->status.compare_exchange_strong(expect, KISS_HIP_E_NOMEM);
                mc->bar.abort();
            }
            if (started_all) body(0);
            for (auto &t : th) t.join();
            if (!started_all) mc->bar.reset(G);
        }
        if (mc->status.load() != KISS_HIP_OK) break;
        if (!mc->deep.load()) break;
        if (attempt || n < 4ull * MG_EXACT_H0 + 1024) { // (texts that short never tie that deep)
            mc->status.store(KINTERNAL());
            break;
        }
        sh.h0 = MG_EXACT_H0;
        sh.k = MG_EXACT_H0;
        sh.depth = kiss_depth_of(n, sh.k);
    }
    const int rc = mc->status.load();
    (void)hipSetDevice(mc->ctx[0]->device);
    if (rc == KISS_HIP_OK) {
        auto ms = [&](int a, int b) { return std::chrono::duration<float, std::milli>(marks[b] - marks[a]).count(); };
        mc->stats.ms_pack = ms(0, 1);
        mc->stats.ms_classify = ms(1, 2);
        mc->stats.ms_partition = ms(2, 3);
        mc->stats.ms_exchange = ms(3, 4);
        mc->stats.ms_sort = ms(4, 5);
        mc->stats.ms_gather = ms(5, 6);
        mc->stats.ms_induce = ms(6, 7);
        mc->stats.ms_total = std::chrono::duration<float, std::milli>(clk::now() - t_begin).count();
        mc->stats.refine_depth = sh.h0;
    }
    return rc;
}

void multi_free(kiss_hip_multi *mc)
{
    for (int r = 0; r < mc->G; r++) {
        if (mc->ctx[r]) {
            (void)hipSetDevice(mc->dev[r]);
            if (mc->d_hist[r]) (void)hipFree(mc->d_hist[r]);
            if (mc->h_hist[r]) (void)hipHostFree(mc->h_hist[r]);
            if (mc->d_near[r]) (void)hipFree(mc->d_near[r]);
            kiss_hip_ctx_destroy(mc->ctx[r]);
        }
    }
    delete mc;
}

} // namespace

int kiss_host_sort(kiss_hip_ctx *ctx, const uint8_t *S, uint64_t n, uint32_t k, uint32_t *SA,
                   int (*sort)(void *, const uint8_t *, uint32_t *), void *arg); // api.hip

extern "C" {

// host only (no device is touched): the key-range rule of the multi-device sort on a caller's histogram -- what
// tests/test_multi_abi.py compares with kiss_amd/multi_gpu.py::choose_splitters / group_counts on CPU
int kiss_hip_debug_splitters(const uint64_t *hist, uint64_t bins, int groups, uint32_t *splitters_out,
                             uint64_t *group_counts_out)
{
    if (!hist || bins == 0 || bins > (1ull << 24) || groups < 1 || groups > MG_MAX_DEV || !splitters_out || !group_counts_out)
        return KISS_HIP_E_INVALID;
    std::vector<uint64_t> h(hist, hist + bins);
    choose_splitters(h, groups, splitters_out);
    group_counts(hist, (size_t)bins, splitters_out, groups, group_counts_out);
    return KISS_HIP_OK;
}

int kiss_hip_multi_create(kiss_hip_multi **out, const int *devices, int ndev, uint64_t max_n)
{
    if (!out || !devices || ndev < 1 || ndev > MG_MAX_DEV || max_n > KISS_HIP_MAX_N) return KISS_HIP_E_INVALID;
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return KISS_HIP_E_NO_DEVICE;
    for (int r = 0; r < ndev; r++)
        if (devices[r] < 0 || devices[r] >= visible) return KISS_HIP_E_NO_DEVICE;
    kiss_hip_multi *mc = new (std::nothrow) kiss_hip_multi();
    if (!mc) return KISS_HIP_E_NOMEM;
    mc->G = ndev;
    mc->max_n = max_n;
    int rc = KISS_HIP_OK;
    for (int r = 0; r < ndev && !rc; r++) {
        mc->dev[r] = devices[r];
        // device 0 holds the whole sorted list (default reservation); the others about 1/G of the LMS suffixes (their
        // arrays regrow on demand: skewed key ranges)
        uint64_t cap = 0;
        if (r > 0) cap = (uint64_t)(0.32 * (double)max_n / (double)ndev * 1.25) + 65536;
        rc = kiss_hip_ctx_create_sized(&mc->ctx[r], devices[r], max_n, cap);
        if (rc) break;
        if (hipSetDevice(devices[r]) != hipSuccess) rc = KISS_HIP_E_HIP;
        const size_t hb = sizeof(uint64_t) << MG_HIST_BITS;
        if (!rc && hipMalloc((void **)&mc->d_hist[r], hb) != hipSuccess) rc = KISS_HIP_E_NOMEM;
        if (!rc && hipHostMalloc((void **)&mc->h_hist[r], hb, hipHostMallocDefault) != hipSuccess) rc = KISS_HIP_E_NOMEM;
    }
    // direct copies over xGMI where the devices can reach each other (otherwise the runtime stages them through the host)
    for (int a = 0; a < ndev && !rc; a++)
        for (int b = 0; b < ndev; b++) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                (void)hipSetDevice(devices[a]);
                (void)hipDeviceEnablePeerAccess(devices[b], 0); // "already enabled" is fine
            }
            (void)hipGetLastError();
        }
    // devices[0] runs the induction: its context words (4 bytes per base) are allocated here, not inside the first sort
    if (!rc && hipSetDevice(devices[0]) == hipSuccess) rc = kiss_need_ctx_words(mc->ctx[0]);
    if (rc) {
        multi_free(mc);
        return rc;
    }
    (void)hipSetDevice(devices[0]);
    *out = mc;
    return KISS_HIP_OK;
}

int kiss_hip_multi_destroy(kiss_hip_multi *mc)
{
    if (!mc) return KISS_HIP_E_INVALID;
    multi_free(mc);
    return KISS_HIP_OK;
}

kiss_hip_ctx *kiss_hip_multi_ctx(kiss_hip_multi *mc, int rank)
{
    if (!mc || rank < 0 || rank >= mc->G) return nullptr;
    return mc->ctx[rank];
}

int kiss_hip_multi_get_stats(const kiss_hip_multi *mc, kiss_hip_multi_stats *out)
{
    if (!mc || !out) return KISS_HIP_E_INVALID;
    *out = mc->stats;
    return KISS_HIP_OK;
}

int kiss_hip_multi_suffix_sort_dna_u32_dev(kiss_hip_multi *mc, const uint8_t *d_S, uint64_t n, uint32_t k, int algo,
                                           uint32_t *d_SA)
{
    return multi_sort_dev(mc, d_S, n, k, algo, d_SA);
}

int kiss_hip_multi_suffix_sort_dna_u32(kiss_hip_multi *mc, const uint8_t *S, uint64_t n, uint32_t k, int algo, uint32_t *SA)
{
    if (!mc || !SA || (n && !S)) return KISS_HIP_E_INVALID;
    if (n > mc->max_n) return KISS_HIP_E_INVALID;
    struct Arg {
        kiss_hip_multi *mc;
        uint64_t n;
        uint32_t k;
        int algo;
    } a{mc, n, k, algo};
    return kiss_host_sort(mc->ctx[0], S, n, k, SA,
                          [](void *p, const uint8_t *d_S, uint32_t *d_SA) {
                              Arg *q = static_cast<Arg *>(p);
                              return multi_sort_dev(q->mc, d_S, q->n, q->k, q->algo, d_SA);
                          },
                          &a);
}

int kiss_hip_suffix_sort_dna_u32_multi(const uint8_t *S, uint64_t n, uint32_t k, int algo, uint32_t *SA, const int *devices,
                                       int ndev)
{
    if (!SA || (n && !S) || !devices || ndev < 1) return KISS_HIP_E_INVALID;
    if (n == 0) { // kiss1_core.hpp:237-238: no device work at all
        SA[0] = 0;
        return KISS_HIP_OK;
    }
    kiss_hip_multi *mc = nullptr;
    int rc = kiss_hip_multi_create(&mc, devices, ndev, n);
    if (rc) return rc;
    rc = kiss_hip_multi_suffix_sort_dna_u32(mc, S, n, k, algo, SA);
    kiss_hip_multi_destroy(mc);
    return rc;
}

} // extern "C"
