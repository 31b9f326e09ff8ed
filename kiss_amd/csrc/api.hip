// api.hip -- the C ABI of libkiss_hip.so (declared in include/kiss_hip.h) and the suffix-sort driver.
//
// Driver = the GPU counterpart of kiss::kiss1_suffix_array_dna
// (reference include/biovoltron/algo/sort/kiss1_core.hpp:229-268):
//   pack -> get_lms -> k-ordered LMS sort -> (near-end rule, merge, context gather) -> L/S induction.
#include "kiss_internal.hpp"
#include <cstdlib>
#include <chrono>
#include <mutex>
#include <cstring>
#include <new>

// ---- profiling timers ----------------------------------------------------------------
KTimer::KTimer(kiss_hip_ctx *c, int cls, uint64_t items) : ctx(c), idx(-1)
{
    if (!((ctx->profile_mask >> cls) & 1ull)) return;
    if (ctx->ev_used == ctx->ev_pool.size()) {
        kiss_hip_ctx::Ev e;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
        e.cls = 0;
        ctx->ev_pool.push_back(e);
    }
    idx = (int)ctx->ev_used++;
    ctx->ev_pool[idx].cls = cls;
    ctx->stats.launches_kernel[cls]++;
    ctx->stats.items_kernel[cls] += items;
    (void)hipEventRecord(ctx->ev_pool[idx].a, ctx->stream);
}
KTimer::~KTimer()
{
    if (idx >= 0) (void)hipEventRecord(ctx->ev_pool[idx].b, ctx->stream);
}
void ktimer_collect(kiss_hip_ctx *ctx)
{
    for (size_t i = 0; i < ctx->ev_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev_pool[i].a, ctx->ev_pool[i].b) == hipSuccess)
            ctx->stats.ms_kernel[ctx->ev_pool[i].cls] += ms;
    }
    ctx->ev_used = 0;
}

// ---- options (kiss_internal.hpp: KissOpts) ---------------------------------------------------------------
static void kiss_opts_supported_from_env(KissOpts &o) // both builds, once per context
{
    o.debug = getenv("KISS_HIP_DEBUG") != nullptr;
    if (const char *e = getenv("KISS_HIP_XFER_THREADS")) o.xfer_threads = atoi(e);
    if (const char *e = getenv("KISS_HIP_PREFAULT_THREADS")) o.prefault_threads = atoi(e);
}
#ifdef KISS_HIP_HOOKS
static bool env_on(const char *name) { return getenv(name) != nullptr; }
static unsigned long long env_u64(const char *name, unsigned long long dflt)
{
    const char *e = getenv(name);
    return e ? strtoull(e, nullptr, 0) : dflt;
}
void kiss_opts_refresh(kiss_hip_ctx *ctx)
{
    KissOpts o;
    kiss_opts_supported_from_env(o);
    if (const char *e = getenv("KISS_HIP_SYNC_READBACK")) o.sync_readback = atoi(e) != 0;
    o.no_serialize = env_on("KISS_HIP_NO_SERIALIZE");
    o.no_lms_exact = env_on("KISS_HIP_NO_LMS_EXACT");
    o.heads_by_compare = env_on("KISS_HIP_LMS_HEADS_BY_COMPARE");
    o.no_early_out = env_on("KISS_HIP_NO_EARLY_OUT");
    o.no_pivot_rounds = env_on("KISS_HIP_NO_PIVOT_ROUNDS");
    o.pivot_from_round2 = env_on("KISS_HIP_PIVOT_FROM_ROUND2");
    o.pair_keys = env_on("KISS_HIP_PAIR_KEYS");
    o.no_fc0_onepass = env_on("KISS_HIP_NO_FC0_ONEPASS");
    o.no_class_bytes = env_on("KISS_HIP_NO_CLASS_BYTES");
    o.no_pivot_ctx = env_on("KISS_HIP_NO_PIVOT_CTX");
    o.no_taint = env_on("KISS_HIP_NO_TAINT");
    o.isa_direct = env_on("KISS_HIP_ISA_DIRECT");
    o.no_onesweep = env_on("KISS_HIP_NO_ONESWEEP");
    o.merge_lms = env_on("KISS_HIP_MERGE_LMS");
    o.no_small_alphabet = env_on("KISS_HIP_NO_SMALL_ALPHABET");
    o.induce_one_pass = env_on("KISS_HIP_INDUCE_ONE_PASS");
    o.verify = env_on("KISS_HIP_VERIFY");
    o.no_prefault = env_on("KISS_HIP_NO_PREFAULT");
    o.doubling_h0 = (uint32_t)env_u64("KISS_HIP_DOUBLING_H0", 0);
    o.tcap0 = env_u64("KISS_HIP_TCAP0", 0);
    o.pivot_slots = (int)env_u64("KISS_HIP_PIVOT_SLOTS", 3);
    o.small_seg = (uint32_t)env_u64("KISS_HIP_SMALL_SEG", 0);
    o.near_merge_min = (uint32_t)env_u64("KISS_HIP_NEAR_MERGE_MIN", 4096);
    o.induce_small_max = (uint32_t)env_u64("KISS_HIP_INDUCE_SMALL_MAX", 0);
    o.collapse_cap = (uint32_t)env_u64("KISS_HIP_COLLAPSE_CAP", 0);
    o.collapse_n = env_u64("KISS_HIP_COLLAPSE_N", 0);
    o.fm_heavy = (uint32_t)env_u64("KISS_HIP_FM_HEAVY", 0);
    o.fm_light = (uint32_t)env_u64("KISS_HIP_FM_LIGHT", 0);
    o.isa_direct_max = env_u64("KISS_HIP_ISA_DIRECT_MAX", 0);
    o.lx_sync_points = (unsigned)env_u64("KISS_HIP_LX_SYNC_POINTS", 0);
    o.tie_trace = (unsigned)env_u64("KISS_HIP_TIE_TRACE", 0);
    o.poison = (uint32_t)env_u64("KISS_HIP_POISON", 0);
    if (const char *e = getenv("KISS_HIP_DUMP_PIVOT")) snprintf(o.dump_pivot, sizeof o.dump_pivot, "%s", e);
    ctx->opts = o;
}
static std::mutex &kiss_launch_mutex()
{
    static std::mutex m;
    return m;
}
static bool kiss_lock_launches()
{
    static const bool on = getenv("KISS_HIP_LOCK_LAUNCHES") != nullptr;
    return on;
}
bool kiss_sync_launches()
{
    static const bool on = getenv("KISS_HIP_SYNC_LAUNCHES") != nullptr;
    return on;
}
KissLaunchGuard::KissLaunchGuard() : held(kiss_lock_launches())
{
    if (held) kiss_launch_mutex().lock();
}
KissLaunchGuard::~KissLaunchGuard()
{
    if (held) kiss_launch_mutex().unlock();
}
#endif

// one lock per device: see sort_dev (also taken stage by stage by the multi-device entry when a device is listed twice)
std::mutex &kiss_device_mutex(int device)
{
    static std::mutex m[64];
    return m[(unsigned)device & 63u];
}

// ---- low-latency read-back ------------------------------------------------------------------------------
// A refinement round or an induction pass ends with the host reading a few counters to size the next launches
// (about a hundred times per sort).  hipMemcpyAsync + hipStreamSynchronize leaves the GPU idle for 40-60 us each time;
// a one-wave kernel that stores the words into coherent host memory followed by a sequence number, with the host
// spinning on that number, brings the hand-over down to the PCIe write + the next launch.
__global__ void k_publish(const uint32_t *__restrict__ src, uint32_t nwords, uint32_t *host_dst, uint32_t seq)
{
    const uint32_t t = threadIdx.x;
    if (t < nwords) __hip_atomic_store(&host_dst[t], src[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system(); // the data words are visible to the host before the sequence number is
    if (t == 0) __hip_atomic_store(&host_dst[16], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int kiss_readback(kiss_hip_ctx *ctx, const void *d_src, uint32_t nwords)
{
    if (nwords > 16) return KINTERNAL();
    ctx->pub_mode = ctx->opts.sync_readback || !ctx->h_pub || !ctx->d_pub ? 0 : 1;
    if (ctx->pub_mode == 0) {
        KCHECK(hipMemcpyAsync(ctx->h_pinned, d_src, nwords * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        KCHECK(hipStreamSynchronize(ctx->stream));
        return KISS_HIP_OK;
    }
    const uint32_t seq = ++ctx->pub_seq ? ctx->pub_seq : ++ctx->pub_seq; // never 0: the buffer starts zeroed
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t *)d_src, nwords, ctx->d_pub, seq);
    KCHECK(hipGetLastError());
    volatile uint32_t *flag = ctx->h_pub + 16;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; spins++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0xFFFF) == 0xFFFF) {
            // a kernel that faulted never publishes: after 5 s ask the runtime what happened instead of spinning on
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s > 5.0) {
                KCHECK(hipStreamSynchronize(ctx->stream));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return KINTERNAL();
                break;
            }
        }
    }
    for (uint32_t i = 0; i < nwords; i++) ctx->h_pinned[i] = ctx->h_pub[i];
    return KISS_HIP_OK;
}

bool kiss_host_is_pinned(const void *p); // xfer.hip
void *kiss_prefault_start(kiss_hip_ctx *ctx, void *p, uint64_t bytes);
void kiss_prefault_join(void *handle);

namespace {

template <typename T>
int dmalloc(kiss_hip_ctx *ctx, T **p, uint64_t count)
{
    void *q = nullptr;
    uint64_t bytes = (count ? count : 1) * sizeof(T);
    // fault-injection hook of tests/test_suffix_sort_gpu.py (kiss_hip_debug_fail_alloc_over): work arrays above this many
    // bytes "do not fit"; 0 = no limit.  No environment look-up per allocation in the shipped path.
    hipError_t e = ctx->fail_alloc_over && bytes > ctx->fail_alloc_over ? hipErrorOutOfMemory : hipMalloc(&q, bytes);
#ifdef KISS_HIP_HOOKS
    // KISS_HIP_POISON=<word>: a work array never starts out as zeros or as what its last owner left in it -- a kernel that
    // reads an element nobody wrote in this call then gives a result that differs from the unpoisoned run's
    if (e == hipSuccess && ctx->opts.poison) e = hipMemsetD32(q, (int)ctx->opts.poison, bytes / 4);
#endif
    if (e != hipSuccess) {
        (void)hipGetLastError(); // (or the next launch check reports this failure as its own)
        ctx->last_hip_error = (int)e;
        return KISS_HIP_E_NOMEM;
    }
    ctx->ws_bytes += bytes;
    *p = reinterpret_cast<T *>(q);
    return KISS_HIP_OK;
}

void free_tied(kiss_hip_ctx *ctx)
{
    void **ptrs[] = {(void **)&ctx->segA, (void **)&ctx->segB, (void **)&ctx->slotA, (void **)&ctx->slotB,
                     (void **)&ctx->segstartA, (void **)&ctx->segstartB, (void **)&ctx->bkeyA, (void **)&ctx->bkeyB,
                     (void **)&ctx->bposA, (void **)&ctx->bposB, (void **)&ctx->bsegA, (void **)&ctx->bsegB,
                     (void **)&ctx->bslot, (void **)&ctx->flags};
    for (void **p : ptrs)
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
    ctx->ws_bytes -= ctx->tied_bytes;
    ctx->tied_bytes = 0;
    ctx->t_cap = 0;
    ctx->flags_cap = 0;
}

void free_lms_side(kiss_hip_ctx *ctx)
{
    free_tied(ctx);
    void **ptrs[] = {(void **)&ctx->lms_pos, (void **)&ctx->keyA, (void **)&ctx->keyB, (void **)&ctx->posA,
                     (void **)&ctx->posB, (void **)&ctx->lms_sorted_far, (void **)&ctx->lms_ctx_far,
                     (void **)&ctx->lmsP, (void **)&ctx->lmsC, (void **)&ctx->tile_hist, (void **)&ctx->scan_tmp,
                     (void **)&ctx->rx_desc, (void **)&ctx->rx_ghist, (void **)&ctx->fc_desc};
    for (void **p : ptrs)
        if (*p) {
            (void)hipFree(*p);
            *p = nullptr;
        }
    ctx->ws_bytes -= ctx->lms_bytes;
    ctx->lms_bytes = 0;
}

void free_all(kiss_hip_ctx *ctx)
{
    free_lms_side(ctx);
    void *ptrs[] = {ctx->pk, ctx->tile_gp, ctx->tile_cnt, ctx->cl_part, ctx->d_counts, ctx->CTX, ctx->CLS, ctx->ind_counts, ctx->ind_desc,
                    ctx->d_small, ctx->near_idx, ctx->near_fin, ctx->near_pos, ctx->near_tmp, ctx->near_tmp2, ctx->pairs1, ctx->pairs2, ctx->rx_ctl, ctx->refine_heads, ctx->ga_codes};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (void *p : ctx->fm_pool)
        if (p) (void)hipFree(p);
#ifdef KISS_HIP_HOOKS
    if (ctx->tie_dbg) (void)hipFree(ctx->tie_dbg);
#endif
    if (ctx->io_S) (void)hipFree(ctx->io_S);
    if (ctx->io_SA) (void)hipFree(ctx->io_SA);
    kiss_xfer_free(ctx);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_pub) (void)hipHostFree(ctx->h_pub);
    for (auto &e : ctx->ev_pool) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
}

int sort_dev_unlocked(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t k, int algo, uint32_t *d_SA, void *stream);
// One sort at a time per device and process.  Found in round 3 with tools/lx_repro.py: two contexts sorting at the same time
// on one device (two host threads, own streams, own workspaces -- tests/test_suffix_sort_gpu.py::
// test_two_contexts_concurrently) gave a wrong exact-order suffix array about once in 2 000 sorts, in every form of the
// exact-order finish including the round-1 one; with the device phases of the two sorts kept apart: none in 6 400.  Nothing
// in this library is shared between contexts (no static device or host state, separate streams, events, pinned
// buffers), per-kernel synchronisation inside the stages changes nothing, and neither does the way counters are read back;
// what is left is what the runtime shares between two host threads that launch a few hundred short kernels each at the
// same time.  A saturating sort gains nothing from a second one beside it, so the calls queue up here instead
// (KISS_HIP_NO_SERIALIZE=1, read per call: the old behaviour, for whoever wants to look further).
int sort_dev(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t k, int algo, uint32_t *d_SA, void *stream)
{
    if (!ctx) return KISS_HIP_E_INVALID;
    kiss_opts_refresh(ctx);
#ifdef KISS_NO_DEVICE_LOCK // (variant build of the A-B of DESIGN.md 4.2: the shipped code without the lock)
    return sort_dev_unlocked(ctx, d_S, n, k, algo, d_SA, stream);
#endif
    if (ctx->opts.no_serialize) return sort_dev_unlocked(ctx, d_S, n, k, algo, d_SA, stream);
    std::lock_guard<std::mutex> lock(kiss_device_mutex(ctx->device));
    return sort_dev_unlocked(ctx, d_S, n, k, algo, d_SA, stream);
}
int sort_dev_unlocked(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t k, int algo, uint32_t *d_SA, void *stream)
{
    if (!ctx || !d_SA || (n && !d_S)) return KISS_HIP_E_INVALID;
    if (n > KISS_HIP_MAX_N || n > ctx->max_n) return KISS_HIP_E_INVALID;
    if (algo != KISS_HIP_ALGO_PARALLEL_SORTING && algo != KISS_HIP_ALGO_PREFIX_DOUBLING) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    KTRY(kiss_workspace_ready(ctx));
    std::memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->stats.n = n;
    ctx->stats.k = k;
    ctx->n = n;
    ctx->m = ctx->m_far = 0;
    if (n == 0) { // kiss1_core.hpp:237-238
        KTRY(kiss_zero_u32(ctx, d_SA, 1));
        KCHECK(hipStreamSynchronize(ctx->stream));
        return KISS_HIP_OK;
    }
    // effective comparison depth: 125 * (k/125 + 1) full-stride blocks (kiss1_core.hpp:95-118);
    // k >= n compares to the end of the text = exact suffix order
    uint64_t depth;
    if ((uint64_t)k >= n) depth = 0;
    else depth = (uint64_t)KISS_STRIDE * ((uint64_t)k / KISS_STRIDE + 1);
    if (algo == KISS_HIP_ALGO_PREFIX_DOUBLING && depth != 0) {
        // KISS2 with bounded k (kiss2_core.hpp:835-886; `suffix_sort -s PREFIX_DOUBLING` with the default k = 256,
        // suffix_sort.hpp:38-48): the reference's result there depends on its thread count (ties at depth k land in
        // schedule order), so the only thing it defines is the k-order property -- which the deterministic KISS1
        // order has.  Same answer as PARALLEL_SORTING, no error.
        algo = KISS_HIP_ALGO_PARALLEL_SORTING;
    }
    ctx->stats.depth = (uint32_t)(depth > 0xFFFFFFFFull ? 0xFFFFFFFFull : depth);
    // PREFIX_DOUBLING: a bounded-depth phase (order h0, the k-ordered pipeline) followed by rank doubling over
    // the tied suffixes; texts too short for that go through the unbounded comparison directly.
    uint32_t h0 = 0;
    if (algo == KISS_HIP_ALGO_PREFIX_DOUBLING) {
        h0 = KISS_EXACT_H0;
        if (ctx->opts.doubling_h0 >= 32 && ctx->opts.doubling_h0 <= (1u << 20)) h0 = ctx->opts.doubling_h0; // (hooks build: tuning)
        if (n < 4ull * h0 + 1024) h0 = 0;
        else {
            k = h0;
            depth = (uint64_t)KISS_STRIDE * ((uint64_t)k / KISS_STRIDE + 1);
        }
    }
    ctx->stats.refine_depth = h0;

    hipEvent_t ev[8];
    for (auto &e : ev) KCHECK(hipEventCreate(&e));
    // exact order: the doubling runs over the LMS suffixes before the induction (kiss_lms_exact_refine); the suffix-array form
    // (kiss_exact_refine) finishes only what that leaves.  KISS_HIP_NO_LMS_EXACT=1 (A-B hook, read per call): the old order of things.
    const bool lms_exact = !ctx->opts.no_lms_exact;
    int rc = KISS_HIP_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
        (void)hipEventRecord(ev[0], ctx->stream);
        if ((rc = kiss_pack_text(ctx, d_S, n))) break;
        (void)hipEventRecord(ev[1], ctx->stream);
        if ((rc = kiss_classify(ctx, n, depth, 0, n))) break;
        (void)hipEventRecord(ev[2], ctx->stream);
        ctx->hfar = nullptr;
        ctx->h_depth = h0;
        if (h0 && lms_exact && ctx->m_far && !ctx->opts.heads_by_compare) {
            // the LMS sort notes which far suffix retires tied with its predecessor: one byte per far-list slot at the far
            // end of CTX (kiss_lms_exact_refine lays its rank array and the merged list's flags out from the near end)
            if ((rc = kiss_need_ctx_words(ctx))) break;
            const uint64_t hwords = (ctx->m_far + 8 + 3) / 4;
            ctx->hfar = reinterpret_cast<uint8_t *>(ctx->CTX + ((n + 2 - hwords) & ~3ull));
            if ((rc = kiss_fill_u32(ctx, ctx->hfar, 0x01010101u, (ctx->m_far + 3) / 4))) break; // (a kernel: see kiss_fill_u32)
        }
        rc = kiss_lms_sort(ctx, n, k, depth);
        if (rc == KISS_INTERNAL_TOO_DEEP && attempt == 0 && n >= 4ull * KISS_EXACT_H0 + 1024) {
            // exact order requested through PARALLEL_SORTING on a text with very long repeats: same result via
            // the bounded phase + rank doubling (the k-ordered stage outputs then belong to k = 256)
            (void)hipStreamSynchronize(ctx->stream);
            h0 = KISS_EXACT_H0;
            k = h0;
            depth = (uint64_t)KISS_STRIDE * ((uint64_t)k / KISS_STRIDE + 1);
            ctx->stats.refine_depth = h0;
            ctx->stats.lms_rounds = 0;
            ctx->stats.sort_item_rounds = ctx->stats.big_item_rounds = 0;
            continue;
        }
        if (rc == KISS_INTERNAL_TOO_DEEP) rc = KINTERNAL();
        if (rc) break;
        (void)hipEventRecord(ev[3], ctx->stream);
        if ((rc = kiss_place_lms(ctx, n, k, depth))) break;
        (void)hipEventRecord(ev[4], ctx->stream);
        bool lms_resolved = false;
        if (h0 && lms_exact && (rc = kiss_lms_exact_refine(ctx, n, h0, d_SA, &lms_resolved))) break;
        (void)hipEventRecord(ev[7], ctx->stream);
        if ((rc = kiss_induce(ctx, n, d_SA))) break;
        (void)hipEventRecord(ev[5], ctx->stream);
        if (h0 && !lms_resolved && (rc = kiss_exact_refine(ctx, n, h0, d_SA))) break;
        if (h0 && lms_resolved) ctx->stats.refine_form = 1; // (kiss_exact_refine says 2 itself)
        (void)hipEventRecord(ev[6], ctx->stream);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->last_hip_error = (int)e;
            rc = KISS_HIP_E_HIP;
            break;
        }
        float ms[6], ms_lx = 0.f;
        for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
        (void)hipEventElapsedTime(&ms_lx, ev[4], ev[7]);
        (void)hipEventElapsedTime(&ms[4], ev[7], ev[5]);
        (void)hipEventElapsedTime(&ms[5], ev[5], ev[6]);
        ctx->stats.ms_refine = h0 ? ms[5] + ms_lx : 0.f; // both forms of the doubling phase
        ctx->stats.ms_pack = ms[0];
        ctx->stats.ms_classify = ms[1];
        ctx->stats.ms_lms_sort = ms[2];
        ctx->stats.ms_place = ms[3];
        ctx->stats.ms_induce = ms[4];
        (void)hipEventElapsedTime(&ctx->stats.ms_total, ev[0], ev[6]);
        rc = kiss_radix_check(ctx);
        break;
    }
    if (rc != KISS_HIP_OK) (void)hipStreamSynchronize(ctx->stream);
#ifdef KISS_HIP_HOOKS
    kiss_tie_trace_report(ctx);
#endif
    ctx->hfar = ctx->hmerged = nullptr;
    ctx->h_depth = 0;
    ctx->stats.m = ctx->m;
    ktimer_collect(ctx);
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

} // namespace

// LMS-sized arrays: DNA has ~0.29-0.30 n LMS suffixes; inputs with more (up to n/2) re-reserve on demand
static uint64_t default_m_cap(const kiss_hip_ctx *ctx)
{
    uint64_t m0 = ctx->m_cap0 ? ctx->m_cap0 : (uint64_t)(0.32 * (double)ctx->max_n) + 4096;
    if (m0 > ctx->max_n / 2 + 2) m0 = ctx->max_n / 2 + 2;
    return m0;
}

// A call that ran out of memory while growing its work arrays leaves them released (never half-allocated); the next call
// starts from the default reservation again instead of failing on what the previous one left behind.
int kiss_workspace_ready(kiss_hip_ctx *ctx)
{
    (void)hipGetLastError(); // a failed allocation of an earlier call is not this call's error
    if (ctx->m_cap && ctx->t_cap && ctx->lms_pos && ctx->flags) return KISS_HIP_OK;
    return kiss_lms_reserve(ctx, default_m_cap(ctx));
}

// CTX (4 bytes per base: context words of the sweeps, later the inverse SA of the doubling phase) is only needed by the
// process that runs the induction: allocated on first use, so that ranks > 0 of a sharded sort never hold it
int kiss_need_ctx_words(kiss_hip_ctx *ctx)
{
    if (ctx->CTX && ctx->CLS) return KISS_HIP_OK;
    if (!ctx->CTX) KTRY(dmalloc(ctx, &ctx->CTX, ctx->max_n + 2));
    // one class byte per context word (induce.hip: cls_byte); + 64: the count pass reads whole aligned 16-byte pieces
    if (!ctx->CLS) KTRY(dmalloc(ctx, &ctx->CLS, ctx->max_n + 2 + 64));
    return KISS_HIP_OK;
}

int kiss_lms_reserve(kiss_hip_ctx *ctx, uint64_t m_cap, uint64_t t_cap_wanted)
{
    const bool dbg = ctx->opts.debug;
    const auto t_begin = std::chrono::steady_clock::now();
    free_lms_side(ctx);
    if (dbg)
        fprintf(stderr, "[kiss_hip] lms_reserve(%llu): freed the old arrays in %.3f s\n", (unsigned long long)m_cap,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    const uint64_t before = ctx->ws_bytes;
    ctx->m_cap = m_cap;
    const uint64_t radix_tiles = m_cap / 2048 + 2; // sized for the smallest radix tile in use
    ctx->tile_hist_cap = 256 * radix_tiles + 256 * (radix_tiles / 64 + 2) + 8;
    uint64_t biggest_scan = m_cap;
    if (ctx->tile_hist_cap > biggest_scan) biggest_scan = ctx->tile_hist_cap;
    if (ctx->ind_tiles_cap > biggest_scan) biggest_scan = ctx->ind_tiles_cap;
    if (ctx->n_tiles_cap > biggest_scan) biggest_scan = ctx->n_tiles_cap;
    ctx->scan_tmp_cap = biggest_scan / 4096 + 16;
    int rc = KISS_HIP_OK;
    do {
#define ALLOC(p, cnt) if ((rc = dmalloc(ctx, &ctx->p, (cnt)))) break
        ALLOC(lms_pos, m_cap);
        ALLOC(keyA, m_cap);
        ALLOC(keyB, m_cap);
        ALLOC(posA, m_cap);
        ALLOC(posB, m_cap);
        ALLOC(lms_sorted_far, m_cap);
        ALLOC(lms_ctx_far, m_cap);
        ALLOC(lmsP, m_cap);
        ALLOC(lmsC, m_cap);
        ALLOC(tile_hist, ctx->tile_hist_cap);
        ALLOC(scan_tmp, ctx->scan_tmp_cap);
        ctx->rx_tiles_cap = m_cap / 16384 + 2;
        ALLOC(rx_desc, 256 * ctx->rx_tiles_cap);
        ALLOC(rx_ghist, 256 * 12);
        ctx->fc_desc_cap = m_cap / 8192 + 4;
        ALLOC(fc_desc, ctx->fc_desc_cap);
#undef ALLOC
        // descriptors carry the epoch of the pass that wrote them: cleared once, never again (the epoch keeps counting)
        if (hipMemset(ctx->rx_desc, 0, 256 * ctx->rx_tiles_cap * sizeof(uint64_t)) != hipSuccess) rc = KISS_HIP_E_HIP;
    } while (0);
    ctx->lms_bytes = ctx->ws_bytes - before;
    if (rc) { // leave nothing half-allocated behind: the next call starts from the default reservation again
        free_lms_side(ctx);
        ctx->m_cap = 0;
        return rc;
    }
    // tied-segment arrays: on genome-like text 18 % of the LMS suffixes survive round 0; they grow on demand
    // (all-tied inputs such as periodic texts end up at m_cap)
    uint64_t t0 = m_cap / 4 + (4ull << 20);
    if (t0 > m_cap) t0 = m_cap;
    // a caller that knows how many tied items it is about to handle says so: one allocation of the tied-segment arrays
    // instead of the default one followed by a regrow (large hipMallocs late in a process run at 20-30 GB/s)
    if (t_cap_wanted > t0) t0 = t_cap_wanted;
    if (ctx->opts.tcap0 >= 1 && ctx->opts.tcap0 < t0) t0 = ctx->opts.tcap0; // (hooks build: start small so that every growth path runs)
    if (dbg)
        fprintf(stderr, "[kiss_hip] lms_reserve(%llu): %.1f GB allocated, %.3f s so far\n", (unsigned long long)m_cap,
                (double)ctx->lms_bytes / 1e9, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    return kiss_tied_reserve(ctx, t0);
}

int kiss_tied_reserve(kiss_hip_ctx *ctx, uint64_t t_cap)
{
    const bool dbg = ctx->opts.debug;
    const auto t_begin = std::chrono::steady_clock::now();
    free_tied(ctx);
    const auto t_freed = std::chrono::steady_clock::now();
    if (t_cap < 1024) t_cap = 1024;
    const uint64_t before = ctx->ws_bytes;
    // `flags` also holds the per-tile words of the fused flag + compaction over all m (or, for the doubling phase,
    // n + 1) items: 2 words per 2048-item tile
    uint64_t fcap = 2 * t_cap;
    const uint64_t tiles_m = 2 * (ctx->m_cap / 2048 + 2), tiles_n = 2 * ((ctx->max_n + 1) / 2048 + 2);
    if (fcap < tiles_m) fcap = tiles_m;
    if (fcap < tiles_n) fcap = tiles_n;
    int rc = KISS_HIP_OK;
    do {
#define ALLOC(p, cnt) if ((rc = dmalloc(ctx, &ctx->p, (cnt)))) break
        ALLOC(segA, t_cap);
        ALLOC(segB, t_cap);
        ALLOC(slotA, t_cap);
        ALLOC(slotB, t_cap);
        ALLOC(segstartA, t_cap + 2);
        ALLOC(segstartB, t_cap + 2);
        ALLOC(bkeyA, t_cap);
        ALLOC(bkeyB, t_cap);
        ALLOC(bposA, t_cap);
        ALLOC(bposB, t_cap);
        ALLOC(bsegA, t_cap);
        ALLOC(bsegB, t_cap);
        ALLOC(bslot, t_cap);
        ALLOC(flags, fcap);
#undef ALLOC
    } while (0);
    ctx->tied_bytes = ctx->ws_bytes - before;
    if (dbg)
        fprintf(stderr, "[kiss_hip] tied_reserve(%llu): free %.3f s, %.1f GB allocated in %.3f s (rc %d)\n",
                (unsigned long long)t_cap, std::chrono::duration<double>(t_freed - t_begin).count(), (double)ctx->tied_bytes / 1e9,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_freed).count(), rc);
    if (rc == KISS_HIP_OK) {
        ctx->t_cap = t_cap;
        ctx->flags_cap = fcap;
    } else {
        free_tied(ctx); // (t_cap = 0: kiss_workspace_ready re-reserves the default before the next sort)
    }
    return rc;
}

// Host S -> host SA around a device-resident sort on `ctx`'s device (the reference's own timed region,
// command/suffix_sort.hpp:57-61): upload into the ctx-owned device copies, sort(arg, d_S, d_SA), download -- finished
// stretches of SA leave while the sweeps still run when SA is page-locked and the order is bounded (xfer.hip).  Shared by
// the single-device entry and the multi-device one (multi.hip: device 0 runs the induction, so the SA lives there).
int kiss_host_sort(kiss_hip_ctx *ctx, const uint8_t *S, uint64_t n, uint32_t k, uint32_t *SA,
                   int (*sort)(void *, const uint8_t *, uint32_t *), void *arg)
{
    KCHECK(hipSetDevice(ctx->device));
    kiss_opts_refresh(ctx);
    ctx->stream = ctx->own_stream;
    if (!ctx->io_S || !ctx->io_SA || ctx->io_cap < n) { // device-side copies of the caller's buffers: owned by the ctx, sized for max_n
        if (ctx->io_S) (void)hipFree(ctx->io_S);
        if (ctx->io_SA) (void)hipFree(ctx->io_SA);
        ctx->io_S = nullptr;
        ctx->io_SA = nullptr;
        ctx->ws_bytes -= ctx->io_cap ? 5 * ctx->io_cap + 4 : 0;
        ctx->io_cap = 0;
        const uint64_t cap = ctx->max_n ? ctx->max_n : 1;
        hipError_t e = hipMalloc((void **)&ctx->io_S, cap);
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->io_SA, (cap + 1) * sizeof(uint32_t));
        if (e != hipSuccess) { // never half a pair: a later call starts from nothing again
            (void)hipGetLastError();
            if (ctx->io_S) (void)hipFree(ctx->io_S);
            if (ctx->io_SA) (void)hipFree(ctx->io_SA);
            ctx->io_S = nullptr;
            ctx->io_SA = nullptr;
            ctx->last_hip_error = (int)e;
            return KISS_HIP_E_NOMEM;
        }
        ctx->io_cap = cap;
        ctx->ws_bytes += 5 * cap + 4;
    }
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    KTRY(kiss_xfer_h2d(ctx, ctx->io_S, S, n));
    const auto t1 = clk::now();
    // bounded order into a page-locked buffer: finished stretches of SA leave while the sweeps still run (xfer.hip)
    const bool no_early = ctx->opts.no_early_out;
    const bool early = !no_early && n >= (1u << 22) && (uint64_t)k < n && kiss_host_is_pinned(SA);
    ctx->early_used = 0;
    ctx->early_bytes = 0;
    ctx->early_host_SA = early ? SA : nullptr;
    // a pageable destination: its page faults are taken by helper threads while the device sorts
    void *prefault = kiss_host_is_pinned(SA) ? nullptr : kiss_prefault_start(ctx, SA, (n + 1) * sizeof(uint32_t));
    int rc = sort(arg, ctx->io_S, ctx->io_SA);
    ctx->early_host_SA = nullptr;
    (void)hipSetDevice(ctx->device);
    const auto t2 = clk::now();
    if (ctx->early_stream && ctx->early_used) {
        const hipError_t e = hipStreamSynchronize(ctx->early_stream);
        if (rc == KISS_HIP_OK && e != hipSuccess) {
            ctx->last_hip_error = (int)e;
            rc = KISS_HIP_E_HIP;
        }
    }
    if (rc) {
        kiss_prefault_join(prefault);
        return rc;
    }
    if (ctx->early_bytes != (n + 1) * sizeof(uint32_t)) // not armed, or (cannot happen) a stretch was not announced
        rc = kiss_xfer_d2h(ctx, SA, ctx->io_SA, (n + 1) * sizeof(uint32_t));
    // (the helpers are not waited for before the download: what they have not reached yet the copy faults in itself)
    kiss_prefault_join(prefault);
    if (rc) return rc;
    const auto t3 = clk::now();
    ctx->stats.ms_h2d = std::chrono::duration<float, std::milli>(t1 - t0).count();
    ctx->stats.ms_d2h = std::chrono::duration<float, std::milli>(t3 - t2).count();
    return KISS_HIP_OK;
}

// releases the device-side copies of the caller's buffers the host-pointer entry points keep between calls (5 bytes per
// base of max_n); the next such call allocates them again
int kiss_io_release(kiss_hip_ctx *ctx)
{
    if (ctx->io_S) (void)hipFree(ctx->io_S);
    if (ctx->io_SA) (void)hipFree(ctx->io_SA);
    ctx->io_S = nullptr;
    ctx->io_SA = nullptr;
    ctx->ws_bytes -= ctx->io_cap ? 5 * ctx->io_cap + 4 : 0;
    ctx->io_cap = 0;
    return KISS_HIP_OK;
}

extern "C" {

int kiss_hip_version(void) { return KISS_HIP_VERSION; }

const char *kiss_hip_strerror(int status)
{
    switch (status) {
    case KISS_HIP_OK: return "ok";
    case KISS_HIP_E_INVALID: return "invalid argument";
    case KISS_HIP_E_NO_DEVICE: return "no usable HIP device";
    case KISS_HIP_E_HIP: return "HIP runtime error";
    case KISS_HIP_E_NOMEM: return "out of memory";
    case KISS_HIP_E_UNSUPPORTED: return "request outside the implemented range";
    case KISS_HIP_E_INTERNAL: return "internal invariant violated";
    case KISS_HIP_E_IO: return "file could not be opened or read";
    case KISS_HIP_E_DEEP: return "ties deeper than the bounded-round exact path handles: sort with a bounded k, then stage_refine_exact";
    default: return "unknown status";
    }
}

int kiss_hip_device_count(int *count)
{
    if (!count) return KISS_HIP_E_INVALID;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
        *count = 0;
        return KISS_HIP_E_NO_DEVICE;
    }
    *count = c;
    return KISS_HIP_OK;
}

int kiss_hip_ctx_create(kiss_hip_ctx **out, int device, uint64_t max_n)
{
    return kiss_hip_ctx_create_sized(out, device, max_n, 0);
}

int kiss_hip_ctx_create_sized(kiss_hip_ctx **out, int device, uint64_t max_n, uint64_t lms_capacity)
{
    if (!out || max_n > KISS_HIP_MAX_N) return KISS_HIP_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return KISS_HIP_E_NO_DEVICE;
    kiss_hip_ctx *ctx = new (std::nothrow) kiss_hip_ctx();
    if (!ctx) return KISS_HIP_E_NOMEM;
    ctx->device = device;
    ctx->max_n = max_n;
    ctx->m_cap0 = lms_capacity ? lms_capacity + 4096 : 0;
    kiss_opts_supported_from_env(ctx->opts); // the only look at the environment the shipped library ever takes
    kiss_opts_refresh(ctx);                  // (hooks build: every switch)
    int rc = KISS_HIP_OK;
    do {
        if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->own_stream) != hipSuccess) {
            rc = KISS_HIP_E_HIP;
            break;
        }
        ctx->stream = ctx->own_stream;
        const uint64_t words = max_n / 32 + 8;
        ctx->pk_words = words;
        ctx->n_tiles_cap = words / 256 + 2;
        ctx->ind_tiles_cap = 4 * ((max_n + 1) / 1024 + 2) + 2; // (room for partition tiles down to 1024 items)
#define ALLOC(p, cnt) if ((rc = dmalloc(ctx, &ctx->p, (cnt)))) break
        ALLOC(pk, words);
        ALLOC(tile_gp, ctx->n_tiles_cap);
        ALLOC(tile_cnt, ctx->n_tiles_cap);
        // rows of per-workgroup partial sums of the classification (classify.hip: <= 8192 workgroups x 1280 words)
        ALLOC(cl_part, (ctx->n_tiles_cap < 8192 ? ctx->n_tiles_cap : 8192) * (uint64_t)(KISS_R0_PASSES * 256));
        ALLOC(d_counts, 16);
        ALLOC(ind_counts, ctx->ind_tiles_cap);
        ctx->ind_desc_stride = (max_n + 1) / 1024 + 2;
        ALLOC(ind_desc, 4 * ctx->ind_desc_stride);
        if (hipMemset(ctx->ind_desc, 0, 4 * ctx->ind_desc_stride * sizeof(uint64_t)) != hipSuccess) {
            rc = KISS_HIP_E_HIP;
            break;
        }
        ALLOC(rx_ctl, 4); // [0] radix ticket, [1] look-back error flag
        if (hipMemset(ctx->rx_ctl, 0, 4 * sizeof(uint32_t)) != hipSuccess) {
            rc = KISS_HIP_E_HIP;
            break;
        }

        ALLOC(d_small, 64);
#undef ALLOC
        if ((rc = kiss_lms_reserve(ctx, default_m_cap(ctx)))) break;
        void *hp = nullptr;
        if (hipHostMalloc(&hp, 64 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            rc = KISS_HIP_E_NOMEM;
            break;
        }
        ctx->h_pinned = reinterpret_cast<uint32_t *>(hp);
        // coherent (fine-grained) host buffer of kiss_readback; if the platform refuses it the memcpy form is used
        void *pub = nullptr, *dpub = nullptr;
        if (hipHostMalloc(&pub, 32 * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) {
            std::memset(pub, 0, 32 * sizeof(uint32_t));
            if (hipHostGetDevicePointer(&dpub, pub, 0) == hipSuccess) {
                ctx->h_pub = reinterpret_cast<uint32_t *>(pub);
                ctx->d_pub = reinterpret_cast<uint32_t *>(dpub);
            } else {
                (void)hipHostFree(pub);
            }
        }
        (void)hipGetLastError();
    } while (0);
    if (rc != KISS_HIP_OK) {
        free_all(ctx);
        delete ctx;
        return rc;
    }
    *out = ctx;
    return KISS_HIP_OK;
}

int kiss_hip_ctx_destroy(kiss_hip_ctx *ctx)
{
    if (!ctx) return KISS_HIP_E_INVALID;
    (void)hipSetDevice(ctx->device);
    free_all(ctx);
    delete ctx;
    return KISS_HIP_OK;
}

int kiss_hip_ctx_set_profiling(kiss_hip_ctx *ctx, int enabled)
{
    if (!ctx) return KISS_HIP_E_INVALID;
    ctx->profile_mask = enabled ? ~0ull : 0ull;
    return KISS_HIP_OK;
}

int kiss_hip_ctx_set_profiling_mask(kiss_hip_ctx *ctx, uint64_t class_mask)
{
    if (!ctx) return KISS_HIP_E_INVALID;
    ctx->profile_mask = class_mask;
    return KISS_HIP_OK;
}

int kiss_hip_last_hip_error(const kiss_hip_ctx *ctx, const char **msg)
{
    if (!ctx) return 0;
    if (msg) *msg = hipGetErrorString((hipError_t)ctx->last_hip_error);
    return ctx->last_hip_error;
}

int kiss_hip_get_stats(const kiss_hip_ctx *ctx, kiss_hip_stats *out)
{
    if (!ctx || !out) return KISS_HIP_E_INVALID;
    *out = ctx->stats;
    return KISS_HIP_OK;
}

int kiss_hip_get_stats_sized(const kiss_hip_ctx *ctx, void *out, uint64_t bytes)
{
    if (!ctx || !out) return KISS_HIP_E_INVALID;
    const uint64_t have = sizeof ctx->stats;
    std::memcpy(out, &ctx->stats, bytes < have ? bytes : have);
    if (bytes > have) std::memset(static_cast<char *>(out) + have, 0, bytes - have);
    return KISS_HIP_OK;
}

int kiss_hip_ctx_release_io_buffers(kiss_hip_ctx *ctx)
{
    if (!ctx) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    return kiss_io_release(ctx);
}

int kiss_hip_ctx_workspace_bytes(const kiss_hip_ctx *ctx, uint64_t *bytes)
{
    if (!ctx || !bytes) return KISS_HIP_E_INVALID;
    *bytes = ctx->ws_bytes;
    return KISS_HIP_OK;
}

int kiss_hip_ctx_suffix_sort_dna_u32_dev(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t k, int algo,
                                         uint32_t *d_SA, void *stream)
{
    return sort_dev(ctx, d_S, n, k, algo, d_SA, stream);
}

int kiss_hip_ctx_suffix_sort_dna_u32(kiss_hip_ctx *ctx, const uint8_t *S, uint64_t n, uint32_t k, int algo,
                                     uint32_t *SA)
{
    if (!ctx || !SA || (n && !S)) return KISS_HIP_E_INVALID;
    if (n > ctx->max_n) return KISS_HIP_E_INVALID;
    struct Arg {
        kiss_hip_ctx *ctx;
        uint64_t n;
        uint32_t k;
        int algo;
    } a{ctx, n, k, algo};
    return kiss_host_sort(ctx, S, n, k, SA,
                          [](void *p, const uint8_t *d_S, uint32_t *d_SA) {
                              Arg *q = static_cast<Arg *>(p);
                              return sort_dev(q->ctx, d_S, q->n, q->k, q->algo, d_SA, nullptr);
                          },
                          &a);
}

// ---- one-shot calls (the form the C++ facade KissHipSorter binds, kiss_hip_sorter.hpp) ------------------------------------
// A context for a chm13-size text is 77 GB of work arrays + 15.6 GB of text / SA copies: allocating and freeing that per
// call costs more than the sort.  The one-shot entry points keep ONE context per device between calls (grown when a longer
// text arrives, never shrunk) and hand it back through kiss_hip_release_cached_contexts().  One one-shot call at a time per
// device (a sort fills the GPU; the reference's static facade is not re-entrant either).
namespace {
struct CachedCtx {
    std::mutex m;
    kiss_hip_ctx *ctx = nullptr;
};
CachedCtx *cached_ctx(int device)
{
    static CachedCtx c[64];
    return &c[(unsigned)device & 63u];
}
// the device's cached context, grown to hold n bases (lock held by the caller)
int cached_ctx_for(CachedCtx &c, int device, uint64_t n, kiss_hip_ctx **out)
{
    if (c.ctx && c.ctx->max_n < n) {
        kiss_hip_ctx_destroy(c.ctx);
        c.ctx = nullptr;
    }
    if (!c.ctx) {
        int rc = kiss_hip_ctx_create(&c.ctx, device, n);
        if (rc) {
            c.ctx = nullptr;
            return rc;
        }
    }
    *out = c.ctx;
    return KISS_HIP_OK;
}
} // namespace

int kiss_hip_suffix_sort_dna_u32(const uint8_t *S, uint64_t n, uint32_t k, int algo, uint32_t *SA, int device)
{
    if (!SA || (n && !S)) return KISS_HIP_E_INVALID;
    if (n == 0) { // kiss1_core.hpp:237-238: no device work at all
        SA[0] = 0;
        return KISS_HIP_OK;
    }
    if (device < 0 || device >= 64) return KISS_HIP_E_NO_DEVICE;
    CachedCtx &c = *cached_ctx(device);
    std::lock_guard<std::mutex> lock(c.m);
    kiss_hip_ctx *ctx = nullptr;
    int rc = cached_ctx_for(c, device, n, &ctx);
    if (rc) return rc;
    rc = kiss_hip_ctx_suffix_sort_dna_u32(ctx, S, n, k, algo, SA);
    if (rc == KISS_HIP_E_NOMEM || rc == KISS_HIP_E_HIP) { // do not keep a context that may be half grown or on a faulted device
        kiss_hip_ctx_destroy(c.ctx);
        c.ctx = nullptr;
    }
    return rc;
}

int kiss_hip_release_cached_contexts(void)
{
    for (int d = 0; d < 64; d++) {
        CachedCtx &c = *cached_ctx(d);
        std::lock_guard<std::mutex> lock(c.m);
        if (c.ctx) {
            kiss_hip_ctx_destroy(c.ctx);
            c.ctx = nullptr;
        }
    }
    return KISS_HIP_OK;
}

/* ---- test hooks: the sorting / scanning primitives on caller data (host pointers) ---------------------- */
int kiss_hip_debug_radix_sort(kiss_hip_ctx *ctx, uint64_t *keys, uint32_t *pos, uint64_t count, int key_lo_bit)
{
    if (!ctx || !keys || !pos || count > ctx->m_cap) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    ctx->stream = ctx->own_stream;
    KCHECK(hipMemcpy(ctx->keyA, keys, count * 8, hipMemcpyHostToDevice));
    KCHECK(hipMemcpy(ctx->posA, pos, count * 4, hipMemcpyHostToDevice));
    RadixBufs rb;
    rb.key[0] = ctx->keyA;
    rb.key[1] = ctx->keyB;
    rb.pos[0] = ctx->posA;
    rb.pos[1] = ctx->posB;
    rb.seg[0] = rb.seg[1] = nullptr;
    int res = 0;
    KTRY(kiss_radix_sort(ctx, rb, count, key_lo_bit, 0, &res));
    KTRY(kiss_radix_check(ctx));
    ktimer_collect(ctx); // kernel-class times of this call show up in kiss_hip_get_stats (tools/radix_probe.py)
    KCHECK(hipMemcpy(keys, rb.key[res], count * 8, hipMemcpyDeviceToHost));
    KCHECK(hipMemcpy(pos, rb.pos[res], count * 4, hipMemcpyDeviceToHost));
    return KISS_HIP_OK;
}

int kiss_hip_debug_fail_alloc_over(kiss_hip_ctx *ctx, uint64_t bytes)
{
    if (!ctx) return KISS_HIP_E_INVALID;
#ifdef KISS_HIP_HOOKS
    ctx->fail_alloc_over = bytes;
    return KISS_HIP_OK;
#else
    (void)bytes;
    return KISS_HIP_E_UNSUPPORTED; // fault injection exists in the hooks build only (libkiss_hip_hooks.so)
#endif
}

/* 1 in the hooks build (libkiss_hip_hooks.so: environment switches, fault injection, tracing), 0 in the shipped one */
int kiss_hip_has_hooks(void)
{
#ifdef KISS_HIP_HOOKS
    return 1;
#else
    return 0;
#endif
}

int kiss_hip_debug_scan_u32(kiss_hip_ctx *ctx, uint32_t *data, uint64_t count)
{
    if (!ctx || !data || count > ctx->m_cap) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    ctx->stream = ctx->own_stream;
    KCHECK(hipMemcpy(ctx->posA, data, count * 4, hipMemcpyHostToDevice));
    KTRY(kiss_scan_u32(ctx, ctx->posA, ctx->posB, count));
    KCHECK(hipStreamSynchronize(ctx->stream));
    KCHECK(hipMemcpy(data, ctx->posB, count * 4, hipMemcpyDeviceToHost));
    return KISS_HIP_OK;
}

int kiss_hip_ctx_get_stage_outputs(kiss_hip_ctx *ctx, uint32_t *lms_ascending, uint32_t *lms_sorted, uint64_t *counts)
{
    if (!ctx) return KISS_HIP_E_INVALID;
    KCHECK(hipSetDevice(ctx->device));
    if (lms_ascending && ctx->m)
        KCHECK(hipMemcpy(lms_ascending, ctx->lms_pos, ctx->m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (lms_sorted && ctx->m) {
        ctx->stream = ctx->own_stream;
        KTRY(kiss_merge_lms(ctx)); // the sort itself never builds the merged list (place.hip)
        KCHECK(hipStreamSynchronize(ctx->stream));
    }
    if (lms_sorted && ctx->m)
        KCHECK(hipMemcpy(lms_sorted, ctx->lmsP, ctx->m * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (counts)
        for (int i = 0; i < 12; i++) counts[i] = ctx->counts[i];
    return KISS_HIP_OK;
}

} // extern "C"
