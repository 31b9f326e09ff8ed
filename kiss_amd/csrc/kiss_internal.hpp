// kiss_internal.hpp -- shared declarations of libkiss_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <mutex>
#include <vector>
#include "../../include/kiss_hip.h"

// ---- options ---------------------------------------------------------------------------------------------
// The shipped library (default build) reads the environment ONCE, in kiss_hip_ctx_create, and only for the three
// supported knobs below; no result path depends on a variable of the host application's environment.  Everything else --
// A-B switches, tuning sweeps, fault injection, tracing -- exists only in the hooks build (-DKISS_HIP_HOOKS,
// kiss_amd/libkiss_hip_hooks.so: what the tests that force the rare paths load), which re-reads the environment at
// the start of every API call (kiss_opts_refresh) so that a test can flip a switch between two calls on one context.
struct KissOpts {
    // supported (both builds, read once per context)
    bool debug = false;            // KISS_HIP_DEBUG: progress lines on stderr
    int xfer_threads = 0;          // KISS_HIP_XFER_THREADS: copy threads of the host-pointer entry points (0 = default)
    int prefault_threads = 0;      // KISS_HIP_PREFAULT_THREADS: helper threads that fault in a pageable destination (0 = default)
    // hooks build only (the defaults are the product's behaviour)
    bool sync_readback = false;    // KISS_HIP_SYNC_READBACK: counters by memcpy + stream synchronise instead of publish + spin
    bool no_lms_exact = false;     // KISS_HIP_NO_LMS_EXACT: exact order by doubling over SA (the round-2 form)
    bool heads_by_compare = false; // KISS_HIP_LMS_HEADS_BY_COMPARE: tie flags of the LMS-level doubling by comparison
    bool no_early_out = false;     // KISS_HIP_NO_EARLY_OUT
    bool no_pivot_rounds = false;  // KISS_HIP_NO_PIVOT_ROUNDS: 32-base rounds only
    bool pivot_from_round2 = false;// KISS_HIP_PIVOT_FROM_ROUND2
    bool pair_keys = false;        // KISS_HIP_PAIR_KEYS: gather the round's key for pairs as well
    bool no_fc0_onepass = false;   // KISS_HIP_NO_FC0_ONEPASS: count + scan + compact after round 0
    bool no_class_bytes = false;   // KISS_HIP_NO_CLASS_BYTES: the induction's count pass reads the context words (rounds 1-3)
    bool no_pivot_ctx = false;     // KISS_HIP_NO_PIVOT_CTX
    bool no_taint = false;         // KISS_HIP_NO_TAINT: the suffix-array form compares every neighbour pair
    bool isa_direct = false;       // KISS_HIP_ISA_DIRECT: inverse SA by plain random scatter
    bool no_onesweep = false;      // KISS_HIP_NO_ONESWEEP: histogram + offsets + scatter radix passes
    bool merge_lms = false;        // KISS_HIP_MERGE_LMS: the merged copy of the LMS list (round-1 form)
    bool no_small_alphabet = false;// KISS_HIP_NO_SMALL_ALPHABET (general.hip)
    bool induce_one_pass = false;  // KISS_HIP_INDUCE_ONE_PASS: a source segment partitioned in one pass with a look-back (slower: DESIGN.md 4)
    bool verify = false;           // KISS_HIP_VERIFY: check sums and per-bucket checks inside the induction
    bool no_prefault = false;      // KISS_HIP_NO_PREFAULT
    bool no_serialize = false;     // KISS_HIP_NO_SERIALIZE: no per-device lock around the device phase of a sort
    bool lock_launches = false;    // KISS_HIP_LOCK_LAUNCHES: every launch under one process-wide lock
    bool sync_launches = false;    // KISS_HIP_SYNC_LAUNCHES: the launching thread waits for its stream after every launch
    uint32_t doubling_h0 = 0;      // KISS_HIP_DOUBLING_H0 (0 = KISS_EXACT_H0)
    uint64_t tcap0 = 0;            // KISS_HIP_TCAP0: first reservation of the tied-segment arrays (0 = default)
    int pivot_slots = 3;           // KISS_HIP_PIVOT_SLOTS
    uint32_t small_seg = 0;        // KISS_HIP_SMALL_SEG (0 = LMS_SMALL_SEG)
    uint32_t near_merge_min = 4096;// KISS_HIP_NEAR_MERGE_MIN
    uint32_t induce_small_max = 0; // KISS_HIP_INDUCE_SMALL_MAX (0 = default)
    uint32_t collapse_cap = 0;     // KISS_HIP_COLLAPSE_CAP (0 = default)
    uint64_t collapse_n = 0;       // KISS_HIP_COLLAPSE_N (0 = default)
    uint32_t fm_heavy = 0, fm_light = 0; // KISS_HIP_FM_HEAVY / KISS_HIP_FM_LIGHT (0 = default)
    uint64_t isa_direct_max = 0;   // KISS_HIP_ISA_DIRECT_MAX (0 = default)
    unsigned lx_sync_points = 0;   // KISS_HIP_LX_SYNC_POINTS
    unsigned tie_trace = 0;        // KISS_HIP_TIE_TRACE: flight recorder of the near-end tie marks (place.hip)
    uint32_t poison = 0;           // KISS_HIP_POISON: fill every freshly allocated work array with this word (0 = off)
    char dump_pivot[256] = {0};    // KISS_HIP_DUMP_PIVOT: file name prefix
};
struct kiss_hip_ctx;
#ifdef KISS_HIP_HOOKS
void kiss_opts_refresh(kiss_hip_ctx *ctx); // api.hip: re-reads the environment (hooks build only)
// every kernel launch of the library under one process-wide lock / followed by a wait for its stream
struct KissLaunchGuard {
    bool held;
    KissLaunchGuard();
    ~KissLaunchGuard();
};
bool kiss_sync_launches();
#define KISS_ARG4_(a, b, c, d, ...) d
#ifdef hipLaunchKernelGGL
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                                                                            \
    do {                                                                                                               \
        KissLaunchGuard kiss_launch_guard_;                                                                            \
        hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);                                                         \
        if (kiss_sync_launches()) (void)hipStreamSynchronize(KISS_ARG4_(__VA_ARGS__, 0, 0, 0, 0));                     \
    } while (0)
#endif
#else
static inline void kiss_opts_refresh(kiss_hip_ctx *) {}
#endif

#define KISS_EMPTY_CTX 1u      // context word with no bases left (marker bit only)
#define KISS_CTX_BASES 15u     // bases carried in a 32-bit context word
// Bit 31 of a context word (15 bases + marker use bits 0..30): "taint".  Set by the LMS sort / placement on every LMS
// suffix whose place in the k-ordered list might not be its place in the exact order (tied through the depth with
// another one, or ranked by the near-end rule), inherited unchanged by everything induced from it.  The exact-order
// finish (kiss_exact_refine) then looks for tie groups only among tainted neighbours of the suffix array.
#define KISS_CTX_TAINT 0x80000000u
#define KISS_CTX_WORD(c) ((c) & 0x7FFFFFFFu)
#define KISS_STRIDE 125u       // KISS1_SPLIT_SORT_STRIDE_DNA, algo/sort/constant.hpp:29

// ---- error plumbing -----------------------------------------------------------
#define KCHECK(call)                                   \
    do {                                               \
        hipError_t e__ = (call);                       \
        if (e__ != hipSuccess) {                       \
            ctx->last_hip_error = (int)e__;            \
            return (e__ == hipErrorOutOfMemory) ? KISS_HIP_E_NOMEM : KISS_HIP_E_HIP; \
        }                                              \
    } while (0)
#define KTRY(expr)                 \
    do {                           \
        int s__ = (expr);          \
        if (s__ != KISS_HIP_OK) return s__; \
    } while (0)

// an internal invariant failed: say where (this is a bug report, not a user error)
#include <cstdio>
#define KINTERNAL() (fprintf(stderr, "[kiss_hip] internal check failed at %s:%d\n", __FILE__, __LINE__), KISS_HIP_E_INTERNAL)

// ---- device workspace -----------------------------------------------------------
struct kiss_hip_ctx {
    int device = 0;
    int last_hip_error = 0;
    KissOpts opts;
#ifdef KISS_HIP_HOOKS
    uint32_t *tie_dbg = nullptr; // place.hip: flight recorder of the near-end tie marks
    bool tie_dbg_on = false;
    uint32_t tie_dbg_E = 0, tie_dbg_k = 0;
#endif
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // stream of the current call
    uint64_t max_n = 0;
    uint64_t m_cap = 0;            // capacity of the per-LMS arrays (every LMS suffix of the text)
    uint64_t fail_alloc_over = 0;  // test hook (kiss_hip_debug_fail_alloc_over): see dmalloc
    uint64_t m_cap0 = 0;           // != 0: the caller's choice of the default reservation (kiss_hip_ctx_create_sized)
    uint64_t t_cap = 0;            // capacity of the tied-segment arrays (suffixes still tied after round 0)
    uint64_t flags_cap = 0;        // u64 entries in `flags`
    uint64_t tied_bytes = 0;
    uint64_t ws_bytes = 0;
    uint64_t lms_bytes = 0;

    // text
    uint64_t *pk = nullptr;        // 2-bit packed text, base i at bits [63-2(i%32), 62-2(i%32)] of word i/32
    uint64_t pk_words = 0;
    // classification scratch
    uint32_t *tile_gp = nullptr;   // per 256-word tile: bit0 = G, bit1 = P  -> later: carry-in of the tile
    uint32_t *tile_cnt = nullptr;  // per tile LMS count -> exclusive offsets
    uint8_t *CLS = nullptr;        // one class byte per word of CTX (induce.hip: cls_byte), allocated with it
    uint32_t *cl_part = nullptr;   // classification: one row of partial sums per workgroup (classify.hip: k_sum_rows)
    uint64_t n_tiles_cap = 0;
    uint32_t *d_counts = nullptr;  // 16 x u32: cnt[4], cntS[4], cntLMS[4], far_lms, spare
    // LMS arrays
    uint32_t *lms_pos = nullptr;   // ascending LMS positions (kept for stage output)
    bool lms_pos_complete = false; // ... all m of them (a whole-text classification on this ctx: not in the multi-device entry)
    uint64_t *keyA = nullptr, *keyB = nullptr;
    uint32_t *posA = nullptr, *posB = nullptr;
    uint32_t *segA = nullptr, *segB = nullptr;
    uint32_t *slotA = nullptr, *slotB = nullptr;
    uint32_t *segstartA = nullptr, *segstartB = nullptr; // first active index of every tied segment (+1 end entry)
    uint64_t *bkeyA = nullptr, *bkeyB = nullptr;         // radix path of big segments in rounds >= 1
    uint32_t *bposA = nullptr, *bposB = nullptr, *bsegA = nullptr, *bsegB = nullptr, *bslot = nullptr;
    uint64_t *flags = nullptr;     // per active item: (survivor << 32) | surviving-head, then its exclusive scan
    uint32_t *lms_sorted_far = nullptr; // far LMS suffixes in k-order
    uint32_t *lms_ctx_far = nullptr;    // parallel to lms_sorted_far: context word from the key payload, 0 = gather it
    uint32_t *lmsP = nullptr;      // all LMS suffixes in k-order (sentinel excluded)
    uint32_t *lmsC = nullptr;      // their context words
    // radix / scan scratch
    uint32_t *tile_hist = nullptr; // 256 x tiles (+1)
    uint64_t tile_hist_cap = 0;
    // one-sweep radix passes (radix.hip): look-back descriptors, digit histograms of all passes, [ticket, error]
    uint64_t *rx_desc = nullptr;
    uint32_t *rx_ghist = nullptr, *rx_ctl = nullptr;
    uint64_t rx_ghist_count = 0; // != 0: rx_ghist already holds the round-0 digit counts of that many keys (classify.hip)
    uint64_t rx_tiles_cap = 0, rx_epoch = 0;
    uint64_t *fc_desc = nullptr;  // round 0 flag + compact in one pass: one descriptor per 8192-item tile (+ ticket)
    uint64_t fc_desc_cap = 0;
    uint32_t rx_ticket_base = 0;
    uint64_t *scan_tmp = nullptr;  // block sums for scans
    uint64_t scan_tmp_cap = 0;
    // induce
    uint32_t *CTX = nullptr;       // context words parallel to SA (n+1)
    bool ctx_words_valid = false;  // CTX holds the words of the last kiss_induce on this ctx (taint bits for the exact finish)
    uint32_t *ind_counts = nullptr;// 4 x tiles + 1
    uint64_t ind_tiles_cap = 0;
    // one-pass partition (induce.hip: k_induce_onepass): look-back descriptors, 4 classes x ind_desc_stride tiles, tagged with
    // the pass epoch (cleared once); the tile ticket is rx_ctl[2]
    uint64_t *ind_desc = nullptr;
    uint64_t ind_desc_stride = 0, ind_epoch = 0;
    uint32_t ind_ticket_base = 0;
    uint32_t *d_small = nullptr;   // small scratch (64 u32) for single-workgroup kernels
    uint32_t *h_pinned = nullptr;  // 64 u32 pinned host scratch
    // low-latency read-back of a few words (api.hip: kiss_readback): a one-wave kernel stores them into this coherent
    // host buffer and then a sequence number; the host spins on the sequence number instead of synchronising the stream
    uint32_t *h_pub = nullptr;     // 32 u32: [0,16) data, [16] sequence number
    uint32_t *d_pub = nullptr;     // device-side address of h_pub
    uint32_t pub_seq = 0;
    int pub_mode = -1;             // -1 undecided, 0 = memcpy + stream synchronise, 1 = publish + spin
    // PREFIX_DOUBLING: (position, index) pairs of the binned inverse-suffix-array build (isa.hip), allocated on first use
    uint64_t *pairs1 = nullptr, *pairs2 = nullptr;
    uint64_t pairs_cap = 0;
    // scratch of kiss_hip_fmi_query_batch_dev, kept between calls (fm.hip)
    void *fm_pool[13] = {};
    uint64_t fm_pool_cap[13] = {};
    // near-end
    uint32_t *near_idx = nullptr, *near_fin = nullptr, *near_pos = nullptr, *near_tmp = nullptr, *near_tmp2 = nullptr; // place.hip: near_reserve
    uint64_t near_cap = 0;
    // The induction reads the sorted far list in place plus this table of the near-end suffixes (induce.hip: LmsRemap);
    // lmsP / lmsC are filled only on demand (kiss_merge_lms: stage outputs, KISS_HIP_MERGE_LMS=1)
    uint8_t *ga_codes = nullptr;     // general.hip: a byte text over <= 4 values, mapped to codes 0..3 (on first use)
    uint64_t ga_codes_cap = 0;
    uint8_t *refine_heads = nullptr; // exact-order finish: tie flags, one byte per SA entry (allocated on first use)
    // exact order through the LMS-level doubling: the LMS sort itself says which far suffix retired tied with its predecessor
    // (one byte per far-list slot, 1 = starts a group; inside CTX, which nothing else uses before the induction), and
    // kiss_merge_lms carries the bytes over to the merged list.  Null outside such a call.
    uint8_t *hfar = nullptr, *hmerged = nullptr;
    uint32_t h_depth = 0; // the bases the members of such a group share at least (= the order of the bounded phase)
    bool lms_merged = false;
    int near_form = 0;                    // 0: none, 1: pairwise ranks (text order), 2: merge-sorted
    const uint32_t *near_sorted = nullptr; // form 2: the near-end suffixes in k-order
    const uint32_t *rm_fin = nullptr, *rm_pos = nullptr;
    uint32_t rm_E = 0;

    // host-pointer entry points: device-side copies of the caller's S / SA (allocated on first use, api.hip) and the
    // page-locked bounce buffers + streams of the staged transfers (xfer.hip)
    uint8_t *io_S = nullptr;
    uint32_t *io_SA = nullptr;
    uint64_t io_cap = 0; // bases
    // early download (api.hip / induce.hip): while the sweeps still run, finished stretches of SA leave for a page-locked
    // host buffer on a copy stream of their own
    uint32_t *early_host_SA = nullptr; // non-null only inside a host-pointer call whose SA buffer is page-locked
    hipStream_t early_stream = nullptr;
    std::vector<hipEvent_t> early_events;
    size_t early_used = 0;
    uint64_t early_bytes = 0;
    void *xf_pin[16][2] = {};
    hipEvent_t xf_done[16][2] = {};
    hipStream_t xf_stream[16] = {};
    int xf_ready = 0; // copy threads whose streams / bounce buffers exist

    // state of the last call (for stage outputs / stats)
    uint64_t n = 0, m = 0, m_far = 0;
    uint64_t counts[12] = {0};
    kiss_hip_stats stats{};
    uint64_t profile_mask = 0; // bit i: kernel class i is timed with a pair of HIP events per launch
    struct Ev { hipEvent_t a, b; int cls; };
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
};

// RAII-less helper: times one kernel class when profiling is on
struct KTimer {
    kiss_hip_ctx *ctx;
    int idx;
    KTimer(kiss_hip_ctx *c, int cls, uint64_t items);
    ~KTimer();
};
void ktimer_collect(kiss_hip_ctx *ctx);

static inline uint64_t div_up(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// ---- stages (host drivers) -------------------------------------------------------
// (re)allocates every LMS-sized array for at least m_cap suffixes
int kiss_lms_reserve(kiss_hip_ctx *ctx, uint64_t m_cap, uint64_t t_cap_wanted = 0);
// (re)allocates the tied-segment arrays (seg*, slot*, segstart*, bkey*, bpos*, bseg*, bslot, flags) for t_cap items;
// their contents are lost
int kiss_tied_reserve(kiss_hip_ctx *ctx, uint64_t t_cap);
// re-reserves the default work arrays if an earlier call released them on running out of memory
int kiss_workspace_ready(kiss_hip_ctx *ctx);
// allocates ctx->CTX (max_n + 2 words) on first use: only the process that runs the induction / doubling phase holds it
int kiss_need_ctx_words(kiss_hip_ctx *ctx);
int kiss_pack_text(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n);
// classification: fills ctx->lms_pos, ctx->keyA (first 32 bases of each LMS), ctx->counts, ctx->m, ctx->m_far
// only LMS suffixes / histogram contributions of text positions in [win_lo, win_hi) are produced (sharded runs)
int kiss_classify(kiss_hip_ctx *ctx, uint64_t n, uint64_t depth /*0 = unbounded*/, uint64_t win_lo, uint64_t win_hi);
// copies nwords (<= 16) 32-bit words from device memory to ctx->h_pinned[0 .. nwords) after everything queued on the
// ctx stream so far has finished; returns when they are there (the per-round / per-pass control values of the drivers)
int kiss_readback(kiss_hip_ctx *ctx, const void *d_src, uint32_t nwords);
// the lock that keeps the device phases of two sorts on one device apart (api.hip: sort_dev)
std::mutex &kiss_device_mutex(int device);
// exclusive scans (in place allowed: out may equal in)
int kiss_scan_u32(kiss_hip_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t count);
int kiss_scan_u64(kiss_hip_ctx *ctx, const uint64_t *in, uint64_t *out, uint64_t count);
// zero count_u32 32-bit words at p with a kernel on the ctx stream
int kiss_zero_u32(kiss_hip_ctx *ctx, void *p, uint64_t count_u32);
// fill with any 32-bit value, also as a kernel (hipMemsetAsync is not reliably ordered against the kernels around it when
// several contexts work at once: found the hard way, twice).
int kiss_fill_u32(kiss_hip_ctx *ctx, void *p, uint32_t value, uint64_t count_u32);
// stable LSD radix sort of (key64[, seg32], pos32) tuples; bits [key_lo_bit,64) of key then seg_bits of seg.
// Buffers ping-pong; on return *in_is_result tells which pair holds the sorted data (true = the A buffers).
struct RadixBufs {
    uint64_t *key[2];
    uint32_t *seg[2]; // may be null when seg_bits == 0
    uint32_t *pos[2];
    const uint32_t *first_pos = nullptr; // optional: the first pass reads its positions from here instead of pos[0]
};
int kiss_radix_sort(kiss_hip_ctx *ctx, RadixBufs &b, uint64_t count, int key_lo_bit, int seg_bits, int *result_idx);
int kiss_radix_check(kiss_hip_ctx *ctx); // synchronises; KISS_HIP_E_INTERNAL if a look-back wait ran out
// stages.hip: pieces of the sharded form shared with multi.hip (all queued on ctx->stream, not synchronised)
uint64_t kiss_depth_of(uint64_t n, uint32_t k);
int kiss_key_hist(kiss_hip_ctx *ctx, const uint64_t *d_keys, uint64_t count, int bits, uint64_t *d_hist);
int kiss_partition_by_splitters(kiss_hip_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_pos, uint64_t count, int bits,
                                const uint32_t *splitters, int groups, uint64_t *d_keys_out, uint32_t *d_pos_out);
// k-ordered LMS sort of the far suffixes -> ctx->lms_sorted_far
int kiss_lms_sort(kiss_hip_ctx *ctx, uint64_t n, uint32_t k, uint64_t depth);
// kiss_lms_sort in exact mode gives up (no output) once suffixes still tie after this many bases
constexpr uint64_t KISS_EXACT_MSD_MAX_DEPTH = 32768;
constexpr int KISS_INTERNAL_TOO_DEEP = 1000; // never crosses the ABI
// Order of the bounded phase in front of the rank doubling (PREFIX_DOUBLING at k >= n, and the fall-back of exact order on
// very deep ties).  Round 3 sweep at chm13 size, whole exact sort: h0 = 128 / 256 / 512 / 1024 -> 224 / 211 / 205 / 213 ms
// (a deeper bounded phase costs pivot-round work, a shallower one leaves more tied suffixes to the doubling rounds).
constexpr uint32_t KISS_EXACT_H0 = 512;
// exact order from an h0-ordered SA by rank doubling over the full suffix array (lms_sort.hip)
int kiss_exact_refine(kiss_hip_ctx *ctx, uint64_t n, uint32_t h0, uint32_t *d_SA, uint8_t *heads_in = nullptr);
// isa[SA[i]] = i for a permutation SA of [0, total) (isa.hip)
int kiss_isa_build(kiss_hip_ctx *ctx, const uint32_t *SA, uint64_t total, uint32_t *isa);
// rank[L[i] >> 1] = i for the m LMS positions of L (lms_asc: the same positions ascending); scratch from the caller (isa.hip)
int kiss_rank_build_lms(kiss_hip_ctx *ctx, const uint32_t *L, const uint32_t *lms_asc, uint64_t m, uint64_t n,
                        uint32_t *rank, uint64_t *pairs1, uint64_t *pairs2, uint32_t *small, uint64_t small_words);
// Exact order of the LMS suffixes BEFORE the induction (lms_sort.hip): the merged h0-ordered list ctx->lmsP / ctx->lmsC is
// refined by rank doubling over the LMS suffixes alone.  `scratch`: the (n + 1)-word suffix array buffer (not yet written).
// *resolved = false: the list is as kiss_merge_lms left it (still h0-ordered, taint bits intact) and kiss_exact_refine has
// to finish the job on the suffix array.
int kiss_lms_exact_refine(kiss_hip_ctx *ctx, uint64_t n, uint32_t h0, uint32_t *scratch, bool *resolved);
// fills ctx->lmsP / ctx->lmsC from the far list and the near-end ranks of the last kiss_place_lms
int kiss_merge_lms(kiss_hip_ctx *ctx);
#ifdef KISS_HIP_HOOKS
void kiss_tie_trace_report(kiss_hip_ctx *ctx); // place.hip
#endif
// near-end ranking, merge, context gather -> ctx->lmsP / ctx->lmsC
int kiss_place_lms(kiss_hip_ctx *ctx, uint64_t n, uint32_t k, uint64_t depth);
// host <-> device legs of the host-pointer entry points (xfer.hip); both return after the bytes have arrived
int kiss_xfer_h2d(kiss_hip_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int kiss_xfer_d2h(kiss_hip_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
void kiss_xfer_free(kiss_hip_ctx *ctx);
// induced sort sweeps -> d_SA
int kiss_induce(kiss_hip_ctx *ctx, uint64_t n, uint32_t *d_SA);
// SA[lo, hi) is final: if an early download is armed (ctx->early_host_SA), send it off on the copy stream now
int kiss_early_out(kiss_hip_ctx *ctx, const uint32_t *d_SA, uint64_t lo, uint64_t hi);

// ---- device helpers ----------------------------------------------------------------
#ifdef __HIPCC__
// ---- loads from the packed text: NATURALLY ALIGNED, always (round 4) -------------------------------------
// A key of 32 bases at an arbitrary position needs two consecutive words; written as one 16-byte load at an 8-byte aligned
// address (rounds 1-3; hipcc also merges `pk[w], pk[w + 1]` into exactly that) the compare loop of k_near_tie_runs returned
// wrong answers for a whole wave whenever another stream kept the memory system busy -- 17.6 % of the launches of
// tools/repro/near_tie_runs_glitch.hip, none with the loads below (same kernel, same noise).  The rule of this code base since:
// a vector load of the packed text starts at a multiple of its own size.  Two words = the 16-byte aligned pair that holds word
// w + the word behind that pair (24 bytes instead of 16; pk carries >= 7 spare zero words behind the text).
__device__ __forceinline__ void kiss_words2(const uint64_t *__restrict__ pk, uint64_t w, uint64_t &a, uint64_t &b)
{
    const uint64_t w0 = w & ~1ull;
    const ulonglong2 ab = *reinterpret_cast<const ulonglong2 *>(__builtin_assume_aligned(pk + w0, 16));
    const uint64_t c = pk[w0 + 2];
    const bool odd = (w & 1ull) != 0;
    a = odd ? ab.y : ab.x;
    b = odd ? c : ab.y;
}
// five consecutive words from word w on (four 32-base keys at any base offset): three aligned 16-byte loads
__device__ __forceinline__ void kiss_words5(const uint64_t *__restrict__ pk, uint64_t w, uint64_t x[5])
{
    const uint64_t w0 = w & ~1ull;
    const ulonglong2 p0 = *reinterpret_cast<const ulonglong2 *>(__builtin_assume_aligned(pk + w0, 16));
    const ulonglong2 p1 = *reinterpret_cast<const ulonglong2 *>(__builtin_assume_aligned(pk + w0 + 2, 16));
    const ulonglong2 p2 = *reinterpret_cast<const ulonglong2 *>(__builtin_assume_aligned(pk + w0 + 4, 16));
    const bool odd = (w & 1ull) != 0;
    x[0] = odd ? p0.y : p0.x;
    x[1] = odd ? p1.x : p0.y;
    x[2] = odd ? p1.y : p1.x;
    x[3] = odd ? p2.x : p1.y;
    x[4] = odd ? p2.y : p2.x;
}
// 32 bases starting at base index p (zero = 'A' padding past the end)
__device__ __forceinline__ uint64_t kiss_key32(const uint64_t *__restrict__ pk, uint64_t p)
{
    const uint32_t s = (uint32_t)(p & 31u) * 2u;
    uint64_t a, b;
    kiss_words2(pk, p >> 5, a, b);
    return (a << s) | ((b >> 1) >> (63u - s)); // branch-free: s == 0 gives a
}
__device__ __forceinline__ uint32_t kiss_base(const uint64_t *__restrict__ pk, uint64_t p)
{
    return (uint32_t)(pk[p >> 5] >> (62u - 2u * (uint32_t)(p & 31u))) & 3u;
}
// context word of position v: bases v-1 (bits 1:0), v-2 (bits 3:2), ... up to nb bases, marker bit above
__device__ __forceinline__ uint32_t kiss_load_ctx_n(const uint64_t *__restrict__ pk, uint64_t v, uint32_t nb)
{
    if (v >= nb) {
        uint64_t k = kiss_key32(pk, v - nb);
        return (uint32_t)(k >> (64u - 2u * nb)) | (1u << (2u * nb));
    }
    if (v == 0) return KISS_EMPTY_CTX;
    uint64_t w0 = pk[0];
    return (uint32_t)(w0 >> (64u - 2u * (uint32_t)v)) | (1u << (2u * (uint32_t)v));
}
__device__ __forceinline__ uint32_t kiss_load_ctx(const uint64_t *__restrict__ pk, uint64_t v)
{
    return kiss_load_ctx_n(pk, v, KISS_CTX_BASES);
}
// Round 0 of the LMS sort orders by the first 20 bases = key bits [24, 64).  The low 24 bits of the key emitted by
// the classification carry a short context word instead (11 bases + marker): a suffix that round 0 already makes
// unique (82 % on genome-like text) gets its context word without the random text gather of the placement step.
constexpr uint32_t KISS_KEY_CTX_BASES = 11;
constexpr uint64_t KISS_KEY_CTX_MASK = 0xFFFFFFull;
// round 0 of the LMS sort: 20 bases = key bits 24..63 = five 8-bit radix passes
constexpr int KISS_R0_SHIFT = 24;
constexpr int KISS_R0_PASSES = 5;
// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for this wave's
// outstanding global stores (vmcnt), so a store burst overlaps the next LDS staging step
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt()
{
    return (1ull << (threadIdx.x & 63u)) - 1ull;
}
#endif
