/*
 * kiss_hip.h -- C ABI of libkiss_hip.so: the MI355X (gfx950) implementation of the
 * jhhung/kISS hot path (k-ordered induced suffix sorting of DNA + FM-index queries).
 *
 * The reference has no FFI: its seam is the compile-time C++ facade
 *   biovoltron::KISS1Sorter<size_type>::get_suffix_array_dna(S, k, num_threads)
 *     (include/biovoltron/algo/sort/kiss1_sorter.hpp:20-26) ->
 *   kiss::kiss1_suffix_array_dna<uint8_t,uint32_t>(S, SA, k, num_threads)
 *     (include/biovoltron/algo/sort/kiss1_core.hpp:229-268)
 * and, for queries,
 *   FMIndex<4,uint32_t,KISS1Sorter<uint32_t>>::get_range / get_offsets
 *     (include/biovoltron/algo/align/exact_match/fm_index.hpp:453-501,553-584).
 * The entry points below are what a cgo/ctypes/C++ binding for that path binds
 * (see INTEGRATION.md).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions: every function returns 0 (KISS_HIP_OK) or a negative kiss_hip_status.
 * No exceptions cross the ABI.  Host buffers are owned by the caller.  Device
 * workspace is owned by a kiss_hip_ctx.  A ctx is bound to one HIP device and must
 * not be used from two threads at once; distinct ctxs are independent.  Inside one process the device phases of
 * sorts on ONE device -- kiss_hip_*suffix_sort*, every kiss_hip_stage_* call, every phase of kiss_hip_multi_* -- queue up
 * behind a per-device lock (a sort fills the GPU, two at once gain nothing); uploads, downloads, verification and queries
 * run side by side.  The lock is per PROCESS: two processes may sort on one GPU at the same time, and so may the caller's own
 * kernels on other streams; the results are the same (what round 3 saw go wrong there was a 16-byte load at
 * an 8-byte aligned address in one kernel, gone since round 4; the near-end tie runs are verified before they are used,
 * kiss_hip_stats.tie_run_retries).
 */
#ifndef KISS_HIP_H
#define KISS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KISS_HIP_VERSION 103 /* 0.1.3: kiss_hip_get_stats_sized, kiss_hip_has_hooks, kiss_hip_release_cached_contexts; from here on
                              * kiss_hip_stats only ever grows at its END (0.1.2 put three fields in the middle: callers built
                              * against 0.1.0 must be rebuilt) */

typedef enum kiss_hip_status {
    KISS_HIP_OK = 0,
    KISS_HIP_E_INVALID = -1,     /* bad argument (null pointer, n too large, unknown algo) */
    KISS_HIP_E_NO_DEVICE = -2,   /* no HIP device / device index out of range */
    KISS_HIP_E_HIP = -3,         /* a HIP runtime call failed (see kiss_hip_last_hip_error) */
    KISS_HIP_E_NOMEM = -4,       /* device or host allocation failed */
    KISS_HIP_E_UNSUPPORTED = -5, /* valid request outside the implemented range (documented) */
    KISS_HIP_E_INTERNAL = -6,    /* an internal invariant failed (bug) */
    KISS_HIP_E_IO = -7,          /* a file could not be opened / read */
    KISS_HIP_E_DEEP = -8         /* stage_sort only: exact order (k >= n) asked of the 32-bases-per-round path on ties
                                    deeper than 32768 bases; sort with a bounded k (512) and finish with stage_refine_exact
                                    (the single-call entry points do this by themselves) */
} kiss_hip_status;

/* Sorting algorithm selector; mirrors kISS::SortingAlgorithm
 * (include/utils/constant.hpp, command/suffix_sort.hpp:38-48). */
#define KISS_HIP_ALGO_PARALLEL_SORTING 0 /* KISS1: k-ordered LMS sort (kiss1_core.hpp)   */
#define KISS_HIP_ALGO_PREFIX_DOUBLING 1  /* KISS2: k >= n -> exact SA by rank doubling; bounded k -> the (deterministic)
                                            k-ordered SA of PARALLEL_SORTING: the reference's own bounded-k KISS2 output
                                            depends on its thread count, only the k-order property is defined */

/* the reference's size_type is uint32_t and EMPTY = 0xFFFFFFFF, so n + 19 < 2^32
 * (algo/sort/structs.hpp:94, constant.hpp:19-20).  Here 4096 less: several kernels run one thread per suffix-array
 * entry, a HIP grid holds fewer than 2^32 threads per dimension, and n + 1 rounded up to a workgroup has to stay
 * below that (found by tools/stress_verify.py at n = 2^32 - 20: "invalid configuration argument") */
#define KISS_HIP_MAX_N 4294963200ull

typedef struct kiss_hip_ctx kiss_hip_ctx;

/* Per-call statistics (filled by kiss_hip_get_stats after a sort on that ctx). */
typedef struct kiss_hip_stats {
    uint64_t n;              /* text length */
    uint64_t m;              /* number of LMS suffixes (without the sentinel) */
    uint32_t k;              /* requested order */
    uint32_t depth;          /* effective comparison depth D (0 = unbounded) */
    uint32_t lms_rounds;     /* 32-base refinement rounds executed */
    uint32_t induce_passes;  /* stable-partition passes executed by the two sweeps */
    uint64_t near_end;       /* LMS suffixes ranked by the near-end rule */
    uint64_t sort_item_rounds; /* sum over rounds of active LMS items */
    uint64_t big_item_rounds;  /* of those, items that went through the radix path in rounds >= 1 */
    float ms_total;          /* device time of the whole call (HIP events) */
    float ms_pack;           /* 2-bit packing */
    float ms_classify;       /* get_lms: classification + LMS extraction */
    float ms_lms_sort;       /* k-ordered LMS sort (all rounds) */
    float ms_place;          /* near-end ranking + merge + context gather */
    float ms_induce;         /* L and S sweeps */
    /* live per-kernel-class timing, only when profiling is enabled on the ctx */
    float ms_kernel[16];
    uint64_t launches_kernel[16];
    uint64_t items_kernel[16]; /* units processed (items for radix/induce, bases for classify) */
    /* PREFIX_DOUBLING (exact order) only: */
    uint64_t refine_items;    /* suffixes still tied after the bounded-depth phase (depth = refine_depth) */
    uint32_t refine_depth;    /* bases the bounded phase ordered by (0: doubling phase not used) */
    uint32_t doubling_rounds; /* rank-doubling rounds executed */
    float ms_refine;          /* device time of the doubling phase (included in ms_total) */
    /* host-pointer entry points only: wall-clock time of the two PCIe legs (the reference's timed region,
     * command/suffix_sort.hpp:57-61, is host S -> host SA) */
    float ms_h2d;             /* host S -> device */
    float ms_d2h;             /* device SA -> host */
    uint32_t refine_form;     /* exact order, how it was reached: 0 = doubling phase not used, 1 = rank doubling over the LMS
                               * suffixes before the induction (KISS2's order of things, kiss2_core.hpp:835-886), 2 = rank doubling
                               * over the whole suffix array after it (the LMS form gave up or is switched off) */
    /* kiss_hip_fmi_query_batch_dev with KISS_HIP_K_FM_QUERY timed: the two halves of ms_kernel[KISS_HIP_K_FM_QUERY] */
    float ms_fm_range;        /* backward search (get_range) */
    float ms_fm_locate;       /* get_offsets */
    /* appended in 0.1.3 (fields are only ever appended from here on) */
    uint32_t tie_run_retries; /* near-end placement: searches of the tie runs that failed their verification and were
                               * repeated (0 on an undisturbed device; see DESIGN.md 4.2 when it is not) */
    uint32_t reserved_tail_;
} kiss_hip_stats;

/* kernel classes for ms_kernel[] / launches_kernel[] */
enum {
    KISS_HIP_K_PACK = 0,
    KISS_HIP_K_CLASSIFY = 1,
    KISS_HIP_K_RADIX_HIST = 2,
    KISS_HIP_K_RADIX_SCATTER = 3,
    KISS_HIP_K_SCAN = 4,
    KISS_HIP_K_KEYGATHER = 5,
    KISS_HIP_K_FLAG_COMPACT = 6,
    KISS_HIP_K_PLACE = 7,
    KISS_HIP_K_INDUCE_COUNT = 8,
    KISS_HIP_K_INDUCE_SCATTER = 9,
    KISS_HIP_K_INDUCE_SMALL = 10,
    KISS_HIP_K_FM_QUERY = 11,
    KISS_HIP_K_FM_BUILD = 12,
    KISS_HIP_K_SEGRANK = 13,
    KISS_HIP_K_GROUP_HEADS = 14, /* PREFIX_DOUBLING: tie detection over the bounded-depth SA */
    KISS_HIP_K_ISA = 15,         /* PREFIX_DOUBLING: inverse suffix array init / rank updates */
    KISS_HIP_K_NCLASSES = 16
};

int kiss_hip_version(void);
/* 0 for the shipped library.  1 for the hooks build of the same sources (libkiss_hip_hooks.so, -DKISS_HIP_HOOKS): test
 * infrastructure in which KISS_HIP_* environment variables switch A-B forms, tuning values, fault injection and tracing,
 * re-read at the start of every call.  The shipped library looks at the environment once per context, in
 * kiss_hip_ctx_create, and only for KISS_HIP_DEBUG (progress lines on stderr), KISS_HIP_XFER_THREADS and
 * KISS_HIP_PREFAULT_THREADS (host-side copy / page-fault helper threads): no variable changes a result path. */
int kiss_hip_has_hooks(void);
const char *kiss_hip_strerror(int status);
/* number of visible HIP devices (does not initialise a context) */
int kiss_hip_device_count(int *count);

/* ---- context: device workspace sized for texts up to max_n bases ------------- */
int kiss_hip_ctx_create(kiss_hip_ctx **out, int device, uint64_t max_n);
/* the same with the capacity of the per-LMS-suffix work arrays chosen by the caller instead of 0.32 max_n: a rank of a
 * sharded sort (ranks > 0 hold about 1/G of the LMS suffixes; the arrays regrow on demand).  0 = the default. */
int kiss_hip_ctx_create_sized(kiss_hip_ctx **out, int device, uint64_t max_n, uint64_t lms_capacity);
int kiss_hip_ctx_destroy(kiss_hip_ctx *ctx);
/* enable/disable per-kernel HIP-event timing (adds event overhead; off by default) */
int kiss_hip_ctx_set_profiling(kiss_hip_ctx *ctx, int enabled);
/* the same for chosen kernel classes only: bit i of class_mask = class KISS_HIP_K_* number i.  A pair of events
 * per launch costs a few microseconds of stream time; timing one class leaves the others back to back. */
int kiss_hip_ctx_set_profiling_mask(kiss_hip_ctx *ctx, uint64_t class_mask);
/* last hipError_t seen by this ctx (0 = hipSuccess) and its string */
int kiss_hip_last_hip_error(const kiss_hip_ctx *ctx, const char **msg);
int kiss_hip_get_stats(const kiss_hip_ctx *ctx, kiss_hip_stats *out);
/* the same for a caller built against an older (shorter) or newer (longer) kiss_hip_stats: copies min(bytes, sizeof the
 * library's struct) bytes and zero-fills the rest of the caller's -- fields are only appended (see KISS_HIP_VERSION) */
int kiss_hip_get_stats_sized(const kiss_hip_ctx *ctx, void *out, uint64_t bytes);
/* bytes of device workspace the ctx holds */
int kiss_hip_ctx_workspace_bytes(const kiss_hip_ctx *ctx, uint64_t *bytes);
/* the host-pointer entry points keep device-side copies of the caller's buffers between calls (5 bytes per base of
 * max_n, counted in workspace_bytes); this releases them (the next such call allocates them again) */
int kiss_hip_ctx_release_io_buffers(kiss_hip_ctx *ctx);

/* ---- suffix sorting --------------------------------------------------------- */
/*
 * Replaces KISS1Sorter<uint32_t>::get_suffix_array_dna(S, k, num_threads)
 * (kiss1_sorter.hpp:20-26) / KISS2Sorter (kiss2_sorter.hpp:20-26).
 *   S  : n bytes, each in 0..3 (A C G T), host memory.  Only the low 2 bits are used.
 *   k  : order; 0xFFFFFFFF (the CLI's -k -1) or any k >= n means the exact suffix array.
 *   SA : caller-allocated, n+1 entries; SA[0] = n (sentinel), SA[1..n] a permutation.
 * One-shot: uploads, sorts and downloads on a context the library keeps for `device` between one-shot calls (created by
 * the first call, grown when a longer text arrives; a chm13-size context is 80 GB of work arrays and the reference's
 * facade allocates its 5 bytes per base per call too, kiss1_core.hpp:243-257 -- at 26 bytes per base that is not free).
 * One one-shot call at a time per device; kiss_hip_release_cached_contexts() gives the memory back.
 * n == 0 yields SA = {0} (kiss1_core.hpp:237-238).
 */
int kiss_hip_suffix_sort_dna_u32(const uint8_t *S, uint64_t n, uint32_t k, int algo, uint32_t *SA, int device);
/* frees the contexts the one-shot calls keep (all devices); the next one-shot call creates one again */
int kiss_hip_release_cached_contexts(void);

/* Same, on an existing ctx with host buffers (upload + sort + download).  The device-side copies of S and SA belong
 * to the ctx (allocated on the first call, kept for the next ones).  Page-locked host buffers (hipHostMalloc /
 * hipHostRegister) travel at the PCIe rate, and for a bounded k the download of SA starts while the induction sweeps are
 * still running (finished stretches leave on a copy stream); pageable ones are moved by 8 threads through page-locked
 * bounce buffers of the ctx.  kiss_hip_get_stats reports the wall time of both legs (ms_h2d, ms_d2h). */
int kiss_hip_ctx_suffix_sort_dna_u32(kiss_hip_ctx *ctx, const uint8_t *S, uint64_t n, uint32_t k, int algo,
                                     uint32_t *SA);

/* Device-resident form: d_S (n bytes) and d_SA (n+1 u32) are DEVICE pointers on the
 * ctx's device.  stream is a hipStream_t (NULL = the ctx's own stream).  The call
 * returns after the work on `stream` has completed. */
int kiss_hip_ctx_suffix_sort_dna_u32_dev(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t k, int algo,
                                         uint32_t *d_SA, void *stream);

/* ---- several GPUs of one node driven by ONE process (`kiss_hip_opts{ngpus, device ids}`) ----
 * What a `suffix_sort_main`-shaped caller (reference include/command/suffix_sort.hpp:37-61) binds to use more than one
 * device: same arguments as kiss_hip_suffix_sort_dna_u32 plus the device list.  The LMS sort is sharded by key range
 * over the devices: devices[0] packs the text and the others pull it, every device classifies
 * a slice of the text, the LMS list is exchanged by direct peer copies over xGMI (every pair of devices its own link,
 * receiver keeps source order = text order), every device sorts its key range, devices[0] pulls the sorted pieces and
 * runs the induction (one dependency chain: it does not shard).  Host threads inside the call, one per device; no
 * fork, no exec.  A device may be listed more than once (two shares on one GPU: how the path is tested on a one-GPU
 * box).  ndev = 1 runs the same staged pipeline on one device and moves nothing. */
typedef struct kiss_hip_multi kiss_hip_multi;
typedef struct kiss_hip_multi_stats {
    uint64_t n, m;          /* text length, LMS suffixes */
    uint32_t ndev;
    uint32_t refine_depth;  /* != 0: exact order finished by rank doubling from this order on devices[0] */
    uint64_t piece[8];      /* far LMS suffixes sorted by each of the first 8 devices */
    /* wall-clock phases of the last call (barrier to barrier, i.e. the slowest device of each phase) */
    float ms_total, ms_pack, ms_classify, ms_partition, ms_exchange, ms_sort, ms_gather, ms_induce;
} kiss_hip_multi_stats;
int kiss_hip_multi_create(kiss_hip_multi **out, const int *devices, int ndev, uint64_t max_n);
int kiss_hip_multi_destroy(kiss_hip_multi *mc);
/* host S -> host SA (the reference's timed region); the SA is downloaded from devices[0] */
int kiss_hip_multi_suffix_sort_dna_u32(kiss_hip_multi *mc, const uint8_t *S, uint64_t n, uint32_t k, int algo, uint32_t *SA);
/* d_S (n bytes) and d_SA (n + 1 u32) are device pointers on devices[0] */
int kiss_hip_multi_suffix_sort_dna_u32_dev(kiss_hip_multi *mc, const uint8_t *d_S, uint64_t n, uint32_t k, int algo,
                                           uint32_t *d_SA);
int kiss_hip_multi_get_stats(const kiss_hip_multi *mc, kiss_hip_multi_stats *out);
/* the per-device context of share `rank` (statistics, profiling switches); owned by mc */
kiss_hip_ctx *kiss_hip_multi_ctx(kiss_hip_multi *mc, int rank);
/* one-shot: create, upload, sort, download, free */
int kiss_hip_suffix_sort_dna_u32_multi(const uint8_t *S, uint64_t n, uint32_t k, int algo, uint32_t *SA, const int *devices,
                                       int ndev);

/* ---- verification (device side; independent of the sort kernels: reads only the caller's text and SA) --------
 * Checks that d_SA (n+1 entries) is what the reference's own tests require of a k-ordered suffix array
 * (tests/kiss.cpp:26-28): SA[0] = n, a permutation of [0, n], and for every i >= 1
 *   S.substr(SA[i-1], k) <= S.substr(SA[i], k).
 * For k >= n the stronger linear-time proof of exactness is used instead (inverse SA; first character, then the rank
 * of the following suffix), which holds iff d_SA is THE suffix array.  Text bytes compare as unsigned values, so the
 * call serves both the DNA codes 0..3 and byte texts (kiss_hip_suffix_sort_u8).  `digest` is an order-sensitive
 * 64-bit sum that any host can recompute (kiss_hip_sa_digest_host).  Allocates its own scratch (n/8 bytes, plus
 * 4(n+1) bytes for k >= n) and frees it before returning; the ctx is only used for the device and the stream. */
typedef struct kiss_hip_verify_report {
    uint64_t n;
    uint32_t k;
    uint32_t exact;            /* 1: the k >= n proof was used */
    uint32_t ok;               /* 1: every check passed */
    uint32_t sa0_ok;           /* SA[0] == n */
    uint64_t out_of_range;     /* entries > n */
    uint64_t duplicates;       /* entries whose value occurred before */
    uint64_t order_violations; /* adjacent pairs in the wrong order */
    uint64_t first_violation;  /* smallest such index i (pair SA[i-1], SA[i]); meaningful if order_violations > 0 */
    uint64_t tied_pairs;       /* bounded k: adjacent pairs equal through k bases (their order is not constrained) */
    uint64_t digest;
    float ms;                  /* device time of the checks */
    uint32_t reserved_;
} kiss_hip_verify_report;
int kiss_hip_ctx_verify_sa_dev(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t k, const uint32_t *d_SA,
                               kiss_hip_verify_report *out, void *stream);
/* host helpers (no device): the digest above, and FNV-1a-64 over raw bytes (seed 0xcbf29ce484222325 to start, or the
 * previous return value to continue over the next chunk) */
uint64_t kiss_hip_sa_digest_host(const uint32_t *SA, uint64_t count);
uint64_t kiss_hip_fnv1a64_host(const void *data, uint64_t bytes, uint64_t seed);

/* Stage outputs of the LAST sort on this ctx, for stage-level parity tests
 * (get_lms: kiss_common.hpp:543-579; lms_suffix_direct_sort_dna: kiss1_core.hpp:24-145).
 *   lms_ascending : m entries (text order), may be NULL
 *   lms_sorted    : m entries (k-order, sentinel excluded), may be NULL
 *   counts        : 12 entries {count[c], s_type_count[c], lms_count[c]} c = A,C,G,T, may be NULL
 * Host pointers. */
int kiss_hip_ctx_get_stage_outputs(kiss_hip_ctx *ctx, uint32_t *lms_ascending, uint32_t *lms_sorted,
                                   uint64_t *counts);

/* ---- stage-level entry points for the sharded (one process per GPU) suffix sort: SURVEY.md section 8(e) --------
 * The exchange between ranks (histogram all-reduce, all-to-all of the LMS list, gather of the sorted pieces) is done
 * by the host with RCCL (torch.distributed); these calls do the arithmetic.  All data pointers are DEVICE pointers.
 *   stage_classify  : pack the whole text, emit the LMS suffixes of text positions [lo, hi) (ascending) with their
 *                     first 32-base key; counts13 = {count[c], s_count[c], lms_count[c] (c = A,C,G,T), far LMS count}
 *                     restricted to the window (sum over ranks = global)       (get_lms, kiss_common.hpp:543-579)
 *   stage_local_lms : sizes of that list (m_local, of which the first m_far_local are far) and a copy of it
 *   stage_key_hist  : histogram (u64[2^bits]) of the first `bits` key bits of count items
 *   stage_partition : stable partition by destination group g = #{splitters <= first `bits` key bits}
 *   stage_sort      : k-ordered sort of `count` far LMS suffixes (ascending position order inside equal keys on input)
 *                                                                        (lms_suffix_direct_sort_dna, kiss1_core.hpp:24-145)
 *   stage_induce    : near-end rule + placement + L/S induction from the concatenated sorted far list, the near-end
 *                     suffixes (ascending positions) and the global counts -> SA     (kiss1_core.hpp:259-267)        */
int kiss_hip_stage_classify(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t k, uint64_t lo, uint64_t hi,
                            uint64_t counts13[13], void *stream);
int kiss_hip_stage_local_lms(kiss_hip_ctx *ctx, uint64_t *d_keys_out /* may be NULL: sizes only */,
                             uint32_t *d_pos_out, uint64_t *m_local, uint64_t *m_far_local);
/* Zero-copy hand-over between the stages: device pointers of the ctx's own work arrays (capacity = entries each holds).
 * A caller that passes these very pointers to the stage calls gets no copy in and no copy out:
 *   LOCAL_KEYS / LOCAL_POS  : what stage_classify emitted (u64 / u32), input of stage_partition, destination of the
 *                             exchange (the receiver's list) and input of stage_sort
 *   PART_KEYS / PART_POS    : output of stage_partition = source of the exchange
 *   SORTED / SORTED_CTX     : output of stage_sort (u32 / u32) = source of the gather; on the rank that runs the
 *                             induction also its destination and the input of stage_induce
 * kiss_hip_stage_reserve: makes the arrays hold lms_capacity entries; their CONTENTS ARE LOST when it has to regrow
 * (call it before stage_classify, or classify again).  The pointers change then: ask for the views afterwards. */
enum {
    KISS_HIP_VIEW_LOCAL_KEYS = 0,
    KISS_HIP_VIEW_LOCAL_POS = 1,
    KISS_HIP_VIEW_PART_KEYS = 2,
    KISS_HIP_VIEW_PART_POS = 3,
    KISS_HIP_VIEW_SORTED = 4,
    KISS_HIP_VIEW_SORTED_CTX = 5
};
int kiss_hip_stage_view(kiss_hip_ctx *ctx, int which, void **d_ptr, uint64_t *capacity);
int kiss_hip_stage_reserve(kiss_hip_ctx *ctx, uint64_t lms_capacity);
int kiss_hip_stage_key_hist(kiss_hip_ctx *ctx, const uint64_t *d_keys, uint64_t count, int bits, uint64_t *d_hist,
                            void *stream);
int kiss_hip_stage_partition(kiss_hip_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_pos, uint64_t count, int bits,
                             const uint32_t *splitters, int groups, uint64_t *d_keys_out, uint32_t *d_pos_out,
                             void *stream);
/* d_ctx_out / d_far_ctx (both optional): the context words the sort derives from the key payload, parallel to the
 * sorted positions (0 = not available, the induction gathers it from the text); shipping them with the pieces
 * spares rank 0 a random text gather per LMS suffix. */
int kiss_hip_stage_sort(kiss_hip_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_pos, uint64_t count, uint64_t n,
                        uint32_t k, uint32_t *d_sorted_out, uint32_t *d_ctx_out, void *stream);
int kiss_hip_stage_induce(kiss_hip_ctx *ctx, uint64_t n, uint32_t k, const uint32_t *d_far_sorted,
                          const uint32_t *d_far_ctx, uint64_t m_far, const uint32_t *d_near_pos, uint64_t near_count,
                          const uint64_t counts12[12], uint32_t *d_SA, void *stream);
/* stage_induce_exact: stage_induce for a list the stages have ordered by h0 bases (k = h0), with the exact-order finish of
 * kiss2_suffix_array_dna in front of the induction: rank doubling over the LMS suffixes, then ONE induction
 * (kiss2_core.hpp:835-886).  *exact_out = 1: d_SA is the exact suffix array; 0: it is h0-ordered and stage_refine_exact
 * has to finish the job (texts whose LMS suffixes are further apart than any window: DESIGN.md 2.3). */
int kiss_hip_stage_induce_exact(kiss_hip_ctx *ctx, uint64_t n, uint32_t h0, const uint32_t *d_far_sorted,
                                const uint32_t *d_far_ctx, uint64_t m_far, const uint32_t *d_near_pos, uint64_t near_count,
                                const uint64_t counts12[12], uint32_t *d_SA, void *stream, int *exact_out);
/* stage_refine_exact: turns the h0-ordered suffix array of the text packed by stage_classify (h0 = 512, say: the output of the
 * stages run with k = h0) into the exact suffix array by rank doubling over the tied suffixes -- what
 * kiss2_suffix_array_dna's prefix doubling yields for k = -1 (kiss2_core.hpp:728-797, 835-886).  Needs n >= 4 h0 + 1024. */
int kiss_hip_stage_refine_exact(kiss_hip_ctx *ctx, uint64_t n, uint32_t h0, uint32_t *d_SA, void *stream);

/* Test hooks (used by tests/ only): the library's stable LSD radix sort on bits [key_lo_bit, 64) of keys with a
 * 32-bit payload, and its exclusive u32 scan, run on caller data in host memory (count <= ctx LMS capacity). */
int kiss_hip_debug_radix_sort(kiss_hip_ctx *ctx, uint64_t *keys, uint32_t *pos, uint64_t count, int key_lo_bit);
int kiss_hip_debug_scan_u32(kiss_hip_ctx *ctx, uint32_t *data, uint64_t count);
/* fault injection: from now on work-array allocations of this ctx above `bytes` fail with KISS_HIP_E_NOMEM (0 = off) */
int kiss_hip_debug_fail_alloc_over(kiss_hip_ctx *ctx, uint64_t bytes);
/* host only: the key-range rule of the multi-device sort (kiss_hip_multi_*) on a caller's histogram of `bins` entries:
 * groups - 1 splitters (group of a bin = number of splitters <= bin) and the resulting items per group */
int kiss_hip_debug_splitters(const uint64_t *hist, uint64_t bins, int groups, uint32_t *splitters_out,
                             uint64_t *group_counts_out);

/* ---- FM-index (biovoltron FMIndex<4,uint32_t,...>{LOOKUP_LEN=0}) --------------- */
/* Raw views of the arrays of the .fmi layout (fm_index.hpp:591-615).
 * For the *_dev call every pointer is a device pointer. */
typedef struct kiss_hip_fmi_view {
    uint64_t n_sa;        /* N = n + 1 */
    uint32_t cnt[4];      /* fm_index.hpp:296-307 */
    uint32_t pri;         /* SA index i with SA[i] == 0 (:324-325) */
    uint32_t sa_intv;     /* SA sampling interval (4) */
    const uint8_t *bwt;   /* ceil(N/4) bytes, dibit i at bits 2(i%4) of byte i/4 (:317-328) */
    const uint32_t *occ1; /* (N/256+1) x 4 (:280,283-301) */
    const uint8_t *occ2;  /* (N/16+1) x 4 (:281,286-287) */
    const uint32_t *sa;   /* sampled SA values, ceil(N/4) (:358-369) */
    const uint64_t *b;    /* bit i = (SA[i] % sa_intv == 0), ceil(N/64) words (:338-350) */
    const uint32_t *b_occ;/* N/64+1 (:339,352-356) */
} kiss_hip_fmi_view;

/*
 * Batched backward search + locate; replaces the per-pattern loop of
 * fmindex_query_main (include/command/fmindex_query.hpp:79-95):
 *   get_range(pattern) (fm_index.hpp:553-584) then get_offsets(beg,end) (:453-501).
 *   patterns : Q x L bytes in 0..3, row-major (device)
 *   beg,end  : Q entries each (device), the SA range per pattern
 *   hit_count_total, checksum : host pointers; sum of (end-beg) and sum of all hit positions
 *   offsets / offsets_index : optional device buffers; offsets_index has Q+1 entries
 *       (exclusive prefix of hit counts), offsets receives the hit positions of pattern q
 *       at [offsets_index[q], offsets_index[q+1]) in get_offsets order; pass NULL to skip.
 *   offsets_capacity : entries available in `offsets`
 */
int kiss_hip_fmi_query_batch_dev(kiss_hip_ctx *ctx, const kiss_hip_fmi_view *fmi, const uint8_t *patterns, uint32_t L,
                                 uint64_t Q, uint32_t *beg, uint32_t *end, uint64_t *hit_count_total,
                                 uint64_t *checksum, uint32_t *offsets, uint64_t *offsets_index,
                                 uint64_t offsets_capacity, void *stream);

/*
 * FM-index construction from a text and its suffix array, both device resident
 * (FMIndex::build(ref, ori_sa), fm_index.hpp:390-451).  Output arrays are device
 * buffers sized as in kiss_hip_fmi_view; cnt/pri are written to the host struct.
 */
int kiss_hip_fmi_build_dev(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, const uint32_t *d_SA, uint32_t sa_intv,
                           uint8_t *d_bwt, uint32_t *d_occ1, uint8_t *d_occ2, uint32_t *d_sa_sampled, uint64_t *d_b,
                           uint32_t *d_b_occ, uint32_t cnt_out[4], uint32_t *pri_out, void *stream);

/* Host-pointer forms of the two FM-index calls (create a ctx on `device`, upload, run, download, free): what a host
 * that does not link HIP binds -- the `kiss` CLI (kiss_amd/csrc/host/kiss_cli.cpp), cgo / ctypes callers.
 * kiss_hip_fmi_build_host: SA_or_null == NULL sorts with k = 32 first, like FMIndex::build(ref) (fm_index.hpp:379-387).
 * Array sizes for a text of n bases: kiss_hip_fmi_sizes_for (the .fmi layout of fm_index.hpp:591-615). */
typedef struct kiss_hip_fmi_sizes {
    uint64_t n_sa, bwt_bytes, occ1_entries, occ2_bytes, sa_entries, b_words, b_occ_entries;
} kiss_hip_fmi_sizes;
int kiss_hip_fmi_sizes_for(uint64_t n, kiss_hip_fmi_sizes *out);
int kiss_hip_fmi_build_host(const uint8_t *S, uint64_t n, const uint32_t *SA_or_null, uint8_t *bwt, uint32_t *occ1,
                            uint8_t *occ2, uint32_t *sa, uint64_t *b, uint32_t *b_occ, uint32_t cnt_out[4],
                            uint32_t *pri_out, int device);
int kiss_hip_fmi_query_batch_host(const kiss_hip_fmi_view *fmi, const uint8_t *patterns, uint32_t L, uint64_t Q,
                                  uint32_t *beg, uint32_t *end, uint64_t *hit_count_total, uint64_t *checksum,
                                  uint32_t *offsets, uint64_t *offsets_index, uint64_t offsets_capacity, int device);

/* ---- General alphabet (bytes): exact suffix array ---------------------------------
 * Replaces KISS1Sorter::get_suffix_array -> kiss1_suffix_array (kiss1_core.hpp:270-311), reachable only from the
 * reference's tests / experiments.  For that entry only the k-order property is defined (its comparator has no
 * index tie-break); the exact suffix array (shorter suffix first on a tie, SA[0] = n, n + 1 entries) satisfies it for
 * every k.  7-character keys + rank doubling over all suffixes; no induction.  n <= ctx max_n. */
int kiss_hip_suffix_sort_u8(const uint8_t *S, uint64_t n, uint32_t *SA, int device);
int kiss_hip_ctx_suffix_sort_u8_dev(kiss_hip_ctx *ctx, const uint8_t *d_S, uint64_t n, uint32_t *d_SA, void *stream);

/* ---- FASTA / plain-text input parsed on the device (replaces read_sequence, include/utils/io.hpp:6-18, and the
 * `c % 4` of command/suffix_sort.hpp:33; record rules of biovoltron/file_io/fasta.hpp:117-151) ------------------
 * The file is FASTA iff its first byte is '>'.  Header lines are dropped, every other byte except '\n' is a base:
 * A/a 0, C/c 1, G/g 2, T/t 3, anything else 0 (the reference maps it to 4 and reduces % 4; that includes '\r').
 * kiss_hip_ctx_parse_text_dev: raw file bytes already in device memory -> codes in d_S (capacity >= bytes), *n_out
 *   = number of bases.  The ctx must have been created with max_n >= bytes / 1000.
 * kiss_hip_ctx_load_text_file: streams the file through pinned buffers into device memory and parses it there;
 *   *d_S_out is a device buffer owned by the caller (kiss_hip_free_dev).  No base is touched on the host.
 * kiss_hip_alloc_dev / kiss_hip_copy_to_host / kiss_hip_free_dev: for hosts that do not link HIP (the CLI, ctypes);
 *   they act on the current device of the calling thread (the one the last ctx call selected). */
int kiss_hip_file_size(const char *path, uint64_t *bytes);
int kiss_hip_ctx_parse_text_dev(kiss_hip_ctx *ctx, const uint8_t *d_raw, uint64_t bytes, uint8_t *d_S, uint64_t *n_out,
                                void *stream);
int kiss_hip_ctx_load_text_file(kiss_hip_ctx *ctx, const char *path, uint8_t **d_S_out, uint64_t *n_out);
int kiss_hip_alloc_dev(void **d_out, uint64_t bytes);
int kiss_hip_copy_to_host(void *dst, const void *d_src, uint64_t bytes);
int kiss_hip_free_dev(void *p);

#ifdef __cplusplus
}
#endif
#endif /* KISS_HIP_H */
