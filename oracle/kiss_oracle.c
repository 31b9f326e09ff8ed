/*
 * kiss_oracle.c -- CPU restatement (plain C) of the jhhung/kISS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (kiss_amd/, the
 * libkiss_hip.so C-ABI) may include, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS: pinned against the reference's own compiled code for get_lms,
 * the packed-text loads, put_lms_suffix and the induction sweeps; NOT pinned by
 * reference code for the control flow of the LMS sort (kiss1_core.hpp:41-144), for
 * KISS2 and for the FM-index ("parity unpinned" for those three).  Detail:
 *   - kiss1_core.hpp, kiss2_core.hpp and fm_index.hpp include <spdlog/spdlog.h>, an
 *     un-vendored submodule absent from /root/reference/submodules/spdlog and from
 *     this image; they are unbuildable here and no stand-in header is written.
 *   - kiss_common.hpp, structs.hpp, utils.hpp, constant.hpp and xbit_vector.hpp
 *     need nothing the image lacks: oracle/ref_driver.cpp compiles them where they
 *     lie into oracle/_ref/libkiss_ref.so, and tests/test_ref_pin.py +
 *     tests/golden/ref_pins.json (made by tests/golden/make_ref_golden.py) check
 *     this file against them: LMS list + 5x256 histograms (1/3/8 threads), 10-mer
 *     and 125-base loads, and the SA produced by the reference's put_lms_suffix +
 *     induced_sort from this file's LMS order -- bit for bit.
 *   - the comparator (kiss1_core.hpp:94-135) is restated twice, independently: here
 *     with byte compares, in ref_driver.cpp with AVX2 block compares on the
 *     reference's PackedDNAString loads under libstdc++'s std::sort; the two agree
 *     on every pinned shape.  The reference's own tests (tests/kiss.cpp:26-28) hold
 *     only the k-order property; test_oracle.py checks it, and for k = 2^32-1
 *     equality with a naive sorter (the exact SA is unique).
 *
 * All citations are relative to /root/reference/include/biovoltron/.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KO_EMPTY 0xFFFFFFFFu /* algo/sort/constant.hpp:19-20 */
#define KO_PREFIX 10         /* KISS1_SPLIT_SORT_PREFIX_SIZE, constant.hpp:27 */
#define KO_NBUCKET (1u << 20)
#define KO_STRIDE 125u /* KISS1_SPLIT_SORT_STRIDE_DNA, constant.hpp:29 */

/* ------------------------------------------------------------------------- */
/* get_lms : algo/sort/kiss_common.hpp:483-579                                 */
/* Types (Appendix A.1 of SURVEY.md): type(n-1)=L; S if S[i]<S[i+1]; L if >;   */
/* else type(i+1).  LMS(i) <=> i>=1, type(i)=S, type(i-1)=L.                   */
/* Output: ascending LMS positions, then the sentinel n appended (:578);       */
/* returns m (count incl. sentinel).  hist[5*256] summed over threads:         */
/* hist[t*256+c] for pair type t=2*type(i)+type(i-1) (i>=1) (:516-519) and      */
/* hist[4*256+c] = #positions with char c (:531,536).                          */
/* ------------------------------------------------------------------------- */
uint32_t ko_get_lms(const uint8_t *S, uint32_t n, uint32_t *lms, uint32_t *hist /* 5*256 or NULL */)
{
    uint32_t m = 0;
    uint32_t local_hist[5 * 256];
    memset(local_hist, 0, sizeof local_hist);
    if (n == 0) {
        lms[0] = 0;
        if (hist) memcpy(hist, local_hist, sizeof local_hist);
        return 1;
    }
    /* backward scan; `type` keeps (type(i+1), type(i)) in its low bits like :512 */
    /* first collect descending, then reverse (the reference reverses per thread, :569-572) */
    uint32_t type = 0; /* L_TYPE for position n-1 */
    uint8_t c0 = S[n - 1];
    for (uint32_t i = n - 1; i >= 1; i--) {
        local_hist[4 * 256 + c0]++;
        uint8_t c1 = S[i - 1];
        /* type of i-1 : c1<c0 -> S(1); c1>c0 -> L(0); equal -> type(i) */
        type = ((type << 1) + (c1 == c0 ? (type & 1u) : (uint32_t)(c1 < c0))) & 3u;
        /* type bits now: bit1 = type(i), bit0 = type(i-1) */
        local_hist[type * 256 + c0]++;
        if (type == 2u) lms[m++] = i; /* LMS_TYPE = 0b10 : i is S, i-1 is L */
        c0 = c1;
    }
    local_hist[4 * 256 + c0]++;
    /* reverse to ascending */
    for (uint32_t a = 0, b = m ? m - 1 : 0; a < b; a++, b--) {
        uint32_t t = lms[a];
        lms[a] = lms[b];
        lms[b] = t;
    }
    lms[m++] = n; /* sentinel, kiss_common.hpp:578 */
    if (hist) memcpy(hist, local_hist, sizeof local_hist);
    return m;
}

/* ------------------------------------------------------------------------- */
/* LMS comparator : algo/sort/kiss1_core.hpp:94-135 (the strict order cmp)      */
/* load_prefix_length_125 (algo/sort/structs.hpp:122-169) + the AVX2 byte       */
/* compare is "compare 125 consecutive bases lexicographically".               */
/* ------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *S;
    uint32_t n;
    uint32_t k;
} ko_cmp_ctx;

static int ko_less(uint32_t i, uint32_t j, const ko_cmp_ctx *c)
{
    const uint8_t *S = c->S;
    const uint64_t n = c->n, k = c->k;
    uint64_t sl = 0, a = i, b = j;
    while (sl <= k && a + KO_STRIDE <= n && b + KO_STRIDE <= n) {
        int r = memcmp(S + a, S + b, KO_STRIDE); /* bytes are 0..3: memcmp == lexicographic */
        if (r != 0) return r < 0;
        sl += KO_STRIDE;
        a += KO_STRIDE;
        b += KO_STRIDE;
    }
    while (sl + 1 <= k && a < n && b < n) {
        if (S[a] != S[b]) return S[a] < S[b];
        sl++;
        a++;
        b++;
    }
    if (sl >= k) return a < b;
    return a == n;
}

static int ko_qsort_cmp(const void *pa, const void *pb, void *ctx)
{
    uint32_t a = *(const uint32_t *)pa, b = *(const uint32_t *)pb;
    if (a == b) return 0;
    return ko_less(a, b, (const ko_cmp_ctx *)ctx) ? -1 : 1;
}

/* 10-mer prefix with zero ('A') padding past the end:
 * structs.hpp:94-96 (>=16 zero dibits of padding), :175-184 */
static inline uint32_t ko_prefix10(const uint8_t *S, uint32_t n, uint32_t p)
{
    uint32_t v = 0;
    for (uint32_t t = 0; t < KO_PREFIX; t++) {
        uint64_t q = (uint64_t)p + t;
        v = (v << 2) | (q < n ? (uint32_t)(S[q] & 3u) : 0u);
    }
    return v;
}

/* ------------------------------------------------------------------------- */
/* lms_suffix_direct_sort_dna : kiss1_core.hpp:24-145                            */
/* in : lms[0..m) = ascending LMS positions followed by the sentinel n           */
/* out: lms[0..m) sorted by (10-mer bucket, cmp); stable bucketing (:72-83)      */
/* tmp: scratch of m entries                                                    */
/* ------------------------------------------------------------------------- */
void ko_lms_sort(const uint8_t *S, uint32_t n, uint32_t k, uint32_t *lms, uint32_t m, uint32_t *tmp)
{
    uint32_t *start = (uint32_t *)calloc((size_t)KO_NBUCKET + 1, sizeof(uint32_t));
    uint32_t *pref = (uint32_t *)malloc((size_t)m * sizeof(uint32_t));
    for (uint32_t i = 0; i < m; i++) {
        pref[i] = ko_prefix10(S, n, lms[i]); /* sentinel n -> all padding -> bucket 0 */
        start[pref[i] + 1]++;
    }
    for (uint32_t b = 0; b < KO_NBUCKET; b++) start[b + 1] += start[b];
    {
        uint32_t *fill = (uint32_t *)malloc((size_t)KO_NBUCKET * sizeof(uint32_t));
        memcpy(fill, start, (size_t)KO_NBUCKET * sizeof(uint32_t));
        for (uint32_t i = 0; i < m; i++) tmp[fill[pref[i]]++] = lms[i];
        free(fill);
    }
    ko_cmp_ctx ctx = {S, n, k};
#pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t b = 0; b < KO_NBUCKET; b++) {
        uint32_t len = start[b + 1] - start[b];
        if (len > 1) qsort_r(tmp + start[b], len, sizeof(uint32_t), ko_qsort_cmp, &ctx);
    }
    memcpy(lms, tmp, (size_t)m * sizeof(uint32_t)); /* :144 */
    free(pref);
    free(start);
}

/* ------------------------------------------------------------------------- */
/* put_lms_suffix : kiss_common.hpp:445-481                                     */
/* sorted = sorted LMS list incl. sentinel at sorted[0]; SA has n+1 slots.      */
/* ------------------------------------------------------------------------- */
static void ko_put_lms(const uint8_t *S, uint32_t n, uint32_t *SA, const uint32_t *sorted, uint32_t m,
                       const uint32_t *count /*[4]*/, const uint32_t *lms_count /*[4]*/)
{
    (void)S;
    uint64_t end[4], acc = 1;
    for (int c = 0; c < 4; c++) {
        acc += count[c];
        end[c] = acc; /* inclusive_scan with init 1 (:463) */
    }
    for (uint64_t i = 1; i <= n; i++) SA[i] = KO_EMPTY;
    SA[0] = sorted[0]; /* == n */
    uint32_t mm = m;   /* SA1 = sorted[1..m) */
    for (int c = 3; c >= 0; c--) {
        uint32_t num = lms_count[c];
        if (!num) continue;
        mm -= num;
        memcpy(&SA[end[c] - num], &sorted[mm], (size_t)num * sizeof(uint32_t));
    }
}

/* induced_L : kiss_common.hpp:226-246, 372-403 (sequential semantics) */
static void ko_induce_L(const uint8_t *S, uint32_t n, uint32_t *SA, const uint32_t *count)
{
    uint64_t head[4], acc = 1;
    for (int c = 0; c < 4; c++) {
        head[c] = acc; /* exclusive_scan with init 1 (:27-38) */
        acc += count[c];
    }
    for (uint64_t i = 0; i <= n; i++) {
        uint32_t v = SA[i];
        if (v == KO_EMPTY || v == 0) continue;
        uint32_t u = v - 1;
        uint8_t cv = (v == n) ? 0 : S[v];
        uint8_t cu = S[u];
        SA[head[cu]++] = (cu < cv) ? KO_EMPTY : u; /* :241 incl. the dummy EMPTY write */
    }
}

/* induced_clear : kiss_common.hpp:405-420 */
static void ko_induce_clear(uint32_t n, uint32_t *SA, const uint32_t *count, const uint32_t *lms_count)
{
    (void)n;
    uint64_t acc = 1;
    for (int c = 0; c < 4; c++) {
        acc += count[c];
        for (uint64_t i = acc - lms_count[c]; i < acc; i++) SA[i] = KO_EMPTY;
    }
}

/* induced_S : kiss_common.hpp:40-61, 192-224 (sequential semantics; slot 0 never visited) */
static void ko_induce_S(const uint8_t *S, uint32_t n, uint32_t *SA, const uint32_t *count)
{
    uint64_t tail[4], acc = 1;
    for (int c = 0; c < 4; c++) {
        acc += count[c];
        tail[c] = acc; /* inclusive_scan with init 1 (:14-25) */
    }
    for (uint64_t i = n; i >= 1; i--) {
        uint32_t v = SA[i];
        if (v == KO_EMPTY || v == 0) continue;
        uint32_t u = v - 1;
        uint8_t cu = S[u], cv = S[v];
        if (cu <= cv) SA[--tail[cu]] = u; /* :55-56 */
    }
}

/* ------------------------------------------------------------------------- */
/* kiss1_suffix_array_dna : kiss1_core.hpp:229-268                               */
/* S: n bytes in 0..3; SA: n+1 entries out.  Optionally exports the stage        */
/* intermediates for stage-level parity checks.                                  */
/* ------------------------------------------------------------------------- */
int ko_suffix_sort_dna(const uint8_t *S, uint32_t n, uint32_t k, uint32_t *SA,
                       uint32_t *lms_sorted_out /* NULL or n/2+2 entries */, uint32_t *m_out /* NULL ok */)
{
    if (n == 0) { /* :237-238 */
        SA[0] = 0;
        if (m_out) *m_out = 1;
        if (lms_sorted_out) lms_sorted_out[0] = 0;
        return 0;
    }
    size_t cap = (size_t)n / 2 + 2;
    uint32_t *lms = (uint32_t *)malloc(cap * sizeof(uint32_t));
    uint32_t *tmp = (uint32_t *)malloc(cap * sizeof(uint32_t));
    uint32_t hist[5 * 256];
    if (!lms || !tmp) return -1;
    uint32_t m = ko_get_lms(S, n, lms, hist);
    ko_lms_sort(S, n, k, lms, m, tmp);
    if (lms_sorted_out) memcpy(lms_sorted_out, lms, (size_t)m * sizeof(uint32_t));
    if (m_out) *m_out = m;
    uint32_t count[4], lms_count[4];
    for (int c = 0; c < 4; c++) {
        count[c] = hist[4 * 256 + c];
        lms_count[c] = hist[2 * 256 + c];
    }
    ko_put_lms(S, n, SA, lms, m, count, lms_count);
    ko_induce_L(S, n, SA, count);
    ko_induce_clear(n, SA, count, lms_count);
    ko_induce_S(S, n, SA, count);
    free(lms);
    free(tmp);
    return 0;
}

/* FNV-1a-64 over the little-endian bytes of a u32 array (fixture hashing) */
uint64_t ko_fnv1a64_u32(const uint32_t *a, uint64_t cnt)
{
    uint64_t h = 0xcbf29ce484222325ull;
    const uint8_t *p = (const uint8_t *)a;
    for (uint64_t i = 0; i < cnt * 4; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

int ko_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
