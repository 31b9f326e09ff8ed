/*
 * kiss_oracle_fm.c -- CPU restatement (plain C) of the FM-index part of the jhhung/kISS hot path:
 * FMIndex<SA_INTV=4, uint32_t, KISS1Sorter<uint32_t>>{.LOOKUP_LEN = 0}
 * (reference include/biovoltron/algo/align/exact_match/fm_index.hpp).
 *
 * TEST INFRASTRUCTURE ONLY (see kiss_oracle.c).  PARITY STATUS: "parity unpinned" -- the reference
 * holds no test or fixture for FMIndex at all and can not be built here; the
 * oracle is pinned by brute-force hit sets (tests/test_oracle.py) and every function cites the
 * reference lines it restates.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OCC1_INTV 256 /* fm_index.hpp:113 */
#define OCC2_INTV 16  /* fm_index.hpp:106,114 */
#define B_OCC_INTV 64 /* fm_index.hpp:143 */
#define SA_INTV 4     /* include/command/fmindex_build.hpp:27 */

typedef struct ko_fmi {
    uint64_t N;        /* SA length = n + 1 */
    uint32_t cnt[4];
    uint32_t pri;
    uint8_t *bwt;      /* ceil(N/4) bytes, dibit i at bits 2(i%4) of byte i/4 */
    uint32_t *occ1;    /* (N/256 + 1) x 4 */
    uint8_t *occ2;     /* (N/16 + 1) x 4 */
    uint32_t *sa;      /* (N + 3) / 4 sampled values */
    uint64_t *b;       /* ceil(N/64) words */
    uint32_t *b_occ;   /* N/64 + 1 */
} ko_fmi;

static inline uint32_t bwt_get(const ko_fmi *f, uint64_t i) { return (f->bwt[i >> 2] >> (2 * (i & 3))) & 3u; }

/* build(ref, ori_sa): fm_index.hpp:277-371, 390-451 */
ko_fmi *ko_fmi_build(const uint8_t *S, uint32_t n, const uint32_t *SA)
{
    ko_fmi *f = (ko_fmi *)calloc(1, sizeof(ko_fmi));
    uint64_t N = (uint64_t)n + 1;
    f->N = N;
    uint64_t n1 = N / OCC1_INTV + 1, n2 = N / OCC2_INTV + 1;
    f->occ1 = (uint32_t *)calloc(n1 * 4, sizeof(uint32_t));
    f->occ2 = (uint8_t *)calloc(n2 * 4, 1);
    f->bwt = (uint8_t *)calloc((N + 3) / 4, 1);
    f->b = (uint64_t *)calloc((N + 63) / 64, sizeof(uint64_t));
    f->b_occ = (uint32_t *)calloc(N / B_OCC_INTV + 1, sizeof(uint32_t));
    f->sa = (uint32_t *)calloc((N + SA_INTV - 1) / SA_INTV, sizeof(uint32_t));
    /* build_occ :277-308 */
    for (uint64_t beg = 0; beg < N; beg += OCC1_INTV) {
        uint32_t *cnt = &f->occ1[(beg / OCC1_INTV) * 4];
        for (uint64_t i = beg; i < beg + OCC1_INTV; i++) {
            if (i % OCC2_INTV == 0)
                for (int j = 0; j < 4; j++) f->occ2[(i / OCC2_INTV) * 4 + j] = (uint8_t)cnt[j];
            if (i == N) break;
            uint32_t v = SA[i];
            if (v != 0) cnt[S[v - 1] & 3]++;
        }
    }
    uint32_t tot[4] = {0, 0, 0, 0};
    for (uint64_t x = 0; x < n1; x++)
        for (int j = 0; j < 4; j++) {
            tot[j] += f->occ1[x * 4 + j];
            f->occ1[x * 4 + j] = tot[j] - f->occ1[x * 4 + j];
        }
    uint32_t sum = 1;
    for (int j = 0; j < 4; j++) {
        sum += tot[j];
        f->cnt[j] = sum - tot[j];
    }
    /* build_bwt :310-329 */
    for (uint64_t i = 0; i < N; i++) {
        uint32_t v = SA[i];
        uint32_t c = 0;
        if (v != 0) c = S[v - 1] & 3;
        else f->pri = (uint32_t)i;
        f->bwt[i >> 2] |= (uint8_t)(c << (2 * (i & 3)));
    }
    /* build_sa :331-371 */
    for (uint64_t i = 0; i < N; i++)
        if (SA[i] % SA_INTV == 0) {
            f->b[i >> 6] |= 1ull << (i & 63);
            f->b_occ[i / B_OCC_INTV]++;
        }
    uint32_t s = 0;
    for (uint64_t x = 0; x < N / B_OCC_INTV + 1; x++) {
        s += f->b_occ[x];
        f->b_occ[x] = s - f->b_occ[x];
    }
    uint64_t p = 0;
    for (uint64_t i = 0; i < N; i++)
        if (SA[i] % SA_INTV == 0) f->sa[p++] = SA[i];
    return f;
}

void ko_fmi_free(ko_fmi *f)
{
    if (!f) return;
    free(f->bwt);
    free(f->occ1);
    free(f->occ2);
    free(f->sa);
    free(f->b);
    free(f->b_occ);
    free(f);
}

/* accessors for the Python binding */
uint64_t ko_fmi_N(const ko_fmi *f) { return f->N; }
uint32_t ko_fmi_pri(const ko_fmi *f) { return f->pri; }
const uint32_t *ko_fmi_cnt(const ko_fmi *f) { return f->cnt; }
const uint8_t *ko_fmi_bwt(const ko_fmi *f) { return f->bwt; }
const uint32_t *ko_fmi_occ1(const ko_fmi *f) { return f->occ1; }
const uint8_t *ko_fmi_occ2(const ko_fmi *f) { return f->occ2; }
const uint32_t *ko_fmi_sa(const ko_fmi *f) { return f->sa; }
const uint64_t *ko_fmi_b(const ko_fmi *f) { return f->b; }
const uint32_t *ko_fmi_bocc(const ko_fmi *f) { return f->b_occ; }

/* compute_occ :166-182 */
static uint32_t ko_occ(const ko_fmi *f, uint32_t c, uint64_t i)
{
    uint64_t o1 = i / OCC1_INTV, o2 = i / OCC2_INTV;
    uint64_t beg = o2 * OCC2_INTV;
    uint32_t cnt = 0;
    int pass_pri = (c == 0 && beg <= f->pri && f->pri < i);
    for (; beg < i; beg++)
        if (bwt_get(f, beg) == c) cnt++;
    return f->occ1[o1 * 4 + c] + f->occ2[o2 * 4 + c] + cnt - (uint32_t)pass_pri;
}
static inline uint64_t ko_lf(const ko_fmi *f, uint32_t c, uint64_t i) { return (uint64_t)f->cnt[c] + ko_occ(f, c, i); }

/* compute_b_occ :189-208 */
static uint32_t ko_b_occ(const ko_fmi *f, uint64_t i)
{
    uint64_t w = i / B_OCC_INTV;
    uint64_t mask = (i & 63) ? ((1ull << (i & 63)) - 1) : 0;
    return f->b_occ[w] + (uint32_t)__builtin_popcountll((i & 63) ? (f->b[w] & mask) : 0ull);
}

/* get_range(seed) with LOOKUP_LEN = 0, stop_cnt = 0 : fm_index.hpp:553-584, 224-235 */
void ko_fmi_range(const ko_fmi *f, const uint8_t *pat, uint32_t L, uint32_t *beg_out, uint32_t *end_out)
{
    uint64_t beg = 0, end = f->N;
    uint32_t len = L;
    if (!(end == beg || len == 0)) {
        while (len > 0) {
            if (end - beg < 1) break;
            uint32_t c = pat[len - 1] & 3;
            beg = ko_lf(f, c, beg);
            end = ko_lf(f, c, end);
            len--;
        }
    }
    *beg_out = (uint32_t)beg;
    *end_out = (uint32_t)end;
}

/* get_offsets(beg, end) for SA_INTV != 1 : fm_index.hpp:453-501 (FIFO queue, same visiting order) */
uint64_t ko_fmi_offsets(const ko_fmi *f, uint32_t beg, uint32_t end, uint32_t *out, uint64_t cap)
{
    uint64_t want = (uint64_t)end - beg, got = 0;
    typedef struct { uint64_t b, e; int d; } ent;
    uint64_t qcap = 4 * want + 64, qh = 0, qt = 0;
    ent *q = (ent *)malloc(qcap * sizeof(ent));
    q[qt++] = (ent){beg, end, 0};
    while (qh < qt && got < want) {
        ent cur = q[qh++];
        uint32_t ob = ko_b_occ(f, cur.b), oe = ko_b_occ(f, cur.e);
        for (uint32_t i = ob; i < oe; i++) {
            if (got < cap) out[got] = f->sa[i] + (uint32_t)cur.d;
            got++;
        }
        int nd = cur.d + 1;
        if (nd == SA_INTV) continue;
        if (qt + 4 >= qcap) {
            qcap *= 2;
            q = (ent *)realloc(q, qcap * sizeof(ent));
        }
        if (cur.b + 1 == cur.e) {
            uint64_t nb = ko_lf(f, bwt_get(f, cur.b), cur.b);
            q[qt++] = (ent){nb, nb + 1, nd};
        } else {
            for (uint32_t c = 0; c < 4; c++) {
                uint64_t nb = ko_lf(f, c, cur.b), ne = ko_lf(f, c, cur.e);
                if (nb != ne) q[qt++] = (ent){nb, ne, nd};
            }
        }
    }
    free(q);
    return got;
}

/* the batch loop of fmindex_query_main (include/command/fmindex_query.hpp:79-95):
 * patterns: Q x L bytes 0..3; beg/end per pattern; returns total hits, checksum = sum of all positions */
void ko_fmi_query_batch(const ko_fmi *f, const uint8_t *patterns, uint32_t L, uint64_t Q, uint32_t *beg, uint32_t *end,
                        uint64_t *occ_total, uint64_t *checksum, uint32_t *offsets /* or NULL */,
                        uint64_t *offsets_index /* Q+1 or NULL */)
{
    uint64_t occ = 0, sum = 0, cap = 1024;
    uint32_t *tmp = (uint32_t *)malloc(cap * sizeof(uint32_t));
    for (uint64_t q = 0; q < Q; q++) {
        uint32_t b, e;
        ko_fmi_range(f, patterns + q * L, L, &b, &e);
        beg[q] = b;
        end[q] = e;
        uint64_t want = (uint64_t)e - b;
        if (want + 8 > cap) {
            cap = 2 * (want + 8);
            tmp = (uint32_t *)realloc(tmp, cap * sizeof(uint32_t));
        }
        uint64_t got = ko_fmi_offsets(f, b, e, tmp, cap);
        if (offsets_index) offsets_index[q] = occ;
        for (uint64_t i = 0; i < got; i++) {
            sum += tmp[i];
            if (offsets) offsets[occ + i] = tmp[i];
        }
        occ += got;
    }
    if (offsets_index) offsets_index[Q] = occ;
    *occ_total = occ;
    *checksum = sum;
    free(tmp);
}

/* Serializer::save layout (utility/archive/serializer.hpp:92-109): u64 count then raw bytes,
 * nothing at all when count == 0.  Returns bytes written (buf may be NULL to size). */
static uint64_t put(uint8_t *buf, uint64_t off, const void *p, uint64_t bytes)
{
    if (buf) memcpy(buf + off, p, bytes);
    return off + bytes;
}
static uint64_t put_vec(uint8_t *buf, uint64_t off, uint64_t count, const void *p, uint64_t bytes)
{
    if (count == 0) return off;
    off = put(buf, off, &count, 8);
    return put(buf, off, p, bytes);
}
uint64_t ko_fmi_serialize(const ko_fmi *f, uint8_t *buf)
{
    uint64_t N = f->N, off = 0;
    off = put(buf, off, f->cnt, 16);
    off = put(buf, off, &f->pri, 4);
    off = put_vec(buf, off, N, f->bwt, (N + 3) / 4);                             /* DibitVector: count = #dibits */
    off = put_vec(buf, off, N / OCC1_INTV + 1, f->occ1, (N / OCC1_INTV + 1) * 16);
    off = put_vec(buf, off, N / OCC2_INTV + 1, f->occ2, (N / OCC2_INTV + 1) * 4);
    off = put_vec(buf, off, (N + SA_INTV - 1) / SA_INTV, f->sa, ((N + SA_INTV - 1) / SA_INTV) * 4);
    uint32_t lookup[2] = {0, (uint32_t)N};                                      /* build_lookup with LOOKUP_LEN = 0 */
    off = put_vec(buf, off, 2, lookup, 8);
    off = put_vec(buf, off, N, f->b, ((N + 63) / 64) * 8);                        /* XbitVector<1, u64>: count = #bits */
    off = put_vec(buf, off, N / B_OCC_INTV + 1, f->b_occ, (N / B_OCC_INTV + 1) * 4);
    return off;
}
