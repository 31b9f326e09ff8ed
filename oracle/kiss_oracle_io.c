/*
 * kiss_oracle_io.c -- CPU restatement (plain C) of the reference's input reader: which bytes of a FASTA or
 * plain-text file become bases, and their codes.
 *
 * TEST INFRASTRUCTURE ONLY (see kiss_oracle.c).  PARITY STATUS: "parity unpinned" -- the reference has no test
 * or fixture for its reader and can not be built here; every step cites the reference lines it restates and
 * follows their control flow (getline / peek) literally instead of a derived rule.
 */
#include <stdint.h>

/* Codec::ints (include/biovoltron/utility/istring.hpp:28-36) followed by `c % 4`
 * (include/command/suffix_sort.hpp:33): bytes >= 128 index the table out of range in the reference (undefined);
 * they are treated like every other non-ACGT byte here. */
static uint8_t ko_code(uint8_t c)
{
    switch (c) {
    case 'a': case 'A': return 0;
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'T': return 3;
    default: return 4 % 4;
    }
}

static int ko_isspace(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

/* std::getline: the line is [*pos, end of line), the '\n' is consumed; fails (returns 0) at end of file */
static int ko_getline(const uint8_t *raw, uint64_t bytes, uint64_t *pos, uint64_t *lo, uint64_t *hi)
{
    if (*pos >= bytes) return 0;
    *lo = *pos;
    uint64_t p = *pos;
    while (p < bytes && raw[p] != '\n') p++;
    *hi = p;
    *pos = p < bytes ? p + 1 : p;
    return 1;
}

/* read_sequence (include/utils/io.hpp:6-18); out must hold `bytes` codes; returns n */
uint64_t ko_read_sequence(const uint8_t *raw, uint64_t bytes, uint8_t *out)
{
    uint64_t n = 0, pos = 0, lo, hi;
    if (bytes > 0 && raw[0] == '>') { /* io.hpp:8 fasta mode: istream_view<FastaRecord<true>> */
        for (;;) {
            /* operator>> (biovoltron/file_io/fasta.hpp:119-123): skip white space, the record must start with '>' */
            while (pos < bytes && ko_isspace(raw[pos])) pos++;
            if (pos >= bytes || raw[pos] != '>') break;
            if (!ko_getline(raw, bytes, &pos, &lo, &hi)) break; /* :126 the header line (name not needed) */
            for (;;) {                                         /* :128 for (...; getline(is, line);) */
                if (!ko_getline(raw, bytes, &pos, &lo, &hi)) break;
                for (uint64_t i = lo; i < hi; i++) out[n++] = ko_code(raw[i]); /* :130 seq += to_istring(line) */
                if (pos < bytes && raw[pos] == '>') break;     /* :137 peek() == START_SYMBOL -> return */
            }
            /* :149 is.clear(): the record is delivered also when the loop ended at end of file */
        }
    } else { /* io.hpp:13-16 text mode */
        while (ko_getline(raw, bytes, &pos, &lo, &hi))
            for (uint64_t i = lo; i < hi; i++) out[n++] = ko_code(raw[i]);
    }
    return n;
}
