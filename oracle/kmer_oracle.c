/*
 * kmer_oracle.c -- CPU restatement of the reference's k-mer hot path (see kmer_oracle.h).
 * TEST INFRASTRUCTURE ONLY: checker + timed CPU baseline; never linked into the product.
 *
 * Written from the behaviour of /root/reference/dna.c (cited per function), not copied:
 * PostgreSQL's palloc/ereport/fmgr plumbing is replaced by plain C buffers and error codes.
 */
#include "kmer_oracle.h"

#include <stdlib.h>
#include <string.h>

const char *orc_strerror(int code)
{
    switch (code) {
    case ORC_OK: return "ok";
    case ORC_ERR_INVALID_K: return "Invalid k value: must be between 1 and 32";           /* dna.c:773 */
    case ORC_ERR_QKMER_LEN_MISMATCH: return "Qkmer pattern and kmer lengths do not match"; /* dna.c:1107 */
    case ORC_ERR_PREFIX_TOO_LONG: return "Prefix length cannot exceed kmer length";       /* dna.c:855 */
    case ORC_ERR_DNA_EMPTY: return "DNA sequence cannot be empty";                        /* dna.c:161 */
    case ORC_ERR_DNA_BAD_CHAR: return "Invalid character in DNA sequence";                /* dna.c:166 */
    case ORC_ERR_KMER_EMPTY: return "K-mer sequence cannot be empty";                     /* dna.c:461 */
    case ORC_ERR_KMER_TOO_LONG: return "K-mer length cannot exceed 32 nucleotides";       /* dna.c:467 */
    case ORC_ERR_KMER_BAD_CHAR: return "Invalid character in K-mer sequence";             /* dna.c:473 */
    case ORC_ERR_QKMER_EMPTY: return "qkmer pattern cannot be empty";                     /* dna.c:878 */
    case ORC_ERR_QKMER_TOO_LONG: return "Qkmer pattern length cannot exceed 32 characters"; /* dna.c:884 */
    case ORC_ERR_QKMER_BAD_CHAR: return "Invalid character in qkmer pattern";             /* dna.c:894 */
    case ORC_ERR_NOMEM: return "out of memory";
    }
    return "unknown error";
}

/* code <-> letter tables: A=00 T=01 C=10 G=11 (dna.c:120-123, 144-147) */
static const char CODE2CHAR[4] = { 'A', 'T', 'C', 'G' };

/* ------------------------------------------------------------------ dna */

uint64_t orc_dna_num_words(uint64_t n_bases)
{
    return (n_bases * 2 + 63) / 64; /* dna.c:181-182 */
}

/* dna.c:159-171 (validation: non-empty, only A/T/C/G) then dna.c:114-128 (LSB-first packing:
 * base i -> word i/32, bit offset (2i mod 64)). */
int orc_dna_encode(const char *seq, uint64_t *words)
{
    if (seq == NULL || *seq == '\0')
        return ORC_ERR_DNA_EMPTY;
    for (const char *p = seq; *p; p++)
        if (*p != 'A' && *p != 'T' && *p != 'C' && *p != 'G')
            return ORC_ERR_DNA_BAD_CHAR;
    uint64_t n = (uint64_t)strlen(seq);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t off = (i * 2) % 64, idx = i / 32;
        switch (seq[i]) {
        case 'A': break;
        case 'T': words[idx] |= (uint64_t)1 << off; break;
        case 'C': words[idx] |= (uint64_t)2 << off; break;
        case 'G': words[idx] |= (uint64_t)3 << off; break;
        }
    }
    return ORC_OK;
}

/* dna.c:135-152 */
void orc_dna_decode(const uint64_t *words, uint64_t n_bases, char *out)
{
    for (uint64_t i = 0; i < n_bases; i++)
        out[i] = CODE2CHAR[(words[i / 32] >> ((i * 2) % 64)) & 3];
    out[n_bases] = '\0';
}

/* ------------------------------------------------------------------ kmer */

/* validate_kmer_sequence dna.c:457-479 (non-empty, <= 32, chars A/T/C/G/X) + encode_kmer
 * dna.c:397-420 ('X' encodes as 00, like 'A'). */
int orc_kmer_encode(const char *seq, int32_t *length, uint64_t *bits)
{
    if (seq == NULL || *seq == '\0')
        return ORC_ERR_KMER_EMPTY;
    size_t len = strlen(seq);
    if (len > 32)
        return ORC_ERR_KMER_TOO_LONG;
    uint64_t b = 0;
    for (size_t i = 0; i < len; i++) {
        int off = (int)i * 2;
        switch (seq[i]) {
        case 'A': break;
        case 'T': b |= (uint64_t)1 << off; break;
        case 'C': b |= (uint64_t)2 << off; break;
        case 'G': b |= (uint64_t)3 << off; break;
        case 'X': break;
        default: return ORC_ERR_KMER_BAD_CHAR;
        }
    }
    *length = (int32_t)len;
    *bits = b;
    return ORC_OK;
}

/* dna.c:428-452 */
void orc_kmer_decode(uint64_t bits, int k, char *out)
{
    for (int i = 0; i < k; i++)
        out[i] = CODE2CHAR[(bits >> (i * 2)) & 3];
    out[k] = '\0';
}

/* ------------------------------------------------------------------ generate_kmers */

/* dna.c:771-773 (1 <= k <= 32) and dna.c:781 (max_calls = length - k + 1); the reference's
 * unsigned underflow for length < k - 1 is NOT reproduced: 0 rows. */
int orc_generate_kmers_count(uint64_t n_bases, int k, uint64_t *n_kmers)
{
    if (k <= 0 || k > 32)
        return ORC_ERR_INVALID_K;
    *n_kmers = n_bases >= (uint64_t)k ? n_bases - (uint64_t)k + 1 : 0;
    return ORC_OK;
}

/* The per-row body of generate_kmers, dna.c:803-825: decode k bases to text (loop A), then
 * kmer_make -> validate (loop B) -> encode_kmer (loop C).  Positions are 64-bit here. */
int orc_generate_kmers(const uint64_t *words, uint64_t n_bases, int k,
                       uint64_t first, uint64_t count, uint64_t *out_keys)
{
    uint64_t total;
    int rc = orc_generate_kmers_count(n_bases, k, &total);
    if (rc)
        return rc;
    if (first > total || count > total - first)
        count = first > total ? 0 : total - first;
    char text[33];
    for (uint64_t r = 0; r < count; r++) {
        uint64_t pos = first + r;
        for (int i = 0; i < k; i++) {                     /* loop A, dna.c:803-820 */
            uint64_t ni = pos + (uint64_t)i;
            uint64_t bits = (words[ni / 32] >> ((ni * 2) % 64)) & 3;
            text[i] = CODE2CHAR[bits];
        }
        text[k] = '\0';
        int32_t len;
        uint64_t key;
        rc = orc_kmer_encode(text, &len, &key);           /* kmer_make, dna.c:825 */
        if (rc)
            return rc;
        out_keys[r] = key;
    }
    return ORC_OK;
}

/* k-mer at position p = bits [2p, 2p+2k) of the packed stream. */
static inline uint64_t window_at(const uint64_t *words, uint64_t n_words, uint64_t pos, int k)
{
    uint64_t w = pos / 32;
    unsigned sh = (unsigned)((pos % 32) * 2);
    uint64_t lo = words[w] >> sh;
    if (sh != 0 && w + 1 < n_words)
        lo |= words[w + 1] << (64 - sh);
    return k == 32 ? lo : (lo & (((uint64_t)1 << (2 * k)) - 1));
}

int orc_generate_kmers_fast(const uint64_t *words, uint64_t n_bases, int k,
                            uint64_t first, uint64_t count, uint64_t *out_keys)
{
    uint64_t total;
    int rc = orc_generate_kmers_count(n_bases, k, &total);
    if (rc)
        return rc;
    if (first > total || count > total - first)
        count = first > total ? 0 : total - first;
    uint64_t nw = orc_dna_num_words(n_bases);
    for (uint64_t r = 0; r < count; r++)
        out_keys[r] = window_at(words, nw, first + r, k);
    return ORC_OK;
}

/* The second counting shape of the reference: FROM a table of sequences, LATERAL generate_kmers(d.sequence, k)
 * (test.sql:140-150: one SRF call per row of the table, every call per dna.c:743-837, so no k-mer spans two rows).
 * The table is given as ONE concatenated packed stream and the first base of every sequence (starts[0 .. n_seqs], starts[0]
 * = 0, starts[n_seqs] = n_bases, ascending; equal neighbours = an empty sequence): sequence i yields the rows of
 * generate_kmers over its own bases -- the windows [starts[i], starts[i] + len_i - k] of the stream -- in table order;
 * a sequence shorter than k yields none (the 64-bit restatement of dna.c:781, as in orc_generate_kmers).  faithful != 0
 * takes the per-base form of dna.c:803-825 for every row.  out_keys: room for orc_table_kmers_count rows. */
int orc_table_kmers_count(const uint64_t *starts, uint64_t n_seqs, int k, uint64_t *n_rows)
{
    uint64_t probe, rows = 0;
    int rc = orc_generate_kmers_count(0, k, &probe);           /* (the k check of dna.c:771-773) */
    if (rc)
        return rc;
    for (uint64_t i = 0; i < n_seqs; i++) {
        uint64_t len = starts[i + 1] - starts[i];
        if (len >= (uint64_t)k)
            rows += len - (uint64_t)k + 1;
    }
    *n_rows = rows;
    return ORC_OK;
}

int orc_generate_kmers_table(const uint64_t *words, uint64_t n_bases, const uint64_t *starts, uint64_t n_seqs, int k,
                             int faithful, uint64_t *out_keys)
{
    uint64_t at = 0;
    for (uint64_t i = 0; i < n_seqs; i++) {
        if (starts[i + 1] < starts[i] || starts[i + 1] > n_bases)
            return ORC_ERR_NOMEM;
        uint64_t len = starts[i + 1] - starts[i];
        if (len < (uint64_t)k)
            continue;
        uint64_t rows = len - (uint64_t)k + 1;
        /* the sequence's own rows: the stream is cut behind its last base, so that the row rule of dna.c:781 is its own */
        int rc = faithful ? orc_generate_kmers(words, starts[i + 1], k, starts[i], rows, out_keys + at)
                          : orc_generate_kmers_fast(words, starts[i + 1], k, starts[i], rows, out_keys + at);
        if (rc)
            return rc;
        at += rows;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ operators */

/* dna.c:655-668 */
int orc_kmer_eq(int32_t len1, uint64_t bits1, int32_t len2, uint64_t bits2)
{
    return len1 == len2 && bits1 == bits2;
}

/* PostgreSQL 16 src/common/hashfn.c (not in /root/reference; reached through dna.c:10 and
 * dna.c:732): Bob Jenkins' lookup3 as PostgreSQL adapts it.  a = b = c = 0x9e3779b9 + len +
 * 3923095; for an aligned 8-byte key on a little-endian host the tail switch adds word 1 to b and
 * word 0 to a, then final() mixes; c is the result. */
#define ORC_ROT(x, k) (((x) << (k)) | ((x) >> (32 - (k))))
#define ORC_FINAL(a, b, c) do {            \
        c ^= b; c -= ORC_ROT(b, 14);       \
        a ^= c; a -= ORC_ROT(c, 11);       \
        b ^= a; b -= ORC_ROT(a, 25);       \
        c ^= b; c -= ORC_ROT(b, 16);       \
        a ^= c; a -= ORC_ROT(c, 4);        \
        b ^= a; b -= ORC_ROT(a, 14);       \
        c ^= b; c -= ORC_ROT(b, 24);       \
    } while (0)

/* dna.c:722-735: hash_any over the 8 bytes of bit_sequence only (length is not hashed). */
uint32_t orc_kmer_hash(uint64_t bits)
{
    uint32_t a, b, c;
    a = b = c = 0x9e3779b9u + 8u + 3923095u;
    b += (uint32_t)(bits >> 32);
    a += (uint32_t)bits;
    ORC_FINAL(a, b, c);
    return c;
}

/* hash_bytes_uint32 of the same file: anchors the restated final() to PostgreSQL's well-known
 * hashint4(1) = -1905060026. */
uint32_t orc_pg_hash_uint32(uint32_t k)
{
    uint32_t a, b, c;
    a = b = c = 0x9e3779b9u + (uint32_t)sizeof(uint32_t) + 3923095u;
    a += k;
    ORC_FINAL(a, b, c);
    return c;
}

/* dna.c:842-866.  Reference defect not reproduced: plen = 32 shifts by 64 there (UB; false on
 * x86 even for identical k-mers); here the mask is all ones. */
int orc_starts_with(int32_t klen, uint64_t kbits, int32_t plen, uint64_t pbits, int *result)
{
    if (plen > klen)
        return ORC_ERR_PREFIX_TOO_LONG;
    uint64_t mask = plen >= 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * plen)) - 1);
    *result = pbits == (kbits & mask);
    return ORC_OK;
}

/* dna.c:876-900 */
int orc_qkmer_validate(const char *pattern)
{
    if (pattern == NULL || *pattern == '\0')
        return ORC_ERR_QKMER_EMPTY;
    if (strlen(pattern) > 32)
        return ORC_ERR_QKMER_TOO_LONG;
    for (const char *p = pattern; *p; p++) {
        switch (*p) {
        case 'A': case 'T': case 'C': case 'G': case 'U': case 'W': case 'S': case 'M':
        case 'K': case 'R': case 'Y': case 'B': case 'D': case 'H': case 'V': case 'N':
            break;
        default:
            return ORC_ERR_QKMER_BAD_CHAR;
        }
    }
    return ORC_OK;
}

/* dna.c:1064-1086.  'U' compares the decoded base with 'U', which never occurs: matches nothing. */
static int nucleotide_matches(char n, char iupac)
{
    switch (iupac) {
    case 'A': return n == 'A';
    case 'T': return n == 'T';
    case 'C': return n == 'C';
    case 'G': return n == 'G';
    case 'U': return n == 'U';
    case 'W': return n == 'A' || n == 'T';
    case 'S': return n == 'C' || n == 'G';
    case 'M': return n == 'A' || n == 'C';
    case 'K': return n == 'G' || n == 'T';
    case 'R': return n == 'A' || n == 'G';
    case 'Y': return n == 'C' || n == 'T';
    case 'B': return n == 'C' || n == 'G' || n == 'T';
    case 'D': return n == 'A' || n == 'G' || n == 'T';
    case 'H': return n == 'A' || n == 'C' || n == 'T';
    case 'V': return n == 'A' || n == 'C' || n == 'G';
    case 'N': return 1;
    }
    return 0;
}

/* dna.c:1091-1135: length mismatch is an ERROR (1106-1108), then per-base decode + match with
 * early exit. */
int orc_contains(const char *pattern, int32_t klen, uint64_t kbits, int *result)
{
    int qlen = (int)strlen(pattern);
    if (qlen != klen)
        return ORC_ERR_QKMER_LEN_MISMATCH;
    for (int i = 0; i < qlen; i++) {
        char n = CODE2CHAR[(kbits >> (i * 2)) & 3];
        if (!nucleotide_matches(n, pattern[i])) {
            *result = 0;
            return ORC_OK;
        }
    }
    *result = 1;
    return ORC_OK;
}

/* generate_kmers + WHERE pattern @> kmer, rows in position order (test.sql:86-92). */
int orc_generate_kmers_contains(const uint64_t *words, uint64_t n_bases, int k, const char *pattern,
                                uint64_t *out_keys, uint64_t *out_pos, uint64_t cap, uint64_t *n_out)
{
    uint64_t total;
    int rc = orc_generate_kmers_count(n_bases, k, &total);
    if (rc)
        return rc;
    rc = orc_qkmer_validate(pattern);
    if (rc)
        return rc;
    if ((int)strlen(pattern) != k && total > 0)
        return ORC_ERR_QKMER_LEN_MISMATCH;   /* raised by the first row's contains() call */
    uint64_t nw = orc_dna_num_words(n_bases), n = 0;
    for (uint64_t p = 0; p < total; p++) {
        uint64_t key = window_at(words, nw, p, k);
        int m;
        rc = orc_contains(pattern, k, key, &m);
        if (rc)
            return rc;
        if (m) {
            if (n < cap) {
                if (out_keys) out_keys[n] = key;
                if (out_pos) out_pos[n] = p;
            }
            n++;
        }
    }
    *n_out = n;
    return ORC_OK;
}

/* generate_kmers + WHERE kmer ^@ prefix (test.sql:67-73) */
int orc_generate_kmers_starts_with(const uint64_t *words, uint64_t n_bases, int k,
                                   int32_t plen, uint64_t pbits,
                                   uint64_t *out_keys, uint64_t *out_pos, uint64_t cap, uint64_t *n_out)
{
    uint64_t total;
    int rc = orc_generate_kmers_count(n_bases, k, &total);
    if (rc)
        return rc;
    if (plen > k && total > 0)
        return ORC_ERR_PREFIX_TOO_LONG;
    uint64_t nw = orc_dna_num_words(n_bases), n = 0;
    for (uint64_t p = 0; p < total; p++) {
        uint64_t key = window_at(words, nw, p, k);
        int m;
        rc = orc_starts_with(k, key, plen, pbits, &m);
        if (rc)
            return rc;
        if (m) {
            if (n < cap) {
                if (out_keys) out_keys[n] = key;
                if (out_pos) out_pos[n] = p;
            }
            n++;
        }
    }
    *n_out = n;
    return ORC_OK;
}

/* generate_kmers + WHERE kmer = q (test.sql:61-65): kmer_eq compares length and bits */
int orc_generate_kmers_equals(const uint64_t *words, uint64_t n_bases, int k,
                              int32_t qlen, uint64_t qbits,
                              uint64_t *out_keys, uint64_t *out_pos, uint64_t cap, uint64_t *n_out)
{
    uint64_t total;
    int rc = orc_generate_kmers_count(n_bases, k, &total);
    if (rc)
        return rc;
    uint64_t nw = orc_dna_num_words(n_bases), n = 0;
    for (uint64_t p = 0; p < total; p++) {
        uint64_t key = window_at(words, nw, p, k);
        if (orc_kmer_eq(k, key, qlen, qbits)) {
            if (n < cap) {
                if (out_keys) out_keys[n] = key;
                if (out_pos) out_pos[n] = p;
            }
            n++;
        }
    }
    *n_out = n;
    return ORC_OK;
}

/* ------------------------------------------------------------------ GROUP BY count(*) */

static int cmp_u64_pair(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

/* PostgreSQL HashAggregate shape (nodeAgg.c/execGrouping.c, not in the reference): one table
 * probe per row using the opclass's hash (kmer_hash, dna.c:722-735) and equality (kmer_eq,
 * dna.c:686-696; all rows of one generate_kmers call share a length, so equality is on bits),
 * transition = count + 1.  Groups are then sorted by key for a canonical result. */
int orc_count_keys(const uint64_t *keys, uint64_t n, uint64_t **out_keys, uint64_t **out_counts,
                   uint64_t *n_distinct)
{
    uint64_t cap = 16;
    while (cap < n * 2 + 2)
        cap <<= 1;
    uint64_t *tk = (uint64_t *)malloc(cap * sizeof(uint64_t));
    uint64_t *tc = (uint64_t *)calloc(cap, sizeof(uint64_t));
    if (!tk || !tc) {
        free(tk); free(tc);
        return ORC_ERR_NOMEM;
    }
    uint64_t mask = cap - 1, d = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t key = keys[i];
        uint64_t h = orc_kmer_hash(key);
        h = (h * 0x9E3779B97F4A7C15ull) >> 20;   /* spread the 32-bit hash over a large table */
        uint64_t s = h & mask;
        for (;;) {
            if (tc[s] == 0) { tk[s] = key; tc[s] = 1; d++; break; }
            if (tk[s] == key) { tc[s]++; break; }
            s = (s + 1) & mask;
        }
    }
    uint64_t *pairs = (uint64_t *)malloc((d ? d : 1) * 2 * sizeof(uint64_t));
    uint64_t *ok = (uint64_t *)malloc((d ? d : 1) * sizeof(uint64_t));
    uint64_t *oc = (uint64_t *)malloc((d ? d : 1) * sizeof(uint64_t));
    if (!pairs || !ok || !oc) {
        free(tk); free(tc); free(pairs); free(ok); free(oc);
        return ORC_ERR_NOMEM;
    }
    uint64_t j = 0;
    for (uint64_t s = 0; s < cap; s++)
        if (tc[s]) { pairs[2 * j] = tk[s]; pairs[2 * j + 1] = tc[s]; j++; }
    free(tk); free(tc);
    qsort(pairs, d, 2 * sizeof(uint64_t), cmp_u64_pair);
    for (uint64_t i = 0; i < d; i++) { ok[i] = pairs[2 * i]; oc[i] = pairs[2 * i + 1]; }
    free(pairs);
    *out_keys = ok; *out_counts = oc; *n_distinct = d;
    return ORC_OK;
}

int orc_count_kmers(const uint64_t *words, uint64_t n_bases, int k, int faithful,
                    uint64_t **out_keys, uint64_t **out_counts, uint64_t *n_distinct)
{
    uint64_t total;
    int rc = orc_generate_kmers_count(n_bases, k, &total);
    if (rc)
        return rc;
    uint64_t *keys = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
    if (!keys)
        return ORC_ERR_NOMEM;
    rc = faithful ? orc_generate_kmers(words, n_bases, k, 0, total, keys)
                  : orc_generate_kmers_fast(words, n_bases, k, 0, total, keys);
    if (!rc)
        rc = orc_count_keys(keys, total, out_keys, out_counts, n_distinct);
    free(keys);
    return rc;
}

/* The same GROUP BY restricted to the k-mers of one slice of the key space: slice = (mix(key) >> 32) * n_slices
 * >> 32 with mix = splitmix64 -- every copy of a k-mer falls into the same slice, so the groups of the n_slices
 * slices are disjoint and their union is orc_count_kmers' result; total / distinct / unique / checksum of the whole
 * histogram are the sums over the slices.  This is how tools/make_digests.py counts sequences whose 8-byte keys would
 * not fit this container's memory at once (3 Gbase: 24 GB of keys).  Extraction is the word-arithmetic form
 * (cross-checked against the faithful per-base form in tests/test_oracle_golden.py); the aggregate is
 * orc_count_keys (kmer_hash + kmer_eq, dna.c:722-735, 655-668). */
int orc_count_kmers_slice(const uint64_t *words, uint64_t n_bases, int k, uint32_t slice, uint32_t n_slices,
                          uint64_t **out_keys, uint64_t **out_counts, uint64_t *n_distinct)
{
    uint64_t total;
    int rc = orc_generate_kmers_count(n_bases, k, &total);
    if (rc)
        return rc;
    if (n_slices == 0 || slice >= n_slices)
        return ORC_ERR_NOMEM;
    uint64_t nw = orc_dna_num_words(n_bases);
    uint64_t cap = total / n_slices + total / (8 * (uint64_t)n_slices) + 65536, n = 0;
    uint64_t *keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    if (!keys)
        return ORC_ERR_NOMEM;
    for (uint64_t p = 0; p < total; p++) {
        uint64_t key = window_at(words, nw, p, k);
        uint64_t s = ((orc_splitmix64(key) >> 32) * (uint64_t)n_slices) >> 32;
        if (s != slice)
            continue;
        if (n == cap) {
            cap += cap / 2;
            uint64_t *g = (uint64_t *)realloc(keys, cap * sizeof(uint64_t));
            if (!g) { free(keys); return ORC_ERR_NOMEM; }
            keys = g;
        }
        keys[n++] = key;
    }
    rc = orc_count_keys(keys, n, out_keys, out_counts, n_distinct);
    free(keys);
    return rc;
}

void orc_free(void *p) { free(p); }

/* order-independent digest of one (key, count) group */
uint64_t orc_pair_mix(uint64_t key, uint64_t count)
{
    uint64_t x = orc_splitmix64(key ^ 0x6a09e667f3bcc909ull);
    return x * (2 * count + 1) + orc_splitmix64(count);
}

/* sum(count), count(*) FILTER (WHERE count = 1) (test.sql:112-114) + wrapping sum of pair digests */
void orc_hist_summary(const uint64_t *keys, const uint64_t *counts, uint64_t n_distinct,
                      uint64_t *total, uint64_t *unique, uint64_t *checksum)
{
    uint64_t t = 0, u = 0, c = 0;
    for (uint64_t i = 0; i < n_distinct; i++) {
        t += counts[i];
        u += counts[i] == 1;
        c += orc_pair_mix(keys[i], counts[i]);
    }
    if (total) *total = t;
    if (unique) *unique = u;
    if (checksum) *checksum = c;
}

/* ------------------------------------------------------------------ synthetic input */

uint64_t orc_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* i.i.d. uniform 2-bit codes = the distribution of data/create_dna.py:27-33; tail bits of the
 * last word are zero like palloc0'd storage (dna.c:186). */
void orc_synth_words(uint64_t seed, uint64_t n_bases, uint64_t *words)
{
    uint64_t nw = orc_dna_num_words(n_bases);
    for (uint64_t w = 0; w < nw; w++)
        words[w] = orc_splitmix64(seed + w);
    unsigned tail = (unsigned)(n_bases % 32);
    if (nw && tail)
        words[nw - 1] &= (((uint64_t)1 << (2 * tail)) - 1);
}

static inline uint64_t synth_base(uint64_t seed, uint64_t i)
{
    return (orc_splitmix64(seed + i / 32) >> ((i % 32) * 2)) & 3;
}

void orc_synth_words_repeat(uint64_t seed, uint64_t n_bases, uint64_t motif_len, uint64_t *words)
{
    uint64_t nw = orc_dna_num_words(n_bases), half = n_bases / 2;
    if (motif_len == 0)
        motif_len = 1;
    for (uint64_t w = 0; w < nw; w++) {
        uint64_t v = 0;
        for (unsigned j = 0; j < 32; j++) {
            uint64_t i = w * 32 + j;
            if (i >= n_bases)
                break;
            uint64_t src = i < half ? i : (i - half) % motif_len;
            v |= synth_base(seed, src) << (2 * j);
        }
        words[w] = v;
    }
}

/* ---- binary wire image: dna_send/dna_recv (dna.c:244-291), kmer_send/kmer_recv (dna.c:552-597) ---- */
static void put_be64(unsigned char *p, uint64_t v)
{
    for (int i = 0; i < 8; i++)
        p[i] = (unsigned char)(v >> (56 - 8 * i));          /* pq_sendint64: network byte order */
}
static uint64_t get_be64(const unsigned char *p)
{
    uint64_t v = 0;
    for (int i = 0; i < 8; i++)
        v = (v << 8) | p[i];
    return v;
}

uint64_t orc_dna_wire_size(uint64_t n_bases) { return 8 + 8 * orc_dna_num_words(n_bases); }

void orc_dna_to_wire(const uint64_t *words, uint64_t n_bases, unsigned char *wire)
{
    uint64_t bit_length = orc_dna_num_words(n_bases);        /* dna.c:279 */
    put_be64(wire, n_bases);                                 /* dna.c:282 (as an int64) */
    for (uint64_t i = 0; i < bit_length; i++)                /* dna.c:284-286 */
        put_be64(wire + 8 + 8 * i, words[i]);
}

int orc_dna_from_wire(const unsigned char *wire, uint64_t wire_bytes, uint64_t *n_bases, uint64_t *words)
{
    if (wire_bytes < 8)
        return ORC_ERR_NOMEM;
    uint64_t length = get_be64(wire);                        /* dna.c:251 */
    if (length == 0)
        return ORC_ERR_DNA_EMPTY;                            /* the type has no empty value (dna.c:160-161) */
    uint64_t bit_length = orc_dna_num_words(length);         /* dna.c:252-253 */
    if (length > (uint64_t)1 << 40 || wire_bytes != 8 + 8 * bit_length)
        return ORC_ERR_NOMEM;
    *n_bases = length;
    if (words) {
        for (uint64_t i = 0; i < bit_length; i++)            /* dna.c:263-265 */
            words[i] = get_be64(wire + 8 + 8 * i);
        if (length % 32)                                     /* keep the palloc0 invariant: tail bits zero */
            words[bit_length - 1] &= (((uint64_t)1 << (2 * (length % 32))) - 1);
    }
    return ORC_OK;
}

void orc_kmer_to_wire(int32_t length, uint64_t bits, unsigned char wire[12])
{
    uint32_t l = (uint32_t)length;                           /* dna.c:588: pq_sendint(.., sizeof(int)) */
    wire[0] = (unsigned char)(l >> 24); wire[1] = (unsigned char)(l >> 16);
    wire[2] = (unsigned char)(l >> 8);  wire[3] = (unsigned char)l;
    put_be64(wire + 4, bits);                                /* dna.c:591 */
}

int orc_kmer_from_wire(const unsigned char wire[12], int32_t *length, uint64_t *bits)
{
    int32_t l = (int32_t)(((uint32_t)wire[0] << 24) | ((uint32_t)wire[1] << 16) | ((uint32_t)wire[2] << 8) | wire[3]);
    if (l <= 0 || l > 32)
        return ORC_ERR_KMER_TOO_LONG;                        /* dna.c:566-568 */
    *length = l;
    *bits = get_be64(wire + 4);                              /* dna.c:571 */
    return ORC_OK;
}
