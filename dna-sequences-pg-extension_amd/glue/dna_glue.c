/*
 * dna_glue.c -- see dna_glue.h.  Written against the behaviour of /root/reference/dna.c (cited per
 * function); PostgreSQL plumbing replaced by plain C.  Heavy lifting: include/dnagpu.h.
 */
#include "dna_glue.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dnagpu.h"

static __thread char g_msg[256];
static dnagpu_ctx *g_ctx;          /* one per process, created lazily (post-fork) */
static int g_device;

static void ereport_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_msg, sizeof g_msg, fmt, ap);
    va_end(ap);
}

const char *dna_glue_errmsg(void) { return g_msg; }
void dna_glue_set_device(int device) { g_device = device; }

static void multi_shutdown(void);

void dna_glue_shutdown(void)
{
    multi_shutdown();
    if (g_ctx) {
        dnagpu_destroy(g_ctx);
        g_ctx = NULL;
    }
}

/* status of a dnagpu call -> the reference's message (codes 1..3) or the library's detail */
static bool gpu_ok(int rc)
{
    if (rc == DNAGPU_OK)
        return true;
    if (rc <= DNAGPU_ERR_QKMER_INVALID)
        ereport_error("%s", dnagpu_strerror(rc));
    else
        ereport_error("%s: %s", dnagpu_strerror(rc), dnagpu_last_error());
    return false;
}

static dnagpu_ctx *ctx(void)
{
    if (!g_ctx && !gpu_ok(dnagpu_init(g_device, &g_ctx)))
        return NULL;
    return g_ctx;
}

static dnagpu_dna *device_dna(Dna *dna)
{
    if (!dna->dev) {
        dnagpu_dna *d = NULL;
        if (!ctx() || !gpu_ok(dnagpu_dna_upload(g_ctx, dna->bit_sequence, dna->length, &d)))
            return NULL;
        dna->dev = d;
    }
    return (dnagpu_dna *)dna->dev;
}

/* ------------------------------------------------------------------ dna */

/* dna_in -> dna_make -> validate_dna_sequence + encode_dna (dna.c:220-228, 178-202, 159-171, 114-128) */
Dna *dna_in(const char *str)
{
    if (str == NULL || *str == '\0') {
        ereport_error("DNA sequence cannot be empty");                    /* dna.c:161 */
        return NULL;
    }
    for (const char *p = str; *p; p++)
        if (*p != 'A' && *p != 'T' && *p != 'C' && *p != 'G') {
            ereport_error("Invalid character in DNA sequence: %c", *p);   /* dna.c:166 */
            return NULL;
        }
    uint64_t length = (uint64_t)strlen(str);
    uint64_t n_words = (length * 2 + 63) / 64;                            /* dna.c:181-182 */
    Dna *dna = (Dna *)calloc(1, sizeof(Dna));
    uint64_t *w = (uint64_t *)calloc(n_words ? n_words : 1, sizeof(uint64_t));   /* palloc0, dna.c:186 */
    if (!dna || !w) {
        free(dna);
        free(w);
        ereport_error("out of memory");
        return NULL;
    }
    for (uint64_t i = 0; i < length; i++) {                               /* dna.c:115-127 */
        uint64_t code = str[i] == 'A' ? 0 : str[i] == 'T' ? 1 : str[i] == 'C' ? 2 : 3;
        w[i / 32] |= code << ((i * 2) % 64);
    }
    dna->length = length;
    dna->bit_sequence = w;
    return dna;
}

/* dna_out -> decode_dna (dna.c:230-242, 135-152) */
char *dna_out(const Dna *dna)
{
    static const char L[4] = { 'A', 'T', 'C', 'G' };
    char *s = (char *)malloc(dna->length + 1);
    if (!s)
        return NULL;
    for (uint64_t i = 0; i < dna->length; i++)
        s[i] = L[(dna->bit_sequence[i / 32] >> ((i * 2) % 64)) & 3];
    s[dna->length] = '\0';
    return s;
}

/* ---- binary I/O: per-datum host loops like dna_in/dna_out (bulk, device-side: dnagpu_dna_from_wire) ---- */
static void put_be64(unsigned char *p, uint64_t v)
{
    for (int i = 0; i < 8; i++)
        p[i] = (unsigned char)(v >> (56 - 8 * i));                        /* pq_sendint64 */
}
static uint64_t get_be64(const unsigned char *p)
{
    uint64_t v = 0;
    for (int i = 0; i < 8; i++)
        v = (v << 8) | p[i];                                              /* pq_getmsgint64 */
    return v;
}

/* dna_send (dna.c:270-291) */
unsigned char *dna_send(const Dna *dna, size_t *wire_bytes)
{
    uint64_t bit_length = (dna->length * 2 + 63) / 64;                    /* dna.c:279 */
    unsigned char *buf = (unsigned char *)malloc(8 + 8 * bit_length);
    if (!buf) {
        ereport_error("out of memory");
        return NULL;
    }
    put_be64(buf, dna->length);                                           /* dna.c:282, as an int64 */
    for (uint64_t i = 0; i < bit_length; i++)                             /* dna.c:284-286 */
        put_be64(buf + 8 + 8 * i, dna->bit_sequence[i]);
    *wire_bytes = 8 + 8 * bit_length;
    return buf;
}

/* dna_recv (dna.c:244-268) */
Dna *dna_recv(const unsigned char *wire, size_t wire_bytes)
{
    if (wire == NULL || wire_bytes < 8) {
        ereport_error("insufficient data left in message");               /* pq_getmsgint64's own ERROR */
        return NULL;
    }
    uint64_t length = get_be64(wire);                                     /* dna.c:251 */
    if (length == 0) {
        ereport_error("DNA sequence cannot be empty");                    /* the type's invariant, dna.c:161 */
        return NULL;
    }
    uint64_t bit_length = (length * 2 + 63) / 64;                         /* dna.c:252-253 */
    if (length > ((uint64_t)1 << 40) || wire_bytes != 8 + 8 * bit_length) {
        ereport_error("insufficient data left in message");
        return NULL;
    }
    Dna *dna = (Dna *)calloc(1, sizeof(Dna));
    uint64_t *w = (uint64_t *)calloc(bit_length, sizeof(uint64_t));       /* palloc0, dna.c:257 */
    if (!dna || !w) {
        free(dna);
        free(w);
        ereport_error("out of memory");
        return NULL;
    }
    for (uint64_t i = 0; i < bit_length; i++)                             /* dna.c:263-265 */
        w[i] = get_be64(wire + 8 + 8 * i);
    if (length % 32)
        w[bit_length - 1] &= (((uint64_t)1 << (2 * (length % 32))) - 1);  /* keep the tail bits zero (dna.c:186) */
    dna->length = length;
    dna->bit_sequence = w;
    return dna;
}

/* kmer_send (dna.c:579-597) */
void kmer_send(const Kmer *kmer, unsigned char wire[12])
{
    uint32_t l = (uint32_t)kmer->length;                                  /* dna.c:588 */
    wire[0] = (unsigned char)(l >> 24);
    wire[1] = (unsigned char)(l >> 16);
    wire[2] = (unsigned char)(l >> 8);
    wire[3] = (unsigned char)l;
    put_be64(wire + 4, kmer->bit_sequence);                               /* dna.c:591 */
}

/* kmer_recv (dna.c:552-574) */
bool kmer_recv(const unsigned char wire[12], Kmer *out)
{
    int32_t length = (int32_t)(((uint32_t)wire[0] << 24) | ((uint32_t)wire[1] << 16) |
                               ((uint32_t)wire[2] << 8) | wire[3]);       /* dna.c:559 */
    if (length <= 0 || length > 32) {
        ereport_error("Invalid K-mer length: must be between 1 and 32");  /* dna.c:566-568 */
        return false;
    }
    out->length = length;
    out->bit_sequence = get_be64(wire + 4);                               /* dna.c:571 */
    return true;
}

void dna_free(Dna *dna)
{
    if (!dna)
        return;
    if (dna->dev)
        dnagpu_dna_free(g_ctx, (dnagpu_dna *)dna->dev);
    free(dna->bit_sequence);
    free(dna);
}

uint64_t dna_length(const Dna *dna) { return dna->length; }

/* ------------------------------------------------------------------ kmer / qkmer text */

/* kmer_in -> kmer_make -> validate_kmer_sequence + encode_kmer (dna.c:528-536, 487-515, 457-479, 397-420) */
bool kmer_in(const char *str, Kmer *out)
{
    if (str == NULL) {
        ereport_error("K-mer sequence cannot be NULL");                   /* dna.c:496 */
        return false;
    }
    if (*str == '\0') {
        ereport_error("K-mer sequence cannot be empty");                  /* dna.c:461 */
        return false;
    }
    size_t len = strlen(str);
    if (len > 32) {
        ereport_error("K-mer length cannot exceed 32 nucleotides");       /* dna.c:467 */
        return false;
    }
    uint64_t bits = 0;
    for (size_t i = 0; i < len; i++) {
        uint64_t code;
        switch (str[i]) {
        case 'A': case 'X': code = 0; break;                              /* 'X' is 00 like 'A', dna.c:413 */
        case 'T': code = 1; break;
        case 'C': code = 2; break;
        case 'G': code = 3; break;
        default:
            ereport_error("Invalid character in K-mer sequence: '%c'", str[i]);   /* dna.c:473 */
            return false;
        }
        bits |= code << (2 * i);
    }
    out->length = (int32_t)len;
    out->bit_sequence = bits;
    return true;
}

/* kmer_out -> decode_kmer (dna.c:538-546, 428-452) */
char *kmer_out(const Kmer *kmer)
{
    static const char L[4] = { 'A', 'T', 'C', 'G' };
    if (kmer->length <= 0 || kmer->length > 32) {
        ereport_error("K-mer length must be between 1 and 32 nucleotides");   /* dna.c:434 */
        return NULL;
    }
    char *s = (char *)malloc((size_t)kmer->length + 1);
    if (!s)
        return NULL;
    for (int i = 0; i < kmer->length; i++)
        s[i] = L[(kmer->bit_sequence >> (2 * i)) & 3];
    s[kmer->length] = '\0';
    return s;
}

/* qkmer_in -> qkmer_make -> validate_qkmer_pattern (dna.c:932-940, 908-930, 876-900) */
bool qkmer_in(const char *str, Qkmer *out)
{
    if (str == NULL || *str == '\0') {
        ereport_error("qkmer pattern cannot be empty");                   /* dna.c:878 */
        return false;
    }
    if (strlen(str) > 32) {
        ereport_error("Qkmer pattern length cannot exceed 32 characters");   /* dna.c:884 */
        return false;
    }
    for (const char *p = str; *p; p++)
        if (!strchr("ATCGUWSMKRYBDHVN", *p)) {
            ereport_error("Invalid character in qkmer pattern: %c", *p);  /* dna.c:894 */
            return false;
        }
    strcpy(out->sequence, str);
    return true;
}

/* ------------------------------------------------------------------ per-datum operators */

bool kmer_eq(const Kmer *a, const Kmer *b)                                /* kmer_eq_internal, dna.c:655-668 */
{
    return a->length == b->length && a->bit_sequence == b->bit_sequence;
}

bool kmer_ne(const Kmer *a, const Kmer *b) { return !kmer_eq(a, b); }

/* dna.c:722-735: hash_any over the 8 bytes of bit_sequence (PostgreSQL hash_bytes, lookup3). */
int32_t kmer_hash(const Kmer *kmer)
{
#define ROT(x, k) (((x) << (k)) | ((x) >> (32 - (k))))
    uint32_t a, b, c;
    a = b = c = 0x9e3779b9u + 8u + 3923095u;
    b += (uint32_t)(kmer->bit_sequence >> 32);
    a += (uint32_t)kmer->bit_sequence;
    c ^= b; c -= ROT(b, 14);
    a ^= c; a -= ROT(c, 11);
    b ^= a; b -= ROT(a, 25);
    c ^= b; c -= ROT(b, 16);
    a ^= c; a -= ROT(c, 4);
    b ^= a; b -= ROT(a, 14);
    c ^= b; c -= ROT(b, 24);
#undef ROT
    return (int32_t)c;
}

/* dna.c:842-866; a 32-base prefix compares all 64 bits (the reference shifts by 64 there: UB) */
int starts_with(const Kmer *kmer, const Kmer *prefix)
{
    if (prefix->length > kmer->length) {
        ereport_error("Prefix length cannot exceed kmer length");        /* dna.c:855 */
        return -1;
    }
    uint64_t mask = prefix->length >= 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * prefix->length)) - 1);
    return prefix->bit_sequence == (kmer->bit_sequence & mask);
}

/* nucleotide_matches (dna.c:1064-1086) as a set lookup: bit0 = A, bit1 = T, bit2 = C, bit3 = G */
static int iupac_set(char c)
{
    switch (c) {
    case 'A': return 1; case 'T': return 2; case 'C': return 4; case 'G': return 8;
    case 'U': return 0;                                                   /* never equals a decoded base */
    case 'W': return 3; case 'S': return 12; case 'M': return 5; case 'K': return 10;
    case 'R': return 9; case 'Y': return 6; case 'B': return 14; case 'D': return 11;
    case 'H': return 7; case 'V': return 13; case 'N': return 15;
    }
    return -1;
}

/* dna.c:1091-1135 */
int contains(const Qkmer *pattern, const Kmer *kmer)
{
    int qlen = (int)strlen(pattern->sequence);
    if (qlen != kmer->length) {
        ereport_error("Qkmer pattern and kmer lengths do not match");    /* dna.c:1107 */
        return -1;
    }
    for (int i = 0; i < qlen; i++) {
        int code = (int)((kmer->bit_sequence >> (2 * i)) & 3);
        int set = iupac_set(pattern->sequence[i]);
        if (set < 0) {
            ereport_error("Invalid character in pattern: %c!", pattern->sequence[i]);   /* dna.c:1083 */
            return -1;
        }
        if (!(set & (1 << code)))
            return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ generate_kmers */

#define GK_WINDOW ((uint64_t)1 << 22)     /* rows fetched from the GPU per refill (32 MiB of keys) */

struct GenerateKmers {
    Dna *dna;
    int k;
    uint64_t total;        /* max_calls, dna.c:781 */
    uint64_t next_row;     /* call_cntr */
    /* host window */
    uint64_t *keys;
    uint64_t win_first, win_count;
    /* filtered mode */
    bool filtered;
    dnagpu_filter filter;
    uint64_t scan_pos;     /* next unscanned row of generate_kmers */
    uint64_t win_used;
    bool failed;           /* an ERROR was raised: the statement is aborted */
};

static GenerateKmers *gk_new(Dna *dna, int k)
{
    uint64_t total = 0;
    if (!gpu_ok(dnagpu_kmer_count(dna->length, k, &total)))              /* dna.c:771-773 */
        return NULL;
    GenerateKmers *g = (GenerateKmers *)calloc(1, sizeof *g);
    if (!g) {
        ereport_error("out of memory");
        return NULL;
    }
    g->dna = dna;
    g->k = k;
    g->total = total;
    g->keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(total < GK_WINDOW ? (total ? total : 1) : GK_WINDOW));
    if (!g->keys) {
        free(g);
        ereport_error("out of memory");
        return NULL;
    }
    return g;
}

GenerateKmers *generate_kmers_begin(Dna *dna, int k)
{
    return gk_new(dna, k);
}

GenerateKmers *generate_kmers_where_begin(Dna *dna, int k, char op, const Kmer *rhs, const Qkmer *rhs_pattern)
{
    GenerateKmers *g = gk_new(dna, k);
    if (!g)
        return NULL;
    g->filtered = true;
    memset(&g->filter, 0, sizeof g->filter);
    switch (op) {
    case '=':
        g->filter.kind = DNAGPU_FILTER_EQUALS;
        g->filter.length = rhs->length;
        g->filter.bits = rhs->bit_sequence;
        break;
    case '^':
        g->filter.kind = DNAGPU_FILTER_STARTS_WITH;
        g->filter.length = rhs->length;
        g->filter.bits = rhs->bit_sequence;
        break;
    case '@':
        g->filter.kind = DNAGPU_FILTER_CONTAINS;
        strncpy(g->filter.pattern, rhs_pattern->sequence, sizeof g->filter.pattern - 1);
        break;
    default:
        ereport_error("unknown operator");
        generate_kmers_end(g);
        return NULL;
    }
    return g;
}

bool generate_kmers_next(GenerateKmers *g, Kmer *out)
{
    if (!g->filtered) {
        if (g->next_row >= g->total)
            return false;                                                 /* SRF_RETURN_DONE */
        if (g->next_row >= g->win_first + g->win_count) {
            dnagpu_dna *d = device_dna(g->dna);
            uint64_t n = g->total - g->next_row < GK_WINDOW ? g->total - g->next_row : GK_WINDOW;
            if (!d || !gpu_ok(dnagpu_generate_kmers(g_ctx, d, g->k, g->next_row, n, g->keys, 0))) {
                g->failed = true;
                return false;
            }
            g->win_first = g->next_row;
            g->win_count = n;
        }
        out->length = g->k;
        out->bit_sequence = g->keys[g->next_row - g->win_first];
        g->next_row++;
        return true;
    }
    /* filtered: refill with the next window of source rows until one has matches */
    while (g->win_used >= g->win_count) {
        if (g->scan_pos >= g->total)
            return false;
        dnagpu_dna *d = device_dna(g->dna);
        uint64_t n = g->total - g->scan_pos < GK_WINDOW ? g->total - g->scan_pos : GK_WINDOW;
        uint64_t n_out = 0;
        if (!d || !gpu_ok(dnagpu_generate_kmers_filtered(g_ctx, d, g->k, &g->filter, g->scan_pos, n, g->keys,
                                                          NULL, GK_WINDOW, &n_out, 0))) {
            g->scan_pos = g->total;                                       /* the ERROR aborts the statement */
            g->failed = true;
            return false;
        }
        g->scan_pos += n;
        g->win_count = n_out;
        g->win_used = 0;
    }
    out->length = g->k;
    out->bit_sequence = g->keys[g->win_used++];
    return true;
}

bool generate_kmers_failed(const GenerateKmers *g) { return g->failed; }

void generate_kmers_end(GenerateKmers *g)
{
    if (!g)
        return;
    free(g->keys);
    free(g);
}

/* ------------------------------------------------------------------ count_kmers */

#define CK_WINDOW ((uint64_t)1 << 21)
#define CK_MAX_RANKS 64

/* one histogram per rank (one rank unless dna_glue_set_gpus asked for more); rank order = ascending key ranges */
struct CountKmers {
    int k;
    int n_ranks, rank;                       /* rank = the histogram rows are being served from */
    dnagpu_hist *hist[CK_MAX_RANKS];
    dnagpu_ctx *hctx[CK_MAX_RANKS];
    uint64_t rank_distinct, next, win_first, win_count;   /* cursor inside hist[rank] */
    uint64_t distinct;
    uint64_t *keys, *counts;
    uint64_t total, unique;
};

/* multi-GPU configuration of this backend: count_kmers shards over these devices (dnagpu_count_multi) */
static int g_n_gpus = 1, g_transport = DNAGPU_MULTI_AUTO;
static int g_gpu_list[CK_MAX_RANKS];
static dnagpu_multi *g_multi;

void dna_glue_set_gpus(int n_gpus, const int *devices, int transport)
{
    if (g_multi) {
        dnagpu_multi_destroy(g_multi);
        g_multi = NULL;
    }
    g_n_gpus = n_gpus < 1 ? 1 : (n_gpus > CK_MAX_RANKS ? CK_MAX_RANKS : n_gpus);
    for (int r = 0; r < g_n_gpus; r++)
        g_gpu_list[r] = devices ? devices[r] : r;
    g_transport = transport;
}

static void multi_shutdown(void)
{
    if (g_multi) {
        dnagpu_multi_destroy(g_multi);
        g_multi = NULL;
    }
}

static dnagpu_multi *multi(void)
{
    if (!g_multi && !gpu_ok(dnagpu_multi_init(g_gpu_list, g_n_gpus, g_transport, &g_multi)))
        return NULL;
    return g_multi;
}

CountKmers *count_kmers_begin(Dna *dna, int k)
{
    uint64_t n_rows = 0;
    if (!gpu_ok(dnagpu_kmer_count(dna->length, k, &n_rows)))
        return NULL;
    CountKmers *c = (CountKmers *)calloc(1, sizeof *c);
    if (!c) {
        ereport_error("out of memory");
        return NULL;
    }
    c->k = k;
    bool ok;
    if (g_n_gpus > 1) {
        /* the sequence goes to the ranks as contiguous word chunks; one all-gather + per-rank owner counts */
        dnagpu_multi *m = multi();
        dnagpu_multi_dna *md = NULL;
        ok = m && gpu_ok(dnagpu_multi_dna_upload(m, dna->bit_sequence, dna->length, &md));
        if (ok) {
            c->n_ranks = g_n_gpus;
            for (int r = 0; r < c->n_ranks; r++)
                c->hctx[r] = dnagpu_multi_ctx(m, r);
            ok = gpu_ok(dnagpu_count_multi_unordered(m, md, k, 0, n_rows, c->hist));   /* GROUP BY promises no order */
            dnagpu_multi_dna_free(m, md);
        }
    } else {
        dnagpu_dna *d = device_dna(dna);
        ok = d != NULL;
        if (ok) {
            c->n_ranks = 1;
            c->hctx[0] = g_ctx;
            /* GROUP BY promises no order (test.sql:95-104): the entry point that may partition by super-k-mers */
            ok = gpu_ok(dnagpu_count_kmers_unordered(g_ctx, d, k, 0, n_rows, &c->hist[0]));
        }
    }
    for (int r = 0; ok && r < c->n_ranks; r++) {
        uint64_t t = 0, u = 0, checksum;
        ok = gpu_ok(dnagpu_hist_summary(c->hctx[r], c->hist[r], &t, &u, &checksum));
        c->total += t;
        c->unique += u;
        c->distinct += dnagpu_hist_distinct(c->hist[r]);
    }
    if (!ok) {
        count_kmers_end(c);
        return NULL;
    }
    c->rank = 0;
    c->rank_distinct = dnagpu_hist_distinct(c->hist[0]);
    c->keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)CK_WINDOW);
    c->counts = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)CK_WINDOW);
    if (!c->keys || !c->counts) {
        ereport_error("out of memory");
        count_kmers_end(c);
        return NULL;
    }
    return c;
}

bool count_kmers_next(CountKmers *c, Kmer *kmer, int64_t *count)
{
    while (c->next >= c->rank_distinct) {            /* this rank's key range is served: on to the next one */
        if (c->rank + 1 >= c->n_ranks)
            return false;
        c->rank++;
        c->rank_distinct = dnagpu_hist_distinct(c->hist[c->rank]);
        c->next = c->win_first = c->win_count = 0;
    }
    if (c->next >= c->win_first + c->win_count) {
        uint64_t n = c->rank_distinct - c->next < CK_WINDOW ? c->rank_distinct - c->next : CK_WINDOW;
        if (!gpu_ok(dnagpu_hist_download(c->hctx[c->rank], c->hist[c->rank], c->next, n, c->keys, c->counts)))
            return false;
        c->win_first = c->next;
        c->win_count = n;
    }
    kmer->length = c->k;
    kmer->bit_sequence = c->keys[c->next - c->win_first];
    *count = (int64_t)c->counts[c->next - c->win_first];
    c->next++;
    return true;
}

void count_kmers_totals(const CountKmers *c, int64_t *total, int64_t *distinct, int64_t *unique)
{
    if (total) *total = (int64_t)c->total;
    if (distinct) *distinct = (int64_t)c->distinct;
    if (unique) *unique = (int64_t)c->unique;
}

void count_kmers_end(CountKmers *c)
{
    if (!c)
        return;
    for (int r = 0; r < c->n_ranks; r++)
        if (c->hist[r])
            dnagpu_hist_free(c->hctx[r], c->hist[r]);
    free(c->keys);
    free(c->counts);
    free(c);
}
