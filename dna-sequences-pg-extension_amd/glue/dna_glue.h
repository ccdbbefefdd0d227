/*
 * dna_glue.h -- host-side mirror of the reference's SQL-visible functions on the k-mer path.
 *
 * This is what the fmgr glue of dna.c looks like once its per-base loops call include/dnagpu.h:
 * the same function names, argument meaning and error text as the reference, with PostgreSQL's
 * plumbing (palloc, ereport/longjmp, the value-per-call SRF protocol, varlena headers) replaced by
 * plain C so that it builds and is testable without a PostgreSQL tree.  INTEGRATION.md maps every
 * function here to the PG_FUNCTION it stands for.
 *
 * Per-datum scalar operators (kmer_eq, starts_with, contains, kmer_hash on ONE value) stay host
 * code exactly as in the reference -- they are O(1) and run once per row inside the executor.
 * Everything that loops over a sequence (generate_kmers, its fused WHERE forms, the GROUP BY count)
 * runs on the GPU through libdnagpu.so; there is no CPU implementation of those here.
 *
 * Errors: a failing call returns NULL / false / a negative value and leaves the text the reference
 * would have passed to ereport(ERROR, errmsg(...)) in dna_glue_errmsg().
 */
#ifndef DNA_GLUE_H
#define DNA_GLUE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dna.c:42-47 without the varlena header; `dev` caches the device-resident copy */
typedef struct Dna {
    uint64_t length;            /* nucleotides */
    uint64_t *bit_sequence;     /* ceil(length/32) words, 2 bits per base, LSB first */
    void *dev;                  /* dnagpu_dna*, created on first GPU use */
} Dna;

/* dna.c:61-65 */
typedef struct Kmer {
    int32_t length;
    uint64_t bit_sequence;
} Kmer;

/* dna.c:81-84 */
typedef struct Qkmer {
    char sequence[33];
} Qkmer;

const char *dna_glue_errmsg(void);      /* text of the last "ereport(ERROR)" on this thread */
/* Device used by this process (default 0); must be called before the first GPU call. */
void dna_glue_set_device(int device);
/* GPUs count_kmers shards over (default 1 = the single device above): ranks on HIP devices devices[0..n)
 * (NULL = 0..n-1; a device may repeat, which rehearses the path on a one-GPU box) and the exchange
 * transport (DNAGPU_MULTI_AUTO / _RCCL / _COPY of include/dnagpu.h).  Takes effect at the next count. */
void dna_glue_set_gpus(int n_gpus, const int *devices, int transport);
/* Releases the GPU context (end of backend). */
void dna_glue_shutdown(void);

/* ---- type input/output (dna.c:220-242, 528-546, 932-951) ---- */
Dna *dna_in(const char *str);
char *dna_out(const Dna *dna);          /* malloc'd */
void dna_free(Dna *dna);
uint64_t dna_length(const Dna *dna);    /* length(dna), dna.c:366-384 */
bool kmer_in(const char *str, Kmer *out);
char *kmer_out(const Kmer *kmer);       /* malloc'd */
bool qkmer_in(const char *str, Qkmer *out);

/* ---- binary I/O (dna_recv/dna_send dna.c:244-291, kmer_recv/kmer_send dna.c:552-597) ----
 * The wire image the reference describes -- length, then the packed words through pq_sendint64 --
 * with a length field that works (the reference's pq_sendint(.., 8) is rejected by PostgreSQL).
 * dna: int64 length + ceil(length/32) int64 words, network byte order; kmer: int32 length + int64. */
unsigned char *dna_send(const Dna *dna, size_t *wire_bytes);       /* malloc'd */
Dna *dna_recv(const unsigned char *wire, size_t wire_bytes);
void kmer_send(const Kmer *kmer, unsigned char wire[12]);
bool kmer_recv(const unsigned char wire[12], Kmer *out);           /* ERROR "Invalid K-mer length: must be between 1 and 32" */

/* ---- per-datum operators (host, as in the reference) ---- */
bool kmer_eq(const Kmer *a, const Kmer *b);                         /* dna.c:686-696  `=`  */
bool kmer_ne(const Kmer *a, const Kmer *b);                         /* dna.c:708-720  `<>` */
int32_t kmer_hash(const Kmer *kmer);                                /* dna.c:722-735       */
int starts_with(const Kmer *kmer, const Kmer *prefix);              /* dna.c:842-866  `^@`: 1/0, -1 = ERROR */
int contains(const Qkmer *pattern, const Kmer *kmer);               /* dna.c:1091-1135 `@>`: 1/0, -1 = ERROR */

/* ---- generate_kmers(dna, k) RETURNS SETOF kmer (dna.c:743-837) ----
 * begin = the SRF_IS_FIRSTCALL block (validates k, dna.c:771-773); next = one SRF_RETURN_NEXT
 * (false = SRF_RETURN_DONE).  Rows are produced on the GPU in windows and served from a host
 * buffer, one per call, in position order. */
typedef struct GenerateKmers GenerateKmers;
GenerateKmers *generate_kmers_begin(Dna *dna, int k);
bool generate_kmers_next(GenerateKmers *g, Kmer *out);
/* true when generate_kmers_next stopped because an ERROR was raised (text in dna_glue_errmsg) */
bool generate_kmers_failed(const GenerateKmers *g);
void generate_kmers_end(GenerateKmers *g);

/* generate_kmers(dna,k) AS k(kmer) WHERE <op>: the filter is evaluated inside the extraction kernel.
 * op: '=' (rhs kmer), '^' (kmer ^@ rhs kmer), '@' (rhs_pattern @> kmer).  Same rows, same order and
 * same ERRORs as evaluating the operator row by row (test.sql:61-92). */
GenerateKmers *generate_kmers_where_begin(Dna *dna, int k, char op, const Kmer *rhs, const Qkmer *rhs_pattern);

/* ---- SELECT kmer, count(*) FROM generate_kmers(dna,k) GROUP BY kmer (test.sql:95-119) as one
 * set-returning call: count_kmers(dna, k) RETURNS SETOF (kmer, bigint) ---- */
typedef struct CountKmers CountKmers;
CountKmers *count_kmers_begin(Dna *dna, int k);
bool count_kmers_next(CountKmers *c, Kmer *kmer, int64_t *count);
/* sum(count), count(*), count(*) FILTER (WHERE count = 1) over the groups (test.sql:112-114) */
void count_kmers_totals(const CountKmers *c, int64_t *total, int64_t *distinct, int64_t *unique);
void count_kmers_end(CountKmers *c);

#ifdef __cplusplus
}
#endif
#endif
