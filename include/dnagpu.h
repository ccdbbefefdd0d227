/*
 * dnagpu.h -- C-ABI of the MI355X (gfx950) k-mer back-end.
 *
 * This is the drop-in boundary for the k-mer generation / filtering / counting path of the
 * PostgreSQL extension sid2364/dna-sequences-pg-extension.  The extension's fmgr glue (dna.c)
 * stays as it is; where its functions loop over bases on the CPU they call these entry points
 * instead (INTEGRATION.md shows the glue).  Plain C: pointers, sizes, status codes.  No
 * exceptions and no longjmp ever cross this boundary; no PostgreSQL, torch or C++ types appear.
 *
 * Data formats are the reference's own:
 *   packed dna   uint64 words, base i in word i/32 at bits (2i mod 64)..+1, LSB first,
 *                A=00 T=01 C=10 G=11, tail bits zero        (dna.c:42-47, 114-128)
 *   kmer key     uint64, base i of the k-mer at bits 2i..2i+1 (dna.c:61-65, 397-420); the k-mer
 *                at position p is bits [2p, 2p+2k) of the packed stream
 *   qkmer        NUL-terminated IUPAC text, <= 32 chars      (dna.c:81-84, 876-900)
 *
 * Threading / process model: one context per process and device, created lazily AFTER fork (a
 * PostgreSQL backend must not inherit a HIP runtime from the postmaster); calls on one context
 * are synchronous with respect to the caller unless documented otherwise and not re-entrant.
 * Every pointer named dev_* is device memory on the context's GPU; every other pointer is host
 * memory owned by the caller.  The library never keeps a caller pointer after the call returns,
 * except dnagpu_dna_wrap (documented there).
 */
#ifndef DNAGPU_H
#define DNAGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: histograms of several parts (dnagpu_hist_parts: what dnagpu_count_multi_unordered returns with its default of three
 * bucket groups per owner -- dnagpu_hist_device_keys / _counts are NULL for those), the dnagpu_multi_* options, the
 * table-of-sequences count (dnagpu_count_kmers_batch, dnagpu_dna_set_sequences + dnagpu_count_kmers_table,
 * dnagpu_hist_merge) */
#define DNAGPU_ABI_VERSION 2

/* ---- status codes ------------------------------------------------------------------------
 * 1..3 are the reference's own ERROR conditions on this path; dnagpu_strerror() returns the
 * reference's exact message text for them so the glue can ereport() it unchanged. */
#define DNAGPU_OK                       0
#define DNAGPU_ERR_INVALID_K            1   /* dna.c:772-773 "Invalid k value: must be between 1 and 32" */
#define DNAGPU_ERR_QKMER_LEN_MISMATCH   2   /* dna.c:1106-1108 "Qkmer pattern and kmer lengths do not match" */
#define DNAGPU_ERR_PREFIX_TOO_LONG      3   /* dna.c:854-856 "Prefix length cannot exceed kmer length" */
#define DNAGPU_ERR_QKMER_INVALID        4   /* dna.c:876-900 pattern empty / > 32 / non-IUPAC character */
#define DNAGPU_ERR_BAD_ARG              5   /* NULL pointer, range outside the sequence, bad filter kind */
#define DNAGPU_ERR_TOO_LARGE            6   /* more than 2^32 - 1 k-mers in one call */
#define DNAGPU_ERR_NO_DEVICE            7   /* no usable gfx950 device / HIP runtime unavailable */
#define DNAGPU_ERR_OOM                  8   /* device or host allocation failed */
#define DNAGPU_ERR_HIP                  9   /* a HIP call or kernel failed; see dnagpu_last_error() */
#define DNAGPU_ERR_INTERNAL             10
#define DNAGPU_ERR_DNA_EMPTY            11  /* dna.c:160-161 "DNA sequence cannot be empty" */
#define DNAGPU_ERR_DNA_INVALID_CHAR     12  /* dna.c:165-166 "Invalid character in DNA sequence: %c" */

const char *dnagpu_strerror(int status);
/* detail text of the most recent failing call on the calling thread (HIP error string, etc.) */
const char *dnagpu_last_error(void);
int dnagpu_abi_version(void);
/* Number of HIP devices this process can use (0 when there is none or the runtime cannot start).  Does not create a
 * context. */
int dnagpu_device_count(void);

/* ---- context ------------------------------------------------------------------------------ */
typedef struct dnagpu_ctx dnagpu_ctx;

/* Creates a context on HIP device `device` (0-based).  Lazy, post-fork.  Owns one HIP stream and
 * a pool of device work buffers that are recycled between calls. */
int dnagpu_init(int device, dnagpu_ctx **out_ctx);
void dnagpu_destroy(dnagpu_ctx *ctx);
/* Blocks until all work queued on the context's stream has finished. */
int dnagpu_synchronize(dnagpu_ctx *ctx);
/* The context's hipStream_t (as void*), for callers that enqueue their own HIP work around it. */
void *dnagpu_stream(dnagpu_ctx *ctx);
/* Releases every pooled device buffer that is not in use. */
int dnagpu_trim(dnagpu_ctx *ctx);
/* Bytes of device memory currently held by the context (in use + pooled). */
uint64_t dnagpu_device_bytes(dnagpu_ctx *ctx);

/* ---- device-resident dna (the detoasted Dna* of dna.c:768, moved to HBM once) --------------- */
typedef struct dnagpu_dna dnagpu_dna;

/* Copies ceil(n_bases/32) packed words host -> device.  Replaces nothing in the reference by
 * itself: it is the hand-off of dna->bit_sequence / dna->length (dna.c:45-46). */
int dnagpu_dna_upload(dnagpu_ctx *ctx, const uint64_t *words, uint64_t n_bases, dnagpu_dna **out);
/* Wraps n_words packed words already in device memory (no copy; the caller keeps them alive and
 * unchanged until dnagpu_dna_free).  n_words >= ceil(n_bases/32). */
int dnagpu_dna_wrap(dnagpu_ctx *ctx, const uint64_t *dev_words, uint64_t n_words, uint64_t n_bases,
                    dnagpu_dna **out);
/* Synthetic sequence generated on the device: word w = splitmix64(seed + w) (i.i.d. uniform bases,
 * the distribution of data/create_dna.py:27-33), tail bits zero.  motif_len > 0 gives the
 * repeat-rich variant: the second half of the sequence tiles the first motif_len bases. */
int dnagpu_dna_synth(dnagpu_ctx *ctx, uint64_t seed, uint64_t n_bases, uint64_t motif_len,
                     dnagpu_dna **out);
int dnagpu_dna_download(dnagpu_ctx *ctx, const dnagpu_dna *dna, uint64_t *words);
uint64_t dnagpu_dna_length(const dnagpu_dna *dna);
const uint64_t *dnagpu_dna_device_words(const dnagpu_dna *dna);
void dnagpu_dna_free(dnagpu_ctx *ctx, dnagpu_dna *dna);

/* ---- text <-> packed forms on the device (the steps either side of the path) ----------------
 * dna_in's work, validate_dna_sequence + encode_dna (dna.c:159-171, 114-128): n_bases characters
 * (host memory, or device memory when text_on_device != 0; no terminating NUL needed) become a
 * device-resident dna.  An empty text is DNAGPU_ERR_DNA_EMPTY; a character other than A/T/C/G is
 * DNAGPU_ERR_DNA_INVALID_CHAR with *bad_pos / *bad_char (either may be NULL) = the FIRST offending
 * character, the one the reference's message names. */
int dnagpu_dna_pack(dnagpu_ctx *ctx, const char *text, uint64_t n_bases, int text_on_device,
                    dnagpu_dna **out, uint64_t *bad_pos, char *bad_char);
/* decode_dna (dna.c:135-152): bases [first, first+count) as `count` characters (no NUL). */
int dnagpu_dna_unpack(dnagpu_ctx *ctx, const dnagpu_dna *dna, uint64_t first, uint64_t count,
                      char *out_text, int out_on_device);
/* ---- binary wire image: dna_recv / dna_send (dna.c:244-268, 270-291) ---------------------------
 * Wire image of a dna: int64 length in bases, then ceil(length/32) packed words, every field in
 * network byte order (what pq_sendint64 writes and pq_getmsgint64 reads).  The reference moves its
 * length field with pq_getmsgint / pq_sendint of size 8, which PostgreSQL rejects ("unsupported
 * integer size 8"), so its binary I/O cannot work as written; this is the format it describes, with
 * a length field that does.  from_wire: wire_bytes must equal dnagpu_dna_wire_size(length) else
 * DNAGPU_ERR_BAD_ARG; length 0 is DNAGPU_ERR_DNA_EMPTY (the type has no empty value, dna.c:160-161);
 * bits behind the last base are cleared (dna.c:186).  A device-side wire buffer must be 8-byte
 * aligned. */
uint64_t dnagpu_dna_wire_size(uint64_t n_bases);
int dnagpu_dna_from_wire(dnagpu_ctx *ctx, const void *wire, uint64_t wire_bytes, int wire_on_device,
                         dnagpu_dna **out);
int dnagpu_dna_to_wire(dnagpu_ctx *ctx, const dnagpu_dna *dna, void *wire, uint64_t wire_cap,
                       int wire_on_device);

/* decode_kmer / kmer_out (dna.c:428-452, 538-546) for n keys of k bases: n records of k characters
 * followed by a NUL (record stride k+1).  keys and out_text both host, or both device. */
int dnagpu_kmers_to_text(dnagpu_ctx *ctx, const uint64_t *keys, uint64_t n, int k, char *out_text,
                         int on_device);

/* ---- generate_kmers(dna, k)  (dna.c:743-837, dna--1.0.sql:188-191) -------------------------- */

/* Row count of generate_kmers: n_bases - k + 1 (dna.c:781), 0 when n_bases < k (the reference
 * underflows there).  DNAGPU_ERR_INVALID_K unless 1 <= k <= 32 (dna.c:771-773). */
int dnagpu_kmer_count(uint64_t n_bases, int k, uint64_t *n_kmers);

/* Writes the keys of rows [first, first+count) of generate_kmers(dna, k), in position order,
 * duplicates kept, to out_keys (host memory, or device memory when out_on_device != 0).
 * Replaces the per-row body dna.c:803-825 (decode to text, kmer_make, encode). */
int dnagpu_generate_kmers(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k,
                          uint64_t first, uint64_t count, uint64_t *out_keys, int out_on_device);

/* ---- generate_kmers fused with a WHERE operator --------------------------------------------- */
#define DNAGPU_FILTER_EQUALS       1   /* kmer = q         kmer_eq,     dna.c:655-668, 686-696 */
#define DNAGPU_FILTER_STARTS_WITH  2   /* kmer ^@ prefix   starts_with, dna.c:842-866           */
#define DNAGPU_FILTER_CONTAINS     3   /* qkmer @> kmer    contains,    dna.c:1064-1135         */

typedef struct dnagpu_filter {
    int32_t  kind;          /* DNAGPU_FILTER_*                                                   */
    int32_t  length;        /* EQUALS / STARTS_WITH: length (bases) of the right-hand kmer       */
    uint64_t bits;          /* EQUALS / STARTS_WITH: its bit_sequence                            */
    char     pattern[36];   /* CONTAINS: the qkmer text, NUL-terminated                          */
    int32_t  reserved;
} dnagpu_filter;

/* Rows of generate_kmers(dna,k) restricted to [first, first+count) that satisfy `filter`, in
 * position order (the reference's row order, test.sql:86-92).  Keys go to out_keys and their
 * positions to out_pos (either may be NULL), at most `cap` of each; *n_out receives the total
 * number of matching rows even when it exceeds cap.  Errors exactly where the reference's operator
 * raises them on the first row: CONTAINS with strlen(pattern) != k -> QKMER_LEN_MISMATCH,
 * STARTS_WITH with length > k -> PREFIX_TOO_LONG (only when there is at least one row).  An EQUALS
 * filter of another length matches nothing.  A 32-base prefix compares all 64 bits (the
 * reference's shift-by-64 at dna.c:862 is undefined behaviour). */
int dnagpu_generate_kmers_filtered(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k,
                                   const dnagpu_filter *filter, uint64_t first, uint64_t count,
                                   uint64_t *out_keys, uint64_t *out_pos, uint64_t cap,
                                   uint64_t *n_out, int out_on_device);

/* ---- GROUP BY kmer, count(*)  (kmer_hash dna.c:722-735 + kmer_eq dna.c:686-696 drive
 * PostgreSQL's HashAggregate in the reference; test.sql:95-119) ------------------------------ */
typedef struct dnagpu_hist dnagpu_hist;

/* Groups rows [first, first+count) of generate_kmers(dna,k) by key.  The result lives in device
 * memory: n_distinct (key, count) pairs in two dense arrays (uint64 keys, uint32 counts).  Group order in those arrays is
 * unspecified, as it is in PostgreSQL (the arrays hold key-range segments in the order the GPU
 * finished them, keys ascending inside a segment); dnagpu_hist_download serves the groups in
 * ascending key order through the segment directory.  At most 2^32-1 rows per call. */
int dnagpu_count_kmers(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k,
                       uint64_t first, uint64_t count, dnagpu_hist **out);
/* The same groups when the caller does not need them in key order -- PostgreSQL's own GROUP BY order is
 * unspecified (test.sql:95-104), so this is what the SQL entry point calls.  Long k-mers (k >= 20) of long
 * sequences (2^25 rows or more; k = 20, 21 and 22: where that is the faster engine, up to 2^28 / 2^29 / 2^31 rows) are then
 * partitioned as super-k-mers (runs of consecutive k-mers sharing a minimizer, 16 bytes per
 * ~9 k-mers) instead of 8-byte keys: less than a third of the partition traffic.  dnagpu_hist_download and
 * dnagpu_hist_sorted_view then serve the groups bucket by bucket (keys ascending inside a bucket only);
 * dnagpu_hist_is_sorted tells which kind a histogram is. */
int dnagpu_count_kmers_unordered(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k,
                                 uint64_t first, uint64_t count, dnagpu_hist **out);
int dnagpu_hist_is_sorted(const dnagpu_hist *h);
/* GROUP BY kmer, count(*) over a TABLE of sequences -- the reference's second counting shape,
 *   SELECT k.kmer, count(*) FROM dna_sequences d, LATERAL generate_kmers(d.sequence, K) AS k(kmer) GROUP BY k.kmer
 * (test.sql:140-150: one generate_kmers call per row of the table, dna.c:743-837; data/create_dna.py:36-49 writes such
 * tables of read-like rows).  The table travels as ONE packed stream `dna` (the sequences' bases back to back, no
 * padding) and seq_starts[0 .. n_seqs] (host memory): the first base of every sequence, seq_starts[0] = 0,
 * seq_starts[n_seqs] = dnagpu_dna_length(dna), ascending (equal neighbours: an empty sequence).  The rows are those of
 * every sequence's own generate_kmers: no k-mer spans two sequences, a sequence shorter than k has none (the 64-bit
 * restatement of dna.c:781).  Same result object as dnagpu_count_kmers_unordered (group order unspecified;
 * dnagpu_hist_is_sorted tells); dnagpu_hist_total = the rows of the table.  At most 2^32 - 1 bases per call: callers
 * that stream larger tables count batch by batch and add the histograms up (dnagpu_hist_merge). */
int dnagpu_count_kmers_batch(dnagpu_ctx *ctx, const dnagpu_dna *dna, const uint64_t *seq_starts, uint64_t n_seqs,
                             int k, dnagpu_hist **out);
/* The same table kept RESIDENT: dnagpu_dna_set_sequences validates seq_starts (as above), uploads them and builds the
 * boundary marks once, beside the packed stream (replacing an earlier set; n_seqs = 0 over an empty stream clears it;
 * released with dnagpu_dna_free); dnagpu_count_kmers_table then counts the table for any k with nothing crossing the bus
 * -- a glue that caches the packed table across queries pays the boundaries' upload (8 bytes per sequence: 80 MB and
 * 5.8 ms of a 14.4 ms call at 10^7 reads) once instead of per query.  dnagpu_dna_sequences: the sequences set, 0 if none.
 * dnagpu_count_kmers_table without a set (and a non-empty stream): DNAGPU_ERR_BAD_ARG. */
int dnagpu_dna_set_sequences(dnagpu_ctx *ctx, dnagpu_dna *dna, const uint64_t *seq_starts, uint64_t n_seqs);
uint64_t dnagpu_dna_sequences(const dnagpu_dna *dna);
int dnagpu_count_kmers_table(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, dnagpu_hist **out);
/* Same over an arbitrary array of n keys of k bases already in device memory.  dev_keys is used as
 * scratch and its contents are unspecified afterwards. */
int dnagpu_count_keys(dnagpu_ctx *ctx, uint64_t *dev_keys, uint64_t n, int k, dnagpu_hist **out);
/* Same, when the caller knows that every key lies in [key_min, key_max] (an owner's key range after
 * the multi-GPU exchange): the bits the two bounds share are not partitioned on again. */
int dnagpu_count_keys_in_range(dnagpu_ctx *ctx, uint64_t *dev_keys, uint64_t n, int k,
                               uint64_t key_min, uint64_t key_max, dnagpu_hist **out);

uint64_t dnagpu_hist_distinct(const dnagpu_hist *h);   /* count(*) over groups                   */
uint64_t dnagpu_hist_total(const dnagpu_hist *h);      /* sum(count)                              */
/* The groups in device memory, in the order they were produced (NOT ascending: see dnagpu_hist_download /
 * dnagpu_hist_sorted_view for that).  An ordered histogram stores its dnagpu_hist_distinct groups back to back.  An
 * unordered one (dnagpu_hist_is_sorted == 0) stores them bucket by bucket over dnagpu_hist_extent slots, and a
 * bucket that held copies of a k-mer is followed by padding slots whose COUNT IS 0 (their keys are undefined): skip
 * them. */
const uint64_t *dnagpu_hist_device_keys(const dnagpu_hist *h);
uint64_t dnagpu_hist_extent(const dnagpu_hist *h);     /* slots of the two device arrays in use (>= distinct) */
/* counts are 32-bit in device memory (one call covers at most 2^32 - 1 rows); dnagpu_hist_download
 * widens them to the 64-bit count(*) of SQL */
const uint32_t *dnagpu_hist_device_counts(const dnagpu_hist *h);
/* A histogram may consist of several PARTS, each with device arrays of its own (the pipelined multi-GPU count makes
 * them: one per bucket group of an owner).  dnagpu_hist_parts is 1 for every other histogram.  For a histogram of
 * several parts dnagpu_hist_device_keys / _counts are NULL -- walk the parts instead (dnagpu_hist_part(h, i): a
 * borrowed view, freed with h); distinct / total / extent / summary / download / sorted_view cover all parts, the
 * groups of part i before those of part i + 1. */
uint32_t dnagpu_hist_parts(const dnagpu_hist *h);
const dnagpu_hist *dnagpu_hist_part(const dnagpu_hist *h, uint32_t i);
/* Copies groups [first, first+count) of the ASCENDING-KEY order to host arrays (either may be
 * NULL). */
int dnagpu_hist_download(dnagpu_ctx *ctx, const dnagpu_hist *h, uint64_t first, uint64_t count,
                         uint64_t *keys, uint64_t *counts);
/* Same view into device memory: groups [first, first+count) of the ascending-key order as two dense
 * arrays of `count` uint64 each (either may be NULL), e.g. for a caller that post-processes on the GPU. */
int dnagpu_hist_sorted_view(dnagpu_ctx *ctx, const dnagpu_hist *h, uint64_t first, uint64_t count,
                            uint64_t *dev_keys, uint64_t *dev_counts);
/* total = sum(count), unique = count(*) FILTER (WHERE count = 1) (test.sql:112-114), checksum =
 * wrapping sum over groups of an order-independent digest of (key, count), computed on device. */
int dnagpu_hist_summary(dnagpu_ctx *ctx, const dnagpu_hist *h, uint64_t *total, uint64_t *unique,
                        uint64_t *checksum);
/* *out = the groups of a and b added up (equal keys: counts summed) -- what a caller that counts a large table batch by
 * batch (dnagpu_count_kmers_batch) does with the batches' histograms; also a PostgreSQL aggregate's combine step.  a and
 * b stay as they are; all three live on ctx's device.  The result is an unordered histogram of one part (dnagpu_hist_is_sorted
 * == 0; here the groups are in no order at all, also inside its one segment).  DNAGPU_ERR_TOO_LARGE when the totals pass
 * 2^32 - 1 (counts are 32-bit in device memory).  A utility (one random probe of a device-memory table per group), not
 * part of the streaming path. */
int dnagpu_hist_merge(dnagpu_ctx *ctx, const dnagpu_hist *a, const dnagpu_hist *b, dnagpu_hist **out);
void dnagpu_hist_free(dnagpu_ctx *ctx, dnagpu_hist *h);

/* ---- multi-GPU sharding of the count (one process per GPU; the exchange itself is the
 * caller's collective, e.g. RCCL all-to-all) -------------------------------------------------
 * Step 1 on every rank: the keys of rows [first, first+count) partitioned by owner.  Owner o of
 * n_owners owns the keys whose top `owner_bits` bits d satisfy (d * n_owners) >> owner_bits == o
 * (contiguous, ascending key ranges).  *dev_keys receives a device buffer of `count` keys grouped
 * by owner, owner_offsets[0..n_owners] the group boundaries.  Free with dnagpu_buffer_free.
 * Step 2 (caller): exchange the groups.  Step 3: dnagpu_count_keys on what was received; the
 * concatenation of the owners' downloads in owner order is the global result, keys ascending. */
/* Alternative without moving keys (the default of sharded.py): every rank holds the whole packed
 * sequence (an all-gather of 2 bits per base instead of an all-to-all of 64 bits per k-mer), scans
 * all of it and keeps only the keys it owns (same owner rule as above).  The histogram covers
 * exactly this owner's key range; dnagpu_hist_total() is the number of rows it owns. */
int dnagpu_count_kmers_owned(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, uint64_t first,
                             uint64_t count, int owner, int n_owners, dnagpu_hist **out);
int dnagpu_partition_kmers(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k,
                           uint64_t first, uint64_t count, int n_owners,
                           uint64_t **dev_keys, uint64_t *owner_offsets);
/* Device buffers from the context's pool.  Stream rule for every pooled buffer a caller touches with its
 * own HIP work (these, *dev_keys of dnagpu_partition_kmers, the arrays of a dnagpu_hist): the pool recycles a
 * block as soon as it is freed and orders reuse only on dnagpu_stream().  So (1) work a caller queues on
 * another stream must be ordered behind dnagpu_stream() before it first touches a buffer, and (2) must have
 * finished (or dnagpu_stream() must wait for it) before dnagpu_buffer_free / dnagpu_hist_free. */
int dnagpu_buffer_alloc(dnagpu_ctx *ctx, uint64_t bytes, void **dev_ptr);
void dnagpu_buffer_free(dnagpu_ctx *ctx, void *dev_ptr);
/* Copies between host memory and a device buffer on the context's stream and waits for the copy
 * (what the glue does with device-resident results before it builds Datums from them). */
int dnagpu_buffer_download(dnagpu_ctx *ctx, const void *dev_ptr, uint64_t bytes, void *host);
int dnagpu_buffer_upload(dnagpu_ctx *ctx, void *dev_ptr, const void *host, uint64_t bytes);

/* ---- the unordered count in two halves, for rows that live on several GPUs (the real exchange step
 * of the sharded path; dna-sequences-pg-extension_amd/sharded.py: count_sharded_exchange_records) -------------------
 * The super-k-mer engine (dnagpu_count_kmers_unordered) cuts the rows into 16-byte RECORDS -- runs of consecutive
 * k-mers that share their minimizer, ~9 k-mers each at k = 31 -- and sends every record to a coarse bucket that
 * depends on the k-mers' content alone.  A rank therefore cuts only its OWN rows (dnagpu_sk_records), ships every
 * bucket to its owner (1.8 bytes per k-mer instead of 8; the caller's all-to-all), and counts what it receives
 * (dnagpu_count_records): equal k-mers meet because they share the bucket.  global_rows = the rows of the whole
 * count: every rank must pass the same value (it fixes the bucket geometry, and the digits are part of the
 * records).  k in [20, 32] (minimizers of 15 bases for k >= 23, of 13 for k = 21 and 22, of 12 for k = 20). */
typedef struct dnagpu_records dnagpu_records;
/* number of coarse buckets of a count of global_rows rows (0: arguments out of range) */
int dnagpu_sk_buckets(const dnagpu_ctx *ctx, uint64_t global_rows, int k);
/* records of rows [first, first+count) of generate_kmers(dna, k), grouped by bucket in device memory */
int dnagpu_sk_records(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, uint64_t first, uint64_t count,
                      uint64_t global_rows, dnagpu_records **out);
uint32_t dnagpu_records_buckets(const dnagpu_records *r);
/* offsets[b] .. offsets[b+1] = bucket b's records (dnagpu_records_buckets(r) + 1 entries, host memory) */
int dnagpu_records_offsets(const dnagpu_records *r, uint64_t *offsets);
void *dnagpu_records_device(const dnagpu_records *r);      /* 16 bytes per record; pooled (stream rule above) */
void dnagpu_records_free(dnagpu_ctx *ctx, dnagpu_records *r);
/* GROUP BY kmer, count(*) over the k-mers of the given records: pieces[i] = piece_len[i] records of bucket
 * piece_bucket[i] in device memory (any order, several pieces per bucket allowed: one per sending rank); every
 * record of a bucket that exists anywhere must be among them, or equal k-mers are counted apart.  The histogram is
 * unordered (dnagpu_hist_is_sorted == 0); dnagpu_hist_total = the k-mers the records held.  The pieces are copied
 * before the call returns. */
int dnagpu_count_records(dnagpu_ctx *ctx, const void *const *pieces, const uint64_t *piece_len,
                         const uint32_t *piece_bucket, uint32_t n_pieces, int k, uint64_t global_rows,
                         dnagpu_hist **out);

/* ---- multi-GPU count from one process (the call a PostgreSQL backend's glue makes; SURVEY.md 8(b)
 * dnagpu_count_multi) ----------------------------------------------------------------------------------
 * A dnagpu_multi holds one context per rank (rank r on HIP device devices[r]; devices == NULL means 0..n-1;
 * the same device may appear more than once, which is how the path is rehearsed on a one-GPU box) and the
 * exchange transport: DNAGPU_MULTI_RCCL = ncclCommInitAll + ncclAllGather over xGMI (librccl.so is loaded on
 * demand; needs distinct devices), DNAGPU_MULTI_COPY = peer copies, DNAGPU_MULTI_AUTO = RCCL when n_gpus > 1
 * distinct devices and the library loads, else copies. */
typedef struct dnagpu_multi dnagpu_multi;
#define DNAGPU_MULTI_AUTO 0
#define DNAGPU_MULTI_RCCL 1
#define DNAGPU_MULTI_COPY 2
int dnagpu_multi_init(const int *devices, int n_gpus, int transport, dnagpu_multi **out);
void dnagpu_multi_destroy(dnagpu_multi *m);
int dnagpu_multi_size(const dnagpu_multi *m);
dnagpu_ctx *dnagpu_multi_ctx(dnagpu_multi *m, int rank);
/* "rccl" when the communicator exists (ncclCommInitAll succeeded: the ordered paths' all-gather / reduce use it, and the
 * record exchange can, see DNAGPU_MULTI_OPT_EXCHANGE_RCCL), else "copy".  This says what is AVAILABLE; what a count actually
 * moved its data with is dnagpu_multi_exchange_transport. */
const char *dnagpu_multi_transport(const dnagpu_multi *m);
/* How the most recent dnagpu_count_multi / dnagpu_count_multi_unordered call on m moved data between ranks: "peer-copy"
 * (hipMemcpyPeerAsync pulls; plain device copies when ranks share a device), "rccl-sendrecv" (the record exchange through
 * ncclSend / ncclRecv), "rccl-allgather" / "rccl-reduce" (the ordered paths), "none" (one rank, or no call yet). */
const char *dnagpu_multi_exchange_transport(const dnagpu_multi *m);
/* ranks of the RCCL communicator (0 when there is none) */
int dnagpu_multi_rccl_ranks(const dnagpu_multi *m);

/* The packed sequence sharded by contiguous word chunk: rank r is resident with words
 * [r*per, (r+1)*per), per = ceil(ceil(n_bases/32) / n).  upload: from the host words of dna->bit_sequence
 * (dna.c:45-46); synth: every rank generates its own chunk (dnagpu_dna_synth's sequence). */
typedef struct dnagpu_multi_dna dnagpu_multi_dna;
int dnagpu_multi_dna_upload(dnagpu_multi *m, const uint64_t *words, uint64_t n_bases, dnagpu_multi_dna **out);
int dnagpu_multi_dna_synth(dnagpu_multi *m, uint64_t seed, uint64_t n_bases, uint64_t motif_len,
                           dnagpu_multi_dna **out);
uint64_t dnagpu_multi_dna_length(const dnagpu_multi_dna *d);
void dnagpu_multi_dna_free(dnagpu_multi *m, dnagpu_multi_dna *d);

/* GROUP BY kmer, count(*) over rows [first, first+count) of generate_kmers(dna, k) on all ranks: one
 * all-gather of the packed chunks (2 bits per base), then rank r counts the keys it owns (owner rule of
 * dnagpu_count_kmers_owned) in its own host thread.  hists[r] (n entries, caller's array) = rank r's
 * histogram, living on rank r's device; the ranks' ascending downloads concatenated in rank order are the
 * global result in ascending key order, sum of dnagpu_hist_total = count.  Free each with
 * dnagpu_hist_free(dnagpu_multi_ctx(m, r), hists[r]).
 * Short k-mers (k <= 9 with enough rows, the single-GPU table path's rule) gather nothing: rank r counts the rows
 * that start in its own chunk into a table of 4^k counters (one word of halo from its neighbour), the tables
 * are summed onto rank 0 (ncclReduce, or peer copies + adds) and compacted there: hists[0] then holds the whole
 * result, the other ranks' histograms are empty -- the same concatenation property. */
int dnagpu_count_multi(dnagpu_multi *m, const dnagpu_multi_dna *dna, int k, uint64_t first, uint64_t count,
                       dnagpu_hist **hists);
/* The same groups with no order promise (PostgreSQL's GROUP BY makes none, test.sql:95-104), for long k-mers (k >= 20;
 * shorter ones go through dnagpu_count_multi): the record exchange above from one process.  Nothing is gathered:
 * rank r cuts the super-k-mer records of the rows that start in its own chunk, the owner of a coarse bucket pulls the
 * bucket's pieces from every rank (peer copies of 16-byte records, 1.8 B per k-mer at k = 31) and counts them.
 * The exchange is pipelined with the count: an owner's buckets travel in DNAGPU_MULTI_OPT_PARTS groups on a transfer
 * stream of their own and group g is counted while group g + 1 is still in flight.
 * hists[r] = the groups of rank r's buckets (a histogram of up to that many parts, dnagpu_hist_parts): disjoint between
 * ranks, dnagpu_hist_is_sorted == 0, sum of dnagpu_hist_total = count. */
int dnagpu_count_multi_unordered(dnagpu_multi *m, const dnagpu_multi_dna *dna, int k, uint64_t first, uint64_t count,
                                 dnagpu_hist **hists);

/* Options of a dnagpu_multi (dnagpu_multi_set_option):
 *   DNAGPU_MULTI_OPT_PARTS             bucket groups per owner of dnagpu_count_multi_unordered's pipelined exchange, 1 ..
 *                                      DNAGPU_MULTI_MAX_PARTS (1 = all pieces first, then one count)
 *   DNAGPU_MULTI_OPT_EMULATE_LINK_GBS  rehearsal aid for ranks that share one device: every group of inbound pieces is
 *                                      followed, on the transfer stream, by the time the same bytes would take at this
 *                                      many GB/s (0 = off, the default) -- timing only, results are unaffected
 *   DNAGPU_MULTI_OPT_PROBE_OWNER       TIMING PROBE, RESULTS ARE PARTIAL: only this owner (0 .. n-1) pulls and counts its
 *                                      buckets, the other ranks' histograms stay EMPTY (the sum of dnagpu_hist_total is then
 *                                      smaller than `count`), so that on a shared device the call's times are one owner's
 *                                      own (-1 = off, the default: every owner counts).  Never set it in a caller that uses
 *                                      the groups.
 *   DNAGPU_MULTI_OPT_EXCHANGE_RCCL     how dnagpu_count_multi_unordered moves the records between ranks: 0 (default) = every
 *                                      owner pulls its buckets' pieces with peer copies; 1 = every piece is one ncclSend on its
 *                                      rank's transfer stream and one ncclRecv on its owner's, one group call per bucket group
 *                                      (needs dnagpu_multi_transport() == "rccl", else DNAGPU_ERR_BAD_ARG); 2 = as 1, and a
 *                                      rank's own pieces go through RCCL too (so that one rank alone exercises the path) */
#define DNAGPU_MULTI_OPT_PARTS             1
#define DNAGPU_MULTI_OPT_EMULATE_LINK_GBS  2
#define DNAGPU_MULTI_OPT_PROBE_OWNER       3
#define DNAGPU_MULTI_OPT_EXCHANGE_RCCL     4
#define DNAGPU_MULTI_MAX_PARTS             8
#define DNAGPU_MULTI_DEFAULT_PARTS         3
int dnagpu_multi_set_option(dnagpu_multi *m, int option, double value);

/* Host wall-clock milliseconds of the most recent dnagpu_count_multi_unordered call on m, per phase, the slowest rank of
 * each phase: records_ms = every rank cutting the records of its own rows, exchange_ms = the owners' copies of their
 * buckets' pieces (until the last piece of the slowest owner has landed; the part of it that ran while the owner was
 * already counting earlier buckets is hidden_ms; both from HIP events on the owner's streams), count_ms = the owners'
 * phase as a whole (copies queued, groups counted as they land), total_ms = the whole call.  All 0 before the first call
 * or when the call took another path (short k-mers). */
typedef struct dnagpu_multi_times {
    double records_ms, exchange_ms, hidden_ms, count_ms, total_ms;
    uint64_t bytes_moved;     /* record bytes copied between ranks (pieces that stayed on their rank not counted) */
    int parts;                /* bucket groups per owner of the pipelined exchange */
} dnagpu_multi_times;
int dnagpu_multi_last_times(const dnagpu_multi *m, dnagpu_multi_times *out);

/* ---- batched operators over arrays of keys (bulk scans of stored kmer columns) -------------- */

/* kmer_hash (dna.c:722-735): PostgreSQL hash_any over the 8 bytes of bit_sequence. n keys in,
 * n uint32 out; both host, or both device when on_device != 0. */
int dnagpu_kmer_hash(dnagpu_ctx *ctx, const uint64_t *keys, uint64_t n, uint32_t *out, int on_device);
/* flags[i] = 1 when keys[i] (a kmer of k bases) satisfies `filter`, else 0.  Same error rules as
 * dnagpu_generate_kmers_filtered. */
int dnagpu_kmer_match(dnagpu_ctx *ctx, const uint64_t *keys, uint64_t n, int k,
                      const dnagpu_filter *filter, uint8_t *flags, int on_device);

/* ---- debug aids (off by default) ---------------------------------------------------------------
 * DNAGPU_DEBUG_POISON_POOL: every work buffer the pool hands out -- fresh or recycled, internal or through
 * dnagpu_buffer_alloc -- is first filled with 0xFF bytes on the context's stream, so a kernel that reads a
 * work buffer before writing it sees garbage in every run, not only in a warm context. */
#define DNAGPU_DEBUG_POISON_POOL 1u
/* DNAGPU_DEBUG_FORCE_SUPERKMER: dnagpu_count_kmers_unordered takes the super-k-mer engine for every k it supports
 * (20..32) and every sequence length, not only where it is the faster one (tests of its shorter windows). */
#define DNAGPU_DEBUG_FORCE_SUPERKMER 2u
/* DNAGPU_DEBUG_HEAVY_EXPAND: the super-k-mer engine expands its heavy mid buckets to keys for the ordinary levels (the
 * older path, still what records received from other ranks take when most of them are heavy) instead of splitting
 * them with the chunked level kernels. */
#define DNAGPU_DEBUG_HEAVY_EXPAND 4u
/* DNAGPU_DEBUG_GUARD_POOL: every work buffer the pool hands out is followed by a 256-byte guard band of 0xA5 bytes, placed
 * right behind the bytes that were asked for; dnagpu_synchronize() then checks every band (of buffers in use and of
 * buffers already returned) and fails with DNAGPU_ERR_INTERNAL -- dnagpu_last_error() names the buffer's size -- if a
 * kernel wrote past the end of one.  Complements DNAGPU_DEBUG_POISON_POOL (reads of memory a call never wrote). */
#define DNAGPU_DEBUG_GUARD_POOL 8u
/* Level 1 of the super-k-mer engine normally runs WITHOUT its histogram: every mid bucket is a region of the record buffer
 * sized from its parent (mean + 12.5 % + 72 records), the sweep reserves slots from cursors, and a region that overflows
 * (repeats) sends the level through the exact path (histogram, prefix, sweep).  DNAGPU_DEBUG_NO_SPEC1 always takes the exact
 * path; DNAGPU_DEBUG_SPEC1_OVERFLOW runs the speculative sweep and then treats it as overflowed (tests of the fall-back). */
#define DNAGPU_DEBUG_NO_SPEC1 16u
#define DNAGPU_DEBUG_SPEC1_OVERFLOW 32u
/* Level 0 likewise runs WITHOUT its histogram sweep on long sequences (from 2^29 rows): a histogram over 1/64 of the rows
 * sizes, per coarse bucket, the slots every chunk of rows reserves in the bucket's region; slots a chunk does not use
 * hold NULL records, which level 1 skips; a chunk that runs out of slots (or uneven buckets in the sample: repeats) sends
 * the level through the exact pair (histogram sweep, prefix, scatter sweep).  DNAGPU_DEBUG_SLAB0 takes the slab sweep for
 * every length, DNAGPU_DEBUG_NO_SLAB0 never, DNAGPU_DEBUG_SLAB0_OVERFLOW runs it and then treats it as overflowed. */
#define DNAGPU_DEBUG_SLAB0 64u
#define DNAGPU_DEBUG_NO_SLAB0 128u
#define DNAGPU_DEBUG_SLAB0_OVERFLOW 256u
/* Over uneven coarse buckets (repeats; the coarse buckets only show them on long sequences) the regions of level 1's
 * speculative sweep come from a sampled histogram -- one piece of 1024 records in every eight -- instead of the parents' sizes.
 * DNAGPU_DEBUG_SAMPLE1 takes the sampled regions whatever the coarse buckets look like (tests on short sequences). */
#define DNAGPU_DEBUG_SAMPLE1 512u
int dnagpu_set_debug(dnagpu_ctx *ctx, unsigned flags);

/* ---- instrumentation ------------------------------------------------------------------------
 * Device time in milliseconds (HIP events on the context's stream) of the phases of the most
 * recent dnagpu_count_kmers / dnagpu_count_keys call.  names[i] are static strings. */
#define DNAGPU_MAX_PHASES 48
typedef struct dnagpu_phase_times {
    int         n;
    const char *names[DNAGPU_MAX_PHASES];
    float       ms[DNAGPU_MAX_PHASES];
} dnagpu_phase_times;
int dnagpu_last_phase_times(dnagpu_ctx *ctx, dnagpu_phase_times *out);
/* The same for rank `rank` of the most recent dnagpu_count_multi_unordered on m: the phases of its record pass (level 0)
 * followed by those of its owner phase (a name appears once per bucket group there). */
int dnagpu_multi_last_phase_times(dnagpu_multi *m, int rank, dnagpu_phase_times *out);
/* Enables (1) / disables (0) per-phase event timing; off by default (it adds event records only). */
int dnagpu_set_profiling(dnagpu_ctx *ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif /* DNAGPU_H */
