// kernels.hpp -- host-callable launchers of the gfx950 kernels (internal; the public surface is
// include/dnagpu.h).  Every launcher enqueues on `stream` and returns hipGetLastError().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmer_device.hpp"

namespace dnagpu {

// ---------------------------------------------------------------- extract_kernels.hip
// words[w] of the synthetic sequence of n_bases bases for w in [w_begin, w_end) (words = the sequence's word 0)
hipError_t launch_synth(u64 *words, u64 w_begin, u64 w_end, u64 n_bases, u64 seed, u64 motif_len, hipStream_t s);
hipError_t launch_extract(const u64 *words, u64 n_words, u64 first, u64 count, int k, u64 *out_keys,
                          hipStream_t s);

// position-ordered filtered extraction (filter_kernels.hip), two sweeps over the (tiny) packed input:
//   count sweep -> matches per workgroup range (filter_bits_geometry groups, group_counts[g]);
//   write sweep -> every group sums the counts before it, then writes keys / positions (either may be
//   null) of rows whose output index is < cap; *total_out (may be null, may be host-mapped memory) receives
//   the number of matching rows
constexpr int FILTER_MAX_GROUPS = 8192;
void filter_bits_geometry(u64 count, u32 *n_groups, u32 *tiles_per_group);
hipError_t launch_filter_bits_count(const u64 *words, u64 n_words, u64 first, u64 count, const FilterBits &fb,
                                    u32 *group_counts, hipStream_t s);
hipError_t launch_filter_bits_write(const u64 *words, u64 n_words, u64 first, u64 count, int k, const FilterBits &fb,
                                    const u32 *group_counts, u64 *out_keys, u64 *out_pos, u64 cap, u64 *total_out,
                                    hipStream_t s);

hipError_t launch_hash_batch(const u64 *keys, u64 n, u32 *out, hipStream_t s);
hipError_t launch_match_batch(const u64 *keys, u64 n, const FilterDev &f, uint8_t *flags, hipStream_t s);
// result[0] += sum(count), result[1] += #(count == 1), result[2] += sum(pair_mix); zero it first
hipError_t launch_hist_summary(const u64 *keys, const u32 *counts, u64 n, u64 *result3, hipStream_t s);

// text <-> packed forms.  *bad_pos must hold ~0 before launch_pack; afterwards the position of the first
// character that is not A/T/C/G (or still ~0)
hipError_t launch_pack(const unsigned char *text, u64 n_bases, u64 *words, u64 *bad_pos, hipStream_t s);
hipError_t launch_unpack(const u64 *words, u64 first, u64 count, unsigned char *text, hipStream_t s);
hipError_t launch_kmers_to_text(const u64 *keys, u64 n, int k, unsigned char *text, hipStream_t s);
// dst[i] = bswap64(src[i]); the last word is ANDed with last_mask (wire <-> packed words)
hipError_t launch_wire_swap(const u64 *src, u64 *dst, u64 n_words, u64 last_mask, hipStream_t s);

// exclusive scan of n u32 values (in != out allowed, in == out allowed); *total receives the sum.
// tmp must hold scan_tmp_words(n) u32 values.
u64 scan_tmp_words(u64 n);
// up to four exclusive scans of the same length n in three launches: in[a] -> out[a] (in == out allowed), *total[a] = the
// sum (may be null); tmp holds n_arrays x scan_tmp_words(n) words
struct ScanSet {
    const u32 *in[4];
    u32 *out[4];
    u32 *total[4];
    u32 *tmp;
};
hipError_t launch_scan_u32_multi(const ScanSet &set, int n_arrays, u64 n, hipStream_t s);
hipError_t launch_scan_u32(const u32 *in, u32 *out, u64 n, u32 *tmp, u32 *total, hipStream_t s);

// ---------------------------------------------------------------- count_kernels.hip
// MSD radix tree over the keys (DESIGN.md "count"): every level splits the oversize nodes on their
// next most significant bits; leaves (<= LEAF_CAP keys, or no bits left) are sorted and run-length
// encoded in LDS, in key order, with a chained scan giving each leaf its output offset.
constexpr int LEAF_CAP = 6144;        // max keys a leaf workgroup sorts in LDS (48 KB of keys, six per thread)
constexpr int LEAF_CAP_TINY = 1024;   // leaves up to here take 256-thread workgroups (eight per CU)
constexpr int LEAF_CAP_SMALL = 4096;  // leaves up to here take the four-keys-per-thread kernel
constexpr int LEAF_TARGET = 5800;     // planned leaf size: a split aims at means in (2900, 5800]; a leaf that still exceeds LEAF_CAP is halved by one more level
constexpr int MAX_SPLIT_BITS = 10;    // widest digit of one level (1024 children)
constexpr int ROW_STRIDE = 1 << MAX_SPLIT_BITS;

struct Node {           // 32 bytes
    u32 start;          // index of the node's first key in its buffer (root over dna: 0)
    u32 len;            // keys in the node
    u32 meta;           // bits 0-7: remaining (not yet fixed) key bits; bit 8: buffer; bit 9: terminal
    u32 split;          // plan: digit width this level (0 = leaf)
    u64 prefix;         // the fixed high bits of every key in the node (low `rem` bits zero)
    u32 child_base;     // plan: index of the first child (or of the node itself) in the next list
    u32 chunk_base;     // plan: index of the node's first chunk
};
constexpr u32 NODE_BUF = 1u << 8;
constexpr u32 NODE_TERMINAL = 1u << 9;
constexpr u32 NODE_SKIP = 1u << 10;     // this level only: all keys share the digit, the scatter leaves the node where it is
constexpr u32 NODE_PEEL = 1u << 11;     // this level only: the node is split three ways around its dominant first key
// per node and key-source level (level_hist -> level_children -> peel_scatter): varying low bits, keys equal
// to the node's first key, cursors of the two moved parts of a peeled node
constexpr int NODE_STAT_WORDS = 8;

struct Chunk {          // 16 bytes: a contiguous piece of one node, the unit of work of a level
    u32 node;
    u32 off;            // offset inside the node
    u32 len;
    u32 pad;
};

struct LevelCounters {  // read back by the host once per level
    u32 n_next;         // nodes in the next list
    u32 n_chunks;
    u32 n_split;        // nodes being split this level
    u32 n_scatter;      // of which non-terminal (their keys move)
    u32 max_bits;       // widest split of this level
    u32 n_big;          // unsplit nodes that sort more than LEAF_CAP_SMALL keys (the six-keys-per-thread leaves kernel)
    u32 n_small;        // unsplit nodes that sort LEAF_CAP_TINY < keys <= LEAF_CAP_SMALL
    u32 n_tiny;         // unsplit nodes that sort at most LEAF_CAP_TINY keys; the rest are single-key or empty nodes
    u32 n_over;         // (levels >= 2) nodes over the leaf capacity, counted before the plan
};

hipError_t launch_plan(Node *nodes, u32 n_nodes, int level, u32 chunk_len, u32 *outc, u32 *nch,
                       LevelCounters *ctr, hipStream_t s, int skew_from = 2);
// plan + both scans + counters of a level (one launch for levels of at most 1024 nodes)
// skew_from: the first level at which an oversize node means skew and is fanned out wider than its size asks
hipError_t launch_plan_level(Node *nodes, u32 n_nodes, int level, u32 chunk_len, u32 *outc, u32 *nch, u32 *scan_tmp,
                             LevelCounters *ctr, hipStream_t s, int skew_from = 2);
hipError_t launch_fill_chunks(const Node *nodes, u32 n_nodes, u32 chunk_len, const u32 *child_base,
                              const u32 *chunk_base, Node *nodes_rw, Chunk *chunks, hipStream_t s);
// src_dna != 0: the (single) node being split is the root over the packed sequence; flt_lo/flt_span:
// owner filter on the root's digits ((d - lo) < span keeps a key; span = ~0 keeps all; tb > 0: lo/span are an
// aligned power-of-two range and only the digit's top tb bits need testing)
hipError_t launch_level_hist(const Node *nodes, const Chunk *chunks, u32 n_chunks, int src_dna,
                             const u64 *words, u64 n_words, u64 first, int k, const u64 *buf0,
                             const u64 *buf1, u32 *hist, u32 flt_lo, u32 flt_span, u32 flt_tb, u32 *vary, int multi_ref,
                             u32 stat_min_len, hipStream_t s);
// per split node and 64-digit group: chunk rows -> exclusive prefixes over chunks; per-digit totals
// into tot[row of the node's first chunk]
hipError_t launch_level_prefix(const Node *nodes, const Chunk *chunks, u32 n_chunks, u32 n_split_nodes, u32 chunk_len,
                               u32 *hist, u32 *tot, hipStream_t s, u32 n_nodes = 0);
// per node: leaf -> copied to the next list; split -> totals scanned over digits, children appended
// in digit (= key) order, tot row overwritten with each digit's absolute base
// vary (may be null): per node NODE_STAT_WORDS words -- the number of low key bits that vary inside it, keys below /
// equal to its first key -- as level_hist found (key-source levels); a node whose keys all share this level's digit is flagged NODE_SKIP and stays where it is
// (stat_min_len: the statistics exist for nodes of at least that many keys only)
hipError_t launch_level_children(Node *nodes, u32 n_nodes, u32 *tot, Node *next, u32 *vary, const u64 *buf0,
                                 const u64 *buf1, u32 stat_min_len, hipStream_t s);
// the chunks of nodes that level_children flagged NODE_PEEL (stat = the NODE_STAT_WORDS-per-node table)
hipError_t launch_peel_scatter(const Node *nodes, u32 n_nodes, const Chunk *chunks, u32 n_chunks, Node *next, u64 *buf0,
                               u64 *buf1, u32 *stat, hipStream_t s);
hipError_t launch_level_scatter(const Node *nodes, const Chunk *chunks, u32 n_chunks, int src_dna,
                                const u64 *words, u64 n_words, u64 first, int k, u64 *buf0, u64 *buf1,
                                const u32 *hist, const u32 *tot, u32 flt_lo, u32 flt_span, u32 flt_tb, u32 max_bits,
                                hipStream_t s);
// leaves -> (key, count) groups appended densely to out_keys/out_counts at offsets taken from *cursor
// (zeroed; holds the group count afterwards); seg_off/seg_cnt[l] = where leaf l landed.
// hashed: the groups of a leaf need no order -- leaves of at most LEAF_CAP_SMALL keys are counted in an LDS hash table
// (flags / scan_tmp / list are then always needed).
// n_tiny / n_small / n_big: leaves per sorting class; the other leaves hold one distinct key (or none) and are emitted
// in bulk; flags / scan_tmp / list (n_leaves + 1, scan_tmp_words(n_leaves) and n_leaves u32) are needed unless all
// nodes are sorting leaves of one class
hipError_t launch_leaves(const Node *leaves, u32 n_leaves, u32 n_tiny, u32 n_small, u32 n_big, const u64 *buf0, const u64 *buf1,
                         u64 *cursor, u64 *seg_off, u32 *seg_cnt, u64 *out_keys, u32 *out_counts, u32 *flags,
                         u32 *scan_tmp, u32 *list, hipStream_t s, bool hashed = false, u64 out_base = 0);
// (out_base: the value *cursor holds at the call -- the first output slot these leaves may use)
// dense count of short k-mers straight from the packed sequence (2k = bits <= dense_max_bits()): table must hold
// 2^bits u32 counters, out_keys/out_counts 2^bits entries; *n_out = distinct keys; results ascending
int dense_max_bits();
hipError_t launch_dense_count(const u64 *words, u64 n_words, u64 first, u64 count, int bits, u32 *table, u64 *out_keys,
                              u32 *out_counts, u64 *n_out, hipStream_t s);
// the two halves of it (a multi-GPU dense count sums the ranks' tables in between), and dst[i] += src[i]
hipError_t launch_dense_table(const u64 *words, u64 n_words, u64 first, u64 count, int bits, u32 *table, hipStream_t s);
hipError_t launch_dense_compact(const u32 *table, int bits, u64 *out_keys, u32 *out_counts, u64 *n_out, hipStream_t s);
hipError_t launch_table_add(u32 *dst, const u32 *src, u32 n, hipStream_t s);
// groups [first, first+count) of the ascending-key view -> dst arrays
hipError_t launch_gather_sorted(const u64 *seg_off, const u32 *seg_cnt, const u32 *seg_pre, u32 n_leaves,
                                u64 first, u64 count, const u64 *keys, const u32 *counts, u64 *dst_keys,
                                u64 *dst_counts, hipStream_t s);

// ---- dnagpu_hist_merge (extract_kernels.hip): groups (key, count; count 0 = padding) added into an open-addressing table of
// t_slots (a power of two) 8-byte keys (all ones = free) + 32-bit counts; the all-ones key's count goes to *ones.  Then the
// table's groups dense, in table order; *cursor (zeroed by the caller) ends as their number.
hipError_t launch_merge_insert(const u64 *keys, const u32 *counts, u64 n, u64 *tkeys, u32 *tcnt, u64 t_slots,
                               unsigned long long *ones, hipStream_t s);
hipError_t launch_merge_compact(const u64 *tkeys, const u32 *tcnt, u64 t_slots, u64 *out_keys, u32 *out_counts,
                                unsigned long long *cursor, hipStream_t s);

// ---- a table of sequences in one packed stream (extract_kernels.hip)
// marks: bit b set where a sequence starts at base b (starts[1 .. n_seqs - 1]; the buffer is zeroed here)
hipError_t launch_batch_marks(const u64 *starts, u64 n_seqs, u32 *marks, u64 n_mark_words, hipStream_t s);
// *rows = sum over the sequences of max(0, length - k + 1)
hipError_t launch_batch_rows(const u64 *starts, u64 n_seqs, int k, u64 *rows, hipStream_t s);
// the keys of the rows of the table -- windows [p, p + k) of rows [0, n_rows) that reach across no mark -- in no
// particular order; *cursor (zeroed here) ends as their number
hipError_t launch_batch_keys(const u64 *words, u64 n_words, const u32 *marks, u64 n_mark_words, u64 n_rows, int k, u64 *out_keys,
                             unsigned long long *cursor, hipStream_t s);

// ---------------------------------------------------------------- superkmer_kernels.hip
// super-k-mer (minimizer) partitioning for long k-mers: the dna sweeps (level 0: records per coarse digit of every
// chunk of rows / records scattered into the coarse buckets), the record level (level 1: d1), and the expansion of
// mid buckets into key nodes.  k in [sk_min_k(), 32]; c0n = coarse buckets, b1bits = bits of d1, r0bits = split
// bits of the root (2^r0bits >= c0n); records are 16 bytes each.
int sk_min_k();
int sk_minimizer_len(int k);          // m: 15 for k >= 23, 13 for k = 21, 22, 12 for k = 20
int sk_tile_rows();
int sk_max_c0();          // most coarse buckets the level-0 sweeps support
// aux: the histogram sweep adds its chunks' counts into ONE row aux[0 .. 2^r0bits) instead of hist rows (a sampled
// histogram); the scatter sweep takes aux as its slab table (sk_slab_words() words, launch_sk_slab_init): level 0 WITHOUT
// the histogram sweep -- every chunk reserves sk_slab_cap(est ...) slots of every digit's region, unused slots become NULL
// records (bit 63 of the second word; level 1 skips them), aux[17 * sk_max_c0() + d] = records stored per digit,
// aux[18 * sk_max_c0()] = 1 if a chunk ran out of slots, aux[18 * sk_max_c0() + 1] = slots of all regions.
// marks (or null): one bit per base of the packed stream, set where a sequence of a TABLE of sequences starts
// (dnagpu_count_kmers_batch): rows whose k-mer reaches across a start are rows of no sequence and make no record.
hipError_t launch_sk_level0(bool scatter, const Chunk *chunks, u32 n_chunks, const u64 *words, u64 n_words, u64 first, int k,
                            u32 c0n, u32 b1bits, u32 r0bits, u32 *hist, const u32 *tot, void *recs, hipStream_t s,
                            u32 *aux = nullptr, const u32 *marks = nullptr, u64 n_mark_words = 0);
hipError_t launch_sk_sample_chunks(Chunk *chunks, u32 n_chunks, u32 stride, u32 len, u32 n, hipStream_t s);
int sk_slab_words();
u32 sk_slab_cap(u32 est, u64 chunk_rows, u64 sampled);
hipError_t launch_sk_slab_init(const u32 *est, u32 r0bits, u32 chunk_rows, u32 sampled, u32 n_chunks, u32 *slab, Node *nodes,
                               hipStream_t s);
hipError_t launch_sk_hist1(const Node *nodes, const Chunk *chunks, u32 n_chunks, const void *recs, u32 *hist, u32 *kcount,
                           hipStream_t s, bool by_d2 = false);
hipError_t launch_sk_scatter1(const Node *nodes, const Chunk *chunks, u32 n_chunks, const void *src, void *dst, const u32 *hist,
                              const u32 *tot, hipStream_t s, bool by_d2 = false, u32 *gcur = nullptr);
// (by_d2: the same kernels on the 4-bit digit d2 -- the heavy mid buckets' split into final buckets)
// Level 1 WITHOUT its histogram (speculative): every mid bucket is a region of the destination buffer, sk_spec_span(len, bits)
// slots for a node of len records (host and device share the formula); spec = 2 words per node, out = 3 words (slots of all
// regions; 1 if they pass 2^32; the sweep's overflow flag), gcur = rows of ROW_STRIDE cursors (row = the node's first chunk).
// The sweep counts the k-mers per mid bucket into kcount; launch_sk_spec_nodes writes the mid nodes and raises *over too.
u64 sk_spec_span(u32 len, int bits);
// Over UNEVEN coarse buckets (repeats) the regions come from a sampled histogram instead: one piece of sk_sample1_len()
// records in every sk_sample1_every() pieces of a node's records (the host lists the pieces as chunks), rcap / rstart per mid
// bucket (launch_sk_sampled_regions), handed to the three launchers above as rstart / rcap.
hipError_t launch_sk_spec_regions(const Node *nodes, u32 n_nodes, u32 *spec, u32 *out, u32 *gcur, hipStream_t s,
                                  const u32 *rstart = nullptr);
hipError_t launch_sk_scatter1_spec(const Node *nodes, const Chunk *chunks, u32 n_chunks, const void *src, void *dst, u32 *gcur,
                                   const u32 *spec, u32 *kcount, u32 *over, hipStream_t s, const u32 *rstart = nullptr,
                                   const u32 *rcap = nullptr);
hipError_t launch_sk_spec_nodes(const Node *nodes, u32 n_nodes, const u32 *spec, const u32 *gcur, Node *next, u32 *over,
                                hipStream_t s, const u32 *rstart = nullptr, const u32 *rcap = nullptr);
u32 sk_sample1_len();
u32 sk_sample1_every();
hipError_t launch_sk_sampled_regions(const Node *nodes, const Chunk *samples, u32 n_samples, const void *recs, u32 n_mid, u32 *est,
                                     u32 *rcap, u32 *rstart, u32 *scan_tmp, u32 *total, hipStream_t s);
hipError_t launch_sk_heavy_finals(const Node *kids, u32 n, const u32 *kcount, Node *out, hipStream_t s);
// buckets sk_count does not take, expanded to keys in record order (no host step): slices of sk_flat_slice() records
// per bucket (n_slices[i], then after an exclusive scan slice_first[i]; slice_rec0 = first record, slice_nrec = records),
// k-mers per slice, after a scan the keys of slice s at key_base + slice_koff[s] and one key node per bucket
int sk_flat_slice();
// out[j] = mids[idx[j]] with child_base = kcount[idx[j]]; mids[idx[j]].len = 0
hipError_t launch_sk_take_heavy(Node *mids, const u32 *idx, u32 n_heavy, const u32 *kcount, Node *out, hipStream_t s);
hipError_t launch_sk_slice_count(const Node *buckets, u32 nb, u32 *n_slices, hipStream_t s);
hipError_t launch_sk_slice_fill(const Node *buckets, u32 nb, const u32 *slice_first, u32 *slice_rec0, u32 *slice_nrec, hipStream_t s);
hipError_t launch_sk_slice_kmers(const void *recs, const u32 *slice_rec0, const u32 *slice_nrec, u32 n_slices, u32 *slice_km,
                                 hipStream_t s);
hipError_t launch_sk_slice_nodes(const Node *buckets, u32 nb, const u32 *slice_first, const u32 *slice_koff, u32 n_slices,
                                 const u32 *total_keys, u32 key_base, int k, bool check_kmers, Node *out, u64 *n_bad, hipStream_t s);
// keys[first .. *end): key_unmix (the groups of the tree over the expansion's mixed keys; at most max_groups of them)
hipError_t launch_sk_unmix(u64 *keys, u64 first, const u64 *end, u64 max_groups, int k, hipStream_t s);
hipError_t launch_sk_expand_flat(const void *recs, const u32 *slice_rec0, const u32 *slice_nrec, const u32 *slice_koff, u32 key_base,
                                 u32 n_slices, int k, u64 *keys, hipStream_t s);
// every mid bucket's records regrouped by d2 from src into the same range of dst; out_nodes[16 i + j] = final bucket:
// start / len in records, child_base = its k-mers
int sk_regroup_tile();   // records one workgroup regroups in one go; longer mid buckets need any_long (a second kernel)
hipError_t launch_sk_regroup(const Node *mids, u32 n_mids, const void *src, void *dst, Node *out_nodes, bool any_long, hipStream_t s);
// final buckets list[0..n_list) (indices into fin) counted from their records in an LDS hash table; groups appended
// at *cursor, seg_off / seg_cnt[bucket] = where they went
// list_off[i] = first output slot of bucket list[i] (its range is as long as its k-mers; slots past its distinct keys are
// count-0 padding); *n_groups += the groups written
// (left: 2 n_list + 1 words of work memory -- the buckets the record test leaves to the k-mer table kernel)
hipError_t launch_sk_count(const Node *fin, const u32 *list, const u32 *list_off, u32 n_list, const void *recs, int k, u64 *n_groups,
                           u64 *seg_off, u32 *seg_cnt, u64 *out_keys, u32 *out_counts, u32 *left, hipStream_t s);

int sk_count_cap();       // most k-mers a final bucket may hold to be counted from its records
hipError_t launch_sk_select_flags(const Node *fin, u32 n_fin, u32 cap, u32 big_limit, u32 *f_small, u32 *f_big, u32 *k_range,
                                  u32 *k_big, hipStream_t s);
hipError_t launch_sk_select_lists(const Node *fin, u32 n_fin, u32 cap, u32 big_limit, const u32 *p_small, const u32 *p_big,
                                  const u32 *kb_range, u32 *list_small, u32 *off_small, u32 *list_big, u32 *off_big, hipStream_t s);
// buckets that go through the expansion: too many k-mers for sk_count_big, or given up by it (big_status[p] != 0)
hipError_t launch_sk_over_flags(const Node *fin, u32 n_fin, u32 cap, u32 big_limit, const u32 *p_big, const u32 *big_status,
                                u32 *f_over, u32 *k_over, hipStream_t s);
hipError_t launch_sk_over_list(const Node *fin, u32 n_fin, const u32 *f_raw, const u32 *p_over, const u32 *kb_over, Node *over_nodes,
                               u32 *over_kbase, hipStream_t s);
// long buckets of few distinct keys (repeats): one workgroup per bucket, one LDS table for the whole bucket; status[i] = 1
// where the distinct keys outgrew the table (the bucket's range is then all padding)
// slices of SKB records per big bucket: nsl[p] = its slices, msl[p] = the same if more than one (else 0); after exclusive
// scans: sl_bucket / sl_idx[first slice of p + j] = (p, j).  Partial areas of sk_big_partial_slots() groups each, one per
// slice of a sliced bucket (mfirst = scan of msl); part_n[area] = its groups (~0: the slice gave up).
hipError_t launch_sk_node_lens(const Node *nodes, u32 n, u32 *lens, hipStream_t s);
u32 sk_big_slice_records();
u64 sk_big_partial_slots();
hipError_t launch_sk_big_slices(const Node *fin, const u32 *list, u32 n_list, u32 *nsl, u32 *msl, hipStream_t s);
hipError_t launch_sk_big_slice_fill(const u32 *nsl_raw, const u32 *sfirst, u32 n_list, u32 *sl_bucket, u32 *sl_idx, hipStream_t s);
hipError_t launch_sk_count_big(const Node *fin, const u32 *list, const u32 *list_off, const u32 *nsl, const u32 *mfirst,
                               const u32 *sl_bucket, const u32 *sl_idx, u32 n_slices, u32 n_list, const void *recs, int k,
                               u64 *n_groups, u64 *seg_off, u32 *seg_cnt, u64 *out_keys, u32 *out_counts, u32 *status, u64 *part_keys,
                               u32 *part_cnts, u32 *part_n, bool any_sliced, hipStream_t s);

// scatter-only microbenchmark entry (bench tooling): one level over a key array
int scatter_tile_keys();
int scatter_threads();

}  // namespace dnagpu
