// superkmer_kernels.hip -- GROUP BY kmer, count(*) for long k-mers through super-k-mer (minimizer) partitioning.
//
// The MSD radix tree of count_kernels.hip moves every 8-byte key through two partition passes
// (3.5x the algorithmic traffic at 3 Gbase).  Here the partition passes move PACKED RUNS instead:
//
//   minimizer  of a k-mer = the m-mer (m = 15) among its w = k - m + 1 with the smallest 32-bit hash; a function
//              of the k-mer's content alone, so equal k-mers share it wherever they occur
//   bucket     = three digits (d0 < C0, d1 < 2^b1, d2 < 16) cut from a multiplicative re-mix of that hash
//   record     = a run of consecutive k-mers with the same minimizer hash (on random data 9 k-mers on
//              average, never more than SK_LMAX): 16 bytes = up to 54 bases (108 bits) + length + d1 + d2
//
//   sk_hist0 / sk_scatter0   sweep the packed dna: hashes, window minima, runs -> records scattered
//                            into C0 coarse buckets (1.8 B per k-mer instead of 8)
//   sk_hist1 / sk_scatter1   records of a coarse bucket -> 2^b1 mid buckets
//   sk_expand                a mid bucket's records -> keys, grouped by d2 into 16 nodes of ~3,800 keys
//   ... then the ordinary tree takes over at level 2: plan -> leaves (an oversize node is split further
//   by the generic levels, skew handling included).
//
// Keys meet their equals because the bucket is a function of the key; which bucket that is never shows in
// the result.  Group order: nodes in bucket order, ascending inside a node -- NOT globally ascending, which
// is why this engine sits behind dnagpu_count_kmers_unordered (PostgreSQL's GROUP BY order is unspecified,
// test.sql:95-104).
#include "kernels.hpp"

namespace dnagpu {

constexpr int SK_NT = 256;                       // threads of a front-end workgroup
constexpr int SK_TILE_ROWS = (SK_NT - 1) * 32;   // rows (k-mers) a tile emits records for: the last thread only supplies hashes
constexpr int SK_STAGE = 2048;                   // descriptors staged per scatter round

int sk_tile_rows() { return SK_TILE_ROWS; }

typedef unsigned long long ull2_t __attribute__((ext_vector_type(2)));

// bijective 32-bit mix of the 30-bit m-mer value: distinct m-mers never tie
__device__ __forceinline__ u32 sk_mix(u32 h)
{
    h += h << 10;
    h ^= h >> 6;
    h += h << 3;
    h ^= h >> 11;
    h += h << 15;
    h ^= h >> 16;
    return h;
}

// the three bucket digits of a minimizer hash.  The minimum of w hashes is small, i.e. its high bits are
// biased; a multiplication by an odd constant spreads any smooth density evenly over the product's top bits.
struct SkDigits {
    u32 d0, d1, d2;
};
__device__ __forceinline__ SkDigits sk_digits(u32 hmin, u32 c0, u32 b1mask)
{
    const u32 g = hmin * 0x9E3779B1u;
    SkDigits r;
    r.d0 = ((g >> 16) * c0) >> 16;
    r.d1 = (g >> 6) & b1mask;
    r.d2 = (g >> 2) & 15u;
    return r;
}

__device__ __forceinline__ u32 wave_incl_max(u32 x)
{
    int v = (int)x;                               // values are row indices + 1: small and non-negative
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return (u32)v;
}

// ------------------------------------------------------------------------------------------------
// Front end of both dna sweeps: one tile = rows [row0, row0 + n_rows), n_rows <= SK_TILE_ROWS, thread t owns
// the 32 rows row0 + 32 t ...  Every thread ends with the window-minimum hash of each of its rows in hm[]
// and the state of the record that is open at its first row (`c0` = rows of that record before it).  Records are
// cut at tile boundaries (one extra record per 8160 rows).
template <int W>
struct SkFront {
    u32 hm[32];
    u32 next_first;        // hm[0] of the next thread (undefined for the tile's last row owner: forced end there)
    u32 c0;                // length so far of the record open at this thread's first row
    u32 n_valid;           // rows of this thread that exist (0..32)
    u32 n_rows;            // rows of the tile
};

template <int W>
__device__ __forceinline__ void sk_front(SkFront<W> &f, const u64 *__restrict__ words, u64 n_words, u64 pos0, u32 n_rows,
                                         u32 lmax, u32 *hs /* [(W-1) * SK_NT] */, u32 *hx /* [2 * SK_NT + 8] */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the 64 bases from this thread's first row on (pos0 + 32 t): 4 dwords
    const u64 pos = pos0 + (u64)tid * 32;
    const u64 w = pos >> 5;
    const unsigned sh = (unsigned)(pos & 31) * 2;          // workgroup-uniform
    const u64 w0 = w < n_words ? words[w] : 0, w1 = w + 1 < n_words ? words[w + 1] : 0;
    u64 lo = w0, hi = w1;
    if (sh) {
        const u64 w2 = w + 2 < n_words ? words[w + 2] : 0;
        lo = (w0 >> sh) | (w1 << (64 - sh));
        hi = (w1 >> sh) | (w2 << (64 - sh));
    }
    const u32 d[4] = {(u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32)};
    u32 a[32 + W - 1];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int q = (2 * j) >> 5, s = (2 * j) & 31;
        const u32 v = __builtin_amdgcn_alignbit(d[q + 1], d[q], s) & 0x3FFFFFFFu;   // 15 bases
        a[j] = sk_mix(v);
    }
    // the next thread's first W-1 hashes complete this thread's windows
#pragma unroll
    for (int j = 0; j < W - 1; j++)
        hs[j * SK_NT + tid] = a[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < W - 1; j++)
        a[32 + j] = tid + 1 < SK_NT ? hs[j * SK_NT + tid + 1] : 0u;
    // window minima by doubling: after steps 1, 2, 4, ... a[i] = min over [i, i + P), P = largest power of two <= W
    constexpr int P = W >= 16 ? 16 : 8;
    static_assert(W >= 9 && W <= 18, "window lengths of k = 23 .. 32 at m = 15");
#pragma unroll
    for (int s = 1; s < P; s *= 2)
#pragma unroll
        for (int i = 0; i + s < 32 + W - 1; i++)
            a[i] = min(a[i], a[i + s]);
#pragma unroll
    for (int j = 0; j < 32; j++)
        f.hm[j] = min(a[j], a[j + W - P]);

    const u32 r0 = (u32)tid * 32;
    f.n_valid = r0 >= n_rows ? 0u : (n_rows - r0 < 32u ? n_rows - r0 : 32u);
    f.n_rows = n_rows;
    // neighbours' boundary minima
    hx[tid] = f.hm[0];
    hx[SK_NT + tid] = f.hm[31];
    __syncthreads();
    f.next_first = tid + 1 < SK_NT ? hx[tid + 1] : 0u;
    const u32 prev_last = tid > 0 ? hx[SK_NT + tid - 1] : 0u;
    // last natural break (a row whose minimum differs from the row before; the tile's first row counts) among this
    // thread's rows, as row index + 1 (0 = none) -> the run that is open at the next thread's first row
    u32 lb = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const bool brk = j == 0 ? (tid == 0 || f.hm[0] != prev_last) : (f.hm[j] != f.hm[j - 1]);
        if (brk)
            lb = r0 + (u32)j + 1u;
    }
    const u32 inc = wave_incl_max(lb);
    __syncthreads();                                       // hx is reused for the wave maxima
    if (lane == 63)
        hx[2 * SK_NT + wave] = inc;
    __syncthreads();
    u32 before = (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x138, 0xf, 0xf, false);   // wave_shr:1: the lanes before this one
    if (lane == 0)
        before = 0;
    for (int q = 0; q < wave; q++)
        before = max(before, hx[2 * SK_NT + q]);
    // `before` >= 1 for every thread but thread 0 (row 0 is a break); thread 0's own row 0 is a break too
    const u32 ns0 = before ? before - 1u : 0u;             // start row of the natural run that reaches r0 from the left
    const bool first_break = tid == 0 || f.hm[0] != prev_last;
    f.c0 = first_break ? 0u : (r0 - ns0) % lmax;
    __syncthreads();                                       // hs / hx may be rewritten by the caller
}

// walks the thread's rows in order and calls emit(end_row, len, hmin) for every record that ENDS in them
template <int W, typename Emit>
__device__ __forceinline__ void sk_records(const SkFront<W> &f, u32 lmax, Emit &&emit)
{
    const u32 r0 = (u32)threadIdx.x * 32;
    u32 c = f.c0;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        if ((u32)j < f.n_valid) {
            if (j > 0 && f.hm[j] != f.hm[j - 1])
                c = 0;                                     // (a break at j == 0 is already in c0)
            const u32 nxt = j < 31 ? f.hm[j + 1] : f.next_first;
            const bool last = r0 + (u32)j + 1 == f.n_rows;   // records are cut at the tile's end
            const bool end = last || nxt != f.hm[j] || c + 1 == lmax;
            if (end) {
                emit(r0 + (u32)j, c + 1, f.hm[j]);
                c = 0;
            } else {
                c++;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sk_hist0: records per coarse digit of every chunk of rows (the histogram the generic prefix kernels take)
template <int W>
__global__ __launch_bounds__(SK_NT) void sk_hist0_kernel(const Chunk *__restrict__ chunks, u32 n_chunks,
                                                         const u64 *__restrict__ words, u64 n_words, u64 first,
                                                         u32 lmax, u32 c0n, u32 b1mask, u32 r0n /* digits of the root's split */,
                                                         u32 *__restrict__ hist)
{
    __shared__ u32 hs[(W - 1) * SK_NT];
    __shared__ u32 hx[2 * SK_NT + 8];
    __shared__ u32 h[ROW_STRIDE];
    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    for (u32 d = threadIdx.x; d < r0n; d += SK_NT)
        h[d] = 0;
    __syncthreads();
    for (u32 t0 = 0; t0 < ch.len; t0 += SK_TILE_ROWS) {
        const u32 n_rows = ch.len - t0 < (u32)SK_TILE_ROWS ? ch.len - t0 : (u32)SK_TILE_ROWS;
        SkFront<W> f;
        sk_front<W>(f, words, n_words, first + ch.off + t0, n_rows, lmax, hs, hx);
        if (threadIdx.x < SK_NT - 1)
            sk_records<W>(f, lmax, [&](u32, u32, u32 hmin) { atomicAdd(&h[sk_digits(hmin, c0n, b1mask).d0], 1u); });
    }
    __syncthreads();
    u32 *row = hist + (u64)blockIdx.x * ROW_STRIDE;
    for (u32 d = threadIdx.x; d < r0n; d += SK_NT)
        row[d] = h[d];
}

// ------------------------------------------------------------------------------------------------
// sk_scatter0: the same sweep; the records of a tile are staged digit-sorted in LDS (as descriptors) and written
// out run by run: 16 bytes per lane, a digit's records consecutive.
//   descriptor = hmin << 32 | start row in tile << 5 | (len - 1)
//   record     = lo: bases 0..31 of the run; hi: bases 32..53 (bits 0..43) | (len-1) << 44 | d1 << 49 | d2 << 59
template <int W>
__global__ __launch_bounds__(SK_NT) void sk_scatter0_kernel(const Chunk *__restrict__ chunks, u32 n_chunks,
                                                            const u64 *__restrict__ words, u64 n_words, u64 first, int k,
                                                            u32 lmax, u32 c0n, u32 b1mask, u32 r0n,
                                                            const u32 *__restrict__ hist, const u32 *__restrict__ tot,
                                                            ull2_t *__restrict__ recs)
{
    __shared__ u32 hs[(W - 1) * SK_NT];
    __shared__ u32 hx[2 * SK_NT + 8];
    __shared__ u32 cnt[ROW_STRIDE];               // records per digit of this round -> exclusive offsets in the stage
    __shared__ u32 cur[ROW_STRIDE];               // placement cursors
    __shared__ u32 gpos[ROW_STRIDE];              // where the chunk's next record of each digit goes
    __shared__ u64 stage[SK_STAGE];
    __shared__ u32 wtmp[SK_NT / 64];
    __shared__ u32 round_total;
    if (blockIdx.x >= n_chunks)
        return;
    const int tid = threadIdx.x;
    const Chunk ch = chunks[blockIdx.x];
    const u32 *hrow = hist + (u64)blockIdx.x * ROW_STRIDE;
    const u32 *trow = tot;                         // the root is node 0: its totals row is row 0
    for (u32 d = tid; d < r0n; d += SK_NT)
        gpos[d] = trow[d] + hrow[d];
    __syncthreads();
    for (u32 t0 = 0; t0 < ch.len; t0 += SK_TILE_ROWS) {
        const u32 n_rows = ch.len - t0 < (u32)SK_TILE_ROWS ? ch.len - t0 : (u32)SK_TILE_ROWS;
        const u64 tile_pos = first + ch.off + t0;
        SkFront<W> f;
        sk_front<W>(f, words, n_words, tile_pos, n_rows, lmax, hs, hx);
        // a tile holds at most 8160 records, usually ~900: one round when they fit the stage, else one round per wave
        // (a wave's 64 threads end at most 2048 records)
        u32 mine = 0;
        if (tid < SK_NT - 1)
            sk_records<W>(f, lmax, [&](u32, u32, u32) { mine++; });
        const u32 tot_recs = wave_sum(mine);
        if ((tid & 63) == 0)
            wtmp[tid >> 6] = tot_recs;
        __syncthreads();
        u32 all = 0;
        for (int q = 0; q < SK_NT / 64; q++)
            all += wtmp[q];
        const int n_rounds = all <= (u32)SK_STAGE ? 1 : SK_NT / 64;
        for (int round = 0; round < n_rounds; round++) {
            const bool in_round = tid < SK_NT - 1 && (n_rounds == 1 || (tid >> 6) == round);
            for (u32 d = tid; d < r0n; d += SK_NT)
                cnt[d] = 0;
            __syncthreads();
            if (in_round)
                sk_records<W>(f, lmax, [&](u32, u32, u32 hmin) { atomicAdd(&cnt[sk_digits(hmin, c0n, b1mask).d0], 1u); });
            __syncthreads();
            const u32 n_stage = block_scan_inplace<SK_NT>(cnt, (int)r0n, wtmp);   // cnt -> exclusive offsets
            for (u32 d = tid; d < r0n; d += SK_NT)
                cur[d] = cnt[d];
            if (tid == 0)
                round_total = n_stage;
            __syncthreads();
            if (in_round)
                sk_records<W>(f, lmax, [&](u32 end_row, u32 len, u32 hmin) {
                    const u32 slot = atomicAdd(&cur[sk_digits(hmin, c0n, b1mask).d0], 1u);
                    stage[slot] = ((u64)hmin << 32) | ((u64)(end_row + 1 - len) << 5) | (u64)(len - 1);
                });
            __syncthreads();
            // write out: slot i -> recs[gpos[d0] + (i - cnt[d0])]
            for (u32 i = tid; i < round_total; i += SK_NT) {
                const u64 desc = stage[i];
                const u32 hmin = (u32)(desc >> 32), len = (u32)(desc & 31) + 1, start = (u32)(desc >> 5) & 0x3FFFu;
                const SkDigits dg = sk_digits(hmin, c0n, b1mask);
                const u64 p = tile_pos + start;                     // stream position of the run's first base
                const u64 wi = p >> 5;
                const unsigned s = (unsigned)(p & 31) * 2;
                const u64 a0 = wi < n_words ? words[wi] : 0, a1 = wi + 1 < n_words ? words[wi + 1] : 0,
                          a2 = wi + 2 < n_words ? words[wi + 2] : 0;
                u64 lo = funnel(a0, a1, s), hi = funnel(a1, a2, s);
                const u32 nb = len + (u32)k - 1;                    // bases of the run: <= 54
                if (nb < 32) {
                    lo &= ((u64)1 << (2 * nb)) - 1;
                    hi = 0;
                } else {
                    hi &= ((u64)1 << (2 * (nb - 32))) - 1;         // nb - 32 <= 22
                }
                hi |= ((u64)(len - 1) << 44) | ((u64)dg.d1 << 49) | ((u64)dg.d2 << 59);
                ull2_t r;
                r.x = lo;
                r.y = hi;
                recs[gpos[dg.d0] + (i - cnt[dg.d0])] = r;
            }
            __syncthreads();
            // advance the chunk's cursors by this round's counts: cur[d] ended at cnt[d] + count[d]
            for (u32 d = tid; d < r0n; d += SK_NT)
                gpos[d] += cur[d] - cnt[d];
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sk_hist1: records of a chunk of a coarse node by d1; also the k-mers of every (node, d1) = mid bucket
constexpr int SK1_NT = 1024;
__global__ __launch_bounds__(SK1_NT) void sk_hist1_kernel(const Node *__restrict__ nodes, const Chunk *__restrict__ chunks,
                                                          u32 n_chunks, const ull2_t *__restrict__ recs,
                                                          u32 *__restrict__ hist, u32 *__restrict__ kcount)
{
    __shared__ u32 h[ROW_STRIDE];
    __shared__ u32 kc[ROW_STRIDE];
    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    const u32 R = 1u << nd.split;
    for (u32 d = threadIdx.x; d < R; d += SK1_NT) {
        h[d] = 0;
        kc[d] = 0;
    }
    __syncthreads();
    const u64 *hi = reinterpret_cast<const u64 *>(recs + (u64)nd.start + ch.off) + 1;
    for (u32 i = threadIdx.x; i < ch.len; i += SK1_NT) {
        const u64 m = __builtin_nontemporal_load(&hi[(u64)i * 2]);
        const u32 d1 = (u32)(m >> 49) & (R - 1);
        atomicAdd(&h[d1], 1u);
        atomicAdd(&kc[d1], (u32)((m >> 44) & 31) + 1u);
    }
    __syncthreads();
    u32 *row = hist + (u64)blockIdx.x * ROW_STRIDE;
    for (u32 d = threadIdx.x; d < R; d += SK1_NT) {
        row[d] = h[d];
        if (kc[d])
            atomicAdd(&kcount[nd.child_base + d], kc[d]);
    }
}

// sk_scatter1: tiles of 8192 records; the tile's records are ranked by d1 in LDS as an index list and copied
// to their mid buckets 16 bytes per lane (8 records per digit and tile on average: 128-byte runs)
constexpr int SK1_ITEMS = 8;
constexpr int SK1_TILE = SK1_NT * SK1_ITEMS;
__global__ __launch_bounds__(SK1_NT) void sk_scatter1_kernel(const Node *__restrict__ nodes, const Chunk *__restrict__ chunks,
                                                             u32 n_chunks, const ull2_t *__restrict__ src_all,
                                                             ull2_t *__restrict__ dst_all, const u32 *__restrict__ hist,
                                                             const u32 *__restrict__ tot)
{
    __shared__ u32 cnt[ROW_STRIDE];
    __shared__ u32 gpos[ROW_STRIDE];
    __shared__ unsigned short idx[SK1_TILE];
    __shared__ u32 wtmp[SK1_NT / 64];
    if (blockIdx.x >= n_chunks)
        return;
    const int tid = threadIdx.x;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    const u32 R = 1u << nd.split;
    const u32 *hrow = hist + (u64)blockIdx.x * ROW_STRIDE;
    const u32 *trow = tot + (u64)nd.chunk_base * ROW_STRIDE;          // absolute base of every digit of the node
    for (u32 d = tid; d < R; d += SK1_NT)
        gpos[d] = trow[d] + hrow[d];
    const ull2_t *src = src_all + (u64)nd.start + ch.off;
    for (u32 t0 = 0; t0 < ch.len; t0 += SK1_TILE) {
        const u32 n_tile = ch.len - t0 < (u32)SK1_TILE ? ch.len - t0 : (u32)SK1_TILE;
        for (u32 d = tid; d < R; d += SK1_NT)
            cnt[d] = 0;
        __syncthreads();
        u32 dig[SK1_ITEMS], rank[SK1_ITEMS];
#pragma unroll
        for (int j = 0; j < SK1_ITEMS; j++) {
            const u32 i = tid + j * SK1_NT;
            dig[j] = 0;
            rank[j] = 0;
            if (i < n_tile) {
                const u64 m = reinterpret_cast<const u64 *>(src + t0 + i)[1];
                dig[j] = (u32)(m >> 49) & (R - 1);
                rank[j] = atomicAdd(&cnt[dig[j]], 1u);
            }
        }
        __syncthreads();
        block_scan_inplace<SK1_NT>(cnt, (int)R, wtmp);                     // cnt -> exclusive offsets
#pragma unroll
        for (int j = 0; j < SK1_ITEMS; j++) {
            const u32 i = tid + j * SK1_NT;
            if (i < n_tile)
                idx[cnt[dig[j]] + rank[j]] = (unsigned short)i;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SK1_ITEMS; j++) {
            const u32 s = tid + j * SK1_NT;
            if (s < n_tile) {
                const ull2_t r = src[t0 + idx[s]];
                const u32 d = (u32)(r.y >> 49) & (R - 1);
                __builtin_nontemporal_store(r, &dst_all[gpos[d] + (s - cnt[d])]);
            }
        }
        __syncthreads();
        // advance: digit d held (next offset - its offset) records
        for (u32 d = tid; d < R; d += SK1_NT) {
            const u32 end = d + 1 < R ? cnt[d + 1] : n_tile;
            gpos[d] += end - cnt[d];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// sk_expand: one workgroup per mid bucket.  Sweep A counts the bucket's k-mers per d2; the bucket's key range
// (key_base[node], from the scan of kcount) is cut into 16 nodes accordingly.  Sweep B takes 256 records at a
// time, stages their keys grouped by d2 in LDS and copies every group to its node's range, coalesced.
constexpr int SKX_NT = 256;
constexpr int SKX_STAGE = SKX_NT * 32;            // keys of 256 records of at most 32 k-mers
__global__ __launch_bounds__(SKX_NT) void sk_expand_kernel(const Node *__restrict__ mids, u32 n_mids,
                                                           const ull2_t *__restrict__ recs,
                                                           const u32 *__restrict__ key_base, int k,
                                                           u64 *__restrict__ keys, Node *__restrict__ out_nodes)
{
    __shared__ u32 kc[64][17];                    // sweep A: per-lane copies of the 16 k-mer counters
    __shared__ u32 nbase[17];                     // node j's keys start at key_base + nbase[j]
    __shared__ u32 done[16];                      // keys of node j already written
    __shared__ u32 sc[16], sb[17];                // this batch: keys per d2, their offsets in the stage
    __shared__ u32 scur[16];
    __shared__ u64 stage[SKX_STAGE];
    const u32 i = blockIdx.x;
    if (i >= n_mids)
        return;
    const int tid = threadIdx.x, lane = tid & 63;
    const Node nd = mids[i];
    const u64 kmask = kmer_mask(k);
    const ull2_t *src = recs + (u64)nd.start;
    for (int q = tid; q < 64 * 17; q += SKX_NT)
        (&kc[0][0])[q] = 0;
    if (tid < 16)
        done[tid] = 0;
    __syncthreads();
    for (u32 r = tid; r < nd.len; r += SKX_NT) {
        const u64 m = reinterpret_cast<const u64 *>(src + r)[1];
        atomicAdd(&kc[lane][(u32)(m >> 59) & 15u], (u32)((m >> 44) & 31) + 1u);
    }
    __syncthreads();
    if (tid < 16) {
        u32 s = 0;
        for (int q = 0; q < 64; q++)
            s += kc[q][tid];
        sc[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (int j = 0; j < 16; j++) {
            nbase[j] = run;
            run += sc[j];
        }
        nbase[16] = run;
    }
    __syncthreads();
    const u32 kb = key_base[i];
    if (tid < 16) {
        Node o;
        o.start = kb + nbase[tid];
        o.len = nbase[tid + 1] - nbase[tid];
        o.meta = (u32)(2 * k);                    // no key bit is fixed: the leaves sort on the top bits; buffer 0
        o.split = 0;
        o.prefix = 0;
        o.child_base = 0;
        o.chunk_base = 0;
        out_nodes[(u64)i * 16 + tid] = o;
    }
    for (u32 b0 = 0; b0 < nd.len; b0 += SKX_NT) {
        if (tid < 16) {
            sc[tid] = 0;
        }
        __syncthreads();
        const u32 r = b0 + tid;
        ull2_t rec;
        rec.x = 0;
        rec.y = 0;
        u32 len = 0, d2 = 0;
        if (r < nd.len) {
            rec = src[r];
            len = (u32)((rec.y >> 44) & 31) + 1u;
            d2 = (u32)(rec.y >> 59) & 15u;
            atomicAdd(&sc[d2], len);
        }
        __syncthreads();
        if (tid == 0) {
            u32 run = 0;
            for (int j = 0; j < 16; j++) {
                sb[j] = run;
                scur[j] = run;
                run += sc[j];
            }
            sb[16] = run;
        }
        __syncthreads();
        if (len) {
            u32 o = atomicAdd(&scur[d2], len);
            const u64 hi = rec.y & (((u64)1 << 44) - 1);
            for (u32 j = 0; j < len; j++)
                stage[o + j] = funnel(rec.x, hi, 2 * j) & kmask;
        }
        __syncthreads();
        const u32 n_keys = sb[16];
        for (u32 s = tid; s < n_keys; s += SKX_NT) {
            u32 j = 0;
#pragma unroll
            for (int q = 1; q < 16; q++)
                j += s >= sb[q] ? 1u : 0u;
            __builtin_nontemporal_store(stage[s], &keys[(u64)kb + nbase[j] + done[j] + (s - sb[j])]);
        }
        __syncthreads();
        if (tid < 16)
            done[tid] += sc[tid];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
template <int W>
static void launch_front(bool scatter, u32 n_chunks, hipStream_t s, const Chunk *chunks, const u64 *words, u64 n_words,
                         u64 first, int k, u32 lmax, u32 c0n, u32 b1mask, u32 r0n, u32 *hist, const u32 *tot, void *recs)
{
    if (scatter)
        hipLaunchKernelGGL(sk_scatter0_kernel<W>, dim3(n_chunks), dim3(SK_NT), 0, s, chunks, n_chunks, words, n_words, first,
                           k, lmax, c0n, b1mask, r0n, hist, tot, reinterpret_cast<ull2_t *>(recs));
    else
        hipLaunchKernelGGL(sk_hist0_kernel<W>, dim3(n_chunks), dim3(SK_NT), 0, s, chunks, n_chunks, words, n_words, first,
                           lmax, c0n, b1mask, r0n, hist);
}

int sk_min_k() { return 15 + 9 - 1; }              // window lengths 9 .. 18: k = 23 .. 32

hipError_t launch_sk_level0(bool scatter, const Chunk *chunks, u32 n_chunks, const u64 *words, u64 n_words, u64 first, int k,
                            u32 c0n, u32 b1bits, u32 r0bits, u32 *hist, const u32 *tot, void *recs, hipStream_t s)
{
    if (n_chunks == 0)
        return hipSuccess;
    const int w = k - 15 + 1;
    u32 lmax = (u32)(54 - k + 1);                  // a record holds 54 bases
    if (lmax > 32)
        lmax = 32;
    const u32 b1mask = (1u << b1bits) - 1, r0n = 1u << r0bits;
#define SK_CASE(W_) case W_: launch_front<W_>(scatter, n_chunks, s, chunks, words, n_words, first, k, lmax, c0n, b1mask, r0n, hist, tot, recs); break;
    switch (w) {
        SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16) SK_CASE(17) SK_CASE(18)
    default: return hipErrorInvalidValue;
    }
#undef SK_CASE
    return hipGetLastError();
}

hipError_t launch_sk_hist1(const Node *nodes, const Chunk *chunks, u32 n_chunks, const void *recs, u32 *hist, u32 *kcount,
                           hipStream_t s)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_hist1_kernel, dim3(n_chunks), dim3(SK1_NT), 0, s, nodes, chunks, n_chunks,
                       reinterpret_cast<const ull2_t *>(recs), hist, kcount);
    return hipGetLastError();
}

hipError_t launch_sk_scatter1(const Node *nodes, const Chunk *chunks, u32 n_chunks, const void *src, void *dst, const u32 *hist,
                              const u32 *tot, hipStream_t s)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_scatter1_kernel, dim3(n_chunks), dim3(SK1_NT), 0, s, nodes, chunks, n_chunks,
                       reinterpret_cast<const ull2_t *>(src), reinterpret_cast<ull2_t *>(dst), hist, tot);
    return hipGetLastError();
}

hipError_t launch_sk_expand(const Node *mids, u32 n_mids, const void *recs, const u32 *key_base, int k, u64 *keys,
                            Node *out_nodes, hipStream_t s)
{
    if (n_mids == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_expand_kernel, dim3(n_mids), dim3(SKX_NT), 0, s, mids, n_mids, reinterpret_cast<const ull2_t *>(recs),
                       key_base, k, keys, out_nodes);
    return hipGetLastError();
}

}  // namespace dnagpu
