// count_kernels.hip -- GROUP BY kmer, count(*) on gfx950: an MSD radix tree over the keys.
//
// The reference counts through PostgreSQL's HashAggregate: one random-access table probe per
// k-mer (kmer_hash dna.c:722-735 + kmer_eq dna.c:686-696).  On MI355X random 8-byte probes into a
// multi-GB table run at a few percent of HBM bandwidth, so the count is restated as
// partition -> local sort -> run-length encode, every pass streaming:
//
//   level l   each node (a set of keys sharing their high bits) larger than LEAF_CAP is split on its
//             next `split` most significant bits: per-chunk digit histograms in LDS (level_hist),
//             a prefix over chunks and digits (level_prefix, level_children), then a scatter that
//             ranks a tile of keys in LDS, stages it digit-sorted and writes each digit's run
//             contiguously (level_scatter).  The level-0 node reads the packed dna directly: the
//             extraction is fused, the raw keys are never materialised.
//   leaves    a node of <= LEAF_CAP keys is sorted in LDS (counting sort on its top 12 free bits,
//             then exact ranks inside each small bin), run-length encoded, and written at the
//             offset a chained scan over the leaves (in key order) hands it.
//   A node whose bits are exhausted holds one distinct key: it is emitted as (key, len) directly,
//   which is how heavy hitters and small k terminate.
//
// Output: groups in ascending key order; bit-exact against the oracle's sorted hash-aggregate.
#include "kernels.hpp"

namespace dnagpu {

constexpr int SC_THREADS = 512;               // level_hist / level_scatter workgroup
constexpr int SC_ITEMS = 16;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS;   // 8192 keys staged in LDS per tile
constexpr int LF_THREADS = 512;               // leaf workgroup
constexpr int LF_ITEMS = LEAF_CAP / LF_THREADS;  // 8
constexpr int LF_SUB_BITS = 12;               // counting-sort bins per leaf: 4096

int scatter_tile_keys() { return SC_TILE; }
int scatter_threads() { return SC_THREADS; }

__device__ __forceinline__ int ceil_log2_u32(u32 x)
{
    return x <= 1 ? 0 : 32 - __clz(x - 1);
}

// In-place exclusive scan of arr[0..n) in LDS by NT threads; returns the total.  The caller has
// synchronised before the call; the function synchronises before returning.
template <int NT>
__device__ __forceinline__ u32 block_scan_inplace(u32 *arr, int n, u32 *wtmp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + NT - 1) / NT;
    const int b = tid * per;
    const int e = (b + per < n) ? b + per : n;
    u32 sum = 0;
    for (int i = b; i < e; i++)
        sum += arr[i];
    u32 inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
        u32 t = __shfl_up(inc, off);
        if (lane >= off)
            inc += t;
    }
    if (lane == 63)
        wtmp[wave] = inc;
    __syncthreads();
    u32 wbase = 0, total = 0;
    for (int w = 0; w < NT / 64; w++) {
        u32 t = wtmp[w];
        if (w < wave)
            wbase += t;
        total += t;
    }
    u32 run = wbase + inc - sum;
    for (int i = b; i < e; i++) {
        u32 v = arr[i];
        arr[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

// ------------------------------------------------------------------------------------------------
// plan: one thread per node decides leaf / split width (see DESIGN.md "level plan")
__global__ __launch_bounds__(256) void plan_kernel(Node *__restrict__ nodes, u32 n_nodes, int level,
                                                   u32 chunk_len, u32 *__restrict__ outc,
                                                   u32 *__restrict__ nch, LevelCounters *__restrict__ ctr)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes)
        return;
    Node nd = nodes[i];
    int rem = (int)(nd.meta & 0xff);
    int bits = 0;
    if (level < 0) {
        // forced level (multi-GPU owner partition): split the node on -level bits whatever its size
        bits = -level;
        if (bits > rem) bits = rem;
        if (nd.len == 0) bits = 0;
    } else if (nd.len > (u32)LEAF_CAP && rem > 0 && !(nd.meta & NODE_TERMINAL)) {
        int want = ceil_log2_u32((nd.len + LEAF_TARGET - 1) / LEAF_TARGET);
        if (level == 0) {
            if (want > MAX_SPLIT_BITS)
                want = (want + 1) / 2;          // two balanced levels
        } else if (level >= 2) {
            want += 4;                          // an oversize survivor is skewed: fan out harder
        }
        bits = want;
        if (bits > MAX_SPLIT_BITS) bits = MAX_SPLIT_BITS;
        if (bits > rem) bits = rem;
        if (bits < 1) bits = 1;
    }
    nodes[i].split = (u32)bits;
    outc[i] = bits ? (1u << bits) : 1u;
    nch[i] = bits ? (nd.len + chunk_len - 1) / chunk_len : 0u;
    if (bits) {
        atomicAdd(&ctr->n_split, 1u);
        if (bits < rem)
            atomicAdd(&ctr->n_scatter, 1u);
    }
}

hipError_t launch_plan(Node *nodes, u32 n_nodes, int level, u32 chunk_len, u32 *outc, u32 *nch,
                       LevelCounters *ctr, hipStream_t s)
{
    if (n_nodes == 0)
        return hipSuccess;
    hipLaunchKernelGGL(plan_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, nodes, n_nodes, level,
                       chunk_len, outc, nch, ctr);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void fill_chunks_kernel(const Node *__restrict__ nodes, u32 n_nodes,
                                                          u32 chunk_len, const u32 *__restrict__ child_base,
                                                          const u32 *__restrict__ chunk_base,
                                                          Node *__restrict__ nodes_rw, Chunk *__restrict__ chunks)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes)
        return;
    u32 cb = child_base[i], kb = chunk_base[i];
    nodes_rw[i].child_base = cb;
    nodes_rw[i].chunk_base = kb;
    if (nodes[i].split) {
        u32 len = nodes[i].len;
        u32 nc = (len + chunk_len - 1) / chunk_len;
        for (u32 c = 0; c < nc; c++) {
            Chunk ch;
            ch.node = i;
            ch.off = c * chunk_len;
            ch.len = (len - ch.off < chunk_len) ? len - ch.off : chunk_len;
            ch.pad = 0;
            chunks[kb + c] = ch;
        }
    }
}

hipError_t launch_fill_chunks(const Node *nodes, u32 n_nodes, u32 chunk_len, const u32 *child_base,
                              const u32 *chunk_base, Node *nodes_rw, Chunk *chunks, hipStream_t s)
{
    if (n_nodes == 0)
        return hipSuccess;
    hipLaunchKernelGGL(fill_chunks_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, nodes, n_nodes,
                       chunk_len, child_base, chunk_base, nodes_rw, chunks);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// 16 consecutive windows starting at base `pos` from three packed words (pos..pos+15 start in words
// w, w+1; a window may reach into w+2).  Cheaper than 16 independent key_at calls.
struct Win16 {
    u64 w0, w1, w2;
    unsigned rel0;
};
__device__ __forceinline__ Win16 win16_load(const u64 *__restrict__ words, u64 n_words, u64 pos)
{
    Win16 r;
    u64 w = pos >> 5;
    r.rel0 = (unsigned)(pos & 31);
    r.w0 = w < n_words ? words[w] : 0;
    r.w1 = w + 1 < n_words ? words[w + 1] : 0;
    r.w2 = w + 2 < n_words ? words[w + 2] : 0;
    return r;
}
__device__ __forceinline__ u64 win16_key(const Win16 &r, int j, u64 mask)
{
    unsigned rel = r.rel0 + (unsigned)j;           // 0..46
    u64 lo = rel < 32 ? r.w0 : r.w1;
    u64 hi = rel < 32 ? r.w1 : r.w2;
    return funnel(lo, hi, (rel & 31) * 2) & mask;
}

// ------------------------------------------------------------------------------------------------
// level_hist: one workgroup per chunk -> hist[chunk][digit]
template <bool SRC_DNA>
__global__ __launch_bounds__(SC_THREADS) void level_hist_kernel(const Node *__restrict__ nodes,
                                                                const Chunk *__restrict__ chunks, u32 n_chunks,
                                                                const u64 *__restrict__ words, u64 n_words,
                                                                u64 first, u64 mask,
                                                                const u64 *__restrict__ buf0,
                                                                const u64 *__restrict__ buf1,
                                                                u32 *__restrict__ hist)
{
    __shared__ u32 h[ROW_STRIDE];
    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    const int bits = (int)nd.split;
    const int shift = (int)(nd.meta & 0xff) - bits;
    const u32 R = 1u << bits, dmask = R - 1;
    for (u32 d = threadIdx.x; d < R; d += SC_THREADS)
        h[d] = 0;
    __syncthreads();
    const u64 origin = (u64)nd.start + ch.off;
    if (SRC_DNA) {
        for (u32 i0 = threadIdx.x * 16; i0 < ch.len; i0 += SC_THREADS * 16) {
            Win16 w = win16_load(words, n_words, first + origin + i0);
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (i0 + j < ch.len)
                    atomicAdd(&h[(u32)(win16_key(w, j, mask) >> shift) & dmask], 1u);
        }
    } else {
        const u64 *__restrict__ src = ((nd.meta & NODE_BUF) ? buf1 : buf0) + origin;
        for (u32 i = threadIdx.x; i < ch.len; i += SC_THREADS)
            atomicAdd(&h[(u32)(src[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
    u32 *row = hist + (u64)blockIdx.x * ROW_STRIDE;
    for (u32 d = threadIdx.x; d < R; d += SC_THREADS)
        row[d] = h[d];
}

hipError_t launch_level_hist(const Node *nodes, const Chunk *chunks, u32 n_chunks, int src_dna,
                             const u64 *words, u64 n_words, u64 first, int k, const u64 *buf0,
                             const u64 *buf1, u32 *hist, hipStream_t s)
{
    if (n_chunks == 0)
        return hipSuccess;
    if (src_dna)
        hipLaunchKernelGGL(level_hist_kernel<true>, dim3(n_chunks), dim3(SC_THREADS), 0, s, nodes, chunks,
                           n_chunks, words, n_words, first, kmer_mask(k), buf0, buf1, hist);
    else
        hipLaunchKernelGGL(level_hist_kernel<false>, dim3(n_chunks), dim3(SC_THREADS), 0, s, nodes, chunks,
                           n_chunks, words, n_words, first, kmer_mask(k), buf0, buf1, hist);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// level_prefix: for every split node and 64-digit group, turn the node's chunk rows into exclusive
// prefixes over chunks (per digit) and store the per-digit totals in tot[row of the first chunk].
// Grid = (n_chunks, ROW_STRIDE/64); only the workgroup of a node's first chunk works.
__global__ __launch_bounds__(256) void level_prefix_kernel(const Node *__restrict__ nodes,
                                                           const Chunk *__restrict__ chunks, u32 n_chunks,
                                                           u32 chunk_len, u32 *__restrict__ hist,
                                                           u32 *__restrict__ tot)
{
    __shared__ u32 part[4][64];
    const u32 c0 = blockIdx.x;
    if (c0 >= n_chunks)
        return;
    const Chunk ch = chunks[c0];
    if (ch.off != 0)
        return;                                   // not the node's first chunk
    const Node nd = nodes[ch.node];
    const u32 R = 1u << nd.split;
    const u32 d = blockIdx.y * 64 + (threadIdx.x & 63);
    if (blockIdx.y * 64 >= R)
        return;
    const u32 nc = (nd.len + chunk_len - 1) / chunk_len;
    const u32 slice = threadIdx.x >> 6;
    const u32 per = (nc + 3) / 4;
    const u32 cb = slice * per < nc ? slice * per : nc;
    const u32 ce = cb + per < nc ? cb + per : nc;
    const bool live = d < R;
    u32 sum = 0;
    if (live)
        for (u32 c = cb; c < ce; c++)
            sum += hist[(u64)(c0 + c) * ROW_STRIDE + d];
    part[slice][threadIdx.x & 63] = sum;
    __syncthreads();
    u32 base = 0, total = 0;
    for (u32 s = 0; s < 4; s++) {
        u32 t = part[s][threadIdx.x & 63];
        if (s < slice)
            base += t;
        total += t;
    }
    if (live) {
        u32 run = base;
        for (u32 c = cb; c < ce; c++) {
            u64 idx = (u64)(c0 + c) * ROW_STRIDE + d;
            u32 v = hist[idx];
            hist[idx] = run;
            run += v;
        }
        if (slice == 0)
            tot[(u64)c0 * ROW_STRIDE + d] = total;
    }
}

// level_children: one workgroup per node.  Leaf: copied to its slot in the next list.  Split: the
// per-digit totals are scanned over digits; child d becomes a node at start + excl[d]; the tot row
// is overwritten with the absolute base of every digit (read by the scatter).
__global__ __launch_bounds__(256) void level_children_kernel(const Node *__restrict__ nodes, u32 n_nodes,
                                                             u32 *__restrict__ tot, Node *__restrict__ next)
{
    __shared__ u32 ex[ROW_STRIDE];
    __shared__ u32 wtmp[4];
    const u32 i = blockIdx.x;
    if (i >= n_nodes)
        return;
    const Node nd = nodes[i];
    if (nd.split == 0) {
        if (threadIdx.x == 0) {
            Node o = nd;
            o.split = 0;
            next[nd.child_base] = o;
        }
        return;
    }
    const int bits = (int)nd.split;
    const int rem = (int)(nd.meta & 0xff);
    const u32 R = 1u << bits;
    u32 *row = tot + (u64)nd.chunk_base * ROW_STRIDE;
    for (u32 d = threadIdx.x; d < R; d += 256)
        ex[d] = row[d];
    __syncthreads();
    // keep the totals: child len = total, read before the in-place scan overwrites them
    u32 lens[ROW_STRIDE / 256];
#pragma unroll
    for (int q = 0; q < ROW_STRIDE / 256; q++) {
        u32 d = threadIdx.x + q * 256;
        lens[q] = d < R ? ex[d] : 0;
    }
    __syncthreads();
    block_scan_inplace<256>(ex, (int)R, wtmp);
    const u32 child_meta = (u32)(rem - bits) | ((nd.meta & NODE_BUF) ^ NODE_BUF) |
                           ((bits == rem) ? NODE_TERMINAL : 0u);
    const bool root_dna_child_buf0 = false;
    (void)root_dna_child_buf0;
#pragma unroll
    for (int q = 0; q < ROW_STRIDE / 256; q++) {
        u32 d = threadIdx.x + q * 256;
        if (d < R) {
            Node c;
            c.start = nd.start + ex[d];
            c.len = lens[q];
            c.meta = child_meta;
            c.split = 0;
            c.prefix = nd.prefix | ((u64)d << (rem - bits));
            c.child_base = 0;
            c.chunk_base = 0;
            next[nd.child_base + d] = c;
            row[d] = nd.start + ex[d];
        }
    }
}

hipError_t launch_level_children(const Node *nodes, u32 n_nodes, u32 *hist, Node *next, hipStream_t s)
{
    // `hist` here is the tot table (same geometry as the hist table); the prefix kernel is launched
    // separately by the host through launch_level_prefix
    if (n_nodes == 0)
        return hipSuccess;
    hipLaunchKernelGGL(level_children_kernel, dim3(n_nodes), dim3(256), 0, s, nodes, n_nodes, hist, next);
    return hipGetLastError();
}

hipError_t launch_level_prefix(const Node *nodes, const Chunk *chunks, u32 n_chunks, u32 chunk_len,
                               u32 *hist, u32 *tot, hipStream_t s)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(level_prefix_kernel, dim3(n_chunks, ROW_STRIDE / 64), dim3(256), 0, s, nodes, chunks,
                       n_chunks, chunk_len, hist, tot);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// level_scatter: one workgroup per chunk of a non-terminal split node.  Per tile of 8192 keys:
//   rank   LDS atomic per key on its digit counter (order inside a digit is irrelevant: every
//          later stage sorts)
//   scan   exclusive scan of the counters
//   stage  keys written digit-sorted into LDS
//   write  thread i copies staged key i to out[base[d] + (i - excl[d])]: each digit's run is one
//          contiguous, coalesced segment
template <bool SRC_DNA>
__global__ __launch_bounds__(SC_THREADS, 4) void level_scatter_kernel(const Node *__restrict__ nodes,
                                                                   const Chunk *__restrict__ chunks, u32 n_chunks,
                                                                   const u64 *__restrict__ words, u64 n_words,
                                                                   u64 first, u64 mask, u64 *__restrict__ buf0,
                                                                   u64 *__restrict__ buf1,
                                                                   const u32 *__restrict__ hist,
                                                                   const u32 *__restrict__ tot)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *stage = reinterpret_cast<u64 *>(smem);                         // SC_TILE keys
    u32 *excl = reinterpret_cast<u32 *>(smem + (size_t)SC_TILE * 8);    // ROW_STRIDE + 1
    u32 *offs = excl + ROW_STRIDE + 4;                                  // ROW_STRIDE
    u32 *wtmp = offs + ROW_STRIDE;                                      // SC_THREADS / 64

    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    const int bits = (int)nd.split;
    const int rem = (int)(nd.meta & 0xff);
    if (bits == rem)
        return;                                   // terminal split: children carry (key, count) already
    const int shift = rem - bits;
    const u32 R = 1u << bits, dmask = R - 1;
    const u64 origin = (u64)nd.start + ch.off;
    const u64 *__restrict__ src = SRC_DNA ? nullptr : (((nd.meta & NODE_BUF) ? buf1 : buf0) + origin);
    // children of the dna root go to buffer 0; otherwise to the other buffer
    u64 *__restrict__ dst = SRC_DNA ? buf0 : ((nd.meta & NODE_BUF) ? buf0 : buf1);

    const u32 *hrow = hist + (u64)blockIdx.x * ROW_STRIDE;
    const u32 *trow = tot + (u64)nd.chunk_base * ROW_STRIDE;
    for (u32 d = threadIdx.x; d < R; d += SC_THREADS)
        offs[d] = hrow[d] + trow[d];

    for (u32 t0 = 0; t0 < ch.len; t0 += SC_TILE) {
        const u32 tn = ch.len - t0 < (u32)SC_TILE ? ch.len - t0 : (u32)SC_TILE;
        for (u32 d = threadIdx.x; d <= R; d += SC_THREADS)
            excl[d] = 0;
        // keys of this thread: the dna root recomputes them from three packed words (cheaper than
        // holding 16 keys in registers); key nodes load them once
        Win16 w;
        u64 key[SRC_DNA ? 1 : SC_ITEMS];
        if (SRC_DNA) {
            w = win16_load(words, n_words, first + origin + t0 + threadIdx.x * 16);
        } else {
#pragma unroll
            for (int j = 0; j < SC_ITEMS; j++) {
                u32 i = threadIdx.x + j * SC_THREADS;
                key[SRC_DNA ? 0 : j] = i < tn ? src[t0 + i] : 0;
            }
        }
        __syncthreads();
        u32 rank[SC_ITEMS];
#pragma unroll
        for (int j = 0; j < SC_ITEMS; j++) {
            u32 i = SRC_DNA ? threadIdx.x * 16 + j : threadIdx.x + j * SC_THREADS;
            u64 kv = SRC_DNA ? win16_key(w, j, mask) : key[SRC_DNA ? 0 : j];
            rank[j] = 0;
            if (i < tn)
                rank[j] = atomicAdd(&excl[(u32)(kv >> shift) & dmask], 1u);
        }
        __syncthreads();
        block_scan_inplace<SC_THREADS>(excl, (int)R, wtmp);
        if (threadIdx.x == 0)
            excl[R] = tn;
#pragma unroll
        for (int j = 0; j < SC_ITEMS; j++) {
            u32 i = SRC_DNA ? threadIdx.x * 16 + j : threadIdx.x + j * SC_THREADS;
            u64 kv = SRC_DNA ? win16_key(w, j, mask) : key[SRC_DNA ? 0 : j];
            if (i < tn)
                stage[excl[(u32)(kv >> shift) & dmask] + rank[j]] = kv;
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < tn; i += SC_THREADS) {
            u64 kv = stage[i];
            u32 d = (u32)(kv >> shift) & dmask;
            dst[(u64)offs[d] + (i - excl[d])] = kv;
        }
        __syncthreads();
        for (u32 d = threadIdx.x; d < R; d += SC_THREADS)
            offs[d] += excl[d + 1] - excl[d];
        __syncthreads();
    }
}

constexpr size_t SC_SMEM = (size_t)SC_TILE * 8 + (size_t)(ROW_STRIDE + 4) * 4 + (size_t)ROW_STRIDE * 4 +
                           (SC_THREADS / 64) * 4;

hipError_t launch_level_scatter(const Node *nodes, const Chunk *chunks, u32 n_chunks, int src_dna,
                                const u64 *words, u64 n_words, u64 first, int k, u64 *buf0, u64 *buf1,
                                const u32 *hist, const u32 *tot, hipStream_t s)
{
    if (n_chunks == 0)
        return hipSuccess;
    if (src_dna)
        hipLaunchKernelGGL(level_scatter_kernel<true>, dim3(n_chunks), dim3(SC_THREADS), SC_SMEM, s, nodes,
                           chunks, n_chunks, words, n_words, first, kmer_mask(k), buf0, buf1, hist, tot);
    else
        hipLaunchKernelGGL(level_scatter_kernel<false>, dim3(n_chunks), dim3(SC_THREADS), SC_SMEM, s, nodes,
                           chunks, n_chunks, words, n_words, first, kmer_mask(k), buf0, buf1, hist, tot);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// leaves: sort + run-length encode each leaf in LDS, emit groups in key order.
//
// Workgroups take leaves in ticket order (an atomic counter), so the chained scan below only ever
// waits on workgroups that started earlier: no dependency on dispatch order or placement.  The
// status word of a leaf is one self-validating 8-byte granule {flag:2, value:62} written and read
// with relaxed agent-scope atomics (write-through / L1-bypassing); no other data is handed between
// workgroups, so no fence is needed.
constexpr u64 ST_AGG = (u64)1 << 62;     // value = this leaf's group count
constexpr u64 ST_INC = (u64)2 << 62;     // value = group count of all leaves up to and including this
constexpr u64 ST_VAL = ((u64)1 << 62) - 1;

__global__ __launch_bounds__(LF_THREADS, 6) void leaves_kernel(const Node *__restrict__ leaves, u32 n_leaves,
                                                            const u64 *__restrict__ buf0,
                                                            const u64 *__restrict__ buf1,
                                                            u64 *__restrict__ status, u32 *__restrict__ ticket,
                                                            u64 *__restrict__ out_keys,
                                                            u64 *__restrict__ out_counts)
{
    __shared__ __attribute__((aligned(16))) u64 A[LEAF_CAP];
    __shared__ u32 H[LEAF_CAP + 8];          // bins -> exclusive offsets, later head positions
    __shared__ u32 rowcnt[LF_ITEMS * (LF_THREADS / 64)];   // 64 entries: [row][wave]
    __shared__ u32 wtmp[LF_THREADS / 64];
    __shared__ u32 sh_ticket;
    __shared__ u64 sh_excl;
    __shared__ u32 sh_D;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0)
        sh_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 li = sh_ticket;
    if (li >= n_leaves)
        return;
    const Node nd = leaves[li];
    const u32 len = nd.len;
    const int rem = (int)(nd.meta & 0xff);
    u32 D = 0;                                 // groups in this leaf
    const bool single = (len > 0) && (rem == 0 || (nd.meta & NODE_TERMINAL));

    if (single) {
        D = 1;
    } else if (len > 0) {
        const u64 *__restrict__ src = ((nd.meta & NODE_BUF) ? buf1 : buf0) + nd.start;
        const int sb = rem < LF_SUB_BITS ? rem : LF_SUB_BITS;
        const int sshift = rem - sb;
        const u32 S = 1u << sb, smask = S - 1;
        for (u32 d = tid; d <= S; d += LF_THREADS)
            H[d] = 0;
        __syncthreads();
        u64 key[LF_ITEMS];
        u32 rank[LF_ITEMS];
#pragma unroll
        for (int j = 0; j < LF_ITEMS; j++) {
            u32 i = tid + j * LF_THREADS;
            key[j] = 0;
            rank[j] = 0;
            if (i < len) {
                key[j] = src[i];
                rank[j] = atomicAdd(&H[(u32)(key[j] >> sshift) & smask], 1u);
            }
        }
        __syncthreads();
        block_scan_inplace<LF_THREADS>(H, (int)S, wtmp);
        if (tid == 0)
            H[S] = len;
#pragma unroll
        for (int j = 0; j < LF_ITEMS; j++) {
            u32 i = tid + j * LF_THREADS;
            if (i < len)
                A[H[(u32)(key[j] >> sshift) & smask] + rank[j]] = key[j];
        }
        __syncthreads();
        if (rem > sb) {
            // exact rank inside each (small) bin: #smaller + #equal-before
            u32 fin[LF_ITEMS];
#pragma unroll
            for (int j = 0; j < LF_ITEMS; j++) {
                u32 i = tid + j * LF_THREADS;
                fin[j] = 0;
                if (i < len) {
                    u64 kv = A[i];
                    key[j] = kv;
                    u32 b = (u32)(kv >> sshift) & smask;
                    u32 b0 = H[b], b1 = H[b + 1];
                    u32 r = b0;
#pragma unroll 1
                    for (u32 m = b0; m < b1; m++) {
                        u64 o = A[m];
                        r += (o < kv) || (o == kv && m < i);
                    }
                    fin[j] = r;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < LF_ITEMS; j++) {
                u32 i = tid + j * LF_THREADS;
                if (i < len)
                    A[fin[j]] = key[j];
            }
            __syncthreads();
        }
        // run heads in sorted order; row j = positions [j*LF_THREADS, (j+1)*LF_THREADS)
        u64 hb[LF_ITEMS];
#pragma unroll
        for (int j = 0; j < LF_ITEMS; j++) {
            u32 i = tid + j * LF_THREADS;
            bool head = i < len && (i == 0 || A[i] != A[i - 1]);
            hb[j] = __ballot(head);
            if (lane == 0)
                rowcnt[j * (LF_THREADS / 64) + wave] = (u32)__popcll(hb[j]);
        }
        __syncthreads();
        if (wave == 0) {
            u32 v = rowcnt[lane], inc = v;
            for (int off = 1; off < 64; off <<= 1) {
                u32 t = __shfl_up(inc, off);
                if (lane >= off)
                    inc += t;
            }
            rowcnt[lane] = inc - v;
            if (lane == 63)
                sh_D = inc;
        }
        __syncthreads();
        D = sh_D;
        const u64 below = ((u64)1 << lane) - 1;
#pragma unroll
        for (int j = 0; j < LF_ITEMS; j++)
            if ((hb[j] >> lane) & 1)
                H[rowcnt[j * (LF_THREADS / 64) + wave] + (u32)__popcll(hb[j] & below)] = tid + j * LF_THREADS;
        if (tid == 0)
            H[D] = len;
        // (H as bin offsets is dead: every read of it happened before the barriers above)
    }

    // ---- chained scan over leaves: exclusive group count of all earlier leaves
    if (wave == 0) {
        if (lane == 0)
            st_agent(&status[li], (li == 0 ? ST_INC : ST_AGG) | (u64)D);
        u64 excl = 0;
        if (li > 0) {
            long long j = (long long)li - 1;
            u32 spins = 0;
            for (;;) {
                long long idx = j - lane;
                u64 sv = idx >= 0 ? ld_agent(&status[idx]) : ST_INC;
                u32 flag = (u32)(sv >> 62);
                if (__any(flag == 0)) {
                    // every earlier ticket is held by a running or finished workgroup, so this wait
                    // ends; the bound only turns a logic error into a reported failure, not a hang
                    if (++spins > (1u << 24)) {
                        if (lane == 0)
                            atomicOr(ticket + 1, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                u64 incmask = __ballot(flag == 2);
                u64 val = sv & ST_VAL;
                if (incmask) {
                    int firstinc = __ffsll((long long)incmask) - 1;
                    if (lane > firstinc)
                        val = 0;
                }
                for (int off = 32; off > 0; off >>= 1)
                    val += __shfl_down(val, off);
                val = __shfl(val, 0);
                excl += val;
                if (incmask)
                    break;
                j -= 64;
            }
            if (lane == 0)
                st_agent(&status[li], ST_INC | (excl + D));
        }
        if (lane == 0)
            sh_excl = excl;
    }
    __syncthreads();
    const u64 obase = sh_excl;
    if (single) {
        if (tid == 0) {
            out_keys[obase] = nd.prefix;
            out_counts[obase] = len;
        }
    } else {
        for (u32 q = tid; q < D; q += LF_THREADS) {
            u32 p = H[q];
            out_keys[obase + q] = A[p];
            out_counts[obase + q] = H[q + 1] - p;
        }
    }
}

hipError_t launch_leaves(const Node *leaves, u32 n_leaves, const u64 *buf0, const u64 *buf1,
                         u64 *status, u32 *ticket, u64 *out_keys, u64 *out_counts, hipStream_t s)
{
    if (n_leaves == 0)
        return hipSuccess;
    hipLaunchKernelGGL(leaves_kernel, dim3(n_leaves), dim3(LF_THREADS), 0, s, leaves, n_leaves, buf0, buf1,
                       status, ticket, out_keys, out_counts);
    return hipGetLastError();
}

}  // namespace dnagpu
