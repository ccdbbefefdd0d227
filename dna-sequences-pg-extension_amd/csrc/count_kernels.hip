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
//   leaves    a node of <= LEAF_CAP keys is sorted in LDS (counting sort on its top 13 free bits,
//             then exact ranks inside each small bin), run-length encoded, and written at the
//             offset a chained scan over the leaves (in key order) hands it.
//   A node whose bits are exhausted holds one distinct key: it is emitted as (key, len) directly,
//   which is how heavy hitters and small k terminate.
//
// Output: groups in ascending key order; bit-exact against the oracle's sorted hash-aggregate.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "kernels.hpp"

#ifndef DNAGPU_NT
#define DNAGPU_NT 1
#endif
#if DNAGPU_NT
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#define NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define NT_LOAD(p) (*(p))
#define NT_STORE(v, p) (*(p) = (v))
#endif

namespace dnagpu {

// Diagnostic build only (make STAMPS=1): thread 0 of every workgroup accumulates shader-clock cycles
// per phase of the scatter / leaves kernels; DNAGPU_STAMPS=1 prints the totals after each launch.
// The same build honours the experiment switches read through diag_env() -- timing ablations that
// make results INVALID (DNAGPU_DEBUG_SCATTER / DNAGPU_DEBUG_LEAVES) and shape experiments that are
// only valid for some trees (DNAGPU_WC_NTH, DNAGPU_L1_BITS).  The product build ignores them.
#ifdef DNAGPU_STAMPS
static inline const char *diag_env(const char *name) { return getenv(name); }
#else
static inline const char *diag_env(const char *) { return nullptr; }
#endif
#ifdef DNAGPU_STAMPS
__device__ unsigned long long g_stamps[32];
#define STAMP_DECL unsigned long long st_acc[12] = {0}; unsigned long long st_last = __builtin_readcyclecounter();
#define STAMP(k) do { if (threadIdx.x == 0) { unsigned long long st_now = __builtin_readcyclecounter(); st_acc[k] += st_now - st_last; st_last = st_now; } } while (0)
#define STAMP_FLUSH(base) do { if (threadIdx.x == 0) { for (int st_i = 0; st_i < 12; st_i++) atomicAdd(&g_stamps[(base) + st_i], st_acc[st_i]); } } while (0)
static void stamps_report(const char *what, int base, hipStream_t s)
{
    const char *e = getenv("DNAGPU_STAMPS");
    if (!e || !atoi(e))
        return;
    unsigned long long h[32], z[32] = {0};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z);
    unsigned long long tot = 0;
    for (int i = 0; i < 12; i++)
        tot += h[base + i];
    fprintf(stderr, "[stamps] %s:", what);
    for (int i = 0; i < 12; i++)
        fprintf(stderr, " p%d=%.1f%%", i, tot ? 100.0 * h[base + i] / tot : 0.0);
    fprintf(stderr, " total=%.3e cycles\n", (double)tot);
}
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH(base)
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting: remembered per HIP device, so a
// process that opens contexts on several GPUs sets it on each
enum { FA_SCATTER = 0, FA_LEAVES = 1, FA_DENSE = 2, FA_COUNT = 3 };
static bool g_func_attrs[64][FA_COUNT];
static inline bool func_attrs_ready(int what)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return false;
    return g_func_attrs[dev][what];
}
static inline void func_attrs_mark(int what)
{
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64)
        g_func_attrs[dev][what] = true;
}

constexpr int SC_THREADS = 1024;              // level_hist workgroup
constexpr int SC_TILE = 8192;                 // keys staged in LDS per scatter tile

int scatter_tile_keys() { return SC_TILE; }
int scatter_threads() { return SC_THREADS; }

__device__ __forceinline__ int ceil_log2_u32(u32 x)
{
    return x <= 1 ? 0 : 32 - __clz(x - 1);
}

// ------------------------------------------------------------------------------------------------
// plan: one thread per node decides leaf / split width (see DESIGN.md "level plan")
// nodes of the level that must split (read by plan_kernel to size the fan-out of deep levels)
__global__ __launch_bounds__(256) void plan_count_kernel(const Node *__restrict__ nodes, u32 n_nodes,
                                                         LevelCounters *__restrict__ ctr)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes)
        return;
    const Node nd = nodes[i];
    if (nd.len > (u32)LEAF_CAP && (nd.meta & 0xff) > 0 && !(nd.meta & NODE_TERMINAL))
        atomicAdd(&ctr->n_over, 1u);
}

// split width of one node (0 = not split at this level); n_over = the level's oversize nodes (levels >= 2)
// skew_from: the first level whose oversize nodes are skew (2; one more if the levels start from the super-k-mer engine's
// bucket nodes, whose first split is by size like a root's)
__device__ __forceinline__ int plan_bits(const Node &nd, int level, u32 n_over, int l1_cap, int skew_from)
{
    const int rem = (int)(nd.meta & 0xff);
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
        } else if (level >= skew_from) {
            // An oversize node this deep is skew, not chance (planned leaves sit >= 4.5 sigma under the capacity):
            // typically heavy k-mers plus a leaf's worth of others.  Fan out wider than the size asks for, so
            // that a heavy key is soon alone (and its node recognised as constant by level_hist) -- as wide as
            // a budget of 2^19 new nodes over all the level's oversize nodes allows: a few such nodes split
            // 1024 ways (two more levels instead of ten), a hundred thousand of them only 4 ways (every child
            // is a leaves-kernel iteration: 1024 ways there cost 75 ms at 1 Gbase).
            const u32 over = n_over ? n_over : 1u;
            const int extra = 31 - __builtin_clz(((1u << 19) / over) | 1u);
            if (want < extra)
                want = extra;
        }
        bits = want;
        if (rem <= MAX_SPLIT_BITS) bits = rem;  // the rest of the key fits one digit: terminal split, no key moves
        else if (rem <= 2 * MAX_SPLIT_BITS && bits < rem - MAX_SPLIT_BITS)
            bits = rem - MAX_SPLIT_BITS;        // ... fits two: leave one digit, so the next level is terminal
        if (bits > MAX_SPLIT_BITS) bits = MAX_SPLIT_BITS;
        if (level >= 1 && bits > l1_cap) bits = l1_cap;
        if (bits > rem) bits = rem;
        if (bits < 1) bits = 1;
    }
    return bits;
}

// the plan's counters of one node into `ctr` (global or LDS)
__device__ __forceinline__ void plan_account(const Node &nd, int bits, LevelCounters *ctr)
{
    const int rem = (int)(nd.meta & 0xff);
    // (fixed addresses, so that the compiler folds each into one atomic per wave)
    const bool sorts = bits == 0 && nd.len > 0 && rem > 0 && !(nd.meta & NODE_TERMINAL);
    if (sorts && nd.len > (u32)LEAF_CAP_SMALL)
        atomicAdd(&ctr->n_big, 1u);
    if (sorts && nd.len <= (u32)LEAF_CAP_SMALL && nd.len > (u32)LEAF_CAP_TINY)
        atomicAdd(&ctr->n_small, 1u);
    if (sorts && nd.len <= (u32)LEAF_CAP_TINY)
        atomicAdd(&ctr->n_tiny, 1u);
    if (bits) {
        atomicMax(&ctr->max_bits, (u32)bits);
        atomicAdd(&ctr->n_split, 1u);
        if (bits < rem)
            atomicAdd(&ctr->n_scatter, 1u);
    }
}

__global__ __launch_bounds__(256) void plan_kernel(Node *__restrict__ nodes, u32 n_nodes, int level,
                                                   u32 chunk_len, u32 *__restrict__ outc,
                                                   u32 *__restrict__ nch, LevelCounters *__restrict__ ctr, int l1_cap,
                                                   int skew_from)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes)
        return;
    const Node nd = nodes[i];
    const int bits = plan_bits(nd, level, level >= 2 ? ctr->n_over : 0u, l1_cap, skew_from);
    plan_account(nd, bits, ctr);
    nodes[i].split = (u32)bits;
    outc[i] = bits ? (1u << bits) : 1u;
    nch[i] = bits ? (nd.len + chunk_len - 1) / chunk_len : 0u;
}

// Levels of at most 1024 nodes (the root, level 1, every level of the record engine's begin): the plan, both exclusive
// scans (children and chunks per node) and the level's counters in ONE launch of one workgroup -- instead of a memset,
// the plan and two three-kernel scans (~45 us of launches per level; a count at 100 Mbase has three such levels).
__global__ __launch_bounds__(1024) void plan_small_kernel(Node *__restrict__ nodes, u32 n_nodes, int level, u32 chunk_len,
                                                          u32 *__restrict__ outc, u32 *__restrict__ nch,
                                                          LevelCounters *__restrict__ ctr, int l1_cap, int skew_from)
{
    __shared__ LevelCounters lc;
    __shared__ u32 sa[1024], sb[1024], wtmp[16];
    const u32 i = threadIdx.x;
    if (i == 0) {
        lc.n_next = lc.n_chunks = lc.n_split = lc.n_scatter = lc.max_bits = 0;
        lc.n_big = lc.n_small = lc.n_tiny = lc.n_over = 0;
    }
    __syncthreads();
    Node nd;
    nd.len = 0;
    nd.meta = 0;
    if (i < n_nodes)
        nd = nodes[i];
    if (level >= 2) {
        if (i < n_nodes && nd.len > (u32)LEAF_CAP && (nd.meta & 0xff) > 0 && !(nd.meta & NODE_TERMINAL))
            atomicAdd(&lc.n_over, 1u);
        __syncthreads();
    }
    int bits = 0;
    if (i < n_nodes) {
        bits = plan_bits(nd, level, lc.n_over, l1_cap, skew_from);
        plan_account(nd, bits, &lc);
        nodes[i].split = (u32)bits;
    }
    sa[i] = i < n_nodes ? (bits ? (1u << bits) : 1u) : 0u;
    sb[i] = i < n_nodes && bits ? (nd.len + chunk_len - 1) / chunk_len : 0u;
    __syncthreads();
    const u32 ta = block_scan_inplace<1024>(sa, 1024, wtmp);
    const u32 tb = block_scan_inplace<1024>(sb, 1024, wtmp);
    if (i < n_nodes) {
        outc[i] = sa[i];
        nch[i] = sb[i];
    }
    if (i == 0) {
        lc.n_next = ta;
        lc.n_chunks = tb;
        *ctr = lc;
    }
}

// experiment (diagnostic build): cap the split width of levels >= 1.  Initialised once, thread-safe (the rank threads of
// dnagpu_count_multi* plan concurrently)
static int plan_l1_cap()
{
    static const int cap = [] {
        const char *e = diag_env("DNAGPU_L1_BITS");
        return e ? atoi(e) : MAX_SPLIT_BITS;
    }();
    return cap;
}

hipError_t launch_plan(Node *nodes, u32 n_nodes, int level, u32 chunk_len, u32 *outc, u32 *nch,
                       LevelCounters *ctr, hipStream_t s, int skew_from)
{
    if (n_nodes == 0)
        return hipSuccess;
    const int l1_cap = plan_l1_cap();
    if (level >= 2)
        hipLaunchKernelGGL(plan_count_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, nodes, n_nodes, ctr);
    hipLaunchKernelGGL(plan_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, nodes, n_nodes, level,
                       chunk_len, outc, nch, ctr, l1_cap, skew_from);
    return hipGetLastError();
}

// The whole planning step of a level: counters zeroed, every node planned, outc / nch turned into exclusive scans with
// their totals in ctr->n_next / ctr->n_chunks.  scan_tmp: scan_tmp_words(n_nodes) words.
hipError_t launch_plan_level(Node *nodes, u32 n_nodes, int level, u32 chunk_len, u32 *outc, u32 *nch, u32 *scan_tmp,
                             LevelCounters *ctr, hipStream_t s, int skew_from)
{
    if (n_nodes > 0 && n_nodes <= 1024) {
        const int l1_cap = plan_l1_cap();
        hipLaunchKernelGGL(plan_small_kernel, dim3(1), dim3(1024), 0, s, nodes, n_nodes, level, chunk_len, outc, nch, ctr, l1_cap,
                           skew_from);
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(ctr, 0, sizeof(LevelCounters), s);
    if (e == hipSuccess) e = launch_plan(nodes, n_nodes, level, chunk_len, outc, nch, ctr, s, skew_from);
    if (e == hipSuccess) e = launch_scan_u32(outc, outc, n_nodes, scan_tmp, &ctr->n_next, s);
    if (e == hipSuccess) e = launch_scan_u32(nch, nch, n_nodes, scan_tmp, &ctr->n_chunks, s);
    return e;
}

// (nodes and nodes_rw are the same array: no __restrict__ on them)
__global__ __launch_bounds__(256) void fill_chunks_kernel(const Node *nodes, u32 n_nodes,
                                                          u32 chunk_len, const u32 *__restrict__ child_base,
                                                          const u32 *__restrict__ chunk_base,
                                                          Node *nodes_rw, Chunk *__restrict__ chunks)
{
    // one node per blockIdx.x-row of 64 lanes: the lanes share the node's chunks (the dna root alone has thousands)
    const u32 i = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const u32 lane = threadIdx.x & 63;
    if (i >= n_nodes)
        return;
    const u32 cb = child_base[i], kb = chunk_base[i];
    if (lane == 0) {
        nodes_rw[i].child_base = cb;
        nodes_rw[i].chunk_base = kb;
    }
    if (nodes[i].split) {
        const u32 len = nodes[i].len;
        const u32 nc = (len + chunk_len - 1) / chunk_len;
        for (u32 c = lane + blockIdx.y * 64; c < nc; c += 64 * gridDim.y) {
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
    hipLaunchKernelGGL(fill_chunks_kernel, dim3((n_nodes + 3) / 4, n_nodes < 64 ? 16 : 1), dim3(256), 0, s, nodes, n_nodes,
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

// The level-0 DIGITS of 16 consecutive windows from one 64-bit funnel: digit j of the window at
// pos + j is bits [2(pos+j) + shift, +bits) of the packed stream, so all 16 sit in one 64-bit
// value at stride 2 (2*15 + bits <= 64).  Histograms, the counting sweep and the owner filter need
// only the digit -- about a tenth of the work of extracting sixteen full keys.
__device__ __forceinline__ u64 dig16_load(const u64 *__restrict__ words, u64 n_words, u64 pos, int shift)
{
    const u64 bit0 = 2 * pos + (u64)shift;
    const u64 w = bit0 >> 6;
    const u64 lo = w < n_words ? words[w] : 0;
    const u64 hi = w + 1 < n_words ? words[w + 1] : 0;
    return funnel(lo, hi, (unsigned)(bit0 & 63));
}

// Owner filter of a sharded count: at the dna root only the keys whose level-0 digit d satisfies
// (d - lo) < span (unsigned) exist; span = ~0 keeps everything.
struct DigitFilter {
    u32 lo, span;
    u32 tb;         // > 0: span is a power of two and lo a multiple of it: only the digit's top tb bits decide
};

// Which of the 16 windows of a dig16_load belong to the owner: bit 2j set = window j kept (j < nvalid).
// With an aligned power-of-two range (every power-of-two GPU count) the test is bit-parallel: the
// top tb bits of all 16 digits are compared in 2*tb + 2 operations on the 64-bit funnel, instead of
// a shift, a mask, a subtract and a compare per window -- the owner sweep visits W times more
// windows than it keeps, so this test is most of a sharded level 0.
__device__ __forceinline__ u32 owner_bits16(u64 dv, int bits, u32 dmask, const DigitFilter &flt, u32 nvalid)
{
    const u32 vmask = nvalid >= 16 ? 0x55555555u : (u32)(((u64)1 << (2 * nvalid)) - 1) & 0x55555555u;
    if (flt.tb) {
        const int sh = bits - (int)flt.tb;
        const u64 y = dv >> sh;
        const u32 pat = flt.lo >> sh;
        u32 m = vmask;
        for (u32 b = 0; b < flt.tb; b++)
            m &= ((pat >> b) & 1) ? (u32)(y >> b) : ~(u32)(y >> b);
        return m;
    }
    u32 m = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const u32 d = (u32)(dv >> (2 * j)) & dmask;
        m |= (d - flt.lo < flt.span) ? (1u << (2 * j)) : 0u;
    }
    return m & vmask;
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
                                                                u32 *__restrict__ hist, DigitFilter flt,
                                                                u32 *__restrict__ vary_all, int multi_ref,
                                                                u32 stat_min_len)
{
    __shared__ u32 h[ROW_STRIDE];
    __shared__ u32 sh_vary, sh_eq, sh_eq1, sh_eq2;
    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    const int bits = (int)nd.split;
    const int shift = (int)(nd.meta & 0xff) - bits;
    const u32 R = 1u << bits, dmask = R - 1;
    // node statistics (below) only for nodes of at least stat_min_len keys: at level 1 that is "far above
    // the mean" -- nothing in uniform data, whose histogram then costs what it did without them
    u32 *__restrict__ vary = nd.len >= stat_min_len ? vary_all : nullptr;
    for (u32 d = threadIdx.x; d < R; d += SC_THREADS)
        h[d] = 0;
    __syncthreads();
    const u64 origin = (u64)nd.start + ch.off;
    if (SRC_DNA) {
        for (u32 i0 = threadIdx.x * 16; i0 < ch.len; i0 += SC_THREADS * 16) {
            const u64 dv = dig16_load(words, n_words, first + origin + i0, shift);
            if (flt.span == ~0u) {                  // no owner filter: every window counts
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    u32 d = (u32)(dv >> (2 * j)) & dmask;
                    if (i0 + j < ch.len)
                        atomicAdd(&h[d], 1u);
                }
            } else {
                u32 m = owner_bits16(dv, bits, dmask, flt, ch.len - i0);
                while (m) {
                    const int j2 = __ffs((int)m) - 1;
                    m &= m - 1;
                    atomicAdd(&h[(u32)(dv >> j2) & dmask], 1u);
                }
            }
        }
    } else {
        // Besides the digit counts: how many low key bits VARY inside the node (64 - clz of the OR of
        // key ^ the node's first key).  A node whose keys all share this level's digit -- many copies of
        // one k-mer, or of k-mers that differ only further down -- is then not moved at all: its single
        // child stays in place and resumes at the first varying bit (level_children), terminal if none varies.
        const u64 *__restrict__ src = ((nd.meta & NODE_BUF) ? buf1 : buf0) + origin;
        const u64 ref = ((nd.meta & NODE_BUF) ? buf1 : buf0)[nd.start];    // the node's first key
        // ... and how many keys EQUAL that first key: a node dominated by it (half its keys or more) is split
        // three ways around it instead (level_children, peel_scatter_kernel).
        // Deep levels (multi_ref: every node there is oversize because of skew) try three candidates -- the
        // node's first, middle and last key -- so that a heavy key is missed only if it is none of them.
        if (threadIdx.x == 0) {
            sh_vary = 0;
            sh_eq = 0;
            sh_eq1 = 0;
            sh_eq2 = 0;
        }
        u64 v = 0;
        u32 neq = 0, neq1 = 0, neq2 = 0;
        if (multi_ref) {
            const u64 *__restrict__ nb = (nd.meta & NODE_BUF) ? buf1 : buf0;
            const u64 ref1 = nb[nd.start + nd.len / 2], ref2 = nb[nd.start + nd.len - 1];
            for (u32 i = threadIdx.x; i < ch.len; i += SC_THREADS) {
                const u64 key = NT_LOAD(&src[i]);
                const u64 x = key ^ ref;
                v |= x;
                neq += x == 0 ? 1u : 0u;
                neq1 += key == ref1 ? 1u : 0u;
                neq2 += key == ref2 ? 1u : 0u;
                atomicAdd(&h[(u32)(key >> shift) & dmask], 1u);
            }
        } else if (vary) {
            for (u32 i = threadIdx.x; i < ch.len; i += SC_THREADS) {
                const u64 key = NT_LOAD(&src[i]);
                const u64 x = key ^ ref;
                v |= x;
                neq += x == 0 ? 1u : 0u;
                atomicAdd(&h[(u32)(key >> shift) & dmask], 1u);
            }
        } else {
            for (u32 i = threadIdx.x; i < ch.len; i += SC_THREADS)
                atomicAdd(&h[(u32)(NT_LOAD(&src[i]) >> shift) & dmask], 1u);
        }
        if (vary) {
            __syncthreads();
            const u32 hb = v ? 64u - (u32)__builtin_clzll(v) : 0u;
            if (hb > 0)                                 // (skipped by whole workgroups of identical keys)
                atomicMax(&sh_vary, hb);
            const u32 weq = wave_sum(neq);
            if ((threadIdx.x & 63) == 0 && weq)
                atomicAdd(&sh_eq, weq);
            if (multi_ref) {
                const u32 weq1 = wave_sum(neq1), weq2 = wave_sum(neq2);
                if ((threadIdx.x & 63) == 0) {
                    if (weq1)
                        atomicAdd(&sh_eq1, weq1);
                    if (weq2)
                        atomicAdd(&sh_eq2, weq2);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                u32 *st = vary + (size_t)ch.node * NODE_STAT_WORDS;
                if (sh_vary > 0)
                    atomicMax(&st[0], sh_vary);
                if (sh_eq)
                    atomicAdd(&st[2], sh_eq);
                if (sh_eq1)
                    atomicAdd(&st[5], sh_eq1);
                if (sh_eq2)
                    atomicAdd(&st[6], sh_eq2);
            }
        }
    }
    __syncthreads();
    u32 *row = hist + (u64)blockIdx.x * ROW_STRIDE;
    for (u32 d = threadIdx.x; d < R; d += SC_THREADS)
        row[d] = h[d];
}

hipError_t launch_level_hist(const Node *nodes, const Chunk *chunks, u32 n_chunks, int src_dna,
                             const u64 *words, u64 n_words, u64 first, int k, const u64 *buf0,
                             const u64 *buf1, u32 *hist, u32 flt_lo, u32 flt_span, u32 flt_tb, u32 *vary, int multi_ref,
                             u32 stat_min_len, hipStream_t s)
{
    const DigitFilter flt{flt_lo, flt_span, flt_tb};
    if (n_chunks == 0)
        return hipSuccess;
    if (src_dna)
        hipLaunchKernelGGL(level_hist_kernel<true>, dim3(n_chunks), dim3(SC_THREADS), 0, s, nodes, chunks,
                           n_chunks, words, n_words, first, kmer_mask(k), buf0, buf1, hist, flt, (u32 *)nullptr, 0, 0u);
    else
        hipLaunchKernelGGL(level_hist_kernel<false>, dim3(n_chunks), dim3(SC_THREADS), 0, s, nodes, chunks,
                           n_chunks, words, n_words, first, kmer_mask(k), buf0, buf1, hist, flt, vary, multi_ref, stat_min_len);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// level_prefix: for every split node and 64-digit group, turn the node's chunk rows into exclusive
// prefixes over chunks (per digit) and store the per-digit totals in tot[row of the first chunk].
// Grid = (n_chunks, ROW_STRIDE/64); only the workgroup of a node's first chunk works.
constexpr int PF_SLICES = 16;                           // chunk slices per digit (threads = 64 digits x slices)
// DIG digits per workgroup: 64 for levels of many nodes; 16 when one node holds thousands of chunks (a root): four times
// the workgroups and a quarter of the chunks per thread -- the kernel is a chain of dependent L2 reads per thread (the
// root's 4096 rows at 64 x 16: 85 - 116 us per level; at 16 x 64: ~25 us).
template <int DIG>
__global__ __launch_bounds__(1024) void level_prefix_kernel(int n_slices, const Node *__restrict__ nodes,
                                                           const Chunk *__restrict__ chunks, u32 n_chunks,
                                                           u32 chunk_len, u32 *__restrict__ hist,
                                                           u32 *__restrict__ tot, u32 n_by_node)
{
    __shared__ u32 part[1024 / DIG][DIG];
    // blockIdx.x = a chunk (only a node's first chunk works), or -- n_by_node > 0: levels of few nodes with many chunks
    // each, where most of a chunk-indexed grid would only look and leave -- a node
    u32 c0 = blockIdx.x;
    Node nd;
    if (n_by_node) {
        if (c0 >= n_by_node)
            return;
        nd = nodes[c0];
        if (nd.split == 0 || nd.len == 0)
            return;
        c0 = nd.chunk_base;
    } else {
        if (c0 >= n_chunks)
            return;
        const Chunk ch = chunks[c0];
        if (ch.off != 0)
            return;                               // not the node's first chunk
        nd = nodes[ch.node];
    }
    const u32 R = 1u << nd.split;
    const u32 dl = threadIdx.x % DIG;
    const u32 d = blockIdx.y * DIG + dl;
    if (blockIdx.y * DIG >= R)
        return;
    const u32 nc = (nd.len + chunk_len - 1) / chunk_len;
    const u32 slice = threadIdx.x / DIG;
    const u32 per = (nc + n_slices - 1) / n_slices;
    const u32 cb = slice * per < nc ? slice * per : nc;
    const u32 ce = cb + per < nc ? cb + per : nc;
    const bool live = d < R;
    u32 sum = 0;
    if (live)
        for (u32 c = cb; c < ce; c++)
            sum += hist[(u64)(c0 + c) * ROW_STRIDE + d];
    part[slice][dl] = sum;
    __syncthreads();
    u32 base = 0, total = 0;
    for (u32 s = 0; s < (u32)n_slices; s++) {
        u32 t = part[s][dl];
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
// vary != null (key-source levels): a node whose keys all share this level's digit is not scattered
// (NODE_SKIP); its one non-empty child is the node itself, in place, with only the varying low bits left.
__global__ __launch_bounds__(256) void level_children_kernel(Node *nodes, u32 n_nodes,
                                                             u32 *__restrict__ tot, Node *__restrict__ next,
                                                             u32 *__restrict__ vary,
                                                             const u64 *__restrict__ buf0,
                                                             const u64 *__restrict__ buf1, u32 stat_min_len)
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
    u32 *st = vary && nd.len >= stat_min_len ? vary + (size_t)i * NODE_STAT_WORDS : nullptr;   // (as level_hist)
    const u32 hv = st ? st[0] : 64u;
    u32 n_eq = st ? st[2] : 0u;
    u32 ridx = 0;                                 // the candidate with most copies (deep levels count three)
    if (st && st[5] > n_eq) {
        n_eq = st[5];
        ridx = 1;
    }
    if (st && st[6] > n_eq) {
        n_eq = st[6];
        ridx = 2;
    }
    // all keys equal, or half of them (or more) equal to the first key with a different one among them: the
    // node is split around that key -- {below, the key itself (terminal), above} take the first three child
    // slots, in key order; the heavy key's copies are not moved again.  Else, all keys in one digit: in place.
    const bool peel = st && nd.len > 0 && bits >= 2 && bits < rem && hv > 0 && 2ull * n_eq >= nd.len;
    const bool stay = !peel && nd.len > 0 && hv <= (u32)(rem - bits) && bits < rem;
    u64 ref = 0;
    if (stay || peel) {
        const u32 roff = !peel || ridx == 0 ? 0u : (ridx == 1 ? nd.len / 2 : nd.len - 1);
        ref = ((nd.meta & NODE_BUF) ? buf1 : buf0)[nd.start + roff];
        if (threadIdx.x == 0) {
            nodes[i].meta = nd.meta | (peel ? NODE_PEEL : NODE_SKIP);
            if (peel) {
                st[3] = 0;                        // cursors of the two moved parts (peel_scatter_kernel)
                st[4] = 0;
                st[7] = roff;                     // where the pivot key is
            }
        }
    }
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
            if (stay && c.len) {                  // the single non-empty child: the node's own keys, where they are
                c.meta = hv | (nd.meta & NODE_BUF) | (hv == 0 ? NODE_TERMINAL : 0u);
                c.prefix = (ref >> hv) << hv;
            }
            if (peel) {
                c.prefix = nd.prefix;             // the moved parts keep every free bit of the node
                c.meta = (u32)rem | ((nd.meta & NODE_BUF) ^ NODE_BUF);
                if (d == 0 || d == 2) {           // below / above: sizes known once the keys have moved (peel_fix_kernel)
                    c.start = nd.start;
                    c.len = 0;
                } else if (d == 1) {
                    c.start = nd.start;
                    c.len = n_eq;
                    c.meta = NODE_TERMINAL;
                    c.prefix = ref;
                } else {
                    c.start = nd.start + nd.len;
                    c.len = 0;
                }
            }
            c.child_base = 0;
            c.chunk_base = 0;
            next[nd.child_base + d] = c;
            row[d] = nd.start + ex[d];
        }
    }
}

hipError_t launch_level_children(Node *nodes, u32 n_nodes, u32 *hist, Node *next, u32 *vary, const u64 *buf0,
                                 const u64 *buf1, u32 stat_min_len, hipStream_t s)
{
    // `hist` here is the tot table (same geometry as the hist table); the prefix kernel is launched
    // separately by the host through launch_level_prefix
    if (n_nodes == 0)
        return hipSuccess;
    hipLaunchKernelGGL(level_children_kernel, dim3(n_nodes), dim3(256), 0, s, nodes, n_nodes, hist, next, vary, buf0, buf1,
                       stat_min_len);
    return hipGetLastError();
}

hipError_t launch_level_prefix(const Node *nodes, const Chunk *chunks, u32 n_chunks, u32 n_split_nodes, u32 chunk_len,
                               u32 *hist, u32 *tot, hipStream_t s, u32 n_nodes)
{
    if (n_chunks == 0)
        return hipSuccess;
    // few nodes with many chunks each (the dna root): 64 chunk slices of 16 digits; else 64 digits x 16 or 4 slices
    if (n_split_nodes == 1 && n_chunks > 512) {       // (one split node: its chunks are all the chunks, the first is chunk 0)
        hipLaunchKernelGGL(level_prefix_kernel<16>, dim3(1, ROW_STRIDE / 16), dim3(1024), 0, s, 64, nodes, chunks, n_chunks,
                           chunk_len, hist, tot, 0u);
        return hipGetLastError();
    }
    const int slices = n_split_nodes > 0 && n_chunks / n_split_nodes > 32 ? PF_SLICES : 4;
    const u32 by_node = n_nodes > 0 && n_nodes * 4 <= n_chunks ? n_nodes : 0u;      // (the record engine's level 1: 256 nodes, 4 K chunks)
    hipLaunchKernelGGL(level_prefix_kernel<64>, dim3(by_node ? by_node : n_chunks, ROW_STRIDE / 64), dim3(64 * slices), 0, s, slices,
                       nodes, chunks, n_chunks, chunk_len, hist, tot, by_node);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// level_scatter with aligned write combining (the default).
//
// Measured on MI355X (tools/microbench/scatter_runs.hip): scattered 8-byte-granular runs of 64 B
// reach 0.84 TB/s, the same bytes as 64-byte ALIGNED units 2.7 TB/s.  So a digit's keys leave the
// tile only in whole 64-byte units of its destination: after staging, digit d flushes the keys up
// to the last 64-byte boundary of its output stream and carries the rest (< 8 keys) into the next
// tile.  Thread d owns digit d's carry in registers (R <= 1024 = workgroup size), so the LDS holds
// one 16,384-key stage (128 KB) plus small tables; a tile takes as many new keys as the stage has
// room for beside the carried ones.  (Storing the carried keys straight from registers instead of
// re-staging them frees that room but was measured 3.5x slower: 8-byte stores to 1024 different
// lines per wave-instruction.)
constexpr int WC_THREADS = 1024;
constexpr int WC_ITEMS_KEYS = 16;                       // 16384 staged keys when the source is a key buffer
constexpr int WC_PAD_SLOTS = 1024;                      // stage slots beyond the tile: parity padding (see below)
constexpr size_t wc_smem(int items, int nth = WC_THREADS)
{
    return (size_t)(nth * items + nth) * 8 + (size_t)(nth + 8) * 4 * 5 + 32 * 4;
}
typedef unsigned long long ull2_t __attribute__((ext_vector_type(2)));

// Write-out of a staged tile, two slots (16 bytes) per lane.  Measured with per-phase cycle stamps:
// with one 8-byte store per lane the write-out was 55 % of the scatter's time and ran at 5.4 B/clk
// per CU -- the store path is bound by store INSTRUCTIONS, so each lane stores 16 bytes.  For that,
// every digit's run starts in the stage on the parity of its destination (one padding slot at most
// before, one after: regions have even starts and sizes), so an even slot pair is a 16-byte aligned
// pair of the destination whenever both slots belong to the same flushed run.
// flsh[d] = {delta, e0 | f << 16}: staged slot i of digit d goes to dst[i + delta] while i - e0 < f.
template <int NTH>
__device__ __forceinline__ void wc_write_out(const u64 *stage, const u32 *flsh, u32 staged, u64 *__restrict__ dst,
                                             int shift, u32 dmask, int dbg, u64 lin = 0)
{
#pragma unroll 4
    for (u32 g = threadIdx.x; 2 * g < staged; g += NTH) {
        const u32 i = 2 * g;
        const ull2_t kk = *reinterpret_cast<const ull2_t *>(&stage[i]);
        const u32 d0 = (u32)(kk.x >> shift) & dmask, d1 = (u32)(kk.y >> shift) & dmask;
        const uint2 f0 = reinterpret_cast<const uint2 *>(flsh)[d0];
        const uint2 f1 = reinterpret_cast<const uint2 *>(flsh)[d1];
        const bool v0 = (u32)(i - (f0.y & 0xffffu)) < (f0.y >> 16);
        const bool v1 = (u32)(i + 1 - (f1.y & 0xffffu)) < (f1.y >> 16);
        if (dbg & 1)
            continue;
        if (dbg & 2) {                              // timing ablation: same bytes, consecutive addresses
            __builtin_nontemporal_store(kk, reinterpret_cast<ull2_t *>(&dst[lin + i]));
            continue;
        }
        if (v0 && v1 && d0 == d1) {
            __builtin_nontemporal_store(kk, reinterpret_cast<ull2_t *>(&dst[(u64)(u32)(i + f0.x)]));
        } else {
            if (v0)
                NT_STORE((u64)kk.x, &dst[(u64)(u32)(i + f0.x)]);
            if (v1)
                NT_STORE((u64)kk.y, &dst[(u64)(u32)(i + 1 + f1.x)]);
        }
    }
}

template <bool SRC_DNA, int WC_ITEMS, int NTH, int FLUSH>
__global__ __launch_bounds__(NTH, 4) void level_scatter_wc_kernel(const Node *__restrict__ nodes,
                                                                         const Chunk *__restrict__ chunks,
                                                                         u32 n_chunks,
                                                                         const u64 *__restrict__ words, u64 n_words,
                                                                         u64 first, u64 mask,
                                                                         u64 *__restrict__ buf0,
                                                                         u64 *__restrict__ buf1,
                                                                         const u32 *__restrict__ hist,
                                                                         const u32 *__restrict__ tot, int dbg)
{
    constexpr int WC_TILE = NTH * WC_ITEMS;
    constexpr int TS = NTH + 8;                    // table stride: a workgroup of NTH threads splits on at most NTH digits
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 WC_STAGE = WC_TILE + NTH;
    u64 *stage = reinterpret_cast<u64 *>(smem);                          // WC_STAGE slots
    u32 *excl = reinterpret_cast<u32 *>(smem + (size_t)WC_STAGE * 8);    // R + 2 (counts, then padded offsets)
    u32 *offs = excl + TS;                                   // next output index per digit
    u32 *flsh = offs + TS;                                   // {delta, limit} per digit (uint2)
    u32 *curs = flsh + 2 * TS;                             // staging cursor per digit
    u32 *wtmp = curs + TS;                                   // 16 + carry total

    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    const int bits = (int)nd.split;
    const int rem = (int)(nd.meta & 0xff);
    if (bits == rem || (nd.meta & (NODE_SKIP | NODE_PEEL)))
        return;                                   // terminal split: nothing moves
    const int shift = rem - bits;
    const u32 R = 1u << bits, dmask = R - 1;
    const u32 tid = threadIdx.x;
    const u64 origin = (u64)nd.start + ch.off;
    const u64 *__restrict__ src = SRC_DNA ? nullptr : (((nd.meta & NODE_BUF) ? buf1 : buf0) + origin);
    u64 *__restrict__ dst = SRC_DNA ? buf0 : ((nd.meta & NODE_BUF) ? buf0 : buf1);
    const u32 abase = (u32)((reinterpret_cast<uintptr_t>(dst) >> 3) & (u32)(FLUSH - 1));   // phase of the buffer inside a flush unit
    u32 *sh_carry = wtmp + 16;                    // carried keys in total (sets the next tile's size)

    if (tid < R)
        offs[tid] = hist[(u64)blockIdx.x * ROW_STRIDE + tid] + tot[(u64)nd.chunk_base * ROW_STRIDE + tid];
    if (tid == 0)
        *sh_carry = 0;
    u64 carry[FLUSH - 1];
    u32 ccnt = 0;
#pragma unroll
    for (int c = 0; c < FLUSH - 1; c++)
        carry[c] = 0;

    // A tile = the carried keys + as many new keys as still fit the 16,384-slot stage.  The new keys
    // of tile t+1 are requested as soon as tile t is staged, so their latency hides behind tile t's
    // write-out (one workgroup per CU: nobody else would hide it).
    u32 t0 = 0;
    // (a tile leaves room for the parity padding: at most two slots per digit)
    const u32 room0 = WC_STAGE - 2 * R < (u32)WC_TILE ? WC_STAGE - 2 * R : (u32)WC_TILE;
    // Odd workgroups open with a half tile: neighbouring CUs then alternate between their LDS phases
    // and their write-out instead of all storing (and all not storing) at the same time.
    const u32 first_tile = ((blockIdx.x & 1) && !(dbg & 32)) ? room0 / 2 : room0;
    u32 tn = ch.len < first_tile ? ch.len : first_tile;
    u32 per = (tn + NTH - 1) / NTH;         // dna root: consecutive windows per thread
    u64 key[SRC_DNA ? 1 : WC_ITEMS];
    Win16 w;
    if (SRC_DNA) {
        w = win16_load(words, n_words, first + origin + tid * per);
    } else {
#pragma unroll
        for (int j = 0; j < WC_ITEMS; j++) {
            u32 i = tid + j * NTH;
            key[SRC_DNA ? 0 : j] = NT_LOAD(&src[i < tn ? i : tn - 1]);
        }
    }

    STAMP_DECL
    STAMP(0);                                       // prologue: descriptors, first tile's loads issued
    for (;;) {
        const bool last = t0 + tn >= ch.len;
        // digit counters start at the carried count, so ranks of new keys land behind the carry
        if (tid < R)
            excl[tid] = ccnt;
        if (tid == 0)
            excl[R] = 0;
        __syncthreads();
        STAMP(1);                                   // counter init + barrier
        // count (non-returning LDS adds: no rank registers are kept); the dna root needs only the
        // windows' digits here: one funnel for all 16
        const u64 dv = SRC_DNA ? dig16_load(words, n_words, first + origin + t0 + tid * per, shift) : 0;
#pragma unroll
        for (int j = 0; j < WC_ITEMS; j++) {
            u32 i = SRC_DNA ? ((u32)j < per ? tid * per + j : tn) : tid + j * NTH;
            u32 d = SRC_DNA ? (u32)(dv >> (2 * j)) & dmask : (u32)(key[SRC_DNA ? 0 : j] >> shift) & dmask;
            if (i < tn)                             // (no dummy digit: same-address LDS atomics serialise)
                atomicAdd(&excl[d], 1u);
        }
        __syncthreads();
        STAMP(2);                                   // count (incl. waiting for the tile's keys)
        // digit d's region: even start, even size, first key on the parity of its destination
        u32 m = 0, par = 0, e0 = 0;
        if (tid < R) {
            m = excl[tid];                          // carried + new keys of this digit
            par = (abase + offs[tid]) & 1u;
            excl[tid] = (m + par + 1u) & ~1u;       // (read back by this thread only: no barrier needed)
        }
        {
            const u32 staged_total = block_scan_inplace<NTH>(excl, (int)R, wtmp);
            if (tid == 0)
                excl[R] = staged_total;
        }
        // digit d's cursor starts behind its carried keys; thread d re-stages those itself
        if (tid < R) {
            e0 = excl[tid] + par;
            curs[tid] = e0 + ccnt;
        }
        __syncthreads();
        STAMP(3);                                   // scan + cursors
        // place: a returning add on the digit's cursor is the staged slot.  All the adds first, then
        // all the stores: written as one loop, every store waited for its own add's round trip.
        u32 slot[WC_ITEMS];
        u64 kvs[SRC_DNA ? WC_ITEMS : 1];            // dna root: the extracted windows are kept for the stores
#pragma unroll
        for (int j = 0; j < WC_ITEMS; j++) {
            u32 i = SRC_DNA ? ((u32)j < per ? tid * per + j : tn) : tid + j * NTH;
            u64 kv = SRC_DNA ? win16_key(w, j, mask) : key[SRC_DNA ? 0 : j];
            if (SRC_DNA)
                kvs[SRC_DNA ? j : 0] = kv;
            slot[j] = ~0u;
            if (i < tn)
                slot[j] = atomicAdd(&curs[(u32)(kv >> shift) & dmask], 1u);
        }
#pragma unroll
        for (int j = 0; j < WC_ITEMS; j++) {
            u64 kv = SRC_DNA ? kvs[SRC_DNA ? j : 0] : key[SRC_DNA ? 0 : j];
            if (slot[j] != ~0u)
                stage[slot[j]] = kv;
        }
        __syncthreads();                            // excl[R] (thread 0) and every stage write are visible
        STAMP(4);                                   // place
        u32 f = 0;
        if (tid < R) {
#pragma unroll
            for (int c = 0; c < FLUSH - 1; c++)
                if ((u32)c < ccnt)
                    stage[e0 + c] = carry[c];
            const u32 o = offs[tid];
            const u32 tail = (abase + o + m) & (u32)(FLUSH - 1);  // keys past the last flush-unit boundary
            f = last ? m : (m >= tail ? m - tail : 0u);
            reinterpret_cast<uint2 *>(flsh)[tid] = make_uint2(o - e0, e0 | (f << 16));
        }
        {   // total carried into the next tile
            const u32 cn = wave_sum(m - f);
            if ((tid & 63) == 0 && cn)
                atomicAdd(sh_carry, cn);
        }
        __syncthreads();
        STAMP(5);                                   // flush table, carry re-stage
        const u32 t0n = t0 + tn;
        const u32 room = room0 - *sh_carry;
        const u32 tnn = ch.len - t0n < room ? ch.len - t0n : room;
        if (tnn > 0) {                              // request the next tile's keys now
            per = (tnn + NTH - 1) / NTH;
            if (SRC_DNA) {
                w = win16_load(words, n_words, first + origin + t0n + tid * per);
            } else {
#pragma unroll
                for (int j = 0; j < WC_ITEMS; j++) {
                    u32 i = tid + j * NTH;
                    key[SRC_DNA ? 0 : j] = NT_LOAD(&src[t0n + (i < tnn ? i : tnn - 1)]);
                }
            }
        }
        STAMP(6);                                   // next tile's loads issued
        wc_write_out<NTH>(stage, flsh, excl[R], dst, shift, dmask, dbg, (origin + t0) & ~(u64)1);
        STAMP(7);                                   // write-out
        if (tid < R) {
            ccnt = m - f;                           // <= 7
#pragma unroll
            for (int c = 0; c < FLUSH - 1; c++)
                if ((u32)c < ccnt)
                    carry[c] = stage[e0 + f + c];
        }
        __syncthreads();
        STAMP(8);                                   // carry reload + barrier
        if (tid < R)
            offs[tid] += f;
        if (tid == 0)
            *sh_carry = 0;
        if (tnn == 0)
            break;
        t0 = t0n;
        tn = tnn;
    }
    STAMP_FLUSH(SRC_DNA ? 0 : 12);
}

// ------------------------------------------------------------------------------------------------
// The dna root's scatter: same aligned write combining, but a tile is filled from as many positions
// of the packed sequence as it takes.  Without an owner filter that is "as many as the stage has
// room for"; with one (sharded count: this GPU keeps 1/W of the key space and scans the whole
// sequence) a tile sweeps about W x 16,384 positions, counting and then placing only its own keys.
// Windows are recomputed in both sweeps (three packed words per 16 windows, a few VALU ops each):
// far cheaper than a second LDS buffer.  If the keys found exceed the room (skewed data), the sweep
// is halved and recounted; a sweep of `room` positions always fits.
__global__ __launch_bounds__(WC_THREADS, 4) void level_scatter_wc_dna_kernel(
    const Node *__restrict__ nodes, const Chunk *__restrict__ chunks, u32 n_chunks,
    const u64 *__restrict__ words, u64 n_words, u64 first, u64 mask, u64 *__restrict__ buf0,
    const u32 *__restrict__ hist, const u32 *__restrict__ tot, DigitFilter flt, int dbg)
{
    constexpr int WC_TILE = WC_THREADS * 16;
    constexpr u32 BATCH = WC_THREADS * 16;        // positions per sweep step: 16 windows per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 WC_STAGE = WC_TILE + WC_PAD_SLOTS;
    u64 *stage = reinterpret_cast<u64 *>(smem);
    u32 *excl = reinterpret_cast<u32 *>(smem + (size_t)WC_STAGE * 8);
    u32 *offs = excl + ROW_STRIDE + 8;
    u32 *flsh = offs + ROW_STRIDE + 8;
    u32 *curs = flsh + 2 * (ROW_STRIDE + 8);
    u32 *wtmp = curs + ROW_STRIDE + 8;
    u32 *sh_carry = wtmp + 16;

    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    const int bits = (int)nd.split;
    const int rem = (int)(nd.meta & 0xff);
    if (bits == rem || (nd.meta & (NODE_SKIP | NODE_PEEL)))
        return;                                   // terminal split: nothing moves
    const int shift = rem - bits;
    const u32 R = 1u << bits, dmask = R - 1;
    const u32 tid = threadIdx.x;
    const u64 origin = (u64)nd.start + ch.off;
    u64 *__restrict__ dst = buf0;
    const u32 abase = (u32)((reinterpret_cast<uintptr_t>(dst) >> 3) & 7u);
    // positions swept per key kept, about (the filter keeps span of R digits)
    const u32 inv = flt.span >= R ? 1u : R / (flt.span ? flt.span : 1u);

    if (tid < R)
        offs[tid] = hist[(u64)blockIdx.x * ROW_STRIDE + tid] + tot[(u64)nd.chunk_base * ROW_STRIDE + tid];
    if (tid == 0)
        *sh_carry = 0;
    u64 carry[7];
    u32 ccnt = 0;
#pragma unroll
    for (int c = 0; c < 7; c++)
        carry[c] = 0;
    __syncthreads();

    u32 t0 = 0;
    while (t0 < ch.len) {
        // (room for the parity padding of wc_write_out: at most two slots per digit)
        const u32 room0 = WC_STAGE - 2 * R < (u32)WC_TILE ? WC_STAGE - 2 * R : (u32)WC_TILE;
        const u32 room = room0 - *sh_carry;
        const u32 remaining = ch.len - t0;
        u32 P = room * inv;                       // positions of this tile
        if (P > room)
            P = P / BATCH * BATCH;
        if (P > remaining)
            P = remaining;
        u32 staged_total, m = 0, par = 0, e0 = 0;
        for (;;) {
            __syncthreads();                      // previous readers of excl are done
            if (tid < R)
                excl[tid] = ccnt;                 // counters start behind the carried keys
            if (tid == 0) {
                excl[R] = 0;
                *sh_carry = 0;                    // (every thread has read `room`)
            }
            __syncthreads();
            for (u32 b = 0; b < P; b += BATCH) {  // count sweep
                const u32 bl = P - b < BATCH ? P - b : BATCH;
                const u32 per = (bl + WC_THREADS - 1) / WC_THREADS;
                const u64 dv = dig16_load(words, n_words, first + origin + t0 + b + tid * per, shift);
                const u32 i0 = tid * per;
                u32 mk = owner_bits16(dv, bits, dmask, flt, i0 < bl ? (bl - i0 < per ? bl - i0 : per) : 0u);
                while (mk) {
                    const int j2 = __ffs((int)mk) - 1;
                    mk &= mk - 1;
                    atomicAdd(&excl[(u32)(dv >> j2) & dmask], 1u);
                }
            }
            __syncthreads();
            if (tid < R) {                          // even start, even size, first key on its destination's parity
                m = excl[tid];
                par = (abase + offs[tid]) & 1u;
                excl[tid] = (m + par + 1u) & ~1u;
            }
            staged_total = block_scan_inplace<WC_THREADS>(excl, (int)R, wtmp);
            if (staged_total <= WC_STAGE || P <= room)
                break;
            P = P / 2;                            // more keys than room: sweep half as far
            if (P < room)
                P = room;
            else if (P > BATCH)
                P = P / BATCH * BATCH;
        }
        const bool last = t0 + P >= ch.len;
        if (tid == 0)
            excl[R] = staged_total;
        if (tid < R) {
            e0 = excl[tid] + par;
            curs[tid] = e0 + ccnt;
        }
        __syncthreads();
        for (u32 b = 0; b < P; b += BATCH) {      // place sweep: same windows, same filter
            const u32 bl = P - b < BATCH ? P - b : BATCH;
            const u32 per = (bl + WC_THREADS - 1) / WC_THREADS;
            const u64 pos0 = first + origin + t0 + b + tid * per;
            const u64 dv = dig16_load(words, n_words, pos0, shift);
            const u32 i0 = tid * per;
            // bit 2j: window j is this owner's
            u32 acc = owner_bits16(dv, bits, dmask, flt, i0 < bl ? (bl - i0 < per ? bl - i0 : per) : 0u);
            if (acc) {                            // full keys only for the windows that are kept
                const Win16 w = win16_load(words, n_words, pos0);
                while (acc) {
                    const int j = (__ffs((int)acc) - 1) >> 1;
                    acc &= acc - 1;
                    const u64 kv = win16_key(w, j, mask);
                    stage[atomicAdd(&curs[(u32)(kv >> shift) & dmask], 1u)] = kv;
                }
            }
        }
        __syncthreads();
        u32 f = 0;
        if (tid < R) {
#pragma unroll
            for (int c = 0; c < 7; c++)
                if ((u32)c < ccnt)
                    stage[e0 + c] = carry[c];
            const u32 o = offs[tid];
            const u32 tail = (abase + o + m) & 7u;  // keys past the last 64-byte boundary
            f = last ? m : (m >= tail ? m - tail : 0u);
            reinterpret_cast<uint2 *>(flsh)[tid] = make_uint2(o - e0, e0 | (f << 16));
        }
        {   // total carried into the next tile
            const u32 cn = wave_sum(m - f);
            if ((tid & 63) == 0 && cn)
                atomicAdd(sh_carry, cn);
        }
        __syncthreads();
        wc_write_out<WC_THREADS>(stage, flsh, staged_total, dst, shift, dmask, dbg);
        if (tid < R) {
            ccnt = m - f;                           // <= 7
#pragma unroll
            for (int c = 0; c < 7; c++)
                if ((u32)c < ccnt)
                    carry[c] = stage[e0 + f + c];
            offs[tid] += f;
        }
        __syncthreads();
        t0 += P;
    }
}

// ------------------------------------------------------------------------------------------------
// peel_scatter: the chunks of NODE_PEEL nodes (dominated by their first key `ref`).  Keys below ref fill the
// node's range from its start upwards, keys above it from its end downwards; copies of ref stay behind (their
// count is the terminal child).  Order inside a part is irrelevant, so a tile takes its two output ranges
// from the node's cursors with one atomic each; peel_fix_kernel then gives the two children their sizes.
constexpr int PEEL_THREADS = 1024;
constexpr int PEEL_ITEMS = 4;
__global__ __launch_bounds__(PEEL_THREADS) void peel_scatter_kernel(const Node *__restrict__ nodes,
                                                                    const Chunk *__restrict__ chunks, u32 n_chunks,
                                                                    u64 *__restrict__ buf0, u64 *__restrict__ buf1,
                                                                    u32 *__restrict__ stat)
{
    __shared__ u32 wt[PEEL_THREADS / 64][2];
    __shared__ u32 base[2], tsum[2];
    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    if (!(nd.meta & NODE_PEEL))
        return;
    const u64 *__restrict__ srcb = (nd.meta & NODE_BUF) ? buf1 : buf0;
    u64 *__restrict__ dst = (nd.meta & NODE_BUF) ? buf0 : buf1;
    u32 *st = stat + (size_t)ch.node * NODE_STAT_WORDS;
    const u64 ref = srcb[nd.start + st[7]];
    const u64 *__restrict__ src = srcb + nd.start + ch.off;
    const u32 lt_start = nd.start, gt_end = nd.start + nd.len;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (u32 t0 = 0; t0 < ch.len; t0 += PEEL_THREADS * PEEL_ITEMS) {
        u64 key[PEEL_ITEMS];
        u32 nl = 0, ng = 0;
#pragma unroll
        for (int j = 0; j < PEEL_ITEMS; j++) {
            const u32 i = t0 + tid + j * PEEL_THREADS;
            key[j] = i < ch.len ? NT_LOAD(&src[i]) : ref;     // (past the end: counts as a copy of ref, not moved)
            nl += key[j] < ref ? 1u : 0u;
            ng += key[j] > ref ? 1u : 0u;
        }
        const u32 il = wave_incl_scan(nl), ig = wave_incl_scan(ng);
        if (lane == 63) {
            wt[wave][0] = il;
            wt[wave][1] = ig;
        }
        __syncthreads();
        if (tid < 2) {                              // thread 0: below, thread 1: above
            u32 sum = 0;
            for (int w = 0; w < PEEL_THREADS / 64; w++) {
                const u32 t = wt[w][tid];
                wt[w][tid] = sum;
                sum += t;
            }
            base[tid] = sum ? atomicAdd(&st[3 + tid], sum) : 0u;
            tsum[tid] = sum;
        }
        __syncthreads();
        u32 pl = lt_start + base[0] + wt[wave][0] + il - nl;
        u32 pg = gt_end - base[1] - tsum[1] + wt[wave][1] + ig - ng;
#pragma unroll
        for (int j = 0; j < PEEL_ITEMS; j++) {
            if (key[j] < ref)
                dst[pl++] = key[j];
            else if (key[j] > ref)
                dst[pg++] = key[j];
        }
        __syncthreads();                            // wt / base are rewritten by the next tile
    }
}

__global__ __launch_bounds__(256) void peel_fix_kernel(const Node *__restrict__ nodes, u32 n_nodes,
                                                       Node *__restrict__ next, const u32 *__restrict__ stat)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes)
        return;
    const Node nd = nodes[i];
    if (!(nd.meta & NODE_PEEL))
        return;
    const u32 *st = stat + (size_t)i * NODE_STAT_WORDS;
    next[nd.child_base].len = st[3];
    next[nd.child_base + 2].start = nd.start + nd.len - st[4];
    next[nd.child_base + 2].len = st[4];
}

hipError_t launch_peel_scatter(const Node *nodes, u32 n_nodes, const Chunk *chunks, u32 n_chunks, Node *next, u64 *buf0,
                               u64 *buf1, u32 *stat, hipStream_t s)
{
    if (n_chunks == 0 || n_nodes == 0 || !stat)
        return hipSuccess;
    hipLaunchKernelGGL(peel_scatter_kernel, dim3(n_chunks), dim3(PEEL_THREADS), 0, s, nodes, chunks, n_chunks, buf0, buf1,
                       stat);
    hipLaunchKernelGGL(peel_fix_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, nodes, n_nodes, next, stat);
    return hipGetLastError();
}

hipError_t launch_level_scatter(const Node *nodes, const Chunk *chunks, u32 n_chunks, int src_dna,
                                const u64 *words, u64 n_words, u64 first, int k, u64 *buf0, u64 *buf1,
                                const u32 *hist, const u32 *tot, u32 flt_lo, u32 flt_span, u32 flt_tb, u32 max_bits,
                                hipStream_t s)
{
    if (n_chunks == 0)
        return hipSuccess;
    const DigitFilter flt{flt_lo, flt_span, flt_tb};
    const u64 mask = kmer_mask(k);
    // 16384-key tiles with aligned 64-/128-byte write combining, next tile prefetched: every write request a
    // full flush unit; one workgroup per CU.  (8192-key tiles without combining, two workgroups per CU, were
    // the first version: 16-18 ms per level at 3 Gbase against 10.6-11.2 now.)
    hipError_t ae = hipSuccess;
    if (!func_attrs_ready(FA_SCATTER)) {
        auto set = [&](const void *fn, size_t bytes) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess && ae == hipSuccess)
                ae = e;
        };
        set(reinterpret_cast<const void *>(level_scatter_wc_dna_kernel), wc_smem(16));
        set(reinterpret_cast<const void *>(level_scatter_wc_kernel<true, WC_ITEMS_KEYS, 1024, 8>), wc_smem(WC_ITEMS_KEYS));
        set(reinterpret_cast<const void *>(level_scatter_wc_kernel<false, WC_ITEMS_KEYS, 1024, 8>), wc_smem(WC_ITEMS_KEYS));
        set(reinterpret_cast<const void *>(level_scatter_wc_kernel<false, WC_ITEMS_KEYS, 512, 8>), wc_smem(WC_ITEMS_KEYS, 512));
        set(reinterpret_cast<const void *>(level_scatter_wc_kernel<false, WC_ITEMS_KEYS, 1024, 16>), wc_smem(WC_ITEMS_KEYS));
        if (ae != hipSuccess)
            return ae;
        func_attrs_mark(FA_SCATTER);
    }
    static int nth_keys = -1, wdbg = -1;
    if (nth_keys < 0) {
        const char *e = diag_env("DNAGPU_WC_NTH");          // experiment: 512-thread workgroups (needs splits <= 9 bits)
        nth_keys = e ? atoi(e) : 1024;
        e = diag_env("DNAGPU_DEBUG_SCATTER");               // timing ablations only; results invalid when set
        wdbg = e ? atoi(e) : 0;
    }
    if (src_dna && flt_span != ~0u)           // sharded count: sweep the whole sequence, keep this owner's keys
        hipLaunchKernelGGL(level_scatter_wc_dna_kernel, dim3(n_chunks), dim3(WC_THREADS), wc_smem(16), s, nodes,
                           chunks, n_chunks, words, n_words, first, mask, buf0, hist, tot, flt, wdbg);
    else if (src_dna)                         // unfiltered root: one position per staged key, next tile
        hipLaunchKernelGGL((level_scatter_wc_kernel<true, WC_ITEMS_KEYS, 1024, 8>), dim3(n_chunks), dim3(WC_THREADS),
                           wc_smem(WC_ITEMS_KEYS), s, nodes, chunks, n_chunks, words, n_words, first, mask, buf0,
                           buf1, hist, tot, wdbg);
    else if (nth_keys == 16 || (nth_keys == 1024 && max_bits <= 9))   // at most 512 digits: the carry of 128-byte flush units fits
                                              // (the dna root, write only, measured slower with them: 3.0 vs 2.65 ms at 1 Gbase)
        hipLaunchKernelGGL((level_scatter_wc_kernel<false, WC_ITEMS_KEYS, 1024, 16>), dim3(n_chunks), dim3(WC_THREADS),
                           wc_smem(WC_ITEMS_KEYS), s, nodes, chunks, n_chunks, words, n_words, first, mask, buf0,
                           buf1, hist, tot, wdbg);
    else if (nth_keys == 512)
        hipLaunchKernelGGL((level_scatter_wc_kernel<false, WC_ITEMS_KEYS, 512, 8>), dim3(n_chunks), dim3(512),
                           wc_smem(WC_ITEMS_KEYS, 512), s, nodes, chunks, n_chunks, words, n_words, first, mask, buf0,
                           buf1, hist, tot, wdbg);
    else
        hipLaunchKernelGGL((level_scatter_wc_kernel<false, WC_ITEMS_KEYS, 1024, 8>), dim3(n_chunks), dim3(WC_THREADS),
                           wc_smem(WC_ITEMS_KEYS), s, nodes, chunks, n_chunks, words, n_words, first, mask, buf0,
                           buf1, hist, tot, wdbg);
#ifdef DNAGPU_STAMPS
    if (!(src_dna && flt_span != ~0u))
        stamps_report(src_dna ? "scatter_wc<dna>" : "scatter_wc<keys>", src_dna ? 0 : 12, s);
#endif
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// leaves: sort + run-length encode each leaf in LDS and append its groups to the output.
//
// Persistent workgroups, static assignment (workgroup b takes leaves b, b + G, b + 2G, ...): while a
// leaf is being sorted the next leaf's descriptor is already in scalar registers and, from the
// middle of the leaf on, its keys are in flight into vector registers.
//
// Output placement: one returning atomic add per leaf on a 64-bit cursor hands the leaf a dense
// range of the output arrays.  Workgroups never wait for each other (an ordered chained scan was
// measured: on this chip its cross-XCD look-back cost 35-50 % of the kernel).  Leaves therefore land
// in completion order; inside a leaf groups ascend, and the segment directory (seg_off/seg_cnt, in
// leaf = key order) gives the globally ascending view that dnagpu_hist_download serves.
// The in-bin ranking of leaves_kernel.  Thread t takes the keys staged at positions t + j*NT (its
// neighbours in the wave hold the neighbouring positions, so the LDS reads below are conflict-free)
// and returns in key[j] / pos[j] the key and its sorted position: the bin's start + the members
// that precede it in (key, staged position) order.
//   * Straight-line code: all rows' keys, then all bin bounds, then four members of every bin are
//     read as three batches; a loop per key paid one LDS round trip per member and key, one after
//     the other (cycle stamps: 27 % of the kernel).  Bins of more than four members (0.6 % of the
//     keys of a random leaf) finish in a loop.
//   * A member slot past the bin's end reads the key's own slot, which never precedes itself: no
//     validity masks.  A row past the leaf's end takes the last key and is never written back.
//   * T = u32 (members of a bin differ only below bit 32): "precedes" is ONE 64-bit compare of
//     {low dword, staged position} pairs.
template <int NT, int ITEMS, typename T>
__device__ __forceinline__ void rank_in_bins(const u64 *A, const unsigned short *H, const u32 *uniform_bits,
                                             const unsigned short *P, u32 len,
                                             int sshift,
                                             u32 smask, u32 nbig, u64 (&key)[ITEMS], u32 (&pos)[ITEMS])
{
    const int tid = threadIdx.x;
    u32 lo[ITEMS], size[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const u32 i = tid + j * NT;
        key[j] = A[i < len ? i : len - 1];
    }
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const u32 b = (u32)(key[j] >> sshift) & smask;
        const u32 l = H[b], h = H[b + 1];
        lo[j] = l;
        // (slots past the end of the leaf rank nothing: they hold a copy of the last key, and if that is a
        // heavy one its whole bin would be walked for nothing -- 955 us for a leaf of 5000 equal keys)
        size[j] = (u32)(tid + j * NT) < len ? h - l : 0u;
    }
    if (nbig > 0) {                                 // rare: the members of a long bin that a wave has placed already
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const u32 b = (u32)(key[j] >> sshift) & smask;
            const u32 i = tid + j * NT;
            if (i < len && ((uniform_bits[b >> 5] >> (b & 31)) & 1)) {
                const u32 p = P[i];
                if (p != 0xffffu) {
                    lo[j] = p;
                    size[j] = 0;
                }
            }
        }
    }
    u32 more = 0;                                   // bit j: row j's bin has more than eight members
    u32 any5 = 0;                                   // some row of this thread has more than four
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        pos[j] = lo[j];
        any5 |= size[j] > 4 ? 1u : 0u;
        more |= size[j] > 8 ? (1u << j) : 0u;
    }
    const int tiers = __any(any5 != 0) ? 2 : 1;     // members 0-3 of every bin, then -- if any lane of the
#pragma unroll 1                                    // wave needs them -- members 4-7, again as one batch
    for (int g = 0; g < tiers; g++) {
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const u32 i = tid + j * NT < len ? tid + j * NT : len - 1;
            T o[4];
            u32 idx[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                idx[t] = (u32)(4 * g + t) < size[j] ? lo[j] + 4 * g + t : i;
                o[t] = *reinterpret_cast<const T *>(&A[idx[t]]);     // (little endian: low dword first)
            }
            u32 cnt = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (sizeof(T) == 4)
                    cnt += (((u64)o[t] << 32) | idx[t]) < (((u64)(u32)key[j] << 32) | i) ? 1u : 0u;
                else
                    cnt += ((u64)o[t] < key[j] || ((u64)o[t] == key[j] && idx[t] < i)) ? 1u : 0u;
            }
            pos[j] += cnt;
        }
    }
    if (more) {
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            if ((more >> j) & 1) {
                const u32 i = tid + j * NT < len ? tid + j * NT : len - 1;
                u32 cnt = 0;
#pragma unroll 1
                for (u32 m = lo[j] + 8; m < lo[j] + size[j]; m++) {
                    const u64 o = A[m];
                    cnt += (o < key[j]) || (o == key[j] && m < i);
                }
                pos[j] += cnt;
            }
        }
    }
}

// A leaf that sorts (not a single-key node) belongs to the class of the kernel whose capacity it needs.
// (classes: 0 = up to LEAF_CAP_TINY keys, 1 = up to LEAF_CAP_SMALL, 2 = up to LEAF_CAP)
__device__ __forceinline__ bool leaf_in_class(const Node &nd, int cls)
{
    const bool sorts = nd.len > 0 && (nd.meta & 0xff) != 0 && !(nd.meta & NODE_TERMINAL);
    const int c = nd.len > (u32)LEAF_CAP_SMALL ? 2 : (nd.len > (u32)LEAF_CAP_TINY ? 1 : 0);
    if (cls == 3)                                  // hashed leaves take both classes up to LEAF_CAP_SMALL
        return sorts && c <= 1;
    return sorts && c == cls;                      // (single-key and empty nodes: emit_singles_kernel)
}

// CAP = 4096 (four keys per thread at 1024 threads), 6144 (six) or 1024 (four at 256 threads, SB = 11): the
// host launches one kernel per class present.  Bigger leaves let the level above split on half as many
// digits (15 % cheaper per key), smaller ones cost less per key here.
// MIXED: the node list also holds other classes' leaves or nodes that do not sort; the launch then walks the
// index list of its class (launches over a single-class list use the MIXED = false instantiation).
template <int NT, int MINW, int CAP, bool MIXED, int SB = 13>
__global__ __launch_bounds__(NT, MINW) void leaves_kernel(const Node *__restrict__ leaves, u32 n_leaves,
                                                          const u64 *__restrict__ buf0,
                                                          const u64 *__restrict__ buf1,
                                                          unsigned long long *__restrict__ cursor,
                                                          u64 *__restrict__ seg_off, u32 *__restrict__ seg_cnt,
                                                          u64 *__restrict__ out_keys,
                                                          u32 *__restrict__ out_counts, int dbg,
                                                          const u32 *__restrict__ list, u32 n_list)
{
    // MIXED: the node list also holds leaves of the other class or nodes that do not sort; `list` then
    // holds the indices of this launch's leaves (class_list_kernel) and workgroup b takes entries b, b + grid, ...
    constexpr int ITEMS = CAP / NT;            // keys per thread
    constexpr int BINS = 1 << SB;              // counting-sort bins (8192: mean occupancy 0.35-0.7), 16-bit counters:
    constexpr int WPT = BINS / 2 / NT;         // two bins per LDS word, WPT consecutive words per thread in the scan
    constexpr int WAVES = NT / 64;
    constexpr int ROWS = ITEMS * WAVES;        // 64-position rows of the staged leaf
    constexpr u32 BIG_BIN = 24;
    static_assert(ROWS * 64 == CAP && ROWS <= 128, "one wave scans the rows' head counts, two per lane");
    static_assert((1 << SB) == BINS && WPT % 4 == 0 && CAP < 65536, "bins");

    __shared__ __attribute__((aligned(16))) u64 A[CAP];
    // bins (two 16-bit counters per word: a leaf holds fewer than 65536 keys, so a returning 32-bit add
    // of 1 or 1 << 16 never carries between the halves) -> offsets.  Zeroed for the NEXT leaf as soon as
    // this leaf's ranking is done (the head positions have their own table), so a leaf starts counting
    // right behind its first barrier.
    __shared__ __attribute__((aligned(16))) u32 H[BINS / 2 + 8];
    __shared__ unsigned short P[CAP + 8];      // position of the q-th run head
    const unsigned short *H16 = reinterpret_cast<const unsigned short *>(H);
    __shared__ u32 rowcnt[ROWS];               // [row][wave] head counts -> exclusive offsets
    __shared__ u32 wtmp[WAVES];
    __shared__ u32 sh_D;
    __shared__ u64 sh_obase;
    __shared__ u32 big_n;                      // nonzero: some bin of this leaf has more than BIG_BIN members
    __shared__ u32 big_cnt;
    __shared__ u32 big_list[64];
    __shared__ u32 uniform_bits[BINS / 32];   // bit b: bin b is long and a wave has placed its heavy members (P)

    int tid = threadIdx.x;
    u32 lq = blockIdx.x;                       // MIXED: position in `list`
    u32 li = blockIdx.x;
    if (MIXED)
        li = lq < n_list ? list[lq] : n_leaves;
    if (li >= n_leaves)
        return;
    u32 ln_list = n_leaves;                    // MIXED: the leaf after this one, fetched one iteration ahead
    if (MIXED) {
        lq += gridDim.x;
        ln_list = lq < n_list ? list[lq] : n_leaves;
    }
    Node nd = leaves[li];
    u64 key[ITEMS];
    if (nd.len > 0 && (nd.meta & 0xff) != 0 && !(nd.meta & NODE_TERMINAL)) {
        const u64 *__restrict__ src = ((nd.meta & NODE_BUF) ? buf1 : buf0) + nd.start;
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            u32 i = tid + j * NT;
            key[j] = NT_LOAD(&src[i < nd.len ? i : nd.len - 1]);
        }
    }

#pragma unroll
    for (int q = 0; q < WPT / 4; q++)              // (every later leaf zeroes the bins for its successor)
        reinterpret_cast<uint4 *>(H)[tid * (WPT / 4) + q] = make_uint4(0, 0, 0, 0);
    if (tid == 0)
        big_n = 0;
    STAMP_DECL
    for (;;) {
        u32 ln = li + gridDim.x;
        if (MIXED) {
            ln = ln_list;
            lq += gridDim.x;
            ln_list = lq < n_list ? list[lq] : n_leaves;
        }
        const bool has_next = ln < n_leaves;
        const Node nn = leaves[has_next ? ln : li];           // wave-uniform: a scalar load, used later
        asm volatile("" : "+v"(tid));              // (nothing derived from the thread index is held across leaves)
        const int lane = tid & 63, wave = tid >> 6;
        __syncthreads();                           // A/H of the previous leaf are dead
        // Every path through an iteration "uses" the prefetched keys here.  Without this a leaf that
        // never reads them (a single-key leaf) leaves loads in flight, and the compiler guards the
        // next prefetch into the same registers with s_waitcnt vmcnt(0) -- on every path, which
        // also waits for the placement atomic issued just before it.
#pragma unroll
        for (int j = 0; j < ITEMS; j++)
            asm volatile("" : "+v"(key[j]));
        STAMP(0);  // waited for the previous leaf / descriptor
        const u32 len = nd.len;
        const int rem = (int)(nd.meta & 0xff);
        u32 D = 0;                                 // groups in this leaf
        const bool single = (len > 0) && (rem == 0 || (nd.meta & NODE_TERMINAL));
        const bool sorted_path = len > 0 && !single;
        const int sb = rem < SB ? rem : SB;
        const int sshift = rem - sb;
        const u32 smask = (1u << sb) - 1;
        u32 headbits = 0;                          // bit j: this thread's slot j starts a run

        if (sorted_path) {
            u32 rank[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                u32 i = tid + j * NT;
                rank[j] = 0;
                // (slots past the end of the leaf take no part: an atomic on one shared dummy bin
                // would serialise a third of the workgroup on a single LDS address)
                if (i < len) {
                    const u32 b = (u32)(key[j] >> sshift) & smask;
                    const u32 sh16 = (b & 1u) * 16u;
                    rank[j] = (atomicAdd(&H[b >> 1], 1u << sh16) >> sh16) & 0xffffu;
                }
            }
            __syncthreads();
            STAMP(2);  // count (incl. waiting for the keys)
            {   // exclusive scan of the bins, 2 * WPT consecutive bins per thread (16-byte LDS accesses)
                uint4 v[WPT / 4];
                u32 sum = 0, cmax = 0;
#pragma unroll
                for (int q = 0; q < WPT / 4; q++) {
                    v[q] = reinterpret_cast<uint4 *>(H)[tid * (WPT / 4) + q];
                    const u32 w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        sum += (w[e] & 0xffffu) + (w[e] >> 16);
                        cmax = max(cmax, max(w[e] & 0xffffu, w[e] >> 16));
                    }
                }
                if (cmax > BIG_BIN)                 // rare: some bin holds many copies of few keys
                    big_n = 1;                      // (benign race: every writer stores 1)
                const u32 inc = wave_incl_scan(sum);
                if (lane == 63)
                    wtmp[wave] = inc;
                __syncthreads();
                u32 base = inc - sum;
                {   // every wave scans the wave totals itself: one LDS read + one DPP scan
                    const u32 ws = wave_incl_scan(lane < WAVES ? wtmp[lane] : 0u);
                    const int wv = __builtin_amdgcn_readfirstlane(wave);
                    if (wv)
                        base += (u32)__builtin_amdgcn_readlane((int)ws, wv - 1);
                }
#pragma unroll
                for (int q = 0; q < WPT / 4; q++) {
                    u32 w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const u32 c0 = w[e] & 0xffffu, c1 = w[e] >> 16;
                        w[e] = base | ((base + c0) << 16);
                        base += c0 + c1;
                    }
                    reinterpret_cast<uint4 *>(H)[tid * (WPT / 4) + q] = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if (tid == 0)
                    H[BINS / 2] = len;              // H16[BINS]: the end of the last bin
            }
            __syncthreads();
            STAMP(3);  // scan
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                u32 i = tid + j * NT;
                if (i < len)
                    A[H16[(u32)(key[j] >> sshift) & smask] + rank[j]] = key[j];
            }
            __syncthreads();
            STAMP(4);  // place
            const bool any_big = big_n != 0;           // (written before the scan's barriers)
            u32 nbig = 0;
            if (rem > sb && any_big) {
                // A bin of hundreds of copies of ONE key (tandem repeats, satellites) would cost
                // members^2 in the all-pairs ranking below; one wave checks "all members equal"
                // in members/64 steps and such bins then keep their staged order.
                for (u32 w = tid; w < BINS / 32; w += NT)
                    uniform_bits[w] = 0;
                if (tid == 0)
                    big_cnt = 0;
                __syncthreads();
                for (u32 b = tid; b < (u32)BINS; b += NT)
                    if ((u32)H16[b + 1] - (u32)H16[b] > BIG_BIN) {
                        u32 slot = atomicAdd(&big_cnt, 1u);
                        if (slot < 64)
                            big_list[slot] = b;
                    }
                __syncthreads();
                nbig = big_cnt < 64 ? big_cnt : 64;
                // One wave per long bin.  Up to MAX_PIVOTS times: the first member without a position is the
                // pivot; count the members below it, then give every member EQUAL to it its final position
                // (staged order among equals) -- two sweeps of members/64 steps per distinct heavy key.  Members
                // still without a position after that (P = NO_POS: the bin holds many distinct keys) are ranked
                // all-pairs by their own threads, as in a short bin.
                constexpr u32 NO_POS = 0xffffu;
                constexpr int MAX_PIVOTS = 32;
                const u64 lane_lt = (1ull << lane) - 1ull;
                for (u32 e = wave; e < nbig; e += WAVES) {
                    const u32 b = big_list[e];
                    const u32 s0 = H16[b], s1 = H16[b + 1];
                    for (u32 m = s0 + lane; m < s1; m += 64)
                        P[m] = (unsigned short)NO_POS;
                    __builtin_amdgcn_wave_barrier();
                    u32 cursor = s0, placed = 0;
                    for (int it = 0; it < MAX_PIVOTS && placed < s1 - s0; it++) {
                        u32 pi = s1;
                        for (u32 m0 = cursor; m0 < s1; m0 += 64) {
                            const u32 m = m0 + lane;
                            const u64 un = __ballot(m < s1 && P[m] == NO_POS);
                            if (un) {
                                pi = m0 + (u32)__builtin_ctzll(un);
                                break;
                            }
                        }
                        if (pi >= s1)
                            break;
                        cursor = pi + 1;
                        const u64 k0 = A[pi];
                        // (four independent LDS loads per step: the sweeps are latency bound)
                        u32 below = 0;
                        for (u32 m0 = s0; m0 < s1; m0 += 256) {
                            u64 v[4];
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const u32 m = m0 + q * 64 + lane;
                                v[q] = A[m < s1 ? m : s0];
                            }
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                below += (u32)__popcll(__ballot(m0 + q * 64 + lane < s1 && v[q] < k0));
                        }
                        u32 run = s0 + below;
                        for (u32 m0 = s0; m0 < s1; m0 += 256) {
                            u64 v[4];
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const u32 m = m0 + q * 64 + lane;
                                v[q] = A[m < s1 ? m : s0];
                            }
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const u32 m = m0 + q * 64 + lane;
                                const bool eq = m < s1 && v[q] == k0;
                                const u64 em = __ballot(eq);
                                if (eq)
                                    P[m] = (unsigned short)(run + (u32)__popcll(em & lane_lt));
                                run += (u32)__popcll(em);
                            }
                        }
                        placed += run - (s0 + below);
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (lane == 0)
                        atomicOr(&uniform_bits[b >> 5], 1u << (b & 31));
                }
                __syncthreads();
            }
            STAMP(11); // long bins (rare)
            if (rem > sb && !(dbg & 1)) {
                // exact position inside each (small) bin: thread i ranks the key staged at i
                if (sshift <= 32)
                    rank_in_bins<NT, ITEMS, u32>(A, H16, uniform_bits, P, len, sshift, smask, nbig, key, rank);
                else
                    rank_in_bins<NT, ITEMS, u64>(A, H16, uniform_bits, P, len, sshift, smask, nbig, key, rank);
                __syncthreads();
                STAMP(5);  // in-bin rank
#pragma unroll
                for (int j = 0; j < ITEMS; j++)
                    if (tid + j * NT < len)
                        A[rank[j]] = key[j];
            }
            // the bins are dead (every read of them is behind a barrier): zero them for the next leaf
#pragma unroll
            for (int q = 0; q < WPT / 4; q++)
                reinterpret_cast<uint4 *>(H)[tid * (WPT / 4) + q] = make_uint4(0, 0, 0, 0);
            if (tid == 0)
                big_n = 0;
            __syncthreads();
            STAMP(6);  // write back sorted
        }

        // ---- the sort no longer needs key[]: the next leaf's loads are issued below, behind wave
        // 0's placement atomic (a wave's vector-memory operations retire in order: an atomic issued
        // behind the HBM loads would wait for all of them), from ONE code site for every wave (with a
        // second site for wave 0 the compiler's wait-count pass guards the key registers of the
        // other site with a vmcnt(0), i.e. waits for the atomic on the spot).  The loads are
        // unconditional (no next leaf: one harmless word of the leaf list itself): one straight-line
        // site, no branch for the wait-count pass to merge over.
        const bool next_loads = has_next && nn.len > 0 && (nn.meta & 0xff) != 0 && !(nn.meta & NODE_TERMINAL);
        const u64 *__restrict__ nsrc = next_loads ? ((nn.meta & NODE_BUF) ? buf1 : buf0) + nn.start
                                                  : reinterpret_cast<const u64 *>(leaves);
        const u32 nlim = next_loads ? nn.len : 1u;

        u64 ob_reg = 0;                            // wave 0, lane 63: the leaf's output base
        if (sorted_path) {
            // run heads in sorted order; row j = positions [j*NT, (j+1)*NT).  All the reads first.
            u64 cur[ITEMS], prv[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                const u32 i = tid + j * NT;
                cur[j] = A[i];
                prv[j] = A[i ? i - 1 : 0];
            }
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                const u32 i = tid + j * NT;
                const bool head = (i < len) & ((i == 0) | (cur[j] != prv[j]));
                headbits |= head ? (1u << j) : 0u;
                const u64 hbj = __ballot(head);
                if (lane == 0)
                    rowcnt[j * WAVES + wave] = (u32)__popcll(hbj);
            }
            __syncthreads();
            STAMP(7);  // next loads issued + heads + ballots
        }
        if (wave == 0) {
            if (sorted_path) {
                const u32 v0 = lane < ROWS ? rowcnt[lane] : 0u;
                const u32 v1 = lane + 64 < ROWS ? rowcnt[lane + 64] : 0u;
                const u32 inc0 = wave_incl_scan(v0);
                const u32 tot0 = (u32)__builtin_amdgcn_readlane((int)inc0, 63);
                const u32 inc1 = wave_incl_scan(v1) + tot0;
                if (lane < ROWS)
                    rowcnt[lane] = inc0 - v0;
                if (lane + 64 < ROWS)
                    rowcnt[lane + 64] = inc1 - v1;
                D = (u32)__builtin_amdgcn_readlane((int)inc1, 63);
            } else if (single) {
                D = 1;
            }
            // ---- placement: one atomic per leaf, issued as early as D is known; its result is
            // only needed at the last barrier before the output.  The add is written in assembly so
            // that nothing waits for its result here: left to the compiler, a wave-uniform atomic
            // goes through its atomic optimizer (one lane adds, s_waitcnt vmcnt(0), readfirstlane),
            // and a plain one still got a vmcnt(0) from register reuse.  The result is awaited just
            // before the last barrier, one phase later.
            if (lane == 63) {
                sh_D = D;
                if (D > 0) {
                    if (dbg & 2) {
                        ob_reg = (u64)nd.start;
                    } else {
                        const unsigned long long dd = D;
                        asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0"
                                     : "=&v"(ob_reg)
                                     : "v"(cursor), "v"(dd)
                                     : "memory");
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);         // (keep the atomic ahead of wave 0's loads)
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            u32 i = tid + j * NT;
            key[j] = NT_LOAD(&nsrc[i < nlim ? i : nlim - 1]);
        }
        if (sorted_path) {
            __syncthreads();
            STAMP(8);  // row scan
            D = sh_D;
            const u64 below = ((u64)1 << lane) - 1;
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                const bool head = (headbits >> j) & 1;
                const u64 hbj = __ballot(head);    // (recomputed: eight live ballots cost 16 SGPRs)
                if (head)
                    P[rowcnt[j * WAVES + wave] + (u32)__popcll(hbj & below)] = (unsigned short)(tid + j * NT);
            }
            if (tid == 0)
                P[D] = (unsigned short)len;
            // (H as bin offsets is dead: every read of it happened before the barriers above)
        }
        if (wave == 0 && lane == 63) {
            // (vmcnt(0), not a count of the loads issued behind the atomic: a register-spill reload
            // the compiler may place in between would make a counted wait return too early)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ob_reg) : : "memory");
            sh_obase = ob_reg;
            seg_off[li] = ob_reg;
            seg_cnt[li] = D;
        }
        __syncthreads();
        STAMP(9);  // atomic + head positions
        D = sh_D;
        const u64 obase = sh_obase;
        if (!(dbg & 4)) {
            if (single) {
                if (tid == 0) {
                    out_keys[obase] = nd.prefix;
                    out_counts[obase] = len;
                }
            } else {
                for (u32 q2 = tid; q2 < D; q2 += NT) {
                    const u32 p = P[q2];
                    NT_STORE(A[p], &out_keys[obase + q2]);
                    NT_STORE((u32)P[q2 + 1] - p, &out_counts[obase + q2]);
                }
            }
        }
        STAMP(10);  // output
        if (!has_next)
            break;
        li = ln;
        nd = nn;
    }
    STAMP_FLUSH(0);
}

static u32 leaves_grid(u32 n_leaves, int per_cu)
{
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
        if (n_cu <= 0)
            n_cu = 256;
    }
    u32 g = (u32)n_cu * (u32)per_cu;
    return n_leaves < g ? n_leaves : g;
}

// Nodes that hold one distinct key (their bits are exhausted: heavy hitters, short k-mers after a
// terminal split) or none need no sorting: they are emitted in bulk, in node (= key) order, into the
// first slots of the output; the sorting leaves then take their ranges from the cursor behind them.
// (Sent through the leaves kernel one workgroup iteration each, a million of them cost 10 ms.)
__global__ __launch_bounds__(256) void single_flags_kernel(const Node *__restrict__ leaves, u32 n_leaves,
                                                           u32 *__restrict__ flags)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves)
        return;
    const Node nd = leaves[i];
    const bool sorts = nd.len > 0 && (nd.meta & 0xff) != 0 && !(nd.meta & NODE_TERMINAL);
    flags[i] = (!sorts && nd.len > 0) ? 1u : 0u;
}

// The leaves of one class among a mixed node list: flags -> exclusive scan -> compact index list.
__global__ __launch_bounds__(256) void class_flags_kernel(const Node *__restrict__ leaves, u32 n_leaves, int big_class,
                                                          u32 *__restrict__ flags)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_leaves)
        flags[i] = leaf_in_class(leaves[i], big_class) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void class_list_kernel(const Node *__restrict__ leaves, u32 n_leaves, int big_class,
                                                         const u32 *__restrict__ pre, u32 *__restrict__ list)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_leaves && leaf_in_class(leaves[i], big_class))
        list[pre[i]] = i;
}

__global__ __launch_bounds__(256) void emit_singles_kernel(const Node *__restrict__ leaves, u32 n_leaves,
                                                           const u32 *__restrict__ pre, const u32 *__restrict__ total,
                                                           unsigned long long *__restrict__ cursor, u64 out_base,
                                                           u64 *__restrict__ seg_off, u32 *__restrict__ seg_cnt,
                                                           u64 *__restrict__ out_keys, u32 *__restrict__ out_counts)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0)
        *cursor = out_base + *total;               // the sorting leaves allocate behind the singles
    if (i >= n_leaves)
        return;
    const Node nd = leaves[i];
    const bool sorts = nd.len > 0 && (nd.meta & 0xff) != 0 && !(nd.meta & NODE_TERMINAL);
    if (sorts)
        return;
    const u64 o = out_base + pre[i];
    seg_off[i] = o;
    seg_cnt[i] = nd.len > 0 ? 1u : 0u;
    if (nd.len > 0) {
        out_keys[o] = nd.prefix;
        out_counts[o] = nd.len;
    }
}

// ------------------------------------------------------------------------------------------------
// hash_leaves: the leaves of a histogram whose groups need no order (dnagpu_count_kmers_unordered).  A leaf of at
// most 4096 keys is counted in an LDS hash table of 8192 slots (linear probing, 64-bit compare-and-swap; counts
// in 16-bit halves beside it) instead of being sorted: four barriers per leaf instead of ten, one LDS round trip
// per key and probe instead of the sort's chain.  Groups leave in slot order: every wave compacts its eighth of
// the table with ballots, so a store instruction writes one contiguous run.  The table is cleaned slot by slot
// as it is emitted.  Same persistent structure as leaves_kernel: static leaf assignment through the class list,
// the next leaf's keys requested while this one is emitted, output ranges from the cursor.
constexpr int HL_NT = 512;
constexpr int HL_ITEMS = LEAF_CAP_SMALL / HL_NT;      // 8 keys per thread
constexpr int HL_SLOTS = 127 * 64;                   // 8128: with the 16-bit counts 79.4 KiB, two workgroups per CU
constexpr u64 HL_EMPTY = ~(u64)0;                     // (the 32-base k-mer GG..G is counted beside the table)

__global__ __launch_bounds__(HL_NT, 4) void hash_leaves_kernel(const Node *__restrict__ leaves, u32 n_leaves,
                                                               const u64 *__restrict__ buf0, const u64 *__restrict__ buf1,
                                                               unsigned long long *__restrict__ cursor,
                                                               u64 *__restrict__ seg_off, u32 *__restrict__ seg_cnt,
                                                               u64 *__restrict__ out_keys, u32 *__restrict__ out_counts,
                                                               const u32 *__restrict__ list, u32 n_list)
{
    constexpr int WAVES = HL_NT / 64;
    constexpr int CHUNKS = HL_SLOTS / 64;          // wave w emits the 64-slot chunks w, w + WAVES, ...
    __shared__ __attribute__((aligned(16))) u64 tab[HL_SLOTS];
    __shared__ __attribute__((aligned(16))) u32 cnt2[HL_SLOTS / 2];
    unsigned short *cnt16 = reinterpret_cast<unsigned short *>(cnt2);
    __shared__ u32 wtot[WAVES];
    __shared__ u64 sh_obase;
    __shared__ u32 ones_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 lq = blockIdx.x;
    if (lq >= n_list)
        return;
    for (int q = tid; q < HL_SLOTS; q += HL_NT)
        tab[q] = HL_EMPTY;
    for (int q = tid; q < HL_SLOTS / 2; q += HL_NT)
        cnt2[q] = 0;
    if (tid == 0)
        ones_cnt = 0;
    u32 li = list[lq];
    Node nd = leaves[li];
    u64 key[HL_ITEMS];
    {
        const u64 *__restrict__ src = ((nd.meta & NODE_BUF) ? buf1 : buf0) + nd.start;
#pragma unroll
        for (int j = 0; j < HL_ITEMS; j++) {
            const u32 i = tid + j * HL_NT;
            key[j] = NT_LOAD(&src[i < nd.len ? i : nd.len - 1]);
        }
    }
    for (;;) {
        const u32 lq_next = lq + gridDim.x;
        const bool has_next = lq_next < n_list;
        const u32 ln = has_next ? list[lq_next] : li;
        const Node nn = leaves[ln];
        __syncthreads();                           // the table is clean, the previous leaf's emit is done
        const u32 len = nd.len;
#pragma unroll
        for (int j = 0; j < HL_ITEMS; j++) {
            const u32 i = tid + j * HL_NT;
            if (i < len) {
                const u64 kv = key[j];
                if (kv == HL_EMPTY) {
                    atomicAdd(&ones_cnt, 1u);
                } else {
                    u32 slot = (((((u32)kv ^ (u32)(kv >> 32)) * 0x9E3779B1u) >> 16) * (u32)HL_SLOTS) >> 16;
                    for (;;) {
                        const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&tab[slot]),
                                                  (unsigned long long)HL_EMPTY, (unsigned long long)kv);
                        if (old == HL_EMPTY || old == kv)
                            break;
                        slot = slot + 1 == (u32)HL_SLOTS ? 0u : slot + 1;
                    }
                    atomicAdd(&cnt2[slot >> 1], 1u << ((slot & 1u) * 16u));
                }
            }
        }
        // the next leaf's keys: in flight while this one is emitted
        if (has_next) {
            const u64 *__restrict__ src = ((nn.meta & NODE_BUF) ? buf1 : buf0) + nn.start;
#pragma unroll
            for (int j = 0; j < HL_ITEMS; j++) {
                const u32 i = tid + j * HL_NT;
                key[j] = NT_LOAD(&src[i < nn.len ? i : nn.len - 1]);
            }
        }
        __syncthreads();
        // occupied slots of this wave's part of the table
        u32 mine = 0;
        for (int ch = wave; ch < CHUNKS; ch += WAVES)
            mine += (u32)__popcll(__ballot(cnt16[ch * 64 + lane] != 0));
        if (lane == 0)
            wtot[wave] = mine;
        __syncthreads();
        u32 before = 0, D = 0;
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
            const u32 t = wtot[w];
            before += w < wave ? t : 0u;
            D += t;
        }
        const u32 ones = ones_cnt;
        if (tid == 0) {
            const u32 groups = D + (ones ? 1u : 0u);
            const u64 ob = atomicAdd(cursor, (unsigned long long)groups);
            sh_obase = ob;
            seg_off[li] = ob;
            seg_cnt[li] = groups;
        }
        __syncthreads();
        const u64 obase = sh_obase;
        u32 run = before;
        for (int ch = wave; ch < CHUNKS; ch += WAVES) {
            const u32 slot = (u32)(ch * 64 + lane);
            const u32 c = cnt16[slot];
            const u64 b = __ballot(c != 0);
            if (c) {
                const u32 r = run + (u32)__popcll(b & (((u64)1 << lane) - 1));
                NT_STORE(tab[slot], &out_keys[obase + r]);
                NT_STORE(c, &out_counts[obase + r]);
                tab[slot] = HL_EMPTY;
                cnt16[slot] = 0;
            }
            run += (u32)__popcll(b);
        }
        if (ones && tid == 0) {
            out_keys[obase + D] = HL_EMPTY;
            out_counts[obase + D] = ones;
            ones_cnt = 0;
        }
        if (!has_next)
            break;
        lq = lq_next;
        li = ln;
        nd = nn;
    }
}

hipError_t launch_leaves(const Node *leaves, u32 n_leaves, u32 n_tiny, u32 n_small, u32 n_big, const u64 *buf0,
                         const u64 *buf1,
                         u64 *cursor, u64 *seg_off, u32 *seg_cnt, u64 *out_keys, u32 *out_counts, u32 *flags,
                         u32 *scan_tmp, u32 *list, hipStream_t s, bool hashed, u64 out_base)
{
    if (n_leaves == 0)
        return hipSuccess;
    static int dbg = -1, variant = 0, mult = 1;
    if (dbg < 0) {
        const char *e = diag_env("DNAGPU_DEBUG_LEAVES");   // timing ablations only; results are invalid when set
        dbg = e ? atoi(e) : 0;
        const char *v = diag_env("DNAGPU_LEAVES_VARIANT");  // experiment switches: diagnostic build only
        variant = v ? atoi(v) : 0;                       // 1024 threads x 4 keys, 64 VGPRs: 2 workgroups = 32 waves per CU
        const char *m = diag_env("DNAGPU_LEAVES_GRIDMULT");
        mult = m ? atoi(m) : 1;
        if (mult < 1)
            mult = 1;
    }
    unsigned long long *cur = reinterpret_cast<unsigned long long *>(cursor);
    if (n_tiny + n_small + n_big < n_leaves) {           // single-key and empty nodes
        const u32 g = (n_leaves + 255) / 256;
        hipLaunchKernelGGL(single_flags_kernel, dim3(g), dim3(256), 0, s, leaves, n_leaves, flags);
        hipError_t e = launch_scan_u32(flags, flags, n_leaves, scan_tmp, flags + n_leaves, s);   // (flags holds n_leaves + 1)
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(emit_singles_kernel, dim3(g), dim3(256), 0, s, leaves, n_leaves, flags, flags + n_leaves, cur,
                           out_base, seg_off, seg_cnt, out_keys, out_counts);
    }
#ifdef DNAGPU_STAMPS
#define LEAVES_STAMPS(name) stamps_report("leaves " name, 0, s)
#else
#define LEAVES_STAMPS(name)
#endif
    const bool mixed_tiny = n_tiny < n_leaves, mixed_small = n_small < n_leaves, mixed_big = n_big < n_leaves;
    // (flags / scan_tmp / list are reused by the second class: same stream, the first launch is done with them)
#define LAUNCH_LEAVES(NT_, MINW_, CAP_, PER_CU_, MIXED_, SB_)                                                                   \
    do {                                                                                                              \
        if (MIXED_) {                                                                                                 \
            const int cls = (CAP_) > LEAF_CAP_SMALL ? 2 : ((CAP_) > LEAF_CAP_TINY ? 1 : 0);                           \
            const u32 n_cls = cls == 2 ? n_big : (cls == 1 ? n_small : n_tiny);                                       \
            const u32 g = (n_leaves + 255) / 256;                                                                     \
            hipLaunchKernelGGL(class_flags_kernel, dim3(g), dim3(256), 0, s, leaves, n_leaves, cls, flags);           \
            hipError_t e2 = launch_scan_u32(flags, flags, n_leaves, scan_tmp, flags + n_leaves, s);                   \
            if (e2 != hipSuccess)                                                                                     \
                return e2;                                                                                            \
            hipLaunchKernelGGL(class_list_kernel, dim3(g), dim3(256), 0, s, leaves, n_leaves, cls, flags, list);      \
            hipLaunchKernelGGL((leaves_kernel<NT_, MINW_, CAP_, true, SB_>), dim3(leaves_grid(n_cls, PER_CU_ * mult)), \
                               dim3(NT_), 0, s, leaves, n_leaves, buf0, buf1, cur, seg_off, seg_cnt, out_keys,        \
                               out_counts, dbg, list, n_cls);                                                         \
        } else                                                                                                        \
            hipLaunchKernelGGL((leaves_kernel<NT_, MINW_, CAP_, false, SB_>), dim3(leaves_grid(n_leaves, PER_CU_ * mult)), \
                               dim3(NT_), 0, s, leaves, n_leaves, buf0, buf1, cur, seg_off, seg_cnt, out_keys,        \
                               out_counts, dbg, (const u32 *)nullptr, 0u);                                            \
        LEAVES_STAMPS(#CAP_);                                                                                         \
    } while (0)
    if (hashed && n_tiny + n_small > 0) {
        // unordered histogram: every leaf of at most LEAF_CAP_SMALL keys is counted in an LDS hash table
        const u32 g = (n_leaves + 255) / 256;
        hipLaunchKernelGGL(class_flags_kernel, dim3(g), dim3(256), 0, s, leaves, n_leaves, 3, flags);
        hipError_t e3 = launch_scan_u32(flags, flags, n_leaves, scan_tmp, flags + n_leaves, s);
        if (e3 != hipSuccess)
            return e3;
        hipLaunchKernelGGL(class_list_kernel, dim3(g), dim3(256), 0, s, leaves, n_leaves, 3, flags, list);
        hipLaunchKernelGGL(hash_leaves_kernel, dim3(leaves_grid(n_tiny + n_small, 2 * mult)), dim3(HL_NT), 0, s, leaves,
                           n_leaves, buf0, buf1, cur, seg_off, seg_cnt, out_keys, out_counts, list, n_tiny + n_small);
        LEAVES_STAMPS("hashed");
        n_tiny = 0;
        n_small = 0;
    }
    if (n_tiny > 0)                                      // leaves of at most LEAF_CAP_TINY keys: 256-thread workgroups,
        LAUNCH_LEAVES(256, 8, LEAF_CAP_TINY, 8, mixed_tiny, 11);   // eight per CU, 2048 bins (what skew leaves behind)
    if (n_small > 0) {                                   // leaves of at most LEAF_CAP_SMALL keys
        if (variant == 2)
            LAUNCH_LEAVES(1024, 4, LEAF_CAP_SMALL, 1, mixed_small, 13);
        else if (variant == 3)
            LAUNCH_LEAVES(512, 4, LEAF_CAP_SMALL, 2, mixed_small, 13);
        else
            LAUNCH_LEAVES(1024, 8, LEAF_CAP_SMALL, 2, mixed_small, 13);
    }
    if (n_big > 0)                                       // leaves of up to LEAF_CAP keys
        LAUNCH_LEAVES(1024, 8, LEAF_CAP, 2, mixed_big, 13);
#undef LAUNCH_LEAVES
#undef LEAVES_STAMPS
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Dense count for short k-mers (2k <= 18 bits: k <= 9, the range test.sql's GROUP BY examples use).  With
// at most 262,144 possible keys the histogram IS a table of counters: persistent workgroups count their
// share of the windows into 32,768 LDS counters (16 windows per 64-bit funnel, the keys are never
// written anywhere), add the non-zero counters to a global table, and a second small kernel turns the
// table into the ascending (key, count) arrays.  Keys of 16-18 bits take 2-8 passes over the packed
// sequence (one per 32,768-key slice of the table), which still costs far less than moving 8-byte
// keys through the tree (1 Gbase: k = 5: 5.3 -> 0.35 ms; k = 8: 5.5 -> 0.85 ms).
constexpr int DENSE_LDS_BINS = 32768;
constexpr int DENSE_MAX_BITS = 18;

__global__ __launch_bounds__(1024) void dense_count_kernel(const u64 *__restrict__ words, u64 n_words, u64 first,
                                                           u64 count, int bits, u32 pass, u32 *__restrict__ table)
{
    extern __shared__ u32 T[];
    const u32 n_bins = bits >= 15 ? (u32)DENSE_LDS_BINS : (1u << bits);
    const u32 kmask = (bits >= 32) ? ~0u : ((1u << bits) - 1u);
    for (u32 b = threadIdx.x; b < n_bins; b += 1024)
        T[b] = 0;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * 1024 * 16;
    for (u64 i0 = ((u64)blockIdx.x * 1024 + threadIdx.x) * 16; i0 < count; i0 += stride) {
        const u64 dv = dig16_load(words, n_words, first + i0, 0);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const u32 key = (u32)(dv >> (2 * j)) & kmask;
            if (i0 + j < count && (key >> 15) == pass)
                atomicAdd(&T[key & (u32)(DENSE_LDS_BINS - 1)], 1u);
        }
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < n_bins; b += 1024) {
        const u32 c = T[b];
        if (c)
            atomicAdd(&table[pass * (u32)DENSE_LDS_BINS + b], c);
    }
}

// table[0 .. 2^bits) -> ascending (key, count) arrays; *n_out = number of non-zero counters.  One workgroup.
__global__ __launch_bounds__(1024) void dense_compact_kernel(const u32 *__restrict__ table, int bits,
                                                             u64 *__restrict__ out_keys, u32 *__restrict__ out_counts,
                                                             u64 *__restrict__ n_out)
{
    __shared__ u32 part[1024];
    __shared__ u32 wtmp[16];
    const u32 n_bins = 1u << bits;
    const u32 per = (n_bins + 1023) / 1024;
    const u32 b0 = threadIdx.x * per, b1 = b0 + per < n_bins ? b0 + per : n_bins;
    u32 nz = 0;
    for (u32 b = b0; b < b1; b++)
        nz += table[b] != 0;
    part[threadIdx.x] = nz;
    __syncthreads();
    const u32 total = block_scan_inplace<1024>(part, 1024, wtmp);
    u32 o = part[threadIdx.x];
    for (u32 b = b0; b < b1; b++) {
        const u32 c = table[b];
        if (c) {
            out_keys[o] = b;
            out_counts[o] = c;
            o++;
        }
    }
    if (threadIdx.x == 0)
        *n_out = total;
}

int dense_max_bits() { return DENSE_MAX_BITS; }

// table[key] = windows of [first, first + count) that hold key (the table is zeroed here)
hipError_t launch_dense_table(const u64 *words, u64 n_words, u64 first, u64 count, int bits, u32 *table, hipStream_t s)
{
    if (!func_attrs_ready(FA_DENSE)) {
        const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(dense_count_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS_BINS * 4);
        if (ae != hipSuccess)
            return ae;
        func_attrs_mark(FA_DENSE);
    }
    hipError_t e = hipMemsetAsync(table, 0, ((size_t)1 << bits) * 4, s);
    if (e != hipSuccess)
        return e;
    if (count == 0)
        return hipSuccess;
    const u32 passes = bits > 15 ? (1u << (bits - 15)) : 1u;
    const u64 per_wg = (u64)1024 * 16;
    u32 grid = (u32)std::min<u64>((count + per_wg - 1) / per_wg, 256);
    if (grid == 0)
        grid = 1;
    const size_t lds = (size_t)(bits >= 15 ? DENSE_LDS_BINS : (1 << bits)) * 4;
    for (u32 pass = 0; pass < passes; pass++)
        hipLaunchKernelGGL(dense_count_kernel, dim3(grid), dim3(1024), lds, s, words, n_words, first, count, bits, pass,
                           table);
    return hipGetLastError();
}

hipError_t launch_dense_compact(const u32 *table, int bits, u64 *out_keys, u32 *out_counts, u64 *n_out, hipStream_t s)
{
    hipLaunchKernelGGL(dense_compact_kernel, dim3(1), dim3(1024), 0, s, table, bits, out_keys, out_counts, n_out);
    return hipGetLastError();
}

hipError_t launch_dense_count(const u64 *words, u64 n_words, u64 first, u64 count, int bits, u32 *table, u64 *out_keys,
                              u32 *out_counts, u64 *n_out, hipStream_t s)
{
    const hipError_t e = launch_dense_table(words, n_words, first, count, bits, table, s);
    return e != hipSuccess ? e : launch_dense_compact(table, bits, out_keys, out_counts, n_out, s);
}

// dst[i] += src[i] (the partial tables of a multi-GPU dense count without RCCL)
__global__ __launch_bounds__(256) void table_add_kernel(u32 *__restrict__ dst, const u32 *__restrict__ src, u32 n)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] += src[i];
}

hipError_t launch_table_add(u32 *dst, const u32 *src, u32 n, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(table_add_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// ascending view of a histogram: leaf l (leaves are in key order) holds seg_cnt[l] groups at
// seg_off[l]; seg_pre is the exclusive scan of seg_cnt.  Copies groups [first, first+count) of the
// ascending order into dst arrays (index 0 = group `first`).  One workgroup per leaf.
__global__ __launch_bounds__(256) void gather_sorted_kernel(const u64 *__restrict__ seg_off,
                                                            const u32 *__restrict__ seg_cnt,
                                                            const u32 *__restrict__ seg_pre, u32 n_leaves,
                                                            u64 first, u64 count, const u64 *__restrict__ keys,
                                                            const u32 *__restrict__ counts,
                                                            u64 *__restrict__ dst_keys, u64 *__restrict__ dst_counts)
{
    const u32 l = blockIdx.x;
    if (l >= n_leaves)
        return;
    const u64 p0 = seg_pre[l], c = seg_cnt[l];
    if (c == 0 || p0 + c <= first || p0 >= first + count)
        return;
    const u64 lo = p0 < first ? first - p0 : 0;
    const u64 hi = p0 + c > first + count ? first + count - p0 : c;
    const u64 so = seg_off[l];
    for (u64 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        if (dst_keys)
            dst_keys[p0 + i - first] = keys[so + i];
        if (dst_counts)
            dst_counts[p0 + i - first] = counts[so + i];
    }
}

hipError_t launch_gather_sorted(const u64 *seg_off, const u32 *seg_cnt, const u32 *seg_pre, u32 n_leaves,
                                u64 first, u64 count, const u64 *keys, const u32 *counts, u64 *dst_keys,
                                u64 *dst_counts, hipStream_t s)
{
    if (n_leaves == 0 || count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(gather_sorted_kernel, dim3(n_leaves), dim3(256), 0, s, seg_off, seg_cnt, seg_pre, n_leaves,
                       first, count, keys, counts, dst_keys, dst_counts);
    return hipGetLastError();
}

}  // namespace dnagpu
