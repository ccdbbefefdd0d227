// superkmer_kernels.hip -- GROUP BY kmer, count(*) for long k-mers through super-k-mer (minimizer) partitioning.
//
// The MSD radix tree of count_kernels.hip moves every 8-byte key through two partition passes
// (3.5x the algorithmic traffic at 3 Gbase).  Here the partition passes move PACKED RUNS instead:
//
//   minimizer  of a k-mer = the m-mer (m = 15) among its w = k - m + 1 with the smallest 32-bit hash; a function
//              of the k-mer's content alone, so equal k-mers share it wherever they occur
//   bucket     = three digits (d0 < C0, d1 < 2^b1, d2 < 16) cut from a multiplicative re-mix of that hash
//   record     = a run of consecutive k-mers with the same minimizer hash (on random data 9 k-mers on
//              average, never more than SK_LMAX): 16 bytes = up to 54 bases (108 bits) + length + d1 + d2.
//              A record cut from a PLAIN tile (all of random sequence) is a run of k-mers whose leftmost minimum m-mer is
//              the SAME OCCURRENCE (same hash, same position): at most w k-mers, at most 2k - m bases, and the record
//              carries that m-mer's offset (SK_POS_*); every other record (low-complexity stretches, the partial tiles at
//              a sequence's end) has SK_REC_MULTI set.  sk_count tests records instead of k-mers with this (see there).
//
//   sk_hist0 / sk_scatter0   sweep the packed dna: hashes, window minima, runs -> records scattered
//                            into C0 coarse buckets (1.8 B per k-mer instead of 8).  Long sequences: the scatter sweep
//                            alone, every chunk of rows reserving slabs sized from a sampled histogram (unused slots =
//                            NULL records, bit 63 of the second word, dropped by level 1); the exact pair is the fall-back
//   sk_hist1 / sk_scatter1   records of a coarse bucket -> 2^b1 mid buckets, every record read once; normally without
//                            the histogram: mid buckets are regions sized from their parent, tiles reserve slots from
//                            global cursors (sk_spec_*), the exact level is the fall-back
//   sk_regroup               a mid bucket's records regrouped by d2 (read once): 16 final buckets of ~2,700 k-mers
//   sk_count                 a final bucket counted from its records in an LDS hash table (fingerprint slots); its groups go
//                            to its own output range
//   sk_count_big / sk_big_merge   long final buckets of few distinct keys (repeats): one table per bucket, slices of records
//   sk_slice_kmers / sk_expand_flat   the buckets sk_count does not take (oversize; the heavy buckets of repeats) -> keys;
//                            the ordinary tree takes over at level 2 (skew handling included)
//
// Keys meet their equals because the bucket is a function of the key; which bucket that is never shows in
// the result.  Group order: nodes in bucket order, ascending inside a node -- NOT globally ascending, which
// is why this engine sits behind dnagpu_count_kmers_unordered (PostgreSQL's GROUP BY order is unspecified,
// test.sql:95-104).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"

namespace dnagpu {

constexpr int SK_NT = 256;                       // threads of a front-end workgroup: four waves, each working tile after tile on its own
constexpr int SKW_ROWS = 63 * 32;                // rows (k-mers) a WAVE tile emits records for: lane 63 only supplies hashes
constexpr int SK_TILE_ROWS = (SK_NT / 64) * SKW_ROWS;   // one round of the workgroup's waves (chunks are cut at multiples of it)
constexpr int SK_MAX_C0 = 256;                   // most coarse buckets (digits of level 0): 2^32 rows need 195
constexpr int SKW_LIST = 768;                    // records of a wave tile listed in LDS (a tile of random bases has ~225 at k = 31, ~400 at k = 21)
// a full tile of at most this many (hash, position) runs is cut into exactly those: 1.7 times what random sequence gives
// (2016 rows in runs of (w + 1) / 2)
__host__ __device__ constexpr u32 sk_plain_max(int w) { return (u32)(2 * SKW_ROWS * 17 / (10 * (w + 1))) < (u32)SKW_LIST ? (u32)(2 * SKW_ROWS * 17 / (10 * (w + 1))) : (u32)SKW_LIST; }

int sk_tile_rows() { return SK_TILE_ROWS; }
int sk_max_c0() { return SK_MAX_C0; }
int sk_count_cap() { return 4096; }              // distinct keys of a bucket <= its k-mers <= half the 8128-slot table

typedef unsigned long long ull2_t __attribute__((ext_vector_type(2)));

// Second word of a record: bases 32..53 (bits 0..43) | (len - 1) << 44 | d1 << 49 | d2 << 59 | bit 63: NULL record.
// Bit 58 (SK_REC_MULTI): the record's k-mers do NOT all have their leftmost minimum m-mer at one place (nothing is known
// about where it is).  Clear: they do, and bits 39..43 (free: such a record has at most 2k - m <= 49 bases, i.e. payload
// bits 0..97 = bits 0..33 of this word) hold that m-mer's offset from the record's first base, 0 .. w - 1 <= 17.
constexpr int SK_REC_MULTI_BIT = 58;
constexpr int SK_POS_SHIFT = 39;                 // (k-mers of the record never reach these bits: consumers mask keys with kmask)
constexpr u32 SK_POS_MASK = 31u;
constexpr u32 SK_GPOS_MASK = 63u;                // low bits of a window minimum: the position of the minimum (see sk_front)

// 32-bit hash of an m-mer value (2 m <= 30 bits; bits above them ignored: t may carry a 16th base): two full-rate 24-bit
// multiplies, t[0..24) C1 + t[24..2m) C2 -- v_mul_u32_u24 + v_mad_u32_u24 behind one v_bfe (round 3's shift / add / xor mix
// took four operations behind a mask: the sweeps of level 0 are bound by VALU issue).  Its top 25 bits order the m-mers
// (sk_front): on random sequence that order makes runs as long as the old one (8.997 k-mers per record at k = 31 against
// 8.995) and the coarse buckets as even (tools/hash_eval2.py).  Not injective on its 25 bits, and it need not be: records
// are compared by their m-mers' VALUES wherever it matters (sk_count_clean, the walks by value).
__device__ __forceinline__ u32 sk_mix_raw(u32 t, u32 hi_bits /* 2 m - 24 */)
{
    return __umul24(t, 0x9E3779u) + __umul24(__builtin_amdgcn_ubfe(t, 24u, hi_bits), 0x85EBCBu);
}
__device__ __forceinline__ u32 sk_mix(u32 v) { return sk_mix_raw(v, 8u); }       // (v < 2^30)

// The three bucket digits of a k-mer, functions of its minimum m-mer (so equal k-mers share them).
//   d0 (coarse, < c0 <= 256) comes from the m-mer's HASH as the window minima carry it (hmin = hash bits 7..31 | 64 |
//   position): the level-0 histogram needs it for every row, and the hash is what a row has.  The minimum of w hashes is
//   small -- its high bits are biased, ~21 bits of it vary -- which is plenty for 256 coarse buckets and far too little for
//   the ~10^6 final ones (tried: final buckets of ~5 hash classes each, a third of them over sk_count's stage).
//   d1 (< 2^b1) and d2 (< 16) therefore come from the m-mer's 30-bit VALUE, which a record's builder cuts from the tile's
//   words at the minimum's position (once per record, not per row).
// v_mul_u32_u24 is a full-rate instruction; the 32-bit multiply of sk_fine_word runs once per record.
struct SkDigits {
    u32 d0, d1, d2;
};
__device__ __forceinline__ u32 sk_digit_word(u32 hmin) { return __umul24(hmin >> 7, 0x9E3779u); }
__device__ __forceinline__ u32 sk_digit0(u32 g, u32 c0) { return __umul24(g >> 16, c0) >> 16; }
__device__ __forceinline__ u32 sk_fine_word(u32 v) { return v * 0x9E3779B1u; }
__device__ __forceinline__ SkDigits sk_digits(u32 hmin, u32 v, u32 c0, u32 b1mask)
{
    const u32 f = sk_fine_word(v);
    SkDigits r;
    r.d0 = sk_digit0(sk_digit_word(hmin), c0);
    r.d1 = (f >> 23) & b1mask;
    r.d2 = (f >> 19) & 15u;
    return r;
}
// the window minimum an m-mer of value v makes (its position bits clear): what sk_front computes for it
__device__ __forceinline__ u32 sk_hash_of(u32 v) { return (sk_mix(v) & ~SK_GPOS_MASK) | 64u; }

// timing ablations (results invalid) exist in the diagnostic build (make STAMPS=1) only
#ifdef DNAGPU_STAMPS
#define SK_DBG(bit) ((dbg & (bit)) != 0)
#else
#define SK_DBG(bit) false
#endif

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

// LDS written by one lane and read by another of the same wave: DS operations of a wave execute in program
// order, so only the compiler has to be kept from reordering them
__device__ __forceinline__ void sk_wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lane i <- lane i + 1 (lane 63 <- 0) / lane i <- lane i - 1 (lane 0 <- 0): one VALU operation each
__device__ __forceinline__ u32 wave_next(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, false); }   // wave_shl:1
__device__ __forceinline__ u32 wave_prev(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false); }   // wave_shr:1

// ------------------------------------------------------------------------------------------------
// Front end of both dna sweeps, ONE WAVE per tile and nothing shared between waves: a tile = rows [row0, row0 + n_rows),
// n_rows <= SKW_ROWS; lane l owns the 32 rows row0 + 32 l ... (lane 63 owns none: it supplies the hashes that complete
// lane 62's windows).  What a lane needs of its neighbours -- the next lane's first W-1 hashes, the boundary minima, the
// last run break before its rows -- moves by DPP wave shifts and a DPP scan: no LDS, no barrier (round 2's workgroup
// tile exchanged them through LDS behind five workgroup barriers; the two sweeps together took 7.9 ms at 3 Gbase).
// Every lane ends with the window minimum of each of its rows in hm[]; sk_front_open adds the state of the record
// that is open at its first row.  Records are cut at tile boundaries (one extra record per 2016 rows: +0.4 %).
//   A minimum is hash bits 7..31 | 64 | position: the m-mers' hashes carry, in their low six bits, their own position
//   relative to the lane's first row (0 .. 31 + W - 1), so the minimum of a window is its smallest 25-bit hash and, among
//   equal ones, the LEFTMOST -- a function of the k-mer's content, like the hash.  Two neighbouring rows with equal hm
//   have the same m-mer occurrence as their minimum.  (Distinct m-mers may now tie on the 25 bits: 2^-25 per pair; the
//   bucket digits use the hash part only, so equal k-mers still share their bucket.)  Bit 6 is set in every real hash, so
//   that 0 is smaller than all of them:
//   BATCH (a table of sequences in one packed stream, dnagpu_count_kmers_batch: `marks` = one bit per base, set where a
//   sequence starts): an m-mer that reaches across a sequence start hashes to 0.  A k-mer's window holds such an m-mer
//   exactly when the k-mer itself reaches across a start, so the rows that are NOT rows of the table -- and only they --
//   have a minimum below 64 (hash part 0); they form runs of their own, which are counted and stored nowhere.
template <int W>
struct SkFront {
    u32 hm[32];
    u32 next_first;        // hm[0] of the next lane, its position in THIS lane's terms (+ 32)
    u32 prev_last;         // hm[31] of the lane before, its position in this lane's terms (- 32: a position before this lane's
                           // first row borrows from the hash part -- it then equals no minimum of this lane, rightly)
    u32 n_valid;           // rows of this lane that exist (0..32)
    u32 n_rows;            // rows of the tile
    bool plain;            // wave-uniform: a full tile whose records are its natural runs (see sk_front)
    // sk_front_open:
    u32 ns0;               // start row of the natural run that is open at this lane's first row
    u32 c0;                // length so far of the record open at this lane's first row
};

template <int W, bool BATCH = false>
__device__ __forceinline__ void sk_front(SkFront<W> &f, const u64 *__restrict__ words, u64 n_words, u64 pos0, u32 n_rows,
                                         u32 lmax, u32 mmask /* 2 m ones: the m-mer */,
                                         u64 *wsh /* this wave's [66] or null: the tile's words, word 0 = the one holding pos0 */,
                                         const u32 *__restrict__ marks = nullptr, u64 n_mark_words = 0)
{
    const int lane = threadIdx.x & 63;
    // the 64 bases from this lane's first row on (pos0 + 32 lane): 4 dwords
    const u64 pos = pos0 + (u64)lane * 32;
    const u64 w = pos >> 5;
    const unsigned sh = (unsigned)(pos & 31) * 2;          // wave-uniform
    const u64 w0 = w < n_words ? words[w] : 0, w1 = w + 1 < n_words ? words[w + 1] : 0;
    u64 lo = w0, hi = w1;
    u64 w2 = 0;
    if (sh || (wsh && lane == 63)) {
        w2 = w + 2 < n_words ? words[w + 2] : 0;
        if (sh) {
            lo = (w0 >> sh) | (w1 << (64 - sh));
            hi = (w1 >> sh) | (w2 << (64 - sh));
        }
    }
    if (wsh) {                                             // (the previous tile's payload reads are this wave's own, earlier in program order)
        wsh[lane] = w0;
        if (lane == 63) {
            wsh[64] = w1;
            wsh[65] = w2;
        }
    }
    const u32 d[4] = {(u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32)};
    constexpr int B = W - 1;                               // a window = W hashes = a[i .. i + B]
    constexpr int N = 32 + B;
    static_assert(W >= 9 && W <= 18, "window lengths of k = 23 .. 32 at m = 15");
    u32 a[N];
    const u32 hi_bits = (u32)__popc(mmask) - 24u;          // (wave-uniform: 6 for m = 15, 2 for m = 13)
    // BATCH: bit j of `within` = this lane's m-mer j reaches across no sequence start (no mark among the m - 1 bases behind
    // its first): the marks of the 64 bases from the lane's first on, smeared over m - 1 places
    u32 within = ~0u;
    if (BATCH) {
        const u64 mw = pos >> 5;
        const unsigned msh = (unsigned)(pos & 31);                                 // wave-uniform
        const u32 b0 = mw < n_mark_words ? marks[mw] : 0u, b1 = mw + 1 < n_mark_words ? marks[mw + 1] : 0u,
                  b2 = mw + 2 < n_mark_words ? marks[mw + 2] : 0u;
        const u64 S = ((u64)__builtin_amdgcn_alignbit(b2, b1, msh) << 32) | __builtin_amdgcn_alignbit(b1, b0, msh);
        u64 r = S >> 1;                            // bit i: a sequence starts at base i + 1 of the lane's
        r |= r >> 1;
        r |= r >> 2;
        r |= r >> 4;                               // ... starts among bases i + 1 .. i + 8
        const unsigned mm1 = (unsigned)__popc(mmask) / 2u - 1u;                    // m - 1 (12 or 14)
        r |= r >> (mm1 - 8u);                      // ... among bases i + 1 .. i + m - 1
        within = ~(u32)r;
    }
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int q = (2 * j) >> 5, s = (2 * j) & 31;
        const u32 t = __builtin_amdgcn_alignbit(d[q + 1], d[q], s);                // 16 bases: the m-mer's and more
        a[j] = (sk_mix_raw(t, hi_bits) & ~SK_GPOS_MASK) | (64u | (u32)j);          // (one v_and_or)
        if (BATCH)
            a[j] &= (u32)((int)(within << (31 - j)) >> 31);                        // (v_bfe_i32 + v_and)
    }
    // the next lane's first W-1 hashes complete this lane's windows (their positions: 32 ... in this lane's terms)
#pragma unroll
    for (int j = 0; j < B; j++)
        a[32 + j] = wave_next(a[j]) + 32u;
    // Window minima in three operations per row whatever W (van Herk / Gil-Werman): cut a[] into blocks of B; with
    // suf[i] = min of a[i .. end of i's block] and pre[i] = min of a[start of i's block .. i], the window [i, i + B]
    // is suf[i] and pre[i + B] together (i + B sits in the next block at i's offset).  (Round 2 took the minima by
    // doubling: 4 x 47 + 32 operations for 32 rows; this is 2 x 30 + 32.)
    u32 suf[N], pre[N];
#pragma unroll
    for (int i = N - 1; i >= 0; i--)
        suf[i] = (i == N - 1 || (i + 1) % B == 0) ? a[i] : min(a[i], suf[i + 1]);
#pragma unroll
    for (int i = 0; i < N; i++)
        pre[i] = (i % B == 0) ? a[i] : min(a[i], pre[i - 1]);
#pragma unroll
    for (int j = 0; j < 32; j++)
        f.hm[j] = min(suf[j], pre[j + B]);

    const u32 r0 = (u32)lane * 32;
    f.n_valid = r0 >= n_rows ? 0u : (n_rows - r0 < 32u ? n_rows - r0 : 32u);
    f.n_rows = n_rows;
    // neighbours' boundary minima
    f.next_first = wave_next(f.hm[0]) + 32u;
    f.prev_last = wave_prev(f.hm[31]) - 32u;
    // A tile may be PLAIN if it is full.  Its records are then the runs of equal minima, hash AND position (cut at the
    // tile's end): one m-mer occurrence stays the minimum of at most W <= 18 windows, so such a record never outgrows lmax,
    // and all its k-mers have their leftmost minimum m-mer at one place.  On random sequence a tile has ~225 of them.
    // Where one m-mer repeats (low-complexity sequence) its leftmost occurrence moves on row after row: a record per row.
    // A short stretch of that costs a few short records and nothing else; a tile that is MOSTLY that -- more than
    // sk_plain_max(W) records, which its callers count -- goes through the general walk instead, which cuts by the m-mer's
    // VALUE and every lmax rows (SK_REC_MULTI records).  (Round 3 and the first half of round 4 tested for a run of one
    // hash over 21 rows on every fourth row -- 55 operations per lane and tile -- and sent every tile with ONE such stretch
    // through the general walk: on real sequence that is every tenth tile, all of its records SK_REC_MULTI.)
    f.plain = n_rows == (u32)SKW_ROWS;
}

// bits [sh, sh + 64) of hi:lo, sh in 0 .. 127
__device__ __forceinline__ u64 sk_shr128(u64 lo, u64 hi, u32 sh)
{
    return sh < 64u ? funnel(lo, hi, sh) : hi >> (sh - 64u);
}

// ---- the general walk (a partial tile, low-complexity sequence, a plain tile whose list overflowed).  Runs are cut where a
// row's RUN KEY differs from the row before; a record is a run, cut every lmax rows and at the tile's end.
//   exact keys (a plain tile whose list overflowed, a partial tile of plain sequence): the minima themselves, hash and
//     position -- the same records as the branch-free walk of a plain tile;
//   value keys (SkValueKeys: everything else): the VALUE of the row's minimum m-mer.  Where one m-mer repeats (poly-A,
//     tandem repeats) the minimum's position moves on row after row while its value stays: runs by value keep such
//     stretches in records of lmax k-mers instead of one record per row.  All k-mers of such a record still share the
//     m-mer's value -- and so all three bucket digits -- but not its place: SK_REC_MULTI.
// The keys are computed row by row as the walk goes (a value key is a 128-bit funnel shift of the lane's bases: kept as
// an array, the 32 of them cost sk_scatter0 65 spilled registers).
template <int W, bool BATCH, bool BYV>
struct SkKeys {
    const SkFront<W> &f;
    u64 lo, hi;                    // BYV: the lane's 64 bases from its first row on
    u32 mmask;
    u32 next_first, prev_last;     // the neighbouring lanes' boundary keys, comparable with this lane's

    __device__ __forceinline__ SkKeys(const SkFront<W> &f_, const u64 *__restrict__ words, u64 n_words, u64 pos0, u32 mmask_)
        : f(f_), lo(0), hi(0), mmask(mmask_)
    {
        if (BYV) {
            // (what sk_front cut its m-mers from, read again: this is the rare path)
            const u64 pos = pos0 + (u64)(threadIdx.x & 63) * 32;
            const u64 w = pos >> 5;
            const unsigned sh = (unsigned)(pos & 31) * 2;
            const u64 w0 = w < n_words ? words[w] : 0, w1 = w + 1 < n_words ? words[w + 1] : 0;
            lo = w0;
            hi = w1;
            if (sh) {
                const u64 w2 = w + 2 < n_words ? words[w + 2] : 0;
                lo = (w0 >> sh) | (w1 << (64 - sh));
                hi = (w1 >> sh) | (w2 << (64 - sh));
            }
            next_first = wave_next((*this)(0));
            prev_last = wave_prev((*this)(31));
        } else {
            next_first = f.next_first;
            prev_last = f.prev_last;
        }
    }
    // (between two sweeps over the rows: the value keys are computed again, not kept -- 32 live keys spill)
    __device__ __forceinline__ void forget()
    {
        if (BYV)
            asm volatile("" : "+v"(lo), "+v"(hi));
    }
    __device__ __forceinline__ u32 operator()(int j) const
    {
        if (!BYV)
            return f.hm[j];
        if (BATCH && f.hm[j] < 64u)
            return ~0u;                            // (a row across a sequence start: no m-mer value is all ones)
        const u32 p = f.hm[j] & SK_GPOS_MASK;     // (<= 31 + W - 1 <= 48: the m-mer lies inside the lane's 64 bases)
        return (u32)sk_shr128(lo, hi, 2u * p) & mmask;
    }
};

// what the walk needs of the lanes before this one: the last break (a row whose key differs from the row before; the
// tile's first row counts) among the lanes' rows -> ns0 = the start row of the run that reaches this lane's first row from
// the left, and c0 = the length so far of the RECORD open there (records are cut every lmax rows of a run)
template <int W, typename Keys>
__device__ __forceinline__ void sk_front_open(SkFront<W> &f, const Keys &key, u32 lmax)
{
    const int lane = threadIdx.x & 63;
    // (opaque to the compiler: else the 32 row numbers r0 + j -- and, where they are widened, their 64-bit forms -- are
    // hoisted out of the callers' tile loops as loop invariants and spilled: sk_scatter0 carried 22 spilled VGPRs and
    // 92 bytes of scratch per lane for them, 0.4 ms of its 3.9 at 3 Gbase)
    u32 r0 = (u32)lane * 32;
    asm volatile("" : "+v"(r0));
    u32 lb = 0;
    u32 kprev = key.prev_last;
    const u32 k0 = key(0);
    u32 kcur = k0;
#pragma unroll 8
    for (int j = 0; j < 32; j++) {
        const bool brk = j == 0 ? (lane == 0 || kcur != kprev) : kcur != kprev;
        if (brk)
            lb = r0 + (u32)j + 1u;
        kprev = kcur;
        if (j < 31)
            kcur = key(j + 1);
    }
    const u32 before = wave_prev(wave_incl_max(lb));       // over the lanes before this one (lane 0: 0)
    // `before` >= 1 for every lane but lane 0 (row 0 is a break); lane 0's own row 0 is a break too
    f.ns0 = before ? before - 1u : 0u;
    const bool first_break = lane == 0 || k0 != key.prev_last;
    f.c0 = first_break ? 0u : (r0 - f.ns0) % lmax;
    if (first_break)
        f.ns0 = r0;
}

// the walk with the callback at EVERY row position (end = a record ends at this thread's row j; false for rows
// that do not exist), so that the callback may use wave-wide operations: emit(j, end, row, len, hmin, key)
template <int W, typename Keys, typename Emit>
__device__ __forceinline__ void sk_records_all(const SkFront<W> &f, const Keys &key, u32 lmax, Emit &&emit)
{
    u32 r0 = (u32)(threadIdx.x & 63) * 32;
    asm volatile("" : "+v"(r0));                  // (see sk_front_open)
    u32 c = f.c0;
    u32 kprev = 0, kcur = key(0);
#pragma unroll 8
    for (int j = 0; j < 32; j++) {
        const u32 nxt = j < 31 ? key(j + 1) : key.next_first;
        bool end = false;
        if ((u32)j < f.n_valid) {
            if (j > 0 && kcur != kprev)
                c = 0;
            end = r0 + (u32)j + 1 == f.n_rows || nxt != kcur || c + 1 == lmax;
        }
        emit(j, end, r0 + (u32)j, c + 1, f.hm[j], kcur);
        c = end ? 0u : c + 1;
        kprev = kcur;
        kcur = nxt;
    }
}

// walks the thread's rows in order and calls emit(j, end_row, len, hmin, key) for every record that ENDS in them (j = the
// row's index among the thread's 32).  (These walks run out of line -- sk_hist0_general, sk_scatter0_general -- over a memory
// copy of the SkFront.  Unrolled eight rows at a time: fully rolled, every row waits for its minimum to come back from scratch
// memory -- a sequence that is half poly-A took 31 ms instead of 24 --; fully unrolled, the walks' registers cost the kernels
// around them theirs.)
template <int W, typename Keys, typename Emit>
__device__ __forceinline__ void sk_records(const SkFront<W> &f, const Keys &key, u32 lmax, Emit &&emit)
{
    u32 r0 = (u32)(threadIdx.x & 63) * 32;
    asm volatile("" : "+v"(r0));                  // (see sk_front_open)
    u32 c = f.c0;
    u32 kprev = 0, kcur = key(0);
#pragma unroll 8
    for (int j = 0; j < 32; j++) {
        const u32 nxt = j < 31 ? key(j + 1) : key.next_first;
        if ((u32)j < f.n_valid) {
            if (j > 0 && kcur != kprev)
                c = 0;                                     // (a break at j == 0 is already in c0)
            const bool last = r0 + (u32)j + 1 == f.n_rows;   // records are cut at the tile's end
            const bool end = last || nxt != kcur || c + 1 == lmax;
            if (end) {
                emit(j, r0 + (u32)j, c + 1, f.hm[j], kcur);
                c = 0;
            } else {
                c++;
            }
        }
        kprev = kcur;
        kcur = nxt;
    }
}

// A FULL tile whose rows all have ONE minimum m-mer value -- poly-A, a microsatellite (the smallest rotation of the unit is
// every window's minimum), a tandem repeat shorter than a window -- needs no walk: by value it is one run, cut every lmax
// rows from the tile's first (exactly what the general walk makes of it, at a few percent of its cost: a sequence that is
// half poly-A spent 5.5 + 7.3 ms in the two level-0 kernels' walks).  Returns that value, or ~0u where the tile is no such
// tile (no m-mer value is all ones).  Equal hashes first (cheap; unequal ones settle it), then the values themselves, cut from
// the lane's bases at every row's minimum: a hash of 25 bits does not vouch for them.  All lanes of the wave call it.
template <int W, bool BATCH>
__device__ __forceinline__ u32 sk_uniform_value(const SkFront<W> &f, const u64 *__restrict__ words, u64 n_words, u64 pos0, u32 mmask)
{
    const int lane = threadIdx.x & 63;
    // (called over the MEMORY copy of the front the general walk takes, loops rolled eight rows at a time: over the front's
    // registers, fully unrolled, it cost sk_scatter0 70 spilled registers and sk_hist0 a wave per SIMD)
    u32 x = 0, y = f.hm[0];
#pragma unroll 2
    for (int j = 1; j < 32; j++) {
        x |= f.hm[j] ^ f.hm[0];
        if (BATCH)
            y &= f.hm[j];
    }
    const u32 h0 = (u32)__builtin_amdgcn_readfirstlane((int)f.hm[0]) >> 7;
    // (BATCH: bit 6 is clear in the minimum of a row that reaches across a sequence start -- such a tile is not uniform)
    const bool same = lane == 63 || ((x >> 7) == 0 && (f.hm[0] >> 7) == h0 && (!BATCH || (y & 64u)));
    if (__ballot(!same))
        return ~0u;
    const u64 pos = pos0 + (u64)lane * 32;
    const u64 w = pos >> 5;
    const unsigned sh = (unsigned)(pos & 31) * 2;
    const u64 w0 = w < n_words ? words[w] : 0, w1 = w + 1 < n_words ? words[w + 1] : 0;
    u64 lo = w0, hi = w1;
    if (sh) {
        const u64 w2 = w + 2 < n_words ? words[w + 2] : 0;
        lo = (w0 >> sh) | (w1 << (64 - sh));
        hi = (w1 >> sh) | (w2 << (64 - sh));
    }
    const u32 v0 = (u32)sk_shr128(lo, hi, 2u * (f.hm[0] & SK_GPOS_MASK)) & mmask;
    u32 d = 0;
#pragma unroll 2
    for (int j = 1; j < 32; j++)
        d |= ((u32)sk_shr128(lo, hi, 2u * (f.hm[j] & SK_GPOS_MASK)) & mmask) ^ v0;
    const u32 V = (u32)__builtin_amdgcn_readfirstlane((int)v0);
    const bool ok = lane == 63 || (d == 0 && v0 == V);
    return __ballot(!ok) ? ~0u : V;
}

// The general walk of sk_hist0 over a MEMORY copy of the front, its loops over the rows unrolled eight at a time: non-plain
// tiles are rare (none on random sequence), and fully unrolled over the front's registers the walk's registers were the
// kernel's (85 -> 128 VGPRs with spills: one wave less per SIMD for every tile).  (As a function of its own -- tried -- every
// call saves and restores the callee-saved registers through scratch: a sequence that is half poly-A took 31 ms.)
template <int W, bool BATCH>
__device__ __forceinline__ void sk_hist0_general(SkFront<W> &f, const u64 *__restrict__ words, u64 n_words, u64 pos0, u32 lmax, u32 mmask,
                                              u32 c0n, u32 *h /* LDS */)
{
    SkKeys<W, BATCH, true> vk(f, words, n_words, pos0, mmask);
    sk_front_open<W>(f, vk, lmax);
    vk.forget();
    if ((threadIdx.x & 63) < 63)
        sk_records<W>(f, vk, lmax, [&](int, u32, u32, u32 hmin, u32 v) {
            if (!BATCH || v != ~0u)
                atomicAdd(&h[sk_digit0(sk_digit_word(hmin), c0n)], 1u);
        });
}

// ------------------------------------------------------------------------------------------------
// sk_hist0: records per coarse digit of every chunk of rows (the histogram the generic prefix kernels take).  The
// chunk's wave tiles go round the workgroup's four waves; the waves meet only at the histogram (LDS adds).
template <int W, bool BATCH>
__global__ __launch_bounds__(SK_NT, 4) void sk_hist0_kernel(const Chunk *__restrict__ chunks, u32 n_chunks,
                                                         const u64 *__restrict__ words, u64 n_words, u64 first,
                                                         u32 lmax, u32 mmask, u32 c0n, u32 b1mask, u32 r0n /* digits of the root's split */,
                                                         u32 *__restrict__ hist, u32 *__restrict__ accum /* or null */,
                                                         const u32 *__restrict__ marks, u64 n_mark_words)
{
    __shared__ u32 h[SK_MAX_C0 + 64];             // (+ a word per lane for the adds that count nothing)
    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (u32 d = threadIdx.x; d < r0n; d += SK_NT)
        h[d] = 0;
    __syncthreads();
    for (u32 t0 = (u32)wave * SKW_ROWS; t0 < ch.len; t0 += SK_TILE_ROWS) {
        const u32 n_rows = ch.len - t0 < (u32)SKW_ROWS ? ch.len - t0 : (u32)SKW_ROWS;
        SkFront<W> f;
        sk_front<W, BATCH>(f, words, n_words, first + ch.off + t0, n_rows, lmax, mmask, nullptr, marks, n_mark_words);
        bool plain = f.plain;
        const u32 nf = lane == 62 ? ~f.hm[31] : f.next_first;
        if (plain) {                               // (the tile's records if it is cut as a plain tile: sk_scatter0 counts the same)
            u32 cnt = 0;
#pragma unroll
            for (int j = 0; j < 32; j++)
                cnt += ((j < 31 ? f.hm[j + 1] : nf) != f.hm[j]) ? 1u : 0u;
            plain = wave_sum(lane < 63 ? cnt : 0u) <= sk_plain_max(W);
        }
        if (plain) {
            // a record ends at every row whose successor has another minimum -- hash or position -- (lane 62's last row: the
            // tile's end).  No branch per row: a lane whose row ends nothing adds to a word of its own behind the histogram.
            if (lane < 63) {
#pragma unroll
                for (int j = 0; j < 32; j++) {
                    const u32 nxt = j < 31 ? f.hm[j + 1] : nf;
                    const u32 d0 = sk_digit0(sk_digit_word(f.hm[j]), c0n);
                    // (BATCH: a minimum below 64 is an m-mer across a sequence start -- 0 in its own lane, 32 + j seen from the
                    // lane before: such runs are counted and stored nowhere)
                    atomicAdd(&h[nxt != f.hm[j] && (!BATCH || f.hm[j] >= 64u) ? d0 : (u32)SK_MAX_C0 + (u32)lane], 1u);
                }
            }
        } else {
            SkFront<W> fm = f;                     // (a copy in memory for what follows: f itself stays in registers)
            const u32 uv = fm.plain ? sk_uniform_value<W, BATCH>(fm, words, n_words, first + ch.off + t0, mmask) : ~0u;
            if (uv != ~0u) {                       // one run of one m-mer value: its records are the cuts every lmax rows
                if (lane == 0)
                    atomicAdd(&h[sk_digit0(sk_digit_word(fm.hm[0]), c0n)], ((u32)SKW_ROWS + lmax - 1u) / lmax);
            } else {
                sk_hist0_general<W, BATCH>(fm, words, n_words, first + ch.off + t0, lmax, mmask, c0n, h);
            }
        }
    }
    __syncthreads();
    if (accum) {                                   // (the sampled histogram behind a slab sweep: one row for all chunks)
        for (u32 d = threadIdx.x; d < r0n; d += SK_NT)
            if (h[d])
                atomicAdd(&accum[d], h[d]);
        return;
    }
    u32 *row = hist + (u64)blockIdx.x * ROW_STRIDE;
    for (u32 d = threadIdx.x; d < r0n; d += SK_NT)
        row[d] = h[d];
}

// ------------------------------------------------------------------------------------------------
// What building a tile's records needs of its kernel (sk_scatter0): the wave's LDS list and words, the
// chunk's cursors, the record buffer.
struct SkBuild {
    u64 *wl;                // list entries (see sk_build)
    const u64 *wsh;         // the tile's packed words (word 0 = the one holding its first base)
    u32 *gpos;              // per coarse digit: the chunk's next slot
    const u32 *gend;        // ... and the end of its slots (slab mode; else all ones)
    ull2_t *recs;
    u32 fo, c0n, b1mask, mmask;
    int k, dbg;
};

// Every lane builds and stores the records of its share of the list.  exact: the listed records are runs of one minimum,
// hash and position (key = that minimum) -- they carry the minimum m-mer's offset, and the m-mer's value (for d1 / d2) is
// cut from the tile's words there; else (the walk by the m-mer's value: key = that value) SK_REC_MULTI.  Returns true if
// a record found its digit's slots used up (slab mode).
// mode 0: entries key << 32 | start row << 16 | end row, key = the m-mer's value (the walk by value);
// mode 1: the same, key = the minimum itself (the exact walk in four passes);
// mode 2: entries key << 32 | end row IN ROW ORDER, key = the minimum itself (a plain tile): a record starts where its
//         predecessor ended.
template <bool BATCH>
__device__ __forceinline__ bool sk_build(const SkBuild &b, u32 wrun, int mode)
{
    const bool exact = mode != 0;
    const int dbg = b.dbg;
    (void)dbg;
    bool dropped = false;
    if (SK_DBG(2))
        return false;
    for (u32 e = threadIdx.x & 63u; e < wrun; e += 64) {
        const u64 en = b.wl[e];
        const u32 key = (u32)(en >> 32);
        if (BATCH && (exact ? key < 64u : key == ~0u))
            continue;                              // (a run of rows across a sequence start: not rows of the table)
        const u32 end_row = (u32)en & 0xFFFFu;
        u32 v = key, hmin = key;
        if (exact) {                               // the minimum's position counts from the first row of the lane of end_row
            const u32 xq = (end_row & ~31u) + (key & SK_GPOS_MASK) + b.fo;
            v = (u32)funnel(b.wsh[xq >> 5], b.wsh[(xq >> 5) + 1], (xq & 31u) * 2u) & b.mmask;
        } else {
            hmin = sk_hash_of(v);
        }
        const SkDigits dg = sk_digits(hmin, v, b.c0n, b.b1mask);
        const u32 gslot = atomicAdd(&b.gpos[dg.d0], 1u);
        u32 start = ((u32)en >> 16) & 0xFFFFu;
        if (mode == 2)
            start = e ? ((u32)b.wl[e - 1] & 0xFFFFu) + 1u : 0u;
        const u32 len = end_row - start + 1u;
        // the record's 108 payload bits from bit 2 q of the tile's words: six dwords, four funnel shifts
        const u32 q = start + b.fo;                // first base of the run, relative to the tile's first word
        const u32 wi = q >> 5, sh = (q & 15u) * 2u;
        const u64 a0 = b.wsh[wi], a1 = b.wsh[wi + 1], a2 = b.wsh[wi + 2];
        const bool odd = (q & 16u) != 0;
        const u32 e0 = odd ? (u32)(a0 >> 32) : (u32)a0, e1 = odd ? (u32)a1 : (u32)(a0 >> 32),
                  e2 = odd ? (u32)(a1 >> 32) : (u32)a1, e3 = odd ? (u32)a2 : (u32)(a1 >> 32),
                  e4 = odd ? (u32)(a2 >> 32) : (u32)a2;
        u64 lo = ((u64)__builtin_amdgcn_alignbit(e2, e1, sh) << 32) | __builtin_amdgcn_alignbit(e1, e0, sh);
        u64 hi = ((u64)__builtin_amdgcn_alignbit(e4, e3, sh) << 32) | __builtin_amdgcn_alignbit(e3, e2, sh);
        const u32 nb = len + (u32)b.k - 1;
        if (nb < 32) {
            lo &= ((u64)1 << (2 * nb)) - 1;
            hi = 0;
        } else {
            hi &= ((u64)1 << (2 * (nb - 32))) - 1;
        }
        const u32 mpos = (end_row & ~31u) + (key & SK_GPOS_MASK) - start;
        const u32 tag = exact ? (mpos & SK_POS_MASK) << (SK_POS_SHIFT - 32) : 1u << (SK_REC_MULTI_BIT - 32);
        hi |= ((u64)(len - 1) << 44) | ((u64)dg.d1 << 49) | ((u64)dg.d2 << 59) | ((u64)tag << 32);
        ull2_t r;
        r.x = lo;
        r.y = hi;
        if (gslot >= b.gend[dg.d0])
            dropped = true;                        // (slab mode: the chunk's slots of this digit are used up)
        else if (!SK_DBG(1))
            b.recs[gslot] = r;
        else if (r.x == 0x1234567 && r.y == 0x89)
            b.recs[0] = r;
    }
    return dropped;
}

// The general walk of sk_scatter0 (over a memory copy of the front: see sk_hist0_general): a partial tile, or a full one of
// more than sk_plain_max(W) (hash, position) runs (low-complexity sequence) -- records = runs of one m-mer VALUE, cut every lmax
// rows (BYV; the walk by the minima themselves, !BYV, is kept for reference: nothing calls it now).  A list that overflows
// in one pass is redone in four passes of eight row positions each, which always fit.
template <int W, bool BATCH, bool BYV>
__device__ __forceinline__ bool sk_scatter0_walk(SkFront<W> &f, const SkBuild &bx, const u64 *__restrict__ words, u64 n_words,
                                                 u64 tile_pos, u32 lmax)
{
    const u64 below = ((u64)1 << (threadIdx.x & 63)) - 1;
    u64 *wl = bx.wl;
    bool dropped = false;
    SkKeys<W, BATCH, BYV> key(f, words, n_words, tile_pos, bx.mmask);
    sk_front_open<W>(f, key, lmax);
    for (int n_pass = BYV ? 1 : 4, pass = 0; pass < n_pass; pass++) {
        const int jlo = pass * (32 / n_pass), jhi = jlo + 32 / n_pass;
        u32 wrun = 0;
        key.forget();
        sk_records_all<W>(f, key, lmax, [&](int j, bool end_any, u32 end_row, u32 len, u32, u32 kj) {
            const bool end = end_any && j >= jlo && j < jhi;
            const u64 b = __ballot(end);
            if (end) {
                const u32 pos = wrun + (u32)__popcll(b & below);
                if (pos < (u32)SKW_LIST)
                    wl[pos] = ((u64)kj << 32) | ((u64)(end_row + 1 - len) << 16) | (u64)end_row;
            }
            wrun += (u32)__popcll(b);
        });
        if (wrun > (u32)SKW_LIST) {                // (only possible in a single pass)
            n_pass = 4;
            pass = -1;
            sk_wave_fence();
            continue;
        }
        sk_wave_fence();                           // list and words written by other lanes of this wave
        if (sk_build<BATCH>(bx, wrun, BYV ? 0 : 1))
            dropped = true;
        sk_wave_fence();                           // wsh / list are rewritten by the next pass / tile
    }
    return dropped;
}
template <int W, bool BATCH>
__device__ __forceinline__ bool sk_scatter0_general(SkFront<W> &f, const SkBuild &bx, const u64 *__restrict__ words, u64 n_words,
                                                    u64 tile_pos, u32 lmax)
{
    return sk_scatter0_walk<W, BATCH, true>(f, bx, words, n_words, tile_pos, lmax);
}

// ------------------------------------------------------------------------------------------------
// sk_scatter0: the same sweep; every record goes straight to its coarse bucket.  The chunk owns a range of
// every bucket (histogram prefix), so a per-digit cursor in LDS hands out consecutive slots; with at most a few
// dozen coarse digits the open cache lines of a workgroup are few and the 16-byte stores combine in L2.
//   record = lo: bases 0..31 of the run; hi: bases 32..53 (bits 0..43) | (len-1) << 44 | d1 << 49 | d2 << 59
template <int W, bool BATCH>
__global__ __launch_bounds__(SK_NT, 4) void sk_scatter0_kernel(const Chunk *__restrict__ chunks, u32 n_chunks,
                                                            const u64 *__restrict__ words, u64 n_words, u64 first, int k,
                                                            u32 lmax, u32 mmask, u32 c0n, u32 b1mask, u32 r0n,
                                                            const u32 *__restrict__ hist, const u32 *__restrict__ tot,
                                                            ull2_t *__restrict__ recs, int dbg,
                                                            u32 *__restrict__ slab /* or null: see below */,
                                                            const u32 *__restrict__ marks, u64 n_mark_words)
{
    // slab != null: level 0 WITHOUT its histogram sweep.  slab[0 .. r0n) = slots a chunk reserves of every digit's region
    // (the host's estimate from a sampled histogram + slack), slab[SK_MAX_C0 + 16 d] = the digit's global cursor (starts
    // at its region's start), slab[17 SK_MAX_C0 + d] += records the chunk stored, slab[18 SK_MAX_C0] = 1 if a chunk ran out
    // of slots (the caller then runs the exact pair).  The slots a chunk does not use are filled with NULL records (bit 63
    // of the second word), which level 1 skips.
    __shared__ u32 gpos[SK_MAX_C0];               // where the chunk's next record of each digit goes (all waves: LDS adds)
    __shared__ u32 gend[SK_MAX_C0], gbeg[SK_MAX_C0];
    __shared__ u64 wsh_all[SK_NT / 64][66];       // per wave: the tile's packed words -- record payloads are cut from here
    __shared__ u64 list_all[SK_NT / 64][SKW_LIST + 64];   // per wave: the tile's records (see below); + an entry per lane for writes that list nothing
    if (blockIdx.x >= n_chunks)
        return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const Chunk ch = chunks[blockIdx.x];
    if (slab) {
        for (u32 d = tid; d < r0n; d += SK_NT) {
            const u32 cap = slab[d];
            const u32 base = cap ? atomicAdd(&slab[SK_MAX_C0 + 16 * d], cap) : 0u;
            gpos[d] = base;
            gbeg[d] = base;
            gend[d] = base + cap;
        }
    } else {
        const u32 *hrow = hist + (u64)blockIdx.x * ROW_STRIDE;
        const u32 *trow = tot;                     // the root is node 0: its totals row is row 0
        for (u32 d = tid; d < r0n; d += SK_NT) {
            gpos[d] = trow[d] + hrow[d];
            gend[d] = ~0u;
        }
    }
    __syncthreads();
    u64 *wsh = wsh_all[wave], *wl = list_all[wave];
    bool dropped = false;
    SkBuild bx;
    bx.wl = wl;
    bx.wsh = wsh;
    bx.gpos = gpos;
    bx.gend = gend;
    bx.recs = recs;
    bx.fo = 0;
    bx.c0n = c0n;
    bx.b1mask = b1mask;
    bx.mmask = mmask;
    bx.k = k;
    bx.dbg = dbg;
    for (u32 t0 = (u32)wave * SKW_ROWS; t0 < ch.len; t0 += SK_TILE_ROWS) {
        const u32 n_rows = ch.len - t0 < (u32)SKW_ROWS ? ch.len - t0 : (u32)SKW_ROWS;
        const u64 tile_pos = first + ch.off + t0;
        const u32 fo = (u32)(tile_pos & 31);
        SkFront<W> f;
        sk_front<W, BATCH>(f, words, n_words, tile_pos, n_rows, lmax, mmask, wsh, marks, n_mark_words);
        // The records that end in this tile (usually ~225) are listed in the wave's LDS list (ballot + prefix count per
        // row position: no atomics), and every lane then builds the payloads of its share (sk_build) -- building them where
        // they end would run the payload code for all 32 row positions, ~9 times the work.  A tile that would overflow the
        // list (very short runs: low-complexity sequence) is redone in four passes of eight row positions each, which always
        // fit (8 x 63 records).
        bx.fo = fo;
        auto build = [&](u32 wrun, int mode) {
            if (sk_build<BATCH>(bx, wrun, mode))
                dropped = true;
        };
        if (f.plain) {
            // records = natural runs, listed IN ROW ORDER: every lane counts the records that end in its rows (a compare and
            // an add-with-carry per row), one wave scan gives it its first list slot, and a second sweep over its rows writes
            // the entries -- key << 32 | end row -- there; a row that ends nothing (and lane 63, which owns no rows) writes to
            // an entry of its own behind the list.  A record's first row is its predecessor's last + 1 (sk_build), so no
            // start is tracked here.  (Round 3 ranked the ends of every row position across the wave -- ballot, two mbcnt,
            // a popcount -- and tracked every lane's open run: ten vector and three scalar operations per row, seven now.)
            const u32 nf = lane == 62 ? ~f.hm[31] : f.next_first;
            u32 cnt = 0;
#pragma unroll
            for (int j = 0; j < 32; j++)
                cnt += ((j < 31 ? f.hm[j + 1] : nf) != f.hm[j]) ? 1u : 0u;
            cnt = lane < 63 ? cnt : 0u;
            const u32 incl = wave_incl_scan(cnt);
            const u32 wrun = (u32)__builtin_amdgcn_readlane((int)incl, 63);   // wave-uniform: records of the tile
            if (wrun <= sk_plain_max(W)) {
                u32 p = incl - cnt;
                const u32 dummy = (u32)SKW_LIST + (u32)lane;
                u32 r0 = (u32)lane * 32;
                asm volatile("" : "+v"(r0));       // (else the 32 row numbers are held across the tile loop: see sk_front_open)
#pragma unroll
                for (int j = 0; j < 32; j++) {
                    const bool end = ((j < 31 ? f.hm[j + 1] : nf) != f.hm[j]) && lane < 63;
                    wl[end ? p : dummy] = ((u64)f.hm[j] << 32) | (u64)(r0 + (u32)j);
                    p += end ? 1u : 0u;
                }
                sk_wave_fence();                   // list and words written by other lanes of this wave
                build(wrun, 2);
                sk_wave_fence();                   // wsh / list are rewritten by the next tile
                continue;
            }
        }
        // the general walk: a partial tile, or one of too many (hash, position) runs
        {
            SkFront<W> fm = f;                     // (a copy in memory for what follows: f itself stays in registers)
            const SkBuild bm = bx;
            const u32 uv = fm.plain ? sk_uniform_value<W, BATCH>(fm, words, n_words, tile_pos, mmask) : ~0u;
            if (uv != ~0u) {
                // one run of one m-mer value (sk_uniform_value): the records the walk by value would list -- value << 32 |
                // start row << 16 | end row, cut every lmax rows -- written directly
                const u32 n_rec = ((u32)SKW_ROWS + lmax - 1u) / lmax;
                for (u32 e = (u32)lane; e < n_rec; e += 64) {
                    const u32 s0 = e * lmax, e1 = (s0 + lmax < (u32)SKW_ROWS ? s0 + lmax : (u32)SKW_ROWS) - 1u;
                    wl[e] = ((u64)uv << 32) | (u64)((s0 << 16) | e1);
                }
                sk_wave_fence();
                if (sk_build<BATCH>(bm, n_rec, 0))
                    dropped = true;
                sk_wave_fence();
            } else if (sk_scatter0_general<W, BATCH>(fm, bm, words, n_words, tile_pos, lmax)) {
                dropped = true;
            }
        }
    }
    if (slab) {
        if (dropped)
            slab[18 * SK_MAX_C0] = 1u;
        __syncthreads();
        ull2_t null_rec;
        null_rec.x = 0;
        null_rec.y = (unsigned long long)1 << 63;
        for (u32 d = (u32)wave; d < r0n; d += SK_NT / 64) {        // (a wave per digit: four short store chains instead of one long)
            const u32 e = gend[d], u = gpos[d] < e ? gpos[d] : e;
            for (u32 i = u + (u32)lane; i < e; i += 64)
                __builtin_nontemporal_store(null_rec, &recs[i]);
            if (lane == 0 && u > gbeg[d])
                atomicAdd(&slab[17 * SK_MAX_C0 + d], u - gbeg[d]);
        }
    }
}

// the sampled chunks of a slab sweep's estimate: chunk i = rows [i * stride, i * stride + len) of the root
__global__ __launch_bounds__(256) void sk_sample_chunks_kernel(Chunk *__restrict__ chunks, u32 n_chunks, u32 stride, u32 len, u32 n)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks)
        return;
    const u64 off = (u64)i * stride;
    Chunk c;
    c.node = 0;
    c.off = (u32)off;
    c.len = off >= (u64)n ? 0u : (u32)((u64)len < (u64)n - off ? (u64)len : (u64)n - off);
    c.pad = 0;
    chunks[i] = c;
}

// Slots a chunk reserves of a digit's region: the digit's share of a chunk's records by the sampled histogram (est records
// among `sampled` rows; chunks of chunk_rows rows), + 1/32 + six standard deviations + 24, whole 128-byte lines.  Host and
// device share the formula (the host sizes the buffers with it).
__host__ __device__ inline u32 sk_isqrt(u64 v)
{
    u64 r = 0, bit = (u64)1 << 62;
    while (bit > v)
        bit >>= 2;
    while (bit) {
        if (v >= r + bit) {
            v -= r + bit;
            r = (r >> 1) + bit;
        } else {
            r >>= 1;
        }
        bit >>= 2;
    }
    return (u32)r;
}
__host__ __device__ inline u32 sk_slab_cap_of(u32 est, u64 chunk_rows, u64 sampled)
{
    if (est == 0)
        return 0;                                  // (a digit the sample never saw: a record of it overflows at once -> exact pair)
    const u64 e = ((u64)est * chunk_rows) / sampled + 1;   // (a FULL chunk's share: the last chunk of a sequence may be shorter)
    return (u32)((e + e / 32 + 6 * (u64)sk_isqrt(e) + 24 + 7) & ~(u64)7);
}
u32 sk_slab_cap(u32 est, u64 chunk_rows, u64 sampled) { return sk_slab_cap_of(est, chunk_rows, sampled); }

// slab[] from the sampled histogram est[] (see sk_scatter0_kernel): per-chunk slots, cursors at the regions' starts, the
// counters zeroed; nodes[d] = the coarse node of digit d (its region: n_chunks slabs, NULL records included)
__global__ __launch_bounds__(SK_MAX_C0) void sk_slab_init_kernel(const u32 *__restrict__ est, u32 r0n, u32 chunk_rows, u32 sampled,
                                                                u32 n_chunks, u32 child_meta, u32 *__restrict__ slab,
                                                                Node *__restrict__ nodes)
{
    __shared__ u64 sc[SK_MAX_C0];
    const u32 d = threadIdx.x;
    const u32 cap = d < r0n ? sk_slab_cap_of(est[d], chunk_rows, sampled) : 0u;
    sc[d] = (u64)cap * n_chunks;
    __syncthreads();
    if (d == 0) {
        u64 run = 0;
        for (u32 i = 0; i < (u32)SK_MAX_C0; i++) {
            const u64 c = sc[i];
            sc[i] = run;
            run += c;
        }
        slab[18 * SK_MAX_C0] = 0;
        slab[18 * SK_MAX_C0 + 1] = (u32)(run < 0xFFFFFFFFull ? run : 0xFFFFFFFFull);
    }
    __syncthreads();
    if (d < r0n) {
        const u32 start = (u32)(sc[d] < 0xFFFFFFFFull ? sc[d] : 0xFFFFFFFFull);
        slab[d] = cap;
        slab[SK_MAX_C0 + 16 * d] = start;
        slab[17 * SK_MAX_C0 + d] = 0;
        Node o;
        o.start = start;
        o.len = cap * n_chunks;
        o.meta = child_meta;
        o.split = 0;
        o.prefix = 0;
        o.child_base = 0;
        o.chunk_base = 0;
        nodes[d] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// sk_hist1: records of a chunk of a coarse node by d1; also the k-mers of every (node, d1) = mid bucket
constexpr int SK1_NT = 1024;
__global__ __launch_bounds__(SK1_NT) void sk_hist1_kernel(const Node *__restrict__ nodes, const Chunk *__restrict__ chunks,
                                                          u32 n_chunks, const ull2_t *__restrict__ recs,
                                                          u32 *__restrict__ hist, u32 *__restrict__ kcount, int shift)
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
    u32 i = threadIdx.x;
    for (; i + 3u * SK1_NT < ch.len; i += 4u * SK1_NT) {           // four loads in flight per thread
        u64 m[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            m[j] = __builtin_nontemporal_load(&hi[(u64)(i + (u32)j * SK1_NT) * 2]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (m[j] >> 63)
                continue;                          // (a NULL record: an unused slot of a slab sweep)
            const u32 d1 = (u32)(m[j] >> shift) & (R - 1);
            atomicAdd(&h[d1], 1u);
            atomicAdd(&kc[d1], (u32)((m[j] >> 44) & 31) + 1u);
        }
    }
    for (; i < ch.len; i += SK1_NT) {
        const u64 m = __builtin_nontemporal_load(&hi[(u64)i * 2]);
        if (m >> 63)
            continue;
        const u32 d1 = (u32)(m >> shift) & (R - 1);
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

// A SAMPLE of the same histogram, for the regions of a speculative level 1 over uneven coarse buckets (repeats): the chunks
// are pieces of SK1_SAMPLE_LEN records, one in every SK1_SAMPLE_EVERY of a node's records (the host lists them); est[mid
// bucket] += the records of the piece that fall into it.
constexpr u32 SK1_SAMPLE_LEN = 1024, SK1_SAMPLE_EVERY = 8;
u32 sk_sample1_len() { return SK1_SAMPLE_LEN; }
u32 sk_sample1_every() { return SK1_SAMPLE_EVERY; }
__global__ __launch_bounds__(SK1_NT) void sk_sample1_kernel(const Node *__restrict__ nodes, const Chunk *__restrict__ chunks,
                                                            u32 n_chunks, const ull2_t *__restrict__ recs, u32 *__restrict__ est)
{
    __shared__ u32 h[ROW_STRIDE];
    if (blockIdx.x >= n_chunks)
        return;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    const u32 R = 1u << nd.split;
    for (u32 d = threadIdx.x; d < R; d += SK1_NT)
        h[d] = 0;
    __syncthreads();
    const u64 *hi = reinterpret_cast<const u64 *>(recs + (u64)nd.start + ch.off) + 1;
    for (u32 i = threadIdx.x; i < ch.len; i += SK1_NT) {
        const u64 m = __builtin_nontemporal_load(&hi[(u64)i * 2]);
        if (!(m >> 63))
            atomicAdd(&h[(u32)(m >> 49) & (R - 1)], 1u);
    }
    __syncthreads();
    for (u32 d = threadIdx.x; d < R; d += SK1_NT)
        if (h[d])
            atomicAdd(&est[nd.child_base + d], h[d]);
}

// Slots of a mid bucket whose sample held est records: the estimate, five standard deviations of it (a Poisson count
// scaled by the sampling factor) and 128, whole 128-byte lines.  (A region's unused slots cost memory only: the mid
// node's length is what the sweep drew.)
__global__ __launch_bounds__(256) void sk_sampled_caps_kernel(const u32 *__restrict__ est, u32 n_mid, u32 *__restrict__ rcap)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_mid)
        return;
    const u64 e = est[i];
    const u64 cap = e * SK1_SAMPLE_EVERY + 5ull * SK1_SAMPLE_EVERY * sk_isqrt(e) + 128ull + 7ull;
    rcap[i] = (u32)(cap < 0xFFFFFFF8ull ? cap : 0xFFFFFFF8ull) & ~7u;
}

// sk_scatter1: tiles of 8192 records, grouped by d1 in LDS and copied to their mid buckets in sorted order, 16 bytes per
// lane (16 records per digit and tile on average: 256-byte runs).  Full tiles keep their records in registers between the
// one read and the staging; the chunk's last, partial tile ranks an index list and gathers.
constexpr int SK1_ITEMS = 8;
constexpr int SK1_TILE = SK1_NT * SK1_ITEMS;
constexpr int SK1_STAGE = SK1_TILE / 2 + 64;     // records the stage holds: half a tile and a digit's worth of slack
// SPEC (level 1 without its histogram): the mid buckets are REGIONS of the destination buffer -- spec[2 node] = start of
// the node's first, spec[2 node + 1] = records each holds, the cursors start at the regions' starts (sk_spec_*_kernel below)
// -- and the sweep counts the k-mers per mid bucket itself (kcount).  A record that does not fit its region is dropped and
// raises *over: the caller then runs the exact level (histogram, prefix, this kernel without SPEC).
template <bool SPEC>
__global__ __launch_bounds__(SK1_NT, 8) void sk_scatter1_kernel(const Node *__restrict__ nodes, const Chunk *__restrict__ chunks,
                                                             u32 n_chunks, const ull2_t *__restrict__ src_all,
                                                             ull2_t *__restrict__ dst_all, const u32 *__restrict__ hist,
                                                             const u32 *__restrict__ tot, int shift, u32 *__restrict__ gcur,
                                                             const u32 *__restrict__ spec, u32 *__restrict__ kcount,
                                                             u32 *__restrict__ over, const u32 *__restrict__ rstart,
                                                             const u32 *__restrict__ rcapv /* or null: see sk_spec_* below */)
{
    __shared__ u32 cnt[ROW_STRIDE];
    __shared__ u32 glim[SPEC ? ROW_STRIDE : 1];    // SPEC: the end of digit d's region
    // gadj[d]: where the tile's sorted slot sl of digit d goes = gadj[d] + sl (the digit's base in the destination minus its
    // offset in the tile).  gpos[d] (only without global cursors: the by-d2 split of heavy buckets): the chunk's running
    // position in digit d.
    __shared__ u32 gadj[ROW_STRIDE];
    __shared__ u32 gpos[SPEC ? 1 : ROW_STRIDE];
    // a full tile: about half of its records at a time, in sorted order; a partial tile: the index list (16 KB of it)
    __shared__ __attribute__((aligned(16))) ull2_t stage[SK1_STAGE];
    unsigned short *idx = reinterpret_cast<unsigned short *>(stage);
    // SPEC: while a tile is counted the stage's first words hold, per digit, records | k-mers << 32 -- ONE 64-bit LDS add
    // per record counts both (round 3: two 32-bit adds); thread d keeps digit d's k-mers of the whole chunk in a register
    u64 *c64 = reinterpret_cast<u64 *>(stage);
    u32 kacc = 0;
    __shared__ u32 wtmp[SK1_NT / 64];
    __shared__ u32 split[4];                       // a full tile's halves: [0] first digit of the second, [1] its offset, [2] skew
    if (blockIdx.x >= n_chunks)
        return;
    int tid = threadIdx.x;
    const Chunk ch = chunks[blockIdx.x];
    const Node nd = nodes[ch.node];
    const u32 R = 1u << nd.split;
    const u32 *hrow = hist + (u64)blockIdx.x * ROW_STRIDE;
    const u32 *trow = tot + (u64)nd.chunk_base * ROW_STRIDE;          // absolute base of every digit of the node
    // gcur != null: the node's digits have GLOBAL cursors (a copy of `tot`: they start at the digits' absolute bases) and
    // every tile reserves its slots there, one returning add per digit, in flight while the tile is staged.  The chunks of a
    // node then fill a mid bucket in arrival order instead of each into a range of its own: a run's partial first and last
    // cache lines are completed by whichever workgroup writes to the bucket next, soon, while they are still in L2, and
    // the per-tile cursor update with its barrier is gone (3.27 -> 2.92 ms at 3 Gbase).  ~300 adds per address and count.
    u32 *gc = gcur ? gcur + (u64)nd.chunk_base * ROW_STRIDE : nullptr;
    if (!SPEC && !gc)
        for (u32 d = tid; d < R; d += SK1_NT)
            gpos[d] = trow[d] + hrow[d];
    const ull2_t *src = src_all + (u64)nd.start + ch.off;
    const u32 dmask = R - 1;
    bool dropped = false;
    if (SPEC) {
        // digit d's region: [reg0 + d rcap, reg0 + (d + 1) rcap) from its parent's size, or -- over uneven coarse buckets --
        // rstart / rcapv per mid bucket from a sampled histogram
        const u32 reg0 = spec[2 * ch.node], rcap = spec[2 * ch.node + 1];
        for (u32 d = tid; d < R; d += SK1_NT)
            glim[d] = rcapv ? rstart[nd.child_base + d] + rcapv[nd.child_base + d] : reg0 + (d + 1) * rcap;
    }
    // after the tile's digit counts have become offsets (cnt[d]; cnt[R] = the tile's records): digit tid's slots in the
    // destination -- reserved from its global cursor, or the chunk's own running position -- as gadj[tid]
    // The reservation is ISSUED here (a returning add on a global cursor: microseconds) and FINISHED -- gadj written -- only
    // where the tile's first write-out needs it: the staging of the first half runs while it is in flight.
    u32 pl_base = 0;
    bool pl_pending = false;
    auto place_issue = [&]() {
        if ((u32)tid < R) {
            const u32 lo = cnt[tid], c = cnt[tid + 1] - lo;
            pl_base = 0;
            if (gc) {
                if (c)
                    pl_base = atomicAdd(&gc[tid], c);
            } else if (!SPEC) {
                pl_base = gpos[tid];
                gpos[tid] = pl_base + c;
            }
            gadj[tid] = 0u - lo;
            pl_pending = true;
        }
    };
    auto place_finish = [&]() {
        if (pl_pending) {
            gadj[tid] += pl_base;                  // (SPEC: a record that falls behind its region's end is dropped, and
            pl_pending = false;                    // reported, where it is written)
        }
    };
    // the tile's digit counts (cnt[], or -- SPEC -- c64[] with the k-mers in the high words) -> exclusive offsets in cnt[],
    // cnt[R] = the tile's records
    auto scan_counts = [&]() {
        u32 v = 0;
        if ((u32)tid <= R) {
            if (SPEC) {
                const u64 w = c64[tid];
                v = (u32)w;
                kacc += (u32)(w >> 32);
            } else {
                v = cnt[tid];
            }
        }
        block_scan_value<SK1_NT>(v, cnt, (int)R + 1, wtmp, tid);       // (R + 1 <= 513)
    };
    for (u32 t0 = 0; t0 < ch.len; t0 += SK1_TILE) {
        const u32 n_tile = ch.len - t0 < (u32)SK1_TILE ? ch.len - t0 : (u32)SK1_TILE;
        asm volatile("" : "+v"(tid));              // (thread-derived addresses recomputed per tile, not held: the kernel lives on 64 registers)
        for (u32 d = tid; d <= R; d += SK1_NT) {   // (bin R: NULL records -- unused slots of a slab sweep -- sort behind all digits)
            if (SPEC)
                c64[d] = 0;
            else
                cnt[d] = 0;
        }
        if (tid == 0) {
            split[0] = R;                          // (a tile of at most SK1_STAGE records: one half)
            split[2] = 0;
        }
        __syncthreads();
        bool gather = n_tile != (u32)SK1_TILE, placed = false;
        if (!gather) {
            // ---- a full tile (all tiles of a chunk but its last): every record is read ONCE, 16 bytes per lane and all
            // eight of a thread in flight; it waits in registers while the tile's digits are counted, then takes its slot in
            // the sorted order from its digit's LDS cursor and goes through the stage, digits [0, s) first and [s, R) second
            // (s = the digit that holds sorted slot SK1_STAGE: both halves fit unless that one digit has more than 128
            // records in this tile -- skew, the gather below), and leaves in runs -- 16 bytes per lane, consecutive lanes to
            // consecutive addresses.  No sorted position is ever kept in a register (round 3 kept eight, and spilled 23
            // VGPRs around the scan: 96 bytes of scratch per lane and tile).  (Reading the digits first and gathering the
            // records by index afterwards -- the partial tile's way below -- brings every record in twice: 16 GB moved for 10.7.)
            ull2_t rec[SK1_ITEMS];
#pragma unroll
            for (int j = 0; j < SK1_ITEMS; j++)
                rec[j] = src[t0 + tid + j * SK1_NT];
#pragma unroll
            for (int j = 0; j < SK1_ITEMS; j++) {
                const bool null = (rec[j].y >> 63) != 0;
                const u32 dg = null ? R : (u32)(rec[j].y >> shift) & dmask;
                if (SPEC)
                    atomicAdd(reinterpret_cast<unsigned long long *>(&c64[dg]),
                              null ? 1ull : 1ull | ((unsigned long long)(((u32)(rec[j].y >> 44) & 31u) + 1u) << 32));
                else
                    atomicAdd(&cnt[dg], 1u);
            }
            __syncthreads();
            scan_counts();
            const u32 n_valid = cnt[R];
            if ((u32)tid < R && cnt[tid] <= (u32)SK1_STAGE && cnt[tid + 1] > (u32)SK1_STAGE) {
                split[0] = (u32)tid;               // (exactly one digit holds slot SK1_STAGE, if the tile has that many)
                split[1] = cnt[tid];
                split[2] = n_valid - cnt[tid] > (u32)SK1_STAGE ? 1u : 0u;
            }
            place_issue();
            placed = true;
            __syncthreads();                       // (the offsets have been read: the cursors below move them)
            const u32 s_dig = split[0];
            gather = split[2] != 0;
            if (!gather) {
                const u32 s_off = s_dig < R ? split[1] : n_valid;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const u32 lo = h ? s_off : 0u, hi = h ? n_valid : s_off;   // sorted slots of this half
                    if (lo == hi)
                        continue;                  // (wave-uniform, workgroup-uniform)
#pragma unroll
                    for (int j = 0; j < SK1_ITEMS; j++) {
                        const u32 dg = (u32)(rec[j].y >> shift) & dmask;
                        if (!(rec[j].y >> 63) && (dg >= s_dig) == (h != 0))
                            stage[atomicAdd(&cnt[dg], 1u) - lo] = rec[j];
                    }
                    place_finish();
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < (SK1_STAGE + SK1_NT - 1) / SK1_NT; j++) {
                        const u32 sl = lo + (u32)tid + (u32)j * SK1_NT;
                        if (sl >= hi)
                            continue;
                        const ull2_t r = stage[sl - lo];
                        const u32 d = (u32)(r.y >> shift) & dmask;
                        // (a plain store: a run's first and last cache lines are partial, and the same digit's next run --
                        // this workgroup's next tile -- completes them; kept in L2 they merge more often: 3.44 -> 3.30 ms)
                        const u32 p = gadj[d] + sl;
                        if (!SPEC || p < glim[d])
                            dst_all[p] = r;
                        else
                            dropped = true;
                    }
                    __syncthreads();
                }
            } else {
                place_finish();
                for (u32 d = tid; d <= R; d += SK1_NT)
                    cnt[d] = 0;                    // (the gather ranks the tile again; its slots are reserved already)
                __syncthreads();
            }
        }
        if (gather) {
            u32 dig[SK1_ITEMS], rank[SK1_ITEMS];
#pragma unroll
            for (int j = 0; j < SK1_ITEMS; j++) {
                const u32 i = tid + j * SK1_NT;
                dig[j] = 0;
                rank[j] = 0;
                if (i < n_tile) {
                    const u64 m = reinterpret_cast<const u64 *>(src + t0 + i)[1];
                    const bool null = (m >> 63) != 0;
                    dig[j] = null ? R : (u32)(m >> shift) & dmask;
                    if (SPEC && !placed) {         // (records | k-mers << 32: the returned low word is the rank)
                        rank[j] = (u32)atomicAdd(reinterpret_cast<unsigned long long *>(&c64[dig[j]]),
                                                 null ? 1ull : 1ull | ((unsigned long long)(((u32)(m >> 44) & 31u) + 1u) << 32));
                    } else {
                        rank[j] = atomicAdd(&cnt[dig[j]], 1u);
                    }
                }
            }
            __syncthreads();
            if (SPEC && !placed)
                scan_counts();
            else
                block_scan_small<SK1_NT>(cnt, (int)R + 1, wtmp, tid);  // cnt -> exclusive offsets; cnt[R] = the tile's records
            if (!placed) {
                place_issue();
                place_finish();
            }
#pragma unroll
            for (int j = 0; j < SK1_ITEMS; j++) {
                const u32 i = tid + j * SK1_NT;
                if (i < n_tile)
                    idx[cnt[dig[j]] + rank[j]] = (unsigned short)i;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < SK1_ITEMS; j++) {
                const u32 sl = tid + j * SK1_NT;
                if (sl < cnt[R]) {
                    const ull2_t r = src[t0 + idx[sl]];
                    const u32 d = (u32)(r.y >> shift) & dmask;
                    const u32 p = gadj[d] + sl;
                    if (!SPEC || p < glim[d])
                        __builtin_nontemporal_store(r, &dst_all[p]);
                    else
                        dropped = true;
                }
            }
            __syncthreads();
        }
    }
    if (SPEC) {
        if ((u32)tid < R && kacc)
            atomicAdd(&kcount[nd.child_base + (u32)tid], kacc);
        if (dropped)
            *over = 1u;
    }
}

// The regions of a speculative level 1: a node of len records split R ways gives each child len / R + 12.5 % + 72 slots,
// whole 128-byte lines (on random sequence a mid bucket is Poisson around ~4,800: the slack is > 8 sigma).  Host and
// device use this one formula (the host sizes the buffers with it).
__host__ __device__ inline u32 sk_spec_cap(u32 len, u32 R)
{
    return ((u32)(((u64)len + (len >> 3)) / R) + 72u + 7u) & ~7u;
}
u64 sk_spec_span(u32 len, int bits) { return len ? (u64)sk_spec_cap(len, 1u << bits) << bits : 0; }

// spec[2 i] = first slot of node i's regions, spec[2 i + 1] = slots per child; out[0] = slots of all, out[1] = 1 if they
// pass 2^32.  One workgroup (at most 256 nodes: the coarse buckets).
__global__ __launch_bounds__(SK_MAX_C0) void sk_spec_caps_kernel(const Node *__restrict__ nodes, u32 n_nodes, u32 *__restrict__ spec,
                                                                u32 *__restrict__ out)
{
    __shared__ u64 span[SK_MAX_C0];
    const u32 i = threadIdx.x;
    u32 cap = 0;
    u64 sp = 0;
    if (i < n_nodes && nodes[i].split) {
        cap = sk_spec_cap(nodes[i].len, 1u << nodes[i].split);
        sp = (u64)cap << nodes[i].split;
    }
    span[i] = sp;
    __syncthreads();
    if (i == 0) {
        u64 run = 0;
        for (u32 q = 0; q < (u32)SK_MAX_C0; q++) {
            const u64 c = span[q];
            span[q] = run;
            run += c;
        }
        out[0] = (u32)(run < 0xFFFFFFFFull ? run : 0xFFFFFFFFull);
        out[1] = run > 0xFFFFFFFFull ? 1u : 0u;
        out[2] = 0;                                // (the sweep's overflow flag)
    }
    __syncthreads();
    if (i < n_nodes) {
        spec[2 * i] = (u32)(span[i] < 0xFFFFFFFFull ? span[i] : 0xFFFFFFFFull);
        spec[2 * i + 1] = cap;
    }
}

// the cursors at the regions' starts (a workgroup per node, a thread per child)
__global__ __launch_bounds__(ROW_STRIDE) void sk_spec_init_kernel(const Node *__restrict__ nodes, u32 n_nodes, const u32 *__restrict__ spec,
                                                                 u32 *__restrict__ gcur, const u32 *__restrict__ rstart)
{
    const u32 i = blockIdx.x, d = threadIdx.x;
    if (i >= n_nodes)
        return;
    const Node nd = nodes[i];
    if (nd.split && d < (1u << nd.split))
        gcur[(u64)nd.chunk_base * ROW_STRIDE + d] = rstart ? rstart[nd.child_base + d] : spec[2 * i] + d * spec[2 * i + 1];
}

// the mid nodes a speculative sweep leaves (what level_children makes of a histogram): start = region start, len = records
// drawn; a cursor past its region's end raises *over
__global__ __launch_bounds__(ROW_STRIDE) void sk_spec_nodes_kernel(const Node *__restrict__ nodes, u32 n_nodes, const u32 *__restrict__ spec,
                                                                  const u32 *__restrict__ gcur, Node *__restrict__ next,
                                                                  u32 *__restrict__ over, const u32 *__restrict__ rstart,
                                                                  const u32 *__restrict__ rcapv)
{
    const u32 i = blockIdx.x, d = threadIdx.x;
    if (i >= n_nodes)
        return;
    const Node nd = nodes[i];
    if (nd.split == 0) {
        if (d == 0) {
            Node o = nd;
            o.split = 0;
            next[nd.child_base] = o;
        }
        return;
    }
    if (d >= (1u << nd.split))
        return;
    const int bits = (int)nd.split, rem = (int)(nd.meta & 0xff);
    const u32 cap = rcapv ? rcapv[nd.child_base + d] : spec[2 * i + 1];
    const u32 start = rcapv ? rstart[nd.child_base + d] : spec[2 * i] + d * cap;
    u32 used = gcur[(u64)nd.chunk_base * ROW_STRIDE + d] - start;
    if (used > cap) {
        used = cap;
        *over = 1u;
    }
    Node c;
    c.start = start;
    c.len = used;
    c.meta = (u32)(rem - bits) | ((nd.meta & NODE_BUF) ^ NODE_BUF) | ((bits == rem) ? NODE_TERMINAL : 0u);
    c.split = 0;
    c.prefix = nd.prefix | ((u64)d << (rem - bits));
    c.child_base = 0;
    c.chunk_base = 0;
    next[nd.child_base + d] = c;
}

// exclusive scan over lanes 0..15 (one DPP row); lanes >= 16 get garbage
__device__ __forceinline__ u32 row16_excl_scan(u32 x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    return (u32)v - x;
}

// ------------------------------------------------------------------------------------------------
// Buckets that sk_count does not take (more k-mers, records or quads than its tables hold; the heavy mid buckets of long
// repeats with millions of k-mers) are expanded to keys in record order by many waves at once, with no host step
// in between: a bucket's records are cut into slices of SKF_SLICE records (sk_slice_count -> scan -> sk_slice_fill),
// sk_slice_kmers sums every slice's k-mers, a scan of the sums gives every slice its key offset and every bucket its
// key range (sk_slice_nodes: one key node per bucket, which the ordinary levels split with their skew paths), and
// sk_expand_flat expands slice by slice (64 records at a time: lengths -> wave prefix -> keys staged in LDS -> one
// contiguous copy).
constexpr int SKF_SLICE = 4096;
int sk_flat_slice() { return SKF_SLICE; }

__global__ __launch_bounds__(256) void sk_slice_count_kernel(const Node *__restrict__ buckets, u32 nb, u32 *__restrict__ n_slices)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb)
        n_slices[i] = (buckets[i].len + (u32)SKF_SLICE - 1u) / (u32)SKF_SLICE;
}

// one wave per bucket: its slices' first records and record counts, at slice_first[bucket]
__global__ __launch_bounds__(256) void sk_slice_fill_kernel(const Node *__restrict__ buckets, u32 nb,
                                                            const u32 *__restrict__ slice_first, u32 *__restrict__ slice_rec0,
                                                            u32 *__restrict__ slice_nrec)
{
    const u32 lane = threadIdx.x & 63;
    const u32 i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nb)
        return;
    const u32 start = buckets[i].start, len = buckets[i].len, s0 = slice_first[i];
    const u32 ns = (len + (u32)SKF_SLICE - 1u) / (u32)SKF_SLICE;
    for (u32 j = lane; j < ns; j += 64) {
        slice_rec0[s0 + j] = start + j * (u32)SKF_SLICE;
        slice_nrec[s0 + j] = min((u32)SKF_SLICE, len - j * (u32)SKF_SLICE);
    }
}

__global__ __launch_bounds__(256) void sk_slice_kmers_kernel(const ull2_t *__restrict__ recs, const u32 *__restrict__ slice_rec0,
                                                             const u32 *__restrict__ slice_nrec, u32 n_slices,
                                                             u32 *__restrict__ slice_km)
{
    const int lane = threadIdx.x & 63;
    const u32 si = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (si >= n_slices)
        return;
    const u64 *hi = reinterpret_cast<const u64 *>(recs + (u64)slice_rec0[si]) + 1;
    const u32 n = slice_nrec[si];
    u32 sum = 0;
    for (u32 r = lane; r < n; r += 64)
        sum += (u32)((__builtin_nontemporal_load(&hi[(u64)r * 2]) >> 44) & 31) + 1u;
    sum = wave_sum(sum);
    if (lane == 0)
        slice_km[si] = sum;
}

// key node of bucket i: the keys of its slices, [key_base + slice_koff[first slice], ... of the next bucket).
// expect_kmers (buckets[i].child_base, the partition's own count) must agree: *n_bad counts the buckets where not.
__global__ __launch_bounds__(256) void sk_slice_nodes_kernel(const Node *__restrict__ buckets, u32 nb,
                                                             const u32 *__restrict__ slice_first, const u32 *__restrict__ slice_koff,
                                                             u32 n_slices, const u32 *__restrict__ total_keys, u32 key_base, int k,
                                                             int check_kmers, Node *__restrict__ out,
                                                             unsigned long long *__restrict__ n_bad)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb)
        return;
    const u32 s0 = slice_first[i], s1 = i + 1 < nb ? slice_first[i + 1] : n_slices;
    const u32 k0 = s0 < n_slices ? slice_koff[s0] : *total_keys, k1 = s1 < n_slices ? slice_koff[s1] : *total_keys;
    Node nd;
    nd.prefix = 0;
    nd.start = key_base + k0;
    nd.len = k1 - k0;
    nd.meta = (u32)(2 * k);                      // no key bit is fixed; buffer 0
    nd.split = 0;
    nd.child_base = 0;
    nd.chunk_base = 0;
    out[i] = nd;
    if (check_kmers && buckets[i].child_base != k1 - k0)
        atomicAdd(n_bad, 1ull);
}

__global__ __launch_bounds__(256) void sk_expand_flat_kernel(const ull2_t *__restrict__ recs, const u32 *__restrict__ slice_rec0,
                                                             const u32 *__restrict__ slice_nrec,
                                                             const u32 *__restrict__ slice_koff, u32 key_base, u32 n_slices, int k,
                                                             u32 lmax, u64 *__restrict__ keys)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char skf_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 si = blockIdx.x * 4 + (u32)wave;
    if (si >= n_slices)
        return;                                   // (whole waves: nothing below synchronises across waves)
    u64 *stage = reinterpret_cast<u64 *>(skf_smem) + (size_t)wave * 64 * lmax;
    const ull2_t *src = recs + (u64)slice_rec0[si];
    const u32 n = slice_nrec[si];
    const u64 kmask = kmer_mask(k);
    u64 out = (u64)key_base + slice_koff[si];
    ull2_t nxt;
    nxt.x = nxt.y = 0;
    if ((u32)lane < n)
        nxt = src[lane];
    for (u32 b0 = 0; b0 < n; b0 += 64) {
        const ull2_t rec = nxt;
        const bool have = b0 + (u32)lane < n;
        if (b0 + 64 + (u32)lane < n)
            nxt = src[b0 + 64 + lane];
        const u32 len = have ? (u32)((rec.y >> 44) & 31) + 1u : 0u;
        const u32 inc = wave_incl_scan(len);
        const u32 n_keys = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        const u32 o = inc - len;
        const u64 hi = rec.y & (((u64)1 << 44) - 1);
        for (u32 j = 0; j < len; j++)
            stage[o + j] = key_mix(funnel(rec.x, hi, 2 * j) & kmask, k, kmask);   // (see key_mix: the levels split mixed keys)
        sk_wave_fence();
        for (u32 s = lane; s < n_keys; s += 64)
            __builtin_nontemporal_store(stage[s], &keys[out + s]);
        sk_wave_fence();
        out += n_keys;
    }
}

// the groups of the expansion's tree, keys[first .. *end), back from key_mix
__global__ __launch_bounds__(256) void sk_unmix_kernel(u64 *__restrict__ keys, u64 first, const unsigned long long *__restrict__ end,
                                                       int k)
{
    const u64 e = *end, mask = kmer_mask(k);
    for (u64 i = first + (u64)blockIdx.x * 256 + threadIdx.x; i < e; i += (u64)gridDim.x * 256)
        keys[i] = key_unmix(keys[i], k, mask);
}

hipError_t launch_sk_unmix(u64 *keys, u64 first, const u64 *end, u64 max_groups, int k, hipStream_t s)
{
    if (max_groups == 0)
        return hipSuccess;
    const u64 g = (max_groups + 1023) / 1024;
    hipLaunchKernelGGL(sk_unmix_kernel, dim3((u32)(g < 4096 ? g : 4096)), dim3(256), 0, s, keys, first,
                       reinterpret_cast<const unsigned long long *>(end), k);
    return hipGetLastError();
}

// heavy mid buckets leave the record path: out[j] = mids[idx[j]] with child_base = its k-mers; the list entry is emptied
__global__ __launch_bounds__(256) void sk_take_heavy_kernel(Node *__restrict__ mids, const u32 *__restrict__ idx, u32 n_heavy,
                                                            const u32 *__restrict__ kcount, Node *__restrict__ out)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_heavy)
        return;
    const u32 i = idx[j];
    Node nd = mids[i];
    nd.child_base = kcount[i];
    out[j] = nd;
    mids[i].len = 0;
}

hipError_t launch_sk_take_heavy(Node *mids, const u32 *idx, u32 n_heavy, const u32 *kcount, Node *out, hipStream_t s)
{
    if (n_heavy == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_take_heavy_kernel, dim3((n_heavy + 255) / 256), dim3(256), 0, s, mids, idx, n_heavy, kcount, out);
    return hipGetLastError();
}

hipError_t launch_sk_slice_count(const Node *buckets, u32 nb, u32 *n_slices, hipStream_t s)
{
    if (nb == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_slice_count_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, buckets, nb, n_slices);
    return hipGetLastError();
}

hipError_t launch_sk_slice_fill(const Node *buckets, u32 nb, const u32 *slice_first, u32 *slice_rec0, u32 *slice_nrec, hipStream_t s)
{
    if (nb == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_slice_fill_kernel, dim3((nb + 3) / 4), dim3(256), 0, s, buckets, nb, slice_first, slice_rec0, slice_nrec);
    return hipGetLastError();
}

hipError_t launch_sk_slice_kmers(const void *recs, const u32 *slice_rec0, const u32 *slice_nrec, u32 n_slices, u32 *slice_km,
                                 hipStream_t s)
{
    if (n_slices == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_slice_kmers_kernel, dim3((n_slices + 3) / 4), dim3(256), 0, s, reinterpret_cast<const ull2_t *>(recs),
                       slice_rec0, slice_nrec, n_slices, slice_km);
    return hipGetLastError();
}

hipError_t launch_sk_slice_nodes(const Node *buckets, u32 nb, const u32 *slice_first, const u32 *slice_koff, u32 n_slices,
                                 const u32 *total_keys, u32 key_base, int k, bool check_kmers, Node *out, u64 *n_bad, hipStream_t s)
{
    if (nb == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_slice_nodes_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, buckets, nb, slice_first, slice_koff, n_slices,
                       total_keys, key_base, k, check_kmers ? 1 : 0, out, reinterpret_cast<unsigned long long *>(n_bad));
    return hipGetLastError();
}

hipError_t launch_sk_expand_flat(const void *recs, const u32 *slice_rec0, const u32 *slice_nrec, const u32 *slice_koff, u32 key_base,
                                 u32 n_slices, int k, u64 *keys, hipStream_t s)
{
    if (n_slices == 0)
        return hipSuccess;
    u32 lmax = (u32)(54 - k + 1);
    if (lmax > 32)
        lmax = 32;
    const size_t smem = (size_t)4 * 64 * lmax * 8;
    const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void *>(sk_expand_flat_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)4 * 64 * 32 * 8));
    if (ae != hipSuccess)
        return ae;
    hipLaunchKernelGGL(sk_expand_flat_kernel, dim3((n_slices + 3) / 4), dim3(256), smem, s, reinterpret_cast<const ull2_t *>(recs),
                       slice_rec0, slice_nrec, slice_koff, key_base, n_slices, k, lmax, keys);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sk_regroup: inside every mid bucket the records are regrouped by d2 (16 final buckets), from one record buffer
// into the same range of the other: a workgroup per mid bucket counts d2 (records and k-mers), then copies tile
// by tile with the same index staging as sk_scatter1.  out_nodes[16 i + j] = final bucket j of mid bucket i:
// start / len in RECORDS, child_base = its k-mers, chunk_base = its quads (sk_count's work items).
constexpr int SKR_NT = 1024;
constexpr int SKR_ITEMS = 8;
constexpr int SKR_TILE = SKR_NT * SKR_ITEMS;
// LONG = false: the mid buckets of at most one tile (all of them on random sequence; the others are left alone);
// LONG = true: the longer ones only (launched when the host has seen one: the records per mid bucket are on the host).
// Two kernels, so that the common one carries neither the long path's code nor its registers (as one kernel it showed
// 76 bytes of scratch per lane: the records' registers spilled around the long path's loop).
template <bool LONG>
__global__ __launch_bounds__(SKR_NT, 8) void sk_regroup_kernel(const Node *__restrict__ mids, u32 n_mids,
                                                            const ull2_t *__restrict__ src_all, ull2_t *__restrict__ dst_all,
                                                            Node *__restrict__ out_nodes)
{
    __shared__ u32 rc[64][17], kc[64][17], qc[64][17];   // per-lane copies of the 16 record / k-mer / quad counters
    __shared__ u32 gpos[16], tcnt[17];
    // a bucket of one tile: half a tile of records at a time, in regrouped order (64 KB); longer buckets: the index list
    __shared__ __attribute__((aligned(16))) ull2_t stage[SKR_TILE / 2];
    unsigned short *idx = reinterpret_cast<unsigned short *>(stage);
    const u32 i = blockIdx.x;
    if (i >= n_mids)
        return;
    const int tid = threadIdx.x, lane = tid & 63;
    const Node nd = mids[i];
    if ((nd.len > (u32)SKR_TILE) != LONG)
        return;
    const ull2_t *src = src_all + (u64)nd.start;
    for (int q = tid; q < 64 * 17; q += SKR_NT) {
        (&rc[0][0])[q] = 0;
        (&kc[0][0])[q] = 0;
        (&qc[0][0])[q] = 0;
    }
    if (tid < 17)
        tcnt[tid] = 0;
    __syncthreads();
    // A mid bucket of at most one tile (the planned size is ~5,000 records): its digit words are read ONCE -- counted
    // for the output nodes and ranked for the copy in the same sweep.
    constexpr bool one_tile = !LONG;
    const u32 last = nd.len ? nd.len - 1u : 0u;
    // A bucket of one tile reads every record ONCE, 16 bytes per lane, all of a thread's loads in flight (no bounds test:
    // a slot past the end reads the last record again and drops it); the records wait in registers while they are counted
    // and ranked, and leave through LDS in regrouped order (as sk_scatter1's full tiles do).
    ull2_t rec[LONG ? 1 : SKR_ITEMS];
    u32 pos[LONG ? 1 : SKR_ITEMS];
    if constexpr (one_tile) {
      if (nd.len) {
#pragma unroll
        for (int j = 0; j < SKR_ITEMS; j++) {
            const u32 r = tid + j * SKR_NT;
            rec[j] = src[r < nd.len ? r : last];
        }
#pragma unroll
        for (int j = 0; j < SKR_ITEMS; j++) {
            const u32 r = tid + j * SKR_NT;
            pos[j] = 0;
            if (r < nd.len) {
                const u64 m = rec[j].y;
                const u32 d2 = (u32)(m >> 59) & 15u;
                const u32 len = (u32)((m >> 44) & 31) + 1u;
                atomicAdd(&rc[lane][d2], 1u);
                atomicAdd(&kc[lane][d2], len);
                atomicAdd(&qc[lane][d2], (len + 3u) >> 2);
                pos[j] = atomicAdd(&tcnt[d2], 1u);
            }
        }
      }
    } else {
        for (u32 r = tid; r < nd.len; r += SKR_NT) {
            const u64 m = reinterpret_cast<const u64 *>(src + r)[1];
            const u32 d2 = (u32)(m >> 59) & 15u;
            const u32 len = (u32)((m >> 44) & 31) + 1u;
            atomicAdd(&rc[lane][d2], 1u);
            atomicAdd(&kc[lane][d2], len);
            atomicAdd(&qc[lane][d2], (len + 3u) >> 2);   // quads of sk_count: groups of four k-mers of one record
        }
    }
    __syncthreads();
    if (tid < 48) {                                // threads 0..15: records, 16..31: k-mers, 32..47: quads of d2 = tid % 16
        const u32 (*tab)[17] = tid < 16 ? rc : (tid < 32 ? kc : qc);
        u32 sum = 0;
        for (int q = 0; q < 64; q++)
            sum += tab[q][tid & 15];
        if (tid < 16)
            rc[0][tid] = sum;
        else if (tid < 32)
            kc[0][tid & 15] = sum;
        else
            qc[0][tid & 15] = sum;
    }
    __syncthreads();
    if (tid < 16) {                                // (a thread per final bucket: thread 0 alone held sixteen nodes' worth of
        const u32 c = rc[0][tid];                  // registers while every thread's records waited -- 18 of them spilled)
        const u32 run = row16_excl_scan(c);
        gpos[tid] = nd.start + run;
        Node o;
        o.start = nd.start + run;
        o.len = c;
        o.meta = 0;
        o.split = 0;
        o.prefix = 0;
        o.child_base = kc[0][tid];
        o.chunk_base = qc[0][tid];
        out_nodes[(u64)i * 16 + tid] = o;
        if (one_tile) {
            tcnt[tid] = run;                       // the tile's digit offsets (its counts are the bucket's)
            if (tid == 15)
                tcnt[16] = run + c;
        }
    }
    __syncthreads();
    if constexpr (one_tile) {
        if (nd.len == 0)
            return;
#pragma unroll
        for (int j = 0; j < SKR_ITEMS; j++)
            pos[j] += tcnt[(u32)(rec[j].y >> 59) & 15u];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if ((u32)h * (SKR_TILE / 2) >= nd.len)
                break;
#pragma unroll
            for (int j = 0; j < SKR_ITEMS; j++) {
                const u32 r = tid + j * SKR_NT;
                if (r < nd.len && (pos[j] >> 12) == (u32)h)
                    stage[pos[j] & 4095u] = rec[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < SKR_ITEMS / 2; j++) {
                const u32 s2 = (u32)h * (SKR_TILE / 2) + tid + j * SKR_NT;
                if (s2 < nd.len)                   // (the final buckets lie back to back in the mid bucket's own range)
                    __builtin_nontemporal_store(stage[tid + j * SKR_NT], &dst_all[nd.start + s2]);
            }
            __syncthreads();
        }
        return;
    }
    for (u32 t0 = 0; t0 < nd.len; t0 += SKR_TILE) {
        const u32 n_tile = nd.len - t0 < (u32)SKR_TILE ? nd.len - t0 : (u32)SKR_TILE;
        u32 dig[SKR_ITEMS], rank[SKR_ITEMS];
#pragma unroll
        for (int j = 0; j < SKR_ITEMS; j++) {
            const u32 r = tid + j * SKR_NT;
            dig[j] = 0;
            rank[j] = 0;
            const bool in = r < n_tile;
            if (in)
                dig[j] = (u32)(reinterpret_cast<const u64 *>(src + t0 + r)[1] >> 59) & 15u;
            // one atomic per wave and digit present: a long bucket is a repeat's, its records share the m-mer and with it
            // d2 -- a lane's own atomic would be 8,192 additions to ONE address per tile
            u64 todo = __ballot(in);
            while (todo) {
                const u32 d = (u32)__builtin_amdgcn_readlane((int)dig[j], (int)__builtin_ctzll(todo));
                const u64 same = __ballot(in && dig[j] == d);
                u32 base = 0;
                if (lane == (int)__builtin_ctzll(same))
                    base = atomicAdd(&tcnt[d], (u32)__popcll(same));
                base = (u32)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(same));
                if (in && dig[j] == d)
                    rank[j] = base + (u32)__popcll(same & (((u64)1 << lane) - 1));
                todo &= ~same;
            }
        }
        __syncthreads();
        if (tid == 0) {                             // counts -> exclusive offsets, tcnt[16] = n_tile
            u32 run = 0;
            for (int j = 0; j < 16; j++) {
                const u32 c = tcnt[j];
                tcnt[j] = run;
                run += c;
            }
            tcnt[16] = run;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SKR_ITEMS; j++) {
            const u32 r = tid + j * SKR_NT;
            if (r < n_tile)
                idx[tcnt[dig[j]] + rank[j]] = (unsigned short)r;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SKR_ITEMS; j++) {
            const u32 s = tid + j * SKR_NT;
            if (s < n_tile) {
                const ull2_t r = src[t0 + idx[s]];
                const u32 d = (u32)(r.y >> 59) & 15u;
                __builtin_nontemporal_store(r, &dst_all[gpos[d] + (s - tcnt[d])]);
            }
        }
        __syncthreads();
        // (lane t reads tcnt[t + 1], which lane t + 1 is about to zero: both values into registers first, then the store)
        u32 adv = 0;
        if (tid < 16)
            adv = tcnt[tid + 1] - tcnt[tid];
        __syncthreads();
        if (tid < 16) {
            gpos[tid] += adv;
            tcnt[tid] = 0;
        }
        __syncthreads();
    }
}

int sk_regroup_tile() { return SKR_TILE; }

hipError_t launch_sk_regroup(const Node *mids, u32 n_mids, const void *src, void *dst, Node *out_nodes, bool any_long, hipStream_t s)
{
    if (n_mids == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_regroup_kernel<false>, dim3(n_mids), dim3(SKR_NT), 0, s, mids, n_mids, reinterpret_cast<const ull2_t *>(src),
                       reinterpret_cast<ull2_t *>(dst), out_nodes);
    if (any_long)
        hipLaunchKernelGGL(sk_regroup_kernel<true>, dim3(n_mids), dim3(SKR_NT), 0, s, mids, n_mids, reinterpret_cast<const ull2_t *>(src),
                           reinterpret_cast<ull2_t *>(dst), out_nodes);
    return hipGetLastError();
}

// {sum of vals[0 .. wave), sum of all n <= 16 values} from one LDS read per lane and a 16-lane DPP scan (instead of a
// loop of n LDS reads in every thread)
__device__ __forceinline__ void sk_wave_prefix16(const u32 *vals, int n, int wave, int lane, u32 &before, u32 &total)
{
    const u32 v = lane < n ? vals[lane] : 0u;
    const u32 ex = row16_excl_scan(v);             // lanes 0..15
    total = (u32)__builtin_amdgcn_readlane((int)(ex + v), 15);
    before = (u32)__builtin_amdgcn_readlane((int)ex, wave < 16 ? wave : 15);
}

// ------------------------------------------------------------------------------------------------
// sk_count: the leaves of the super-k-mer engine for the buckets that may hold COPIES (round 4: sk_count_clean below takes
// every small bucket first and leaves these).  A final bucket of at most sk_count_cap() k-mers in at most
// SKC_MAXREC records is counted straight from its records in the workgroup's LDS hash table: linear probing over
// four-byte slots claimed by a 32-bit compare-and-swap, a slot holding 19 bits of fingerprint and the 12-bit id of the
// k-mer that claimed it; a probe that meets its own fingerprint re-derives the claimant's key from the staged records and
// compares in full (a copy is counted in a 16-bit counter at the claimant's id).  The keys a thread was first to insert
// stay in its registers and leave, with their counts, one bucket later; a bucket's output range is the exclusive scan of
// the buckets' k-mer counts -- no cursor.  No key of such a bucket ever reaches HBM as a key.
//   Work split: a record's k-mers are cut into groups of SKC_KPT ("quads"); a prefix sum over the records' quad counts
//   gives every quad a thread, and every record writes its quads' owner entries itself -- no search.  Records and
//   owner table live in LDS, so that nothing between two barriers waits for global memory: the next bucket's records
//   are requested a whole bucket ahead.
//   What bounds it (PMC of three variants, round 3): the time follows the kernel's VALU + SALU instruction count
//   (9.3 G wave instructions at 3 Gbase: VALU issue 77 % busy, the CU's scalar unit 47 %), not the LDS (23 % busy).
//   Measured and dropped: the quad's four first probes issued back to back (more registers live, a spill); eight k-mers
//   per thread at 512 threads (half the waves); a binary search over the prefix instead of the owner table; branch-free
//   probe loops that keep finished lanes probing a private word (more VALU than the SALU they save: 12.7 vs 12.1 ms);
//   a per-lane `claimed` flag as a bool (it lives in scalar registers: +44 % SALU, 13.8 ms).
// Other buckets (flagged by the host's selection) are expanded to keys and counted by the ordinary levels.
constexpr int SKC_NT = 1024;                     // two workgroups = 32 waves per CU: the kernel lives on hidden latency
constexpr int SKC_SLOTS = 236 * 64;              // 15104 four-byte slots (load 0.2 at 3000 keys): with the tables below 79.0 KiB, two workgroups per CU
constexpr int SKC_MAXREC = 512;                  // records of a bucket (a bucket of 3300 k-mers of random sequence has ~370)
constexpr int SKC_KPT = 4;                       // k-mers per quad
constexpr int SKC_MAXQ = SKC_NT;                 // quads of a bucket: one per thread (the selection sends buckets with more elsewhere)
constexpr u32 SKC_FREE = ~0u;                    // an empty slot

// k-mer j of a record
__device__ __forceinline__ u64 sk_record_kmer(const ull2_t rec, u32 j, u64 kmask)
{
    return funnel(rec.x, rec.y & (((u64)1 << 44) - 1), 2 * j) & kmask;
}

__global__ __launch_bounds__(SKC_NT, 8) void sk_count_kernel(const Node *__restrict__ fin, const u32 *__restrict__ list,
                                                             const u32 *__restrict__ list_off, u32 n_list_arg,
                                                             const u32 *__restrict__ n_list_dev /* or null */,
                                                             const ull2_t *__restrict__ recs, int k,
                                                             unsigned long long *__restrict__ n_groups,
                                                             u64 *__restrict__ seg_off, u32 *__restrict__ seg_cnt,
                                                             u64 *__restrict__ out_keys, u32 *__restrict__ out_counts, int dbg)
{
    constexpr int WAVES = SKC_NT / 64;
    constexpr int RWAVES = SKC_MAXREC / 64;        // waves that hold records
    // A slot holds 19 bits of fingerprint (under a clear top bit: no word equals the all-ones of a free slot) and the
    // 12-bit id of the k-mer that claimed it (thread << 2 | position in its quad) -- four bytes instead of the key's
    // eight, so the same LDS gives twice the slots and probe chains half as long.  An insert that meets its own
    // fingerprint re-derives the claimant's key from the staged records (owner table -> record -> shift) and compares in
    // full: equal = a copy, counted at the claimant's id; different = a fingerprint collision (one comparison in 2^19),
    // which just probes on.  No key value is reserved.
    __shared__ __attribute__((aligned(16))) u32 tab[SKC_SLOTS];
    __shared__ __attribute__((aligned(16))) u32 cop2[SKC_NT * SKC_KPT / 2];    // copies per claimant id, 16-bit halves
    unsigned short *cop16 = reinterpret_cast<unsigned short *>(cop2);
    __shared__ __attribute__((aligned(16))) ull2_t lrec[SKC_MAXREC];
    __shared__ unsigned short ownq[SKC_MAXQ];      // quad -> record | first k-mer / SKC_KPT << 9
    __shared__ u32 wclaim[2][WAVES], wq[RWAVES];
    __shared__ u32 copy_seen[2];
    int tid = threadIdx.x;
    int lane = tid & 63, wave = tid >> 6;
    const u32 n_list = n_list_dev ? *n_list_dev : n_list_arg;      // (the buckets sk_count_clean left: counted on the device)
    u32 lq = blockIdx.x;
    if (lq >= n_list)
        return;
    for (int q = tid; q < SKC_SLOTS; q += SKC_NT)
        tab[q] = SKC_FREE;
    for (int q = tid; q < SKC_NT * SKC_KPT / 2; q += SKC_NT)
        cop2[q] = 0;
    if (tid < 2)
        copy_seen[tid] = 0;
    const u64 kmask = kmer_mask(k);
    u32 li = list[lq];
    u32 off = list_off[lq];                        // the bucket's output range is [off, off + its k-mers): the exclusive scan of
    Node nd = fin[li];                             // the buckets' k-mer counts, so no cursor is involved
    ull2_t myrec;                                  // record tid of the bucket (threads 0 .. SKC_MAXREC-1)
    myrec.x = myrec.y = 0;
    if ((u32)tid < nd.len)
        myrec = recs[(u64)nd.start + tid];

    // What a thread keeps of a bucket until its output range has arrived: the keys of its quad whose slots it claimed,
    // their counts, its wave's offset.
    constexpr int KEEP = SKC_KPT;
    u32 pkl[KEEP], pkh[KEEP];
    u32 pc[KEEP];
    u32 p_mask = 0, p_before = 0, p_groups = 0, p_li = 0, p_off = 0, p_kmers = 0;
    bool have_prev = false, p_copy = false;
    u64 my_groups = 0;                             // (thread 0: groups of all this workgroup's buckets)
    int par = 0;

    auto emit_prev = [&]() {
        // every wave's claimed keys go out position by position: lanes that claimed their q-th key write one
        // contiguous run per store instruction (wave-uniform base pointers, 32-bit offsets)
        const u64 obase = p_off;
        u64 *ok = out_keys + obase;
        u32 *oc = out_counts + obase;
        u32 run = p_before;
#pragma unroll
        for (int q = 0; q < KEEP; q++) {
            const bool mine = (p_mask >> q) & 1u;
            const u64 b = __ballot(mine);
            if (mine && !SK_DBG(64)) {
                const u32 o = run + __builtin_amdgcn_mbcnt_hi((u32)(b >> 32), __builtin_amdgcn_mbcnt_lo((u32)b, 0u));
                ok[o] = ((u64)pkh[q] << 32) | pkl[q];
                oc[o] = pc[q];
            }
            run += (u32)__popcll(b);
        }
        if (p_copy)                                // a bucket that held copies: the rest of its range is count-0 padding
            for (u32 i = p_groups + (u32)tid; i < p_kmers; i += SKC_NT)
                oc[i] = 0;
        if (tid == 0) {
            seg_off[p_li] = obase;
            seg_cnt[p_li] = p_groups;
        }
    };

    // Two barriers per bucket.  The bucket's records and owner table are staged (registers -> LDS) right behind the
    // previous bucket's barrier B, from records requested a bucket earlier and a quad prefix the record waves computed
    // at the end of the previous insert phase; descriptors run two buckets ahead.
    const u32 step = gridDim.x;
    u32 n_quads = 0;
    {   // the first bucket: staged here (one extra barrier)
        const u32 len = (u32)tid < nd.len ? (u32)((myrec.y >> 44) & 31) + 1u : 0u;
        const u32 nq = (len + SKC_KPT - 1) / SKC_KPT;
        u32 qinc = 0;
        if (tid < SKC_MAXREC) {
            lrec[tid] = myrec;
            qinc = wave_incl_scan(nq);
            if (lane == 63)
                wq[wave] = qinc;
        }
        __syncthreads();
        u32 qb = 0;
        sk_wave_prefix16(wq, RWAVES, wave, lane, qb, n_quads);
        if (tid < SKC_MAXREC) {
            const u32 q0 = qb + qinc - nq;
            for (u32 q = 0; q < nq; q++)
                ownq[q0 + q] = (unsigned short)((u32)tid | (q << 9));
        }
    }
    bool has_next = lq + step < n_list;
    u32 ln = has_next ? list[lq + step] : li;
    u32 off_next = has_next ? list_off[lq + step] : off;
    Node nn = fin[ln];
    myrec.x = myrec.y = 0;
    if (has_next && (u32)tid < nn.len)
        myrec = recs[(u64)nn.start + tid];         // the second bucket's records: in flight
    for (;;) {
        const u32 lq2 = lq + 2 * step;
        const bool has_next2 = lq2 < n_list;
        const u32 ln2 = has_next2 ? list[lq2] : ln;                     // (used at the end of this iteration)
        const u32 off2 = has_next2 ? list_off[lq2] : off_next;
        const Node nn2 = fin[ln2];
        // (opaque per bucket: the LDS addresses and masks derived from the thread index are then recomputed -- one or two
        // operations each -- instead of being held across the loop: 21 of the 64 VGPRs were such invariants; 54 are used
        // now, and the kernel issues fewer instructions: 11.47 -> 11.29 ms with the next item in an A/B on one box)
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        wave = tid >> 6;
        __syncthreads();                           // A2: lrec[] and ownq[] complete, every slot of the previous bucket reset
        // ---- this thread's quad.  32-bit arithmetic throughout (the kernel's time follows its instruction count: 343 vector
        // + 214 scalar instructions per wave and bucket in round 2's form): the quad's k-mers are cut
        // from a three-dword window of the record that moves on two bits per k-mer (three v_alignbit), slot and
        // fingerprint come from one 32-bit product (the slot through a full-rate 24-bit multiply), and the probe loop
        // carries the slot's byte address only -- what a claim leaves behind is assigned once, after the loop.
        u32 ckl[KEEP], ckh[KEEP];
        u32 cslot[KEEP];                           // byte address of the claimed slot
        u32 c_mask = 0;
        // (only positions whose c_mask bit is set are ever used: the others stay whatever their registers hold -- defined
        // for the compiler by an empty asm, so that no instruction initialises them: 12 moves per thread and bucket)
#pragma unroll
        for (int q = 0; q < SKC_KPT; q++) {
            asm volatile("" : "=v"(ckl[q]));
            asm volatile("" : "=v"(ckh[q]));
            asm volatile("" : "=v"(cslot[q]));
        }
        if ((u32)tid < n_quads) {
            const u32 e = ownq[tid];
            const ull2_t rec = lrec[e & 511u];
            const u32 p0 = (u32)rec.x, p1 = (u32)(rec.x >> 32), p2 = (u32)rec.y, p3h = (u32)(rec.y >> 32);
            const u32 rl = ((p3h >> 12) & 31u) + 1u;
            const u32 p3 = p3h & 0xFFFu;
            const u32 j0 = (e >> 9) * SKC_KPT;
            const bool up = j0 >= 16;                            // the quad starts in the record's second dword
            const u32 sh = (2 * j0) & 31u;
            const u32 a0 = up ? p1 : p0, a1 = up ? p2 : p1, a2 = up ? p3 : p2, a3 = up ? 0u : p3;
            u32 w0 = __builtin_amdgcn_alignbit(a1, a0, sh), w1 = __builtin_amdgcn_alignbit(a2, a1, sh),
                w2 = __builtin_amdgcn_alignbit(a3, a2, sh);
            const u32 hmask = (u32)(kmask >> 32);                // (k >= 21: the low dword is whole)
            const u32 id0 = (u32)tid << 2;
            // all four keys of the quad first, straight into the registers that keep them (the window moves on two bits per
            // k-mer whether the position exists or not: no copies where the branches of the inserts meet)
#pragma unroll
            for (int q = 0; q < SKC_KPT; q++) {
                ckl[q] = w0;
                ckh[q] = w1 & hmask;
                w0 = __builtin_amdgcn_alignbit(w1, w0, 2);
                w1 = __builtin_amdgcn_alignbit(w2, w1, 2);
                w2 >>= 2;
            }
#pragma unroll
            for (int q = 0; q < SKC_KPT; q++) {
                if (j0 + (u32)q < rl) {
                    const u32 kl = ckl[q], kh = ckh[q];
                    if (SK_DBG(32)) {
                        c_mask |= (kl & 1) ? 1u << q : 0u;
                    } else {
                        const u32 x = (kl ^ kh) * 0x9E3779B1u;
                        u32 &sa = cslot[q];                      // byte address of the probed slot: where a claim leaves it, it stays
                        sa = ((u32)__umul24(x >> 16, (u32)SKC_SLOTS) >> 16) << 2;
                        const u32 y = x ^ (x >> 15);             // (the product's low bits alone depend on the key's low bits only)
                        // 19 bits of fingerprint under a clear top bit: no word equals the all-ones of a free slot
                        const u32 word = ((y & 0x7FFFFu) << 12) | id0 | (u32)q;
                        for (;;) {
                            const u32 old = atomicCAS(reinterpret_cast<u32 *>(reinterpret_cast<char *>(tab) + sa), SKC_FREE, word);
                            if (old == SKC_FREE) {
                                c_mask |= 1u << q;
                                break;
                            }
                            if ((old ^ word) < 4096u) {
                                // the claimant's key, from the staged records (its owner entry and record do not change before B)
                                const u32 oid = old & 0xFFFu;
                                const u32 oe = ownq[oid >> 2];
                                const u64 okey = sk_record_kmer(lrec[oe & 511u], (oe >> 9) * SKC_KPT + (oid & 3u), kmask);
                                if (okey == (((u64)kh << 32) | kl)) {  // a copy: counted at the claimant's id
                                    atomicAdd(&cop2[oid >> 1], 1u << ((oid & 1u) * 16u));
                                    copy_seen[par] = 1u;
                                    break;
                                }
                            }
                            sa = sa + 4 == (u32)SKC_SLOTS * 4u ? 0u : sa + 4;
                        }
                    }
                }
            }
        }
        const u32 wc = wave_sum((u32)__popc(c_mask));
        if (lane == 0)
            wclaim[par][wave] = wc;
        // the next bucket's quad prefix (record waves; its records were requested a bucket ago)
        const u32 len_n = has_next && (u32)tid < nn.len ? (u32)((myrec.y >> 44) & 31) + 1u : 0u;
        const u32 nq_n = (len_n + SKC_KPT - 1) / SKC_KPT;
        u32 qinc_n = 0;
        if (tid < SKC_MAXREC) {
            qinc_n = wave_incl_scan(nq_n);
            if (lane == 63)
                wq[wave] = qinc_n;
        }
        __syncthreads();                           // B: all inserts done
        if (have_prev)
            emit_prev();                           // (a bucket behind: its stores overlap this bucket's tail and the next one's head)
        // ---- this bucket: counts of the claimed slots, table cleaned; emitted next round
        u32 before = 0, D = 0;
        sk_wave_prefix16(wclaim[par], WAVES, wave, lane, before, D);
        const bool any_copy = copy_seen[par] != 0; // (random sequence: no bucket has one, and the copy counters are not read)
#pragma unroll
        for (int q = 0; q < KEEP; q++) {
            pkl[q] = ckl[q];
            pkh[q] = ckh[q];
            pc[q] = 0;
            if ((c_mask >> q) & 1u) {
                pc[q] = 1u;
                *reinterpret_cast<u32 *>(reinterpret_cast<char *>(tab) + cslot[q]) = SKC_FREE;
                if (any_copy) {
                    const u32 copies = cop16[(u32)tid * SKC_KPT + (u32)q];
                    pc[q] = 1u + copies;
                    if (copies)
                        cop16[(u32)tid * SKC_KPT + (u32)q] = 0;
                }
            }
        }
        p_mask = c_mask;
        p_before = before;
        p_groups = D;
        p_li = li;
        p_off = off;
        p_kmers = nd.child_base;
        p_copy = any_copy;
        if (tid == 0) {
            my_groups += D;
            copy_seen[par ^ 1] = 0;                // (the other parity's flag: its bucket is done with it)
        }
        have_prev = true;
        par ^= 1;
        if (!has_next)
            break;
        // ---- the next bucket is staged (nobody reads lrec / ownq between B and A2), the one after it requested
        {
            u32 qb = 0;
            sk_wave_prefix16(wq, RWAVES, wave, lane, qb, n_quads);
            if (tid < SKC_MAXREC) {
                lrec[tid] = myrec;
                const u32 q0 = qb + qinc_n - nq_n;
                for (u32 q = 0; q < nq_n; q++)
                    ownq[q0 + q] = (unsigned short)((u32)tid | (q << 9));
            }
            myrec.x = myrec.y = 0;
            if (has_next2 && (u32)tid < nn2.len)
                myrec = recs[(u64)nn2.start + tid];
        }
        lq += step;
        li = ln;
        off = off_next;
        nd = nn;
        has_next = has_next2;
        ln = ln2;
        off_next = off2;
        nn = nn2;
    }
    emit_prev();                                   // the last bucket's output
    if (tid == 0)
        atomicAdd(n_groups, (unsigned long long)my_groups);
}

// Could records A and B -- both cut from plain tiles (SK_REC_MULTI clear), their minimum m-mers equal, at offsets a and b --
// hold an equal k-mer?  All k-mers of such a record have their leftmost minimum m-mer at the record's one place, and that
// offset inside a k-mer is a function of the k-mer's content: equal k-mers K = A[j1 ..] = B[j2 ..] have a - j1 = b - j2.  So
// the records can only agree where they are aligned at their m-mers, and K covers the m-mer, L' bases before it and R'
// behind it with L' + R' = k - m: an equal k-mer exists iff the bases agree over L before and R behind the m-mer (as far
// as BOTH records reach: ML, MR) with L + R >= k - m.  Exact, not a filter.
__device__ __forceinline__ bool sk_records_share_kmer(const ull2_t A, u32 a, u32 nba, const ull2_t B, u32 b, u32 nbb, u32 m, u32 kmm)
{
    const u64 pm = ((u64)1 << 34) - 1;             // payload bits of the second word (at most 49 bases)
    const u64 ah = A.y & pm, bh = B.y & pm;
    const u32 ML = a < b ? a : b;
    const u32 ra = nba - a - m, rb = nbb - b - m;
    const u32 MR = ra < rb ? ra : rb;               // <= k - m <= 17
    const u64 xr = (sk_shr128(A.x, ah, 2u * (a + m)) ^ sk_shr128(B.x, bh, 2u * (b + m))) & (((u64)1 << (2u * MR)) - 1);
    const u32 R = xr ? (u32)__builtin_ctzll(xr) >> 1 : MR;
    if (R + ML < kmm)
        return false;
    const u64 xl = (funnel(A.x, ah, 2u * (a - ML)) ^ funnel(B.x, bh, 2u * (b - ML))) & (((u64)1 << (2u * ML)) - 1);
    const u32 L = xl ? ML - 1u - ((63u - (u32)__builtin_clzll(xl)) >> 1) : ML;
    return L + R >= kmm;
}


// ------------------------------------------------------------------------------------------------
// sk_count_clean (round 4): RECORDS ARE TESTED BEFORE K-MERS.  Equal k-mers share their minimum m-mer AND its offset inside
// the k-mer (both are functions of the k-mer's content), so two records cut from plain tiles can hold an equal k-mer only
// if their m-mers are equal and they agree around them -- over L bases before and R behind it with L + R >= k - m
// (sk_records_share_kmer: exact).  A bucket's ~300 records go into a small LDS table keyed by the m-mer: an entry is
//   low word  = 22 bits of a hash of the m-mer | record index << 22 (bit 31 clear: no entry is all ones)
//   high word = the F bases before the m-mer | the F bases behind it << 16, F = min(8, (k - m) / 2); a side the record
//               does not have F bases of holds a value of the record's own (0x8000 | index) instead
// (linear probing: a record passes every earlier entry of its m-mer on the way to its own slot: ~1.4 of them at 3 Gbase,
// 0.1 at 250 Mbase).  An equal k-mer covers the m-mer and k - m >= 2 F more bases around it, so it needs the m-mer's hash
// and one of the two halves equal: two XORs per passed entry; the few pairs that pass (4^-F per side) take the exact test.
// A bucket in which no two records share a k-mer holds every k-mer ONCE: its k-mers are cut from the records and leave as
// (key, 1) groups, one k-mer per thread and round, 8 + 4 bytes per lane to consecutive addresses -- no k-mer hash, no probe,
// no table.  A bucket in which SOME records share k-mers (a stretch that occurs again elsewhere: a handful of buckets per
// million on random sequence, nearly every bucket of a real genome) keeps that path for all its other records and counts
// only the sharing records' k-mers, up to SKQ_DIRTY_MAX of them, in the record table's slots once the records are done
// with them (a k-mer of a record that shares nothing has no copy; a copy of a sharing record's k-mer lies in a record that
// shares with it).  What is left -- a SK_REC_MULTI record of a low-complexity stretch, more copies than that -- is
// appended to a list that sk_count takes afterwards, exact as before.
//   Why: round 3's sk_count spent 265 vector + 200 scalar instructions per wave and bucket, a third of them probing the
//   k-mer table for k-mers that never had a copy, the rest bookkeeping around sixteen waves (quads, claim prefixes, keys
//   kept a bucket long, four ballot rounds of stores).  This kernel has eight waves per bucket, no table to clear, and
//   26 KB of LDS: four workgroups per CU.
constexpr int SKQ_NT = 512;
static_assert(SKQ_NT == SKC_MAXREC, "a thread per record");
constexpr int SKQ_KMERS = 4096;                  // sk_count_cap(): most k-mers of a small bucket
constexpr int SKQ_VSLOTS = 1024;                 // record table (at most 512 entries)
constexpr int SKQ_DIRTY_MAX = 640;               // most k-mers of sharing records that are counted in those slots

// (A/B builds only: tools/build_variant.sh WORK x.so -DSKQ_STORE_PLAIN / -DSKQ_KEYS_PLAIN / -DSKQ_COUNTS_PLAIN)
#if defined(SKQ_STORE_PLAIN) || defined(SKQ_KEYS_PLAIN)
#define SKQ_STORE(v, p) (*(p) = (v))
#else
#define SKQ_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif
#if defined(SKQ_STORE_PLAIN) || defined(SKQ_COUNTS_PLAIN)
#define SKQ_STOREC(v, p) (*(p) = (v))
#else
#define SKQ_STOREC(v, p) __builtin_nontemporal_store(v, p)
#endif
__global__ __launch_bounds__(SKQ_NT, 8) void sk_count_clean_kernel(const Node *__restrict__ fin, const u32 *__restrict__ list,
                                                                   const u32 *__restrict__ list_off, u32 n_list,
                                                                   const ull2_t *__restrict__ recs, int k,
                                                                   unsigned long long *__restrict__ n_groups,
                                                                   u64 *__restrict__ seg_off, u32 *__restrict__ seg_cnt,
                                                                   u64 *__restrict__ out_keys, u32 *__restrict__ out_counts,
                                                                   u32 *__restrict__ left_list, u32 *__restrict__ left_off,
                                                                   u32 *__restrict__ n_left, int dbg)
{
    constexpr int WAVES = SKQ_NT / 64;
    __shared__ __attribute__((aligned(16))) ull2_t lrec[SKC_MAXREC];
    __shared__ __attribute__((aligned(16))) u64 vtab[SKQ_VSLOTS];
    __shared__ unsigned short own[SKQ_KMERS];      // k-mer of the bucket -> record | position in it << 9
    __shared__ u32 wk[WAVES], wk2[WAVES];
    __shared__ u32 shared_flag[2], multi_flag[2];  // by bucket parity: two records share a k-mer; a SK_REC_MULTI record
    __shared__ u32 dcnt2[SKQ_VSLOTS / 2];          // (buckets with copies) the counts beside vtab's k-mer keys, 16 bits each
    __shared__ unsigned short cpos[SKC_MAXREC];    // ... a record's first output slot, or 0xFFFF: its k-mers are counted
    __shared__ unsigned char rdb[SKC_MAXREC];      // ... the record shares a k-mer with another
    __shared__ u32 d_ones;
    unsigned short *dcnt = reinterpret_cast<unsigned short *>(dcnt2);
    int tid = threadIdx.x;
    u32 lq = blockIdx.x;
    if (lq >= n_list)
        return;
    vtab[tid] = ~(u64)0;
    vtab[tid + SKQ_NT] = ~(u64)0;
    dcnt2[tid] = 0;
    rdb[tid] = 0;
    if (tid < 2)
        shared_flag[tid] = multi_flag[tid] = 0;
    if (tid == 0)
        d_ones = 0;
    const u32 mlen = k >= 23 ? 15u : (k >= 21 ? 13u : 12u);   // sk_minimizer_len
    const u32 vmask = (1u << (2u * mlen)) - 1u, kmm = (u32)k - mlen;
    const u32 flank = kmm / 2u < 8u ? kmm / 2u : 8u, fmask = (1u << (2u * flank)) - 1u;
    const u32 hmask = (u32)(kmer_mask(k) >> 32);   // (k >= 21: the low dword of a key is whole)
    const u32 step = gridDim.x;
    u32 li = list[lq];
    Node nd = fin[li];
    ull2_t myrec;
    myrec.x = myrec.y = 0;
    if ((u32)tid < nd.len)
        myrec = recs[(u64)nd.start + tid];
    u64 my_groups = 0;
    int par = 0;
    for (;;) {
        const u32 off = list_off[lq];
        const bool has_next = lq + step < n_list;
        const u32 ln = has_next ? list[lq + step] : li;
        const Node nn = fin[ln];
        asm volatile("" : "+v"(tid));              // (nothing derived from the thread index is held across buckets)
        const int lane = tid & 63, wave = tid >> 6;
        // ---- this bucket's records into LDS, the next bucket's requested
        const ull2_t me = myrec;
        lrec[tid] = me;
        myrec.x = myrec.y = 0;
        if (has_next && (u32)tid < nn.len)
            myrec = recs[(u64)nn.start + tid];
        const bool have = (u32)tid < nd.len;
        const u32 yh = (u32)(me.y >> 32);
        const u32 len = have ? ((yh >> 12) & 31u) + 1u : 0u;
        const u32 kinc = wave_incl_scan(len);
        if (lane == 63)
            wk[wave] = kinc;
        __syncthreads();                           // (1) lrec / wk complete; the previous bucket's readers are done (barrier 3)
        u32 kbase = 0, n_km = 0;
        sk_wave_prefix16(wk, WAVES, wave, lane, kbase, n_km);
        // ---- every k-mer's owner; the records against each other
        u32 vslot = ~0u;
        if (have) {
            const u32 k0 = kbase + kinc - len;
            if (!SK_DBG(1024))
                for (u32 j = 0; j < len; j++)
                    own[k0 + j] = (unsigned short)((u32)tid | (j << 9));
            if (SK_DBG(256)) {
            } else if (yh & (1u << (SK_REC_MULTI_BIT - 32))) {
                multi_flag[par] = 1u;              // (nothing is known about where this record's m-mer is)
            } else {
                const u32 p0 = (u32)me.x, p1 = (u32)(me.x >> 32), p2 = (u32)me.y, p3 = yh & 3u;
                const u32 a = (yh >> (SK_POS_SHIFT - 32)) & SK_POS_MASK;
                const u32 nba = len + (u32)k - 1u;                                // bases (<= 49: payload bits 0 .. 97)
                const u32 sv = 2u * a, sl = 2u * (a - flank), sr = 2u * (a + mlen);   // bit offsets (sl: only used if a >= flank)
                const u32 v = __builtin_amdgcn_alignbit(sv >= 32u ? p2 : p1, sv >= 32u ? p1 : p0, sv) & vmask;
                u32 fl = __builtin_amdgcn_alignbit(sl >= 32u ? p2 : p1, sl >= 32u ? p1 : p0, sl) & fmask;
                const u32 r_lo = sr >= 64u ? p2 : (sr >= 32u ? p1 : p0), r_hi = sr >= 64u ? p3 : (sr >= 32u ? p2 : p1);
                u32 fr = __builtin_amdgcn_alignbit(r_hi, r_lo, sr) & fmask;
                const u32 mine_own = 0x8000u | (u32)tid;
                fl = a >= flank ? fl : mine_own;
                fr = nba - a - mlen >= flank ? fr : mine_own;
                const u32 hv = (v ^ (v >> 13)) * 0x85EBCA6Bu;                    // (not sk_fine_word: the bucket's records share its top bits)
                const u32 lo32 = (hv >> 10) | ((u32)tid << 22), hi32 = fl | (fr << 16);
                const u64 mine = (u64)lo32 | ((u64)hi32 << 32);
                u32 slot = hv >> 22;
                for (;;) {
                    const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&vtab[slot]), ~0ull, (unsigned long long)mine);
                    if (old == ~(u64)0)
                        break;
                    const u32 xl = (u32)old ^ lo32, xh = (u32)(old >> 32) ^ hi32;
                    if ((xl & 0x3FFFFFu) == 0 && ((xh & 0xFFFFu) == 0 || (xh >> 16) == 0)) {
                        // (rare: 4^-F per side and pair) the exact test
                        const u32 oid = ((u32)old >> 22) & 511u;
                        const ull2_t ot = lrec[oid];
                        const u32 oyh = (u32)(ot.y >> 32);
                        if (sk_records_share_kmer(me, a, nba, ot, (oyh >> (SK_POS_SHIFT - 32)) & SK_POS_MASK,
                                                  ((oyh >> 12) & 31u) + (u32)k, mlen, kmm)) {
                            shared_flag[par] = 1u;
                            rdb[tid] = 1;
                            rdb[oid] = 1;
                        }
                    }
                    slot = (slot + 1u) & (u32)(SKQ_VSLOTS - 1);
                }
                vslot = slot;
            }
        }
        __syncthreads();                           // (2) owners and the verdict complete
        if (vslot != ~0u)
            vtab[vslot] = ~(u64)0;                 // (the table is clean again for the next bucket)
        u64 *ok = out_keys + off;
        u32 *oc = out_counts + off;
        auto key_cut = [&](const ull2_t r, u32 e) -> u64 {
            const u32 p0 = (u32)r.x, p1 = (u32)(r.x >> 32), p2 = (u32)r.y, p3 = (u32)(r.y >> 32) & 3u;
            const u32 sh = 2u * (e >> 9);          // (<= 2 (w - 1) = 34)
            const bool up = sh >= 32u;
            const u32 a0 = up ? p1 : p0, a1 = up ? p2 : p1, a2 = up ? p3 : p2;
            const u32 kl = __builtin_amdgcn_alignbit(a1, a0, sh), kh = __builtin_amdgcn_alignbit(a2, a1, sh) & hmask;
            return ((u64)kh << 32) | kl;
        };
        auto key_of = [&](u32 i) -> u64 {
            const u32 e = own[i];
            return key_cut(lrec[e & 511u], e);
        };
        const u32 flag = shared_flag[par] | (multi_flag[par] << 1);
        if (flag) {
            // ---- some records share k-mers with others (a copy of a stretch elsewhere: the rule on real genomes, a handful
            // of buckets per million on random sequence).  Only THOSE records' k-mers can have copies: every other record's
            // k-mers leave as (key, 1) as below, compacted by a prefix over the records; the sharing records' k-mers -- up to
            // SKQ_DIRTY_MAX of them -- are counted in the record table's slots (empty again by now) and leave behind them.
            // A bucket with a SK_REC_MULTI record or with more k-mers to count goes to sk_count's list.
            const bool dirty = have && rdb[tid] != 0;
            rdb[tid] = 0;
            const u32 pk = dirty ? (len << 16) : len;
            const u32 pinc = wave_incl_scan(pk);
            if (lane == 63)
                wk2[wave] = pinc;
            __syncthreads();                       // (A) wk2; every slot of vtab is empty again
            u32 pbase = 0, ptot = 0;
            sk_wave_prefix16(wk2, WAVES, wave, lane, pbase, ptot);
            const u32 n_clean = ptot & 0xFFFFu, n_dirty = ptot >> 16;
            if ((flag & 2u) || n_dirty > (u32)SKQ_DIRTY_MAX) {
                if (tid == 0) {
                    const u32 d = atomicAdd(n_left, 1u);
                    left_list[d] = li;
                    left_off[d] = off;
                }
            } else {
                cpos[tid] = dirty ? (unsigned short)0xFFFFu : (unsigned short)((pbase + pinc - pk) & 0xFFFFu);
                __syncthreads();                   // (B) cpos
                for (u32 i = (u32)tid; i < n_km; i += SKQ_NT) {
                    const u32 e = own[i];
                    const u32 c = cpos[e & 511u];
                    const u64 key = key_of(i);
                    if (c != 0xFFFFu) {
                        SKQ_STORE(key, &ok[c + (e >> 9)]);
                        SKQ_STOREC(1u, &oc[c + (e >> 9)]);
                    } else if (key == ~(u64)0) {
                        atomicAdd(&d_ones, 1u);    // (the 32-base k-mer GG..G: the empty slot's value)
                    } else {
                        u32 slot = ((((u32)key ^ (u32)(key >> 32)) * 0x9E3779B1u) >> 22) & (u32)(SKQ_VSLOTS - 1);
                        for (;;) {
                            const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&vtab[slot]), ~0ull, (unsigned long long)key);
                            if (old == ~(u64)0 || old == key)
                                break;
                            slot = (slot + 1u) & (u32)(SKQ_VSLOTS - 1);
                        }
                        atomicAdd(&dcnt2[slot >> 1], 1u << ((slot & 1u) * 16u));
                    }
                }
                __syncthreads();                   // (C) the table complete
                const u32 c0 = dcnt[tid], c1 = dcnt[tid + SKQ_NT], ones = d_ones;
                const u64 b0 = __ballot(c0 != 0), b1 = __ballot(c1 != 0);
                if (lane == 0)
                    wk2[wave] = (u32)__popcll(b0) + (u32)__popcll(b1);
                __syncthreads();                   // (D)
                u32 before = 0, D = 0;
                sk_wave_prefix16(wk2, WAVES, wave, lane, before, D);
                const u64 lt = ((u64)1 << lane) - 1;
                u64 *dk = ok + n_clean;
                u32 *dc = oc + n_clean;
                if (c0) {
                    const u32 r = before + (u32)__popcll(b0 & lt);
                    dk[r] = vtab[tid];
                    dc[r] = c0;
                    vtab[tid] = ~(u64)0;
                    dcnt[tid] = 0;
                }
                if (c1) {
                    const u32 r = before + (u32)__popcll(b0) + (u32)__popcll(b1 & lt);
                    dk[r] = vtab[tid + SKQ_NT];
                    dc[r] = c1;
                    vtab[tid + SKQ_NT] = ~(u64)0;
                    dcnt[tid + SKQ_NT] = 0;
                }
                const u32 groups = n_clean + D + (ones ? 1u : 0u);
                for (u32 i = groups + (u32)tid; i < n_km; i += SKQ_NT)
                    oc[i] = 0;                     // (the rest of the bucket's range is padding: count 0)
                if (tid == 0) {
                    if (ones) {
                        dk[D] = ~(u64)0;
                        dc[D] = ones;
                        d_ones = 0;
                    }
                    seg_off[li] = off;
                    seg_cnt[li] = groups;
                    my_groups += groups;
                }
            }
        } else {
            // ---- every k-mer of the bucket is the only one of its kind: (key, 1) groups, k-mer i at slot off + i
            // Two k-mers per thread and round: 16 bytes of keys and 8 of counts per lane, nontemporal (a plain store stream
            // tops out near 3.3 TB/s on this chip).
            // The pairs start on a 32-slot boundary of the arrays (256 bytes of keys, 128 of counts), so that a wave's stores
            // cover whole lines; the k-mers before that boundary leave one per thread.  (Measured, 3 Gbase, 14 runs each on
            // one box: pairs from the first even slot 8.07 - 8.34 ms with runs of 10.0 - 10.8 in between -- typically a box's
            // first -- against 7.93 - 8.04 with one of 8.7, when every kernel was 5 - 8 % slower; plain stores throughout
            // 8.45 - 9.3; plain stores for the partial lines at a bucket's ends: no gain.)
            const u32 head = (32u - (off & 31u)) & 31u;
            if ((u32)tid < head && (u32)tid < n_km) {
                SKQ_STORE(key_of((u32)tid), &ok[tid]);
                SKQ_STOREC(1u, &oc[tid]);
            }
            // (both owners, then both records: two LDS round trips per pair instead of four in a row -- and the NEXT pair's
            // reads are issued before this pair's keys are cut and stored)
            u32 i = head + 2u * (u32)tid;
            u32 e0 = 0, e1 = 0;
            ull2_t r0, r1;
            r0.x = r0.y = r1.x = r1.y = 0;
            auto fetch = [&](u32 at) {
                e0 = own[at];
                e1 = own[at + 1 < n_km ? at + 1 : at];
                asm volatile("" : "+v"(e0), "+v"(e1));
                r0 = lrec[e0 & 511u];
                r1 = lrec[e1 & 511u];
                asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r1.x), "+v"(r1.y));
            };
            if (i < n_km)
                fetch(i);
            while (i < n_km) {
                const u32 ce0 = e0, ce1 = e1;
                const ull2_t c0 = r0, c1 = r1;
                const u32 nxt = i + 2u * SKQ_NT;
                if (nxt < n_km)
                    fetch(nxt);
                const u64 k0 = key_cut(c0, ce0);
                if (SK_DBG(512)) {
                    if (k0 == 0x123456789ull)
                        ok[0] = k0;
                } else if (i + 1 < n_km) {
                    ull2_t kk;
                    kk.x = k0;
                    kk.y = key_cut(c1, ce1);
                    SKQ_STORE(kk, reinterpret_cast<ull2_t *>(&ok[i]));
                    SKQ_STOREC((u64)0x100000001ull, reinterpret_cast<u64 *>(&oc[i]));
                } else {
                    SKQ_STORE(k0, &ok[i]);
                    SKQ_STOREC(1u, &oc[i]);
                }
                i = nxt;
            }
            if (tid == 0) {
                seg_off[li] = off;
                seg_cnt[li] = n_km;
                my_groups += n_km;
            }
        }
        if (tid == 0)
            shared_flag[par ^ 1] = multi_flag[par ^ 1] = 0;   // (the other parity's flags: its bucket is done with them)
        par ^= 1;
        if (!has_next)
            break;
        __syncthreads();                           // (3) lrec / own are rewritten by the next bucket
        lq += step;
        li = ln;
        nd = nn;
    }
    if (tid == 0 && my_groups)
        atomicAdd(n_groups, (unsigned long long)my_groups);
}

static int sk_dbg();

hipError_t launch_sk_count(const Node *fin, const u32 *list, const u32 *list_off, u32 n_list, const void *recs, int k, u64 *n_groups,
                           u64 *seg_off, u32 *seg_cnt, u64 *out_keys, u32 *out_counts, u32 *left /* 2 n_list + 1 words */,
                           hipStream_t s)
{
    if (n_list == 0)
        return hipSuccess;
    int dev = 0, n_cu = 256, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        n_cu = v;
    // every small bucket through the record test first; the ones that may hold copies are listed (left[0 .. n), left[n_list ..
    // n_list + n) = their output offsets, left[2 n_list] = n) and counted by the k-mer table kernel behind it
    u32 *left_list = left, *left_off = left + n_list, *n_left = left + 2 * (size_t)n_list;
    hipError_t e = hipMemsetAsync(n_left, 0, sizeof(u32), s);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(sk_count_clean_kernel, dim3(std::min<u32>(n_list, (u32)n_cu * 4u)), dim3(SKQ_NT), 0, s, fin, list, list_off, n_list,
                       reinterpret_cast<const ull2_t *>(recs), k, reinterpret_cast<unsigned long long *>(n_groups), seg_off, seg_cnt,
                       out_keys, out_counts, left_list, left_off, n_left, sk_dbg());
    const u32 grid = std::min<u32>(n_list, (u32)n_cu * 2u);
    hipLaunchKernelGGL(sk_count_kernel, dim3(grid), dim3(SKC_NT), 0, s, fin, left_list, left_off, 0u, n_left,
                       reinterpret_cast<const ull2_t *>(recs), k, reinterpret_cast<unsigned long long *>(n_groups), seg_off, seg_cnt,
                       out_keys, out_counts, sk_dbg());
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// sk_count_big: final buckets that are too long for sk_count but hold FEW DISTINCT keys -- what repeats make: a
// minimizer of a repeated stretch brings thousands to millions of copies of a handful of k-mers into one bucket.  One
// workgroup per bucket sweeps its records tile by tile (512 records) into ONE LDS table that is kept for the whole
// bucket: 8192 eight-byte key slots with 32-bit counts.  Groups leave once, by a sweep of the table, into the
// bucket's output range (same placement rule as sk_count: the scan of the buckets' k-mer counts).  A bucket whose
// distinct keys outgrow the table raises its status: the host sends it through the expansion + tree path instead.
constexpr int SKB_NT = 1024;
constexpr int SKB_SLOTS = 8192;
constexpr int SKB_REC = 512;                     // records per tile
constexpr int SKB_MAXQ = SKB_REC * 8;            // quads of a tile (a record holds at most 32 k-mers)
constexpr u32 SKB_ROUND_CAP = SKB_SLOTS - SKB_NT * SKC_KPT - 64;   // a round of inserts (<= 4096 new keys) starts below this
constexpr u64 SKB_EMPTY = ~(u64)0;
constexpr int SKB_RTAB_BITS = 10;
static_assert((1 << SKB_RTAB_BITS) == SKB_NT, "one record-table slot per thread");

// the table's groups, compacted: eight consecutive slots per thread, wave by wave; returns the number of groups (a count
// of the all-ones key, which the table cannot hold, goes last).  All threads of the workgroup call it.
__device__ __forceinline__ u32 skb_emit(const u64 *tab, const u32 *cnt, u32 n_ones, u32 *wtot, u64 *dst_keys, u32 *dst_counts)
{
    constexpr int WAVES = SKB_NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u64 myk[8];
    u32 myc[8], mine = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        myk[j] = tab[tid * 8 + j];
        myc[j] = cnt[tid * 8 + j];
        mine += myk[j] != SKB_EMPTY ? 1u : 0u;
    }
    const u32 inc = wave_incl_scan(mine);
    if (lane == 63)
        wtot[wave] = inc;
    __syncthreads();
    u32 before = 0, D = 0;
    sk_wave_prefix16(wtot, WAVES, wave, lane, before, D);
    u64 o = before + inc - mine;
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (myk[j] != SKB_EMPTY) {
            dst_keys[o] = myk[j];
            dst_counts[o] = myc[j];
            o++;
        }
    if (n_ones && tid == 0) {
        dst_keys[D] = SKB_EMPTY;
        dst_counts[D] = n_ones;
    }
    return D + (n_ones ? 1u : 0u);
}

// Work item = a SLICE of a bucket (SKB_SLICE_REC records): a bucket of one slice is counted and emitted here; a longer
// one (millions of copies of a few k-mers) is swept by several workgroups at once, each leaving the groups of its slice
// in a partial area, and sk_big_merge adds the slices' groups up.
constexpr u32 SKB_SLICE_REC = 128 * SKB_REC;
constexpr u32 SKB_FAILED = ~0u;

__global__ __launch_bounds__(256) void sk_big_slices_kernel(const Node *__restrict__ fin, const u32 *__restrict__ list, u32 n_list,
                                                            u32 *__restrict__ nsl, u32 *__restrict__ msl)
{
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_list)
        return;
    const u32 ns = (fin[list[p]].len + SKB_SLICE_REC - 1) / SKB_SLICE_REC;
    nsl[p] = ns;
    msl[p] = ns > 1 ? ns : 0u;
}

__global__ __launch_bounds__(256) void sk_big_slice_fill_kernel(const u32 *__restrict__ nsl_raw, const u32 *__restrict__ sfirst,
                                                                u32 n_list, u32 *__restrict__ sl_bucket, u32 *__restrict__ sl_idx)
{
    const u32 lane = threadIdx.x & 63;
    const u32 p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_list)
        return;
    const u32 ns = nsl_raw[p], s0 = sfirst[p];
    for (u32 j = lane; j < ns; j += 64) {
        sl_bucket[s0 + j] = p;
        sl_idx[s0 + j] = j;
    }
}

__global__ __launch_bounds__(SKB_NT) void sk_count_big_kernel(const Node *__restrict__ fin, const u32 *__restrict__ list,
                                                              const u32 *__restrict__ list_off, const u32 *__restrict__ nsl,
                                                              const u32 *__restrict__ mfirst, const u32 *__restrict__ sl_bucket,
                                                              const u32 *__restrict__ sl_idx, u32 n_slices,
                                                              const ull2_t *__restrict__ recs, int k,
                                                              unsigned long long *__restrict__ n_groups,
                                                              u64 *__restrict__ seg_off, u32 *__restrict__ seg_cnt,
                                                              u64 *__restrict__ out_keys, u32 *__restrict__ out_counts,
                                                              u32 *__restrict__ status, u64 *__restrict__ part_keys,
                                                              u32 *__restrict__ part_cnts, u32 *__restrict__ part_n)
{
    constexpr int WAVES = SKB_NT / 64, RWAVES = SKB_REC / 64;
    __shared__ __attribute__((aligned(16))) u64 tab[SKB_SLOTS];
    __shared__ u32 cnt[SKB_SLOTS];
    __shared__ __attribute__((aligned(16))) ull2_t lrec[SKB_REC];
    __shared__ unsigned short ownq[SKB_MAXQ];      // quad -> record | first k-mer / SKC_KPT << 9
    __shared__ u32 wq[RWAVES], wtot[WAVES], wnew[2][WAVES];
    __shared__ u32 mult[SKB_REC];                  // records of the tile equal to this one (counted at the first to arrive)
    __shared__ u32 rtab[SKB_NT];                   // the tile's distinct records: record indices, hashed by content
    __shared__ u32 ones;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 kmask = kmer_mask(k);
    u64 my_groups = 0;
    for (u32 sq = blockIdx.x; sq < n_slices; sq += gridDim.x) {
        const u32 lq = sl_bucket[sq], sj = sl_idx[sq], ns = nsl[lq];
        const u32 li = list[lq];
        const u64 obase = list_off[lq];
        const Node nd = fin[li];
        const u32 r_lo = sj * SKB_SLICE_REC, r_hi = nd.len - r_lo < SKB_SLICE_REC ? nd.len : r_lo + SKB_SLICE_REC;
        if (ns > 1) {
            // the padding of a sliced bucket's output range (as long as its k-mers: hundreds of MB for a bucket of a long
            // repeat) is written by all its slices, a share each, before the merge puts the groups at its start -- one
            // workgroup zeroing 800 MB alone took 13 ms
            const u64 z_lo = (u64)nd.child_base * sj / ns, z_hi = (u64)nd.child_base * (sj + 1) / ns;
            for (u64 i = z_lo + (u64)tid; i < z_hi; i += SKB_NT)
                out_counts[obase + i] = 0;
        }
        for (int q = tid; q < SKB_SLOTS; q += SKB_NT) {
            tab[q] = SKB_EMPTY;
            cnt[q] = 0;
        }
        if (tid == 0)
            ones = 0;
        __syncthreads();
        bool failed = false;
        u32 distinct = 0, rpar = 0;                // (every thread keeps the same count: no shared counter to race on)
        for (u32 t0 = r_lo; t0 < r_hi && !failed; t0 += SKB_REC) {
            // ---- the tile's records into LDS.  A repeat brings the SAME record again and again (the same stretch of
            // sequence cut at the same places): equal records of a tile are expanded once, with a multiplicity.
            u32 nq = 0, qinc = 0;
            ull2_t myr;
            myr.x = myr.y = 0;
            u32 mylen = 0;
            if (tid < SKB_REC) {
                if (t0 + (u32)tid < r_hi) {
                    myr = recs[(u64)nd.start + t0 + tid];
                    mylen = (u32)((myr.y >> 44) & 31) + 1u;
                }
                lrec[tid] = myr;
                mult[tid] = 1u;
            }
            rtab[tid] = SKC_FREE;                  // (SKB_RTAB == SKB_NT slots)
            __syncthreads();
            // equal records of the tile, wherever they stand (the copies of a repeat's records alternate when several of them
            // share a bucket): a small table of record INDICES; a record that finds an equal one ahead of it adds to
            // that one's multiplicity and is not expanded
            bool head = false;
            if (tid < SKB_REC && mylen != 0) {
                u32 hslot = (((u32)myr.x ^ (u32)(myr.x >> 32) ^ (u32)myr.y) * 0x9E3779B1u) >> (32 - SKB_RTAB_BITS);
                for (;;) {
                    const u32 old = atomicCAS(&rtab[hslot], SKC_FREE, (u32)tid);
                    if (old == SKC_FREE) {
                        head = true;
                        break;
                    }
                    const ull2_t other = lrec[old];
                    if (other.x == myr.x && other.y == myr.y) {
                        atomicAdd(&mult[old], 1u);
                        break;
                    }
                    hslot = (hslot + 1) & (u32)(SKB_NT - 1);
                }
            }
            if (tid < SKB_REC) {
                nq = head ? (mylen + SKC_KPT - 1) / SKC_KPT : 0u;
                qinc = wave_incl_scan(nq);
                if (lane == 63)
                    wq[wave] = qinc;
            }
            __syncthreads();
            u32 n_quads = 0, qb = 0;
            sk_wave_prefix16(wq, RWAVES, wave, lane, qb, n_quads);
            if (tid < SKB_REC) {
                const u32 q0 = qb + qinc - nq;
                for (u32 q = 0; q < nq; q++)
                    ownq[q0 + q] = (unsigned short)((u32)tid | (q << 9));
            }
            __syncthreads();
            // ---- rounds of one quad per thread
            for (u32 r0 = 0; r0 < n_quads; r0 += SKB_NT) {
                if (distinct > SKB_ROUND_CAP) {    // the table could fill up inside this round
                    failed = true;
                    break;
                }
                u32 claimed = 0;
                const u32 qi = r0 + (u32)tid;
                if (qi < n_quads) {
                    const u32 e = ownq[qi];
                    const ull2_t rec = lrec[e & 511u];
                    const u32 m = mult[e & 511u];
                    const u32 rl = (u32)((rec.y >> 44) & 31) + 1u;
                    const u32 j0 = (e >> 9) * SKC_KPT;
                    const u64 hi44 = rec.y & (((u64)1 << 44) - 1);
                    u64 slo = funnel(rec.x, hi44, 2 * j0), shi = hi44 >> (2 * j0);
#pragma unroll
                    for (int q = 0; q < SKC_KPT; q++) {
                        if (j0 + (u32)q < rl) {
                            const u64 kv = slo & kmask;
                            slo = (slo >> 2) | (shi << 62);
                            shi >>= 2;
                            if (kv == SKB_EMPTY) {
                                atomicAdd(&ones, m);
                            } else {
                                u32 slot = ((((u32)kv ^ (u32)(kv >> 32)) * 0x9E3779B1u) >> 19) & (u32)(SKB_SLOTS - 1);
                                for (;;) {
                                    const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&tab[slot]),
                                                              (unsigned long long)SKB_EMPTY, (unsigned long long)kv);
                                    if (old == SKB_EMPTY || old == kv) {
                                        atomicAdd(&cnt[slot], m);
                                        claimed += old == SKB_EMPTY ? 1u : 0u;
                                        break;
                                    }
                                    slot = (slot + 1) & (u32)(SKB_SLOTS - 1);
                                }
                            }
                        }
                    }
                }
                const u32 wc = wave_sum(claimed);
                if (lane == 0)
                    wnew[rpar][wave] = wc;
                __syncthreads();
                u32 nb = 0, nt = 0;
                sk_wave_prefix16(wnew[rpar], WAVES, wave, lane, nb, nt);
                distinct += nt;
                rpar ^= 1;                         // (the next round writes the other row: slow readers of this one are safe)
            }
            __syncthreads();                       // lrec / ownq are rewritten by the next tile
        }
        if (ns == 1) {
            // ---- the whole bucket was this slice: its groups go to its output range
            if (!failed) {
                const u32 groups = skb_emit(tab, cnt, ones, wtot, out_keys + obase, out_counts + obase);
                for (u32 i = groups + (u32)tid; i < nd.child_base; i += SKB_NT)
                    out_counts[obase + i] = 0;     // padding: the range is as long as the bucket's k-mers
                if (tid == 0) {
                    seg_off[li] = obase;
                    seg_cnt[li] = groups;
                    status[lq] = 0;
                    my_groups += groups;
                }
            } else {
                for (u32 i = tid; i < nd.child_base; i += SKB_NT)
                    out_counts[obase + i] = 0;     // the whole range is padding: the bucket is counted elsewhere
                if (tid == 0) {
                    seg_off[li] = obase;
                    seg_cnt[li] = 0;
                    status[lq] = 1;
                }
            }
        } else {
            // ---- one slice of several: its groups go to its partial area
            const u64 pa = (u64)(mfirst[lq] + sj);
            u32 groups = SKB_FAILED;
            if (!failed)
                groups = skb_emit(tab, cnt, ones, wtot, part_keys + pa * SKB_SLOTS, part_cnts + pa * SKB_SLOTS);
            if (tid == 0)
                part_n[pa] = groups;
        }
        __syncthreads();                           // (tab / cnt / counters are reset by the next slice)
    }
    if (tid == 0 && my_groups)
        atomicAdd(n_groups, (unsigned long long)my_groups);
}

// the groups of a sliced bucket: the slices' partial groups added up in one table (one workgroup per bucket)
__global__ __launch_bounds__(SKB_NT) void sk_big_merge_kernel(const Node *__restrict__ fin, const u32 *__restrict__ list,
                                                              const u32 *__restrict__ list_off, const u32 *__restrict__ nsl,
                                                              const u32 *__restrict__ mfirst, u32 n_list,
                                                              unsigned long long *__restrict__ n_groups,
                                                              u64 *__restrict__ seg_off, u32 *__restrict__ seg_cnt,
                                                              u64 *__restrict__ out_keys, u32 *__restrict__ out_counts,
                                                              u32 *__restrict__ status, const u64 *__restrict__ part_keys,
                                                              const u32 *__restrict__ part_cnts, const u32 *__restrict__ part_n)
{
    constexpr int WAVES = SKB_NT / 64;
    __shared__ __attribute__((aligned(16))) u64 tab[SKB_SLOTS];
    __shared__ u32 cnt[SKB_SLOTS];
    __shared__ u32 wtot[WAVES], wnew[2][WAVES];
    __shared__ u32 ones;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 lq = blockIdx.x;
    if (lq >= n_list)
        return;
    const u32 ns = nsl[lq];
    if (ns <= 1)
        return;
    const u32 li = list[lq];
    const u64 obase = list_off[lq];
    for (int q = tid; q < SKB_SLOTS; q += SKB_NT) {
        tab[q] = SKB_EMPTY;
        cnt[q] = 0;
    }
    if (tid == 0)
        ones = 0;
    __syncthreads();
    bool failed = false;
    u32 distinct = 0, rpar = 0;
    for (u32 j = 0; j < ns && !failed; j++) {
        const u64 pa = (u64)(mfirst[lq] + j);
        const u32 n_e = part_n[pa];
        if (n_e == SKB_FAILED) {
            failed = true;
            break;
        }
        for (u32 r0 = 0; r0 < n_e; r0 += SKB_NT) {
            if (distinct > (u32)(SKB_SLOTS - SKB_NT - 64)) {
                failed = true;
                break;
            }
            u32 claimed = 0;
            const u32 e = r0 + (u32)tid;
            if (e < n_e) {
                const u64 kv = part_keys[pa * SKB_SLOTS + e];
                const u32 m = part_cnts[pa * SKB_SLOTS + e];
                if (kv == SKB_EMPTY) {
                    atomicAdd(&ones, m);
                } else {
                    u32 slot = ((((u32)kv ^ (u32)(kv >> 32)) * 0x9E3779B1u) >> 19) & (u32)(SKB_SLOTS - 1);
                    for (;;) {
                        const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&tab[slot]), (unsigned long long)SKB_EMPTY,
                                                  (unsigned long long)kv);
                        if (old == SKB_EMPTY || old == kv) {
                            atomicAdd(&cnt[slot], m);
                            claimed = old == SKB_EMPTY ? 1u : 0u;
                            break;
                        }
                        slot = (slot + 1) & (u32)(SKB_SLOTS - 1);
                    }
                }
            }
            const u32 wc = wave_sum(claimed);
            if (lane == 0)
                wnew[rpar][wave] = wc;
            __syncthreads();
            u32 nb = 0, nt = 0;
            sk_wave_prefix16(wnew[rpar], WAVES, wave, lane, nb, nt);
            distinct += nt;
            rpar ^= 1;
        }
    }
    __syncthreads();
    if (!failed) {
        const u32 groups = skb_emit(tab, cnt, ones, wtot, out_keys + obase, out_counts + obase);   // (the rest of the range: zeroed by the slices)
        if (tid == 0) {
            seg_off[li] = obase;
            seg_cnt[li] = groups;
            status[lq] = 0;
            atomicAdd(n_groups, (unsigned long long)groups);
        }
    } else {
        if (tid == 0) {
            seg_off[li] = obase;
            seg_cnt[li] = 0;
            status[lq] = 1;
        }
    }
}

u32 sk_big_slice_records() { return SKB_SLICE_REC; }
u64 sk_big_partial_slots() { return SKB_SLOTS; }

// lens[i] = nodes[i].len (what the host needs of a node list: 4 of its 32 bytes)
__global__ __launch_bounds__(256) void sk_node_lens_kernel(const Node *__restrict__ nodes, u32 n, u32 *__restrict__ lens)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        lens[i] = nodes[i].len;
}

hipError_t launch_sk_node_lens(const Node *nodes, u32 n, u32 *lens, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_node_lens_kernel, dim3((n + 255) / 256), dim3(256), 0, s, nodes, n, lens);
    return hipGetLastError();
}

hipError_t launch_sk_big_slices(const Node *fin, const u32 *list, u32 n_list, u32 *nsl, u32 *msl, hipStream_t s)
{
    if (n_list == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_big_slices_kernel, dim3((n_list + 255) / 256), dim3(256), 0, s, fin, list, n_list, nsl, msl);
    return hipGetLastError();
}

hipError_t launch_sk_big_slice_fill(const u32 *nsl_raw, const u32 *sfirst, u32 n_list, u32 *sl_bucket, u32 *sl_idx, hipStream_t s)
{
    if (n_list == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_big_slice_fill_kernel, dim3((n_list + 3) / 4), dim3(256), 0, s, nsl_raw, sfirst, n_list, sl_bucket, sl_idx);
    return hipGetLastError();
}

hipError_t launch_sk_count_big(const Node *fin, const u32 *list, const u32 *list_off, const u32 *nsl, const u32 *mfirst,
                               const u32 *sl_bucket, const u32 *sl_idx, u32 n_slices, u32 n_list, const void *recs, int k,
                               u64 *n_groups, u64 *seg_off, u32 *seg_cnt, u64 *out_keys, u32 *out_counts, u32 *status, u64 *part_keys,
                               u32 *part_cnts, u32 *part_n, bool any_sliced, hipStream_t s)
{
    if (n_slices == 0)
        return hipSuccess;
    int dev = 0, n_cu = 256, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        n_cu = v;
    const u32 grid = std::min<u32>(n_slices, (u32)n_cu);
    hipLaunchKernelGGL(sk_count_big_kernel, dim3(grid), dim3(SKB_NT), 0, s, fin, list, list_off, nsl, mfirst, sl_bucket, sl_idx,
                       n_slices, reinterpret_cast<const ull2_t *>(recs), k, reinterpret_cast<unsigned long long *>(n_groups), seg_off,
                       seg_cnt, out_keys, out_counts, status, part_keys, part_cnts, part_n);
    if (any_sliced)
        hipLaunchKernelGGL(sk_big_merge_kernel, dim3(n_list), dim3(SKB_NT), 0, s, fin, list, list_off, nsl, mfirst, n_list,
                           reinterpret_cast<unsigned long long *>(n_groups), seg_off, seg_cnt, out_keys, out_counts, status,
                           part_keys, part_cnts, part_n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// selection of the final buckets.  class 1 "small": sk_count takes it (k-mers, records and quads within its stage); class 2
// "big": more than twice sk_count's k-mers and up to big_limit, tried by sk_count_big; class 3 "over": expanded to keys
// for the ordinary levels.
__device__ __forceinline__ u32 sk_bucket_class(const Node &nd, u32 cap, u32 big_limit)
{
    const u32 km = nd.child_base;
    if (km == 0)
        return 0u;
    if (km <= cap && nd.len <= (u32)SKC_MAXREC && nd.chunk_base <= (u32)SKC_MAXQ)
        return 1u;
    // (buckets just over sk_count's stage -- the tail of the size distribution of non-repetitive sequence, mostly distinct
    // keys -- are cheaper through the expansion: 0.55 against 0.8 ms at 3 Gbase uniform)
    return km > 2 * cap && km <= big_limit ? 2u : 3u;
}

// flags (to be scanned in place) and the k-mers that take output slots (small and big buckets: k_range) / key slots (over)
__global__ __launch_bounds__(256) void sk_select_flags_kernel(const Node *__restrict__ fin, u32 n_fin, u32 cap, u32 big_limit,
                                                              u32 *__restrict__ f_small, u32 *__restrict__ f_big,
                                                              u32 *__restrict__ k_range, u32 *__restrict__ k_big)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_fin)
        return;
    const u32 c = sk_bucket_class(fin[i], cap, big_limit);
    f_small[i] = c == 1u ? 1u : 0u;
    f_big[i] = c == 2u ? 1u : 0u;
    k_range[i] = c == 1u || c == 2u ? fin[i].child_base : 0u;
    k_big[i] = c == 2u ? fin[i].child_base : 0u;
}

__global__ __launch_bounds__(256) void sk_select_lists_kernel(const Node *__restrict__ fin, u32 n_fin, u32 cap, u32 big_limit,
                                                              const u32 *__restrict__ p_small, const u32 *__restrict__ p_big,
                                                              const u32 *__restrict__ kb_range, u32 *__restrict__ list_small,
                                                              u32 *__restrict__ off_small, u32 *__restrict__ list_big,
                                                              u32 *__restrict__ off_big)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_fin)
        return;
    const u32 c = sk_bucket_class(fin[i], cap, big_limit);
    if (c == 1u) {
        list_small[p_small[i]] = i;
        off_small[p_small[i]] = kb_range[i];
    } else if (c == 2u) {
        list_big[p_big[i]] = i;
        off_big[p_big[i]] = kb_range[i];
    }
}

// the buckets that go through the expansion: class 3, and the big ones sk_count_big gave up on
__global__ __launch_bounds__(256) void sk_over_flags_kernel(const Node *__restrict__ fin, u32 n_fin, u32 cap, u32 big_limit,
                                                            const u32 *__restrict__ p_big, const u32 *__restrict__ big_status,
                                                            u32 *__restrict__ f_over, u32 *__restrict__ k_over)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_fin)
        return;
    const u32 c = sk_bucket_class(fin[i], cap, big_limit);
    const bool over = c == 3u || (c == 2u && big_status[p_big[i]] != 0);
    f_over[i] = over ? 1u : 0u;
    k_over[i] = over ? fin[i].child_base : 0u;
}

__global__ __launch_bounds__(256) void sk_over_list_kernel(const Node *__restrict__ fin, u32 n_fin, const u32 *__restrict__ f_raw,
                                                           const u32 *__restrict__ p_over, const u32 *__restrict__ kb_over,
                                                           Node *__restrict__ over_nodes, u32 *__restrict__ over_kbase)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_fin || !f_raw[i])
        return;
    over_nodes[p_over[i]] = fin[i];
    over_kbase[p_over[i]] = kb_over[i];
}

hipError_t launch_sk_select_flags(const Node *fin, u32 n_fin, u32 cap, u32 big_limit, u32 *f_small, u32 *f_big, u32 *k_range,
                                  u32 *k_big, hipStream_t s)
{
    if (n_fin == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_select_flags_kernel, dim3((n_fin + 255) / 256), dim3(256), 0, s, fin, n_fin, cap, big_limit, f_small, f_big,
                       k_range, k_big);
    return hipGetLastError();
}

hipError_t launch_sk_select_lists(const Node *fin, u32 n_fin, u32 cap, u32 big_limit, const u32 *p_small, const u32 *p_big,
                                  const u32 *kb_range, u32 *list_small, u32 *off_small, u32 *list_big, u32 *off_big, hipStream_t s)
{
    if (n_fin == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_select_lists_kernel, dim3((n_fin + 255) / 256), dim3(256), 0, s, fin, n_fin, cap, big_limit, p_small, p_big,
                       kb_range, list_small, off_small, list_big, off_big);
    return hipGetLastError();
}

hipError_t launch_sk_over_flags(const Node *fin, u32 n_fin, u32 cap, u32 big_limit, const u32 *p_big, const u32 *big_status,
                                u32 *f_over, u32 *k_over, hipStream_t s)
{
    if (n_fin == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_over_flags_kernel, dim3((n_fin + 255) / 256), dim3(256), 0, s, fin, n_fin, cap, big_limit, p_big, big_status,
                       f_over, k_over);
    return hipGetLastError();
}

hipError_t launch_sk_over_list(const Node *fin, u32 n_fin, const u32 *f_raw, const u32 *p_over, const u32 *kb_over, Node *over_nodes,
                               u32 *over_kbase, hipStream_t s)
{
    if (n_fin == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_over_list_kernel, dim3((n_fin + 255) / 256), dim3(256), 0, s, fin, n_fin, f_raw, p_over, kb_over, over_nodes,
                       over_kbase);
    return hipGetLastError();
}

// timing ablations (results invalid): diagnostic build (make STAMPS=1) only
static int sk_dbg()
{
#ifdef DNAGPU_STAMPS
    const char *e = getenv("DNAGPU_DEBUG_SK");
    return e ? atoi(e) : 0;
#else
    return 0;
#endif
}

// ------------------------------------------------------------------------------------------------
// launchers
template <int W, bool BATCH>
static void launch_front(bool scatter, u32 n_chunks, hipStream_t s, const Chunk *chunks, const u64 *words, u64 n_words,
                         u64 first, int k, u32 lmax, u32 mmask, u32 c0n, u32 b1mask, u32 r0n, u32 *hist, const u32 *tot, void *recs,
                         u32 *aux, const u32 *marks, u64 n_mark_words)
{
    if (scatter)
        hipLaunchKernelGGL((sk_scatter0_kernel<W, BATCH>), dim3(n_chunks), dim3(SK_NT), 0, s, chunks, n_chunks, words, n_words, first,
                           k, lmax, mmask, c0n, b1mask, r0n, hist, tot, reinterpret_cast<ull2_t *>(recs), sk_dbg(), aux, marks,
                           n_mark_words);
    else
        hipLaunchKernelGGL((sk_hist0_kernel<W, BATCH>), dim3(n_chunks), dim3(SK_NT), 0, s, chunks, n_chunks, words, n_words, first,
                           lmax, mmask, c0n, b1mask, r0n, hist, aux, marks, n_mark_words);
}

// (k = 20, round 4: 12 bases, a window of 9 -- 4^12 / 5 ~ 3 M effective minimizer values: even final buckets up to ~2^28 rows,
// which is where the engine is the default for it; longer sequences of 20-mers stay on the tree)
// The minimizer's length m: 15 for k >= 23 (windows of 9 .. 18 m-mers), 13 for k = 21 and 22 (windows of 9 and 10: runs of
// 5 - 5.5 k-mers, 3 bytes per k-mer; 4^13 / 9 ~ 7 M effective minimizer values keep the final buckets even up to 2^32
// rows).  A function of k alone, so every rank of a sharded count cuts the same records.
int sk_min_k() { return 20; }
int sk_minimizer_len(int k) { return k >= 23 ? 15 : (k >= 21 ? 13 : 12); }

hipError_t launch_sk_level0(bool scatter, const Chunk *chunks, u32 n_chunks, const u64 *words, u64 n_words, u64 first, int k,
                            u32 c0n, u32 b1bits, u32 r0bits, u32 *hist, const u32 *tot, void *recs, hipStream_t s, u32 *aux,
                            const u32 *marks, u64 n_mark_words)
{
    if (n_chunks == 0)
        return hipSuccess;
    const int m = sk_minimizer_len(k);
    const int w = k - m + 1;
    const u32 mmask = (1u << (2 * m)) - 1u;
    u32 lmax = (u32)(54 - k + 1);                  // a record holds 54 bases
    if (lmax > 32)
        lmax = 32;
    const u32 b1mask = (1u << b1bits) - 1, r0n = 1u << r0bits;
    if (r0n > (u32)SK_MAX_C0)
        return hipErrorInvalidValue;
#define SK_CASE(W_) case W_: \
        if (marks) launch_front<W_, true>(scatter, n_chunks, s, chunks, words, n_words, first, k, lmax, mmask, c0n, b1mask, r0n, hist, tot, recs, aux, marks, n_mark_words); \
        else launch_front<W_, false>(scatter, n_chunks, s, chunks, words, n_words, first, k, lmax, mmask, c0n, b1mask, r0n, hist, tot, recs, aux, nullptr, 0); \
        break;
    switch (w) {
        SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16) SK_CASE(17) SK_CASE(18)
    default: return hipErrorInvalidValue;
    }
#undef SK_CASE
    return hipGetLastError();
}

hipError_t launch_sk_sample_chunks(Chunk *chunks, u32 n_chunks, u32 stride, u32 len, u32 n, hipStream_t s)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_sample_chunks_kernel, dim3((n_chunks + 255) / 256), dim3(256), 0, s, chunks, n_chunks, stride, len, n);
    return hipGetLastError();
}

int sk_slab_words() { return 18 * SK_MAX_C0 + 4; }

hipError_t launch_sk_slab_init(const u32 *est, u32 r0bits, u32 chunk_rows, u32 sampled, u32 n_chunks, u32 *slab, Node *nodes,
                               hipStream_t s)
{
    hipLaunchKernelGGL(sk_slab_init_kernel, dim3(1), dim3(SK_MAX_C0), 0, s, est, 1u << r0bits, chunk_rows, sampled, n_chunks,
                       (u32)(32 - r0bits), slab, nodes);
    return hipGetLastError();
}

hipError_t launch_sk_hist1(const Node *nodes, const Chunk *chunks, u32 n_chunks, const void *recs, u32 *hist, u32 *kcount,
                           hipStream_t s, bool by_d2)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_hist1_kernel, dim3(n_chunks), dim3(SK1_NT), 0, s, nodes, chunks, n_chunks,
                       reinterpret_cast<const ull2_t *>(recs), hist, kcount, by_d2 ? 59 : 49);
    return hipGetLastError();
}

hipError_t launch_sk_scatter1(const Node *nodes, const Chunk *chunks, u32 n_chunks, const void *src, void *dst, const u32 *hist,
                              const u32 *tot, hipStream_t s, bool by_d2, u32 *gcur)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_scatter1_kernel<false>, dim3(n_chunks), dim3(SK1_NT), 0, s, nodes, chunks, n_chunks,
                       reinterpret_cast<const ull2_t *>(src), reinterpret_cast<ull2_t *>(dst), hist, tot, by_d2 ? 59 : 49, gcur,
                       nullptr, nullptr, nullptr, nullptr, nullptr);
    return hipGetLastError();
}

// level 1 without its histogram: regions (spec: 2 words per node; out: 3 words -- slots of all regions, 1 if past 2^32, the
// sweep's overflow flag), cursors, the sweep (k-mers per mid bucket into kcount), the mid nodes
// (rstart != null: the regions are per mid bucket -- launch_sk_sampled_regions -- and spec / out are only cleared)
hipError_t launch_sk_spec_regions(const Node *nodes, u32 n_nodes, u32 *spec, u32 *out, u32 *gcur, hipStream_t s, const u32 *rstart)
{
    if (n_nodes == 0 || n_nodes > (u32)SK_MAX_C0)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(sk_spec_caps_kernel, dim3(1), dim3(SK_MAX_C0), 0, s, nodes, n_nodes, spec, out);
    hipLaunchKernelGGL(sk_spec_init_kernel, dim3(n_nodes), dim3(ROW_STRIDE), 0, s, nodes, n_nodes, spec, gcur, rstart);
    return hipGetLastError();
}

// The regions of a speculative level 1 from a SAMPLED histogram (uneven coarse buckets: repeats): est[] (zeroed here) <- the
// records of the sample pieces per mid bucket, rcap[] <- slots per mid bucket, rstart[] <- their exclusive scan, *total <-
// slots of all (saturating).  scan_tmp: scan_tmp_words(n_mid) words.
hipError_t launch_sk_sampled_regions(const Node *nodes, const Chunk *samples, u32 n_samples, const void *recs, u32 n_mid, u32 *est,
                                     u32 *rcap, u32 *rstart, u32 *scan_tmp, u32 *total, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(est, 0, (size_t)n_mid * sizeof(u32), s);
    if (e != hipSuccess)
        return e;
    if (n_samples)
        hipLaunchKernelGGL(sk_sample1_kernel, dim3(n_samples), dim3(SK1_NT), 0, s, nodes, samples, n_samples,
                           reinterpret_cast<const ull2_t *>(recs), est);
    hipLaunchKernelGGL(sk_sampled_caps_kernel, dim3((n_mid + 255) / 256), dim3(256), 0, s, est, n_mid, rcap);
    e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    return launch_scan_u32(rcap, rstart, n_mid, scan_tmp, total, s);
}

hipError_t launch_sk_scatter1_spec(const Node *nodes, const Chunk *chunks, u32 n_chunks, const void *src, void *dst, u32 *gcur,
                                   const u32 *spec, u32 *kcount, u32 *over, hipStream_t s, const u32 *rstart, const u32 *rcap)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_scatter1_kernel<true>, dim3(n_chunks), dim3(SK1_NT), 0, s, nodes, chunks, n_chunks,
                       reinterpret_cast<const ull2_t *>(src), reinterpret_cast<ull2_t *>(dst), nullptr, nullptr, 49, gcur, spec,
                       kcount, over, rstart, rcap);
    return hipGetLastError();
}

hipError_t launch_sk_spec_nodes(const Node *nodes, u32 n_nodes, const u32 *spec, const u32 *gcur, Node *next, u32 *over,
                                hipStream_t s, const u32 *rstart, const u32 *rcap)
{
    if (n_nodes == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_spec_nodes_kernel, dim3(n_nodes), dim3(ROW_STRIDE), 0, s, nodes, n_nodes, spec, gcur, next, over, rstart,
                       rcap);
    return hipGetLastError();
}

// the 16 children of the heavy mid buckets (split by d2 with the chunked level kernels) as final-bucket nodes: start / len
// in records, child_base = k-mers; their quads are not counted (chunk_base = ~0: never sk_count's)
__global__ __launch_bounds__(256) void sk_heavy_finals_kernel(const Node *__restrict__ kids, u32 n, const u32 *__restrict__ kcount,
                                                              Node *__restrict__ out)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    Node o;
    o.start = kids[i].start;
    o.len = kids[i].len;
    o.meta = 0;
    o.split = 0;
    o.prefix = 0;
    o.child_base = kids[i].len ? kcount[i] : 0u;
    o.chunk_base = ~0u;
    out[i] = o;
}

hipError_t launch_sk_heavy_finals(const Node *kids, u32 n, const u32 *kcount, Node *out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(sk_heavy_finals_kernel, dim3((n + 255) / 256), dim3(256), 0, s, kids, n, kcount, out);
    return hipGetLastError();
}


}  // namespace dnagpu
