// kmer_device.hpp -- device-side bit arithmetic shared by every kernel (gfx950 only).
//
// Formats are the reference's: packed dna = uint64 words, base i at bits (2i mod 64) of word i/32
// (dna.c:114-128); kmer key = uint64 with base i at bits 2i..2i+1 (dna.c:397-420).  The k-mer at
// position p is therefore bits [2p, 2p+2k) of the word stream: no per-base loop is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dnagpu {

typedef uint32_t u32;
typedef uint64_t u64;

__host__ __device__ __forceinline__ u64 kmer_mask(int k)
{
    return k >= 32 ? ~(u64)0 : (((u64)1 << (2 * k)) - 1);
}

// bits [sh, sh+64) of the 128-bit value hi:lo, sh in 0..62
__device__ __forceinline__ u64 funnel(u64 lo, u64 hi, unsigned sh)
{
    return (lo >> sh) | ((hi << 1) << (63u - sh));
}

// A bijection of the 2k-bit keys and its inverse (k in 20..32): the keys of one minimizer's bucket are near-copies of each
// other (the same m-mer at the same offset, a repeat family's consensus around it), and a most-significant-digit split of
// such keys separates nothing for levels on end.  The super-k-mer engine's expansion therefore hands the levels
// key_mix(key) -- equal keys stay equal, everything else is spread evenly over the 2k bits -- and the groups are turned
// back (key_unmix) once the leaves have written them: an unordered histogram has no order to keep.
// (x ^= x >> k is its own inverse on 2k bits; the multipliers are odd, so they have inverses modulo any power of two)
__host__ __device__ __forceinline__ u64 key_mix(u64 x, int k, u64 mask)
{
    x ^= x >> k;
    x = (x * 0x9E3779B97F4A7C15ull) & mask;
    x ^= x >> k;
    x = (x * 0xD6E8FEB86659FD93ull) & mask;
    x ^= x >> k;
    return x;
}
__host__ __device__ __forceinline__ u64 key_unmix(u64 x, int k, u64 mask)
{
    x ^= x >> k;
    x = (x * 0xCFEE444D8B59A89Bull) & mask;
    x ^= x >> k;
    x = (x * 0xF1DE83E19937733Dull) & mask;
    x ^= x >> k;
    return x;
}

// key of the k-mer starting at base `pos` (masked); words[n_words] is never read
__device__ __forceinline__ u64 key_at(const u64 *__restrict__ words, u64 n_words, u64 pos, u64 mask)
{
    u64 w = pos >> 5;
    unsigned sh = (unsigned)(pos & 31) * 2;
    u64 lo = words[w];
    u64 hi = (w + 1 < n_words) ? words[w + 1] : 0;
    return funnel(lo, hi, sh) & mask;
}

__host__ __device__ __forceinline__ u64 splitmix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// kmer_hash (dna.c:722-735) = PostgreSQL hash_any over the 8 key bytes: lookup3 final() with
// a = b = c = 0x9e3779b9 + 8 + 3923095, b += high word, a += low word.
__host__ __device__ __forceinline__ u32 rot32(u32 x, int k) { return (x << k) | (x >> (32 - k)); }
__host__ __device__ __forceinline__ u32 pg_kmer_hash(u64 bits)
{
    u32 a, b, c;
    a = b = c = 0x9e3779b9u + 8u + 3923095u;
    b += (u32)(bits >> 32);
    a += (u32)bits;
    c ^= b; c -= rot32(b, 14);
    a ^= c; a -= rot32(c, 11);
    b ^= a; b -= rot32(a, 25);
    c ^= b; c -= rot32(b, 16);
    a ^= c; a -= rot32(c, 4);
    b ^= a; b -= rot32(a, 14);
    c ^= b; c -= rot32(b, 24);
    return c;
}

// order-independent digest of one (key, count) group; mirrored by the oracle's orc_pair_mix
__host__ __device__ __forceinline__ u64 pair_mix(u64 key, u64 count)
{
    u64 x = splitmix64(key ^ 0x6a09e667f3bcc909ull);
    return x * (2 * count + 1) + splitmix64(count);
}

// ---- WHERE operators as branch-free bit tests -----------------------------------------------
// One descriptor covers `=`, `^@` and `@>`:
//   masked compare  (key & and_mask) == eq_value          kmer_eq / starts_with (dna.c:655-668, 862-863)
//   IUPAC planes    deny[c] has bit 2i set when code c is NOT in the pattern's set at position i
//                   (nucleotide_matches, dna.c:1064-1086; 'U' denies all four codes)
struct FilterDev {
    u64 and_mask;
    u64 eq_value;
    u64 deny[4];   // indexed by 2-bit code: A=0 T=1 C=2 G=3
    int use_planes;
};

__device__ __forceinline__ bool filter_match(const FilterDev &f, u64 key)
{
    if (!f.use_planes)
        return (key & f.and_mask) == f.eq_value;
    const u64 EVEN = 0x5555555555555555ull;
    u64 lo = key & EVEN, hi = (key >> 1) & EVEN;
    u64 isT = lo & ~hi, isC = hi & ~lo, isG = lo & hi, isA = ~(lo | hi) & EVEN;
    u64 bad = (isA & f.deny[0]) | (isT & f.deny[1]) | (isC & f.deny[2]) | (isG & f.deny[3]);
    return bad == 0;
}

// The same three operators as per-position sets for the bit-sliced stream test of
// filter_kernels.hip: position i of the k-mer must carry a code of sets[i] (4-bit mask, bit c = code c
// allowed; 0xF = any).  `=` is k singletons, `^@` a prefix of singletons, `@>` the IUPAC text.
struct FilterBits {
    u32 sets[4];   // 32 positions x 4 bits, position i at bits 4*(i%8) of sets[i/8]
    int k;
};

// relaxed, agent-scope 8-byte accesses for words shared between workgroups inside one launch
// (self-validating {flag, value} granules: the only inter-workgroup hand-off used here)
__device__ __forceinline__ u64 ld_agent(const u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(u64 *p, u64 v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Inclusive scan over the 64 lanes of a wave with DPP adds (row shifts inside rows of 16, then two row
// broadcasts): six VALU instructions.  The __shfl_up form is six ds_bpermute round trips through
// the LDS crossbar -- measured with cycle stamps, the 1024-element scan of a scatter tile cost
// 3.2 K of the tile's 33 K cycles that way.
__device__ __forceinline__ u32 wave_incl_scan(u32 x)
{
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return (u32)v;
}
// Sum over the wave, returned in every lane.
__device__ __forceinline__ u32 wave_sum(u32 x)
{
    return (u32)__builtin_amdgcn_readlane((int)wave_incl_scan(x), 63);
}

// In-place exclusive scan of arr[0..n), n <= NT, in LDS by NT threads (one element per thread); returns the total.  The
// caller has synchronised before the call; the function synchronises before returning.  tid = the caller's thread index
// (callers that keep it opaque per tile -- so that nothing derived from it is held across their tile loops -- pass that one).
template <int NT>
__device__ __forceinline__ u32 block_scan_value(u32 v, u32 *arr, int n, u32 *wtmp, int tid);
template <int NT>
__device__ __forceinline__ u32 block_scan_small(u32 *arr, int n, u32 *wtmp, int tid)
{
    return block_scan_value<NT>(tid < n ? arr[tid] : 0u, arr, n, wtmp, tid);
}
// the same with thread tid's element handed over as a value (0 for tid >= n): arr[tid] = the sum of the elements before it
template <int NT>
__device__ __forceinline__ u32 block_scan_value(u32 v, u32 *arr, int n, u32 *wtmp, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const u32 inc = wave_incl_scan(v);
    if (lane == 63)
        wtmp[wave] = inc;
    __syncthreads();
    // every wave scans the (at most 16) wave totals itself: one LDS read, one DPP scan
    const u32 ws = wave_incl_scan(lane < NT / 64 ? wtmp[lane] : 0u);
    const int wv1 = __builtin_amdgcn_readfirstlane(wave);
    const u32 total1 = (u32)__builtin_amdgcn_readlane((int)ws, NT / 64 - 1);
    const u32 wbase1 = wv1 ? (u32)__builtin_amdgcn_readlane((int)ws, wv1 - 1) : 0u;
    if (tid < n)
        arr[tid] = wbase1 + inc - v;
    __syncthreads();
    return total1;
}

// In-place exclusive scan of arr[0..n) in LDS by NT threads; returns the total.  The caller has
// synchronised before the call; the function synchronises before returning.
template <int NT>
__device__ __forceinline__ u32 block_scan_inplace(u32 *arr, int n, u32 *wtmp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n <= NT) {                          // one element per thread: no per-thread loops
        const u32 v = tid < n ? arr[tid] : 0;
        const u32 inc = wave_incl_scan(v);
        if (lane == 63)
            wtmp[wave] = inc;
        __syncthreads();
        // every wave scans the (at most 16) wave totals itself: one LDS read, one DPP scan
        const u32 ws = wave_incl_scan(lane < NT / 64 ? wtmp[lane] : 0u);
        const int wv1 = __builtin_amdgcn_readfirstlane(wave);
        const u32 total1 = (u32)__builtin_amdgcn_readlane((int)ws, NT / 64 - 1);
        const u32 wbase1 = wv1 ? (u32)__builtin_amdgcn_readlane((int)ws, wv1 - 1) : 0u;
        if (tid < n)
            arr[tid] = wbase1 + inc - v;
        __syncthreads();
        return total1;
    }
    const int per = (n + NT - 1) / NT;
    const int b = tid * per;
    const int e = (b + per < n) ? b + per : n;
    u32 sum = 0;
#pragma unroll 1
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
    // wave-uniform trip count: keeps the prefix in scalar code instead of NT/64 hoisted predicates
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    u32 wbase = 0, total = 0;
    for (int w = 0; w < wv; w++)
        wbase += wtmp[w];
    for (int w = 0; w < NT / 64; w++)
        total += wtmp[w];
    u32 run = wbase + inc - sum;
#pragma unroll 1
    for (int i = b; i < e; i++) {
        u32 v = arr[i];
        arr[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

}  // namespace dnagpu
