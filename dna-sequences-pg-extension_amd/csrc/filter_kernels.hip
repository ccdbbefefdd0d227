// filter_kernels.hip -- generate_kmers fused with a WHERE operator (`=`, `^@`, `@>`), bit-sliced.
//
// The reference evaluates the operator once per row, base by base (kmer_eq dna.c:655-668,
// starts_with dna.c:842-866, contains + nucleotide_matches dna.c:1064-1135).  Here one thread tests
// 32 consecutive rows at once on the packed STREAM, never forming a key for a row that fails:
//
//   S            = the 64 bases (128 bits) starting at the thread's first row
//   x_i, y_i     = S >> 2i and S >> (2i+1): bit 2j of x_i / y_i is the low / high code bit of base j+i,
//                  i.e. of base i of the k-mer of row j
//   allowed_i    = the pattern position's set as a boolean function of (x_i, y_i): one or two
//                  32-bit operations per dword (codes A=00 T=01 C=10 G=11, dna.c:120-123)
//   match        = AND over the pattern's non-N positions of allowed_i, on the even bits
//
// ~10 VALU operations per non-N pattern position per 32 rows.  All three operators are such
// patterns: `=` is k singleton sets, `^@` is a prefix of singletons followed by N, `@>` is the IUPAC
// text itself.  Rows are produced in position order (the reference's row order, test.sql:86-92) by
// two sweeps over the (tiny) packed input: workgroup b counts the matches of its contiguous range
// of tiles; then every workgroup sums the counts of the workgroups before it (<= 2048 values),
// lists each tile's matching rows in LDS in row order and writes keys and positions with
// 16 bytes per lane, 1 KiB per wave-instruction.  Keys are formed only for matching rows.
#include "kernels.hpp"

namespace dnagpu {

constexpr int FB_THREADS = 256;
constexpr int FB_WAVES = FB_THREADS / 64;
constexpr int FB_ROWS = 32;                          // rows per thread and tile
constexpr int FB_TILE = FB_THREADS * FB_ROWS;        // 8192 rows
constexpr int FB_MAX_GROUPS = 8192;                  // workgroups of one sweep (8 per CU)
static_assert(FB_MAX_GROUPS == FILTER_MAX_GROUPS, "kernels.hpp names the group limit");

typedef unsigned long long ull2_t __attribute__((ext_vector_type(2)));

// the 64 bases starting at stream word w (+ a wave-uniform shift of sh = 2 * (first % 32) bits) as 4 dwords
struct Stream4 {
    u32 d[4];
};
struct Words3 {
    u64 w0, w1, w2;
};

__device__ __forceinline__ Words3 words_load(const u64 *__restrict__ words, u64 n_words, u64 w, unsigned sh)
{
    Words3 r;
    r.w0 = w < n_words ? words[w] : 0;
    r.w1 = w + 1 < n_words ? words[w + 1] : 0;
    r.w2 = (sh && w + 2 < n_words) ? words[w + 2] : 0;
    return r;
}

__device__ __forceinline__ Stream4 stream_of(const Words3 &r, unsigned sh)
{
    u64 lo = r.w0, hi = r.w1;
    if (sh) {                                        // wave-uniform
        lo = (r.w0 >> sh) | (r.w1 << (64 - sh));
        hi = (r.w1 >> sh) | (r.w2 << (64 - sh));
    }
    Stream4 s;
    s.d[0] = (u32)lo;
    s.d[1] = (u32)(lo >> 32);
    s.d[2] = (u32)hi;
    s.d[3] = (u32)(hi >> 32);
    return s;
}

// One pattern position on both dwords of the 32 rows: a &= allowed(set; x, y), where bit 2j of x / y is the
// low / high code bit of the base the position looks at for row j (odd bit positions carry garbage that
// the caller masks off).  set: bit0 = A(00), bit1 = T(01), bit2 = C(10), bit3 = G(11) -- nucleotide_matches,
// dna.c:1064-1086.  Every case is one three-input v_bitop3_b32 per dword.
__device__ __forceinline__ void and_allowed(u32 set, u32 x0, u32 y0, u32 x1, u32 y1, u32 &a0, u32 &a1)
{
#define FB_CASE(S, EXPR0, EXPR1) case S: a0 &= (EXPR0); a1 &= (EXPR1); break;
    switch (set) {
        FB_CASE(0x1, ~(x0 | y0), ~(x1 | y1))        // A
        FB_CASE(0x2, x0 & ~y0, x1 & ~y1)            // T
        FB_CASE(0x4, ~x0 & y0, ~x1 & y1)            // C
        FB_CASE(0x8, x0 & y0, x1 & y1)              // G
        FB_CASE(0x3, ~y0, ~y1)                      // W = A,T
        FB_CASE(0xC, y0, y1)                        // S = C,G
        FB_CASE(0x5, ~x0, ~x1)                      // M = A,C
        FB_CASE(0xA, x0, x1)                        // K = G,T
        FB_CASE(0x9, ~(x0 ^ y0), ~(x1 ^ y1))        // R = A,G
        FB_CASE(0x6, x0 ^ y0, x1 ^ y1)              // Y = C,T
        FB_CASE(0xE, x0 | y0, x1 | y1)              // B = not A
        FB_CASE(0xB, x0 | ~y0, x1 | ~y1)            // D = not C
        FB_CASE(0x7, ~(x0 & y0), ~(x1 & y1))        // H = not G
        FB_CASE(0xD, ~x0 | y0, ~x1 | y1)            // V = not T
        FB_CASE(0x0, 0u, 0u)                        // U: matches nothing (dna.c:1070)
    default: break;                                 // N
    }
#undef FB_CASE
}

// the eight pattern positions 8*W .. 8*W+7 (sets in `sw`, 4 bits each); W is a compile-time constant, so the
// dwords the shifts read and the shift bases are too
template <int W>
__device__ __forceinline__ void match_word(u32 sw, const Stream4 &s, u32 &a0, u32 &a1)
{
    if (sw == 0xFFFFFFFFu)                          // eight N in a row: nothing to test
        return;
    constexpr int D = W >> 1;                       // positions 0..15 shift inside dwords 0..2, 16..31 inside 1..3
    const u32 lo = s.d[D], mid = s.d[D + 1], hi = s.d[D + 2];
#pragma unroll 1
    for (u32 j = 0; j < 8; j++, sw >>= 4) {
        const u32 set = sw & 15u;
        if (set == 15u)
            continue;
        const u32 sh = 2u * ((u32)(W & 1) * 8u + j);
        const u32 x0 = __builtin_amdgcn_alignbit(mid, lo, sh), x1 = __builtin_amdgcn_alignbit(hi, mid, sh);
        const u32 y0 = __builtin_amdgcn_alignbit(mid, lo, sh + 1), y1 = __builtin_amdgcn_alignbit(hi, mid, sh + 1);
        and_allowed(set, x0, y0, x1, y1, a0, a1);
    }
}

// Even bit 2j of the result = row j (of the thread's 32) satisfies the pattern.  s0..s3 are fb.sets[] held in
// scalar registers by the caller (positions >= k are N).
__device__ __forceinline__ u64 match_rows(const Stream4 &s, u32 s0, u32 s1, u32 s2, u32 s3)
{
    u32 a0 = ~0u, a1 = ~0u;
    match_word<0>(s0, s, a0, a1);
    match_word<1>(s1, s, a0, a1);
    match_word<2>(s2, s, a0, a1);
    match_word<3>(s3, s, a0, a1);
    return (((u64)a1 << 32) | a0) & 0x5555555555555555ull;
}

// even-bit mask of the rows of a thread that exist: `left` rows remain from the thread's first row on
__device__ __forceinline__ u64 valid_rows(long long left)
{
    const u64 EVEN = 0x5555555555555555ull;
    if (left >= FB_ROWS)
        return EVEN;
    if (left <= 0)
        return 0;
    return EVEN & (((u64)1 << (2 * (unsigned)left)) - 1);
}

// bit 2j -> bit j
__device__ __forceinline__ u32 compact_even(u64 m)
{
    u32 lo = (u32)m, hi = (u32)(m >> 32);
    lo = (lo | (lo >> 1)) & 0x33333333u;
    hi = (hi | (hi >> 1)) & 0x33333333u;
    lo = (lo | (lo >> 2)) & 0x0f0f0f0fu;
    hi = (hi | (hi >> 2)) & 0x0f0f0f0fu;
    lo = (lo | (lo >> 4)) & 0x00ff00ffu;
    hi = (hi | (hi >> 4)) & 0x00ff00ffu;
    lo = (lo | (lo >> 8)) & 0x0000ffffu;
    hi = (hi | (hi >> 8));
    return lo | (hi << 16);
}

// ---- sweep 1: matches per workgroup range ----------------------------------------------------
__global__ __launch_bounds__(FB_THREADS) void fb_count_kernel(const u64 *__restrict__ words, u64 n_words, u64 first,
                                                              u64 count, FilterBits fb, u32 tiles_per_group,
                                                              u32 *__restrict__ group_counts)
{
    __shared__ u32 wsum[FB_WAVES];
    const u32 s0 = fb.sets[0], s1 = fb.sets[1], s2 = fb.sets[2], s3 = fb.sets[3];
    const unsigned sh = (unsigned)(first & 31) * 2;
    const u64 w_first = first >> 5;
    const u64 n_tiles = (count + FB_TILE - 1) / FB_TILE;
    u64 t0 = (u64)blockIdx.x * tiles_per_group, t1 = t0 + tiles_per_group;
    if (t1 > n_tiles)
        t1 = n_tiles;
    u32 c = 0;
    // the next tile's words are requested before the current tile is tested
    Words3 nxt = words_load(words, n_words, w_first + t0 * (FB_TILE / 32) + threadIdx.x, sh);
    for (u64 t = t0; t < t1; t++) {
        const Words3 cur = nxt;
        if (t + 1 < t1)
            nxt = words_load(words, n_words, w_first + (t + 1) * (FB_TILE / 32) + threadIdx.x, sh);
        const u64 row0 = t * FB_TILE + (u64)threadIdx.x * FB_ROWS;
        const u64 m = match_rows(stream_of(cur, sh), s0, s1, s2, s3);
        c += (u32)__popcll(m & valid_rows(row0 < count ? (long long)(count - row0) : 0));
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 g = 0;
        for (int w = 0; w < FB_WAVES; w++)
            g += wsum[w];
        group_counts[blockIdx.x] = g;
    }
}

// ---- sweep 2: position-ordered keys and positions ---------------------------------------------
// WIDE: out_keys / out_pos are 16-byte aligned at the same index parity `par`, so two consecutive
// slots go out as one 16-byte store.
// (A wave-granular variant -- every wave lists and writes its own 2048 rows, no workgroup barrier -- was
// measured 8-10 % slower at selectivity 1/4: four times as many, four times shorter output bursts.)
template <bool HAS_KEYS, bool HAS_POS, bool WIDE>
__global__ __launch_bounds__(FB_THREADS) void fb_write_kernel(const u64 *__restrict__ words, u64 n_words, u64 first,
                                                              u64 count, u64 mask, FilterBits fb, u32 tiles_per_group,
                                                              const u32 *__restrict__ group_counts,
                                                              u64 *__restrict__ out_keys, u64 *__restrict__ out_pos,
                                                              u64 cap, u32 par, u64 *__restrict__ total_out)
{
    __shared__ unsigned short list[FB_TILE];         // tile-local rows of the matches, in row order
    __shared__ u64 wsh[FB_THREADS + 2];              // the tile's packed words: keys are cut from here
    __shared__ u32 wtot[FB_WAVES];
    __shared__ u64 base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // matches before this workgroup's range
    {
        u32 part = 0;
        for (u32 g = tid; g < blockIdx.x; g += FB_THREADS)
            part += group_counts[g];
        // partial sums of up to 2^32-1 rows cannot overflow 32 bits in total
        part = wave_sum(part);
        if (lane == 0)
            wtot[wave] = part;
        __syncthreads();
        if (tid == 0) {
            u64 b = 0;
            for (int w = 0; w < FB_WAVES; w++)
                b += wtot[w];
            base_sh = b;
        }
        __syncthreads();
    }
    u64 off = base_sh;

    const u32 s0 = fb.sets[0], s1 = fb.sets[1], s2 = fb.sets[2], s3 = fb.sets[3];
    const u32 fo = (u32)(first & 31);
    const unsigned sh = fo * 2;
    const u64 w_first = first >> 5;
    const u64 n_tiles = (count + FB_TILE - 1) / FB_TILE;
    u64 t0 = (u64)blockIdx.x * tiles_per_group, t1 = t0 + tiles_per_group;
    if (t1 > n_tiles)
        t1 = n_tiles;
    Words3 nxt = words_load(words, n_words, w_first + t0 * (FB_TILE / 32) + tid, sh);
    for (u64 t = t0; t < t1; t++) {
        const u64 tile_row0 = t * FB_TILE;
        const u64 row0 = tile_row0 + (u64)tid * FB_ROWS;
        const Words3 cur = nxt;
        if (t + 1 < t1)
            nxt = words_load(words, n_words, w_first + (t + 1) * (FB_TILE / 32) + tid, sh);
        u32 m = compact_even(match_rows(stream_of(cur, sh), s0, s1, s2, s3) &
                             valid_rows(row0 < count ? (long long)(count - row0) : 0));
        const u32 c = (u32)__popc(m);
        const u32 inc = wave_incl_scan(c);
        if (lane == 63)
            wtot[wave] = inc;
        __syncthreads();                             // also: the previous tile's list and words have been read
        u32 wbase = 0, tile_cnt = 0;
#pragma unroll
        for (int w = 0; w < FB_WAVES; w++) {
            const u32 v = wtot[w];
            wbase += w < wave ? v : 0u;
            tile_cnt += v;
        }
        wsh[tid] = cur.w0;
        if (tid == FB_THREADS - 1) {
            wsh[FB_THREADS] = cur.w1;
            wsh[FB_THREADS + 1] = cur.w2;
        }
        u32 r = wbase + inc - c;
        const u32 row_in_tile = (u32)tid * FB_ROWS;
        while (m) {
            const u32 j = (u32)__builtin_ctz(m);
            m &= m - 1;
            list[r++] = (unsigned short)(row_in_tile + j);
        }
        __syncthreads();

        // key of tile-local row r: bits [2q, 2q + 2k) of the tile's words, q = r + first % 32
        auto key_of = [&](u32 row) -> u64 {
            const u32 q = row + fo;
            return funnel(wsh[q >> 5], wsh[(q >> 5) + 1], (q & 31u) * 2u) & mask;
        };
        if (WIDE) {
            // slot pairs (s, s+1) with (off + s + par) even: 16-byte aligned in both arrays
            const int lead = (int)((off + par) & 1);
            for (int s = 2 * tid - lead; s < (int)tile_cnt; s += 2 * FB_THREADS) {
                const bool v0 = s >= 0, v1 = s + 1 < (int)tile_cnt;
                const u32 r0 = list[v0 ? s : 0], r1 = list[v1 ? s + 1 : s];
                const u64 p0 = first + tile_row0 + r0, p1 = first + tile_row0 + r1;
                const u64 i0 = off + (u64)(long long)s;
                if (v0 && v1 && i0 + 1 < cap) {
                    if (HAS_KEYS) {
                        ull2_t kv;
                        kv.x = key_of(r0);
                        kv.y = key_of(r1);
                        __builtin_nontemporal_store(kv, reinterpret_cast<ull2_t *>(out_keys + i0));
                    }
                    if (HAS_POS) {
                        ull2_t pv;
                        pv.x = p0;
                        pv.y = p1;
                        __builtin_nontemporal_store(pv, reinterpret_cast<ull2_t *>(out_pos + i0));
                    }
                } else {
                    if (v0 && i0 < cap) {
                        if (HAS_KEYS)
                            __builtin_nontemporal_store(key_of(r0), &out_keys[i0]);
                        if (HAS_POS)
                            __builtin_nontemporal_store(p0, &out_pos[i0]);
                    }
                    if (v1 && i0 + 1 < cap) {
                        if (HAS_KEYS)
                            __builtin_nontemporal_store(key_of(r1), &out_keys[i0 + 1]);
                        if (HAS_POS)
                            __builtin_nontemporal_store(p1, &out_pos[i0 + 1]);
                    }
                }
            }
        } else {
            for (u32 s = tid; s < tile_cnt; s += FB_THREADS) {
                const u32 r0 = list[s];
                const u64 p0 = first + tile_row0 + r0;
                const u64 i0 = off + s;
                if (i0 < cap) {
                    if (HAS_KEYS)
                        __builtin_nontemporal_store(key_of(r0), &out_keys[i0]);
                    if (HAS_POS)
                        __builtin_nontemporal_store(p0, &out_pos[i0]);
                }
            }
        }
        off += tile_cnt;
    }
    if (total_out && blockIdx.x == gridDim.x - 1 && tid == 0)
        *total_out = off;
}

// groups of the two sweeps for `count` rows: every group takes the same number of consecutive tiles
void filter_bits_geometry(u64 count, u32 *n_groups, u32 *tiles_per_group)
{
    const u64 n_tiles = (count + FB_TILE - 1) / FB_TILE;
    u64 tpg = (n_tiles + FB_MAX_GROUPS - 1) / FB_MAX_GROUPS;
    if (tpg == 0)
        tpg = 1;
    *tiles_per_group = (u32)tpg;
    *n_groups = (u32)((n_tiles + tpg - 1) / tpg);
}

hipError_t launch_filter_bits_count(const u64 *words, u64 n_words, u64 first, u64 count, const FilterBits &fb,
                                    u32 *group_counts, hipStream_t s)
{
    if (count == 0)
        return hipSuccess;
    u32 groups, tpg;
    filter_bits_geometry(count, &groups, &tpg);
    hipLaunchKernelGGL(fb_count_kernel, dim3(groups), dim3(FB_THREADS), 0, s, words, n_words, first, count, fb, tpg,
                       group_counts);
    return hipGetLastError();
}

template <bool HK, bool HP>
static void launch_write_variant(bool wide, dim3 grid, hipStream_t s, const u64 *words, u64 n_words, u64 first, u64 count,
                                 u64 mask, const FilterBits &fb, u32 tpw, const u32 *group_counts,
                                 u64 *out_keys, u64 *out_pos, u64 cap, u32 par, u64 *total_out)
{
    if (wide)
        hipLaunchKernelGGL((fb_write_kernel<HK, HP, true>), grid, dim3(FB_THREADS), 0, s, words, n_words, first, count,
                           mask, fb, tpw, group_counts, out_keys, out_pos, cap, par, total_out);
    else
        hipLaunchKernelGGL((fb_write_kernel<HK, HP, false>), grid, dim3(FB_THREADS), 0, s, words, n_words, first, count,
                           mask, fb, tpw, group_counts, out_keys, out_pos, cap, par, total_out);
}

hipError_t launch_filter_bits_write(const u64 *words, u64 n_words, u64 first, u64 count, int k, const FilterBits &fb,
                                    const u32 *group_counts, u64 *out_keys, u64 *out_pos, u64 cap, u64 *total_out,
                                    hipStream_t s)
{
    if (count == 0)
        return hipSuccess;
    u32 groups, tpw;
    filter_bits_geometry(count, &groups, &tpw);
    // 16-byte stores need both arrays 8-byte aligned with the same index parity at 16-byte boundaries
    const uintptr_t ak = reinterpret_cast<uintptr_t>(out_keys), ap = reinterpret_cast<uintptr_t>(out_pos);
    const uintptr_t ref = out_keys ? ak : ap;
    const bool wide = (ref & 7) == 0 && (!out_keys || !out_pos || ((ak ^ ap) & 15) == 0);
    const u32 par = (u32)((ref >> 3) & 1);
    const u64 mask = kmer_mask(k);
    const dim3 grid(groups);
    if (out_keys && out_pos)
        launch_write_variant<true, true>(wide, grid, s, words, n_words, first, count, mask, fb, tpw, group_counts,
                                         out_keys, out_pos, cap, par, total_out);
    else if (out_keys)
        launch_write_variant<true, false>(wide, grid, s, words, n_words, first, count, mask, fb, tpw, group_counts,
                                          out_keys, out_pos, cap, par, total_out);
    else
        launch_write_variant<false, true>(wide, grid, s, words, n_words, first, count, mask, fb, tpw, group_counts,
                                          out_keys, out_pos, cap, par, total_out);
    return hipGetLastError();
}

}  // namespace dnagpu
