// extract_kernels.hip -- generate_kmers and its fused WHERE operators on gfx950.
//
// All of these are HBM-bound integer kernels: 0.25 B of packed input per k-mer against 8 B (or,
// filtered, 8-16 B per match) of output.  The packed words a wave needs for 128 consecutive
// positions are 4-5 words, so the input side is served by L1 broadcasts; the design target is the
// store side: every wave-instruction writes 1 KiB of consecutive keys (16 B per lane).
#include "kernels.hpp"

namespace dnagpu {

// ------------------------------------------------------------------------------------------------
// synthetic packed dna: word w = splitmix64(seed + w); repeat variant tiles the first motif_len
// bases over the second half.  Mirrors oracle/kmer_oracle.c orc_synth_words{,_repeat}.
// words[w] for w in [w_begin, w_end): a rank of the multi-GPU path generates only its own word chunk
__global__ __launch_bounds__(256) void synth_kernel(u64 *__restrict__ words, u64 w_begin, u64 w_end, u64 n_bases,
                                                    u64 seed, u64 motif_len)
{
    u64 w = w_begin + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= w_end)
        return;
    u64 v;
    u64 half = n_bases / 2;
    if (motif_len == 0 || (w + 1) * 32 <= half) {
        v = splitmix64(seed + w);
    } else {
        v = 0;
        for (unsigned j = 0; j < 32; j++) {
            u64 i = w * 32 + j;
            if (i >= n_bases)
                break;
            u64 src = i < half ? i : (i - half) % motif_len;
            v |= ((splitmix64(seed + (src >> 5)) >> ((src & 31) * 2)) & 3) << (2 * j);
        }
    }
    u64 end = (w + 1) * 32;
    if (end > n_bases) {
        unsigned tail = (unsigned)(n_bases - w * 32);   // 1..31 valid bases in the last word
        v &= (((u64)1 << (2 * tail)) - 1);
    }
    words[w] = v;
}

hipError_t launch_synth(u64 *words, u64 w_begin, u64 w_end, u64 n_bases, u64 seed, u64 motif_len, hipStream_t s)
{
    if (w_end <= w_begin)
        return hipSuccess;
    unsigned grid = (unsigned)((w_end - w_begin + 255) / 256);
    hipLaunchKernelGGL(synth_kernel, dim3(grid), dim3(256), 0, s, words, w_begin, w_end, n_bases, seed, motif_len);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// generate_kmers rows [first, first+count) -> out_keys[0..count)  (dna.c:803-825 per row)
// One workgroup = 4096 consecutive rows; lane l of each wave-instruction stores rows 2l, 2l+1 of
// a 128-row group as one 16-byte store: 1 KiB contiguous per wave-instruction.
constexpr int EXTRACT_TILE = 4096;
typedef unsigned long long ull2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void extract_kernel(const u64 *__restrict__ words, u64 n_words, u64 first,
                                                      u64 count, u64 mask, u64 *__restrict__ out)
{
    const u64 base = (u64)blockIdx.x * EXTRACT_TILE;
    const bool aligned = (((uintptr_t)out) & 15) == 0;
#pragma unroll
    for (int j = 0; j < EXTRACT_TILE / 512; j++) {
        u64 i = base + (u64)j * 512 + 2 * threadIdx.x;
        if (i >= count)
            break;
        u64 k0 = key_at(words, n_words, first + i, mask);
        if (i + 1 < count) {
            u64 k1 = key_at(words, n_words, first + i + 1, mask);
            if (aligned) {
                ull2_t v;
                v.x = k0;
                v.y = k1;
                __builtin_nontemporal_store(v, reinterpret_cast<ull2_t *>(out + i));
            } else {
                __builtin_nontemporal_store(k0, &out[i]);
                __builtin_nontemporal_store(k1, &out[i + 1]);
            }
        } else {
            __builtin_nontemporal_store(k0, &out[i]);
        }
    }
}

hipError_t launch_extract(const u64 *words, u64 n_words, u64 first, u64 count, int k, u64 *out_keys,
                          hipStream_t s)
{
    if (count == 0)
        return hipSuccess;
    unsigned grid = (unsigned)((count + EXTRACT_TILE - 1) / EXTRACT_TILE);
    hipLaunchKernelGGL(extract_kernel, dim3(grid), dim3(256), 0, s, words, n_words, first, count,
                       kmer_mask(k), out_keys);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// batched operators over key arrays

__global__ __launch_bounds__(256) void hash_batch_kernel(const u64 *__restrict__ keys, u64 n,
                                                         u32 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        __builtin_nontemporal_store(pg_kmer_hash(__builtin_nontemporal_load(&keys[i])), &out[i]);
}

// ------------------------------------------------------------------------------------------------
// A table of sequences in one packed stream (GROUP BY over FROM table, LATERAL generate_kmers(sequence, k):
// test.sql:140-150).  marks: one bit per base, set where a sequence starts.
__global__ __launch_bounds__(256) void batch_marks_kernel(const u64 *__restrict__ starts, u64 n_seqs, u32 *__restrict__ marks,
                                                          u64 n_mark_words)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x + 1;      // starts[0] = 0 marks nothing
    if (i >= n_seqs)
        return;
    const u64 b = starts[i];
    if ((b >> 5) < n_mark_words)
        atomicOr(&marks[b >> 5], 1u << (b & 31));
}

hipError_t launch_batch_marks(const u64 *starts, u64 n_seqs, u32 *marks, u64 n_mark_words, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(marks, 0, (size_t)n_mark_words * sizeof(u32), s);
    if (e != hipSuccess || n_seqs < 2)
        return e;
    hipLaunchKernelGGL(batch_marks_kernel, dim3((unsigned)((n_seqs - 1 + 255) / 256)), dim3(256), 0, s, starts, n_seqs, marks,
                       n_mark_words);
    return hipGetLastError();
}

// rows of the table for one k: the sum over the sequences of max(0, length - k + 1) (a resident table is counted for any
// k without its starts travelling again)
__global__ __launch_bounds__(256) void batch_rows_kernel(const u64 *__restrict__ starts, u64 n_seqs, u32 k,
                                                         unsigned long long *__restrict__ rows)
{
    u64 sum = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_seqs; i += (u64)gridDim.x * blockDim.x) {
        const u64 len = starts[i + 1] - starts[i];
        sum += len >= k ? len - k + 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0 && sum)
        atomicAdd(rows, (unsigned long long)sum);
}

hipError_t launch_batch_rows(const u64 *starts, u64 n_seqs, int k, u64 *rows, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(rows, 0, sizeof(u64), s);
    if (e != hipSuccess || n_seqs == 0)
        return e;
    const u64 blocks = (n_seqs + 1023) / 1024;
    hipLaunchKernelGGL(batch_rows_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, starts, n_seqs, (u32)k,
                       reinterpret_cast<unsigned long long *>(rows));
    return hipGetLastError();
}

// The keys of the table's rows for the engines that take keys (short k-mers, short tables): a wave owns 1024 consecutive
// rows, 16 rounds of one row per lane; a row is kept when no sequence starts among the k - 1 bases behind its first (its
// window lies in one sequence).  The wave's kept keys wait in registers, one returning add reserves their slots, and every
// round's keys go out as one contiguous run.  Order of the keys: none (they are counted).
constexpr int BK_ROUNDS = 16;
__global__ __launch_bounds__(256) void batch_keys_kernel(const u64 *__restrict__ words, u64 n_words, const u32 *__restrict__ marks,
                                                         u64 n_mark_words, u64 n_rows, int k, u64 *__restrict__ out_keys,
                                                         unsigned long long *__restrict__ cursor)
{
    const int lane = threadIdx.x & 63;
    const u64 wave = (u64)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const u64 row0 = wave * (u64)(64 * BK_ROUNDS);
    if (row0 >= n_rows)
        return;
    const u64 kmask = kmer_mask(k);
    const u32 span = (u32)k - 1u;                                      // marks at bases p + 1 .. p + k - 1 spoil row p
    u64 key[BK_ROUNDS];
    u32 keep = 0, before[BK_ROUNDS], total = 0;
#pragma unroll
    for (int r = 0; r < BK_ROUNDS; r++) {
        const u64 p = row0 + (u64)r * 64 + lane;
        bool ok = p < n_rows;
        key[r] = 0;
        if (ok) {
            key[r] = key_at(words, n_words, p, kmask);
            if (span) {
                const u64 q = p + 1, mw = q >> 5;
                const unsigned sh = (unsigned)(q & 31);
                const u64 m0 = mw < n_mark_words ? marks[mw] : 0u, m1 = mw + 1 < n_mark_words ? marks[mw + 1] : 0u;
                const u32 bits = (u32)(((m1 << 32) | m0) >> sh);       // marks of bases q .. q + 31
                ok = (bits & (span >= 32u ? ~0u : (1u << span) - 1u)) == 0;
            }
        }
        const u64 b = __ballot(ok);
        before[r] = total + (u32)__popcll(b & (((u64)1 << lane) - 1));
        total += (u32)__popcll(b);
        keep |= ok ? 1u << r : 0u;
    }
    unsigned long long base = 0;
    if (lane == 0 && total)
        base = atomicAdd(cursor, (unsigned long long)total);
    base = ((unsigned long long)(u32)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
           (u32)__builtin_amdgcn_readfirstlane((int)(u32)base);
#pragma unroll
    for (int r = 0; r < BK_ROUNDS; r++)
        if ((keep >> r) & 1u)
            __builtin_nontemporal_store(key[r], &out_keys[base + before[r]]);
}

hipError_t launch_batch_keys(const u64 *words, u64 n_words, const u32 *marks, u64 n_mark_words, u64 n_rows, int k, u64 *out_keys,
                             unsigned long long *cursor, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(cursor, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || n_rows == 0)
        return e;
    const u64 per_block = (u64)4 * 64 * BK_ROUNDS;
    hipLaunchKernelGGL(batch_keys_kernel, dim3((unsigned)((n_rows + per_block - 1) / per_block)), dim3(256), 0, s, words, n_words, marks,
                       n_mark_words, n_rows, k, out_keys, cursor);
    return hipGetLastError();
}

hipError_t launch_hash_batch(const u64 *keys, u64 n, u32 *out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    u64 blocks = (n + 255) / 256;
    unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(hash_batch_kernel, dim3(grid), dim3(256), 0, s, keys, n, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void match_batch_kernel(const u64 *__restrict__ keys, u64 n, FilterDev f,
                                                          uint8_t *__restrict__ flags)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        flags[i] = filter_match(f, __builtin_nontemporal_load(&keys[i])) ? 1 : 0;
}

hipError_t launch_match_batch(const u64 *keys, u64 n, const FilterDev &f, uint8_t *flags, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    u64 blocks = (n + 255) / 256;
    unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(match_batch_kernel, dim3(grid), dim3(256), 0, s, keys, n, f, flags);
    return hipGetLastError();
}

// sum(count), #(count == 1), wrapping sum of pair digests: three wave reductions + 3 atomics per block
__global__ __launch_bounds__(256) void hist_summary_kernel(const u64 *__restrict__ keys,
                                                           const u32 *__restrict__ counts, u64 n,
                                                           u64 *__restrict__ result)
{
    __shared__ u64 part[3][4];
    u64 t = 0, u = 0, c = 0;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        u64 cnt = counts[i];
        if (cnt == 0)                            // padding of an unordered histogram (a bucket that held copies)
            continue;
        t += cnt;
        u += cnt == 1;
        c += pair_mix(keys[i], cnt);
    }
    for (int off = 32; off > 0; off >>= 1) {
        t += __shfl_down(t, off);
        u += __shfl_down(u, off);
        c += __shfl_down(c, off);
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = t;
        part[1][threadIdx.x >> 6] = u;
        part[2][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        u64 v = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
        atomicAdd(reinterpret_cast<unsigned long long *>(result + threadIdx.x), (unsigned long long)v);
    }
}

// ------------------------------------------------------------------------------------------------
// dnagpu_hist_merge: the groups of two histograms added up through one open-addressing table in device memory (8-byte
// keys claimed by compare-and-swap, 32-bit counts added): a utility for callers that count a large table batch by batch,
// not a hot path -- every group is one random probe.  The all-ones key (32 G's) cannot live in the table: its count is
// kept apart.
__global__ __launch_bounds__(256) void merge_insert_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ counts, u64 n,
                                                           u64 *__restrict__ tkeys, u32 *__restrict__ tcnt, u64 tmask,
                                                           unsigned long long *__restrict__ ones)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const u32 c = counts[i];
    if (c == 0)
        return;                                    // (a padding slot of an unordered histogram)
    const u64 key = keys[i];
    if (key == ~(u64)0) {
        atomicAdd(ones, (unsigned long long)c);
        return;
    }
    u64 h = splitmix64(key) & tmask;
    for (;;) {
        const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&tkeys[h]), ~0ull, (unsigned long long)key);
        if (old == ~(u64)0 || old == key) {
            atomicAdd(&tcnt[h], c);
            return;
        }
        h = (h + 1) & tmask;
    }
}

// the table's groups, dense, in table order: a workgroup reserves the slots of its 4096 table entries with one add
__global__ __launch_bounds__(256) void merge_compact_kernel(const u64 *__restrict__ tkeys, const u32 *__restrict__ tcnt, u64 t_slots,
                                                            u64 *__restrict__ out_keys, u32 *__restrict__ out_counts,
                                                            unsigned long long *__restrict__ cursor)
{
    __shared__ u32 wsum[4];
    __shared__ unsigned long long base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 i0 = (u64)blockIdx.x * 4096 + (u64)threadIdx.x * 16;
    u64 k[16];
    u32 c[16], mine = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        k[j] = i0 + j < t_slots ? tkeys[i0 + j] : ~(u64)0;
        c[j] = i0 + j < t_slots ? tcnt[i0 + j] : 0u;
        mine += k[j] != ~(u64)0 ? 1u : 0u;
    }
    const u32 inc = wave_incl_scan(mine);
    if (lane == 63)
        wsum[wave] = inc;
    __syncthreads();
    u32 before = 0, total = 0;
    for (int w = 0; w < 4; w++) {
        before += w < wave ? wsum[w] : 0u;
        total += wsum[w];
    }
    if (threadIdx.x == 0)
        base_s = total ? atomicAdd(cursor, (unsigned long long)total) : 0ull;
    __syncthreads();
    u64 o = base_s + before + inc - mine;
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (k[j] != ~(u64)0) {
            out_keys[o] = k[j];
            out_counts[o] = c[j];
            o++;
        }
}

hipError_t launch_merge_insert(const u64 *keys, const u32 *counts, u64 n, u64 *tkeys, u32 *tcnt, u64 t_slots,
                               unsigned long long *ones, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(merge_insert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, counts, n, tkeys, tcnt, t_slots - 1,
                       ones);
    return hipGetLastError();
}

hipError_t launch_merge_compact(const u64 *tkeys, const u32 *tcnt, u64 t_slots, u64 *out_keys, u32 *out_counts,
                                unsigned long long *cursor, hipStream_t s)
{
    hipLaunchKernelGGL(merge_compact_kernel, dim3((unsigned)((t_slots + 4095) / 4096)), dim3(256), 0, s, tkeys, tcnt, t_slots, out_keys,
                       out_counts, cursor);
    return hipGetLastError();
}

hipError_t launch_hist_summary(const u64 *keys, const u32 *counts, u64 n, u64 *result3, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    u64 blocks = (n + 255) / 256;
    unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
    hipLaunchKernelGGL(hist_summary_kernel, dim3(grid), dim3(256), 0, s, keys, counts, n, result3);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// text <-> packed forms (the steps either side of the path: encode_dna dna.c:114-128 with the
// validation of dna.c:159-171, decode_dna dna.c:135-152, decode_kmer dna.c:428-452)

// 'A','T','C','G' -> 0,1,2,3 (dna.c:120-123); anything else -> 4
__device__ __forceinline__ u32 base_code(u32 c)
{
    u32 t = (c >> 1) & 3;                        // A=0 C=1 T=2 G=3
    u32 code = ((t & 1) << 1) | (t >> 1);        // A=0 T=1 C=2 G=3
    u32 expect = (0x47544341u >> (8 * t)) & 0xff;   // "ACTG"[t]
    return c == expect ? code : 4u;
}

// one thread per output word: 32 characters in, one packed word out; the position of the first
// invalid character (what the reference reports) is kept with an atomic minimum
__global__ __launch_bounds__(256) void pack_kernel(const unsigned char *__restrict__ text, u64 n_bases,
                                                   u64 *__restrict__ words, unsigned long long *__restrict__ bad_pos)
{
    u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 n_words = (n_bases + 31) / 32;
    if (w >= n_words)
        return;
    const u64 base = w * 32;
    u32 chunk[8];
    if (base + 32 <= n_bases && ((reinterpret_cast<uintptr_t>(text) & 15) == 0)) {
        const uint4 *p = reinterpret_cast<const uint4 *>(text + base);
        uint4 a = p[0], b = p[1];
        chunk[0] = a.x; chunk[1] = a.y; chunk[2] = a.z; chunk[3] = a.w;
        chunk[4] = b.x; chunk[5] = b.y; chunk[6] = b.z; chunk[7] = b.w;
    } else {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            u32 v = 0;
            for (int r = 0; r < 4; r++) {
                u64 i = base + q * 4 + r;
                v |= (i < n_bases ? (u32)text[i] : (u32)'A') << (8 * r);   // tail bits stay zero: 'A' = 00
            }
            chunk[q] = v;
        }
    }
    u64 out = 0;
    u32 first_bad = 32;
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            u32 code = base_code((chunk[q] >> (8 * r)) & 0xff);
            int j = q * 4 + r;
            if (code > 3 && first_bad == 32)
                first_bad = j;
            out |= (u64)(code & 3) << (2 * j);
        }
    if (first_bad != 32)
        atomicMin(bad_pos, (unsigned long long)(base + first_bad));
    words[w] = out;
}

hipError_t launch_pack(const unsigned char *text, u64 n_bases, u64 *words, u64 *bad_pos, hipStream_t s)
{
    if (n_bases == 0)
        return hipSuccess;
    u64 n_words = (n_bases + 31) / 32;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, text, n_bases, words,
                       reinterpret_cast<unsigned long long *>(bad_pos));
    return hipGetLastError();
}

// bases [first, first+count) of the packed sequence -> count characters
__global__ __launch_bounds__(256) void unpack_kernel(const u64 *__restrict__ words, u64 first, u64 count,
                                                     unsigned char *__restrict__ text)
{
    u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i0 >= count)
        return;
    const u32 L = 0x47435441u;                   // codes 0..3 -> 'A','T','C','G'
    unsigned char buf[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        u64 p = first + i0 + j;
        u32 code = (u32)(words[p >> 5] >> ((p & 31) * 2)) & 3;
        buf[j] = (unsigned char)((L >> (8 * code)) & 0xff);
    }
    if (i0 + 16 <= count && ((reinterpret_cast<uintptr_t>(text + i0) & 15) == 0)) {
        uint4 v;                                 // one 16-byte store per lane: 1 KiB per wave-instruction
        v.x = buf[0] | (buf[1] << 8) | (buf[2] << 16) | ((u32)buf[3] << 24);
        v.y = buf[4] | (buf[5] << 8) | (buf[6] << 16) | ((u32)buf[7] << 24);
        v.z = buf[8] | (buf[9] << 8) | (buf[10] << 16) | ((u32)buf[11] << 24);
        v.w = buf[12] | (buf[13] << 8) | (buf[14] << 16) | ((u32)buf[15] << 24);
        *reinterpret_cast<uint4 *>(text + i0) = v;
    } else {
        for (int j = 0; j < 16 && i0 + j < count; j++)
            text[i0 + j] = buf[j];
    }
}

hipError_t launch_unpack(const u64 *words, u64 first, u64 count, unsigned char *text, hipStream_t s)
{
    if (count == 0)
        return hipSuccess;
    u64 threads = (count + 15) / 16;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, words, first, count,
                       text);
    return hipGetLastError();
}

// dna_recv / dna_send (dna.c:244-291): the wire carries every packed word through pq_sendint64, i.e.
// byte-swapped on this little-endian machine.  One 16-byte access per lane either way; dna_recv also
// clears the bits behind the last base (the type's palloc0 invariant, dna.c:186).
__global__ __launch_bounds__(256) void wire_swap_kernel(const u64 *__restrict__ src, u64 *__restrict__ dst, u64 n_words,
                                                        u64 last_mask)
{
    const u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n_words)
        return;
    u64 a = __builtin_bswap64(__builtin_nontemporal_load(&src[i]));
    if (i + 1 < n_words) {
        u64 b = __builtin_bswap64(__builtin_nontemporal_load(&src[i + 1]));
        if (i + 2 == n_words)
            b &= last_mask;
        __builtin_nontemporal_store(a, &dst[i]);
        __builtin_nontemporal_store(b, &dst[i + 1]);
    } else {
        dst[i] = a & last_mask;
    }
}

hipError_t launch_wire_swap(const u64 *src, u64 *dst, u64 n_words, u64 last_mask, hipStream_t s)
{
    if (n_words == 0)
        return hipSuccess;
    const u64 threads = (n_words + 1) / 2;
    hipLaunchKernelGGL(wire_swap_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, src, dst, n_words,
                       last_mask);
    return hipGetLastError();
}

// n keys of k bases -> n records of k characters + NUL (stride k+1), kmer_out's text
__global__ __launch_bounds__(256) void kmers_to_text_kernel(const u64 *__restrict__ keys, u64 n, int k,
                                                            unsigned char *__restrict__ text)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const u32 L = 0x47435441u;
    u64 key = keys[i];
    unsigned char *o = text + i * (u64)(k + 1);
    for (int j = 0; j < k; j++)
        o[j] = (unsigned char)((L >> (8 * ((key >> (2 * j)) & 3))) & 0xff);
    o[k] = 0;
}

hipError_t launch_kmers_to_text(const u64 *keys, u64 n, int k, unsigned char *text, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(kmers_to_text_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, n, k, text);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of u32 arrays (tile offsets, node lists): reduce -> scan of block sums -> apply.
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;   // 2048 values per workgroup

u64 scan_tmp_words(u64 n)
{
    return (n + SCAN_TILE - 1) / SCAN_TILE + 1;
}

__device__ __forceinline__ u32 block_reduce_256(u32 v, u32 *sh4)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0)
        sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    u32 r = sh4[0] + sh4[1] + sh4[2] + sh4[3];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_kernel(const u32 *__restrict__ in, u64 n,
                                                                 u32 *__restrict__ sums)
{
    __shared__ u32 sh4[4];
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q++)
        if (base + q < n)
            v += in[base + q];
    u32 r = block_reduce_256(v, sh4);
    if (threadIdx.x == 0)
        sums[blockIdx.x] = r;
}

// one workgroup scans the block sums in place (exclusive) and stores the grand total
__global__ __launch_bounds__(1024) void scan_sums_kernel(u32 *__restrict__ sums, u64 nb, u32 *__restrict__ total)
{
    __shared__ u32 wtot[16];
    __shared__ u32 carry_sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry_sh = 0;
    __syncthreads();
    for (u64 b0 = 0; b0 < nb; b0 += 1024) {
        u64 i = b0 + threadIdx.x;
        u32 v = i < nb ? sums[i] : 0, inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            u32 t = __shfl_up(inc, off);
            if (lane >= off)
                inc += t;
        }
        if (lane == 63)
            wtot[wave] = inc;
        __syncthreads();
        u32 wbase = 0, all = 0;
        for (int w = 0; w < 16; w++) {
            u32 t = wtot[w];
            if (w < wave)
                wbase += t;
            all += t;
        }
        u32 carry = carry_sh;
        if (i < nb)
            sums[i] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0)
            carry_sh = carry + all;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total)
        *total = carry_sh;
}

// (in == out is allowed by launch_scan_u32's contract, so neither pointer is __restrict__: every value of a
// thread's eight is loaded before its first store)
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const u32 *in, u32 *out, u64 n,
                                                                const u32 *__restrict__ sums)
{
    __shared__ u32 wtot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 sum = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q++) {
        v[q] = (base + q < n) ? in[base + q] : 0;
        sum += v[q];
    }
    u32 inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
        u32 t = __shfl_up(inc, off);
        if (lane >= off)
            inc += t;
    }
    if (lane == 63)
        wtot[wave] = inc;
    __syncthreads();
    u32 wbase = 0;
    for (int w = 0; w < wave; w++)
        wbase += wtot[w];
    u32 run = sums[blockIdx.x] + wbase + inc - sum;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q++) {
        if (base + q < n)
            out[base + q] = run;
        run += v[q];
    }
}

// up to four scans of the same length in the three launches of one (blockIdx.y = which array): the selections of the
// record engine scan four flag / size arrays over the final buckets back to back
__global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_multi_kernel(ScanSet set, u64 n, u64 tmp_stride)
{
    __shared__ u32 sh4[4];
    const u32 *in = set.in[blockIdx.y];
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q++)
        if (base + q < n)
            v += in[base + q];
    u32 r = block_reduce_256(v, sh4);
    if (threadIdx.x == 0)
        set.tmp[blockIdx.y * tmp_stride + blockIdx.x] = r;
}

__global__ __launch_bounds__(1024) void scan_sums_multi_kernel(ScanSet set, u64 nb, u64 tmp_stride)
{
    __shared__ u32 wtot[16];
    __shared__ u32 carry_sh;
    u32 *sums = set.tmp + blockIdx.y * tmp_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry_sh = 0;
    __syncthreads();
    for (u64 b0 = 0; b0 < nb; b0 += 1024) {
        u64 i = b0 + threadIdx.x;
        u32 v = i < nb ? sums[i] : 0, inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            u32 t = __shfl_up(inc, off);
            if (lane >= off)
                inc += t;
        }
        if (lane == 63)
            wtot[wave] = inc;
        __syncthreads();
        u32 wbase = 0, all = 0;
        for (int w = 0; w < 16; w++) {
            u32 t = wtot[w];
            if (w < wave)
                wbase += t;
            all += t;
        }
        u32 carry = carry_sh;
        if (i < nb)
            sums[i] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0)
            carry_sh = carry + all;
        __syncthreads();
    }
    if (threadIdx.x == 0 && set.total[blockIdx.y])
        *set.total[blockIdx.y] = carry_sh;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_multi_kernel(ScanSet set, u64 n, u64 tmp_stride)
{
    __shared__ u32 wtot[4];
    const u32 *in = set.in[blockIdx.y];          // (in == out allowed: every value of a thread's eight is loaded first)
    u32 *out = set.out[blockIdx.y];
    const u32 *sums = set.tmp + blockIdx.y * tmp_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 sum = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q++) {
        v[q] = (base + q < n) ? in[base + q] : 0;
        sum += v[q];
    }
    u32 inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
        u32 t = __shfl_up(inc, off);
        if (lane >= off)
            inc += t;
    }
    if (lane == 63)
        wtot[wave] = inc;
    __syncthreads();
    u32 wbase = 0;
    for (int w = 0; w < wave; w++)
        wbase += wtot[w];
    u32 run = sums[blockIdx.x] + wbase + inc - sum;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q++) {
        if (base + q < n)
            out[base + q] = run;
        run += v[q];
    }
}

hipError_t launch_scan_u32_multi(const ScanSet &set, int n_arrays, u64 n, hipStream_t s)
{
    if (n_arrays < 1 || n_arrays > 4)
        return hipErrorInvalidValue;
    if (n == 0) {
        for (int a = 0; a < n_arrays; a++)
            if (set.total[a]) {
                const hipError_t e = hipMemsetAsync(set.total[a], 0, sizeof(u32), s);
                if (e != hipSuccess)
                    return e;
            }
        return hipSuccess;
    }
    const u64 nb = (n + SCAN_TILE - 1) / SCAN_TILE, stride = scan_tmp_words(n);
    hipLaunchKernelGGL(scan_reduce_multi_kernel, dim3((unsigned)nb, (unsigned)n_arrays), dim3(SCAN_BLOCK), 0, s, set, n, stride);
    hipLaunchKernelGGL(scan_sums_multi_kernel, dim3(1, (unsigned)n_arrays), dim3(1024), 0, s, set, nb, stride);
    hipLaunchKernelGGL(scan_apply_multi_kernel, dim3((unsigned)nb, (unsigned)n_arrays), dim3(SCAN_BLOCK), 0, s, set, n, stride);
    return hipGetLastError();
}

hipError_t launch_scan_u32(const u32 *in, u32 *out, u64 n, u32 *tmp, u32 *total, hipStream_t s)
{
    if (n == 0) {
        if (total)
            return hipMemsetAsync(total, 0, sizeof(u32), s);
        return hipSuccess;
    }
    u64 nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, s, in, n, tmp);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, tmp, nb, total);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, s, in, out, n, tmp);
    return hipGetLastError();
}

}  // namespace dnagpu
