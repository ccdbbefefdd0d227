// dnagpu_api.hip -- the C-ABI of include/dnagpu.h over the gfx950 kernels.
//
// Host side only: argument checks in the reference's terms (same conditions, same message text as
// the ereport() sites of dna.c), device buffer pool, the level loop of the count, event timing.
// There is no CPU fallback: without a HIP device every entry point fails with DNAGPU_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/dnagpu.h"
#include "kernels.hpp"

using namespace dnagpu;

// ------------------------------------------------------------------------------------------------
// errors
static thread_local char g_err[512] = "";

static void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *dnagpu_strerror(int status)
{
    switch (status) {
    case DNAGPU_OK: return "ok";
    case DNAGPU_ERR_INVALID_K: return "Invalid k value: must be between 1 and 32";             // dna.c:773
    case DNAGPU_ERR_QKMER_LEN_MISMATCH: return "Qkmer pattern and kmer lengths do not match";   // dna.c:1107
    case DNAGPU_ERR_PREFIX_TOO_LONG: return "Prefix length cannot exceed kmer length";         // dna.c:855
    case DNAGPU_ERR_QKMER_INVALID: return "Invalid Qkmer sequence: must contain valid IUPAC nucleotide codes";  // dna.c:916
    case DNAGPU_ERR_BAD_ARG: return "bad argument";
    case DNAGPU_ERR_TOO_LARGE: return "too many k-mers for one call (limit 2^32 - 1)";
    case DNAGPU_ERR_NO_DEVICE: return "no usable HIP device (gfx950 required)";
    case DNAGPU_ERR_OOM: return "out of memory";
    case DNAGPU_ERR_HIP: return "HIP runtime or kernel failure";
    case DNAGPU_ERR_INTERNAL: return "internal error";
    case DNAGPU_ERR_DNA_EMPTY: return "DNA sequence cannot be empty";                         // dna.c:161
    case DNAGPU_ERR_DNA_INVALID_CHAR: return "Invalid character in DNA sequence";              // dna.c:166 (+ ": %c")
    }
    return "unknown status";
}

extern "C" const char *dnagpu_last_error(void) { return g_err; }
extern "C" int dnagpu_abi_version(void) { return DNAGPU_ABI_VERSION; }
extern "C" int dnagpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n > 0 ? n : 0;
}

#ifdef DNAGPU_STAMPS
static inline const char *diag_env(const char *name) { return getenv(name); }
#else
static inline const char *diag_env(const char *) { return nullptr; }
#endif

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? DNAGPU_ERR_OOM : DNAGPU_ERR_HIP;              \
        }                                                                                    \
    } while (0)

#define RC_TRY(expr)           \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != DNAGPU_OK)  \
            return rc_;        \
    } while (0)

// No C++ exception may cross the C-ABI (a PostgreSQL backend would die in std::terminate): every
// extern "C" entry point that can allocate on the host (pool bookkeeping, event lists, node lists) runs its
// body inside this guard.
template <typename F>
static int guarded(F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        set_err("host allocation failed");
        return DNAGPU_ERR_OOM;
    } catch (...) {
        set_err("unexpected C++ exception");
        return DNAGPU_ERR_INTERNAL;
    }
}

// ------------------------------------------------------------------------------------------------
// context + device buffer pool
struct PoolBlock {
    void *ptr;
    size_t size;
    bool in_use;
    size_t guard_at = 0;      // DNAGPU_DEBUG_GUARD_POOL: offset of the block's guard band (0 = none)
};
constexpr size_t POOL_GUARD = 256;               // bytes of 0xA5 behind the bytes a caller asked for

struct dnagpu_ctx {
    int device;
    hipStream_t stream;
    std::vector<PoolBlock> pool;
    bool profiling;
    dnagpu_phase_times last_times;
    // per-call event list (profiling)
    std::vector<hipEvent_t> ev;
    std::vector<const char *> ev_names;
    // pinned, device-visible host words: small results (totals, per-group counts) land here without a
    // staging copy; read after hipStreamSynchronize
    u64 *mailbox;
    u64 mailbox_seq = 0;      // sequence number of the last flagged read-back (read_back)
    unsigned debug_flags;     // DNAGPU_DEBUG_*
    // a count over a TABLE of sequences (dnagpu_count_kmers_batch): one bit per base of the packed stream, set where a
    // sequence starts; null otherwise.  Level 0 of the super-k-mer engine makes no record of rows that reach across a mark.
    const u32 *batch_marks = nullptr;
    u64 batch_mark_words = 0;
};
constexpr size_t MAILBOX_BYTES = (size_t)1 << 20;

// Up to 32 bytes from the host into device memory as a kernel argument: no staging copy, and no wait for a stack
// variable to be consumed.
struct Poke32 {
    u32 w[8];
};
__global__ void poke_kernel(u32 *dst, Poke32 v, int n_words)
{
    if ((int)threadIdx.x < n_words)
        dst[threadIdx.x] = v.w[threadIdx.x];
}
static hipError_t poke(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    Poke32 v;
    memset(&v, 0, sizeof v);
    memcpy(&v, src, bytes <= sizeof v ? bytes : sizeof v);
    hipLaunchKernelGGL(poke_kernel, dim3(1), dim3(8), 0, st, static_cast<u32 *>(dst), v, (int)((bytes + 3) / 4));
    return hipGetLastError();
}

// Small results the host needs between launches (level counters, child lists, totals): copied into the context's pinned
// mailbox and read from there.  A copy into pageable memory (a stack variable, a std::vector) goes through the
// runtime's staging path, which costs tens of microseconds per call -- a count makes about ten of them.  Waits for
// the stream.
// up to 256 bytes go by a one-wave kernel that stores them into the mailbox and then raises a flag word there (system
// scope): the host polls the flag for a few tens of microseconds -- no completion signal, no wake-up: the copy + wait
// of a tiny result costs ~20 us through the runtime and a third of that this way -- and falls back to waiting for the
// stream when the work queued before it takes longer (so a backend does not burn a core through millisecond kernels).
constexpr size_t MAILBOX_FLAG_WORD = MAILBOX_BYTES / 8 - 1;    // the mailbox's last word
__global__ void mailbox_kernel(u64 *mailbox, const u32 *src, int n_words, u64 flag_word, u64 seq)
{
    u32 *dst = reinterpret_cast<u32 *>(mailbox);
    if ((int)threadIdx.x < n_words)
        __hip_atomic_store(&dst[threadIdx.x], src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&mailbox[flag_word], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static int read_back(dnagpu_ctx *ctx, void *host, const void *dev, size_t bytes)
{
    if (bytes == 0)
        return DNAGPU_OK;
    if (bytes <= 256 && (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(dev) & 3) == 0) {
        const u64 seq = ++ctx->mailbox_seq;
        hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->mailbox, static_cast<const u32 *>(dev),
                           (int)(bytes / 4), (u64)MAILBOX_FLAG_WORD, seq);
        HIP_TRY(hipGetLastError());
        volatile u64 *flag = ctx->mailbox + MAILBOX_FLAG_WORD;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (u32 spins = 0;; spins++) {
            if (*flag == seq) {
                seen = true;
                break;
            }
            if ((spins & 63) == 63 &&
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 60.0)
                break;
        }
        if (!seen)
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (*flag != seq) {
            set_err("mailbox: the result of a read-back did not arrive");
            return DNAGPU_ERR_INTERNAL;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        memcpy(host, ctx->mailbox, bytes);
        return DNAGPU_OK;
    }
    void *to = bytes <= MAILBOX_BYTES - 8 ? static_cast<void *>(ctx->mailbox) : host;
    HIP_TRY(hipMemcpyAsync(to, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (to != host)
        memcpy(host, to, bytes);
    return DNAGPU_OK;
}

struct dnagpu_dna {
    u64 *words;
    u64 n_words;
    u64 n_bases;
    bool owned;
    // a TABLE of sequences (dnagpu_dna_set_sequences): where every sequence starts, resident beside the packed stream
    u64 *seq_starts = nullptr;    // n_seqs + 1 offsets (pool memory)
    u32 *seq_marks = nullptr;     // one bit per base, set where a sequence starts (pool memory)
    u64 n_seqs = 0, n_mark_words = 0;
};

struct dnagpu_hist {
    u64 *keys;        // n_distinct groups, dense; stored leaf by leaf in completion order
    u32 *counts;      // a count never exceeds the 2^32 - 1 rows of one call: 4 bytes in HBM, widened on download
    u64 n_distinct;
    u64 total;
    // segment directory: leaf l (leaves are in ascending key order) = seg_cnt[l] groups at seg_off[l]
    u64 *seg_off;
    u32 *seg_cnt;
    u32 *seg_pre;     // exclusive scan of seg_cnt, built on first ordered download
    u32 n_segs;
    bool sorted;      // the segments are consecutive key ranges (true unless the super-k-mer engine made them)
    u64 extent;       // slots of keys / counts in use: n_distinct, or more when an unordered histogram holds count-0 padding
                      // between segments (0 = n_distinct)
    // A histogram made of several (dnagpu_hist_parts): the pipelined record exchange counts an owner's buckets group by
    // group, each group into arrays of its own.  The head then owns no arrays (keys == nullptr); n_distinct / total /
    // extent are the sums over the parts, the groups of part i come before those of part i + 1 in every ordered read.
    std::vector<dnagpu_hist *> parts;
};

// DNAGPU_DEBUG_POISON_POOL: no work buffer starts out zeroed (fresh hipMalloc memory) or holding a
// previous call's values (a recycled block); both hide reads of data the call never wrote
static int pool_poison(dnagpu_ctx *ctx, void *p, size_t bytes)
{
    if (ctx->debug_flags & DNAGPU_DEBUG_POISON_POOL)
        HIP_TRY(hipMemsetAsync(p, 0xFF, bytes, ctx->stream));
    return DNAGPU_OK;
}

static int pool_alloc(dnagpu_ctx *ctx, size_t bytes, void **out)
{
    if (bytes == 0)
        bytes = 256;
    const size_t asked = bytes;
    const bool guard = (ctx->debug_flags & DNAGPU_DEBUG_GUARD_POOL) != 0;
    bytes = (bytes + (guard ? POOL_GUARD : 0) + 255) & ~(size_t)255;
    // size classes (eight per power of two above 64 KB: at most 12.5 % more than asked): a work buffer whose size moves a
    // little from call to call (key ranges of the oversize buckets, sampled regions) finds the block of the call before
    // instead of a hipMalloc -- which costs milliseconds to a second for buffers of gigabytes
    if (bytes > ((size_t)1 << 16)) {
        const size_t g = (size_t)1 << (60 - __builtin_clzll((unsigned long long)bytes));
        bytes = (bytes + g - 1) & ~(g - 1);
    }
    const size_t want = bytes;
    auto finish = [&](PoolBlock &b) -> int {
        b.in_use = true;
        b.guard_at = 0;
        *out = b.ptr;
        RC_TRY(pool_poison(ctx, b.ptr, b.size));
        if (guard) {      // the band sits right behind the bytes asked for (a recycled block may be larger)
            HIP_TRY(hipMemsetAsync(static_cast<char *>(b.ptr) + asked, 0xA5, POOL_GUARD, ctx->stream));
            b.guard_at = asked;
        }
        return DNAGPU_OK;
    };
    int best = -1;
    for (size_t i = 0; i < ctx->pool.size(); i++) {
        PoolBlock &b = ctx->pool[i];
        if (!b.in_use && b.size >= want && b.size <= want * 2 + (1u << 20))
            if (best < 0 || b.size < ctx->pool[best].size)
                best = (int)i;
    }
    if (best >= 0)
        return finish(ctx->pool[best]);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
#ifdef DNAGPU_STAMPS
    fprintf(stderr, "[pool] hipMalloc %zu bytes (%zu blocks pooled)\n", want, ctx->pool.size());
#endif
    if (e != hipSuccess) {
        // give pooled-but-idle memory back and retry once
        (void)hipGetLastError();
        hipStreamSynchronize(ctx->stream);
        for (size_t i = 0; i < ctx->pool.size();) {
            if (!ctx->pool[i].in_use) {
                hipFree(ctx->pool[i].ptr);
                ctx->pool.erase(ctx->pool.begin() + i);
            } else {
                i++;
            }
        }
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_err("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return DNAGPU_ERR_OOM;
    }
    ctx->pool.push_back(PoolBlock{p, want, true});
    return finish(ctx->pool.back());
}

// DNAGPU_DEBUG_GUARD_POOL: every guard band must still hold its pattern (checked with the stream idle)
static int pool_check_guards(dnagpu_ctx *ctx)
{
    if (!(ctx->debug_flags & DNAGPU_DEBUG_GUARD_POOL))
        return DNAGPU_OK;
    unsigned char band[POOL_GUARD];
    for (const PoolBlock &b : ctx->pool) {
        if (!b.guard_at)
            continue;
        HIP_TRY(hipMemcpy(band, static_cast<const char *>(b.ptr) + b.guard_at, POOL_GUARD, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < POOL_GUARD; i++)
            if (band[i] != 0xA5) {
                set_err("pool guard: a kernel wrote %zu bytes past the end of a %zu-byte work buffer (%s)", i + 1, b.guard_at,
                        b.in_use ? "in use" : "already returned");
                return DNAGPU_ERR_INTERNAL;
            }
    }
    return DNAGPU_OK;
}

// Buffers go back to the pool while kernels that use them may still be queued: every later user is
// queued on the same stream, behind them.
static void pool_free(dnagpu_ctx *ctx, void *p)
{
    if (!p)
        return;
    for (PoolBlock &b : ctx->pool)
        if (b.ptr == p) {
            b.in_use = false;
            return;
        }
}

template <typename T>
static int pool_alloc_t(dnagpu_ctx *ctx, size_t n, T **out)
{
    void *p = nullptr;
    int rc = pool_alloc(ctx, n * sizeof(T), &p);
    *out = static_cast<T *>(p);
    return rc;
}

// frees a set of pool buffers at scope exit
struct PoolScope {
    dnagpu_ctx *ctx;
    std::vector<void *> ptrs;
    explicit PoolScope(dnagpu_ctx *c) : ctx(c) {}
    ~PoolScope()
    {
        for (void *p : ptrs)
            pool_free(ctx, p);
    }
    template <typename T>
    int alloc(size_t n, T **out)
    {
        int rc = pool_alloc_t(ctx, n, out);
        if (rc == DNAGPU_OK)
            ptrs.push_back(*out);
        return rc;
    }
    void release(void *p)   // hand ownership to the caller
    {
        ptrs.erase(std::remove(ptrs.begin(), ptrs.end(), p), ptrs.end());
    }
    void free_now(void *p)
    {
        release(p);
        pool_free(ctx, p);
    }
};

extern "C" int dnagpu_init(int device, dnagpu_ctx **out_ctx)
{
    return guarded([&]() -> int {
    if (!out_ctx)
        return DNAGPU_ERR_BAD_ARG;
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_err("hipGetDeviceCount: %s (%d devices)", hipGetErrorString(e), n);
        return DNAGPU_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_err("device %d out of range (%d devices)", device, n);
        return DNAGPU_ERR_NO_DEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    dnagpu_ctx *ctx = new (std::nothrow) dnagpu_ctx();
    if (!ctx)
        return DNAGPU_ERR_OOM;
    ctx->device = device;
    ctx->profiling = false;
    ctx->last_times.n = 0;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_err("hipStreamCreate: %s", hipGetErrorString(e));
        delete ctx;
        return DNAGPU_ERR_HIP;
    }
    ctx->mailbox = nullptr;
    ctx->debug_flags = 0;
    e = hipHostMalloc(reinterpret_cast<void **>(&ctx->mailbox), MAILBOX_BYTES, hipHostMallocDefault);
    if (e == hipSuccess)
        memset(ctx->mailbox, 0, MAILBOX_BYTES);
    if (e != hipSuccess) {
        set_err("hipHostMalloc: %s", hipGetErrorString(e));
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return DNAGPU_ERR_OOM;
    }
    *out_ctx = ctx;
    return DNAGPU_OK;
    });
}

extern "C" void dnagpu_destroy(dnagpu_ctx *ctx)
{
    if (!ctx)
        return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (PoolBlock &b : ctx->pool)
        hipFree(b.ptr);
    for (hipEvent_t e : ctx->ev)
        hipEventDestroy(e);
    hipStreamDestroy(ctx->stream);
    if (ctx->mailbox)
        hipHostFree(ctx->mailbox);
    delete ctx;
}

extern "C" int dnagpu_synchronize(dnagpu_ctx *ctx)
{
    return guarded([&]() -> int {
    if (!ctx)
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return pool_check_guards(ctx);                  // (DNAGPU_DEBUG_GUARD_POOL only)
    });
}

extern "C" void *dnagpu_stream(dnagpu_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int dnagpu_trim(dnagpu_ctx *ctx)
{
    return guarded([&]() -> int {
    if (!ctx)
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->pool.size();) {
        if (!ctx->pool[i].in_use) {
            hipFree(ctx->pool[i].ptr);
            ctx->pool.erase(ctx->pool.begin() + i);
        } else {
            i++;
        }
    }
    return DNAGPU_OK;
    });
}

extern "C" uint64_t dnagpu_device_bytes(dnagpu_ctx *ctx)
{
    uint64_t t = 0;
    if (ctx)
        for (PoolBlock &b : ctx->pool)
            t += b.size;
    return t;
}

extern "C" int dnagpu_buffer_alloc(dnagpu_ctx *ctx, uint64_t bytes, void **dev_ptr)
{
    return guarded([&]() -> int {
    if (!ctx || !dev_ptr)
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    return pool_alloc(ctx, (size_t)bytes, dev_ptr);
    });
}

extern "C" void dnagpu_buffer_free(dnagpu_ctx *ctx, void *dev_ptr)
{
    if (ctx)
        pool_free(ctx, dev_ptr);
}

extern "C" int dnagpu_buffer_download(dnagpu_ctx *ctx, const void *dev_ptr, uint64_t bytes, void *host)
{
    return guarded([&]() -> int {
    if (!ctx || (bytes && (!dev_ptr || !host)))
        return DNAGPU_ERR_BAD_ARG;
    if (bytes == 0)
        return DNAGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(host, dev_ptr, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_buffer_upload(dnagpu_ctx *ctx, void *dev_ptr, const void *host, uint64_t bytes)
{
    return guarded([&]() -> int {
    if (!ctx || (bytes && (!dev_ptr || !host)))
        return DNAGPU_ERR_BAD_ARG;
    if (bytes == 0)
        return DNAGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dev_ptr, host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DNAGPU_OK;
    });
}

// ------------------------------------------------------------------------------------------------
// profiling helpers
static void prof_begin(dnagpu_ctx *ctx)
{
    ctx->ev_names.clear();
}

static void prof_mark(dnagpu_ctx *ctx, const char *name)
{
    if (!ctx->profiling)
        return;
    size_t i = ctx->ev_names.size();
    if (i >= ctx->ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess)
            return;
        ctx->ev.push_back(e);
    }
    hipEventRecord(ctx->ev[i], ctx->stream);
    ctx->ev_names.push_back(name);
}

// names[i] labels the interval [mark i, mark i+1)
static void prof_end(dnagpu_ctx *ctx)
{
    ctx->last_times.n = 0;
    if (!ctx->profiling || ctx->ev_names.size() < 2)
        return;
    hipStreamSynchronize(ctx->stream);
    int n = 0;
    for (size_t i = 0; i + 1 < ctx->ev_names.size() && n < DNAGPU_MAX_PHASES; i++) {
        float ms = 0;
        hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]);
        // merge intervals that carry the same label (levels beyond the table of names)
        if (n > 0 && ctx->last_times.names[n - 1] == ctx->ev_names[i]) {
            ctx->last_times.ms[n - 1] += ms;
        } else {
            ctx->last_times.names[n] = ctx->ev_names[i];
            ctx->last_times.ms[n] = ms;
            n++;
        }
    }
    ctx->last_times.n = n;
}

extern "C" int dnagpu_set_debug(dnagpu_ctx *ctx, unsigned flags)
{
    if (!ctx)
        return DNAGPU_ERR_BAD_ARG;
    ctx->debug_flags = flags;
    return DNAGPU_OK;
}

extern "C" int dnagpu_last_phase_times(dnagpu_ctx *ctx, dnagpu_phase_times *out)
{
    return guarded([&]() -> int {
    if (!ctx || !out)
        return DNAGPU_ERR_BAD_ARG;
    *out = ctx->last_times;
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_set_profiling(dnagpu_ctx *ctx, int enabled)
{
    return guarded([&]() -> int {
    if (!ctx)
        return DNAGPU_ERR_BAD_ARG;
    ctx->profiling = enabled != 0;
    return DNAGPU_OK;
    });
}

// ------------------------------------------------------------------------------------------------
// dna
static u64 words_for(u64 n_bases) { return (n_bases + 31) / 32; }

extern "C" int dnagpu_dna_upload(dnagpu_ctx *ctx, const uint64_t *words, uint64_t n_bases, dnagpu_dna **out)
{
    return guarded([&]() -> int {
    if (!ctx || !out || (!words && n_bases))
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    u64 nw = words_for(n_bases);
    u64 *d = nullptr;
    RC_TRY(pool_alloc_t(ctx, (size_t)std::max<u64>(nw, 1), &d));
    if (nw) {
        hipError_t e = hipMemcpyAsync(d, words, nw * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            pool_free(ctx, d);
            set_err("upload: %s", hipGetErrorString(e));
            return DNAGPU_ERR_HIP;
        }
    }
    dnagpu_dna *h = new (std::nothrow) dnagpu_dna{d, nw, n_bases, true};
    if (!h) {
        pool_free(ctx, d);
        return DNAGPU_ERR_OOM;
    }
    *out = h;
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_dna_wrap(dnagpu_ctx *ctx, const uint64_t *dev_words, uint64_t n_words,
                               uint64_t n_bases, dnagpu_dna **out)
{
    return guarded([&]() -> int {
    if (!ctx || !out || (!dev_words && n_bases) || n_words < words_for(n_bases))
        return DNAGPU_ERR_BAD_ARG;
    dnagpu_dna *h = new (std::nothrow) dnagpu_dna{const_cast<u64 *>(dev_words), n_words, n_bases, false};
    if (!h)
        return DNAGPU_ERR_OOM;
    *out = h;
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_dna_synth(dnagpu_ctx *ctx, uint64_t seed, uint64_t n_bases, uint64_t motif_len,
                                dnagpu_dna **out)
{
    return guarded([&]() -> int {
    if (!ctx || !out)
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    u64 nw = words_for(n_bases);
    u64 *d = nullptr;
    RC_TRY(pool_alloc_t(ctx, (size_t)std::max<u64>(nw, 1), &d));
    hipError_t e = launch_synth(d, 0, nw, n_bases, seed, motif_len, ctx->stream);
    if (e != hipSuccess) {
        pool_free(ctx, d);
        set_err("synth: %s", hipGetErrorString(e));
        return DNAGPU_ERR_HIP;
    }
    dnagpu_dna *h = new (std::nothrow) dnagpu_dna{d, nw, n_bases, true};
    if (!h) {
        pool_free(ctx, d);
        return DNAGPU_ERR_OOM;
    }
    *out = h;
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_dna_download(dnagpu_ctx *ctx, const dnagpu_dna *dna, uint64_t *words)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !words)
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    u64 nw = words_for(dna->n_bases);
    if (nw) {
        HIP_TRY(hipMemcpyAsync(words, dna->words, nw * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_dna_pack(dnagpu_ctx *ctx, const char *text, uint64_t n_bases, int text_on_device,
                               dnagpu_dna **out, uint64_t *bad_pos, char *bad_char)
{
    return guarded([&]() -> int {
    if (!ctx || !out)
        return DNAGPU_ERR_BAD_ARG;
    if (n_bases == 0)
        return DNAGPU_ERR_DNA_EMPTY;
    if (!text)
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    PoolScope ps(ctx);
    const unsigned char *dtext = reinterpret_cast<const unsigned char *>(text);
    if (!text_on_device) {
        unsigned char *stage = nullptr;
        RC_TRY(ps.alloc((size_t)n_bases, &stage));
        HIP_TRY(hipMemcpyAsync(stage, text, n_bases, hipMemcpyHostToDevice, ctx->stream));
        dtext = stage;
    }
    u64 nw = words_for(n_bases);
    u64 *words = nullptr, *bad = nullptr;
    RC_TRY(ps.alloc((size_t)nw, &words));
    RC_TRY(ps.alloc(1, &bad));
    HIP_TRY(hipMemsetAsync(bad, 0xff, 8, ctx->stream));
    HIP_TRY(launch_pack(dtext, n_bases, words, bad, ctx->stream));
    u64 hbad = 0;
    HIP_TRY(hipMemcpyAsync(&hbad, bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (hbad != ~(u64)0) {
        char c = 0;
        HIP_TRY(hipMemcpy(&c, dtext + hbad, 1, hipMemcpyDeviceToHost));
        if (bad_pos) *bad_pos = hbad;
        if (bad_char) *bad_char = c;
        set_err("Invalid character in DNA sequence: %c", c);
        return DNAGPU_ERR_DNA_INVALID_CHAR;
    }
    dnagpu_dna *h = new (std::nothrow) dnagpu_dna{words, nw, n_bases, true};
    if (!h)
        return DNAGPU_ERR_OOM;
    ps.release(words);
    *out = h;
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_dna_unpack(dnagpu_ctx *ctx, const dnagpu_dna *dna, uint64_t first, uint64_t count,
                                 char *out_text, int out_on_device)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || (count && !out_text))
        return DNAGPU_ERR_BAD_ARG;
    if (first > dna->n_bases || count > dna->n_bases - first)
        return DNAGPU_ERR_BAD_ARG;
    if (count == 0)
        return DNAGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    PoolScope ps(ctx);
    unsigned char *dt = reinterpret_cast<unsigned char *>(out_text);
    if (!out_on_device)
        RC_TRY(ps.alloc((size_t)count, &dt));
    HIP_TRY(launch_unpack(dna->words, first, count, dt, ctx->stream));
    if (!out_on_device)
        HIP_TRY(hipMemcpyAsync(out_text, dt, count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DNAGPU_OK;
    });
}

extern "C" uint64_t dnagpu_dna_wire_size(uint64_t n_bases) { return 8 + 8 * words_for(n_bases); }

extern "C" int dnagpu_dna_from_wire(dnagpu_ctx *ctx, const void *wire, uint64_t wire_bytes, int wire_on_device,
                                    dnagpu_dna **out)
{
    return guarded([&]() -> int {
    if (!ctx || !wire || !out || wire_bytes < 8)
        return DNAGPU_ERR_BAD_ARG;
    if (wire_on_device && (reinterpret_cast<uintptr_t>(wire) & 7))
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned char hdr[8];
    if (wire_on_device) {
        HIP_TRY(hipMemcpyAsync(hdr, wire, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    } else {
        memcpy(hdr, wire, 8);
    }
    u64 n_bases = 0;
    for (int i = 0; i < 8; i++)
        n_bases = (n_bases << 8) | hdr[i];                  // pq_getmsgint64: network byte order (dna.c:251)
    if (n_bases == 0)
        return DNAGPU_ERR_DNA_EMPTY;
    if (n_bases > ((u64)1 << 40) || wire_bytes != dnagpu_dna_wire_size(n_bases))
        return DNAGPU_ERR_BAD_ARG;
    const u64 nw = words_for(n_bases);
    u64 *d = nullptr;
    RC_TRY(pool_alloc_t(ctx, (size_t)nw, &d));
    int rc = DNAGPU_OK;
    {
        PoolScope ps(ctx);
        const u64 *src = reinterpret_cast<const u64 *>(static_cast<const unsigned char *>(wire) + 8);
        u64 *stage = nullptr;
        hipError_t e = hipSuccess;
        if (!wire_on_device) {
            rc = ps.alloc((size_t)nw, &stage);
            if (rc == DNAGPU_OK)
                e = hipMemcpyAsync(stage, src, nw * 8, hipMemcpyHostToDevice, ctx->stream);
            src = stage;
        }
        const u64 last_mask = (n_bases % 32) ? (((u64)1 << (2 * (n_bases % 32))) - 1) : ~(u64)0;
        if (rc == DNAGPU_OK && e == hipSuccess)
            e = launch_wire_swap(src, d, nw, last_mask, ctx->stream);
        if (rc == DNAGPU_OK && e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (rc == DNAGPU_OK && e != hipSuccess) {
            set_err("from_wire: %s", hipGetErrorString(e));
            rc = DNAGPU_ERR_HIP;
        }
    }
    dnagpu_dna *h = rc == DNAGPU_OK ? new (std::nothrow) dnagpu_dna{d, nw, n_bases, true} : nullptr;
    if (!h) {
        pool_free(ctx, d);
        return rc == DNAGPU_OK ? DNAGPU_ERR_OOM : rc;
    }
    *out = h;
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_dna_to_wire(dnagpu_ctx *ctx, const dnagpu_dna *dna, void *wire, uint64_t wire_cap,
                                  int wire_on_device)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !wire || wire_cap < dnagpu_dna_wire_size(dna->n_bases))
        return DNAGPU_ERR_BAD_ARG;
    if (wire_on_device && (reinterpret_cast<uintptr_t>(wire) & 7))
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned char hdr[8];
    for (int i = 0; i < 8; i++)
        hdr[i] = (unsigned char)(dna->n_bases >> (56 - 8 * i));     // pq_sendint64 (dna.c:282, as an int64)
    const u64 nw = words_for(dna->n_bases);
    PoolScope ps(ctx);
    u64 *dst = reinterpret_cast<u64 *>(static_cast<unsigned char *>(wire) + 8);
    u64 *stage = nullptr;
    if (!wire_on_device) {
        RC_TRY(ps.alloc((size_t)std::max<u64>(nw, 1), &stage));
        memcpy(wire, hdr, 8);
    } else {
        HIP_TRY(hipMemcpyAsync(wire, hdr, 8, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(launch_wire_swap(dna->words, wire_on_device ? dst : stage, nw, ~(u64)0, ctx->stream));
    if (!wire_on_device && nw)
        HIP_TRY(hipMemcpyAsync(dst, stage, nw * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_kmers_to_text(dnagpu_ctx *ctx, const uint64_t *keys, uint64_t n, int k, char *out_text,
                                    int on_device)
{
    return guarded([&]() -> int {
    if (!ctx || (n && (!keys || !out_text)))
        return DNAGPU_ERR_BAD_ARG;
    if (k <= 0 || k > 32)
        return DNAGPU_ERR_INVALID_K;
    if (n == 0)
        return DNAGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    PoolScope ps(ctx);
    const u64 *dk = keys;
    unsigned char *dt = reinterpret_cast<unsigned char *>(out_text);
    if (!on_device) {
        u64 *tk = nullptr;
        RC_TRY(ps.alloc((size_t)n, &tk));
        RC_TRY(ps.alloc((size_t)n * (k + 1), &dt));
        HIP_TRY(hipMemcpyAsync(tk, keys, n * 8, hipMemcpyHostToDevice, ctx->stream));
        dk = tk;
    }
    HIP_TRY(launch_kmers_to_text(dk, n, k, dt, ctx->stream));
    if (!on_device)
        HIP_TRY(hipMemcpyAsync(out_text, dt, n * (u64)(k + 1), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DNAGPU_OK;
    });
}

extern "C" uint64_t dnagpu_dna_length(const dnagpu_dna *dna) { return dna ? dna->n_bases : 0; }
extern "C" const uint64_t *dnagpu_dna_device_words(const dnagpu_dna *dna) { return dna ? dna->words : nullptr; }

extern "C" void dnagpu_dna_free(dnagpu_ctx *ctx, dnagpu_dna *dna)
{
    if (!dna)
        return;
    if (dna->owned && ctx)
        pool_free(ctx, dna->words);
    if (ctx && dna->seq_starts)
        pool_free(ctx, dna->seq_starts);
    if (ctx && dna->seq_marks)
        pool_free(ctx, dna->seq_marks);
    delete dna;
}

// ------------------------------------------------------------------------------------------------
// generate_kmers
extern "C" int dnagpu_kmer_count(uint64_t n_bases, int k, uint64_t *n_kmers)
{
    return guarded([&]() -> int {
    if (k <= 0 || k > 32)                      // dna.c:772
        return DNAGPU_ERR_INVALID_K;
    if (!n_kmers)
        return DNAGPU_ERR_BAD_ARG;
    *n_kmers = n_bases >= (u64)k ? n_bases - (u64)k + 1 : 0;   // dna.c:781 without the underflow
    return DNAGPU_OK;
    });
}

// validates [first, first+count) against the row count of generate_kmers(dna, k)
static int check_range(const dnagpu_dna *dna, int k, u64 first, u64 count)
{
    u64 total = 0;
    RC_TRY(dnagpu_kmer_count(dna->n_bases, k, &total));
    if (first > total || count > total - first) {
        set_err("rows [%llu, +%llu) outside generate_kmers' %llu rows", (unsigned long long)first,
                (unsigned long long)count, (unsigned long long)total);
        return DNAGPU_ERR_BAD_ARG;
    }
    return DNAGPU_OK;
}

extern "C" int dnagpu_generate_kmers(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, uint64_t first,
                                     uint64_t count, uint64_t *out_keys, int out_on_device)
{
    return guarded([&]() -> int {
    if (!ctx || !dna)
        return DNAGPU_ERR_BAD_ARG;
    RC_TRY(check_range(dna, k, first, count));
    if (count == 0)
        return DNAGPU_OK;
    if (!out_keys)
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (out_on_device) {
        HIP_TRY(launch_extract(dna->words, dna->n_words, first, count, k, out_keys, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return DNAGPU_OK;
    }
    // host destination: batches through a device staging buffer, so a PostgreSQL caller can
    // CHECK_FOR_INTERRUPTS between batches by asking for windows itself
    PoolScope ps(ctx);
    const u64 BATCH = (u64)1 << 26;            // 64 Mi keys = 512 MiB staging
    u64 *stage = nullptr;
    RC_TRY(ps.alloc((size_t)std::min(count, BATCH), &stage));
    for (u64 done = 0; done < count; done += BATCH) {
        u64 nb = std::min(BATCH, count - done);
        HIP_TRY(launch_extract(dna->words, dna->n_words, first + done, nb, k, stage, ctx->stream));
        HIP_TRY(hipMemcpyAsync(out_keys + done, stage, nb * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return DNAGPU_OK;
    });
}

// ------------------------------------------------------------------------------------------------
// filters: reference operator semantics -> FilterDev
// IUPAC sets as 4-bit masks over codes (bit0 = A, bit1 = T, bit2 = C, bit3 = G), dna.c:1064-1081.
static int iupac_set(char c)
{
    switch (c) {
    case 'A': return 0x1;
    case 'T': return 0x2;
    case 'C': return 0x4;
    case 'G': return 0x8;
    case 'U': return 0x0;   // compares the decoded base with 'U': never true (dna.c:1070)
    case 'W': return 0x3;
    case 'S': return 0xC;
    case 'M': return 0x5;
    case 'K': return 0xA;
    case 'R': return 0x9;
    case 'Y': return 0x6;
    case 'B': return 0xE;
    case 'D': return 0xB;
    case 'H': return 0x7;
    case 'V': return 0xD;
    case 'N': return 0xF;
    }
    return -1;
}

// has_rows: the reference raises operator errors only when the operator is actually evaluated
static int build_filter(const dnagpu_filter *f, int k, bool has_rows, FilterDev *out)
{
    if (!f)
        return DNAGPU_ERR_BAD_ARG;
    FilterDev d;
    memset(&d, 0, sizeof d);
    switch (f->kind) {
    case DNAGPU_FILTER_EQUALS:
        // kmer_eq_internal: lengths must be equal, then bits (dna.c:655-668)
        if (f->length != k) {
            d.and_mask = 0;
            d.eq_value = 1;                     // (key & 0) == 1: matches nothing
        } else {
            d.and_mask = ~(u64)0;
            d.eq_value = f->bits;
        }
        break;
    case DNAGPU_FILTER_STARTS_WITH:
        if (f->length < 0)
            return DNAGPU_ERR_BAD_ARG;
        if (f->length > k) {                    // dna.c:854-856
            if (has_rows)
                return DNAGPU_ERR_PREFIX_TOO_LONG;
            d.and_mask = 0;
            d.eq_value = 1;
            break;
        }
        d.and_mask = kmer_mask(f->length);      // dna.c:862, defined for length 32 too
        if (f->length == 0)
            d.and_mask = 0;
        d.eq_value = f->bits;
        break;
    case DNAGPU_FILTER_CONTAINS: {
        size_t len = strnlen(f->pattern, sizeof f->pattern);
        if (len == 0 || len > 32)               // dna.c:877-886
            return DNAGPU_ERR_QKMER_INVALID;
        for (size_t i = 0; i < len; i++)
            if (iupac_set(f->pattern[i]) < 0)   // dna.c:888-896
                return DNAGPU_ERR_QKMER_INVALID;
        if ((int)len != k) {                    // dna.c:1106-1108
            if (has_rows)
                return DNAGPU_ERR_QKMER_LEN_MISMATCH;
            d.and_mask = 0;
            d.eq_value = 1;
            break;
        }
        d.use_planes = 1;
        for (size_t i = 0; i < len; i++) {
            int set = iupac_set(f->pattern[i]);
            for (int c = 0; c < 4; c++)
                if (!(set & (1 << c)))
                    d.deny[c] |= (u64)1 << (2 * i);
        }
        break;
    }
    default:
        return DNAGPU_ERR_BAD_ARG;
    }
    *out = d;
    return DNAGPU_OK;
}

// The same operator as per-position sets for the bit-sliced stream kernels.  *none: no row can match
// (an `=` of another length, stray bits behind the right-hand kmer's length, a 'U' in the pattern).
static int build_filter_bits(const dnagpu_filter *f, int k, bool has_rows, FilterBits *out, bool *none)
{
    FilterDev fd;
    RC_TRY(build_filter(f, k, has_rows, &fd));          // argument checks and the reference's ERRORs
    FilterBits fb;
    for (int q = 0; q < 4; q++)
        fb.sets[q] = 0xFFFFFFFFu;                        // N everywhere
    fb.k = k;
    *none = false;
    auto put = [&](int i, u32 set) { fb.sets[i >> 3] = (fb.sets[i >> 3] & ~(15u << ((i & 7) * 4))) | (set << ((i & 7) * 4)); };
    switch (f->kind) {
    case DNAGPU_FILTER_EQUALS:
    case DNAGPU_FILTER_STARTS_WITH: {
        const int len = f->length;
        if (fd.and_mask == 0 && fd.eq_value != 0) {      // build_filter's "matches nothing", or an empty prefix with bits
            *none = true;
            break;
        }
        if (len < 32 && len >= 0 && (f->bits >> (2 * len)) != 0) {
            *none = true;                                // bits behind the right-hand kmer's own length never compare equal
            break;
        }
        for (int i = 0; i < len && i < k; i++)
            put(i, 1u << ((f->bits >> (2 * i)) & 3));
        break;
    }
    case DNAGPU_FILTER_CONTAINS: {
        size_t len = strnlen(f->pattern, sizeof f->pattern);
        if ((int)len != k) {                             // only reachable without rows
            *none = true;
            break;
        }
        for (int i = 0; i < k; i++) {
            int set = iupac_set(f->pattern[i]);
            if (set == 0)
                *none = true;                            // 'U' matches no base (dna.c:1070)
            put(i, (u32)set);
        }
        break;
    }
    }
    *out = fb;
    return DNAGPU_OK;
}

extern "C" int dnagpu_generate_kmers_filtered(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k,
                                              const dnagpu_filter *filter, uint64_t first, uint64_t count,
                                              uint64_t *out_keys, uint64_t *out_pos, uint64_t cap,
                                              uint64_t *n_out, int out_on_device)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !n_out)
        return DNAGPU_ERR_BAD_ARG;
    RC_TRY(check_range(dna, k, first, count));
    FilterBits fb;
    bool none = false;
    RC_TRY(build_filter_bits(filter, k, count > 0, &fb, &none));
    *n_out = 0;
    if (count == 0 || none)
        return DNAGPU_OK;
    if (count > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    HIP_TRY(hipSetDevice(ctx->device));
    PoolScope ps(ctx);
    u32 n_groups = 0, tpg = 0;
    filter_bits_geometry(count, &n_groups, &tpg);
    u32 *group_counts = nullptr;
    RC_TRY(ps.alloc((size_t)n_groups, &group_counts));
    prof_begin(ctx);
    prof_mark(ctx, "filter_count");
    HIP_TRY(launch_filter_bits_count(dna->words, dna->n_words, first, count, fb, group_counts, ctx->stream));
    const bool want = cap > 0 && (out_keys || out_pos);
    if (want && out_on_device) {
        // both sweeps queued back to back; the total arrives in the pinned mailbox
        prof_mark(ctx, "filter_write");
        HIP_TRY(launch_filter_bits_write(dna->words, dna->n_words, first, count, k, fb, group_counts, out_keys,
                                         out_pos, cap, ctx->mailbox, ctx->stream));
        prof_mark(ctx, "end");
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        prof_end(ctx);
        *n_out = ctx->mailbox[0];
        return DNAGPU_OK;
    }
    static_assert(MAILBOX_BYTES >= FILTER_MAX_GROUPS * sizeof(u32), "mailbox holds one count per group");
    u32 *hc = reinterpret_cast<u32 *>(ctx->mailbox);
    HIP_TRY(hipMemcpyAsync(hc, group_counts, (size_t)n_groups * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    u64 total = 0;
    for (u32 g = 0; g < n_groups; g++)
        total += hc[g];
    *n_out = total;
    u64 nwrite = std::min<u64>(total, cap);
    if (nwrite == 0 || !want)
        return DNAGPU_OK;
    u64 *dk = nullptr, *dp = nullptr;
    if (out_keys)
        RC_TRY(ps.alloc((size_t)nwrite, &dk));
    if (out_pos)
        RC_TRY(ps.alloc((size_t)nwrite, &dp));
    HIP_TRY(launch_filter_bits_write(dna->words, dna->n_words, first, count, k, fb, group_counts, dk, dp, nwrite,
                                     nullptr, ctx->stream));
    if (out_keys)
        HIP_TRY(hipMemcpyAsync(out_keys, dk, nwrite * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_pos)
        HIP_TRY(hipMemcpyAsync(out_pos, dp, nwrite * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DNAGPU_OK;
    });
}

// ------------------------------------------------------------------------------------------------
// batched operators
extern "C" int dnagpu_kmer_hash(dnagpu_ctx *ctx, const uint64_t *keys, uint64_t n, uint32_t *out, int on_device)
{
    return guarded([&]() -> int {
    if (!ctx || (n && (!keys || !out)))
        return DNAGPU_ERR_BAD_ARG;
    if (n == 0)
        return DNAGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (on_device) {
        HIP_TRY(launch_hash_batch(keys, n, out, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return DNAGPU_OK;
    }
    PoolScope ps(ctx);
    u64 *dk = nullptr;
    u32 *dh = nullptr;
    RC_TRY(ps.alloc((size_t)n, &dk));
    RC_TRY(ps.alloc((size_t)n, &dh));
    HIP_TRY(hipMemcpyAsync(dk, keys, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_hash_batch(dk, n, dh, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, dh, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_kmer_match(dnagpu_ctx *ctx, const uint64_t *keys, uint64_t n, int k,
                                 const dnagpu_filter *filter, uint8_t *flags, int on_device)
{
    return guarded([&]() -> int {
    if (!ctx || (n && (!keys || !flags)))
        return DNAGPU_ERR_BAD_ARG;
    if (k <= 0 || k > 32)
        return DNAGPU_ERR_INVALID_K;
    FilterDev fd;
    RC_TRY(build_filter(filter, k, n > 0, &fd));
    if (n == 0)
        return DNAGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (on_device) {
        HIP_TRY(launch_match_batch(keys, n, fd, flags, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return DNAGPU_OK;
    }
    PoolScope ps(ctx);
    u64 *dk = nullptr;
    uint8_t *df = nullptr;
    RC_TRY(ps.alloc((size_t)n, &dk));
    RC_TRY(ps.alloc((size_t)n, &df));
    HIP_TRY(hipMemcpyAsync(dk, keys, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_match_batch(dk, n, fd, df, ctx->stream));
    HIP_TRY(hipMemcpyAsync(flags, df, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DNAGPU_OK;
    });
}

// ------------------------------------------------------------------------------------------------
// GROUP BY kmer, count(*): the level loop
static const char *const LEVEL_HIST_NAMES[] = {"level0_hist", "level1_hist", "level2_hist", "level3_hist", "levelN_hist"};
static const char *const LEVEL_PREFIX_NAMES[] = {"level0_prefix", "level1_prefix", "level2_prefix", "level3_prefix", "levelN_prefix"};
static const char *const LEVEL_SCATTER_NAMES[] = {"level0_scatter", "level1_scatter", "level2_scatter", "level3_scatter", "levelN_scatter"};
static const char *const LEVEL_PLAN_NAMES[] = {"level0_plan", "level1_plan", "level2_plan", "level3_plan", "levelN_plan"};

struct TreeResult {
    Node *nodes;      // final node list (leaves, or the children of a forced level)
    u32 n_nodes;
    u32 n_big;        // leaves that sort more than LEAF_CAP_SMALL keys among them
    u32 n_small;      // leaves that sort up to LEAF_CAP_SMALL keys
    u32 n_tiny;       // leaves that sort at most LEAF_CAP_TINY keys (the rest: single-key or empty nodes)
    u64 n_keys;       // keys in the tree (== n unless an owner filter dropped some at the dna root)
    u64 *buf0;
    u64 *buf1;        // may be null if never needed
};

// Runs levels until every node is a leaf (force_bits == 0), or exactly one forced level of
// `force_bits` bits on the root (force_bits > 0).  dna != null: root over the packed sequence
// (keys land in buf0, allocated here); else root over keys_in (used as buf0).
// init_nodes != null: the levels start at `start_level` from that node list over keys_in (pool memory of `ps`; the
// nodes' key ranges need not share key bits: the super-k-mer engine enters here with its bucket nodes)
static int run_tree(dnagpu_ctx *ctx, PoolScope &ps, const dnagpu_dna *dna, u64 first, u64 n, int k,
                    u64 *keys_in, int force_bits, TreeResult *res, int fixed_bits = 0, u64 fixed_prefix = 0,
                    bool single_level = true, u32 flt_lo = 0, u32 flt_span = ~0u, u32 flt_tb = 0,
                    Node *init_nodes = nullptr, u32 init_n = 0, int start_level = 0)
{
    hipStream_t st = ctx->stream;
    u64 *buf0 = keys_in, *buf1 = nullptr;
    Node root;
    memset(&root, 0, sizeof root);
    root.start = 0;
    root.len = (u32)n;
    root.meta = (u32)(2 * k - fixed_bits);      // bits every key is known to share are not split on
    root.prefix = fixed_prefix;
    bool src_dna = false;
    u64 n_keys = n;
    if (dna) {
        if (n <= (u64)LEAF_CAP && force_bits == 0) {
            RC_TRY(ps.alloc((size_t)n, &buf0));
            HIP_TRY(launch_extract(dna->words, dna->n_words, first, n, k, buf0, st));
        } else {
            src_dna = true;                 // buffer 0 is allocated once level 0 knows how many keys it keeps
            root.meta |= NODE_BUF;          // children of the dna root go to buffer 0
        }
    }
    Node *cur = init_nodes;
    u32 n_nodes = init_nodes ? init_n : 1u, n_big = 0, n_small = 0, n_tiny = 0;
    if (!init_nodes) {
        RC_TRY(ps.alloc(1, &cur));
        static_assert(sizeof(Node) == 32, "a node travels as one kernel argument");
        HIP_TRY(poke(cur, &root, sizeof root, st));
    }
    u32 n_nonempty = 1;           // nodes of the current level that uniform data would fill (all, or an owner's share)

    static u64 chunk_target = 0;                 // chunks per level (work units of the hist/scatter kernels)
    if (chunk_target == 0) {
        const char *e = diag_env("DNAGPU_CHUNKS");       // experiment switch: diagnostic build (make STAMPS=1) only
        chunk_target = e ? (u64)atoll(e) : 4096;
        if (chunk_target < 256)
            chunk_target = 256;
    }
    u32 chunk_len = (u32)std::max<u64>(4 * (u64)scatter_tile_keys(), (n + chunk_target - 1) / chunk_target);
    chunk_len = (chunk_len + scatter_tile_keys() - 1) / scatter_tile_keys() * scatter_tile_keys();

    for (int level = start_level;; level++) {
        const int li = std::min(level, 4);
        prof_mark(ctx, LEVEL_PLAN_NAMES[li]);
        u32 *outc = nullptr, *nch = nullptr, *scan_tmp = nullptr;
        LevelCounters *ctr = nullptr;
        RC_TRY(ps.alloc(n_nodes, &outc));
        RC_TRY(ps.alloc(n_nodes, &nch));
        RC_TRY(ps.alloc((size_t)scan_tmp_words(n_nodes), &scan_tmp));
        RC_TRY(ps.alloc(1, &ctr));
        HIP_TRY(launch_plan_level(cur, n_nodes, (force_bits > 0 && level == 0) ? -force_bits : level, chunk_len, outc, nch,
                                  scan_tmp, ctr, st, init_nodes ? start_level + 1 : 2));
        LevelCounters hc;
        RC_TRY(read_back(ctx, &hc, ctr, sizeof hc));
        if (hc.n_split == 0) {
            n_big = hc.n_big;                    // every node of the final list was planned (and counted) here
            n_small = hc.n_small;
            n_tiny = hc.n_tiny;
            ps.free_now(outc);
            ps.free_now(nch);
            ps.free_now(scan_tmp);
            ps.free_now(ctr);
            break;
        }
        u32 n_nonempty_next = 0;
        Chunk *chunks = nullptr;
        u32 *hist = nullptr, *tot = nullptr;
        Node *next = nullptr;
        RC_TRY(ps.alloc(hc.n_chunks, &chunks));
        RC_TRY(ps.alloc((size_t)hc.n_chunks * ROW_STRIDE, &hist));
        RC_TRY(ps.alloc((size_t)hc.n_chunks * ROW_STRIDE, &tot));
        RC_TRY(ps.alloc(hc.n_next, &next));
        // key-source levels: level_hist also finds, per node, how many low key bits vary (see level_children)
        // (levels 0-1: only for nodes a quarter or more above the level's mean size -- none in uniform data, whose
        // level-1 histogram then costs 4.05 instead of 4.3 ms at 3 Gbase; deeper: every node)
        const u32 stat_min_len = level >= 2 ? 0u : (u32)std::min<u64>((n_keys / n_nonempty + 1) * 5 / 4, 0xffffffffull);
        u32 *vary = nullptr;
        if (!src_dna && !(force_bits > 0 && level == 0)) {
            RC_TRY(ps.alloc((size_t)n_nodes * NODE_STAT_WORDS, &vary));
            HIP_TRY(hipMemsetAsync(vary, 0, (size_t)n_nodes * NODE_STAT_WORDS * sizeof(u32), st));
        }
        HIP_TRY(launch_fill_chunks(cur, n_nodes, chunk_len, outc, nch, cur, chunks, st));
        prof_mark(ctx, LEVEL_HIST_NAMES[li]);
        HIP_TRY(launch_level_hist(cur, chunks, hc.n_chunks, src_dna, dna ? dna->words : nullptr,
                                  dna ? dna->n_words : 0, first, k, buf0, buf1, hist, src_dna ? flt_lo : 0u,
                                  src_dna ? flt_span : ~0u, src_dna ? flt_tb : 0u, vary, level >= 2 ? 1 : 0, stat_min_len, st));
        prof_mark(ctx, LEVEL_PREFIX_NAMES[li]);
        HIP_TRY(launch_level_prefix(cur, chunks, hc.n_chunks, hc.n_split, chunk_len, hist, tot, st, n_nodes));
        HIP_TRY(launch_level_children(cur, n_nodes, tot, next, vary, buf0, buf1, stat_min_len, st));
        if (src_dna) {
            // the dna root's children say how many keys survive the owner filter
            std::vector<Node> kids(hc.n_next);
            RC_TRY(read_back(ctx, kids.data(), next, (size_t)hc.n_next * sizeof(Node)));
            n_keys = 0;
            for (const Node &c : kids)
                n_keys += c.len;
            if (flt_span != ~0u && flt_span > 0 && flt_span < hc.n_next)
                n_nonempty_next = flt_span;       // an owner's digits: the other children are empty by construction
            RC_TRY(ps.alloc((size_t)std::max<u64>(n_keys, 1), &buf0));
        }
        if (hc.n_scatter) {
            if (!src_dna && !buf1)
                RC_TRY(ps.alloc((size_t)std::max<u64>(n_keys, 1), &buf1));
            prof_mark(ctx, LEVEL_SCATTER_NAMES[li]);
            HIP_TRY(launch_level_scatter(cur, chunks, hc.n_chunks, src_dna, dna ? dna->words : nullptr,
                                         dna ? dna->n_words : 0, first, k, buf0, buf1, hist, tot,
                                         src_dna ? flt_lo : 0u, src_dna ? flt_span : ~0u, src_dna ? flt_tb : 0u, hc.max_bits, st));
            if (vary)                   // nodes dominated by one key: three-way split around it (returns at once if none)
                HIP_TRY(launch_peel_scatter(cur, n_nodes, chunks, hc.n_chunks, next, buf0, buf1, vary, st));
        }
        ps.free_now(outc);
        ps.free_now(nch);
        ps.free_now(scan_tmp);
        ps.free_now(ctr);
        ps.free_now(chunks);
        ps.free_now(hist);
        ps.free_now(tot);
        if (vary)
            ps.free_now(vary);
        ps.free_now(cur);
        cur = next;
        n_nodes = hc.n_next;
        n_nonempty = n_nonempty_next ? n_nonempty_next : (n_nodes ? n_nodes : 1u);
        src_dna = false;
        if (force_bits > 0 && single_level)
            break;
    }
    res->nodes = cur;
    res->n_nodes = n_nodes;
    res->n_big = n_big;
    res->n_small = n_small;
    res->n_tiny = n_tiny;
    res->n_keys = n_keys;
    res->buf0 = buf0;
    res->buf1 = buf1;
    return DNAGPU_OK;
}

// ---- super-k-mer engine (superkmer_kernels.hip): the partition passes move 16-byte records of ~9 k-mers
// instead of 8-byte keys, and a final bucket is counted from its records in an LDS hash table: no key of it is
// ever written to HBM.
// DNAGPU_SK_SKEWED: a bucket is too heavy (low-complexity input): the caller counts with the ordinary tree
// instead, which has the skew paths.
constexpr int DNAGPU_SK_SKEWED = -1;
constexpr u64 SK_LEAF_MEAN = 2500;               // planned k-mers per final bucket: ~770 quads of four k-mers -- 1024 (sk_count's threads: the buckets with copies) is 4 sigma above, so next to no bucket takes the expansion path (A/B on one box, 3 Gbase: 2700 18.7 - 18.8 ms, 2500 18.2, 2300 18.1 - 18.4)
// A mid bucket of more than SK_MID_LIMIT k-mers (planned: 16 x SK_LEAF_MEAN) is "heavy" and leaves the record path for the
// expansion; below that it is regrouped like the others, and its long final buckets (thousands to millions of copies of a
// few k-mers) are what sk_count_big is for.  Final buckets beyond SK_BIG_LIMIT k-mers are expanded without trying.
constexpr u64 SK_MID_LIMIT = (u64)1 << 27;
// (a mid bucket is regrouped by ONE workgroup, tile after tile, twice: beyond eight tiles the chunked split below, many
// workgroups per bucket, is faster -- 249 Mbase of a tiled 1000-base motif: sk_regroup 0.64 ms at 2^19, sk_heavy_split 0.31 at 2^16)
#ifndef SK_MID_RECORDS_LOG2
#define SK_MID_RECORDS_LOG2 16
#endif
constexpr u32 SK_MID_RECORDS = 1u << SK_MID_RECORDS_LOG2;
constexpr u64 SK_BIG_LIMIT = 0xFFFFFFFFull;
// Level 1 splits a coarse bucket 512 ways, not 1024: a tile of 8192 records then leaves in runs of 16 records (256
// bytes) instead of 8 -- sk_scatter1 3.6 - 3.9 instead of 4.9 - 5.2 ms at 3 Gbase (A/B on one box) -- and level 0 takes
// the bit over (136 coarse buckets at 3 Gbase: its 16-byte stores still combine in L2, 2.2 MB of open lines per XCD).
constexpr int SK_B1_MAX = 9;

struct SkLevel {                                 // what one forced partition level leaves behind
    Node *next;
    u32 n_next;
    u32 *hist, *tot;
    Chunk *chunks;
    u32 n_chunks;
};

// plan (forced split on `bits` bits) + chunk list + histogram tables of one level over `cur`
static int sk_level_begin(dnagpu_ctx *ctx, PoolScope &ps, Node *cur, u32 n_nodes, int bits, u32 chunk_len, SkLevel *lv)
{
    hipStream_t st = ctx->stream;
    u32 *outc = nullptr, *nch = nullptr, *scan_tmp = nullptr;
    LevelCounters *ctr = nullptr;
    RC_TRY(ps.alloc(n_nodes, &outc));
    RC_TRY(ps.alloc(n_nodes, &nch));
    RC_TRY(ps.alloc((size_t)scan_tmp_words(n_nodes), &scan_tmp));
    RC_TRY(ps.alloc(1, &ctr));
    HIP_TRY(launch_plan_level(cur, n_nodes, -bits, chunk_len, outc, nch, scan_tmp, ctr, st));
    LevelCounters hc;
    RC_TRY(read_back(ctx, &hc, ctr, sizeof hc));
    lv->n_next = hc.n_next;
    lv->n_chunks = hc.n_chunks;
    RC_TRY(ps.alloc(std::max<u32>(hc.n_chunks, 1), &lv->chunks));
    RC_TRY(ps.alloc((size_t)std::max<u32>(hc.n_chunks, 1) * ROW_STRIDE, &lv->hist));
    RC_TRY(ps.alloc((size_t)std::max<u32>(hc.n_chunks, 1) * ROW_STRIDE, &lv->tot));
    RC_TRY(ps.alloc(std::max<u32>(hc.n_next, 1), &lv->next));
    HIP_TRY(launch_fill_chunks(cur, n_nodes, chunk_len, outc, nch, cur, lv->chunks, st));
    // (outc / nch / scan_tmp / ctr go back to the pool when the scope ends: later users queue behind this stream)
    return DNAGPU_OK;
}

// The three partition levels.  On success: *recs = the record buffer holding the final buckets, *fin / *n_fin =
// their nodes (start / len in records, child_base = k-mers), all pool memory of `ps`.
// Heavy mid buckets (more than SK_MID_LIMIT k-mers: the minimizers of repeats) are taken out of the record path: their
// nodes come back in *heavy (device copies, start / len in records of *heavy_recs), their k-mer counts in heavy_kc.
struct SkHeavy {
    Node *nodes = nullptr;      // device, n entries: start / len in records, child_base = k-mers (if counted)
    u32 n = 0;
    bool counted = true;        // child_base holds the bucket's k-mers (checked against its expansion)
    void *recs = nullptr;       // the record buffer they live in
    u64 total = 0;              // k-mers of all
};
// geometry of a count of n rows: final buckets of ~SK_LEAF_MEAN k-mers = 16 per mid bucket; mid buckets = c0n coarse x 2^b1.
// A multi-GPU count derives it from the GLOBAL row count on every rank (the digits are part of the records).
struct SkGeom {
    int b1, r0bits;
    u32 c0n;
    u64 mid_limit;
};
static SkGeom sk_geometry(const dnagpu_ctx *ctx, u64 n, int k)
{
    SkGeom g;
    // short windows make short runs ((k - m + 2) / 2 k-mers per record on random sequence): the buckets shrink with them
    // so that a bucket's records (~450) still fit sk_count's 512-record stage
    const u64 leaf_mean = std::min<u64>(SK_LEAF_MEAN, 225 * (u64)(k - sk_minimizer_len(k) + 2));
    const u64 n_final = std::max<u64>(n / leaf_mean, 16);
    const u64 n_mid = (n_final + 15) / 16;
    g.b1 = 1;
    while (g.b1 < SK_B1_MAX && ((u64)1 << g.b1) < n_mid)
        g.b1++;
    g.c0n = (u32)std::min<u64>((n_mid + ((u64)1 << g.b1) - 1) >> g.b1, (u64)sk_max_c0());
    g.r0bits = 1;
    while ((1u << g.r0bits) < g.c0n)
        g.r0bits++;
    // (the forced engine of the tests calls a bucket heavy at three times the mean, so that short sequences take that path too)
    g.mid_limit = (ctx->debug_flags & DNAGPU_DEBUG_FORCE_SUPERKMER) ? 3 * (n / ((u64)g.c0n << g.b1) + 1) : SK_MID_LIMIT;
    return g;
}

// Level 0: rows [first, first + n) of the packed sequence -> records in the coarse buckets of geometry g.
// *rec0 = the record buffer (pool memory of ps), *coarse / *n_coarse = the 2^r0bits coarse nodes (device; start / len in
// records, in digit order), kids = the same on the host.
// (rec0_cap != null: the buffer is made large enough for the regions of a speculative level 1 -- sk_levels12 -- and
// *rec0_cap = the records it holds)
static int sk_level0(dnagpu_ctx *ctx, PoolScope &ps, const dnagpu_dna *dna, u64 first, u64 n, int k, const SkGeom &g, void **rec0_out,
                     Node **coarse, u32 *n_coarse, std::vector<Node> *kids_out, u64 *n_recs_out, u64 *rec0_cap = nullptr)
{
    hipStream_t st = ctx->stream;
    const int b1 = g.b1, r0bits = g.r0bits;
    const u32 c0n = g.c0n;
    Node root;
    memset(&root, 0, sizeof root);
    root.len = (u32)n;
    root.meta = 32;                              // "remaining bits" of the bucket digits: r0bits + b1 <= 20 of them are split on
    Node *cur = nullptr;
    RC_TRY(ps.alloc(1, &cur));
    HIP_TRY(poke(cur, &root, sizeof root, st));
    const u64 tile = (u64)sk_tile_rows();
    u64 chunk_rows = std::max<u64>(4 * tile, (n + 4095) / 4096);
    chunk_rows = (chunk_rows + tile - 1) / tile * tile;
    prof_mark(ctx, "sk_plan0");
    SkLevel l0;
    RC_TRY(sk_level_begin(ctx, ps, cur, 1, r0bits, (u32)chunk_rows, &l0));
    prof_mark(ctx, "sk_hist0");
    HIP_TRY(launch_sk_level0(false, l0.chunks, l0.n_chunks, dna->words, dna->n_words, first, k, c0n, (u32)b1, (u32)r0bits,
                             l0.hist, nullptr, nullptr, st, nullptr, ctx->batch_marks, ctx->batch_mark_words));
    prof_mark(ctx, "sk_prefix0");
    HIP_TRY(launch_level_prefix(cur, l0.chunks, l0.n_chunks, 1, (u32)chunk_rows, l0.hist, l0.tot, st));
    HIP_TRY(launch_level_children(cur, 1, l0.tot, l0.next, nullptr, nullptr, nullptr, 0, st));
    std::vector<Node> &kids = *kids_out;
    kids.resize(l0.n_next);
    RC_TRY(read_back(ctx, kids.data(), l0.next, (size_t)l0.n_next * sizeof(Node)));
    u64 n_recs = 0;
    for (const Node &c : kids)
        n_recs += c.len;
    if (n_recs > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    void *rec0 = nullptr;
    u64 cap = std::max<u64>(n_recs, 1);
    if (rec0_cap) {
        u64 span = 0, big = 0, used = 0;
        for (const Node &c : kids) {
            span += sk_spec_span(c.len, b1);
            big = std::max<u64>(big, c.len);
            used += c.len ? 1 : 0;
        }
        if (span <= 0xFFFFFFFFull)
            cap = std::max(cap, span);
        // uneven coarse buckets (repeats): level 1's regions will come from a sampled histogram (sk_levels12), ~20 % of slack
        // on an ordinary mid bucket: room for a third more than the records
        if ((ctx->debug_flags & DNAGPU_DEBUG_SAMPLE1) ||
            (used && (double)big * (double)used > 1.02 * (double)n_recs + 64.0 * (double)used)) {
            const u64 roomy = n_recs + n_recs / 3 + ((u64)kids.size() << b1) * 136;
            if (roomy <= 0xFFFFFFFFull)
                cap = std::max(cap, roomy);
        }
        *rec0_cap = cap;
    }
    RC_TRY(pool_alloc(ctx, (size_t)cap * 16, &rec0));
    ps.ptrs.push_back(rec0);
    prof_mark(ctx, "sk_scatter0");
    HIP_TRY(launch_sk_level0(true, l0.chunks, l0.n_chunks, dna->words, dna->n_words, first, k, c0n, (u32)b1, (u32)r0bits,
                             l0.hist, l0.tot, rec0, st, nullptr, ctx->batch_marks, ctx->batch_mark_words));
    *rec0_out = rec0;
    *coarse = l0.next;
    *n_coarse = l0.n_next;
    *n_recs_out = n_recs;
    return DNAGPU_OK;
}

// Levels 1 and 2 over coarse nodes (records of rec0, which this takes over).  n = the k-mers the records must hold
// (0 = not known: records received from other ranks).  On success *n_kmers = the k-mers found.
// Level 0 WITHOUT its histogram sweep (the window minima are computed once): a histogram over 1/64 of the rows (chunks of
// four tiles, evenly spaced) gives every coarse bucket's share of a chunk's records; every chunk then reserves that share
// + 1/32 + six standard deviations + 24 slots in the bucket's region (one returning add per digit and chunk) and fills
// them as the exact sweep fills its histogram ranges; what it does not use becomes NULL records, which level 1 skips
// (~10 % of the slots at 3 Gbase).  *ok = false (nothing usable produced: the caller runs the exact pair) when the sampled
// buckets are uneven (repeats), when the regions pass 2^32 slots, or when a chunk ran out of slots.  On success the coarse
// nodes cover their whole regions (lens[d] slots, NULL records included) and *n_recs / *rec0_cap are slots.
constexpr u64 SK_SLAB_MIN_ROWS = (u64)1 << 29;     // (measured: 249 Mbase 2.16 vs 2.12 ms, 1 Gbase 8.31 vs 8.42, 3 Gbase 21.8 vs 22.4)
static int sk_level0_slab(dnagpu_ctx *ctx, PoolScope &ps, const dnagpu_dna *dna, u64 first, u64 n, int k, const SkGeom &g,
                          void **rec0_out, Node **coarse, u32 *n_coarse, std::vector<u32> *lens, u64 *n_recs, u64 *rec0_cap, bool *ok)
{
    hipStream_t st = ctx->stream;
    *ok = false;
    const u32 r0n = 1u << g.r0bits;
    Node root;
    memset(&root, 0, sizeof root);
    root.len = (u32)n;
    root.meta = 32;
    Node *cur = nullptr;
    RC_TRY(ps.alloc(1, &cur));
    HIP_TRY(poke(cur, &root, sizeof root, st));
    const u64 tile = (u64)sk_tile_rows();
    // (1024 chunks = one resident set of workgroups: a chunk's share of a bucket is then ~2,400 records, and the six
    // standard deviations of slack it needs are 12 % of them -- with the exact pair's 4096 chunks they would be 25 %)
    u64 chunk_rows = std::max<u64>(4 * tile, (n + 1023) / 1024);
    chunk_rows = (chunk_rows + tile - 1) / tile * tile;
    prof_mark(ctx, "sk_plan0");
    SkLevel l0;
    RC_TRY(sk_level_begin(ctx, ps, cur, 1, g.r0bits, (u32)chunk_rows, &l0));
    if (l0.n_chunks == 0)
        return DNAGPU_OK;
    // ---- the sample
    const u64 samp_len = 4 * tile, samp_stride = 64 * samp_len;
    const u32 n_samp = (u32)((n + samp_stride - 1) / samp_stride);
    u64 sampled = 0;
    for (u32 i = 0; i < n_samp; i++)
        sampled += std::min<u64>(samp_len, n - (u64)i * samp_stride);
    Chunk *samp = nullptr;
    u32 *est = nullptr, *slab = nullptr;
    Node *nodes = nullptr;
    RC_TRY(ps.alloc((size_t)n_samp, &samp));
    RC_TRY(ps.alloc((size_t)sk_max_c0(), &est));
    RC_TRY(ps.alloc((size_t)sk_slab_words(), &slab));
    RC_TRY(ps.alloc((size_t)r0n, &nodes));
    prof_mark(ctx, "sk_sample0");
    HIP_TRY(hipMemsetAsync(est, 0, (size_t)sk_max_c0() * sizeof(u32), st));
    HIP_TRY(launch_sk_sample_chunks(samp, n_samp, (u32)samp_stride, (u32)samp_len, (u32)n, st));
    HIP_TRY(launch_sk_level0(false, samp, n_samp, dna->words, dna->n_words, first, k, g.c0n, (u32)g.b1, (u32)g.r0bits, nullptr, nullptr,
                             nullptr, st, est, ctx->batch_marks, ctx->batch_mark_words));
    HIP_TRY(launch_sk_slab_init(est, (u32)g.r0bits, (u32)chunk_rows, (u32)sampled, l0.n_chunks, slab, nodes, st));
    std::vector<u32> h_est(r0n);
    RC_TRY(read_back(ctx, h_est.data(), est, (size_t)r0n * sizeof(u32)));
    // even buckets?  (a repeated stretch sends its records to the few buckets of its minimizers: see sk_levels12)
    u64 tot = 0, big = 0, used = 0, span = 0, span1 = 0;
    lens->assign(r0n, 0);
    for (u32 d = 0; d < r0n; d++) {
        tot += h_est[d];
        big = std::max<u64>(big, h_est[d]);
        used += h_est[d] ? 1 : 0;
        const u64 len = (u64)sk_slab_cap(h_est[d], chunk_rows, sampled) * l0.n_chunks;
        span += len;
        if (len > 0xFFFFFFFFull)
            return DNAGPU_OK;
        (*lens)[d] = (u32)len;
        span1 += sk_spec_span((u32)len, g.b1);
    }
    if (used == 0 || (double)big * (double)used > 1.05 * (double)tot + 64.0 * (double)used || span > 0xFFFFFFFFull)
        return DNAGPU_OK;
    const u64 cap = std::max<u64>(span, span1 <= 0xFFFFFFFFull ? span1 : 0);
    void *rec0 = nullptr;
    RC_TRY(pool_alloc(ctx, (size_t)std::max<u64>(cap, 1) * 16, &rec0));
    prof_mark(ctx, "sk_scatter0");
    const hipError_t e = launch_sk_level0(true, l0.chunks, l0.n_chunks, dna->words, dna->n_words, first, k, g.c0n, (u32)g.b1,
                                          (u32)g.r0bits, nullptr, nullptr, rec0, st, slab, ctx->batch_marks, ctx->batch_mark_words);
    u32 status[2] = {1, 0};
    int rc = e == hipSuccess ? read_back(ctx, status, slab + 18 * (size_t)sk_max_c0(), sizeof status) : DNAGPU_ERR_HIP;
    if (rc != DNAGPU_OK || status[0] || status[1] != (u32)span || (ctx->debug_flags & DNAGPU_DEBUG_SLAB0_OVERFLOW)) {
        pool_free(ctx, rec0);                      // (ordered behind the sweep on this stream)
        if (e != hipSuccess)
            set_err("sk_scatter0 (slabs): %s", hipGetErrorString(e));
        return rc;
    }
    ps.ptrs.push_back(rec0);
    *rec0_out = rec0;
    *coarse = nodes;
    *n_coarse = r0n;
    *n_recs = span;
    *rec0_cap = cap;
    *ok = true;
    return DNAGPU_OK;
}

// host_lens / rec0_cap (optional): the coarse nodes' record counts on the host and the records rec0 has room for -- with
// both, level 1 runs WITHOUT its histogram where the regions fit (see sk_spec_span): mid buckets are regions of len / 2^b1
// + 12.5 % + 72 slots, the sweep reserves slots from cursors and counts the k-mers per mid bucket itself; a region that
// overflows (repeats) sends the level through the exact path (histogram, prefix, sweep).
static int sk_levels12(dnagpu_ctx *ctx, PoolScope &ps, const SkGeom &g, Node *coarse, u32 n_coarse, void *rec0, u64 n_recs, u64 n,
                       void **recs, Node **fin, u32 *n_fin, SkHeavy *heavy, u64 *n_kmers, const u32 *host_lens = nullptr,
                       u64 rec0_cap = 0)
{
    hipStream_t st = ctx->stream;
    const int b1 = g.b1;
    const u64 mid_limit = g.mid_limit;
    void *rec1 = nullptr;
    SkLevel l0;                                  // (only the node list of level 0 is used below)
    l0.next = coarse;
    l0.n_next = n_coarse;

    // ---- level 1: records of every coarse bucket -> 2^b1 mid buckets; k-mers per mid bucket on the way
    u64 chunk_recs = std::max<u64>(4 * 8192, (n_recs + 4095) / 4096);
    chunk_recs = (chunk_recs + 8191) / 8192 * 8192;
    prof_mark(ctx, "sk_plan1");
    SkLevel l1;
    RC_TRY(sk_level_begin(ctx, ps, l0.next, l0.n_next, b1, (u32)chunk_recs, &l1));
    u32 *kcount = nullptr;
    RC_TRY(ps.alloc(std::max<u32>(l1.n_next, 1), &kcount));
    HIP_TRY(hipMemsetAsync(kcount, 0, (size_t)std::max<u32>(l1.n_next, 1) * sizeof(u32), st));
    u32 *d_lens = nullptr;
    RC_TRY(ps.alloc((size_t)l1.n_next + 4, &d_lens));      // (+ the speculative sweep's three status words)
    u32 *gcur = nullptr;
    RC_TRY(ps.alloc((size_t)std::max<u32>(l1.n_chunks, 1) * ROW_STRIDE, &gcur));
    std::vector<u32> kc(l1.n_next), rcn((size_t)l1.n_next + 4);
    // mid-bucket k-mer and record counts to the host (the list is short), with `extra` words behind the record counts
    auto lens_to_host = [&](u32 extra) -> int {
        const size_t nb = (size_t)l1.n_next * sizeof(u32), nb2 = nb + (size_t)extra * sizeof(u32);
        if (nb + nb2 <= MAILBOX_BYTES - 8) {      // both lists through the pinned mailbox, one wait (its last word is read_back's flag)
            char *mb = reinterpret_cast<char *>(ctx->mailbox);
            HIP_TRY(hipMemcpyAsync(mb, kcount, nb, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(mb + nb, d_lens, nb2, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            memcpy(kc.data(), mb, nb);
            memcpy(rcn.data(), mb + nb, nb2);
        } else {
            HIP_TRY(hipMemcpyAsync(kc.data(), kcount, nb, hipMemcpyDeviceToHost, st));
            RC_TRY(read_back(ctx, rcn.data(), d_lens, nb2));
        }
        return DNAGPU_OK;
    };
    // ---- speculative: no histogram.  The regions must fit both record buffers (level 2 writes a mid bucket's final
    // buckets back into its range of rec0).
    u64 span = 0;
    bool even = true;
    if (host_lens) {
        // Repeats show at the coarse level already: a repeated stretch sends its records to the few buckets of its minimizers
        // (half a sequence of one tiled 1000-base motif doubles ~110 of 136 coarse buckets; random sequence fills them to
        // within 0.3 %).  Uneven coarse buckets (2 % over the mean of the non-empty ones) take the exact level at once,
        // instead of paying for a speculative sweep that will overflow.
        u64 tot = 0, big = 0, used = 0;
        for (u32 i = 0; i < n_coarse; i++) {
            span += sk_spec_span(host_lens[i], b1);
            tot += host_lens[i];
            big = std::max<u64>(big, host_lens[i]);
            used += host_lens[i] ? 1 : 0;
        }
        even = used == 0 || (double)big * (double)used <= 1.02 * (double)tot + 64.0 * (double)used;
    }
    const bool can_spec = host_lens && !(ctx->debug_flags & DNAGPU_DEBUG_NO_SPEC1) && n_coarse <= (u32)sk_max_c0() && l1.n_chunks > 0;
    const bool force_sample = (ctx->debug_flags & DNAGPU_DEBUG_SAMPLE1) != 0;
    bool spec = can_spec && even && !force_sample && span <= rec0_cap && span <= 0xFFFFFFFFull;
    // ---- uneven coarse buckets (repeats): the regions from a SAMPLED histogram -- one piece of 1024 records in every eight
    // of a coarse bucket's, read once (an eighth of the records: ~0.15 ms at 3 Gbase against the exact histogram's 1.0 - 1.3)
    // -- estimate + five standard deviations + 128 slots per mid bucket, so that a heavy mid bucket gets a region of its
    // size.  One more wait for the host (the regions' total decides the buffer); a region that overflows all the same
    // (bursts the sample missed) falls back to the exact level like every speculative sweep.
    u32 *rstart = nullptr, *rcapv = nullptr;
    bool sampled = false;
    if (can_spec && (!even || force_sample) && !spec) {
        std::vector<Chunk> samp;
        const u32 slen = sk_sample1_len(), sstep = slen * sk_sample1_every();
        for (u32 i = 0; i < n_coarse; i++)
            for (u64 off = 0; off < host_lens[i]; off += sstep) {
                Chunk c;
                c.node = i;
                c.off = (u32)off;
                c.len = (u32)std::min<u64>(slen, host_lens[i] - off);
                c.pad = 0;
                samp.push_back(c);
            }
        Chunk *d_samp = nullptr;
        u32 *est = nullptr, *stmp = nullptr, *tot1 = nullptr;
        RC_TRY(ps.alloc(std::max<size_t>(samp.size(), 1), &d_samp));
        RC_TRY(ps.alloc((size_t)l1.n_next, &est));
        RC_TRY(ps.alloc((size_t)l1.n_next, &rcapv));
        RC_TRY(ps.alloc((size_t)l1.n_next, &rstart));
        RC_TRY(ps.alloc((size_t)scan_tmp_words(l1.n_next), &stmp));
        RC_TRY(ps.alloc(1, &tot1));
        prof_mark(ctx, "sk_sample1");
        if (!samp.empty())
            HIP_TRY(hipMemcpyAsync(d_samp, samp.data(), samp.size() * sizeof(Chunk), hipMemcpyHostToDevice, st));
        HIP_TRY(launch_sk_sampled_regions(l0.next, d_samp, (u32)samp.size(), rec0, l1.n_next, est, rcapv, rstart, stmp, tot1, st));
        u32 total = 0;
        RC_TRY(read_back(ctx, &total, tot1, sizeof total));      // (also: samp has been consumed)
        // (the scan's total wraps past 2^32: regions that large are out of reach of 32-bit slots anyway -- the check below
        // compares against rec0's room, which is below 2^32)
        u64 chk = 0;
        for (u32 i = 0; i < n_coarse; i++)
            chk += host_lens[i];
        if ((u64)total >= chk && (u64)total <= rec0_cap) {
            span = total;
            spec = sampled = true;
        }
    }
    bool moved = false;
    if (spec) {
        u32 *sp = nullptr;
        RC_TRY(ps.alloc((size_t)2 * n_coarse, &sp));
        u32 *status = d_lens + l1.n_next;         // [0] slots of all regions, [1] past 2^32, [2] overflow
        prof_mark(ctx, "sk_spec1");
        HIP_TRY(launch_sk_spec_regions(l0.next, n_coarse, sp, status, gcur, st, sampled ? rstart : nullptr));
        RC_TRY(pool_alloc(ctx, (size_t)std::max<u64>(std::max(n_recs, span), 1) * 16, &rec1));
        ps.ptrs.push_back(rec1);
        prof_mark(ctx, "sk_scatter1");
        HIP_TRY(launch_sk_scatter1_spec(l0.next, l1.chunks, l1.n_chunks, rec0, rec1, gcur, sp, kcount, status + 2, st,
                                        sampled ? rstart : nullptr, sampled ? rcapv : nullptr));
        HIP_TRY(launch_sk_spec_nodes(l0.next, l0.n_next, sp, gcur, l1.next, status + 2, st, sampled ? rstart : nullptr,
                                     sampled ? rcapv : nullptr));
        HIP_TRY(launch_sk_node_lens(l1.next, l1.n_next, d_lens, st));
        RC_TRY(lens_to_host(3));
        const u32 *stw = rcn.data() + l1.n_next;
        if ((!sampled && stw[0] != (u32)span) || (!sampled && stw[1]) || stw[2] || (ctx->debug_flags & DNAGPU_DEBUG_SPEC1_OVERFLOW)) {
            spec = false;                          // (a region overflowed, or the test flag says so: the exact level, into the same rec1)
            HIP_TRY(hipMemsetAsync(kcount, 0, (size_t)std::max<u32>(l1.n_next, 1) * sizeof(u32), st));
        } else {
            moved = true;
        }
    }
    if (!spec) {
        prof_mark(ctx, "sk_hist1");
        HIP_TRY(launch_sk_hist1(l0.next, l1.chunks, l1.n_chunks, rec0, l1.hist, kcount, st));
        prof_mark(ctx, "sk_prefix1");
        HIP_TRY(launch_level_prefix(l0.next, l1.chunks, l1.n_chunks, n_coarse, (u32)chunk_recs, l1.hist, l1.tot, st, n_coarse));
        HIP_TRY(launch_level_children(l0.next, l0.n_next, l1.tot, l1.next, nullptr, nullptr, nullptr, 0, st));
        // ---- skew check on the k-mers per mid bucket, before their records move
        HIP_TRY(launch_sk_node_lens(l1.next, l1.n_next, d_lens, st));
        RC_TRY(lens_to_host(0));
    }
    // heavy: too many k-mers, or too many records for the one workgroup that regroups a mid bucket (its tiles are serial)
    const bool forced = (ctx->debug_flags & DNAGPU_DEBUG_FORCE_SUPERKMER) != 0;
    auto is_heavy = [&](u32 i) { return kc[i] > mid_limit || (!forced && rcn[i] > SK_MID_RECORDS); };
    u64 run = 0, heaviest = 0;
    for (u32 i = 0; i < l1.n_next; i++) {
        run += kc[i];
        if (is_heavy(i))
            heaviest = std::max<u64>(heaviest, std::max<u64>(kc[i], mid_limit + 1));
    }
    if (n != 0 && run != n) {
        set_err("super-k-mer partition lost rows: %llu of %llu", (unsigned long long)run, (unsigned long long)n);
        return DNAGPU_ERR_INTERNAL;
    }
    if (run > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    *n_kmers = run;
    std::vector<u32> heavy_idx;
    if (heaviest > mid_limit) {
        for (u32 i = 0; i < l1.n_next; i++)
            if (is_heavy(i)) {
                heavy_idx.push_back(i);
                heavy->total += kc[i];
            }
        // DNAGPU_SK_SKEWED leaves this function in two cases only: with DNAGPU_DEBUG_HEAVY_EXPAND (the older path, kept for
        // the tests: heavy mid buckets are expanded as a whole, and a set that is mostly heavy is cheaper through the tree
        // from scratch -- count_core -- or as one key node per coarse bucket -- count_sk_received; rec0 is still what it
        // was: level 1 only reads it), or with more heavy mid buckets than the chunked split below plans for (32768: not
        // reachable with 2^32 rows, a guard).
        if (((ctx->debug_flags & DNAGPU_DEBUG_HEAVY_EXPAND) && heavy->total * 2 > run) || heavy_idx.size() > 32768)
            return DNAGPU_SK_SKEWED;
    }

    if (!moved) {
        if (!rec1) {
            RC_TRY(pool_alloc(ctx, (size_t)std::max<u64>(n_recs, 1) * 16, &rec1));
            ps.ptrs.push_back(rec1);
        }
        prof_mark(ctx, "sk_scatter1");
        // the mid buckets' cursors start at their exact bases (the prefix of the histogram); tiles reserve their slots there
        HIP_TRY(hipMemcpyAsync(gcur, l1.tot, (size_t)std::max<u32>(l1.n_chunks, 1) * ROW_STRIDE * sizeof(u32), hipMemcpyDeviceToDevice, st));
        HIP_TRY(launch_sk_scatter1(l0.next, l1.chunks, l1.n_chunks, rec0, rec1, l1.hist, l1.tot, st, false, gcur));
    }

    const u32 nh = (u32)heavy_idx.size();
    const bool heavy_expand = (ctx->debug_flags & DNAGPU_DEBUG_HEAVY_EXPAND) != 0;
    SkLevel lh;
    memset(&lh, 0, sizeof lh);
    u32 *kcount2 = nullptr;
    Node *hnodes = nullptr;
    if (nh) {
        // the heavy buckets leave the list here (empty nodes stay behind): one workgroup could not regroup them in time
        u32 *d_idx = nullptr;
        RC_TRY(ps.alloc((size_t)nh, &d_idx));
        RC_TRY(ps.alloc((size_t)nh, &hnodes));
        HIP_TRY(hipMemcpyAsync(d_idx, heavy_idx.data(), (size_t)nh * sizeof(u32), hipMemcpyHostToDevice, st));
        HIP_TRY(launch_sk_take_heavy(l1.next, d_idx, nh, kcount, hnodes, st));
        HIP_TRY(hipStreamSynchronize(st));       // (heavy_idx is a host vector)
        if (heavy_expand) {                      // (tests: the expansion of whole mid buckets)
            heavy->nodes = hnodes;
            heavy->n = nh;
            heavy->recs = rec1;
        } else {
            // They are split by d2 with the CHUNKED level kernels instead (many workgroups per bucket: plan, histogram,
            // prefix, children, scatter rec1 -> rec0 into the range the bucket would have been regrouped into); their
            // sixteen children join the final buckets, where sk_count_big takes the long ones slice by slice.
            u64 hrecs = 0;
            for (u32 i : heavy_idx)
                hrecs += rcn[i];
            u64 chunk_h = std::max<u64>(4 * 8192, (hrecs + 4095) / 4096);
            chunk_h = (chunk_h + 8191) / 8192 * 8192;
            prof_mark(ctx, "sk_heavy_split");
            RC_TRY(sk_level_begin(ctx, ps, hnodes, nh, 4, (u32)chunk_h, &lh));
            RC_TRY(ps.alloc(std::max<u32>(lh.n_next, 1), &kcount2));
            HIP_TRY(hipMemsetAsync(kcount2, 0, (size_t)std::max<u32>(lh.n_next, 1) * sizeof(u32), st));
            HIP_TRY(launch_sk_hist1(hnodes, lh.chunks, lh.n_chunks, rec1, lh.hist, kcount2, st, true));
            HIP_TRY(launch_level_prefix(hnodes, lh.chunks, lh.n_chunks, nh, (u32)chunk_h, lh.hist, lh.tot, st, nh));
            HIP_TRY(launch_level_children(hnodes, nh, lh.tot, lh.next, nullptr, nullptr, nullptr, 0, st));
            HIP_TRY(launch_sk_scatter1(hnodes, lh.chunks, lh.n_chunks, rec1, rec0, lh.hist, lh.tot, st, true));
            heavy->total = 0;                    // (nothing is left for the expansion of mid buckets)
        }
    }
    // ---- level 2: every mid bucket regrouped by d2 (rec1 -> rec0): 16 final buckets each
    Node *fn = nullptr;
    RC_TRY(ps.alloc((size_t)l1.n_next * 16 + lh.n_next, &fn));
    prof_mark(ctx, "sk_regroup");
    bool any_long = false;                       // (mid buckets of more than one regroup tile: repeats)
    for (u32 i = 0; i < l1.n_next && !any_long; i++)
        any_long = rcn[i] > (u32)sk_regroup_tile();
    HIP_TRY(launch_sk_regroup(l1.next, l1.n_next, rec1, rec0, fn, any_long, st));
    if (lh.n_next)
        HIP_TRY(launch_sk_heavy_finals(lh.next, lh.n_next, kcount2, fn + (size_t)l1.n_next * 16, st));
    if (nh == 0 || !heavy_expand)
        ps.free_now(rec1);
    *recs = rec0;
    *fin = fn;
    *n_fin = l1.n_next * 16 + lh.n_next;
    return DNAGPU_OK;
}

// The whole count: partition, then final buckets of at most sk_count_cap() k-mers are counted from their records
// (sk_count), the others expanded to keys and counted by the ordinary levels.  Fills h on success.
static int count_sk_tail(dnagpu_ctx *ctx, PoolScope &ps, void *recs, Node *fin, u32 n_fin, const SkHeavy &heavy, u64 n, int k,
                         dnagpu_hist *h);

// n = rows swept; n_kmers_expected = the k-mers they hold (fewer over a table of sequences: ctx->batch_marks)
static int count_sk(dnagpu_ctx *ctx, const dnagpu_dna *dna, u64 first, u64 n, int k, dnagpu_hist *h, u64 n_kmers_expected)
{
    PoolScope ps(ctx);
    const SkGeom g = sk_geometry(ctx, std::max<u64>(n_kmers_expected, 1), k);
    void *rec0 = nullptr, *recs = nullptr;
    Node *coarse = nullptr, *fin = nullptr;
    u32 n_coarse = 0, n_fin = 0;
    u64 n_recs = 0, n_kmers = 0;
    std::vector<Node> kids;
    SkHeavy heavy;
    u64 rec0_cap = 0;
    std::vector<u32> lens;
    bool slabs = false;
    const unsigned dbg = ctx->debug_flags;
    if (!(dbg & DNAGPU_DEBUG_NO_SLAB0) && (n >= SK_SLAB_MIN_ROWS || (dbg & (DNAGPU_DEBUG_SLAB0 | DNAGPU_DEBUG_SLAB0_OVERFLOW))))
        RC_TRY(sk_level0_slab(ctx, ps, dna, first, n, k, g, &rec0, &coarse, &n_coarse, &lens, &n_recs, &rec0_cap, &slabs));
    if (!slabs) {
        RC_TRY(sk_level0(ctx, ps, dna, first, n, k, g, &rec0, &coarse, &n_coarse, &kids, &n_recs, &rec0_cap));
        lens.resize(kids.size());
        for (size_t i = 0; i < kids.size(); i++)
            lens[i] = kids[i].len;
    }
    RC_TRY(sk_levels12(ctx, ps, g, coarse, n_coarse, rec0, n_recs, n_kmers_expected, &recs, &fin, &n_fin, &heavy, &n_kmers,
                       lens.size() == n_coarse ? lens.data() : nullptr, rec0_cap));
    return count_sk_tail(ctx, ps, recs, fin, n_fin, heavy, n_kmers_expected, k, h);
}

// Records that arrive from elsewhere (the multi-GPU exchange: every rank cuts the records of its own chunk and ships each
// coarse bucket to its owner): pieces[i] = piece_len[i] records of coarse bucket piece_bucket[i], device memory.  The
// pieces are copied bucket by bucket into one buffer (equal k-mers share the bucket, so its pieces must form ONE node),
// then levels 1-2 and the counting as in count_sk.  A skewed set (more than half the k-mers in heavy mid buckets)
// cannot fall back to the sequence here: all of it is expanded to keys for the ordinary levels instead.
// rec0 (pool memory; this takes it over and returns it to the pool) = the records of the coarse buckets, bucket after
// bucket: bucket d = blen[d] records at boff[d] (n_coarse = 2^r0bits entries).  Everything queued on ctx->stream
// behind whatever filled rec0.
// the records a landing buffer of the buckets blen[] should have room for, so that level 1 can run without its histogram
static u64 sk_received_cap(const std::vector<u64> &blen, u32 n_coarse, const SkGeom &g)
{
    u64 n_recs = 0, span = 0;
    for (u32 d = 0; d < n_coarse; d++) {
        n_recs += blen[d];
        span += sk_spec_span((u32)std::min<u64>(blen[d], 0xFFFFFFFFull), g.b1);
    }
    return span <= 0xFFFFFFFFull ? std::max(n_recs, span) : n_recs;
}

static int count_sk_received(dnagpu_ctx *ctx, void *rec0, const std::vector<u64> &boff, const std::vector<u64> &blen, const SkGeom &g,
                             int k, dnagpu_hist *h, u64 rec0_cap = 0)
{
    hipStream_t st = ctx->stream;
    PoolScope ps(ctx);
    ps.ptrs.push_back(rec0);
    const u32 n_coarse = 1u << g.r0bits;
    const u64 n_recs = boff[n_coarse];
    std::vector<Node> hn(n_coarse);
    for (u32 d = 0; d < n_coarse; d++) {
        memset(&hn[d], 0, sizeof(Node));
        hn[d].start = (u32)boff[d];
        hn[d].len = (u32)blen[d];
        hn[d].meta = (u32)(32 - g.r0bits);       // (what level_children leaves a child of the 32-"bit" root)
    }
    Node *coarse = nullptr;
    RC_TRY(ps.alloc((size_t)n_coarse, &coarse));
    HIP_TRY(hipMemcpyAsync(coarse, hn.data(), (size_t)n_coarse * sizeof(Node), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));           // (hn, and the caller's pieces, are free again)
    void *recs = nullptr;
    Node *fin = nullptr;
    u32 n_fin = 0;
    u64 n_kmers = 0;
    SkHeavy heavy;
    std::vector<u32> lens(n_coarse);
    for (u32 d = 0; d < n_coarse; d++)
        lens[d] = (u32)blen[d];
    int rc = sk_levels12(ctx, ps, g, coarse, n_coarse, rec0, n_recs, 0, &recs, &fin, &n_fin, &heavy, &n_kmers, lens.data(), rec0_cap);
    if (rc == DNAGPU_SK_SKEWED) {
        // every coarse bucket as one "heavy" bucket: keys, then the ordinary levels (sk_levels12 has not moved anything yet)
        heavy = SkHeavy();
        heavy.nodes = coarse;
        heavy.n = n_coarse;
        heavy.recs = rec0;
        heavy.total = n_kmers;
        heavy.counted = false;
        n_fin = 0;
        rc = DNAGPU_OK;
    }
    RC_TRY(rc);
    return count_sk_tail(ctx, ps, recs, fin, n_fin, heavy, n_kmers, k, h);
}

static int count_sk_records(dnagpu_ctx *ctx, const void *const *pieces, const u64 *piece_len, const u32 *piece_bucket, u32 n_pieces,
                            int k, u64 global_rows, dnagpu_hist *h)
{
    hipStream_t st = ctx->stream;
    const SkGeom g = sk_geometry(ctx, global_rows, k);
    const u32 n_coarse = 1u << g.r0bits;
    std::vector<u64> blen(n_coarse, 0), boff(n_coarse + 1, 0);
    for (u32 i = 0; i < n_pieces; i++) {
        if (piece_bucket[i] >= g.c0n || (piece_len[i] && !pieces[i]))
            return DNAGPU_ERR_BAD_ARG;
        blen[piece_bucket[i]] += piece_len[i];
    }
    for (u32 d = 0; d < n_coarse; d++)
        boff[d + 1] = boff[d] + blen[d];
    const u64 n_recs = boff[n_coarse];
    if (n_recs > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    if (n_recs == 0) {
        h->total = 0;
        return DNAGPU_OK;
    }
    void *rec0 = nullptr;
    const u64 cap0 = sk_received_cap(blen, n_coarse, g);    // (room for the regions of a level 1 without its histogram)
    RC_TRY(pool_alloc(ctx, (size_t)cap0 * 16, &rec0));
    std::vector<u64> fill(boff.begin(), boff.end() - 1);
    for (u32 i = 0; i < n_pieces; i++)
        if (piece_len[i]) {
            const hipError_t e = hipMemcpyAsync(static_cast<char *>(rec0) + fill[piece_bucket[i]] * 16, pieces[i], (size_t)piece_len[i] * 16,
                                                hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) {
                pool_free(ctx, rec0);
                set_err("record pieces: %s", hipGetErrorString(e));
                return DNAGPU_ERR_HIP;
            }
            fill[piece_bucket[i]] += piece_len[i];
        }
    return count_sk_received(ctx, rec0, boff, blen, g, k, h, cap0);
}

static int count_sk_tail(dnagpu_ctx *ctx, PoolScope &ps, void *recs, Node *fin, u32 n_fin, const SkHeavy &heavy, u64 n, int k,
                         dnagpu_hist *h)
{
    hipStream_t st = ctx->stream;
    const u32 n_heavy = heavy.n;
    prof_mark(ctx, "sk_select");
    const u32 cap = (u32)sk_count_cap();
    const u32 big_limit = (u32)SK_BIG_LIMIT;
    u32 *f_small = nullptr, *f_big = nullptr, *f_over = nullptr, *f_over_raw = nullptr, *k_over = nullptr, *k_range = nullptr,
        *scan_tmp = nullptr, *totals = nullptr, *list_small = nullptr, *off_small = nullptr, *list_big = nullptr, *off_big = nullptr;
    RC_TRY(ps.alloc((size_t)n_fin, &f_small));
    RC_TRY(ps.alloc((size_t)n_fin, &f_big));
    RC_TRY(ps.alloc((size_t)n_fin, &k_range));
    RC_TRY(ps.alloc((size_t)scan_tmp_words(n_fin) * 4, &scan_tmp));      // (four scans at a time: launch_scan_u32_multi)
    RC_TRY(ps.alloc(8, &totals));
    RC_TRY(ps.alloc((size_t)n_fin, &list_small));
    RC_TRY(ps.alloc((size_t)n_fin, &off_small));
    RC_TRY(ps.alloc((size_t)n_fin, &f_over));       // (first: the k-mers of the big buckets, summed)
    HIP_TRY(launch_sk_select_flags(fin, n_fin, cap, big_limit, f_small, f_big, k_range, f_over, st));
    {
        ScanSet ss;
        u32 *arr[4] = {f_small, f_big, k_range, f_over};
        for (int a = 0; a < 4; a++) {
            ss.in[a] = arr[a];
            ss.out[a] = arr[a];
            ss.total[a] = totals + a;
        }
        ss.tmp = scan_tmp;
        HIP_TRY(launch_scan_u32_multi(ss, 4, n_fin, st));
    }
    u32 ht[4] = {0, 0, 0, 0};
    RC_TRY(read_back(ctx, ht, totals, sizeof ht));
    const u32 n_small = ht[0], n_big = ht[1];
    const u64 small_keys = ht[2];                // the output slots of the small and big buckets: one per k-mer, in bucket order
    // a big bucket that sk_count_big gives up on is counted again through the expansion: its groups land behind the
    // ranges while its own range stays padding, so the arrays hold up to n + the big buckets' k-mers
    const u64 out_cap = n + ht[3];
    RC_TRY(ps.alloc((size_t)std::max<u32>(n_big, 1), &list_big));
    RC_TRY(ps.alloc((size_t)std::max<u32>(n_big, 1), &off_big));
    HIP_TRY(launch_sk_select_lists(fin, n_fin, cap, big_limit, f_small, f_big, k_range, list_small, off_small, list_big, off_big, st));

    // output arrays and the segment directory: final buckets first, the nodes of the oversize buckets' tree behind them
    u64 *cursor = nullptr, *ok = nullptr;
    u32 *oc = nullptr;
    // [0] next free output slot of the leaves behind the buckets' ranges; [1] buckets whose expansion disagrees with the
    // partition's count; [2] groups sk_count and sk_count_big wrote
    RC_TRY(ps.alloc(3, &cursor));
    RC_TRY(ps.alloc((size_t)out_cap, &ok));
    RC_TRY(ps.alloc((size_t)out_cap, &oc));
    {
        const u64 init[3] = {small_keys, 0, 0};
        HIP_TRY(poke(cursor, init, sizeof init, st));
    }
    u64 *seg_off = nullptr;
    u32 *seg_cnt = nullptr;
    // (the directory of the final buckets; the tree's nodes get a second one behind it once their number is known)
    u64 *seg_off_fin = nullptr;
    u32 *seg_cnt_fin = nullptr;
    RC_TRY(ps.alloc((size_t)std::max<u32>(n_fin, 1), &seg_off_fin));
    RC_TRY(ps.alloc((size_t)std::max<u32>(n_fin, 1), &seg_cnt_fin));
    HIP_TRY(hipMemsetAsync(seg_cnt_fin, 0, (size_t)n_fin * sizeof(u32), st));     // empty and expanded buckets: no groups of their own
    HIP_TRY(hipMemsetAsync(seg_off_fin, 0, (size_t)n_fin * sizeof(u64), st));
    // ---- long buckets of few distinct keys (repeats): one table per bucket; what outgrows it joins the expansion below
    u32 *big_status = nullptr;
    RC_TRY(ps.alloc((size_t)std::max<u32>(n_big, 1), &big_status));
    if (n_big) {
        prof_mark(ctx, "sk_count_big");
        // work items: slices of the buckets' records; a bucket of several slices gets a partial area per slice
        u32 *nsl = nullptr, *sfirst = nullptr, *mfirst = nullptr, *sl_bucket = nullptr, *sl_idx = nullptr, *part_n = nullptr, *part_cnts = nullptr;
        u64 *part_keys = nullptr;
        RC_TRY(ps.alloc((size_t)n_big, &nsl));
        RC_TRY(ps.alloc((size_t)n_big, &sfirst));
        RC_TRY(ps.alloc((size_t)n_big, &mfirst));
        HIP_TRY(launch_sk_big_slices(fin, list_big, n_big, nsl, mfirst, st));
        HIP_TRY(launch_scan_u32(nsl, sfirst, n_big, scan_tmp, totals + 6, st));
        HIP_TRY(launch_scan_u32(mfirst, mfirst, n_big, scan_tmp, totals + 7, st));
        u32 hs[2] = {0, 0};
        RC_TRY(read_back(ctx, hs, totals + 6, sizeof hs));
        const u32 n_slices = hs[0], n_part = hs[1];
        RC_TRY(ps.alloc((size_t)std::max<u32>(n_slices, 1), &sl_bucket));
        RC_TRY(ps.alloc((size_t)std::max<u32>(n_slices, 1), &sl_idx));
        RC_TRY(ps.alloc((size_t)std::max<u32>(n_part, 1), &part_n));
        RC_TRY(ps.alloc((size_t)std::max<u32>(n_part, 1) * sk_big_partial_slots(), &part_keys));
        RC_TRY(ps.alloc((size_t)std::max<u32>(n_part, 1) * sk_big_partial_slots(), &part_cnts));
        HIP_TRY(launch_sk_big_slice_fill(nsl, sfirst, n_big, sl_bucket, sl_idx, st));
        HIP_TRY(launch_sk_count_big(fin, list_big, off_big, nsl, mfirst, sl_bucket, sl_idx, n_slices, n_big, recs, k, cursor + 2,
                                    seg_off_fin, seg_cnt_fin, ok, oc, big_status, part_keys, part_cnts, part_n, n_part > 0, st));
    }
    prof_mark(ctx, "sk_select_over");
    RC_TRY(ps.alloc((size_t)n_fin, &f_over_raw));
    RC_TRY(ps.alloc((size_t)n_fin, &k_over));
    HIP_TRY(launch_sk_over_flags(fin, n_fin, cap, big_limit, f_big, big_status, f_over_raw, k_over, st));
    {
        ScanSet ss;
        memset(&ss, 0, sizeof ss);
        ss.in[0] = f_over_raw;
        ss.out[0] = f_over;
        ss.total[0] = totals + 4;
        ss.in[1] = k_over;
        ss.out[1] = k_over;
        ss.total[1] = totals + 5;
        ss.tmp = scan_tmp;
        HIP_TRY(launch_scan_u32_multi(ss, 2, n_fin, st));
    }
    u32 ho[2] = {0, 0};
    RC_TRY(read_back(ctx, ho, totals + 4, sizeof ho));
    const u32 n_over = ho[0];
    const u64 over_keys = ho[1];
    Node *over_nodes = nullptr;
    u32 *over_kbase = nullptr;
    RC_TRY(ps.alloc((size_t)std::max<u32>(n_over, 1), &over_nodes));
    RC_TRY(ps.alloc((size_t)std::max<u32>(n_over, 1), &over_kbase));
    HIP_TRY(launch_sk_over_list(fin, n_fin, f_over_raw, f_over, k_over, over_nodes, over_kbase, st));
    TreeResult tr;
    memset(&tr, 0, sizeof tr);
    if (n_over + n_heavy > 0) {
        // oversize final buckets (the tail of the size distribution, moderate repeats) and heavy mid buckets (the
        // minimizers of long repeats): keys, then the ordinary levels with their skew paths.  Every such bucket becomes
        // one key node; its records are expanded in slices by many waves at once.
        const u64 tree_keys = over_keys + heavy.total;
        if (tree_keys > 0xFFFFFFFFull)
            return DNAGPU_ERR_TOO_LARGE;
        const u32 n_tree = n_over + n_heavy;
        u64 *kbuf = nullptr;
        Node *knodes = nullptr;
        RC_TRY(ps.alloc((size_t)tree_keys, &kbuf));
        RC_TRY(ps.alloc((size_t)n_tree, &knodes));
        prof_mark(ctx, "sk_expand_flat");
        // the final buckets live in `recs`, the heavy mid buckets in heavy.recs -> two slice lists; the key ranges: the
        // oversize final buckets in list order, the heavy buckets behind them
        for (int part = 0; part < 2; part++) {
            const u32 nb = part == 0 ? n_over : n_heavy;
            if (nb == 0)
                continue;
            const void *rbuf = part == 0 ? recs : heavy.recs;
            const Node *bk = part == 0 ? over_nodes : heavy.nodes;     // (start / len in records, child_base = k-mers)
            u32 *sfirst = nullptr, *stmp = nullptr, *stot = nullptr;
            RC_TRY(ps.alloc((size_t)nb, &sfirst));
            RC_TRY(ps.alloc((size_t)scan_tmp_words(nb), &stmp));
            RC_TRY(ps.alloc(2, &stot));
            HIP_TRY(launch_sk_slice_count(bk, nb, sfirst, st));
            HIP_TRY(launch_scan_u32(sfirst, sfirst, nb, stmp, stot, st));
            u32 n_slices = 0;
            RC_TRY(read_back(ctx, &n_slices, stot, 4));
            u32 *d_r0 = nullptr, *d_nr = nullptr, *d_ko = nullptr, *ktmp = nullptr;
            RC_TRY(ps.alloc((size_t)std::max<u32>(n_slices, 1), &d_r0));
            RC_TRY(ps.alloc((size_t)std::max<u32>(n_slices, 1), &d_nr));
            RC_TRY(ps.alloc((size_t)std::max<u32>(n_slices, 1), &d_ko));
            RC_TRY(ps.alloc((size_t)scan_tmp_words(std::max<u32>(n_slices, 1)), &ktmp));
            HIP_TRY(launch_sk_slice_fill(bk, nb, sfirst, d_r0, d_nr, st));
            HIP_TRY(launch_sk_slice_kmers(rbuf, d_r0, d_nr, n_slices, d_ko, st));
            HIP_TRY(launch_scan_u32(d_ko, d_ko, n_slices, ktmp, stot + 1, st));
            const u32 key_base = part == 0 ? 0u : (u32)over_keys;
            HIP_TRY(launch_sk_slice_nodes(bk, nb, sfirst, d_ko, n_slices, stot + 1, key_base, k, part == 0 || heavy.counted,
                                          knodes + (part == 0 ? 0 : n_over), cursor + 1, st));
            HIP_TRY(launch_sk_expand_flat(rbuf, d_r0, d_nr, d_ko, key_base, n_slices, k, kbuf, st));
        }
        RC_TRY(run_tree(ctx, ps, nullptr, 0, tree_keys, k, kbuf, 0, &tr, 0, 0, true, 0, ~0u, 0, knodes, n_tree, 2));
    }
#ifdef DNAGPU_STAMPS
    fprintf(stderr, "[sk select] n_fin %u small %u big %u (k-mers of big %u) over %u (keys %llu) heavy %u (keys %llu) tree nodes %u tiny %u small %u big %u\n",
            n_fin, n_small, n_big, ht[3], n_over, (unsigned long long)over_keys, n_heavy, (unsigned long long)heavy.total, tr.n_nodes,
            tr.n_tiny, tr.n_small, tr.n_big);
#endif
    const u32 n_segs = n_fin + tr.n_nodes;
    RC_TRY(ps.alloc((size_t)n_segs, &seg_off));
    RC_TRY(ps.alloc((size_t)n_segs, &seg_cnt));
    HIP_TRY(hipMemcpyAsync(seg_cnt, seg_cnt_fin, (size_t)n_fin * sizeof(u32), hipMemcpyDeviceToDevice, st));   // (sk_count_big's entries)
    HIP_TRY(hipMemcpyAsync(seg_off, seg_off_fin, (size_t)n_fin * sizeof(u64), hipMemcpyDeviceToDevice, st));
    if (tr.n_nodes > 0) {
        u32 *flags = nullptr, *ltmp = nullptr, *cls_list = nullptr;
        RC_TRY(ps.alloc((size_t)tr.n_nodes + 1, &flags));
        RC_TRY(ps.alloc((size_t)scan_tmp_words(tr.n_nodes), &ltmp));
        RC_TRY(ps.alloc((size_t)tr.n_nodes, &cls_list));
        prof_mark(ctx, "leaves");
        HIP_TRY(launch_leaves(tr.nodes, tr.n_nodes, tr.n_tiny, tr.n_small, tr.n_big, tr.buf0, tr.buf1, cursor, seg_off + n_fin,
                              seg_cnt + n_fin, ok, oc, flags, ltmp, cls_list, st, true, small_keys));
        HIP_TRY(launch_sk_unmix(ok, small_keys, cursor, tr.n_keys, k, st));      // (sk_expand_flat wrote key_mix(key))
    }
    prof_mark(ctx, "sk_count");
    u32 *left = nullptr;
    RC_TRY(ps.alloc((size_t)2 * n_small + 1, &left));
    HIP_TRY(launch_sk_count(fin, list_small, off_small, n_small, recs, k, cursor + 2, seg_off, seg_cnt, ok, oc, left, st));
#ifdef DNAGPU_STAMPS
    {
        u32 nl = 0;
        RC_TRY(read_back(ctx, &nl, left + 2 * (size_t)n_small, 4));
        fprintf(stderr, "[sk count] %u small buckets, %u left to sk_count by sk_count_clean\n", n_small, nl);
    }
#endif
    prof_mark(ctx, "end");
    u64 fin_ctr[3] = {0, 0, 0};
    RC_TRY(read_back(ctx, fin_ctr, cursor, 24));
    const u64 extent = fin_ctr[0];
    const u64 total_groups = fin_ctr[2] + (extent - small_keys);
    if (fin_ctr[1] != 0) {
        set_err("super-k-mer count: %llu buckets whose records expand to a different number of k-mers than the partition counted",
                (unsigned long long)fin_ctr[1]);
        return DNAGPU_ERR_INTERNAL;
    }
    if (extent > out_cap) {
        set_err("super-k-mer count: %llu output slots used, %llu allocated", (unsigned long long)extent, (unsigned long long)out_cap);
        return DNAGPU_ERR_INTERNAL;
    }
    if (total_groups > n) {
        set_err("super-k-mer count: %llu groups for %llu rows", (unsigned long long)total_groups, (unsigned long long)n);
        return DNAGPU_ERR_INTERNAL;
    }
    h->total = n;
    h->n_distinct = total_groups;
    h->extent = extent;
    h->keys = ok;
    h->counts = oc;
    h->seg_off = seg_off;
    h->seg_cnt = seg_cnt;
    h->n_segs = n_segs;
    h->sorted = false;
    ps.release(ok);
    ps.release(oc);
    ps.release(seg_off);
    ps.release(seg_cnt);
    return DNAGPU_OK;
}

// short k-mers (2k <= dense_max_bits()) of enough rows to pay for the table passes and for compacting the table: no tree
static bool dense_pays(u64 n, int k)
{
    return 2 * k <= dense_max_bits() && n > (u64)LEAF_CAP && n >= ((u64)(2 * k > 16 ? 64 : 4) << (2 * k));
}

// any_order: the caller does not need ascending keys across the whole result (dnagpu_count_kmers_unordered): long
// k-mers of long sequences then go through the super-k-mer engine
constexpr u64 SK_MIN_ROWS = (u64)1 << 25;
// the engine pays once the runs are long enough (mean (k - 13) / 2 k-mers per record) and the sequence is: measured at
// 1 Gbase, tree vs this engine: k = 23 13.4 vs 13.1 ms, 24 13.2 vs 12.4, 25 13.2 vs 12.0, 27 13.2 vs 11.5, 29 13.0 vs 10.8;
// k = 31: 16 Mbase 0.63 vs 0.63 ms, 64 Mbase 1.32 vs 1.11, 250 Mbase 3.70 vs 3.21, 3 Gbase 42.0 vs 30.5
// k = 21 and 22 (13-base minimizers: runs of 5 - 5.5 k-mers; the multi-GPU record exchange uses the engine from k = 21
// whatever the size) gain less, and lose on the longest sequences, where the tree's passes run at their best
// (tools/engine_probe.py, tree vs records: k = 21 0.93 vs 0.91 ms at 50 Mbase, 1.64 vs 1.58 at 100 Mbase, 3.53 vs 3.34 at
// 250 Mbase, 13.1 vs 13.2 at 1 Gbase, 42.2 vs 44.7 at 3 Gbase; k = 22 0.90 vs 0.86, 1.62 vs 1.50, 3.52 vs 3.12, 13.1 vs 12.2,
// 42.1 vs 43.4; k = 23 1.62 vs 1.43, 13.0 vs 11.7, 42.1 vs 36.8): they take the engine up to 2^29 (k = 21) and 2^31 (k = 22) rows
constexpr int SK_MIN_K = 23;
static bool sk_is_default(u64 n, int k)
{
    if (n < SK_MIN_ROWS)
        return false;
    if (k >= SK_MIN_K)
        return true;
    // (k = 20: 12-base minimizers keep the final buckets even up to ~2^28 rows)
    return (k == 22 && n <= ((u64)1 << 31)) || (k == 21 && n <= ((u64)1 << 29)) || (k == 20 && n <= ((u64)1 << 28));
}
static int count_core(dnagpu_ctx *ctx, const dnagpu_dna *dna, u64 first, u64 n, int k, u64 *keys_in,
                      dnagpu_hist **out, int fixed_bits = 0, u64 fixed_prefix = 0, int owner = 0, int n_owners = 1,
                      bool any_order = false)
{
    if (n > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    dnagpu_hist *h = new (std::nothrow) dnagpu_hist{nullptr, nullptr, 0, n, nullptr, nullptr, nullptr, 0, true};
    if (!h)
        return DNAGPU_ERR_OOM;
    if (n == 0) {
        *out = h;
        return DNAGPU_OK;
    }
    prof_begin(ctx);
    int rc;
    const bool force_sk = (ctx->debug_flags & DNAGPU_DEBUG_FORCE_SUPERKMER) && k >= sk_min_k() && n >= 64;
    if (any_order && dna && fixed_bits == 0 && n_owners == 1 && (sk_is_default(n, k) || force_sk)) {
        rc = count_sk(ctx, dna, first, n, k, h, n);
        if (rc != DNAGPU_SK_SKEWED) {
            prof_end(ctx);
            if (rc != DNAGPU_OK) {
                delete h;
                return rc;
            }
            *out = h;
            return DNAGPU_OK;
        }
        prof_begin(ctx);                         // a bucket too heavy for the engine: the ordinary tree from scratch
    }
    if (dna && n_owners == 1 && fixed_bits == 0 && dense_pays(n, k)) {
        // short k-mers: the histogram is a table of at most 262,144 counters filled straight from the
        // packed sequence (no key is ever written); one segment, keys ascending
        PoolScope ps(ctx);
        const int bits = 2 * k;
        const size_t n_bins = (size_t)1 << bits;
        u32 *table = nullptr, *oc = nullptr, *seg_cnt = nullptr;
        u64 *ok = nullptr, *seg_off = nullptr, *n_out = nullptr;
        rc = ps.alloc(n_bins, &table);
        if (rc == DNAGPU_OK) rc = ps.alloc(n_bins, &ok);
        if (rc == DNAGPU_OK) rc = ps.alloc(n_bins, &oc);
        if (rc == DNAGPU_OK) rc = ps.alloc(1, &seg_off);
        if (rc == DNAGPU_OK) rc = ps.alloc(1, &seg_cnt);
        if (rc == DNAGPU_OK) rc = ps.alloc(1, &n_out);
        u64 D = 0;
        if (rc == DNAGPU_OK) {
            prof_mark(ctx, "dense_count");
            hipError_t e = launch_dense_count(dna->words, dna->n_words, first, n, bits, table, ok, oc, n_out, ctx->stream);
            prof_mark(ctx, "end");
            if (e == hipSuccess)
                e = hipMemcpyAsync(&D, n_out, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(ctx->stream);
            const u64 zero = 0;
            const u32 d32 = (u32)D;
            if (e == hipSuccess)
                e = hipMemcpyAsync(seg_off, &zero, 8, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(seg_cnt, &d32, 4, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) {
                set_err("dense count: %s", hipGetErrorString(e));
                rc = DNAGPU_ERR_HIP;
            }
        }
        if (rc == DNAGPU_OK) {
            h->total = n;
            h->n_distinct = D;
            h->keys = ok;
            h->counts = oc;
            h->seg_off = seg_off;
            h->seg_cnt = seg_cnt;
            h->n_segs = 1;
            ps.release(ok);
            ps.release(oc);
            ps.release(seg_off);
            ps.release(seg_cnt);
        }
    } else {
        PoolScope ps(ctx);
        TreeResult tr;
        if (n_owners > 1) {
            // sharded count: level 0 is forced onto the owner digits and keeps only this owner's keys
            const int obits = std::min(2 * k, MAX_SPLIT_BITS);
            const u32 R = 1u << obits;
            const u32 d_lo = (u32)(((u64)owner * R + n_owners - 1) / n_owners);
            const u32 d_hi = (u32)(((u64)(owner + 1) * R + n_owners - 1) / n_owners);
            const u32 span = d_hi > d_lo ? d_hi - d_lo : 0u;
            // aligned power-of-two range (every power-of-two GPU count): the kernels test top bits only
            u32 tb = 0;
            if (span && (span & (span - 1)) == 0 && d_lo % span == 0 && span < R) {
                int lg = 0;
                while ((1u << lg) < span)
                    lg++;
                tb = (u32)(obits - lg);
            }
            rc = run_tree(ctx, ps, dna, first, n, k, keys_in, obits, &tr, 0, 0, false, d_lo, span, tb);
        } else {
            rc = run_tree(ctx, ps, dna, first, n, k, keys_in, 0, &tr, fixed_bits, fixed_prefix);
        }
        u64 *cursor = nullptr, *seg_off = nullptr;
        u32 *seg_cnt = nullptr;
        u64 *ok = nullptr;
        u32 *oc = nullptr;
        // a k-mer of k bases has at most 4^k distinct values
        u64 cap = rc == DNAGPU_OK ? std::max<u64>(tr.n_keys, 1) : 1;
        if (k < 16)
            cap = std::min<u64>(cap, (u64)1 << (2 * k));
        if (rc == DNAGPU_OK) rc = ps.alloc(1, &cursor);
        if (rc == DNAGPU_OK) rc = ps.alloc(tr.n_nodes, &seg_off);
        if (rc == DNAGPU_OK) rc = ps.alloc(tr.n_nodes, &seg_cnt);
        if (rc == DNAGPU_OK) rc = ps.alloc((size_t)cap, &ok);
        if (rc == DNAGPU_OK) rc = ps.alloc((size_t)cap, &oc);
        u32 *flags = nullptr, *scan_tmp = nullptr, *cls_list = nullptr;
        if (rc == DNAGPU_OK && tr.n_tiny != tr.n_nodes && tr.n_small != tr.n_nodes && tr.n_big != tr.n_nodes) {
            // a mixed node list: single-key / empty nodes are emitted in bulk, each leaf class gets an index list
            rc = ps.alloc((size_t)tr.n_nodes + 1, &flags);
            if (rc == DNAGPU_OK) rc = ps.alloc((size_t)scan_tmp_words(tr.n_nodes), &scan_tmp);
            if (rc == DNAGPU_OK) rc = ps.alloc((size_t)tr.n_nodes, &cls_list);
        }
        hipError_t e = hipSuccess;
        u64 total_groups = 0;
        if (rc == DNAGPU_OK) {
            prof_mark(ctx, "leaves");
            e = hipMemsetAsync(cursor, 0, 8, ctx->stream);
            if (e == hipSuccess)
                e = launch_leaves(tr.nodes, tr.n_nodes, tr.n_tiny, tr.n_small, tr.n_big, tr.buf0, tr.buf1, cursor, seg_off, seg_cnt,
                                  ok, oc, flags, scan_tmp, cls_list, ctx->stream, false);
            prof_mark(ctx, "end");
            if (e == hipSuccess)
                e = hipMemcpyAsync(ctx->mailbox, cursor, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(ctx->stream);
            total_groups = ctx->mailbox[0];
            if (e != hipSuccess) {
                set_err("leaves: %s", hipGetErrorString(e));
                rc = DNAGPU_ERR_HIP;
            } else if (total_groups > cap) {
                set_err("leaves: %llu groups exceed the output capacity %llu", (unsigned long long)total_groups,
                        (unsigned long long)cap);
                rc = DNAGPU_ERR_INTERNAL;
            }
        }
        if (rc == DNAGPU_OK) {
            h->total = tr.n_keys;               // rows counted (an owner filter keeps only this owner's rows)
            h->n_distinct = total_groups;
            h->keys = ok;
            h->counts = oc;
            h->seg_off = seg_off;
            h->seg_cnt = seg_cnt;
            h->n_segs = tr.n_nodes;
            h->sorted = true;                   // segments in ascending key order
            ps.release(ok);
            ps.release(oc);
            ps.release(seg_off);
            ps.release(seg_cnt);
        }
    }
    prof_end(ctx);
    if (rc != DNAGPU_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return DNAGPU_OK;
}

extern "C" int dnagpu_count_kmers(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, uint64_t first,
                                  uint64_t count, dnagpu_hist **out)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !out)
        return DNAGPU_ERR_BAD_ARG;
    RC_TRY(check_range(dna, k, first, count));
    HIP_TRY(hipSetDevice(ctx->device));
    return count_core(ctx, dna, first, count, k, nullptr, out);
    });
}

extern "C" int dnagpu_count_kmers_unordered(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, uint64_t first,
                                            uint64_t count, dnagpu_hist **out)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !out)
        return DNAGPU_ERR_BAD_ARG;
    RC_TRY(check_range(dna, k, first, count));
    HIP_TRY(hipSetDevice(ctx->device));
    return count_core(ctx, dna, first, count, k, nullptr, out, 0, 0, 0, 1, true);
    });
}

// ---- GROUP BY kmer, count(*) FROM a table of sequences, LATERAL generate_kmers(sequence, k) (test.sql:140-150)
// the rows of a table: every sequence's own generate_kmers rows (none for a sequence shorter than k); BAD_ARG unless the
// starts are ascending from 0 to the stream's length
static int table_rows_host(const dnagpu_dna *dna, const uint64_t *seq_starts, uint64_t n_seqs, int k, u64 *rows_out)
{
    u64 rows = 0;
    for (u64 i = 0; i < n_seqs; i++) {
        if (seq_starts[i + 1] < seq_starts[i])
            return DNAGPU_ERR_BAD_ARG;
        const u64 len = seq_starts[i + 1] - seq_starts[i];
        if (len >= (u64)k)
            rows += len - (u64)k + 1;
    }
    if (n_seqs && (seq_starts[0] != 0 || seq_starts[n_seqs] != dna->n_bases))
        return DNAGPU_ERR_BAD_ARG;
    if (n_seqs == 0 && dna->n_bases != 0)
        return DNAGPU_ERR_BAD_ARG;
    *rows_out = rows;
    return DNAGPU_OK;
}

// the count over a table whose marks are in device memory (the caller's PoolScope or the dna's own)
static int count_table(dnagpu_ctx *ctx, const dnagpu_dna *dna, const u32 *marks, u64 n_mark_words, u64 rows, int k, dnagpu_hist **out)
{
    PoolScope ps(ctx);
    hipStream_t st = ctx->stream;
    const u64 n_windows = dna->n_bases - (u64)k + 1;   // (rows > 0: some sequence has k bases)
    // ---- long k-mers of long tables: the super-k-mer engine, its level 0 blind to the rows across sequence starts
    const bool force_sk = (ctx->debug_flags & DNAGPU_DEBUG_FORCE_SUPERKMER) && k >= sk_min_k() && rows >= 64;
    if (sk_is_default(rows, k) || force_sk) {
        dnagpu_hist *h = new (std::nothrow) dnagpu_hist{nullptr, nullptr, 0, rows, nullptr, nullptr, nullptr, 0, true};
        if (!h)
            return DNAGPU_ERR_OOM;
        prof_begin(ctx);
        ctx->batch_marks = marks;
        ctx->batch_mark_words = n_mark_words;
        const int rc = count_sk(ctx, dna, 0, n_windows, k, h, rows);
        ctx->batch_marks = nullptr;
        ctx->batch_mark_words = 0;
        prof_end(ctx);
        if (rc == DNAGPU_OK) {
            *out = h;
            return DNAGPU_OK;
        }
        delete h;
        if (rc != DNAGPU_SK_SKEWED)
            return rc;                             // (else: the keys below)
    }
    // ---- every other case: the keys of the table's rows, compacted, then the ordinary count over keys
    u64 *keys = nullptr;
    unsigned long long *cursor = nullptr;
    RC_TRY(ps.alloc((size_t)rows, &keys));
    RC_TRY(ps.alloc(1, &cursor));
    HIP_TRY(launch_batch_keys(dna->words, dna->n_words, marks, n_mark_words, n_windows, k, keys,
                              cursor, st));
    u64 got = 0;
    RC_TRY(read_back(ctx, &got, cursor, 8));
    if (got != rows) {
        set_err("table count: %llu rows kept, %llu expected", (unsigned long long)got, (unsigned long long)rows);
        return DNAGPU_ERR_INTERNAL;
    }
    return count_core(ctx, nullptr, 0, rows, k, keys, out);
}

extern "C" int dnagpu_count_kmers_batch(dnagpu_ctx *ctx, const dnagpu_dna *dna, const uint64_t *seq_starts, uint64_t n_seqs,
                                        int k, dnagpu_hist **out)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !out || (n_seqs && !seq_starts))
        return DNAGPU_ERR_BAD_ARG;
    if (k < 1 || k > 32)
        return DNAGPU_ERR_INVALID_K;               // dna.c:771-773, raised by the first row's generate_kmers call
    *out = nullptr;
    u64 rows = 0;
    RC_TRY(table_rows_host(dna, seq_starts, n_seqs, k, &rows));
    if (rows > 0xFFFFFFFFull || dna->n_bases > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    HIP_TRY(hipSetDevice(ctx->device));
    if (rows == 0)
        return count_core(ctx, nullptr, 0, 0, k, nullptr, out);
    if (n_seqs == 1)                               // one sequence: the plain count
        return count_core(ctx, dna, 0, rows, k, nullptr, out, 0, 0, 0, 1, true);
    PoolScope ps(ctx);
    hipStream_t st = ctx->stream;
    // ---- the marks: one bit per base, set where a sequence starts
    const u64 n_mark_words = dna->n_bases / 32 + 3;
    u32 *marks = nullptr;
    u64 *d_starts = nullptr;
    RC_TRY(ps.alloc((size_t)n_mark_words, &marks));
    RC_TRY(ps.alloc((size_t)n_seqs + 1, &d_starts));
    HIP_TRY(hipMemcpyAsync(d_starts, seq_starts, (size_t)(n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(launch_batch_marks(d_starts, n_seqs, marks, n_mark_words, st));
    HIP_TRY(hipStreamSynchronize(st));             // (seq_starts is the caller's: not kept behind the call)
    return count_table(ctx, dna, marks, n_mark_words, rows, k, out);
    });
}

// The table's boundaries made resident: validated, uploaded, and the marks built ONCE; dnagpu_count_kmers_table then counts
// it for any k with nothing crossing the bus (at 10^7 reads the starts are 80 MB: 5.8 ms of a 14.4 ms call).
extern "C" int dnagpu_dna_set_sequences(dnagpu_ctx *ctx, dnagpu_dna *dna, const uint64_t *seq_starts, uint64_t n_seqs)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || (n_seqs && !seq_starts))
        return DNAGPU_ERR_BAD_ARG;
    u64 rows1 = 0;
    RC_TRY(table_rows_host(dna, seq_starts, n_seqs, 1, &rows1));
    if (dna->n_bases > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    HIP_TRY(hipSetDevice(ctx->device));
    if (dna->seq_starts)
        pool_free(ctx, dna->seq_starts);
    if (dna->seq_marks)
        pool_free(ctx, dna->seq_marks);
    dna->seq_starts = nullptr;
    dna->seq_marks = nullptr;
    dna->n_seqs = dna->n_mark_words = 0;
    if (n_seqs == 0)
        return DNAGPU_OK;                          // (an empty table over an empty stream: nothing to keep)
    hipStream_t st = ctx->stream;
    const u64 n_mark_words = dna->n_bases / 32 + 3;
    void *ds = nullptr, *dm = nullptr;
    RC_TRY(pool_alloc(ctx, (size_t)(n_seqs + 1) * 8, &ds));
    int rc = pool_alloc(ctx, (size_t)n_mark_words * 4, &dm);
    if (rc != DNAGPU_OK) {
        pool_free(ctx, ds);
        return rc;
    }
    hipError_t e = hipMemcpyAsync(ds, seq_starts, (size_t)(n_seqs + 1) * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = launch_batch_marks(static_cast<const u64 *>(ds), n_seqs, static_cast<u32 *>(dm), n_mark_words, st);
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);              // (seq_starts is the caller's: not kept behind the call)
    if (e != hipSuccess) {
        pool_free(ctx, ds);
        pool_free(ctx, dm);
        set_err("set_sequences: %s", hipGetErrorString(e));
        return DNAGPU_ERR_HIP;
    }
    dna->seq_starts = static_cast<u64 *>(ds);
    dna->seq_marks = static_cast<u32 *>(dm);
    dna->n_seqs = n_seqs;
    dna->n_mark_words = n_mark_words;
    return DNAGPU_OK;
    });
}

extern "C" uint64_t dnagpu_dna_sequences(const dnagpu_dna *dna) { return dna ? dna->n_seqs : 0; }

extern "C" int dnagpu_count_kmers_table(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, dnagpu_hist **out)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !out)
        return DNAGPU_ERR_BAD_ARG;
    if (k < 1 || k > 32)
        return DNAGPU_ERR_INVALID_K;
    *out = nullptr;
    if (dna->n_seqs == 0 && dna->n_bases != 0)
        return DNAGPU_ERR_BAD_ARG;                 // (no dnagpu_dna_set_sequences before)
    HIP_TRY(hipSetDevice(ctx->device));
    u64 rows = 0;
    if (dna->n_seqs) {
        PoolScope ps(ctx);
        u64 *d_rows = nullptr;
        RC_TRY(ps.alloc(1, &d_rows));
        HIP_TRY(launch_batch_rows(dna->seq_starts, dna->n_seqs, k, d_rows, ctx->stream));
        RC_TRY(read_back(ctx, &rows, d_rows, 8));
    }
    if (rows > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    if (rows == 0)
        return count_core(ctx, nullptr, 0, 0, k, nullptr, out);
    if (dna->n_seqs == 1)
        return count_core(ctx, dna, 0, rows, k, nullptr, out, 0, 0, 0, 1, true);
    return count_table(ctx, dna, dna->seq_marks, dna->n_mark_words, rows, k, out);
    });
}

extern "C" int dnagpu_hist_is_sorted(const dnagpu_hist *h) { return h && h->sorted ? 1 : 0; }

// ---- the two halves of the unordered count, for a count whose rows live on several GPUs: records of a rank's own rows,
// grouped by coarse bucket (to be shipped to the buckets' owners), and the count of the records a rank has received
struct dnagpu_records {
    void *recs;                 // pool memory: 16 bytes per record, bucket after bucket
    std::vector<u64> off;       // n_buckets + 1 offsets (records)
};

extern "C" int dnagpu_sk_buckets(const dnagpu_ctx *ctx, uint64_t global_rows, int k)
{
    if (!ctx || k < sk_min_k() || k > 32 || global_rows == 0 || global_rows > 0xFFFFFFFFull)
        return 0;
    return (int)sk_geometry(ctx, global_rows, k).c0n;
}

extern "C" int dnagpu_sk_records(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, uint64_t first, uint64_t count,
                                 uint64_t global_rows, dnagpu_records **out)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !out || k < sk_min_k() || k > 32 || global_rows < count || global_rows == 0 || global_rows > 0xFFFFFFFFull)
        return DNAGPU_ERR_BAD_ARG;
    RC_TRY(check_range(dna, k, first, count));
    HIP_TRY(hipSetDevice(ctx->device));
    const SkGeom g = sk_geometry(ctx, global_rows, k);
    dnagpu_records *r = new (std::nothrow) dnagpu_records();
    if (!r)
        return DNAGPU_ERR_OOM;
    r->recs = nullptr;
    r->off.assign((size_t)g.c0n + 1, 0);
    if (count > 0) {
        PoolScope ps(ctx);
        void *rec0 = nullptr;
        Node *coarse = nullptr;
        u32 n_coarse = 0;
        u64 n_recs = 0;
        std::vector<Node> kids;
        prof_begin(ctx);
        const int rc = sk_level0(ctx, ps, dna, first, count, k, g, &rec0, &coarse, &n_coarse, &kids, &n_recs);
        prof_mark(ctx, "end");
        prof_end(ctx);
        if (rc != DNAGPU_OK) {
            delete r;
            return rc;
        }
        for (u32 d = 0; d < g.c0n; d++)
            r->off[d + 1] = r->off[d] + (d < kids.size() ? kids[d].len : 0);
        const hipError_t se = hipStreamSynchronize(ctx->stream);
        if (r->off[g.c0n] != n_recs || se != hipSuccess) {
            set_err("super-k-mer level 0: %llu records in the buckets, %llu counted (%s)", (unsigned long long)r->off[g.c0n],
                    (unsigned long long)n_recs, hipGetErrorString(se));
            delete r;
            return se != hipSuccess ? DNAGPU_ERR_HIP : DNAGPU_ERR_INTERNAL;
        }
        ps.release(rec0);
        r->recs = rec0;
    }
    *out = r;
    return DNAGPU_OK;
    });
}

extern "C" uint32_t dnagpu_records_buckets(const dnagpu_records *r) { return r ? (uint32_t)(r->off.size() - 1) : 0; }
extern "C" int dnagpu_records_offsets(const dnagpu_records *r, uint64_t *offsets)
{
    if (!r || !offsets)
        return DNAGPU_ERR_BAD_ARG;
    for (size_t i = 0; i < r->off.size(); i++)
        offsets[i] = r->off[i];
    return DNAGPU_OK;
}
extern "C" void *dnagpu_records_device(const dnagpu_records *r) { return r ? r->recs : nullptr; }
extern "C" void dnagpu_records_free(dnagpu_ctx *ctx, dnagpu_records *r)
{
    if (!r)
        return;
    if (ctx)
        pool_free(ctx, r->recs);
    delete r;
}

extern "C" int dnagpu_count_records(dnagpu_ctx *ctx, const void *const *pieces, const uint64_t *piece_len,
                                    const uint32_t *piece_bucket, uint32_t n_pieces, int k, uint64_t global_rows, dnagpu_hist **out)
{
    return guarded([&]() -> int {
    if (!ctx || !out || (n_pieces && (!pieces || !piece_len || !piece_bucket)) || k < sk_min_k() || k > 32 || global_rows == 0 ||
        global_rows > 0xFFFFFFFFull)
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    dnagpu_hist *h = new (std::nothrow) dnagpu_hist{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, false};
    if (!h)
        return DNAGPU_ERR_OOM;
    prof_begin(ctx);
    const int rc = count_sk_records(ctx, pieces, piece_len, piece_bucket, n_pieces, k, global_rows, h);
    prof_end(ctx);
    if (rc != DNAGPU_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_count_kmers_owned(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, uint64_t first,
                                        uint64_t count, int owner, int n_owners, dnagpu_hist **out)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !out || n_owners < 1 || owner < 0 || owner >= n_owners)
        return DNAGPU_ERR_BAD_ARG;
    RC_TRY(check_range(dna, k, first, count));
    HIP_TRY(hipSetDevice(ctx->device));
    return count_core(ctx, dna, first, count, k, nullptr, out, 0, 0, owner, n_owners);
    });
}

extern "C" int dnagpu_count_keys(dnagpu_ctx *ctx, uint64_t *dev_keys, uint64_t n, int k, dnagpu_hist **out)
{
    return guarded([&]() -> int {
    if (!ctx || !out || (n && !dev_keys))
        return DNAGPU_ERR_BAD_ARG;
    if (k <= 0 || k > 32)
        return DNAGPU_ERR_INVALID_K;
    HIP_TRY(hipSetDevice(ctx->device));
    return count_core(ctx, nullptr, 0, n, k, dev_keys, out);
    });
}

extern "C" int dnagpu_count_keys_in_range(dnagpu_ctx *ctx, uint64_t *dev_keys, uint64_t n, int k,
                                          uint64_t key_min, uint64_t key_max, dnagpu_hist **out)
{
    return guarded([&]() -> int {
    if (!ctx || !out || (n && !dev_keys) || key_min > key_max)
        return DNAGPU_ERR_BAD_ARG;
    if (k <= 0 || k > 32)
        return DNAGPU_ERR_INVALID_K;
    HIP_TRY(hipSetDevice(ctx->device));
    // number of leading bits (of the 2k key bits) that key_min and key_max share
    const int nbits = 2 * k;
    u64 diff = (key_min ^ key_max) & kmer_mask(k);
    int free_bits = 0;
    while (free_bits < nbits && (diff >> free_bits) != 0)
        free_bits++;
    const int fixed = nbits - free_bits;
    const u64 prefix = free_bits >= 64 ? 0 : (key_min >> free_bits) << free_bits;
    // a single possible key (fixed == 2k) still runs through the generic path: rem = 0 leaf
    return count_core(ctx, nullptr, 0, n, k, dev_keys, out, fixed, prefix);
    });
}

extern "C" uint64_t dnagpu_hist_distinct(const dnagpu_hist *h) { return h ? h->n_distinct : 0; }
extern "C" uint64_t dnagpu_hist_total(const dnagpu_hist *h) { return h ? h->total : 0; }
extern "C" const uint64_t *dnagpu_hist_device_keys(const dnagpu_hist *h) { return h ? h->keys : nullptr; }
extern "C" uint64_t dnagpu_hist_extent(const dnagpu_hist *h) { return !h ? 0 : (h->extent ? h->extent : h->n_distinct); }
extern "C" const uint32_t *dnagpu_hist_device_counts(const dnagpu_hist *h) { return h ? h->counts : nullptr; }
extern "C" uint32_t dnagpu_hist_parts(const dnagpu_hist *h) { return !h ? 0 : (h->parts.empty() ? 1u : (uint32_t)h->parts.size()); }
extern "C" const dnagpu_hist *dnagpu_hist_part(const dnagpu_hist *h, uint32_t i)
{
    if (!h)
        return nullptr;
    if (h->parts.empty())
        return i == 0 ? h : nullptr;
    return i < h->parts.size() ? h->parts[i] : nullptr;
}

// Ascending-key order through the segment directory: groups are gathered on the device into a
// staging window, then copied to the host.
// exclusive scan of the segment sizes, built on the first ordered read of a histogram
static int ensure_seg_pre(dnagpu_ctx *ctx, dnagpu_hist *h)
{
    if (h->seg_pre)
        return DNAGPU_OK;
    PoolScope ps(ctx);
    u32 *pre = nullptr, *tmp = nullptr;
    RC_TRY(pool_alloc_t(ctx, (size_t)h->n_segs + 1, &pre));
    int rc = ps.alloc((size_t)scan_tmp_words(h->n_segs), &tmp);
    hipError_t e = rc == DNAGPU_OK ? launch_scan_u32(h->seg_cnt, pre, h->n_segs, tmp, pre + h->n_segs, ctx->stream)
                                   : hipSuccess;
    if (rc != DNAGPU_OK || e != hipSuccess) {
        pool_free(ctx, pre);
        if (rc == DNAGPU_OK) {
            set_err("segment scan: %s", hipGetErrorString(e));
            rc = DNAGPU_ERR_HIP;
        }
        return rc;
    }
    h->seg_pre = pre;
    return DNAGPU_OK;
}

// groups [first, first + count) of a histogram in its read order = the same range cut along the parts
template <typename F>
static int for_parts(dnagpu_hist *h, u64 first, u64 count, F &&f)
{
    if (h->parts.empty())
        return f(h, first, count, (u64)0);
    u64 base = 0, done = 0;
    for (dnagpu_hist *p : h->parts) {
        const u64 lo = std::max(first, base), hi = std::min(first + count, base + p->n_distinct);
        if (hi > lo) {
            RC_TRY(f(p, lo - base, hi - lo, done));
            done += hi - lo;
        }
        base += p->n_distinct;
    }
    return DNAGPU_OK;
}

extern "C" int dnagpu_hist_sorted_view(dnagpu_ctx *ctx, const dnagpu_hist *h_c, uint64_t first, uint64_t count,
                                       uint64_t *dev_keys, uint64_t *dev_counts)
{
    return guarded([&]() -> int {
    dnagpu_hist *h = const_cast<dnagpu_hist *>(h_c);
    if (!ctx || !h)
        return DNAGPU_ERR_BAD_ARG;
    if (first > h->n_distinct || count > h->n_distinct - first)
        return DNAGPU_ERR_BAD_ARG;
    if (count == 0 || (!dev_keys && !dev_counts))
        return DNAGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    RC_TRY(for_parts(h, first, count, [&](dnagpu_hist *p, u64 pf, u64 pc, u64 out_at) -> int {
        RC_TRY(ensure_seg_pre(ctx, p));
        HIP_TRY(launch_gather_sorted(p->seg_off, p->seg_cnt, p->seg_pre, p->n_segs, pf, pc, p->keys, p->counts,
                                     dev_keys ? dev_keys + out_at : nullptr, dev_counts ? dev_counts + out_at : nullptr, ctx->stream));
        return DNAGPU_OK;
    }));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_hist_download(dnagpu_ctx *ctx, const dnagpu_hist *h_c, uint64_t first, uint64_t count,
                                    uint64_t *keys, uint64_t *counts)
{
    return guarded([&]() -> int {
    dnagpu_hist *h = const_cast<dnagpu_hist *>(h_c);
    if (!ctx || !h)
        return DNAGPU_ERR_BAD_ARG;
    if (first > h->n_distinct || count > h->n_distinct - first)
        return DNAGPU_ERR_BAD_ARG;
    if (count == 0 || (!keys && !counts))
        return DNAGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    PoolScope ps(ctx);
    const u64 BATCH = (u64)1 << 25;            // 32 Mi groups = 2 x 256 MiB staging
    u64 *sk = nullptr, *sc = nullptr;
    if (keys)
        RC_TRY(ps.alloc((size_t)std::min(count, BATCH), &sk));
    if (counts)
        RC_TRY(ps.alloc((size_t)std::min(count, BATCH), &sc));
    return for_parts(h, first, count, [&](dnagpu_hist *p, u64 pf, u64 pc, u64 out_at) -> int {
        RC_TRY(ensure_seg_pre(ctx, p));
        for (u64 done = 0; done < pc; done += BATCH) {
            u64 nb = std::min(BATCH, pc - done);
            HIP_TRY(launch_gather_sorted(p->seg_off, p->seg_cnt, p->seg_pre, p->n_segs, pf + done, nb, p->keys,
                                         p->counts, sk, sc, ctx->stream));
            if (keys)
                HIP_TRY(hipMemcpyAsync(keys + out_at + done, sk, nb * 8, hipMemcpyDeviceToHost, ctx->stream));
            if (counts)
                HIP_TRY(hipMemcpyAsync(counts + out_at + done, sc, nb * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        return DNAGPU_OK;
    });
    });
}

extern "C" int dnagpu_hist_summary(dnagpu_ctx *ctx, const dnagpu_hist *h, uint64_t *total, uint64_t *unique,
                                   uint64_t *checksum)
{
    return guarded([&]() -> int {
    if (!ctx || !h)
        return DNAGPU_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    PoolScope ps(ctx);
    u64 *res = nullptr;
    RC_TRY(ps.alloc(4, &res));
    HIP_TRY(hipMemsetAsync(res, 0, 32, ctx->stream));
    if (h->parts.empty()) {
        HIP_TRY(launch_hist_summary(h->keys, h->counts, h->extent ? h->extent : h->n_distinct, res, ctx->stream));
    } else {
        for (const dnagpu_hist *p : h->parts)       // (the kernel adds into res: wrapping sums over all parts)
            HIP_TRY(launch_hist_summary(p->keys, p->counts, p->extent ? p->extent : p->n_distinct, res, ctx->stream));
    }
    u64 r[3] = {0, 0, 0};
    RC_TRY(read_back(ctx, r, res, 24));
    if (total) *total = r[0];
    if (unique) *unique = r[1];
    if (checksum) *checksum = r[2];
    return DNAGPU_OK;
    });
}

// The groups of a and b added up: equal keys' counts are summed.  Both histograms must live on ctx's device; they are
// left as they are.
extern "C" int dnagpu_hist_merge(dnagpu_ctx *ctx, const dnagpu_hist *a, const dnagpu_hist *b, dnagpu_hist **out)
{
    return guarded([&]() -> int {
    if (!ctx || !a || !b || !out)
        return DNAGPU_ERR_BAD_ARG;
    *out = nullptr;
    if (a->total + b->total > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;               // (counts are 32-bit in device memory)
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    PoolScope ps(ctx);
    const u64 n_max = a->n_distinct + b->n_distinct;
    u64 t_slots = 1024;
    while (t_slots < 2 * n_max)
        t_slots <<= 1;
    u64 *tkeys = nullptr, *ok = nullptr, *seg_off = nullptr;
    u32 *tcnt = nullptr, *oc = nullptr, *seg_cnt = nullptr;
    unsigned long long *ctr = nullptr;             // [0] the all-ones key's count, [1] the groups written
    RC_TRY(ps.alloc((size_t)t_slots, &tkeys));
    RC_TRY(ps.alloc((size_t)t_slots, &tcnt));
    RC_TRY(ps.alloc(2, &ctr));
    RC_TRY(ps.alloc((size_t)std::max<u64>(n_max, 1), &ok));
    RC_TRY(ps.alloc((size_t)std::max<u64>(n_max, 1), &oc));
    RC_TRY(ps.alloc(1, &seg_off));
    RC_TRY(ps.alloc(1, &seg_cnt));
    HIP_TRY(hipMemsetAsync(tkeys, 0xFF, (size_t)t_slots * 8, st));
    HIP_TRY(hipMemsetAsync(tcnt, 0, (size_t)t_slots * 4, st));
    HIP_TRY(hipMemsetAsync(ctr, 0, 16, st));
    for (const dnagpu_hist *h : {a, b}) {
        if (h->parts.empty()) {
            HIP_TRY(launch_merge_insert(h->keys, h->counts, h->extent ? h->extent : h->n_distinct, tkeys, tcnt, t_slots, ctr, st));
        } else {
            for (const dnagpu_hist *p : h->parts)
                HIP_TRY(launch_merge_insert(p->keys, p->counts, p->extent ? p->extent : p->n_distinct, tkeys, tcnt, t_slots, ctr, st));
        }
    }
    HIP_TRY(launch_merge_compact(tkeys, tcnt, t_slots, ok, oc, ctr + 1, st));
    u64 res[2] = {0, 0};
    RC_TRY(read_back(ctx, res, ctr, 16));
    u64 D = res[1];
    if (D > n_max) {
        set_err("hist merge: %llu groups out of %llu", (unsigned long long)D, (unsigned long long)n_max);
        return DNAGPU_ERR_INTERNAL;
    }
    if (res[0]) {                                  // the all-ones key goes last (n_max has room: it was a group of a or b)
        const u64 kk = ~(u64)0;
        const u32 cc = (u32)res[0];
        HIP_TRY(hipMemcpyAsync(ok + D, &kk, 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(oc + D, &cc, 4, hipMemcpyHostToDevice, st));
        D++;
    }
    const u64 zero = 0;
    const u32 d32 = (u32)D;
    if (D > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    HIP_TRY(hipMemcpyAsync(seg_off, &zero, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(seg_cnt, &d32, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    // group order: the table's -- unspecified, like every unordered histogram's (one segment, not ascending inside)
    dnagpu_hist *h = new (std::nothrow) dnagpu_hist{ok, oc, D, a->total + b->total, seg_off, seg_cnt, nullptr, 1, false};
    if (!h)
        return DNAGPU_ERR_OOM;
    ps.release(ok);
    ps.release(oc);
    ps.release(seg_off);
    ps.release(seg_cnt);
    *out = h;
    return DNAGPU_OK;
    });
}

extern "C" void dnagpu_hist_free(dnagpu_ctx *ctx, dnagpu_hist *h)
{
    if (!h)
        return;
    for (dnagpu_hist *p : h->parts)
        dnagpu_hist_free(ctx, p);
    if (ctx) {
        pool_free(ctx, h->keys);
        pool_free(ctx, h->counts);
        pool_free(ctx, h->seg_off);
        pool_free(ctx, h->seg_cnt);
        pool_free(ctx, h->seg_pre);
    }
    delete h;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU step 1: one forced level over the dna root, children grouped by owner
extern "C" int dnagpu_partition_kmers(dnagpu_ctx *ctx, const dnagpu_dna *dna, int k, uint64_t first,
                                      uint64_t count, int n_owners, uint64_t **dev_keys,
                                      uint64_t *owner_offsets)
{
    return guarded([&]() -> int {
    if (!ctx || !dna || !dev_keys || !owner_offsets || n_owners < 1 || n_owners > (1 << MAX_SPLIT_BITS))
        return DNAGPU_ERR_BAD_ARG;
    RC_TRY(check_range(dna, k, first, count));
    if (count > 0xFFFFFFFFull)
        return DNAGPU_ERR_TOO_LARGE;
    HIP_TRY(hipSetDevice(ctx->device));
    *dev_keys = nullptr;
    for (int o = 0; o <= n_owners; o++)
        owner_offsets[o] = 0;
    if (count == 0)
        return DNAGPU_OK;
    const int bits = std::min(2 * k, MAX_SPLIT_BITS);
    const u32 R = 1u << bits;
    prof_begin(ctx);
    PoolScope ps(ctx);
    TreeResult tr;
    RC_TRY(run_tree(ctx, ps, dna, first, count, k, nullptr, bits, &tr));
    if (tr.n_nodes != R) {
        set_err("partition: expected %u children, got %u", R, tr.n_nodes);
        return DNAGPU_ERR_INTERNAL;
    }
    std::vector<Node> kids(R);
    HIP_TRY(hipMemcpyAsync(kids.data(), tr.nodes, (size_t)R * sizeof(Node), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // owner o owns digits d with (d * n_owners) >> bits == o: first digit = ceil(o * R / n_owners)
    for (int o = 0; o <= n_owners; o++) {
        u64 d0 = ((u64)o * R + n_owners - 1) / n_owners;
        owner_offsets[o] = d0 >= R ? count : kids[d0].start;
    }
    if (2 * k == bits) {
        // terminal level: nothing was scattered (children carry key = prefix, count = len); expand
        // is not needed by any caller today: k <= 5 counts run on one GPU
        set_err("partition: k too small to shard (2k <= %d bits)", MAX_SPLIT_BITS);
        return DNAGPU_ERR_BAD_ARG;
    }
    ps.release(tr.buf0);
    *dev_keys = tr.buf0;
    prof_end(ctx);
    return DNAGPU_OK;
    });
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU count from ONE process (what a PostgreSQL backend's glue can call): N contexts, one per
// rank, the sequence resident as contiguous word chunks, one all-gather of the packed words (RCCL
// over xGMI, or peer copies), then every rank counts the key range it owns in its own host thread.
// Same algorithm and ownership rule as the process-per-GPU path of sharded.py (bench.py --gpus N).
#include <dlfcn.h>
#include <pthread.h>
#include <rccl/rccl.h>
#include <signal.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if (lib)
            return true;
        // loaded on demand: a single-GPU backend never maps RCCL
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib)
                break;
        }
        if (!lib)
            return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        Reduce = reinterpret_cast<decltype(Reduce)>(dlsym(lib, "ncclReduce"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(lib, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(lib, "ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllGather || !Reduce || !Send || !Recv || !GroupStart || !GroupEnd || !GetErrorString) {
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        return true;
    }
};
}  // namespace

// One host thread per rank >= 1, kept for the life of the dnagpu_multi (rank 0's work runs on the caller's thread): a
// count drives every rank from its own thread because the level loops read counters back between launches.  No
// exception leaves a worker (std::terminate would take the PostgreSQL backend down): a job that throws marks its rank
// failed.  If the threads cannot be created the ranks' jobs run one after the other on the caller's thread.
struct MultiPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::function<void(int)> *job = nullptr;
    std::vector<int> threw;               // per rank: the job ended in a C++ exception (1 = bad_alloc, 2 = other)
    unsigned long long gen = 0;
    int pending = 0;
    bool stop = false, started = false, serial = false;

    static int run_guarded(const std::function<void(int)> &f, int r) noexcept
    {
        try {
            f(r);
            return 0;
        } catch (const std::bad_alloc &) {
            return 1;
        } catch (...) {
            return 2;
        }
    }
    void worker(int r)
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)> *f = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return stop || gen != seen; });
                if (stop)
                    return;
                seen = gen;
                f = job;
            }
            const int t = run_guarded(*f, r);
            {
                std::lock_guard<std::mutex> lk(mu);
                threw[(size_t)r] = t;
                if (--pending == 0)
                    cv_done.notify_all();
            }
        }
    }
    void start(int n) noexcept
    {
        if (started)
            return;
        started = true;
        // The workers must never run the host program's signal handlers: a PostgreSQL backend's handlers (SIGINT cancel,
        // SIGUSR1 latch, SIGTERM) are not thread-safe, and the kernel may deliver a process-directed signal to ANY thread
        // that does not block it.  Threads inherit the creating thread's mask: every signal is blocked around the creation
        // and the caller's mask restored right after, so the workers block everything for their whole life.
        sigset_t all, old_mask;
        sigfillset(&all);
        const bool masked = pthread_sigmask(SIG_BLOCK, &all, &old_mask) == 0;
        try {
            threw.assign((size_t)n, 0);
            th.reserve((size_t)n);
            for (int r = 1; r < n; r++)
                th.emplace_back(&MultiPool::worker, this, r);
        } catch (...) {
            shutdown();                       // joins the threads that did start
            serial = true;
        }
        if (masked)
            (void)pthread_sigmask(SIG_SETMASK, &old_mask, nullptr);
    }
    // runs f(r) for r = 0 .. n-1, rank 0 here; returns 0, or DNAGPU_ERR_OOM / DNAGPU_ERR_INTERNAL if a job threw
    int run(int n, const std::function<void(int)> &f) noexcept
    {
        start(n);
        int bad = 0;
        if (serial || n == 1) {
            for (int r = 0; r < n; r++)
                bad = std::max(bad, run_guarded(f, r));
        } else {
            {
                std::lock_guard<std::mutex> lk(mu);
                job = &f;
                pending = n - 1;
                gen++;
            }
            cv_go.notify_all();
            bad = run_guarded(f, 0);
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return pending == 0; });
            for (int r = 1; r < n; r++)
                bad = std::max(bad, threw[(size_t)r]);
        }
        return bad == 0 ? DNAGPU_OK : (bad == 1 ? DNAGPU_ERR_OOM : DNAGPU_ERR_INTERNAL);
    }
    void shutdown() noexcept
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_go.notify_all();
        for (std::thread &t : th)
            if (t.joinable())
                t.join();
        th.clear();
        stop = false;
    }
};

struct dnagpu_multi {
    MultiPool workers;
    int n;
    std::vector<int> dev;
    std::vector<dnagpu_ctx *> ctx;
    bool rccl;
    RcclApi api;
    std::vector<ncclComm_t> comms;
    dnagpu_multi_times last{};            // host clock of the most recent dnagpu_count_multi_unordered
    std::vector<hipStream_t> xfer;        // per rank: the stream its inbound record copies are queued on
    int parts = DNAGPU_MULTI_DEFAULT_PARTS;   // bucket groups per owner of the pipelined exchange
    double emulate_gbs = 0;               // rehearsal: same-device "transfers" are held to this rate (0 = off)
    int probe_owner = -1;                 // rehearsal: only this owner pulls and counts (-1 = all), so that its time is its own
    int exchange_rccl = 0;                // record exchange: 0 = owners pull with peer copies, 1 = ncclSend / ncclRecv per piece,
                                          // 2 = as 1 and a rank's own pieces travel through RCCL too (tests with one rank)
    const char *last_exchange = "none";   // what the most recent dnagpu_count_multi_unordered moved its records with
    std::vector<dnagpu_phase_times> rec_phases;   // per rank: device phases of its record pass (most recent unordered count)
};

struct dnagpu_multi_dna {
    u64 n_bases, n_words, per;            // per = words per rank chunk; every rank's buffer holds per * n words
    std::vector<u64 *> full;              // rank r: chunk r resident at full[r] + r * per; the rest is gather space
    std::vector<dnagpu_dna *> view;       // full[r] as a dnagpu_dna of n_bases bases
};

extern "C" void dnagpu_multi_destroy(dnagpu_multi *m)
{
    if (!m)
        return;
    m->workers.shutdown();
    for (size_t r = 0; r < m->xfer.size(); r++)
        if (hipSetDevice(m->ctx[r]->device) == hipSuccess) {
            (void)hipStreamSynchronize(m->xfer[r]);
            (void)hipStreamDestroy(m->xfer[r]);
        }
    for (size_t r = 0; r < m->comms.size(); r++)
        if (m->comms[r])
            m->api.CommDestroy(m->comms[r]);
    for (dnagpu_ctx *c : m->ctx)
        dnagpu_destroy(c);
    delete m;
}

extern "C" int dnagpu_multi_init(const int *devices, int n_gpus, int transport, dnagpu_multi **out)
{
    return guarded([&]() -> int {
    if (!out || n_gpus < 1 || n_gpus > 64 || transport < DNAGPU_MULTI_AUTO || transport > DNAGPU_MULTI_COPY)
        return DNAGPU_ERR_BAD_ARG;
    *out = nullptr;
    dnagpu_multi *m = new (std::nothrow) dnagpu_multi();
    if (!m)
        return DNAGPU_ERR_OOM;
    m->n = n_gpus;
    m->rccl = false;
    bool distinct = true;
    for (int r = 0; r < n_gpus; r++) {
        const int d = devices ? devices[r] : r;
        for (int q = 0; q < r; q++)
            distinct = distinct && m->dev[q] != d;
        m->dev.push_back(d);
    }
    for (int r = 0; r < n_gpus; r++) {
        dnagpu_ctx *c = nullptr;
        const int rc = dnagpu_init(m->dev[r], &c);
        if (rc != DNAGPU_OK) {
            dnagpu_multi_destroy(m);
            return rc;
        }
        m->ctx.push_back(c);
        // the rank's transfer stream, made right behind its context's stream: the runtime hands its hardware queues out
        // round-robin in creation order, and two streams on one queue would not overlap (seen in the one-device rehearsal
        // with eight ranks: the pipelined exchange hid nothing when the streams were made in two batches)
        hipStream_t xs = nullptr;
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipStreamCreateWithFlags(&xs, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            dnagpu_multi_destroy(m);
            return DNAGPU_ERR_HIP;
        }
        m->xfer.push_back(xs);
    }
    // peer access for the copy transport and for RCCL's direct xGMI paths (failure is not fatal: copies stage)
    for (int a = 0; a < n_gpus; a++)
        for (int b = 0; b < n_gpus; b++)
            if (m->dev[a] != m->dev[b]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, m->dev[a], m->dev[b]) == hipSuccess && can) {
                    (void)hipSetDevice(m->dev[a]);
                    const hipError_t e = hipDeviceEnablePeerAccess(m->dev[b], 0);
                    if (e != hipSuccess)
                        (void)hipGetLastError();      // already enabled, or not supported
                }
            }
    const bool want_rccl = transport == DNAGPU_MULTI_RCCL || (transport == DNAGPU_MULTI_AUTO && n_gpus > 1 && distinct);
    if (want_rccl) {
        if (!distinct) {
            set_err("RCCL transport needs %d distinct devices", n_gpus);
            dnagpu_multi_destroy(m);
            return DNAGPU_ERR_BAD_ARG;
        }
        if (!m->api.load()) {
            if (transport == DNAGPU_MULTI_RCCL) {
                set_err("librccl.so could not be loaded: %s", dlerror());
                dnagpu_multi_destroy(m);
                return DNAGPU_ERR_HIP;
            }
        } else {
            m->comms.assign((size_t)n_gpus, nullptr);
            const ncclResult_t nr = m->api.CommInitAll(m->comms.data(), n_gpus, m->dev.data());
            if (nr != ncclSuccess) {
                set_err("ncclCommInitAll: %s", m->api.GetErrorString(nr));
                m->comms.clear();
                (void)hipGetLastError();
                if (transport == DNAGPU_MULTI_RCCL) {
                    dnagpu_multi_destroy(m);
                    return DNAGPU_ERR_HIP;
                }
                // DNAGPU_MULTI_AUTO: "RCCL when ... the library loads, else copies" -- a communicator that cannot be made
                // (e.g. another ROCm runtime already in the process) leaves the copy transport
            } else {
                m->rccl = true;
            }
        }
    }
    *out = m;
    return DNAGPU_OK;
    });
}

extern "C" int dnagpu_multi_size(const dnagpu_multi *m) { return m ? m->n : 0; }
extern "C" dnagpu_ctx *dnagpu_multi_ctx(dnagpu_multi *m, int rank)
{
    return (m && rank >= 0 && rank < m->n) ? m->ctx[(size_t)rank] : nullptr;
}
extern "C" const char *dnagpu_multi_transport(const dnagpu_multi *m) { return !m ? "" : (m->rccl ? "rccl" : "copy"); }
extern "C" const char *dnagpu_multi_exchange_transport(const dnagpu_multi *m) { return !m ? "" : m->last_exchange; }
extern "C" int dnagpu_multi_rccl_ranks(const dnagpu_multi *m) { return (m && m->rccl) ? m->n : 0; }
extern "C" int dnagpu_multi_last_phase_times(dnagpu_multi *m, int rank, dnagpu_phase_times *out)
{
    if (!m || !out || rank < 0 || rank >= m->n)
        return DNAGPU_ERR_BAD_ARG;
    // the record pass's phases (kept by the call: the owner phase starts a new session on the rank's context), then the
    // owner phase's
    dnagpu_phase_times t{};
    if ((size_t)rank < m->rec_phases.size())
        t = m->rec_phases[(size_t)rank];
    const dnagpu_phase_times &o = m->ctx[(size_t)rank]->last_times;
    for (int i = 0; i < o.n && t.n < DNAGPU_MAX_PHASES; i++) {
        t.names[t.n] = o.names[i];
        t.ms[t.n] = o.ms[i];
        t.n++;
    }
    *out = t;
    return DNAGPU_OK;
}
extern "C" int dnagpu_multi_last_times(const dnagpu_multi *m, dnagpu_multi_times *out)
{
    if (!m || !out)
        return DNAGPU_ERR_BAD_ARG;
    *out = m->last;
    return DNAGPU_OK;
}

extern "C" void dnagpu_multi_dna_free(dnagpu_multi *m, dnagpu_multi_dna *d)
{
    if (!d)
        return;
    for (size_t r = 0; r < d->view.size(); r++)
        if (d->view[r])
            dnagpu_dna_free(m ? m->ctx[r] : nullptr, d->view[r]);
    if (m)
        for (size_t r = 0; r < d->full.size(); r++)
            pool_free(m->ctx[r], d->full[r]);
    delete d;
}

// allocates every rank's buffer and wraps it; fill(r, w_lo, w_hi) makes rank r's own chunk resident
template <typename Fill>
static int multi_dna_make(dnagpu_multi *m, u64 n_bases, dnagpu_multi_dna **out, Fill &&fill)
{
    dnagpu_multi_dna *d = new (std::nothrow) dnagpu_multi_dna();
    if (!d)
        return DNAGPU_ERR_OOM;
    d->n_bases = n_bases;
    d->n_words = words_for(n_bases);
    d->per = (d->n_words + (u64)m->n - 1) / (u64)m->n;
    if (d->per == 0)
        d->per = 1;
    int rc = DNAGPU_OK;
    for (int r = 0; r < m->n && rc == DNAGPU_OK; r++) {
        dnagpu_ctx *c = m->ctx[(size_t)r];
        hipError_t e = hipSetDevice(c->device);
        u64 *buf = nullptr;
        if (e == hipSuccess)
            rc = pool_alloc_t(c, (size_t)(d->per * (u64)m->n), &buf);
        if (e != hipSuccess || rc != DNAGPU_OK) {
            if (e != hipSuccess) {
                set_err("hipSetDevice: %s", hipGetErrorString(e));
                rc = DNAGPU_ERR_HIP;
            }
            break;
        }
        d->full.push_back(buf);
        d->view.push_back(nullptr);
        const u64 lo = std::min((u64)r * d->per, d->n_words), hi = std::min((u64)(r + 1) * d->per, d->n_words);
        // gather space behind the last word of the sequence stays zero (never read as bases: n_words bounds every sweep)
        e = hipMemsetAsync(buf + d->n_words, 0, (size_t)(d->per * (u64)m->n - d->n_words) * 8, c->stream);
        if (e == hipSuccess)
            e = fill(r, c, buf, lo, hi);
        if (e != hipSuccess) {
            set_err("multi dna: %s", hipGetErrorString(e));
            rc = DNAGPU_ERR_HIP;
            break;
        }
        rc = dnagpu_dna_wrap(c, buf, d->per * (u64)m->n, n_bases, &d->view[(size_t)r]);
    }
    for (int r = 0; r < m->n && rc == DNAGPU_OK; r++)
        if (hipSetDevice(m->ctx[(size_t)r]->device) != hipSuccess || hipStreamSynchronize(m->ctx[(size_t)r]->stream) != hipSuccess)
            rc = DNAGPU_ERR_HIP;
    if (rc != DNAGPU_OK) {
        dnagpu_multi_dna_free(m, d);
        return rc;
    }
    *out = d;
    return DNAGPU_OK;
}

extern "C" int dnagpu_multi_dna_upload(dnagpu_multi *m, const uint64_t *words, uint64_t n_bases, dnagpu_multi_dna **out)
{
    return guarded([&]() -> int {
    if (!m || !out || (n_bases && !words))
        return DNAGPU_ERR_BAD_ARG;
    return multi_dna_make(m, n_bases, out, [&](int, dnagpu_ctx *c, u64 *buf, u64 lo, u64 hi) -> hipError_t {
        if (hi <= lo)
            return hipSuccess;
        return hipMemcpyAsync(buf + lo, words + lo, (size_t)(hi - lo) * 8, hipMemcpyHostToDevice, c->stream);
    });
    });
}

extern "C" int dnagpu_multi_dna_synth(dnagpu_multi *m, uint64_t seed, uint64_t n_bases, uint64_t motif_len,
                                      dnagpu_multi_dna **out)
{
    return guarded([&]() -> int {
    if (!m || !out)
        return DNAGPU_ERR_BAD_ARG;
    return multi_dna_make(m, n_bases, out, [&](int, dnagpu_ctx *c, u64 *buf, u64 lo, u64 hi) -> hipError_t {
        return launch_synth(buf, lo, hi, n_bases, seed, motif_len, c->stream);
    });
    });
}

extern "C" uint64_t dnagpu_multi_dna_length(const dnagpu_multi_dna *d) { return d ? d->n_bases : 0; }

// every rank's buffer receives the other ranks' chunks, ordered on each rank's own stream
static int multi_gather(dnagpu_multi *m, const dnagpu_multi_dna *d)
{
    if (m->n == 1)
        return DNAGPU_OK;
    const size_t per_bytes = (size_t)d->per * 8;
    if (m->rccl) {
        ncclResult_t nr = m->api.GroupStart();
        for (int r = 0; r < m->n && nr == ncclSuccess; r++)     // in place: send = recv + rank * count
            nr = m->api.AllGather(d->full[(size_t)r] + (u64)r * d->per, d->full[(size_t)r], (size_t)d->per, ncclUint64,
                                  m->comms[(size_t)r], m->ctx[(size_t)r]->stream);
        const ncclResult_t ne = m->api.GroupEnd();
        if (nr != ncclSuccess || ne != ncclSuccess) {
            set_err("ncclAllGather: %s", m->api.GetErrorString(nr != ncclSuccess ? nr : ne));
            return DNAGPU_ERR_HIP;
        }
        return DNAGPU_OK;
    }
    for (int dst = 0; dst < m->n; dst++) {
        dnagpu_ctx *c = m->ctx[(size_t)dst];
        HIP_TRY(hipSetDevice(c->device));
        for (int q = 1; q < m->n; q++) {                          // start at the neighbour: spreads the link load
            const int src = (dst + q) % m->n;
            u64 *to = d->full[(size_t)dst] + (u64)src * d->per;
            const u64 *from = d->full[(size_t)src] + (u64)src * d->per;
            if (m->dev[(size_t)src] == m->dev[(size_t)dst])
                HIP_TRY(hipMemcpyAsync(to, from, per_bytes, hipMemcpyDeviceToDevice, c->stream));
            else
                HIP_TRY(hipMemcpyPeerAsync(to, m->dev[(size_t)dst], from, m->dev[(size_t)src], per_bytes, c->stream));
        }
    }
    return DNAGPU_OK;
}

// Short k-mers on N ranks (a sum-reduce of the 4^k table): nothing is gathered.  Rank r counts the
// rows that START in its own chunk into a table of 4^k counters (the k-1 <= 8 bases a row may reach into the next
// chunk are one word, copied from the neighbour), the tables are summed onto rank 0 (ncclReduce, or peer copies and
// adds), and rank 0 compacts: hists[0] holds the whole result in ascending key order, the other ranks' are empty.
static int multi_count_dense(dnagpu_multi *m, const dnagpu_multi_dna *d, int k, u64 first, u64 count, dnagpu_hist **hists)
{
    const int bits = 2 * k;
    const size_t n_bins = (size_t)1 << bits;
    std::vector<u32 *> table((size_t)m->n, nullptr);
    u32 *scratch = nullptr;
    int rc = DNAGPU_OK;
    auto cleanup = [&]() {
        for (int r = 0; r < m->n; r++)
            pool_free(m->ctx[(size_t)r], table[(size_t)r]);
        pool_free(m->ctx[0], scratch);
    };
    for (int r = 0; r < m->n && rc == DNAGPU_OK; r++) {
        dnagpu_ctx *c = m->ctx[(size_t)r];
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess)
            rc = pool_alloc_t(c, n_bins, &table[(size_t)r]);
        if (e == hipSuccess && rc == DNAGPU_OK) {
            const u64 w_lo = std::min((u64)r * d->per, d->n_words), w_hi = std::min((u64)(r + 1) * d->per, d->n_words);
            const u64 row_lo = std::max(first, w_lo * 32), row_hi = std::min(first + count, w_hi * 32);
            if (r + 1 < m->n && w_hi < d->n_words && row_hi > row_lo) {
                // the neighbour's first word (its chunk is resident since the upload; gather space on this rank)
                const int src = r + 1;
                u64 *to = d->full[(size_t)r] + w_hi;
                const u64 *from = d->full[(size_t)src] + w_hi;
                e = m->dev[(size_t)src] == m->dev[(size_t)r]
                        ? hipMemcpyAsync(to, from, 8, hipMemcpyDeviceToDevice, c->stream)
                        : hipMemcpyPeerAsync(to, m->dev[(size_t)r], from, m->dev[(size_t)src], 8, c->stream);
            }
            if (e == hipSuccess)
                e = launch_dense_table(d->full[(size_t)r], d->n_words, row_lo, row_hi > row_lo ? row_hi - row_lo : 0, bits,
                                       table[(size_t)r], c->stream);
        }
        if (e != hipSuccess) {
            set_err("dense multi count (rank %d): %s", r, hipGetErrorString(e));
            rc = DNAGPU_ERR_HIP;
        }
    }
    if (rc == DNAGPU_OK && m->n > 1) {
        if (m->rccl) {
            ncclResult_t nr = m->api.GroupStart();
            for (int r = 0; r < m->n && nr == ncclSuccess; r++)
                nr = m->api.Reduce(table[(size_t)r], table[(size_t)r], n_bins, ncclUint32, ncclSum, 0, m->comms[(size_t)r],
                                   m->ctx[(size_t)r]->stream);
            const ncclResult_t ne = m->api.GroupEnd();
            if (nr != ncclSuccess || ne != ncclSuccess) {
                set_err("ncclReduce: %s", m->api.GetErrorString(nr != ncclSuccess ? nr : ne));
                rc = DNAGPU_ERR_HIP;
            }
        } else {
            dnagpu_ctx *c0 = m->ctx[0];
            hipError_t e = hipSuccess;
            for (int r = 1; r < m->n && e == hipSuccess; r++) {       // (the partial table of rank r is complete)
                e = hipSetDevice(m->ctx[(size_t)r]->device);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(m->ctx[(size_t)r]->stream);
            }
            if (e == hipSuccess)
                e = hipSetDevice(c0->device);
            if (e == hipSuccess)
                rc = pool_alloc_t(c0, n_bins, &scratch);
            for (int r = 1; r < m->n && e == hipSuccess && rc == DNAGPU_OK; r++) {
                e = m->dev[(size_t)r] == m->dev[0]
                        ? hipMemcpyAsync(scratch, table[(size_t)r], n_bins * 4, hipMemcpyDeviceToDevice, c0->stream)
                        : hipMemcpyPeerAsync(scratch, m->dev[0], table[(size_t)r], m->dev[(size_t)r], n_bins * 4, c0->stream);
                if (e == hipSuccess)
                    e = launch_table_add(table[0], scratch, (u32)n_bins, c0->stream);
            }
            if (e != hipSuccess) {
                set_err("dense multi count (sum): %s", hipGetErrorString(e));
                rc = DNAGPU_ERR_HIP;
            }
        }
    }
    // rank 0: the table -> ascending (key, count) groups, one segment
    if (rc == DNAGPU_OK) {
        dnagpu_ctx *c0 = m->ctx[0];
        PoolScope ps(c0);
        u32 *oc = nullptr, *seg_cnt = nullptr;
        u64 *ok = nullptr, *seg_off = nullptr, *n_out = nullptr;
        hipError_t e = hipSetDevice(c0->device);
        rc = ps.alloc(n_bins, &ok);
        if (rc == DNAGPU_OK) rc = ps.alloc(n_bins, &oc);
        if (rc == DNAGPU_OK) rc = ps.alloc(1, &seg_off);
        if (rc == DNAGPU_OK) rc = ps.alloc(1, &seg_cnt);
        if (rc == DNAGPU_OK) rc = ps.alloc(1, &n_out);
        u64 D = 0;
        if (rc == DNAGPU_OK) {
            if (e == hipSuccess)
                e = launch_dense_compact(table[0], bits, ok, oc, n_out, c0->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(&D, n_out, 8, hipMemcpyDeviceToHost, c0->stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(c0->stream);
            const u64 zero = 0;
            const u32 d32 = (u32)D;
            if (e == hipSuccess)
                e = hipMemcpyAsync(seg_off, &zero, 8, hipMemcpyHostToDevice, c0->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(seg_cnt, &d32, 4, hipMemcpyHostToDevice, c0->stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(c0->stream);
            if (e != hipSuccess) {
                set_err("dense multi count (compact): %s", hipGetErrorString(e));
                rc = DNAGPU_ERR_HIP;
            }
        }
        for (int r = 0; r < m->n && rc == DNAGPU_OK; r++) {
            hists[r] = new (std::nothrow) dnagpu_hist{nullptr, nullptr, 0, r == 0 ? count : 0, nullptr, nullptr, nullptr, 0, true};
            if (!hists[r])
                rc = DNAGPU_ERR_OOM;
        }
        if (rc == DNAGPU_OK) {
            dnagpu_hist *h = hists[0];
            h->n_distinct = D;
            h->keys = ok;
            h->counts = oc;
            h->seg_off = seg_off;
            h->seg_cnt = seg_cnt;
            h->n_segs = 1;
            ps.release(ok);
            ps.release(oc);
            ps.release(seg_off);
            ps.release(seg_cnt);
        } else {
            for (int r = 0; r < m->n; r++) {
                delete hists[r];
                hists[r] = nullptr;
            }
        }
    }
    // the other ranks' streams may still hold the reduce: their tables go back to the pools behind it
    for (int r = 1; r < m->n; r++)
        if (hipSetDevice(m->ctx[(size_t)r]->device) == hipSuccess)
            (void)hipStreamSynchronize(m->ctx[(size_t)r]->stream);
    (void)hipSetDevice(m->ctx[0]->device);
    cleanup();
    return rc;
}

extern "C" int dnagpu_count_multi(dnagpu_multi *m, const dnagpu_multi_dna *dna, int k, uint64_t first, uint64_t count,
                                  dnagpu_hist **hists)
{
    return guarded([&]() -> int {
    if (!m || !dna || !hists || (int)dna->view.size() != m->n)
        return DNAGPU_ERR_BAD_ARG;
    for (int r = 0; r < m->n; r++)
        hists[r] = nullptr;
    RC_TRY(check_range(dna->view[0], k, first, count));
    if (dense_pays(count, k)) {
        m->last_exchange = m->n == 1 ? "none" : (m->rccl ? "rccl-reduce" : "peer-copy");
        return multi_count_dense(m, dna, k, first, count, hists);
    }
    m->last_exchange = m->n == 1 ? "none" : (m->rccl ? "rccl-allgather" : "peer-copy");
    RC_TRY(multi_gather(m, dna));
    // one host thread per rank: the level loop of a count reads counters back between levels, so the ranks
    // only run concurrently when each is driven by its own thread (device selection is per thread)
    std::vector<int> rcs((size_t)m->n, DNAGPU_OK);
    std::vector<std::string> errs((size_t)m->n);
    const std::function<void(int)> work = [&](int r) {
        rcs[(size_t)r] = dnagpu_count_kmers_owned(m->ctx[(size_t)r], dna->view[(size_t)r], k, first, count, r, m->n,
                                                 &hists[r]);
        if (rcs[(size_t)r] != DNAGPU_OK)
            errs[(size_t)r] = dnagpu_last_error();                // the error text is per thread
    };
    const int wrc = m->workers.run(m->n, work);
    if (wrc != DNAGPU_OK) {
        set_err("a rank's count ended in a C++ exception");
        for (int q = 0; q < m->n; q++) {
            dnagpu_hist_free(m->ctx[(size_t)q], hists[q]);
            hists[q] = nullptr;
        }
        return wrc;
    }
    for (int r = 0; r < m->n; r++)
        if (rcs[(size_t)r] != DNAGPU_OK) {
            set_err("rank %d: %s", r, errs[(size_t)r].c_str());
            for (int q = 0; q < m->n; q++) {
                dnagpu_hist_free(m->ctx[(size_t)q], hists[q]);
                hists[q] = nullptr;
            }
            return rcs[(size_t)r];
        }
    return DNAGPU_OK;
    });
}

// ---- the same count without any order promise, for long k-mers (k >= 21): the record exchange from one process.
// Rank r cuts the records of the rows that start in its own chunk (one word of halo from its neighbour), every coarse
// bucket's pieces are pulled by the bucket's owner (peer copies of 16-byte records, 1.8 B per k-mer at k = 31; nothing is
// gathered and no rank sweeps rows of another), and the owner counts them.  The exchange is PIPELINED with the count: an
// owner's buckets are cut into `parts` groups; all copies are queued at once on the owner's transfer stream, group after
// group with an event behind each, and the counting of group g (on the context's stream) waits for event g only -- the
// pieces of group g + 1 arrive while group g is counted.  hists[r] = the groups of rank r's buckets (a histogram of
// `parts` parts): disjoint between ranks, in no key order.

// rehearsal aid: holds a stream for `ticks` of the 100 MHz wall clock (the time a copy of that size would take on a link
// of the emulated bandwidth); one wave, every lane leaves the loop when the clock passes the deadline
__global__ __launch_bounds__(64) void link_delay_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(32);
}

extern "C" int dnagpu_multi_set_option(dnagpu_multi *m, int option, double value)
{
    if (!m)
        return DNAGPU_ERR_BAD_ARG;
    switch (option) {
    case DNAGPU_MULTI_OPT_PARTS:
        if (value < 1 || value > DNAGPU_MULTI_MAX_PARTS)
            return DNAGPU_ERR_BAD_ARG;
        m->parts = (int)value;
        return DNAGPU_OK;
    case DNAGPU_MULTI_OPT_EMULATE_LINK_GBS:
        if (value < 0)
            return DNAGPU_ERR_BAD_ARG;
        m->emulate_gbs = value;
        return DNAGPU_OK;
    case DNAGPU_MULTI_OPT_PROBE_OWNER:
        if (value < -1 || value >= m->n)
            return DNAGPU_ERR_BAD_ARG;
        m->probe_owner = (int)value;
        return DNAGPU_OK;
    case DNAGPU_MULTI_OPT_EXCHANGE_RCCL:
        if (value < 0 || value > 2)
            return DNAGPU_ERR_BAD_ARG;
        if (value > 0 && !m->rccl) {
            set_err("the RCCL record exchange needs the RCCL transport (dnagpu_multi_transport() is \"%s\")", m->rccl ? "rccl" : "copy");
            return DNAGPU_ERR_BAD_ARG;
        }
        m->exchange_rccl = (int)value;
        return DNAGPU_OK;
    }
    return DNAGPU_ERR_BAD_ARG;
}

namespace {
double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
struct EventSet {                               // timing events of one owner, destroyed with the scope
    std::vector<hipEvent_t> ev;
    ~EventSet()
    {
        for (hipEvent_t e : ev)
            (void)hipEventDestroy(e);
    }
    hipError_t make(hipEvent_t *out)
    {
        hipEvent_t e;
        const hipError_t r = hipEventCreate(&e);
        if (r != hipSuccess)
            return r;
        ev.push_back(e);
        *out = e;
        return hipSuccess;
    }
};
}  // namespace

extern "C" int dnagpu_count_multi_unordered(dnagpu_multi *m, const dnagpu_multi_dna *dna, int k, uint64_t first, uint64_t count,
                                            dnagpu_hist **hists)
{
    return guarded([&]() -> int {
    if (!m || !dna || !hists || (int)dna->view.size() != m->n)
        return DNAGPU_ERR_BAD_ARG;
    for (int r = 0; r < m->n; r++)
        hists[r] = nullptr;
    m->last = dnagpu_multi_times{};
    RC_TRY(check_range(dna->view[0], k, first, count));
    if (k < sk_min_k() || count == 0)
        return dnagpu_count_multi(m, dna, k, first, count, hists);         // (short k-mers: the ordered paths)
    const auto t_call = std::chrono::steady_clock::now();
    const int W = m->n;
    std::vector<dnagpu_records *> recs((size_t)W, nullptr);
    std::vector<int> rcs((size_t)W, DNAGPU_OK);
    std::vector<std::string> errs((size_t)W);
    std::vector<double> t_rec((size_t)W, 0.0), t_cnt((size_t)W, 0.0), t_xfer((size_t)W, 0.0), t_hidden((size_t)W, 0.0);
    std::vector<u64> moved((size_t)W, 0);
    auto fail = [&](int r, int rc, const char *what) {
        rcs[(size_t)r] = rc;
        errs[(size_t)r] = what;
    };
    // ---- every rank: the records of its own rows
    const std::function<void(int)> cut = [&](int r) {
        const auto t0 = std::chrono::steady_clock::now();
        dnagpu_ctx *c = m->ctx[(size_t)r];
        const u64 w_lo = std::min((u64)r * dna->per, dna->n_words), w_hi = std::min((u64)(r + 1) * dna->per, dna->n_words);
        const u64 row_lo = std::max<u64>(first, w_lo * 32), row_hi = std::min<u64>(first + count, w_hi * 32);
        hipError_t e = hipSetDevice(c->device);
        if (e == hipSuccess && r + 1 < W && w_hi < dna->n_words && row_hi > row_lo) {
            const int src = r + 1;                 // the k-1 <= 31 bases a row reaches into the next chunk: one word
            u64 *to = dna->full[(size_t)r] + w_hi;
            const u64 *from = dna->full[(size_t)src] + w_hi;
            e = m->dev[(size_t)src] == m->dev[(size_t)r] ? hipMemcpyAsync(to, from, 8, hipMemcpyDeviceToDevice, c->stream)
                                                        : hipMemcpyPeerAsync(to, m->dev[(size_t)r], from, m->dev[(size_t)src], 8, c->stream);
        }
        if (e != hipSuccess)
            return fail(r, DNAGPU_ERR_HIP, hipGetErrorString(e));
        rcs[(size_t)r] = dnagpu_sk_records(c, dna->view[(size_t)r], k, row_hi > row_lo ? row_lo : 0, row_hi > row_lo ? row_hi - row_lo : 0,
                                           count, &recs[(size_t)r]);
        if (rcs[(size_t)r] != DNAGPU_OK)
            errs[(size_t)r] = dnagpu_last_error();
        m->rec_phases[(size_t)r] = c->last_times;  // (the owner phase below starts a new profiling session on this context)
        t_rec[(size_t)r] = ms_since(t0);
    };
    m->rec_phases.assign((size_t)W, dnagpu_phase_times{});
    int rc = m->workers.run(W, cut);
    if (rc != DNAGPU_OK)
        set_err("a rank's record pass ended in a C++ exception");
    for (int r = 0; r < W && rc == DNAGPU_OK; r++)
        if (rcs[(size_t)r] != DNAGPU_OK) {
            set_err("rank %d (records): %s", r, errs[(size_t)r].c_str());
            rc = rcs[(size_t)r];
        }
    // ---- every owner: its buckets' pieces from all ranks, group by group, counted as they land
    if (rc == DNAGPU_OK) {
        const SkGeom g = sk_geometry(m->ctx[0], count, k);
        const u32 nb = dnagpu_records_buckets(recs[0]);
        const u32 n_coarse = 1u << g.r0bits;
        // owners: contiguous bucket ranges balanced by the records the buckets hold on all ranks (shard_math.py:
        // bucket_owner_ranges_weighted -- a bucket goes to the side its middle falls on)
        std::vector<u64> wgt(nb, 0);
        u64 wtotal = 0;
        for (int r = 0; r < W; r++)
            for (u32 b = 0; b < nb; b++) {
                wgt[b] += recs[(size_t)r]->off[b + 1] - recs[(size_t)r]->off[b];
                wtotal += recs[(size_t)r]->off[b + 1] - recs[(size_t)r]->off[b];
            }
        // cuts[j] for j = 0 .. W * P: owner o's group p = buckets [cuts[o * P + p], cuts[o * P + p + 1])
        const int P = std::max(1, std::min(m->parts, (int)DNAGPU_MULTI_MAX_PARTS));
        const int WP = W * P;
        std::vector<u32> cuts((size_t)WP + 1, 0);
        cuts[(size_t)WP] = nb;
        if (wtotal == 0) {
            for (int j = 1; j < WP; j++)
                cuts[(size_t)j] = (u32)(((u64)j * nb + (u64)WP - 1) / (u64)WP);
        } else {
            // owners first (the rule the process-per-GPU path uses), then every owner's range into P groups the same way
            std::vector<u32> ocut((size_t)W + 1, 0);
            ocut[(size_t)W] = nb;
            auto split = [&](u32 lo, u32 hi, int ways, u32 *out /* ways + 1 entries, out[0] = lo, out[ways] = hi */) {
                u64 tot = 0;
                for (u32 b = lo; b < hi; b++)
                    tot += wgt[b];
                out[0] = lo;
                out[ways] = hi;
                u64 run = 0;
                u32 b = lo;
                for (int j = 1; j < ways; j++) {
                    const double target = (double)tot * j / ways;
                    while (b < hi && (double)run + (double)wgt[b] / 2 <= target) {
                        run += wgt[b];
                        b++;
                    }
                    out[j] = b;
                }
            };
            split(0, nb, W, ocut.data());
            // An owner's groups grow geometrically (1 : 3 : 9 ...): the first one lands -- and its counting starts --
            // after a small share of the transfer, and every later group is still in flight while a group a third of its
            // size is being counted.
            for (int o = 0; o < W; o++) {
                const u32 lo = ocut[(size_t)o], hi = std::max(ocut[(size_t)o + 1], ocut[(size_t)o]);
                u64 tot = 0;
                for (u32 b = lo; b < hi; b++)
                    tot += wgt[b];
                double wsum = 0, acc = 0, wp = 1;
                for (int p = 0; p < P; p++, wp *= 3)
                    wsum += wp;
                u32 *out = &cuts[(size_t)o * P];
                out[0] = lo;
                u64 run = 0;
                u32 b = lo;
                wp = 1;
                for (int p = 1; p < P; p++, wp *= 3) {
                    acc += wp;
                    const double target = (double)tot * acc / wsum;
                    while (b < hi && (double)run + (double)wgt[b] / 2 <= target) {
                        run += wgt[b];
                        b++;
                    }
                    out[p] = b;
                }
                cuts[(size_t)(o + 1) * P] = hi;
            }
        }
        m->last.parts = P;
        // How the remote pieces travel.  Default: the owner PULLS every piece with a peer copy on its transfer stream.
        // DNAGPU_MULTI_OPT_EXCHANGE_RCCL: every piece is one ncclSend on its rank's transfer stream and one ncclRecv on its
        // owner's, a group call per bucket group (round p: a rank sends what the other owners' groups p hold of its records
        // and receives its own group p; between two ranks the pieces are issued in ascending bucket order on both sides).
        // Needs every rank driven by its own thread (the ranks' group calls meet each other) and all owners active.
        const bool via_rccl = m->exchange_rccl > 0 && m->rccl && !m->workers.serial && m->probe_owner < 0;
        const bool rccl_self = via_rccl && m->exchange_rccl == 2;
        m->last_exchange = via_rccl ? "rccl-sendrecv" : "peer-copy";
        const std::function<void(int)> own = [&](int o) {
            const auto t0 = std::chrono::steady_clock::now();
            dnagpu_ctx *c = m->ctx[(size_t)o];
            if (m->probe_owner >= 0 && o != m->probe_owner) {     // rehearsal probe: this owner's buckets are not counted
                hists[o] = new (std::nothrow) dnagpu_hist{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, false};
                if (!hists[o])
                    fail(o, DNAGPU_ERR_OOM, "host allocation failed");
                return;
            }
            hipStream_t xs = m->xfer[(size_t)o];
            hipError_t e = hipSetDevice(c->device);
            if (e != hipSuccess)
                return fail(o, DNAGPU_ERR_HIP, hipGetErrorString(e));
            dnagpu_hist *head = new (std::nothrow) dnagpu_hist{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, false};
            if (!head)
                return fail(o, DNAGPU_ERR_OOM, "host allocation failed");
            hists[o] = head;
            EventSet evs;
            hipEvent_t x0 = nullptr, x1 = nullptr, c0 = nullptr;
            std::vector<hipEvent_t> landed((size_t)P, nullptr);
            std::vector<void *> bufs((size_t)P, nullptr);
            std::vector<std::vector<u64>> boffs((size_t)P), blens((size_t)P);
            auto drop = [&](int rc_, const char *what) {          // error exit: nothing of this owner's buffers is in flight afterwards
                (void)hipStreamSynchronize(xs);
                (void)hipStreamSynchronize(c->stream);
                for (void *b : bufs)
                    pool_free(c, b);
                fail(o, rc_, what);
            };
            hipEvent_t ready = nullptr;
            e = evs.make(&x0);
            if (e == hipSuccess) e = evs.make(&x1);
            if (e == hipSuccess) e = evs.make(&c0);
            if (e == hipSuccess) e = evs.make(&ready);
            for (int p = 0; p < P && e == hipSuccess; p++)
                e = evs.make(&landed[(size_t)p]);
            if (e != hipSuccess)
                return drop(DNAGPU_ERR_HIP, hipGetErrorString(e));
            // ---- every group's landing buffer
            for (int p = 0; p < P; p++) {
                const u32 b_lo = cuts[(size_t)o * P + p], b_hi = std::max(cuts[(size_t)o * P + p + 1], b_lo);
                std::vector<u64> &blen = blens[(size_t)p], &boff = boffs[(size_t)p];
                blen.assign(n_coarse, 0);
                boff.assign((size_t)n_coarse + 1, 0);
                for (u32 b = b_lo; b < b_hi; b++)
                    blen[b] = wgt[b];
                for (u32 d = 0; d < n_coarse; d++)
                    boff[d + 1] = boff[d] + blen[d];
                const u64 n_recs = boff[n_coarse];
                if (n_recs > 0xFFFFFFFFull)
                    return drop(DNAGPU_ERR_TOO_LARGE, "too many records for one owner");
                if (n_recs) {
                    const int arc = pool_alloc(c, (size_t)sk_received_cap(blen, n_coarse, g) * 16, &bufs[(size_t)p]);
                    if (arc != DNAGPU_OK)
                        return drop(arc, dnagpu_last_error());
                }
            }
            // The pool orders reuse on the context's stream only (and poisons there when asked to): the transfer stream
            // starts behind everything queued on it so far -- the owner's own record pass included, whose pieces are read
            // from this device; the other ranks' passes were synchronised by dnagpu_sk_records.
            e = hipEventRecord(ready, c->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(xs, ready, 0);
            if (e == hipSuccess) e = hipEventRecord(x0, xs);
            if (e != hipSuccess)
                return drop(DNAGPU_ERR_HIP, hipGetErrorString(e));
            // ---- all copies, group after group, an event behind each group
            for (int p = 0; p < P; p++) {
                const u32 b_lo = cuts[(size_t)o * P + p], b_hi = std::max(cuts[(size_t)o * P + p + 1], b_lo);
                const std::vector<u64> &boff = boffs[(size_t)p];
                if (via_rccl) {
                    ncclResult_t nr = m->api.GroupStart();
                    // this rank's records of the other owners' groups p (its own pieces too when asked: one-rank tests)
                    const dnagpu_records *mine = recs[(size_t)o];
                    for (int q = 0; q < W && nr == ncclSuccess; q++) {
                        const int dst = (o + q) % W;
                        if (dst == o && !rccl_self)
                            continue;
                        const u32 d_lo = cuts[(size_t)dst * P + p], d_hi = std::max(cuts[(size_t)dst * P + p + 1], d_lo);
                        for (u32 b = d_lo; b < d_hi && nr == ncclSuccess; b++) {
                            const u64 n_b = mine->off[b + 1] - mine->off[b];
                            if (n_b)
                                nr = m->api.Send(static_cast<const char *>(mine->recs) + mine->off[b] * 16, (size_t)n_b * 2, ncclUint64,
                                                 dst, m->comms[(size_t)o], xs);
                        }
                    }
                    for (u32 b = b_lo; b < b_hi && nr == ncclSuccess && bufs[(size_t)p]; b++) {
                        u64 at = boff[b];
                        for (int q = 0; q < W && nr == ncclSuccess; q++) {
                            const int src = (o + q) % W;
                            const dnagpu_records *rr = recs[(size_t)src];
                            const u64 n_b = rr->off[b + 1] - rr->off[b];
                            if (!n_b)
                                continue;
                            char *to = static_cast<char *>(bufs[(size_t)p]) + at * 16;
                            if (src == o && !rccl_self) {
                                if (hipMemcpyAsync(to, static_cast<const char *>(rr->recs) + rr->off[b] * 16, (size_t)n_b * 16,
                                                   hipMemcpyDeviceToDevice, xs) != hipSuccess)
                                    nr = ncclUnhandledCudaError;
                            } else {
                                nr = m->api.Recv(to, (size_t)n_b * 2, ncclUint64, src, m->comms[(size_t)o], xs);
                                if (src != o)
                                    moved[(size_t)o] += n_b * 16;
                            }
                            at += n_b;
                        }
                    }
                    const ncclResult_t ne = m->api.GroupEnd();
                    if (nr != ncclSuccess || ne != ncclSuccess) {
                        (void)hipGetLastError();
                        return drop(DNAGPU_ERR_HIP, m->api.GetErrorString(nr != ncclSuccess ? nr : ne));
                    }
                } else if (bufs[(size_t)p]) {
                    u64 delay_bytes = 0;
                    for (u32 b = b_lo; b < b_hi; b++) {
                        u64 at = boff[b];
                        for (int q = 0; q < W; q++) {
                            const int src = (o + q) % W;             // own pieces first, then round the ranks: spreads the link load
                            const dnagpu_records *rr = recs[(size_t)src];
                            const u64 n_b = rr->off[b + 1] - rr->off[b];
                            if (!n_b)
                                continue;
                            char *to = static_cast<char *>(bufs[(size_t)p]) + at * 16;
                            const char *from = static_cast<const char *>(rr->recs) + rr->off[b] * 16;
                            if (m->dev[(size_t)src] == m->dev[(size_t)o])
                                e = hipMemcpyAsync(to, from, (size_t)n_b * 16, hipMemcpyDeviceToDevice, xs);
                            else
                                e = hipMemcpyPeerAsync(to, m->dev[(size_t)o], from, m->dev[(size_t)src], (size_t)n_b * 16, xs);
                            if (e != hipSuccess)
                                return drop(DNAGPU_ERR_HIP, hipGetErrorString(e));
                            if (src != o) {
                                moved[(size_t)o] += n_b * 16;
                                delay_bytes += n_b * 16;
                            }
                            at += n_b;
                        }
                    }
                    if (m->emulate_gbs > 0 && delay_bytes) {
                        // rehearsal on one device: the group's inbound bytes at the emulated rate, on the transfer stream
                        const double us = (double)delay_bytes / (m->emulate_gbs * 1e3);
                        const unsigned long long ticks = (unsigned long long)std::min(us, 50000.0) * 100ull;
                        hipLaunchKernelGGL(link_delay_kernel, dim3(1), dim3(64), 0, xs, ticks);
                    }
                }
                e = hipEventRecord(landed[(size_t)p], xs);
                if (e != hipSuccess)
                    return drop(DNAGPU_ERR_HIP, hipGetErrorString(e));
            }
            e = hipEventRecord(x1, xs);
            if (e == hipSuccess) e = hipEventRecord(c0, c->stream);
            if (e != hipSuccess)
                return drop(DNAGPU_ERR_HIP, hipGetErrorString(e));
            // ---- count group p behind its event
            prof_begin(c);
            for (int p = 0; p < P; p++) {
                if (!bufs[(size_t)p])
                    continue;
                e = hipStreamWaitEvent(c->stream, landed[(size_t)p], 0);
                if (e != hipSuccess)
                    return drop(DNAGPU_ERR_HIP, hipGetErrorString(e));
                dnagpu_hist *part = new (std::nothrow) dnagpu_hist{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, false};
                if (!part)
                    return drop(DNAGPU_ERR_OOM, "host allocation failed");
                void *buf = bufs[(size_t)p];
                bufs[(size_t)p] = nullptr;                        // (count_sk_received takes the buffer over)
                const int crc = count_sk_received(c, buf, boffs[(size_t)p], blens[(size_t)p], g, k, part,
                                                  sk_received_cap(blens[(size_t)p], n_coarse, g));
                if (crc != DNAGPU_OK) {
                    delete part;
                    return drop(crc, dnagpu_last_error());
                }
                head->parts.push_back(part);
                head->n_distinct += part->n_distinct;
                head->total += part->total;
                head->extent += part->extent ? part->extent : part->n_distinct;
            }
            prof_end(c);
            e = hipStreamSynchronize(xs);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess)
                return drop(DNAGPU_ERR_HIP, hipGetErrorString(e));
            float x_ms = 0, c_at = 0;
            (void)hipEventElapsedTime(&x_ms, x0, x1);             // first copy queued -> last piece landed
            (void)hipEventElapsedTime(&c_at, x0, c0);             // ... -> the owner's stream was free to count
            t_xfer[(size_t)o] = x_ms;
            // the counting starts when the first group has landed; what the transfer stream did after that ran beside it
            float first_ms = 0;
            (void)hipEventElapsedTime(&first_ms, x0, landed[0]);
            t_hidden[(size_t)o] = std::max(0.0f, x_ms - std::max(first_ms, c_at));
            if (head->parts.size() == 1) {                        // one group: a plain histogram, no head
                dnagpu_hist *only = head->parts[0];
                head->parts.clear();
                delete head;
                hists[o] = only;
            }
            t_cnt[(size_t)o] = ms_since(t0);
        };
        const auto t_own = std::chrono::steady_clock::now();
        rc = m->workers.run(W, own);
        if (rc != DNAGPU_OK)
            set_err("an owner's count ended in a C++ exception");
        for (int r = 0; r < W && rc == DNAGPU_OK; r++)
            if (rcs[(size_t)r] != DNAGPU_OK) {
                set_err("rank %d (count): %s", r, errs[(size_t)r].c_str());
                rc = rcs[(size_t)r];
            }
        m->last.records_ms = *std::max_element(t_rec.begin(), t_rec.end());
        m->last.exchange_ms = *std::max_element(t_xfer.begin(), t_xfer.end());
        m->last.hidden_ms = m->probe_owner >= 0 ? t_hidden[(size_t)m->probe_owner] : *std::min_element(t_hidden.begin(), t_hidden.end());
        m->last.count_ms = ms_since(t_own);
        for (int r = 0; r < W; r++)
            m->last.bytes_moved += moved[(size_t)r];
    }
    for (int r = 0; r < W; r++) {
        (void)hipSetDevice(m->ctx[(size_t)r]->device);
        dnagpu_records_free(m->ctx[(size_t)r], recs[(size_t)r]);
        if (rc != DNAGPU_OK) {
            dnagpu_hist_free(m->ctx[(size_t)r], hists[r]);
            hists[r] = nullptr;
        }
    }
    (void)hipSetDevice(m->ctx[0]->device);
    m->last.total_ms = ms_since(t_call);
    return rc;
    });
}

