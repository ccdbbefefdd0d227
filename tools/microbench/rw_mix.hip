// Microbenchmark: HBM throughput of the access mixes the count's passes produce.
//   read    sequential 8 B / lane loads only
//   write   sequential 8 B / lane stores only
//   copy    sequential loads + sequential stores (8 B / lane and 16 B / lane)
//   scat    sequential loads + stores in aligned runs of RUN keys at pseudo-random places (the
//           scatter's write-out: RUN = 8 or 16 keys = 64 or 128 B)
// Prints GB/s of (bytes read + bytes written).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_read(const uint64_t *src, uint64_t n, uint64_t *sink)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (uint64_t i = gid; i < n; i += stride)
        acc ^= __builtin_nontemporal_load(&src[i]);
    if (acc == 0x1234567)
        sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(uint64_t *dst, uint64_t n)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = gid; i < n; i += stride)
        __builtin_nontemporal_store(i, &dst[i]);
}
__global__ __launch_bounds__(1024) void k_write_wg1024(uint64_t *dst, uint64_t n)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = gid; i < n; i += stride)
        __builtin_nontemporal_store(i, &dst[i]);
}
__global__ __launch_bounds__(256) void k_copy8(const uint64_t *src, uint64_t *dst, uint64_t n)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = gid; i < n; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
typedef unsigned long long ull2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_copy16(const ull2_t *src, ull2_t *dst, uint64_t n2)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = gid; i < n2; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
// reads n keys sequentially; every run of 2^run_log consecutive keys is stored, aligned, at a random run slot
// (shifts and masks only: a 64-bit division per key would make this ALU-bound)
__global__ __launch_bounds__(256) void k_scat(const uint64_t *src, uint64_t *dst, uint64_t n, int run_log, int read_too)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t run_mask = ((uint64_t)1 << run_log) - 1;
    const uint64_t n_runs_mask = (((uint64_t)1 << 31) >> run_log) - 1;       // power-of-two run slots inside 2^31 keys
    for (uint64_t i = gid; i < n; i += stride) {
        uint64_t v = read_too ? __builtin_nontemporal_load(&src[i]) : i;
        uint64_t run = i >> run_log, within = i & run_mask;
        uint64_t where = mix(run) & n_runs_mask;
        __builtin_nontemporal_store(v, &dst[(where << run_log) + within]);
    }
}

int main()
{
    const uint64_t n = (uint64_t)3 << 30;       // 3 Gi keys = 24 GiB per buffer
    uint64_t *a, *b, *sink;
    if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipMemset(a, 1, n * 8);
    hipMemset(b, 0, n * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, double bytes, auto launch) {
        float best = 1e9;
        for (int it = 0; it < 3; it++) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-34s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
        fflush(stdout);
    };
    dim3 g(256 * 16), t(256);
    run("read 8B/lane", n * 8.0, [&] { hipLaunchKernelGGL(k_read, g, t, 0, 0, a, n, sink); });
    run("write 8B/lane", n * 8.0, [&] { hipLaunchKernelGGL(k_write, g, t, 0, 0, b, n); });
    // per-CU or chip limit?  the same stores from 1024-thread workgroups on a fraction of the CUs
    for (int blocks : {32, 64, 128, 256, 512}) {
        char nm[64];
        snprintf(nm, sizeof nm, "write 8B/lane, %d x 1024 threads", blocks);
        run(nm, n * 8.0, [&] { hipLaunchKernelGGL(k_write_wg1024, dim3(blocks), dim3(1024), 0, 0, b, n); });
    }
    run("copy 8B/lane (R+W)", n * 16.0, [&] { hipLaunchKernelGGL(k_copy8, g, t, 0, 0, a, b, n); });
    run("copy 16B/lane (R+W)", n * 16.0, [&] { hipLaunchKernelGGL(k_copy16, g, t, 0, 0, (const ull2_t *)a, (ull2_t *)b, n / 2); });
    for (int rl : {2, 3, 4, 5, 7}) {
        const int r = 1 << rl;
        char nm[64];
        snprintf(nm, sizeof nm, "write scattered runs of %d keys", r);
        run(nm, n * 8.0, [&] { hipLaunchKernelGGL(k_scat, g, t, 0, 0, a, b, n, rl, 0); });
        snprintf(nm, sizeof nm, "read seq + write runs of %d (R+W)", r);
        run(nm, n * 16.0, [&] { hipLaunchKernelGGL(k_scat, g, t, 0, 0, a, b, n, rl, 1); });
    }
    return 0;
}
