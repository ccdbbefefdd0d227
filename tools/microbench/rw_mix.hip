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
// reads n keys sequentially; every run of run_keys consecutive keys is stored, aligned, at a random run slot
__global__ __launch_bounds__(256) void k_scat(const uint64_t *src, uint64_t *dst, uint64_t n, int run_keys, int read_too)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t n_runs = n / run_keys;
    for (uint64_t i = gid; i < n; i += stride) {
        uint64_t v = read_too ? __builtin_nontemporal_load(&src[i]) : i;
        uint64_t run = i / run_keys, within = i % run_keys;
        uint64_t where = mix(run) % n_runs;
        __builtin_nontemporal_store(v, &dst[where * run_keys + within]);
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
    run("copy 8B/lane (R+W)", n * 16.0, [&] { hipLaunchKernelGGL(k_copy8, g, t, 0, 0, a, b, n); });
    run("copy 16B/lane (R+W)", n * 16.0, [&] { hipLaunchKernelGGL(k_copy16, g, t, 0, 0, (const ull2_t *)a, (ull2_t *)b, n / 2); });
    for (int r : {8, 16, 32, 128}) {
        char nm[64];
        snprintf(nm, sizeof nm, "write scattered runs of %d keys", r);
        run(nm, n * 8.0, [&] { hipLaunchKernelGGL(k_scat, g, t, 0, 0, a, b, n, r, 0); });
        snprintf(nm, sizeof nm, "read seq + write runs of %d (R+W)", r);
        run(nm, n * 16.0, [&] { hipLaunchKernelGGL(k_scat, g, t, 0, 0, a, b, n, r, 1); });
    }
    return 0;
}
