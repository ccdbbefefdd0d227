// Microbenchmark: returning atomic adds from every CU on a FEW global addresses (what per-digit output cursors of a
// one-pass scatter would be): D cursors (one per 128-byte line), every workgroup does ITER rounds of one wave-wide
// instruction in which lane d adds to cursor d.  Prints atomics per microsecond per address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_atom(unsigned long long *cur, int D, int iters, unsigned long long *sink, int dep)
{
    const int lane = threadIdx.x & 63;
    unsigned long long acc = 0;
    if ((threadIdx.x >> 6) == 0 && lane < D) {
        for (int i = 0; i < iters; i++) {
            const unsigned long long v = atomicAdd(&cur[(size_t)lane * 16], 14ull + (dep ? (acc & 1) : 0));
            acc += v;
        }
    }
    if (acc == 0x1234567)
        sink[0] = acc;
}

int main()
{
    unsigned long long *cur, *sink;
    hipMalloc(&cur, 128 * 64 * 8);
    hipMalloc(&sink, 64);
    hipMemset(cur, 0, 128 * 64 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int dep = 0; dep < 2; dep++)
        for (int D : {1, 16, 64}) {
            for (int grid : {1024, 4096}) {
                const int iters = 256;
                hipLaunchKernelGGL(k_atom, dim3(grid), dim3(256), 0, 0, cur, D, 8, sink, dep);
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k_atom, dim3(grid), dim3(256), 0, 0, cur, D, iters, sink, dep);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double per_addr = (double)grid * iters;
                printf("dependent=%d D=%2d grid=%4d: %8.3f ms, %7.1f atomics/us/address (%.1f ns each), %.1f M atomics/ms total\n", dep, D,
                       grid, ms, per_addr / (ms * 1e3), ms * 1e6 / per_addr, per_addr * D / ms / 1e6);
            }
        }
    return 0;
}
