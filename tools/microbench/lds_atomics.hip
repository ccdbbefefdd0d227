// Microbenchmark: LDS throughput of the operations a workgroup-local hash table / counting sort is made of, at random
// addresses (lanes per clock per CU).  1024-thread workgroups, two per CU (the sk_count shape); every thread does
// ITER dependent-free operations on a table of SLOTS 8-byte slots.
//   cas64   ds_cmpst_rtn_b64 (returning compare-and-swap)       cas32   ds_cmpst_rtn_b32
//   add32r  ds_add_rtn_u32 (returning)                           add32   ds_add_u32 (non-returning)
//   rd64 / wr64 / rd32 / wr32   plain loads / stores
// Prints lane-operations per clock per CU (clock: 2.4 GHz nominal) from the kernel time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

constexpr int NT = 1024, SLOTS = 6784, ITER = 64;

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    return x ^ (x >> 16);
}

template <int MODE>
__global__ __launch_bounds__(NT, 8) void k_lds(uint64_t *sink, int rounds)
{
    __shared__ uint64_t tab[SLOTS];
    uint32_t *tab32 = reinterpret_cast<uint32_t *>(tab);
    for (int i = threadIdx.x; i < SLOTS; i += NT)
        tab[i] = ~0ull;
    __syncthreads();
    uint64_t acc = 0;
    uint32_t h = mix32(blockIdx.x * NT + threadIdx.x);
    for (int r = 0; r < rounds; r++) {
#pragma unroll 4
        for (int i = 0; i < ITER; i++) {
            h = h * 1664525u + 1013904223u;
            const uint32_t slot = ((h >> 16) * (uint32_t)SLOTS) >> 16;
            if (MODE == 0)
                acc += atomicCAS(reinterpret_cast<unsigned long long *>(&tab[slot]), ~0ull, (unsigned long long)h);
            else if (MODE == 1)
                acc += atomicCAS(&tab32[slot * 2], ~0u, h);
            else if (MODE == 2)
                acc += atomicAdd(&tab32[slot * 2], 1u);
            else if (MODE == 3)
                atomicAdd(&tab32[slot * 2], 1u);
            else if (MODE == 4)
                acc += tab[slot];
            else if (MODE == 5)
                tab[slot] = h;
            else if (MODE == 6)
                acc += tab32[slot * 2];
            else if (MODE == 7)
                tab32[slot * 2] = h;
            else if (MODE == 8)
                acc += tab32[slot];                // 4-byte slots: all 2 x SLOTS dwords
            else if (MODE == 9)
                acc += atomicAdd(&tab32[slot], 1u);
        }
        __syncthreads();
    }
    if (acc == 0x1234567u)
        sink[0] = acc + tab[threadIdx.x];
}

template <int MODE>
static void run(const char *name, uint64_t *sink, int n_cu)
{
    const int rounds = 64, grid = n_cu * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_lds<MODE>, dim3(grid), dim3(NT), 0, 0, sink, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_lds<MODE>, dim3(grid), dim3(NT), 0, 0, sink, rounds);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops_per_cu = 2.0 * NT * (double)ITER * rounds;
    printf("%-7s %8.3f ms  %6.2f lanes/clk/CU (2.4 GHz)  %6.1f clk per wave-op\n", name, ms, ops_per_cu / (ms * 1e-3 * 2.4e9),
           64.0 / (ops_per_cu / (ms * 1e-3 * 2.4e9)));
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    uint64_t *sink;
    hipMalloc(&sink, 64);
    printf("%s, %d CUs, clock %d kHz\n", p.name, n_cu, p.clockRate);
    run<0>("cas64", sink, n_cu);
    run<1>("cas32", sink, n_cu);
    run<2>("add32r", sink, n_cu);
    run<3>("add32", sink, n_cu);
    run<4>("rd64", sink, n_cu);
    run<5>("wr64", sink, n_cu);
    run<6>("rd32", sink, n_cu);
    run<7>("wr32", sink, n_cu);
    run<8>("rd32d", sink, n_cu);
    run<9>("add32rd", sink, n_cu);
    return 0;
}
