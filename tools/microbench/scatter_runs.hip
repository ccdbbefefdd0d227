// Microbenchmark: HBM write bandwidth when every wave-store instruction (64 lanes x 8 B) is split into
// runs of RUN bytes landing at pseudo-random, run-aligned or unaligned places of a large buffer.
// Informs the tile/radix choice of level_scatter (keys per digit per tile = run length).
// NOTE: this kernel divides by run_keys per key and uses temporal stores, so its plateau (3.0 TB/s)
// is ALU/temporal-store bound, not the memory system's: rw_mix.hip (shifts, nontemporal stores)
// supersedes it -- sequential writes reach 6.0-6.5 TB/s, 128-byte runs 4.3 TB/s.  The ratio between
// aligned and unaligned runs measured here stands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// total_keys keys of 8 B are written; each group of run_keys consecutive lanes forms one run
__global__ __launch_bounds__(256) void scatter_runs(uint64_t *dst, uint64_t n_slots, uint64_t total_keys,
                                                    int run_keys, int misalign)
{
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t n_runs = n_slots / run_keys - 2;
    for (uint64_t i = gid; i < total_keys; i += stride) {
        uint64_t run = i / run_keys;            // global run id
        uint64_t within = i % run_keys;
        uint64_t where = mix(run) % n_runs;     // random destination run slot
        uint64_t off = misalign ? (mix(run ^ 0x55) % run_keys) : 0;   // 8-byte granular misalignment
        dst[where * run_keys + off + within] = i;
    }
}

int main(int argc, char **argv)
{
    uint64_t n_slots = (uint64_t)1 << 31;       // 16 GiB destination
    uint64_t total = (uint64_t)1 << 30;         // 8 GiB written per launch
    uint64_t *dst;
    if (hipMalloc(&dst, n_slots * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(dst, 0, n_slots * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int runs[] = {1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 4096};
    for (int mis = 0; mis < 2; mis++)
        for (int r : runs) {
            float best = 1e9;
            for (int it = 0; it < 3; it++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(scatter_runs, dim3(256 * 16), dim3(256), 0, 0, dst, n_slots, total, r, mis);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("run %5d keys (%6d B) %s: %8.3f ms  %7.1f GB/s\n", r, r * 8, mis ? "unaligned" : "aligned  ",
                   best, total * 8 / best / 1e6);
            fflush(stdout);
        }
    return 0;
}
