// How many 256-thread workgroups (18 KB of LDS each) meet at a grid barrier under hipLaunchCooperativeKernel?
// hipcc --offload-arch=gfx950 -O3 coop_barrier.hip -o coop_barrier && ./coop_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned *sync, unsigned *out, unsigned limit)
{
    __shared__ unsigned pad[4608];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > limit) {
                __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        out[blockIdx.x] = spins + pad[5];
    }
}
int main()
{
    int dev = 0, cus = 0, coop = 0, nb = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k), 256, 0);
    printf("CUs %d coop %d occupancy %d blocks/CU\n", cus, coop, nb);
    unsigned *sync, *out;
    hipMalloc(&sync, 16);
    hipMalloc(&out, 1 << 20);
    for (int per = 1; per <= 9; per++)
        for (int mode = 0; mode < 2; mode++) {
            unsigned grid = (unsigned)(per * cus), limit = 1u << 18;
            hipMemset(sync, 0, 16);
            hipError_t e;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a, 0);
            if (mode == 0) {
                void *args[] = {&sync, &out, &limit};
                e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k), dim3(grid), dim3(256), args, 0, 0);
            } else {
                hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, sync, out, limit);
                e = hipGetLastError();
            }
            hipEventRecord(b, 0);
            hipError_t se = hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            unsigned h[4];
            hipMemcpy(h, sync, 16, hipMemcpyDeviceToHost);
            printf("%s grid %5u (%d per CU): launch %s sync %s arrived %u timed_out %u  %.3f ms\n", mode == 0 ? "coop " : "plain", grid, per,
                   hipGetErrorName(e), hipGetErrorName(se), h[0], h[1], ms);
            (void)hipGetLastError();
        }
    return 0;
}
