// micro-benchmark: cost of a device-wide barrier between two dependent phases against a kernel boundary (dependent launches on one stream), for grids that
// are co-resident (cooperative launch).  Usage: hipcc --offload-arch=gfx950 -O3 -o grid_sync grid_sync.hip && ./grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_cg(unsigned* buf, int iters, unsigned* bad)
{
    cg::grid_group g = cg::this_grid();
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    unsigned wrong = 0;
    for (int c = 0; c < iters; c++) {
        buf[(c & 1) * n + gid] = gid + c;
        g.sync();
        const unsigned o = (gid + blockDim.x * 7 + 13) % n;  // a value written by another workgroup
        wrong += __hip_atomic_load(&buf[(c & 1) * n + o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != o + c;
    }
    if (wrong) atomicAdd(bad, wrong);
}

// hand-made barrier: one arrival counter + generation word (agent scope), thread 0 of every workgroup spins
__device__ __forceinline__ void my_sync(unsigned* bar, unsigned nblocks)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(bar + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}
__global__ void k_my(unsigned* buf, int iters, unsigned* bad, unsigned* bar)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    unsigned wrong = 0;
    for (int c = 0; c < iters; c++) {
        buf[(c & 1) * n + gid] = gid + c;
        my_sync(bar, gridDim.x);
        const unsigned o = (gid + blockDim.x * 7 + 13) % n;
        wrong += __hip_atomic_load(&buf[(c & 1) * n + o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != o + c;
    }
    if (wrong) atomicAdd(bad, wrong);
}
__global__ void k_phase(unsigned* buf, int c)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    buf[(c & 1) * n + gid] = buf[((c + 1) & 1) * n + (gid + 2999) % n] + 1;
}

int main()
{
    unsigned *buf, *bad, *bar;
    hipMalloc(&buf, sizeof(unsigned) * 2 * 1024 * 256);
    hipMalloc(&bad, 4); hipMalloc(&bar, 8);
    hipMemset(bad, 0, 4); hipMemset(bar, 0, 8);
    hipMemset(buf, 0, sizeof(unsigned) * 2 * 1024 * 256);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int iters = 2000;
    for (int blocks : {8, 64, 128, 256, 512}) {
        float ms_cg = 0, ms_my = 0, ms_k = 0;
        for (int rep = 0; rep < 3; rep++) {
            void* args[] = {&buf, &iters, &bad};
            hipEventRecord(a);
            hipError_t e = hipLaunchCooperativeKernel((const void*)k_cg, dim3(blocks), dim3(256), args, 0, 0);
            hipEventRecord(b); hipEventSynchronize(b);
            if (e != hipSuccess) { printf("cooperative launch of %d blocks: %s\n", blocks, hipGetErrorString(e)); break; }
            hipEventElapsedTime(&ms_cg, a, b);
            void* args2[] = {&buf, &iters, &bad, &bar};
            hipEventRecord(a);
            e = hipLaunchCooperativeKernel((const void*)k_my, dim3(blocks), dim3(256), args2, 0, 0);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms_my, a, b);
            hipEventRecord(a);
            for (int c = 0; c < iters; c++) hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(256), 0, 0, buf, c);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms_k, a, b);
        }
        unsigned h = 0;
        hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("blocks %4d x 256: grid.sync %.2f us, hand-made barrier %.2f us, kernel boundary %.2f us per phase (wrong reads: %u)\n", blocks,
               1e3 * ms_cg / iters, 1e3 * ms_my / iters, 1e3 * ms_k / iters, h);
    }
    return 0;
}
