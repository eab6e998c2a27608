// micro-benchmark: does data written (or read) by one kernel stay in the XCD's L2 for the next kernel on the same stream?
// Workgroup b runs on XCD b % 8 (round-robin dispatch).  A 16 MB buffer is cut in 8 slices of 2 MB; kernel W writes slice (b % 8), kernel R reads slice
// ((b + shift) % 8): shift 0 = the XCD that wrote (or last read) it, shift 1 = a neighbour.  Usage: hipcc --offload-arch=gfx950 -O3 -o l2_persist l2_persist.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define SLICE_DW (512 * 1024)   // 2 MB of dwords per XCD slice
__global__ void k_write(unsigned* buf, unsigned v)
{
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nb = gridDim.x >> 3;
    unsigned* s = buf + (size_t)xcd * SLICE_DW;
    for (unsigned i = j * blockDim.x + threadIdx.x; i < SLICE_DW; i += nb * blockDim.x) s[i] = v + i;
}
__global__ void k_read(const unsigned* buf, unsigned shift, unsigned* out)
{
    const unsigned xcd = (blockIdx.x + shift) & 7, j = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const unsigned* s = buf + (size_t)xcd * SLICE_DW;
    unsigned acc = 0;
    for (unsigned i = j * blockDim.x + threadIdx.x; i < SLICE_DW; i += nb * blockDim.x) acc += s[i];
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    unsigned *buf, *out;
    hipMalloc(&buf, sizeof(unsigned) * 8 * SLICE_DW);
    hipMalloc(&out, 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 2048, threads = 256;
    for (int mode = 0; mode < 4; mode++) {
        // mode 0: write, read same XCD; 1: write, read other XCD; 2: read (warm), read same XCD again; 3: read (warm), read other XCD
        float best = 1e9f;
        for (int rep = 0; rep < 20; rep++) {
            if (mode < 2) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(threads), 0, 0, buf, (unsigned)rep);
            else hipLaunchKernelGGL(k_read, dim3(blocks), dim3(threads), 0, 0, buf, 0u, out);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_read, dim3(blocks), dim3(threads), 0, 0, buf, (unsigned)(mode & 1), out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep > 2 && ms < best) best = ms;
        }
        const char* names[] = {"written by the previous kernel, read on the SAME XCD", "written by the previous kernel, read on ANOTHER XCD",
                               "read by the previous kernel, read again on the SAME XCD", "read by the previous kernel, read on ANOTHER XCD"};
        printf("%-58s: %.1f us for 16 MB = %.0f GB/s\n", names[mode], 1e3 * best, 16.777216 / best);
    }
    return 0;
}
