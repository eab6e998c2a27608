// micro-benchmark: what rocprofv3's FETCH_SIZE / WRITE_SIZE report for KNOWN byte counts, by access width.  MI355X_MICROARCH.md (HBM section): on gfx950
// FETCH_SIZE shows half the bytes of a wide (16 B / lane) coalesced streaming read; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a
// known byte count in your own access pattern".  The image kernels of this library read bytes as dwords / dword pairs (k_roi_warp: two dwords per source
// row and lane, k_pyr_down: dword runs) and write packed dwords, so this program streams a 1 GiB buffer (4 x the 256 MiB Infinity Cache) once per kernel
// with 1 / 4 / 8 / 16 byte loads per lane, and writes 1 GiB with 4 / 16 byte stores.  Run it under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/ubench/fetch_calib      (and again with --pmc WRITE_SIZE)
// and divide the counter (KiB) of each kernel by 2^20 KiB: tools/summarize_profiles.py files the factors as profiles/rNN_fetch_calibration.json.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ src, size_t n, unsigned* sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = src[i];
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
#pragma unroll
        for (int k = 0; k < (int)sizeof(T); k += (sizeof(T) >= 4 ? 4 : 1)) acc += b[k];
    }
    if (acc == 0xdeadbeefu) *sink = acc;  // (never true for the fill pattern: keeps the loads alive without a store stream)
}
template <typename T>
__global__ __launch_bounds__(256) void k_write(T* __restrict__ dst, size_t n, T v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
// the access shape of k_roi_warp's gather: every lane reads TWO adjacent dwords of a row (8 source bytes for 8 output pixels), lanes of a wavefront cover
// 4 rows of 128 pixels, rows `stride` bytes apart
__global__ __launch_bounds__(256) void k_read_rows(const unsigned* __restrict__ src, int w4, int h, size_t stride4, unsigned* sink)
{
    const int x = (blockIdx.x * 16 + (threadIdx.x & 15)) * 2, y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x + 1 >= w4 || y >= h) return;
    const unsigned* r = src + (size_t)y * stride4 + x;
    const unsigned a = r[0] + r[1];
    if (a == 0xdeadbeefu) *sink = a;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void* buf;
    unsigned* sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 4);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 16;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_read<unsigned char>, dim3(grid), dim3(256), 0, 0, (const unsigned char*)buf, bytes, sink);
        hipLaunchKernelGGL(k_read<unsigned>, dim3(grid), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(k_read<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, sink);
        hipLaunchKernelGGL(k_read<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
        // 1 GiB as 16384 rows of 65536 bytes, every dword pair read once
        hipLaunchKernelGGL(k_read_rows, dim3(65536 / 4 / 32, 16384 / 16), dim3(256), 0, 0, (const unsigned*)buf, 65536 / 4, 16384, (size_t)65536 / 4, sink);
        hipLaunchKernelGGL(k_write<unsigned>, dim3(grid), dim3(256), 0, 0, (unsigned*)buf, bytes / 4, 0x01010101u);
        hipLaunchKernelGGL(k_write<uint4>, dim3(grid), dim3(256), 0, 0, (uint4*)buf, bytes / 16, make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u));
    }
    hipDeviceSynchronize();
    printf("{\"bytes_per_kernel\": %zu}\n", bytes);
    hipFree(buf);
    hipFree(sink);
    return 0;
}
