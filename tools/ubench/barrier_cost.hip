// micro-benchmark: what one "pivot step" skeleton costs inside a single workgroup (barriers, dependent LDS reads, f64 divide)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, int iters)
{
    __shared__ double s[256];
    __shared__ unsigned long long key[2];
    const int tid = threadIdx.x;
    if (tid < 256) s[tid] = 1.0 + tid;
    if (tid < 2) key[tid] = 5;
    __syncthreads();
    double acc = 0.0;
    for (int c = 0; c < iters; c++) {
        const int p = (int)(key[c & 1] & 0xff);
        if (MODE >= 1) {
            if (tid == 0) key[(c + 1) & 1] = (unsigned long long)((p * 7 + 3) & 0xff);
            if (tid == p) s[p] = 1.0 + acc * 1e-9 + c;
        }
        __syncthreads();
        double inv = s[p];
        if (MODE >= 2) inv = 1.0 / inv;
        if (MODE >= 3) {
#pragma unroll
            for (int q = 0; q < 16; q++) acc += inv * s[(tid + q) & 255];
        } else acc += inv;
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + tid] = acc;
}
template <int MODE>
static void run(int threads, int iters)
{
    double* d;
    hipMalloc(&d, sizeof(double) * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("mode %d threads %4d: %.3f us per step\n", MODE, threads, 1e3 * ms / iters);
    }
    hipFree(d);
}
int main()
{
    for (int t : {64, 256, 1024}) { run<0>(t, 2000); run<1>(t, 2000); run<2>(t, 2000); run<3>(t, 2000); }
    return 0;
}
