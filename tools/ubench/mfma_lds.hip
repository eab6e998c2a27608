// What does an LDS operand read cost next to v_mfma_f64_16x16x4_f64 on gfx950?  One wavefront per SIMD (256-thread workgroups, one per CU), a loop of
// steps: R ds_read_b64 of the NEXT step's operands, then M independent MFMAs on the current ones (the software pipeline of k_ba_syrk_mfma /
// k_ba_chol_left).  Prints ns and cycles per MFMA for several (R, M); USE = 1: the MFMAs consume the values read, 0: they use loop-invariant registers
// and the reads only land in registers.   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 mfma_lds.hip -o mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int R, int M, int USE>
__global__ __launch_bounds__(256) void k(double* out, int steps)
{
    __shared__ double lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 256) lds[i] = 1.0 + (i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    d4 acc[M];
#pragma unroll
    for (int m = 0; m < M; m++) acc[m] = d4{0, 0, 0, 0};
    double o[R > 0 ? R : 1], n[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; r++) o[r] = lds[r * 64 + lane];
    const double ca = 1.0 + lane, cb = 2.0 - lane;
    double sink = 0.0;
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int r = 0; r < R; r++) n[r] = lds[((s & 7) * 8 + r) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < M; m++) {
            const double a = USE && R > 0 ? o[m % (R > 0 ? R : 1)] : ca, b = USE && R > 1 ? o[(m + 1) % (R > 0 ? R : 1)] : cb;
            acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (!USE) sink += o[r] * 0.0;
            o[r] = n[r];
        }
    }
    double t = sink;
#pragma unroll
    for (int m = 0; m < M; m++) t += acc[m].x + acc[m].w;
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int R, int M, int USE>
static void run(double* d, int steps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<R, M, USE>), dim3(256), dim3(256), 0, 0, d, steps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, M, USE>), dim3(256), dim3(256), 0, 0, d, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = 1e6 * ms / ((double)steps * M);
    printf("  {\"reads_per_step\": %d, \"mfma_per_step\": %d, \"consumed\": %d, \"ns_per_mfma\": %.2f, \"cycles_per_mfma_at_2p4\": %.1f},\n", R, M, USE, ns, ns * 2.4);
}

int main()
{
    double* d;
    hipMalloc(&d, 256 * 256 * sizeof(double));
    const int steps = 20000;
    printf("[\n");
    run<0, 4, 0>(d, steps); run<2, 4, 0>(d, steps); run<6, 4, 0>(d, steps); run<6, 4, 1>(d, steps); run<8, 16, 0>(d, steps); run<8, 16, 1>(d, steps);
    run<3, 2, 1>(d, steps); run<12, 4, 1>(d, steps); run<16, 16, 1>(d, steps);
    printf("  {}\n]\n");
    return 0;
}
