// Can a wavefront's VALU stream overlap another wavefront's v_mfma_f64_16x16x4_f64 stream on the same SIMD of gfx950?
// One workgroup of 8 wavefronts per CU (2 per SIMD): waves 0-3 issue MFMAs, waves 4-7 issue a VALU stream of one opcode class.
// Prints the time of each stream alone and of both together.   hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // VALU stream: 0 = v_add_u32, 1 = v_fma_f64, 2 = v_perm_b32 (half-rate integer), 3 = ds_read_b64 + v_add
__global__ __launch_bounds__(512) void k(double* out, int n_mfma, int n_valu, int run_mfma, int run_valu)
{
    __shared__ double lds[1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = (double)threadIdx.x;
    lds[threadIdx.x + 512] = 1.0;
    __syncthreads();
    double r = 0.0;
    if (wave < 4) {
        if (run_mfma) {
            d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
            const double a = 1.0 + lane, b = 2.0 - lane;
            for (int i = 0; i < n_mfma; i++) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
            }
            r = acc0.x + acc1.y + acc2.z + acc3.w;
        }
    } else if (run_valu) {
        if (MODE == 0) {
            unsigned x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
            for (int i = 0; i < n_valu; i++) {
#pragma unroll
                for (int u = 0; u < 8; u++) { x0 += 0x9e37u; x1 += x0 * 0 + 0x79b9u; x2 += 0x7f4au; x3 += 0x7c15u; }
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            }
            r = (double)(x0 ^ x1 ^ x2 ^ x3);
        } else if (MODE == 1) {
            double x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
            const double m = 1.0000001, c = 0.5;
            for (int i = 0; i < n_valu; i++) {
#pragma unroll
                for (int u = 0; u < 8; u++) { x0 = __builtin_fma(x0, m, c); x1 = __builtin_fma(x1, m, c); x2 = __builtin_fma(x2, m, c); x3 = __builtin_fma(x3, m, c); }
            }
            r = x0 + x1 + x2 + x3;
        } else if (MODE == 2) {
            unsigned x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
            for (int i = 0; i < n_valu; i++) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    x0 = __builtin_amdgcn_perm(x0, x1, 0x06050403u); x1 = __builtin_amdgcn_perm(x1, x2, 0x06050403u);
                    x2 = __builtin_amdgcn_perm(x2, x3, 0x06050403u); x3 = __builtin_amdgcn_perm(x3, x0, 0x06050403u);
                }
            }
            r = (double)(x0 ^ x1 ^ x2 ^ x3);
        } else {
            double x0 = 0;
            int idx = lane;
            for (int i = 0; i < n_valu; i++) {
#pragma unroll
                for (int u = 0; u < 8; u++) { x0 += lds[(idx + 64 * u) & 1023]; }
                idx += 7;
            }
            r = x0;
        }
    }
    if (r == 12345.678) out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE>
static float run(double* d, int nm, int nv, int rm, int rv)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, nm, nv, rm, rv);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, nm, nv, rm, rv);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main()
{
    double* d;
    hipMalloc(&d, 256 * 512 * 8);
    const int nm = 20000;  // x4 MFMAs
    const char* names[4] = {"v_add_u32", "v_fma_f64", "v_perm_b32", "ds_read_b64+v_add_f64"};
    const int nv[4] = {40000, 20000, 20000, 20000};
    printf("{\n");
    {
        const float tm = run<0>(d, nm, 0, 1, 0);
        printf(" \"mfma_alone_ms\": %.3f, \"mfma_cycles_per_inst\": %.1f,\n", tm, tm * 1e-3 * 2.4e9 / (4.0 * nm));
        float tv[4], tb[4];
        tv[0] = run<0>(d, nm, nv[0], 0, 1); tb[0] = run<0>(d, nm, nv[0], 1, 1);
        tv[1] = run<1>(d, nm, nv[1], 0, 1); tb[1] = run<1>(d, nm, nv[1], 1, 1);
        tv[2] = run<2>(d, nm, nv[2], 0, 1); tb[2] = run<2>(d, nm, nv[2], 1, 1);
        tv[3] = run<3>(d, nm, nv[3], 0, 1); tb[3] = run<3>(d, nm, nv[3], 1, 1);
        for (int m = 0; m < 4; m++)
            printf(" \"%s\": {\"valu_alone_ms\": %.3f, \"both_ms\": %.3f, \"sum_ms\": %.3f, \"max_ms\": %.3f}%s\n", names[m], tv[m], tb[m], tv[m] + tm,
                   tv[m] > tm ? tv[m] : tm, m < 3 ? "," : "");
    }
    printf("}\n");
    return 0;
}
