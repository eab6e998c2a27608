// micro-benchmark: the shader clock a launch really runs at, by load.  s_memtime counts shader-clock cycles, s_memrealtime the constant 100 MHz reference:
// their ratio over a long dependent VALU chain is the clock in units of 100 MHz.  Launch shapes: ONE workgroup (what a one-window BA solve's
// k_ba_solve_mfma or a single-stream tracker launch looks like to the power manager), one workgroup per CU, and the full chip -- each as a single long
// launch and as a train of ~30 us launches (short kernels with gaps are what the latency-bound paths issue).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/clock_probe tools/ubench/clock_probe.hip ; prints one JSON object.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(unsigned long long* out, int iters)
{
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned v = threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) v = v * 1664525u + 1013904223u;  // dependent chain: the loop cannot be shortened
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = c1 - c0;
        out[3 * blockIdx.x + 1] = r1 - r0;
        out[3 * blockIdx.x + 2] = v;
    }
}

static double probe(int blocks, int iters, int launches)
{
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 3 * blocks);
    double ghz = 0.0;
    for (int rep = 0; rep < 3; rep++) {  // the last repetition counts
        for (int l = 0; l < launches; l++) hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(3 * blocks);
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * 3 * blocks, hipMemcpyDeviceToHost);
        double c = 0, r = 0;
        for (int b = 0; b < blocks; b++) { c += (double)h[3 * b]; r += (double)h[3 * b + 1]; }
        ghz = r > 0 ? 0.1 * c / r : 0.0;
    }
    hipFree(d);
    return ghz;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clockRate_khz\": %d,\n", p.name, cus, p.clockRate);
    // ~30 us of dependent work at 2.4 GHz: 16 ops x 4 cycles x iters ~ 72000 cycles -> iters ~ 1100; long launch: 400x that
    const int shapes[3] = {1, cus, 8 * cus};
    const char* names[3] = {"one_workgroup", "one_workgroup_per_cu", "eight_workgroups_per_cu"};
    for (int s = 0; s < 3; s++) {
        const double train = probe(shapes[s], 1100, 400), single = probe(shapes[s], 1100 * 400, 1);
        printf(" \"%s\": {\"train_of_30us_launches_ghz\": %.3f, \"one_long_launch_ghz\": %.3f}%s\n", names[s], train, single, s < 2 ? "," : "");
    }
    printf("}\n");
    return 0;
}
