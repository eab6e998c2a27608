// Micro-benchmark: VALU issue rate of gfx950 (MI355X) for the instruction classes the LK kernels are made of.
//
// Question it settles ("What's weak"): does a wave64 VALU instruction occupy its SIMD for 2 cycles (32 lanes / clk, as
// /opt/skills/guides/MI355X_MICROARCH.md states for the FP32 FMA rate) or for 4 cycles (16 lanes / clk), per instruction class?
// bench.py prices the LK kernels' VALU work against the answer (profiles/r02_valu_rate.json).
//
// Method: every wavefront runs `iters` rounds of CHAINS independent accumulator chains of ONE instruction (inline asm, so the compiler
// can neither fold nor reorder them), waves per SIMD are set through the dynamic-LDS footprint of a 256-thread workgroup (4 waves = 1
// per SIMD; k workgroups per CU = k waves per SIMD).  Reported per (instruction, waves/SIMD):
//   cyc_per_inst  = shader cycles (s_memtime, per wave, loop only) * waves_per_SIMD / instructions per wave   -> SIMD occupancy per wave64 instruction
//   lanes_per_clk = 64 / cyc_per_inst
//   glane_s       = lane-instructions / wall time (HIP events): the chip-level rate actually sustained (includes clock throttling)
//
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate > valu_rate.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

enum Op { DOT2_I16, PERM, ADD_U32, MAD_I24, FMA_F32, PK_FMA_F32, PK_ADD_U16, ALIGNBYTE, MUL_LO, LSHL_ADD, DPP_ADD,
          AND_B32, LSHLREV, ASHRREV, BFE_U32, ADD3, MAD_U24, MUL_I24, SUB_U32, MAX_I32, CNDMASK, CVT_F32_I32, DOT4_I8, PK_MUL_LO, PK_MAD_I16, AND_OR, MOV_DPP,
          ADD_F32, MUL_F32, FMA_F64, SAD_U8, LSHL_OR, MAD_U32_U16,
          DOT2C, DOT2C_DPP, MUL_I24_SDWA, LSHRREV, MOV, MIN_I32, BFE_I32, BITOP3, PK_SUB_I16, PK_MAD_U16, OR_B32, XOR_B32, SUB_F32, CVT_I32_F32, RNDNE_F32, CMP_CND, ADD_F64, MUL_F64, CVT_F64_I32, LSHL_ADD_U64, MAD_U64_U32, READLANE_W, N_OPS };
static const char* kNames[] = {"v_dot2_i32_i16", "v_perm_b32",       "v_add_u32",      "v_mad_i32_i24", "v_fma_f32", "v_pk_fma_f32",
                               "v_pk_add_u16",   "v_alignbyte_b32", "v_mul_lo_u32",   "v_lshl_add_u32", "v_add_u32 row_shr:1 (DPP)",
                               "v_and_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_bfe_u32", "v_add3_u32", "v_mad_u32_u24", "v_mul_i32_i24", "v_sub_u32", "v_max_i32",
                               "v_cndmask_b32", "v_cvt_f32_i32", "v_dot4_i32_i8", "v_pk_mul_lo_u16", "v_pk_mad_i16", "v_and_or_b32", "v_mov_b32 row_shr:1 (DPP)",
                               "v_add_f32", "v_mul_f32", "v_fma_f64", "v_sad_u8", "v_lshl_or_b32", "v_mad_u32_u16",
                               "v_dot2c_i32_i16 (VOP2)", "v_dot2c_i32_i16 row_shl:1 (DPP)", "v_mul_i32_i24 (SDWA)", "v_lshrrev_b32", "v_mov_b32", "v_min_i32", "v_bfe_i32", "v_bitop3_b32", "v_pk_sub_i16", "v_pk_mad_u16", "v_or_b32", "v_xor_b32", "v_sub_f32", "v_cvt_i32_f32", "v_rndne_f32", "v_cmp_lt_i32 + v_cndmask_b32 (pair)", "v_add_f64", "v_mul_f64", "v_cvt_f64_i32", "v_lshl_add_u64", "v_mad_u64_u32", "v_readlane_b32 + v_writelane_b32 (pair)"};

template <int OP>
__device__ __forceinline__ void one(unsigned& a, unsigned b, unsigned c, float& fa, float fb, float fc, float2& pa, float2 pb, float2 pc)
{
    if (OP == DOT2_I16) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
    if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == MAD_I24) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
    if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fa) : "v"(fb), "v"(fc));
    if (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa) : "v"(pb), "v"(pc));
    if (OP == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
    if (OP == DPP_ADD) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
    if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));
    if (OP == ASHRREV) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a));
    if (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a));
    if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
    if (OP == MUL_I24) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
    if (OP == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a));
    if (OP == DOT4_I8) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
    if (OP == PK_MUL_LO) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == PK_MAD_I16) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
    if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));
    if (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));
    if (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));
    if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(pa) : "v"(pb), "v"(pc));
    if (OP == SAD_U8) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
    if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
    if (OP == MAD_U32_U16) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
    if (OP == DOT2C) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == DOT2C_DPP) asm volatile("v_dot2c_i32_i16_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b), "v"(c));
    if (OP == MUL_I24_SDWA) asm volatile("v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0" : "+v"(a) : "v"(b));
    if (OP == LSHRREV) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a));
    if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
    if (OP == MIN_I32) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == BFE_I32) asm volatile("v_bfe_i32 %0, %0, 3, 17" : "+v"(a));
    if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(a) : "v"(b), "v"(c));
    if (OP == PK_SUB_I16) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == PK_MAD_U16) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OR_B32) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));
    if (OP == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a));
    if (OP == RNDNE_F32) asm volatile("v_rndne_f32 %0, %0" : "+v"(fa));
    if (OP == CMP_CND) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");
    if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(pa) : "v"(pb));
    if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(pa) : "v"(pb));
    if (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(pa) : "v"(b));
    if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(pa) : "v"(pb));
    if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(pa) : "v"(b), "v"(c) : "vcc");
    if (OP == READLANE_W) { unsigned t_; asm volatile("v_readlane_b32 %1, %0, 3\n\ts_nop 0\n\tv_writelane_b32 %0, %1, 5" : "+v"(a), "=s"(t_)); }
}

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void k_rate(unsigned* out, unsigned long long* cyc, int iters)
{
    extern __shared__ char lds_pad[];  // footprint only: limits the workgroups per CU
    unsigned a[CHAINS];
    float fa[CHAINS];
    float2 pa[CHAINS];
    const unsigned b = threadIdx.x * 2654435761u + 12345u, c = 0x03020100u ^ (threadIdx.x & 3);
    const float fb = 1.0f + 1e-7f * threadIdx.x, fc = 1e-9f;
    const float2 pb = make_float2(fb, fb), pc = make_float2(fc, fc);
#pragma unroll
    for (int j = 0; j < CHAINS; j++) { a[j] = threadIdx.x + j; fa[j] = (float)j; pa[j] = make_float2((float)j, 1.f); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 4) {  // 4 x CHAINS instructions per loop trip: the s_add / s_cmp / s_cbranch overhead stays < 5 %
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int j = 0; j < CHAINS; j++) one<OP>(a[j], b, c, fa[j], fb, fc, pa[j], pb, pc);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < CHAINS; j++) s += a[j] + __float_as_uint(fa[j]) + __float_as_uint(pa[j].x) + __float_as_uint(pa[j].y);
    if (lds_pad[0] == 77 && s == 0x12345678u) out[0] = s;  // keep everything alive without a store on the normal path
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP, int CHAINS>
static void run(int wps, int iters, unsigned* d_out, unsigned long long* d_cyc, int ncu, double memtime_hz, bool first)
{
    // k workgroups of 4 waves per CU  <=>  k waves per SIMD: give each workgroup 1/k of the 160 KB LDS (minus a margin)
    const size_t lds = wps >= 8 ? 0 : (size_t)(160 * 1024 / wps - 2048);
    CHECK(hipFuncSetAttribute((const void*)k_rate<OP, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    const int blocks = ncu * wps;  // exactly one resident round
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<OP, CHAINS>), dim3(blocks), dim3(256), lds, 0, d_out, d_cyc, iters / 8);  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<OP, CHAINS>), dim3(blocks), dim3(256), lds, 0, d_out, d_cyc, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc((size_t)blocks * 4);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : cyc) mean += (double)v;
    mean /= cyc.size();
    const double inst_per_wave = (double)iters * CHAINS;
    const double lane_inst = inst_per_wave * 64.0 * blocks * 4;
    // s_memtime ticks -> shader cycles is calibrated by main() (memtime_hz); wall-clock rate needs no calibration at all
    printf("%s    {\"inst\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"memtime_ticks_per_inst_x_waves\": %.4f, \"wall_ms\": %.4f, "
           "\"glane_inst_per_s\": %.1f, \"lanes_per_ns_per_simd\": %.3f}",
           first ? "" : ",\n", kNames[OP], CHAINS, wps, mean * wps / inst_per_wave, ms, lane_inst / (ms * 1e-3) / 1e9,
           lane_inst / (ms * 1e-3) / 1e9 / (ncu * 4.0));
    (void)memtime_hz;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

template <int OP>
static void sweep(unsigned* d_out, unsigned long long* d_cyc, int ncu, double hz, bool& first)
{
    for (int wps : {1, 2, 4}) {
        run<OP, 16>(wps, 16384, d_out, d_cyc, ncu, hz, first);
        first = false;
    }
    run<OP, 1>(1, 65536, d_out, d_cyc, ncu, hz, first);  // one dependent chain, one wave per SIMD: the instruction's issue-to-issue latency
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    unsigned* d_out;
    unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_out, (size_t)ncu * 8 * 256 * 4));
    CHECK(hipMalloc(&d_cyc, (size_t)ncu * 8 * 4 * 8));
    printf("{\n  \"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d,\n", prop.name, prop.gcnArchName, ncu, prop.clockRate);
    printf("  \"note\": \"lanes_per_ns_per_simd / (clock GHz) = lanes per cycle per SIMD; 16 independent chains per wave unless chains = 1\",\n");
    printf("  \"results\": [\n");
    bool first = true;
    sweep<DOT2_I16>(d_out, d_cyc, ncu, 0, first);
    sweep<PERM>(d_out, d_cyc, ncu, 0, first);
    sweep<ADD_U32>(d_out, d_cyc, ncu, 0, first);
    sweep<MAD_I24>(d_out, d_cyc, ncu, 0, first);
    sweep<FMA_F32>(d_out, d_cyc, ncu, 0, first);
    sweep<PK_FMA_F32>(d_out, d_cyc, ncu, 0, first);
    sweep<PK_ADD_U16>(d_out, d_cyc, ncu, 0, first);
    sweep<ALIGNBYTE>(d_out, d_cyc, ncu, 0, first);
    sweep<MUL_LO>(d_out, d_cyc, ncu, 0, first);
    sweep<LSHL_ADD>(d_out, d_cyc, ncu, 0, first);
    sweep<DPP_ADD>(d_out, d_cyc, ncu, 0, first);
    sweep<AND_B32>(d_out, d_cyc, ncu, 0, first);
    sweep<LSHLREV>(d_out, d_cyc, ncu, 0, first);
    sweep<ASHRREV>(d_out, d_cyc, ncu, 0, first);
    sweep<BFE_U32>(d_out, d_cyc, ncu, 0, first);
    sweep<ADD3>(d_out, d_cyc, ncu, 0, first);
    sweep<MAD_U24>(d_out, d_cyc, ncu, 0, first);
    sweep<MUL_I24>(d_out, d_cyc, ncu, 0, first);
    sweep<SUB_U32>(d_out, d_cyc, ncu, 0, first);
    sweep<MAX_I32>(d_out, d_cyc, ncu, 0, first);
    sweep<CNDMASK>(d_out, d_cyc, ncu, 0, first);
    sweep<CVT_F32_I32>(d_out, d_cyc, ncu, 0, first);
    sweep<DOT4_I8>(d_out, d_cyc, ncu, 0, first);
    sweep<PK_MUL_LO>(d_out, d_cyc, ncu, 0, first);
    sweep<PK_MAD_I16>(d_out, d_cyc, ncu, 0, first);
    sweep<AND_OR>(d_out, d_cyc, ncu, 0, first);
    sweep<MOV_DPP>(d_out, d_cyc, ncu, 0, first);
    sweep<ADD_F32>(d_out, d_cyc, ncu, 0, first);
    sweep<MUL_F32>(d_out, d_cyc, ncu, 0, first);
    sweep<FMA_F64>(d_out, d_cyc, ncu, 0, first);
    sweep<SAD_U8>(d_out, d_cyc, ncu, 0, first);
    sweep<LSHL_OR>(d_out, d_cyc, ncu, 0, first);
    sweep<MAD_U32_U16>(d_out, d_cyc, ncu, 0, first);
    sweep<DOT2C>(d_out, d_cyc, ncu, 0, first);
    sweep<DOT2C_DPP>(d_out, d_cyc, ncu, 0, first);
    sweep<MUL_I24_SDWA>(d_out, d_cyc, ncu, 0, first);
    sweep<LSHRREV>(d_out, d_cyc, ncu, 0, first);
    sweep<MOV>(d_out, d_cyc, ncu, 0, first);
    sweep<MIN_I32>(d_out, d_cyc, ncu, 0, first);
    sweep<BFE_I32>(d_out, d_cyc, ncu, 0, first);
    sweep<BITOP3>(d_out, d_cyc, ncu, 0, first);
    sweep<PK_SUB_I16>(d_out, d_cyc, ncu, 0, first);
    sweep<PK_MAD_U16>(d_out, d_cyc, ncu, 0, first);
    sweep<OR_B32>(d_out, d_cyc, ncu, 0, first);
    sweep<XOR_B32>(d_out, d_cyc, ncu, 0, first);
    sweep<SUB_F32>(d_out, d_cyc, ncu, 0, first);
    sweep<CVT_I32_F32>(d_out, d_cyc, ncu, 0, first);
    sweep<RNDNE_F32>(d_out, d_cyc, ncu, 0, first);
    sweep<CMP_CND>(d_out, d_cyc, ncu, 0, first);
    sweep<ADD_F64>(d_out, d_cyc, ncu, 0, first);
    sweep<MUL_F64>(d_out, d_cyc, ncu, 0, first);
    sweep<CVT_F64_I32>(d_out, d_cyc, ncu, 0, first);
    sweep<LSHL_ADD_U64>(d_out, d_cyc, ncu, 0, first);
    sweep<MAD_U64_U32>(d_out, d_cyc, ncu, 0, first);
    sweep<READLANE_W>(d_out, d_cyc, ncu, 0, first);
    printf("\n  ]\n}\n");
    return 0;
}
