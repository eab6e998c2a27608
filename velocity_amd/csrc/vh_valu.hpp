// Small gfx950 VALU idioms shared by the image and track kernels: packed int16 dot products, 24-bit multiply-adds and byte / half-word shuffles
// in the forms hipcc does not select by itself.
#pragma once
#include <hip/hip_runtime.h>

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack16(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
__device__ __forceinline__ short2v as_s2(unsigned v) { return __builtin_bit_cast(short2v, v); }
__device__ __forceinline__ int dot2(unsigned a, unsigned b, int c) { return __builtin_amdgcn_sdot2(as_s2(a), as_s2(b), c, false); }
// First link of a dot2 chain.  For the builtin hipcc selects v_dot2c_i32_i16 (accumulator tied to the destination), which needs a v_mov to
// seed every chain; the VOP3P form takes the inline constant 0 as its accumulator.
__device__ __forceinline__ int dot2_first(unsigned a, unsigned b)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a * K + c for 24-bit a and an inline-constant K as ONE v_mad_i32_i24 (hipcc emits v_mul_i32_i24 + v_add for `__mul24(a, K) + c`); the _s form takes a
// wave-uniform addend from an SGPR
template <int K>
__device__ __forceinline__ int mad24_v(int a, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(K), "v"(c));
    return d;
}
template <int K>
__device__ __forceinline__ int mad24_s(int a, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(K), "s"(c));
    return d;
}
// dot2 whose accumulator must SURVIVE (the start value of a chain that is needed again: the Newton accumulators restart at -c every iteration): the
// three-address VOP3P form reads it as src2; hipcc's v_dot2c would copy it first (a v_mov per chain and iteration)
__device__ __forceinline__ int dot2_keep(unsigned a, unsigned b, int c)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// dot2 with a wave-uniform accumulator taken from an SGPR (a non-inline constant would otherwise cost a v_mov into the tied accumulator of v_dot2c)
__device__ __forceinline__ int dot2_s(unsigned a, unsigned b, int c)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
// (lo >> 16) & 0xffff | (hi >> 16) << 16 in one v_perm: descale-by-shift and int16 packing of two values that were scaled so that the
// wanted 16 bits sit in the upper half
__device__ __forceinline__ unsigned pack_hi16(int lo, int hi) { return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x07060302u); }
