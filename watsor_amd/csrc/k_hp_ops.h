// Arithmetic shared by the split-operand block kernels (k_mbconv_hp.hip: one launch per block; k_mbconv_hp2.hip: the two-launch form
// of the 10x10 blocks): the hi + lo split, the chunk buffer's 16-bit codes (linear unorm16 / float form) and the packed depthwise FMAs.
#pragma once
#include "wz_common.h"

typedef __attribute__((ext_vector_type(2))) unsigned short wz_us2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int wz_u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int wz_u32x2_t;

__device__ __forceinline__ void wz_hp_split(const float v[8], half8_t& hi, half8_t& lo) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        hi[r] = (half_t)v[r];
        lo[r] = (half_t)(v[r] - (float)hi[r]);
    }
}

typedef __attribute__((ext_vector_type(2))) float wz_f32x2_t;

// WZ_HP_SKELETON=1 (a measurement build, never shipped: tools/micro/README.md): every global load, LDS access,
// barrier and store of the kernel stays, the ARITHMETIC goes -- a matrix instruction becomes one add that consumes its operands, a
// depthwise tap one packed add, the decoder nothing.  What such a launch takes is what this decomposition of the block into tiles
// and chunks costs in data movement, synchronisation and latency alone: the ceiling the arithmetic could at best hide under.
#ifndef WZ_HP_SKELETON
#define WZ_HP_SKELETON 0
#endif
#if WZ_HP_SKELETON
__device__ __forceinline__ float4_t wz_hp_skel_mfma(const half8_t a, const half8_t b, float4_t c) {
    c[0] += (float)a[0] + (float)b[0];
    return c;
}
#define WZ_HP_MFMA(a, b, c) wz_hp_skel_mfma(a, b, c)
#else
#define WZ_HP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif

// The 16-bit FLOAT form of the chunk buffer (QE, the robust program): a value z = relu6(v) / 6 in [0, 1] is kept as
//     t = C + z * K,   C = 2^-7,  K = (2 - 2^-13) - C      (t in [2^-7, 2): exactly eight binades)
// rounded to 13 mantissa bits; the code is bits 10 .. 25 of t's fp32 pattern (3 exponent bits + 13 mantissa bits), i.e. a relative
// step of 2^-13 whatever the channel's scale is, where unorm16 of z has an absolute step (a channel living at 0.03 keeps 8 bits) and
// unorm16 of sqrt(z) (round 3) 10 - 11 bits for such a channel: 1.4e-3 of the scores at two decades of channel spread against 7e-4
// (tools/err_budget.py, modes q / T).  z = 0 is code 0 (C is a power of two), so out-of-frame halo pixels stay all-zero words.
// Decoding is two integer operations per value and NO arithmetic: the depthwise weights carry 6 / K and the depthwise bias
// -(6 C / K) * (sum of the channel's nine taps) (watsor_amd/engine.py), exact also where taps fall on padding (code 0 decodes to C).
#ifndef WZ_HP_ASM_DEC
#define WZ_HP_ASM_DEC 1   // 0: the decoder as the compiler writes it (three instructions for the low half)
#endif
#define WZ_HP_FC 0.0078125f
#define WZ_HP_FK ((2.0f - 0.0001220703125f) - WZ_HP_FC)
#define WZ_HP_FE 0x3C000000u   // exponent field of 2^-7 (120 << 23): code 0

// eight 16-bit codes -> four pairs of floats.  Linear: v_cvt_f32_u32 with SDWA word select; float form: shift + mask-or per value.
template <bool QE = false>
__device__ __forceinline__ void wz_hp_unpack(const wz_u32x4_t t, wz_f32x2_t x[4]) {
#if WZ_HP_SKELETON
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = (wz_f32x2_t){__uint_as_float(t[r]), 0.0f};
    return;
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if constexpr (QE && !WZ_HP_ASM_DEC) {
            x[r] = (wz_f32x2_t){__uint_as_float(((t[r] << 10) & 0x03FFFC00u) | WZ_HP_FE), __uint_as_float(((t[r] >> 6) & 0x03FFFC00u) | WZ_HP_FE)};
        } else if constexpr (QE) {
            // two instructions per value: isolate the 16 bits (v_and_b32 / v_bfe_u32), then v_lshl_or_b32 puts them at bits 10 .. 25
            // under the exponent bits of 2^-7.  (Written as `(t << 10) & mask | FE` the compiler takes three for the low half.)
            unsigned lo16, xlo;
            const unsigned fe = WZ_HP_FE;
            asm("v_and_b32 %0, 0xffff, %1" : "=v"(lo16) : "v"(t[r]));
            asm("v_lshl_or_b32 %0, %1, 10, %2" : "=v"(xlo) : "v"(lo16), "v"(fe));
            x[r] = (wz_f32x2_t){__uint_as_float(xlo), __uint_as_float(((t[r] >> 16) << 10) | WZ_HP_FE)};
        } else
            x[r] = (wz_f32x2_t){(float)(t[r] & 0xffffu), (float)(t[r] >> 16)};
    }
}
// four values d = v / 6 (before the clamp) -> two words of float-form codes
// (3.5 instructions per value: the clamp (v_max_f32 with the clamp modifier), a packed fma, then per value the rounding add that also
// takes the exponent bias off, a shift, and one v_and_or_b32 per pair.  The clamp as the output modifier of a PACKED multiply by
// one -- `v_pk_mul_f32 ..., 1.0 clamp`, half an instruction per value -- assembles and does not clamp on this part: measured, garbage
// codes for negative pre-activations, profiles/r04_robust_program_variants.txt.)
__device__ __forceinline__ wz_f32x2_t wz_hp_clamp01_pk(const wz_f32x2_t d) {
    return (wz_f32x2_t){__builtin_amdgcn_fmed3f(d[0], 0.0f, 1.0f), __builtin_amdgcn_fmed3f(d[1], 0.0f, 1.0f)};
}
__device__ __forceinline__ wz_u32x2_t wz_hp_fenc4(const float4_t d) {
    const wz_f32x2_t k2 = {WZ_HP_FK, WZ_HP_FK}, c2 = {WZ_HP_FC, WZ_HP_FC};
    const wz_f32x2_t t01 = __builtin_elementwise_fma(wz_hp_clamp01_pk((wz_f32x2_t){d[0], d[1]}), k2, c2);
    const wz_f32x2_t t23 = __builtin_elementwise_fma(wz_hp_clamp01_pk((wz_f32x2_t){d[2], d[3]}), k2, c2);
    // round to 13 mantissa bits and take the exponent bias off: the code, 0 .. 65535, is then bits 10 .. 25
    const unsigned a0 = __float_as_uint(t01[0]) + (512u - WZ_HP_FE), a1 = __float_as_uint(t01[1]) + (512u - WZ_HP_FE);
    const unsigned a2 = __float_as_uint(t23[0]) + (512u - WZ_HP_FE), a3 = __float_as_uint(t23[1]) + (512u - WZ_HP_FE);
    return (wz_u32x2_t){(a0 >> 10) | ((a1 << 6) & 0xffff0000u), (a2 >> 10) | ((a3 << 6) & 0xffff0000u)};
}

// d[0..3] += x[0..3] * (w0, w1) as four v_pk_fma_f32: the depthwise stage is bound by VALU issue, and a packed FMA
// is one issue for two of them
__device__ __forceinline__ void wz_hp_fma8(wz_f32x2_t d[4], const wz_f32x2_t x[4], const float4_t w0, const float4_t w1) {
#if WZ_HP_SKELETON
    d[0] = d[0] + x[0] + x[1];
    d[1] = d[1] + x[2] + x[3];
    d[2] = d[2] + (wz_f32x2_t){w0[0], w1[0]};
    return;
#endif
    d[0] = __builtin_elementwise_fma(x[0], __builtin_shufflevector(w0, w0, 0, 1), d[0]);
    d[1] = __builtin_elementwise_fma(x[1], __builtin_shufflevector(w0, w0, 2, 3), d[1]);
    d[2] = __builtin_elementwise_fma(x[2], __builtin_shufflevector(w1, w1, 0, 1), d[2]);
    d[3] = __builtin_elementwise_fma(x[3], __builtin_shufflevector(w1, w1, 2, 3), d[3]);
}
