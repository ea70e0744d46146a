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
//     t = z * T,   T = 2^-120 * (2 - 2^-13)     -- an fp32 number at the very BOTTOM of the format's range: exponent field 0 .. 7 --
// rounded to 13 mantissa bits; the code is bits 10 .. 25 of t's fp32 pattern (3 exponent bits + 13 mantissa bits; everything above bit
// 25 is zero).  Seven binades with a relative step of 2^-13 (z >= 2^-7), below them the subnormal range with an absolute step of
// 2^-20 of full scale -- gradual underflow gives the same resolution as round 4's t = C + z K (eight binades above 2^-7), where unorm16
// of z has an absolute step (a channel living at 0.03 keeps 8 bits): 1.4e-3 of the scores at two decades of channel spread against
// 7e-4 (tools/err_budget.py).  z = 0 is code 0: out-of-frame halo pixels stay all-zero words and need no correction.
// DECODING IS ONE INSTRUCTION PER VALUE (round 6; round 4's form took two -- mask, then shift-or of the exponent bits of 2^-7 -- and
// the depthwise stage of these blocks is bound by VALU issue): v_lshlrev_b32 with an SDWA word select puts the code at bits 10 .. 25
// and that IS the float (kernels are compiled with fp32 denormals on: .amdhsa_float_denorm_mode_32 3).  The depthwise taps carry
// 6 * 2^60 / (2 - 2^-13) (watsor_amd/engine.py), the tap sum lives around 2^-60 and is scaled back by an exact 2^60 together with
// the bias add (wz_hp_dw_finish: four packed FMAs per tile) -- the factor is split so that a large folded depthwise weight cannot
// overflow (6 / T alone is 4e36).
#ifndef WZ_HP_ASM_DEC
#define WZ_HP_ASM_DEC 1   // 0: the decoder as the compiler writes it (shift + mask: two instructions per value)
#endif
#define WZ_HP_FT 0x1.fff8p-120f          // T
#define WZ_HP_FS 0x1p+60f                // what the depthwise tap sum is scaled back by

// eight 16-bit codes -> four pairs of floats.  Linear: v_cvt_f32_u32 with SDWA word select; float form: v_lshlrev_b32 with SDWA word select.
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
            x[r] = (wz_f32x2_t){__uint_as_float((t[r] << 10) & 0x03FFFC00u), __uint_as_float((t[r] >> 6) & 0x03FFFC00u)};
        } else if constexpr (QE) {
            unsigned xlo, xhi;
            const unsigned ten = 10u;
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(xlo) : "v"(ten), "v"(t[r]));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(xhi) : "v"(ten), "v"(t[r]));
            x[r] = (wz_f32x2_t){__uint_as_float(xlo), __uint_as_float(xhi)};
        } else
            x[r] = (wz_f32x2_t){(float)(t[r] & 0xffffu), (float)(t[r] >> 16)};
    }
}
// four values d = v / 6 (before the clamp) -> two words of float-form codes
// (3.5 instructions per value: the clamp, a packed multiply by T, then per value the rounding add, a shift, and one v_and_or_b32 per
// pair.  The clamp as the output modifier of a PACKED multiply by one -- `v_pk_mul_f32 ..., 1.0 clamp`, half an instruction per value --
// assembles and does not clamp on this part: measured, garbage codes for negative pre-activations, profiles/r04_robust_program_variants.txt.)
__device__ __forceinline__ wz_f32x2_t wz_hp_clamp01_pk(const wz_f32x2_t d) {
    return (wz_f32x2_t){__builtin_amdgcn_fmed3f(d[0], 0.0f, 1.0f), __builtin_amdgcn_fmed3f(d[1], 0.0f, 1.0f)};
}
__device__ __forceinline__ wz_u32x2_t wz_hp_fenc4(const float4_t d) {
    const wz_f32x2_t t2 = {WZ_HP_FT, WZ_HP_FT};
    const wz_f32x2_t t01 = wz_hp_clamp01_pk((wz_f32x2_t){d[0], d[1]}) * t2;
    const wz_f32x2_t t23 = wz_hp_clamp01_pk((wz_f32x2_t){d[2], d[3]}) * t2;
    // round to 13 mantissa bits (half of the kept ulp, whatever the exponent: an integer add on the pattern; T's own pattern ends in ten
    // zeros, so the largest code does not carry): the code, 0 .. 65535, is then bits 10 .. 25
    const unsigned a0 = __float_as_uint(t01[0]) + 512u, a1 = __float_as_uint(t01[1]) + 512u;
    const unsigned a2 = __float_as_uint(t23[0]) + 512u, a3 = __float_as_uint(t23[1]) + 512u;
    return (wz_u32x2_t){(a0 >> 10) | ((a1 << 6) & 0xffff0000u), (a2 >> 10) | ((a3 << 6) & 0xffff0000u)};
}
// the depthwise accumulators of one output: QE -- tap sum (around 2^-60) * 2^60 + bias; else the bias was the initial value already
template <bool QE>
__device__ __forceinline__ void wz_hp_dw_finish(wz_f32x2_t dd[4], const float4_t b0, const float4_t b1) {
    if constexpr (QE) {
        const wz_f32x2_t s2 = {WZ_HP_FS, WZ_HP_FS};
        dd[0] = __builtin_elementwise_fma(dd[0], s2, __builtin_shufflevector(b0, b0, 0, 1));
        dd[1] = __builtin_elementwise_fma(dd[1], s2, __builtin_shufflevector(b0, b0, 2, 3));
        dd[2] = __builtin_elementwise_fma(dd[2], s2, __builtin_shufflevector(b1, b1, 0, 1));
        dd[3] = __builtin_elementwise_fma(dd[3], s2, __builtin_shufflevector(b1, b1, 2, 3));
    }
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
