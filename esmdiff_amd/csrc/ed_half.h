// ed_half.h — the 16-bit operand type of the throughput path, chosen per translation unit.
//
// The MFMA kernels (gemm*.hip, attention.hip), the row kernels that feed them (norm.hip, geom.hip) and the weight
// conversion (convert.hip) are written against the few primitives below and compiled TWICE by esmdiff_amd/build.py:
//   as-is                      namespace ed     operands are bfloat16  (8-bit significand; the reference's GPU dtype for stock
//                                               ESM3 and BASELINE configs[1]'s "bf16")
//   -DED_F16 -Ded=ed16         namespace ed16   operands are IEEE half (11-bit significand) on the f16 forms of the same MFMA
//                                               instructions: same rate, same bytes, 1/8 of the rounding error
// f16 has 5 exponent bits: weights (|w| <= 65504, precision floor 6e-8 absolute) and LayerNorm / attention / SwiGLU outputs
// of this network sit comfortably inside; conversions saturate to +-65504 instead of producing inf (ed_sat).
// esmdiff_config.precision = ESMDIFF_PRECISION_F16 selects the ed16 kernels at engine create.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(2))) float ed_f32x2;
#ifdef ED_F16
typedef _Float16 ed_half_t;
#define ED_HALF_IS_F16 1
#define ED_MFMA_32x32x16_ASM "v_mfma_f32_32x32x16_f16"
#define ED_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 ed_half_t;
#define ED_HALF_IS_F16 0
#define ED_MFMA_32x32x16_ASM "v_mfma_f32_32x32x16_bf16"
#define ED_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(2))) ed_half_t ed_half2;
typedef __attribute__((ext_vector_type(4))) ed_half_t ed_half4;
typedef __attribute__((ext_vector_type(8))) ed_half_t ed_half8;

// raw 16 bits (low half of h) -> f32
__device__ __forceinline__ float ed_h2f(uint32_t h) {
#ifdef ED_F16
  const uint16_t b = (uint16_t)h;
  _Float16 v;
  __builtin_memcpy(&v, &b, 2);
  return (float)v;
#else
  return __uint_as_float(h << 16);
#endif
}
__device__ __forceinline__ float ed_sat(float a) {
#ifdef ED_F16
  return __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);   // one v_med3_f32: clamp to the finite f16 range
#else
  return a;
#endif
}
// two f32 -> two packed 16-bit values, round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32); saturating in the f16 build
__device__ __forceinline__ uint32_t ed_pack2(float a, float b) {
  const ed_half2 v = __builtin_convertvector(ed_f32x2{ed_sat(a), ed_sat(b)}, ed_half2);
  uint32_t u;
  __builtin_memcpy(&u, &v, 4);
  return u;
}
// ... without the clamp, for values bounded by construction (softmax weights <= 2^8, convex combinations of V rows)
__device__ __forceinline__ uint32_t ed_pack2_bounded(float a, float b) {
  const ed_half2 v = __builtin_convertvector(ed_f32x2{a, b}, ed_half2);
  uint32_t u;
  __builtin_memcpy(&u, &v, 4);
  return u;
}
__device__ __forceinline__ uint16_t ed_f2h(float a) { return (uint16_t)(ed_pack2(a, 0.f) & 0xffffu); }
