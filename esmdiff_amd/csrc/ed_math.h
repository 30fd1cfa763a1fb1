/*
 * ed_math.h — exp/log in float32 built ONLY from IEEE-754 single operations whose
 * result is uniquely defined (add, mul, fma, round-to-nearest-even, int<->float
 * conversion, bit casts).  The same source compiles for the gfx950 device (HIP) and
 * for the host (gcc, used by oracle/csrc), with contraction disabled on both, so the
 * two sides produce bit-identical results.  That is what makes the fused sampler
 * kernel bit-exact against the CPU oracle ("fix a canonical reduction
 * order shared by the CPU restatement and the HIP kernel").
 *
 * Accuracy: Cephes-style minimax polynomials, ~1 ulp; the reference (torch CPU,
 * model.py:28,529,602) uses the platform's vector libm, which differs from any other
 * libm in the last ulp as well — ids can differ only at exact near-ties.
 */
#ifndef ED_MATH_H
#define ED_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define ED_HD __host__ __device__ inline __attribute__((always_inline))
#else
#include <math.h>
#include <string.h>
#define ED_HD static inline
#endif

ED_HD float ed_bits_to_float(uint32_t u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}

ED_HD uint32_t ed_float_to_bits(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}

ED_HD float ed_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

/* round to nearest even for |t| < 2^22 via the magic-constant trick (two IEEE adds) */
ED_HD float ed_rint(float t) {
  const float magic = 12582912.0f; /* 1.5 * 2^23 */
  float s = t + magic; /* not foldable without fast-math; both builds forbid it */
  return s - magic;
}

/* exp(x), x finite.  Returns 0 below the subnormal range, +inf above overflow. */
ED_HD float ed_expf(float x) {
  if (x < -104.0f) return 0.0f;
  if (x > 88.72283f) return ed_bits_to_float(0x7f800000u);
  float n = ed_rint(x * 1.44269504088896341f);
  float r = ed_fma(n, -0.693359375f, x);          /* ln2 hi (9 significant bits) */
  r = ed_fma(n, 2.12194440e-4f, r);               /* -ln2 lo */
  float p = 1.9875691500e-4f;
  p = ed_fma(p, r, 1.3981999507e-3f);
  p = ed_fma(p, r, 8.3334519073e-3f);
  p = ed_fma(p, r, 4.1665795894e-2f);
  p = ed_fma(p, r, 1.6666665459e-1f);
  p = ed_fma(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  float y = ed_fma(p, r2, r);
  y = y + 1.0f;
  int ni = (int)n;
  if (ni < -126) { /* result may be subnormal: scale in two exact-power-of-two steps */
    y = y * ed_bits_to_float((uint32_t)(ni + 100 + 127) << 23);
    return y * 7.888609052210118e-31f; /* 2^-100 */
  }
  if (ni > 127) {
    y = y * ed_bits_to_float((uint32_t)(ni - 1 + 127) << 23);
    return y * 2.0f;
  }
  return y * ed_bits_to_float((uint32_t)(ni + 127) << 23);
}

/* log(x) for normal positive x */
ED_HD float ed_logf(float x) {
  uint32_t ix = ed_float_to_bits(x);
  int e = (int)(ix >> 23) - 127;
  float m = ed_bits_to_float((ix & 0x007fffffu) | 0x3f800000u); /* [1,2) */
  if (m > 1.41421356237f) {
    m = m * 0.5f;
    e = e + 1;
  }
  float f = m - 1.0f;
  float z = f * f;
  float p = 7.0376836292e-2f;
  p = ed_fma(p, f, -1.1514610310e-1f);
  p = ed_fma(p, f, 1.1676998740e-1f);
  p = ed_fma(p, f, -1.2420140846e-1f);
  p = ed_fma(p, f, 1.4249322787e-1f);
  p = ed_fma(p, f, -1.6668057665e-1f);
  p = ed_fma(p, f, 2.0000714765e-1f);
  p = ed_fma(p, f, -2.4999993993e-1f);
  p = ed_fma(p, f, 3.3333331174e-1f);
  float y = (f * z) * p;
  float fe = (float)e;
  y = ed_fma(fe, -2.12194440e-4f, y);
  y = ed_fma(-0.5f, z, y);
  float r = f + y;
  return ed_fma(fe, 0.693359375f, r);
}

/* ---- Philox4x32-10 (Salmon et al., SC'11), counter-based noise for the sampler ---- */
typedef struct { uint32_t v[4]; } ed_u32x4;

ED_HD ed_u32x4 ed_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  ed_u32x4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

/* 24-bit uniform in [0,1): the format torch's CPU float32 rand uses */
ED_HD float ed_u32_to_uniform(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }

/* the uniform of (sample, step, position l, vocab id v) — esmdiff_rng contract in esmdiff_hip.h */
ED_HD float ed_philox_uniform(uint64_t seed, uint64_t sample, uint32_t step, uint32_t l, uint32_t v) {
  ed_u32x4 o = ed_philox4x32_10(v >> 2, l, (uint32_t)sample, step ^ ((uint32_t)(sample >> 32) << 16),
                                (uint32_t)seed, (uint32_t)(seed >> 32));
  return ed_u32_to_uniform(o.v[v & 3]);
}

#endif /* ED_MATH_H */
