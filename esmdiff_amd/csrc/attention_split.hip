// attention_split.hip — float32-grade multi-head attention on the f16 MFMA (precision = ESMDIFF_PRECISION_F32_SPLIT).
//
// The strict path's attention (strict.hip::attention_f32_kernel: one query per lane, VALU fmaf chains) was 55 ms of a 310 ms
// F32_SPLIT forward at configs[1]'s batch once the linears ran on the split GEMM.  This is attention.hip's flash-style kernel
// (K / V tiles of 64 keys through LDS by LDS-DMA, S^T = K Q^T so that a query lives in a lane, online softmax, V^T fragments
// by ds_read_b64_tr_b16) with every operand split into two f16 numbers, x ~ hi + lo to 2^-22 (gemm_split.hip), and every
// product as three MFMAs, hi.lo + lo.hi + hi.hi, into the same f32 accumulator:
//   S^T  = K Q^T          2 x 4 x 3 v_mfma_f32_32x32x16_f16 per 64-key tile
//   O^T += V^T P^T        the same count; P = exp2(...) is split in registers (P <= 2^4 by the lazy-rescale threshold)
// f16 x f16 products are exact in f32; softmax statistics, the running sum and the output are f32.
//
// Scales.  f16 has 5 exponent bits, so q, k and v are multiplied by powers of two before they are split, chosen ONCE PER LAYER AT
// ENGINE CREATE from rigorous bounds (no runtime reduction, nothing can overflow):
//   q, k  come out of a bias-free LayerNorm over D followed by a rotation: |x| <= sqrt(2 (D - 1)) max|ln weight|
//   v     = Linear(LayerNorm(x) * g + b): |v| <= (sqrt(D) max|g| + |b|_2) max_row |W_v row|_2   (Cauchy-Schwarz)
// and undone on the f32 side (one fma per score: exp2(s * r - m); one factor in the final normalisation).
//
// Kernels: qk_norm_rope_split (strict.hip's q / k LayerNorm + rotary, written as [hi | lo] f16 rows, q pre-multiplied by
// log2(e) / 8), v_split (the V third of the QKV output as [hi | lo]), attention_split.
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace ed {

typedef __attribute__((ext_vector_type(8))) _Float16 sf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 sf16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 sf16x2;
typedef __attribute__((ext_vector_type(2))) float sf32x2;
typedef __attribute__((ext_vector_type(4))) float sf32x4;
typedef __attribute__((ext_vector_type(16))) float sf32x16;

namespace {

constexpr int KV_TILE = 64;
constexpr int PLANE = KV_TILE * 128;   // 8 KiB: 64 rows x 64 f16
constexpr int STAGE = 4 * PLANE;       // K hi | K lo | V hi | V lo
constexpr float P_SCALE = 1024.0f;     // p <= 2^THR = 16 -> p * 2^10 <= 2^14; lo parts normal for p >= 2^-13

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(4))) short ss16x4;
__device__ __forceinline__ sf16x4 lds_read_tr16(const char* p) {  // ds_read_b64_tr_b16
  const ss16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ss16x4*)p);
  sf16x4 r;
  __builtin_memcpy(&r, &v, 8);
  return r;
}

// two f32 -> (hi pair, lo pair) as packed f16x2 words, round to nearest even
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  const sf16x2 h = __builtin_convertvector(sf32x2{a, b}, sf16x2);
  const sf32x2 hf = __builtin_convertvector(h, sf32x2);
  const sf16x2 l = __builtin_convertvector(sf32x2{a - hf[0], b - hf[1]}, sf16x2);
  __builtin_memcpy(&hi, &h, 4);
  __builtin_memcpy(&lo, &l, 4);
}

// ---------------------------------------------------------------------------------------------------------------
// q / k: full-width LayerNorm (no bias) + rotate-half rotary in strict.hip's arithmetic, then x * scale as a [hi | lo] f16 row
// (row stride 2 D).  One wave per (token, q | k).
template <int NV, bool FULL>
__global__ __launch_bounds__(256) void qk_norm_rope_split_kernel(const float* __restrict__ qkv, const float* __restrict__ qw,
                                                                 const float* __restrict__ kw, const float* __restrict__ rcos,
                                                                 const float* __restrict__ rsin, uint16_t* __restrict__ q2,
                                                                 uint16_t* __restrict__ k2, int M, int L, int D, float q_scale,
                                                                 float k_scale) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= 2 * M) return;
  const int row = item >> 1, which = item & 1;
  const float* src = qkv + (int64_t)row * 3 * D + which * D;
  const float* w = which ? kw : qw;
  uint16_t* dst = (which ? k2 : q2) + (int64_t)row * 2 * D;
  const float sc = which ? k_scale : q_scale;
  const int l = row % L;
  sf32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    if (FULL || c < D) {
      v[j] = *reinterpret_cast<const sf32x4*>(src + c);
      s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    } else {
      v[j] = sf32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wsum64(s) / (float)D;
  float qq = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    if (FULL || c < D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[j][e] - mean;
        qq += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wsum64(qq) / (float)D + 1e-5f);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    sf32x4 n{0.f, 0.f, 0.f, 0.f};
    if (FULL || c < D) {
      const sf32x4 ww = *reinterpret_cast<const sf32x4*>(w + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) n[e] = (v[j][e] - mean) * rstd * ww[e];
    }
    sf32x4 p;
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = __shfl_xor(n[e], 8, 64);   // the rotary partner column (c +- 32) lives in lane ^ 8
    if (FULL || c < D) {
      const int d = c & 63;
      const bool lo_half = d < 32;
      const sf32x4 cs = *reinterpret_cast<const sf32x4*>(rcos + (int64_t)l * 32 + (d & 31));
      const sf32x4 sn = *reinterpret_cast<const sf32x4*>(rsin + (int64_t)l * 32 + (d & 31));
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (n[e] * cs[e] + (lo_half ? -p[e] : p[e]) * sn[e]) * sc;
      uint2 hi, lo;
      split_pair(o[0], o[1], hi.x, lo.x);
      split_pair(o[2], o[3], hi.y, lo.y);
      *reinterpret_cast<uint2*>(dst + c) = hi;
      *reinterpret_cast<uint2*>(dst + D + c) = lo;
    }
  }
}

// the V third of the QKV output [M, 3 D] f32 -> [hi | lo] f16 rows [M, 2 D] of v * scale
__global__ __launch_bounds__(256) void v_split_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ v2, int64_t n4, int D,
                                                      float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int per_row = D / 4;
  const int64_t row = i / per_row;
  const int c = (int)(i - row * per_row) * 4;
  const sf32x4 v = *reinterpret_cast<const sf32x4*>(qkv + row * 3 * D + 2 * D + c);
  uint2 hi, lo;
  split_pair(v[0] * scale, v[1] * scale, hi.x, lo.x);
  split_pair(v[2] * scale, v[3] * scale, hi.y, lo.y);
  *reinterpret_cast<uint2*>(v2 + row * 2 * D + c) = hi;
  *reinterpret_cast<uint2*>(v2 + row * 2 * D + D + c) = lo;
}

// ---------------------------------------------------------------------------------------------------------------
// q2, k2, v2: [B L, 2 D] f16 = [hi(D) | lo(D)] per token; ctx f32 [B L, D].  Workgroup = NW waves = 32 NW queries of one (batch,
// head); block ids as in attention.hip (the query blocks of a head sit on one XCD).  s_unscale = 1 / (q_scale k_scale) so that
// S_mfma * s_unscale = log2(e) / 8 * q . k;  o_unscale = 1 / (P_SCALE v_scale).
__global__ __launch_bounds__(256) void attention_split_kernel(const uint16_t* __restrict__ q2, const uint16_t* __restrict__ k2,
                                                              const uint16_t* __restrict__ v2, float* __restrict__ ctx, int L, int H,
                                                              int BH, int nqb, float s_unscale, float o_unscale) {
  extern __shared__ __attribute__((aligned(16))) char smem_as[];  // [stage][K hi | K lo | V hi | V lo], 2 stages (1 when L <= 64)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / nqb) * 8 + xcd, qb = slot - (slot / nqb) * nqb;
  if (bh >= BH) return;
  const int b = bh / H, h = bh - b * H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NW = __builtin_amdgcn_readfirstlane((int)blockDim.x >> 6);
  const int q0 = (qb * NW + wave) * 32;
  const bool active = q0 < L;  // wave-uniform
  const int qi = lane & 31, hi = lane >> 5;
  const int D = H * 64, ld = 2 * D;
  const uint16_t* kbase = k2 + (int64_t)b * L * ld + h * 64;
  const uint16_t* vbase = v2 + (int64_t)b * L * ld + h * 64;

  const int srow = lane >> 3;
  const int vchunk = (lane & 7) ^ (2 * (srow & 3));
  auto stage = [&](int buf, int kt) {
    char* base = smem_as + buf * STAGE;
    for (int g = wave; g < 8; g += NW) {  // wave-uniform
      char* d = base + g * 1024;
      const int schunk = (lane & 7) ^ (((lane >> 4) + 4 * (g & 1)) & 7);
      const int tok = min(kt * KV_TILE + g * 8 + srow, L - 1);
      const uint16_t* kr = kbase + (int64_t)tok * ld + schunk * 8;
      const uint16_t* vr = vbase + (int64_t)tok * ld + vchunk * 8;
      glds16(kr, d);
      glds16(kr + D, d + PLANE);
      glds16(vr, d + 2 * PLANE);
      glds16(vr + D, d + 3 * PLANE);
    }
  };

  sf16x8 qh[4], ql[4];
  {
    const uint16_t* qrow = q2 + ((int64_t)b * L + min(q0 + qi, L - 1)) * ld + h * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = *reinterpret_cast<const sf16x8*>(qrow + (ks * 2 + hi) * 8);
      ql[ks] = *reinterpret_cast<const sf16x8*>(qrow + D + (ks * 2 + hi) * 8);
    }
  }
  sf32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int fsw = (qi >> 1) & 7;
  const int tg = lane >> 4, ti = lane & 15;
  const int tr_row = 4 * (tg >> 1) + (ti >> 2);
  const int tr_c = 2 * (tg & 1) + ((ti & 3) >> 1);
  const int tr_swz = 2 * (ti >> 2);
  int tr_off[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) tr_off[d] = tr_row * 128 + (((d * 4 + tr_c) ^ tr_swz) << 4) + (ti & 1) * 8;
  const int nkt = (L + KV_TILE - 1) / KV_TILE;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  constexpr float THR = 4.0f;   // lazy rescale: P <= 2^4 (exact in f32; keeps P * P_SCALE inside f16)
  auto tile = [&](int kt, auto MASKED, auto HALF) {
    constexpr int NT = decltype(HALF)::value ? 1 : 2;
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
    if (active) {
      const char* kh_l = smem_as + cur * STAGE;
      const char* kl_l = kh_l + PLANE;
      const char* vh_l = kh_l + 2 * PLANE;
      const char* vl_l = kh_l + 3 * PLANE;
      sf32x16 s[2];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int off = (t * 32 + qi) * 128 + (((ks * 2 + hi) ^ fsw) << 4);
          const sf16x8 kfh = *reinterpret_cast<const sf16x8*>(kh_l + off);
          const sf16x8 kfl = *reinterpret_cast<const sf16x8*>(kl_l + off);
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl, qh[ks], s[t], 0, 0, 0);   // small terms first
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, ql[ks], s[t], 0, 0, 0);
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh, qh[ks], s[t], 0, 0, 0);
        }
      }
      if constexpr (decltype(MASKED)::value) {
        const int lim = L - kt * KV_TILE - 4 * hi;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t * 32 + (r & 3) + 8 * (r >> 2) >= lim) s[t][r] = -1e30f;
      }
      float mx = s[0][0];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * s_unscale;     // log2 units
      if (!__all(mx - m_run <= THR)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      }
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], s_unscale, -m_run));
          ps += s[t][r];
        }
      l_run += ps;
#pragma unroll
      for (int kk = 0; kk < 2 * NT; ++kk) {
        const int t = kk >> 1, r0 = (kk & 1) * 8;
        union { uint32_t u[4]; sf16x8 v; } ph, pl;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          split_pair(s[t][r0 + 2 * e] * P_SCALE, s[t][r0 + 2 * e + 1] * P_SCALE, ph.u[e], pl.u[e]);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          union { sf16x4 h[2]; sf16x8 v; } vah, val;
          vah.h[0] = lds_read_tr16(vh_l + kk * 16 * 128 + tr_off[d]);
          vah.h[1] = lds_read_tr16(vh_l + (kk * 16 + 8) * 128 + tr_off[d]);
          val.h[0] = lds_read_tr16(vl_l + kk * 16 * 128 + tr_off[d]);
          val.h[1] = lds_read_tr16(vl_l + (kk * 16 + 8) * 128 + tr_off[d]);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(val.v, ph.v, o[d], 0, 0, 0);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vah.v, pl.v, o[d], 0, 0, 0);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vah.v, ph.v, o[d], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int kt = 0; kt + 1 < nkt; ++kt) tile(kt, std::false_type{}, std::false_type{});
  const int tail = L - (nkt - 1) * KV_TILE;
  if (tail <= 32) tile(nkt - 1, std::true_type{}, std::true_type{});
  else if (tail < KV_TILE) tile(nkt - 1, std::true_type{}, std::false_type{});
  else tile(nkt - 1, std::false_type{}, std::false_type{});

  if (!active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = o_unscale / l_tot;
  const int qrow = q0 + qi;
  if (qrow < L) {
    float* dst = ctx + ((int64_t)b * L + qrow) * D + h * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        sf32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = o[d][g * 4 + e] * inv;
        *reinterpret_cast<sf32x4*>(dst + d * 32 + g * 8 + 4 * hi) = r;
      }
  }
}

// max over rows r0 .. r0 + rows - 1 of the row's L2 norm (a [*, K] weight, f32 or bf16): one wave per row
__global__ __launch_bounds__(256) void rownorm_max_kernel(const void* __restrict__ src, int dt, int64_t r0, int rows, int K,
                                                          uint32_t* __restrict__ out_bits) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < K; c += 64) {
    const int64_t i = (r0 + r) * K + c;
    const float a = dt == ESMDIFF_F32 ? reinterpret_cast<const float*>(src)[i]
                                      : __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(src)[i] << 16);
    s += a * a;
  }
  s = wsum64(s);
  if (lane == 0) atomicMax(out_bits, __float_as_uint(sqrtf(s)));
}

}  // namespace

hipError_t launch_qk_norm_rope_split(const float* qkv, const float* q_ln_w, const float* k_ln_w, const float* rope_cos,
                                     const float* rope_sin, uint16_t* q2, uint16_t* k2, int B, int L, int H, float q_scale,
                                     float k_scale, hipStream_t stream) {
  const int M = B * L, D = H * 64;
  if (M <= 0) return hipSuccess;
  if (D > 2048) return hipErrorInvalidValue;
  const int nv = (D + 255) / 256;
  dim3 grid((2 * M + 3) / 4), block(256);
#define ED_QK(N)                                                                                                              \
  do {                                                                                                                        \
    if (D == N * 256) hipLaunchKernelGGL((qk_norm_rope_split_kernel<N, true>), grid, block, 0, stream, qkv, q_ln_w, k_ln_w, rope_cos, rope_sin, q2, k2, M, L, D, q_scale, k_scale); \
    else hipLaunchKernelGGL((qk_norm_rope_split_kernel<N, false>), grid, block, 0, stream, qkv, q_ln_w, k_ln_w, rope_cos, rope_sin, q2, k2, M, L, D, q_scale, k_scale); \
  } while (0)
  switch (nv) {
    case 1: ED_QK(1); break;
    case 2: ED_QK(2); break;
    case 3: ED_QK(3); break;
    case 4: ED_QK(4); break;
    case 5: ED_QK(5); break;
    case 6: ED_QK(6); break;
    case 7: ED_QK(7); break;
    default: ED_QK(8); break;
  }
#undef ED_QK
  return hipGetLastError();
}

hipError_t launch_v_split(const float* qkv, uint16_t* v2, int M, int D, float scale, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if (D % 4) return hipErrorInvalidValue;
  const int64_t n4 = (int64_t)M * (D / 4);
  hipLaunchKernelGGL(v_split_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, qkv, v2, n4, D, scale);
  return hipGetLastError();
}

hipError_t launch_attention_split(const uint16_t* q2, const uint16_t* k2, const uint16_t* v2, float* ctx, int B, int L, int H,
                                  float qk_scale_product, float v_scale, hipStream_t stream) {
  if (B <= 0 || L <= 0) return hipSuccess;
  const int W = std::min(4, (L + 31) / 32);
  const int nw = (L + 31) / 32;
  const int nqb = (nw + W - 1) / W, BH = B * H;
  dim3 grid(8 * nqb * ((BH + 7) / 8)), block(64 * W);
  const int lds = (L <= KV_TILE ? 1 : 2) * STAGE;
  if (const hipError_t a_ = ensure_dynamic_lds((const void*)attention_split_kernel, 2 * STAGE); a_ != hipSuccess) return a_;
  hipLaunchKernelGGL(attention_split_kernel, grid, block, lds, stream, q2, k2, v2, ctx, L, H, BH, nqb, 1.0f / qk_scale_product,
                     1.0f / (P_SCALE * v_scale));
  return hipGetLastError();
}

// create time, synchronous: max L2 norm over rows [r0, r0 + rows) of a [*, K] weight -> *out [host]; scratch_bits: 4 device bytes
hipError_t weight_rownorm_max(const void* src, int src_dtype, int64_t r0, int rows, int K, uint32_t* scratch_bits, float* out) {
  hipError_t s = hipMemset(scratch_bits, 0, 4);
  if (s != hipSuccess) return s;
  hipLaunchKernelGGL(rownorm_max_kernel, dim3((rows + 3) / 4), dim3(256), 0, 0, src, src_dtype, r0, rows, K, scratch_bits);
  uint32_t bits = 0;
  s = hipMemcpy(&bits, scratch_bits, 4, hipMemcpyDeviceToHost);
  if (s != hipSuccess) return s;
  memcpy(out, &bits, 4);
  return hipGetLastError();
}

}  // namespace ed
