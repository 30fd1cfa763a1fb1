// norm.hip — row-wise kernels between the GEMMs (gfx950).  All HBM-bound; one wave64 per token row,
// 16-byte loads, the row lives in registers between the statistics pass and the write.
//
//   layernorm_bf16        LayerNorm(D)[weight(+bias)] of the f32 residual stream -> bf16 GEMM operand
//                         (esm UnifiedTransformerBlock: attn.layernorm_qkv.0, ffn.0; TransformerStack.norm;
//                          constructed at /root/reference/slm/models/net.py:339-346)
//   layernorm_bf16_in     same on a bf16 input (RegressionHead's LayerNorm after GELU, net.py:301)
//   qk_norm_rope          full-width (D) LayerNorm of q and of k (attn.q_ln / attn.k_ln, no bias), rotary
//                         (rotate-half, base 10000) per 64-wide head, q pre-scaled by log2(e)/sqrt(64);
//                         writes token-major q,k [M, D] (same row layout as qkv, contiguous 1 KiB stores)
//                         (v is not touched: attention.hip reads it in place and transposes in its LDS reads)
#include "ed_half.h"
#include "kernels.h"

namespace ed {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// 16-bit operand conversions of this TU (ed_half.h: bf16, or f16 in the ed16 build)
__device__ __forceinline__ float bf2f(uint32_t h) { return ed_h2f(h); }
__device__ __forceinline__ uint32_t pack2(float a, float b) { return ed_pack2(a, b); }

// ---------------------------------------------------------------------------------------------
// LayerNorm: NV = ceil(D / 256) float4 per lane; D % 4 == 0.
template <int NV, bool IN_BF16>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ xin, const float* __restrict__ w,
                                                        const float* __restrict__ b, bf16_t* __restrict__ y, int M,
                                                        int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    if (c < D) {
      if constexpr (IN_BF16) {
        const uint2 p = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(xin) + (int64_t)row * D + c);
        v[j][0] = bf2f(p.x & 0xffffu); v[j][1] = bf2f(p.x >> 16);
        v[j][2] = bf2f(p.y & 0xffffu); v[j][3] = bf2f(p.y >> 16);
      } else {
        v[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(xin) + (int64_t)row * D + c);
      }
      s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    } else {
      v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wsum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    if (c < D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[j][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wsum(q) / (float)D + 1e-5f);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    if (c < D) {
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * ww[e];
      if (b) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += bb[e];
      }
      uint2 p;
      p.x = pack2(o[0], o[1]);
      p.y = pack2(o[2], o[3]);
      *reinterpret_cast<uint2*>(y + (int64_t)row * D + c) = p;
    }
  }
}

template <bool IN_BF16>
static hipError_t launch_ln(const void* x, const float* w, const float* b, bf16_t* y, int M, int D,
                            hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if (D % 4 != 0 || D > 2048) return hipErrorInvalidValue;
  const int nv = (D + 255) / 256;
  dim3 grid((M + 3) / 4), block(256);
#define ED_LN(N) hipLaunchKernelGGL((layernorm_kernel<N, IN_BF16>), grid, block, 0, stream, x, w, b, y, M, D)
  switch (nv) {
    case 1: ED_LN(1); break;
    case 2: ED_LN(2); break;
    case 3: ED_LN(3); break;
    case 4: ED_LN(4); break;
    case 5: ED_LN(5); break;
    case 6: ED_LN(6); break;
    case 7: ED_LN(7); break;
    default: ED_LN(8); break;
  }
#undef ED_LN
  return hipGetLastError();
}

hipError_t launch_layernorm_bf16(const float* x, const float* w, const float* b, bf16_t* y, int M, int D,
                                 hipStream_t stream) {
  return launch_ln<false>(x, w, b, y, M, D, stream);
}
hipError_t launch_layernorm_bf16_in(const bf16_t* x, const float* w, const float* b, bf16_t* y, int M, int D,
                                    hipStream_t stream) {
  return launch_ln<true>(x, w, b, y, M, D, stream);
}

// ---------------------------------------------------------------------------------------------
// Fused residual add + LayerNorm:  x += delta (bf16, the previous GEMM's scaled output; may be null),
// x written back (f32 residual stream), y = LayerNorm(x) * w (+ b) as the next GEMM's bf16 operand.
// This takes the residual read-modify-write out of the out-proj / FFN-down GEMM epilogues (where its
// loads sat exposed behind the MFMA main loop) and into a pure streaming kernel.  D == NV * 256 exactly.
template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_kernel(float* __restrict__ x, const bf16_t* __restrict__ delta,
                                                            const bf16_t* __restrict__ delta2,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            bf16_t* __restrict__ y, int M, int write_x) {
  constexpr int D = NV * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float* xr = x + (int64_t)row * D + lane * 4;
  f32x4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const f32x4*>(xr + j * 256);
  // v = (x + delta) + delta2, in that order: the same f32 sums as adding the two branch outputs one LayerNorm apart
  auto add_bf16 = [&](const bf16_t* dptr) {
    const bf16_t* dr = dptr + (int64_t)row * D + lane * 4;
    uint2 p[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) p[j] = *reinterpret_cast<const uint2*>(dr + j * 256);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      v[j][0] += bf2f(p[j].x & 0xffffu); v[j][1] += bf2f(p[j].x >> 16);
      v[j][2] += bf2f(p[j].y & 0xffffu); v[j][3] += bf2f(p[j].y >> 16);
    }
  };
  if (delta) add_bf16(delta);
  if (delta2) add_bf16(delta2);
  if (write_x && (delta || delta2)) {
#pragma unroll
    for (int j = 0; j < NV; ++j) *reinterpret_cast<f32x4*>(xr + j * 256) = v[j];
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  const float mean = wsum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[j][e] - mean;
      q += d * d;
    }
  const float rstd = rsqrtf(wsum(q) * (1.0f / D) + 1e-5f);
  bf16_t* yr = y + (int64_t)row * D + lane * 4;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + j * 256 + lane * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * ww[e];
    if (b) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b + j * 256 + lane * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += bb[e];
    }
    uint2 pk;
    pk.x = pack2(o[0], o[1]);
    pk.y = pack2(o[2], o[3]);
    *reinterpret_cast<uint2*>(yr + j * 256) = pk;
  }
}

// Small-batch path: the branch linear left its product as S raw f32 K-slice planes (gemm.hip: launch_gemm_partials);
// x += alpha * (P[0] + ... + P[S-1]), summed in that order; x written back; y = LayerNorm(x) * w (+ b).
// A few hundred rows at most, so latency is everything: ONE WORKGROUP PER ROW (256 threads, thread t owns float4 t and
// t + 256 of the row), every plane load of a thread issued before the first add, statistics through 4-wave LDS sums.
template <int S>
__global__ __launch_bounds__(256) void add_partials_layernorm_kernel(float* __restrict__ x, const float* __restrict__ P,
                                                                     int64_t pstride, int ldp, float alpha,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ b,
                                                                     bf16_t* __restrict__ y, int D) {
  __shared__ float red[2][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t row = blockIdx.x;
  const int nv4 = D >> 2;
  const bool has2 = t + 256 < nv4, has1 = t < nv4;
  float* xr = x + row * D;
  const float* pr = P + row * ldp;
  f32x4 xv[2], pv[2][S];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const bool on = c ? has2 : has1;
    const int col = (t + c * 256) * 4;
    if (on) {
      xv[c] = *reinterpret_cast<const f32x4*>(xr + col);
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) pv[c][s2] = *reinterpret_cast<const f32x4*>(pr + s2 * pstride + col);
    } else {
      xv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) pv[c][s2] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  f32x4 v[2];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    f32x4 acc = pv[c][0];
#pragma unroll
    for (int s2 = 1; s2 < S; ++s2) {
      acc[0] += pv[c][s2][0]; acc[1] += pv[c][s2][1]; acc[2] += pv[c][s2][2]; acc[3] += pv[c][s2][3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[c][e] = xv[c][e] + acc[e] * alpha;
    if (c ? has2 : has1) *reinterpret_cast<f32x4*>(xr + (t + c * 256) * 4) = v[c];
    sum += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
  }
  sum = wsum(sum);
  if (lane == 0) red[0][wave] = sum;
  __syncthreads();
  const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c ? has2 : has1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[c][e] - mean;
        q += d * d;
      }
    }
  q = wsum(q);
  if (lane == 0) red[1][wave] = q;
  __syncthreads();
  const float rstd = rsqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)D + 1e-5f);
  bf16_t* yr = y + row * D;
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c ? has2 : has1) {
      const int col = (t + c * 256) * 4;
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + col);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * ww[e];
      if (b) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += bb[e];
      }
      uint2 pk;
      pk.x = pack2(o[0], o[1]);
      pk.y = pack2(o[2], o[3]);
      *reinterpret_cast<uint2*>(yr + col) = pk;
    }
}

hipError_t launch_add_partials_layernorm_bf16(float* x, const GemmPartials& P, int N, float alpha, const float* w,
                                              const float* b, bf16_t* y, int M, int D, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if ((D & 3) || D > 2048 || !P.p || P.S < 1 || P.S > 8 || N < D) return hipErrorInvalidValue;
  dim3 grid(M), block(256);
#define ED_APLN(Sv) \
  hipLaunchKernelGGL(add_partials_layernorm_kernel<Sv>, grid, block, 0, stream, x, P.p, P.stride, N, alpha, w, b, y, D)
  switch (P.S) {
    case 1: ED_APLN(1); break;
    case 2: ED_APLN(2); break;
    case 3: ED_APLN(3); break;
    case 4: ED_APLN(4); break;
    case 5: ED_APLN(5); break;
    case 6: ED_APLN(6); break;
    case 7: ED_APLN(7); break;
    default: ED_APLN(8); break;
  }
#undef ED_APLN
  return hipGetLastError();
}

hipError_t launch_add_layernorm_bf16(float* x, const bf16_t* delta, const bf16_t* delta2, int write_x, const float* w,
                                     const float* b, bf16_t* y, int M, int D, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if (D % 256 != 0 || D > 2048) return hipErrorInvalidValue;
  dim3 grid((M + 3) / 4), block(256);
#define ED_ALN(N) \
  hipLaunchKernelGGL(add_layernorm_kernel<N>, grid, block, 0, stream, x, delta, delta2, w, b, y, M, write_x)
  switch (D / 256) {
    case 1: ED_ALN(1); break;
    case 2: ED_ALN(2); break;
    case 3: ED_ALN(3); break;
    case 4: ED_ALN(4); break;
    case 5: ED_ALN(5); break;
    case 6: ED_ALN(6); break;
    case 7: ED_ALN(7); break;
    default: ED_ALN(8); break;
  }
#undef ED_ALN
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// q/k LayerNorm + rotary.  One wave per token; lane holds 8 contiguous columns per 512-column slab
// (16-byte loads), NS = D / 512 slabs.  A head is 64 columns = 8 lanes; the rotate-half partner of
// column offset o is o +/- 32, i.e. lane ^ 4.
template <int NS>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const bf16_t* __restrict__ qkv,
                                                           const float* __restrict__ q_w,
                                                           const float* __restrict__ k_w,
                                                           const float* __restrict__ rope_cos,
                                                           const float* __restrict__ rope_sin,
                                                           bf16_t* __restrict__ qo, bf16_t* __restrict__ ko, int B,
                                                           int L, int H) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= B * L) return;
  const int D = H * 64;
  const int b = tok / L, l = tok - b * L;
  const float qscale = 0.125f * 1.44269504088896341f;  // 1/sqrt(64) * log2(e): softmax in base 2

  // both rows are requested before anything is reduced (6 x 16 B in flight per lane at D = 1536)
  uint4 raw[2][NS];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const bf16_t* src = qkv + (int64_t)tok * (3 * D) + which * D;
#pragma unroll
    for (int j = 0; j < NS; ++j)  // D = 1280 (the VQ-VAE decoder stack) ends in a half slab: lanes past D hold zeros and are not stored
      raw[which][j] = (j * 512 + lane * 8 < D) ? *reinterpret_cast<const uint4*>(src + j * 512 + lane * 8) : uint4{0, 0, 0, 0};
  }
  // rotary factors of this lane's 8 columns: they depend on the position and on the column offset inside the head only,
  // i.e. they are the same for every slab and for q and k -> four 16-byte loads per token instead of 16 scalar loads per
  // slab and row (r01: the kernel issued ~150 small loads per lane and ran at 4.5 TB/s instead of the copy rate)
  const int o = (lane & 7) * 8;          // column offset inside the head
  const int fi = o & 31;                 // rotary frequency index of element 0
  const bool second = o >= 32;
  float cs[8], sn[8];
  {
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(rope_cos + l * 32 + fi), c1 = *reinterpret_cast<const f32x4*>(rope_cos + l * 32 + fi + 4);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(rope_sin + l * 32 + fi), s1 = *reinterpret_cast<const f32x4*>(rope_sin + l * 32 + fi + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cs[e] = c0[e]; cs[4 + e] = c1[e];
      sn[e] = second ? s0[e] : -s0[e];   // rotate_half: first half gets -x2, second half gets +x1
      sn[4 + e] = second ? s1[e] : -s1[e];
    }
  }

#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const float* w = which ? k_w : q_w;
    float v[NS][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const uint32_t pw[4] = {raw[which][j].x, raw[which][j].y, raw[which][j].z, raw[which][j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[j][2 * e] = bf2f(pw[e] & 0xffffu);
        v[j][2 * e + 1] = bf2f(pw[e] >> 16);
        s += v[j][2 * e] + v[j][2 * e + 1];
      }
    }
    const float mean = wsum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (j * 512 + lane * 8 < D) ? v[j][e] - mean : 0.f;
        q += d * d;
      }
    const float rstd = rsqrtf(wsum(q) / (float)D + 1e-5f);
    bf16_t* dst = which ? ko : qo;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int c0 = j * 512 + lane * 8;
      const bool in = c0 < D;  // wave-uniform per slab except in the half slab; the shuffles below need every lane
      float n[8], r[8];
      const f32x4 w0 = in ? *reinterpret_cast<const f32x4*>(w + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 w1 = in ? *reinterpret_cast<const f32x4*>(w + c0 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 8; ++e) n[e] = (v[j][e] - mean) * rstd * (e < 4 ? w0[e] : w1[e - 4]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float partner = __shfl_xor(n[e], 4, 64);
        r[e] = n[e] * cs[e] + partner * sn[e];
        if (which == 0) r[e] *= qscale;
      }
      if (!in) continue;  // the shuffle partners (lane ^ 4) of a stored lane are always inside D: D % 64 == 0
      uint4 p;
      p.x = pack2(r[0], r[1]); p.y = pack2(r[2], r[3]); p.z = pack2(r[4], r[5]); p.w = pack2(r[6], r[7]);
      *reinterpret_cast<uint4*>(dst + (int64_t)tok * D + c0) = p;  // token-major: 1 KiB contiguous per wave store
    }
  }
}

hipError_t launch_qk_norm_rope(const bf16_t* qkv, const float* q_ln_w, const float* k_ln_w,
                               const float* rope_cos, const float* rope_sin, bf16_t* q, bf16_t* k,
                               int B, int L, int H, hipStream_t stream) {
  const int D = H * 64, M = B * L;
  if (M <= 0) return hipSuccess;
  if (D % 64 != 0 || D > 2048) return hipErrorInvalidValue;
  dim3 grid((M + 3) / 4), block(256);
#define ED_QK(N) \
  hipLaunchKernelGGL(qk_norm_rope_kernel<N>, grid, block, 0, stream, qkv, q_ln_w, k_ln_w, rope_cos, rope_sin, q, k, B, L, H)
  switch ((D + 511) / 512) {
    case 1: ED_QK(1); break;
    case 2: ED_QK(2); break;
    case 3: ED_QK(3); break;
    default: ED_QK(4); break;
  }
#undef ED_QK
  return hipGetLastError();
}

}  // namespace ed
