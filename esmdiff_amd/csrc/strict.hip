// strict.hip — the float32 ("strict") precision path of the engine (gfx950).
//
// Why it exists: the reference runs its hot path in the checkpoint's float32 (checkpoint_utils.py:59-73; decode at
// /root/reference/slm/sample_esmdiff.py:40-61 in the model's dtype), and north_star states its floating-point bar for
// exactly that arithmetic: ids equal under a fixed seed, decoded backbone within 1e-4 A.  The bf16 MFMA path is the
// throughput path (0.014 max logit error, 0.05 A backbone error); this file is the same network with
//   * float32 weights and float32 activations end to end,
//   * every linear on the f32-input matrix instruction v_mfma_f32_32x32x2_f32 (each output element is ONE f32 fmaf chain over
//     k in a fixed, batch-independent order — within a group of 8 the kernel feeds k = 0,4,1,5,2,6,3,7, so it is not the
//     ascending-k chain and an external float32 sum agrees to ~1e-7 relative, not bitwise; 157 TFLOP/s peak = 1/16 of bf16),
//   * LayerNorm / rotary / softmax / SwiGLU / GELU in f32 with correctly rounded divide and sqrt and libm-grade
//     expf / erff (no fast-math, no approximate reciprocals),
//   * a fixed K order per output element, so a row's result does not depend on the batch it is computed in.
// It is selected at engine create (esmdiff_config.precision = ESMDIFF_PRECISION_F32) and is the structure decoder's
// default: decoding 100 samples is 3e13 FLOP, a fraction of a second, outside the timed metric.
//
// Kernels (all one launch per op, no split-K, no atomics):
//   gemm_f32_kernel        out = epi(A[M,K] . W[N,K]^T): 128 x 128 x 32 tiles, 4 waves x (2 x 2) MFMA 32x32 blocks,
//                          global -> registers -> LDS double buffer, one barrier per K-tile
//                          epilogues: store (+bias), bias + exact GELU, residual x = x + acc / scale
//   layernorm_f32_kernel   one wave per row, two-pass statistics, 1 / sqrt(var + 1e-5)
//   swiglu_f32_kernel      mid = silu(gate) * up of the [M, 2 FH] FFN-up output
//   qk_norm_rope_f32       full-width LayerNorm of q and k (no bias) + rotate-half rotary, f32 in / out
//   attention_f32_kernel   one query per lane, K / V tiles of 32 keys in LDS, online softmax in f32
#include "kernels.h"

namespace ed {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM.  A [M, K] row-major (lda), W [N_rows, K] row-major (nn.Linear layout), out [M, ldc].
// Tile 128 x 128, K-tile 32.  LDS rows are padded to 36 floats (144 B): the MFMA operand read is one 16-byte
// ds_read per lane at [row = lane & 31][4 * (lane >> 5) + 8 j], rows 144 B apart -> conflict-free for b128 phases.
constexpr int TM = 128, TN = 128, TK = 32, LDT = TK + 4;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                       float* __restrict__ out, const float* __restrict__ bias, int M,
                                                       int n_rows, int K, int ldc, int n_valid, float div) {
  extern __shared__ __attribute__((aligned(16))) float smem_f32[];   // 2 x (A 128 x 36 | W 128 x 36) floats = 72 KiB
  float(*sA)[TM * LDT] = reinterpret_cast<float(*)[TM * LDT]>(smem_f32);
  float(*sW)[TN * LDT] = reinterpret_cast<float(*)[TN * LDT]>(smem_f32 + 2 * TM * LDT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // global -> register staging: 128 rows x 8 float4 per operand = 1024 float4 = 4 per thread
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;   // rows lrow + 32 i
  const float* ga[4];
  const float* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = min(m0 + lrow + 32 * i, M - 1);
    const int rw = min(n0 + lrow + 32 * i, n_rows - 1);
    ga[i] = A + (int64_t)ra * lda + lcol;
    gw[i] = W + (int64_t)rw * K + lcol;
  }
  f32x4 ra4[4], rw4[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra4[i] = *reinterpret_cast<const f32x4*>(ga[i] + kt * TK);
      rw4[i] = *reinterpret_cast<const f32x4*>(gw[i] + kt * TK);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(&sA[buf][(lrow + 32 * i) * LDT + lcol]) = ra4[i];
      *reinterpret_cast<f32x4*>(&sW[buf][(lrow + 32 * i) * LDT + lcol]) = rw4[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / TK;
  gload(0);
  sstore(0);
  __syncthreads();
  const int fr = lane & 31, fk = (lane >> 5) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // 8 k per j: lanes 0-31 hold k = 8j..8j+3, lanes 32-63 k = 8j+4..8j+7
      f32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const f32x4*>(&sA[buf][(wm + 32 * i + fr) * LDT + 8 * j + fk]);
        b[i] = *reinterpret_cast<const f32x4*>(&sW[buf][(wn + 32 * i + fr) * LDT + 8 * j + fk]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D map of the 32x32 forms: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn + 32 * ni + (lane & 31);
      if (col >= n_valid) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        float v = acc[mi][ni][r];
        float* o = out + (int64_t)row * ldc + col;
        if constexpr (EPI == ESMDIFF_F32EPI_STORE) {
          *o = v + bv;
        } else if constexpr (EPI == ESMDIFF_F32EPI_BIAS_GELU) {
          v += bv;
          *o = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        } else {   // residual: x = x + (acc [+ bias]) / scale   (esm UnifiedTransformerBlock: x + r / scaling_factor)
          *o = *o + (bias ? v + bv : v) / div;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm of f32 rows -> f32.  One wave per row; D % 4 == 0, D <= 2048.
template <int NV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y, int M,
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
      v[j] = *reinterpret_cast<const f32x4*>(x + (int64_t)row * D + c);
      s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    } else {
      v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wsum64(s) / (float)D;
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
  const float rstd = 1.0f / sqrtf(wsum64(q) / (float)D + 1e-5f);
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
      *reinterpret_cast<f32x4*>(y + (int64_t)row * D + c) = o;
    }
  }
}

__global__ __launch_bounds__(256) void swiglu_f32_kernel(const float* __restrict__ gu, float* __restrict__ mid, int64_t n4,
                                                         int FH) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int per_row = FH / 4;
  const int64_t row = i / per_row;
  const int c = (int)(i - row * per_row) * 4;
  const f32x4 g = *reinterpret_cast<const f32x4*>(gu + row * 2 * FH + c);
  const f32x4 u = *reinterpret_cast<const f32x4*>(gu + row * 2 * FH + FH + c);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (g[e] / (1.0f + expf(-g[e]))) * u[e];
  *reinterpret_cast<f32x4*>(mid + row * FH + c) = o;
}

// q/k LayerNorm over the full width D (no bias) + rotary per 64-wide head (rotate-half, cos / sin tables [L][32]).
// One wave per (token, q|k); the row (<= 2048 floats) lives in registers: NV float4 per lane at c = j*256 + lane*4.
template <int NV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void qk_norm_rope_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ qw,
                                                               const float* __restrict__ kw, const float* __restrict__ rcos,
                                                               const float* __restrict__ rsin, float* __restrict__ q,
                                                               float* __restrict__ k, int M, int L, int D) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= 2 * M) return;
  const int row = item >> 1, which = item & 1;
  const float* src = qkv + (int64_t)row * 3 * D + which * D;
  const float* w = which ? kw : qw;
  float* dst = (which ? k : q) + (int64_t)row * D;
  const int l = row % L;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    if (c < D) {
      v[j] = *reinterpret_cast<const f32x4*>(src + c);
      s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    } else {
      v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wsum64(s) / (float)D;
  float qq = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    if (c < D) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[j][e] - mean;
        qq += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wsum64(qq) / (float)D + 1e-5f);
  // normalised values; the rotary partner of column c (d = c % 64) is c + 32 for d < 32 and c - 32 otherwise: lane ^ 8
  // holds it (lane * 4 -> +- 32 columns = +- 8 lanes), same j.
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = j * 256 + lane * 4;
    f32x4 n{0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) n[e] = (v[j][e] - mean) * rstd * ww[e];
    }
    f32x4 p;
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = __shfl_xor(n[e], 8, 64);
    if (c < D) {
      const int d = c & 63;
      const bool lo = d < 32;
      const f32x4 cs = *reinterpret_cast<const f32x4*>(rcos + (int64_t)l * 32 + (d & 31));
      const f32x4 sn = *reinterpret_cast<const f32x4*>(rsin + (int64_t)l * 32 + (d & 31));
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = n[e] * cs[e] + (lo ? -p[e] : p[e]) * sn[e];   // x cos + rotate_half(x) sin
      *reinterpret_cast<f32x4*>(dst + c) = o;
    }
  }
}

// Attention in f32.  Block = one wave = 64 queries of one (batch, head); lane = query.  K and V tiles of 32 keys x 64
// floats are staged in LDS by the wave itself; every lane walks the keys with broadcast LDS reads.
// softmax(q . k / 8) v with the running-maximum form (rescale only when the maximum grows).
__global__ __launch_bounds__(64) void attention_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ qkv, float* __restrict__ ctx, int L,
                                                           int H) {
  constexpr int KT = 32;
  __shared__ float sK[KT * 64];
  __shared__ float sV[KT * 64];
  const int lane = threadIdx.x;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
  const int D = H * 64;
  const int64_t base = (int64_t)b * L;
  const int qi = min(q0 + lane, L - 1);
  float qr[64], o[64];
  {
    const float* qp = q + (base + qi) * D + h * 64;
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(qp + d);
      qr[d] = t[0]; qr[d + 1] = t[1]; qr[d + 2] = t[2]; qr[d + 3] = t[3];
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.f;
  }
  float m = -INFINITY, lsum = 0.f;
  for (int k0 = 0; k0 < L; k0 += KT) {
    const int nk = min(KT, L - k0);
    __syncthreads();
    // stage: 32 rows x 16 float4 per tensor = 512 float4 -> 8 per lane
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 64 + lane;
      const int r = idx >> 4, c4 = (idx & 15) * 4;
      const int kr = min(k0 + r, L - 1);
      *reinterpret_cast<f32x4*>(&sK[r * 64 + c4]) = *reinterpret_cast<const f32x4*>(k + (base + kr) * D + h * 64 + c4);
      *reinterpret_cast<f32x4*>(&sV[r * 64 + c4]) = *reinterpret_cast<const f32x4*>(qkv + (base + kr) * 3 * D + 2 * D + h * 64 + c4);
    }
    __syncthreads();
    for (int j = 0; j < nk; ++j) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int d = 0; d < 64; d += 4) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(&sK[j * 64 + d]);
        s0 = fmaf(qr[d], kk[0], s0);
        s1 = fmaf(qr[d + 1], kk[1], s1);
        s2 = fmaf(qr[d + 2], kk[2], s2);
        s3 = fmaf(qr[d + 3], kk[3], s3);
      }
      const float s = ((s0 + s1) + (s2 + s3)) * 0.125f;
      if (s > m) {
        const float r = expf(m - s);   // exp(-inf) = 0 on the first key
        lsum *= r;
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] *= r;
        m = s;
      }
      const float p = expf(s - m);
      lsum += p;
#pragma unroll
      for (int d = 0; d < 64; d += 4) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(&sV[j * 64 + d]);
        o[d] = fmaf(p, vv[0], o[d]);
        o[d + 1] = fmaf(p, vv[1], o[d + 1]);
        o[d + 2] = fmaf(p, vv[2], o[d + 2]);
        o[d + 3] = fmaf(p, vv[3], o[d + 3]);
      }
    }
  }
  if (q0 + lane < L) {
    float* op = ctx + (base + q0 + lane) * D + h * 64;
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
      f32x4 t{o[d] / lsum, o[d + 1] / lsum, o[d + 2] / lsum, o[d + 3] / lsum};
      *reinterpret_cast<f32x4*>(op + d) = t;
    }
  }
}

}  // namespace

hipError_t launch_gemm_f32(const float* A, int lda, const float* W, float* out, const float* bias, int M, int n_rows, int K,
                           int ldc, int n_valid, float div, int epi, hipStream_t stream) {
  if (M <= 0 || n_rows <= 0) return hipSuccess;
  if (K <= 0 || K % TK || lda % 4 || n_valid > n_rows) return hipErrorInvalidValue;
  dim3 grid((n_valid + TN - 1) / TN, (M + TM - 1) / TM), block(256);
  constexpr int lds = 2 * (TM + TN) * LDT * (int)sizeof(float);
#define ED_GF(E)                                                                                                        \
  do {                                                                                                                  \
    const hipError_t a_ = ensure_dynamic_lds((const void*)gemm_f32_kernel<E>, lds);                                     \
    if (a_ != hipSuccess) return a_;                                                                                    \
    hipLaunchKernelGGL(gemm_f32_kernel<E>, grid, block, lds, stream, A, lda, W, out, bias, M, n_rows, K, ldc, n_valid, div); \
  } while (0)
  switch (epi) {
    case ESMDIFF_F32EPI_STORE: ED_GF(ESMDIFF_F32EPI_STORE); break;
    case ESMDIFF_F32EPI_BIAS_GELU: ED_GF(ESMDIFF_F32EPI_BIAS_GELU); break;
    case ESMDIFF_F32EPI_RESID_DIV: ED_GF(ESMDIFF_F32EPI_RESID_DIV); break;
    default: return hipErrorInvalidValue;
  }
#undef ED_GF
  return hipGetLastError();
}

hipError_t launch_layernorm_f32(const float* x, const float* w, const float* b, float* y, int M, int D, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if (D % 4 != 0 || D > 2048) return hipErrorInvalidValue;
  const int nv = (D + 255) / 256;
  dim3 grid((M + 3) / 4), block(256);
#define ED_LN(N) hipLaunchKernelGGL((layernorm_f32_kernel<N>), grid, block, 0, stream, x, w, b, y, M, D)
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

hipError_t launch_swiglu_f32(const float* gu, float* mid, int M, int FH, hipStream_t stream) {
  if (M <= 0) return hipSuccess;
  if (FH % 4) return hipErrorInvalidValue;
  const int64_t n4 = (int64_t)M * (FH / 4);
  hipLaunchKernelGGL(swiglu_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, gu, mid, n4, FH);
  return hipGetLastError();
}

hipError_t launch_qk_norm_rope_f32(const float* qkv, const float* q_ln_w, const float* k_ln_w, const float* rope_cos,
                                   const float* rope_sin, float* q, float* k, int B, int L, int H, hipStream_t stream) {
  const int M = B * L, D = H * 64;
  if (M <= 0) return hipSuccess;
  if (D > 2048) return hipErrorInvalidValue;
  const int nv = (D + 255) / 256;
  dim3 grid((2 * M + 3) / 4), block(256);
#define ED_QK(N) hipLaunchKernelGGL((qk_norm_rope_f32_kernel<N>), grid, block, 0, stream, qkv, q_ln_w, k_ln_w, rope_cos, rope_sin, q, k, M, L, D)
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

hipError_t launch_attention_f32(const float* q, const float* k, const float* qkv, float* ctx, int B, int L, int H,
                                hipStream_t stream) {
  if (B <= 0 || L <= 0) return hipSuccess;
  dim3 grid((L + 63) / 64, H, B), block(64);
  hipLaunchKernelGGL(attention_f32_kernel, grid, block, 0, stream, q, k, qkv, ctx, L, H);
  return hipGetLastError();
}

}  // namespace ed
