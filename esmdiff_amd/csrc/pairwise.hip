// pairwise.hip — the structure decoder's pairwise confidence head: predicted aligned error and pTM (gfx950).
//
// What the reference's decode path returns as decoder_output["ptm"] / ["predicted_aligned_error"]
// (/root/reference/slm/models/utils.py:64-76 -> esm StructureTokenDecoder.decode [ESM-RECALL]):
//   qk = downproject(x)                      [B, L, 128]   (engine.hip issues this GEMM next to the other heads)
//   f(i, j) = [ q_j * k_i | q_j - k_i ]      128 features of the pair (i, j), q = qk[..., :64], k = qk[..., 64:]
//   logits = linear2(LayerNorm(GELU(linear1 f)))   -> the 64 predicted-aligned-error bins (rows 160..223 of linear2)
//   p = softmax(logits) over the bins; pairs with a special token on either side: uniform
//   PAE(i, j) = sum p * centre;   tm(i) = mean_j sum p / (1 + (centre / d0)^2);   pTM = max_i tm(i)
// The two linears run on the MFMA GEMM of gemm.hip over the materialised pair rows (L^2 rows per sample, processed in
// chunks of whole samples); this file holds the pair-feature builder and the bin reduction.  O(L^2) work per sample,
// ~4.4 GFLOP at L = 258: noise next to the 30-block decoder stack.
#include "kernels.h"

namespace ed {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f_(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t pk2_(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  const bf16x2_t v = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
  uint32_t u;
  __builtin_memcpy(&u, &v, 4);
  return u;
}

// one thread = 8 feature columns of one pair row; rows ordered (sample, i, j)
__global__ __launch_bounds__(256) void pair_features_kernel(const bf16_t* __restrict__ qk, bf16_t* __restrict__ X, int L,
                                                            int64_t n_rows) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_rows * 16) return;
  const int64_t row = idx >> 4;
  const int c = (int)(idx & 15);               // chunk of 8 columns: 0..7 product, 8..15 difference
  const int64_t LL = (int64_t)L * L;
  const int64_t b = row / LL;
  const int ij = (int)(row - b * LL), i = ij / L, j = ij - i * L;
  const int d0 = (c & 7) * 8;
  const uint4 qv = *reinterpret_cast<const uint4*>(qk + ((int64_t)b * L + j) * 128 + d0);
  const uint4 kv = *reinterpret_cast<const uint4*>(qk + ((int64_t)b * L + i) * 128 + 64 + d0);
  const uint32_t qa[4] = {qv.x, qv.y, qv.z, qv.w}, ka[4] = {kv.x, kv.y, kv.z, kv.w};
  uint32_t o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float q0 = bf2f_(qa[e] & 0xffffu), q1 = bf2f_(qa[e] >> 16);
    const float k0 = bf2f_(ka[e] & 0xffffu), k1 = bf2f_(ka[e] >> 16);
    o[e] = c < 8 ? pk2_(q0 * k0, q1 * k1) : pk2_(q0 - k0, q1 - k1);
  }
  *reinterpret_cast<uint4*>(X + row * 128 + c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

// float32 form (precision = F32): one thread = 4 feature columns of one pair row
__global__ __launch_bounds__(256) void pair_features_f32_kernel(const float* __restrict__ qk, float* __restrict__ X, int L,
                                                                int64_t n_rows) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_rows * 32) return;
  const int64_t row = idx >> 5;
  const int c = (int)(idx & 31);               // chunk of 4 columns: 0..15 product, 16..31 difference
  const int64_t LL = (int64_t)L * L;
  const int64_t b = row / LL;
  const int ij = (int)(row - b * LL), i = ij / L, j = ij - i * L;
  const int d0 = (c & 15) * 4;
  const f32x4 q = *reinterpret_cast<const f32x4*>(qk + ((int64_t)b * L + j) * 128 + d0);
  const f32x4 k = *reinterpret_cast<const f32x4*>(qk + ((int64_t)b * L + i) * 128 + 64 + d0);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = c < 16 ? q[e] * k[e] : q[e] - k[e];
  *reinterpret_cast<f32x4*>(X + row * 128 + c * 4) = o;
}

hipError_t launch_pair_features_f32(const float* qk, float* X, int nb, int L, hipStream_t stream) {
  const int64_t n_rows = (int64_t)nb * L * L;
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(pair_features_f32_kernel, dim3((unsigned)((n_rows * 32 + 255) / 256)), dim3(256), 0, stream, qk, X, L, n_rows);
  return hipGetLastError();
}

hipError_t launch_pair_features(const bf16_t* qk, bf16_t* X, int nb, int L, hipStream_t stream) {
  const int64_t n_rows = (int64_t)nb * L * L;
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(pair_features_kernel, dim3((unsigned)((n_rows * 16 + 255) / 256)), dim3(256), 0, stream, qk, X, L, n_rows);
  return hipGetLastError();
}

// Bin centres of esm's _pae_bins(max_bin, 64): linspace(0, max_bin, 63) + step / 2 with step = max_bin / 62, and one more
// centre a step further.
__device__ __forceinline__ float pae_centre(int bin, float max_bin) {
  const float step = max_bin / 62.0f;
  return bin < 63 ? (float)bin * (max_bin / 62.0f) + 0.5f * step : 62.0f * (max_bin / 62.0f) + 1.5f * step;
}

// grid (L, nb): block (i, sample).  16 lanes per pair (one float4 of the 64 bin logits each), 16 pairs per iteration.
__global__ __launch_bounds__(256) void pae_tm_kernel(const float* __restrict__ logits, const int64_t* __restrict__ tokens,
                                                     float* __restrict__ tm_rows, float* __restrict__ pae, int L,
                                                     float max_bin) {
  __shared__ float red[16];
  __shared__ int cnt[4];
  const int i = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int64_t* tok = tokens + (int64_t)b * L;
  // d0 from the number of non-special tokens of this sample
  int n = 0;
  for (int j = t; j < L; j += 256) n += tok[j] < ESMDIFF_MASK_ID ? 1 : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off, 64);
  if ((t & 63) == 0) cnt[t >> 6] = n;
  __syncthreads();
  const int n_valid = cnt[0] + cnt[1] + cnt[2] + cnt[3];
  const float d0 = 1.24f * cbrtf((float)max(n_valid, 19) - 15.0f) - 1.8f;
  const bool vi = tok[i] < ESMDIFF_MASK_ID;  // ids >= 4096 are the special tokens (mask, EOS, BOS, pad, chain break)
  const int c = t & 15, g = t >> 4;
  float fd[4], ce[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ce[e] = pae_centre(c * 4 + e, max_bin);
    const float r = ce[e] / d0;
    fd[e] = 1.0f / (1.0f + r * r);
  }
  const float* lrow = logits + ((int64_t)b * L + i) * L * 64;
  float tm_acc = 0.f;
  for (int j0 = 0; j0 < L; j0 += 16) {
    const int j = j0 + g;
    const bool inb = j < L;
    const bool valid = inb && vi && tok[j] < ESMDIFF_MASK_ID;
    f32x4 v = inb ? *reinterpret_cast<const f32x4*>(lrow + (int64_t)j * 64 + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};   // masked pair: every bin at the same value -> uniform probabilities
    float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float s = 0.f, st = 0.f, sp = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p = __expf(v[e] - mx);
      s += p;
      st += p * fd[e];
      sp += p * ce[e];
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      s += __shfl_xor(s, off, 64);
      st += __shfl_xor(st, off, 64);
      sp += __shfl_xor(sp, off, 64);
    }
    if (c == 0 && inb) {
      if (valid) tm_acc += st / s;
      if (pae) pae[((int64_t)b * L + i) * L + j] = sp / s;
    }
  }
  // sum of the 16 group leaders' partial sums (fixed order), then the masked mean over j
  if (c == 0) red[g] = tm_acc;
  __syncthreads();
  if (t == 0) {
    float tot = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) tot += red[k2];
    tm_rows[(int64_t)b * L + i] = vi ? tot / (1e-10f + (float)n_valid) : 0.f;
  }
}

__global__ __launch_bounds__(256) void row_max_kernel(const float* __restrict__ v, float* __restrict__ out, int L) {
  __shared__ float red[4];
  const int b = blockIdx.x, t = threadIdx.x;
  float m = -1e30f;
  for (int i = t; i < L; i += 256) m = fmaxf(m, v[(int64_t)b * L + i]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  if (t == 0) out[b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

hipError_t launch_pae_tm(const float* logits, const int64_t* tokens, float* tm_rows, float* pae, float* ptm, int nb, int L,
                         float max_bin, hipStream_t stream) {
  if (nb <= 0 || L <= 0) return hipSuccess;
  hipLaunchKernelGGL(pae_tm_kernel, dim3(L, nb), dim3(256), 0, stream, logits, tokens, tm_rows, pae, L, max_bin);
  hipLaunchKernelGGL(row_max_kernel, dim3(nb), dim3(256), 0, stream, tm_rows, ptm, L);
  return hipGetLastError();
}

}  // namespace ed
