// gibbs.hip — one step of entropy-ordered iterative unmasking ("gibbs" mode) on the structure track (gfx950).
//
// Replaces, per step, the per-prompt post-processing of esm.utils.generation.iterative_sampling_raw as the
// reference calls it (/root/reference/slm/sample_esmdiff.py:114-122: GenerationConfig(track="structure",
// num_steps, temperature=1.4, top_p=0.9)).  [ESM-RECALL] That function lives in the un-vendored esm==3.0.4 package;
// the semantics restated here are SURVEY.md Appendix B, in esm's order of operations: for every position, with `raw` the
// WHOLE structure-logit row (W = 4096 columns for the stock head, 4101 for the ESMDiff head the CLI samples from with
// --ckpt and the default mode, sample_esmdiff.py:257-258,265)
//     H    = entropy of softmax(raw)                                   (all W columns)
//     keep = nucleus over all W columns: sorted descending, keep the prefix whose cumulative probability is <= top_p,
//            always the top-1
//     then the special ids (>= 4096) are removed from what was kept ("mask invalid ids" comes AFTER top-p)
//     tok  ~ Categorical(softmax(kept / temperature))
// (if nothing valid survives — a special id alone holds more than top_p of the mass — the best valid id is taken)
// then per prompt the k_t lowest-entropy still-masked positions receive their sampled token, with
// k_t = still_masked - int(cos((t+1)/T * pi/2) * total_to_sample + 0.1)  (host, esmdiff_amd/gibbs.py).
//
// gibbs_row_kernel     one 256-thread workgroup per MASKED (b,l) row; the W logits are read once into
//                      registers (17 per thread, columns >= W idle).  The nucleus needs no sort: element i is kept iff the
//                      probability mass of {j : z_j >= z_i} is <= top_p, which is monotone in z_i, so the cut is
//                      found by a 32-step bitwise search over the order-preserving integer image of the floats,
//                      each step one masked block sum (wave halving tree + 4 partials, the canonical order of
//                      oracle/csrc/sampler_oracle.c).  Draw = exponential race with Philox or explicit uniforms.
// gibbs_select_kernel  one workgroup per prompt: rank of each eligible position by (entropy, index), the k
//                      smallest take their token.  L <= 1280.
// Both are HBM/latency-bound and tiny next to the forward pass.
#include "ed_math.h"
#include "kernels.h"

namespace ed {

constexpr int GNT = 256;
constexpr int G_MASK = ESMDIFF_MASK_ID;
constexpr int G_NVALID = 4096;  // VQ-VAE codebook ids; specials 4096..4100 are invalid draws
constexpr int G_PER = 17;        // 17 x 256 = 4352 >= the widest row (4101)

__device__ __forceinline__ float g_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float g_block_sum(float v, float* red, int lane, int wave) {
  v = g_wave_sum(v);
  __syncthreads();  // red reuse
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ uint32_t g_key(float f) {  // order-preserving float -> uint
  const uint32_t b = ed_float_to_bits(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(GNT) void gibbs_row_kernel(const int64_t* __restrict__ x, const float* __restrict__ logits,
                                                        int ld, int W, float inv_temperature, float top_p,
                                                        const float* __restrict__ u, int use_philox, uint64_t seed,
                                                        uint64_t sample_offset, int step, int L,
                                                        int32_t* __restrict__ sampled, float* __restrict__ entropy,
                                                        int logits_period, int strategy,
                                                        const uint32_t* __restrict__ inv_mask) {
  const int row = blockIdx.x;
  if (x[row] != G_MASK) return;  // only masked positions are candidates
  __shared__ float red[4];
  __shared__ int s_idx[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // logits_period > 0: prompt b reads the logits of prompt b % logits_period (step-0 sharing, see sampler.hip)
  const int lrow = logits_period > 0 ? ((row / L) % logits_period) * L + (row % L) : row;
  const float* z = logits + (int64_t)lrow * ld;

  float zz[G_PER];
  float m = -3.402823466e38f;
#pragma unroll
  for (int j = 0; j < G_PER; ++j) {
    zz[j] = (t + j * GNT < W) ? z[t + j * GNT] : -3.402823466e38f;
    m = fmaxf(m, zz[j]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));

  float e[G_PER];
  float acc = 0.f, accz = 0.f;
#pragma unroll
  for (int j = 0; j < G_PER; ++j) {
    const bool in = t + j * GNT < W;
    const float d = zz[j] - m;
    e[j] = in ? ed_expf(d) : 0.f;
    acc = acc + e[j];
    accz = accz + (in ? e[j] * d : 0.f);
  }
  const float S = g_block_sum(acc, red, lane, wave);
  const float A = g_block_sum(accz, red, lane, wave);
  const float H = ed_logf(S) - A / S;  // -sum p log p

  // nucleus cut: largest key tau with mass{key_j >= tau} > top_p * S ; kept <=> key_j > tau (or the maximum)
  uint32_t tau = 0;
  if (top_p < 1.0f) {
    const float P = top_p * S;
    uint32_t kk[G_PER];
#pragma unroll
    for (int j = 0; j < G_PER; ++j) kk[j] = (t + j * GNT < W) ? g_key(zz[j]) : 0u;   // key 0 is below every candidate cut
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = tau | (1u << bit);
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < G_PER; ++j) part = part + (kk[j] >= cand ? e[j] : 0.f);
      const float mass = g_block_sum(part, red, lane, wave);
      if (mass > P) tau = cand;
    }
  }

  const int b = row / L, l = row - b * L;
  const float* urow = u ? u + (int64_t)row * G_NVALID : nullptr;
  float best = -1.0f;
  int best_i = 0x7fffffff;
  float fb = -3.402823466e38f;  // fall-back: the best VALID logit, used when the nucleus kept special ids only
  int fb_i = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < G_PER; ++j) {
    const int v = t + j * GNT;
    if (v >= G_NVALID) continue;   // special ids are removed after the nucleus cut
    if (inv_mask && ((inv_mask[v >> 5] >> (v & 31)) & 1u)) continue;   // ... and so are the caller's invalid_ids
    if (zz[j] > fb) {
      fb = zz[j];
      fb_i = v;
    }
    const bool keep = (top_p >= 1.0f) || g_key(zz[j]) > tau || zz[j] == m;
    if (keep) {
      float val;
      if (inv_temperature == 0.0f) {   // temperature 0: arg-max of the kept valid logits, no noise (esm's sample_logits)
        val = (zz[j] - m) + 2.0f;      // > the initial -1 for everything within 3 of the row maximum; the maximum itself is kept
      } else {
        const float w = ed_expf((zz[j] - m) * inv_temperature);
        const float uu = use_philox ? ed_philox_uniform(seed, sample_offset + (uint64_t)b, (uint32_t)step, (uint32_t)l, (uint32_t)v)
                                    : urow[v];
        const float g = 1e-10f - ed_logf(uu + 1e-10f);
        val = w / g;
      }
      if (val > best) {
        best = val;
        best_i = v;
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(best_i, off, 64);
    if (ob > best || (ob == best && oi < best_i)) {
      best = ob;
      best_i = oi;
    }
    const float of = __shfl_xor(fb, off, 64);
    const int ofi = __shfl_xor(fb_i, off, 64);
    if (of > fb || (of == fb && ofi < fb_i)) {
      fb = of;
      fb_i = ofi;
    }
  }
  __shared__ float s_fb[4];
  __shared__ int s_fbi[4];
  __syncthreads();
  if (lane == 0) {
    red[wave] = best;
    s_idx[wave] = best_i;
    s_fb[wave] = fb;
    s_fbi[wave] = fb_i;
  }
  __syncthreads();
  if (t == 0) {
    float bb = red[0], ff = s_fb[0];
    int bi = s_idx[0], fi = s_fbi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      if (red[w] > bb || (red[w] == bb && s_idx[w] < bi)) {
        bb = red[w];
        bi = s_idx[w];
      }
      if (s_fb[w] > ff || (s_fb[w] == ff && s_fbi[w] < fi)) {
        ff = s_fb[w];
        fi = s_fbi[w];
      }
    }
    sampled[row] = bi != 0x7fffffff ? bi : fi;
    // the ordering key of gibbs_select_kernel: the entropy (strategy "entropy"), or a uniform per position drawn from a
    // Philox column no token draw uses (strategy "random": the k smallest keys are a uniformly random k-subset)
    entropy[row] = strategy == 1 ? ed_philox_uniform(seed, sample_offset + (uint64_t)b, (uint32_t)step, (uint32_t)l, 4352u) : H;
  }
}

// per prompt: unmask the k[b] lowest-entropy eligible positions (ties -> lower index)
__global__ __launch_bounds__(GNT) void gibbs_select_kernel(int64_t* __restrict__ x, const int64_t* __restrict__ seq,
                                                           const int32_t* __restrict__ sampled,
                                                           const float* __restrict__ entropy,
                                                           const int32_t* __restrict__ n_unmask, int L) {
  __shared__ float s_ent[1280];
  const int b = blockIdx.x, t = threadIdx.x;
  const int k = n_unmask[b];
  if (k <= 0) return;
  for (int l = t; l < L; l += GNT) {
    const int64_t s = seq[(int64_t)b * L + l];
    const bool eligible = x[(int64_t)b * L + l] == G_MASK && s != 0 && s != 1 && s != 2;  // not BOS / PAD / EOS
    s_ent[l] = eligible ? entropy[(int64_t)b * L + l] : 3.402823466e38f;
  }
  __syncthreads();
  for (int i = t; i < L; i += GNT) {
    const float ei = s_ent[i];
    if (ei == 3.402823466e38f) continue;
    int rank = 0;
    for (int j = 0; j < L; ++j) {
      const float ej = s_ent[j];
      rank += (ej < ei || (ej == ei && j < i)) ? 1 : 0;
    }
    if (rank < k) x[(int64_t)b * L + i] = (int64_t)sampled[(int64_t)b * L + i];
  }
}

hipError_t launch_gibbs_step(int64_t* x, const int64_t* seq, const float* logits, int ld, int vocab, float temperature,
                             float top_p, const int32_t* n_unmask, const float* u, int use_philox, uint64_t seed,
                             uint64_t sample_offset, int step, int32_t* sampled, float* entropy, int B, int L,
                             hipStream_t stream, int logits_period, int strategy, const uint32_t* inv_mask) {
  if (B <= 0 || L <= 0) return hipSuccess;
  if (L > 1280 || vocab < G_NVALID || vocab > G_PER * GNT || ld < vocab || !(temperature >= 0.f)) return hipErrorInvalidValue;
  if (strategy != 0 && (strategy != 1 || !use_philox)) return hipErrorInvalidValue;   // random positions need the Philox source
  hipLaunchKernelGGL(gibbs_row_kernel, dim3(B * L), dim3(GNT), 0, stream, x, logits, ld, vocab, temperature > 0.f ? 1.0f / temperature : 0.0f, top_p, u,
                     use_philox, seed, sample_offset, step, L, sampled, entropy, logits_period, strategy, inv_mask);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gibbs_select_kernel, dim3(B), dim3(GNT), 0, stream, x, seq, sampled, entropy, n_unmask, L);
  return hipGetLastError();
}

}  // namespace ed
