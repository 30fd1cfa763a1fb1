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
//
// ROWS (esmdiff_gibbs_step_rows): every prompt carries its own step — Philox sample index, step index, number of positions to
// unmask — so one launch serves prompts that sit at DIFFERENT steps of their chains (the certified sampler's verification batches
// and its fast lane after a roll-back).  Per prompt the arithmetic is the plain kernels', bit for bit.
// MARGIN (certified sampling, esmdiff_amd/certified.py): the same ids, plus a per-prompt report of whether logits that are only
// known up to an error could have decided otherwise.  With R a bound on the error of every logit DIFFERENCE of a row (so every
// probability is known up to the factor exp(+-R)) and E a bound on the error of a row's entropy, the three decisions of a step are
//   nucleus   a token is DEFINITELY kept when mass{z_j >= z_v - R} * exp(R) <= top_p * S (or it is the only token within R of the
//             row maximum), DEFINITELY dropped when mass{z_j >= z_v + R} * exp(-R) > top_p * S and it cannot be the maximum;
//             everything else is "possibly kept".  Two more cuts run in the same 32-step search as the draw's own cut.
//   race      the draw's winner must be definitely kept, be the best of ALL possibly-kept tokens, and lead the runner-up among
//             them by more than exp(R / temperature) (temperature 0: by more than R in the logit)
//   order     (select kernel) the largest selected entropy and the smallest unselected one must be more than 2 E apart
// Only the rows a prompt actually unmasks in this step count (the other rows' draws are discarded).  sample_flags[b] gets bit 0
// (race), bit 1 (nucleus), bit 2 (order); an unflagged prompt's new ids are those ANY logits within the bounds would have produced.
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
// three block sums at once, each in g_block_sum's order (the first one IS the draw's own nucleus mass)
__device__ __forceinline__ void g_block_sum3(float& a, float& b, float& c, float (*red3)[4], int lane, int wave) {
  a = g_wave_sum(a);
  b = g_wave_sum(b);
  c = g_wave_sum(c);
  __syncthreads();
  if (lane == 0) {
    red3[0][wave] = a;
    red3[1][wave] = b;
    red3[2][wave] = c;
  }
  __syncthreads();
  a = (red3[0][0] + red3[0][1]) + (red3[0][2] + red3[0][3]);
  b = (red3[1][0] + red3[1][1]) + (red3[1][2] + red3[1][3]);
  c = (red3[2][0] + red3[2][1]) + (red3[2][2] + red3[2][3]);
}

struct GibbsMargin {     // MARGIN kernels: how far the logits are trusted (host: launch_gibbs_step_rows)
  float R;               // bound on the error of a logit difference
  float mass_lo, mass_hi;   // exp(-R), exp(+R) (with a rounding allowance): factors on a probability mass
  float race;            // exp(R / temperature): factor between two race values (temperature 0: unused, R is the bound)
  float ent;             // 2 E: required distance between the selected and the unselected entropies; < 0: not checked
};

template <bool ROWS, bool MARGIN>
__global__ __launch_bounds__(GNT) void gibbs_row_kernel(const int64_t* __restrict__ x, const float* __restrict__ logits,
                                                        int ld, int W, float inv_temperature, float top_p,
                                                        const float* __restrict__ u, int use_philox, uint64_t seed,
                                                        uint64_t sample_offset, int step, int L,
                                                        int32_t* __restrict__ sampled, float* __restrict__ entropy,
                                                        int logits_period, int strategy,
                                                        const uint32_t* __restrict__ inv_mask,
                                                        const esmdiff_gibbs_sample_step* __restrict__ sp = nullptr,
                                                        GibbsMargin gm = GibbsMargin{}, uint8_t* __restrict__ row_flag = nullptr,
                                                        float* __restrict__ row_gap = nullptr) {
  const int row = blockIdx.x;
  if (x[row] != G_MASK) return;  // only masked positions are candidates
  uint64_t sample_index = sample_offset + (uint64_t)(row / L);
  if constexpr (ROWS) {
    const esmdiff_gibbs_sample_step q = sp[row / L];
    if (q.n_unmask <= 0) return;   // this prompt unmasks nothing in this step: no draw of its rows is used
    sample_index = q.sample_index;
    step = q.step;
  }
  __shared__ float red[4];
  __shared__ float red3[3][4];
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
  [[maybe_unused]] uint32_t tau_t = 0, tau_l = 0;   // MARGIN: the cuts at top_p * exp(-R) (tight) and top_p * exp(+R) (loose)
  [[maybe_unused]] int n_near = 0;                  // MARGIN: tokens within R of the row maximum
  if (top_p < 1.0f) {
    const float P = top_p * S;
    uint32_t kk[G_PER];
#pragma unroll
    for (int j = 0; j < G_PER; ++j) kk[j] = (t + j * GNT < W) ? g_key(zz[j]) : 0u;   // key 0 is below every candidate cut
    if constexpr (!MARGIN) {
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = tau | (1u << bit);
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < G_PER; ++j) part = part + (kk[j] >= cand ? e[j] : 0.f);
        const float mass = g_block_sum(part, red, lane, wave);
        if (mass > P) tau = cand;
      }
    } else {
      const float Pt = P * gm.mass_lo, Pl = P * gm.mass_hi;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = tau | (1u << bit), cand_t = tau_t | (1u << bit), cand_l = tau_l | (1u << bit);
        float part = 0.f, part_t = 0.f, part_l = 0.f;
#pragma unroll
        for (int j = 0; j < G_PER; ++j) {
          part = part + (kk[j] >= cand ? e[j] : 0.f);
          part_t = part_t + (kk[j] >= cand_t ? e[j] : 0.f);
          part_l = part_l + (kk[j] >= cand_l ? e[j] : 0.f);
        }
        g_block_sum3(part, part_t, part_l, red3, lane, wave);
        if (part > P) tau = cand;
        if (part_t > Pt) tau_t = cand_t;
        if (part_l > Pl) tau_l = cand_l;
      }
      float near = 0.f;
#pragma unroll
      for (int j = 0; j < G_PER; ++j) near = near + ((t + j * GNT < W && zz[j] >= m - gm.R) ? 1.f : 0.f);
      n_near = (int)g_block_sum(near, red, lane, wave);
    }
  }

  const int b = row / L, l = row - b * L;
  const float* urow = u ? u + (int64_t)row * G_NVALID : nullptr;
  float best = -1.0f;
  int best_i = 0x7fffffff;
  float fb = -3.402823466e38f;  // fall-back: the best VALID logit, used when the nucleus kept special ids only
  int fb_i = 0x7fffffff;
  // MARGIN: the two best race values over the POSSIBLY kept valid tokens; c1_p = index | (definitely kept << 16)
  [[maybe_unused]] float c1 = -1.0f, c2 = -1.0f;
  [[maybe_unused]] int c1_p = 0xffff;
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
    bool maybe = keep, sure = keep;
    if constexpr (MARGIN) {
      if (top_p < 1.0f) {
        sure = g_key(zz[j] - gm.R) > tau_t || (zz[j] == m && n_near == 1);
        maybe = !(g_key(zz[j] + gm.R) <= tau_l && zz[j] + gm.R < m);
      }
    }
    if (MARGIN ? maybe : keep) {
      float val;
      if (inv_temperature == 0.0f) {   // temperature 0: arg-max of the kept valid logits, no noise (esm's sample_logits)
        val = (zz[j] - m) + 2.0f;      // > the initial -1 for everything within 3 of the row maximum; the maximum itself is kept
      } else {
        const float w = ed_expf((zz[j] - m) * inv_temperature);
        const float uu = use_philox ? ed_philox_uniform(seed, sample_index, (uint32_t)step, (uint32_t)l, (uint32_t)v)
                                    : urow[v];
        const float g = 1e-10f - ed_logf(uu + 1e-10f);
        val = w / g;
      }
      if (keep && val > best) {
        best = val;
        best_i = v;
      }
      if constexpr (MARGIN) {
        if (val > c1) {
          c2 = c1;
          c1 = val;
          c1_p = v | (sure ? 0x10000 : 0);
        } else {
          c2 = fmaxf(c2, val);
        }
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
    if constexpr (MARGIN) {
      const float o1 = __shfl_xor(c1, off, 64), o2 = __shfl_xor(c2, off, 64);
      const int op = __shfl_xor(c1_p, off, 64);
      c2 = fmaxf(fmaxf(c2, o2), fminf(c1, o1));
      if (o1 > c1 || (o1 == c1 && (op & 0xffff) < (c1_p & 0xffff))) {
        c1 = o1;
        c1_p = op;
      }
    }
  }
  __shared__ float s_fb[4];
  __shared__ int s_fbi[4];
  __shared__ float s_c1[4], s_c2[4];
  __shared__ int s_cp[4];
  __syncthreads();
  if (lane == 0) {
    red[wave] = best;
    s_idx[wave] = best_i;
    s_fb[wave] = fb;
    s_fbi[wave] = fb_i;
    if constexpr (MARGIN) {
      s_c1[wave] = c1;
      s_c2[wave] = c2;
      s_cp[wave] = c1_p;
    }
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
    entropy[row] = strategy == 1 ? ed_philox_uniform(seed, sample_index, (uint32_t)step, (uint32_t)l, 4352u) : H;
    if constexpr (MARGIN) {
      float a1 = s_c1[0], a2 = s_c2[0];
      int ap = s_cp[0];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        a2 = fmaxf(fmaxf(a2, s_c2[w]), fminf(a1, s_c1[w]));
        if (s_c1[w] > a1 || (s_c1[w] == a1 && (s_cp[w] & 0xffff) < (ap & 0xffff))) {
          a1 = s_c1[w];
          ap = s_cp[w];
        }
      }
      // nucleus: the draw's winner is the best of everything that MIGHT be kept and is itself certainly kept (the fall-back —
      // nothing valid inside the nucleus — is never certified)
      const bool nuc_ok = bi != 0x7fffffff && (ap & 0xffff) == bi && (ap & 0x10000) != 0;
      const bool any = (ap & 0xffff) != 0xffff;   // some valid token is possibly kept (else: the fall-back, never certified)
      float gap;   // in logit units
      if (!any) gap = 0.f;
      else if (inv_temperature == 0.0f) gap = a1 - a2;
      else gap = a2 > 0.f ? (ed_logf(a1) - ed_logf(a2)) / inv_temperature : 3.402823466e38f;
      const bool race_ok = any && (inv_temperature == 0.0f ? (a1 - a2 > gm.R) : (a1 > a2 * gm.race));
      row_flag[row] = (uint8_t)((race_ok ? 0 : 1) | (nuc_ok ? 0 : 2));
      row_gap[row] = fmaxf(gap, 0.f);
    }
  }
}

// per prompt: unmask the k[b] lowest-entropy eligible positions (ties -> lower index)
template <bool ROWS, bool MARGIN>
__global__ __launch_bounds__(GNT) void gibbs_select_kernel(int64_t* __restrict__ x, const int64_t* __restrict__ seq,
                                                           const int32_t* __restrict__ sampled,
                                                           const float* __restrict__ entropy,
                                                           const int32_t* __restrict__ n_unmask, int L,
                                                           const esmdiff_gibbs_sample_step* __restrict__ sp = nullptr,
                                                           float ent_margin = -1.f, const uint8_t* __restrict__ row_flag = nullptr,
                                                           const float* __restrict__ row_gap = nullptr,
                                                           int32_t* __restrict__ sample_flags = nullptr,
                                                           float* __restrict__ sample_gaps = nullptr) {
  __shared__ float s_ent[1280];
  const int b = blockIdx.x, t = threadIdx.x;
  const int k = ROWS ? sp[b].n_unmask : n_unmask[b];
  if (k <= 0) return;
  [[maybe_unused]] int m_flag = 0;
  [[maybe_unused]] float m_gap = 3.402823466e38f, h_sel = -3.402823466e38f, h_uns = 3.402823466e38f;
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
    if constexpr (MARGIN) {
      if (rank < k) {
        m_flag |= row_flag[(int64_t)b * L + i];
        m_gap = fminf(m_gap, row_gap[(int64_t)b * L + i]);
        h_sel = fmaxf(h_sel, ei);
      } else {
        h_uns = fminf(h_uns, ei);
      }
    }
  }
  if constexpr (MARGIN) {
    __shared__ int s_f[4];
    __shared__ float s_g[4], s_hs[4], s_hu[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      m_flag |= __shfl_xor(m_flag, off, 64);
      m_gap = fminf(m_gap, __shfl_xor(m_gap, off, 64));
      h_sel = fmaxf(h_sel, __shfl_xor(h_sel, off, 64));
      h_uns = fminf(h_uns, __shfl_xor(h_uns, off, 64));
    }
    if ((t & 63) == 0) {
      s_f[t >> 6] = m_flag;
      s_g[t >> 6] = m_gap;
      s_hs[t >> 6] = h_sel;
      s_hu[t >> 6] = h_uns;
    }
    __syncthreads();
    if (t == 0) {
      int f = (s_f[0] | s_f[1]) | (s_f[2] | s_f[3]);
      const float g = fminf(fminf(s_g[0], s_g[1]), fminf(s_g[2], s_g[3]));
      const float hs = fmaxf(fmaxf(s_hs[0], s_hs[1]), fmaxf(s_hs[2], s_hs[3]));
      const float hu = fminf(fminf(s_hu[0], s_hu[1]), fminf(s_hu[2], s_hu[3]));
      // order: some position stays behind and its key is not clearly above the last selected one
      const bool both = hs != -3.402823466e38f && hu != 3.402823466e38f;
      const float hgap = both ? hu - hs : 3.402823466e38f;
      if (ent_margin >= 0.f && both && !(hgap > ent_margin)) f |= 4;
      if (sample_flags) sample_flags[b] = f;
      if (sample_gaps) {
        sample_gaps[2 * b] = g;
        sample_gaps[2 * b + 1] = hgap;
      }
    }
  }
}

hipError_t launch_gibbs_step(int64_t* x, const int64_t* seq, const float* logits, int ld, int vocab, float temperature,
                             float top_p, const int32_t* n_unmask, const float* u, int use_philox, uint64_t seed,
                             uint64_t sample_offset, int step, int32_t* sampled, float* entropy, int B, int L,
                             hipStream_t stream, int logits_period, int strategy, const uint32_t* inv_mask) {
  if (B <= 0 || L <= 0) return hipSuccess;
  if (L > 1280 || vocab < G_NVALID || vocab > G_PER * GNT || ld < vocab || !(temperature >= 0.f)) return hipErrorInvalidValue;
  if (strategy != 0 && (strategy != 1 || !use_philox)) return hipErrorInvalidValue;   // random positions need the Philox source
  hipLaunchKernelGGL((gibbs_row_kernel<false, false>), dim3(B * L), dim3(GNT), 0, stream, x, logits, ld, vocab,
                     temperature > 0.f ? 1.0f / temperature : 0.0f, top_p, u, use_philox, seed, sample_offset, step, L, sampled, entropy,
                     logits_period, strategy, inv_mask);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((gibbs_select_kernel<false, false>), dim3(B), dim3(GNT), 0, stream, x, seq, sampled, entropy, n_unmask, L);
  return hipGetLastError();
}

// the same step with one parameter set per prompt (esmdiff_gibbs_step_rows); pair_bound < 0: plain draws (flags / gaps unused),
// else the MARGIN kernels with R = pair_bound, E = entropy_bound.  row_flag [B*L] bytes, row_gap [B*L] floats: engine scratch.
hipError_t launch_gibbs_step_rows(int64_t* x, const int64_t* seq, const float* logits, int ld, int vocab, float temperature,
                                  float top_p, const esmdiff_gibbs_sample_step* sp, uint64_t seed, int32_t* sampled, float* entropy,
                                  int B, int L, float pair_bound, float entropy_bound, uint8_t* row_flag, float* row_gap,
                                  int32_t* sample_flags, float* sample_gaps, hipStream_t stream, int strategy,
                                  const uint32_t* inv_mask) {
  if (B <= 0 || L <= 0) return hipSuccess;
  if (L > 1280 || vocab < G_NVALID || vocab > G_PER * GNT || ld < vocab || !(temperature >= 0.f)) return hipErrorInvalidValue;
  if (strategy != 0 && strategy != 1) return hipErrorInvalidValue;
  const float inv_t = temperature > 0.f ? 1.0f / temperature : 0.0f;
  if (pair_bound < 0.f) {
    hipLaunchKernelGGL((gibbs_row_kernel<true, false>), dim3(B * L), dim3(GNT), 0, stream, x, logits, ld, vocab, inv_t, top_p, nullptr, 1,
                       seed, (uint64_t)0, 0, L, sampled, entropy, 0, strategy, inv_mask, sp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gibbs_select_kernel<true, false>), dim3(B), dim3(GNT), 0, stream, x, seq, sampled, entropy, nullptr, L, sp);
    return hipGetLastError();
  }
  if (!row_flag || !row_gap || !(entropy_bound >= 0.f)) return hipErrorInvalidValue;
  GibbsMargin gm;
  gm.R = pair_bound;
  // a rounding allowance on the masses (block sums of ~4 000 f32 terms): the cuts move outwards by it
  gm.mass_lo = (float)(exp(-(double)pair_bound) * (1.0 - 4e-6));
  gm.mass_hi = (float)(exp((double)pair_bound) * (1.0 + 4e-6));
  gm.race = (float)(exp((double)pair_bound * (double)inv_t) * (1.0 + 1e-6));
  gm.ent = strategy == 0 ? 2.0f * entropy_bound : -1.0f;   // "random" orders by Philox keys: no logit decides it
  hipLaunchKernelGGL((gibbs_row_kernel<true, true>), dim3(B * L), dim3(GNT), 0, stream, x, logits, ld, vocab, inv_t, top_p, nullptr, 1,
                     seed, (uint64_t)0, 0, L, sampled, entropy, 0, strategy, inv_mask, sp, gm, row_flag, row_gap);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((gibbs_select_kernel<true, true>), dim3(B), dim3(GNT), 0, stream, x, seq, sampled, entropy, nullptr, L, sp, gm.ent,
                     row_flag, row_gap, sample_flags, sample_gaps);
  return hipGetLastError();
}

}  // namespace ed
