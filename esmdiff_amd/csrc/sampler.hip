// sampler.hip — fused reverse-diffusion update for one batch of token rows (gfx950).
//
// Replaces, in ONE pass over the raw network logits,
//   logits_parameterization   /root/reference/slm/models/model.py:527-533
//   q_xs + mask column        /root/reference/slm/models/model.py:602-603
//   _sample_categorical       /root/reference/slm/models/model.py:24-28
//   carry-over                /root/reference/slm/models/model.py:606-607
//   noise-removal argmax      /root/reference/slm/models/model.py:575-579
// which the reference runs as ~10 elementwise/reduction passes over a (B,L,4101) float tensor.
//
// Mapping: one 256-thread workgroup (4 waves) per (b,l) row; the row (4101 floats, 16.4 KB) is read
// ONCE into registers (17 per thread, coalesced dword loads) and every later stage works from
// registers.  Rows whose token is already unmasked exit immediately (they keep their id and, with
// position-indexed noise, consume nothing).  HBM-bound: algorithmic bytes per masked row = 4·V (logits)
// [+ 4·V explicit uniforms in parity mode].
//
// Float-operation order is the canonical one documented in oracle/csrc/sampler_oracle.c; this TU is
// compiled with -ffp-contract=off so that ed_math.h evaluates identically on host and device.
#include "ed_math.h"
#include "kernels.h"

namespace ed {

constexpr int NT = 256;
constexpr int MASK_ID = ESMDIFF_MASK_ID;
constexpr int MAX_PER_THREAD = 20;  // supports V <= 5120

__device__ __forceinline__ float wave_halving_sum(float v) {
  // t[i] = t[i] + t[i+off], off = 32..1; lane 0 ends with the canonical tree value
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// MARGIN (esmdiff_ddpm_step_margin): the same draw, plus the runner-up of the arg-max: sample_flags[b] is set when, for some
// masked row of sample b, the winner does not beat the runner-up by the factor `margin` (final pass: by the difference
// `margin`) — i.e. when a perturbation of the logits of that size could change the id.  The id written is the same bit for bit.
//
// ROWS (esmdiff_ddpm_step_rows): every sample of the batch carries its own update — Philox sample index, move chances, step
// index, final flag — in `sp[b]`, so one launch serves samples that sit at DIFFERENT updates of their chains (the certified
// sampler's verification batches and its fast lane after a roll-back).  Per sample the arithmetic is the plain kernel's, bit
// for bit.  `margin` is then the ratio bound of the updates and `margin_fin` the difference bound of final passes;
// sample_flags may be NULL; sample_gap[b] (optional, initialised to +inf by the caller) receives the smallest winner-to-
// runner-up gap of the sample's masked rows in log units (log ratio for an update, log-probability difference for a final
// pass) — a statistic for choosing eps, not part of the draw.
template <int PER, bool MARGIN = false, bool ROWS = false>
__global__ __launch_bounds__(NT) void ddpm_step_kernel(int64_t* __restrict__ x, const float* __restrict__ logits,
                                                       int ld, int V, float mc_t, float mc_s, int final_,
                                                       const float* __restrict__ u, int use_philox,
                                                       uint64_t seed, uint64_t sample_offset, int step, int L,
                                                       int logits_period, float margin = 0.f,
                                                       int32_t* __restrict__ sample_flags = nullptr,
                                                       const esmdiff_sample_step* __restrict__ sp = nullptr,
                                                       float margin_fin = 0.f, float* __restrict__ sample_gap = nullptr) {
  const int row = blockIdx.x;
  if (x[row] != MASK_ID) return;  // carry-over: copy_flag * x  (model.py:606-607)
  uint64_t sample_index = 0;
  if constexpr (ROWS) {
    const esmdiff_sample_step q = sp[row / L];
    mc_t = q.move_chance_t;
    mc_s = q.move_chance_s;
    step = q.step;
    final_ = q.final;
    sample_index = q.sample_index;
    if (final_) margin = margin_fin;
  }

  __shared__ float s_red[8];
  __shared__ int s_idx[4];
  __shared__ float s_sec[4];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  // logits_period > 0: sample b reads the logits of sample b % logits_period (step-0 sharing: every sample of the batch
  // had identical inputs, engine.hip::esmdiff_ddpm_sample); the noise stays keyed by the sample's own index
  const int lrow = logits_period > 0 ? ((row / L) % logits_period) * L + (row % L) : row;
  const float* z = logits + (int64_t)lrow * ld;

  float zz[PER];
  float m = -3.402823466e38f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int v = t + j * NT;
    float val = -3.402823466e38f;
    if (v < V) {
      val = z[v];
      if (v == MASK_ID) val = val + -1000000.0f;  // logits[:, :, mask] += neg_infinity
      m = fmaxf(m, val);
    }
    zz[j] = val;
  }
  m = wave_max(m);
  if (lane == 0) s_red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));

  float acc = 0.0f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int v = t + j * NT;
    if (v < V) acc = acc + ed_expf(zz[j] - m);
  }
  acc = wave_halving_sum(acc);
  if (lane == 0) s_red[4 + wave] = acc;
  __syncthreads();
  const float s = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
  const float lse = m + ed_logf(s);

  const float d = mc_t - mc_s;
  const int b = row / L, l = row - b * L;
  const float* urow = u ? u + (int64_t)row * V : nullptr;
  float best = -3.402823466e38f, second = -3.402823466e38f;
  int best_i = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int v = t + j * NT;
    if (v < V) {
      const float lp = zz[j] - lse;
      float val;
      if (final_) {
        val = lp;
      } else {
        float q = ed_expf(lp) * d;
        if (v == MASK_ID) q = mc_s;
        float uu;
        if (use_philox)
          uu = ed_philox_uniform(seed, ROWS ? sample_index : sample_offset + (uint64_t)b, (uint32_t)step, (uint32_t)l, (uint32_t)v);
        else
          uu = urow[v];
        const float g = 1e-10f - ed_logf(uu + 1e-10f);
        val = q / g;
      }
      if (best_i == 0x7fffffff || val > best) {
        if constexpr (MARGIN) second = best_i == 0x7fffffff ? second : best;
        best = val;
        best_i = v;
      } else if constexpr (MARGIN) {
        second = fmaxf(second, val);
      }
    }
  }
  // argmax with lowest-index tie-break; (max, min index) is order-independent
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(best_i, off, 64);
    if constexpr (MARGIN) {   // runner-up of the union: the smaller of the two winners, or either side's runner-up
      const float os = __shfl_xor(second, off, 64);
      second = fmaxf(fmaxf(second, os), fminf(best, ob));
    }
    if (ob > best || (ob == best && oi < best_i)) {
      best = ob;
      best_i = oi;
    }
  }
  __syncthreads();  // s_red reuse
  if (lane == 0) {
    s_red[wave] = best;
    s_idx[wave] = best_i;
    if constexpr (MARGIN) s_sec[wave] = second;
  }
  __syncthreads();
  if (t == 0) {
    float bb = s_red[0];
    int bi = s_idx[0];
    float ss = MARGIN ? s_sec[0] : 0.f;
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      if constexpr (MARGIN) ss = fmaxf(fmaxf(ss, s_sec[w]), fminf(bb, s_red[w]));
      if (s_red[w] > bb || (s_red[w] == bb && s_idx[w] < bi)) {
        bb = s_red[w];
        bi = s_idx[w];
      }
    }
    x[row] = (int64_t)bi;
    if constexpr (MARGIN) {
      // non-final: values are q / g > 0, the criterion is a ratio; final: log-probabilities, a difference
      const bool safe = final_ ? (bb - ss > margin) : (bb > ss * margin);
      if constexpr (ROWS) {
        if (!safe && sample_flags) sample_flags[row / L] = 1;
        if (sample_gap) {   // gaps are >= 0, so their float order is their bit order as signed integers
          const float gap = final_ ? bb - ss : ed_logf(bb) - ed_logf(fmaxf(ss, 1e-37f));
          atomicMin(reinterpret_cast<int*>(sample_gap) + row / L, __float_as_int(fmaxf(gap, 0.f)));
        }
      } else {
        if (!safe) sample_flags[row / L] = 1;    // (benign race: every writer stores 1)
      }
    }
  }
}

hipError_t launch_ddpm_step(int64_t* x, const float* logits, int ld, int V, float mc_t, float mc_s, int final_,
                            const float* u, int use_philox, uint64_t seed, uint64_t sample_offset, int step,
                            int B, int L, hipStream_t stream, int logits_period, float margin, int32_t* sample_flags) {
  const int rows = B * L;
  if (rows <= 0) return hipSuccess;
  const int per = (V + NT - 1) / NT;
  if (per > MAX_PER_THREAD) return hipErrorInvalidValue;
  dim3 grid(rows), block(NT);
#define ED_LAUNCH(P)                                                                                                       \
  do {                                                                                                                     \
    if (sample_flags)                                                                                                      \
      hipLaunchKernelGGL((ddpm_step_kernel<P, true>), grid, block, 0, stream, x, logits, ld, V, mc_t, mc_s, final_, u,     \
                         use_philox, seed, sample_offset, step, L, logits_period, margin, sample_flags);                   \
    else                                                                                                                   \
      hipLaunchKernelGGL((ddpm_step_kernel<P, false>), grid, block, 0, stream, x, logits, ld, V, mc_t, mc_s, final_, u,    \
                         use_philox, seed, sample_offset, step, L, logits_period, 0.f, nullptr);                           \
  } while (0)
  if (per <= 1) ED_LAUNCH(1);
  else if (per <= 4) ED_LAUNCH(4);
  else if (per <= 17) ED_LAUNCH(17);
  else ED_LAUNCH(MAX_PER_THREAD);
#undef ED_LAUNCH
  return hipGetLastError();
}

hipError_t launch_ddpm_step_rows(int64_t* x, const float* logits, int ld, int V, const esmdiff_sample_step* sp, uint64_t seed,
                                 int B, int L, float margin_ratio, float margin_diff, int32_t* sample_flags, float* sample_gap,
                                 hipStream_t stream) {
  const int rows = B * L;
  if (rows <= 0) return hipSuccess;
  const int per = (V + NT - 1) / NT;
  if (per > MAX_PER_THREAD) return hipErrorInvalidValue;
  dim3 grid(rows), block(NT);
#define ED_LAUNCH(P)                                                                                                       \
  hipLaunchKernelGGL((ddpm_step_kernel<P, true, true>), grid, block, 0, stream, x, logits, ld, V, 0.f, 0.f, 0, nullptr, 1,  \
                     seed, (uint64_t)0, 0, L, 0, margin_ratio, sample_flags, sp, margin_diff, sample_gap)
  if (per <= 1) ED_LAUNCH(1);
  else if (per <= 4) ED_LAUNCH(4);
  else if (per <= 17) ED_LAUNCH(17);
  else ED_LAUNCH(MAX_PER_THREAD);
#undef ED_LAUNCH
  return hipGetLastError();
}

// Logit-error statistics of one engine against another on the same input (certified sampling: how far the fast engine's logits are
// from the f32-grade ones).  One workgroup per token row; rows that are not MASK (their draw does not read the logits) give zeros.
// e_v = a_v - b_v over the columns a decision reads: all but the MASK column for the ddpm draw (all_columns = 0; that column is
// pushed to -1e6 before anything looks at it, model.py:528), every column for the gibbs step (all_columns = 1: the nucleus and the
// entropy are taken over the whole row).  d_v = e_v - e_{v+1} is the error of the logit difference of two NEIGHBOURING tokens (a
// sample of the pair-error distribution: 4 099 of them per row); the pair that decides a draw is arbitrary, and its error is at
// most the row's RANGE max e - min e.  H = entropy of softmax over the same columns (plain expf / logf: a statistic, not a draw).
// out[row] = { max |e|, sum e^2, max |d|, sum d^2, max e - min e, H(a) - H(b), H(b), 0 }.
constexpr int STATS_W = 8;
__global__ __launch_bounds__(NT) void logit_error_stats_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                               int ldb, const int64_t* __restrict__ x, int V, int all_columns,
                                                               float* __restrict__ out) {
  const int row = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  __shared__ float s_r[4][12];
  if (x[row] != MASK_ID) {
    if (t < STATS_W) out[(int64_t)row * STATS_W + t] = 0.f;
    return;
  }
  const float* za = a + (int64_t)row * lda;
  const float* zb = b + (int64_t)row * ldb;
  float me = 0.f, se = 0.f, md = 0.f, sd = 0.f, emax = -3.402823466e38f, emin = 3.402823466e38f;
  float ma = -3.402823466e38f, mb = -3.402823466e38f;
  for (int v = t; v < V; v += NT) {
    if (!all_columns && v == MASK_ID) continue;
    const float e = za[v] - zb[v];
    me = fmaxf(me, fabsf(e));
    se += e * e;
    emax = fmaxf(emax, e);
    emin = fminf(emin, e);
    ma = fmaxf(ma, za[v]);
    mb = fmaxf(mb, zb[v]);
    const int w = v + 1;
    if (w < V && (all_columns || w != MASK_ID)) {
      const float d = e - (za[w] - zb[w]);
      md = fmaxf(md, fabsf(d));
      sd += d * d;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    me = fmaxf(me, __shfl_xor(me, off, 64));
    md = fmaxf(md, __shfl_xor(md, off, 64));
    emax = fmaxf(emax, __shfl_xor(emax, off, 64));
    emin = fminf(emin, __shfl_xor(emin, off, 64));
    ma = fmaxf(ma, __shfl_xor(ma, off, 64));
    mb = fmaxf(mb, __shfl_xor(mb, off, 64));
    se += __shfl_xor(se, off, 64);
    sd += __shfl_xor(sd, off, 64);
  }
  if (lane == 0) {
    s_r[wave][0] = me; s_r[wave][1] = se; s_r[wave][2] = md; s_r[wave][3] = sd;
    s_r[wave][4] = emax; s_r[wave][5] = emin; s_r[wave][6] = ma; s_r[wave][7] = mb;
  }
  __syncthreads();
  ma = fmaxf(fmaxf(s_r[0][6], s_r[1][6]), fmaxf(s_r[2][6], s_r[3][6]));
  mb = fmaxf(fmaxf(s_r[0][7], s_r[1][7]), fmaxf(s_r[2][7], s_r[3][7]));
  // entropies of both rows: H = log S - A / S with S = sum exp(z - m), A = sum exp(z - m) (z - m)
  float Sa = 0.f, Aa = 0.f, Sb = 0.f, Ab = 0.f;
  for (int v = t; v < V; v += NT) {
    if (!all_columns && v == MASK_ID) continue;
    const float da = za[v] - ma, db = zb[v] - mb;
    const float ea = expf(da), eb = expf(db);
    Sa += ea; Aa += ea * da;
    Sb += eb; Ab += eb * db;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Sa += __shfl_xor(Sa, off, 64);
    Aa += __shfl_xor(Aa, off, 64);
    Sb += __shfl_xor(Sb, off, 64);
    Ab += __shfl_xor(Ab, off, 64);
  }
  if (lane == 0) {
    s_r[wave][8] = Sa; s_r[wave][9] = Aa; s_r[wave][10] = Sb; s_r[wave][11] = Ab;
  }
  __syncthreads();
  if (t == 0) {
    float* o = out + (int64_t)row * STATS_W;
    o[0] = fmaxf(fmaxf(s_r[0][0], s_r[1][0]), fmaxf(s_r[2][0], s_r[3][0]));
    o[1] = (s_r[0][1] + s_r[1][1]) + (s_r[2][1] + s_r[3][1]);
    o[2] = fmaxf(fmaxf(s_r[0][2], s_r[1][2]), fmaxf(s_r[2][2], s_r[3][2]));
    o[3] = (s_r[0][3] + s_r[1][3]) + (s_r[2][3] + s_r[3][3]);
    o[4] = fmaxf(fmaxf(s_r[0][4], s_r[1][4]), fmaxf(s_r[2][4], s_r[3][4])) -
           fminf(fminf(s_r[0][5], s_r[1][5]), fminf(s_r[2][5], s_r[3][5]));
    const float SA = (s_r[0][8] + s_r[1][8]) + (s_r[2][8] + s_r[3][8]), AA = (s_r[0][9] + s_r[1][9]) + (s_r[2][9] + s_r[3][9]);
    const float SB = (s_r[0][10] + s_r[1][10]) + (s_r[2][10] + s_r[3][10]), AB = (s_r[0][11] + s_r[1][11]) + (s_r[2][11] + s_r[3][11]);
    const float Ha = logf(SA) - AA / SA, Hb = logf(SB) - AB / SB;
    o[5] = Ha - Hb;
    o[6] = Hb;
    o[7] = 0.f;
  }
}

hipError_t launch_logit_error_stats(const float* a, int lda, const float* b, int ldb, const int64_t* x, int rows, int V, int all_columns,
                                    float* out, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(logit_error_stats_kernel, dim3(rows), dim3(NT), 0, stream, a, lda, b, ldb, x, V, all_columns, out);
  return hipGetLastError();
}

// flag[0] &= (every row of seq and of x equals row 0): one thread per element of rows 1..B-1.
__global__ void rows_identical_kernel(const int64_t* __restrict__ seq, const int64_t* __restrict__ x, int64_t n, int L,
                                      int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int l = (int)(i % L);
  if (seq[L + i] != seq[l] || x[L + i] != x[l]) flag[0] = 0;
}

// has[b] = 1 if sample b still holds a MASK token, else 0: one wave per sample
__global__ __launch_bounds__(64) void samples_with_mask_kernel(const int64_t* __restrict__ x, int L, int32_t* __restrict__ has) {
  const int b = blockIdx.x;
  int any = 0;
  for (int l = threadIdx.x; l < L; l += 64) any |= x[(int64_t)b * L + l] == MASK_ID;
  any = __any(any);
  if (threadIdx.x == 0) has[b] = any ? 1 : 0;
}

hipError_t launch_samples_with_mask(const int64_t* x, int B, int L, int32_t* has, hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(samples_with_mask_kernel, dim3(B), dim3(64), 0, stream, x, L, has);
  return hipGetLastError();
}

// dst[i, :] = src[idx[i], :] (gather = 1) or dst[idx[i], :] = src[i, :] (gather = 0), rows of L int64; n rows
__global__ void move_token_rows_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ dst, const int32_t* __restrict__ idx,
                                       int n, int L, int gather) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * L) return;
  const int r = (int)(i / L), l = (int)(i - (int64_t)r * L);
  const int64_t sidx = gather ? (int64_t)idx[r] * L + l : i, didx = gather ? i : (int64_t)idx[r] * L + l;
  dst[didx] = src[sidx];
}

hipError_t launch_move_token_rows(const int64_t* src, int64_t* dst, const int32_t* idx, int n, int L, int gather, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int64_t tot = (int64_t)n * L;
  hipLaunchKernelGGL(move_token_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, src, dst, idx, n, L, gather);
  return hipGetLastError();
}

hipError_t launch_rows_identical(const int64_t* seq, const int64_t* x, int B, int L, int32_t* flag, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(flag, 0xff, sizeof(int32_t), stream);   // all ones = true
  if (e != hipSuccess || B <= 1) return e;
  const int64_t n = (int64_t)(B - 1) * L;
  hipLaunchKernelGGL(rows_identical_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, seq, x, n, L, flag);
  return hipGetLastError();
}

}  // namespace ed
