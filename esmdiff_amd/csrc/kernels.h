// kernels.h — internal launch functions of libesmdiff_hip.so (gfx950 only).
// Every function enqueues on `stream` and returns the hipError_t of the launch.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/esmdiff_hip.h"

#include <map>
#include <mutex>
#include <utility>

typedef uint16_t bf16_t;  // raw bfloat16 bits

namespace ed {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: once per (kernel, device), thread-safe,
// return code checked (a `static bool once` left a second device's launches without the attribute).
inline hipError_t ensure_dynamic_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;
  int dev = 0;
  hipError_t s = hipGetDevice(&dev);
  if (s != hipSuccess) return s;
  std::lock_guard<std::mutex> g(mu);
  auto it = done.find({fn, dev});
  if (it != done.end() && it->second >= bytes) return hipSuccess;
  s = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (s == hipSuccess) done[{fn, dev}] = bytes;
  return s;
}

// ---- sampler.hip -----------------------------------------------------------------------------
hipError_t launch_ddpm_step(int64_t* x, const float* logits, int ld, int V, float mc_t, float mc_s, int final_,
                            const float* u, int use_philox, uint64_t seed, uint64_t sample_offset, int step,
                            int B, int L, hipStream_t stream, int logits_period = 0);
// has[b] = 1 if sample b of x [B, L] still holds a MASK token
hipError_t launch_samples_with_mask(const int64_t* x, int B, int L, int32_t* has, hipStream_t stream);
// gather (dst[i] = src[idx[i]]) or scatter (dst[idx[i]] = src[i]) of n token rows of L int64
hipError_t launch_move_token_rows(const int64_t* src, int64_t* dst, const int32_t* idx, int n, int L, int gather, hipStream_t stream);
// flag[0] (device int32) = -1 if every row of seq and x [B, L] equals row 0, else 0
hipError_t launch_rows_identical(const int64_t* seq, const int64_t* x, int B, int L, int32_t* flag, hipStream_t stream);

// ---- gibbs.hip ---------------------------------------------------------------------------------
// one entropy-ordered unmasking step: per masked row nucleus(top_p) + temperature draw + entropy, then per prompt
// the n_unmask[b] lowest-entropy masked positions take their token.  u: [B,L,4096] explicit uniforms or null.
hipError_t launch_gibbs_step(int64_t* x, const int64_t* seq, const float* logits, int ld, int vocab, float temperature,
                             float top_p, const int32_t* n_unmask, const float* u, int use_philox, uint64_t seed,
                             uint64_t sample_offset, int step, int32_t* sampled, float* entropy, int B, int L,
                             hipStream_t stream, int logits_period = 0, int strategy = 0, const uint32_t* inv_mask = nullptr);

// ---- gemm.hip --------------------------------------------------------------------------------
// out = epilogue(A[M,K] · W[N,K]^T); K % 64 == 0, N % 128 == 0 (weights are padded at load time).
// ws (optional): split-K workspace for the small-M path — room for f32 partial tiles, owned by ONE launch queue
// (concurrent launches need their own).  Without it every tile runs its whole K range.
struct GemmWorkspace {
  float* partial;
  size_t partial_floats;
};
hipError_t launch_gemm_bf16(const bf16_t* A, const bf16_t* W, void* out, const float* bias, int M, int N,
                            int K, int ldc, int n_valid, float alpha, int epilogue, hipStream_t stream,
                            const GemmWorkspace* ws = nullptr);

// Row count below which a (sub-)batch takes the small-batch path.  ESMDIFF_SMALL_MAX_ROWS overrides (A/B runs).
inline int small_max_rows() {
  static const int v = [] {
    const char* e = getenv("ESMDIFF_SMALL_MAX_ROWS");
    const int x = e ? atoi(e) : 1152;  // r02: 1 024 -> 1 152 (+4 % at 1 032 - 1 080 rows; 2 048 loses 5 - 20 % from 1 440 rows)
    return x < 128 ? 128 : (x > 8192 ? 8192 : x);
  }();
  return v;
}

// Small-batch path (M < small_max_rows()): the product as S raw f32 K-slice planes in `ws` (plane stride `stride` floats, row stride
// N), consumed by launch_add_partials_layernorm_bf16.  S = gemm_partial_splits(N, K), a function of the shape only.
struct GemmPartials {
  const float* p;
  int S;
  int64_t stride;
};
int gemm_partial_splits(int N, int K);
hipError_t launch_gemm_partials(const bf16_t* A, const bf16_t* W, const GemmWorkspace* ws, int M, int N, int K,
                                hipStream_t stream, GemmPartials* res);

// gemm256.hip: 256x256x64 tiles, 8 waves, counted-vmcnt pipeline; needs N % 256 == 0 (large-M path)
hipError_t launch_gemm256_bf16(const bf16_t* A, const bf16_t* W, void* out, const float* bias, int M, int N,
                               int K, int ldc, float alpha, int epilogue, hipStream_t stream);

// gemm256w4.hip: the same tile with 4 waves x 128x128 wave tiles, hand-placed main loop, accumulators in AGPRs; K % 128 == 0
hipError_t launch_gemm256w4_bf16(const bf16_t* A, const bf16_t* W, void* out, const float* bias, int M, int N,
                                 int K, int ldc, float alpha, int epilogue, hipStream_t stream);

// ---- norm.hip --------------------------------------------------------------------------------
hipError_t launch_layernorm_bf16(const float* x, const float* w, const float* b, bf16_t* y, int M, int D,
                                 hipStream_t stream);
// v = (x + delta) + delta2 (bf16 or null each); x = v if write_x; y = LayerNorm(v) * w (+ b)
hipError_t launch_add_layernorm_bf16(float* x, const bf16_t* delta, const bf16_t* delta2, int write_x, const float* w,
                                     const float* b, bf16_t* y, int M, int D, hipStream_t stream);
// x += alpha * (P[0] + P[1] + ... + P[S-1]) (f32 K-slice planes of a branch linear); x written back;
// y = LayerNorm(x) * w (+ b)
hipError_t launch_add_partials_layernorm_bf16(float* x, const GemmPartials& P, int N, float alpha, const float* w,
                                              const float* b, bf16_t* y, int M, int D, hipStream_t stream);
hipError_t launch_layernorm_bf16_in(const bf16_t* x, const float* w, const float* b, bf16_t* y, int M, int D,
                                    hipStream_t stream);
// qkv bf16 [M,3D] -> q,k bf16 [M,D] token-major (LayerNorm over D, rotary, q pre-scaled); v stays in qkv
hipError_t launch_qk_norm_rope(const bf16_t* qkv, const float* q_ln_w, const float* k_ln_w,
                               const float* rope_cos, const float* rope_sin, bf16_t* q, bf16_t* k,
                               int B, int L, int H, hipStream_t stream);

// ---- attention.hip ---------------------------------------------------------------------------
// q,k [B*L, H*64] token-major, v read in place from qkv [B*L, 3*H*64] (columns 2*H*64 ..) -> ctx bf16 [B*L, H*64]
hipError_t launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* qkv, bf16_t* ctx, int B, int L,
                            int H, hipStream_t stream);

// ---- geom.hip --------------------------------------------------------------------------------
// P bf16 [B*L, 15*VH] (proj output) + frames -> out bf16 [B*L, 3*VH]; w_rot / w_dist = softplus(scale) per head
hipError_t launch_geom_attention(const bf16_t* P, const float* rot, const float* trans, const uint8_t* fmask,
                                 const float* w_rot, const float* w_dist, bf16_t* out, int B, int L, int VH,
                                 hipStream_t stream);

hipError_t launch_geom_attention_f32(const float* P, const float* rot, const float* trans, const uint8_t* fmask,
                                     const float* w_rot, const float* w_dist, float* out, int B, int L, int VH,
                                     hipStream_t stream);

// ---- embed.hip -------------------------------------------------------------------------------
// x[b,l,:] = E_seq[seq] + E_struct[struct'] + c + cond   (net.py:445-466)
hipError_t launch_embed(const int64_t* seq, const int64_t* xtok, const float* e_seq, const float* e_struct,
                        const float* cvec, const float* cond, float* out, int B, int L, int D,
                        hipStream_t stream);
// cond = W2 · silu(W1 · t_freq + b1) + b2   (net.py:489-492,519-522), f32
hipError_t launch_gather_rows(const int64_t* tok, const float* table, float* out, int M, int D, int n_rows,
                              hipStream_t stream);
// v f32 [M, ld] (23 used) -> backbone N/CA/C coordinates f32 [M, 3, 3]
hipError_t launch_dim6_to_backbone(const float* v, int ld, float* out, int M, float trans_scale, hipStream_t stream);
hipError_t launch_delay_us(int us, hipStream_t stream);
// pairwise.hip: the decoder's pairwise confidence head (pair features; PAE / pTM from the 64 PAE bin logits)
hipError_t launch_pair_features(const bf16_t* qk, bf16_t* X, int nb, int L, hipStream_t stream);
hipError_t launch_pair_features_f32(const float* qk, float* X, int nb, int L, hipStream_t stream);
hipError_t launch_pae_tm(const float* logits, const int64_t* tokens, float* tm_rows, float* pae, float* ptm, int nb, int L,
                         float max_bin, hipStream_t stream);
hipError_t launch_plddt_mean(const float* v, int ld, int n_bins, float* out, int M, hipStream_t stream);
hipError_t launch_sigma_mlp(const float* t_freq, const float* w1, const float* b1, const float* w2,
                            const float* b2, float* hidden, float* cond, int F, int D, hipStream_t stream);

// ---- strict.hip: the float32 precision path (esmdiff_config.precision = ESMDIFF_PRECISION_F32) -----------------
hipError_t launch_gemm_f32(const float* A, int lda, const float* W, float* out, const float* bias, int M, int n_rows, int K,
                           int ldc, int n_valid, float div, int epi, hipStream_t stream);
hipError_t launch_layernorm_f32(const float* x, const float* w, const float* b, float* y, int M, int D, hipStream_t stream);
hipError_t launch_swiglu_f32(const float* gu, float* mid, int M, int FH, hipStream_t stream);
hipError_t launch_qk_norm_rope_f32(const float* qkv, const float* q_ln_w, const float* k_ln_w, const float* rope_cos,
                                   const float* rope_sin, float* q, float* k, int B, int L, int H, hipStream_t stream);
hipError_t launch_attention_f32(const float* q, const float* k, const float* qkv, float* ctx, int B, int L, int H,
                                hipStream_t stream);

// ---- gemm_split.hip + gemm256w4.hip (SPLIT): float32-grade linears as three f16 MFMA passes over split operands ----
// A3 f16 [M, 3K] = [hi | lo | hi] scaled per row (rs[M] = 1 / row scale), W3 f16 [N_pad, 3K] = [lo | hi | hi] scaled per
// matrix (w_inv_scale = 1 / scale); out f32 [M, ldc] = epi(rs[m] * w_inv_scale * A . W^T); N % 256 == 0, K % 128 == 0,
// ldc >= N (no column bound in the kernel); epi: ESMDIFF_F32EPI_STORE (+ bias[N] when non-null) or ESMDIFF_F32EPI_RESID_DIV.
hipError_t launch_gemm256w4_split(const uint16_t* A2, const float* rs, const uint16_t* W2, float w_inv_scale, float* out,
                                  const float* bias, int M, int N, int K, int ldc, float div, int epi, hipStream_t stream);
hipError_t launch_split_rows(const float* src, int ld, uint16_t* dst, float* rs, int M, int K, hipStream_t stream);
// LayerNorm (of gelu(x) when gelu_in; of x + delta when delta, a bf16 [M, D] branch output, is non-null) in the strict path's
// arithmetic -> split row (+ the f32 row into y32 when non-null)
hipError_t launch_layernorm_split(const float* x, const float* w, const float* b, uint16_t* dst, float* rs, float* y32,
                                  int M, int D, int gelu_in, hipStream_t stream, const uint16_t* delta = nullptr);
hipError_t launch_swiglu_split(const float* gu, uint16_t* dst, float* rs, int M, int FH, hipStream_t stream);
// create time, synchronous: dst [rows_pad, 3K] (caller zero-fills the padding rows); scratch_bits: 4 device bytes
hipError_t split_weight(const void* src, int src_dtype, uint16_t* dst, int64_t rows, int K, uint32_t* scratch_bits,
                        float* inv_scale_out);

// ---- convert.hip (weight preparation at engine create) ---------------------------------------
hipError_t launch_to_bf16(const void* src, int src_dtype, bf16_t* dst, int64_t n, hipStream_t stream);
hipError_t launch_to_f32(const void* src, int src_dtype, float* dst, int64_t n, hipStream_t stream);
// dst rows: blocks of 32: [gate 32t..32t+31 | up 32t..32t+31]; src [2H, K]: gate rows 0..H-1, up rows H..2H-1
hipError_t launch_interleave_swiglu(const void* src, int src_dtype, bf16_t* dst, int H, int K,
                                    hipStream_t stream);

}  // namespace ed
