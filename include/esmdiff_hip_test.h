/*
 * esmdiff_hip_test.h — entry points of libesmdiff_hip.so that are NOT part of the drop-in surface (esmdiff_hip.h): one kernel
 * at a time for the parity tests (tests/test_gpu_*.py compare each against a float32 / float64 statement of the same op), the
 * per-section device-time profiler bench.py's roofline leg reads, and the measurement aids of -DED_DEBUG builds.  No call site
 * of the reference binds any of them; same conventions as esmdiff_hip.h (device pointers owned by the caller, int status,
 * caller's stream).
 */
#ifndef ESMDIFF_HIP_TEST_H
#define ESMDIFF_HIP_TEST_H

#include "esmdiff_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* C[M,N] (+)= A[M,K] · W[N,K]^T, bf16 in, f32 accumulate.  epilogue: see esmdiff_gemm_epilogue. */
typedef enum {
  ESMDIFF_EPI_BF16 = 0,       /* out bf16 [M,N]                                   */
  ESMDIFF_EPI_RESID_F32 = 1,  /* out f32 [M,N] += acc * alpha   (residual stream)   */
  ESMDIFF_EPI_SWIGLU_BF16 = 2,/* W rows interleaved gate/up in blocks of 32; out bf16 [M,N/2] */
  ESMDIFF_EPI_BIAS_GELU_BF16 = 3, /* out bf16 = gelu(acc + bias[N])                 */
  ESMDIFF_EPI_BIAS_F32 = 4    /* out f32 [M,ldc] = acc + bias, columns >= n_valid skipped */
} esmdiff_gemm_epilogue;

int esmdiff_gemm_bf16(const void* A, const void* W, void* out, const float* bias, int32_t M, int32_t N,
                      int32_t K, int32_t ldc, int32_t n_valid, float alpha, int32_t epilogue, void* stream);

/* The same kernels in their f16 build (csrc/ed_half.h, namespace ed16; what precision = ESMDIFF_PRECISION_F16 engines run): A, W and
 * the 16-bit outputs are IEEE half instead of bfloat16, conversions saturate at +-65504; everything else as esmdiff_gemm_bf16. */
int esmdiff_gemm_f16(const void* A, const void* W, void* out, const float* bias, int32_t M, int32_t N,
                     int32_t K, int32_t ldc, int32_t n_valid, float alpha, int32_t epilogue, void* stream);

/* The strict path's linear (csrc/strict.hip): out f32 [M,ldc] = epi(A f32 [M,K] (row stride lda) . W f32 [N,K]^T), every
 * product and sum in float32 on v_mfma_f32_32x32x2_f32 (fixed, batch-independent K order per output element; not ascending k);
 * K % 32 == 0; columns >= n_valid are not written. */
typedef enum {
  ESMDIFF_F32EPI_STORE = 0,      /* out = acc (+ bias[N] when bias != NULL)                        */
  ESMDIFF_F32EPI_BIAS_GELU = 1,  /* out = gelu(acc + bias), exact (erf) GELU                        */
  ESMDIFF_F32EPI_RESID_DIV = 2   /* out = out + acc / div  (x + branch / scaling_factor, in place)  */
} esmdiff_gemm_f32_epilogue;
int esmdiff_gemm_f32(const float* A, int32_t lda, const float* W, float* out, const float* bias, int32_t M, int32_t N,
                     int32_t K, int32_t ldc, int32_t n_valid, float div, int32_t epilogue, void* stream);

/* The F32_SPLIT path's linear and its operand preparation (csrc/gemm_split.hip, csrc/gemm256w4.hip SPLIT = 1).
 *   esmdiff_split_rows    src f32 [M,K] (row stride ld) -> a3 f16 [M,3K] = [hi | lo | hi] of src * 2^k(row), rs[M] = 2^-k(row)
 *   esmdiff_split_weight  src f32 [N,K] -> w3 f16 [N_pad,3K] = [lo | hi | hi] of src * 2^k (rows N..N_pad-1 zero-filled,
 *                         N_pad a multiple of 256 >= N), *inv_scale_out [host] = 2^-k; synchronous
 *   esmdiff_gemm_split    out f32 [M,ldc] = epi(rs[m] * w_inv_scale * A . W^T [+ bias[N]]); N (= N_pad) % 256 == 0,
 *                         K % 128 == 0, ldc >= N: whole padded rows are written (the kernel carries no column bound);
 *                         epilogue ESMDIFF_F32EPI_STORE (bias optional) or ESMDIFF_F32EPI_RESID_DIV. */
int esmdiff_split_rows(const float* src, int32_t ld, void* a2, float* rs, int32_t M, int32_t K, void* stream);
int esmdiff_split_weight(const float* src, void* w2, int32_t N, int32_t N_pad, int32_t K, float* inv_scale_out);
int esmdiff_gemm_split(const void* a2, const float* rs, const void* w2, float w_inv_scale, float* out, const float* bias,
                       int32_t M, int32_t N, int32_t K, int32_t ldc, float div, int32_t epilogue, void* stream);

/* The same GEMM as the engine issues it: with the engine's split-K workspace, so the small-M path (M < 1152 rows,
 * K >= 2048: FFN-down) runs as K slices + a fixed-order reduce kernel (csrc/gemm.hip).  Not re-entrant per engine. */
int esmdiff_gemm_bf16_ws(esmdiff_engine* eng, const void* A, const void* W, void* out, const float* bias, int32_t M,
                         int32_t N, int32_t K, int32_t ldc, int32_t n_valid, float alpha, int32_t epilogue,
                         void* stream);

/* The small-batch form of a residual branch (M < 1152 rows; csrc/engine.hip::forward): the branch linear A[M,K] W[N,K]^T
 * is left as S raw f32 K-slice planes in the engine's workspace (S = *splits_out, a function of N and K only) and the
 * LayerNorm kernel that follows sums them:  x[M,N] f32 += alpha * (A W^T);  y bf16 [M,N] = LayerNorm(x) * w (+ b).
 * N = d_model of the engine's shapes (N % 128 == 0, N <= 2048), K % 64 == 0.  Not re-entrant per engine. */
int esmdiff_branch_linear_layernorm(esmdiff_engine* eng, const void* A, const void* W, float* x, float alpha,
                                    const float* w, const float* b, void* y, int32_t M, int32_t N, int32_t K,
                                    int32_t* splits_out, void* stream);

/* y bf16 [M,D] = LayerNorm(x f32 [M,D]) * w (+ b); b may be NULL.  eps = 1e-5. */
int esmdiff_layernorm_bf16(const float* x, const float* w, const float* b, void* y, int32_t M, int32_t D,
                           void* stream);

/* Attention over pre-processed heads: qkv bf16 [B*L, 3*D] (GEMM output) -> ctx bf16 [B*L, D].
 * Applies the full-width q/k LayerNorm (weights f32 [D]), rotary, 1/sqrt(64) scaling, non-causal softmax. */
int esmdiff_attention_bf16(esmdiff_engine* eng, const void* qkv, const float* q_ln_w, const float* k_ln_w,
                           void* ctx, int32_t B, int32_t L, void* stream);

/* Accumulated per-section device time of the esmdiff_forward_logits/ddpm_sample calls since profiling was
 * enabled: esmdiff_set_profiling(eng, 1) brackets every launch with HIP events on the launch stream (no sync);
 * mode 2 brackets only the dominant kernel (FFN-up GEMM, section 6), cheap enough for a timed region; 0 = off. sections: 0 embed, 1 layernorm, 2 gemm_qkv,
 * 3 qk_norm_rope, 4 attention, 5 gemm_out, 6 gemm_ffn_up, 7 gemm_ffn_down, 8 head, 9 sampler.
 * ms_out: [16] floats, launches_out: [16] ints [host].  Synchronises the device. */
int esmdiff_set_profiling(esmdiff_engine* eng, int32_t on);
int esmdiff_get_profile(esmdiff_engine* eng, float* ms_out, int32_t* launches_out);

#ifdef ED_DEBUG
/* ---- measurement aids: exported only by libraries built with -DED_DEBUG (ESMDIFF_EXTRA_CXXFLAGS=-DED_DEBUG python -m
 * esmdiff_amd.build); the product library carries neither these nor the ESMDIFF_DEBUG_SKIP launch-skipping switch, and
 * esmdiff_engine_create FAILS when that variable is set.  Used by scratch/ A/B scripts only. ---- */
/* Measurement aid: one forward at (B, L) `n` times as plain launches and as `n` replays of one captured hipGraph of the
 * same launches (engine-owned stream); milliseconds per forward of each [host]. */
int esmdiff_debug_graph_ab(esmdiff_engine* eng, const int64_t* seq, const int64_t* x, int32_t B, int32_t L, int32_t n,
                           float* ms_direct, float* ms_graph);

/* Wall-clock helper for the bench's roofline leg: runs the GEMM `iters` times on `stream` bracketed by
 * HIP events on that stream and returns the average milliseconds per launch in *ms_out [host]. */
int esmdiff_gemm_bf16_timed(const void* A, const void* W, void* out, const float* bias, int32_t M,
                            int32_t N, int32_t K, int32_t ldc, int32_t n_valid, float alpha,
                            int32_t epilogue, int32_t iters, float* ms_out, void* stream);
#endif /* ED_DEBUG */

#ifdef __cplusplus
}
#endif
#endif /* ESMDIFF_HIP_TEST_H */
