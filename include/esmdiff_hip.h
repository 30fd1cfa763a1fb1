/*
 * esmdiff_hip.h — C ABI of libesmdiff_hip.so, the MI355X (gfx950) engine for the
 * ESMDiff sampling hot path.
 *
 * The reference (lujiarui/esmdiff) has no FFI: its hot path is a chain of Python
 * call sites.  Each entry point below is what a binding for ONE of those call
 * sites would bind; the call site it replaces is cited as file:line relative to
 * the reference tree.  INTEGRATION.md shows the ctypes stubs a maintainer of the
 * reference would add.
 *
 * Conventions
 *   ownership  every pointer passed in/out is DEVICE memory owned by the caller
 *              (e.g. a PyTorch-ROCm tensor's data_ptr()) unless marked [host].
 *              The engine owns only its bf16 weight copies and its workspace,
 *              both sized at create time for (max_batch, max_len).
 *   errors     every entry returns int: 0 = ok, <0 = esmdiff_status.  Nothing
 *              throws across the boundary.  esmdiff_last_error() gives the text.
 *   threading  one engine per (process, device); not thread-safe.  All work is
 *              enqueued on the caller's hipStream_t (passed as void*); no entry
 *              synchronises the stream except where stated.
 *   dtypes     token ids int64 (the reference's LongTensor), logits float32,
 *              schedule scalars float32 computed by the HOST exactly as
 *              model.py:564-567,584-595 does and passed in.
 */
#ifndef ESMDIFF_HIP_H
#define ESMDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESMDIFF_ABI_VERSION 8   /* 8: + esmdiff_gibbs_step_rows; esmdiff_logit_error_stats writes 8 floats per row and takes all_columns; 7: + esmdiff_ddpm_step_rows, esmdiff_logit_error_stats (additions only); 6: + esmdiff_ddpm_step_margin, esmdiff_forward_logits_sigmas, esmdiff_set_small_batch_splitk */

/* structure-track vocabulary: esm constants mirrored at model.py:380-381 */
#define ESMDIFF_VOCAB 4101
#define ESMDIFF_MASK_ID 4096
#define ESMDIFF_STRUCT_EOS 4097
#define ESMDIFF_STRUCT_BOS 4098
#define ESMDIFF_STRUCT_PAD 4099
#define ESMDIFF_STRUCT_CHAINBREAK 4100

typedef enum {
  ESMDIFF_OK = 0,
  ESMDIFF_E_INVALID = -1,   /* bad argument (the reference: assert / ValueError)   */
  ESMDIFF_E_MISSING = -2,   /* weight missing from the table (load_state_dict strict) */
  ESMDIFF_E_SHAPE = -3,     /* weight has the wrong shape / dtype                   */
  ESMDIFF_E_HIP = -4,       /* HIP runtime error                                    */
  ESMDIFF_E_CAPACITY = -5,  /* B > max_batch or L > max_len                         */
  ESMDIFF_E_NODEVICE = -6   /* no gfx950 device                                     */
} esmdiff_status;

typedef enum { ESMDIFF_F32 = 0, ESMDIFF_BF16 = 1 } esmdiff_dtype;

/* Arithmetic of the network (esmdiff_config.precision).
 *   BF16  the throughput path: bf16 weights and GEMM operands on the bf16 MFMA, f32 accumulation, f32 residual stream.
 *   F32   the "strict" path (csrc/strict.hip): float32 weights and activations end to end, every linear on the f32-input
 *         MFMA (one f32 fmaf chain per output element, in a fixed batch-independent K order), correctly rounded divide / sqrt — the arithmetic the reference itself runs in
 *         (checkpoint_utils.py:59-73 loads float32; decode at sample_esmdiff.py:40-61), ~1/12 of the bf16 throughput.
 *         It is what north_star's floating-point bars are stated against (ids equal under a fixed seed, decoded backbone
 *         within 1e-4 A) and the structure decoder's default on the Python side.  A row's result does not depend on the
 *         batch it is computed in.  Coordinate conditioning (esmdiff_set_frames) runs in float32 too.
 *   F32_SPLIT (ABI 5; csrc/gemm_split.hip)  the F32 path with every large linear computed as THREE passes of the f16 MFMA
 *         over operands split into two f16 numbers each (x = hi + lo to 2^-22, rows scaled by powers of two so that nothing
 *         can overflow; f16 x f16 products are exact in f32; the dropped lo.lo term is 2^-22): float32-grade products with
 *         f32 accumulation at ~1/3 of the bf16 MFMA rate instead of 1/16.  Attention runs the same way (attention_split.hip).
 *         LayerNorm, rotary and GELU are the F32 path's arithmetic; the softmax exponentials and the SwiGLU (fused into the
 *         FFN-up epilogue) use the hardware's exp2 / reciprocal (1 ulp each) where F32 calls expf and divides.  Logits agree
 *         with the F32 engine to 7e-6 and the id chains are equal at BASELINE configs[1]'s full size; F32 stays the referee.  A
 *         row's result does not depend on the batch it is computed in.
 *   F16 (ABI 5)  the BF16 path's kernels and launch sequence with IEEE half operands instead of bfloat16 (the same sources
 *         compiled a second time, csrc/ed_half.h): the f16 forms of the same MFMA instructions run at the same rate on the
 *         same bytes, and an 11-bit significand instead of 8 cuts every operand rounding — and with it the logit error and
 *         the rate of near-tie id flips against the float32 chain — by 8.  f16's narrower exponent is not an issue for this
 *         network (weights, LayerNorm / attention / SwiGLU outputs are O(1e-3 .. 1e2)); conversions saturate at +-65504. */
typedef enum { ESMDIFF_PRECISION_BF16 = 0, ESMDIFF_PRECISION_F32 = 1, ESMDIFF_PRECISION_F32_SPLIT = 2, ESMDIFF_PRECISION_F16 = 3 } esmdiff_precision;

/* Hyper-parameters of CustomizedESM3 (net.py:322-332) + TimestepEmbedder (net.py:487) +
 * StructureOutputHeads (net.py:299); values for ESM3-open: 1536 / 24 / 48 / 4096 / 4101 / 256. */
typedef struct {
  int32_t d_model;
  int32_t n_heads;      /* head dim must be 64 */
  int32_t n_layers;
  int32_t ffn_hidden;   /* SwiGLU hidden (gate+up is 2x this) */
  int32_t vocab_out;    /* n_structure_heads: 4101 (ESMDiff) or 4096 (stock ESM3) */
  int32_t freq_dim;     /* TimestepEmbedder.frequency_embedding_size */
  int32_t max_batch;
  int32_t max_len;      /* tokens incl. BOS/EOS */
  float residue_scale;  /* sqrt(n_layers/36) — esm TransformerStack */
  int32_t time_conditioning; /* mdlm.yaml:41 */
  int32_t precision;    /* one of the esmdiff_precision values; new in ABI 4 */
  int32_t head_precision; /* ABI 5.  0: the head runs in `precision`.  1 (with precision = BF16): the final LayerNorm and the
                           * output head (net.py:298-308: Linear, GELU, LayerNorm, Linear) run in float32 grade on the
                           * F32_SPLIT kernels from the f32 residual stream — the logits' own rounding (64 % of the bf16
                           * engine's logit-error variance, profiles/r04_head_decomposition.json) goes away for +1 % time */
} esmdiff_config;

/* One state-dict entry.  `name` uses the reference's key layout for the ESMDiff
 * checkpoint ('module' dict, checkpoint_utils.py:62-64): "net.transformer.blocks.0.attn.…",
 * "sigma_embedder.mlp.0.weight", …  `data` is a DEVICE pointer to a contiguous tensor. */
typedef struct {
  const char* name;
  const void* data;
  int32_t dtype;     /* esmdiff_dtype */
  int32_t ndim;
  int64_t shape[4];
} esmdiff_weight;

/* Counter-based noise source (performance mode).  Uniform for (sample, step, position l,
 * vocabulary id v) = Philox4x32-10(key = seed, counter = {v>>2, l, sample, step})[v&3] >> 8 · 2^-24,
 * so results do not depend on batch composition or on how samples are sharded over GPUs. */
typedef struct {
  uint64_t seed;
  uint64_t sample_offset; /* global index of batch row 0 */
} esmdiff_rng;

typedef struct esmdiff_engine esmdiff_engine;

int esmdiff_abi_version(void);

/* What this library is and how an engine will run a batch — for logs (the CLI prints both beside "Sampling token time").
 * The product library reads NO ESMDIFF_* environment variable (one exception: it refuses to create an engine while
 * ESMDIFF_DEBUG_SKIP is set); the tuning switches of the A/B experiments exist in -DED_DEBUG builds only, and
 * esmdiff_get_build_info says which kind this is.  Both write a NUL-terminated text into buf [host, cap bytes] and return
 * the length the full text needs (excluding the NUL), or < 0.
 *   esmdiff_get_build_info   "abi=7 arch=gfx950 debug_env=0 ..."
 *   esmdiff_describe_plan    the dispatch plan of a (B, L) forward on this engine: precision, head precision, number of
 *                            sub-batch streams and their sizes, regular / small-batch path, the GEMM kernel of each block
 *                            linear, the exact shortcuts that are on. */
int esmdiff_get_build_info(char* buf, int32_t cap);
int esmdiff_describe_plan(const esmdiff_engine* eng, int32_t B, int32_t L, char* buf, int32_t cap);

/* Explicit dispatch options (ABI 7) — what used to be environment switches and may legitimately be chosen by a caller.  Neither
 * changes a result bit: the dispatch PATH of a (B, L) forward (the small-batch K-slice kernels below 1 152 rows per sub-batch,
 * the regular kernels above: two summation orders) is what the DEFAULT options choose, a function of (B, L) alone; an option
 * only changes how many launch queues run the batch, and a value that would push the sub-batches across that row count is
 * reduced until they stay on the default path's side (r06; esmdiff_describe_plan shows the count in use).  On one path a row's
 * K order depends on (N, K) only.  tests/test_gpu_kernels.py::test_stream_counts_bit_identical (inside one path),
 * tests/test_gpu_fullwidth.py::test_stream_options_never_change_a_bit (settings that would cross the threshold).
 *   ESMDIFF_OPT_STREAMS           sub-batch launch queues of the 16-bit forward: 1 .. 4 (default 2)
 *   ESMDIFF_OPT_DUAL_MIN_TOKENS   batches of at least this many tokens are cut into sub-batches (default 2200) */
typedef enum { ESMDIFF_OPT_STREAMS = 1, ESMDIFF_OPT_DUAL_MIN_TOKENS = 2 } esmdiff_option;
int esmdiff_set_option(esmdiff_engine* eng, int32_t option, int64_t value);

/* Replaces load_state_dict_from_lightning_ckpt (checkpoint_utils.py:41-74) + hydra instantiate
 * of mdlm.yaml:26-58: builds bf16 device copies (SwiGLU rows interleaved, head padded) of the
 * weights in `table` [host array of n entries] on `device` and allocates the workspace.
 * Synchronous. */
int esmdiff_engine_create(const esmdiff_config* cfg, const esmdiff_weight* table, int32_t n,
                          int32_t device, esmdiff_engine** out);
void esmdiff_engine_destroy(esmdiff_engine* eng);

/* Text of the last error on `eng` (or of the last failed create when eng == NULL). [host] */
const char* esmdiff_last_error(const esmdiff_engine* eng);

/* Replaces self.net(structure_tokens=x, sequence_tokens=seq, auxiliary_embeddings=cond).structure_logits
 * (model.py:475-481 -> net.py:371-483) including conditions = sigma_embedder(sigma) (model.py:466-471,
 * net.py:519-522).  t_freq = TimestepEmbedder.timestep_embedding(sigma, freq_dim) [freq_dim] floats for
 * the single sigma all rows share (model.py:570-571), or NULL for no auxiliary embedding at all (esm's own forward,
 * the gibbs path).  The MLP runs whenever sigma_embedder.* weights were in the table: a model built with
 * time_conditioning = false still adds sigma_embedder(0) in the reference (_process_sigma, model.py:535-541), so its
 * caller passes the sinusoid of sigma = 0.
 * seq, x: [B,L] int64; seq ids outside 0..63 and structure ids outside -1..4100 are clamped (the Python host raises
 * before it gets here).  logits_out: [B,L,ld_logits] float32, ld_logits >= vocab_out. */
int esmdiff_forward_logits(esmdiff_engine* eng, const int64_t* seq, const int64_t* x,
                           const float* t_freq, float* logits_out, int32_t ld_logits,
                           int32_t B, int32_t L, void* stream);

/* The same forward with ONE SIGMA PER SAMPLE, as `_model_wrapper(x, sequence_tokens, sigma)` takes it (model.py:464-481: sigma is
 * a (B,) vector, conditions = sigma_embedder(sigma) is (B, d_model), tiled over each sample's rows): t_freq = the sinusoids
 * [B, freq_dim] (device).  The sampling loops always pass one sigma for the whole batch; this entry is for callers that mix
 * noise levels in one batch (a training-style evaluation of the denoiser, or samples at different updates in one forward). */
int esmdiff_forward_logits_sigmas(esmdiff_engine* eng, const int64_t* seq, const int64_t* x, const float* t_freq,
                                  float* logits_out, int32_t ld_logits, int32_t B, int32_t L, void* stream);

/* F32_SPLIT engines, off by default: forwards of at most 4096 rows run their two residual linears (out-proj, FFN-down: 6 column
 * tiles each, i.e. a few dozen of the 256 CUs at such sizes) as K slices in ONE launch (3 and 4 slices as extra row blocks of the
 * persistent GEMM) plus a pass that sums the slices in order and adds into the residual stream.  Float32 grade as before, but the
 * summation order over K differs from the unsliced kernel's, so with the option ON a sample's last bits depend on whether its
 * batch had more than 4096 rows; OFF (default) the engine stays batch-independent bit for bit.  Used by the certified sampler
 * for its re-runs (small batches by construction). */
int esmdiff_set_small_batch_splitk(esmdiff_engine* eng, int32_t on);

/* ESMOutput.embeddings of the forward that just ran (net.py:468-469, :312-320: the transformer stack's pre-norm hidden
 * state, the second value `self.transformer(...)` returns): out f32 [B,L,d_model].  (B, L) must be the last forward's. */
int esmdiff_get_embeddings(esmdiff_engine* eng, float* out, int32_t B, int32_t L, void* stream);

/* ESMOutput.sequence_logits of the forward that just ran, for networks built with a sequence head (net.py:299-311:
 * StructureOutputHeads(..., n_sequence_heads > 0), a RegressionHead on the same normalised hidden state as the structure head;
 * what _model_wrapper returns next to the structure logits when sequence_prediction is on, model.py:488-490).  The head is
 * created when the weight table holds output_heads.sequence_head.{0,2,3}.{weight,bias} (1 .. 128 outputs) and then runs with
 * every forward; out f32 [B, L, ld_out], ld_out >= n_sequence_heads.  ESMDIFF_E_MISSING without those weights.  (ABI 7) */
int esmdiff_get_sequence_logits(esmdiff_engine* eng, float* out, int32_t ld_out, int32_t B, int32_t L, void* stream);

/* Replaces logits_parameterization + the sampling half of _ddpm_update + _sample_categorical
 * (model.py:527-533, 602-607, 24-28): given RAW network logits, writes x' in place.
 *   final == 0: x' = where(x != MASK, x, argmax_v q_v / (1e-10 - log(u_v + 1e-10)))
 *   final != 0: noise-removal step (model.py:575-579): x' = argmax_v log_p_v
 * Noise: `u` = explicit uniforms [B,L,vocab] row-major (what torch.rand_like draws, model.py:27),
 * or u == NULL and `rng` != NULL for the Philox source at step index `step`. */
int esmdiff_ddpm_step(esmdiff_engine* eng, int64_t* x_inout, const float* logits, int32_t ld_logits,
                      float move_chance_t, float move_chance_s, int32_t final, const float* u,
                      const esmdiff_rng* rng, int32_t step, int32_t B, int32_t L, void* stream);

/* esmdiff_ddpm_step with the Philox source, plus a per-sample report of how close the draw was: sample_flags[b]
 * (device int32 [B], zeroed by the caller) is set to 1 when some masked row of sample b was decided by less than `margin`
 * — update (final == 0): winner <= margin * runner-up of q_v / g_v (margin = exp(2 eps) covers logits known to +-eps:
 * between two tokens the logsumexp cancels and z_a - z_b moves by at most 2 eps; against the mask column z_a moves by
 * eps and the logsumexp by eps); final pass: winner - runner-up <= margin in log-probability (margin = 2 eps).  An unflagged sample's ids are the ids ANY logits within eps of these
 * would have produced.  The ids written are those of esmdiff_ddpm_step, bit for bit.  No reference counterpart: it is
 * what lets a reduced-precision engine hand the few close calls to an f32-grade one (esmdiff_amd/certified.py). */
int esmdiff_ddpm_step_margin(esmdiff_engine* eng, int64_t* x_inout, const float* logits, int32_t ld_logits,
                             float move_chance_t, float move_chance_s, int32_t final, const esmdiff_rng* rng,
                             int32_t step, int32_t B, int32_t L, float margin, int32_t* sample_flags, void* stream);

/* esmdiff_ddpm_step_margin with ONE PARAMETER SET PER SAMPLE (ABI 7).  The reference's loop (model.py:570-573) moves the whole
 * batch through the same update, but nothing couples its samples (model.py:583-607 is row-wise and `_sample_categorical` draws
 * per element), so a batch may hold samples that sit at DIFFERENT updates of their chains: params[b] (DEVICE array of B entries)
 * gives sample b its global Philox sample index (what esmdiff_rng.sample_offset + b is in the plain entry), its move chances,
 * its Philox step index and whether its pass is the noise-removal one.  Per sample the ids written are those esmdiff_ddpm_step
 * would write for that sample alone with the same scalars, bit for bit (tests/test_gpu_kernels.py).  margin_ratio (>= 1) is the
 * bound of the updates, margin_diff (>= 0) of the final passes, as in esmdiff_ddpm_step_margin; sample_flags may be NULL (plain
 * draws).  sample_min_gap (f32 [B], optional, set to +inf by the caller): the smallest winner-over-runner-up gap among the
 * sample's masked rows in log units — log(winner / runner-up) of q_v / g_v for an update, the log-probability difference for a
 * final pass; a sample is flagged exactly when that gap is <= log(margin_ratio) (resp. margin_diff), up to float rounding of the
 * logarithm.  It is a statistic (the re-run share as a function of eps), not an input of the draw.
 * Used by esmdiff_amd/certified.py: verification batches of the f32-grade engine and the fast lane after a roll-back. */
typedef struct {
  uint64_t sample_index;   /* Philox key: the sample's GLOBAL index */
  float move_chance_t;     /* model.py:592-595; ignored when final != 0 */
  float move_chance_s;
  int32_t step;            /* Philox step index = index of the update in the sample's chain */
  int32_t final;           /* != 0: noise-removal pass (model.py:575-579) */
} esmdiff_sample_step;
int esmdiff_ddpm_step_rows(esmdiff_engine* eng, int64_t* x_inout, const float* logits, int32_t ld_logits,
                           const esmdiff_sample_step* params, uint64_t seed, int32_t B, int32_t L, float margin_ratio,
                           float margin_diff, int32_t* sample_flags, float* sample_min_gap, void* stream);

/* How far two engines' logits are apart on the same input (ABI 7, widened in ABI 8; no reference counterpart — the reference
 * has one precision).  a, b: f32 [rows, ld_a / ld_b] raw logits of the same token rows; x: int64 [rows] the tokens that went in.
 * For every row that is MASK (the only rows whose decisions read the logits), over the columns a decision reads — all v < vocab
 * but the MASK column for the ddpm draw (all_columns = 0: model.py:528 pushes that column to -1e6), every v < vocab for the gibbs
 * step (all_columns = 1: nucleus and entropy span the whole row): e_v = a_v - b_v; d_v = e_v - e_(v+1), the error of the logit
 * difference of two NEIGHBOURING tokens (a sample of the pair-error distribution); max e - min e, the bound on the error of the
 * difference of ANY two tokens of the row, which is what decides a draw between them; and the entropy H of softmax over the same
 * columns for both inputs.  out f32 [rows, 8] = { max |e|, sum e^2, max |d|, sum d^2, max e - min e, H(a) - H(b), H(b), 0 };
 * zeros for rows that are not MASK. */
int esmdiff_logit_error_stats(const float* a, int32_t ld_a, const float* b, int32_t ld_b, const int64_t* x, int32_t rows,
                              int32_t vocab, int32_t all_columns, float* out, void* stream);

/* Replaces MaskedDiffusionLanguageModeling.ddpm_sample (model.py:543-581) for one batch, entirely on
 * the device, Philox noise: T updates + the noise-removal pass.  x_inout holds the prior on entry
 * (all MASK, model.py:555, or input_prior, :561) and the sample on return.
 * mc_t, mc_s: [T] move chances per step [host]; t_freq: [(T+1), freq_dim] [host] (row T = noise removal). */
int esmdiff_ddpm_sample(esmdiff_engine* eng, const int64_t* seq, int64_t* x_inout, int32_t B, int32_t L,
                        int32_t T, const float* mc_t, const float* mc_s, const float* t_freq,
                        const esmdiff_rng* rng, void* stream);

/* Step-0 sharing (off by default; exact).  The reference's loop (model.py:570-573) runs the network on the whole batch at
 * every step; at step 0 of a run whose samples all start from the same tokens (the CLI repeats ONE sequence and an all-MASK
 * or identical prior, sample_esmdiff.py:196-209) every sample's inputs — and so its logits — are identical.  With the
 * option on, esmdiff_ddpm_sample / esmdiff_gibbs_sample verify ON THE DEVICE that all rows of seq and x_inout are equal,
 * run the first forward on the smallest sub-batch that takes the same dispatch path as the whole batch (bit-identical
 * logits: tests/test_gpu_fullwidth.py::test_logits_across_dispatch_paths) and let every sample draw from those logits with
 * its own noise: the ids are bit-identical to the unshared run (test_step0_sharing_is_exact) and 1 of the T + 1 forwards
 * shrinks to a fraction.  Costs one 4-byte read-back per sampling call.  Not used with coordinate conditioning.
 * esmdiff_get_counters: network forwards issued and token rows pushed through them since create / the last reset — the
 * work actually executed, for FLOP accounting (bench.py reports the shared run as a separate, labelled figure). */
int esmdiff_set_step0_sharing(esmdiff_engine* eng, int32_t on);
/* The sub-batch size step-0 sharing uses (ABI 8): the smallest number of samples n <= B whose (n, L) forward takes the same
 * dispatch path — and so produces the same logits bit for bit — as a (B, L) forward; B when there is none.  Returns n >= 1, or a
 * negative esmdiff_status.  Callers that drive the loop themselves (esmdiff_amd/certified.py) share their first forward with it. */
int esmdiff_shared_forward_batch(const esmdiff_engine* eng, int32_t B, int32_t L);
/* Exact skip of the noise-removal forward (off by default; exact).  The last of the T + 1 forwards of esmdiff_ddpm_sample only
 * serves `x = argmax(log p)` (model.py:575-579), and for a position that is no longer MASK the re-parameterised log p is 0 at its
 * own token and -1e6 elsewhere (model.py:530-532): a sample without a MASK comes back unchanged.  With the option on, the
 * engine counts on the device which samples still hold a MASK after update T (one B x 4-byte read-back), skips the forward
 * when none does, and otherwise runs it on those samples only — as a sub-batch on the same dispatch path as the whole batch
 * (padded with complete samples when smaller), so the ids are bit-identical to the full run
 * (tests/test_gpu_fullwidth.py::test_final_skip_is_exact).  P(a masked row stays masked at the last update) = mc_s / mc_t. */
int esmdiff_set_final_skip(esmdiff_engine* eng, int32_t on);
int esmdiff_get_counters(esmdiff_engine* eng, int64_t* forwards, int64_t* token_rows, int32_t reset);

/* Replaces, for ONE step, the per-prompt half of esm.utils.generation.iterative_sampling_raw as the reference calls
 * it in "gibbs" mode (sample_esmdiff.py:114-122; [ESM-RECALL]): for every still-masked position
 * entropy of softmax(logits over the 4096 codebook ids), nucleus filter (top_p), temperature, categorical draw; then
 * per prompt the n_unmask[b] lowest-entropy masked positions (never BOS/EOS/PAD of `seq`) take their token.
 * n_unmask: [B] int32 DEVICE; u: explicit uniforms [B,L,4096] or NULL with rng; temperature >= 0 (0: the arg-max
 * of the filtered logits, no noise drawn — esm's sample_logits [ESM-RECALL]). */
int esmdiff_gibbs_step(esmdiff_engine* eng, int64_t* x_inout, const int64_t* seq, const float* logits,
                       int32_t ld_logits, float temperature, float top_p, const int32_t* n_unmask, const float* u,
                       const esmdiff_rng* rng, int32_t step, int32_t B, int32_t L, void* stream);

/* GenerationConfig fields the reference leaves at their defaults (sample_esmdiff.py:116-119 sets only track / num_steps /
 * temperature / top_p) [ESM-RECALL]: strategy 0 = "entropy" (the k lowest-entropy masked positions are unmasked), 1 =
 * "random" (a uniformly random k-subset of the masked positions: the k smallest of one Philox uniform per position; needs the
 * rng noise source); invalid_ids [host, n_invalid]: codebook ids that are never drawn — masked after the nucleus cut exactly
 * like the special ids >= 4096.  Applies to every following esmdiff_gibbs_step / esmdiff_gibbs_sample of this engine;
 * (0, NULL, 0) restores the defaults.  Synchronous (a 512-byte upload). */
int esmdiff_set_gibbs_options(esmdiff_engine* eng, int32_t strategy, const int32_t* invalid_ids, int32_t n_invalid);

/* esmdiff_gibbs_step (Philox source) with ONE PARAMETER SET PER PROMPT (ABI 8), and optionally a report of how close the step's
 * decisions were.  iterative_sampling_raw moves a batch through the same step index, but nothing couples its prompts (every
 * decision above is per row or per prompt), so a batch may hold prompts that sit at DIFFERENT steps of their chains: params[b]
 * (DEVICE array of B entries) gives prompt b its global Philox sample index (what esmdiff_rng.sample_offset + b is in the plain
 * entry), its Philox step index and the number of positions it unmasks.  Per prompt the ids written are those esmdiff_gibbs_step
 * writes for that prompt alone with the same scalars, bit for bit (tests/test_gpu_kernels.py).
 * pair_bound < 0: plain draws (sample_flags / sample_gaps must be NULL).  pair_bound = R >= 0, entropy_bound = E >= 0: the logits
 * are taken to be known only up to an error whose difference between any two columns of a row is at most R (so every probability
 * up to the factor exp(+-R)) and which moves a row's entropy by at most E.  sample_flags[b] (int32 [B], zeroed by the caller) then
 * gets, over the rows prompt b UNMASKS in this step: bit 0 — a draw's winner does not lead the best other possibly-kept token by
 * more than exp(R / temperature) (temperature 0: R); bit 1 — the winner's membership of the nucleus, or a better token's, depends
 * on the error (kept for sure: mass{z_j >= z_v - R} exp(R) <= top_p S; dropped for sure: mass{z_j >= z_v + R} exp(-R) > top_p S),
 * or the fall-back was taken; bit 2 — the largest selected and the smallest unselected entropy are not more than 2 E apart
 * (strategy "entropy" only).  An unflagged prompt's new ids are those ANY logits within the bounds would have produced.
 * sample_gaps (f32 [B, 2], optional, +inf on entry): the smallest race gap of the unmasked rows in logit units, and the entropy
 * distance above — statistics, not inputs of the step.  Used by esmdiff_amd/certified.py (CertifiedSampler.gibbs_sample). */
typedef struct {
  uint64_t sample_index;   /* Philox key: the prompt's GLOBAL index */
  int32_t step;            /* Philox step index = index of the step in the prompt's chain */
  int32_t n_unmask;        /* positions to unmask in this step (<= 0: the prompt is left alone) */
} esmdiff_gibbs_sample_step;
int esmdiff_gibbs_step_rows(esmdiff_engine* eng, int64_t* x_inout, const int64_t* seq, const float* logits, int32_t ld_logits,
                            float temperature, float top_p, const esmdiff_gibbs_sample_step* params, uint64_t seed, int32_t B,
                            int32_t L, float pair_bound, float entropy_bound, int32_t* sample_flags, float* sample_gaps,
                            void* stream);

/* The whole loop of iterative_sampling_raw for one batch on the device (Philox noise, no time conditioning):
 * T x (forward, gibbs step).  n_unmask_table: [T,B] int32 [host] = positions to unmask per step and prompt
 * (cosine schedule, computed by the host: esmdiff_amd/gibbs.py). */
int esmdiff_gibbs_sample(esmdiff_engine* eng, const int64_t* seq, int64_t* x_inout, int32_t B, int32_t L, int32_t T,
                         float temperature, float top_p, const int32_t* n_unmask_table, const esmdiff_rng* rng,
                         void* stream);

/* Coordinate conditioning (block 0's geometric attention).  Replaces what the reference does inside
 * CustomizedESM3.forward when structure_coords is given (/root/reference/slm/models/net.py:433-441, :468):
 * the caller passes the per-residue backbone frames that esm's build_affine3d_from_coordinates derives from the
 * N/CA/C coordinates (esmdiff_amd/geometry.py computes them on the host), and every following forward of this engine
 * with the same (B, L) adds the geometric-attention branch of block 0.  rot: f32 [B,L,3,3] row-major rotation matrices,
 * trans: f32 [B,L,3], has_frame: u8 [B,L] (0 = no coordinates for that residue) — device pointers, copied on `stream`.
 * rot == NULL clears the frames (the branch is then exactly zero, as in the reference with all-NaN coordinates, and is
 * skipped).  Needs the transformer.blocks.0.geom_attn.* weights in the table given to esmdiff_engine_create
 * (ESMDIFF_E_MISSING otherwise). */
int esmdiff_set_frames(esmdiff_engine* eng, const float* rot, const float* trans, const uint8_t* has_frame,
                       int32_t B, int32_t L, void* stream);

/* VQ-VAE structure-token decoder: structure tokens -> backbone coordinates.  Replaces `esm3.decode(...)` as the
 * reference's decode() helper calls it once per sample (/root/reference/slm/sample_esmdiff.py:40-61, :225-230; the
 * reference also moves the whole ESM3 model between CPU and GPU around it, :58, :176).  The decoder is esm's
 * StructureTokenDecoder [ESM-RECALL, SURVEY.md 8f-1]: nn.Embedding(4101, d) -> the same pre-LN block stack as ESM3
 * (no geometric attention, residue scaling 1) -> Dim6RotStructureHead.  cfg: d_model 1280, n_heads 20, n_layers 30,
 * ffn_hidden 3584 for esm3_structure_decoder_v0; vocab_out / freq_dim / time_conditioning are ignored, residue_scale
 * should be 1.  Weight names: embed.weight, decoder_stack.blocks.{i}.<as ESM3>, decoder_stack.norm.weight,
 * affine_output_projection.{ffn1,norm,proj}.{weight,bias}.  Destroy with esmdiff_engine_destroy.
 * Optional: plddt_head.{0,2,3}.{weight,bias} (esm's RegressionHead(d, 50) on the same hidden state) enables `plddt`.
 * esmdiff_decoder_decode: tokens int64 [B,L] INCLUDING BOS (4098) / EOS (4097); bb_coords f32 [B,L,3,3] = N, CA, C
 * per position (rows 0 and L-1 belong to BOS/EOS and are to be dropped); plddt f32 [B,L] or NULL = mean of the
 * categorical mixture over the 50 bins of [0,1] (what ESMProtein.to_pdb writes into the B-factor column,
 * sample_esmdiff.py:56-61); trans_scale = 10 in esm.
 * Optional: pairwise_classification_head.{downproject,linear1,linear2}.weight + .norm.{weight,bias} (esm's
 * PairwisePredictionHead(d, 128, 128, 64 + 96 + 64 bins, bias=False)) enables `ptm` f32 [B] and `pae` f32 [B,L,L]
 * (NULL to skip; pae needs ptm): decoder_output["ptm"] / ["predicted_aligned_error"] of the reference's decode path
 * (/root/reference/slm/models/utils.py:64-76) — softmax over the 64 predicted-aligned-error bins (max bin 31 A) of every
 * token pair, PAE = expected bin centre, pTM = max_i mean_j E[1 / (1 + (e / d0)^2)] over non-special tokens; pairs with
 * BOS/EOS/special tokens carry the uniform-distribution value in `pae`, as esm leaves them.  Only the PAE slice of
 * linear2 is evaluated (the distogram / direction slices are training targets). */
int esmdiff_decoder_create(const esmdiff_config* cfg, const esmdiff_weight* table, int32_t n_weights,
                           int32_t device, esmdiff_engine** out);
int esmdiff_decoder_decode(esmdiff_engine* dec, const int64_t* tokens, float* bb_coords, float* plddt, float* ptm,
                           float* pae, int32_t B, int32_t L, float trans_scale, void* stream);

/* VQ-VAE structure-token ENCODER: backbone frames -> structure tokens.  Replaces `model.encode(ESMProtein(coordinates))`
 * as protseq_to_data calls it for the DDPM inpainting prior (/root/reference/slm/models/utils.py:136-137,
 * /root/reference/slm/sample_esmdiff.py:196-201).  esm's StructureTokenEncoder [ESM-RECALL, SURVEY.md 8f-4]: 16 nearest
 * residues per residue -> relative-position embedding -> 2 x (geometric attention + SwiGLU FFN, with biases) per
 * neighbourhood -> Linear(d, 128) -> nearest of 4096 codebook vectors.  esm3_structure_encoder_v0: d_model 1024, v_heads
 * 128, n_layers 2, ffn_hidden 4096, d_out 128, n_codes 4096, knn 16, relpos_bins 32.  Weight names:
 * relative_positional_embedding.embedding.weight, transformer.blocks.{i}.geom_attn.{s_norm.weight, proj.{weight,bias},
 * out_proj.{weight,bias}, rotation_scale_per_head, distance_scale_per_head}, transformer.blocks.{i}.ffn.{0,1,3}.{weight,
 * bias}, pre_vq_proj.{weight,bias}, codebook.embeddings.
 * precision (ABI 4): ESMDIFF_PRECISION_F32 runs the two blocks and the projection in float32 (csrc/strict.hip) — the codes then
 * equal a float32 encoder's except at exact distance ties; BF16 is the MFMA path (98-99 % identical codes, every difference a
 * near-tie).
 * encode: ca f32 [B,L,3] (CA positions), rot f32 [B,L,3,3], trans f32 [B,L,3], has_frame u8 [B,L] (the frames of
 * esmdiff_set_frames; residues WITHOUT BOS/EOS) -> tokens int64 [B,L] (device), 4096 (MASK) where has_frame == 0.
 * Synchronises `stream` before returning. */
typedef struct esmdiff_encoder esmdiff_encoder;
int esmdiff_encoder_create(int32_t d_model, int32_t v_heads, int32_t n_layers, int32_t ffn_hidden, int32_t d_out,
                           int32_t n_codes, int32_t knn, int32_t relpos_bins, int32_t precision /* esmdiff_precision */,
                           const esmdiff_weight* table, int32_t n_weights, int32_t device, esmdiff_encoder** out);
void esmdiff_encoder_destroy(esmdiff_encoder* enc);
const char* esmdiff_encoder_last_error(const esmdiff_encoder* enc);
int esmdiff_encoder_encode(esmdiff_encoder* enc, const float* ca, const float* rot, const float* trans,
                           const uint8_t* has_frame, int64_t* tokens, int32_t B, int32_t L, void* stream);

/* Ensemble metrics on CA traces (float64, device pointers in, one double out on the host; each call synchronises
 * `stream`).  They replace the numpy / scipy functions of /root/reference/slm/utils/eval_utils.py that score a generated
 * ensemble against a reference ensemble: js_pwd :227-255 (Jensen-Shannon distance of the per-pair CA distance
 * histograms, bins spanning the reference's range, pseudo count 1e-6, mean over pairs), js_rg :290-316, validity
 * :158-173 (fraction of frames without a CA-CA distance below 2*radius - overlap among pairs |i-j| > k_exclusion),
 * bonding_validity :176-188 (fraction of frames whose adjacent CA distances all stay below the reference's maximum
 * + 1e-6).  ca_*: f64 [n, L, 3].  Values are returned unrounded (the reference rounds to 4 decimals last).
 * w_model / w_ref: per-frame histogram weights f64 [n] or NULL (the reference's `weights=` dictionaries, default ones);
 * kl != 0: mean of scipy.special.kl_div over all bins and columns instead of the mean JS distance (`kl=True`).
 * esmdiff_metrics_pwd: the pairwise-distance features alone, out f64 [n, D], D = pairs (i, j >= i + offset) in
 * numpy.triu_indices order (eval_utils.py:90-102) — what js_tica (:258-289) feeds its TICA fit; esmdiff_metrics_js_columns:
 * the histogram + JS / KL tail on any feature matrix [n, D] (js_tica: D = 2 projected coordinates). */
int esmdiff_metrics_js_pwd(const double* ca_model, int32_t n_model, const double* w_model, const double* ca_ref,
                           int32_t n_ref, const double* w_ref, int32_t L, int32_t n_bins, int32_t pwd_offset, int32_t kl,
                           double* js_out, void* stream);
int esmdiff_metrics_js_rg(const double* ca_model, int32_t n_model, const double* w_model, const double* ca_ref,
                          int32_t n_ref, const double* w_ref, int32_t L, int32_t n_bins, int32_t kl, double* js_out,
                          void* stream);
int esmdiff_metrics_pwd(const double* ca, int32_t n, int32_t L, int32_t pwd_offset, double* out, void* stream);
int esmdiff_metrics_js_columns(const double* x_model, int32_t n_model, const double* w_model, const double* x_ref,
                               int32_t n_ref, const double* w_ref, int32_t D, int32_t n_bins, int32_t kl, double* out,
                               void* stream);
int esmdiff_metrics_validity(const double* ca, int32_t n, int32_t L, double ca_vdw_radius, double allowable_overlap,
                             int32_t k_exclusion, double* out, void* stream);
int esmdiff_metrics_bonding_validity(const double* ca_model, int32_t n_model, const double* ca_ref, int32_t n_ref,
                                     int32_t L, double* out, void* stream);

/* Per-kernel entry points of the parity tests, the per-section profiler of bench.py's roofline leg and the -DED_DEBUG
 * measurement aids are declared in esmdiff_hip_test.h: they are exported by the same library but are not part of the surface a
 * binding of the reference's call sites needs. */

#ifdef __cplusplus
}
#endif
#endif /* ESMDIFF_HIP_H */
