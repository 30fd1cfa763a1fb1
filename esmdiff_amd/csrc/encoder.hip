// encoder.hip — VQ-VAE structure-token ENCODER: backbone frames -> structure tokens (gfx950).
//
// Reference call sites: /root/reference/slm/models/utils.py:136-137 (`model.encode(ESMProtein(coordinates=...))` in
// protseq_to_data), reached from /root/reference/slm/sample_esmdiff.py:196-201 to build the DDPM inpainting prior
// (BASELINE configs[4]).  The module is esm==3.0.4's StructureTokenEncoder (un-vendored): restated from memory
// [ESM-RECALL, SURVEY.md 8f-4], PARITY UNPINNED; checked against oracle/encoder_ref.py.
//
//   per residue i: the K = 16 residues nearest to it (CA distance; sequence distance x 100 + 1e6 where coordinates
//   are missing), nearest first, i itself leading -> x[e] = relpos_embedding[clamp(j_e - i, +-32) + 33] ->
//   2 x [ x += geom_attn(x; frames of the 16 residues) ; x += swiglu_ffn(x) ] (both with biases) over that
//   neighbourhood -> row e = 0 -> 0 if i has no frame -> Linear(d, 128) -> nearest codebook vector (4096 x 128);
//   residues without coordinates get MASK (4096).
// The neighbourhoods are independent sequences of length 16: M = B*L*16 rows go through the shared GEMM / LayerNorm /
// geometric-attention kernels; the small glue kernels live here.  Encoding happens once per input structure, so this
// file aims at correctness and reuse, not at the roofline.
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "kernels.h"

namespace ed {
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4e;

__device__ __forceinline__ float bf2f_(bf16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ bf16_t f2bf_(float f) {
  uint32_t u = __float_as_uint(f);
  return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// one workgroup per (b, i): K selections of the smallest key, lowest index first among equals
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ ca, const uint8_t* __restrict__ has, int L, int K,
                                                  int32_t* __restrict__ edges) {
  extern __shared__ float key[];  // [L] + reduction scratch
  __shared__ float rv[256];
  __shared__ int ri[256];
  const int b = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
  const float* c = ca + (int64_t)b * L * 3;
  const uint8_t* h = has + (int64_t)b * L;
  const bool hi_ = h[i] != 0;
  const float xi = hi_ ? c[i * 3] : 0.f, yi = hi_ ? c[i * 3 + 1] : 0.f, zi = hi_ ? c[i * 3 + 2] : 0.f;
  for (int j = tid; j < L; j += 256) {
    float v;
    if (hi_ && h[j]) {
      const float dx = xi - c[j * 3], dy = yi - c[j * 3 + 1], dz = zi - c[j * 3 + 2];
      v = sqrtf(dx * dx + dy * dy + dz * dz);
    } else {
      v = fabsf((float)(i - j)) * 1e2f + 1e6f;
    }
    key[j] = v;
  }
  __syncthreads();
  for (int e = 0; e < K; ++e) {
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < L; j += 256)
      if (key[j] < bv) {  // strided ascending: the first hit of a thread is its lowest index
        bv = key[j];
        bi = j;
      }
    rv[tid] = bv;
    ri[tid] = bi;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
      if (tid < s) {
        const float ov = rv[tid + s];
        const int oi = ri[tid + s];
        if (ov < rv[tid] || (ov == rv[tid] && oi < ri[tid])) {
          rv[tid] = ov;
          ri[tid] = oi;
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      edges[((int64_t)b * L + i) * K + e] = ri[0];
      key[ri[0]] = INFINITY;
    }
    __syncthreads();
  }
}

// x[m] = table[clamp(edge - edge_0, +-bins) + bins + 1]; frames of the neighbours gathered alongside
__global__ __launch_bounds__(256) void neighbourhood_kernel(const int32_t* __restrict__ edges, const float* __restrict__ table,
                                                            const float* __restrict__ rot, const float* __restrict__ trans,
                                                            const uint8_t* __restrict__ has, int L, int K, int D, int bins,
                                                            float* __restrict__ x, float* __restrict__ nrot,
                                                            float* __restrict__ ntrans, uint8_t* __restrict__ nmask) {
  const int64_t m = blockIdx.x;  // (b*L + i)*K + e
  const int64_t bi = m / K;
  const int b = (int)(bi / L);
  const int j = edges[m], j0 = edges[bi * K];
  int diff = j - j0;
  diff = diff < -bins ? -bins : (diff > bins ? bins : diff);
  const float* src = table + (int64_t)(diff + bins + 1) * D;
  float* o = x + m * D;
  for (int c = threadIdx.x * 4; c < D; c += 1024) *reinterpret_cast<f32x4e*>(o + c) = *reinterpret_cast<const f32x4e*>(src + c);
  const int64_t r = (int64_t)b * L + j;
  if (threadIdx.x < 9) nrot[m * 9 + threadIdx.x] = rot[r * 9 + threadIdx.x];
  if (threadIdx.x < 3) ntrans[m * 3 + threadIdx.x] = trans[r * 3 + threadIdx.x];
  if (threadIdx.x == 0) nmask[m] = has[r];
}

__global__ __launch_bounds__(256) void add_bias_bf16_kernel(bf16_t* __restrict__ x, const float* __restrict__ bias, int64_t n, int N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  x[i] = f2bf_(bf2f_(x[i]) + bias[i % N]);
}

// u bf16 [M, 2H] = [gate | up] (bias added here) -> mid bf16 [M, H] = silu(gate) * up
__global__ __launch_bounds__(256) void bias_swiglu_kernel(const bf16_t* __restrict__ u, const float* __restrict__ bias, int64_t M,
                                                          int H, bf16_t* __restrict__ mid) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * H) return;
  const int64_t m = i / H;
  const int c = (int)(i - m * H);
  const float g = bf2f_(u[m * 2 * H + c]) + bias[c], v = bf2f_(u[m * 2 * H + H + c]) + bias[H + c];
  mid[i] = f2bf_(g / (1.0f + expf(-g)) * v);
}

// z[r] = has[r] ? x[r*K] + delta[r*K] : 0  (the query residue's row of its neighbourhood), as the bf16 operand of pre_vq_proj
__global__ __launch_bounds__(256) void query_rows_kernel(const float* __restrict__ x, const bf16_t* __restrict__ delta,
                                                         const uint8_t* __restrict__ has, int K, int D, bf16_t* __restrict__ z) {
  const int64_t r = blockIdx.x;
  const int64_t m = r * K;
  const bool keep = has[r] != 0;
  for (int c = threadIdx.x; c < D; c += 256) z[r * D + c] = keep ? f2bf_(x[m * D + c] + bf2f_(delta[m * D + c])) : (bf16_t)0;
}

// nearest codebook vector (squared Euclidean distance, lowest index among equals); MASK for residues without a frame
__global__ __launch_bounds__(256) void codebook_kernel(const float* __restrict__ z, const float* __restrict__ code,
                                                       const uint8_t* __restrict__ has, int n_codes, int d_out,
                                                       int64_t* __restrict__ tok) {
  extern __shared__ float zs[];
  __shared__ float rv[256];
  __shared__ int ri[256];
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x;
  for (int c = tid; c < d_out; c += 256) zs[c] = z[r * d_out + c];
  __syncthreads();
  float bv = INFINITY;
  int bi = 0x7fffffff;
  for (int k = tid; k < n_codes; k += 256) {
    const float* e = code + (int64_t)k * d_out;
    float s = 0.f;
    for (int c = 0; c < d_out; ++c) {
      const float d = zs[c] - e[c];
      s += d * d;
    }
    if (s < bv) {
      bv = s;
      bi = k;
    }
  }
  rv[tid] = bv;
  ri[tid] = bi;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (tid < s) {
      if (rv[tid + s] < rv[tid] || (rv[tid + s] == rv[tid] && ri[tid + s] < ri[tid])) {
        rv[tid] = rv[tid + s];
        ri[tid] = ri[tid + s];
      }
    }
    __syncthreads();
  }
  if (tid == 0) tok[r] = has[r] ? ri[0] : ESMDIFF_MASK_ID;
}

// z[r] = has[r] ? x[r*K] : 0 in float32 (precision = F32: x already holds every residual)
__global__ __launch_bounds__(256) void query_rows_f32_kernel(const float* __restrict__ x, const uint8_t* __restrict__ has, int K,
                                                             int D, float* __restrict__ z) {
  const int64_t r = blockIdx.x;
  const bool keep = has[r] != 0;
  for (int c = threadIdx.x; c < D; c += 256) z[r * D + c] = keep ? x[r * K * D + c] : 0.f;
}

struct Block {
  float *s_norm_w, *proj_b, *out_b, *w_rot, *w_dist, *ln_w, *ln_b, *b1, *b3;
  bf16_t *proj_w, *out_w, *w1, *w3;
  float *fproj_w, *fout_w, *fw1, *fw3;   // precision = F32
};

}  // namespace
}  // namespace ed

using namespace ed;

struct esmdiff_encoder {
  int device = 0, D = 0, VH = 0, FH = 0, n_layers = 0, d_out = 0, n_codes = 0, knn = 0, bins = 0;
  std::string err;
  std::vector<void*> allocs;
  std::vector<Block> blocks;
  float *relpos = nullptr, *vq_b = nullptr, *code = nullptr;
  bf16_t* vq_w = nullptr;
  float* fvq_w = nullptr;
  bool strict = false;   // precision = F32: float32 weights and activations, the linears on the f32-input MFMA (csrc/strict.hip)
};

namespace {
thread_local std::string g_enc_error;

int efail(esmdiff_encoder* e, int code, const std::string& msg) {
  if (e) e->err = msg; else g_enc_error = msg;
  return code;
}

const esmdiff_weight* find(const esmdiff_weight* t, int n, const std::string& name) {
  for (int i = 0; i < n; ++i)
    if (t[i].name && name == t[i].name) return &t[i];
  return nullptr;
}

template <typename T>
T* dget(esmdiff_encoder* e, size_t n) {
  void* v = nullptr;
  if (hipMalloc(&v, (n ? n : 1) * sizeof(T) + 256) != hipSuccess) return nullptr;
  e->allocs.push_back(v);
  return (T*)v;
}

int load(esmdiff_encoder* e, const esmdiff_weight* t, int n, const std::string& name, std::initializer_list<int64_t> shape,
         float** f32_out, bf16_t** bf16_out) {
  const esmdiff_weight* w = find(t, n, name);
  if (!w) return efail(e, ESMDIFF_E_MISSING, "missing weight '" + name + "'");
  int64_t numel = 1;
  int i = 0;
  if (w->ndim != (int)shape.size()) return efail(e, ESMDIFF_E_SHAPE, "weight '" + name + "': wrong rank");
  for (int64_t s : shape) {
    if (w->shape[i++] != s) return efail(e, ESMDIFF_E_SHAPE, "weight '" + name + "': wrong shape");
    numel *= s;
  }
  if (!w->data || (w->dtype != ESMDIFF_F32 && w->dtype != ESMDIFF_BF16)) return efail(e, ESMDIFF_E_SHAPE, "weight '" + name + "': bad dtype / null");
  if (f32_out) {
    *f32_out = dget<float>(e, numel);
    if (!*f32_out || launch_to_f32(w->data, w->dtype, *f32_out, numel, 0) != hipSuccess) return efail(e, ESMDIFF_E_HIP, "to_f32 failed");
  } else {
    *bf16_out = dget<bf16_t>(e, numel);
    if (!*bf16_out || launch_to_bf16(w->data, w->dtype, *bf16_out, numel, 0) != hipSuccess) return efail(e, ESMDIFF_E_HIP, "to_bf16 failed");
  }
  return 0;
}

struct Tmp {  // per-call device scratch
  std::vector<void*> p;
  ~Tmp() {
    for (void* q : p) hipFree(q);
  }
  template <typename T>
  T* get(size_t n) {
    void* v = nullptr;
    if (hipMalloc(&v, (n ? n : 1) * sizeof(T) + 256) != hipSuccess) return nullptr;
    p.push_back(v);
    return (T*)v;
  }
};
}  // namespace

extern "C" {

const char* esmdiff_encoder_last_error(const esmdiff_encoder* e) { return e ? e->err.c_str() : g_enc_error.c_str(); }

void esmdiff_encoder_destroy(esmdiff_encoder* e) {
  if (!e) return;
  hipSetDevice(e->device);
  hipDeviceSynchronize();
  for (void* p : e->allocs) hipFree(p);
  delete e;
}

int esmdiff_encoder_create(int32_t d_model, int32_t v_heads, int32_t n_layers, int32_t ffn_hidden, int32_t d_out,
                           int32_t n_codes, int32_t knn, int32_t relpos_bins, int32_t precision, const esmdiff_weight* table,
                           int32_t n, int32_t device, esmdiff_encoder** out) {
  if (!table || !out || n <= 0) return efail(nullptr, ESMDIFF_E_INVALID, "null argument");
  if (precision != ESMDIFF_PRECISION_BF16 && precision != ESMDIFF_PRECISION_F32) return efail(nullptr, ESMDIFF_E_INVALID, "precision: 0 (bf16) or 1 (f32)");
  *out = nullptr;
  if (d_model % 256 || d_model > 2048 || v_heads % 128 || ffn_hidden % 128 || d_out % 128 || n_layers <= 0 || knn <= 0 ||
      n_codes <= 0 || relpos_bins <= 0)
    return efail(nullptr, ESMDIFF_E_INVALID, "invalid encoder configuration");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev || hipSetDevice(device) != hipSuccess)
    return efail(nullptr, ESMDIFF_E_NODEVICE, "no such HIP device");
  esmdiff_encoder* e = new esmdiff_encoder;
  e->device = device; e->D = d_model; e->VH = v_heads; e->FH = ffn_hidden; e->n_layers = n_layers; e->d_out = d_out;
  e->n_codes = n_codes; e->knn = knn; e->bins = relpos_bins;
  e->strict = precision == ESMDIFF_PRECISION_F32;
  const bool strict = e->strict;
  const int D = d_model, VH = v_heads, FH = ffn_hidden;
#define ETRY(x)                        \
  do {                                 \
    if (int _r = (x)) {                \
      g_enc_error = e->err;            \
      esmdiff_encoder_destroy(e);      \
      return _r;                       \
    }                                  \
  } while (0)
  ETRY(load(e, table, n, "relative_positional_embedding.embedding.weight", {2 * relpos_bins + 2, D}, &e->relpos, nullptr));
  e->blocks.resize(n_layers);
  for (int i = 0; i < n_layers; ++i) {
    Block& b = e->blocks[i];
    const std::string p = "transformer.blocks." + std::to_string(i) + ".";
    ETRY(load(e, table, n, p + "geom_attn.s_norm.weight", {D}, &b.s_norm_w, nullptr));
    b.proj_w = b.out_w = b.w1 = b.w3 = nullptr;
    b.fproj_w = b.fout_w = b.fw1 = b.fw3 = nullptr;
    ETRY(load(e, table, n, p + "geom_attn.proj.weight", {15 * VH, D}, strict ? &b.fproj_w : nullptr, &b.proj_w));
    ETRY(load(e, table, n, p + "geom_attn.proj.bias", {15 * VH}, &b.proj_b, nullptr));
    ETRY(load(e, table, n, p + "geom_attn.out_proj.weight", {D, 3 * VH}, strict ? &b.fout_w : nullptr, &b.out_w));
    ETRY(load(e, table, n, p + "geom_attn.out_proj.bias", {D}, &b.out_b, nullptr));
    ETRY(load(e, table, n, p + "geom_attn.rotation_scale_per_head", {VH}, &b.w_rot, nullptr));
    ETRY(load(e, table, n, p + "geom_attn.distance_scale_per_head", {VH}, &b.w_dist, nullptr));
    ETRY(load(e, table, n, p + "ffn.0.weight", {D}, &b.ln_w, nullptr));
    ETRY(load(e, table, n, p + "ffn.0.bias", {D}, &b.ln_b, nullptr));
    ETRY(load(e, table, n, p + "ffn.1.weight", {2 * FH, D}, strict ? &b.fw1 : nullptr, &b.w1));
    ETRY(load(e, table, n, p + "ffn.1.bias", {2 * FH}, &b.b1, nullptr));
    ETRY(load(e, table, n, p + "ffn.3.weight", {D, FH}, strict ? &b.fw3 : nullptr, &b.w3));
    ETRY(load(e, table, n, p + "ffn.3.bias", {D}, &b.b3, nullptr));
  }
  ETRY(load(e, table, n, "pre_vq_proj.weight", {d_out, D}, strict ? &e->fvq_w : nullptr, &e->vq_w));
  ETRY(load(e, table, n, "pre_vq_proj.bias", {d_out}, &e->vq_b, nullptr));
  ETRY(load(e, table, n, "codebook.embeddings", {n_codes, d_out}, &e->code, nullptr));
  if (hipDeviceSynchronize() != hipSuccess) ETRY(efail(e, ESMDIFF_E_HIP, "weight conversion failed"));
  std::vector<float> hw(VH);
  for (Block& b : e->blocks)
    for (float* p : {b.w_rot, b.w_dist}) {  // softplus once, on the host
      hipMemcpy(hw.data(), p, VH * 4, hipMemcpyDeviceToHost);
      for (float& v : hw) v = v > 20.f ? v : log1pf(expf(v));
      hipMemcpy(p, hw.data(), VH * 4, hipMemcpyHostToDevice);
    }
#undef ETRY
  *out = e;
  return 0;
}

int esmdiff_encoder_encode(esmdiff_encoder* e, const float* ca, const float* rot, const float* trans, const uint8_t* has_frame,
                           int64_t* tokens, int32_t B, int32_t L, void* stream) {
  if (!e || !ca || !rot || !trans || !has_frame || !tokens || B <= 0 || L <= 0) return efail(e, ESMDIFF_E_INVALID, "invalid argument");
  if (L > 8192) return efail(e, ESMDIFF_E_CAPACITY, "L > 8192");
  if (hipSetDevice(e->device) != hipSuccess) return efail(e, ESMDIFF_E_HIP, "hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  const int D = e->D, VH = e->VH, FH = e->FH, K = e->knn < L ? e->knn : L;
  const int64_t R = (int64_t)B * L, M = R * K;
  if (M > (int64_t)1 << 30) return efail(e, ESMDIFF_E_CAPACITY, "B*L*knn too large");
  Tmp t;
  int32_t* edges = t.get<int32_t>(M);
  float *x = t.get<float>(M * D), *nrot = t.get<float>(M * 9), *ntrans = t.get<float>(M * 3), *zq = t.get<float>(R * e->d_out);
  uint8_t* nmask = t.get<uint8_t>(M);
  bf16_t *h = t.get<bf16_t>(M * D), *P = t.get<bf16_t>(M * 15 * VH), *G = t.get<bf16_t>(M * 3 * VH), *dA = t.get<bf16_t>(M * D),
         *dF = t.get<bf16_t>(M * D), *U = t.get<bf16_t>(M * 2 * FH), *mid = t.get<bf16_t>(M * FH), *z = t.get<bf16_t>(R * D);
  if (!edges || !x || !nrot || !ntrans || !zq || !nmask || !h || !P || !G || !dA || !dF || !U || !mid || !z)
    return efail(e, ESMDIFF_E_HIP, "encoder scratch allocation failed");
#define HT(call)                                                                                  \
  do {                                                                                            \
    hipError_t _s = (call);                                                                       \
    if (_s != hipSuccess) return efail(e, ESMDIFF_E_HIP, std::string(#call) + ": " + hipGetErrorString(_s)); \
  } while (0)
  hipLaunchKernelGGL(knn_kernel, dim3(L, B), dim3(256), (size_t)L * sizeof(float), st, ca, has_frame, L, K, edges);
  hipLaunchKernelGGL(neighbourhood_kernel, dim3((unsigned)M), dim3(256), 0, st, edges, e->relpos, rot, trans, has_frame, L, K, D,
                     e->bins, x, nrot, ntrans, nmask);
  HT(hipGetLastError());
  const int Mi = (int)M;
  if (e->strict) {
    // float32 end to end: x is updated in place by the residual epilogues (x + (acc + bias) / 1), the SwiGLU bias rides the
    // FFN-up GEMM's store epilogue.  Same order of operations as oracle/encoder_ref.py.
    float *fh = t.get<float>(M * D), *fP = t.get<float>(M * 15 * VH), *fG = t.get<float>(M * 3 * VH), *fU = t.get<float>(M * 2 * FH),
          *fmid = t.get<float>(M * FH), *fz = t.get<float>(R * D);
    if (!fh || !fP || !fG || !fU || !fmid || !fz) return efail(e, ESMDIFF_E_HIP, "encoder scratch allocation failed");
    for (const Block& b : e->blocks) {
      HT(launch_layernorm_f32(x, b.s_norm_w, nullptr, fh, Mi, D, st));
      HT(launch_gemm_f32(fh, D, b.fproj_w, fP, b.proj_b, Mi, 15 * VH, D, 15 * VH, 15 * VH, 1.f, ESMDIFF_F32EPI_STORE, st));
      HT(launch_geom_attention_f32(fP, nrot, ntrans, nmask, b.w_rot, b.w_dist, fG, (int)R, K, VH, st));
      HT(launch_gemm_f32(fG, 3 * VH, b.fout_w, x, b.out_b, Mi, D, 3 * VH, D, D, 1.f, ESMDIFF_F32EPI_RESID_DIV, st));
      HT(launch_layernorm_f32(x, b.ln_w, b.ln_b, fh, Mi, D, st));
      HT(launch_gemm_f32(fh, D, b.fw1, fU, b.b1, Mi, 2 * FH, D, 2 * FH, 2 * FH, 1.f, ESMDIFF_F32EPI_STORE, st));
      HT(launch_swiglu_f32(fU, fmid, Mi, FH, st));
      HT(launch_gemm_f32(fmid, FH, b.fw3, x, b.b3, Mi, D, FH, D, D, 1.f, ESMDIFF_F32EPI_RESID_DIV, st));
    }
    hipLaunchKernelGGL(query_rows_f32_kernel, dim3((unsigned)R), dim3(256), 0, st, x, has_frame, K, D, fz);
    HT(launch_gemm_f32(fz, D, e->fvq_w, zq, e->vq_b, (int)R, e->d_out, D, e->d_out, e->d_out, 1.f, ESMDIFF_F32EPI_STORE, st));
    hipLaunchKernelGGL(codebook_kernel, dim3((unsigned)R), dim3(256), (size_t)e->d_out * sizeof(float), st, zq, e->code, has_frame,
                       e->n_codes, e->d_out, tokens);
    HT(hipGetLastError());
    HT(hipStreamSynchronize(st));
    return 0;
  }
  bool pending = false;
  for (const Block& b : e->blocks) {
    // x += dF (previous block's FFN); h = s_norm(x)
    HT(launch_add_layernorm_bf16(x, pending ? dF : nullptr, nullptr, 1, b.s_norm_w, nullptr, h, Mi, D, st));
    HT(launch_gemm_bf16(h, b.proj_w, P, nullptr, Mi, 15 * VH, D, 15 * VH, 15 * VH, 1.f, ESMDIFF_EPI_BF16, st));
    hipLaunchKernelGGL(add_bias_bf16_kernel, dim3((unsigned)((M * 15 * VH + 255) / 256)), dim3(256), 0, st, P, b.proj_b, M * 15 * VH, 15 * VH);
    HT(launch_geom_attention(P, nrot, ntrans, nmask, b.w_rot, b.w_dist, G, (int)R, K, VH, st));
    HT(launch_gemm_bf16(G, b.out_w, dA, nullptr, Mi, D, 3 * VH, D, D, 1.f, ESMDIFF_EPI_BF16, st));
    hipLaunchKernelGGL(add_bias_bf16_kernel, dim3((unsigned)((M * D + 255) / 256)), dim3(256), 0, st, dA, b.out_b, M * D, D);
    // x += dA; h = LN(x); FFN
    HT(launch_add_layernorm_bf16(x, dA, nullptr, 1, b.ln_w, b.ln_b, h, Mi, D, st));
    HT(launch_gemm_bf16(h, b.w1, U, nullptr, Mi, 2 * FH, D, 2 * FH, 2 * FH, 1.f, ESMDIFF_EPI_BF16, st));
    hipLaunchKernelGGL(bias_swiglu_kernel, dim3((unsigned)((M * FH + 255) / 256)), dim3(256), 0, st, U, b.b1, M, FH, mid);
    HT(launch_gemm_bf16(mid, b.w3, dF, nullptr, Mi, D, FH, D, D, 1.f, ESMDIFF_EPI_BF16, st));
    hipLaunchKernelGGL(add_bias_bf16_kernel, dim3((unsigned)((M * D + 255) / 256)), dim3(256), 0, st, dF, b.b3, M * D, D);
    pending = true;
  }
  hipLaunchKernelGGL(query_rows_kernel, dim3((unsigned)R), dim3(256), 0, st, x, dF, has_frame, K, D, z);
  HT(launch_gemm_bf16(z, e->vq_w, zq, e->vq_b, (int)R, e->d_out, D, e->d_out, e->d_out, 1.f, ESMDIFF_EPI_BIAS_F32, st));
  hipLaunchKernelGGL(codebook_kernel, dim3((unsigned)R), dim3(256), (size_t)e->d_out * sizeof(float), st, zq, e->code, has_frame,
                     e->n_codes, e->d_out, tokens);
  HT(hipGetLastError());
  HT(hipStreamSynchronize(st));  // the scratch is freed on return
#undef HT
  return 0;
}

}  // extern "C"
