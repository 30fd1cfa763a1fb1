// geom.hip — block 0's geometric attention: coordinate conditioning of ESM3 (gfx950).
//
// Reference call sites: /root/reference/slm/models/net.py:433-441 (coordinates -> frames), :468 (transformer(x,
// sequence_id, affine, affine_mask, chain_id)), :339-346 (v_heads = 256, mask_and_zero_frameless = True);
// the sampler reaches it only when coordinates are supplied (/root/reference/slm/sample_esmdiff.py:88-96, inpainting
// in the default "gibbs" mode).  The arithmetic is esm==3.0.4's GeometricReasoningOriginalImpl, which is not
// vendored: restated from SURVEY.md A.4 [ESM-RECALL], PARITY UNPINNED; checked against oracle/geom_ref.py.
//
//   p = proj(s_norm(x))                    [M, 15*VH] bf16 (GEMM), columns [q_rot | k_rot | value | q_dist | k_dist],
//                                          each VH heads x 3 components
//   q_rot,k_rot,value = R p                (rotation of the residue's frame);   q_dist,k_dist = R p + t
//   logit[q,k] = softplus(rot_scale_h) (q_rot . k_rot)/sqrt3 - softplus(dist_scale_h) |q_dist - k_dist|/sqrt3
//   keys without a frame are excluded; out[q] = R_q^T sum_k softmax(logit)[k] value[k]; rows without a frame := 0
//
// One 64-thread workgroup per (batch, head): the key side (9 floats per key) is rotated once into LDS, then every
// lane owns a query and walks the keys with an online softmax (all lanes read the same key: LDS broadcast, no bank
// conflicts).  Head dimension 3 leaves nothing for MFMA; the kernel is VALU-bound at ~25 ops per (query, key)
// pair: 1.7e9 pairs per forward at config 2, i.e. about one millisecond, and only when coordinates are given.
#include <algorithm>

#include "ed_half.h"
#include "kernels.h"

namespace ed {

__device__ __forceinline__ float gbf(const bf16_t* p) { return ed_h2f(*p); }   // the TU's 16-bit type (ed_half.h)
__device__ __forceinline__ float gbf(const float* p) { return *p; }

// T = bf16_t: the throughput path (bare v_sqrt_f32 / v_exp_f32, softmax in base 2).  T = float: the strict path
// (esmdiff_config.precision = F32): float32 in and out, correctly rounded sqrt, libm-grade expf, natural-base softmax.
template <typename T>
__global__ __launch_bounds__(64) void geom_attention_kernel(const T* __restrict__ P, const float* __restrict__ rot,
                                                           const float* __restrict__ trans,
                                                           const uint8_t* __restrict__ fmask,
                                                           const float* __restrict__ w_rot,
                                                           const float* __restrict__ w_dist, T* __restrict__ out,
                                                           int L, int VH) {
  constexpr bool STRICT = sizeof(T) == 4;
  extern __shared__ __attribute__((aligned(16))) float kl[];  // [Lk][12]: k_rot 3 | k_dist 3 | value 3 | has-frame | pad 2  (three 16-byte reads per key)
  const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
  const int ldp = 15 * VH;
  const int64_t row0 = (int64_t)b * L;
  const float c = 0.57735026918962576f;  // 1/sqrt(3)
  const float LOG2E = STRICT ? 1.0f : 1.44269504088896341f;  // throughput path: softmax in base 2 (v_exp_f32 is exp2)
  const float wr = w_rot[h] * c * LOG2E, wd = w_dist[h] * c * LOG2E;
  auto gsqrt = [](float v) { return STRICT ? sqrtf(v) : __builtin_amdgcn_sqrtf(v); };
  auto gexp = [](float v) { return STRICT ? expf(v) : __builtin_amdgcn_exp2f(v); };
  const int Lk = (L + 3) & ~3;               // key count padded to the 4-key trip; pad keys carry has-frame = 0

  auto load3 = [&](int64_t row, int col, float* v) {
    const T* p = P + row * ldp + col;
    v[0] = gbf(p); v[1] = gbf(p + 1); v[2] = gbf(p + 2);
  };
  auto rotate = [&](const float* R, const float* v, float* o) {  // o = R v, R row-major
    o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  };
  for (int l = lane; l < L; l += 64) {
    const int64_t row = row0 + l;
    float R[9], t[3], v[3], o[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = rot[row * 9 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = trans[row * 3 + i];
    float* k = kl + l * 12;
    load3(row, 3 * VH + 3 * h, v);   // k_rot
    rotate(R, v, o);
    k[0] = o[0]; k[1] = o[1]; k[2] = o[2];
    load3(row, 12 * VH + 3 * h, v);  // k_dist (second half of the distance block, which starts at 9*VH)
    rotate(R, v, o);
    k[3] = o[0] + t[0]; k[4] = o[1] + t[1]; k[5] = o[2] + t[2];
    load3(row, 6 * VH + 3 * h, v);   // value
    rotate(R, v, o);
    k[6] = o[0]; k[7] = o[1]; k[8] = o[2];
    k[9] = fmask[row] ? 1.0f : 0.0f;
    k[10] = k[11] = 0.0f;
  }
  for (int l = L + lane; l < Lk; l += 64) {
    float* k = kl + l * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) k[i] = 0.0f;
  }
  __syncthreads();

  for (int q = lane; q < L; q += 64) {
    const int64_t row = row0 + q;
    float R[9], t[3], v[3], qr[3], qd[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = rot[row * 9 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = trans[row * 3 + i];
    load3(row, 3 * h, v);            // q_rot
    rotate(R, v, qr);
    load3(row, 9 * VH + 3 * h, v);   // q_dist
    rotate(R, v, qd);
    qd[0] += t[0]; qd[1] += t[1]; qd[2] += t[2];
    float m = -3.0e38f, den = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (fmask[row]) {  // rows without a frame are zeroed below; skip their walk
      // four keys per trip: their LDS reads and score arithmetic are independent (the single-key loop exposed one LDS
      // latency per key), the running maximum moves at most once per trip, keys without a frame score -inf
      for (int k = 0; k < Lk; k += 4) {
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* kk = kl + (k + u) * 12;
          const float dx = qd[0] - kk[3], dy = qd[1] - kk[4], dz = qd[2] - kk[5];
          const float v = wr * (qr[0] * kk[0] + qr[1] * kk[1] + qr[2] * kk[2]) - wd * gsqrt(dx * dx + dy * dy + dz * dz);  // bare v_sqrt_f32 (1 ulp) on the bf16 path
          sc[u] = kk[9] != 0.0f ? v : -__builtin_inff();
        }
        const float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        if (mx > m) {  // lazy rescale: rare after the first few trips
          const float a = gexp(m - mx);
          den *= a; o0 *= a; o1 *= a; o2 *= a;
          m = mx;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* kk = kl + (k + u) * 12;
          const float p = gexp(sc[u] - m);  // bare v_exp_f32 on the bf16 path; exp(-inf) = 0 for keys without a frame
          den += p;
          o0 += p * kk[6]; o1 += p * kk[7]; o2 += p * kk[8];
        }
      }
    }
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    if (den > 0.f) {
      const float inv = 1.0f / den;
      o0 *= inv; o1 *= inv; o2 *= inv;
      r0 = R[0] * o0 + R[3] * o1 + R[6] * o2;  // R^T o: back into the residue's frame
      r1 = R[1] * o0 + R[4] * o1 + R[7] * o2;
      r2 = R[2] * o0 + R[5] * o1 + R[8] * o2;
    }
    T* dst = out + row * (3 * VH) + 3 * h;
    if constexpr (STRICT) {
      dst[0] = r0; dst[1] = r1; dst[2] = r2;
    } else {
      auto f2b = [](float f) { return (bf16_t)ed_f2h(f); };   // round to nearest even
      dst[0] = f2b(r0); dst[1] = f2b(r1); dst[2] = f2b(r2);
    }
  }
}

template <typename T>
static hipError_t launch_geom_t(const T* P, const float* rot, const float* trans, const uint8_t* fmask, const float* w_rot,
                                const float* w_dist, T* out, int B, int L, int VH, hipStream_t stream) {
  if (B <= 0 || L <= 0) return hipSuccess;
  // This translation unit is compiled WITHOUT SLP vectorisation (esmdiff_amd/build.py: -fno-slp-vectorize -fno-vectorize), i.e. with
  // no packed float ops (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32).  Inside a two-queue forward this kernel runs
  // while the other queue's 256x256 GEMM is on the GPU, and with a 12.5 KB LDS request two of its one-wave workgroups fit beside
  // a GEMM workgroup on a CU.  In that co-residency the SLP-vectorised build computed lanes 48-63 of the first query trip wrong in
  // a few (sample, head) workgroups per forward, differently from run to run (r06: found by the configs[4] gibbs soak; bisected
  // to this kernel next to the other queue's GEMM; reproduced outside the engine and reduced to one instruction form: v_pk_fma_f32 /
  // v_pk_mul_f32 whose op_sel takes the LOW result half from the HIGH register of src1 — the SLP vectoriser's form of the 3x3
  // rotations above — lose that half's product in lanes 48-63 whenever a wave sharing the SIMD issues memory instructions between
  // back-to-back MFMAs, as every pipelined GEMM does; a 30-line self-checking reproducer needs none of our kernels
  // (scratch/ubench/pk_opsel.hip beside scratch/ubench/mfma_neighbour.hip; profiles/r06_frames_two_queue_race.txt, third pass).
  // The build without packed ops was right in 40 of 40 rounds where the packed build was wrong in 37.  The first fix
  // (r06) was a 40 KB LDS request, which keeps the kernel off the GEMM's CUs but also runs 3 instead of 12 workgroups per CU
  // (2.95 ms instead of 1.72 ms at 50 x 258); the unpacked build at the size it needs takes 1.44 ms.
  // ED_GEOM_MIN_LDS_KB: A/B builds only (scratch/r06_lds_neighbour.sh).
#ifndef ED_GEOM_MIN_LDS_KB
#define ED_GEOM_MIN_LDS_KB 0
#endif
  const size_t lds = std::max<size_t>((size_t)((L + 3) & ~3) * 12 * sizeof(float), (size_t)ED_GEOM_MIN_LDS_KB * 1024);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  if (const hipError_t a_ = ensure_dynamic_lds((const void*)geom_attention_kernel<T>, 150 * 1024); a_ != hipSuccess) return a_;
  hipLaunchKernelGGL(geom_attention_kernel<T>, dim3(VH, B), dim3(64), lds, stream, P, rot, trans, fmask, w_rot, w_dist, out,
                     L, VH);
  return hipGetLastError();
}

hipError_t launch_geom_attention(const bf16_t* P, const float* rot, const float* trans, const uint8_t* fmask,
                                 const float* w_rot, const float* w_dist, bf16_t* out, int B, int L, int VH,
                                 hipStream_t stream) {
  return launch_geom_t<bf16_t>(P, rot, trans, fmask, w_rot, w_dist, out, B, L, VH, stream);
}
hipError_t launch_geom_attention_f32(const float* P, const float* rot, const float* trans, const uint8_t* fmask,
                                     const float* w_rot, const float* w_dist, float* out, int B, int L, int VH,
                                     hipStream_t stream) {
  return launch_geom_t<float>(P, rot, trans, fmask, w_rot, w_dist, out, B, L, VH, stream);
}

}  // namespace ed
