// attention.hip — non-causal multi-head attention, head dim 64, flash-style (gfx950).
//
// Replaces F.scaled_dot_product_attention(q, k, v) as esm's MultiHeadAttention calls it with
// sequence_id=None (the reference passes None: /root/reference/slm/models/net.py:468), i.e. no mask.
// The (B,24,L,L) score tensor is never materialised: each wave owns 32 queries and streams the keys
// in tiles of 64 through LDS with an online softmax, so L = 1026 (BASELINE config 4; K/V per head =
// 263 KB > 160 KB LDS) needs no special casing — only the number of tiles changes.
//
// Workgroup = 4 waves = 128 queries of one (batch, head).  Per 64-key tile:
//   S^T = K · Q^T      2 x 4 v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = Q^T held in registers)
//     "swapped" product: C[key][query] puts a query in a LANE (col = lane & 31) and its 64 scores in
//     that lane's registers (+ the partner lane ^ 32), so row max / row sum are register reductions
//     plus one cross-lane exchange, and the running rescale factor is a per-lane scalar.
//   O^T += V^T · P^T   2 x 4 MFMAs (A = V^T fragments, B = P^T taken directly from the S^T accumulators: the key
//     order inside a 16-key MFMA k-step is permuted identically on both operands, so no lane shuffles are needed).
//     V is read where the QKV GEMM left it (token-major rows of qkv, 128 B per key and head): the LDS tile is
//     [key][d] and the A fragment (a lane = one d, 4 consecutive keys per 8 bytes) comes out of
//     ds_read_b64_tr_b16, the hardware transpose read: within a 16-lane group lane i supplies the address of
//     V[key0 + i/4][d0 + 4*(i%4) ..+3] and receives V[key0 .. key0+3][d0 + i].  There is no V^T tensor in HBM
//     and no transpose kernel (r01a-e had both: 33 us and 158 MB of traffic per layer at config 2).
// K and V tiles are 64 rows x 128 B, staged by LDS-DMA with a source-side XOR swizzle of the 16-byte chunks
// (K: chunk ^= (row>>1)&7, conflict-free for ds_read_b128 as in the GEMM; V: chunk ^= 2*(row&3), which spreads
// the 4 rows x 32 B of a transpose-read group over distinct banks), double buffered.
// q arrives pre-scaled by log2(e)/8 (qk_norm_rope), so the softmax runs on exp2.
#include <type_traits>

#include "kernels.h"

namespace ed {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int KV_TILE = 64;
constexpr int KV_BYTES = KV_TILE * 128;  // 8 KiB per operand per stage

__device__ __forceinline__ void glds16a(const void* gsrc, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even)
  const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
  uint32_t u;
  __builtin_memcpy(&u, &v, 4);
  return u;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // bare v_exp_f32

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x4 lds_read_tr16(const char* p) {  // ds_read_b64_tr_b16
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  bf16x4 r;
  __builtin_memcpy(&r, &v, 8);
  return r;
}

__global__ __launch_bounds__(256, 2) void attention_kernel(const bf16_t* __restrict__ q,
                                                           const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                           int L, int H) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * KV_BYTES];  // [stage][K 8K | V 8K]

  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bool active = q0 < L;  // wave-uniform
  const int qi = lane & 31, hi = lane >> 5;

  const int D = H * 64;
  const bf16_t* kbase = k + (int64_t)b * L * D + h * 64;                 // K of this (batch, head), token 0; row stride D
  const int ldq = 3 * D;                                                 // qkv row stride (elements)
  const bf16_t* vbase = qkv + (int64_t)b * L * ldq + 2 * H * 64 + h * 64;  // V of this (batch, head), token 0

  // ---- LDS-DMA: per tile 8 instructions for K (64 rows x 128 B) + 8 for Vt; 2 + 2 per wave ------
  // instruction i of this wave covers rows (i*4 + wave)*8 + (lane>>3), 16-byte slot lane&7
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7);
  const int vchunk = (lane & 7) ^ (2 * (srow & 3));  // V: logical chunk stored at slot lane&7 of row r (r & 3 == srow & 3)
  int v_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) v_row[i] = (i * 4 + wave) * 8 + srow;
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * (2 * KV_BYTES);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* d = base + (i * 4 + wave) * 1024;
      const int tok = min(kt * KV_TILE + v_row[i], L - 1);  // keys >= L are masked / get weight 0; any finite row will do
      glds16a(kbase + (int64_t)tok * D + schunk * 8, d);
      glds16a(vbase + (int64_t)tok * ldq + vchunk * 8, d + KV_BYTES);
    }
  };

  // ---- Q^T fragments (B operand): lane -> query qi, d chunk ks*2 + hi -------------------------
  bf16x8 qf[4];
  {
    const bf16_t* qrow = q + ((int64_t)b * L + min(q0 + qi, L - 1)) * D + h * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + (ks * 2 + hi) * 8);
  }

  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int fsw = (qi >> 1) & 7;
  // transpose-read addressing: lane (g = lane>>4, i = lane&15) supplies V[key0 + 4*(g>>1) + (i>>2)][d0 + 16*(g&1) +
  // 4*(i&3) ..+3] and receives d = d0 + (lane&31), keys key0 + 4*hi + 0..3 (key0 a multiple of 8, so row&3 = i>>2)
  const int tg = lane >> 4, ti = lane & 15;
  const int tr_row = 4 * (tg >> 1) + (ti >> 2);                    // + key0
  const int tr_c = 2 * (tg & 1) + ((ti & 3) >> 1);                 // 16-byte chunk within the 32-d half
  const int tr_swz = 2 * (ti >> 2);                                // 2 * (row & 3)
  int tr_off[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) tr_off[d] = tr_row * 128 + (((d * 4 + tr_c) ^ tr_swz) << 4) + (ti & 1) * 8;
  const int nkt = (L + KV_TILE - 1) / KV_TILE;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // Rescale threshold (log2 units): the running max is only raised when a tile's max exceeds it by more than
  // THR, so P <= 2^THR instead of <= 1 (exact in f32 accumulation; bf16 P keeps its relative precision) and the
  // 32-register O rescale is skipped on almost every tile.  The decision sits between S and P of the SAME tile
  // and no P·V is pending across it, so everything at the old scale (O and l) is rescaled exactly once.
  constexpr float THR = 8.0f;
  auto tile = [&](int kt, auto MASKED, auto HALF) {  // HALF: at most 32 valid keys left (e.g. L = 258: keys 256, 257)
    constexpr int NT = decltype(HALF)::value ? 1 : 2;
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
    if (active) {
      const char* kl = smem + cur * (2 * KV_BYTES);
      const char* vl = kl + KV_BYTES;
      f32x16 s[2];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 kf =
              *reinterpret_cast<const bf16x8*>(kl + (t * 32 + qi) * 128 + (((ks * 2 + hi) ^ fsw) << 4));
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t], 0, 0, 0);
        }
      }
      if constexpr (decltype(MASKED)::value) {  // keys >= L exist only in the last tile
        const int lim = L - kt * KV_TILE - 4 * hi;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t * 32 + (r & 3) + 8 * (r >> 2) >= lim) s[t][r] = -1e30f;
      }
      float mx = s[0][0];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (!__all(mx - m_run <= THR)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = fast_exp2(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      }
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[t][r] = fast_exp2(s[t][r] - m_run);
          ps += s[t][r];
        }
      l_run += ps;
#pragma unroll
      for (int kk = 0; kk < 2 * NT; ++kk) {
        const int t = kk >> 1, r0 = (kk & 1) * 8;
        union { uint32_t u[4]; bf16x8 v; } pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) pb.u[e] = pk_bf16(s[t][r0 + 2 * e], s[t][r0 + 2 * e + 1]);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          union { bf16x4 h[2]; bf16x8 v; } va;
          va.h[0] = lds_read_tr16(vl + kk * 16 * 128 + tr_off[d]);        // keys kk*16 + 4*hi + 0..3
          va.h[1] = lds_read_tr16(vl + (kk * 16 + 8) * 128 + tr_off[d]);  // keys kk*16 + 8 + 4*hi + 0..3
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va.v, pb.v, o[d], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int kt = 0; kt + 1 < nkt; ++kt) tile(kt, std::false_type{}, std::false_type{});
  const int tail = L - (nkt - 1) * KV_TILE;  // valid keys of the last tile, 1..64
  if (tail <= 32) tile(nkt - 1, std::true_type{}, std::true_type{});
  else if (tail < KV_TILE) tile(nkt - 1, std::true_type{}, std::false_type{});
  else tile(nkt - 1, std::false_type{}, std::false_type{});

  if (!active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + qi;
  if (qrow < L) {
    bf16_t* dst = ctx + ((int64_t)b * L + qrow) * (H * 64) + h * 64;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 p;
        p.x = pk_bf16(o[d][g * 4 + 0] * inv, o[d][g * 4 + 1] * inv);
        p.y = pk_bf16(o[d][g * 4 + 2] * inv, o[d][g * 4 + 3] * inv);
        *reinterpret_cast<uint2*>(dst + d * 32 + g * 8 + 4 * hi) = p;
      }
  }
}

hipError_t launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* qkv, bf16_t* ctx, int B, int L,
                            int H, hipStream_t stream) {
  if (B <= 0 || L <= 0) return hipSuccess;
  dim3 grid((L + 127) / 128, B * H), block(256);
  hipLaunchKernelGGL(attention_kernel, grid, block, 0, stream, q, k, qkv, ctx, L, H);
  return hipGetLastError();
}

}  // namespace ed
