// gemm.hip — bf16 MFMA GEMM for gfx950 with fused epilogues:  out = epi(A[M,K] · W[N,K]^T).
//
// These are the linears of the ESM3 block the reference reaches through esm's TransformerStack
// (/root/reference/slm/models/net.py:339-346, :468) and of RegressionHead (net.py:301): QKV, attention
// out-projection, SwiGLU FFN up/down, and the 4101-way structure head.  The reference runs them as
// separate cuBLAS/rocBLAS GEMMs followed by separate elementwise kernels (residual add and scale,
// SwiGLU, GELU, bias); here each epilogue is fused into the GEMM that produces its input.
//
// Both operands are K-contiguous (activations [M,K], nn.Linear weights [N,K]), which is exactly the
// MFMA fragment shape: every lane's A and B fragment is 8 contiguous bf16 (one 16-byte LDS read).
//
// Tile: 128(M) x 128(N) x 64(K), 256 threads = 4 waves in 2x2, each wave 64x64 = 2x2 tiles of
// v_mfma_f32_32x32x16_bf16.  Staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave instruction = 8 tile rows), double buffered, next tile issued before the current tile's MFMAs.
// LDS image: rows of 128 B (8 chunks of 16 B).  LDS-DMA writes lane-linear, so the bank-conflict
// swizzle is applied to the per-lane SOURCE address and again on the fragment read
// (chunk' = chunk ^ ((row >> 1) & 7)): the 16 lanes of every ds_read_b128 group then hit 16 distinct
// 16-byte slots of the 256-byte bank row (conflict free for the 32x32x16 fragment pattern).
// Epilogue: accumulators go through the (now idle) LDS so that global stores are whole 128/256-byte
// row segments instead of 2-4 byte scatters.
// Grid: one workgroup per tile, XCD-aware (block b runs on XCD b%8, so each XCD is given a contiguous
// run of tiles) and rasterised in 8x8 super-tiles so the 64 tiles resident on an XCD share 8 A panels
// and 8 W panels in that XCD's L2.
//
// Small M (< 1 152 rows, small_max_rows(): a handful of short sequences).  There are then only tiles_n..8*tiles_n tiles, so most CUs
// idle while a few stream whole weight matrices, and with two stages each workgroup waits one full HBM latency per
// K-tile (M = 240, N = 1536, K = 4096: 24 workgroups, 63 us for 12.6 MB).  Two changes on this path:
//   NST = 4  four LDS stages (128 KiB, one workgroup per CU), three K-tiles in flight, counted vmcnt waits;
//   split-K  when the caller provides a workspace, the K range is cut into S slices (S is a function of N and K only,
//            never of M, so a sample's logits do not depend on the batch it is in).  Slice s writes its raw f32
//            tile to partial[s]; `splitk_reduce_kernel` adds the slices in the fixed order s = 0..S-1 and applies
//            the epilogue.  Deterministic, no atomics.  (A single-kernel variant with per-tile tickets and
//            __threadfence() — "last arriver reduces" — was 2.7x SLOWER than no split at all: the agent-scope
//            release/acquire costs an L2 write-back/invalidate per workgroup on this multi-XCD part.)
#include <stdlib.h>

#include "ed_half.h"
#include <stdio.h>
#include "kernels.h"

namespace ed {

typedef ed_half8 bf16x8;   // 8 operand elements of the TU's 16-bit type (ed_half.h: bf16, or f16 in the ed16 build)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage
constexpr int GROUP_M = 8;

// the weight stream with the non-temporal policy (aux = 2): when only a few row tiles re-read a weight tile it lands
// ~18 % sooner (configs[0]: +3 %); with many row tiles sharing it through L2 it loses (M = 774: -7 %), see launch_tiles
__device__ __forceinline__ void glds16_nt(const void* gsrc, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 2);
}
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

// silu(g) * u with v_exp_f32 + v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence: the result is
// rounded to bf16 (8 bits) right after
__device__ __forceinline__ float silu_mul(float g, float u) {
  return g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.44269504088896341f)) * u;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) { return ed_pack2(a, b); }   // the TU's 16-bit type (ed_half.h)

// MI = 32-row blocks per wave: 2 -> 128-row tiles (waves 2x2 of 64x64), 1 -> 64-row tiles (waves 2x2 of 32x64) for the
// small-M path, where the 128-row grid leaves most CUs idle.  The per-element K order is the same, so the tile height
// never changes a result bit.
template <int EPI, int NST, int MI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const bf16_t* __restrict__ A,
                                                           const bf16_t* __restrict__ W, void* __restrict__ out,
                                                           const float* __restrict__ bias, int M, int N, int K,
                                                           int ldc, int n_valid, float alpha, int tiles_m,
                                                           int tiles_n, int ksplit, float* __restrict__ partial,
                                                           int64_t pstride, int w_row_stride, int w_kt_stride,
                                                           int64_t w_nblk_stride, int w_nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NST stages][A 8K*MI | B 16K]
  constexpr int BM = 64 * MI, A_BYTES = BM * BK * 2, STAGE = A_BYTES + TILE_BYTES;

  // ---- tile assignment: XCD-contiguous, 8x8 super-tile raster; K slices are the slow index ------
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lin_all = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int n_tiles = tiles_m * tiles_n;
  const int ks_id = lin_all / n_tiles, lin = lin_all - ks_id * n_tiles;
  const int per_group = GROUP_M * tiles_n;
  const int grp = lin / per_group, in_grp = lin - grp * per_group;
  const int gm0 = grp * GROUP_M;
  const int gsz = min(GROUP_M, tiles_m - gm0);
  const int mt = gm0 + in_grp % gsz, nt = in_grp / gsz;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- LDS-DMA source pointers (4 instructions per operand per wave per K-tile) ---------------
  // instruction i covers tile rows (i*4 + wave)*8 .. +8 ; lane -> row + (lane>>3), 16-byte slot lane&7
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7);  // logical chunk stored at slot lane&7
  const bf16_t* a_src[2 * MI];
  const bf16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 4 + wave) * 8 + srow;
    if (i < 2 * MI) {
      const int am = min(m0 + r, M - 1);
      a_src[i] = A + (int64_t)am * K + schunk * 8;
    }
    w_src[i] = W + (int64_t)nt * w_nblk_stride + (int64_t)r * w_row_stride + schunk * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      char* da = base + (i * 4 + wave) * 1024;
      if (i < 2 * MI) glds16(a_src[i] + kt * BK, da);
      if (w_nt) glds16_nt(w_src[i] + (int64_t)kt * w_kt_stride, da + A_BYTES);
      else glds16(w_src[i] + (int64_t)kt * w_kt_stride, da + A_BYTES);
    }
  };

  // ---- fragment read offsets ------------------------------------------------------------------
  const int frow = lane & 31, khalf = lane >> 5;
  const int fsw = (frow >> 1) & 7;
  int a_off[MI], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i < MI) a_off[i] = (wm * (32 * MI) + i * 32 + frow) * 128;
    b_off[i] = A_BYTES + (wn * 64 + i * 32 + frow) * 128;
  }

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk_per = (K / BK) / ksplit;  // the launcher guarantees divisibility
  const int kt0 = ks_id * nk_per;
  auto compute = [&](int buf) {
    const char* base = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((ks * 2 + khalf) ^ fsw) << 4;
      bf16x8 a[MI], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < MI) a[i] = *reinterpret_cast<const bf16x8*>(base + a_off[i] + coff);
        b[i] = *reinterpret_cast<const bf16x8*>(base + b_off[i] + coff);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = ED_MFMA_32x32x16(a[i], b[j], acc[i][j]);
    }
  };
  if constexpr (NST == 2) {
    stage(0, kt0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < nk_per; ++i) {
      const int cur = i & 1;
      if (i + 1 < nk_per) stage(cur ^ 1, kt0 + i + 1);
      compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // NST - 1 K-tiles in flight; every tile is 8 (MI = 2) or 6 (MI = 1) LDS-DMA instructions per wave, so "tile i has
    // landed" is vmcnt(that many * tiles issued after it).  One barrier per K-tile: it publishes tile i and retires buffer (i-1) % NST,
    // which the refill issued right after it overwrites.
    // (six stages of the 64-row tile, 144 KiB, measured no faster than four: 15.1 vs 14.6 us on the FFN-up shape at M = 240)
    static_assert(NST == 4, "stage ring is instantiated for 4 stages");
    constexpr int PER = 2 * MI + 4;  // LDS-DMA instructions per wave and K-tile
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
      if (p < nk_per) stage(p, kt0 + p);
    int cur = 0, fill = NST - 1;  // ring positions of K-tile i and of the K-tile issued in iteration i
    for (int i = 0; i < nk_per; ++i) {
      const int after = min(NST - 2, nk_per - 1 - i);
      // vmcnt(PER * after): immediates only
      if (after >= 4) {
        if constexpr (NST == 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      } else if (after == 3) {
        if constexpr (NST == 6) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      } else if (after == 2) {
        if constexpr (PER == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else if (after == 1) {
        if constexpr (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // raw barrier: __syncthreads() would add vmcnt(0) and drain the ring
      if (i + NST - 1 < nk_per) stage(fill, kt0 + i + NST - 1);
      compute(cur);
      cur = cur + 1 == NST ? 0 : cur + 1;
      fill = fill + 1 == NST ? 0 : fill + 1;
    }
    __syncthreads();  // the epilogue slabs alias the stages
  }

  // ---- epilogue: registers -> this wave's LDS slab -> whole-row global stores -------------------
  constexpr int WR = 32 * MI;  // rows of a wave's tile
  float* slab = reinterpret_cast<float*>(smem) + wave * (WR * 64);
  const int ccol = lane & 31, rhalf = lane >> 5;
  if (partial != nullptr) {
    // split-K slice: raw f32 tile -> partial[ks_id][m][n] (row stride N; rows >= M are never written or read)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
          slab[row * 64 + j * 32 + ccol] = acc[i][j][r];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int prow = lane >> 4, pc4 = (lane & 15) * 4;  // 16 lanes per 64-float row, 4 rows per iteration
    float* pbase = partial + (int64_t)ks_id * pstride + (int64_t)(m0 + wm * WR) * N + n0 + wn * 64;
#pragma unroll 4
    for (int it = 0; it < WR / 4; ++it) {
      const int row = it * 4 + prow;
      if (m0 + wm * WR + row < M)
        *reinterpret_cast<f32x4*>(pbase + (int64_t)row * N + pc4) = *reinterpret_cast<const f32x4*>(slab + row * 64 + pc4);
    }
    return;
  }
  constexpr int SW = (EPI == ESMDIFF_EPI_SWIGLU_BF16) ? 32 : 64;  // slab width in floats
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if constexpr (EPI == ESMDIFF_EPI_SWIGLU_BF16) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
        const float g = acc[i][0][r], u = acc[i][1][r];
        slab[row * SW + ccol] = silu_mul(g, u);  // silu(gate) * up
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rhalf;
          slab[row * SW + j * 32 + ccol] = acc[i][j][r];
        }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS writes done (slab is private to the wave)
  __builtin_amdgcn_wave_barrier();

  constexpr int LPR = SW / 4;            // lanes per row (float4 each)
  constexpr int RPI = 64 / LPR;          // rows per iteration
  const int rr_ = lane / LPR, c4 = (lane % LPR) * 4;
  const int ncol0 = (EPI == ESMDIFF_EPI_SWIGLU_BF16) ? (n0 + wn * 64) / 2 : (n0 + wn * 64);
#pragma unroll 4
  for (int it = 0; it < WR / RPI; ++it) {
    const int row = it * RPI + rr_;
    const int m = m0 + wm * WR + row;
    if (m >= M) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * SW + c4);
    const int n = ncol0 + c4;
    if constexpr (EPI == ESMDIFF_EPI_BF16 || EPI == ESMDIFF_EPI_SWIGLU_BF16) {
      const float sc = (EPI == ESMDIFF_EPI_BF16) ? alpha : 1.0f;
      uint2 p;
      p.x = pack_bf16x2(v[0] * sc, v[1] * sc);
      p.y = pack_bf16x2(v[2] * sc, v[3] * sc);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (int64_t)m * ldc + n) = p;
    } else if constexpr (EPI == ESMDIFF_EPI_RESID_F32) {
      float* o = reinterpret_cast<float*>(out) + (int64_t)m * ldc + n;
      f32x4 x = *reinterpret_cast<const f32x4*>(o);
      x[0] += v[0] * alpha; x[1] += v[1] * alpha; x[2] += v[2] * alpha; x[3] += v[3] * alpha;
      *reinterpret_cast<f32x4*>(o) = x;
    } else if constexpr (EPI == ESMDIFF_EPI_BIAS_GELU_BF16) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n);
      uint2 p;
      p.x = pack_bf16x2(gelu_erf(v[0] + bb[0]), gelu_erf(v[1] + bb[1]));
      p.y = pack_bf16x2(gelu_erf(v[2] + bb[2]), gelu_erf(v[3] + bb[3]));
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (int64_t)m * ldc + n) = p;
    } else {  // ESMDIFF_EPI_BIAS_F32
      if (n + 4 <= ldc) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n);
        f32x4 x;
        x[0] = v[0] + bb[0]; x[1] = v[1] + bb[1]; x[2] = v[2] + bb[2]; x[3] = v[3] + bb[3];
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (int64_t)m * ldc + n) = x;
      }
    }
  }
  (void)n_valid;
}

// split-K second pass: out = epi(sum_s partial[s]) ; one thread = 4 consecutive output columns of one row
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int S, int64_t pstride,
                                                            void* __restrict__ out, const float* __restrict__ bias,
                                                            int M, int N, int ldc, float alpha) {
  constexpr bool SWIGLU = EPI == ESMDIFF_EPI_SWIGLU_BF16;
  const int ow4 = (SWIGLU ? N / 2 : N) / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * ow4) return;
  const int m = (int)(idx / ow4), c = (int)(idx - (int64_t)m * ow4) * 4;
  const int n_in = SWIGLU ? (c >> 5) * 64 + (c & 31) : c;  // gate columns of the interleaved [32 gate | 32 up] blocks
  const float* p = partial + (int64_t)m * N + n_in;
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
  for (int s2 = 1; s2 < S; ++s2) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + s2 * pstride);
    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
  }
  if constexpr (SWIGLU) {
    f32x4 u = *reinterpret_cast<const f32x4*>(p + 32);
    for (int s2 = 1; s2 < S; ++s2) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + 32 + s2 * pstride);
      u[0] += t[0]; u[1] += t[1]; u[2] += t[2]; u[3] += t[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = silu_mul(v[e], u[e]);
  }
  const int n = c;
  if constexpr (EPI == ESMDIFF_EPI_BF16 || SWIGLU) {
    const float sc = (EPI == ESMDIFF_EPI_BF16) ? alpha : 1.0f;
    uint2 pk;
    pk.x = pack_bf16x2(v[0] * sc, v[1] * sc);
    pk.y = pack_bf16x2(v[2] * sc, v[3] * sc);
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (int64_t)m * ldc + n) = pk;
  } else if constexpr (EPI == ESMDIFF_EPI_RESID_F32) {
    float* o = reinterpret_cast<float*>(out) + (int64_t)m * ldc + n;
    f32x4 x = *reinterpret_cast<const f32x4*>(o);
    x[0] += v[0] * alpha; x[1] += v[1] * alpha; x[2] += v[2] * alpha; x[3] += v[3] * alpha;
    *reinterpret_cast<f32x4*>(o) = x;
  } else if constexpr (EPI == ESMDIFF_EPI_BIAS_GELU_BF16) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n);
    uint2 pk;
    pk.x = pack_bf16x2(gelu_erf(v[0] + bb[0]), gelu_erf(v[1] + bb[1]));
    pk.y = pack_bf16x2(gelu_erf(v[2] + bb[2]), gelu_erf(v[3] + bb[3]));
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (int64_t)m * ldc + n) = pk;
  } else {  // ESMDIFF_EPI_BIAS_F32
    if (n + 4 <= ldc) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n);
      f32x4 x;
      x[0] = v[0] + bb[0]; x[1] = v[1] + bb[1]; x[2] = v[2] + bb[2]; x[3] = v[3] + bb[3];
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (int64_t)m * ldc + n) = x;
    }
  }
}

// One launch of the 128-column-tile kernel.  mi: 32-row blocks per wave (tile height 64 * mi); S > 1: K slices into
// `partial` (S planes of pstride floats, row stride N), reduced by splitk_reduce_kernel unless keep_partials.
static hipError_t launch_tiles(const bf16_t* A, const bf16_t* W, void* out, const float* bias, int M, int N, int K,
                               int ldc, int n_valid, float alpha, int epilogue, hipStream_t stream, int mi, int S,
                               float* partial, int64_t pstride, bool keep_partials) {
  const int bm = 64 * mi;
  const int tiles_m = (M + bm - 1) / bm, tiles_n = N / BN;
  // four-stage ring with one workgroup per CU while everything is resident at once; two stages x two (or more)
  // workgroups per CU beyond that
  const bool four = tiles_m * tiles_n * S <= 256;
  dim3 grid(tiles_m * tiles_n * S), block(256);
  const int w_nt = tiles_m <= 8;          // few row tiles per weight tile: stream the weights non-temporally
  const int w_rs = K, w_ks = BK;          // W strides: row, K-tile, 128-row block (row-major [N, K])
  const int64_t w_ns = (int64_t)BN * K;
#define ED_LAUNCH(E, NST, MI)                                                                                       \
  do {                                                                                                              \
    constexpr int lds = NST * (64 * MI * BK * 2 + TILE_BYTES);                                                      \
    if (lds > 65536) {                                                                                              \
      const hipError_t a_ = ensure_dynamic_lds((const void*)gemm_bf16_kernel<E, NST, MI>, lds);                     \
      if (a_ != hipSuccess) return a_;                                                                              \
    }                                                                                                               \
    hipLaunchKernelGGL((gemm_bf16_kernel<E, NST, MI>), grid, block, lds, stream, A, W, out, bias, M, N, K, ldc,     \
                       n_valid, alpha, tiles_m, tiles_n, S, partial, pstride, w_rs, w_ks, w_ns, w_nt);              \
  } while (0)
#define ED_GEMM(E)                                                                                                  \
  do {                                                                                                              \
    if (four && mi == 2) ED_LAUNCH(E, 4, 2);                                                                        \
    else if (four) ED_LAUNCH(E, 4, 1);                                                                              \
    else if (mi == 2) ED_LAUNCH(E, 2, 2);                                                                           \
    else ED_LAUNCH(E, 2, 1);                                                                                        \
    if (S > 1 && !keep_partials) {                                                                                  \
      const int64_t n_thr = (int64_t)M * ((E == ESMDIFF_EPI_SWIGLU_BF16 ? N / 2 : N) / 4);                          \
      hipLaunchKernelGGL(splitk_reduce_kernel<E>, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, stream,      \
                         partial, S, pstride, out, bias, M, N, ldc, alpha);                                         \
    }                                                                                                               \
  } while (0)
  switch (epilogue) {
    case ESMDIFF_EPI_BF16: ED_GEMM(ESMDIFF_EPI_BF16); break;
    case ESMDIFF_EPI_RESID_F32: ED_GEMM(ESMDIFF_EPI_RESID_F32); break;
    case ESMDIFF_EPI_SWIGLU_BF16: ED_GEMM(ESMDIFF_EPI_SWIGLU_BF16); break;
    case ESMDIFF_EPI_BIAS_GELU_BF16: ED_GEMM(ESMDIFF_EPI_BIAS_GELU_BF16); break;
    case ESMDIFF_EPI_BIAS_F32: ED_GEMM(ESMDIFF_EPI_BIAS_F32); break;
    default: return hipErrorInvalidValue;
  }
#undef ED_GEMM
#undef ED_LAUNCH
  return hipGetLastError();
}

// Tile height of the small-M path: 64-row tiles while the 128-row grid (x K slices) would leave half the CUs without a
// workgroup.  The K order per output element is the same, so this never changes a result bit.
// ESMDIFF_GEMM_SMALL_BM=64|128 forces one (A/B runs).
static int small_tile_mi(int M, int N, int S) {
  static const int forced = [] {
    const char* e = ed_dbg_env("ESMDIFF_GEMM_SMALL_BM");
    return e ? atoi(e) : 0;
  }();
  if (forced == 64) return 1;
  if (forced == 128) return 2;
  return ((M + 127) / 128) * (N / BN) * S <= 128 ? 1 : 2;
}

// Which kernel a shape runs on — ONE function for the launcher and for the text esmdiff_describe_plan prints beside the results
// (two copies of the rule drift apart as soon as one is edited).
//   w4      the 256x256 four-wave kernel (gemm256w4.hip; one persistent workgroup per CU)
//   mi, S   else: this file's 128-column kernel with 64 * mi rows per tile and S K-slices (S > 1 needs the split-K workspace)
struct GemmChoice {
  bool w4;
  int mi, S;
};
static GemmChoice choose_gemm(int M, int N, int K, bool has_ws, size_t ws_floats) {
  // Tile selection: the 256x256 kernel once it has >= 128 tiles to hand out, the 128x128 kernel otherwise.  Measured crossovers
  // (us, 128 vs 256): N = 1536, K = 4096: M = 4128 67 / 81, M = 6192 109 / 87; N = 4608: M = 1032 28 / 37, M = 2064 48 / 42;
  // N = 8192: M = 1032 47 / 40.  ESMDIFF_GEMM_TILE=128|256 forces one (-DED_DEBUG builds: A/B benchmarking).
  // (Measured and rejected: splitting the rows so that 256-row tiles fill whole 256-CU rounds and the leftover
  // rows go through this kernel — 93 + 29 us apart, 138 us back to back, vs 132 us unsplit at N = K = 1536.)
  static const int forced = [] {
    const char* e = ed_dbg_env("ESMDIFF_GEMM_TILE");
    return e ? atoi(e) : 0;
  }();
  static const int min_tiles = [] {
    const char* e = ed_dbg_env("ESMDIFF_GEMM_256_MIN_TILES");
    return e ? atoi(e) : 128;
  }();
  // (r02, inside the two-stream forward: the N = 1536 linears of a 2 817 .. 5 376-row sub-batch — 72 .. 126 tiles — also
  // run better on the persistent 256x256 kernel, which leaves the other CUs to the other stream: B = 24 .. 40 at L_tok = 258
  // +1 .. 3 %; at 7 - 8 row tiles (QKV of a 1 548-row sub-batch, 126 tiles) the 128-column kernel still wins, B = 12 -3 %)
  // The 256x256 kernel walks K in pairs of 64-wide tiles and needs >= 6 of them; other K (none in ESM3-open, the decoder or the
  // encoder: 1536, 4096, 1280, 3584, 768) stay on this kernel.  (The r01 eight-wave 256x256 kernel that used to take those
  // shapes left the product in r05: scratch/gemm256_8wave_kernel.hip.txt of the round-5 tree, git commit 8ee0514.)
  const int t256m = (M + 255) / 256, t256 = t256m * (N / 256);
  if (N % 256 == 0 && K % (2 * BK) == 0 && K >= 6 * BK &&
      (forced == 256 || (forced == 0 && (t256 >= min_tiles || (t256 >= 72 && t256m >= 12)))))
    return GemmChoice{true, 2, 1};
  const int tiles_n = N / BN;
  // split-K factor, a function of (N, K) only: for K >= 2048 (FFN-down) the largest divisor of K/64 that is <= 8 and
  // keeps tiles_n * S <= 96.  Measured (us per launch, M = 240 / 774): FFN-down K = 4096 S = 8: 42 -> 17 / 41 -> 35;
  // but K = 1536 shapes lose at the larger M (out-proj S = 8: 21 -> 16 / 20 -> 27; QKV S = 2: 21 -> 20 / 22 -> 33) —
  // the f32 partials (S x M x N x 4 B, written and re-read) outweigh the shorter K loop — so they are not split.
  int S = 1;
  if (M < small_max_rows() && has_ws && K >= 2048) {
    const int nk = K / BK;
    for (int c = 8; c >= 2; --c)
      if (nk % c == 0 && tiles_n * c <= 96) {
        S = c;
        break;
      }
    if ((size_t)S * (size_t)((M + 127) / 128) * 128 * N > ws_floats) S = 1;
  }
  return GemmChoice{false, M < small_max_rows() ? small_tile_mi(M, N, S) : 2, S};
}

hipError_t launch_gemm_bf16(const bf16_t* A, const bf16_t* W, void* out, const float* bias, int M, int N,
                            int K, int ldc, int n_valid, float alpha, int epilogue, hipStream_t stream,
                            const GemmWorkspace* ws) {
  if (M <= 0) return hipSuccess;
  if (N % BN != 0 || K % BK != 0 || (ldc & 3)) return hipErrorInvalidValue;
  const bool has_ws = ws && ws->partial;
  const GemmChoice c = choose_gemm(M, N, K, has_ws, has_ws ? ws->partial_floats : 0);
  if (c.w4) return launch_gemm256w4_bf16(A, W, out, bias, M, N, K, ldc, alpha, epilogue, stream);
  const int64_t pstride = (int64_t)((M + 127) / 128) * 128 * N;
  return launch_tiles(A, W, out, bias, M, N, K, ldc, n_valid, alpha, epilogue, stream, c.mi, c.S,
                      c.S > 1 ? ws->partial : nullptr, pstride, false);
}

// Which kernel launch_gemm_bf16 picks for a shape, as text (esmdiff_describe_plan): "256x256w4" or "128x<rows>[/S<k-slices>]".
// ws_floats: the size of the split-K workspace the launch will be handed (0: none).
void describe_gemm(int M, int N, int K, size_t ws_floats, char* out, size_t cap) {
  const GemmChoice c = choose_gemm(M, N, K, ws_floats > 0, ws_floats);
  if (c.w4) snprintf(out, cap, "256x256w4");
  else if (c.S > 1) snprintf(out, cap, "128x%d/S%d", 64 * c.mi, c.S);
  else snprintf(out, cap, "128x%d", 64 * c.mi);
}

// ---- residual-branch linears of the small-batch path ------------------------------------------------------------
// Below small_max_rows() (1 152) rows the out-projection and FFN-down products are left as S raw f32 K-slice planes; the add+LayerNorm
// kernel that consumes them (norm.hip: launch_add_partials_layernorm_bf16) sums the planes in the order s = 0..S-1,
// scales and adds them to the residual stream — so no reduce kernel runs and the K = 1536 out-projection can be sliced
// as well (with a separate reduce pass its slices did not pay, see above).  S is a function of (N, K) only: the
// largest divisor of K/64 <= 8 that leaves every slice >= 6 K-tiles and tiles_n * S <= 96 (N = 1536: K = 1536 -> 4,
// K = 4096 -> 8).  ESMDIFF_GEMM_PSPLIT=<S> overrides (A/B runs; must divide K/64).
int gemm_partial_splits(int N, int K) {
  static const int forced = [] {
    const char* e = ed_dbg_env("ESMDIFF_GEMM_PSPLIT");
    return e ? atoi(e) : 0;
  }();
  const int nk = K / BK, tiles_n = N / BN;
  if (forced > 0 && nk % forced == 0) return forced;
  for (int c = 8; c >= 2; --c)
    if (nk % c == 0 && nk / c >= 6 && tiles_n * c <= 96) return c;
  return 1;
}

hipError_t launch_gemm_partials(const bf16_t* A, const bf16_t* W, const GemmWorkspace* ws, int M, int N, int K,
                                hipStream_t stream, GemmPartials* res) {
  if (M <= 0 || M >= small_max_rows() || !ws || !ws->partial || !res || N % BN != 0 || K % BK != 0) return hipErrorInvalidValue;
  const int S = gemm_partial_splits(N, K);
  const int64_t pstride = (int64_t)((M + 127) / 128) * 128 * N;
  if ((size_t)S * pstride > ws->partial_floats) return hipErrorInvalidValue;
  *res = GemmPartials{ws->partial, S, pstride};
  // (S = 1 still goes through `partial`: one raw f32 plane)
  return launch_tiles(A, W, nullptr, nullptr, M, N, K, N, N, 1.f, ESMDIFF_EPI_BF16, stream, small_tile_mi(M, N, S), S,
                      ws->partial, pstride, true);
}

}  // namespace ed
