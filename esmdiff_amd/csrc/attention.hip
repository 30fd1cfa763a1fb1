// attention.hip — non-causal multi-head attention, head dim 64, flash-style (gfx950).
//
// Replaces F.scaled_dot_product_attention(q, k, v) as esm's MultiHeadAttention calls it with
// sequence_id=None (the reference passes None: /root/reference/slm/models/net.py:468), i.e. no mask.
// The (B,24,L,L) score tensor is never materialised: each wave owns 32 queries and streams the keys
// in tiles of 64 through LDS with an online softmax, so L = 1026 (BASELINE config 4; K/V per head =
// 263 KB > 160 KB LDS) needs no special casing — only the number of tiles changes.
//
// Workgroup = W waves (1..10; default min(4, ceil(L/32)), ESMDIFF_ATTN_WAVES overrides) = 32 W queries of one (batch,
// head); every wave owns 32 queries and all of them share the staged K/V tiles.  Measured at L_tok = 258 (9 query waves,
// which W = 4 cuts into 4 + 4 + 1, the third workgroup staging five K/V tiles for two queries) in ms of attention per
// forward, one box: W = 4 / 3 waves per SIMD (the r01 kernel) 5.33, W = 3 (no idle wave) 5.30-5.32, W = 5 6.46, W = 9
// (one workgroup per head, each K/V tile staged once) 6.20, W = 4 / 4 waves per SIMD 5.11 (kept); L_tok = 1026: 14.3 /
// 14.7-15.2 / 18.2 / 18.2 / 13.6.  So idle waves and repeated staging are NOT what the kernel waits for (a parked wave
// costs a slot, not cycles); wider workgroups lose to the 9-wave barrier.  What bounds it is the serial S -> softmax -> PV
// chain inside each wave (MFMA pipe ~19 % busy, VALU ~45 %).
// Per 64-key tile:
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
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "ed_half.h"
#include "kernels.h"

namespace ed {

typedef ed_half8 bf16x8;   // the TU's 16-bit operand type (ed_half.h: bf16, or f16 in the ed16 build)
typedef ed_half4 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int KV_TILE = 64;
constexpr int KV_BYTES = KV_TILE * 128;  // 8 KiB per operand per stage

__device__ __forceinline__ void glds16a(const void* gsrc, void* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) { return ed_pack2_bounded(a, b); }  // v_cvt_pk_{bf16,f16}_f32, nearest even
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // bare v_exp_f32

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x4 lds_read_tr16(const char* p) {  // ds_read_b64_tr_b16
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  bf16x4 r;
  __builtin_memcpy(&r, &v, 8);
  return r;
}

// Two register budgets of the same body: 3 waves per SIMD (137 VGPRs, no spill) and 4 waves per SIMD (128 VGPRs, a
// handful of spilled address registers outside the tile loop); the launcher picks (ESMDIFF_ATTN_OCC overrides).
// Ablation builds (-DED_ATTN_ABL=<bits>, wrong results by construction): 1 no v_exp_f32, 2 K/V staged once (tile 0 is
// reused: no LDS-DMA in the loop), 4 no vmcnt wait / barrier per tile, 8 no P·V MFMAs.
#ifndef ED_ATTN_ABL
#define ED_ATTN_ABL 0
#endif
#ifdef ED_ATTN_TRACE
__device__ unsigned long long g_attn_trace[512 * 4 * 32];
extern "C" int esmdiff_debug_attn_trace(unsigned long long* out_host) {   // debug builds only (scratch/attn_trace.py)
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_attn_trace), sizeof(g_attn_trace));
}
#endif
#define ED_ATTN_NAME attention_kernel_occ3
#define ED_ATTN_WPE 3
#include "attention_kernel.inc"
#undef ED_ATTN_NAME
#undef ED_ATTN_WPE
#define ED_ATTN_NAME attention_kernel_occ4
#define ED_ATTN_WPE 4
#include "attention_kernel.inc"
#undef ED_ATTN_NAME
#undef ED_ATTN_WPE

// (ESMDIFF_ATTN_WAVES=-1 only: the occupancy model that was tried and measured slower, see the header)
// Workgroup width for L tokens: nw = ceil(L/32) query waves per (batch, head) are cut into nWG workgroups of W waves.  A CU
// holds k = min(4 occ / W, LDS / 32 KiB) of them (occ = waves per SIMD of the build), of which nw / (nWG W) of the waves
// are working: pick the W that keeps the most working waves resident (ties: the widest, it stages each K/V tile fewer
// times).  occ 4: L = 258 -> 3 (3 workgroups of 3, 15 working waves per CU; the old fixed 4 at occ 3 gave 9),
// L = 1026 -> 3 (11 workgroups, 15), L = 60 -> 2.
static int attention_waves(int L, int occ) {
  const int nw = (L + 31) / 32, max_wg = L <= KV_TILE ? 10 : 5;  // LDS: 16 / 32 KiB per workgroup of 160
  int best_w = 1, best_score = -1;
  for (int W = 1; W <= 10; ++W) {
    if (W > nw && W > 1) break;
    const int nwg = (nw + W - 1) / W;
    const int k = std::min(4 * occ / W, max_wg);
    const int score = k * nw * 1000 / nwg;  // working waves per CU x 1000
    if (score >= best_score) {
      best_score = score;
      best_w = W;
    }
  }
  return best_w;
}

hipError_t launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* qkv, bf16_t* ctx, int B, int L,
                            int H, hipStream_t stream) {
  if (B <= 0 || L <= 0) return hipSuccess;
  static const int forced = [] {
    const char* e = ed_dbg_env("ESMDIFF_ATTN_WAVES");
    return e ? atoi(e) : 0;
  }();
  static const int occ = [] {
    const char* e = ed_dbg_env("ESMDIFF_ATTN_OCC");
    return e && atoi(e) == 3 ? 3 : 4;
  }();
  const int W = forced > 0 ? std::min(forced, 10) : (forced < 0 ? attention_waves(L, occ) : std::min(4, (L + 31) / 32));
  const int nw = (L + 31) / 32;
  const int nqb = (nw + W - 1) / W, BH = B * H;
  dim3 grid(8 * nqb * ((BH + 7) / 8)), block(64 * W);
  const size_t lds = (L <= KV_TILE ? 1 : 2) * 2 * KV_BYTES;
  if (occ == 3) hipLaunchKernelGGL(attention_kernel_occ3, grid, block, lds, stream, q, k, qkv, ctx, L, H, BH, nqb);
  else hipLaunchKernelGGL(attention_kernel_occ4, grid, block, lds, stream, q, k, qkv, ctx, L, H, BH, nqb);
  return hipGetLastError();
}

}  // namespace ed
