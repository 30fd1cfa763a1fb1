// gemm256.hip — the large-M bf16 MFMA GEMM: 256x256x64 tiles, 8 waves, register-prefetched pipeline (gfx950).
//
// Same contract and epilogues as gemm.hip (out = epi(A[M,K] · W[N,K]^T)); used when N % 256 == 0 and M is
// large (the four per-block linears of the ESM3 stack at B*L = 25 800 rows).  What changes is the schedule:
//
//   workgroup  512 threads = 8 waves as 2 (M) x 4 (N); one workgroup per CU (128 KiB LDS, <=256 VGPR/wave,
//              two waves per SIMD).  Wave tile 128 x 64 = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16.
//   LDS        2 stages x [A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255], each 128 rows x 128 B
//              (16 KiB), filled by LDS-DMA (global_load_lds_dwordx4), source-side XOR swizzle as in gemm.hip.
//   K-tile     8 groups of 4 MFMAs per wave; a group is one 64x32 quadrant of the wave tile over HALF of the
//              K-tile (two k-steps of 16), visited in a snake so that consecutive groups share an operand:
//                g1 C00+=A0h0·B0h0 | g2 C01+=A0h0·B1h0 | g3 C11+=A1h0·B1h0 | g4 C10+=A1h0·B0h0
//                g5 C10+=A1h1·B0h1 | g6 C11+=A1h1·B1h1 | g7 C01+=A0h1·B1h1 | g8 C00+=A0h1·B0h1
//              Operand half-sets (4 x ds_read_b128 for A, 2 for B) are read TWO groups before their first use
//              (<= 64 operand VGPRs live + 128 accumulators at 2 waves/SIMD).  The reads are inline-asm
//              ds_read_b128 and the waits are counted s_waitcnt lgkmcnt(N): hipcc would turn every wait into
//              lgkmcnt(0) while an LDS-DMA is pending.
//              One barrier per K-tile, after g7: every wave has then finished reading this stage and K-tile
//              t+1 has landed.  The 8 LDS-DMA instructions per wave that refill the stage with K-tile t+2 are
//              spread over g8, g1', g2', g3' (2 each): measured with s_memtime, the CU's address path takes
//              ~25-45 cycles per 1-KiB LDS-DMA, and a burst of 64 of them right after the barrier stalled the
//              younger four waves for 1.2-2k cycles in instruction issue.
//   tiles      persistent workgroups (one per CU) walk tiles b, b+G, ...; the K-tile LDS-DMA stream runs across the
//              tile boundary (K-tile 0 of the next tile lands in stage 0 during the last K-tile, K-tile 1 is issued
//              into stage 1 right after the K loop, under the epilogue), so a tile pays neither the DMA latency of
//              its first K-tiles nor a workgroup relaunch, and there is no barrier between two tiles.
//   epilogue   register-direct: the MFMAs compute the TRANSPOSED 32x32 blocks (W fragment as first operand), which
//              gives every lane 4 consecutive output columns per register group; v_permlane32_swap widens that to 8
//              (one 16-byte bf16 store per lane, 32 contiguous bytes per row).  No LDS staging, no barriers.
//   raster     XCD-contiguous, 8(M) x 4(N) super-tiles: the 32 tiles resident on one XCD share 8 A panels and
//              4 W panels in that XCD's L2.
#include <stdlib.h>

#include <type_traits>

#include "ed_half.h"
#include "kernels.h"

namespace ed {

typedef ed_half8 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace g256 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;     // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;  // 64 KiB
constexpr int GROUP_M = 8;

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
#define ED_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
// Ablation builds: -DED_ABL=<bits> compiles parts of the kernel out (results are then wrong by construction; only the
// times mean something).  Bits: 1 no fragment reads, 2 no LDS-DMA in the main loop, 4 no stores, 8 no vmcnt wait, 16 no
// barrier, 64 no MFMA, 128 no lgkmcnt waits, 256 every workgroup streams the panels of tile (0,0) (perfect L2 locality).
// Compile-time on purpose: the r01 run-time switch (a kernel argument tested around every statement) made the debug
// build itself 1.5x slower than the product build, so its differences said little about the product kernel.
// -DED_GEMM_DEBUG adds s_memtime stamps of one workgroup into the `bias` buffer of an EPI_BF16 launch.
#ifndef ED_ABL
#define ED_ABL 0
#endif
#define ED_DBG(bit) (((ED_ABL) & (bit)) != 0)

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                         void* __restrict__ out, const float* __restrict__ bias,
                                                         int M, int N, int K, int ldc, float alpha, int tiles_m,
                                                         int tiles_n, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x 64 KiB

  // Persistent workgroups: workgroup b (on XCD b % 8) walks the virtual tile indices b, b + G, b + 2G, ... (G = grid
  // size, a multiple of 8, so every index it visits maps to its own XCD's contiguous run of tiles).
  const int n_tiles = tiles_m * tiles_n, bid = blockIdx.x;
  auto tile_origin = [&](int vt, int& m0_, int& n0_) {
    const int xcd = vt & 7, qq = n_tiles >> 3, rr = n_tiles & 7;
    const int lin = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (vt >> 3);
    const int per_group = GROUP_M * tiles_n;
    const int grp = lin / per_group, in_grp = lin - grp * per_group;
    const int gm0 = grp * GROUP_M;
    const int gsz = min(GROUP_M, tiles_m - gm0);
    m0_ = (gm0 + in_grp % gsz) * BM;
    n0_ = (in_grp / gsz) * BN;
  };
  int m0, n0;
  tile_origin(bid, m0, n0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- LDS-DMA sources: half-tile = 128 rows = 16 wave-instructions of 8 rows; wave issues i = 0,1 ----
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7);
  uint32_t a_off[2][2], w_off[2][2];      // this tile
  uint32_t a_offn[2][2], w_offn[2][2];    // next tile of this workgroup (its K-tile 0 is prefetched across the boundary)
  auto set_offsets = [&](uint32_t (&ao)[2][2], uint32_t (&wo)[2][2], int m0_, int n0_) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = h * 128 + (i * 8 + wave) * 8 + srow;
        const int am = min((ED_DBG(256) ? 0 : m0_) + r, M - 1);
        ao[h][i] = (uint32_t)(((int64_t)am * K + schunk * 8) * 2);
        wo[h][i] = (uint32_t)(((int64_t)((ED_DBG(256) ? 0 : n0_) + r) * K + schunk * 8) * 2);
      }
  };
  set_offsets(a_off, w_off, m0, n0);
  constexpr int kstride = BK * 2;
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Wb = reinterpret_cast<const char*>(W);
  // main-loop LDS-DMA: scalar base (operand + K-tile offset) + 32-bit per-lane offset, M0 = wave-uniform LDS
  // destination.  Written as inline asm because hipcc otherwise materialises a 64-bit per-lane address
  // (v_lshl_add_u64 + a VGPR pair) for every instruction.
  const uint32_t lds_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  auto glds16s = [&](const char* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(voff), "s"(sbase)
                 : "memory");
  };
  // piece = 128 rows of one operand (2 instructions per wave).  v is a K-tile index of the CURRENT tile, or nk + 0 for
  // K-tile 0 of the NEXT tile (xnext), which keeps the DMA stream running across the tile boundary.
  bool xnext = false;
  int nk_ = K / BK;
  // -DED_KROT=1 (experiment): tile column tn starts its K loop at K-tile (tn & 3) * nk / 4 and wraps, so that the four
  // workgroups of an XCD that share an A panel do not ask L2 for the same lines at the same moment (a function of the
  // COLUMN tile only: a row's result must not depend on where its sample sits in the batch)
#ifndef ED_KROT
#define ED_KROT 0
#endif
  int rot = 0, rotn = 0;
  auto krot_of = [&](int n0_) { return ED_KROT ? ((n0_ / BN) & 3) * (nk_ >> 2) : 0; };
  auto kt_of = [&](int v, int r) {
    int kk = v + r;
    return kk >= nk_ ? kk - nk_ : kk;
  };
  rot = krot_of(n0);
  auto issue_Ah = [&](int h, int buf, int v) {
    if (v < nk_) {
      const char* sb = Ab + (size_t)kt_of(v, rot) * kstride;
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16s(sb, a_off[h][i], lds_base + (h * 2 + buf) * HALF_BYTES + (i * 8 + wave) * 1024);
    } else if (xnext && v == nk_) {
      const char* sb = Ab + (size_t)rotn * kstride;
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16s(sb, a_offn[h][i], lds_base + (h * 2 + buf) * HALF_BYTES + (i * 8 + wave) * 1024);
    }
  };
  auto issue_Wh = [&](int h, int buf, int v) {
    if (v < nk_) {
      const char* sb = Wb + (size_t)kt_of(v, rot) * kstride;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        glds16s(sb, w_off[h][i], lds_base + ((2 + h) * 2 + buf) * HALF_BYTES + (i * 8 + wave) * 1024);
    } else if (xnext && v == nk_) {
      const char* sb = Wb + (size_t)rotn * kstride;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        glds16s(sb, w_offn[h][i], lds_base + ((2 + h) * 2 + buf) * HALF_BYTES + (i * 8 + wave) * 1024);
    }
  };
  auto issue_A = [&](int buf, int kt) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        glds16(Ab + (size_t)kt_of(kt, rot) * kstride + a_off[h][i], smem + (h * 2 + buf) * HALF_BYTES + (i * 8 + wave) * 1024);
  };
  auto issue_W = [&](int buf, int kt) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        glds16(Wb + (size_t)kt_of(kt, rot) * kstride + w_off[h][i],
               smem + ((2 + h) * 2 + buf) * HALF_BYTES + (i * 8 + wave) * 1024);
  };

  // ---- fragment read addressing -----------------------------------------------------------------
  // Fragment reads are inline-asm ds_read_b128 so that the waits can be COUNTED: hipcc turns every
  // lgkmcnt wait into lgkmcnt(0) while an LDS-DMA is pending (it models global_load_lds as a FLAT access
  // that may touch LDS), which would make each group wait for the prefetch it has just issued.
  const int frow = lane & 31, khalf = lane >> 5, fsw = (frow >> 1) & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  uint32_t offA[4], offB[4];  // stage-0 byte addresses of this lane's 16-byte fragment piece, per k-step
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint32_t lo = frow * 128 + (((ks * 2 + khalf) ^ fsw) << 4);
    offA[ks] = lds0 + (wm * 2) * HALF_BYTES + lo;
    offB[ks] = lds0 + ((2 + (wn >> 1)) * 2) * HALF_BYTES + (wn & 1) * (64 * 128) + lo;
  }
#define ED_DSR(dst, addr, imm)                                                                       \
  do {                                                                                               \
    if (!ED_DBG(1)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm)); \
    else asm volatile("" : "=v"(dst));                                                               \
  } while (0)
  // half-sets: A[f][j] = rows (mh*64 + f*32 ..), k-step 2*h + j ; B[j] = cols (nh*32 ..), k-step 2*h + j
  auto read_A = [&](bf16x8 (&a)[2][2], auto P, auto MH, auto H) {
    constexpr int mh = decltype(MH)::value, h = decltype(H)::value;
    constexpr int st = decltype(P)::value * HALF_BYTES;
    const uint32_t a0_ = offA[2 * h], a1_ = offA[2 * h + 1];
    ED_DSR(a[0][0], a0_, st + (mh * 64) * 128);
    ED_DSR(a[0][1], a1_, st + (mh * 64) * 128);
    ED_DSR(a[1][0], a0_, st + (mh * 64 + 32) * 128);
    ED_DSR(a[1][1], a1_, st + (mh * 64 + 32) * 128);
  };
  auto read_B = [&](bf16x8 (&b)[2], auto P, auto NH, auto H) {
    constexpr int nh = decltype(NH)::value, h = decltype(H)::value;
    constexpr int st = decltype(P)::value * HALF_BYTES;
    const uint32_t b0_ = offB[2 * h], b1_ = offB[2 * h + 1];
    ED_DSR(b[0], b0_, st + (nh * 32) * 128);
    ED_DSR(b[1], b1_, st + (nh * 32) * 128);
  };
  // counted wait that also makes the retired registers' readiness visible to the compiler ("+v")
#define ED_WAIT_A(N, a)                                                                                       \
  do {                                                                                                        \
    if (!ED_DBG(128)) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]));   \
    ED_PHASE_FENCE();                                                                                         \
  } while (0)
#define ED_WAIT_B(N, b)                                                   \
  do {                                                                    \
    if (!ED_DBG(128)) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(b[0]), "+v"(b[1]));   \
    ED_PHASE_FENCE();                                                     \
  } while (0)
#define ED_WAIT_AB(N, a, b)                                                                                  \
  do {                                                                                                       \
    if (!ED_DBG(128)) asm volatile("s_waitcnt lgkmcnt(" #N ")"                                               \
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[0]), "+v"(b[1]));      \
    ED_PHASE_FENCE();                                                                                        \
  } while (0)

  f32x16 acc[2][2][2];  // [mh][f][nh]

  auto mma = [&](int mh, int nh, const bf16x8 (&a)[2][2], const bf16x8 (&b)[2]) {
    if (ED_DBG(64)) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int f = 0; f < 2; ++f)
        acc[mh][f][nh] = ED_MFMA_32x32x16(b[j], a[f][j], acc[mh][f][nh]);  // C^T: see epilogue
    __builtin_amdgcn_s_setprio(0);
    ED_PHASE_FENCE();
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  bf16x8 A0a[2][2], A0b[2][2], A1a[2][2], A1b[2][2], B0a[2], B0b[2], B1a[2], B1b[2];
  const int nk = nk_;

  // One K-tile at stage P.  Read sets are issued two groups ahead of their first use; the LGKM queue holds
  // only these ds_reads, in order, so each wait is "all but the N youngest".
  //   queue on entry: [A0a 4][B0a 2][B1a 2]
#ifdef ED_GEMM_DEBUG
  unsigned long long* trace = nullptr;
  if constexpr (EPI == ESMDIFF_EPI_BF16) {
    if (bias != nullptr && blockIdx.x == 40) trace = reinterpret_cast<unsigned long long*>(const_cast<float*>(bias));
  }
#define ED_STAMP(i)                                                                     \
  do {                                                                                  \
    if (trace && t < 32 && lane == 0) trace[(wave * 32 + t) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define ED_STAMP(i) do { } while (0)
#endif
  auto ktile = [&](int t, auto P) {
    using Q = std::integral_constant<int, 1 - decltype(P)::value>;
    ED_STAMP(0);
    // g1  C00 += A0h0·B0h0   (+ second quarter of the refill of the other stage with K-tile t+1)
    if (t >= 1 && !ED_DBG(2)) issue_Wh(1, Q::value, t + 1);
    read_A(A1a, P, I1{}, I0{});            // queue 12
    ED_WAIT_AB(6, A0a, B0a);
    mma(0, 0, A0a, B0a);
    // g2  C01 += A0h0·B1h0
    if (t >= 1 && !ED_DBG(2)) issue_Ah(0, Q::value, t + 1);
    read_A(A1b, P, I1{}, I1{});            // [B1a 2][A1a 4][A1b 4]
    ED_WAIT_B(8, B1a);
    mma(0, 1, A0a, B1a);
    // g3  C11 += A1h0·B1h0
    if (t >= 1 && !ED_DBG(2)) issue_Ah(1, Q::value, t + 1);
    read_B(B0b, P, I0{}, I1{});            // [A1a 4][A1b 4][B0b 2]
    ED_WAIT_A(6, A1a);
    mma(1, 1, A1a, B1a);
    // g4  C10 += A1h0·B0h0
    read_B(B1b, P, I1{}, I1{});            // [A1b 4][B0b 2][B1b 2]
    mma(1, 0, A1a, B0a);
    ED_STAMP(1);
    // g5  C10 += A1h1·B0h1
    read_A(A0b, P, I0{}, I1{});            // [A1b 4][B0b 2][B1b 2][A0b 4]
    ED_WAIT_AB(6, A1b, B0b);
    mma(1, 0, A1b, B0b);
    // g6  C11 += A1h1·B1h1
    ED_WAIT_B(4, B1b);                                     // [B1b 2][A0b 4]
    mma(1, 1, A1b, B1b);
    // g7  C01 += A0h1·B1h1
    ED_WAIT_A(0, A0b);
    mma(0, 1, A0b, B1b);
    ED_STAMP(2);
    if (!ED_DBG(8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K-tile t+1 (issued one K-tile ago) has landed
    ED_STAMP(3);
    if (!ED_DBG(16)) __builtin_amdgcn_s_barrier();    // ... for every wave; and stage P is no longer read
    ED_STAMP(4);
    ED_PHASE_FENCE();
    // g8  C00 += A0h1·B0h1
    if (!ED_DBG(2)) issue_Wh(0, decltype(P)::value, t + 2);
    if (t + 1 < nk) {
      read_A(A0a, Q{}, I0{}, I0{});
      read_B(B0a, Q{}, I0{}, I0{});
      read_B(B1a, Q{}, I1{}, I0{});
    }
    ED_STAMP(5);
    ED_PHASE_FENCE();
    mma(0, 0, A0b, B0b);
    ED_STAMP(6);
  };

  // ---- tile loop -----------------------------------------------------------------------------------------
  const bool can_xprefetch = (nk % 2 == 0);  // the last K-tile must sit in stage 1 so that stage 0 can take the next tile
  bool have_k0 = false;                      // K-tile 0 of the current tile already landed in stage 0
#ifdef ED_GEMM_DEBUG
  int tileno = 0;
#define ED_TSTAMP(i)                                                                                              \
  do {                                                                                                            \
    if (trace && tileno < 8 && lane == 0) trace[2048 + (wave * 8 + tileno) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define ED_TSTAMP(i) do { } while (0)
#endif
  // (A 0-3 x 2k-cycle start skew between neighbouring workgroups paid +1 % while the epilogue went through LDS slabs and
  // tile-boundary barriers; with the register-direct epilogue it measures -0.5 % and is gone.)
  for (int vt = bid; vt < n_tiles; vt += gridDim.x) {
    const int vtn = vt + gridDim.x;
    ED_TSTAMP(0);
    xnext = can_xprefetch && vtn < n_tiles;
    int m0n = 0, n0n = 0;
    if (xnext) {
      tile_origin(vtn, m0n, n0n);
      set_offsets(a_offn, w_offn, m0n, n0n);
      rotn = krot_of(n0n);
    }
    if (!have_k0) {
      issue_A(0, 0);
      issue_W(0, 0);
      if (nk > 1) {
        issue_A(1, 1);
        issue_W(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }  // else: K-tiles 0 and 1 were issued by the previous tile of this workgroup (see its epilogue)
    ED_PHASE_FENCE();
    read_A(A0a, I0{}, I0{}, I0{});
    read_B(B0a, I0{}, I0{}, I0{});
    read_B(B1a, I0{}, I1{}, I0{});
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) (&acc[0][0][0])[i][r] = 0.0f;
    ED_PHASE_FENCE();
    ED_TSTAMP(1);

    int t = 0;
    for (; t + 1 < nk; t += 2) {
      ktile(t, I0{});
      ktile(t + 1, I1{});
    }
    if (t < nk) ktile(t, I0{});
    ED_TSTAMP(2);

    ED_TSTAMP(3);
    // Every wave is past the barrier of the last K-tile and reads no more LDS for this tile: stage 1 is free, so
    // K-tile 1 of the next tile starts streaming now, under the epilogue.
    if (xnext && nk > 1) {
      const size_t k1 = (size_t)kt_of(1, rotn) * kstride;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          glds16s(Ab + k1, a_offn[h][i], lds_base + (h * 2 + 1) * HALF_BYTES + (i * 8 + wave) * 1024);
          glds16s(Wb + k1, w_offn[h][i], lds_base + ((2 + h) * 2 + 1) * HALF_BYTES + (i * 8 + wave) * 1024);
        }
    }

    // ---- epilogue: straight from the accumulators, no LDS.  The MFMAs take the W fragment as their first operand,
    // so each 32x32 accumulator is the TRANSPOSED output block: a lane holds one output row (m = lane & 31) and
    // every group of four registers is four consecutive columns, n = 8*(r>>2) + 4*(lane>>5) + (r&3).  For the bf16
    // outputs v_permlane32_swap trades packed register groups between lane l and l+32 so that a lane owns 8
    // consecutive columns = one 16-byte store; lanes l / l+32 then cover 32 contiguous bytes of a row.
    const int lrow = lane & 31, lhi = lane >> 5;
    auto swap_halves = [](uint32_t& lo_keep, uint32_t& hi_keep) {  // lo_keep[32:63] <-> hi_keep[0:31]
      const auto r = __builtin_amdgcn_permlane32_swap(lo_keep, hi_keep, false, false);
      lo_keep = r[0];
      hi_keep = r[1];
    };
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int m = m0 + wm * 128 + mh * 64 + f * 32 + lrow;
        const bool live = m < M && !ED_DBG(4);
        if constexpr (EPI == ESMDIFF_EPI_SWIGLU_BF16) {
          bf16_t* orow = reinterpret_cast<bf16_t*>(out) + (int64_t)m * ldc + (n0 + wn * 64) / 2 + lhi * 8;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float h[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float g = acc[mh][f][0][gp * 8 + i], u = acc[mh][f][1][gp * 8 + i];
              h[i] = silu_mul(g, u);
            }
            uint32_t p0 = pack_bf16x2(h[0], h[1]), p1 = pack_bf16x2(h[2], h[3]);
            uint32_t q0 = pack_bf16x2(h[4], h[5]), q1 = pack_bf16x2(h[6], h[7]);
            swap_halves(p0, q0);
            swap_halves(p1, q1);
            if (live) *reinterpret_cast<uint4*>(orow + gp * 16) = make_uint4(p0, p1, q0, q1);
          }
        } else if constexpr (EPI == ESMDIFF_EPI_BF16 || EPI == ESMDIFF_EPI_BIAS_GELU_BF16) {
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            const int nb = n0 + wn * 64 + nh * 32;
            bf16_t* orow = reinterpret_cast<bf16_t*>(out) + (int64_t)m * ldc + nb + lhi * 8;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              float h[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) h[i] = acc[mh][f][nh][gp * 8 + i];
              if constexpr (EPI == ESMDIFF_EPI_BF16) {
#pragma unroll
                for (int i = 0; i < 8; ++i) h[i] *= alpha;
              } else {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + nb + gp * 16 + lhi * 4);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + nb + gp * 16 + 8 + lhi * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  h[i] = gelu_erf(h[i] + b0[i]);
                  h[4 + i] = gelu_erf(h[4 + i] + b1[i]);
                }
              }
              uint32_t p0 = pack_bf16x2(h[0], h[1]), p1 = pack_bf16x2(h[2], h[3]);
              uint32_t q0 = pack_bf16x2(h[4], h[5]), q1 = pack_bf16x2(h[6], h[7]);
              swap_halves(p0, q0);
              swap_halves(p1, q1);
              if (live) *reinterpret_cast<uint4*>(orow + gp * 16) = make_uint4(p0, p1, q0, q1);
            }
          }
        } else {  // f32 outputs: 16 bytes per lane per register group, 32 contiguous bytes per row per store
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int n = n0 + wn * 64 + nh * 32 + g * 8 + lhi * 4;
              f32x4 v;
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = acc[mh][f][nh][g * 4 + i];
              if (!live) continue;
              float* o = reinterpret_cast<float*>(out) + (int64_t)m * ldc + n;
              if constexpr (EPI == ESMDIFF_EPI_RESID_F32) {
                f32x4 x = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] += v[i] * alpha;
                *reinterpret_cast<f32x4*>(o) = x;
              } else {
                if (n + 4 <= ldc) {
                  const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                  for (int i = 0; i < 4; ++i) v[i] += bb[i];
                  *reinterpret_cast<f32x4*>(o) = v;
                }
              }
            }
        }
      }

    ED_TSTAMP(4);
#ifdef ED_GEMM_DEBUG
    ++tileno;
#endif
    // ---- next tile of this workgroup ---------------------------------------------------------------
    if (vtn < n_tiles) {
      have_k0 = xnext;
      if (xnext) {
        m0 = m0n;
        n0 = n0n;
        rot = rotn;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            a_off[h][i] = a_offn[h][i];
            w_off[h][i] = w_offn[h][i];
          }
      } else {
        tile_origin(vtn, m0, n0);
        set_offsets(a_off, w_off, m0, n0);
        rot = krot_of(n0);
      }
    }
  }
}
}  // namespace g256

hipError_t launch_gemm256_bf16(const bf16_t* A, const bf16_t* W, void* out, const float* bias, int M, int N,
                               int K, int ldc, float alpha, int epilogue, hipStream_t stream) {
  using namespace g256;
  if (M <= 0) return hipSuccess;
  if (N % BN != 0 || K % BK != 0 || (ldc & 3)) return hipErrorInvalidValue;
  // default: the four-wave kernel (gemm256w4.hip: 128x128 wave tiles, hand-placed main loop; 4-5 % faster on the block
  // linears, r02); this eight-wave kernel stays for odd K-tile counts and for A/B runs (ESMDIFF_GEMM_W4=0)
  static const int w4 = [] {
    const char* e = getenv("ESMDIFF_GEMM_W4");
    return e ? atoi(e) : 1;
  }();
  if (w4 && K % (2 * BK) == 0 && K >= 6 * BK) return launch_gemm256w4_bf16(A, W, out, bias, M, N, K, ldc, alpha, epilogue, stream);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
  // persistent grid: one workgroup per CU (LDS allows no more), each walking tiles b, b + G, ...  G must be a
  // multiple of 8 (XCD-contiguous tile runs).  ESMDIFF_GEMM_PERSIST=0 launches one workgroup per tile instead.
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? (n / 8) * 8 : 8;
  }();
  static const int persist = [] {
    const char* e = getenv("ESMDIFF_GEMM_PERSIST");
    return e ? atoi(e) : 1;
  }();
  const int n_tiles = tiles_m * tiles_n;
  dim3 grid(persist ? (n_tiles < n_cu ? n_tiles : n_cu) : n_tiles), block(512);
  const size_t lds = 2 * STAGE_BYTES;
  static const int dbg = [] {
    const char* e = getenv("ESMDIFF_GEMM_DBG");
    return e ? atoi(e) : 0;
  }();
#define ED_GEMM(E)                                                                                         \
  do {                                                                                                     \
    {                                                                                                      \
      const hipError_t a_ = ensure_dynamic_lds((const void*)gemm256_kernel<E>, (int)lds);                  \
      if (a_ != hipSuccess) return a_;                                                                     \
    }                                                                                                      \
    hipLaunchKernelGGL(gemm256_kernel<E>, grid, block, lds, stream, A, W, out, bias, M, N, K, ldc, alpha,   \
                       tiles_m, tiles_n, dbg);                                                             \
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
  return hipGetLastError();
}

}  // namespace ed
