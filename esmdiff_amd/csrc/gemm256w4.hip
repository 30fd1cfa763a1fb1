// gemm256w4.hip — the 256x256x64 bf16 MFMA GEMM with FOUR waves (one per SIMD) and 128x128 wave tiles (gfx950).
//
// Same contract, LDS image, raster, persistence and register-direct epilogue as gemm256.hip; what changes is the wave
// decomposition and that the main loop is a hand-placed instruction stream:
//
//   workgroup  256 threads = 4 waves as 2 (M) x 2 (N), one wave per SIMD, one workgroup per CU.  Wave tile 128 x 128 =
//              4 x 4 accumulators of v_mfma_f32_32x32x16_bf16 = 256 accumulator registers, kept in AGPRs (the MFMAs are
//              inline asm with "+a" operands); operands, addresses and everything else live in the 256 VGPRs.
//   why        r02 ablations of the 8-wave kernel (profiles/r02_gemm_ablations.txt): the fragment reads cost 22 % of the
//              FFN-up launch (about 15 matrix-pipe cycles per ds_read_b128 per SIMD, i.e. the time to move 1 KiB from LDS
//              into the register file), LDS-DMA issue 11 %, barriers and vmcnt waits nothing.  A 128 x 64 wave tile reads
//              6 fragments per 8 MFMAs, a 128 x 128 one 8 per 16: a third fewer LDS bytes per MFMA.
//   k-step     16 MFMAs (j = B fragment outer, i = A fragment inner: 16 independent accumulators back to back, the same
//              accumulator again 16 MFMAs later).  The 8 fragment reads of the NEXT k-step are placed one per MFMA gap
//              behind the first 8 MFMAs, into the other operand set; one s_waitcnt lgkmcnt(0) at the k-step boundary
//              (the reads were issued >= 8 MFMAs = 256 cycles earlier).
//   K-tile     4 k-steps.  Barrier between k-steps 2 and 3: by then every wave holds the fragments of k-step 3 in
//              registers (stage P is no longer read) and its own LDS-DMA of K-tile t+1 has landed (vmcnt(0): issued
//              during k-step 3 of K-tile t-1 and k-step 0 of K-tile t, i.e. >= 2 k-steps = 1024 matrix cycles ago).
//              The 16 LDS-DMA instructions per wave that refill stage P with K-tile t+2 go one per two MFMAs into k-step
//              3 of K-tile t (A pieces) and k-step 0 of K-tile t+1 (W pieces).
//   tiles      persistent; the DMA stream runs across tile boundaries exactly as the K-tile indices continue (v = nk is
//              K-tile 0 of the workgroup's next tile).  Needs an even number of K-tiles (K % 128 == 0).
#include <stdlib.h>

#include <type_traits>

#include "ed_half.h"
#include "kernels.h"

namespace ed {

typedef ed_half8 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace g4 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;     // 16 KiB
constexpr int STAGE_BYTES = 4 * HALF_BYTES;  // 64 KiB
// Raster of the tiles an XCD walks (its 32 CUs run 32 consecutive tiles at a time and share that XCD's 4 MB L2):
//   ED_W4_RASTER 0  groups of GROUP_M tile rows, column by column: a round is GROUP_M rows x 32 / GROUP_M columns
//   ED_W4_RASTER 1  groups of GROUP_M tile COLUMNS, row by row: a round is 32 / GROUP_M rows x GROUP_M columns — the group's W
//                   panels (GROUP_M x 0.79 MB at K = 1536) stay in L2 while A streams through
// (A/B builds: -DED_W4_GROUP_M=<n> -DED_W4_RASTER=<0|1>; the r06 traffic measurements are profiles/r06_gemm_raster_ab.txt)
#ifndef ED_W4_GROUP_M
#define ED_W4_GROUP_M 8
#endif
#ifndef ED_W4_RASTER
#define ED_W4_RASTER 0
#endif
constexpr int GROUP_M = ED_W4_GROUP_M;
// ablation builds (-DED_ABL4=<bits>, wrong results by construction): 1 no fragment reads, 2 no LDS-DMA, 4 no MFMA,
// 8 no global stores (epilogue arithmetic kept), 16 no epilogue at all
#ifndef ED_ABL4
#define ED_ABL4 0
#endif
#define W4_ABL(bit) (((ED_ABL4) & (bit)) != 0)
// Where the 16 LDS-DMA instructions of a K-tile sit among the MFMAs (-DED_W4_PLAN=<n>): the instruction index issued
// behind MFMA n of k-step ks (the refill of stage P with K-tile t+2 starts in k-step 3 of K-tile t and continues in k-steps
// 0.. of K-tile t+1), or -1.
#ifndef ED_W4_PLAN
#define ED_W4_PLAN 0
#endif
constexpr int dma_idx(int ks, int n) {
  if (ED_W4_PLAN == 0) {  // one per two MFMAs over all of k-steps 3 and 0 (shares the first gaps with the fragment reads)
    if (ks == 3) return (n & 1) ? (n >> 1) : -1;
    if (ks == 0) return (n & 1) ? 8 + (n >> 1) : -1;
    return -1;
  } else if (ED_W4_PLAN == 1) {  // one per MFMA in the read-free second half of k-steps 3 and 0
    if (ks == 3) return n >= 8 ? n - 8 : -1;
    if (ks == 0) return n >= 8 ? n : -1;
    return -1;
  } else if (ED_W4_PLAN == 2) {  // read-free gaps of three k-steps: 6 + 5 + 5 (k-step 2 is the landing slack)
    if (ks == 3) return n >= 10 ? n - 10 : -1;
    if (ks == 0) return n >= 11 ? 6 + (n - 11) : -1;
    if (ks == 1) return n >= 11 ? 11 + (n - 11) : -1;
    return -1;
  } else if (ED_W4_PLAN == 4) {  // 6 + 6 + 4 over two and a half k-steps, one per two MFMAs
    if (ks == 3) return (n >= 4 && !(n & 1)) ? (n - 4) >> 1 : -1;          // 0..5 at n = 4,6,..,14
    if (ks == 0) return ((n & 1) && n <= 11) ? 6 + (n >> 1) : -1;          // 6..11 at n = 1,3,..,11
    if (ks == 1) return ((n & 1) && n <= 7) ? 12 + (n >> 1) : -1;          // 12..15 at n = 1,3,5,7
    return -1;
  } else if (ED_W4_PLAN == 5) {  // 5 + 6 + 5 over three k-steps, one per three MFMAs
    if (ks == 3) return (n >= 1 && n % 3 == 1) ? (n - 1) / 3 : -1;         // 0..4 at n = 1,4,7,10,13
    if (ks == 0) return (n % 3 == 0) ? 5 + n / 3 : -1;                     // 5..10 at n = 0,3,..,15
    if (ks == 1) return (n % 3 == 1 && n <= 13) ? 11 + (n - 1) / 3 : -1;   // 11..15 at n = 1,4,7,10,13
    return -1;
  } else {  // 3: read-free gaps, one per two MFMAs, of all four k-steps is impossible (landing); 4 + 4 + 4 + 4 over ks 3,0,1 + late ks 3
    if (ks == 3) return n >= 8 ? ((n & 1) ? -1 : (n - 8) >> 1) : -1;               // 0..3 at n = 8,10,12,14
    if (ks == 0) return n >= 4 ? ((n & 1) ? -1 : 4 + ((n - 4) >> 1)) : -1;         // 4..9 at n = 4..14 even
    if (ks == 1) return n >= 4 ? ((n & 1) ? -1 : 10 + ((n - 4) >> 1)) : -1;        // 10..15
    return -1;
  }
}
constexpr int dma_first_count() {  // instructions of a K-tile issued in k-step 3 (the prologue issues as many of K-tile 1)
  int c = 0;
  for (int n = 0; n < 16; ++n) c += dma_idx(3, n) >= 0 ? 1 : 0;
  return c;
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_mul(float g, float u) {
  return g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.44269504088896341f)) * u;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) { return ed_pack2(a, b); }   // the TU's 16-bit type (ed_half.h)

// SPLIT = 1 (gemm_split.hip's launcher, the float32-grade "split" linears of the strict path): the operands are f16 plane
// triples per row — A [M, 3K1] = [hi | lo | hi], W [N, 3K1] = [lo | hi | hi] with x ~ hi + lo to 2^-22 — so that ONE linear
// walk over K = 3 K1 accumulates A_hi.W_lo + A_lo.W_hi + A_hi.W_hi (small terms first) into the same f32 accumulators on
// v_mfma_f32_32x32x16_f16: the main loop is the bf16 one with another opcode and not one scalar more (a two-plane layout
// with a jump back for the third pass cost 5 SGPRs, the kernel spilled, and hipcc's v_readlane reloads landed directly in
// front of the inline-asm LDS-DMA that consumed them: a VALU-writes-SGPR -> VMEM hazard it does not pad inside asm).
// EPI is then one of esmdiff_gemm_f32_epilogue, outputs are f32, scaled per row by rs[m] * alpha (powers of two: exact).
template <int EPI, int SPLIT = 0>
__global__ __launch_bounds__(256, 1) void gemm256w4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                           void* __restrict__ out, const float* __restrict__ bias, int M,
                                                           int N, int K, int ldc, float alpha, int tiles_m, int tiles_n,
                                                           const float* __restrict__ rs = nullptr, float div = 1.f,
                                                           int ld_ab = 0, int m_pad = 0, int m_real = 0) {
  // SPLIT == 2 (launch_gemm256w4_splitk): K slices of the SPLIT == 1 product as extra row blocks.  The grid walks
  // S * m_pad "virtual" rows; virtual row block s = m0 / m_pad reads columns s*K .. (s+1)*K of the physical rows
  // m0 - s*m_pad .. of A and of every W row (row stride ld_ab = the whole 3 K1 walk), and its f32 partial products go to
  // rows m0 .. of `out` ([S * m_pad, N]).  Everything but the source addresses below is the SPLIT == 1 kernel.
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x 64 KiB

  const int n_tiles = tiles_m * tiles_n, bid = blockIdx.x;
  auto tile_origin = [&](int vt, int& m0_, int& n0_) {
    const int xcd = vt & 7, qq = n_tiles >> 3, rr = n_tiles & 7;
    const int lin = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (vt >> 3);
    if constexpr (ED_W4_RASTER == 0) {
      const int per_group = GROUP_M * tiles_n;
      const int grp = lin / per_group, in_grp = lin - grp * per_group;
      const int gm0 = grp * GROUP_M;
      const int gsz = min(GROUP_M, tiles_m - gm0);
      m0_ = (gm0 + in_grp % gsz) * BM;
      n0_ = (in_grp / gsz) * BN;
    } else {
      const int per_group = GROUP_M * tiles_m;
      const int grp = lin / per_group, in_grp = lin - grp * per_group;
      const int gn0 = grp * GROUP_M;
      const int gsz = min(GROUP_M, tiles_n - gn0);
      n0_ = (gn0 + in_grp % gsz) * BN;
      m0_ = (in_grp / gsz) * BM;
    }
  };
  int m0, n0;
  tile_origin(bid, m0, n0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- LDS-DMA sources.  Piece p = 0,1: A rows 0-127 / 128-255; p = 2,3: W rows.  A piece is 16 instructions of 8 rows;
  // this wave issues i = 0..3, instruction i covering rows (i*4 + wave)*8 + (lane>>3) of the piece. ------------------
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);  // (row >> 1) & 7 of that row
  uint32_t a_off[2][4], w_off[2][4], a_offn[2][4], w_offn[2][4];     // current tile / the workgroup's next tile
  auto set_offsets = [&](uint32_t (&ao)[2][4], uint32_t (&wo)[2][4], int m0_, int n0_) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (SPLIT == 2) {
          const int slice = m0_ / m_pad;
          const int am = min(m0_ - slice * m_pad + h * 128 + (i * 4 + wave) * 8 + srow, m_real - 1);
          const int64_t col = (int64_t)slice * K + schunk * 8;
          ao[h][i] = (uint32_t)(((int64_t)am * ld_ab + col) * 2);
          wo[h][i] = (uint32_t)(((int64_t)(n0_ + h * 128 + (i * 4 + wave) * 8 + srow) * ld_ab + col) * 2);
        } else {
          const int am = min(m0_ + h * 128 + (i * 4 + wave) * 8 + srow, M - 1);
          ao[h][i] = (uint32_t)(((int64_t)am * K + schunk * 8) * 2);
          wo[h][i] = (uint32_t)(((int64_t)(n0_ + h * 128 + (i * 4 + wave) * 8 + srow) * K + schunk * 8) * 2);
        }
      }
    }
  };
  set_offsets(a_off, w_off, m0, n0);
  constexpr int kstride = BK * 2;
  const char* Ab = reinterpret_cast<const char*>(A);
  const char* Wb = reinterpret_cast<const char*>(W);
  const uint32_t lds_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  // M0 (the LDS destination) is written and consumed inside one statement; nothing else in this kernel uses M0 (plain
  // ds_read / ds_write do not), so it is not saved and restored around each of the 16 instructions per K-tile
  // (ED_W4_M0SPLIT: the s_mov m0 goes in front of the MFMA of its gap and the load behind it — the MFMA is the wait state
  // the pair needs, which saves the s_nop; nothing else may write M0 in between, and nothing does)
#ifndef ED_W4_M0SPLIT
#define ED_W4_M0SPLIT 1
#endif
  // cache policy of the operand streams (-DED_W4_POL_A / -DED_W4_POL_W = "" | " nt" | " sc1" ...; A/B builds only: the
  // default policy measured best, profiles/r03_gemm_cache_policy.txt)
#ifndef ED_W4_POL_A
#define ED_W4_POL_A ""
#endif
#ifndef ED_W4_POL_W
#define ED_W4_POL_W ""
#endif
  auto glds = [&](const char* sbase, uint32_t voff, uint32_t lds_dst, int phase, bool is_w) {
    if (phase == 0) {
      if (is_w) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ED_W4_POL_W : : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
      else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ED_W4_POL_A : : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
    } else if (phase == 1) {
      asm volatile("s_mov_b32 m0, %0" : : "s"(lds_dst) : "memory");
    } else {
      if (is_w) asm volatile("global_load_lds_dwordx4 %0, %1" ED_W4_POL_W : : "v"(voff), "s"(sbase) : "memory");
      else asm volatile("global_load_lds_dwordx4 %0, %1" ED_W4_POL_A : : "v"(voff), "s"(sbase) : "memory");
    }
  };
  int xnext = 0;  // wave-uniform: this workgroup has another tile after the current one
  const int nk = K / BK;
  // LDS-DMA instruction idx = p*4 + i of K-tile v into stage `buf`.  NEXT: v counts K-tiles of the workgroup's NEXT tile
  // (only the last two K-tiles of a tile stream the next tile's first two: the steady-state loop has no condition at all).
  auto dma1 = [&](auto NEXT, int idx, int buf, int v, int phase = 0) {
    if (W4_ABL(2)) return;
    const int p = idx >> 2, i = idx & 3;
    const uint32_t dst = lds_base + (p * 2 + buf) * HALF_BYTES + (i * 4 + wave) * 1024;
    const char* sb = (p < 2 ? Ab : Wb) + (size_t)v * kstride;  // one scalar base per operand and K-tile
    if constexpr (!decltype(NEXT)::value) {
      glds(sb, p < 2 ? a_off[p][i] : w_off[p - 2][i], dst, phase, p >= 2);
    } else {
      if (xnext) glds(sb, p < 2 ? a_offn[p][i] : w_offn[p - 2][i], dst, phase, p >= 2);
    }
  };

  // ---- fragment read addressing (inline-asm ds_read_b128, immediates carry stage / fragment offsets) --------------
  const int frow = lane & 31, khalf = lane >> 5, fsw = (frow >> 1) & 7;
  uint32_t offA[4], offB[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint32_t lo = frow * 128 + (((ks * 2 + khalf) ^ fsw) << 4);
    offA[ks] = lds_base + (wm * 2) * HALF_BYTES + lo;
    offB[ks] = lds_base + ((2 + wn) * 2) * HALF_BYTES + lo;
  }
#define W4_DSR(dst, addr, imm)                                                                    \
  do {                                                                                            \
    if (!W4_ABL(1)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm)); \
    else asm volatile("" : "=v"(dst));                                                            \
  } while (0)
#define W4_MFMA(acc_, b_, a_)                                                                         \
  do {                                                                                                \
    if (W4_ABL(4)) asm volatile("" : "+a"(acc_) : "v"(b_), "v"(a_));                                  \
    else if constexpr (SPLIT) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc_) : "v"(b_), "v"(a_)); \
    else asm volatile(ED_MFMA_32x32x16_ASM " %0, %1, %2, %0" : "+a"(acc_) : "v"(b_), "v"(a_));        \
  } while (0)
// first k-step of a tile: D = B x A + 0 (no zeroing pass over the 256 accumulator registers)
#define W4_MFMA0(acc_, b_, a_)                                                                        \
  do {                                                                                                \
    if (W4_ABL(4)) asm volatile("" : "=a"(acc_) : "v"(b_), "v"(a_));                                  \
    else if constexpr (SPLIT) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc_) : "v"(b_), "v"(a_)); \
    else asm volatile(ED_MFMA_32x32x16_ASM " %0, %1, %2, 0" : "=a"(acc_) : "v"(b_), "v"(a_));         \
  } while (0)
#define W4_WAIT_LGKM0(s_)                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                              \
               : "+v"(s_.a[0]), "+v"(s_.a[1]), "+v"(s_.a[2]), "+v"(s_.a[3]), "+v"(s_.b[0]), "+v"(s_.b[1]), "+v"(s_.b[2]), \
                 "+v"(s_.b[3]))

  struct OpSet {
    bf16x8 a[4], b[4];
  };
  OpSet X, Y;
  f32x16 acc[4][4];  // [i: A row block][j: W column block]; holds the TRANSPOSED 32x32 block (W fragment is operand 1)

  // read fragment r (0..3: A row block r, 4..7: W column block r-4) of k-step ks, stage buf, into set S
  auto read_frag = [&](OpSet& S, auto R, auto KS, auto BUF) {
    constexpr int r = decltype(R)::value, ks = decltype(KS)::value, buf = decltype(BUF)::value;
    const uint32_t addr = r < 4 ? offA[ks] : offB[ks];
    bf16x8& dst = r < 4 ? S.a[r & 3] : S.b[r & 3];
    W4_DSR(dst, addr, buf * HALF_BYTES + (r & 3) * 4096);
  };

  // One k-step: 16 MFMAs on set C; the reads of the next k-step (NKS of stage NBUF) into set Nx behind MFMAs 0..7;
  // DMA_BASE >= 0: LDS-DMA instructions DMA_BASE .. DMA_BASE+7 of K-tile dma_v into stage DMA_BUF behind the odd MFMAs.
  auto kstep = [&](OpSet& C, OpSet& Nx, auto NKS, auto NBUF, auto KS, auto DMA_BUF, auto DMA_NEXT, int dma_v,
                   auto FIRST) {
    constexpr int ks_ = decltype(KS)::value, dbuf = decltype(DMA_BUF)::value;
#define W4_STEP(n)                                                                               \
  {                                                                                              \
    constexpr int j_ = (n) >> 2, i_ = (n)&3;                                                     \
    if constexpr (ED_W4_M0SPLIT && dma_idx(ks_, (n)) >= 0) dma1(DMA_NEXT, dma_idx(ks_, (n)), dbuf, dma_v, 1); \
    if constexpr (decltype(FIRST)::value) W4_MFMA0(acc[i_][j_], C.b[j_], C.a[i_]);               \
    else W4_MFMA(acc[i_][j_], C.b[j_], C.a[i_]);                                                 \
    if constexpr ((n) < 8) read_frag(Nx, std::integral_constant<int, (n)>{}, NKS, NBUF);         \
    if constexpr (dma_idx(ks_, (n)) >= 0) dma1(DMA_NEXT, dma_idx(ks_, (n)), dbuf, dma_v, ED_W4_M0SPLIT ? 2 : 0); \
  }
    W4_STEP(0) W4_STEP(1) W4_STEP(2) W4_STEP(3) W4_STEP(4) W4_STEP(5) W4_STEP(6) W4_STEP(7)
    W4_STEP(8) W4_STEP(9) W4_STEP(10) W4_STEP(11) W4_STEP(12) W4_STEP(13) W4_STEP(14) W4_STEP(15)
#undef W4_STEP
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  using TF = std::false_type;
  using TT = std::true_type;
  // One K-tile at stage P.  Entry: X holds the fragments of (t, k-step 0).  N0 / N3: the LDS-DMA issued in k-step 0 / 3
  // belongs to the workgroup's next tile (K-tile v0 / v3 of it) instead of K-tile t+1 / t+2 of this one.
  auto ktile = [&](int t, auto P, auto N0, auto N3, int v0, int v3, auto FIRST) {
    using Q = std::integral_constant<int, 1 - decltype(P)::value>;
    // k-step 0 (+ second half of the LDS-DMA of the next K-tile into the other stage)
    kstep(X, Y, I1{}, P, I0{}, Q{}, N0, v0, FIRST);
    W4_WAIT_LGKM0(Y);
    kstep(Y, X, I2{}, P, I1{}, Q{}, N0, v0, TF{});
    W4_WAIT_LGKM0(X);
    kstep(X, Y, I3{}, P, I2{}, Q{}, N0, v0, TF{});
    W4_WAIT_LGKM0(Y);                                  // ... and stage P is fully read by this wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next K-tile has landed
    __builtin_amdgcn_s_barrier();
    // k-step 3: first fragments of the next K-tile (other stage; after the tile's last K-tile they are simply not used) +
    // first half of the LDS-DMA of the K-tile after it into this stage
    kstep(Y, X, I0{}, Q{}, I3{}, P, N3, v3, TF{});
    W4_WAIT_LGKM0(X);
  };

  bool have_k0 = false;
  const uint32_t lhi = lane >> 5, lrow = lane & 31;
  for (int vt = bid; vt < n_tiles; vt += gridDim.x) {
    const int vtn = vt + gridDim.x;
    xnext = __builtin_amdgcn_readfirstlane(vtn < n_tiles ? 1 : 0);
    int m0n = 0, n0n = 0;
    if (xnext) {
      tile_origin(vtn, m0n, n0n);
      set_offsets(a_offn, w_offn, m0n, n0n);
    }
    if (!have_k0) {  // first tile of this workgroup: K-tile 0 and the first half of K-tile 1
#pragma unroll
      for (int idx = 0; idx < 16; ++idx) dma1(TF{}, idx, 0, 0);
#pragma unroll
      for (int idx = 0; idx < dma_first_count(); ++idx) dma1(TF{}, idx, 1, 1);
      if constexpr (dma_first_count() == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (dma_first_count() == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (dma_first_count() == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if constexpr (dma_first_count() == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }  // else: the previous tile's K loop streamed them (and its last barrier made K-tile 0 visible)
    if (!have_k0) {  // (a following tile finds the fragments of its (K-tile 0, k-step 0) in X: the last k-step of the
                     // previous tile read them from stage 0, where the stream had already put this tile's K-tile 0)
      read_frag(X, I0{}, I0{}, I0{});
      read_frag(X, I1{}, I0{}, I0{});
      read_frag(X, I2{}, I0{}, I0{});
      read_frag(X, I3{}, I0{}, I0{});
      read_frag(X, std::integral_constant<int, 4>{}, I0{}, I0{});
      read_frag(X, std::integral_constant<int, 5>{}, I0{}, I0{});
      read_frag(X, std::integral_constant<int, 6>{}, I0{}, I0{});
      read_frag(X, std::integral_constant<int, 7>{}, I0{}, I0{});
      W4_WAIT_LGKM0(X);
    }

    ktile(0, I0{}, TF{}, TF{}, 1, 2, TT{});          // first k-step writes the accumulators (C = 0)
    ktile(1, I1{}, TF{}, TF{}, 2, 3, TF{});
    for (int t = 2; t < nk - 2; t += 2) {            // steady state: no condition anywhere
      ktile(t, I0{}, TF{}, TF{}, t + 1, t + 2, TF{});
      ktile(t + 1, I1{}, TF{}, TF{}, t + 2, t + 3, TF{});
    }
    ktile(nk - 2, I0{}, TF{}, TT{}, nk - 1, 0, TF{});  // k-step 3 starts streaming the next tile's K-tile 0
    ktile(nk - 1, I1{}, TT{}, TT{}, 0, 1, TF{});        // ... finishes it, and starts its K-tile 1
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results before the compiler's accvgpr reads

    // ---- epilogue: register-direct, as gemm256.hip (each 32x32 accumulator is the transposed output block: a lane
    // holds one output row and 4 consecutive columns per register group; v_permlane32_swap widens that to 8) -----------
    auto swap_halves = [](uint32_t& lo_keep, uint32_t& hi_keep) {
      const auto r = __builtin_amdgcn_permlane32_swap(lo_keep, hi_keep, false, false);
      lo_keep = r[0];
      hi_keep = r[1];
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (W4_ABL(16)) {
        asm volatile("" ::"a"(acc[i][0]), "a"(acc[i][1]), "a"(acc[i][2]), "a"(acc[i][3]));
        continue;
      }
      const int m = m0 + wm * 128 + i * 32 + lrow;
      const bool live = m < M && !(W4_ABL(8) && alpha != 12345.f);
      if constexpr (SPLIT && EPI == 4) {
        // F32_SPLIT FFN-up with the SwiGLU fused (W rows interleaved gate / up in blocks of 32, as in the bf16 path): mid =
        // silu(g) * u in f32, written as f32 [M, FH] (ldc >= FH).  exp and the reciprocal are the hardware's (1 ulp each).
        // The split row the FFN-down GEMM reads is made from it by split_rows_kernel with the row's OWN power-of-two scale
        // (r05).  r04 wrote the split row right here with one scale per layer taken from the a-priori bound |mid| <= B^2:
        // on weights with trained statistics (LayerNorm gains of 30, FFN units with 50x row norm) that bound sits 2^22 ..
        // 2^33 above the typical element, the f16 pair underflows, and the engine's logits were 100x further from a float64
        // evaluation than the exact-f32 engine's (profiles/r05_split_vs_f64.txt).
        const float sc = (rs && m < M) ? rs[m] * alpha : alpha;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float gt = acc[i][2 * jp][g * 4 + e] * sc, up = acc[i][2 * jp + 1][g * 4 + e] * sc;
              v[e] = gt * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gt * -1.44269504088896341f)) * up;
            }
            if (!live) continue;
            float* o = reinterpret_cast<float*>(out) + (int64_t)m * ldc + (n0 + wn * 128 + jp * 64) / 2 + g * 8 + lhi * 4;
            *reinterpret_cast<f32x4*>(o) = v;
          }
      } else if constexpr (SPLIT) {   // f32 outputs of the split linears: acc * (row scale * weight scale), both powers of two
        const float sc = (rs && m < M) ? rs[m] * alpha : alpha;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // no column bound: the launcher requires ldc >= N (a bound check per 8-column group becomes 16 hoisted lane
            // masks or scalar flags = 32+ SGPRs, the kernel spills, and hipcc's v_readlane reloads land in front of the
            // inline-asm LDS-DMA that reads them: a VALU-writes-SGPR -> VMEM hazard nobody pads inside asm)
            const int n = n0 + wn * 128 + j * 32 + g * 8 + lhi * 4;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e] * sc;
            if (!live) continue;
            float* o = reinterpret_cast<float*>(out) + (int64_t)m * ldc + n;
            if constexpr (EPI == ESMDIFF_F32EPI_RESID_DIV) {   // x = x + r / scaling_factor (esm's own expression)
              f32x4 x = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] = x[e] + v[e] / div;
              *reinterpret_cast<f32x4*>(o) = x;
            } else {
              if constexpr (EPI != ESMDIFF_F32EPI_STORE) {   // 3: the launcher's code for STORE with a bias
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bb[e];
              }
              *reinterpret_cast<f32x4*>(o) = v;
            }
          }
      } else if constexpr (EPI == ESMDIFF_EPI_SWIGLU_BF16) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          bf16_t* orow = reinterpret_cast<bf16_t*>(out) + (int64_t)m * ldc + (n0 + wn * 128 + jp * 64) / 2 + lhi * 8;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = silu_mul(acc[i][2 * jp][gp * 8 + e], acc[i][2 * jp + 1][gp * 8 + e]);
            uint32_t p0 = pack_bf16x2(h[0], h[1]), p1 = pack_bf16x2(h[2], h[3]);
            uint32_t q0 = pack_bf16x2(h[4], h[5]), q1 = pack_bf16x2(h[6], h[7]);
            swap_halves(p0, q0);
            swap_halves(p1, q1);
            if (live) *reinterpret_cast<uint4*>(orow + gp * 16) = make_uint4(p0, p1, q0, q1);
          }
        }
      } else if constexpr (EPI == ESMDIFF_EPI_BF16 || EPI == ESMDIFF_EPI_BIAS_GELU_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nb = n0 + wn * 128 + j * 32;
          bf16_t* orow = reinterpret_cast<bf16_t*>(out) + (int64_t)m * ldc + nb + lhi * 8;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = acc[i][j][gp * 8 + e];
            if constexpr (EPI == ESMDIFF_EPI_BF16) {
#pragma unroll
              for (int e = 0; e < 8; ++e) h[e] *= alpha;
            } else {
              const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + nb + gp * 16 + lhi * 4);
              const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + nb + gp * 16 + 8 + lhi * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                h[e] = gelu_erf(h[e] + b0[e]);
                h[4 + e] = gelu_erf(h[4 + e] + b1[e]);
              }
            }
            uint32_t p0 = pack_bf16x2(h[0], h[1]), p1 = pack_bf16x2(h[2], h[3]);
            uint32_t q0 = pack_bf16x2(h[4], h[5]), q1 = pack_bf16x2(h[6], h[7]);
            swap_halves(p0, q0);
            swap_halves(p1, q1);
            if (live) *reinterpret_cast<uint4*>(orow + gp * 16) = make_uint4(p0, p1, q0, q1);
          }
        }
      } else {  // f32 outputs
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * 128 + j * 32 + g * 8 + lhi * 4;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e];
            if (!live) continue;
            float* o = reinterpret_cast<float*>(out) + (int64_t)m * ldc + n;
            if constexpr (EPI == ESMDIFF_EPI_RESID_F32) {
              f32x4 x = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
              for (int e = 0; e < 4; ++e) x[e] += v[e] * alpha;
              *reinterpret_cast<f32x4*>(o) = x;
            } else {
              if (n + 4 <= ldc) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bb[e];
                *reinterpret_cast<f32x4*>(o) = v;
              }
            }
          }
      }
    }

    if (xnext) {
      have_k0 = true;
      m0 = m0n;
      n0 = n0n;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a_off[h][i] = a_offn[h][i];
          w_off[h][i] = w_offn[h][i];
        }
    }
  }
}
}  // namespace g4

// The split linears (see the kernel's SPLIT note; operand preparation: gemm_split.hip).  A3 f16 [M, 3K] = [hi | lo | hi] with
// per-row scale rs[M] (NULL: 1), W3 f16 [N, 3K] = [lo | hi | hi] scaled by 1 / w_scale; out f32 [M, ldc]; N % 256 == 0,
// K % 128 == 0 (so that 3 K / 64 is even).
hipError_t launch_gemm256w4_split(const uint16_t* A2, const float* rs, const uint16_t* W2, float w_scale, float* out,
                                  const float* bias, int M, int N, int K, int ldc, float div, int epi, hipStream_t stream) {
  using namespace g4;
  if (M <= 0) return hipSuccess;
  if (N % BN != 0 || K % (2 * BK) != 0 || (ldc & 3) || (epi != 4 && ldc < N) || (epi == 4 && ldc < N / 2)) return hipErrorInvalidValue;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
  static const int n_cu = [] {   // (one device model per process: every gfx950 in a node has the same CU count)
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? (n / 8) * 8 : 8;
  }();
  const int n_tiles = tiles_m * tiles_n;
  dim3 grid(n_tiles < n_cu ? n_tiles : n_cu), block(256);
  const size_t lds = 2 * STAGE_BYTES;
  // (no bias + GELU epilogue here: erff on 256 accumulators spills; the consumer LayerNorm applies the GELU on load)
#define ED_GEMM_S(E)                                                                                                \
  do {                                                                                                              \
    {                                                                                                               \
      const hipError_t a_ = ensure_dynamic_lds((const void*)gemm256w4_kernel<E, 1>, (int)lds);                      \
      if (a_ != hipSuccess) return a_;                                                                              \
    }                                                                                                               \
    hipLaunchKernelGGL((gemm256w4_kernel<E, 1>), grid, block, lds, stream, A2, W2, (void*)out, bias, M, N, 3 * K, ldc, \
                       w_scale, tiles_m, tiles_n, rs, div);                                                      \
  } while (0)
  switch (epi) {
    case ESMDIFF_F32EPI_STORE:
      if (bias) ED_GEMM_S(3);
      else ED_GEMM_S(ESMDIFF_F32EPI_STORE);
      break;
    case ESMDIFF_F32EPI_RESID_DIV: ED_GEMM_S(ESMDIFF_F32EPI_RESID_DIV); break;
    case 4: ED_GEMM_S(4); break;   // fused SwiGLU -> mid as f32 [M, ldc >= N / 2] (W rows interleaved gate / up)
    default: return hipErrorInvalidValue;
  }
#undef ED_GEMM_S
  return hipGetLastError();
}

// The same product cut into S slices of the 3 K walk, for launches with too few tiles to fill the chip: parts[s] ([m_pad, N]
// f32, m_pad = M rounded up to 256) = (slice s of A2) . (slice s of W2)^T * w_scale, row scales NOT applied; the caller sums
// the slices in order and applies rs (gemm_split.hip::launch_splitk_reduce_resid).  3 K / S must be a multiple of 128.
hipError_t launch_gemm256w4_splitk(const uint16_t* A2, const uint16_t* W2, float w_scale, float* parts, int M, int N, int K,
                                   int S, hipStream_t stream) {
  using namespace g4;
  if (M <= 0) return hipSuccess;
  if (S < 2 || N % BN != 0 || (3 * K) % S != 0 || ((3 * K) / S) % (2 * BK) != 0 || (3 * K) / S < 6 * BK) return hipErrorInvalidValue;
  const int tiles_m_phys = (M + BM - 1) / BM, tiles_n = N / BN, m_pad = tiles_m_phys * BM;
  const int tiles_m = S * tiles_m_phys;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? (n / 8) * 8 : 8;
  }();
  const int n_tiles = tiles_m * tiles_n;
  dim3 grid(n_tiles < n_cu ? n_tiles : n_cu), block(256);
  const size_t lds = 2 * STAGE_BYTES;
  const hipError_t a_ = ensure_dynamic_lds((const void*)gemm256w4_kernel<ESMDIFF_F32EPI_STORE, 2>, (int)lds);
  if (a_ != hipSuccess) return a_;
  hipLaunchKernelGGL((gemm256w4_kernel<ESMDIFF_F32EPI_STORE, 2>), grid, block, lds, stream, A2, W2, (void*)parts, (const float*)nullptr,
                     S * m_pad, N, (3 * K) / S, N, w_scale, tiles_m, tiles_n, (const float*)nullptr, 1.f, 3 * K, m_pad, M);
  return hipGetLastError();
}

hipError_t launch_gemm256w4_bf16(const bf16_t* A, const bf16_t* W, void* out, const float* bias, int M, int N, int K,
                                 int ldc, float alpha, int epilogue, hipStream_t stream) {
  using namespace g4;
  if (M <= 0) return hipSuccess;
  if (N % BN != 0 || K % (2 * BK) != 0 || K < 6 * BK || (ldc & 3)) return hipErrorInvalidValue;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? (n / 8) * 8 : 8;
  }();
  const int n_tiles = tiles_m * tiles_n;
  dim3 grid(n_tiles < n_cu ? n_tiles : n_cu), block(256);
  const size_t lds = 2 * STAGE_BYTES;
#define ED_GEMM(E)                                                                                                  \
  do {                                                                                                              \
    {                                                                                                               \
      const hipError_t a_ = ensure_dynamic_lds((const void*)gemm256w4_kernel<E>, (int)lds);                         \
      if (a_ != hipSuccess) return a_;                                                                              \
    }                                                                                                               \
    hipLaunchKernelGGL(gemm256w4_kernel<E>, grid, block, lds, stream, A, W, out, bias, M, N, K, ldc, alpha, tiles_m, \
                       tiles_n);                                                                                    \
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
