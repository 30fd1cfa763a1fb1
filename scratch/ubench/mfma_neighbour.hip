// r06: a neighbour with the 256x256 GEMM's footprint on a CU (256 threads, 256 accumulator registers, 128 KB of LDS requested, one
// workgroup per CU) that either executes MFMAs (the GEMM's 16-per-k-step pattern on register operands) or only holds the registers,
// for `iters` k-steps — to separate "an MFMA wave EXECUTING on the SIMD" from "a big wave being LAUNCHED on the SIMD" as the thing
// that disturbs a co-resident wave's packed float ops (scratch/r06_geom_isa.py NEIGHBOUR=...).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/ubench/mfma_neighbour.hip -o scratch/ubench/mfma_neighbour.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>   // 1: MFMA   0: hold the accumulators, s_nop in place of each MFMA
                      // 2: MFMA on one operand set while 8 ds_read_b128 refill the other (the GEMM's k-step), 3: those ds_reads without the MFMAs
                      // 4: MFMA bursts of 16 with an idle gap (s_sleep) behind each, 5: bursts of 16 with an s_barrier behind each,
                      // 6: bursts of 16 with a dependent global load behind each (a memory-latency gap)
                      // 7: 16 MFMAs + 8 LDS-DMA loads (global_load_lds_dwordx4, M0 = destination) per k-step, vmcnt(0) behind them; 8: those LDS-DMA loads without the MFMAs
                      // 9: the GEMM's own interleave: s_mov m0 in front of every second MFMA, the LDS-DMA load behind it, destinations over 128 KB
__global__ __launch_bounds__(256) void nb_kernel(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef NB_PAD_VGPR   // arch VGPR count NB_PAD_VGPR + 1 (+ 256 accumulators): 151 gives the product GEMM's 408 registers per lane
#define NB_STR2(x) #x
#define NB_STR(x) NB_STR2(x)
  asm volatile("v_mov_b32 v" NB_STR(NB_PAD_VGPR) ", 0" ::: "v" NB_STR(NB_PAD_VGPR));
#endif
  bf16x8 A[4], B[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      A[i][e] = (__bf16)(0.001f * (float)((lane * 7 + i * 3 + e) % 13 - 6));
      B[i][e] = (__bf16)(0.001f * (float)((lane * 5 + i * 11 + e) % 17 - 8));
    }
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (MODE == 2 || MODE == 3) {
    for (int i = threadIdx.x; i < 16384; i += 256) ((uint32_t*)smem)[i] = 0x3c003c00u + (uint32_t)i * 2654435761u % 0x00400040u;
    __syncthreads();
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const uint32_t off = lds0 + (lane & 31) * 128 + ((lane >> 5) << 4);
  bf16x8 A2[4], B2[4];
  typedef __attribute__((ext_vector_type(4))) float f32x4_;
  f32x4_ ldtmp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) { A2[i] = A[i]; B2[i] = B[i]; }
  auto kstep = [&](bf16x8 (&RA)[4], bf16x8 (&RB)[4], bf16x8 (&MA)[4], bf16x8 (&MB)[4], int it) {   // read into R*, multiply M*
    if (MODE == 2 || MODE == 3) {
      const uint32_t a = off ^ ((it & 7) << 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(RA[i]) : "v"(a), "i"(i * 4096));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(RB[i]) : "v"(a), "i"(32768 + i * 4096));
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n_ = j * 4 + i;
        // A/B switches of mode 9 (-D...): NB9_LOW destinations below 64 KB only; NB9_NOBAR no s_barrier; NB9_M0LATE s_mov m0 behind the MFMA,
        // directly in front of its load; NB9_NODMA the s_mov m0 alone, no load at all; NB9_SGPR the value goes to s40 instead of m0
#ifdef NB9_LOW
        const uint32_t dst_ = lds0 + (uint32_t)((((n_ >> 2) * 2 + (it & 1)) & 1) * 16384 + ((n_ & 3) * 4 + wave) * 1024);
#else
        const uint32_t dst_ = lds0 + (uint32_t)(((n_ >> 2) * 2 + (it & 1)) * 16384 + ((n_ & 3) * 4 + wave) * 1024);
#endif
#if defined(NB9_SGPR)
        if (MODE == 9 && (n_ & 1)) asm volatile("s_mov_b32 s40, %0" : : "s"(dst_) : "memory", "s40");
#elif !defined(NB9_M0LATE)
        if (MODE == 9 && (n_ & 1)) asm volatile("s_mov_b32 m0, %0" : : "s"(dst_) : "memory");
#endif
        if (MODE == 1 || MODE == 2 || (MODE >= 4 && MODE != 8)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(MB[j]), "v"(MA[i]));
#if defined(NB9_M0LATE)
        if (MODE == 9 && (n_ & 1)) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst_), "v"((uint32_t)lane * 16u), "s"(out) : "memory");
#elif defined(NB9_VGPRLOAD)   // an ordinary load into registers in the same place
        if (MODE == 9 && (n_ & 1)) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ldtmp) : "v"((uint32_t)lane * 16u), "s"(out) : "memory");
#elif defined(NB9_DSWRITE)    // an LDS write from registers in the same place
        if (MODE == 9 && (n_ & 1)) asm volatile("ds_write_b128 %0, %1" : : "v"(dst_ + (uint32_t)lane * 16u), "v"(ldtmp) : "memory");
#elif !defined(NB9_NODMA) && !defined(NB9_SGPR)
        if (MODE == 9 && (n_ & 1)) asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"((uint32_t)lane * 16u), "s"(out) : "memory");
#endif
        else asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[i][j]) : "v"(MB[j]), "v"(MA[i]));
      }
    if (MODE == 7 || MODE == 8) {
#pragma unroll
      for (int d = 0; d < 8; ++d)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds0 + (uint32_t)(wave * 8 + d) * 1024u), "v"((uint32_t)lane * 16u), "s"(out) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 9) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(ldtmp) : : "memory");
#ifndef NB9_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
    }
    if (MODE == 4) asm volatile("s_sleep 4" ::: "memory");
    if (MODE == 5) __builtin_amdgcn_s_barrier();
    if (MODE == 6) {
      float g = out[(it * 64 + lane) & 1023];
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(g));
      if (g == 77.f) out[0] = g;
    }
    if (MODE == 2 || MODE == 3)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(RA[0]), "+v"(RA[1]), "+v"(RA[2]), "+v"(RA[3]), "+v"(RB[0]), "+v"(RB[1]), "+v"(RB[2]), "+v"(RB[3]));
  };
  for (int it = 0; it < iters; it += 2) {
    kstep(A2, B2, A, B, it);
    kstep(A, B, A2, B2, it + 1);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" ::"a"(acc[i][j]));
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[0][0][r];
  if (s == 123456.f) out[threadIdx.x] = s;
}

extern "C" int nb_launch(int mode, int iters, int lds_bytes, int grid, void* out, void* stream) {
  const void* fns[10] = {(const void*)nb_kernel<0>, (const void*)nb_kernel<1>, (const void*)nb_kernel<2>, (const void*)nb_kernel<3>,
                        (const void*)nb_kernel<4>, (const void*)nb_kernel<5>, (const void*)nb_kernel<6>, (const void*)nb_kernel<7>, (const void*)nb_kernel<8>, (const void*)nb_kernel<9>};
  if (mode < 0 || mode > 9) return -1;
  hipError_t r = hipFuncSetAttribute(fns[mode], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (r != hipSuccess) return (int)r;
#define NB_GO(M) hipLaunchKernelGGL(nb_kernel<M>, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, (float*)out, iters)
  switch (mode) {
    case 0: NB_GO(0); break;
    case 1: NB_GO(1); break;
    case 2: NB_GO(2); break;
    case 3: NB_GO(3); break;
    case 4: NB_GO(4); break;
    case 5: NB_GO(5); break;
    case 6: NB_GO(6); break;
    case 7: NB_GO(7); break;
    case 8: NB_GO(8); break;
    default: NB_GO(9); break;
  }
  return (int)hipGetLastError();
}
