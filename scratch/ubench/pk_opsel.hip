// r06: self-checking minimal victim.  Each lane loads 32 bytes (A.lo A.hi B.lo B.hi | C.lo C.hi . .), waits for all of it, computes ONE
// packed float op and the same two results with plain float ops from the same registers, and compares bit for bit; `reps` times at
// different addresses.  MODE selects the packed form:
//   0  v_pk_fma_f32 D, A, B, C op_sel:[0,1,0] op_sel_hi:[1,0,1]   lo = A.lo * B.hi + C.lo, hi = A.hi * B.lo + C.hi   (the form that fails in pk_rotate)
//   1  v_pk_fma_f32 D, A, B, C                                      plain
//   2  v_pk_fma_f32 D, A, B, C op_sel_hi:[1,0,1]                    hi half from B.lo
//   3  v_pk_mul_f32 D, A, B op_sel:[0,1] op_sel_hi:[1,0]
//   4  v_pk_add_f32 D, A, A op_sel:[0,1] op_sel_hi:[1,0]            the horizontal add of the LayerNorm kernels
//   5  v_pk_fma_f32 D, A, B, C op_sel:[1,0,0]                       lo = A.hi * B.lo + C.lo
//   6  v_pk_fma_f32 D, A, B, C op_sel:[0,0,1]                       lo = A.lo * B.lo + C.hi
//   7  v_pk_fma_f32 D, A, B, C op_sel:[0,1,0]                       lo = A.lo * B.hi + C.lo, hi half plain
//   8  v_pk_mul_f32 D, s[..], B op_sel:[1,0]                        a scalar pair as src0, its high half in the lo result (the product's only op_sel forms)
//   9 / 10 / 11  v_pk_fma_f16 / v_pk_mul_f16 / v_pk_add_f16 with op_sel:[0,1(,0)] op_sel_hi:[1,0(,1)] (16-bit halves of ONE register swapped)
//      against the plain packed op on a copy of B whose halves were swapped with v_alignbit_b32 (inputs masked to finite f16 values)
// err[q] mismatching (lane, rep) of lane quarter q; err[4] lo-half mismatches, err[5] hi-half mismatches; err[6] lo == C.lo exactly
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/ubench/pk_opsel.hip -o scratch/ubench/pk_opsel.so
// (registers v40..v51: with v100.. the kernel allocates 112 registers and can NOT share a SIMD with a 408-register GEMM wave — the first
// version of this probe could never fail)
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LOADS                                         \
  "global_load_dwordx4 v[40:43], %4, off\n\t"       \
  "global_load_dwordx4 v[44:47], %4, off offset:16\n\t" \
  "s_waitcnt vmcnt(0)\n\t"
#define OUTS                     \
  "s_nop 4\n\t"                  \
  "v_mov_b32 %0, v48\n\t"       \
  "v_mov_b32 %1, v49\n\t"       \
  "v_mov_b32 %2, v50\n\t"       \
  "v_mov_b32 %3, v51\n\t"
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory"

template <int MODE>
__global__ __launch_bounds__(64) void opsel_kernel(const float4* __restrict__ G, uint32_t n8, int reps, unsigned* err) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const uint32_t wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (lane == 0) lds[0] = 0.f;
  uint32_t idx = (wg * 64u + lane) % n8;
  unsigned bad = 0, blo = 0, bhi = 0, clo = 0;
  for (int r = 0; r < reps; ++r) {
    const float4* p = G + 2 * (size_t)idx;
    uint32_t a0, a1, b0, b1;
    if (MODE == 0)
      asm volatile(LOADS "v_pk_fma_f32 v[48:49], v[40:41], v[42:43], v[44:45] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
                         "v_fma_f32 v50, v40, v43, v44\n\tv_fma_f32 v51, v41, v42, v45\n\t" OUTS
                   : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
    else if (MODE == 1)
      asm volatile(LOADS "v_pk_fma_f32 v[48:49], v[40:41], v[42:43], v[44:45]\n\t"
                         "v_fma_f32 v50, v40, v42, v44\n\tv_fma_f32 v51, v41, v43, v45\n\t" OUTS
                   : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
    else if (MODE == 2)
      asm volatile(LOADS "v_pk_fma_f32 v[48:49], v[40:41], v[42:43], v[44:45] op_sel_hi:[1,0,1]\n\t"
                         "v_fma_f32 v50, v40, v42, v44\n\tv_fma_f32 v51, v41, v42, v45\n\t" OUTS
                   : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
    else if (MODE == 3)
      asm volatile(LOADS "v_pk_mul_f32 v[48:49], v[40:41], v[42:43] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_mul_f32 v50, v40, v43\n\tv_mul_f32 v51, v41, v42\n\t" OUTS
                   : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
    else if (MODE == 4)
      asm volatile(LOADS "v_pk_add_f32 v[48:49], v[40:41], v[40:41] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_add_f32 v50, v40, v41\n\tv_add_f32 v51, v41, v40\n\t" OUTS
                   : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
    else if (MODE == 6)
      asm volatile(LOADS "v_pk_fma_f32 v[48:49], v[40:41], v[42:43], v[44:45] op_sel:[0,0,1]\n\t"
                         "v_fma_f32 v50, v40, v42, v45\n\tv_fma_f32 v51, v41, v43, v45\n\t" OUTS
                   : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
    else if (MODE == 7)
      asm volatile(LOADS "v_pk_fma_f32 v[48:49], v[40:41], v[42:43], v[44:45] op_sel:[0,1,0]\n\t"
                         "v_fma_f32 v50, v40, v43, v44\n\tv_fma_f32 v51, v41, v43, v45\n\t" OUTS
                   : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
    else if (MODE >= 9 && MODE <= 11) {
      // a0 / b0: the op_sel form / the plain form on swapped B; a1 = b1 = 0
#define F16PREP "v_and_b32 v40, 0x3bff3bff, v40\n\tv_and_b32 v42, 0x3bff3bff, v42\n\tv_and_b32 v44, 0x3bff3bff, v44\n\tv_alignbit_b32 v46, v42, v42, 16\n\t"
      if (MODE == 9)
        asm volatile(LOADS F16PREP "v_pk_fma_f16 v48, v40, v42, v44 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\tv_pk_fma_f16 v50, v40, v46, v44\n\t"
                           "v_mov_b32 v49, 0\n\tv_mov_b32 v51, 0\n\t" OUTS
                     : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
      else if (MODE == 10)
        asm volatile(LOADS F16PREP "v_pk_mul_f16 v48, v40, v42 op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_mul_f16 v50, v40, v46\n\t"
                           "v_mov_b32 v49, 0\n\tv_mov_b32 v51, 0\n\t" OUTS
                     : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
      else
        asm volatile(LOADS F16PREP "v_pk_add_f16 v48, v40, v42 op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f16 v50, v40, v46\n\t"
                           "v_mov_b32 v49, 0\n\tv_mov_b32 v51, 0\n\t" OUTS
                     : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
    } else if (MODE == 8)
      asm volatile(LOADS "v_readfirstlane_b32 s20, v40\n\tv_readfirstlane_b32 s21, v41\n\ts_nop 4\n\t"
                         "v_pk_mul_f32 v[48:49], s[20:21], v[42:43] op_sel:[1,0]\n\t"
                         "v_mul_f32 v50, s21, v42\n\tv_mul_f32 v51, s21, v43\n\t" OUTS
                   : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB, "s20", "s21");
    else
      asm volatile(LOADS "v_pk_fma_f32 v[48:49], v[40:41], v[42:43], v[44:45] op_sel:[1,0,0]\n\t"
                         "v_fma_f32 v50, v41, v42, v44\n\tv_fma_f32 v51, v41, v43, v45\n\t" OUTS
                   : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1) : "v"(p) : CLOB);
    if (a0 != b0 || a1 != b1) {
      ++bad;
      blo += a0 != b0;
      bhi += a1 != b1;
      clo += (MODE == 3) ? ((a0 << 1) == 0u) : (a0 == __float_as_uint(p[1].x));   // MODE 3: the lo product is +-0
    }
    idx = (idx * 1664525u + 1013904223u + wg) % n8;
  }
  if (bad) {
    atomicAdd(&err[lane >> 4], bad);
    atomicAdd(&err[4], blo);
    atomicAdd(&err[5], bhi);
    atomicAdd(&err[6], clo);
    atomicAdd(&err[7], 1u);
  }
}

extern "C" int opsel_launch(int mode, const void* G, uint32_t n8, int B, int VH, int reps, int lds_bytes, void* err, void* stream) {
#define GO(M) hipLaunchKernelGGL(opsel_kernel<M>, dim3(VH, B), dim3(64), lds_bytes, (hipStream_t)stream, (const float4*)G, n8, reps, (unsigned*)err)
  switch (mode) {
    case 0: GO(0); break;
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 3: GO(3); break;
    case 4: GO(4); break;
    case 6: GO(6); break;
    case 7: GO(7); break;
    case 8: GO(8); break;
    case 9: GO(9); break;
    case 10: GO(10); break;
    case 11: GO(11); break;
    default: GO(5); break;
  }
  return (int)hipGetLastError();
}
