// r06: a neighbour with the 256x256 GEMM's footprint on a CU (256 threads, 256 accumulator registers, 128 KB of LDS requested, one
// workgroup per CU) that either executes MFMAs (the GEMM's 16-per-k-step pattern on register operands) or only holds the registers,
// for `iters` k-steps — to separate "an MFMA wave EXECUTING on the SIMD" from "a big wave being LAUNCHED on the SIMD" as the thing
// that disturbs a co-resident wave's packed float ops (scratch/r06_geom_isa.py NEIGHBOUR=...).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/ubench/mfma_neighbour.hip -o scratch/ubench/mfma_neighbour.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>   // 1: MFMA   0: hold the accumulators, s_nop in place of each MFMA   2: MFMA on VGPR accumulators? (not built)
__global__ __launch_bounds__(256) void nb_kernel(float* out, int iters) {
  const int lane = threadIdx.x & 63;
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
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(B[j]), "v"(A[i]));
        else asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[i][j]) : "v"(B[j]), "v"(A[i]));
      }
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
  const void* fn = mode == 1 ? (const void*)nb_kernel<1> : (const void*)nb_kernel<0>;
  hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (r != hipSuccess) return (int)r;
  if (mode == 1) hipLaunchKernelGGL(nb_kernel<1>, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, (float*)out, iters);
  else hipLaunchKernelGGL(nb_kernel<0>, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, (float*)out, iters);
  return (int)hipGetLastError();
}
