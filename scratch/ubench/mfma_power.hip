// What does the package power cap allow the matrix pipes alone?  One wave per SIMD (4 per CU, 256 CUs), the four-wave
// GEMM's inner pattern without its global side: per k-step 16 v_mfma_f32_32x32x16_bf16 on 16 accumulators (4 x 4), operands
//   mode 0  zero operands held in registers
//   mode 1  random bf16 operands held in registers (8 fragments, reused every k-step)
//   mode 2  random operands re-read from LDS every k-step: 8 ds_read_b128 per 16 MFMAs (the GEMM's ratio), addresses
//           walking through 64 KB of random data
// Runs each mode for ~`secs` seconds in back-to-back launches and prints TFLOP/s per launch; sample rocm-smi beside it
// (scratch/ubench/mfma_power.sh).     hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 64 KB
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) {
    uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // two random bf16 with magnitudes in [0.25, 2): random sign, random mantissa, exponent 0x3e8..0x3ff
    const uint32_t lo = (h & 0x807f) | 0x3f00, hi = ((h >> 16) & 0x807f) | 0x3e80;
    ((uint32_t*)smem)[i] = MODE == 0 ? 0u : (lo | (hi << 16));
  }
  __syncthreads();
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const int frow = lane & 31, khalf = lane >> 5, fsw = (frow >> 1) & 7;
  uint32_t off = lds0 + frow * 128 + ((khalf ^ fsw) << 4);
  bf16x8 A[4], B[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A[i] = *reinterpret_cast<const bf16x8*>(smem + (frow + 32 * i) * 128 + (khalf << 4));
    B[i] = *reinterpret_cast<const bf16x8*>(smem + 32768 + (frow + 32 * i) * 128 + (khalf << 4));
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
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE == 2) {
        const uint32_t a = off ^ (((it * 4 + ks) & 7) << 4);   // another 16-byte chunk of the same rows each k-step
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[i]) : "v"(a), "i"(i * 4096));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(B[i]) : "v"(a), "i"(32768 + i * 4096));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B[j], A[i], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(float secs, float* o) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000;  // 40000 x 64 MFMAs x 32 cycles = 82 M cycles ~ 45 ms per launch
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 65536, 0, o, 1000);
  hipDeviceSynchronize();
  float total = 0.f;
  int n = 0;
  double last = 0;
  while (total < secs * 1000.f) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 65536, 0, o, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    total += ms;
    ++n;
    last = 256.0 * 4 * iters * 64 * 32768.0 / (ms * 1e-3) / 1e12;
    if (n <= 3 || n % 10 == 0) printf("mode %d launch %3d: %7.2f ms  %7.1f TFLOP/s\n", MODE, n, ms, last);
  }
  printf("mode %d: last launch %7.1f TFLOP/s after %.1f s\n", MODE, last, total / 1000.f);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const float secs = argc > 1 ? atof(argv[1]) : 3.0f;
  float* o;
  hipMalloc(&o, 4);
  run<0>(secs, o);
  run<1>(secs, o);
  run<2>(secs, o);
  return 0;
}
