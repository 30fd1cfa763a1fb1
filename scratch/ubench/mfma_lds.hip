// micro-benchmark: matrix-pipe throughput vs interleaved ds_read_b128, destination/accumulator register class
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int R, int DST_AGPR, int ACC_AGPR>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + threadIdx.x * 16;
  for (int i = threadIdx.x; i < 8192; i += 512) ((float*)smem)[i] = 1.0f;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)1.0f; b[e] = (__bf16)(lane * 0.001f); }
  u32x4 d[4];
  for (int j = 0; j < 4; ++j) d[j] = u32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (DST_AGPR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d[r]) : "v"(base), "i"(r * 8192));
      else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[r]) : "v"(base), "i"(r * 8192));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ACC_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int j = 0; j < 4; ++j) s += (float)(d[j][0] & 1u);
  if (s == 12345.678f) out[0] = s;
}

template <int R, int D, int A>
static float run(int iters) {
  float* o; hipMalloc(&o, 4);
  hipFuncSetAttribute((const void*)k<R, D, A>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<R, D, A>), dim3(256), dim3(512), 65536, 0, o, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<R, D, A>), dim3(256), dim3(512), 65536, 0, o, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(o);
  return ms;
}

extern "C" void ubench(int iters, float* ms /*[5][2][2]*/) {
#define RUN(R) ms[(R)*4+0] = run<R,0,0>(iters); ms[(R)*4+1] = run<R,0,1>(iters); ms[(R)*4+2] = run<R,1,0>(iters); ms[(R)*4+3] = run<R,1,1>(iters);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
}
