// convert.hip — weight preparation at engine create (gfx950): dtype conversion and the row interleave
// that lets the SwiGLU epilogue of the FFN-up GEMM pair gate and up columns inside one wave's registers.
#include "ed_half.h"
#include "kernels.h"

namespace ed {

// f32 -> the TU's 16-bit operand type, round to nearest even (ed_half.h: bf16, or f16 — saturating — in the ed16 build)
__device__ __forceinline__ uint32_t cv_f2bf(float a) { return ed_f2h(a); }
// a source tensor tagged ESMDIFF_BF16 is bfloat16 in BOTH builds (the caller's dtype, not the engine's)
__device__ __forceinline__ uint32_t cv_bf2h(bf16_t b) {
#ifdef ED_F16
  return ed_f2h(__uint_as_float((uint32_t)b << 16));
#else
  return b;
#endif
}

__global__ void to_bf16_kernel(const void* __restrict__ src, int src_dtype, bf16_t* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src_dtype == ESMDIFF_F32 ? (bf16_t)cv_f2bf(reinterpret_cast<const float*>(src)[i])
                                      : (bf16_t)cv_bf2h(reinterpret_cast<const bf16_t*>(src)[i]);
}

__global__ void to_f32_kernel(const void* __restrict__ src, int src_dtype, float* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src_dtype == ESMDIFF_F32 ? reinterpret_cast<const float*>(src)[i]
                                      : __uint_as_float((uint32_t)reinterpret_cast<const bf16_t*>(src)[i] << 16);
}

// dst row rho: block = rho/32, hidden = (block/2)*32 + rho%32, source row = hidden + (block&1 ? H : 0)
__global__ void interleave_swiglu_kernel(const void* __restrict__ src, int src_dtype, bf16_t* __restrict__ dst, int H,
                                         int K) {
  const int rho = blockIdx.x;
  const int blk = rho >> 5;
  const int srow = (blk >> 1) * 32 + (rho & 31) + ((blk & 1) ? H : 0);
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const int64_t si = (int64_t)srow * K + k;
    dst[(int64_t)rho * K + k] = src_dtype == ESMDIFF_F32 ? (bf16_t)cv_f2bf(reinterpret_cast<const float*>(src)[si])
                                                         : (bf16_t)cv_bf2h(reinterpret_cast<const bf16_t*>(src)[si]);
  }
}

hipError_t launch_to_bf16(const void* src, int src_dtype, bf16_t* dst, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(to_bf16_kernel, dim3(blocks), dim3(256), 0, stream, src, src_dtype, dst, n);
  return hipGetLastError();
}
hipError_t launch_to_f32(const void* src, int src_dtype, float* dst, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(to_f32_kernel, dim3(blocks), dim3(256), 0, stream, src, src_dtype, dst, n);
  return hipGetLastError();
}
hipError_t launch_interleave_swiglu(const void* src, int src_dtype, bf16_t* dst, int H, int K, hipStream_t stream) {
  if (H % 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(interleave_swiglu_kernel, dim3(2 * H), dim3(256), 0, stream, src, src_dtype, dst, H, K);
  return hipGetLastError();
}

}  // namespace ed
