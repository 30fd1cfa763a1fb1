// r06: minimal victim?  One-wave workgroups; each lane loads 16 bytes from global memory, waits for all of it, multiplies the two
// halves with ONE packed float op and with two plain float ops, and compares the results bit for bit — `reps` times at different
// addresses.  err[q] counts mismatches of lane quarter q, err[4 + trip range] when they happened.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/ubench/pk_minimal.hip -o scratch/ubench/pk_minimal.so
// (registers v40..v51: with v100.. the kernel allocates 112 registers and can NOT share a SIMD with a 408-register GEMM wave — the first
// version of this probe could never fail)
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void pkmin_kernel(const float4* __restrict__ G, uint32_t n4, int reps, unsigned* err) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const uint32_t wg = blockIdx.y * gridDim.x + blockIdx.x;
  lds[lane] = (float)lane;   // (the LDS request only steers co-residency)
  uint32_t idx = (wg * 64u + lane) % n4;
  unsigned bad = 0, first_rep = 0xffffffffu;
  for (int r = 0; r < reps; ++r) {
    const float4* p = G + idx;
    uint32_t a0, a1, b0, b1;
    asm volatile(
        "global_load_dwordx4 v[40:43], %4, off\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_pk_mul_f32 v[44:45], v[40:41], v[42:43]\n\t"
        "v_mul_f32 v46, v40, v42\n\t"
        "v_mul_f32 v47, v41, v43\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %0, v44\n\t"
        "v_mov_b32 %1, v45\n\t"
        "v_mov_b32 %2, v46\n\t"
        "v_mov_b32 %3, v47\n\t"
        : "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1)
        : "v"(p)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
    if (a0 != b0 || a1 != b1) {
      ++bad;
      if (first_rep == 0xffffffffu) first_rep = r;
    }
    idx = (idx * 1664525u + 1013904223u + wg) % n4;
  }
  if (bad) {
    atomicAdd(&err[lane >> 4], bad);
    atomicAdd(&err[4 + min(first_rep * 8 / (unsigned)reps, 7u)], 1u);
    atomicAdd(&err[12], 1u);
  }
  if (lds[(lane + 1) & 63] < -1.f) err[15] = 1;
}

extern "C" int pkmin_launch(const void* G, uint32_t n4, int B, int VH, int reps, int lds_bytes, void* err, void* stream) {
  hipError_t r = hipFuncSetAttribute((const void*)pkmin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  if (r != hipSuccess) return (int)r;
  hipLaunchKernelGGL(pkmin_kernel, dim3(VH, B), dim3(64), lds_bytes, (hipStream_t)stream, (const float4*)G, n4, reps, (unsigned*)err);
  return (int)hipGetLastError();
}
