// r06: which side of "one-wave LDS kernel next to a 128 KB LDS-DMA GEMM workgroup on the same CU" is the victim?
// A checker kernel with geom_attention_kernel's shape (64-thread workgroups, [Lk][12] floats of dynamic LDS written once and then
// read as three 16-byte broadcasts per key, 2-byte scalar global reads at a 6-byte stride), except that every word it reads — from
// global memory and from its own LDS — is a function of its index, so each read is verified on the spot.  Run on one stream while
// the real 256x256 GEMM runs on another (scratch/r06_lds_neighbour.py), which verifies the GEMM's output against a solo run.
//
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/ubench/lds_neighbour.hip -o scratch/ubench/lds_neighbour.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}
__device__ __forceinline__ uint16_t gword(uint64_t i) { return (uint16_t)(mix((uint32_t)i ^ (uint32_t)(i >> 32) * 0x9e3779b9u) >> 7); }

__global__ void fill_kernel(uint16_t* G, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) G[i] = gword(i);
}

// err[0]: wrong global words, err[1]: wrong LDS words, err[2..7]: first wrong LDS word (wg, key, word, got, expected, rep)
__global__ __launch_bounds__(64) void checker_kernel(const uint16_t* __restrict__ G, int ldp, int L, int VH, int reps, unsigned* err) {
  extern __shared__ __attribute__((aligned(16))) uint32_t kl[];
  const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
  const uint32_t wg = (uint32_t)(b * VH + h);
  const int Lk = (L + 3) & ~3;
  unsigned ge = 0, le = 0;
  for (int rep = 0; rep < reps; ++rep) {
    const uint32_t salt = mix(wg * 0x9e3779b9u + rep);
    for (int l = lane; l < Lk; l += 64) {
      if (l < L) {
        const uint64_t row = (uint64_t)b * L + l;
#pragma unroll
        for (int blk = 0; blk < 3; ++blk) {
          const uint64_t i0 = row * ldp + (blk * 3 + 3) * VH + 3 * h;
#pragma unroll
          for (int c = 0; c < 3; ++c) ge += G[i0 + c] != gword(i0 + c);
        }
      }
#pragma unroll
      for (int j = 0; j < 12; ++j) kl[l * 12 + j] = mix(salt + l * 12 + j);
    }
    __syncthreads();
    for (int q = lane; q < L; q += 64) {
      for (int k = 0; k < Lk; ++k) {
        const uint4 a = *reinterpret_cast<const uint4*>(kl + k * 12), c = *reinterpret_cast<const uint4*>(kl + k * 12 + 4),
                    d = *reinterpret_cast<const uint4*>(kl + k * 12 + 8);
        const uint32_t got[12] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const uint32_t want = mix(salt + k * 12 + j);
          if (got[j] != want) {
            if (le == 0 && atomicAdd(&err[8], 1u) == 0) {
              err[2] = wg; err[3] = k; err[4] = j; err[5] = got[j]; err[6] = want; err[7] = rep;
            }
            ++le;
          }
        }
      }
    }
    __syncthreads();
  }
  if (ge) atomicAdd(&err[0], ge);
  if (le) atomicAdd(&err[1], le);
}

// the same occupancy and VALU load without any LDS (control for "the GEMM is disturbed by a neighbour as such")
__global__ __launch_bounds__(64) void nolds_kernel(int L, int reps, unsigned* sink) {
  uint32_t a = blockIdx.x * 977 + blockIdx.y * 131 + threadIdx.x;
  for (int rep = 0; rep < reps; ++rep)
    for (int q = threadIdx.x; q < L; q += 64)
      for (int k = 0; k < L * 12; ++k) a = mix(a + k);
  if (a == 0x12345678u) sink[9] = a;
}

extern "C" int nb_fill(void* G, uint64_t n, void* stream) {
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, (uint16_t*)G, n);
  return (int)hipGetLastError();
}
extern "C" int nb_checker(const void* G, int ldp, int B, int L, int VH, int lds_bytes, int reps, void* err, void* stream) {
  const int need = ((L + 3) & ~3) * 12 * 4;
  if (lds_bytes < need) lds_bytes = need;
  hipError_t r = hipFuncSetAttribute((const void*)checker_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  if (r != hipSuccess) return (int)r;
  hipLaunchKernelGGL(checker_kernel, dim3(VH, B), dim3(64), lds_bytes, (hipStream_t)stream, (const uint16_t*)G, ldp, L, VH, reps, (unsigned*)err);
  return (int)hipGetLastError();
}
extern "C" int nb_nolds(int B, int L, int VH, int reps, void* err, void* stream) {
  hipLaunchKernelGGL(nolds_kernel, dim3(VH, B), dim3(64), 0, (hipStream_t)stream, L, reps, (unsigned*)err);
  return (int)hipGetLastError();
}
