// r06: geom_attention_kernel's key walk as a stand-alone probe with full-precision per-lane outputs (running maximum, denominator,
// three accumulators), compiled twice — as hipcc compiles it (SLP-vectorised: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) and
// with -fno-slp-vectorize -fno-vectorize (no packed float ops) — to be run alone and beside the 256x256 GEMM on another stream
// (scratch/r06_pk_neighbour.py): a run beside the GEMM must equal the run alone bit for bit.
//
//   hipcc --offload-arch=gfx950 -O3 -fno-fast-math -shared -fPIC scratch/ubench/pk_neighbour.hip -o scratch/ubench/pk_neighbour.so
//   hipcc ... -fno-slp-vectorize -fno-vectorize -DNB_NAME=nopk ... -o scratch/ubench/pk_neighbour_nopk.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}
__device__ __forceinline__ float unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }   // [-1, 1)

// out[(wg * L + q) * 8 + {0..4}] = m, den, o0, o1, o2
__global__ __launch_bounds__(64) void walk_kernel(float* __restrict__ out, int L, int VH, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) float kl[];
  const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
  const uint32_t wg = (uint32_t)(b * VH + h);
  const int Lk = (L + 3) & ~3;
  for (int l = lane; l < Lk; l += 64) {
    float* k = kl + l * 12;
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = unit(mix(wg * 0x9e3779b9u + l * 16 + j)) * (j >= 3 && j < 6 ? 20.f : 1.f);
    k[9] = (l < L && (l % 37) != 5) ? 1.0f : 0.0f;
    k[10] = k[11] = 0.0f;
  }
  __syncthreads();
  const float wr = 0.83f, wd = 0.41f;
  for (int q = lane; q < L; q += 64) {
    float qr[3], qd[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      qr[j] = unit(mix(wg * 0x85ebca6bu + q * 8 + j));
      qd[j] = unit(mix(wg * 0x85ebca6bu + q * 8 + 4 + j)) * 20.f;
    }
    float m = -3.0e38f, den = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if ((q % 41) != 7) {
      for (int k = 0; k < Lk; k += 4) {
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* kk = kl + (k + u) * 12;
          const float dx = qd[0] - kk[3], dy = qd[1] - kk[4], dz = qd[2] - kk[5];
          const float v = wr * (qr[0] * kk[0] + qr[1] * kk[1] + qr[2] * kk[2]) - wd * __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
          sc[u] = kk[9] != 0.0f ? v : -__builtin_inff();
        }
        const float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        if (mx > m) {
          const float a = __builtin_amdgcn_exp2f(m - mx);
          den *= a; o0 *= a; o1 *= a; o2 *= a;
          m = mx;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* kk = kl + (k + u) * 12;
          const float p = __builtin_amdgcn_exp2f(sc[u] - m);
          den += p;
          o0 += p * kk[6]; o1 += p * kk[7]; o2 += p * kk[8];
        }
      }
    }
    float* dst = out + ((size_t)wg * L + q) * 8;
    dst[0] = m; dst[1] = den; dst[2] = o0; dst[3] = o1; dst[4] = o2;
  }
}

#ifndef NB_NAME
#define NB_NAME pk
#endif
#define NB_CAT2(a, b) a##b
#define NB_CAT(a, b) NB_CAT2(a, b)
extern "C" int NB_CAT(nb_walk_, NB_NAME)(void* out, int B, int L, int VH, int lds_bytes, void* stream) {
  const int need = ((L + 3) & ~3) * 12 * 4;
  if (lds_bytes < need) lds_bytes = need;
  hipError_t r = hipFuncSetAttribute((const void*)walk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  if (r != hipSuccess) return (int)r;
  hipLaunchKernelGGL(walk_kernel, dim3(VH, B), dim3(64), lds_bytes, (hipStream_t)stream, (float*)out, L, VH, 1);
  return (int)hipGetLastError();
}
