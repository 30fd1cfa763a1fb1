// r06: the reduced victim as a stand-alone kernel.  Per (sample, head) workgroup and query row: load a 3x3 rotation R, a translation
// t and two 3-vectors from global memory, store R v (+ 0 * (R u + t), which keeps the second rotation alive) — what is left of
// geom_attention_kernel's query-side prologue when the key fill and the key walk are removed (scratch/ubench/isa/mk.py: e2).  hipcc's
// SLP vectoriser turns the rotations into v_pk_mul_f32 / v_pk_fma_f32 / v_pk_mov_b32; -fno-slp-vectorize gives the plain form.
// A run beside the 256x256 GEMM on another stream must equal the run alone bit for bit (scratch/r06_pk_rotate.py); with packed ops
// it does not: lanes 48-63 of some waves come out wrong.
//   NT     threads per workgroup (64: one wave, as the product kernel; 256: four waves, each wave its own 64 rows)
//   reps   the whole row loop is repeated `reps` times into out[rep] (a long-lived wave: WHEN in its life do the errors fall?)
//   hipcc --offload-arch=gfx950 -O3 -fno-fast-math -shared -fPIC scratch/ubench/pk_rotate.hip -o scratch/ubench/pk_rotate.so
//   hipcc ... -fno-slp-vectorize -fno-vectorize -DPR_NAME=nopk -o scratch/ubench/pk_rotate_nopk.so
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int NT>
__global__ __launch_bounds__(NT) void rotate_kernel(const float* __restrict__ P, const float* __restrict__ rot, const float* __restrict__ trans,
                                                    const uint8_t* __restrict__ fmask, float* __restrict__ out, int L, int VH, int reps,
                                                    int64_t rep_stride) {
  extern __shared__ float lds[];
  const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
  const int ldp = 15 * VH;
  const int64_t row0 = (int64_t)b * L;
  if (lane == 0) lds[0] = 0.f;   // (the LDS request only steers co-residency)
  for (int rep = 0; rep < reps; ++rep) {
    for (int q = lane; q < L; q += NT) {
      const int64_t row = row0 + q;
      float R[9], t[3], v[3], u[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = rot[row * 9 + i];
#pragma unroll
      for (int i = 0; i < 3; ++i) t[i] = trans[row * 3 + i];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        v[i] = P[row * ldp + 3 * h + i];
        u[i] = P[row * ldp + 9 * VH + 3 * h + i];
      }
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
      if (fmask[row]) {
        const float a0 = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], a1 = R[3] * v[0] + R[4] * v[1] + R[5] * v[2], a2 = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
        const float d0 = R[0] * u[0] + R[1] * u[1] + R[2] * u[2] + t[0], d1 = R[3] * u[0] + R[4] * u[1] + R[5] * u[2] + t[1],
                    d2 = R[6] * u[0] + R[7] * u[1] + R[8] * u[2] + t[2];
        r0 = a0 + 0.f * d0; r1 = a1 + 0.f * d1; r2 = a2 + 0.f * d2;
      }
      float* dst = out + rep * rep_stride + row * (3 * VH) + 3 * h;
      dst[0] = r0; dst[1] = r1; dst[2] = r2;
    }
  }
}

#ifndef PR_NAME
#define PR_NAME pk
#endif
#define PR_CAT2(a, b) a##b
#define PR_CAT(a, b) PR_CAT2(a, b)
extern "C" int PR_CAT(pr_rotate_, PR_NAME)(const float* P, const float* rot, const float* trans, const uint8_t* fmask, float* out, int B, int L, int VH,
                                        int reps, int64_t rep_stride, int nt, int lds_bytes, void* stream) {
  if (nt == 64) hipLaunchKernelGGL(rotate_kernel<64>, dim3(VH, B), dim3(64), lds_bytes, (hipStream_t)stream, P, rot, trans, fmask, out, L, VH, reps, rep_stride);
  else hipLaunchKernelGGL(rotate_kernel<256>, dim3(VH, B), dim3(256), lds_bytes, (hipStream_t)stream, P, rot, trans, fmask, out, L, VH, reps, rep_stride);
  return (int)hipGetLastError();
}
