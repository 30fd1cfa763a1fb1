// r06: the product's geom.hip as a stand-alone library (its own launcher, a 12.5 KB LDS request), to run geom_attention_kernel beside
// the 256x256 GEMM outside the engine (scratch/r06_geom_alone.py).
//   hipcc <esmdiff_amd.build COMMON flags> -Iinclude -Iesmdiff_amd/csrc -DED_GEOM_MIN_LDS_KB=0 -shared scratch/ubench/geom_alone.hip -o scratch/ubench/geom_alone.so
//   ... -fno-slp-vectorize -fno-vectorize -DGA_NAME=nopk -o scratch/ubench/geom_alone_nopk.so
#include "../../esmdiff_amd/csrc/geom.hip"
#ifndef GA_NAME
#define GA_NAME pk
#endif
#define GA_CAT2(a, b) a##b
#define GA_CAT(a, b) GA_CAT2(a, b)
extern "C" int GA_CAT(ga_geom_f32_, GA_NAME)(const float* P, const float* rot, const float* trans, const uint8_t* fmask, const float* w_rot,
                                          const float* w_dist, float* out, int B, int L, int VH, void* stream) {
  return (int)ed::launch_geom_attention_f32(P, rot, trans, fmask, w_rot, w_dist, out, B, L, VH, (hipStream_t)stream);
}
extern "C" int GA_CAT(ga_geom_bf16_, GA_NAME)(const void* P, const float* rot, const float* trans, const uint8_t* fmask, const float* w_rot,
                                           const float* w_dist, void* out, int B, int L, int VH, void* stream) {
  return (int)ed::launch_geom_attention((const bf16_t*)P, rot, trans, fmask, w_rot, w_dist, (bf16_t*)out, B, L, VH, (hipStream_t)stream);
}
