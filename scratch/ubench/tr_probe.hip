#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const unsigned short* in, unsigned short* out, const int* addr) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  std::vector<unsigned short> h(4096); for (int i = 0; i < 4096; ++i) h[i] = i;
  std::vector<int> a(64);
  // V tile [key][64 d], 128 B per key.  lane l: group g=l>>4, i=l&15: key = 4*(g>>1) + (i>>2), d = 16*(g&1) + 4*(i&3)
  for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; a[l] = (4 * (g >> 1) + (i >> 2)) * 128 + (16 * (g & 1) + 4 * (i & 3)) * 2; }
  unsigned short *din, *dout; int* da;
  hipMalloc(&din, 8192); hipMalloc(&dout, 512); hipMalloc(&da, 256);
  hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, dout, da);
  std::vector<unsigned short> o(256); hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (k%d,d%d)", o[l*4+j] / 64, o[l*4+j] % 64); printf("\n"); }
  return 0;
}
