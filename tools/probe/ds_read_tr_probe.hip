#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int *out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  // every 16-lane group reads 128 contiguous bytes (64 elements): lane l' at element 4 l' of its group's tile
  const int grp = lane >> 4, lg = lane & 15;
  const unsigned short *p = lds + grp * 256 + 4 * lg;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
  int *d; hipMalloc(&d, 64 * 4 * 4);
  k<<<1, 64>>>(d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  return 0;
}
