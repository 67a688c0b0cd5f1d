// checks wave_transpose_reduce32 (nf_simple.hip, k_radial_step): every lane of a 32-lane half-wave holds 32 values; on
// return lane l31 holds the sum over the half's 32 lanes of value number l31.  Five stages: v_permlane16_swap (rows), then
// DPP row_ror:8, row_half_mirror, quad_perm xor 2, xor 1 -- 77 instructions for 32 x 32 values.
// build + run: hipcc --offload-arch=gfx950 -O2 tools/probe/transpose_reduce_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float dppf(float v, int ctrl) {
  switch (ctrl) {
    case 0: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));  // row_ror:8
    case 1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    case 2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
  }
}
__device__ __forceinline__ float wave_transpose_reduce32(const float (&v)[32], int lane) {
  float w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    unsigned a = __builtin_bit_cast(unsigned, v[i]), b = __builtin_bit_cast(unsigned, v[i + 16]);
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    unsigned r0 = r[0], r1 = r[1];
    asm volatile("s_nop 1" : "+v"(r0), "+v"(r1));
    w[i] = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
  }
  float u[8], t[4], s[2];
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = (b3 ? w[i + 8] : w[i]) + dppf(b3 ? w[i] : w[i + 8], 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = (b2 ? u[i + 4] : u[i]) + dppf(b2 ? u[i] : u[i + 4], 1);
#pragma unroll
  for (int i = 0; i < 2; ++i) s[i] = (b1 ? t[i + 2] : t[i]) + dppf(b1 ? t[i] : t[i + 2], 2);
  return (b0 ? s[1] : s[0]) + dppf(b0 ? s[0] : s[1], 3);
}

__global__ void k(float *o) {
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = 1000.f * i + l31 + 0.25f * hi;
  o[lane] = wave_transpose_reduce32(v, lane);
}
int main() {
  float *d, h[64];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const float want = 32.f * 1000.f * (l & 31) + 496.f + 32 * 0.25f * (l >> 5);
    if (h[l] != want) { ++bad; printf("lane %d: got %.2f want %.2f\n", l, h[l], want); }
  }
  printf("%s\n", bad ? "MISMATCH" : "transpose-reduce ok: lane l31 holds the half-wave sum of value l31");
  return bad != 0;
}
