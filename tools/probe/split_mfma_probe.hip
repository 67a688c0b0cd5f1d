// Probe (design aid, not part of the library): is the split whose two subtractions run on the matrix pipe (nf_split16_mfma,
// nf_mfma.h: D = x - h as C + (-selection) x h) BIT-IDENTICAL to the vector-instruction split nf_split2 / nf_split8?
// Every wave takes 32 x 32 blocks of inputs -- normal draws over 80 binades, values on and next to bf16 rounding ties,
// values whose residual is subnormal, +-0, the largest finite floats -- splits them both ways and counts differing words.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I normalizingflows.jl_amd/csrc -I include tools/probe/split_mfma_probe.hip -o tools/probe/split_mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "nf_common.h"
#include "nf_mfma.h"

__global__ void k(const float *in, int nblocks, unsigned long long *bad, unsigned *first) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const SplitSel sel = nf_split_sel(l31, hi);
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    f32x16 x;
    for (int r = 0; r < 16; ++r) x[r] = in[((size_t)b * 16 + r) * 64 + lane];
    nf_u32x4 h[2], m[2], l[2], h2[2], m2[2], l2[2];
    nf_split16_mfma(sel, x, h, m, l);
    for (int g = 0; g < 2; ++g) {
      float v[8];
      for (int j = 0; j < 8; ++j) v[j] = x[8 * g + j];
      nf_split8(v, h2[g], m2[g], l2[g]);
    }
    for (int g = 0; g < 2; ++g)
      for (int p = 0; p < 4; ++p) {
        const bool ne = h[g][p] != h2[g][p] || m[g][p] != m2[g][p] || l[g][p] != l2[g][p];
        if (ne) {
          if (atomicAdd(bad, 1ull) == 0) {
            first[0] = __float_as_uint(x[8 * g + 2 * p]); first[1] = __float_as_uint(x[8 * g + 2 * p + 1]);
            first[2] = h[g][p]; first[3] = h2[g][p]; first[4] = m[g][p]; first[5] = m2[g][p]; first[6] = l[g][p]; first[7] = l2[g][p];
          }
        }
      }
  }
}

int main() {
  const int nblocks = 1 << 14;  // 16 M values
  std::vector<float> x((size_t)nblocks * 1024);
  std::mt19937_64 g(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_int_distribution<int> ex(-60, 40), kind(0, 15);
  std::uniform_int_distribution<unsigned> bits(0, 0xFFFFFFFFu);
  for (size_t i = 0; i < x.size(); ++i) {
    const int kd = kind(g);
    float v;
    if (kd < 9) v = std::ldexp(nd(g), ex(g));
    else if (kd < 12) {  // on / next to a bf16 tie: low 16 bits 0x8000 +- {0, 1}, also after the first level (0x..80 +- 1 patterns)
      unsigned b = bits(g) & 0x7FFFFFFFu;
      if ((b >> 23) >= 0xFD) b &= 0x7E7FFFFFu;
      b = (b & 0xFFFF0000u) | (0x8000u + (unsigned)(int)(bits(g) % 3) - 1u);
      if (bits(g) & 1) b = (b & 0xFFFFFF00u) | (0x80u + (unsigned)(int)(bits(g) % 3) - 1u);
      if (bits(g) & 1) b |= 0x80000000u;
      memcpy(&v, &b, 4);
    } else if (kd == 12) {  // tiny: the residuals fall into the subnormal range
      v = std::ldexp(nd(g), -126 + (int)(bits(g) % 12));
    } else if (kd == 13) {  // huge, but below the largest bf16 (a part that rounds to infinity poisons every row of its column in
      // the matrix-pipe form -- 0 x inf -- and its own element in the vector form: neither splits such a value)
      unsigned b = 0x7E800000u | (bits(g) & 0x007FFFFFu) | ((bits(g) & 1) << 31);
      memcpy(&v, &b, 4);
    } else if (kd == 14) v = (bits(g) & 1) ? 0.f : -0.f;
    else {  // arbitrary finite bit pattern
      unsigned b = bits(g);
      if (((b >> 23) & 0xFF) >= 0xFE) b &= 0xBFFFFFFFu;
      memcpy(&v, &b, 4);
    }
    x[i] = v;
  }
  float *d;
  unsigned long long *bad;
  unsigned *first;
  hipMalloc(&d, x.size() * 4);
  hipMalloc(&bad, 8);
  hipMalloc(&first, 32);
  hipMemset(bad, 0, 8);
  hipMemcpy(d, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  k<<<1024, 64>>>(d, nblocks, bad, first);
  unsigned long long nb = 0;
  unsigned f[8];
  hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
  printf("split on the matrix pipe vs nf_split2: %llu differing packed words of %zu\n", nb, x.size() / 2);
  if (nb) printf("first: x = %08x %08x  h %08x / %08x  m %08x / %08x  l %08x / %08x (mfma / valu)\n", f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
  return nb ? 1 : 0;
}
