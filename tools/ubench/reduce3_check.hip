// Correctness check of the interleaved three-chain DPP wave reduction used by make_widepk (ffx_trace.hip:
// wave_reduce3_nn): four butterfly steps inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row
// results to lane 63.  Compared with a plain shuffle reduction on random unsigned data, min and max.
// build: hipcc --offload-arch=gfx950 -O3 -o reduce3_check reduce3_check.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define R3_STEP(OP, CTRL, MASK)                                              \
  OP " %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"             \
  OP " %1, %1, %1 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"             \
  OP " %2, %2, %2 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"
#define R3_ALL(OP)                                                                                                      \
  "s_nop 1\n\t" R3_STEP(OP, "quad_perm:[1,0,3,2]", "0xf") R3_STEP(OP, "quad_perm:[2,3,0,1]", "0xf") R3_STEP(OP, "row_half_mirror", "0xf") \
      R3_STEP(OP, "row_mirror", "0xf") R3_STEP(OP, "row_bcast:15", "0xa") R3_STEP(OP, "row_bcast:31", "0xc")

template <bool MAX>
__device__ __forceinline__ void wave_reduce3_nn(uint32_t &a, uint32_t &b, uint32_t &c) {
  if (MAX) asm(R3_ALL("v_max_u32_dpp") : "+v"(a), "+v"(b), "+v"(c));
  else asm(R3_ALL("v_min_u32_dpp") : "+v"(a), "+v"(b), "+v"(c));
  a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63);
  b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
  c = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
}

__global__ void k(const uint32_t *in, uint32_t *out, int n_waves) {
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= n_waves) return;
  uint32_t a = in[(w * 3 + 0) * 64 + lane], b = in[(w * 3 + 1) * 64 + lane], c = in[(w * 3 + 2) * 64 + lane];
  uint32_t x = a, y = b, z = c, p = a, q = b, r = c;
  wave_reduce3_nn<false>(x, y, z);
  wave_reduce3_nn<true>(p, q, r);
  if (lane == 0) {
    uint32_t *o = out + w * 6;
    o[0] = x; o[1] = y; o[2] = z; o[3] = p; o[4] = q; o[5] = r;
  }
}

int main() {
  const int n_waves = 4096;
  uint32_t *h = (uint32_t *)malloc(n_waves * 3 * 64 * 4), *ho = (uint32_t *)malloc(n_waves * 6 * 4), *d, *dout;
  srand(1);
  for (int i = 0; i < n_waves * 3 * 64; ++i) h[i] = ((uint32_t)rand() << 16) ^ (uint32_t)rand() ^ ((i % 7 == 0) ? 0xffffffffu : 0u);
  hipMalloc(&d, n_waves * 3 * 64 * 4);
  hipMalloc(&dout, n_waves * 6 * 4);
  hipMemcpy(d, h, n_waves * 3 * 64 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n_waves / 4), dim3(256), 0, 0, d, dout, n_waves);
  hipMemcpy(ho, dout, n_waves * 6 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < n_waves; ++w)
    for (int j = 0; j < 3; ++j) {
      uint32_t mn = 0xffffffffu, mx = 0;
      for (int l = 0; l < 64; ++l) { uint32_t v = h[(w * 3 + j) * 64 + l]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
      if (ho[w * 6 + j] != mn || ho[w * 6 + 3 + j] != mx) ++bad;
    }
  printf("reduce3_check: %d waves x 3 chains x (min, max): %d mismatches\n", n_waves, bad);
  return bad != 0;
}
