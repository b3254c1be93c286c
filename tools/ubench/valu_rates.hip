// Issue-rate microbenchmark for the VALU instruction kinds used by the traversal kernels (gfx950).
// Every wave runs ITER iterations of 32 independent instructions of one kind; 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, float s0, float s1) {
  float a[16], b[16];
  float2_ p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = a[i] * 0.5f + 1.0f; p[i] = (float2_){a[i], b[i]}; }
  float ss0 = __builtin_amdgcn_readfirstlane(__float_as_int(s0)) ? s0 : s1; // keep in SGPR
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) { a[i] = fmaf(a[i], b[i], 1.0f); b[i] = fmaf(b[i], a[i], 0.5f); }                    // v_fma_f32, VGPR operands
      if (KIND == 1) { a[i] = fmaf(ss0, a[i], b[i]); b[i] = fmaf(ss0, b[i], a[i]); }                      // v_fma_f32 with an SGPR operand
      if (KIND == 2) { p[i] = __builtin_elementwise_fma(p[i], p[i], (float2_){1.0f, 0.5f}); p[i] = __builtin_elementwise_fma(p[i], (float2_){0.9f, 0.8f}, p[i]); } // v_pk_fma_f32
      if (KIND == 3) { a[i] = fminf(a[i], b[i]) ; b[i] = fmaxf(b[i], a[i] + 0.f); }                        // v_min / v_max (+add)
      if (KIND == 4) { a[i] = (a[i] > b[i]) ? b[i] : a[i] + 1.0f; b[i] = (b[i] < a[i]) ? a[i] : b[i]; }   // v_cmp + v_cndmask
      if (KIND == 5) { a[i] = (ss0 - a[i]) * b[i]; b[i] = (ss0 - b[i]) * a[i]; }                          // v_sub (SGPR) + v_mul
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += a[i] + b[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
void run(const char *name, int per_iter_instr, int flops_per_instr, float *d) {
  int blocks = 256 * 8; // 8 waves per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.9999f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.9999f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  double winstr = (double)blocks * 4 * ITER * per_iter_instr; // wave-instructions
  double per_simd_cycles = ms * 1e-3 * 2.4e9;                 // at nominal 2.4 GHz
  double instr_per_simd = winstr / 1024.0;
  printf("%-34s %8.3f ms  %6.2f cycles/wave-instr/SIMD (nominal clock)  %7.1f TFLOP/s\n", name, ms, per_simd_cycles / instr_per_simd,
         winstr * 64 * flops_per_instr / (ms * 1e-3) / 1e12);
}

int main() {
  float *d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("v_fma_f32 (VGPR operands)", 32, 2, d);
  run<1>("v_fma_f32 (one SGPR operand)", 32, 2, d);
  run<2>("v_pk_fma_f32", 32, 4, d);
  run<3>("v_min/v_max/v_add mix", 48, 1, d);
  run<4>("v_cmp + v_cndmask (+add)", 80, 1, d);
  run<5>("v_sub(SGPR) + v_mul", 64, 1, d);
  return 0;
}
