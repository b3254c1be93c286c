// Would a node step that tests both child boxes with packed fp32 be cheaper?  (gfx950)
// A: 12 v_fma_f32 (node plane = SGPR operand) + 2 v_max3 + 2 v_min3          — today's step
// B:  6 v_pk_fma_f32 (two children's planes = SGPR pair, ray constants broadcast with op_sel) + the same 4
// Each wave runs ITER iterations; 8 waves per SIMD (2048 workgroups x 256 threads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, float s0, float s1, float s2, float s3) {
  float t = threadIdx.x * 0.001f;
  f2 m01 = {1.0f + t, 1.1f + t}, m23 = {1.2f + t, 1.3f + t}, c01 = {0.1f + t, 0.2f + t}, c23 = {0.3f + t, 0.4f + t};
  float acc0 = 0.f, acc1 = 0.f;
  f2 sp0 = {s0, s1}, sp1 = {s2, s3};
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) {
      float a0, a1, a2, a3, a4, a5, b0, b1, b2, b3, b4, b5, n0, n1, f0, f1;
      asm volatile(
          "v_fma_f32 %0, %16, %20, -%22\n v_fma_f32 %1, %17, %21, -%23\n v_fma_f32 %2, %18, %24, -%26\n"
          "v_fma_f32 %3, %19, %20, -%22\n v_fma_f32 %4, %16, %21, -%23\n v_fma_f32 %5, %17, %24, -%26\n"
          "v_fma_f32 %6, %18, %20, -%22\n v_fma_f32 %7, %19, %21, -%23\n v_fma_f32 %8, %16, %24, -%26\n"
          "v_fma_f32 %9, %17, %20, -%22\n v_fma_f32 %10, %18, %21, -%23\n v_fma_f32 %11, %19, %24, -%26\n"
          "v_max3_f32 %12, %0, %1, %2\n v_min3_f32 %13, %3, %4, %5\n v_max3_f32 %14, %6, %7, %8\n v_min3_f32 %15, %9, %10, %11\n"
          : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(n0),
            "=&v"(f0), "=&v"(n1), "=&v"(f1)
          : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "v"(m01.x), "v"(m01.y), "v"(c01.x), "v"(c01.y), "v"(m23.x), "v"(m23.y), "v"(c23.x), "v"(c23.y));
      acc0 += n0 + f0;
      acc1 += n1 + f1;
    } else {
      f2 r0, r1, r2, r3, r4, r5;
      float n0, n1, f0, f1;
      // src0 = SGPR pair (plane of child 0, plane of child 1); src1/src2 = one ray constant broadcast to both halves
      asm volatile(
          "v_pk_fma_f32 %0, %6, %8, %10 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
          "v_pk_fma_f32 %1, %7, %8, %10 op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
          "v_pk_fma_f32 %2, %6, %9, %11 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
          "v_pk_fma_f32 %3, %7, %9, %11 op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
          "v_pk_fma_f32 %4, %6, %8, %11 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
          "v_pk_fma_f32 %5, %7, %9, %10 op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5)
          : "s"(sp0), "s"(sp1), "v"(m01), "v"(m23), "v"(c01), "v"(c23));
      asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(n0) : "v"(r0.x), "v"(r1.x), "v"(r2.x));
      asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(r3.x), "v"(r4.x), "v"(r5.x));
      asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(n1) : "v"(r0.y), "v"(r1.y), "v"(r2.y));
      asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(r3.y), "v"(r4.y), "v"(r5.y));
      acc0 += n0 + f0;
      acc1 += n1 + f1;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1;
}

template <int KIND>
void run(const char *name, float *d) {
  int blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.9999f, 1.0002f, 0.9998f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.9999f, 1.0002f, 0.9998f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  double steps = (double)blocks * 4 * ITER;
  printf("%-64s %8.3f ms  %6.1f cycles per step per SIMD (nominal 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (steps / 1024.0));
}

int main() {
  float *d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("A: 12 v_fma_f32 + 4 min3/max3 (+4 adds)", d);
  run<1>("B: 6 v_pk_fma_f32 (SGPR pair, op_sel broadcast) + 4 (+4 adds)", d);
  return 0;
}
