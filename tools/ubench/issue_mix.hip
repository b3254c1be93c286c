// Issue-rate microbenchmark, part 2 (gfx950): scalar-ALU rate, VALU/SALU co-issue, and packed fp32 with
// an SGPR-pair operand — the instruction mix of the packet traversal's node step.
// Each wave runs ITER iterations of a fixed asm block; 8 waves per SIMD (2048 workgroups x 256 threads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 4096

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, float s0, float s1, int n) {
  float a = threadIdx.x * 0.001f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
  float e = a + 4.f, f = a + 5.f, g = a + 6.f, h = a + 7.f;
  int x = n, y = n + 1, z = n + 2, w = n + 3;
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) // 16 independent v_fma_f32 with an SGPR operand
      asm volatile(REP4("v_fma_f32 %0, %8, %0, %1\n v_fma_f32 %2, %8, %2, %3\n v_fma_f32 %4, %8, %4, %5\n v_fma_f32 %6, %8, %6, %7\n")
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s0));
    if (KIND == 1) // 16 v_pk_fma_f32 with an SGPR-pair operand (two planes per instruction)
      asm volatile(REP4("v_pk_fma_f32 %0, %4, %0, %1\n v_pk_fma_f32 %1, %4, %1, %2\n v_pk_fma_f32 %2, %4, %2, %3\n v_pk_fma_f32 %3, %4, %3, %0\n")
                   : "+v"(*(double *)&a), "+v"(*(double *)&c), "+v"(*(double *)&e), "+v"(*(double *)&g) : "s"(*(double *)&s0));
    if (KIND == 2) // 16 v_pk_fma_f32, VGPR operands only
      asm volatile(REP4("v_pk_fma_f32 %0, %1, %0, %1\n v_pk_fma_f32 %1, %2, %1, %2\n v_pk_fma_f32 %2, %3, %2, %3\n v_pk_fma_f32 %3, %0, %3, %0\n")
                   : "+v"(*(double *)&a), "+v"(*(double *)&c), "+v"(*(double *)&e), "+v"(*(double *)&g));
    if (KIND == 3) // 16 SALU
      asm volatile(REP4("s_add_i32 %0, %0, 1\n s_xor_b32 %1, %1, %0\n s_add_i32 %2, %2, 3\n s_and_b32 %3, %3, %2\n") : "+s"(x), "+s"(y), "+s"(z), "+s"(w) : : "scc");
    if (KIND == 4) // 16 VALU + 16 SALU interleaved
      asm volatile(REP4("v_fma_f32 %0, %12, %0, %1\n s_add_i32 %8, %8, 1\n v_fma_f32 %2, %12, %2, %3\n s_xor_b32 %9, %9, %8\n v_fma_f32 %4, %12, %4, %5\n s_add_i32 %10, %10, 3\n"
                        "v_fma_f32 %6, %12, %6, %7\n s_and_b32 %11, %11, %10\n")
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+s"(x), "+s"(y), "+s"(z), "+s"(w) : "s"(s0) : "scc");
    if (KIND == 5) // 8 VALU + 24 SALU interleaved
      asm volatile(REP4("v_fma_f32 %0, %12, %0, %1\n s_add_i32 %8, %8, 1\n s_xor_b32 %9, %9, %8\n s_add_i32 %10, %10, 3\n v_fma_f32 %2, %12, %2, %3\n s_and_b32 %11, %11, %10\n"
                        "s_add_i32 %8, %8, 1\n s_xor_b32 %9, %9, %8\n")
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+s"(x), "+s"(y), "+s"(z), "+s"(w) : "s"(s0) : "scc");
    if (KIND == 6) // 16 v_cmp writing an SGPR pair + 16 s_and_b64 consuming it (mask logic)
      asm volatile(REP16("v_cmp_le_f32 vcc, %0, %1\n s_and_b64 %2, %2, vcc\n") : "+v"(a), "+v"(b), "+s"(*(long long *)&x) : : "vcc", "scc");
    if (KIND == 7) // 16 taken branches (s_cbranch to the next instruction)
      asm volatile(REP16("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n") : "+s"(x) : : "scc");
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (float)(x + y + z + w);
}

template <int KIND>
void run(const char *name, int per_iter_instr, float *d) {
  int blocks = 256 * 8; // 8 waves per SIMD
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.9999f, 3);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.9999f, 3);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  double winstr = (double)blocks * 4 * ITER * per_iter_instr;
  double per_simd_cycles = ms * 1e-3 * 2.4e9;
  printf("%-52s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (nominal 2.4 GHz)\n", name, ms, per_simd_cycles / (winstr / 1024.0));
}

int main() {
  float *d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("16 v_fma_f32 (SGPR operand)", 16, d);
  run<1>("16 v_pk_fma_f32 (SGPR-pair operand)", 16, d);
  run<2>("16 v_pk_fma_f32 (VGPR operands)", 16, d);
  run<3>("16 SALU (s_add/s_xor/s_and)", 16, d);
  run<4>("16 VALU + 16 SALU interleaved (32 instr)", 32, d);
  run<5>("8 VALU + 24 SALU interleaved (32 instr)", 32, d);
  run<6>("16 v_cmp->SGPR pair + 16 s_and_b64 (32 instr)", 32, d);
  run<7>("16 x (s_cmp + taken s_cbranch) (32 instr)", 32, d);
  return 0;
}
