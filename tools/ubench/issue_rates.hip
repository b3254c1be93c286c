// Issue-rate microbenchmark for gfx950 (MI355X): what ONE SIMD sustains per instruction kind as a
// function of the number of resident waves.  Every body is an inline-asm block of exactly NI
// instructions of one kind (so the count is the disassembly's count by construction; `make check`
// greps the .s), repeated ITER times by every wave.  Registers are chosen so that consecutive
// instructions are independent (16 destinations in rotation).  Occupancy is pinned with dynamic LDS:
// 64-thread workgroups, 4*W of them per CU, so every SIMD holds exactly W waves.
//
// Output: cycles per wave-instruction per SIMD, using the shader clock measured in the same kernel
// (s_memtime ticks over s_memrealtime's 100 MHz), not a nominal figure.
//
// build: hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ITER 2048

#define R16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)

// destinations v[32..47], sources v[0..31] (never written in the loop): fully independent stream
#define D(i) D##i
#define D0 "v32"
#define D1 "v33"
#define D2 "v34"
#define D3 "v35"
#define D4 "v36"
#define D5 "v37"
#define D6 "v38"
#define D7 "v39"
#define D8 "v40"
#define D9 "v41"
#define D10 "v42"
#define D11 "v43"
#define D12 "v44"
#define D13 "v45"
#define D14 "v46"
#define D15 "v47"
#define A(i) A##i
#define A0 "v0"
#define A1 "v1"
#define A2 "v2"
#define A3 "v3"
#define A4 "v4"
#define A5 "v5"
#define A6 "v6"
#define A7 "v7"
#define A8 "v8"
#define A9 "v9"
#define A10 "v10"
#define A11 "v11"
#define A12 "v12"
#define A13 "v13"
#define A14 "v14"
#define A15 "v15"
#define B(i) B##i
#define B0 "v16"
#define B1 "v17"
#define B2 "v18"
#define B3 "v19"
#define B4 "v20"
#define B5 "v21"
#define B6 "v22"
#define B7 "v23"
#define B8 "v24"
#define B9 "v25"
#define B10 "v26"
#define B11 "v27"
#define B12 "v28"
#define B13 "v29"
#define B14 "v30"
#define B15 "v31"
// even-aligned 64-bit pairs for packed ops: destinations v[48+2i:49+2i]
#define P(i) P##i
#define P0 "v[48:49]"
#define P1 "v[50:51]"
#define P2 "v[52:53]"
#define P3 "v[54:55]"
#define P4 "v[56:57]"
#define P5 "v[58:59]"
#define P6 "v[60:61]"
#define P7 "v[62:63]"
#define P8 "v[48:49]"
#define P9 "v[50:51]"
#define P10 "v[52:53]"
#define P11 "v[54:55]"
#define P12 "v[56:57]"
#define P13 "v[58:59]"
#define P14 "v[60:61]"
#define P15 "v[62:63]"
#define Q(i) Q##i
#define Q0 "v[0:1]"
#define Q1 "v[2:3]"
#define Q2 "v[4:5]"
#define Q3 "v[6:7]"
#define Q4 "v[8:9]"
#define Q5 "v[10:11]"
#define Q6 "v[12:13]"
#define Q7 "v[14:15]"
#define Q8 "v[16:17]"
#define Q9 "v[18:19]"
#define Q10 "v[20:21]"
#define Q11 "v[22:23]"
#define Q12 "v[24:25]"
#define Q13 "v[26:27]"
#define Q14 "v[28:29]"
#define Q15 "v[30:31]"
// SGPR destinations s[40+2i:41+2i]
#define S(i) S##i
#define S0 "s[40:41]"
#define S1 "s[42:43]"
#define S2 "s[44:45]"
#define S3 "s[46:47]"
#define S4 "s[48:49]"
#define S5 "s[50:51]"
#define S6 "s[52:53]"
#define S7 "s[54:55]"
#define S8 "s[56:57]"
#define S9 "s[58:59]"
#define S10 "s[60:61]"
#define S11 "s[62:63]"
#define S12 "s[64:65]"
#define S13 "s[66:67]"
#define S14 "s[68:69]"
#define S15 "s[70:71]"
#define T(i) T##i
#define T0 "s40"
#define T1 "s41"
#define T2 "s42"
#define T3 "s43"
#define T4 "s44"
#define T5 "s45"
#define T6 "s46"
#define T7 "s47"
#define T8 "s48"
#define T9 "s49"
#define T10 "s50"
#define T11 "s51"
#define T12 "s52"
#define T13 "s53"
#define T14 "s54"
#define T15 "s55"

#define K_FMA_VVV(i) "v_fma_f32 " D(i) ", " A(i) ", " B(i) ", " A(i) "\n"
#define K_FMA_SVV(i) "v_fma_f32 " D(i) ", s36, " B(i) ", " A(i) "\n"
#define K_FMA_SVNV(i) "v_fma_f32 " D(i) ", s36, " B(i) ", -" A(i) "\n"
#define K_FMAC(i) "v_fmac_f32 " D(i) ", " A(i) ", " B(i) "\n"
#define K_MUL_VV(i) "v_mul_f32 " D(i) ", " A(i) ", " B(i) "\n"
#define K_MUL_SV(i) "v_mul_f32 " D(i) ", s36, " B(i) "\n"
#define K_ADD_VV(i) "v_add_f32 " D(i) ", " A(i) ", " B(i) "\n"
#define K_MAX3(i) "v_max3_f32 " D(i) ", " A(i) ", " B(i) ", " A(i) "\n"
#define K_MAX3C(i) "v_max3_f32 " D(i) ", " A(i) ", " B(i) ", " A(i) " clamp\n"
#define K_MIN2(i) "v_min_f32 " D(i) ", " A(i) ", " B(i) "\n"
#define K_CMP_S(i) "v_cmp_le_f32 " S(i) ", " A(i) ", " B(i) "\n"
#define K_CMP_VCC(i) "v_cmp_le_f32 vcc, " A(i) ", " B(i) "\n"
#define K_CNDMASK_S(i) "v_cndmask_b32 " D(i) ", " A(i) ", " B(i) ", s[38:39]\n"
#define K_CNDMASK_VCC(i) "v_cndmask_b32 " D(i) ", " A(i) ", " B(i) ", vcc\n"
#define K_PKFMA(i) "v_pk_fma_f32 " P(i) ", " Q(i) ", " Q(i) ", " Q(i) "\n"
#define K_PKMUL(i) "v_pk_mul_f32 " P(i) ", " Q(i) ", " Q(i) "\n"
#define K_MOV(i) "v_mov_b32 " D(i) ", " A(i) "\n"
#define K_RCP(i) "v_rcp_f32 " D(i) ", " A(i) "\n"
#define K_READLANE(i) "v_readlane_b32 " T(i) ", " A(i) ", 3\n"
#define K_WRITELANE(i) "v_writelane_b32 " D(i) ", s36, 3\n"
#define K_SAND64(i) "s_and_b64 " S(i) ", s[36:37], s[38:39]\n"
#define K_SADD(i) "s_add_u32 " T(i) ", s36, s37\n"
#define K_SCMP(i) "s_cmp_lg_u64 s[36:37], 0\n"
#define K_BCNT(i) "s_bcnt1_i32_b64 " T(i) ", s[36:37]\n"
#define K_SNOP(i) "s_nop 0\n"
#define K_DPP_MIN(i) "v_min_u32_dpp " D(i) ", " A(i) ", " B(i) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define K_DPP_MIRROR(i) "v_min_u32_dpp " D(i) ", " A(i) ", " B(i) " row_mirror row_mask:0xf bank_mask:0xf\n"
#define K_DPP_MOV(i) "v_mov_b32_dpp " D(i) ", " A(i) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define K_MIN_U32(i) "v_min_u32 " D(i) ", " A(i) ", " B(i) "\n"
#define K_SDWA_CVT(i) "v_cvt_f32_u32_sdwa " D(i) ", " A(i) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define K_CVT_U32(i) "v_cvt_f32_u32 " D(i) ", " A(i) "\n"
#define K_CVT_UBYTE(i) "v_cvt_f32_ubyte1 " D(i) ", " A(i) "\n"
#define K_AND_OR(i) "v_and_or_b32 " D(i) ", " A(i) ", " B(i) ", " A(i) "\n"
#define K_MBCNT(i) "v_mbcnt_lo_u32_b32 " D(i) ", s36, " A(i) "\n"
#define K_LSHL_ADD(i) "v_lshl_add_u32 " D(i) ", " A(i) ", 3, " B(i) "\n"
#define K_SNOP1(i) "s_nop 1\n"
// the DPP reduction step as used in the kernels: s_nop 1 + one dependent DPP op
#define K_DPP_CHAIN(i) "s_nop 1\n v_min_u32_dpp v32, v32, v32 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
// readlane feeding a scalar op (the VALU -> SGPR -> SALU path of the cross-row combine)
#define K_RL_SMIN(i) "v_readlane_b32 " T(i) ", " A(i) ", 16\n s_min_u32 s40, s40, " T(i) "\n"
// v_cmp -> SGPR pair -> s_and -> s_cmp -> (not taken) branch: the uniform decision chain of the walks
#define K_CMP_BR(i) "v_cmp_le_f32 s[40:41], " A(i) ", " B(i) "\n s_and_b64 s[42:43], s[40:41], s[38:39]\n s_cmp_lg_u64 s[42:43], 0\n s_cbranch_scc0 1f\n"
// VALU/SALU interleave
#define K_MIX11(i) "v_fma_f32 " D(i) ", s36, " B(i) ", " A(i) "\n" "s_and_b64 " S(i) ", s[36:37], s[38:39]\n"
#define K_MIX21(i) "v_fma_f32 " D(i) ", s36, " B(i) ", " A(i) "\n" "v_mul_f32 " D(i) ", s36, " B(i) "\n" "s_and_b64 " S(i) ", s[36:37], s[38:39]\n"
#define K_MIX12(i) "v_fma_f32 " D(i) ", s36, " B(i) ", " A(i) "\n" "s_and_b64 " S(i) ", s[36:37], s[38:39]\n" "s_add_u32 " T(i) ", s36, s37\n"
// dependent chain: each instruction consumes the previous result
#define K_DEP_FMA(i) "v_fma_f32 v32, v32, v16, v0\n"
#define K_DEP_MUL(i) "v_mul_f32 v32, v32, v16\n"
// a never-taken and an always-taken scalar branch
#define K_BR_NT(i) "s_cbranch_scc1 1f\n"
#define K_CMPBR(i) "s_cmp_lg_u32 s36, 0\n s_cbranch_scc0 1f\n"

#define CLOB "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", \
             "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47",   \
             "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",   \
             "s68", "s69", "s70", "s71", "vcc", "scc", "s36", "s37", "s38", "s39", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10",       \
             "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"

struct Stamp { unsigned long long t0, t1, r0, r1; };

#define KERNEL(NAME, BODY)                                                                                                        \
  __global__ void __launch_bounds__(64) NAME(float *out, Stamp *st, int iters) {                                                  \
    extern __shared__ int dyn[];                                                                                                  \
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                                  \
    asm volatile("s_mov_b32 s36, 0x3f800100\n s_mov_b32 s37, 7\n s_mov_b64 s[38:39], 0x5555\n"                                    \
                 "v_mov_b32 v0, 1.0\n v_mov_b32 v1, 1.0\n v_mov_b32 v2, 1.0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 1.0\n"             \
                 "v_mov_b32 v5, 1.0\n v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n"             \
                 "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n"        \
                 "v_mov_b32 v15, 1.0\n v_mov_b32 v16, 0.5\n v_mov_b32 v17, 0.5\n v_mov_b32 v18, 0.5\n v_mov_b32 v19, 0.5\n"        \
                 "v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n v_mov_b32 v24, 0.5\n"        \
                 "v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.5\n v_mov_b32 v27, 0.5\n v_mov_b32 v28, 0.5\n v_mov_b32 v29, 0.5\n"        \
                 "v_mov_b32 v30, 0.5\n v_mov_b32 v31, 0.5\n v_mov_b32 v32, 0.5\n" ::: CLOB);                                      \
    for (int it = 0; it < iters; ++it) { asm volatile(BODY "1:\n" ::: CLOB); }                                                    \
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                                  \
    float v;                                                                                                                      \
    asm volatile("v_mov_b32 %0, v32" : "=v"(v)::CLOB);                                                                            \
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->t0 = t0; st->t1 = t1; st->r0 = r0; st->r1 = r1; }                              \
    if (v == 12345.f) out[0] = v + dyn[0];                                                                                        \
  }

#define X4(M) R16(M) R16(M) R16(M) R16(M)

KERNEL(k_fma_vvv, X4(K_FMA_VVV))
KERNEL(k_fma_svv, X4(K_FMA_SVV))
KERNEL(k_fma_svnv, X4(K_FMA_SVNV))
KERNEL(k_fmac, X4(K_FMAC))
KERNEL(k_mul_vv, X4(K_MUL_VV))
KERNEL(k_mul_sv, X4(K_MUL_SV))
KERNEL(k_add_vv, X4(K_ADD_VV))
KERNEL(k_max3, X4(K_MAX3))
KERNEL(k_max3c, X4(K_MAX3C))
KERNEL(k_min2, X4(K_MIN2))
KERNEL(k_cmp_s, X4(K_CMP_S))
KERNEL(k_cmp_vcc, X4(K_CMP_VCC))
KERNEL(k_cnd_s, X4(K_CNDMASK_S))
KERNEL(k_cnd_vcc, X4(K_CNDMASK_VCC))
KERNEL(k_pkfma, X4(K_PKFMA))
KERNEL(k_pkmul, X4(K_PKMUL))
KERNEL(k_mov, X4(K_MOV))
KERNEL(k_rcp, X4(K_RCP))
KERNEL(k_readlane, X4(K_READLANE))
KERNEL(k_writelane, X4(K_WRITELANE))
KERNEL(k_sand64, X4(K_SAND64))
KERNEL(k_sadd, X4(K_SADD))
KERNEL(k_scmp, X4(K_SCMP))
KERNEL(k_bcnt, X4(K_BCNT))
KERNEL(k_snop, X4(K_SNOP))
KERNEL(k_mix11, X4(K_MIX11))
KERNEL(k_mix21, X4(K_MIX21))
KERNEL(k_mix12, X4(K_MIX12))
KERNEL(k_dep_fma, X4(K_DEP_FMA))
KERNEL(k_dep_mul, X4(K_DEP_MUL))
KERNEL(k_br_nt, "s_cmp_lg_u32 0, 0\n" X4(K_BR_NT))
KERNEL(k_dpp_min, X4(K_DPP_MIN))
KERNEL(k_dpp_mirror, X4(K_DPP_MIRROR))
KERNEL(k_dpp_mov, X4(K_DPP_MOV))
KERNEL(k_min_u32, X4(K_MIN_U32))
KERNEL(k_sdwa_cvt, X4(K_SDWA_CVT))
KERNEL(k_cvt_u32, X4(K_CVT_U32))
KERNEL(k_cvt_ubyte, X4(K_CVT_UBYTE))
KERNEL(k_and_or, X4(K_AND_OR))
KERNEL(k_mbcnt, X4(K_MBCNT))
KERNEL(k_lshl_add, X4(K_LSHL_ADD))
KERNEL(k_snop1, X4(K_SNOP1))
KERNEL(k_dpp_chain, X4(K_DPP_CHAIN))
KERNEL(k_rl_smin, X4(K_RL_SMIN))
KERNEL(k_cmp_br, X4(K_CMP_BR))

// scalar loads from a 4 KB table (scalar-cache hits): 16 x s_load_dwordx16 per iteration, drained once per iteration
__global__ void __launch_bounds__(64) k_sload16(float *out, Stamp *st, int iters, const int *tab) {
  extern __shared__ int dyn[];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_load_dwordx16 s[40:55], %0, 0x0\n s_load_dwordx16 s[56:71], %0, 0x40\n s_load_dwordx16 s[40:55], %0, 0x80\n"
                 "s_load_dwordx16 s[56:71], %0, 0xc0\n s_load_dwordx16 s[40:55], %0, 0x100\n s_load_dwordx16 s[56:71], %0, 0x140\n"
                 "s_load_dwordx16 s[40:55], %0, 0x180\n s_load_dwordx16 s[56:71], %0, 0x1c0\n s_load_dwordx16 s[40:55], %0, 0x200\n"
                 "s_load_dwordx16 s[56:71], %0, 0x240\n s_load_dwordx16 s[40:55], %0, 0x280\n s_load_dwordx16 s[56:71], %0, 0x2c0\n"
                 "s_load_dwordx16 s[40:55], %0, 0x300\n s_load_dwordx16 s[56:71], %0, 0x340\n s_load_dwordx16 s[40:55], %0, 0x380\n"
                 "s_load_dwordx16 s[56:71], %0, 0x3c0\n s_waitcnt lgkmcnt(0)\n" ::"s"(tab) : CLOB);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) { st->t0 = t0; st->t1 = t1; st->r0 = r0; st->r1 = r1; }
  if (iters == 12345) out[0] = dyn[0];
}
// dependent scalar loads (offset chase inside the scalar cache): latency of one s_load_dword
__global__ void __launch_bounds__(64) k_sload_dep(float *out, Stamp *st, int iters, const int *tab) {
  extern __shared__ int dyn[];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_mov_b32 s40, 0" ::: CLOB);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n"
                 "s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n"
                 "s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n"
                 "s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n"
                 "s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n"
                 "s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n"
                 "s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n"
                 "s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n s_load_dword s40, %0, s40\n s_waitcnt lgkmcnt(0)\n" ::"s"(tab) : CLOB);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) { st->t0 = t0; st->t1 = t1; st->r0 = r0; st->r1 = r1; }
  if (iters == 12345) out[0] = dyn[0];
}

typedef void (*kfn)(float *, Stamp *, int);

static float *d_out;
static Stamp *d_st;
static int *d_tab;

template <typename F>
static void time_one(const char *name, int n_per_iter, int waves_per_simd, F launch) {
  // 4*W workgroups of 64 threads per CU, pinned by LDS: each gets 160 KiB / (4 W), minus a little
  const int wg_per_cu = 4 * waves_per_simd;
  size_t lds = (size_t)(160 * 1024) / wg_per_cu;
  lds = lds > 64 ? lds - 64 : lds;
  if (waves_per_simd >= 8) lds = 0; // 32 waves per CU is the cap anyway
  const int blocks = 256 * wg_per_cu;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch(blocks, lds, 64); // warm-up
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch(blocks, lds, ITER);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  Stamp s;
  hipMemcpy(&s, d_st, sizeof s, hipMemcpyDeviceToHost);
  const double clk = (double)(s.t1 - s.t0) / ((double)(s.r1 - s.r0) * 10e-9); // Hz
  const double instr_per_simd = (double)waves_per_simd * ITER * n_per_iter;
  const double cyc = ms * 1e-3 * clk / instr_per_simd;
  const double cyc_in = (double)(s.t1 - s.t0) / instr_per_simd; // from the in-kernel stamps of one wave (all waves run concurrently)
  printf("%-40s W=%d  %8.3f ms  clk %.3f GHz  %6.2f cyc/instr/SIMD (events)  %6.2f (in-kernel)\n", name, waves_per_simd, ms, clk * 1e-9, cyc, cyc_in);
  hipEventDestroy(a);
  hipEventDestroy(b);
}

static const char *g_filter = nullptr;
#define RUN(K, NAME, N)                                                                                                             \
  if (!g_filter || strstr(NAME, g_filter))                                                                                           \
  for (int w : {1, 2, 4, 7, 8})                                                                                                      \
    time_one(NAME, N, w, [&](int blocks, size_t lds, int iters) { hipLaunchKernelGGL(K, dim3(blocks), dim3(64), lds, 0, d_out, d_st, iters); });

int main(int argc, char **argv) {
  if (argc > 1) g_filter = argv[1];
  hipMalloc(&d_out, 1024);
  hipMalloc(&d_st, sizeof(Stamp));
  int h_tab[64 * 16];
  for (int k = 0; k < 64; ++k)
    for (int j = 0; j < 16; ++j) h_tab[k * 16 + j] = ((k + 1) & 63) * 64; // byte offset of the next line
  hipMalloc(&d_tab, sizeof h_tab);
  hipMemcpy(d_tab, h_tab, sizeof h_tab, hipMemcpyHostToDevice);
  RUN(k_fma_vvv, "v_fma_f32 v,v,v,v", 64)
  RUN(k_fma_svv, "v_fma_f32 v,s,v,v", 64)
  RUN(k_fma_svnv, "v_fma_f32 v,s,v,-v", 64)
  RUN(k_fmac, "v_fmac_f32 v,v,v (VOP2)", 64)
  RUN(k_mul_vv, "v_mul_f32 v,v,v (VOP2)", 64)
  RUN(k_mul_sv, "v_mul_f32 v,s,v (VOP2)", 64)
  RUN(k_add_vv, "v_add_f32 v,v,v (VOP2)", 64)
  RUN(k_max3, "v_max3_f32", 64)
  RUN(k_max3c, "v_max3_f32 clamp", 64)
  RUN(k_min2, "v_min_f32 (VOP2)", 64)
  RUN(k_cmp_s, "v_cmp_le_f32 -> SGPR pair (VOP3)", 64)
  RUN(k_cmp_vcc, "v_cmp_le_f32 -> vcc (VOPC)", 64)
  RUN(k_cnd_s, "v_cndmask_b32 SGPR mask (VOP3)", 64)
  RUN(k_cnd_vcc, "v_cndmask_b32 vcc (VOP2)", 64)
  RUN(k_pkfma, "v_pk_fma_f32", 64)
  RUN(k_pkmul, "v_pk_mul_f32", 64)
  RUN(k_mov, "v_mov_b32", 64)
  RUN(k_rcp, "v_rcp_f32", 64)
  RUN(k_readlane, "v_readlane_b32", 64)
  RUN(k_writelane, "v_writelane_b32", 64)
  RUN(k_sand64, "s_and_b64", 64)
  RUN(k_sadd, "s_add_u32", 64)
  RUN(k_scmp, "s_cmp_lg_u64", 64)
  RUN(k_bcnt, "s_bcnt1_i32_b64", 64)
  RUN(k_snop, "s_nop 0", 64)
  RUN(k_mix11, "1 v_fma(s) + 1 s_and_b64", 128)
  RUN(k_mix21, "2 VALU + 1 SALU", 192)
  RUN(k_mix12, "1 VALU + 2 SALU", 192)
  RUN(k_dep_fma, "v_fma_f32 dependent chain", 64)
  RUN(k_dep_mul, "v_mul_f32 dependent chain", 64)
  RUN(k_br_nt, "s_cbranch_scc1 not taken", 65)
  RUN(k_dpp_min, "v_min_u32_dpp quad_perm (indep.)", 64)
  RUN(k_dpp_mirror, "v_min_u32_dpp row_mirror (indep.)", 64)
  RUN(k_dpp_mov, "v_mov_b32_dpp quad_perm", 64)
  RUN(k_min_u32, "v_min_u32 (VOP2)", 64)
  RUN(k_sdwa_cvt, "v_cvt_f32_u32_sdwa WORD_1", 64)
  RUN(k_cvt_u32, "v_cvt_f32_u32", 64)
  RUN(k_cvt_ubyte, "v_cvt_f32_ubyte1", 64)
  RUN(k_and_or, "v_and_or_b32", 64)
  RUN(k_mbcnt, "v_mbcnt_lo_u32_b32 (SGPR mask)", 64)
  RUN(k_lshl_add, "v_lshl_add_u32", 64)
  RUN(k_snop1, "s_nop 1", 64)
  RUN(k_dpp_chain, "s_nop 1 + dependent v_min_u32_dpp (pair)", 64)
  RUN(k_rl_smin, "v_readlane -> s_min_u32 (pair)", 64)
  RUN(k_cmp_br, "v_cmp->s_and->s_cmp->branch (4 instr)", 64)
  for (int w : {1, 2, 4, 7, 8})
    time_one("s_load_dwordx16 (scalar-cache hits)", 16, w,
             [&](int blocks, size_t lds, int iters) { hipLaunchKernelGGL(k_sload16, dim3(blocks), dim3(64), lds, 0, d_out, d_st, iters, d_tab); });
  for (int w : {1, 2, 4, 7, 8})
    time_one("s_load_dword dependent chain", 16, w,
             [&](int blocks, size_t lds, int iters) { hipLaunchKernelGGL(k_sload_dep, dim3(blocks), dim3(64), lds, 0, d_out, d_st, iters, d_tab); });
  return 0;
}
