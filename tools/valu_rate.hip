// Micro-benchmark: issue rate of the integer VALU ops the fill kernel is made of
// (per SIMD, 8 waves resident).  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k(int *out, int iters, int seed) {
  int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < REP; r++) {
      if (OP == 0) { asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 1) { asm volatile("v_min_i32 %0, %0, %1\n v_min_i32 %2, %2, %3\n v_min_i32 %4, %4, %5\n v_min_i32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 2) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc"); }
      if (OP == 3) { asm volatile("v_min3_i32 %0, %0, %1, %2\n v_min3_i32 %2, %2, %3, %4\n v_min3_i32 %4, %4, %5, %6\n v_min3_i32 %6, %6, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 4) { asm volatile("v_min_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_i32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_i32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_i32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 5) { asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %2, %3\n v_cmp_eq_u32 vcc, %4, %5\n v_cmp_eq_u32 vcc, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc"); }
      if (OP == 6) { asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %2, %2, %3, %2\n v_fma_f32 %4, %4, %5, %4\n v_fma_f32 %6, %6, %7, %6" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 8) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n v_cndmask_b32_e64 %4, %4, %5, s[10:11]\n v_cndmask_b32_e64 %6, %6, %7, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s10", "s11"); }
      if (OP == 9) { asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc"); }
      if (OP == 10) { asm volatile("v_cmp_lt_i32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_i32 vcc, %4, %5\n s_nop 1\n v_cndmask_b32 %6, %6, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc"); }
      if (OP == 11) { asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 12) { asm volatile("v_bfe_i32 %0, %0, %1, 8\n v_bfe_i32 %2, %2, %3, 8\n v_sub_u32 %4, %4, %5\n v_sub_u32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 13) { asm volatile("v_readlane_b32 s10, %0, 63\n v_readlane_b32 s11, %2, 63\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7):: "s10", "s11"); }
      if (OP == 20) { asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %2, %2, %3\n v_min_f32 %4, %4, %5\n v_min_f32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 21) { asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %2, %2, %3\n v_max_i32 %4, %4, %5\n v_max_i32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 22) { asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %3\n v_and_b32 %4, %4, %5\n v_and_b32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 23) { asm volatile("v_or_b32 %0, %0, %1\n v_or_b32 %2, %2, %3\n v_or_b32 %4, %4, %5\n v_or_b32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 24) { asm volatile("v_lshlrev_b32 %0, %0, %1\n v_lshlrev_b32 %2, %2, %3\n v_lshlrev_b32 %4, %4, %5\n v_lshlrev_b32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 25) { asm volatile("v_min_i16 %0, %0, %1\n v_min_i16 %2, %2, %3\n v_min_i16 %4, %4, %5\n v_min_i16 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 26) { asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %2, %2, %3\n v_min_u32 %4, %4, %5\n v_min_u32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 27) { asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %4, %4, %5\n v_add_f32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 28) { asm volatile("v_sub_u32 %0, %0, %1\n v_sub_u32 %2, %2, %3\n v_sub_u32 %4, %4, %5\n v_sub_u32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 29) { asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_xor_b32 %4, %4, %5\n v_xor_b32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 30) { asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %3\n v_max_f32 %4, %4, %5\n v_max_f32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 31) { asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 32) { asm volatile("v_ashrrev_i32 %0, %0, %1\n v_ashrrev_i32 %2, %2, %3\n v_ashrrev_i32 %4, %4, %5\n v_ashrrev_i32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 33) { asm volatile("v_add_u16 %0, %0, %1\n v_add_u16 %2, %2, %3\n v_add_u16 %4, %4, %5\n v_add_u16 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 34) { asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %4, %4, %5\n v_pk_add_u16 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 35) { asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %2, %2, %3\n v_pk_max_i16 %4, %4, %5\n v_pk_max_i16 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 40) { asm volatile("v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %2, %2, %3, %4\n v_min3_f32 %4, %4, %5, %6\n v_min3_f32 %6, %6, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 41) { asm volatile("v_min_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 42) { asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 43) { asm volatile("v_cmp_eq_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_eq_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc"); }
      if (OP == 44) { asm volatile("v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %2, %2, %3, %4\n v_perm_b32 %4, %4, %5, %6\n v_perm_b32 %6, %6, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 45) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 46) { asm volatile("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %2, %2, %3, %4\n v_lshl_add_u32 %4, %4, 1, %6\n v_lshl_add_u32 %6, %6, 1, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
      if (OP == 7) { asm volatile("v_pk_add_i16 %0, %0, %1\n v_pk_add_i16 %2, %2, %3\n v_pk_min_i16 %4, %4, %5\n v_pk_min_i16 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char *name, int *out) {
  const int blocks = 256 * 8, iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(out, 10, 1); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, iters, 1); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winstr = (double)blocks * 4 * iters * REP * 4;  // wave-instructions
  double per_simd_per_s = winstr / (ms * 1e-3) / 1024;
  printf("%-14s %8.2f ms  %.3f G wave-instr/s/SIMD  -> %.2f cycles/instr at 2.4 GHz\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}
int main() {
  int *out; hipMalloc(&out, 256 * 8 * 256 * 4);
  run<0>("v_add_u32", out); run<1>("v_min_i32", out); run<2>("v_cndmask", out); run<3>("v_min3_i32", out);
  run<4>("v_min_i32_dpp", out); run<5>("v_cmp_eq", out); run<6>("v_fma_f32", out); run<7>("v_pk_i16", out);
  run<8>("cndmask_e64", out); run<9>("cmp,add,add,cnd", out); run<10>("cmp,nop,cnd x2", out); run<11>("mov_dpp", out); run<12>("bfe,bfe,sub,sub", out); run<13>("readlane x2+add x2", out);
  run<20>("v_min_f32", out); run<21>("v_max_i32", out); run<22>("v_and_b32", out); run<23>("v_or_b32", out); run<24>("v_lshlrev_b32", out); run<25>("v_min_i16", out); run<26>("v_min_u32", out); run<27>("v_add_f32", out); run<28>("v_sub_u32", out); run<29>("v_xor_b32", out); run<30>("v_max_f32", out); run<31>("v_mul_u32_u24", out); run<32>("v_ashrrev_i32", out); run<33>("v_add_u16", out); run<34>("v_pk_add_u16", out); run<35>("v_pk_max_i16", out); run<40>("v_min3_f32", out); run<41>("v_min_f32_dpp", out); run<42>("v_add_u32_dpp", out); run<43>("v_cmp_f32", out); run<44>("bfi/perm", out); run<45>("v_mov_b32", out); run<46>("add3/lshl_add", out);
  return 0;
}
