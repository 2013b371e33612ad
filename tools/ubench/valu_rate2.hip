// valu_rate.hip - issue cost of the integer VALU / DS / scalar instructions the ranking and alignment kernels are made of, on
// gfx950: each kernel runs one instruction pattern ITER x 64 times per wave (8 independent accumulators, no memory), 4 waves per
// SIMD on every CU.  Prints cycles per wave-instruction per SIMD (s_memtime ticks = shader cycles) - the integer-VALU roofline
// DESIGN.md prices k_rank and k_align against.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>

#define ITER 8000

#define BODY8(INS)  \
  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define REP8(X) X X X X X X X X

template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *cyc, unsigned seed)
{
  unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  unsigned b = seed | 1, c = seed * 7 + 3;
  unsigned long long vcc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; ++i) {
    if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 1) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 2) { REP8(asm volatile("v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n v_lshlrev_b32 %4, %8, %4\n v_lshlrev_b32 %5, %8, %5\n v_lshlrev_b32 %6, %8, %6\n v_lshlrev_b32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 3) { REP8(asm volatile("v_bfe_u32 %0, %0, %8, 4\n v_bfe_u32 %1, %1, %8, 4\n v_bfe_u32 %2, %2, %8, 4\n v_bfe_u32 %3, %3, %8, 4\n v_bfe_u32 %4, %4, %8, 4\n v_bfe_u32 %5, %5, %8, 4\n v_bfe_u32 %6, %6, %8, 4\n v_bfe_u32 %7, %7, %8, 4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 4) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 5) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 6) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 7) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 8) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 9) { REP8(asm volatile("v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 10) { REP8(asm volatile("v_alignbit_b32 %0, %0, %8, %9\n v_alignbit_b32 %1, %1, %8, %9\n v_alignbit_b32 %2, %2, %8, %9\n v_alignbit_b32 %3, %3, %8, %9\n v_alignbit_b32 %4, %4, %8, %9\n v_alignbit_b32 %5, %5, %8, %9\n v_alignbit_b32 %6, %6, %8, %9\n v_alignbit_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 12) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 13) { REP8(asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 14) { REP8(asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 15) { REP8(asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 16) { REP8(asm volatile("s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, %1" : "+s"(b) : "s"(c) : "scc");) }
    if (OP == 17) { REP8(asm volatile("v_add_u32 %0, %8, %0\n s_add_u32 %8, %8, %9\n v_add_u32 %1, %8, %1\n s_add_u32 %8, %8, %9\n v_add_u32 %2, %8, %2\n s_add_u32 %8, %8, %9\n v_add_u32 %3, %8, %3\n s_add_u32 %8, %8, %9\n v_add_u32 %4, %8, %4\n s_add_u32 %8, %8, %9\n v_add_u32 %5, %8, %5\n s_add_u32 %8, %8, %9\n v_add_u32 %6, %8, %6\n s_add_u32 %8, %8, %9\n v_add_u32 %7, %8, %7\n s_add_u32 %8, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(b) : "s"(c) : "scc");) }
    if (OP == 18) { REP8(asm volatile("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %8, %1, 3\n v_readlane_b32 %8, %2, 3\n v_readlane_b32 %8, %3, 3\n v_readlane_b32 %8, %4, 3\n v_readlane_b32 %8, %5, 3\n v_readlane_b32 %8, %6, 3\n v_readlane_b32 %8, %7, 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(b));) }
    if (OP == 19) { REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %0, %1" : : "v"((unsigned long long)a0 << 32 | a1), "v"((unsigned long long)a2 << 32 | a3) : "vcc");) }
    if (OP == 20) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(vcc));) }
    if (OP == 21) { REP8(asm volatile("v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 22) { REP8(asm volatile("v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 23) { REP8(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 24) { REP8(asm volatile("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 25) { REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 26) { REP8(asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 27) { REP8(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 28) { REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 29) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 30) { REP8(asm volatile("v_cmp_lt_u32_e64 %8, %0, %9\n v_cmp_lt_u32_e64 %8, %1, %9\n v_cmp_lt_u32_e64 %8, %2, %9\n v_cmp_lt_u32_e64 %8, %3, %9\n v_cmp_lt_u32_e64 %8, %4, %9\n v_cmp_lt_u32_e64 %8, %5, %9\n v_cmp_lt_u32_e64 %8, %6, %9\n v_cmp_lt_u32_e64 %8, %7, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) , "+s"(vcc) : "v"(b));) }
    if (OP == 31) { REP8(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n v_add_u32 %4, 0x12345, %4\n v_add_u32 %5, 0x12345, %5\n v_add_u32 %6, 0x12345, %6\n v_add_u32 %7, 0x12345, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 32) { REP8(asm volatile("v_and_b32 %0, %9, %0\n v_and_b32 %1, %9, %1\n v_and_b32 %2, %9, %2\n v_and_b32 %3, %9, %3\n v_and_b32 %4, %9, %4\n v_and_b32 %5, %9, %5\n v_and_b32 %6, %9, %6\n v_and_b32 %7, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(c));) }
    if (OP == 33) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_add_co_u32 %1, vcc, %1, %8\n v_add_co_u32 %2, vcc, %2, %8\n v_add_co_u32 %3, vcc, %3, %8\n v_add_co_u32 %4, vcc, %4, %8\n v_add_co_u32 %5, vcc, %5, %8\n v_add_co_u32 %6, vcc, %6, %8\n v_add_co_u32 %7, vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 34) { REP8(asm volatile("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 35) { REP8(asm volatile("v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3\n v_lshlrev_b32 %4, 2, %4\n v_lshlrev_b32 %5, 2, %5\n v_lshlrev_b32 %6, 2, %6\n v_lshlrev_b32 %7, 2, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 36) { REP8(asm volatile("v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3\n v_ashrrev_i32 %4, 1, %4\n v_ashrrev_i32 %5, 1, %5\n v_ashrrev_i32 %6, 1, %6\n v_ashrrev_i32 %7, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 37) { REP8(asm volatile("v_add_u32 %0, %8, %8\n v_add_u32 %1, %8, %8\n v_add_u32 %2, %8, %8\n v_add_u32 %3, %8, %8\n v_add_u32 %4, %8, %8\n v_add_u32 %5, %8, %8\n v_add_u32 %6, %8, %8\n v_add_u32 %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 40) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 41) { REP8(asm volatile("v_cmp_lt_u32_e32 vcc, %8, %0\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cmp_lt_u32_e32 vcc, %8, %1\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cmp_lt_u32_e32 vcc, %8, %2\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cmp_lt_u32_e32 vcc, %8, %3\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cmp_lt_u32_e32 vcc, %8, %4\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cmp_lt_u32_e32 vcc, %8, %5\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_lt_u32_e32 vcc, %8, %6\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cmp_lt_u32_e32 vcc, %8, %7\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 42) { REP8(asm volatile("v_cmp_lt_u32_e64 s[10:11], %8, %0\n v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cmp_lt_u32_e64 s[10:11], %8, %1\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cmp_lt_u32_e64 s[10:11], %8, %2\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cmp_lt_u32_e64 s[10:11], %8, %3\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cmp_lt_u32_e64 s[10:11], %8, %4\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cmp_lt_u32_e64 s[10:11], %8, %5\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cmp_lt_u32_e64 s[10:11], %8, %6\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cmp_lt_u32_e64 s[10:11], %8, %7\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s10", "s11");) }
    if (OP == 43) { REP8(asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %8, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %8, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %8, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %8, vcc\n v_addc_co_u32_e32 %4, vcc, %4, %8, vcc\n v_addc_co_u32_e32 %5, vcc, %5, %8, vcc\n v_addc_co_u32_e32 %6, vcc, %6, %8, vcc\n v_addc_co_u32_e32 %7, vcc, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 44) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");) }
    if (OP == 45) { REP8(asm volatile("v_cmp_lt_u32_e64 s[10:11], %8, %0\n v_cmp_lt_u32_e64 s[12:13], %8, %1\n v_cmp_lt_u32_e64 s[14:15], %8, %2\n v_cmp_lt_u32_e64 s[16:17], %8, %3\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[12:13]\n v_cndmask_b32_e64 %6, %6, %8, s[14:15]\n v_cndmask_b32_e64 %7, %7, %8, s[16:17]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");) }
    if (OP == 46) { REP8(asm volatile("v_cmp_lt_u32_e64 s[10:11], %8, %0\n s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc0 1f\n 1:\n v_cmp_lt_u32_e64 s[10:11], %8, %1\n s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc0 2f\n 2:\n v_cmp_lt_u32_e64 s[10:11], %8, %2\n s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc0 3f\n 3:\n v_cmp_lt_u32_e64 s[10:11], %8, %3\n s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc0 4f\n 4:\n v_cmp_lt_u32_e64 s[10:11], %8, %4\n s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc0 5f\n 5:\n v_cmp_lt_u32_e64 s[10:11], %8, %5\n s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc0 6f\n 6:\n v_cmp_lt_u32_e64 s[10:11], %8, %6\n s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc0 7f\n 7:\n v_cmp_lt_u32_e64 s[10:11], %8, %7\n s_cmp_lg_u64 s[10:11], 0\n s_cbranch_scc0 8f\n 8:" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s10", "s11", "scc");) }
    if (OP == 47) { REP8(asm volatile("v_readlane_b32 s10, %0, 3\n s_add_u32 %8, %8, s10\n v_readlane_b32 s10, %1, 3\n s_add_u32 %8, %8, s10\n v_readlane_b32 s10, %2, 3\n s_add_u32 %8, %8, s10\n v_readlane_b32 s10, %3, 3\n s_add_u32 %8, %8, s10\n v_readlane_b32 s10, %4, 3\n s_add_u32 %8, %8, s10\n v_readlane_b32 s10, %5, 3\n s_add_u32 %8, %8, s10\n v_readlane_b32 s10, %6, 3\n s_add_u32 %8, %8, s10\n v_readlane_b32 s10, %7, 3\n s_add_u32 %8, %8, s10" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(b) : : "s10", "scc");) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> double run(const char *name, int wgs_per_cu, unsigned *d_out, unsigned long long *d_cyc, int ncu)
{
  const int grid = ncu * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, 2u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(grid);
  hipMemcpy(c.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto x : c) avg += (double)x; avg /= grid;
  const double insts_per_wave = (double)ITER * 64;
  const int waves_per_simd = wgs_per_cu;                       // 4 waves per workgroup, one per SIMD
  // cycles per wave-instruction per SIMD from the wave's own clock: a wave's loop time / (instructions of ALL waves sharing its SIMD)
  printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_cycles\": %.0f, \"cycles_per_inst_per_simd\": %.2f, \"Ginst_per_s_chip\": %.1f}\n",
         name, waves_per_simd, ms, avg, avg / (insts_per_wave * waves_per_simd), (double)grid * 4 * insts_per_wave / (ms * 1e-3) / 1e9);
  return ms;
}

int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  unsigned *d_out; unsigned long long *d_cyc;
  hipMalloc(&d_out, (size_t)ncu * 8 * 256 * 4); hipMalloc(&d_cyc, (size_t)ncu * 8 * 8);
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", p.gcnArchName, ncu, p.clockRate);
  for (int w : {1, 4}) {
    run<0>("v_add_u32", w, d_out, d_cyc, ncu);
    run<4>("v_cndmask_b32 (e32, vcc never written)", w, d_out, d_cyc, ncu);
    run<20>("v_cndmask_b32_e64 sgpr mask", w, d_out, d_cyc, ncu);
    run<40>("v_cndmask_b32_e64 explicit vcc operand", w, d_out, d_cyc, ncu);
    run<41>("pair: v_cmp_lt_u32_e32 vcc + v_cndmask_e32 vcc (2 inst)", w, d_out, d_cyc, ncu);
    run<42>("pair: v_cmp_lt_u32_e64 s[10:11] + v_cndmask_e64 s[10:11] (2 inst)", w, d_out, d_cyc, ncu);
    run<43>("v_addc_co_u32_e32 (reads+writes vcc)", w, d_out, d_cyc, ncu);
    run<44>("v_cndmask_b32_e32 vcc after s_mov vcc in loop", w, d_out, d_cyc, ncu);
    run<45>("4x v_cmp_e64 to distinct sgpr pairs then 4x v_cndmask_e64 (8 inst)", w, d_out, d_cyc, ncu);
    run<46>("ballot pattern: v_cmp_e64 sgpr + s_cmp_lg_u64 + s_cbranch (not taken) (3 inst)", w, d_out, d_cyc, ncu);
    run<47>("v_readlane_b32 then s_add using it (2 inst)", w, d_out, d_cyc, ncu);
  }
  return 0;
}
