#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE> __global__ __launch_bounds__(1024) void k(unsigned *out, int iters) {
 unsigned a=threadIdx.x,b=a*3+1,c=a^5,d=a+7,e=a+11,f=b+13,g=c+17,h=d+19; double p=a,q=b,r=c,s=d;
 for (int it=0; it<iters; it++) {
  if (MODE==0) { REP8(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==1) { REP8(asm volatile("v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_or_b32 %3, %3, %4\n v_or_b32 %4, %4, %5\n v_or_b32 %5, %5, %6\n v_or_b32 %6, %6, %7\n v_or_b32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==2) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==3) { REP8(asm volatile("v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %4\n v_sub_u32 %4, %4, %5\n v_sub_u32 %5, %5, %6\n v_sub_u32 %6, %6, %7\n v_sub_u32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==4) { REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==5) { REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %4\n v_mul_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_mul_f32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==6) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %2, %2, %3, %3\n v_fma_f32 %3, %3, %4, %4\n v_fma_f32 %4, %4, %5, %5\n v_fma_f32 %5, %5, %6, %6\n v_fma_f32 %6, %6, %7, %7\n v_fma_f32 %7, %7, %0, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==7) { REP8(asm volatile("v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n v_min_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_min_f32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==8) { REP8(asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %4\n v_max_f32 %4, %4, %5\n v_max_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_max_f32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==9) { REP8(asm volatile("v_min_i32 %0, %0, %1\n v_min_i32 %1, %1, %2\n v_min_i32 %2, %2, %3\n v_min_i32 %3, %3, %4\n v_min_i32 %4, %4, %5\n v_min_i32 %5, %5, %6\n v_min_i32 %6, %6, %7\n v_min_i32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==10) { REP8(asm volatile("v_max_u32 %0, %0, %1\n v_max_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_max_u32 %3, %3, %4\n v_max_u32 %4, %4, %5\n v_max_u32 %5, %5, %6\n v_max_u32 %6, %6, %7\n v_max_u32 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==11) { REP8(asm volatile("v_med3_f32 %0, %0, %1, %1\n v_med3_f32 %1, %1, %2, %2\n v_med3_f32 %2, %2, %3, %3\n v_med3_f32 %3, %3, %4, %4\n v_med3_f32 %4, %4, %5, %5\n v_med3_f32 %5, %5, %6, %6\n v_med3_f32 %6, %6, %7, %7\n v_med3_f32 %7, %7, %0, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==12) { REP8(asm volatile("v_min3_f32 %0, %0, %1, %1\n v_min3_f32 %1, %1, %2, %2\n v_min3_f32 %2, %2, %3, %3\n v_min3_f32 %3, %3, %4, %4\n v_min3_f32 %4, %4, %5, %5\n v_min3_f32 %5, %5, %6, %6\n v_min3_f32 %6, %6, %7, %7\n v_min3_f32 %7, %7, %0, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==13) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==14) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==15) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==16) { REP8(asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %1, %2\n v_cmp_eq_u32 vcc, %2, %3\n v_cmp_eq_u32 vcc, %3, %4\n v_cmp_eq_u32 vcc, %4, %5\n v_cmp_eq_u32 vcc, %5, %6\n v_cmp_eq_u32 vcc, %6, %7\n v_cmp_eq_u32 vcc, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==17) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 3, %2\n v_lshl_or_b32 %2, %2, 3, %3\n v_lshl_or_b32 %3, %3, 3, %4\n v_lshl_or_b32 %4, %4, 3, %5\n v_lshl_or_b32 %5, %5, 3, %6\n v_lshl_or_b32 %6, %6, 3, %7\n v_lshl_or_b32 %7, %7, 3, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==18) { REP8(asm volatile("v_ashrrev_i32 %0, 31, %1\n v_ashrrev_i32 %1, 31, %2\n v_ashrrev_i32 %2, 31, %3\n v_ashrrev_i32 %3, 31, %4\n v_ashrrev_i32 %4, 31, %5\n v_ashrrev_i32 %5, 31, %6\n v_ashrrev_i32 %6, 31, %7\n v_ashrrev_i32 %7, 31, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==19) { REP8(asm volatile("v_sub_f32 %0, |%0|, %1\n v_sub_f32 %1, |%1|, %2\n v_sub_f32 %2, |%2|, %3\n v_sub_f32 %3, |%3|, %4\n v_sub_f32 %4, |%4|, %5\n v_sub_f32 %5, |%5|, %6\n v_sub_f32 %6, |%6|, %7\n v_sub_f32 %7, |%7|, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==20) { REP8(asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==21) { REP8(asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, %3, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %4 bitop3:0x96\n v_bitop3_b32 %4, %4, %5, %5 bitop3:0x96\n v_bitop3_b32 %5, %5, %6, %6 bitop3:0x96\n v_bitop3_b32 %6, %6, %7, %7 bitop3:0x96\n v_bitop3_b32 %7, %7, %0, %0 bitop3:0x96" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==22) { REP8(asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==23) { REP8(asm volatile("v_and_b32 %0, 0x7fffffff, %1\n v_and_b32 %1, 0x7fffffff, %2\n v_and_b32 %2, 0x7fffffff, %3\n v_and_b32 %3, 0x7fffffff, %4\n v_and_b32 %4, 0x7fffffff, %5\n v_and_b32 %5, 0x7fffffff, %6\n v_and_b32 %6, 0x7fffffff, %7\n v_and_b32 %7, 0x7fffffff, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==24) { REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0" : "+v"(p),"+v"(q),"+v"(r),"+v"(s));) }
  if (MODE==25) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0" : "+v"(p),"+v"(q),"+v"(r),"+v"(s));) }
  if (MODE==26) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n v_mad_u32_u24 %1, %1, %2, %2\n v_mad_u32_u24 %2, %2, %3, %3\n v_mad_u32_u24 %3, %3, %4, %4\n v_mad_u32_u24 %4, %4, %5, %5\n v_mad_u32_u24 %5, %5, %6, %6\n v_mad_u32_u24 %6, %6, %7, %7\n v_mad_u32_u24 %7, %7, %0, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==27) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %1\n v_perm_b32 %1, %1, %2, %2\n v_perm_b32 %2, %2, %3, %3\n v_perm_b32 %3, %3, %4, %4\n v_perm_b32 %4, %4, %5, %5\n v_perm_b32 %5, %5, %6, %6\n v_perm_b32 %6, %6, %7, %7\n v_perm_b32 %7, %7, %0, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==28) { REP8(asm volatile("v_bfe_u32 %0, %1, 3, 16\n v_bfe_u32 %1, %2, 3, 16\n v_bfe_u32 %2, %3, 3, 16\n v_bfe_u32 %3, %4, 3, 16\n v_bfe_u32 %4, %5, 3, 16\n v_bfe_u32 %5, %6, 3, 16\n v_bfe_u32 %6, %7, 3, 16\n v_bfe_u32 %7, %0, 3, 16" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==29) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %4\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %5, %5, %6\n v_mul_u32_u24 %6, %6, %7\n v_mul_u32_u24 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==30) { REP8(asm volatile("v_max3_f32 %0, %0, %1, %1\n v_max3_f32 %1, %1, %2, %2\n v_max3_f32 %2, %2, %3, %3\n v_max3_f32 %3, %3, %4, %4\n v_max3_f32 %4, %4, %5, %5\n v_max3_f32 %5, %5, %6, %6\n v_max3_f32 %6, %6, %7, %7\n v_max3_f32 %7, %7, %0, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==31) { REP8(asm volatile("v_min_u16 %0, %0, %1\n v_min_u16 %1, %1, %2\n v_min_u16 %2, %2, %3\n v_min_u16 %3, %3, %4\n v_min_u16 %4, %4, %5\n v_min_u16 %5, %5, %6\n v_min_u16 %6, %6, %7\n v_min_u16 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==32) { REP8(asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %1, %1, %2\n v_pk_max_i16 %2, %2, %3\n v_pk_max_i16 %3, %3, %4\n v_pk_max_i16 %4, %4, %5\n v_pk_max_i16 %5, %5, %6\n v_pk_max_i16 %6, %6, %7\n v_pk_max_i16 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==33) { REP8(asm volatile("v_pk_add_i16 %0, %0, %1\n v_pk_add_i16 %1, %1, %2\n v_pk_add_i16 %2, %2, %3\n v_pk_add_i16 %3, %3, %4\n v_pk_add_i16 %4, %4, %5\n v_pk_add_i16 %5, %5, %6\n v_pk_add_i16 %6, %6, %7\n v_pk_add_i16 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
  if (MODE==34) { REP8(asm volatile("v_pk_sub_i16 %0, %0, %1\n v_pk_sub_i16 %1, %1, %2\n v_pk_sub_i16 %2, %2, %3\n v_pk_sub_i16 %3, %3, %4\n v_pk_sub_i16 %4, %4, %5\n v_pk_sub_i16 %5, %5, %6\n v_pk_sub_i16 %6, %6, %7\n v_pk_sub_i16 %7, %7, %0" : "+v"(a),"+v"(b),"+v"(c),"+v"(d),"+v"(e),"+v"(f),"+v"(g),"+v"(h) :: "vcc");) }
 }
 out[blockIdx.x*1024+threadIdx.x]=a^b^c^d^e^f^g^h^(unsigned)(p+q+r+s);
}
template <int MODE> void run(const char *name, int per_iter) {
 unsigned *out; hipMalloc(&out, 256*1024*4); const int iters=4000; hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
 k<MODE><<<256,1024>>>(out,10); hipEventRecord(a); k<MODE><<<256,1024>>>(out,iters); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b);
 printf("%-28s %.3f ms -> %.2f cycles/wave64 instr/SIMD @2.4GHz\n", name, ms, ms*1e-3*2.4e9/((double)iters*per_iter*4)); hipFree(out); }
int main(){
 run<0>("v_and_b32", 64);
 run<1>("v_or_b32", 64);
 run<2>("v_add_u32", 64);
 run<3>("v_sub_u32", 64);
 run<4>("v_add_f32", 64);
 run<5>("v_mul_f32", 64);
 run<6>("v_fma_f32", 64);
 run<7>("v_min_f32", 64);
 run<8>("v_max_f32", 64);
 run<9>("v_min_i32", 64);
 run<10>("v_max_u32", 64);
 run<11>("v_med3_f32", 64);
 run<12>("v_min3_f32", 64);
 run<13>("v_mov_b32", 64);
 run<14>("v_cndmask_b32(vcc)", 64);
 run<15>("v_cmp_lt_f32", 64);
 run<16>("v_cmp_eq_u32", 64);
 run<17>("v_lshl_or_b32", 64);
 run<18>("v_ashrrev_i32", 64);
 run<19>("v_sub_f32 |abs| (VOP3)", 64);
 run<20>("v_add_f32 sdwa", 64);
 run<21>("v_xor3/bitop3", 64);
 run<22>("v_cvt_f32_u32 sdwa W0", 64);
 run<23>("v_and_b32 literal", 64);
 run<24>("v_pk_add_f32 (indep)", 32);
 run<25>("v_pk_mul_f32 (indep)", 32);
 run<26>("v_mad_u32_u24", 64);
 run<27>("v_perm_b32", 64);
 run<28>("v_bfe_u32", 64);
 run<29>("v_mul_u32_u24", 64);
 run<30>("v_max3_f32", 64);
 run<31>("v_min_u16", 64);
 run<32>("v_pk_max_i16", 64);
 run<33>("v_pk_add_i16", 64);
 run<34>("v_pk_sub_i16", 64);
 return 0; }
