// valu_rates.hip -- micro-benchmark: issue rate of the VALU instructions the fused decoder
// uses, per SIMD, with 4 waves per SIMD (1024-thread blocks, one per CU).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned *out, int iters, unsigned s0) {
  unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = threadIdx.x ^ 5, d = threadIdx.x + 7;
  unsigned e = a + 11, f = b + 13, g = c + 17, h = d + 19;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { REP8(asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0));) }
    if (MODE == 1) { REP8(asm volatile("v_med3_u32 %0, %0, %1, %2\n v_med3_u32 %1, %1, %2, %3\n v_med3_u32 %2, %2, %3, %0\n v_med3_u32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 2) { REP8(asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_u32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 3) { REP8(asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_sub_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 4) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %1, %1, %2, 31\n v_alignbit_b32 %2, %2, %3, 31\n v_alignbit_b32 %3, %3, %0, 31" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 5) { REP8(asm volatile("v_cmp_eq_u32 vcc, 3, %0\n v_cndmask_b32 %1, %2, %3, vcc\n v_cmp_eq_u32 vcc, 4, %1\n v_cndmask_b32 %0, %3, %2, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
    if (MODE == 6) { REP8(asm volatile("v_add_u32_sdwa %0, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %1, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_add_u32_sdwa %2, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_add_u32_sdwa %3, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0));) }
    if (MODE == 7) { REP8(asm volatile("v_cvt_f32_i32 %0, %1\n v_cvt_f32_i32 %1, %2\n v_cvt_f32_i32 %2, %3\n v_cvt_f32_i32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 8) { REP8(asm volatile("v_lshlrev_b32 %0, 5, %1\n v_lshlrev_b32 %1, 6, %2\n v_lshlrev_b32 %2, 7, %3\n v_lshlrev_b32 %3, 8, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 9) { REP8(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 10) { REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %0" : "+v"(*(double *)&a), "+v"(*(double *)&e));) }
    if (MODE == 11) { REP8(asm volatile("v_pk_min_u16 %0, %0, %1\n v_pk_min_u16 %1, %1, %2\n v_pk_min_u16 %2, %2, %3\n v_pk_min_u16 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 12) { REP8(asm volatile("v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0));) }
    if (MODE == 13) { REP8(asm volatile("v_cvt_i32_f32 %0, %1\n v_cvt_i32_f32 %1, %2\n v_cvt_i32_f32 %2, %3\n v_cvt_i32_f32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

template <int MODE> void run(const char *name, int per_iter) {
  unsigned *out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 4000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, 1024>>>(out, 10, 0x7fffffffu);
  hipEventRecord(a);
  k<MODE><<<256, 1024>>>(out, iters, 0x7fffffffu);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr_per_simd = (double)iters * per_iter * 4;   // 4 waves per SIMD
  printf("%-28s %.3f ms  -> %.2f cycles per wave64 instruction per SIMD (@2.4 GHz)\n", name, ms,
         ms * 1e-3 * 2.4e9 / instr_per_simd);
  hipFree(out);
}

int main() {
  run<0>("v_and_or_b32", 32); run<1>("v_med3_u32", 32); run<2>("v_min_u32", 32); run<3>("v_sub_f32", 32);
  run<4>("v_alignbit_b32", 32); run<5>("v_cmp_eq_u32+v_cndmask", 32); run<6>("v_add_u32_sdwa", 32);
  run<7>("v_cvt_f32_i32", 32); run<8>("v_lshlrev_b32", 32); run<9>("v_xor_b32", 32); run<10>("v_pk_add_f32", 16);
  run<11>("v_pk_min_u16", 32); run<12>("v_bfi_b32", 32); run<13>("v_cvt_i32_f32", 32);
  return 0;
}
