// valu_rates3.hip -- issue rate per SIMD (4 waves per SIMD, one 1024-thread block per CU) of
// candidate instructions for the fused decoder's slot loops.  Independent 4-chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define K4(ins) asm volatile(ins(%0, %0, %1, %2) "\n" ins(%1, %1, %2, %3) "\n" ins(%2, %2, %3, %0) "\n" ins(%3, %3, %0, %1) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0));
#define I_BITOP3(d, x, y, z) "v_bitop3_b32 " #d ", " #x ", " #y ", " #z " bitop3:0x96"
#define I_ADDU(d, x, y, z) "v_add_u32 " #d ", " #x ", " #y
#define I_SUBU(d, x, y, z) "v_sub_u32 " #d ", " #x ", " #y
#define I_AND(d, x, y, z) "v_and_b32 " #d ", " #x ", " #y
#define I_OR(d, x, y, z) "v_or_b32 " #d ", " #x ", " #y
#define I_FMA(d, x, y, z) "v_fma_f32 " #d ", " #x ", " #y ", " #z
#define I_FMAC(d, x, y, z) "v_fmac_f32 " #d ", " #y ", " #z
#define I_MULF(d, x, y, z) "v_mul_f32 " #d ", " #x ", " #y
#define I_MAXF(d, x, y, z) "v_max_f32 " #d ", " #x ", " #y
#define I_MINF(d, x, y, z) "v_min_f32 " #d ", " #x ", " #y
#define I_MED3F(d, x, y, z) "v_med3_f32 " #d ", " #x ", " #y ", " #z
#define I_MAXI(d, x, y, z) "v_max_i32 " #d ", " #x ", " #y
#define I_MAD24(d, x, y, z) "v_mad_u32_u24 " #d ", " #x ", " #y ", " #z
#define I_MUL24(d, x, y, z) "v_mul_u32_u24 " #d ", " #x ", " #y
#define I_ADD3(d, x, y, z) "v_add3_u32 " #d ", " #x ", " #y ", " #z
#define I_LSHLADD(d, x, y, z) "v_lshl_add_u32 " #d ", " #x ", 1, " #z
#define I_XAD(d, x, y, z) "v_xad_u32 " #d ", " #x ", " #y ", " #z
#define I_MOV(d, x, y, z) "v_mov_b32 " #d ", " #y
#define I_PERM(d, x, y, z) "v_perm_b32 " #d ", " #x ", " #y ", " #z
#define I_BFE(d, x, y, z) "v_bfe_u32 " #d ", " #y ", 16, 16"
#define I_SUBABS(d, x, y, z) "v_sub_f32 " #d ", |" #x "|, " #y
#define I_SUBNEG(d, x, y, z) "v_sub_f32 " #d ", -" #x ", " #y
#define I_CNDS(d, x, y, z) "v_cndmask_b32 " #d ", " #x ", " #y ", vcc"
#define I_CMP(d, x, y, z) "v_cmp_lt_u32 vcc, " #x ", " #y
#define I_CMPS(d, x, y, z) "v_cmp_lt_u32 s[20:21], " #x ", " #y
#define I_ADDCO(d, x, y, z) "v_add_co_u32 " #d ", vcc, " #x ", " #y
#define I_PKADDU16(d, x, y, z) "v_pk_add_u16 " #d ", " #x ", " #y
#define I_PKSUBI16(d, x, y, z) "v_pk_sub_i16 " #d ", " #x ", " #y
#define I_PKMAXI16(d, x, y, z) "v_pk_max_i16 " #d ", " #x ", " #y
#define I_SADU8(d, x, y, z) "v_sad_u8 " #d ", " #x ", " #y ", " #z
#define I_ADDF16(d, x, y, z) "v_add_f16 " #d ", " #x ", " #y
#define I_PKADDF16(d, x, y, z) "v_pk_add_f16 " #d ", " #x ", " #y
#define I_PKMINF16(d, x, y, z) "v_pk_min_f16 " #d ", " #x ", " #y
#define I_DPPXOR(d, x, y, z) "v_xor_b32_dpp " #d ", " #x ", " #y " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define I_SDWAXOR(d, x, y, z) "v_xor_b32_sdwa " #d ", " #x ", " #y " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
#define I_ANDSDWA(d, x, y, z) "v_and_b32_sdwa " #d ", " #x ", " #y " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
#define I_SUBF_SDWA(d, x, y, z) "v_sub_f32_sdwa " #d ", " #x ", " #y " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"
#define I_ASHR(d, x, y, z) "v_ashrrev_i32 " #d ", 31, " #y
#define I_MIN3U(d, x, y, z) "v_min3_u32 " #d ", " #x ", " #y ", " #z
#define I_MAX3F(d, x, y, z) "v_max3_f32 " #d ", " #x ", " #y ", " #z
#define I_LSHR16(d, x, y, z) "v_lshrrev_b32 " #d ", 16, " #y
#define I_ASHR16(d, x, y, z) "v_ashrrev_i32 " #d ", 16, " #y
#define I_LSHL16(d, x, y, z) "v_lshlrev_b32 " #d ", 16, " #y
#define I_LSHRV(d, x, y, z) "v_lshrrev_b32 " #d ", " #x ", " #y
#define I_CVTUB(d, x, y, z) "v_cvt_f32_ubyte0 " #d ", " #y
#define I_CVTF16(d, x, y, z) "v_cvt_f32_f16 " #d ", " #y
#define I_SUBREV(d, x, y, z) "v_subrev_f32 " #d ", " #x ", " #y
#define I_ADDF(d, x, y, z) "v_add_f32 " #d ", " #x ", " #y
#define I_XNOR(d, x, y, z) "v_xnor_b32 " #d ", " #x ", " #y

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned *out, int iters, unsigned s0) {
  unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = threadIdx.x ^ 5, d = threadIdx.x + 7;
  for (int it = 0; it < iters; it++) {
#define CASE(n, INS) if (MODE == n) { REP8(K4(INS)) }
    CASE(0, I_BITOP3) CASE(1, I_ADDU) CASE(2, I_SUBU) CASE(3, I_AND) CASE(4, I_OR) CASE(5, I_FMA) CASE(6, I_FMAC)
    CASE(7, I_MULF) CASE(8, I_MAXF) CASE(9, I_MINF) CASE(10, I_MED3F) CASE(11, I_MAXI) CASE(12, I_MAD24) CASE(13, I_MUL24)
    CASE(14, I_ADD3) CASE(15, I_LSHLADD) CASE(16, I_XAD) CASE(17, I_MOV) CASE(18, I_PERM) CASE(19, I_BFE) CASE(20, I_SUBABS)
    CASE(21, I_SUBNEG) CASE(22, I_CNDS) CASE(23, I_CMP) CASE(24, I_CMPS) CASE(25, I_ADDCO) CASE(26, I_PKADDU16)
    CASE(27, I_PKSUBI16) CASE(28, I_PKMAXI16) CASE(29, I_SADU8) CASE(30, I_ADDF16) CASE(31, I_PKADDF16) CASE(32, I_PKMINF16)
    CASE(33, I_DPPXOR) CASE(34, I_SDWAXOR) CASE(35, I_ANDSDWA) CASE(36, I_SUBF_SDWA) CASE(37, I_ASHR) CASE(38, I_MIN3U) CASE(39, I_MAX3F) CASE(40, I_LSHR16) CASE(41, I_ASHR16) CASE(42, I_LSHL16) CASE(43, I_LSHRV) CASE(44, I_CVTUB) CASE(45, I_CVTF16) CASE(46, I_SUBREV) CASE(47, I_ADDF) CASE(48, I_XNOR)
  }
  out[blockIdx.x * 1024 + threadIdx.x] = a ^ b ^ c ^ d;
}
template <int MODE> void run(const char *name) {
  unsigned *out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 3000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, 1024>>>(out, 10, 0x7fffffffu);
  hipEventRecord(a);
  k<MODE><<<256, 1024>>>(out, iters, 0x7fffffffu);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-22s %.2f cycles per wave64 instruction per SIMD (@2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / ((double)iters * 32 * 4));
  hipFree(out);
}
int main() {
#define R(n, s) run<n>(s);
  R(0, "v_bitop3_b32") R(1, "v_add_u32") R(2, "v_sub_u32") R(3, "v_and_b32") R(4, "v_or_b32") R(5, "v_fma_f32") R(6, "v_fmac_f32")
  R(7, "v_mul_f32") R(8, "v_max_f32") R(9, "v_min_f32") R(10, "v_med3_f32") R(11, "v_max_i32") R(12, "v_mad_u32_u24") R(13, "v_mul_u32_u24")
  R(14, "v_add3_u32") R(15, "v_lshl_add_u32") R(16, "v_xad_u32") R(17, "v_mov_b32") R(18, "v_perm_b32") R(19, "v_bfe_u32") R(20, "v_sub_f32 |x|")
  R(21, "v_sub_f32 -x") R(22, "v_cndmask vcc") R(23, "v_cmp -> vcc") R(24, "v_cmp -> sgpr") R(25, "v_add_co_u32") R(26, "v_pk_add_u16")
  R(27, "v_pk_sub_i16") R(28, "v_pk_max_i16") R(29, "v_sad_u8") R(30, "v_add_f16") R(31, "v_pk_add_f16") R(32, "v_pk_min_f16")
  R(33, "v_xor_b32 dpp") R(34, "v_xor_b32 sdwa") R(35, "v_and_b32 sdwa") R(36, "v_sub_f32 sdwa") R(37, "v_ashrrev_i32") R(38, "v_min3_u32") R(39, "v_max3_f32") R(40, "v_lshrrev_b32 16") R(41, "v_ashrrev_i32 16") R(42, "v_lshlrev_b32 16") R(43, "v_lshrrev_b32 var") R(44, "v_cvt_f32_ubyte0") R(45, "v_cvt_f32_f16") R(46, "v_subrev_f32") R(47, "v_add_f32") R(48, "v_xnor_b32")
  return 0;
}
