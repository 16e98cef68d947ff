// fused_inst.hip -- instantiations of the register-cached fused Q8.8 decoder
// (fused_kernels.hip.h), one kernel per (checks per thread, slot-pair count R2, offset
// shift SH).  Compiled FQ_PARTS times with -DFQ_PART=p so the ~140 kernels build in
// parallel; part p holds the R2 values whose position in the list is p mod FQ_PARTS.
#include "fused_kernels.hip.h"

#ifndef FQ_PART
#error "compile with -DFQ_PART=<0..FQ_PARTS-1>"
#endif

namespace ldpc {

template <int CPT, int R2, int SH>
static FusedKern fq_one() {
  if constexpr (fq_reg_max_block(CPT, R2) == 1024) return fused_q88_reg_kernel<CPT, R2, SH, 1024, 4>;
  else if constexpr (fq_reg_max_block(CPT, R2) == 768) return fused_q88_reg_kernel<CPT, R2, SH, 768, 3>;
  else return nullptr;
}

template <int R2>
static FusedKern fq_pick(int cpt, int sh) {
  switch (sh * 4 + cpt) {
    case 1: return fq_one<1, R2, 0>();
    case 2: return fq_one<2, R2, 0>();
    case 3: return fq_one<3, R2, 0>();
    case 9: return fq_one<1, R2, 2>();
    case 10: return fq_one<2, R2, 2>();
    case 11: return fq_one<3, R2, 2>();
  }
  return nullptr;
}

#define FQ_CASE(pos, r2) \
  case r2:               \
    if ((pos) % FQ_PARTS == FQ_PART) return fq_pick<((pos) % FQ_PARTS == FQ_PART) ? r2 : 0>(cpt, sh); \
    break;

// R2 = 0 is the "not in this part" placeholder: never instantiated for real
template <>
FusedKern fq_pick<0>(int, int) { return nullptr; }

#define FQ_CAT2(a, b) a##b
#define FQ_CAT(a, b) FQ_CAT2(a, b)

FusedKern FQ_CAT(fused_reg_lookup_part, FQ_PART)(int cpt, int r2, int sh) {
  switch (r2) {
    FQ_CASE(0, 2) FQ_CASE(1, 3) FQ_CASE(2, 4) FQ_CASE(3, 5) FQ_CASE(4, 6) FQ_CASE(5, 7) FQ_CASE(6, 8)
    FQ_CASE(7, 9) FQ_CASE(8, 10) FQ_CASE(9, 11) FQ_CASE(10, 12) FQ_CASE(11, 13) FQ_CASE(12, 14)
    FQ_CASE(13, 15) FQ_CASE(14, 16) FQ_CASE(18, 18) FQ_CASE(17, 20) FQ_CASE(16, 22) FQ_CASE(15, 24)
    FQ_CASE(19, 26) FQ_CASE(20, 28) FQ_CASE(21, 30) FQ_CASE(22, 32)
  }
  return nullptr;
}

}  // namespace ldpc
