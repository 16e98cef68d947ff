// lds_schedule.h -- LDS placement / slot schedule of the fused decoder (lds_schedule.cpp)
#pragma once
#include <cstdint>
#include <vector>

#include "graph.h"

struct LdsSchedule {
  int block = 0, rounds = 0, RM = 0, nw = 1;
  int var_words = 0, dummy_base = 0, ndummy = 0, words = 0;
  std::vector<uint16_t> pos;     // [n]  variable -> LDS word
  std::vector<uint16_t> colT;    // [rounds][RM][block]  LDS word visited by (check, slot)
  std::vector<uint32_t> maskT;   // [rounds][block][nw]  sign-mask bits of the real slots
  std::vector<uint8_t> dcT;      // [rounds][block]      row weight (0: no such check)
  long conflicts = 0, accesses = 0;   // extra LDS cycles / half-wave accesses per iteration pass
  int max_dummy_load = 0;             // most pad slots of the workgroup that point at one dummy word
};

// block: threads per workgroup (multiple of 64); RM: slots per check (even, >= rmax);
// ndummy: dummy words (multiple of 32).  optimise = false keeps the natural layout.
int ldpc_build_lds_schedule(const ldpc_code &c, int block, int RM, int ndummy, bool optimise,
                            LdsSchedule &out);

// Re-ranks the variables within their LDS banks (vpass_place.cpp) so that the record reads of
// the fused float kernel's variable pass meet few bank conflicts; rewrites sc.pos / sc.colT.
// *before / *after: modelled extra LDS cycles per iteration.
void ldpc_place_variable_pass(const ldpc_code &c, LdsSchedule &sc, long *before, long *after);
// the same for the wide-row kernel's 8-byte {min1, min2} pairs (two 32-lane groups per ds_read_b64, 32 bank pairs)
void ldpc_place_variable_pass_wide(const ldpc_code &c, LdsSchedule &sc, long *before, long *after);
