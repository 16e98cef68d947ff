// gf_engine.hip -- non-binary engines (GF(2^p) WHT-domain, GF(q) direct).
#include "engine.h"

Engine *ldpc_make_gf_engine(int alg, const ldpc_code &c) {
  (void)alg;
  (void)c;
  return nullptr;
}
