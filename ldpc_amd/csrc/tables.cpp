// tables.cpp -- host-built lookup tables and GF(q) arithmetic for libldpc_hip.
//
// The two integer decoders of the reference are table driven; the tables are
// filled once on the host with libm doubles (exactly what the reference does
// at start-up) and uploaded.  Formulas: inittab() test_code_B.c:52-74 and
// initlogexptab2() GFq_LDPC_NTT.c:196-219.
#include <cmath>
#include <cstdint>
#include <vector>

#include "engine.h"

namespace {

// --- Q8.14 Gallager tables (binary table sum-product) ---
constexpr int kDeci = 14;
constexpr int kFmul = 1 << kDeci;
constexpr double kPrec = 1.0 / kFmul;

int fix14(double x) {   // float2fix, test_code_B.c:36-43
  if (x >= 0) return (int)(x * kFmul + 0.5);
  return -(int)((-x) * kFmul + 0.5);
}
double lt2(double x) { return std::log(std::tanh(std::fabs(x * 0.5))); }          // logtanh2 :27-30
double at2(double x) { return std::log((1.0 + x) / (1.0 - x)); }                  // atanh2 :23-26

}  // namespace

void ldpc_build_spa_tables(std::vector<int32_t> &flogtanh, std::vector<int32_t> &fgallag) {
  const int levels = 1 << (8 + kDeci);
  flogtanh.assign(levels, 0);
  fgallag.assign(levels, 0);
  // Simpson average over each quantisation cell: (4 f(d) + f(d-h/2) + f(d+h/2)) / 6
  flogtanh[0] = -kFmul * 14;
  double right = lt2(1 * kPrec - 0.5 * kPrec);
  for (int i = 1; i < levels; i++) {
    const double d = i * kPrec;
    const double left = lt2(d + 0.5 * kPrec);
    flogtanh[i] = fix14((4 * lt2(d) + right + left) / 6.0);
    right = left;
  }
  fgallag[0] = kFmul * 14;
  right = at2(std::exp(-1 * kPrec - 0.5 * kPrec));
  for (int i = 1; i < levels; i++) {
    const double d = (-i) * kPrec;
    const double mid = at2(std::exp(d));
    const double left = at2(std::exp(d + 0.5 * kPrec));
    fgallag[i] = fix14((4 * mid + right + left) / 6.0);
    right = left;
  }
}

// --- log2 / exp2 tables of the GF(2^p) WHT decoder (INT 6, DECI 14) ---
void ldpc_build_wht_tables(std::vector<int32_t> &flog, std::vector<uint32_t> &fexp) {
  const int levels = 1 << (6 + kDeci);
  auto l2 = [](double x) { return std::log(x) / std::log(2.0); };   // GFq_LDPC_NTT.c:194
  auto e2 = [](double x) { return std::pow(2, x); };               // GFq_LDPC_NTT.c:193
  flog.assign(levels, 0);
  fexp.assign((size_t)levels * 2, 0);   // index i + levels holds Fexp(i), i in [-levels, levels)
  flog[0] = -kFmul * 14;
  double right = l2(1 * kPrec - 0.5 * kPrec);
  for (int i = 1; i < levels; i++) {
    const double d = i * kPrec;
    const double left = l2(d + 0.5 * kPrec);
    flog[i] = fix14((4 * l2(d) + right + left) / 6.0);
    right = left;
  }
  right = e2((-levels) * kPrec - 0.5 * kPrec);
  for (int i = -levels; i < levels; i++) {
    const double d = i * kPrec;
    const double mid = e2(d);
    const double left = e2(d + 0.5 * kPrec);
    if (mid > (double)(1UL << (31 - kDeci))) fexp[(size_t)(i + levels)] = 1U << 31;
    else fexp[(size_t)(i + levels)] = (uint32_t)((4 * mid + right + left) / 6.0 * kFmul + 0.5);
    right = left;
  }
}

// --- GF(q) ---
// GF(2^p): discrete-log tables of the primitive polynomials whose powers are
// the reference's expq literals (GFq_LDPC_NTT.c:25-92); prime q: mod q
// (GFq_LDPC.c:22-24).
void ldpc_gf_tables_host(int q, int *logq, int *expq) {
  static const int poly[9] = {0, 0, 0x7, 0xB, 0x13, 0x25, 0x43, 0x89, 0x11D};
  int p = 0;
  while ((1 << p) < q) p++;
  int x = 1;
  for (int i = 0; i < q - 1; i++) {
    expq[i] = x;
    logq[x] = i;
    x <<= 1;
    if (x & q) x ^= poly[p];
  }
  logq[0] = 0;
}

int ldpc_gf_mul_host(int q, int a, int b) {
  if ((q & (q - 1)) != 0) return (a * b) % q;
  if (q == 2) return a & b;
  if (a == 0 || b == 0) return 0;
  static thread_local int cached_q = 0;
  static thread_local int lg[256], ex[256];
  if (cached_q != q) {
    ldpc_gf_tables_host(q, lg, ex);
    cached_q = q;
  }
  return ex[(lg[a] + lg[b]) % (q - 1)];
}
