// tables.cpp -- host-built lookup tables and GF(q) arithmetic for libldpc_hip.
//
// The two integer decoders of the reference are table driven; the tables are
// filled once on the host with libm doubles (exactly what the reference does
// at start-up) and uploaded.  Formulas: inittab() test_code_B.c:52-74 and
// initlogexptab2() GFq_LDPC_NTT.c:196-219.
#include <cmath>
#include <cstdint>
#include <cstring>
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

// --- compact, exact forms of the two tables (round 2) ---
// The 12 MiB of tables are gathered four times per edge and symbol; every gather of a wavefront
// is 64 separate L2 requests, and that request rate bounded the GF(2^p) kernels (profiles/
// r02_gf4_wht_counters.txt).  Both tables have structure the kernels can use from LDS instead:
//
//  * fexp[i] is, for every one of its 2^21 entries, the rounding of 2^e * S(fr) * 2^14 with
//    i = 16384 e + fr and S(fr) the Simpson mean of 2^x over cell fr of [0, 1): glibc's pow(2, x)
//    scales exactly with the integer part of x.  Only bits 51..20 of S(fr) * 2^52 can reach a
//    result (e <= 17): xexp[16384] of 32 bits, 64 KiB, leading one implied.
//  * flog[i] = round(16384 log2(i)) - 14 * 16384 for all but 60 entries, all below 2048: the
//    first 2048 entries are kept, the rest is evaluated from a 512-segment quadratic of
//    16384 log2(1 + u) (error < 1e-5; results closer than 1e-4 to a rounding boundary are
//    NOT trusted: the kernel then reads the full table) -- lseg[512] = {integer part, fraction,
//    slope, curvature}.
// ldpc_wht_compact_selfcheck() replays the kernels' arithmetic over every index of both tables.
void ldpc_build_wht_compact(WhtCompact &c) {
  c.xexp.assign(kWhtFrac, 0);
  for (int fr = 0; fr < kWhtFrac; fr++) {
    const double d = fr * kPrec;
    const double mid = std::pow(2, d), right = std::pow(2, d - 0.5 * kPrec), left = std::pow(2, d + 0.5 * kPrec);
    const double S = (4 * mid + right + left) / 6.0;              // in [1, 2)
    const uint64_t X = (uint64_t)std::ldexp(S, 52);               // exact: S has 52 fraction bits
    c.xexp[fr] = (uint32_t)(X >> 20);                             // bits 51..20; bit 52 is the implied one
  }
  c.lseg.assign((size_t)kWhtSeg * 4, 0.0f);
  c.lsegi.assign(kWhtSeg, 0);
  const double ln2 = std::log(2.0);
  for (int h = 0; h < kWhtSeg; h++) {
    const double mh = 1.0 + (h + 0.5) / kWhtSeg;
    const double c0 = 16384.0 * std::log(mh) / ln2;
    const double ci = std::floor(c0);
    c.lsegi[h] = (int32_t)ci;
    c.lseg[(size_t)h * 4 + 0] = (float)(c0 - ci);
    c.lseg[(size_t)h * 4 + 1] = (float)(16384.0 / (ln2 * mh));
    c.lseg[(size_t)h * 4 + 2] = (float)(-16384.0 / (2.0 * ln2 * mh * mh));
  }
}

// The kernels' fexp (gf_engine.hip w_fexp_c), restated
uint32_t ldpc_wht_fexp_compact(const WhtCompact &c, int x) {
  const int L = 1 << (6 + kDeci);
  x = x < -L ? -L : x;
  x = x > L - 1 ? L - 1 : x;
  if (x > 17 * kWhtFrac) return 1u << 31;
  const int e = x >> kDeci, fr = x & (kWhtFrac - 1);
  const int sh = 18 - e;                                          // >= 1
  if (sh >= 34) return 0;
  const uint64_t Y = (1ull << 32) | c.xexp[fr];
  return (uint32_t)((Y + (1ull << (sh - 1))) >> sh);
}

// The kernels' flog fast path for 2048 <= i < 2^20; *defer = the result is not trusted
int ldpc_wht_flog_compact(const WhtCompact &c, int i, bool *defer) {
  int e = 31;
  while (!((unsigned)i >> e)) e--;
  const int r = i - (1 << e), sh = e - 9;
  const int h = r >> sh, lo = r & ((1 << sh) - 1);
  const float t = std::ldexp((float)lo, -e) - 0x1p-10f;
  const float *q = &c.lseg[(size_t)h * 4];
  const float w = std::fmaf(t, std::fmaf(t, q[2], q[1]), q[0]);
  const float fl = std::floor(w + 0.5f);
  *defer = std::fabs((w - fl)) > 0.5f - 1e-4f;
  return c.lsegi[h] + kFmul * (e - 14) + (int)fl;
}

long ldpc_wht_compact_selfcheck(long stats[4]) {
  std::vector<int32_t> flog;
  std::vector<uint32_t> fexp;
  ldpc_build_wht_tables(flog, fexp);
  WhtCompact c;
  ldpc_build_wht_compact(c);
  const int L = 1 << (6 + kDeci);
  long bad_exp = 0, bad_log = 0, deferred = 0;
  for (int i = -L; i < L; i++)
    if (ldpc_wht_fexp_compact(c, i) != fexp[(size_t)(i + L)]) bad_exp++;
  for (int i = kWhtSmall; i < L; i++) {
    bool d;
    const int v = ldpc_wht_flog_compact(c, i, &d);
    if (d) deferred++;
    else if (v != flog[i]) bad_log++;
  }
  stats[0] = bad_exp; stats[1] = bad_log; stats[2] = deferred; stats[3] = kWhtSmall;
  return bad_exp + bad_log;
}

// --- compact, exact form of the two Q8.14 sum-product tables (engine.h has the format) ---
// The check pass of the table decoder gathers three table entries per edge and frame from 32 MiB of
// tables: 64 separate L1 requests per wavefront instruction, which is what bounded it (2.7 Gbit/s on
// PEG-16000 where the int32 min-sum pass, same traffic, is bandwidth-bound at 7.8).  The compact image
// is built FROM the full tables (so whatever libm produced is what it holds) and refused when a table
// does not have the structure the format needs; ldpc_spa_compact_selfcheck() replays the kernels'
// arithmetic over all 2^22 arguments of both tables.
bool ldpc_build_spa_compact(const std::vector<int32_t> &flogtanh, const std::vector<int32_t> &fgallag,
                            std::vector<uint32_t> &image) {
  const int levels = 1 << (8 + kDeci);
  if ((int)flogtanh.size() != levels || (int)fgallag.size() != levels) return false;
  image.assign(kTabcBytes / 4, 0);
  for (int t = 0; t < 2; t++) {
    const std::vector<int32_t> &src = t ? fgallag : flogtanh;
    auto mag = [&](int i) { return t ? src[(size_t)i] : -src[(size_t)i]; };
    unsigned char *base = reinterpret_cast<unsigned char *>(image.data()) + (size_t)t * kTabcTable;
    uint32_t *a32 = reinterpret_cast<uint32_t *>(base);
    uint16_t *b16 = reinterpret_cast<uint16_t *>(base + kTabcOffB);
    uint32_t *c32 = reinterpret_cast<uint32_t *>(base + kTabcOffC);
    for (int i = 0; i < kTabcSmall; i++) {
      if (mag(i) < 0) return false;
      a32[i] = (uint32_t)mag(i);
    }
    a32[kTabcSmall] = 0;
    b16[0] = 0;                                             // (entry 2047)
    for (int i = kTabcSmall; i < kTabcSplit; i++) {
      if (mag(i) < 0 || mag(i) > 65535) return false;
      b16[i - (kTabcSmall - 1)] = (uint16_t)mag(i);
    }
    b16[kTabcSplit - (kTabcSmall - 1)] = 0;
    for (int b = 0; b < kTabcBlocks; b++) {
      const int i0 = kTabcSplit + 16 * b;
      const int start = mag(i0 - 1);
      if (start < 0 || start > 65535) return false;
      uint32_t w = (uint32_t)start << 16;
      for (int k = 0; k < 16; k++) {
        const int step = mag(i0 + k - 1) - mag(i0 + k);
        if (step != 0 && step != 1) return false;
        w |= (uint32_t)step << k;
      }
      c32[b] = w;
    }
    for (int i = kTabcEnd - 1; i < levels; i++)
      if (src[(size_t)i] != 0) return false;
  }
  return true;
}

int ldpc_spa_compact_mag(const uint32_t *image, int table, unsigned i) {
  const unsigned char *t = reinterpret_cast<const unsigned char *>(image) + (size_t)table * kTabcTable;
  auto umin = [](unsigned a, unsigned b) { return a < b ? a : b; };
  uint32_t a, w;
  uint16_t bq;
  const unsigned ia = umin(i, kTabcSmall);
  const unsigned ib = i < (unsigned)kTabcSmall - 1 ? (unsigned)kTabcSmall - 1 : umin(i, kTabcSplit);
  const unsigned j = umin(i - (unsigned)kTabcSplit, 16u * kTabcBlocks - 1u);
  std::memcpy(&a, t + 4 * ia, 4);
  std::memcpy(&bq, t + (kTabcOffB - 2 * (kTabcSmall - 1)) + 2 * ib, 2);
  std::memcpy(&w, t + kTabcOffC + ((j >> 2) & ~3u), 4);
  return (int)(a + bq + (w >> 16)) - __builtin_popcount(w & ((2u << (j & 15u)) - 1u));
}

long ldpc_spa_compact_selfcheck(long stats[4]) {
  std::vector<int32_t> lt, ga;
  ldpc_build_spa_tables(lt, ga);
  std::vector<uint32_t> image;
  const bool ok = ldpc_build_spa_compact(lt, ga, image);
  long bad_l = 0, bad_g = 0;
  if (ok)
    for (unsigned i = 0; i < (unsigned)lt.size(); i++) {
      if (-ldpc_spa_compact_mag(image.data(), 0, i) != lt[i]) bad_l++;
      if (ldpc_spa_compact_mag(image.data(), 1, i) != ga[i]) bad_g++;
    }
  stats[0] = ok ? 1 : 0; stats[1] = bad_l; stats[2] = bad_g; stats[3] = kTabcBytes;
  return ok ? bad_l + bad_g : -1;
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
