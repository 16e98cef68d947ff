// vpass_place.cpp -- host-side placement of variables for the variable pass of the fused
// floating-point min-sum kernel (fused_float.hip): see ldpc_place_variable_pass().
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "lds_schedule.h"

// Variable pass placement.  The check pass fixed each variable's LDS bank (word mod 32); its
// rank within the bank is still free, and it decides which 16 variables share a lane group of
// a ds_read_b128 in the variable pass (thread tid rebuilds words tid + B*i: lane = word mod 64,
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32).  A group is served in one
// LDS cycle when its 16 records sit in 16 different bank quads (record index mod 16); random
// ranks give ~3.  Local search over swaps of two variables of one bank, one of them in an over-full
// class of its group: cost = sum over groups and edges of the busiest quad.  Returns the extra LDS
// cycles left (and before).  PEG-16000, 0.7 s: 6 178 -> 1 800 (ten times the moves: 1 250; a rank's
// last variables have a 1 in 4 096 chance of fitting three edges at once, so zero is out of reach).
//
// wide = true: the wide-row kernel's layout instead -- {min1, min2} pairs of 8 bytes read with ds_read_b64, which is
// served in two groups of 32 lanes (lanes 0-31, 32-63), one LDS cycle each when the 32 records sit in 32 different
// bank pairs (record index mod 32).
static void place_variable_pass(const ldpc_code &c, LdsSchedule &sc, long *before, long *after, const bool wide) {
  const int QM = wide ? 32 : 16;          // bank classes of a record
  std::vector<int> row_of(c.E());
  for (int j = 0; j < c.m; j++)
    for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++) row_of[e] = j;
  const int W = sc.var_words, CM = std::max(1, c.cmax), NT = ((W + 63) / 64) * (wide ? 2 : 4);
  std::vector<int> occ(W, -1);
  for (int v = 0; v < c.n; v++) occ[sc.pos[v]] = v;
  auto tuple_of = [wide](int w) {
    const int l = w & 63, b = l & 31;
    if (wide) return (w >> 6) * 2 + (l >> 5);
    const bool a = b <= 3 || (b >= 12 && b <= 15) || (b >= 20 && b <= 27);
    return (w >> 6) * 4 + (l >> 5) * 2 + (a ? 0 : 1);
  };
  std::vector<std::vector<int>> members(NT);
  for (int w = 0; w < W; w++) members[tuple_of(w)].push_back(w);
  std::vector<int8_t> quad((size_t)c.n * CM, -1);
  for (int v = 0; v < c.n; v++)
    for (int kk = 0; kk < c.col_ptr[v + 1] - c.col_ptr[v]; kk++)
      quad[(size_t)v * CM + kk] = (int8_t)(row_of[c.edge_of[c.col_ptr[v] + kk]] & (QM - 1));
  auto cost = [&](int t) {
    int total = 0;
    for (int kk = 0; kk < CM; kk++) {
      int cnt[32] = {0}, mx = 0;
      for (int w : members[t]) {
        const int v = occ[w];
        if (v < 0) continue;
        const int q = quad[(size_t)v * CM + kk];
        if (q >= 0) mx = std::max(mx, ++cnt[q]);
      }
      total += mx > 0 ? mx - 1 : 0;
    }
    return total;
  };
  long sum = 0;
  for (int t = 0; t < NT; t++) sum += cost(t);
  *before = sum;
  // local search on swaps of two variables of one bank.  Energy per (group, edge):
  // 16 * (busiest quad - 1) + number of surplus entries (the second term guides the search
  // where the first is flat).  Counts are kept per (group, edge, quad).
  std::vector<uint8_t> cnt((size_t)NT * CM * QM, 0);
  auto energy_of = [&](int t, int kk) {
    const uint8_t *q = &cnt[((size_t)t * CM + kk) * QM];
    int mx = 0, surplus = 0;
    for (int i = 0; i < QM; i++) { mx = std::max<int>(mx, q[i]); surplus += q[i] > 1 ? q[i] - 1 : 0; }
    return QM * (mx > 0 ? mx - 1 : 0) + surplus;
  };
  auto add = [&](int t, int v, int d) {
    if (v < 0) return;
    for (int kk = 0; kk < CM; kk++) {
      const int q = quad[(size_t)v * CM + kk];
      if (q >= 0) cnt[((size_t)t * CM + kk) * QM + q] = (uint8_t)(cnt[((size_t)t * CM + kk) * QM + q] + d);
    }
  };
  std::vector<int> tw(W);
  for (int w = 0; w < W; w++) { tw[w] = tuple_of(w); add(tw[w], occ[w], +1); }
  auto energy_t = [&](int t) { int e = 0; for (int kk = 0; kk < CM; kk++) e += energy_of(t, kk); return e; };
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 32); };
  const int ranks = W / 32;
  long moves = std::min<long>(150L * W, 8000000L);
  int budget0 = 0;   // greedy with sideways moves: every annealing schedule tried ended higher in the same time (PEG-16000: 12 -> 2 642, 4 -> 2 314, 0 -> 1 969 extra cycles)
  if (const char *e = getenv("LDPC_HIP_VPLACE_MOVES")) moves = (long)(atof(e) * W);      // (tuning: moves per LDS word)
  if (const char *e = getenv("LDPC_HIP_VPLACE_BUDGET")) budget0 = atoi(e);               // (tuning: first uphill threshold)
  // acceptance of a move that changes the energy by d: d <= budget, a threshold that falls linearly to 0 (0 from the start by default)
  // one end of a swap is a variable that sits in an over-full class of some group (moves between two settled
  // variables only shuffle the plateau)
  auto in_conflict = [&](int w) {
    const int v = occ[w];
    if (v < 0) return false;
    for (int kk = 0; kk < CM; kk++) {
      const int q = quad[(size_t)v * CM + kk];
      if (q >= 0 && cnt[((size_t)tw[w] * CM + kk) * QM + q] > 1) return true;
    }
    return false;
  };
  int focus = 1;
  if (const char *e = getenv("LDPC_HIP_VPLACE_FOCUS")) focus = atoi(e);
  int best_of = 1;
  if (const char *e = getenv("LDPC_HIP_VPLACE_BESTOF")) best_of = std::max(1, atoi(e));
  auto swap_counts = [&](int t, int t2, int va, int vb) {   // va leaves t for t2, vb leaves t2 for t
    add(t, va, -1); add(t2, vb, -1);
    add(t, vb, +1); add(t2, va, +1);
  };
  for (long it = 0; it < moves && ranks > 1; it++) {
    int w = (int)(rnd() % (uint32_t)W);
    for (int tries = 0; focus && tries < 16 && !in_conflict(w); tries++) w = (int)(rnd() % (uint32_t)W);
    const int t = tw[w];
    // the best of `best_of` partners in the same bank
    int w2 = -1, d_best = 1 << 30;
    for (int k = 0; k < best_of; k++) {
      const int c2 = (w & 31) + 32 * (int)(rnd() % (uint32_t)ranks);
      const int t2 = tw[c2];
      if (t == t2 || (occ[w] < 0 && occ[c2] < 0)) continue;
      const int e0 = energy_t(t) + energy_t(t2);
      swap_counts(t, t2, occ[w], occ[c2]);
      const int d = energy_t(t) + energy_t(t2) - e0;
      swap_counts(t, t2, occ[c2], occ[w]);
      if (d < d_best) { d_best = d; w2 = c2; }
    }
    if (w2 < 0) continue;
    const int budget = (int)(budget0 * (moves - it) / moves);
    if (d_best <= budget - (int)(rnd() % (uint32_t)(budget + 1))) {
      swap_counts(t, tw[w2], occ[w], occ[w2]);
      std::swap(occ[w], occ[w2]);
    }
  }
  sum = 0;
  for (int t = 0; t < NT; t++) sum += cost(t);
  *after = sum;
  // apply: variables move within their bank, so the check pass's schedule stays conflict free
  std::vector<uint16_t> newword(sc.words);
  for (int w = 0; w < sc.words; w++) newword[w] = (uint16_t)w;
  for (int w = 0; w < W; w++)
    if (occ[w] >= 0) newword[sc.pos[occ[w]]] = (uint16_t)w;
  for (auto &x : sc.colT) x = newword[x];
  for (int w = 0; w < W; w++)
    if (occ[w] >= 0) sc.pos[occ[w]] = (uint16_t)w;
}


void ldpc_place_variable_pass(const ldpc_code &c, LdsSchedule &sc, long *before, long *after) {
  place_variable_pass(c, sc, before, after, false);
}
void ldpc_place_variable_pass_wide(const ldpc_code &c, LdsSchedule &sc, long *before, long *after) {
  place_variable_pass(c, sc, before, after, true);
}
