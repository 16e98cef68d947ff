// vpass_place.cpp -- host-side placement of variables for the variable pass of the fused
// floating-point min-sum kernel (fused_float.hip): see ldpc_place_variable_pass().
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <system_error>
#include <thread>
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
static void place_variable_pass_search(const ldpc_code &c, LdsSchedule &sc, long *before, long *after, const bool wide) {
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
  // QM * (busiest class - 1) + number of surplus entries (the second term guides the search
  // where the first is flat).  Kept incrementally per (group, edge): the count of every class, how many classes hold
  // each count, and the largest count -- a variable entering or leaving a group is O(1) per edge.
  struct Cell { uint8_t cnt[32]; uint8_t hist[40]; uint8_t mx; };   // hist[c] = classes with count c (a group has at most 32 members)
  // (cells of the two lane sets apart, a bank's words together: the two search threads then share no cache line)
  const int NTH = (NT + 1) / 2;
  auto ci = [NTH, wide](int t) { return wide ? t : (t & 1) * NTH + (t >> 1); };
  std::vector<Cell> cell((size_t)2 * NTH * CM);
  for (auto &x : cell) { std::fill(x.cnt, x.cnt + 32, 0); std::fill(x.hist, x.hist + 40, 0); x.hist[0] = (uint8_t)QM; x.mx = 0; }
  auto cell_energy_mx = [&](int mx) { return QM * (mx > 0 ? mx - 1 : 0); };
  auto bump = [&](Cell &x, int q, int d) {   // count of class q changes by d = +-1; returns the change of the cell's energy
    const int c = x.cnt[q], c2 = c + d;
    const int mx0 = x.mx;
    x.hist[c]--; x.hist[c2]++;
    x.cnt[q] = (uint8_t)c2;
    if (c2 > x.mx) x.mx = (uint8_t)c2;
    else if (d < 0 && c == x.mx && x.hist[c] == 0) x.mx = (uint8_t)c2;
    return cell_energy_mx(x.mx) - cell_energy_mx(mx0) + ((c2 > 1 ? c2 - 1 : 0) - (c > 1 ? c - 1 : 0));
  };
  auto add = [&](int t, int v, int d) {
    int de = 0;
    if (v < 0) return de;
    for (int kk = 0; kk < CM; kk++) {
      const int q = quad[(size_t)v * CM + kk];
      if (q >= 0) de += bump(cell[(size_t)ci(t) * CM + kk], q, d);
    }
    return de;
  };
  std::vector<int> tw(W);
  for (int w = 0; w < W; w++) { tw[w] = tuple_of(w); add(tw[w], occ[w], +1); }
  const int ranks = W / 32;
  // Plain descent with sideways moves: every annealing schedule tried ended higher in the same time (PEG-16000, 150
  // moves per word: first uphill threshold 12 -> 2 642 extra cycles, 4 -> 2 314, 0 -> 1 969), and so did the best of K
  // partners per move.  One end of a swap is a variable that sits in an over-full class of its group (moves between
  // two settled variables only shuffle the plateau), taken in turn: 1 969 -> 1 800; 600 moves per word: 1 270.
  // The two lane sets of the ds_read_b128 layout share no bank and no group: two threads, one each.
  const int parts = wide ? 1 : 2;
  long moves = std::min<long>((wide ? 300L : 600L) * W, 32000000L) / parts;
  if (const char *e = getenv("LDPC_HIP_VPLACE_MOVES")) moves = (long)(atof(e) * W) / parts;      // (tuning: moves per LDS word)
  auto swap_counts = [&](int t, int t2, int va, int vb) {   // va leaves t for t2, vb leaves t2 for t
    return add(t, va, -1) + add(t2, vb, -1) + add(t, vb, +1) + add(t2, va, +1);
  };
  // a thread works on its own copy of the word -> variable map (it only ever touches the words of its lane set) and
  // on its own cells; its words are copied back at the end
  auto search = [&](int part, std::vector<int> &o) {
    uint64_t rng = 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull * (uint64_t)part;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 32); };
    auto mine = [&](int w) { return parts == 1 || (tw[w] & 1) == part; };
    auto in_conflict = [&](int w) {
      const int v = o[w];
      if (v < 0) return false;
      for (int kk = 0; kk < CM; kk++) {
        const int q = quad[(size_t)v * CM + kk];
        if (q >= 0 && cell[(size_t)ci(tw[w]) * CM + kk].cnt[q] > 1) return true;
      }
      return false;
    };
    int w = 0;
    for (long it = 0; it < moves && ranks > 1; it++) {
      // (the words in turn, skipping the settled ones: no random probes wasted on them)
      int tries = 0;
      for (; tries < W; tries++) {
        w = w + 1 < W ? w + 1 : 0;
        if (mine(w) && in_conflict(w)) break;
      }
      if (tries == W) break;                                   // nothing left to improve
      const int w2 = (w & 31) + 32 * (int)(rnd() % (uint32_t)ranks);
      const int t = tw[w], t2 = tw[w2];
      if (t == t2 || (o[w] < 0 && o[w2] < 0)) continue;
      if (swap_counts(t, t2, o[w], o[w2]) <= 0) std::swap(o[w], o[w2]);
      else swap_counts(t, t2, o[w2], o[w]);
    }
  };
  if (parts == 2) {
    std::vector<int> o0 = occ, o1 = occ;
    std::thread other;
    try {
      other = std::thread(search, 1, std::ref(o1));
    } catch (const std::system_error &) {      // no thread to be had: one after the other
    }
    search(0, o0);
    if (other.joinable()) other.join();
    else search(1, o1);
    for (int w = 0; w < W; w++) occ[w] = (tw[w] & 1) ? o1[w] : o0[w];
  } else {
    search(0, occ);
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


// The search is deterministic and takes a second or two on the long codes: a process that makes several decoders of
// one code (a sweep over algorithms, the test suite) pays for it once.  Keyed by a hash of the graph and of the
// schedule the search starts from; a handful of entries, dropped together when full.
namespace {
struct PlacedEntry { std::vector<uint16_t> pos, colT; long before, after; };
std::mutex g_placed_mutex;
std::map<uint64_t, PlacedEntry> g_placed;
uint64_t fnv(uint64_t h, const void *p, size_t bytes) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < bytes; i++) h = (h ^ b[i]) * 0x100000001B3ull;
  return h;
}
}  // namespace
static void place_variable_pass(const ldpc_code &c, LdsSchedule &sc, long *before, long *after, const bool wide) {
  uint64_t key = 0xCBF29CE484222325ull ^ (wide ? 0x9E37ull : 0);
  key = fnv(key, c.row_ptr.data(), c.row_ptr.size() * sizeof(int32_t));
  key = fnv(key, c.col_idx.data(), c.col_idx.size() * sizeof(int32_t));
  key = fnv(key, sc.pos.data(), sc.pos.size() * sizeof(uint16_t));
  key = fnv(key, sc.colT.data(), sc.colT.size() * sizeof(uint16_t));
  if (const char *e = getenv("LDPC_HIP_VPLACE_MOVES")) key = fnv(key, e, strlen(e));
  {
    std::lock_guard<std::mutex> lock(g_placed_mutex);
    auto it = g_placed.find(key);
    if (it != g_placed.end() && it->second.pos.size() == sc.pos.size() && it->second.colT.size() == sc.colT.size()) {
      sc.pos = it->second.pos; sc.colT = it->second.colT;
      *before = it->second.before; *after = it->second.after;
      return;
    }
  }
  place_variable_pass_search(c, sc, before, after, wide);
  std::lock_guard<std::mutex> lock(g_placed_mutex);
  if (g_placed.size() >= 16) g_placed.clear();
  g_placed[key] = PlacedEntry{sc.pos, sc.colT, *before, *after};
}

void ldpc_place_variable_pass(const ldpc_code &c, LdsSchedule &sc, long *before, long *after) {
  place_variable_pass(c, sc, before, after, false);
}
void ldpc_place_variable_pass_wide(const ldpc_code &c, LdsSchedule &sc, long *before, long *after) {
  place_variable_pass(c, sc, before, after, true);
}
