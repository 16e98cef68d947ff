// lds_schedule.cpp -- host-side "schedule compiler" of the fused (frame resident
// in LDS) decoder: decides WHERE in LDS each variable's sum lives and in WHICH
// slot each thread visits each edge of its checks, so that the 32 lanes of a
// half-wavefront hit 32 different LDS banks at every step.
//
// The kernel (fused_kernels.hip.h) walks the slots k = 0..RM-1 of its checks in
// lock step: at step k lane i reads, and later atomically updates, the sum of
// the variable in slot k of its check.  A b32 LDS access is served per 32-lane
// half wave, one cycle if the 32 word addresses fall in 32 different banks
// (bank = word mod 32), else one cycle per extra address on the busiest bank.
// With variables laid out in natural order the accesses are random: ~3.5 cycles
// per group, and that, not arithmetic, bounded the kernel.
//
// Two steps, both per code (once, at decoder creation):
//  1. bank assignment: every variable gets a bank such that, for every group of
//     32 checks that run as one half wave, no bank holds more than RM of the
//     group's edges (and banks are evenly filled overall).  Local search on the
//     sum of squared bin loads.
//  2. slot assignment: per group, the edges form a bipartite multigraph
//     lanes x banks of maximum degree <= RM; by Koenig's theorem its edges can
//     be coloured with RM colours so that no lane and no bank sees a colour
//     twice.  Colour = slot: every slot of the group is then conflict free.
//     Unused slots are pointed at dummy words of the banks still free in that
//     slot.
#include "lds_schedule.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>

namespace {

constexpr int NB = 32;   // banks seen by a 32-lane group of a b32 access

struct EdgeCol {
  // proper edge colouring of a bipartite multigraph (left = lanes, right = banks)
  int C;                                // colours available
  std::vector<int> atL, atR;            // [node][colour] -> edge or -1
  std::vector<int> el, er, ec;          // edge endpoints and colour
  EdgeCol(int nl, int nr, int colours) : C(colours), atL((size_t)nl * colours, -1), atR((size_t)nr * colours, -1) {}
  int freeL(int l) const { for (int c = 0; c < C; c++) if (atL[(size_t)l * C + c] < 0) return c; return -1; }
  int freeR(int r) const { for (int c = 0; c < C; c++) if (atR[(size_t)r * C + c] < 0) return c; return -1; }
  void set(int e, int c) { ec[e] = c; atL[(size_t)el[e] * C + c] = e; atR[(size_t)er[e] * C + c] = e; }
  void clr(int e) { int c = ec[e]; atL[(size_t)el[e] * C + c] = -1; atR[(size_t)er[e] * C + c] = -1; ec[e] = -1; }
  bool add(int l, int r) {
    const int e = (int)el.size();
    el.push_back(l); er.push_back(r); ec.push_back(-1);
    const int a = freeL(l), b = freeR(r);
    if (a < 0 || b < 0) return false;                 // degree above C: caller places it by hand
    if (atR[(size_t)r * C + a] < 0) { set(e, a); return true; }
    // a is free at l, b is free at r: flip the a/b alternating path that starts at r with colour a
    std::vector<int> path;
    int x = r, c = a;
    bool right = true;
    for (;;) {
      const int f = right ? atR[(size_t)x * C + c] : atL[(size_t)x * C + c];
      if (f < 0) break;
      path.push_back(f);
      x = right ? el[f] : er[f];
      right = !right;
      c = (c == a) ? b : a;
    }
    std::vector<int> nc(path.size());
    for (size_t i = 0; i < path.size(); i++) nc[i] = ec[path[i]] == a ? b : a;
    for (int f : path) clr(f);
    for (size_t i = 0; i < path.size(); i++) set(path[i], nc[i]);
    set(e, a);
    return true;
  }
};

}  // namespace

int ldpc_build_lds_schedule(const ldpc_code &c, int block, int RM, int ndummy, bool optimise,
                            LdsSchedule &out) {
  const int n = c.n, m = c.m;
  const int rounds = (m + block - 1) / block;
  const int G = (rounds * block + NB - 1) / NB;        // half-wave groups, check j is in group j / 32
  out.block = block; out.rounds = rounds; out.RM = RM;
  out.nw = (RM + 31) / 32;

  // ---- 1. banks
  std::vector<int> bank(n);
  const int capL = (n + NB - 1) / NB + 2;              // words per bank
  if (!optimise) {
    for (int v = 0; v < n; v++) bank[v] = v % NB;
  } else {
    // occurrences of each variable: (group, multiplicity)
    std::vector<std::vector<std::pair<int, int>>> occ(n);
    for (int j = 0; j < m; j++)
      for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++) {
        auto &o = occ[c.col_idx[e]];
        const int g = j / NB;
        bool found = false;
        for (auto &p : o) if (p.first == g) { p.second++; found = true; }
        if (!found) o.push_back({g, 1});
      }
    std::vector<int> cnt((size_t)G * NB, 0), load(NB, 0);
    auto delta_add = [&](int v, int b) {               // change of sum of squares when v joins bank b
      long d = 0;
      for (auto &p : occ[v]) { const long x = cnt[(size_t)p.first * NB + b]; d += (x + p.second) * (x + p.second) - x * x; }
      return d;
    };
    auto apply = [&](int v, int b, int sign) {
      for (auto &p : occ[v]) cnt[(size_t)p.first * NB + b] += sign * p.second;
      load[b] += sign;
    };
    std::mt19937 rng(12345);
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::shuffle(order.begin(), order.end(), rng);
    for (int v : order) {                               // greedy start
      long best = -1; int bb = 0;
      const int off = (int)(rng() % NB);
      for (int t = 0; t < NB; t++) {
        const int b = (t + off) % NB;
        if (load[b] >= capL) continue;
        const long d = delta_add(v, b);
        if (best < 0 || d < best) { best = d; bb = b; }
      }
      bank[v] = bb;
      apply(v, bb, +1);
    }
    for (int sweep = 0; sweep < 60; sweep++) {          // best-move descent
      long moved = 0;
      std::shuffle(order.begin(), order.end(), rng);
      for (int v : order) {
        const int b0 = bank[v];
        apply(v, b0, -1);
        long best = delta_add(v, b0); int bb = b0;
        for (int b = 0; b < NB; b++) {
          if (b == b0 || load[b] >= capL) continue;
          const long d = delta_add(v, b);
          if (d < best) { best = d; bb = b; }
        }
        bank[v] = bb;
        apply(v, bb, +1);
        if (bb != b0) moved++;
      }
      int worst = 0;
      for (int x : cnt) worst = std::max(worst, x);
      if (moved == 0 || (worst <= RM && sweep >= 2)) break;
    }
  }
  // positions: word = bank + 32 * rank within the bank
  std::vector<int> fill(NB, 0);
  out.pos.assign(n, 0);
  int maxfill = 0;
  for (int v = 0; v < n; v++) {
    out.pos[v] = (uint16_t)(bank[v] + NB * fill[bank[v]]);
    maxfill = std::max(maxfill, ++fill[bank[v]]);
  }
  out.var_words = maxfill * NB;
  out.dummy_base = out.var_words;
  out.ndummy = ndummy;                                 // multiple of 32: dummy d has bank d % 32
  out.words = out.var_words + ndummy;
  if (out.words > 65535) return -1;

  // ---- 2. slots
  out.colT.assign((size_t)rounds * RM * block, 0);
  out.maskT.assign((size_t)rounds * block * out.nw, 0);
  out.dcT.assign((size_t)rounds * block, 0);
  out.conflicts = 0;
  out.accesses = 0;
  std::vector<int> slot_of_edge;
  for (int g = 0; g < G; g++) {
    const int j0 = g * NB;
    std::vector<std::vector<int>> slots(NB, std::vector<int>(RM, -1));   // [lane][slot] -> variable
    if (optimise) {
      int maxdeg = RM;
      std::vector<int> deg(NB, 0);
      for (int i = 0; i < NB && j0 + i < m; i++)
        for (int e = c.row_ptr[j0 + i]; e < c.row_ptr[j0 + i + 1]; e++) maxdeg = std::max(maxdeg, ++deg[bank[c.col_idx[e]]]);
      EdgeCol ec(NB, NB, maxdeg);
      std::vector<int> var_of;
      for (int i = 0; i < NB && j0 + i < m; i++)
        for (int e = c.row_ptr[j0 + i]; e < c.row_ptr[j0 + i + 1]; e++) {
          ec.add(i, bank[c.col_idx[e]]);
          var_of.push_back(c.col_idx[e]);
        }
      std::vector<int> late;
      for (size_t e = 0; e < var_of.size(); e++) {
        if (ec.ec[e] >= 0 && ec.ec[e] < RM) slots[ec.el[e]][ec.ec[e]] = var_of[e];
        else late.push_back((int)e);
      }
      for (int e : late) {                             // bank over capacity: any free slot of the lane
        auto &s = slots[ec.el[e]];
        for (int k = 0; k < RM; k++) if (s[k] < 0) { s[k] = var_of[e]; break; }
      }
    } else {
      for (int i = 0; i < NB && j0 + i < m; i++) {
        int k = 0;
        for (int e = c.row_ptr[j0 + i]; e < c.row_ptr[j0 + i + 1]; e++) slots[i][k++] = c.col_idx[e];
      }
    }
    // pads -> dummies of banks unused in that slot; count what conflicts are left
    for (int k = 0; k < RM; k++) {
      int used[NB] = {0};
      for (int i = 0; i < NB; i++) if (slots[i][k] >= 0) used[bank[slots[i][k]]]++;
      int mx = 1;
      for (int b = 0; b < NB; b++) mx = std::max(mx, used[b]);
      int nb = 0;
      for (int i = 0; i < NB; i++) {
        const int j = j0 + i;
        const int r = j / block, t = j % block;
        if (r >= rounds) continue;
        int word;
        if (slots[i][k] >= 0) {
          word = out.pos[slots[i][k]];
          out.maskT[((size_t)r * block + t) * out.nw + k / 32] |= 1u << ((RM < 32 ? RM : 32) - 1 - (k % 32));
        } else {
          while (nb < NB && used[nb]) nb++;
          int b = nb < NB ? nb : i;                    // a free bank if one is left
          if (nb < NB) used[nb] = 1;
          word = out.dummy_base + b + NB * ((g + k) % (ndummy / NB));
        }
        out.colT[((size_t)r * RM + k) * block + t] = (uint16_t)word;
      }
      out.conflicts += mx - 1;
      out.accesses += 1;
    }
    for (int i = 0; i < NB && j0 + i < m; i++) {
      const int j = j0 + i;
      out.dcT[(size_t)(j / block) * block + j % block] = (uint8_t)std::min(255, c.row_ptr[j + 1] - c.row_ptr[j]);
    }
  }
  {
    // pad slots per dummy word: each adds a bounded amount to it in phase B, so the kernels can
    // only promise "huge and positive" dummies up to a load (FusedQ88Engine checks it)
    std::vector<int> load((size_t)ndummy, 0);
    for (uint16_t w : out.colT)
      if (w >= out.dummy_base) load[w - out.dummy_base]++;
    out.max_dummy_load = 0;
    for (int x : load) out.max_dummy_load = std::max(out.max_dummy_load, x);
  }
  if (optimise && out.conflicts > 0) {
    // structured (e.g. quasi-cyclic) codes can be conflict free just as they stand, and a slot
    // count with no slack leaves the colouring no room: keep whichever placement is better
    LdsSchedule nat;
    if (ldpc_build_lds_schedule(c, block, RM, ndummy, false, nat) == 0 && nat.conflicts < out.conflicts) out = nat;
  }
  return 0;
}
