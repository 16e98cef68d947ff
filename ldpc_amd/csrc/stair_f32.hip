// stair_f32.hip -- frame-resident binary32 min-sum decoder for accumulator-chain codes whose frame does
// not fit the LDS layouts of fused_float.hip (DVB-S2 long frames, rate 1/2: 64 800 variables, 32 400 checks).
//
// dec() of test_code_min_sum_B.c:137-208 in binary32 (the arithmetic of fused_float.hip's kernels: every
// floating-point operation is the reference's, on the same operands in the same order, so bits, iteration
// counts, syndromes and posterior sums equal those of a binary32 transcription of the reference, bit for bit: tests/).  The difficulty is the
// order: the reference adds a variable's messages in column-slot order (:146-148) and floating-point sums
// are not associative, so chain_q88.hip's integer trick (fold new - old into the LDS sums with atomics, in
// any order) is not available, and fused_float.hip's variable pass needs every check's record within reach
// of every thread (32 400 records: 380 KB).  Here:
//
//   * check j is lane j % B of round j / B (B = 1024 threads, 32 rounds).  A check's compressed messages
//     {min1, min2, signs | slot of min1} live in a per-workgroup global row (L2 / Infinity Cache), one
//     12-byte record per check, two sets that swap roles every iteration.
//   * staircase variables (weight 2, between check j and check j + 1) have no sum anywhere: their two
//     messages are in the records of checks j and j + 1, which the thread of either check reads (its own
//     record and both neighbours': the same cache lines); posterior = (q0 + c_j) + c_{j+1}, the column's
//     order.  Their channel values sit in a per-workgroup global row, written once per frame.
//     (Tried and taken out: one record set rewritten in place by its own thread, the neighbours' messages
//     by wavefront shifts and a small LDS array across wavefront borders -- 512 instead of 896 KB per
//     workgroup, so all rows stay in the Infinity Cache -- check pass 4.37 -> 4.83 ms per 2 048 frames,
//     rebuild unchanged: the check pass is bound by vector issue, not by where its rows come from.)
//   * the other variables (the information bits) have binary32 sums in LDS (130 KB), q0 of the thread's
//     share in registers.  After the check pass the sums are REBUILT in the reference's order: sum = q0,
//     then the rounds in turn, every check adding its new message to the sums of its variables (a plain
//     LDS read, add, write: one writer per variable and step).  Columns list their checks in rising order
//     (checked on the host), so "round by round" IS column-slot order, except for two checks of one
//     variable in the same round: the later one waits for a second (rarely third) step of that round.
//     DVB-S2 0.5: 9 527 of the 32 400 information bits have such a pair, 32 a triple.  In the steps after the first
//     a check has at most one slot in turn (the host spreads them: three extra steps per round on DVB-S2), so such a
//     step is one LDS read, one add, one write per lane and a barrier.
//
// Slots of a check: 0 = staircase variable shared with the previous check, 1 = the one shared with the next
// check (or a private weight-1 variable: DVB-S2's last parity bit), 2..7 = up to six LDS variables.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <type_traits>

#include "engine.h"
#include "fused_kernels.hip.h"
#include "lds_schedule.h"

namespace ldpc {
namespace {

#define SF_NDUMMY 64
#define SF_CAP 200.0f       /* min_msg starts at 200 (test_code_min_sum_B.c:169) */
#define SF_VPT 32           /* LDS variables per thread (q0 in registers) */
#ifndef SF_ZERO_SET
#define SF_ZERO_SET 0       /* 1: first-iteration records from a shared zero set instead of a branch (slower: see the check pass) */
#endif

struct StairArgs {
  const fq_u4 *tab;          // [J] per check: six 16-bit LDS word numbers, flags
  const fq_u4 *tab2;         // [J][2] per check, for the rebuild of the sums: six words of step 0, three of steps 1..3 (trash word where none), their slots
  const uint16_t *pos;       // [n] variable -> LDS word, or 0x8000 | check whose right-hand variable it is
  const uint16_t *rightvar;  // [J] the variable to the right of check j (0xffff: none)
  unsigned long long nsteps; // steps of round r in the ordered rebuild of the sums, minus one: two bits per round
  float *q0p;                // [grid][J] channel values of the right-hand variables
  const uint32_t *zrec;      // [J][3] zeros: the records "before the first iteration" (:141), shared by all workgroups
  uint32_t *rec;             // [grid][2][J][3] records {min1, min2, signs (slot k at bit 7 - k) | slot of min1 << 8}
  int n, m, J, rounds, words, dummy_base, var_words;
  int nframes, max_iter;
  const void *llr;
  int dt;                    // LDPC_DT_F64 = 0, LDPC_DT_F32 = 1
  const uint8_t *synd;
  uint8_t *bits;
  int32_t *iters;
  uint8_t *sout;
  int32_t *status;
  float *llr_out;
  int *counter;
  int chunk;
  int ablate;                // timing experiments (LDPC_HIP_STAIR_ABLATE; results wrong): 1 = no rebuild rounds, 2 = no check pass
};

// flags of a check (tab row, halfword 6): bits 0..7 = which slots hold a variable, slot k at bit 7 - k (the layout of
// a record's sign bits: v_alignbit shifts them in from the right), bit 8 = the right-hand variable is private (weight 1).
// Halfword 7: the step (0..2) of each LDS slot in the rebuild of the sums, two bits per slot, 3 = no variable.
enum { SF_LEFT = 0x80, SF_RIGHT = 0x40, SF_PRIV = 0x100 };

typedef __amdgpu_buffer_rsrc_t sf_rsrc;
typedef __attribute__((address_space(3))) float sf_lds_float;
__device__ __forceinline__ float ff_pick32(float a, float b, unsigned long long take_b) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(take_b));
  return r;
}
typedef unsigned sf_u3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ sf_rsrc sf_make_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000 /* raw dwords, gfx9 */);
}
__device__ __forceinline__ fq_u4 sf_ld128(sf_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(fq_u4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ sf_u3 sf_ld96(sf_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(sf_u3, __builtin_amdgcn_raw_buffer_load_b96(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float sf_ld32f(sf_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void sf_st96(sf_u3 v, sf_rsrc r, unsigned voff, unsigned soff) {
  typedef unsigned v3 __attribute__((__vector_size__(3 * sizeof(unsigned))));
  __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(v3, v), r, (int)voff, (int)soff, 0);
}

// A barrier that orders LDS traffic only.  __syncthreads() also waits for every global load in flight (vmcnt(0)):
// in the rebuild loop that is the prefetch of the round after next, and the wait put a full memory latency into
// every round (44 of the 84 us of an iteration).
__device__ __forceinline__ void sf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// message of a record to the variable in slot k: min2 where k is the slot of min1, the sign from bit 7 - k
__device__ __forceinline__ float sf_msg(sf_u3 rec, unsigned k) {
  const unsigned mag = ((rec.z >> 8) & 15u) == k ? rec.y : rec.x;
  return __uint_as_float(FQ_ANDOR(rec.z << (24u + k), 0x80000000u, mag));
}

template <int B>
__global__ __launch_bounds__(B) void stair_f32_kernel(StairArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sfs[];   // [words] sums | hard decisions of the right-hand variables | frame word, flags
  constexpr int NWV = B / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int J = a.J;
  // [words] sums | [64] trash words (a word per lane for slots not in turn; the host's tables name them) | hard decisions | frame word, flags
  unsigned long long *hardbits = (unsigned long long *)(sfs + a.words + 64);            // [rounds][NWV]   (words is even)
  int *frame_slot = (int *)(hardbits + a.rounds * NWV);
  float *q0p = a.q0p + (size_t)blockIdx.x * J;
  uint32_t *recbase = a.rec + (size_t)blockIdx.x * 2 * J * 3;
  const sf_rsrc rtab = sf_make_rsrc(a.tab, (unsigned)J * 16u);
  const sf_rsrc rtab2 = sf_make_rsrc(a.tab2, (unsigned)J * 32u);
  const sf_rsrc rq0 = sf_make_rsrc(q0p, (unsigned)J * 4u);
  const sf_rsrc rrec = sf_make_rsrc(recbase, (unsigned)J * 24u);       // both sets: set s starts at s * J * 12
#if SF_ZERO_SET
  const sf_rsrc rzero = sf_make_rsrc(a.zrec, (unsigned)J * 12u);
#endif

  for (int i = a.dummy_base + tid; i < a.words + 64; i += B) sfs[i] = 1e30f;     // pad slots: sign 0, never a minimum
  for (int i = tid; i < a.var_words; i += B) sfs[i] = 0.0f;                 // (words that hold no variable)
  int q_next = 0, q_lim = 0;
  auto take_frame = [&]() {
    if (q_next == q_lim) { q_next = atomicAdd(a.counter, a.chunk); q_lim = q_next + a.chunk; }
    return q_next++;
  };
  if (tid == 0) *frame_slot = take_frame();
  __syncthreads();
  for (;;) {
    const int frame = *frame_slot;
    if (frame >= a.nframes) break;
    int next_frame = 0;
    if (tid == 0) next_frame = take_frame();
    // ---- stage-in: LDS variables to their words, right-hand variables to the global row
    {
      const size_t base = (size_t)frame * a.n;
      if (a.dt == 1) {
        const float *p = (const float *)a.llr + base;
        for (int i = tid; i < a.n; i += B) {
          const unsigned w = a.pos[i];
          if (!(w & 0x8000u)) sfs[w] = __builtin_nontemporal_load(p + i);
        }
        for (int j = tid; j < J; j += B) {
          const unsigned rv = a.rightvar[j];
          q0p[j] = rv != 0xffffu ? p[rv] : 0.0f;
        }
      } else {
        const double *p = (const double *)a.llr + base;
        for (int i = tid; i < a.n; i += B) {
          const unsigned w = a.pos[i];
          if (!(w & 0x8000u)) sfs[w] = (float)__builtin_nontemporal_load(p + i);
        }
        for (int j = tid; j < J; j += B) {
          const unsigned rv = a.rightvar[j];
          q0p[j] = rv != 0xffffu ? (float)p[rv] : 0.0f;
        }
      }
    }
    // syndrome bits of the thread's checks, one per round
    unsigned sbits = 0;
    if (a.synd)
      for (int r = 0; r < a.rounds; r++) {
        const int j = r * B + tid;
        if (j < a.m) sbits |= (unsigned)(a.synd[(size_t)frame * a.m + j] & 1) << r;
      }
    if (tid == 0) { frame_slot[1] = 0; frame_slot[2] = 0; }   // flags of fq_block_or
    __syncthreads();
    // q0 of this thread's LDS variables
    float q0v[SF_VPT];
#pragma unroll
    for (int i = 0; i < SF_VPT; i++) {
      const int w = tid + B * i;
      q0v[i] = w < a.var_words ? sfs[w] : 0.0f;
    }

    int it_done = a.max_iter, ok = 0;
    for (int t = 1; t <= a.max_iter + 1; t++) {
      // ---- check pass on the sums of iteration t-1: records of iteration t-1 in set (t-1)&1, new ones to set t&1
      const unsigned oldset = (unsigned)((t - 1) & 1) * (unsigned)J * 12u, newset = (unsigned)(t & 1) * (unsigned)J * 12u;
      const bool first = t == 1, last = t == a.max_iter + 1;
#if SF_ZERO_SET
      const sf_rsrc rold = first ? rzero : rrec;
      const unsigned oldoff = first ? 0u : oldset;
#endif
      int bad = 0;
      for (int r = 0; r < ((a.ablate & 2) ? 0 : a.rounds); r++) {
        const int j = r * B + tid;
        const unsigned jo = (unsigned)j;
        const fq_u4 row = sf_ld128(rtab, jo * 16u, 0u);
        // Before the first iteration every message is +0.0 (:141).  Two ways to say so, both kept: a wave-uniform
        // branch around the record loads (the default), or -DSF_ZERO_SET=1: the loads always happen and read a
        // shared set of zero records through a selected descriptor.  Measured on DVB-S2, 2 048 frames, 10
        // iterations: branch 7.43 ms, zero set 7.83 ms (check pass 4.37 -> 4.64 ms).
#if SF_ZERO_SET
        const sf_u3 own = sf_ld96(rold, jo * 12u, oldoff);
        const sf_u3 lft = sf_ld96(rold, (jo ? jo - 1u : 0u) * 12u, oldoff);
        const sf_u3 rgt = sf_ld96(rold, (j + 1 < J ? jo + 1u : jo) * 12u, oldoff);
#else
        sf_u3 own = {0u, 0u, 0u}, lft = {0u, 0u, 0u}, rgt = {0u, 0u, 0u};
        if (!first) {
          own = sf_ld96(rrec, jo * 12u, oldset);
          lft = sf_ld96(rrec, (jo ? jo - 1u : 0u) * 12u, oldset);
          rgt = sf_ld96(rrec, (j + 1 < J ? jo + 1u : jo) * 12u, oldset);
        }
#endif
        const float q0R = sf_ld32f(rq0, jo * 4u, 0u);
        const float q0L = sf_ld32f(rq0, (jo ? jo - 1u : 0u) * 4u, 0u);
        const unsigned flags = row.w & 0xffffu;
        // the six LDS slots (pads point at dummy words: 1e30, sign 0, never a minimum)
        float S[8];
        S[2] = sfs[row.x & 0xffffu]; S[3] = sfs[row.x >> 16];
        S[4] = sfs[row.y & 0xffffu]; S[5] = sfs[row.y >> 16];
        S[6] = sfs[row.z & 0xffffu]; S[7] = sfs[row.z >> 16];
        // the slot of the check's previous minimum, one lane mask per bit (scalar registers): which lanes take min2 in
        // slot k is then an AND on the scalar unit (vector compares instead: 4.4 -> 4.6 ms per 2 048 frames)
        unsigned long long ob[3];
#pragma unroll
        for (int b = 0; b < 3; b++) ob[b] = __ballot((own.z >> (8 + b)) & 1u);
        auto own_msg = [&](int k) {
          unsigned long long in_k = (k & 1) ? ob[0] : ~ob[0];
          in_k &= (k & 2) ? ob[1] : ~ob[1];
          in_k &= (k & 4) ? ob[2] : ~ob[2];
          const float mo = ff_pick32(__uint_as_float(own.x), __uint_as_float(own.y), in_k);
          return __uint_as_float(FQ_ANDOR(own.z << (24 + k), 0x80000000u, __float_as_uint(mo)));
        };
        // the staircase variables: (q0 + the earlier check's message) + the later check's, the column's order (:146-148)
        float co[8];
        co[0] = own_msg(0); co[1] = own_msg(1);
        const float S0 = (q0L + sf_msg(lft, 1u)) + co[0];
        const float S1a = q0R + co[1], S1 = (flags & SF_PRIV) ? S1a : S1a + sf_msg(rgt, 0u);
        S[0] = (flags & SF_LEFT) ? S0 : 1e30f;
        S[1] = (flags & SF_RIGHT) ? S1 : 1e30f;
        unsigned pacc = 0u, sg = 0u, arg = 0u;
        float m1 = SF_CAP, m2 = SF_CAP;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          S[k] = S[k] + 0.0f;                                 // (-0.0 -> +0.0: the sign bit is then "sum < 0", :190)
          pacc ^= __float_as_uint(S[k]);
          if (k >= 2) co[k] = own_msg(k);
          const float q = S[k] - co[k];                       // :150
          sg = __builtin_amdgcn_alignbit(sg, __float_as_uint(q), 31);
          const float aq = __builtin_fabsf(q);
          m2 = __builtin_amdgcn_fmed3f(m1, aq, m2);           // :168-176
          const bool lt = aq < m1;
          arg = lt ? (unsigned)k : arg;
          m1 = lt ? aq : m1;
        }
        const unsigned par = pacc >> 31;                      // H*bits of this check (:200-203)
        const unsigned sb = (sbits >> r) & 1u;
        if (j < a.m) bad |= (int)(par ^ sb);
        if (!last) {
          // signs of the new messages: the product of the others' signs and the syndrome bit (:163-166, :178)
          const unsigned tot = (sb ^ (unsigned)__popc(sg)) & 1u;
          const unsigned nsg = (tot ? ~sg : sg) & (flags & 0xffu);
          sf_u3 nr;
          nr.x = __float_as_uint(m1); nr.y = __float_as_uint(m2); nr.z = nsg | (arg << 8);
          sf_st96(nr, rrec, jo * 12u, newset);
        }
        // hard decision of the right-hand variable (bit `lane` of the wavefront's word)
        const unsigned long long hb = __ballot((flags & SF_RIGHT) && S[1] < 0.0f);
        if (lane == 0) hardbits[r * NWV + wave] = hb;
        if (a.llr_out && (flags & SF_RIGHT)) a.llr_out[(size_t)frame * a.n + a.rightvar[j]] = S[1];
        if (a.sout && j < a.m) a.sout[(size_t)frame * a.m + j] = (uint8_t)par;
      }
      const int any_bad = fq_block_or(bad, frame_slot + 1, t);   // (one barrier)
      // the reference tests H*bits == s after each iteration (:200-203), not before the first
      if (t > 1 && !any_bad) { it_done = t - 1; ok = 1; break; }
      if (last) break;
      // ---- the sums, rebuilt in column-slot order: q0, then round by round, step by step
#pragma unroll
      for (int i = 0; i < SF_VPT; i++) {
        const int w = tid + B * i;
        if (w < a.var_words) sfs[w] = q0v[i];
      }
      __syncthreads();
      // (the barriers of the steps keep the wavefronts in lock step, so nothing hides a load's latency but distance:
      // a round's table rows and record are requested two rounds ahead)
      struct StepIn { fq_u4 ra, rb; sf_u3 nr; };
      auto fetch2 = [&](int r) {
        StepIn in;
        const unsigned jo = (unsigned)(r * B + tid);
        in.ra = sf_ld128(rtab2, jo * 32u, 0u);
        in.rb = sf_ld128(rtab2, jo * 32u, 16u);
        in.nr = sf_ld96(rrec, jo * 12u, newset);
        return in;
      };
      auto round_steps = [&](const StepIn &cur, int r) {
        const sf_u3 nr = cur.nr;
        const int ns = 1 + (int)((a.nsteps >> (2 * r)) & 3ull);
        // step 0: every slot whose check is the first of its variable in this round (the others, and empty slots,
        // point at the lane's trash word): six reads, six adds, six writes, no branches
        {
          unsigned ad[6];
          float v[6];
          ad[0] = (cur.ra.x & 0xffffu) * 4u; ad[1] = (cur.ra.x >> 16) * 4u; ad[2] = (cur.ra.y & 0xffffu) * 4u;
          ad[3] = (cur.ra.y >> 16) * 4u; ad[4] = (cur.ra.z & 0xffffu) * 4u; ad[5] = (cur.ra.z >> 16) * 4u;
#pragma unroll
          for (int k = 0; k < 6; k++) v[k] = *(sf_lds_float *)(unsigned long)ad[k];
#pragma unroll
          for (int k = 0; k < 6; k++) *(sf_lds_float *)(unsigned long)ad[k] = v[k] + sf_msg(nr, (unsigned)(k + 2));
          sf_lds_barrier();
        }
        // steps 1..3: a later check of a variable already served in this round; at most one slot per lane and step
        for (int s = 1; s < ns; s++) {
          const unsigned wsel = s == 1 ? (cur.ra.w & 0xffffu) : s == 2 ? (cur.ra.w >> 16) : (cur.rb.x & 0xffffu);
          const unsigned k = (cur.rb.x >> (16 + 3 * (s - 1))) & 7u;
          const unsigned ad = wsel * 4u;
          const float v = *(sf_lds_float *)(unsigned long)ad;
          *(sf_lds_float *)(unsigned long)ad = v + sf_msg(nr, k);
          sf_lds_barrier();
        }
      };
      // three buffers in rotation, no copies between them (a copy needs its data: the wait for the newest load)
      const int nr_rounds = (a.ablate & 1) ? 0 : a.rounds;
      auto clampr = [&](int r) { return r < a.rounds ? r : a.rounds - 1; };
      StepIn b0 = fetch2(0), b1 = fetch2(clampr(1)), b2;
      for (int r = 0; r < nr_rounds; r += 3) {
        b2 = fetch2(clampr(r + 2));
        round_steps(b0, r);
        if (r + 1 >= nr_rounds) break;
        b0 = fetch2(clampr(r + 3));
        round_steps(b1, r + 1);
        if (r + 2 >= nr_rounds) break;
        b1 = fetch2(clampr(r + 4));
        round_steps(b2, r + 2);
      }
    }

    // ---- outputs
    if (a.bits) {
      uint8_t *o = a.bits + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) {
        const unsigned w = a.pos[i];
        if (w & 0x8000u) { const unsigned j = w & 0x7fffu; o[i] = (uint8_t)((hardbits[j >> 6] >> (j & 63u)) & 1ull); }
        else o[i] = sfs[w] < 0.0f;
      }
    }
    if (a.llr_out) {
      float *o = a.llr_out + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) {
        const unsigned w = a.pos[i];
        if (!(w & 0x8000u)) o[i] = sfs[w];
      }
    }
    if (tid == 0) {
      if (a.iters) a.iters[frame] = it_done;
      if (a.status) a.status[frame] = ok ? 0 : 1;
    }
    __syncthreads();
    if (tid == 0) *frame_slot = next_frame;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

struct StairF32Engine : Engine {
  int n = 0, m = 0, block = 0, rounds = 0, J = 0, num_cu = 256, words = 0, dummy_base = 0, var_words = 0;
  size_t lds = 0;
  DevBuf d_tab, d_tab2, d_pos, d_rightvar, d_q0p, d_rec, d_zrec, d_counter;
  unsigned long long nsteps_packed = 0;
  void (*kern)(StairArgs) = nullptr;

  size_t llr_elems_per_frame() const override { return (size_t)n; }

  int init(const ldpc_code &c, const ldpc_opts &o) override {
    n = c.n; m = c.m;
    if (c.q != 2 || n > 65534 || m < 2 || m > 32768 || c.min_row_weight < 2) {
      ldpc_set_error("staircase binary32 decoder: code out of range");
      return LDPC_ERR_UNSUPPORTED;
    }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, o.device));
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    block = 1024;
    if (const char *e = getenv("LDPC_HIP_STAIR_BLOCK")) block = atoi(e) == 256 ? 256 : 1024;
    kern = block == 256 ? stair_f32_kernel<256> : stair_f32_kernel<1024>;
    rounds = (m + block - 1) / block;
    J = rounds * block;
    if (rounds > 32) { ldpc_set_error("staircase binary32 decoder: too many checks"); return LDPC_ERR_UNSUPPORTED; }
    auto check_of = [&](int ed) { return (int)(std::upper_bound(c.row_ptr.begin(), c.row_ptr.end(), ed) - c.row_ptr.begin()) - 1; };
    // staircase: a weight-2 variable whose column lists check j, then check j + 1
    std::vector<int> right(m, -1);
    std::vector<char> is_chain(n, 0), priv(m, 0);
    for (int j = 0; j + 1 < m; j++)
      for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++) {
        const int v = c.col_idx[e];
        if (c.col_ptr[v + 1] - c.col_ptr[v] != 2 || is_chain[v]) continue;
        if (check_of(c.edge_of[c.col_ptr[v]]) == j && check_of(c.edge_of[c.col_ptr[v] + 1]) == j + 1) { right[j] = v; is_chain[v] = 1; break; }
      }
    {
      int nstairs = 0;
      for (int j = 0; j < m; j++) nstairs += right[j] >= 0;
      if (2 * nstairs < m) {
        ldpc_set_error("staircase binary32 decoder: the code has no staircase");
        return LDPC_ERR_UNSUPPORTED;
      }
    }
    // a check without a staircase variable to its right may park one weight-1 variable there
    for (int j = 0; j < m; j++) {
      if (right[j] >= 0) continue;
      for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++) {
        const int v = c.col_idx[e];
        if (c.col_ptr[v + 1] - c.col_ptr[v] == 1) { right[j] = v; priv[j] = 1; is_chain[v] = 1; break; }
      }
    }
    // the other variables: LDS words by the bank schedule of the fused decoders (check j = lane j % B of round j / B);
    // their columns must list the checks in rising order, or "round by round" is not the column's order
    std::vector<int> rank(n, -1);
    int nl = 0;
    for (int v = 0; v < n; v++) {
      if (is_chain[v]) continue;
      rank[v] = nl++;
      int prev = -1;
      for (int p = c.col_ptr[v]; p < c.col_ptr[v + 1]; p++) {
        const int j = check_of(c.edge_of[p]);
        if (j <= prev) { ldpc_set_error("staircase binary32 decoder: a column lists its checks out of order"); return LDPC_ERR_UNSUPPORTED; }
        prev = j;
      }
    }
    ldpc_code vc;
    vc.n = nl; vc.m = m; vc.q = 2;
    vc.row_ptr.assign((size_t)m + 1, 0);
    int most = 0;
    for (int j = 0; j < m; j++) {
      for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++)
        if (rank[c.col_idx[e]] >= 0) vc.col_idx.push_back(rank[c.col_idx[e]]);
      vc.row_ptr[j + 1] = (int32_t)vc.col_idx.size();
      most = std::max(most, vc.row_ptr[j + 1] - vc.row_ptr[j]);
    }
    if (most > 6) { ldpc_set_error("staircase binary32 decoder: more than six variables of a check outside the staircase"); return LDPC_ERR_UNSUPPORTED; }
    if (nl < 1) { ldpc_set_error("staircase binary32 decoder: no variable outside the staircase"); return LDPC_ERR_UNSUPPORTED; }
    LdsSchedule sc;
    int optimise = 1;
    if (const char *e = getenv("LDPC_HIP_FUSED_PLACE")) optimise = atoi(e);
    if (ldpc_build_lds_schedule(vc, block, 6, SF_NDUMMY, optimise != 0, sc) || sc.rounds != rounds) {
      ldpc_set_error("staircase binary32 decoder: the variables outside the staircase do not fit in LDS");
      return LDPC_ERR_UNSUPPORTED;
    }
    words = sc.words; dummy_base = sc.dummy_base; var_words = sc.var_words;
    if (words & 1) words++;
    lds = (size_t)words * 4 + 64 * 4 + (size_t)rounds * (block / 64) * 8 + 16;
    if (lds > 160 * 1024 || words + 64 > 0xffff || var_words > 0x7fff || var_words > SF_VPT * block) {
      ldpc_set_error("staircase binary32 decoder: the variables outside the staircase do not fit in LDS");
      return LDPC_ERR_UNSUPPORTED;
    }
    std::vector<uint16_t> pos(n, 0);
    for (int v = 0; v < n; v++)
      if (rank[v] >= 0) pos[v] = sc.pos[rank[v]];
    std::vector<uint16_t> rightvar((size_t)J, 0xffff);
    for (int j = 0; j < m; j++)
      if (right[j] >= 0) { pos[right[j]] = (uint16_t)(0x8000 | j); rightvar[j] = (uint16_t)right[j]; }
    // slot of every edge to an LDS variable; for the rebuild, its step within the round: 0 for the first check of its
    // variable in that round, else a later step -- after the variable's previous check, and such that a check has at
    // most one slot in each of the steps 1..3
    std::vector<uint16_t> tab((size_t)J * 8, 0), tab2((size_t)J * 16, 0);
    std::vector<uint8_t> nsteps((size_t)rounds, 1);
    std::vector<int> last_round(nl, -1), last_step(nl, 0);
    for (int j = 0; j < J; j++) {
      const int r = j / block, t = j % block;
      uint16_t *row = &tab[(size_t)j * 8], *row2 = &tab2[(size_t)j * 16];
      const uint16_t trashw = (uint16_t)(words + (t & 63));
      unsigned flags = 0, ks = 0;
      if (j < m) {
        if (j >= 1 && right[j - 1] >= 0 && !priv[j - 1]) flags |= SF_LEFT;
        if (right[j] >= 0) flags |= SF_RIGHT;
        if (right[j] >= 0 && priv[j]) flags |= SF_PRIV;
      }
      for (int k = 0; k < 9; k++) row2[k] = trashw;
      int free_step = 1;
      for (int k = 0; k < 6; k++) {
        const int w = sc.colT[((size_t)r * 6 + k) * block + t];
        row[k] = (uint16_t)w;
        if (j >= m || w >= sc.dummy_base) continue;
        // which variable: the one of this row placed at word w
        int lv = -1;
        for (int e = vc.row_ptr[j]; e < vc.row_ptr[j + 1]; e++)
          if (sc.pos[vc.col_idx[e]] == w) lv = vc.col_idx[e];
        if (lv < 0) { ldpc_set_error("staircase binary32 decoder: schedule lost an edge"); return LDPC_ERR_FORMAT; }
        flags |= 0x80u >> (k + 2);
        if (last_round[lv] != r) { last_round[lv] = r; last_step[lv] = 0; row2[k] = (uint16_t)w; continue; }
        const int step = std::max(free_step, last_step[lv] + 1);
        if (step > 3) { ldpc_set_error("staircase binary32 decoder: too many checks of one variable in a round"); return LDPC_ERR_UNSUPPORTED; }
        free_step = step + 1;
        last_step[lv] = step;
        row2[5 + step] = (uint16_t)w;
        ks |= (unsigned)(k + 2) << (3 * (step - 1));
        nsteps[r] = (uint8_t)std::max<int>(nsteps[r], step + 1);
      }
      row[6] = (uint16_t)flags;
      row[7] = 0;
      row2[9] = (uint16_t)ks;
    }
    int rc;
    if ((rc = d_tab.ensure(tab.size() * 2))) return rc;
    if ((rc = d_tab2.ensure(tab2.size() * 2))) return rc;
    if ((rc = d_pos.ensure(pos.size() * 2))) return rc;
    if ((rc = d_rightvar.ensure(rightvar.size() * 2))) return rc;
    if ((rc = d_counter.ensure(256))) return rc;
    if ((rc = d_zrec.ensure((size_t)J * 12))) return rc;
    HIPCHK(hipMemset(d_zrec.p, 0, (size_t)J * 12));
    HIPCHK(hipMemcpy(d_tab.p, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_tab2.p, tab2.data(), tab2.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_pos.p, pos.data(), pos.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_rightvar.p, rightvar.data(), rightvar.size() * 2, hipMemcpyHostToDevice));
    nsteps_packed = 0;
    for (int r = 0; r < rounds; r++) nsteps_packed |= (unsigned long long)(nsteps[r] - 1) << (2 * r);
    hipFuncAttributes fa;
    HIPCHK(hipFuncGetAttributes(&fa, (const void *)kern));
    if (fa.sharedSizeBytes != 0 || lds > 160 * 1024) {   // (the kernel addresses its dynamic LDS from 0)
      ldpc_set_error("staircase binary32 decoder: frame does not fit in LDS");
      return LDPC_ERR_UNSUPPORTED;
    }
    HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes));
    if (getenv("LDPC_HIP_VERBOSE") || getenv("LDPC_HIP_DEBUG")) {
      int steps = 0;
      for (int r = 0; r < rounds; r++) steps += nsteps[r];
      fprintf(stderr, "ldpc_hip: staircase binary32 decoder: %d threads x %d rounds, %d variables in LDS (%d words), %d steps per rebuild, %zu B LDS, %d VGPRs, %zu B spilled\n",
              block, rounds, nl, words, steps, lds, fa.numRegs, (size_t)fa.localSizeBytes);
    }
    return LDPC_OK;
  }

  size_t q0p_bytes(long grid) const { return ((size_t)grid * J * 4 + 255) / 256 * 256; }
  size_t rec_bytes(long grid) const { return (size_t)grid * 2 * J * 12; }

  int launch(int *counter, float *q0p, uint32_t *rec, long grid, int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter,
             uint8_t *bits, int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out, hipStream_t st) {
    HIPCHK(hipMemsetAsync(counter, 0, 4, st));
    StairArgs a;
    a.tab = d_tab.as<fq_u4>();
    a.tab2 = d_tab2.as<fq_u4>();
    a.pos = d_pos.as<uint16_t>();
    a.rightvar = d_rightvar.as<uint16_t>();
    a.nsteps = nsteps_packed;
    a.q0p = q0p; a.rec = rec;
    a.zrec = d_zrec.as<uint32_t>();
    a.n = n; a.m = m; a.J = J; a.rounds = rounds; a.words = words; a.dummy_base = dummy_base; a.var_words = var_words;
    a.nframes = nframes; a.max_iter = max_iter;
    a.llr = llr; a.dt = dt; a.synd = sy;
    a.bits = bits; a.iters = iters_out; a.sout = sout; a.status = status; a.llr_out = llr_out;
    a.counter = counter;
    a.chunk = (int)std::max<long>(1, std::min<long>(nframes / (grid * 16), 16));
    a.ablate = 0;
    if (const char *e = getenv("LDPC_HIP_STAIR_ABLATE")) a.ablate = atoi(e);
    prof_cur_frames = nframes;
    prof_begin(K_FUSED, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(block), lds, st, a);
    LAUNCHCHK("stair_f32_kernel");
    prof_end(st);
    last_iterations = max_iter;
    return LDPC_OK;
  }

  bool args_ok(const void *llr, int dt, int max_iter, float *llr_out, const ldpc_opts &o) {
    if (!llr || max_iter < 0 || max_iter > 65000 || (dt != LDPC_DT_F64 && dt != LDPC_DT_F32)) {
      ldpc_set_error("floating-point min-sum takes F64 or F32 input, 0 <= max_iter <= 65000");
      return false;
    }
    if (llr_out && !o.want_llr) { ldpc_set_error("llr_out needs opts.want_llr"); return false; }
    return true;
  }

  ldpc_opts opts{};
  int decode(int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter, uint8_t *bits,
             int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out, hipStream_t st) override {
    if (nframes <= 0) return LDPC_OK;
    if (!args_ok(llr, dt, max_iter, llr_out, opts)) return LDPC_ERR_ARG;
    const long grid = std::min<long>(num_cu, nframes);
    if (int rc = d_q0p.ensure(q0p_bytes(num_cu))) return rc;
    if (int rc = d_rec.ensure(rec_bytes(num_cu))) return rc;
    return launch(d_counter.as<int>(), d_q0p.as<float>(), d_rec.as<uint32_t>(), grid, nframes, llr, dt, sy, max_iter, bits, iters_out, sout,
                  status, llr_out, st);
  }

  // scratch: the frame queue head (256 B), the channel row and the two record sets of every workgroup in flight
  int decode_ws(void *ws, size_t ws_bytes, int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter,
                uint8_t *bits, int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out,
                hipStream_t st) override {
    if (nframes <= 0) return LDPC_OK;
    if (!args_ok(llr, dt, max_iter, llr_out, opts)) return LDPC_ERR_ARG;
    long grid = std::min<long>(num_cu, nframes);
    if (ws && ws_bytes > 512) grid = std::min<long>(grid, (long)((ws_bytes - 512) / (q0p_bytes(1) + rec_bytes(1))));
    if (!ws || grid < 1 || ws_bytes < 256 + q0p_bytes(grid) + rec_bytes(grid)) {
      ldpc_set_error("scratch too small for the staircase binary32 decoder");
      return LDPC_ERR_ARG;
    }
    return launch((int *)ws, (float *)((char *)ws + 256), (uint32_t *)((char *)ws + 256 + q0p_bytes(grid)), grid, nframes, llr, dt, sy,
                  max_iter, bits, iters_out, sout, status, llr_out, st);
  }
};

}  // namespace
}  // namespace ldpc

Engine *ldpc_make_stair_f32_engine(const ldpc_opts &o) {
  auto *e = new (std::nothrow) ldpc::StairF32Engine;
  if (e) e->opts = o;
  return e;
}
