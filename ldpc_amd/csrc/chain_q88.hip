// chain_q88.hip -- frame-resident Q8.8 min-sum decoder for accumulator-chain codes whose frame
// does not fit the LDS sums of fused_kernels.hip.h (DVB-S2 long frames, rate 1/2: 64 800
// variables, 32 400 checks).
//
// dec() of test_code_min_sum_B_fixed.c:192-316, flooding schedule, bit-exact -- the same
// algorithm and the same arithmetic as fused_q88_reg_kernel (sums as biased integers in LDS,
// a check's messages compressed to min1 / min2 / slot of min1 / one sign per slot, floats that
// hold integers in the slot loops).  What is new is where the state lives:
//
//   * IRA / DVB-S2 style codes have a "staircase": a weight-2 variable between check j and
//     check j+1 for (nearly) every j.  One thread owns R consecutive checks, so such a
//     variable sits between two checks of the same thread and never needs a sum in LDS: its
//     posterior is q0 + the two messages, both of which the thread holds in registers.  Only
//     the variable between two threads' runs is exchanged (two LDS words per thread).
//   * Only the other variables (DVB-S2 0.5: the 32 400 information bits and the last parity
//     bit) have sums in LDS: 130 KB of the CU's 160 KB.
//   * The compressed messages of all of a thread's checks, previous and current iteration,
//     stay in VGPRs, three registers per check (R = 32: 96 of the 128 registers a wave of a
//     1024-thread workgroup owns); the edge -> LDS word table (one 16-byte row per check) is
//     re-read from L2 in both phases; the channel values of the staircase variables are parked
//     in a per-workgroup global scratch row (int16, L2 resident) and re-read once per check.
//
// Slots of a check: 0 = staircase variable shared with the previous check, 1 = the one shared
// with the next check, 2..6 = LDS variables, 7 = an optional sixth LDS variable (walked only by
// wavefronts in which some check has one).  Missing staircase variables read as 2^24 ("huge,
// positive": neutral for sign and minimum), missing LDS slots point at dummy words as in
// fused_kernels.hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <type_traits>

#include "engine.h"
#include "fused_kernels.hip.h"

namespace ldpc {

struct ChainArgs {
  const fq_u4 *tab;          // [R][block] one row per check: 6 LDS words (u16), spare, real-slot mask | flags << 8
  const uint16_t *chainvar;  // [R + 1][block] staircase variable left of check 0, then right of check r (0xFFFF: none)
  const uint16_t *pos;       // [n] variable -> LDS word (0xFFFF: staircase variable)
  const uint32_t *wextra;    // [block / 64] bit r: some check r of the wavefront has slot 7
  short *q0s;                // [grid][R + 1][block] channel values of the staircase variables
  uint32_t *nms;             // [grid][R][block] min1 | min2 << 16 of the current iteration, between the two phases
  int n, m, nframes, max_iter;
  const void *llr;
  int dt;
  const uint8_t *synd;
  uint8_t *bits;
  int32_t *iters;
  uint8_t *sout;
  int32_t *status;
  float *llr_out;
  int *counter;
  int words, dummy_base;
};

#define CQ_BIGF 16777216.0f /* 2^24 */
#define CQ_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ float cq_pick(float a, float b, unsigned long long take_b) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(take_b));
  return r;
}
// lanes whose slot number (transposed into one lane mask per bit) equals k
__device__ __forceinline__ unsigned long long cq_in_slot(const unsigned long long ab[3], int k) {
  unsigned long long m = (k & 1) ? ab[0] : ~ab[0];
  m &= (k & 2) ? ab[1] : ~ab[1];
  m &= (k & 4) ? ab[2] : ~ab[2];
  return m;
}
__device__ __forceinline__ float cq_f(unsigned x) { return __uint_as_float(x); }
__device__ __forceinline__ unsigned cq_u(float x) { return __float_as_uint(x); }

// BLK: workgroup size when it is known at compile time (row offsets become constants), else 0
template <int R, int BLK>
__global__ __launch_bounds__(1024) void chain_q88_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) int fsum[];   // [words] sums + dummies, frame id + flags, exchange rows
  const int tid = threadIdx.x, B = BLK ? BLK : (int)blockDim.x;
  int *frame_slot = fsum + a.words;            // [4]
  float *xR = (float *)(frame_slot + 4);       // [B + 2]  xR[i + 1]: message of thread i's last check to its right variable
  float *xL = xR + B + 2;                      // [B + 2]  xL[i]: message of thread i's first check to its left variable
  const unsigned wex = __builtin_amdgcn_readfirstlane(a.wextra[tid >> 6]);
  short *q0s = a.q0s + (size_t)blockIdx.x * (R + 1) * B + tid;
  // (uniform bases for the loops below: an SGPR pair plus the thread's 32-bit offset per access)
  const short *q0base = a.q0s + (size_t)blockIdx.x * (R + 1) * B;
  uint32_t *nmbase = a.nms + (size_t)blockIdx.x * R * B;

  // compressed messages: om = min1 | min2 << 16 of the previous iteration (the current one's waits
  // in the global row nms between the phases: 96 registers of state did not leave the compiler
  // enough to work with); as = signs | slot of min1 << 8, previous iteration in the low half,
  // current in the high half (signs: slot k in bit 7 - k)
  unsigned om[R], as[R];

  if (tid == 0) *frame_slot = atomicAdd(a.counter, 1);
  __syncthreads();
  for (;;) {
    const int frame = *frame_slot;
    if (frame >= a.nframes) break;
    int next_frame = 0;
    if (tid == 0) next_frame = atomicAdd(a.counter, 1);
    const size_t base = (size_t)frame * a.n;
    auto q0_of = [&](int i) -> int {             // float_to_fix for real-valued inputs, as bsc() :117-120
      if (a.dt == 2) return ((const short *)a.llr)[base + i];
      if (a.dt == 3) {
        const int v = ((const int *)a.llr)[base + i];
        return v > FQ_SAT ? FQ_SAT : (v < -32768 ? -32768 : v);
      }
      if (a.dt == 1) return fq_from_real((double)((const float *)a.llr)[base + i]);
      return fq_from_real(((const double *)a.llr)[base + i]);
    };
    for (int i = tid; i < a.n; i += B) {
      const unsigned w = a.pos[i];
      if (w != 0xffffu) fsum[w] = FQ_BIAS + q0_of(i);
    }
    for (int i = a.dummy_base + tid; i < a.words; i += B) fsum[i] = FQ_BIG;
    for (int r = 0; r <= R; r++) {
      const unsigned v = a.chainvar[(size_t)r * B + tid];
      q0s[(size_t)r * B] = v != 0xffffu ? (short)q0_of((int)v) : (short)0;
    }
    unsigned sbits = 0, parbits = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      om[r] = as[r] = 0;
      const int c = tid * R + r;
      if (a.synd && c < a.m) sbits |= (unsigned)(a.synd[(size_t)frame * a.m + c] & 1) << r;
    }
    if (tid == 0) { frame_slot[1] = 0; frame_slot[2] = 0; }
    for (int i = tid; i < 2 * (B + 2); i += B) xR[i] = 0.0f;
    __syncthreads();

    // the check in turn with its messages to the two staircase variables
    struct Open { float m1, m2, cl, cr; unsigned sgt; unsigned long long ab[3]; };
    auto open = [&](const unsigned mm, const unsigned s, Open &o) {
      o.m1 = (float)(mm & 0xffffu);
      o.m2 = (float)(mm >> 16);
      o.sgt = s << 24;                           // sign of slot 0 in bit 31
#pragma unroll
      for (int b = 0; b < 3; b++) o.ab[b] = __ballot((s >> (8 + b)) & 1u);
      o.cl = cq_f(FQ_ANDOR(o.sgt, 0x80000000u, cq_u(cq_pick(o.m1, o.m2, cq_in_slot(o.ab, 0)))));
      o.cr = cq_f(FQ_ANDOR(o.sgt << 1, 0x80000000u, cq_u(cq_pick(o.m1, o.m2, cq_in_slot(o.ab, 1)))));
    };

    // Phase A: every check of the thread on the sums of the previous iteration.  PO: only the
    // parity of the hard decisions (the verdict after the last iteration).
    auto check_pass = [&](auto po_c) {
      constexpr bool PO = decltype(po_c)::value;
      // (opaque copies: the table rows and channel values are loop-invariant, and hoisted out of the
      // iteration loop they would take a hundred registers)
      const fq_u4 *tab_ = a.tab;
      const short *q0_ = q0base;
      uint32_t *nm_ = nmbase;
      asm volatile("" : "+s"(tab_), "+s"(q0_), "+s"(nm_));
      const CQ_GLOBAL fq_u4 *tabp = (const CQ_GLOBAL fq_u4 *)tab_ + tid;
      const CQ_GLOBAL short *q0p = (const CQ_GLOBAL short *)q0_ + tid;
      CQ_GLOBAL uint32_t *nmp = (CQ_GLOBAL uint32_t *)nm_ + tid;
      Open c, nx;
      open(om[0], as[0], c);
      float Sl = 0.0f;
      unsigned pacc = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const fq_u4 w = tabp[(size_t)r * B];
        const bool ex = (wex >> r) & 1u;          // (wave-uniform)
        int Sw[6];
        Sw[0] = fsum[w.x & 0xffffu]; Sw[1] = fsum[w.x >> 16];
        Sw[2] = fsum[w.y & 0xffffu]; Sw[3] = fsum[w.y >> 16];
        Sw[4] = fsum[w.z & 0xffffu];
        Sw[5] = 0;
        if (ex) Sw[5] = fsum[w.z >> 16];
        const float q0r = (float)q0p[(size_t)(r + 1) * B];
        float cln;
        if (r + 1 < R) { open(om[r + 1], as[r + 1], nx); cln = nx.cl; }
        else cln = xL[tid + 1];
        if (r == 0) {
          Sl = (float)q0p[0] + xR[tid] + c.cl;
          if ((w.w >> 24) & 1u) Sl = CQ_BIGF;
        }
        float Sr = q0r + c.cr + cln;
        if ((w.w >> 25) & 1u) Sr = CQ_BIGF;
        float Sf[6];
#pragma unroll
        for (int j = 0; j < 6; j++) Sf[j] = cq_f((unsigned)Sw[j]) - FQ_BIASF;
        unsigned pp = __builtin_amdgcn_bitop3_b32(cq_u(Sl), cq_u(Sr), cq_u(Sf[0]), 0x96);
        pp = __builtin_amdgcn_bitop3_b32(pp, cq_u(Sf[1]), cq_u(Sf[2]), 0x96);
        pp = __builtin_amdgcn_bitop3_b32(pp, cq_u(Sf[3]), cq_u(Sf[4]), 0x96);
        if (ex) pp ^= cq_u(Sf[5]);
        // parity of check r in bit r (after R steps; R < 32 is moved down afterwards)
        pacc = FQ_ANDOR(pp, 0x80000000u, pacc >> 1);
        if constexpr (!PO) {
          unsigned k1 = 0xffffffffu, k2 = 0xffffffffu, sg = c.sgt;
          auto slot = [&](const float q, const int k) {
            sg = __builtin_amdgcn_alignbit(sg, cq_u(q), 31);
            const unsigned key = FQ_ANDOR(cq_u(q), 0x7fffffffu, (unsigned)k);
            k2 = fq_med3u(k1, key, k2);
            k1 = min(k1, key);
          };
          slot(Sl - c.cl, 0);
          slot(Sr - c.cr, 1);
#pragma unroll
          for (int j = 0; j < 6; j++) {
            if (j == 5 && !ex) { sg <<= 1; break; }
            const float mo = cq_pick(c.m1, c.m2, cq_in_slot(c.ab, 2 + j));
            slot(Sf[j] - cq_f(FQ_ANDOR(sg, 0x80000000u, cq_u(mo))), 2 + j);
          }
          // sgnprod = s ^ XOR(signs); c2v_k < 0 iff sgnprod != sign_k (:278-283)
          const unsigned tot = (unsigned)__popc(sg) + (sbits >> r);
          const unsigned flip = (unsigned)(-(int)(tot & 1u));
          const unsigned allm = (w.w >> 16) & 0xffu;
          const unsigned nsg = __builtin_amdgcn_bitop3_b32(sg, allm, flip, 0x78);   // sg ^ (allm & flip)
          // min over saturated magnitudes = saturated min (:43-53, :269-277)
          // (non-negative floats order like their bit patterns: the clamp is an integer minimum)
          const float n1 = cq_f(min(k1 & ~7u, 0x46fe0000u /* 127.0 * 256 */));
          const float n2 = cq_f(min(k2 & ~7u, 0x46fe0000u));
          nmp[(size_t)r * B] = (unsigned)n1 | ((unsigned)n2 << 16);
          as[r] = (as[r] & 0xffffu) | ((nsg | ((k1 & 7u) << 8)) << 16);
          // materialise the new state here: left alone the compiler sinks this arithmetic of all R
          // checks below the loop and keeps every check's signs and table row alive until then
          asm volatile("" : "+v"(as[r]));
        }
        Sl = Sr;
        if (r + 1 < R) c = nx;
        asm volatile("" : "+v"(pacc), "+v"(Sl));
        __builtin_amdgcn_sched_barrier(0);
      }
      parbits = R < 32 ? pacc >> (32 - R) : pacc;
    };

    int it_done = a.max_iter, ok = 0;
    for (int t = 1; t <= a.max_iter + 1; t++) {
      if (t == a.max_iter + 1) check_pass(std::true_type{});
      else check_pass(std::false_type{});
      const int any_bad = fq_block_or((int)(parbits ^ sbits), frame_slot + 1, t);
      if (!any_bad) { it_done = t - 1; ok = 1; break; }       // H*bits == s after iteration t-1
      if (t == a.max_iter + 1) break;
      for (int i = a.dummy_base + tid; i < a.words; i += B) fsum[i] = FQ_BIG;   // re-arm (see fused_kernels.hip.h)
      // ---- phase B: the new messages replace the old ones in the LDS sums
      const fq_u4 *tab_ = a.tab;
      const uint32_t *nm_ = nmbase;
      asm volatile("" : "+s"(tab_), "+s"(nm_));
      const CQ_GLOBAL fq_u4 *tabp = (const CQ_GLOBAL fq_u4 *)tab_ + tid;
      const CQ_GLOBAL uint32_t *nmp = (const CQ_GLOBAL uint32_t *)nm_ + tid;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const fq_u4 w = tabp[(size_t)r * B];
        const bool ex = (wex >> r) & 1u;
        const unsigned s = as[r];
        const float o1 = (float)(om[r] & 0xffffu), o2 = (float)(om[r] >> 16);
        const unsigned nmr = nmp[(size_t)r * B];
        const float n1 = (float)(nmr & 0xffffu), n2 = (float)(nmr >> 16);
        unsigned long long abo[3], abn[3];
#pragma unroll
        for (int b = 0; b < 3; b++) { abo[b] = __ballot((s >> (8 + b)) & 1u); abn[b] = __ballot((s >> (24 + b)) & 1u); }
        unsigned orot = s << 26, nrot = (s >> 16) << 26;     // sign of slot 2 in bit 31
        const unsigned idx[6] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16, w.z & 0xffffu, w.z >> 16};
#pragma unroll
        for (int j = 0; j < 6; j++) {
          if (j == 5 && !ex) break;
          const float cn = cq_f(FQ_ANDOR(nrot, 0x80000000u, cq_u(cq_pick(n1, n2, cq_in_slot(abn, 2 + j)))));
          const float co = cq_f(FQ_ANDOR(orot, 0x80000000u, cq_u(cq_pick(o1, o2, cq_in_slot(abo, 2 + j)))));
          nrot = fq_shl1(nrot);
          orot = fq_shl1(orot);
          const int delta = (int)(cn - co);
          __hip_atomic_fetch_add(&fsum[idx[j]], delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (r == 0)
          xL[tid] = cq_f(FQ_ANDOR((s >> 16) << 24, 0x80000000u, cq_u(cq_pick(n1, n2, cq_in_slot(abn, 0)))));
        if (r == R - 1)
          xR[tid + 1] = cq_f(FQ_ANDOR((s >> 16) << 25, 0x80000000u, cq_u(cq_pick(n1, n2, cq_in_slot(abn, 1)))));
        om[r] = nmr;
        as[r] = s >> 16;
        asm volatile("" : "+v"(om[r]), "+v"(as[r]));
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }

    // ---- outputs of this frame
    if (a.bits || a.llr_out) {
      for (int i = tid; i < a.n; i += B) {
        const unsigned w = a.pos[i];
        if (w == 0xffffu) continue;
        int s = fsum[w] - FQ_BIAS;
        if (a.bits) a.bits[base + i] = s < 0;
        if (a.llr_out) {
          s = s > FQ_SAT ? FQ_SAT : (s < -32768 ? -32768 : s);      // float_to_fix(sum), :303
          a.llr_out[base + i] = (float)s * (1.0f / 256.0f);
        }
      }
      // staircase variables: posterior = q0 + the messages of the two checks (right variable of every check)
      Open c, nx;
      open(om[0], as[0], c);
#pragma unroll
      for (int r = 0; r < R; r++) {
        float cln;
        if (r + 1 < R) { open(om[r + 1], as[r + 1], nx); cln = nx.cl; }
        else cln = xL[tid + 1];
        const unsigned v = a.chainvar[(size_t)(r + 1) * B + tid];
        if (v != 0xffffu) {
          int s = (int)((float)q0s[(size_t)(r + 1) * B] + c.cr + cln);
          if (a.bits) a.bits[base + v] = s < 0;
          if (a.llr_out) {
            s = s > FQ_SAT ? FQ_SAT : (s < -32768 ? -32768 : s);
            a.llr_out[base + v] = (float)s * (1.0f / 256.0f);
          }
        }
        if (r + 1 < R) c = nx;
      }
    }
    if (a.sout) {
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int cc = tid * R + r;
        if (cc < a.m) a.sout[(size_t)frame * a.m + cc] = (uint8_t)((parbits >> r) & 1u);
      }
    }
    if (tid == 0) {
      if (a.iters) a.iters[frame] = it_done;
      if (a.status) a.status[frame] = ok ? 0 : 1;
    }
    __syncthreads();                           // everyone has read this frame's number, sums and exchange rows
    if (tid == 0) *frame_slot = next_frame;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (see fused_q88_reg_kernel)
    __syncthreads();
  }
}

}  // namespace ldpc

namespace {
using namespace ldpc;

// Host side: finds the staircase, lays the other variables out in LDS and builds the tables.
struct ChainQ88Engine : Engine {
  int n = 0, m = 0, R = 0, block = 0, num_cu = 256, words = 0, dummy_base = 0;
  size_t lds = 0;
  DevBuf d_tab, d_chainvar, d_pos, d_wextra, d_q0s, d_nms, d_counter;
  void (*kern)(ChainArgs) = nullptr;

  size_t llr_elems_per_frame() const override { return (size_t)n; }

  int init(const ldpc_code &c, const ldpc_opts &o) override {
    n = c.n; m = c.m;
    if (c.q != 2 || n > 65534 || c.min_row_weight < 2 || !fq_cmax_ok(c.cmax)) {
      ldpc_set_error("staircase Q8.8 decoder: code out of range");
      return LDPC_ERR_UNSUPPORTED;
    }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, o.device));
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (m <= 4 * 1024) { R = 4; kern = chain_q88_kernel<4, 0>; }
    else if (m > 31 * 1024 && m <= 32 * 1024) { R = 32; kern = chain_q88_kernel<32, 1024>; }
    else { ldpc_set_error("staircase Q8.8 decoder: too many checks"); return LDPC_ERR_UNSUPPORTED; }
    block = ((m + R - 1) / R + 63) / 64 * 64;
    // staircase: a weight-2 variable whose checks are j and j + 1 (the first such in row j)
    std::vector<int> right(m, -1);               // right[j]: variable between j and j + 1
    std::vector<char> is_chain(n, 0);
    for (int j = 0; j + 1 < m; j++)
      for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++) {
        const int v = c.col_idx[e];
        if (c.col_ptr[v + 1] - c.col_ptr[v] != 2 || is_chain[v]) continue;
        // the variable's two checks, from the CSC view (edge ids are check-major)
        const int e0 = c.edge_of[c.col_ptr[v]], e1 = c.edge_of[c.col_ptr[v] + 1];
        auto check_of = [&](int ed) { return (int)(std::upper_bound(c.row_ptr.begin(), c.row_ptr.end(), ed) - c.row_ptr.begin()) - 1; };
        const int ca = check_of(e0), cb = check_of(e1);
        if (std::min(ca, cb) == j && std::max(ca, cb) == j + 1) { right[j] = v; is_chain[v] = 1; break; }
      }
    std::vector<uint16_t> pos(n, 0xffff);
    int nl = 0;
    for (int v = 0; v < n; v++)
      if (!is_chain[v]) pos[v] = (uint16_t)nl++;
    dummy_base = nl;
    words = nl + FQ_NDUMMY;
    lds = (size_t)(words + 4 + 2 * (block + 2)) * 4;
    if (lds > 160 * 1024 || words > 65535) {
      ldpc_set_error("staircase Q8.8 decoder: the variables outside the staircase do not fit in LDS");
      return LDPC_ERR_UNSUPPORTED;
    }
    std::vector<uint16_t> tab((size_t)R * block * 8, 0), chainvar((size_t)(R + 1) * block, 0xffff);
    std::vector<uint32_t> wextra(block / 64, 0);
    long pads = 0;
    for (int t = 0; t < block; t++)
      for (int r = 0; r < R; r++) {
        const int j = t * R + r;
        uint16_t *row = &tab[((size_t)r * block + t) * 8];
        unsigned allm = 0, flags = 0;
        int k = 0;
        const int lv = (j >= 1 && j < m) ? right[j - 1] : -1, rv = j < m ? right[j] : -1;
        if (lv >= 0) allm |= 0x80; else flags |= 1;
        if (rv >= 0) allm |= 0x40; else flags |= 2;
        if (r == 0) chainvar[t] = lv >= 0 ? (uint16_t)lv : 0xffff;
        chainvar[(size_t)(r + 1) * block + t] = rv >= 0 ? (uint16_t)rv : 0xffff;
        if (j < m)
          for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++) {
            const int v = c.col_idx[e];
            if (v == lv || v == rv) continue;
            if (k >= 6) { ldpc_set_error("staircase Q8.8 decoder: more than six variables of a check outside the staircase"); return LDPC_ERR_UNSUPPORTED; }
            row[k] = pos[v];
            allm |= 0x20u >> k;
            k++;
          }
        if (k == 6) wextra[t / 64] |= 1u << r;
        for (; k < 6; k++) { row[k] = (uint16_t)(dummy_base + (t * 7 + r * 13 + k * 41) % FQ_NDUMMY); pads++; }
        row[6] = 0;
        row[7] = (uint16_t)(allm | (flags << 8));
      }
    // (a dummy word must stay "huge, positive" through the adds of one phase B: see FusedQ88Engine)
    if ((pads / FQ_NDUMMY + 1) * (long)FQ_CAP > 10000000L) {
      ldpc_set_error("staircase Q8.8 decoder: too many padded slots per dummy word");
      return LDPC_ERR_UNSUPPORTED;
    }
    int rc;
    if ((rc = d_tab.ensure(tab.size() * 2))) return rc;
    if ((rc = d_chainvar.ensure(chainvar.size() * 2))) return rc;
    if ((rc = d_pos.ensure(pos.size() * 2))) return rc;
    if ((rc = d_wextra.ensure(wextra.size() * 4))) return rc;
    if ((rc = d_counter.ensure(256))) return rc;
    HIPCHK(hipMemcpy(d_tab.p, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_chainvar.p, chainvar.data(), chainvar.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_pos.p, pos.data(), pos.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_wextra.p, wextra.data(), wextra.size() * 4, hipMemcpyHostToDevice));
    hipFuncAttributes fa;
    HIPCHK(hipFuncGetAttributes(&fa, (const void *)kern));
    if (lds + fa.sharedSizeBytes > 160 * 1024) {
      ldpc_set_error("staircase Q8.8 decoder: frame does not fit in LDS");
      return LDPC_ERR_UNSUPPORTED;
    }
    HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes));
    if (getenv("LDPC_HIP_VERBOSE")) {
      int nchain = 0;
      for (int v = 0; v < n; v++) nchain += is_chain[v];
      fprintf(stderr, "ldpc_hip: staircase Q8.8 decoder: %d threads x %d checks, %d staircase variables in registers, %d in LDS, %zu B LDS, %d VGPRs, %zu B spilled\n",
              block, R, nchain, nl, lds, fa.numRegs, (size_t)fa.localSizeBytes);
    }
    return LDPC_OK;
  }

  size_t q0s_bytes(long grid) const { return (size_t)grid * (R + 1) * block * sizeof(short); }

  size_t nms_bytes(long grid) const { return (size_t)grid * R * block * sizeof(uint32_t); }

  int launch(int *counter, short *q0s, uint32_t *nms, long grid, int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter,
             uint8_t *bits, int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out, hipStream_t st) {
    HIPCHK(hipMemsetAsync(counter, 0, 4, st));
    ChainArgs a;
    a.tab = d_tab.as<fq_u4>();
    a.chainvar = d_chainvar.as<uint16_t>();
    a.pos = d_pos.as<uint16_t>();
    a.wextra = d_wextra.as<uint32_t>();
    a.q0s = q0s;
    a.nms = nms;
    a.n = n; a.m = m; a.nframes = nframes; a.max_iter = max_iter;
    a.llr = llr; a.dt = dt; a.synd = sy;
    a.bits = bits; a.iters = iters_out; a.sout = sout; a.status = status; a.llr_out = llr_out;
    a.counter = counter;
    a.words = words; a.dummy_base = dummy_base;
    prof_begin(K_FUSED, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(block), lds, st, a);
    LAUNCHCHK("chain_q88_kernel");
    prof_end(st);
    last_iterations = max_iter;
    return LDPC_OK;
  }

  int decode(int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter, uint8_t *bits,
             int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out, hipStream_t st) override {
    if (nframes <= 0) return LDPC_OK;
    if (!llr || max_iter < 0 || max_iter > 65000 || dt < 0 || dt > 3) {
      ldpc_set_error("bad arguments to the staircase Q8.8 decoder");
      return LDPC_ERR_ARG;
    }
    const long grid = std::min<long>(num_cu, nframes);
    if (int rc = d_q0s.ensure(q0s_bytes(num_cu))) return rc;
    if (int rc = d_nms.ensure(nms_bytes(num_cu))) return rc;
    return launch(d_counter.as<int>(), d_q0s.as<short>(), d_nms.as<uint32_t>(), grid, nframes, llr, dt, sy, max_iter, bits, iters_out, sout,
                  status, llr_out, st);
  }

  // scratch: the frame queue head (256 B), the channel rows of the staircase variables, the rows of new minima
  int decode_ws(void *ws, size_t ws_bytes, int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter,
                uint8_t *bits, int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out,
                hipStream_t st) override {
    if (nframes <= 0) return LDPC_OK;
    if (!llr || max_iter < 0 || max_iter > 65000 || dt < 0 || dt > 3) {
      ldpc_set_error("bad arguments to the staircase Q8.8 decoder");
      return LDPC_ERR_ARG;
    }
    const long grid = std::min<long>(num_cu, nframes);
    const size_t q0b = (q0s_bytes(grid) + 255) / 256 * 256;
    if (!ws || ws_bytes < 256 + q0b + nms_bytes(grid)) { ldpc_set_error("scratch too small for the staircase Q8.8 decoder"); return LDPC_ERR_ARG; }
    return launch((int *)ws, (short *)((char *)ws + 256), (uint32_t *)((char *)ws + 256 + q0b), grid, nframes, llr, dt, sy, max_iter, bits, iters_out, sout,
                  status, llr_out, st);
  }
};

}  // namespace

Engine *ldpc_make_chain_q88_engine() { return new (std::nothrow) ChainQ88Engine; }
