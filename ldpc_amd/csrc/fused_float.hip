// fused_float.hip -- whole-decode kernel of the floating-point min-sum decoder
// (dec() of test_code_min_sum_B.c:137-208, binary64 as the reference computes or its
// binary32 twin) with the frame resident in LDS.
//
// The Q8.8 fused kernel (fused_kernels.hip.h) folds message differences into the
// posterior sums with integer LDS atomics; floating-point sums are not associative, and
// the reference adds a variable's messages in column-slot order (:146-148), so here the
// sums are rebuilt every iteration, in that order, by a variable pass:
//
//   LDS   q0[n], sum[n]                channel LLR and posterior of every variable
//         per check: min1, min2, slot of min1, one sign bit per slot   (its c2v messages,
//         compressed: c2v of slot k is +-min1, or +-min2 in the slot of min1 -- the minimum
//         over "the others" of :168-176)
//   VGPRs the same compressed state of the thread's own checks (previous iteration)
//
// One workgroup decodes one frame, start to finish.  Per iteration:
//   check pass     each check thread walks its slots: S = sum[v], rebuilds its own previous
//                  message c_old, v2c = S - c_old (:150), sign parity with the syndrome bit
//                  (:163-166) and min1/min2 of |v2c| (cap 200, :169) -> new compressed
//                  state to LDS.  sign(S) over the slots is the syndrome of the hard
//                  decisions after the previous iteration (enc() + HamDist, :200-203).
//   barrier        (workgroup-wide OR of "some check unsatisfied": the exit test)
//   variable pass  each variable thread rebuilds sum = q0 + c2v_0 + c2v_1 + ... in slot
//                  order from the compressed check states (:146-148, :190-193).
//   barrier
// Every floating-point operation is the reference's, on the same operands in the same
// order, so decoded bits, iteration counts, syndromes and posteriors are bit-identical
// (binary64: to the reference; binary32: to its binary32 twin).  sum is stored as
// sum + 0.0 (turns -0.0 into +0.0, nothing else): then "sum < 0" is its sign bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <vector>

#include "engine.h"
#include "fused_kernels.hip.h"
#include "lds_schedule.h"

namespace ldpc {

struct FfArgs {
  const uint16_t *colT;    // [rounds][RM][block]  LDS word of (check = round*block + tid, slot)
  const uint32_t *realT;   // [rounds][block][NW]  bit k of word k/32: slot k holds an edge
  const uint32_t *vtab;    // [cmax][var_words]    (check << 8) | slot of the variable's kk-th edge; check == m: none
  const uint16_t *pos;     // [n] variable -> LDS word
  int n, m, RM, rounds, cmax;
  int var_words, words, dummy_base;
  int nframes, max_iter;
  const void *llr;         // [nframes][n] binary64 (dt 0) or binary32 (dt 1), frame-major
  int dt;
  const uint8_t *synd;     // [nframes][m] or null
  uint8_t *bits;           // [nframes][n] or null
  int32_t *iters;          // [nframes] or null
  uint8_t *sout;           // [nframes][m] or null
  int32_t *status;         // [nframes] or null
  float *llr_out;          // [nframes][n] or null
  int *counter;            // frame queue head (zeroed by the host before the launch)
  int chunk;               // frames a workgroup takes from the queue per atomic (>= 1), as in fused_kernels.hip.h
  // register-cached kernel only
  const uint16_t *vtabT;   // [VPT*CM][block]  (check << 5) | shift of its sign bit, per (thread, variable, edge)
  const uint32_t *maskT;   // [rounds][block]  real-slot bits in the kernel's mask layout (slot k at bit RM-1-k)
  const uint32_t *colP;    // binary64 register-cached kernel: packed slot addresses, [rounds][R2 / PK][block][PK]
  const uint32_t *vtabW;   // wide-row kernel: one entry per edge of a thread's variables, [VPT / 2][LPG][block][4]
  // QG kernels (frames whose q0 does not fit beside the sums): the channel LLRs in placed order,
  // one row of var_words values per workgroup, in global memory (L2 / Infinity Cache resident)
  void *q0g;
};

#define FF_NDUMMY 64
#define FF_CAP 200          /* min_msg starts at 200 (test_code_min_sum_B.c:169) */

template <int NW> struct FfSgw { static constexpr int value = NW == 1 ? 2 : 4; };

// Pieces of the generic kernel's check pass, written once for binary32 and binary64: the word that
// holds the sign, a select on a 64-lane mask held in scalar registers, and min / max against |x|
// as single instructions (the C library forms come with NaN-quieting instructions around them;
// every operand here is finite).
__device__ __forceinline__ uint32_t ff_hi(float x) { return __float_as_uint(x); }
__device__ __forceinline__ uint32_t ff_hi(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ float ff_xor_hi(float x, uint32_t m) { return __uint_as_float(__float_as_uint(x) ^ m); }
__device__ __forceinline__ double ff_xor_hi(double x, uint32_t m) {
  return __hiloint2double((int)((uint32_t)__double2hiint(x) ^ m), __double2loint(x));
}
__device__ __forceinline__ float ff_pick(float a, float b, unsigned long long take_b) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(take_b));
  return r;
}
__device__ __forceinline__ double ff_pick(double a, double b, unsigned long long take_b) {
  int lo, hi;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(a)), "v"(__double2loint(b)), "s"(take_b));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(a)), "v"(__double2hiint(b)), "s"(take_b));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ bool ff_abs_lt(float x, float m) { return __builtin_fabsf(x) < m; }
__device__ __forceinline__ bool ff_abs_lt(double x, double m) { return __builtin_fabs(x) < m; }
__device__ __forceinline__ float ff_min_abs(float a, float x) { float r; asm("v_min_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(x)); return r; }
__device__ __forceinline__ float ff_max_abs(float a, float x) { float r; asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(x)); return r; }
__device__ __forceinline__ float ff_min2(float a, float b) { float r; asm("v_min_f32_e64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double ff_min_abs(double a, double x) { double r; asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(x)); return r; }
__device__ __forceinline__ double ff_max_abs(double a, double x) { double r; asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(x)); return r; }
__device__ __forceinline__ double ff_min2(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// QG: the frame's channel LLRs stay in global memory (a.q0g, placed order, written once per frame
// and re-read by every variable pass) instead of LDS, and the check records are packed
// (sign words, then one byte per check for the slot of min1): binary64 sums of a 16 000-variable
// frame are 128 KB, which leaves room for the records but not for a second array of that size.
template <typename T, int CPT, int NW, bool QG = false>
__global__ __launch_bounds__(1024) void fused_ms_kernel(FfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ff_lds[];
  constexpr int SGW = QG ? NW : FfSgw<NW>::value;   // words of sign (/arg) state per check
  T *sumL = (T *)ff_lds;
  T *q0L = QG ? (T *)a.q0g + (size_t)blockIdx.x * a.var_words : sumL + a.words;
  T *magL = QG ? sumL + a.words : q0L + a.var_words;                 // [m + 1][2]  min1, min2
  uint32_t *sgL = (uint32_t *)(magL + 2 * (a.m + 1));   // [m + 1][SGW]  sign words (, slot of min1)
  uint8_t *argL = (uint8_t *)(sgL + SGW * (a.m + 1));   // QG: [m + 1] slot of min1
  int *frame_slot = QG ? (int *)(((size_t)(argL + a.m + 1) + 3) & ~(size_t)3) : (int *)(sgL + SGW * (a.m + 1));
  const int tid = threadIdx.x, B = blockDim.x;
  const T BIG = (T)1e30;

  bool has[CPT];
  uint32_t real[CPT][NW];
#pragma unroll
  for (int r = 0; r < CPT; r++) {
    has[r] = r < a.rounds && r * B + tid < a.m;
#pragma unroll
    for (int w = 0; w < NW; w++) real[r][w] = has[r] ? a.realT[((size_t)r * B + tid) * NW + w] : 0u;
  }
  // constants of the workgroup's life: pad slots read +BIG (sign 0, never a minimum), words
  // that hold no variable, and the "no edge" check state whose message is -0.0 (x + -0.0 = x)
  for (int i = a.dummy_base + tid; i < a.words; i += B) sumL[i] = BIG;
  for (int i = tid; i < a.var_words; i += B) { q0L[i] = (T)0; sumL[i] = (T)0; }
  if (tid == 0) {
    magL[2 * a.m] = (T)0; magL[2 * a.m + 1] = (T)0;
#pragma unroll
    for (int w = 0; w < NW; w++) sgL[SGW * a.m + w] = 1u;
    if (QG) argL[a.m] = 1; else sgL[SGW * a.m + NW] = 1u;
  }
  int q_next = 0, q_lim = 0;                    // (thread 0) the rest of the chunk of frames in hand
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
    if (tid == 0) next_frame = take_frame();              // its round trip hides under the decoding
    {
      const size_t base = (size_t)frame * a.n;
      if (a.dt == 1) {
        const float *p = (const float *)a.llr + base;
        for (int i = tid; i < a.n; i += B) { const T x = (T)p[i]; const int w = a.pos[i]; q0L[w] = x; sumL[w] = x + (T)0; }
      } else {
        const double *p = (const double *)a.llr + base;
        for (int i = tid; i < a.n; i += B) { const T x = (T)p[i]; const int w = a.pos[i]; q0L[w] = x; sumL[w] = x + (T)0; }
      }
    }
    // compressed messages of the thread's checks as of the previous iteration (all +0.0: :141)
    T om1[CPT], om2[CPT];
    uint32_t oarg[CPT], osg[CPT][NW];
    int sbit[CPT], par[CPT];
#pragma unroll
    for (int r = 0; r < CPT; r++) {
      om1[r] = om2[r] = (T)0;
      oarg[r] = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) osg[r][w] = 0;
      sbit[r] = (a.synd && has[r]) ? (a.synd[(size_t)frame * a.m + r * B + tid] & 1) : 0;
      par[r] = 0;
    }
    if (tid == 0) { frame_slot[1] = 0; frame_slot[2] = 0; }   // flags of fq_block_or
    __syncthreads();

    int it_done = a.max_iter, ok = 0;
    for (int t = 1; t <= a.max_iter + 1; t++) {
      // ---- check pass on the sums of iteration t-1
      int bad = 0;
#pragma unroll
      for (int r = 0; r < CPT; r++) {
        if (has[r]) {
          const uint16_t *col = a.colT + (size_t)r * a.RM * B + tid;
          T m1 = (T)FF_CAP, m2 = (T)FF_CAP;
          uint32_t arg = 0, p = 0, sg[NW];
#pragma unroll
          for (int w = 0; w < NW; w++) sg[w] = 0;
          // the low bits of the previous minimum's slot, transposed into lane masks (see the chunk loop)
          unsigned long long ob[3];
#pragma unroll
          for (int b = 0; b < 3; b++) ob[b] = __ballot((oarg[r] >> b) & 1u);
          // eight slots at a time (then the rest in pairs: RM is even): their indices (global, L2)
          // and then their sums (LDS) are all requested before the first is used -- one index
          // load per slot in a dependent chain was most of this kernel's time.
          auto slots = [&](int k0, auto ck_c) {
            constexpr int CK = decltype(ck_c)::value;
            int wv[CK];
            T Sv[CK];
#pragma unroll
            for (int jj = 0; jj < CK; jj++) wv[jj] = (int)col[(size_t)(k0 + jj) * B];
#pragma unroll
            for (int jj = 0; jj < CK; jj++) Sv[jj] = sumL[wv[jj]];
            // Branch-free, and with the per-slot questions answered once per chunk where they can be:
            // k0 is uniform and a multiple of CK, so "is the previous minimum in this chunk" is one
            // compare per chunk (the slot inside it: scalar ANDs of lane masks), the old signs of the
            // chunk are one shift, the new ones are collected in a register of their own and merged
            // into the mask word once.
            constexpr int LBITS = CK == 8 ? 3 : 1;
            const int w = NW == 1 ? 0 : (k0 >> 5);
            uint32_t osw = osg[r][0];
            if (NW > 1 && w) osw = osg[r][NW - 1];
            const uint32_t orot = osw >> (k0 & 31);                 // bit jj: old sign of slot k0 + jj
            const unsigned long long own = __ballot((oarg[r] >> LBITS) == (uint32_t)(k0 >> LBITS));
            uint32_t acc = 0;
#pragma unroll
            for (int jj = 0; jj < CK; jj++) {
              const T S = Sv[jj];
              p ^= ff_hi(S);                                        // (bit 31: S < 0; sums are stored as S + 0)
              unsigned long long in_k = own;
#pragma unroll
              for (int b = 0; b < LBITS; b++) in_k &= ((jj >> b) & 1) ? ob[b] : ~ob[b];
              const T mo = ff_pick(om1[r], om2[r], in_k);           // min2 in the slot of min1
              const T co = ff_xor_hi(mo, (orot << (31 - jj)) & 0x80000000u);
              const T q = S - co;                                   // :150
              acc = (acc >> 1) | (ff_hi(q) & 0x80000000u);
              const bool lt = ff_abs_lt(q, m1);
              m2 = ff_min2(m2, ff_max_abs(m1, q));
              arg = lt ? (uint32_t)(k0 + jj) : arg;
              m1 = ff_min_abs(m1, q);
            }
            const uint32_t fresh = (acc >> (32 - CK)) << (k0 & 31);  // new signs of the chunk, at their slots
            if (NW == 1 || !w) sg[0] |= fresh;
            else sg[NW - 1] |= fresh;
          };
          int k0 = 0;
          for (; k0 + 8 <= a.RM; k0 += 8) slots(k0, std::integral_constant<int, 8>{});
          for (; k0 < a.RM; k0 += 2) slots(k0, std::integral_constant<int, 2>{});
          p >>= 31;
          par[r] = (int)p;
          bad |= (int)p ^ sbit[r];
          // sgnprod = s ^ XOR(signs); c2v_k < 0 iff sgnprod != sign_k (:177-178)
          uint32_t tot = (uint32_t)sbit[r];
#pragma unroll
          for (int w = 0; w < NW; w++) tot ^= (uint32_t)__popc(sg[w]);
          const int j = r * B + tid;
          magL[2 * j] = m1; magL[2 * j + 1] = m2;
#pragma unroll
          for (int w = 0; w < NW; w++) {
            osg[r][w] = (tot & 1u) ? (sg[w] ^ real[r][w]) : sg[w];
            sgL[SGW * j + w] = osg[r][w];
          }
          if (QG) argL[j] = (uint8_t)arg; else sgL[SGW * j + NW] = arg;
          om1[r] = m1; om2[r] = m2; oarg[r] = arg;
        }
      }
      const int any_bad = fq_block_or(bad, frame_slot + 1, t);   // (one barrier; __syncthreads_or is three)
      // the reference tests H*bits == s after each iteration (:200-203), not before the first
      if (t > 1 && !any_bad) { it_done = t - 1; ok = 1; break; }
      if (t == a.max_iter + 1) break;
      // ---- variable pass: sum = q0 + c2v in column-slot order (:146-148)
      // up to four variables and four of their edges at a time, every edge-table entry (global,
      // L2) requested before the first record is read; each variable's messages are still added
      // in slot order.
      auto variables = [&](int w0, auto vu_c) {
        constexpr int VU = decltype(vu_c)::value;
        T sum[VU];
#pragma unroll
        for (int u = 0; u < VU; u++) sum[u] = q0L[w0 + u * B];
        auto edges = [&](int kk0, auto kc_c) {
          constexpr int KC = decltype(kc_c)::value;
          uint32_t ev[VU][KC];
#pragma unroll
          for (int u = 0; u < VU; u++)
#pragma unroll
            for (int c = 0; c < KC; c++) ev[u][c] = a.vtab[(size_t)(kk0 + c) * a.var_words + w0 + u * B];
#pragma unroll
          for (int c = 0; c < KC; c++)
#pragma unroll
            for (int u = 0; u < VU; u++) {
              // (one magnitude is read, after the slot of min1 is known: reading both and choosing in
              // registers doubles the LDS traffic of this pass and measured 12.9 against 17.8 Gbit/s)
              const uint32_t e = ev[u][c];
              const int j = (int)(e >> 8), p = (int)(e & 0xffu);
              const uint32_t sw = sgL[SGW * j + (NW == 1 ? 0 : (p >> 5))];
              const uint32_t arg = QG ? (uint32_t)argL[j] : sgL[SGW * j + NW];
              const T mg = magL[2 * j + (arg == (uint32_t)p ? 1 : 0)];
              sum[u] = sum[u] + ff_xor_hi(mg, (sw << (31 - (p & 31))) & 0x80000000u);
            }
        };
        int kk0 = 0;
        for (; kk0 + 4 <= a.cmax; kk0 += 4) edges(kk0, std::integral_constant<int, 4>{});
        if (a.cmax - kk0 == 3) edges(kk0, std::integral_constant<int, 3>{});
        else if (a.cmax - kk0 == 2) edges(kk0, std::integral_constant<int, 2>{});
        else if (a.cmax - kk0 == 1) edges(kk0, std::integral_constant<int, 1>{});
#pragma unroll
        for (int u = 0; u < VU; u++) sumL[w0 + u * B] = sum[u] + (T)0;
      };
      {
        int w0 = tid;
        for (; w0 + 3 * B < a.var_words; w0 += 4 * B) variables(w0, std::integral_constant<int, 4>{});
        for (; w0 < a.var_words; w0 += B) variables(w0, std::integral_constant<int, 1>{});
      }
      __syncthreads();
    }

    if (a.bits) {
      uint8_t *o = a.bits + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = sumL[a.pos[i]] < (T)0;     // :194-195
    }
    if (a.llr_out) {
      float *o = a.llr_out + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = (float)sumL[a.pos[i]];
    }
    if (a.sout) {
#pragma unroll
      for (int r = 0; r < CPT; r++)
        if (has[r]) a.sout[(size_t)frame * a.m + r * B + tid] = (uint8_t)par[r];
    }
    if (tid == 0) {
      if (a.iters) a.iters[frame] = it_done;
      if (a.status) a.status[frame] = ok ? 0 : 1;
    }
    __syncthreads();                           // everyone has read this frame's number and its sums
    if (tid == 0) *frame_slot = next_frame;
    // publication barrier in line after the write (see fused_kernels.hip.h: at the loop head the
    // compiler dropped the wait between this ds_write and the barrier across the back edge)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------
// Register-cached binary32 variant: the same algorithm with every index the workgroup needs
// held in VGPRs for its (persistent) life -- the LDS addresses of each check's slots (two
// 16-bit absolute byte addresses per register, as in fused_q88_reg_kernel) and, for the
// variable pass, (check, sign-bit shift) of each edge of the thread's VPT variables (two
// 16-bit entries per register) -- with all loops unrolled: an iteration touches no global
// memory.  q0 of the thread's variables lives in registers as well.
//   LDS   sum[words] binary32 (placed by the schedule compiler), then per check one 16-byte
//         record {min1, min2, sign mask, slot of min1} read by the variable pass with one
//         ds_read_b128 per edge.
//   CPT checks per thread, R2 slot pairs per check (RM = 2*R2 <= 32: one sign word),
//   VPT variables per thread, CM edges per variable (pads point at the "no edge" record).
// Sign masks keep slot k at bit RM-1-k (shifted in with v_alignbit); an edge's entry holds
// sh = 32-RM+k, so mask << sh puts its sign at bit 31, and the record stores the slot of min1
// as 32-RM+k too.
typedef unsigned ff_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ff_u4 lds_u4;
typedef __attribute__((address_space(3))) float lds_float;

template <int HI>
__device__ __forceinline__ unsigned ff_rec_addr(unsigned stbase, unsigned packed) {
  unsigned r;   // (entry >> 5) * 16 + stbase; volatile: not hoisted out of the iteration loop
  if (HI) asm volatile("v_bfe_u32 %0, %1, 21, 11" : "=v"(r) : "v"(packed));
  else asm volatile("v_bfe_u32 %0, %1, 5, 11" : "=v"(r) : "v"(packed));
  return (r << 4) + stbase;
}
template <int HI>
__device__ __forceinline__ unsigned ff_rec_shift(unsigned packed) {
  unsigned r;
  if (HI) asm volatile("v_bfe_u32 %0, %1, 16, 5" : "=v"(r) : "v"(packed));
  else asm volatile("v_and_b32 %0, 31, %1" : "=v"(r) : "v"(packed));
  return r;
}

template <int CPT, int R2, int VPT, int CM, int LB, int MINW>
__global__ __launch_bounds__(LB, MINW) void fused_ms_reg_kernel(FfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ffsum[];   // [words] sums, records, frame id
  constexpr int RM = 2 * R2;
  static_assert(RM <= 32, "one sign word per check");
  constexpr int CH = RM < 4 ? RM : 4;          // LDS reads in flight per check
  constexpr int NE = VPT * CM, NE2 = (NE + 1) / 2;
  constexpr int VCH = CM <= 4 ? 2 : 1;         // variables whose records are in flight together
  constexpr int B = LB;                        // (the host launches exactly LB threads: variable w = tid + B*i
  const int tid = threadIdx.x;                 //  is then an immediate offset, not an address register per variable)
  const unsigned lbase = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)ffsum;
  const unsigned stbase = lbase + (unsigned)a.words * 4u;
  ff_u4 *rec = (ff_u4 *)(ffsum + a.words);     // [m + 1]
  float *q0L = (float *)(rec + a.m + 1);       // [var_words] channel LLRs, by LDS word like the sums
  int *frame_slot = (int *)(q0L + a.var_words);

  bool has[CPT];
  unsigned idx[CPT][R2], allm[CPT];
  unsigned vidx[NE2];
#pragma unroll
  for (int r = 0; r < CPT; r++) {
    has[r] = r < a.rounds && r * B + tid < a.m;
#pragma unroll
    for (int k2 = 0; k2 < R2; k2++) {
      unsigned lo = (unsigned)a.dummy_base, hi = (unsigned)a.dummy_base;
      if (r < a.rounds) {
        const uint16_t *col = a.colT + (size_t)r * RM * B + tid;
        lo = col[(size_t)(2 * k2) * B];
        hi = col[(size_t)(2 * k2 + 1) * B];
      }
      lo = lo * 4u + lbase; hi = hi * 4u + lbase;    // absolute addresses (host: they fit 16 bits)
      idx[r][k2] = lo | (hi << 16);
      asm volatile("" : "+v"(idx[r][k2]));   // packed here, once (not re-packed per frame from 2*R2 halves)
    }
    allm[r] = r < a.rounds ? a.maskT[(size_t)r * B + tid] : 0u;
  }
#pragma unroll
  for (int e2 = 0; e2 < NE2; e2++) {
    const unsigned lo = a.vtabT[(size_t)(2 * e2) * B + tid];
    const unsigned hi = 2 * e2 + 1 < NE ? a.vtabT[(size_t)(2 * e2 + 1) * B + tid] : ((unsigned)a.m << 5);
    vidx[e2] = lo | (hi << 16);
    asm volatile("" : "+v"(vidx[e2]));
  }
  const bool wave_full = __ballot(has[CPT - 1]) != 0;   // (wave-uniform) any check in the last round?

  const bool vec_out = a.bits && (a.n & 7) == 0 && ((size_t)a.bits & 7) == 0;
  const bool vec_in = (a.n & 3) == 0 && ((size_t)a.llr & 15) == 0;
  const fq_u4 *pos8 = (const fq_u4 *)a.pos;

  for (int i = a.dummy_base + tid; i < a.words; i += B) ffsum[i] = 1e30f;     // pad slots: sign 0, never a minimum
  for (int i = tid; i < a.var_words; i += B) { ffsum[i] = 0.0f; q0L[i] = 0.0f; }   // (words that hold no variable)
  if (tid == 0) {
    ff_u4 none; none.x = 0u; none.y = 0u; none.z = 0xffffffffu; none.w = 0xffffffffu;   // message -0.0
    rec[a.m] = none;
  }
  int q_next = 0, q_lim = 0;                    // (thread 0) the rest of the chunk of frames in hand
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
    {
      const size_t base = (size_t)frame * a.n;
      if (a.dt == 1 && vec_in) {
        // four variables per thread and step: one 16-byte load of LLRs, one 8-byte load of positions
        typedef float ff_f4 __attribute__((ext_vector_type(4)));
        const ff_f4 *p4 = (const ff_f4 *)((const float *)a.llr + base);
        const fq_u2 *pos4 = (const fq_u2 *)a.pos;
        for (int i = tid; i < (a.n >> 2); i += B) {
          const ff_f4 v = __builtin_nontemporal_load(p4 + i);
          const fq_u2 w = pos4[i];
          const float x0 = v.x + 0.0f, x1 = v.y + 0.0f, x2 = v.z + 0.0f, x3 = v.w + 0.0f;
          ffsum[w.x & 0xffffu] = x0; q0L[w.x & 0xffffu] = x0;
          ffsum[w.x >> 16] = x1; q0L[w.x >> 16] = x1;
          ffsum[w.y & 0xffffu] = x2; q0L[w.y & 0xffffu] = x2;
          ffsum[w.y >> 16] = x3; q0L[w.y >> 16] = x3;
        }
      } else if (a.dt == 1) {
        const float *p = (const float *)a.llr + base;
        for (int i = tid; i < a.n; i += B) { const float x = __builtin_nontemporal_load(p + i) + 0.0f; const int w = a.pos[i]; ffsum[w] = x; q0L[w] = x; }
      } else {
        const double *p = (const double *)a.llr + base;
        for (int i = tid; i < a.n; i += B) { const float x = (float)__builtin_nontemporal_load(p + i) + 0.0f; const int w = a.pos[i]; ffsum[w] = x; q0L[w] = x; }
      }
    }
    float om1[CPT], om2[CPT];
    unsigned oarg[CPT], osg[CPT];
    int sbit[CPT], par[CPT];
#pragma unroll
    for (int r = 0; r < CPT; r++) {
      om1[r] = om2[r] = 0.0f;
      oarg[r] = 0u; osg[r] = 0u;
      sbit[r] = (a.synd && has[r]) ? (a.synd[(size_t)frame * a.m + r * B + tid] & 1) : 0;
      par[r] = 0;
    }
    if (tid == 0) { frame_slot[1] = 0; frame_slot[2] = 0; }   // flags of fq_block_or
    __syncthreads();

    int it_done = a.max_iter, ok = 0;
    for (int t = 1; t <= a.max_iter + 1; t++) {
      int bad = 0;
      auto check_pass = [&](auto na_c) {
        constexpr int NA = decltype(na_c)::value;
        float m1[CPT], m2[CPT];
        unsigned arg[CPT], pp[CPT], sg[CPT];
        // The slot of the previous minimum, transposed: one lane mask per bit, in scalar registers.
        // Which lanes read min2 in slot k is then an AND on the scalar unit, not a vector compare
        // per slot.  (The running minimum's slot stays in a vector register: tracking it the same
        // way costs five scalar operations per slot, and the CU's one scalar unit issues no faster
        // than its vector units -- measured 43.9 against 47.0 Gbit/s.)
        unsigned long long ob[CPT][5];
#pragma unroll
        for (int r = 0; r < NA; r++) {
          m1[r] = m2[r] = (float)FF_CAP;
          pp[r] = 0u; arg[r] = 0u;
          // one register serves both sign masks: the old signs leave at the top (the slot in turn is
          // always bit 31) while the new ones enter at the bottom, one v_alignbit per slot
          sg[r] = osg[r] << (32 - RM);
#pragma unroll
          for (int b = 0; b < 5; b++) ob[r][b] = __ballot((oarg[r] >> b) & 1u);
        }
#pragma unroll
        for (int kc = 0; kc < RM; kc += CH) {
          int S[CPT][CH];
#pragma unroll
          for (int r = 0; r < NA; r++)
#pragma unroll
            for (int j = 0; j < CH; j++) {
              const int k = kc + j;
              if (k >= RM) continue;
              S[r][j] = (k & 1) ? *fq_slot_ptr<1, 0>(lbase, idx[r][k >> 1]) : *fq_slot_ptr<0, 0>(lbase, idx[r][k >> 1]);
            }
#pragma unroll
          for (int j = 0; j < CH; j++)
#pragma unroll
            for (int r = 0; r < NA; r++) {
              const int k = kc + j;
              if (k >= RM) continue;
              // parity of the hard decisions (sums are stored as sum + 0.0: sign bit <=> sum < 0)
              if ((j & 1) == 0) pp[r] = __builtin_amdgcn_bitop3_b32(pp[r], (unsigned)S[r][j], (unsigned)S[r][j + 1], 0x96);
              // old magnitude of this slot: min2 in the slot of min1 (its one-hot mask rotates with the signs)
              unsigned long long in_k = (k & 1) ? ob[r][0] : ~ob[r][0];
#pragma unroll
              for (int b = 1; b < 5; b++) in_k &= ((k >> b) & 1) ? ob[r][b] : ~ob[r][b];
              float mo;
              asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(mo) : "v"(om1[r]), "v"(om2[r]), "s"(in_k));
              const float q = __int_as_float(S[r][j]) - __uint_as_float(FQ_ANDOR(sg[r], 0x80000000u, __float_as_uint(mo)));
              sg[r] = __builtin_amdgcn_alignbit(sg[r], __float_as_uint(q), 31);
              const float aq = __builtin_fabsf(q);
              m2[r] = __builtin_amdgcn_fmed3f(m1[r], aq, m2[r]);
              const bool lt = aq < m1[r];
              arg[r] = lt ? (unsigned)k : arg[r];
              m1[r] = lt ? aq : m1[r];
            }
          // the running values are materialised here: pure arithmetic is not ordered against
          // sched_barrier when the block is first linearised, and a check's whole chain can end
          // up after the last chunk with every sum it read still live
#pragma unroll
          for (int r = 0; r < NA; r++)
            asm volatile("" : "+v"(pp[r]), "+v"(m1[r]), "+v"(m2[r]), "+v"(arg[r]), "+v"(sg[r]));
          __builtin_amdgcn_sched_barrier(0);   // keeps a chunk's work in the chunk (live registers)
        }
#pragma unroll
        for (int r = 0; r < NA; r++) {
          const unsigned p = pp[r] >> 31;
          par[r] = (int)p;
          bad |= has[r] ? ((int)p ^ sbit[r]) : 0;
          const unsigned tot = (unsigned)sbit[r] ^ (unsigned)__popc(sg[r]);
          osg[r] = (tot & 1u) ? (sg[r] ^ allm[r]) : sg[r];
          om1[r] = m1[r]; om2[r] = m2[r]; oarg[r] = arg[r];
          // materialise the compressed state now (see fused_q88_reg_kernel)
          asm volatile("" : "+v"(om1[r]), "+v"(om2[r]), "+v"(oarg[r]), "+v"(osg[r]));
          if (has[r]) {
            ff_u4 st;
            st.x = __float_as_uint(om1[r]); st.y = __float_as_uint(om2[r]);
            st.z = osg[r]; st.w = oarg[r] + (unsigned)(32 - RM);
            rec[r * B + tid] = st;
          }
        }
      };
      // after the last iteration only the verdict is wanted: the parity of the hard decisions
      auto parity_only = [&](auto na_c) {
        constexpr int NA = decltype(na_c)::value;
#pragma unroll
        for (int r = 0; r < NA; r++) {
          unsigned pp = 0;
#pragma unroll
          for (int k = 0; k < RM; k += 2) {
            const int s0 = *fq_slot_ptr<0, 0>(lbase, idx[r][k >> 1]);
            const int s1 = *fq_slot_ptr<1, 0>(lbase, idx[r][k >> 1]);
            pp = __builtin_amdgcn_bitop3_b32(pp, (unsigned)s0, (unsigned)s1, 0x96);
          }
          par[r] = (int)(pp >> 31);
          bad |= has[r] ? (par[r] ^ sbit[r]) : 0;
        }
      };
      if (t == a.max_iter + 1) {
        if (wave_full) parity_only(std::integral_constant<int, CPT>{});
        else parity_only(std::integral_constant<int, CPT - 1>{});
      } else if (wave_full) check_pass(std::integral_constant<int, CPT>{});
      else check_pass(std::integral_constant<int, CPT - 1>{});
      const int any_bad = fq_block_or(bad, frame_slot + 1, t);   // (one barrier; __syncthreads_or is three)
      if (t > 1 && !any_bad) { it_done = t - 1; ok = 1; break; }
      if (t == a.max_iter + 1) break;
      // ---- variable pass
#pragma unroll
      for (int i0 = 0; i0 < VPT; i0 += VCH) {
        ff_u4 st[VCH][CM];
#pragma unroll
        for (int u = 0; u < VCH; u++)
#pragma unroll
          for (int kk = 0; kk < CM; kk++) {
            const int i = i0 + u;
            if (i >= VPT) continue;
            const int e = i * CM + kk;
            const unsigned ad = (e & 1) ? ff_rec_addr<1>(stbase, vidx[e >> 1]) : ff_rec_addr<0>(stbase, vidx[e >> 1]);
            st[u][kk] = *(lds_u4 *)(unsigned long)ad;
          }
#pragma unroll
        for (int u = 0; u < VCH; u++) {
          const int i = i0 + u;
          if (i >= VPT) continue;
          const int w = tid + B * i;
          float sum = (i < VPT - 2 || w < a.var_words) ? q0L[w] : 0.0f;
#pragma unroll
          for (int kk = 0; kk < CM; kk++) {
            const int e = i * CM + kk;
            const unsigned sh = (e & 1) ? ff_rec_shift<1>(vidx[e >> 1]) : ff_rec_shift<0>(vidx[e >> 1]);
            const unsigned mg = (st[u][kk].w == sh) ? st[u][kk].y : st[u][kk].x;
            sum = sum + __uint_as_float(FQ_ANDOR(st[u][kk].z << sh, 0x80000000u, mg));
          }
          if (i < VPT - 2 || w < a.var_words) ffsum[w] = sum + 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);   // bounds the records in flight (live registers)
      }
      __syncthreads();
    }

    if (vec_out) {
      fq_u2 *o8 = (fq_u2 *)(a.bits + (size_t)frame * a.n);
      const unsigned *us = (const unsigned *)ffsum;
      for (int i = tid; i < (a.n >> 3); i += B) {
        const fq_u4 w = pos8[i];
        fq_u2 b;
        b.x = (us[w.x & 0xffffu] >> 31) | ((us[w.x >> 16] >> 31) << 8) | ((us[w.y & 0xffffu] >> 31) << 16) | ((us[w.y >> 16] >> 31) << 24);
        b.y = (us[w.z & 0xffffu] >> 31) | ((us[w.z >> 16] >> 31) << 8) | ((us[w.w & 0xffffu] >> 31) << 16) | ((us[w.w >> 16] >> 31) << 24);
        __builtin_nontemporal_store(b, o8 + i);
      }
    } else if (a.bits) {
      uint8_t *o = a.bits + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = ffsum[a.pos[i]] < 0.0f;
    }
    if (a.llr_out) {
      float *o = a.llr_out + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = ffsum[a.pos[i]];
    }
    if (a.sout) {
#pragma unroll
      for (int r = 0; r < CPT; r++)
        if (has[r]) a.sout[(size_t)frame * a.m + r * B + tid] = (uint8_t)par[r];
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

// ---------------------------------------------------------------------------------------
// Register-cached binary64 variant (round 3): dec() of test_code_min_sum_B.c:137-208 in the
// reference's own arithmetic with every index in VGPRs, as fused_ms_reg_kernel has them for
// binary32.  What differs is where things live, because binary64 sums of a 16 000-variable frame
// are 128 KB of the 160 KB:
//   LDS     sum[words] binary64 (placed by the schedule compiler); per check {min1, min2} as one
//           16-byte pair, its sign word, and one byte for the slot of min1 (the packed records of
//           the QG kernels above)
//   VGPRs   q0 of the thread's VPT variables (there is no LDS left for it, and a per-workgroup row in
//           global memory -- the QG kernels' way, and this kernel's first build -- streams 128 KB per
//           frame-iteration from the Infinity Cache: 4.5 TB/s of fabric traffic, 64 % of the wave cycles
//           parked, 18 Gbit/s), (check, sign shift) of every edge of those variables, the compressed
//           state of the thread's checks
//   global  the slot addresses of the thread's checks (a.colP: two 16-bit entries per word, each the
//           sum's index * 4 -- the byte address is twice that): a read-only table of 100 KB that every
//           workgroup shares, so it lives in L2, read a chunk ahead of its use with 16-byte loads
// Every floating-point operation is the reference's, on the same operands in the same order.
#ifndef FF64_PKMAX
#define FF64_PKMAX 4
#endif
// packed slot-address words per load (2 slots each) of the binary64 register-cached kernel
constexpr int ff64_pk(int r2) { return (r2 % 4 == 0 && FF64_PKMAX >= 4) ? 4 : (r2 % 2 == 0 && FF64_PKMAX >= 2) ? 2 : 1; }
typedef __attribute__((address_space(3))) double lds_double;
typedef double ff_d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) ff_d2 lds_d2;
typedef __attribute__((address_space(3))) unsigned lds_uint;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

template <int HI>
__device__ __forceinline__ lds_double *ff_dslot_ptr(unsigned packed) {
  unsigned r;   // entry = index * 4; the dynamic LDS starts at address 0 (checked on the host)
  if (HI) asm volatile("v_lshrrev_b32 %0, 15, %1\n\tv_and_b32 %0, 0x7fff8, %0" : "=&v"(r) : "v"(packed));
  else asm volatile("v_and_b32 %0, 0xffff, %1\n\tv_add_u32 %0, %0, %0" : "=&v"(r) : "v"(packed));
  return (lds_double *)(unsigned long)r;
}
template <int HI>
__device__ __forceinline__ unsigned ff_check_of(unsigned packed) {
  unsigned r;   // volatile: not hoisted out of the iteration loop (a register per edge)
  if (HI) asm volatile("v_bfe_u32 %0, %1, 21, 11" : "=v"(r) : "v"(packed));
  else asm volatile("v_bfe_u32 %0, %1, 5, 11" : "=v"(r) : "v"(packed));
  return r;
}

template <int CPT, int R2, int VPT, int CM, int LB, int MINW>
__global__ __launch_bounds__(LB, MINW) void fused_ms_reg64_kernel(FfArgs a) {
  extern __shared__ __attribute__((aligned(16))) double ffsum64[];   // [words] sums, then the records
  constexpr int RM = 2 * R2;
  static_assert(RM <= 32, "one sign word per check");
  constexpr int NE = VPT * CM, NE2 = (NE + 1) / 2;
  constexpr int B = LB;
  const int tid = threadIdx.x;
  ff_d2 *mag = (ff_d2 *)(ffsum64 + a.words);                  // [m + 1] {min1, min2}     (words is even: 16-byte aligned)
  unsigned *sgL = (unsigned *)(mag + a.m + 1);                // [m + 1] sign words
  unsigned char *argL = (unsigned char *)(sgL + a.m + 1);     // [m + 1] slot of min1 (as a shift: 32 - RM + slot)
  int *frame_slot = (int *)(((size_t)(argL + a.m + 1) + 3) & ~(size_t)3);
  const unsigned magbase = (unsigned)a.words * 8u;
  const unsigned sgbase = magbase + (unsigned)(a.m + 1) * 16u;
  const unsigned argbase = sgbase + (unsigned)(a.m + 1) * 4u;
  // slot addresses: PK packed words (2 PK slots) per load, [round][chunk][thread][PK] (host: colP)
  constexpr int PK = ff64_pk(R2);
  constexpr int NCH = R2 / PK;                 // chunks per check
  constexpr int CH = 2 * PK;                   // slots per chunk
  typedef unsigned ff_pk __attribute__((ext_vector_type(PK)));
  // (uniform base + one 32-bit lane offset: the chunk's position is scalar arithmetic, not an address register pair per chunk)
  const ff_pk *colP = (const ff_pk *)a.colP;

  bool has[CPT];
  unsigned allm[CPT];
  unsigned vidx[NE2];
#pragma unroll
  for (int r = 0; r < CPT; r++) {
    has[r] = r < a.rounds && r * B + tid < a.m;
    allm[r] = r < a.rounds ? a.maskT[(size_t)r * B + tid] : 0u;
  }
#pragma unroll
  for (int e2 = 0; e2 < NE2; e2++) {
    const unsigned lo = a.vtabT[(size_t)(2 * e2) * B + tid];
    const unsigned hi = 2 * e2 + 1 < NE ? a.vtabT[(size_t)(2 * e2 + 1) * B + tid] : ((unsigned)a.m << 5);
    vidx[e2] = lo | (hi << 16);
    asm volatile("" : "+v"(vidx[e2]));
  }
  const bool wave_full = __ballot(has[CPT - 1]) != 0;
  const bool vec_out = a.bits && (a.n & 7) == 0 && ((size_t)a.bits & 7) == 0;
  const fq_u4 *pos8 = (const fq_u4 *)a.pos;

  for (int i = a.dummy_base + tid; i < a.words; i += B) ffsum64[i] = 1e30;     // pad slots: sign 0, never a minimum
  for (int i = tid; i < a.var_words; i += B) ffsum64[i] = 0.0;                 // (words that hold no variable)
  if (tid == 0) {
    ff_d2 none; none.x = 0.0; none.y = 0.0;                     // "no edge": message -0.0
    mag[a.m] = none; sgL[a.m] = 0xffffffffu; argL[a.m] = 0xff;
  }
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
    {
      const size_t base = (size_t)frame * a.n;
      if (a.dt == 0 && (a.n & 1) == 0 && ((size_t)a.llr & 15) == 0) {
        const ff_d2 *p2 = (const ff_d2 *)((const double *)a.llr + base);
        const unsigned *pos2 = (const unsigned *)a.pos;
        for (int i = tid; i < (a.n >> 1); i += B) {
          const ff_d2 v = __builtin_nontemporal_load(p2 + i);
          const unsigned w = pos2[i];
          ffsum64[w & 0xffffu] = v.x + 0.0;
          ffsum64[w >> 16] = v.y + 0.0;
        }
      } else if (a.dt == 0) {
        const double *p = (const double *)a.llr + base;
        for (int i = tid; i < a.n; i += B) ffsum64[a.pos[i]] = p[i] + 0.0;
      } else {
        const float *p = (const float *)a.llr + base;
        for (int i = tid; i < a.n; i += B) ffsum64[a.pos[i]] = (double)p[i] + 0.0;
      }
    }
    double om1[CPT], om2[CPT];
    unsigned oarg[CPT], osg[CPT];
    int sbit[CPT], par[CPT];
#pragma unroll
    for (int r = 0; r < CPT; r++) {
      om1[r] = om2[r] = 0.0;
      oarg[r] = 0u; osg[r] = 0u;
      sbit[r] = (a.synd && has[r]) ? (a.synd[(size_t)frame * a.m + r * B + tid] & 1) : 0;
      par[r] = 0;
    }
    if (tid == 0) { frame_slot[1] = 0; frame_slot[2] = 0; }   // flags of fq_block_or
    __syncthreads();
    // q0 of this thread's variables (the sums start as q0 + 0.0; the zero's sign never reaches a result)
    double q0v[VPT];
#pragma unroll
    for (int i = 0; i < VPT; i++) {
      const int w = tid + B * i;
      q0v[i] = (i < VPT - 2 || w < a.var_words) ? ffsum64[w] : 0.0;
    }

    int it_done = a.max_iter, ok = 0;
    for (int t = 1; t <= a.max_iter + 1; t++) {
      int bad = 0;
      // one check at a time (the binary32 kernel interleaves a thread's checks; with 64-bit values
      // that working set does not fit beside the index registers -- 73 spilled registers)
      auto check_pass = [&](auto na_c) {
        constexpr int NA = decltype(na_c)::value;
#pragma unroll
        for (int r = 0; r < NA; r++) {
          double m1 = (double)FF_CAP, m2 = (double)FF_CAP;
          unsigned arg = 0u, pp = 0u;
          unsigned sg = osg[r] << (32 - RM);      // old signs leave at the top while the new ones enter at the bottom
          unsigned long long ob[5];               // the slot of the previous minimum, one lane mask per bit (scalar registers)
#pragma unroll
          for (int b = 0; b < 5; b++) ob[b] = __ballot((oarg[r] >> b) & 1u);
          ff_pk pk = colP[(unsigned)((r * NCH) * B) + (unsigned)tid];
#pragma unroll
          for (int kc = 0; kc < RM; kc += CH) {
            // this chunk's slot addresses are here; the next chunk's (the next check's first) are requested now
            unsigned ix[PK];
#pragma unroll
            for (int j = 0; j < PK; j++) ix[j] = pk[j];
            if (kc + CH < RM || r + 1 < NA) pk = colP[(unsigned)((r * NCH + kc / CH + 1) * B) + (unsigned)tid];
            double S[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) S[j] = (j & 1) ? *ff_dslot_ptr<1>(ix[j >> 1]) : *ff_dslot_ptr<0>(ix[j >> 1]);
#pragma unroll
            for (int j = 0; j < CH; j++) {
              const int k = kc + j;
              if (k >= RM) continue;
              if ((j & 1) == 0) pp = __builtin_amdgcn_bitop3_b32(pp, ff_hi(S[j]), ff_hi(S[j + 1]), 0x96);
              unsigned long long in_k = (k & 1) ? ob[0] : ~ob[0];
#pragma unroll
              for (int b = 1; b < 5; b++) in_k &= ((k >> b) & 1) ? ob[b] : ~ob[b];
              const double mo = ff_pick(om1[r], om2[r], in_k);                  // min2 in the slot of min1
              const double co = __hiloint2double((int)FQ_ANDOR(sg, 0x80000000u, (unsigned)__double2hiint(mo)), __double2loint(mo));
              const double q = S[j] - co;                                       // :150
              sg = __builtin_amdgcn_alignbit(sg, ff_hi(q), 31);
              const bool lt = ff_abs_lt(q, m1);
              m2 = ff_min2(m2, ff_max_abs(m1, q));
              arg = lt ? (unsigned)k : arg;
              m1 = ff_min_abs(m1, q);
            }
            asm volatile("" : "+v"(pp), "+v"(m1), "+v"(m2), "+v"(arg), "+v"(sg));
            __builtin_amdgcn_sched_barrier(0);
          }
          const unsigned p = pp >> 31;
          par[r] = (int)p;
          bad |= has[r] ? ((int)p ^ sbit[r]) : 0;
          const unsigned tot = (unsigned)sbit[r] ^ (unsigned)__popc(sg);
          osg[r] = (tot & 1u) ? (sg ^ allm[r]) : sg;
          om1[r] = m1; om2[r] = m2; oarg[r] = arg;
          asm volatile("" : "+v"(om1[r]), "+v"(om2[r]), "+v"(oarg[r]), "+v"(osg[r]));
          if (has[r]) {
            const int j = r * B + tid;
            ff_d2 mm; mm.x = om1[r]; mm.y = om2[r];
            mag[j] = mm; sgL[j] = osg[r]; argL[j] = (unsigned char)(oarg[r] + (unsigned)(32 - RM));
          }
        }
      };
      auto parity_only = [&](auto na_c) {
        constexpr int NA = decltype(na_c)::value;
#pragma unroll
        for (int r = 0; r < NA; r++) {
          unsigned pp = 0;
#pragma unroll
          for (int c = 0; c < NCH; c++) {
            const ff_pk pk = colP[(unsigned)((r * NCH + c) * B) + (unsigned)tid];
#pragma unroll
            for (int j = 0; j < PK; j++) {
              const unsigned ix = pk[j];
              const double s0 = *ff_dslot_ptr<0>(ix);
              const double s1 = *ff_dslot_ptr<1>(ix);
              pp = __builtin_amdgcn_bitop3_b32(pp, ff_hi(s0), ff_hi(s1), 0x96);
            }
          }
          par[r] = (int)(pp >> 31);
          bad |= has[r] ? (par[r] ^ sbit[r]) : 0;
        }
      };
      if (t == a.max_iter + 1) {
        if (wave_full) parity_only(std::integral_constant<int, CPT>{});
        else parity_only(std::integral_constant<int, CPT - 1>{});
      } else if (wave_full) check_pass(std::integral_constant<int, CPT>{});
      else check_pass(std::integral_constant<int, CPT - 1>{});
      const int any_bad = fq_block_or(bad, frame_slot + 1, t);
      if (t > 1 && !any_bad) { it_done = t - 1; ok = 1; break; }
      if (t == a.max_iter + 1) break;
      // ---- variable pass
#pragma unroll
      for (int i = 0; i < VPT; i++) {
        ff_d2 mg[CM];
        unsigned sw[CM], ar[CM];
#pragma unroll
        for (int kk = 0; kk < CM; kk++) {
          const int e = i * CM + kk;
          const unsigned cj = (e & 1) ? ff_check_of<1>(vidx[e >> 1]) : ff_check_of<0>(vidx[e >> 1]);
          mg[kk] = *(lds_d2 *)(unsigned long)(magbase + cj * 16u);
          sw[kk] = *(lds_uint *)(unsigned long)(sgbase + cj * 4u);
          ar[kk] = *(lds_u8 *)(unsigned long)(argbase + cj);
        }
        const int w = tid + B * i;
        double sum = q0v[i];
#pragma unroll
        for (int kk = 0; kk < CM; kk++) {
          const int e = i * CM + kk;
          const unsigned sh = (e & 1) ? ff_rec_shift<1>(vidx[e >> 1]) : ff_rec_shift<0>(vidx[e >> 1]);
          const double m = (ar[kk] == sh) ? mg[kk].y : mg[kk].x;
          sum = sum + __hiloint2double((int)FQ_ANDOR(sw[kk] << sh, 0x80000000u, (unsigned)__double2hiint(m)), __double2loint(m));
        }
        if (i < VPT - 2 || w < a.var_words) ffsum64[w] = sum + 0.0;
        __builtin_amdgcn_sched_barrier(0);   // bounds the records in flight (live registers)
      }
      __syncthreads();
    }

    const unsigned *us = (const unsigned *)ffsum64;      // (sign of sum: bit 31 of its high word)
    if (vec_out) {
      fq_u2 *o8 = (fq_u2 *)(a.bits + (size_t)frame * a.n);
      for (int i = tid; i < (a.n >> 3); i += B) {
        const fq_u4 w = pos8[i];
        fq_u2 b;
        b.x = (us[2 * (w.x & 0xffffu) + 1] >> 31) | ((us[2 * (w.x >> 16) + 1] >> 31) << 8) | ((us[2 * (w.y & 0xffffu) + 1] >> 31) << 16) | ((us[2 * (w.y >> 16) + 1] >> 31) << 24);
        b.y = (us[2 * (w.z & 0xffffu) + 1] >> 31) | ((us[2 * (w.z >> 16) + 1] >> 31) << 8) | ((us[2 * (w.w & 0xffffu) + 1] >> 31) << 16) | ((us[2 * (w.w >> 16) + 1] >> 31) << 24);
        __builtin_nontemporal_store(b, o8 + i);
      }
    } else if (a.bits) {
      uint8_t *o = a.bits + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = ffsum64[a.pos[i]] < 0.0;
    }
    if (a.llr_out) {
      float *o = a.llr_out + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = (float)ffsum64[a.pos[i]];
    }
    if (a.sout) {
#pragma unroll
      for (int r = 0; r < CPT; r++)
        if (has[r]) a.sout[(size_t)frame * a.m + r * B + tid] = (uint8_t)par[r];
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

// ---------------------------------------------------------------------------------------
// Wide-row binary32 variant (round 3): frames whose channel LLRs do not fit in LDS beside the sums and
// whose checks have 33 ... 48 slots -- 32000.2240.3.105: 128 KB of sums, row weights 42 / 43.  The plan of
// the binary64 kernel above (q0 in registers, slot addresses from a shared L2-resident table), plus:
//   LDS     sum[words] binary32; per check {min1, min2} (8 bytes), the signs of slots 0..31 (4 bytes), of
//           slots 32..RM-1 (2 bytes) and the slot of min1 (1 byte), as four arrays: 15 bytes per check --
//           a 16-byte record would put this code 272 bytes over the 160 KB
//   VGPRs   q0 of the thread's VPT variables, the compressed state of its checks
//   global  a.colP: slot addresses of the thread's checks, two per word (byte address in 17 bits | word
//           index << 17), [round][chunk][thread][PK];
//           a.vtabW: one 32-bit entry per edge of the thread's variables, [pair of variables][load][thread][4]:
//           check (12 bits) | byte offset of the sign word that holds the edge's bit (14) | the edge's
//           slot code (6: shift that brings its sign to bit 31, + 32 for slots >= 32); both tables are
//           read a step ahead of their use with 16- / 8-byte loads and shared by every workgroup
// The 16-bit sign words of two neighbouring checks share a 32-bit word; the variable pass reads that
// word, and the entry's shift accounts for the half it wants (so every lane issues the same three LDS
// reads per edge: b64, b32, u8).
typedef float ff_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) ff_f2 lds_f2;

typedef __amdgpu_buffer_rsrc_t ff_rsrc;
__device__ __forceinline__ ff_rsrc ff_make_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000 /* raw dwords, gfx9 */);
}
__device__ __forceinline__ fq_u4 ff_ld128(ff_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(fq_u4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ fq_u2 ff_ld64(ff_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(fq_u2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}

template <int HI>
__device__ __forceinline__ lds_float *ff_wslot_ptr(unsigned packed) {
  unsigned r;   // volatile: not hoisted out of the iteration loop
  if (HI) asm volatile("v_lshrrev_b32 %0, 15, %1\n\tv_and_b32 %0, 0x1fffc, %0" : "=&v"(r) : "v"(packed));
  else asm volatile("v_and_b32 %0, 0x1fffc, %1" : "=v"(r) : "v"(packed));
  return (lds_float *)(unsigned long)r;
}

template <int CPT, int R2, int VPT, int CM, int LB, int MINW>
__global__ __launch_bounds__(LB, MINW) void fused_ms_wide_kernel(FfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ffsumw[];   // [words] sums, then the record arrays
  constexpr int RM = 2 * R2;
  static_assert(RM > 32 && RM <= 48 && R2 % 2 == 0, "two sign words, the second one of at most 16 bits");
  constexpr int RB = RM - 32;                  // slots in the second sign word
  static_assert(VPT % 2 == 0, "variables are walked in pairs");
  constexpr int LPG = (2 * CM + 3) / 4;        // 16-byte loads of entries per pair of variables
  constexpr int B = LB;
  constexpr int PK = (R2 % 4 == 0) ? 4 : 2;    // packed slot-address words per load
  constexpr int NCH = R2 / PK, CH = 2 * PK;    // chunks per check, slots per chunk
  typedef unsigned ff_pk __attribute__((ext_vector_type(PK)));
  const int tid = threadIdx.x;
  ff_f2 *mag = (ff_f2 *)(ffsumw + a.words);                       // [m + 1] {min1, min2}   (words is even)
  unsigned *sgL = (unsigned *)(mag + a.m + 1);                    // [m + 1] signs of slots 0..31 (slot k at bit 31-k)
  unsigned short *sgH = (unsigned short *)(sgL + a.m + 1);        // [m + 1] signs of slots 32..RM-1 (slot k at bit RM-1-k)
  unsigned char *argL = (unsigned char *)(sgH + a.m + 1);         // [m + 1] slot code of min1
  int *frame_slot = (int *)(((size_t)(argL + a.m + 1) + 3) & ~(size_t)3);
  const unsigned magbase = (unsigned)a.words * 4u;                // (the dynamic LDS starts at address 0: checked on the host)
  const unsigned sgbase = magbase + (unsigned)(a.m + 1) * 8u;
  const unsigned argbase = sgbase + (unsigned)(a.m + 1) * 6u;

  bool has[CPT];
#pragma unroll
  for (int r = 0; r < CPT; r++) has[r] = r < a.rounds && r * B + tid < a.m;
  // the three index tables through buffer loads: uniform base in a descriptor, the thread's column as one 32-bit
  // lane offset, the row as a scalar offset (global loads cost a 64-bit vector add and an address pair per access)
  const ff_rsrc rcol = ff_make_rsrc(a.colP, (unsigned)a.rounds * R2 * B * 4u);
  const ff_rsrc rvt = ff_make_rsrc(a.vtabW, (unsigned)(VPT / 2) * LPG * B * 16u);
  const ff_rsrc rmk = ff_make_rsrc(a.maskT, (unsigned)a.rounds * B * 8u);         // [rounds][B] real-slot bits of the two sign words
  auto ld_pk = [&](unsigned soff) -> ff_pk {
    if constexpr (PK == 4) return ff_ld128(rcol, (unsigned)tid * 16u, soff);
    else return ff_ld64(rcol, (unsigned)tid * 8u, soff);
  };
  const bool wave_full = __ballot(has[CPT - 1]) != 0;
  const bool vec_out = a.bits && (a.n & 7) == 0 && ((size_t)a.bits & 7) == 0;
  const bool vec_in = (a.n & 3) == 0 && ((size_t)a.llr & 15) == 0;
  const fq_u4 *pos8 = (const fq_u4 *)a.pos;

  for (int i = a.dummy_base + tid; i < a.words; i += B) ffsumw[i] = 1e30f;     // pad slots: sign 0, never a minimum
  for (int i = tid; i < a.var_words; i += B) ffsumw[i] = 0.0f;                  // (words that hold no variable)
  if (tid == 0) {
    ff_f2 none; none.x = 0.0f; none.y = 0.0f;                     // "no edge": message -0.0
    mag[a.m] = none; sgL[a.m] = 0xffffffffu; sgH[a.m] = 0xffffu; argL[a.m] = 0xff;
  }
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
    {
      const size_t base = (size_t)frame * a.n;
      if (a.dt == 1 && vec_in) {
        typedef float ff_f4 __attribute__((ext_vector_type(4)));
        const ff_f4 *p4 = (const ff_f4 *)((const float *)a.llr + base);
        const fq_u2 *pos4 = (const fq_u2 *)a.pos;
        for (int i = tid; i < (a.n >> 2); i += B) {
          const ff_f4 v = __builtin_nontemporal_load(p4 + i);
          const fq_u2 w = pos4[i];
          ffsumw[w.x & 0xffffu] = v.x + 0.0f;
          ffsumw[w.x >> 16] = v.y + 0.0f;
          ffsumw[w.y & 0xffffu] = v.z + 0.0f;
          ffsumw[w.y >> 16] = v.w + 0.0f;
        }
      } else if (a.dt == 1) {
        const float *p = (const float *)a.llr + base;
        for (int i = tid; i < a.n; i += B) ffsumw[a.pos[i]] = __builtin_nontemporal_load(p + i) + 0.0f;
      } else {
        const double *p = (const double *)a.llr + base;
        for (int i = tid; i < a.n; i += B) ffsumw[a.pos[i]] = (float)__builtin_nontemporal_load(p + i) + 0.0f;
      }
    }
    float om1[CPT], om2[CPT];
    unsigned oarg[CPT], osgA[CPT], osgB[CPT];
    int sbit[CPT], par[CPT];
#pragma unroll
    for (int r = 0; r < CPT; r++) {
      om1[r] = om2[r] = 0.0f;
      oarg[r] = 0u; osgA[r] = 0u; osgB[r] = 0u;
      sbit[r] = (a.synd && has[r]) ? (a.synd[(size_t)frame * a.m + r * B + tid] & 1) : 0;
      par[r] = 0;
    }
    if (tid == 0) { frame_slot[1] = 0; frame_slot[2] = 0; }   // flags of fq_block_or
    __syncthreads();
    // q0 of this thread's variables (the sums start as q0 + 0.0; the zero's sign never reaches a result)
    float q0v[VPT];
#pragma unroll
    for (int i = 0; i < VPT; i++) {
      const int w = tid + B * i;
      q0v[i] = w < a.var_words ? ffsumw[w] : 0.0f;
    }

    int it_done = a.max_iter, ok = 0;
    for (int t = 1; t <= a.max_iter + 1; t++) {
      int bad = 0;
      // (the tables are loop-invariant: a scalar offset that is opaque per iteration keeps their loads in the loop)
      unsigned so = 0u;
      asm volatile("" : "+s"(so));
      auto check_pass = [&](auto na_c) {
        constexpr int NA = decltype(na_c)::value;
#pragma unroll
        for (int r = 0; r < NA; r++) {
          float m1 = (float)FF_CAP, m2 = (float)FF_CAP;
          unsigned arg = 0u, pp = 0u;
          // old signs leave at the top of their word while the new ones enter at the bottom
          unsigned sgA = osgA[r], sgB = osgB[r] << (32 - RB);
          unsigned long long ob[6];               // the slot of the previous minimum, one lane mask per bit (scalar registers)
#pragma unroll
          for (int b = 0; b < 6; b++) ob[b] = __ballot((oarg[r] >> b) & 1u);
          ff_pk pk = ld_pk(so + (unsigned)((r * NCH) * B * PK * 4));
          const fq_u2 allm = ff_ld64(rmk, (unsigned)tid * 8u, so + (unsigned)(r * B * 8));   // (re-read per iteration: two registers per check otherwise)
#pragma unroll
          for (int kc = 0; kc < RM; kc += CH) {
            unsigned ix[PK];
#pragma unroll
            for (int j = 0; j < PK; j++) ix[j] = pk[j];
            if (kc + CH < RM || r + 1 < NA) pk = ld_pk(so + (unsigned)((r * NCH + kc / CH + 1) * B * PK * 4));
            float S[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) S[j] = (j & 1) ? *ff_wslot_ptr<1>(ix[j >> 1]) : *ff_wslot_ptr<0>(ix[j >> 1]);
#pragma unroll
            for (int j = 0; j < CH; j++) {
              const int k = kc + j;
              if ((j & 1) == 0) pp = __builtin_amdgcn_bitop3_b32(pp, __float_as_uint(S[j]), __float_as_uint(S[j + 1]), 0x96);
              unsigned long long in_k = (k & 1) ? ob[0] : ~ob[0];
#pragma unroll
              for (int b = 1; b < 6; b++) in_k &= ((k >> b) & 1) ? ob[b] : ~ob[b];
              const float mo = ff_pick(om1[r], om2[r], in_k);                  // min2 in the slot of min1
              if (k < 32) {
                const float q = S[j] - __uint_as_float(FQ_ANDOR(sgA, 0x80000000u, __float_as_uint(mo)));   // :150
                sgA = __builtin_amdgcn_alignbit(sgA, __float_as_uint(q), 31);
                const float aq = __builtin_fabsf(q);
                m2 = __builtin_amdgcn_fmed3f(m1, aq, m2);
                const bool lt = aq < m1;
                arg = lt ? (unsigned)k : arg;
                m1 = lt ? aq : m1;
              } else {
                const float q = S[j] - __uint_as_float(FQ_ANDOR(sgB, 0x80000000u, __float_as_uint(mo)));
                sgB = __builtin_amdgcn_alignbit(sgB, __float_as_uint(q), 31);
                const float aq = __builtin_fabsf(q);
                m2 = __builtin_amdgcn_fmed3f(m1, aq, m2);
                const bool lt = aq < m1;
                arg = lt ? (unsigned)k : arg;
                m1 = lt ? aq : m1;
              }
            }
            asm volatile("" : "+v"(pp), "+v"(m1), "+v"(m2), "+v"(arg), "+v"(sgA), "+v"(sgB));
            __builtin_amdgcn_sched_barrier(0);
          }
          const unsigned p = pp >> 31;
          par[r] = (int)p;
          bad |= has[r] ? ((int)p ^ sbit[r]) : 0;
          const unsigned tot = (unsigned)sbit[r] ^ (unsigned)__popc(sgA) ^ (unsigned)__popc(sgB);
          osgA[r] = (tot & 1u) ? (sgA ^ allm.x) : sgA;
          osgB[r] = (tot & 1u) ? (sgB ^ allm.y) : sgB;
          om1[r] = m1; om2[r] = m2; oarg[r] = arg;
          asm volatile("" : "+v"(om1[r]), "+v"(om2[r]), "+v"(oarg[r]), "+v"(osgA[r]), "+v"(osgB[r]));
          if (has[r]) {
            const int j = r * B + tid;
            ff_f2 mm; mm.x = om1[r]; mm.y = om2[r];
            mag[j] = mm; sgL[j] = osgA[r]; sgH[j] = (unsigned short)osgB[r];
            // the slot of min1 in the entries' code: slots < 32 by their number; slots >= 32 by 32 + the shift
            // that brings their bit of the shared 32-bit word to bit 31 (odd checks sit in the high half)
            const unsigned code = arg < 32u ? arg : arg + (unsigned)(64 - RM) - ((unsigned)(j & 1) << 4);
            argL[j] = (unsigned char)code;
          }
        }
      };
      auto parity_only = [&](auto na_c) {
        constexpr int NA = decltype(na_c)::value;
#pragma unroll
        for (int r = 0; r < NA; r++) {
          unsigned pp = 0;
#pragma unroll
          for (int c = 0; c < NCH; c++) {
            const ff_pk pk = ld_pk(so + (unsigned)((r * NCH + c) * B * PK * 4));
#pragma unroll
            for (int j = 0; j < PK; j++) {
              const unsigned ix = pk[j];
              const float s0 = *ff_wslot_ptr<0>(ix);
              const float s1 = *ff_wslot_ptr<1>(ix);
              pp = __builtin_amdgcn_bitop3_b32(pp, __float_as_uint(s0), __float_as_uint(s1), 0x96);
            }
          }
          par[r] = (int)(pp >> 31);
          bad |= has[r] ? (par[r] ^ sbit[r]) : 0;
        }
      };
      if (t == a.max_iter + 1) {
        if (wave_full) parity_only(std::integral_constant<int, CPT>{});
        else parity_only(std::integral_constant<int, CPT - 1>{});
      } else if (wave_full) check_pass(std::integral_constant<int, CPT>{});
      else check_pass(std::integral_constant<int, CPT - 1>{});
      const int any_bad = fq_block_or(bad, frame_slot + 1, t);
      if (t > 1 && !any_bad) { it_done = t - 1; ok = 1; break; }
      if (t == a.max_iter + 1) break;
      // ---- variable pass: two variables (LPG 16-byte loads of entries) per step, the next step's entries
      // requested before this step's records
      fq_u4 en[LPG];
#pragma unroll
      for (int l = 0; l < LPG; l++) en[l] = ff_ld128(rvt, (unsigned)tid * 16u, so + (unsigned)(l * B * 16));
#pragma unroll
      for (int g = 0; g < VPT / 2; g++) {
        unsigned ee[4 * LPG];
#pragma unroll
        for (int l = 0; l < LPG; l++) { ee[4 * l] = en[l].x; ee[4 * l + 1] = en[l].y; ee[4 * l + 2] = en[l].z; ee[4 * l + 3] = en[l].w; }
        if (g + 1 < VPT / 2) {
#pragma unroll
          for (int l = 0; l < LPG; l++) en[l] = ff_ld128(rvt, (unsigned)tid * 16u, so + (unsigned)(((g + 1) * LPG + l) * B * 16));
        }
        float mgx[2][CM], mgy[2][CM];
        unsigned sw[2][CM], ar[2][CM];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
          for (int kk = 0; kk < CM; kk++) {
            const unsigned e = ee[u * CM + kk];
            const unsigned cj = e & 0xfffu;
            const ff_f2 mg = *(lds_f2 *)(unsigned long)(magbase + cj * 8u);
            mgx[u][kk] = mg.x; mgy[u][kk] = mg.y;
            sw[u][kk] = *(lds_uint *)(unsigned long)(sgbase + __builtin_amdgcn_ubfe(e, 12, 14));
            ar[u][kk] = *(lds_u8 *)(unsigned long)(argbase + cj);
            // (both minima are wanted in registers: left alone the select below becomes a second, dependent LDS read)
            asm volatile("" : "+v"(mgx[u][kk]), "+v"(mgy[u][kk]));
          }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int i = g * 2 + u;
          const int w = tid + B * i;
          float sum = q0v[i];
#pragma unroll
          for (int kk = 0; kk < CM; kk++) {
            const unsigned code = ee[u * CM + kk] >> 26;
            const unsigned m = (ar[u][kk] == code) ? __float_as_uint(mgy[u][kk]) : __float_as_uint(mgx[u][kk]);
            sum = sum + __uint_as_float(FQ_ANDOR(sw[u][kk] << (code & 31u), 0x80000000u, m));
          }
          if (w < a.var_words) ffsumw[w] = sum + 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);   // bounds the records in flight (live registers)
      }
      __syncthreads();
    }

    if (vec_out) {
      fq_u2 *o8 = (fq_u2 *)(a.bits + (size_t)frame * a.n);
      const unsigned *us = (const unsigned *)ffsumw;
      for (int i = tid; i < (a.n >> 3); i += B) {
        const fq_u4 w = pos8[i];
        fq_u2 b;
        b.x = (us[w.x & 0xffffu] >> 31) | ((us[w.x >> 16] >> 31) << 8) | ((us[w.y & 0xffffu] >> 31) << 16) | ((us[w.y >> 16] >> 31) << 24);
        b.y = (us[w.z & 0xffffu] >> 31) | ((us[w.z >> 16] >> 31) << 8) | ((us[w.w & 0xffffu] >> 31) << 16) | ((us[w.w >> 16] >> 31) << 24);
        __builtin_nontemporal_store(b, o8 + i);
      }
    } else if (a.bits) {
      uint8_t *o = a.bits + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = ffsumw[a.pos[i]] < 0.0f;
    }
    if (a.llr_out) {
      float *o = a.llr_out + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = ffsumw[a.pos[i]];
    }
    if (a.sout) {
#pragma unroll
      for (int r = 0; r < CPT; r++)
        if (has[r]) a.sout[(size_t)frame * a.m + r * B + tid] = (uint8_t)par[r];
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

// the shapes the register-cached kernel is built for: {checks per thread, slot pairs, variables
// per thread, edges per variable}
#define FF_REG_SHAPES(X) \
  X(2, 16, 16, 3, 1024, 4) /* peg_16000_3_0.9 */ \
  X(2, 9, 10, 3, 512, 4)   /* 4095.737.3: two workgroups per CU */ \
  X(1, 7, 6, 3, 256, 4)    /* 1057.244.3: four */ \
  X(1, 3, 2, 3, 128, 4)    /* 204.33.484: eight */ \
  X(1, 5, 2, 7, 1024, 4)   /* Frolov_1024_0.5 (the Verilog trace's code): irregular, column weights to 7 */

// the same for the binary64 register-cached kernel
#define FF_REG64_SHAPES(X) \
  X(2, 16, 16, 3, 1024, 4) /* peg_16000_3_0.9 */ \
  X(2, 9, 10, 3, 512, 4)   /* 4095.737.3 */ \
  X(1, 7, 6, 3, 256, 4)    /* 1057.244.3 */ \
  X(1, 3, 2, 3, 128, 4)    /* 204.33.484 */ \
  X(1, 5, 2, 7, 1024, 4)   /* Frolov_1024_0.5 (the Verilog trace's code): irregular, column weights to 7 */

// the wide-row binary32 kernel (33 ... 48 slots per check)
#define FF_WIDE_SHAPES(X) \
  X(3, 22, 32, 3, 1024, 4) /* 32000.2240.3.105, 16 wavefronts */ \
  X(3, 22, 42, 3, 768, 3)  /* 32000.2240.3.105, 12 wavefronts of 168 registers */

namespace {

typedef void (*FfKern)(FfArgs);

template <typename T, int NW, bool QG = false>
FfKern ff_pick(int cpt) {
  switch (cpt) {
    case 1: return fused_ms_kernel<T, 1, NW, QG>;
    case 2: return fused_ms_kernel<T, 2, NW, QG>;
    case 3: return fused_ms_kernel<T, 3, NW, QG>;
    case 4: return fused_ms_kernel<T, 4, NW, QG>;
    case 5: case 6: return fused_ms_kernel<T, 6, NW, QG>;
    case 7: case 8: return fused_ms_kernel<T, 8, NW, QG>;
  }
  return nullptr;
}


FfKern ff_reg_lookup(int cpt, int r2, int vpt, int cm, int block) {
#define FF_X(C, R, V, M, LB, MW) if (cpt == C && r2 == R && vpt == V && cm == M && block == LB) return fused_ms_reg_kernel<C, R, V, M, LB, MW>;
  FF_REG_SHAPES(FF_X)
#undef FF_X
  return nullptr;
}

FfKern ff_reg64_lookup(int cpt, int r2, int vpt, int cm, int block) {
#define FF_X(C, R, V, M, LB, MW) if (cpt == C && r2 == R && vpt == V && cm == M && block == LB) return fused_ms_reg64_kernel<C, R, V, M, LB, MW>;
  FF_REG64_SHAPES(FF_X)
#undef FF_X
  return nullptr;
}

FfKern ff_wide_lookup(int cpt, int r2, int vpt, int cm, int block) {
#define FF_X(C, R, V, M, LB, MW) if (cpt == C && r2 == R && vpt == V && cm == M && block == LB) return fused_ms_wide_kernel<C, R, V, M, LB, MW>;
  FF_WIDE_SHAPES(FF_X)
#undef FF_X
  return nullptr;
}

template <typename T>
struct FusedFloatEngine : Engine {
  ldpc_opts opts{};
  int n = 0, m = 0, rmax = 0, cmax = 0, RM = 0, NW = 1;
  int block = 0, rounds = 0, wg_per_cu = 1, num_cu = 256;
  size_t lds = 0;
  DevBuf d_colT, d_realT, d_vtab, d_pos, d_counter, d_vtabT, d_maskT, d_colP, d_vtabW;
  LdsSchedule sched;
  FfKern kern = nullptr;
  bool use_reg = false;
  long vconf_before = 0, vconf_after = 0;   // extra LDS cycles of the variable pass per iteration (model)

  static size_t lds_bytes(int words, int var_words, int m, int nw) {
    const int sgw = nw == 1 ? 2 : 4;
    return sizeof(T) * ((size_t)words + var_words + 2 * (size_t)(m + 1)) + 4 * (size_t)sgw * (m + 1) + 16;
  }
  // the QG kernels: no q0 in LDS, packed records
  static size_t lds_bytes_qg(int words, int m, int nw) {
    return sizeof(T) * ((size_t)words + 2 * (size_t)(m + 1)) + 4 * (size_t)nw * (m + 1) + (size_t)(m + 1) + 24;
  }
  // the wide-row binary32 kernel: sums, 15 bytes per check, the frame word and the exit flags
  static size_t lds_bytes_wide(int words, int m) { return 4 * (size_t)words + 15 * (size_t)(m + 1) + 3 + 16; }
  bool qg = false;
  DevBuf d_q0g;
  void *ws_q0g = nullptr;
  // (the schedule may leave a few words of slack per bank: n + 64 bounds var_words closely enough
  // for a first test; init() checks the real figure)
  static bool fits(const ldpc_code &c) {
    if (c.q != 2 || c.n > 65000 || c.rmax > 64 || c.rmax < 1 || c.cmax > 32 || c.m > 8192) return false;
    const int nw = (2 * ((c.rmax + 1) / 2) + 31) / 32;
    return lds_bytes(c.n + 64 + FF_NDUMMY, c.n + 64, c.m, nw) + 1024 <= 160 * 1024 ||
           lds_bytes_qg(c.n + 64 + FF_NDUMMY, c.m, nw) + 256 <= 160 * 1024 ||
           (sizeof(T) == 4 && nw == 2 && lds_bytes_wide(c.n + FF_NDUMMY, c.m) <= 160 * 1024);
  }
  size_t llr_elems_per_frame() const override { return (size_t)n; }

  int init(const ldpc_code &c, const ldpc_opts &o) override {
    opts = o;
    n = c.n; m = c.m; rmax = c.rmax; cmax = c.cmax;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, o.device));
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    RM = std::max(2, 2 * ((rmax + 1) / 2));
    NW = (RM + 31) / 32;
    const int need = (m + 63) / 64;
    int optimise = 1;
    if (const char *e = getenv("LDPC_HIP_FUSED_PLACE")) optimise = atoi(e);
    int env_w = 0, env_reg = 1;
    if (const char *e = getenv("LDPC_HIP_FUSED_WAVES")) env_w = atoi(e);
    if (const char *e = getenv("LDPC_HIP_FUSED_REG")) env_reg = atoi(e);
    std::vector<int> row_of(c.E());
    for (int j = 0; j < m; j++)
      for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++) row_of[e] = j;
    int rc;

    // First choice: the register-cached kernel (binary32 or binary64), if one is built for this shape.  Try
    // the workgroup shapes in order of SIMD balance (wavefronts go to the 4 SIMDs in turn; the
    // busiest sets the pace; a wavefront of the last, partial round runs the shorter stream).
    use_reg = false;
    constexpr bool F64 = sizeof(T) == 8;
    if (env_reg && NW == 1 && m <= 2047 && cmax >= 1) {
      std::vector<std::pair<double, int>> cand;
      for (int W = 1; W <= 16; W++) {
        if (env_w && W != env_w) continue;
        const int r = (need + W - 1) / W;
        if (r > 4) continue;
        const int full = need - (r - 1) * W;
        int load[4] = {0, 0, 0, 0};
        for (int w = 0; w < W; w++) load[w & 3] += w < full ? r : r - 1;
        const int worst = std::max(std::max(load[0], load[1]), std::max(load[2], load[3]));
        cand.push_back({-((double)need / (4.0 * worst) + 0.002 * W), W});
      }
      std::sort(cand.begin(), cand.end());
      for (auto &cw : cand) {
        const int W = cw.second, B = W * 64;
        LdsSchedule sc;
        if (ldpc_build_lds_schedule(c, B, RM, FF_NDUMMY, optimise != 0, sc)) continue;
        int vpt = (sc.var_words + B - 1) / B;
        if ((cmax & 1) && (vpt & 1)) vpt++;
        FfKern k = F64 ? ff_reg64_lookup(sc.rounds, RM / 2, vpt, cmax, B) : ff_reg_lookup(sc.rounds, RM / 2, vpt, cmax, B);
        if (getenv("LDPC_HIP_DEBUG"))
          fprintf(stderr, "fused float: shape {%d checks/thread, %d slot pairs, %d variables/thread, %d edges/variable, %d threads}: %s\n",
                  sc.rounds, RM / 2, vpt, cmax, B, k ? "built" : "no register-cached kernel");
        if (!k) continue;
        hipFuncAttributes fa;
        HIPCHK(hipFuncGetAttributes(&fa, (const void *)k));
        if (F64) {
          // binary64: sums, {min1, min2} pairs, sign words, one byte per check; q0 in global memory.  The
          // kernel addresses its dynamic LDS from 0 and packs index * 4 in 16 bits.
          if (sc.words & 1) sc.words++;                                   // (16-byte alignment of the pairs)
          if (fa.sharedSizeBytes != 0 || (size_t)sc.words * 4 > 65535) continue;
        } else if ((size_t)sc.words * 4 + fa.sharedSizeBytes > 65535) continue;   // 16-bit absolute slot addresses
        const size_t need_lds = F64 ? (size_t)sc.words * 8 + (size_t)(m + 1) * 21 + 3 + 16
                                    : (size_t)sc.words * 4 + (size_t)(m + 1) * 16 + (size_t)sc.var_words * 4 + 16;
        if (need_lds + fa.sharedSizeBytes > 160 * 1024) continue;
        if (optimise) ldpc_place_variable_pass(c, sc, &vconf_before, &vconf_after);
        // edges of the variables each thread rebuilds: (check << 5) | shift of the sign bit
        std::vector<int> inv(sc.var_words, -1);
        for (int v = 0; v < n; v++) inv[sc.pos[v]] = v;
        std::vector<int> slot_of_edge(c.E(), -1);
        for (int j = 0; j < m; j++)
          for (int kk = 0; kk < RM; kk++) {
            const int w = sc.colT[((size_t)(j / B) * RM + kk) * B + j % B];
            if (w >= sc.dummy_base) continue;
            for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++)
              if (sc.pos[c.col_idx[e]] == w) slot_of_edge[e] = kk;
          }
        std::vector<uint16_t> vt((size_t)vpt * cmax * B, (uint16_t)(m << 5));
        bool lost = false;
        for (int t = 0; t < B; t++)
          for (int i = 0; i < vpt; i++) {
            const int w = t + B * i;
            if (w >= sc.var_words || inv[w] < 0) continue;
            const int v = inv[w];
            for (int kk = 0; kk < c.col_ptr[v + 1] - c.col_ptr[v]; kk++) {
              const int e = c.edge_of[c.col_ptr[v] + kk];
              if (slot_of_edge[e] < 0) lost = true;
              vt[((size_t)i * cmax + kk) * B + t] = (uint16_t)((row_of[e] << 5) | (32 - RM + slot_of_edge[e]));
            }
          }
        if (lost) continue;
        sched = sc;
        kern = k; use_reg = true;
        qg = false;
        block = B; rounds = sc.rounds;
        lds = need_lds;
        if (F64) {
          // the binary64 kernel's slot addresses: index * 4 in 16 bits, two per word, PK words per load,
          // [round][chunk][thread][PK]
          const int R2 = RM / 2, PK = ff64_pk(R2);
          std::vector<uint32_t> cp((size_t)sc.rounds * R2 * B, 0u);
          for (int r = 0; r < sc.rounds; r++)
            for (int k2 = 0; k2 < R2; k2++)
              for (int t = 0; t < B; t++) {
                const uint32_t lo = sc.colT[((size_t)r * RM + 2 * k2) * B + t], hi = sc.colT[((size_t)r * RM + 2 * k2 + 1) * B + t];
                cp[(((size_t)r * (R2 / PK) + k2 / PK) * B + t) * PK + k2 % PK] = (lo * 4u) | ((hi * 4u) << 16);
              }
          if ((rc = d_colP.ensure(cp.size() * 4))) return rc;
          HIPCHK(hipMemcpy(d_colP.p, cp.data(), cp.size() * 4, hipMemcpyHostToDevice));
        }
        if ((rc = d_vtabT.ensure(vt.size() * 2))) return rc;
        if ((rc = d_maskT.ensure(sc.maskT.size() * 4))) return rc;
        HIPCHK(hipMemcpy(d_vtabT.p, vt.data(), vt.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_maskT.p, sc.maskT.data(), sc.maskT.size() * 4, hipMemcpyHostToDevice));
        wg_per_cu = (int)std::min<size_t>((160 * 1024) / (lds + 512), (size_t)std::max(1, 16 / W));
        if (wg_per_cu < 1) wg_per_cu = 1;
        if (getenv("LDPC_HIP_DEBUG"))
          fprintf(stderr, "fused float: %d waves, %d checks/thread, %d variables/thread, variable-pass extra LDS cycles %ld -> %ld of %d group reads\n",
                  W, sc.rounds, vpt, vconf_before, vconf_after, ((sc.var_words + 63) / 64) * 4 * cmax);
        break;
      }
    }
    // Second choice, binary32 with 33 ... 48 slots per check: the wide-row kernel (q0 in registers, 15 bytes of
    // record per check, both index tables from L2).
    if (!use_reg && !F64 && env_reg && NW == 2 && RM <= 48 && m <= 4094 && cmax >= 1) {
      for (int W : {16, 12}) {
        if (env_w && W != env_w) continue;
        const int B = W * 64;
        LdsSchedule sc;
        if (ldpc_build_lds_schedule(c, B, RM, FF_NDUMMY, optimise != 0, sc)) continue;
        if (sc.words & 1) sc.words++;                                     // (8-byte alignment of the pairs)
        const int vpt = 2 * ((sc.var_words + 2 * B - 1) / (2 * B));
        FfKern k = ff_wide_lookup(sc.rounds, RM / 2, vpt, cmax, B);
        if (getenv("LDPC_HIP_DEBUG"))
          fprintf(stderr, "fused float (wide rows): shape {%d checks/thread, %d slot pairs, %d variables/thread, %d edges/variable, %d threads}, %d words: %s\n",
                  sc.rounds, RM / 2, vpt, cmax, B, sc.words, k ? "built" : "no kernel");
        if (!k) continue;
        hipFuncAttributes fa;
        HIPCHK(hipFuncGetAttributes(&fa, (const void *)k));
        // the kernel addresses its dynamic LDS from 0; a slot is a 15-bit word index / a 17-bit byte address; an
        // edge's sign word is found by a 14-bit byte offset
        const size_t need_lds = lds_bytes_wide(sc.words, m);
        if (fa.sharedSizeBytes != 0 || sc.words > 32767 || 6 * (size_t)(m + 1) > 16383 || need_lds > 160 * 1024) continue;
        if (optimise) ldpc_place_variable_pass_wide(c, sc, &vconf_before, &vconf_after);
        if (getenv("LDPC_HIP_DEBUG"))
          fprintf(stderr, "fused float (wide rows): variable-pass extra LDS cycles %ld -> %ld of %d group reads\n", vconf_before, vconf_after,
                  ((sc.var_words + 63) / 64) * 2 * cmax);
        const int R2 = RM / 2, PK = (R2 % 4 == 0) ? 4 : 2, RB = RM - 32;
        std::vector<uint32_t> cp((size_t)sc.rounds * R2 * B, 0u), mk((size_t)sc.rounds * B * 2, 0u);
        std::vector<int> slot_of_edge(c.E(), -1);
        for (int j = 0; j < m; j++) {
          const int r = j / B, t = j % B;
          for (int kk = 0; kk < RM; kk++) {
            const int w = sc.colT[((size_t)r * RM + kk) * B + t];
            if (w >= sc.dummy_base) continue;
            // the kernel's masks: slot k < 32 at bit 31-k of the first word, slot k >= 32 at bit RM-1-k of the second
            if (kk < 32) mk[((size_t)r * B + t) * 2] |= 1u << (31 - kk);
            else mk[((size_t)r * B + t) * 2 + 1] |= 1u << (RM - 1 - kk);
            for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++)
              if (sc.pos[c.col_idx[e]] == w) slot_of_edge[e] = kk;
          }
        }
        for (int r = 0; r < sc.rounds; r++)
          for (int k2 = 0; k2 < R2; k2++)
            for (int t = 0; t < B; t++) {
              const uint32_t lo = sc.colT[((size_t)r * RM + 2 * k2) * B + t], hi = sc.colT[((size_t)r * RM + 2 * k2 + 1) * B + t];
              cp[(((size_t)r * (R2 / PK) + k2 / PK) * B + t) * PK + k2 % PK] = (lo * 4u) | (hi << 17);
            }
        std::vector<int> inv(sc.var_words, -1);
        for (int v = 0; v < n; v++) inv[sc.pos[v]] = v;
        const int lpg = (2 * cmax + 3) / 4;                               // 16-byte loads per pair of variables
        std::vector<uint32_t> vw((size_t)(vpt / 2) * lpg * B * 4, (uint32_t)m | ((uint32_t)(4 * m) << 12));   // "no edge": record m, code 0
        bool lost = false;
        for (int t = 0; t < B; t++)
          for (int i = 0; i < vpt; i++) {
            const int w = t + B * i;
            if (w >= sc.var_words || inv[w] < 0) continue;
            const int v = inv[w];
            for (int kk = 0; kk < c.col_ptr[v + 1] - c.col_ptr[v]; kk++) {
              const int e = c.edge_of[c.col_ptr[v] + kk];
              const int s = slot_of_edge[e], j = row_of[e];
              if (s < 0) { lost = true; continue; }
              // sign word: slots < 32 in sgL[j]; the others in the 32-bit word that holds sgH[j] (high half for odd j)
              const uint32_t sa = s < 32 ? 4u * (uint32_t)j : 4u * (uint32_t)(m + 1) + 4u * (uint32_t)(j >> 1);
              const uint32_t code = s < 32 ? (uint32_t)s : (uint32_t)(s + 64 - RM - 16 * (j & 1));
              const int idx = (i & 1) * cmax + kk;                        // within the pair's 4 * lpg entries
              vw[(((size_t)(i / 2) * lpg + idx / 4) * B + t) * 4 + idx % 4] = (uint32_t)j | (sa << 12) | (code << 26);
            }
          }
        if (lost || RB > 16) continue;
        sched = sc;
        kern = k; use_reg = true; qg = false;
        block = B; rounds = sc.rounds;
        lds = need_lds;
        if ((rc = d_colP.ensure(cp.size() * 4))) return rc;
        if ((rc = d_vtabW.ensure(vw.size() * 4))) return rc;
        if ((rc = d_maskT.ensure(mk.size() * 4))) return rc;
        HIPCHK(hipMemcpy(d_colP.p, cp.data(), cp.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_vtabW.p, vw.data(), vw.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_maskT.p, mk.data(), mk.size() * 4, hipMemcpyHostToDevice));
        wg_per_cu = 1;
        break;
      }
    }
    if (!use_reg) {
    // generic kernel: up to 16 wavefronts, every wavefront with the same number of checks
    int W = std::min(16, need);
    rounds = (need + W - 1) / W;
    if (rounds > 8) { ldpc_set_error("fused min-sum: too many checks"); return LDPC_ERR_UNSUPPORTED; }
    W = (need + rounds - 1) / rounds;
    if (env_w >= 1 && env_w <= 16 && (need + env_w - 1) / env_w <= 8) { W = env_w; rounds = (need + W - 1) / W; }
    block = W * 64;
    if (ldpc_build_lds_schedule(c, block, RM, FF_NDUMMY, optimise != 0, sched)) {
      ldpc_set_error("LDS schedule: code too large");
      return LDPC_ERR_UNSUPPORTED;
    }
    rounds = sched.rounds;
    lds = lds_bytes(sched.words, sched.var_words, m, NW);
    qg = false;
    if (lds + 1024 > 160 * 1024) {
      // q0 in global memory, packed records (QG kernels)
      lds = lds_bytes_qg(sched.words, m, NW);
      qg = true;
    }
    if (lds + 256 > 160 * 1024) {
      ldpc_set_error("fused min-sum: frame does not fit in LDS");
      return LDPC_ERR_UNSUPPORTED;
    }
    wg_per_cu = (int)std::min<size_t>((160 * 1024) / (lds + 512), (size_t)std::max(1, 16 / W));
    if (wg_per_cu < 1) wg_per_cu = 1;
    }
    if (const char *e = getenv("LDPC_HIP_FUSED_WG_PER_CU")) {
      const int g = atoi(e);
      if (g >= 1 && (size_t)g * (lds + 512) <= 160 * 1024) wg_per_cu = g;
    }
    if (!use_reg)
    kern = qg ? (NW == 1 ? ff_pick<T, 1, true>(rounds) : ff_pick<T, 2, true>(rounds))
              : (NW == 1 ? ff_pick<T, 1>(rounds) : ff_pick<T, 2>(rounds));
    if (!kern) { ldpc_set_error("no fused min-sum kernel for this code shape"); return LDPC_ERR_UNSUPPORTED; }

    // which slots hold edges (bit k of word k / 32), and where each edge of each variable sits
    std::vector<uint32_t> realT((size_t)rounds * block * NW, 0u);
    std::vector<int> slot_of_edge(c.E(), -1);
    for (int j = 0; j < m; j++) {
      const int r = j / block, t = j % block;
      for (int k = 0; k < RM; k++) {
        const int w = sched.colT[((size_t)r * RM + k) * block + t];
        if (w >= sched.dummy_base) continue;
        realT[((size_t)r * block + t) * NW + k / 32] |= 1u << (k % 32);
        for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++)
          if (sched.pos[c.col_idx[e]] == w) slot_of_edge[e] = k;
      }
    }
    std::vector<uint32_t> vtab((size_t)std::max(1, cmax) * sched.var_words, (uint32_t)m << 8);
    for (int v = 0; v < n; v++)
      for (int kk = 0; kk < c.col_ptr[v + 1] - c.col_ptr[v]; kk++) {
        const int e = c.edge_of[c.col_ptr[v] + kk];
        if (slot_of_edge[e] < 0) { ldpc_set_error("fused min-sum: schedule lost an edge"); return LDPC_ERR_FORMAT; }
        vtab[(size_t)kk * sched.var_words + sched.pos[v]] = ((uint32_t)row_of[e] << 8) | (uint32_t)slot_of_edge[e];
      }
    if ((rc = d_colT.ensure(sched.colT.size() * 2))) return rc;
    if ((rc = d_realT.ensure(realT.size() * 4))) return rc;
    if ((rc = d_vtab.ensure(vtab.size() * 4))) return rc;
    if ((rc = d_pos.ensure(sched.pos.size() * 2))) return rc;
    if ((rc = d_counter.ensure(256))) return rc;
    if (qg && (rc = d_q0g.ensure((size_t)num_cu * wg_per_cu * sched.var_words * sizeof(T)))) return rc;
    HIPCHK(hipMemcpy(d_colT.p, sched.colT.data(), sched.colT.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_realT.p, realT.data(), realT.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_vtab.p, vtab.data(), vtab.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_pos.p, sched.pos.data(), sched.pos.size() * 2, hipMemcpyHostToDevice));
    {
      // (the attribute belongs to the kernel, not to this decoder: another decoder of a smaller
      // code that uses the same instantiation must not lower it, so it is always the maximum)
      hipFuncAttributes fa;
      HIPCHK(hipFuncGetAttributes(&fa, (const void *)kern));
      HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - (int)fa.sharedSizeBytes));
      if (lds + fa.sharedSizeBytes > 160 * 1024) {
        ldpc_set_error("fused min-sum: frame does not fit in LDS");
        return LDPC_ERR_UNSUPPORTED;
      }
    }
    return LDPC_OK;
  }

  int decode(int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter, uint8_t *bits,
             int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out, hipStream_t st) override {
    if (nframes <= 0) return LDPC_OK;
    if (!llr || max_iter < 0 || max_iter > 65000) {
      ldpc_set_error("nframes/max_iter out of range for this decoder");
      return LDPC_ERR_ARG;
    }
    if (dt != LDPC_DT_F64 && dt != LDPC_DT_F32) {
      ldpc_set_error("floating-point min-sum takes F64 or F32 input");
      return LDPC_ERR_ARG;
    }
    if (llr_out && !opts.want_llr) {
      ldpc_set_error("llr_out needs opts.want_llr");
      return LDPC_ERR_ARG;
    }
    int *counter = ws_counter ? ws_counter : d_counter.as<int>();
    HIPCHK(hipMemsetAsync(counter, 0, 4, st));
    FfArgs a;
    a.colT = d_colT.as<uint16_t>();
    a.realT = d_realT.as<uint32_t>();
    a.vtab = d_vtab.as<uint32_t>();
    a.pos = d_pos.as<uint16_t>();
    a.n = n; a.m = m; a.RM = RM; a.rounds = rounds; a.cmax = cmax;
    a.var_words = sched.var_words; a.words = sched.words; a.dummy_base = sched.dummy_base;
    a.nframes = nframes; a.max_iter = max_iter;
    a.llr = llr; a.dt = dt; a.synd = sy;
    a.bits = bits; a.iters = iters_out; a.sout = sout; a.status = status; a.llr_out = llr_out;
    a.counter = counter;
    a.q0g = ws_q0g ? ws_q0g : d_q0g.p;
    a.vtabT = d_vtabT.as<uint16_t>();
    a.maskT = d_maskT.as<uint32_t>();
    a.colP = d_colP.as<uint32_t>();
    a.vtabW = d_vtabW.as<uint32_t>();
    long grid = (long)num_cu * wg_per_cu;
    if (grid > nframes) grid = nframes;
    a.chunk = (int)std::max<long>(1, std::min<long>(std::min<long>(4096 / n, nframes / (grid * 16)), 16));
    if (const char *e = getenv("LDPC_HIP_FUSED_CHUNK")) a.chunk = std::max(1, atoi(e));
    prof_cur_frames = nframes;
    prof_begin(K_FUSED, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(block), lds, st, a);
    LAUNCHCHK("fused_ms_kernel");
    prof_end(st);
    last_iterations = max_iter;
    return LDPC_OK;
  }

  // the kernel's only device scratch is the frame queue head: one word of ws
  int *ws_counter = nullptr;
  int decode_ws(void *ws, size_t ws_bytes, int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter,
                uint8_t *bits, int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out,
                hipStream_t st) override {
    if (!ws || ws_bytes < 256) { ldpc_set_error("scratch too small for the frame queue head"); return LDPC_ERR_ARG; }
    if (qg) {
      // one placed copy of q0 per workgroup in flight
      long grid = (long)num_cu * wg_per_cu;
      if (grid > nframes) grid = nframes;
      const size_t need = 256 + (size_t)grid * sched.var_words * sizeof(T);
      if (need > ws_bytes) {
        ldpc_set_error("allocation-free decode: " + std::to_string(nframes) + " frames of this code need " +
                       std::to_string(need) + " bytes of channel-LLR workspace, the static scratch holds " +
                       std::to_string(ws_bytes) + " (use ldpc_decode_batch_device, or fewer frames per call)");
        return LDPC_ERR_ARG;
      }
      ws_q0g = (char *)ws + 256;
    }
    ws_counter = (int *)ws;
    const int rc = decode(nframes, llr, dt, sy, max_iter, bits, iters_out, sout, status, llr_out, st);
    ws_counter = nullptr;
    ws_q0g = nullptr;
    return rc;
  }
};

}  // namespace
}  // namespace ldpc

// prec: 4 binary32, 8 binary64.  nullptr: the code does not fit (the caller takes the two-pass engine)
Engine *ldpc_make_fused_float_engine(int prec, const ldpc_code &c) {
  if (prec == 4) return ldpc::FusedFloatEngine<float>::fits(c) ? new ldpc::FusedFloatEngine<float> : nullptr;
  return ldpc::FusedFloatEngine<double>::fits(c) ? new ldpc::FusedFloatEngine<double> : nullptr;
}
