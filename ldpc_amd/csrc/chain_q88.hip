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
//   * Of the compressed messages of a thread's checks, the signs and the slot of min1 (previous
//     and current iteration, one register per check: R = 32 of the 128 registers a wave of a
//     1024-thread workgroup owns) stay in VGPRs; min1 | min2 live in two per-workgroup row sets in
//     global memory (L2 / Infinity Cache) that swap roles every iteration, read a check ahead
//     (with them in registers as well the R = 32 kernel spilled 262 registers, and every reload
//     was a full memory wait); the edge -> LDS word table (one 16-byte row per check) is re-read
//     from L2 in both phases; the channel values of the staircase variables are parked in a
//     per-workgroup global row (int16) and re-read once per check.
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
#include "lds_schedule.h"

namespace ldpc {

struct ChainArgs {
  const fq_u4 *tab;          // [R][block] one row per check: 6 LDS words (u16), spare, real-slot mask | flags << 8
  const uint16_t *chainvar;  // [R][block] staircase variable right of check r (0xFFFF: none)
  const uint16_t *pos;       // [n] variable -> LDS word, or 0x8000 | j: the staircase variable right of check j
  const uint32_t *wextra;    // [block / 64][3] per wavefront, bit r: some check r has slot 7 / lacks its right variable; bit 0: some check 0 lacks its left variable
  short *q0s;                // [grid][R][block] channel values of the staircase variables (right of check r)
  uint32_t *nms;             // [grid][2][R][block] min1 | min2 << 16 per check: two row sets per workgroup that swap roles every
                             // iteration (the previous iteration's minima / the current one's)
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
  unsigned long long *timing;   // [16 wavefronts][8] cycle sums per stage, or null (LDPC_HIP_FUSED_TIMING)
};

#define CQ_BIGF 16777216.0f /* 2^24 */
#ifndef CQ_ABLATE
#define CQ_ABLATE 0
#endif
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
// bit r of a wave-uniform word, tested where it is used: computed ahead of the unrolled loops (as
// the compiler does when left alone) the 32 x 3 answers become 64-bit lane masks in spilled scalar
// registers, and every test two v_readlane
__device__ __forceinline__ bool cq_bit(unsigned word, const int r) {
  asm volatile("" : "+s"(word));
  return (word >> r) & 1u;
}
__device__ __forceinline__ unsigned cq_u(float x) { return __float_as_uint(x); }

// LDS byte address of one of the two 16-bit word indices packed in w: one SDWA shift
template <int HI>
__device__ __forceinline__ unsigned cq_addr(unsigned w) {
  unsigned r;
  if (HI) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(w));
  else asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(w));
  return r;
}
// (the dynamic LDS array starts at LDS address 0: the kernel has no static LDS, which the host checks)
__device__ __forceinline__ lds_int *cq_lds(unsigned byte_addr) { return (lds_int *)(unsigned long)byte_addr; }

// as: byte 2 <- the low byte of signs, byte 3 <- slot & 7, the low half kept (two SDWA writes
// instead of mask, shift, and-or, shift-or)
__device__ __forceinline__ unsigned cq_set_new(unsigned as, const unsigned signs, const unsigned key) {
  asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(as) : "v"(signs));
  asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(as) : "v"(key), "v"(7u));
  return as;
}

// The per-check rows (table, channel values, new minima) are [R][block] arrays read at row r, column
// tid: buffer instructions take the uniform base as a descriptor, the row as a scalar byte offset and
// the column as one 32-bit lane offset, so no row costs 64-bit vector address arithmetic or a pair of
// address registers (global loads: two vector adds per row and access).  Reads past `bytes` return 0.
typedef __amdgpu_buffer_rsrc_t cq_rsrc;
__device__ __forceinline__ cq_rsrc cq_make_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000 /* raw dwords, gfx9 */);
}
__device__ __forceinline__ fq_u4 cq_ld128(cq_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(fq_u4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ unsigned cq_ld32(cq_rsrc r, unsigned voff, unsigned soff) {
  return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ int cq_ld16s(cq_rsrc r, unsigned voff, unsigned soff) {
  return (int)(short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void cq_st32(unsigned v, cq_rsrc r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, (int)soff, 0);
}

// BLK: workgroup size when it is known at compile time (row offsets become constants), else 0
template <int R, int BLK>
__global__ __launch_bounds__(1024) void chain_q88_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) int fsum[];   // [words] sums + dummies, frame id + flags, exchange rows
  const int tid = threadIdx.x, B = BLK ? BLK : (int)blockDim.x;
  int *frame_slot = fsum + a.words;            // [4]
  float *xR = (float *)(frame_slot + 4);       // [B + 2]  xR[i + 1]: message of thread i's last check to its right variable
  float *xL = xR + B + 2;                      // [B + 2]  xL[i]: message of thread i's first check to its left variable
  unsigned *cbits = (unsigned *)(xL + B + 2);  // [B]      hard decisions of thread i's staircase variables (outputs)
  // wave-uniform: bit r set if some check r of this wavefront has a slot 7 / lacks its right
  // variable; bit 0 of the third word: some check 0 lacks its left variable
  const unsigned wex = __builtin_amdgcn_readfirstlane(a.wextra[3 * (tid >> 6)]);
  const unsigned wabs = __builtin_amdgcn_readfirstlane(a.wextra[3 * (tid >> 6) + 1]);
  const unsigned wabsl = __builtin_amdgcn_readfirstlane(a.wextra[3 * (tid >> 6) + 2]);
  // (uniform bases for the loops below)
  short *q0base = a.q0s + (size_t)blockIdx.x * R * B;
  uint32_t *nmbase = a.nms + (size_t)blockIdx.x * 2 * R * B;
  const cq_rsrc rtab = cq_make_rsrc(a.tab, (unsigned)R * B * 16u);
  const cq_rsrc rq0 = cq_make_rsrc(q0base, (unsigned)R * B * 2u);
  // (the two sets of minima rows: set k of iteration t's new minima is t & 1)
  auto nm_set = [&](int k) { return cq_make_rsrc(nmbase + (size_t)(k & 1) * R * B, (unsigned)R * B * 4u); };

  // compressed messages: min1 | min2 << 16 in the global row sets; as = signs | slot of min1 << 8,
  // previous iteration in the low half, current in the high half (signs: slot k in bit 7 - k)
  unsigned as[R];

  const bool vec_in = a.dt == 2 && (a.n & 7) == 0 && ((size_t)a.llr & 15) == 0;
  const bool vec_out = a.bits && (a.n & 7) == 0 && ((size_t)a.bits & 7) == 0;
  const fq_u4 *pos8 = (const fq_u4 *)a.pos;

  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (tid == 0) *frame_slot = atomicAdd(a.counter, 1);
  __syncthreads();
  for (;;) {
    const int frame = *frame_slot;
    if (frame >= a.nframes) break;
    long long tk = a.timing ? clock64() : 0;
    auto tick = [&](int i) {                     // (wave-uniform; a handful of scalar instructions per stage)
      if (a.timing) { const long long now = clock64(); tacc[i] += now - tk; tk = now; }
    };
    int next_frame = 0;
    if (tid == 0) next_frame = atomicAdd(a.counter, 1);
    const size_t base = (size_t)frame * a.n;
    {
      // q0 of variable with table entry p: an LDS sum, or (bit 15) the right variable of check
      // p & 0x7fff, parked in the row of the thread that owns that check
      auto put = [&](const unsigned p, const int q) {
        if (p & 0x8000u) {
          const unsigned j = p & 0x7fffu;
          q0base[(size_t)(j % R) * B + j / R] = (short)q;
        } else fsum[p] = FQ_BIAS + q;
      };
      if (vec_in) {
        const fq_u4 *p8 = (const fq_u4 *)((const short *)a.llr + base);
        for (int i = tid; i < (a.n >> 3); i += B) {
          const fq_u4 v = __builtin_nontemporal_load(p8 + i);
          const fq_u4 w = pos8[i];
          put(w.x & 0xffffu, (int)(short)(v.x & 0xffffu)); put(w.x >> 16, (int)v.x >> 16);
          put(w.y & 0xffffu, (int)(short)(v.y & 0xffffu)); put(w.y >> 16, (int)v.y >> 16);
          put(w.z & 0xffffu, (int)(short)(v.z & 0xffffu)); put(w.z >> 16, (int)v.z >> 16);
          put(w.w & 0xffffu, (int)(short)(v.w & 0xffffu)); put(w.w >> 16, (int)v.w >> 16);
        }
      } else if (a.dt == 2) {
        const short *p = (const short *)a.llr + base;
        for (int i = tid; i < a.n; i += B) put(a.pos[i], p[i]);
      } else if (a.dt == 3) {
        const int *p = (const int *)a.llr + base;
        for (int i = tid; i < a.n; i += B) {
          const int v = p[i];
          put(a.pos[i], v > FQ_SAT ? FQ_SAT : (v < -32768 ? -32768 : v));
        }
      } else if (a.dt == 1) {                      // float_to_fix for real-valued inputs, as bsc() :117-120
        const float *p = (const float *)a.llr + base;
        for (int i = tid; i < a.n; i += B) put(a.pos[i], fq_from_real((double)p[i]));
      } else {
        const double *p = (const double *)a.llr + base;
        for (int i = tid; i < a.n; i += B) put(a.pos[i], fq_from_real(p[i]));
      }
    }
    for (int i = a.dummy_base + tid; i < a.words; i += B) fsum[i] = FQ_BIG;
    unsigned sbits = 0, parbits = 0, cmask = 0;   // cmask: hard decisions of the right variables, from the last check pass
#pragma unroll
    for (int r = 0; r < R; r++) {
      as[r] = 0;
      cq_st32(0u, nm_set(0), (unsigned)tid * 4u, (unsigned)r * B * 4u);   // "previous" minima of iteration 1: all +0
      const int c = tid * R + r;
      // (per-check bit words -- syndrome, parity, hard decisions -- hold check r in bit R - 1 - r:
      // the accumulators of the check pass shift left, one v_alignbit per check)
      if (a.synd && c < a.m) sbits |= (unsigned)(a.synd[(size_t)frame * a.m + c] & 1) << (R - 1 - r);
    }
    if (tid == 0) { frame_slot[1] = 0; frame_slot[2] = 0; }
    for (int i = tid; i < 2 * (B + 2); i += B) xR[i] = 0.0f;
    tick(0);
    __syncthreads();
    tick(1);

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
    // channel value of the variable left of the thread's first check: the previous thread's last row
    auto left_q0 = [&]() { return (float)cq_ld16s(rq0, (unsigned)(tid ? tid - 1 : 0) * 2u, (unsigned)(R - 1) * B * 2u); };

    // Phase A: every check of the thread on the sums of the previous iteration.  PO: only the
    // parity of the hard decisions (the verdict after the last iteration).
    auto check_pass = [&](auto po_c, const cq_rsrc rold, const cq_rsrc rnew) {
      constexpr bool PO = decltype(po_c)::value;
      // (an opaque lane offset: the table rows and channel values are loop-invariant, and hoisted out
      // of the iteration loop they would take a hundred registers)
      unsigned t16 = (unsigned)tid * 16u;
      asm volatile("" : "+v"(t16));
      const unsigned t2 = t16 >> 3, t4 = t16 >> 2;
      Open c, nx;
      unsigned omn = cq_ld32(rold, t4, 0);
      fq_u4 w = cq_ld128(rtab, t16, 0);
      int q0w = cq_ld16s(rq0, t2, 0);
      open(omn, as[0], c);
      if (R > 1) omn = cq_ld32(rold, t4, (unsigned)B * 4u);      // (of check r + 1, a check ahead of its use)
      // vl: message of the variable left of the check in turn (q0 + the message of the check on its
      // other side), Sl: its posterior
      float vl = left_q0() + xR[tid];
      unsigned pacc = 0, cacc = 0;
      if (wabsl && ((w.w >> 24) & 1u)) vl = CQ_BIGF;
      float Sl = vl + c.cl;
#pragma unroll
      for (int r = 0; r < R; r++) {
        const bool ex = cq_bit(wex, r);           // (wave-uniform)
        int Sw[6];
        Sw[0] = *cq_lds(cq_addr<0>(w.x)); Sw[1] = *cq_lds(cq_addr<1>(w.x));
        Sw[2] = *cq_lds(cq_addr<0>(w.y)); Sw[3] = *cq_lds(cq_addr<1>(w.y));
        Sw[4] = *cq_lds(cq_addr<0>(w.z));
        Sw[5] = FQ_BIAS;                            // (reads as +0.0 below: no sign, no parity)
        if (ex) Sw[5] = *cq_lds(cq_addr<1>(w.z));
        const unsigned wflags = w.w;
        const float q0r = (float)q0w;
        // the next check's row and channel value while this one is worked on
        if (r + 1 < R) {
          w = cq_ld128(rtab, t16, (unsigned)(r + 1) * B * 16u);
          q0w = cq_ld16s(rq0, t2, (unsigned)(r + 1) * B * 2u);
        }
        float cln;
        if (r + 1 < R) {
          open(omn, as[r + 1], nx);
          cln = nx.cl;
          if (r + 2 < R) omn = cq_ld32(rold, t4, (unsigned)(r + 2) * B * 4u);
        } else cln = xL[tid + 1];
        // the variable right of the check: its message to this check, its posterior, its message to
        // the next check (integer-valued floats: exact in any order)
        float vr = q0r + cln, Sr = vr + c.cr, vln = q0r + c.cr, Sln = Sr;
        if (cq_bit(wabs, r)) {                    // (rare: wave-uniform test first)
          if ((wflags >> 26) & 1u) { vr = q0r; Sr = vln; vln = Sln = CQ_BIGF; }   // private variable: no second check
          if ((wflags >> 25) & 1u) vr = Sr = vln = Sln = CQ_BIGF;                 // no variable
        }
        float Sf[6];
#pragma unroll
        for (int j = 0; j < 6; j++) Sf[j] = cq_f((unsigned)Sw[j]) - FQ_BIASF;
        unsigned pp = __builtin_amdgcn_bitop3_b32(cq_u(Sl), cq_u(Sr), cq_u(Sf[0]), 0x96);
        pp = __builtin_amdgcn_bitop3_b32(pp, cq_u(Sf[1]), cq_u(Sf[2]), 0x96);
        pp = __builtin_amdgcn_bitop3_b32(pp, cq_u(Sf[3]), cq_u(Sf[4]), 0x96);
        pp ^= cq_u(Sf[5]);
        // parity of check r in bit R - 1 - r after R steps
        pacc = __builtin_amdgcn_alignbit(pacc, pp, 31);
        cacc = __builtin_amdgcn_alignbit(cacc, cq_u(Sr), 31);   // (the pass that turns out to be the last one supplies the outputs)
        if constexpr (!PO) {
          unsigned k1 = 0xffffffffu, k2 = 0xffffffffu, sg = c.sgt;
          auto slot = [&](const float q, const int k) {
            sg = __builtin_amdgcn_alignbit(sg, cq_u(q), 31);
            const unsigned key = FQ_ANDOR(cq_u(q), 0x7fffffffu, (unsigned)k);
            k2 = fq_med3u(k1, key, k2);
            k1 = min(k1, key);
          };
          slot(vl, 0);
          slot(vr, 1);
#pragma unroll
          for (int j = 0; j < 6; j++) {
            if (j == 5 && !ex) { sg <<= 1; break; }
            const float mo = cq_pick(c.m1, c.m2, cq_in_slot(c.ab, 2 + j));
            slot(Sf[j] - cq_f(FQ_ANDOR(sg, 0x80000000u, cq_u(mo))), 2 + j);
          }
          // sgnprod = s ^ XOR(signs); c2v_k < 0 iff sgnprod != sign_k (:278-283)
          const unsigned tot = (unsigned)__popc(sg) + (sbits >> (R - 1 - r));
          const unsigned flip = (unsigned)(-(int)(tot & 1u));
          const unsigned allm = (wflags >> 16) & 0xffu;
          const unsigned nsg = __builtin_amdgcn_bitop3_b32(sg, allm, flip, 0x78);   // sg ^ (allm & flip)
          // min over saturated magnitudes = saturated min (:43-53, :269-277)
          // (non-negative floats order like their bit patterns: the clamp is an integer minimum)
          // (the slot number in a key's low bits is a fraction of the values below the cap -- integers
          // under 2^15 -- and the conversion truncates it)
          const float n1 = cq_f(min(k1, 0x46fe0000u /* 127.0 * 256 */));
          const float n2 = cq_f(min(k2, 0x46fe0000u));
          cq_st32((unsigned)n1 | ((unsigned)n2 << 16), rnew, t4, (unsigned)r * B * 4u);
          as[r] = cq_set_new(as[r], nsg, k1);
          // materialise the new state here: left alone the compiler sinks this arithmetic of all R
          // checks below the loop and keeps every check's signs and table row alive until then
          asm volatile("" : "+v"(as[r]));
        }
        Sl = Sln;
        vl = vln;
        if (r + 1 < R) c = nx;
        asm volatile("" : "+v"(pacc), "+v"(cacc), "+v"(Sl), "+v"(vl));
        __builtin_amdgcn_sched_barrier(0);
      }
      parbits = pacc;
      cmask = cacc;
    };

    int it_done = a.max_iter, ok = 0, t_last = 1;
    for (int t = 1; t <= a.max_iter + 1; t++) {
      const cq_rsrc rold = nm_set(t - 1), rnew = nm_set(t);
      t_last = t;
      if (t == a.max_iter + 1) check_pass(std::true_type{}, rold, rnew);
      else check_pass(std::false_type{}, rold, rnew);
      tick(2);
      const int any_bad = fq_block_or((int)(parbits ^ sbits), frame_slot + 1, t);
      tick(3);
      if (!any_bad) { it_done = t - 1; ok = 1; break; }       // H*bits == s after iteration t-1
      if (t == a.max_iter + 1) break;
      for (int i = a.dummy_base + tid; i < a.words; i += B) fsum[i] = FQ_BIG;   // re-arm (see fused_kernels.hip.h)
      // ---- phase B: the new messages replace the old ones in the LDS sums
      unsigned t16 = (unsigned)tid * 16u;
      asm volatile("" : "+v"(t16));
      const unsigned t4 = t16 >> 2;
      fq_u4 w = cq_ld128(rtab, t16, 0);
      unsigned nmr = cq_ld32(rnew, t4, 0), omr = cq_ld32(rold, t4, 0);
#pragma unroll
      for (int r = 0; r < R; r++) {
        const bool ex = cq_bit(wex, r);
        const unsigned s = as[r];
        const float o1 = (float)(omr & 0xffffu), o2 = (float)(omr >> 16);
        const float n1 = (float)(nmr & 0xffffu), n2 = (float)(nmr >> 16);
        unsigned long long abo[3], abn[3];
#pragma unroll
        for (int b = 0; b < 3; b++) { abo[b] = __ballot((s >> (8 + b)) & 1u); abn[b] = __ballot((s >> (24 + b)) & 1u); }
        unsigned orot = s << 26, nrot = (s >> 16) << 26;     // sign of slot 2 in bit 31
        const unsigned adr[6] = {cq_addr<0>(w.x), cq_addr<1>(w.x), cq_addr<0>(w.y), cq_addr<1>(w.y), cq_addr<0>(w.z), cq_addr<1>(w.z)};
        if (r + 1 < R) {
          w = cq_ld128(rtab, t16, (unsigned)(r + 1) * B * 16u);
          nmr = cq_ld32(rnew, t4, (unsigned)(r + 1) * B * 4u);
          omr = cq_ld32(rold, t4, (unsigned)(r + 1) * B * 4u);
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
          if (j == 5 && !ex) break;
          const float cn = cq_f(FQ_ANDOR(nrot, 0x80000000u, cq_u(cq_pick(n1, n2, cq_in_slot(abn, 2 + j)))));
          const float co = cq_f(FQ_ANDOR(orot, 0x80000000u, cq_u(cq_pick(o1, o2, cq_in_slot(abo, 2 + j)))));
          nrot = fq_shl1(nrot);
          orot = fq_shl1(orot);
          const int delta = (int)(cn - co);
#if CQ_ABLATE & 1   /* timing experiments only (results wrong): the fold without its LDS adds */
          asm volatile("" ::"v"(delta), "v"(adr[j]));
#else
          __hip_atomic_fetch_add(cq_lds(adr[j]), delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        }
        if (r == 0)
          xL[tid] = cq_f(FQ_ANDOR((s >> 16) << 24, 0x80000000u, cq_u(cq_pick(n1, n2, cq_in_slot(abn, 0)))));
        if (r == R - 1)
          xR[tid + 1] = cq_f(FQ_ANDOR((s >> 16) << 25, 0x80000000u, cq_u(cq_pick(n1, n2, cq_in_slot(abn, 1)))));
        as[r] = s >> 16;
        asm volatile("" : "+v"(as[r]));
        __builtin_amdgcn_sched_barrier(0);
      }
      tick(4);
      __syncthreads();
      tick(5);
    }

    // ---- outputs of this frame
    if (a.bits || a.llr_out) {
      // staircase variables: the signs the last check pass saw (posterior = q0 + the messages of the
      // two checks); their values only when the caller wants posteriors
      if (a.llr_out) {
        // (the messages in effect are the "previous" ones of the last check pass)
        const cq_rsrc rold = nm_set(t_last - 1);
        Open c, nx;
        open(cq_ld32(rold, (unsigned)tid * 4u, 0), as[0], c);
#pragma unroll
        for (int r = 0; r < R; r++) {
          float cln;
          if (r + 1 < R) { open(cq_ld32(rold, (unsigned)tid * 4u, (unsigned)(r + 1) * B * 4u), as[r + 1], nx); cln = nx.cl; }
          else cln = xL[tid + 1];
          if ((wabs >> r) & 1u) {
            const fq_u4 wr = (a.tab + (size_t)r * B)[tid];
            if ((wr.w >> 26) & 1u) cln = 0.0f;
          }
          const unsigned v = a.chainvar[(size_t)r * B + tid];
          if (v != 0xffffu) {
            int sv = (int)((float)cq_ld16s(rq0, (unsigned)tid * 2u, (unsigned)r * B * 2u) + c.cr + cln);
            sv = sv > FQ_SAT ? FQ_SAT : (sv < -32768 ? -32768 : sv);      // float_to_fix(sum), :303
            a.llr_out[base + v] = (float)sv * (1.0f / 256.0f);
          }
          if (r + 1 < R) c = nx;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      cbits[tid] = cmask;
      __syncthreads();
      // hard decision of the variable with table entry p: one LDS read either way
      const unsigned cb0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned *)cbits;
      auto hard = [&](const unsigned p) -> unsigned {
        const unsigned j = p & 0x7fffu;
        const bool ch = (p & 0x8000u) != 0;
        const unsigned val = (unsigned)*cq_lds(ch ? cb0 + (j / R) * 4u : j * 4u);
        return ch ? (val >> (R - 1 - j % R)) & 1u : (val - (unsigned)FQ_BIAS) >> 31;
      };
      if (vec_out) {
        fq_u2 *o8 = (fq_u2 *)(a.bits + base);
        for (int i = tid; i < (a.n >> 3); i += B) {
          const fq_u4 w = pos8[i];
          fq_u2 b;
          b.x = hard(w.x & 0xffffu) | (hard(w.x >> 16) << 8) | (hard(w.y & 0xffffu) << 16) | (hard(w.y >> 16) << 24);
          b.y = hard(w.z & 0xffffu) | (hard(w.z >> 16) << 8) | (hard(w.w & 0xffffu) << 16) | (hard(w.w >> 16) << 24);
          __builtin_nontemporal_store(b, o8 + i);
        }
      } else if (a.bits) {
        for (int i = tid; i < a.n; i += B) a.bits[base + i] = (uint8_t)hard(a.pos[i]);
      }
      if (a.llr_out) {
        for (int i = tid; i < a.n; i += B) {
          const unsigned p = a.pos[i];
          if (p & 0x8000u) continue;
          int sv = fsum[p] - FQ_BIAS;
          sv = sv > FQ_SAT ? FQ_SAT : (sv < -32768 ? -32768 : sv);
          a.llr_out[base + i] = (float)sv * (1.0f / 256.0f);
        }
      }
    }
    if (a.sout) {
#pragma unroll
      for (int r = 0; r < R; r++) {
        const int cc = tid * R + r;
        if (cc < a.m) a.sout[(size_t)frame * a.m + cc] = (uint8_t)((parbits >> (R - 1 - r)) & 1u);
      }
    }
    if (tid == 0) {
      if (a.iters) a.iters[frame] = it_done;
      if (a.status) a.status[frame] = ok ? 0 : 1;
    }
    tick(6);
    __syncthreads();                           // everyone has read this frame's number, sums and exchange rows
    if (tid == 0) *frame_slot = next_frame;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (see fused_q88_reg_kernel)
    __syncthreads();
    tick(7);
  }
  if (a.timing && (tid & 63) == 0)
    for (int i = 0; i < 8; i++) atomicAdd(a.timing + (tid >> 6) * 8 + i, (unsigned long long)tacc[i]);
}

}  // namespace ldpc

namespace {
using namespace ldpc;

// Host side: finds the staircase, lays the other variables out in LDS and builds the tables.
struct ChainQ88Engine : Engine {
  int n = 0, m = 0, R = 0, block = 0, num_cu = 256, words = 0, dummy_base = 0;
  size_t lds = 0;
  DevBuf d_tab, d_chainvar, d_pos, d_wextra, d_q0s, d_nms, d_counter, d_timing;
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
    // No staircase, no point: without weight-2 variables between consecutive checks this kernel is one
    // workgroup of m / R threads per CU walking LDS sums only, which the two-pass engine beats.  At least
    // half of the checks must hand a variable to their successor (DVB-S2: all but the last).
    {
      int nstairs = 0;
      for (int j = 0; j < m; j++) nstairs += right[j] >= 0;
      if (2 * nstairs < m) {
        ldpc_set_error("staircase Q8.8 decoder: the code has no staircase (" + std::to_string(nstairs) + " of " +
                       std::to_string(m) + " checks share a weight-2 variable with the next check)");
        return LDPC_ERR_UNSUPPORTED;
      }
    }
    // a check without a right staircase variable may park one weight-1 variable there instead
    // ("private": posterior = q0 + the check's own message; DVB-S2's last parity bit)
    std::vector<char> priv(m, 0);
    for (int j = 0; j < m; j++) {
      if (right[j] >= 0) continue;
      for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++) {
        const int v = c.col_idx[e];
        if (c.col_ptr[v + 1] - c.col_ptr[v] == 1) { right[j] = v; priv[j] = 1; is_chain[v] = 1; break; }
      }
    }
    // The other variables get LDS words.  Where, and in which of its slots a check visits them, is
    // the bank-conflict schedule of the fused decoder (lds_schedule.cpp), run on the graph this
    // kernel's LDS accesses see: check t*R + r is lane t of step r, staircase edges removed.
    std::vector<int> rank(n, -1);
    int nl = 0;
    for (int v = 0; v < n; v++)
      if (!is_chain[v]) rank[v] = nl++;
    ldpc_code vc;
    vc.n = nl; vc.m = R * block; vc.q = 2;
    vc.row_ptr.assign((size_t)vc.m + 1, 0);
    int most = 0;
    for (int r = 0; r < R; r++)
      for (int t = 0; t < block; t++) {
        const int j = t * R + r, jv = r * block + t;
        if (j < m)
          for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++)
            if (rank[c.col_idx[e]] >= 0) vc.col_idx.push_back(rank[c.col_idx[e]]);
        vc.row_ptr[jv + 1] = (int32_t)vc.col_idx.size();
        most = std::max(most, vc.row_ptr[jv + 1] - vc.row_ptr[jv]);
      }
    if (most > 6) { ldpc_set_error("staircase Q8.8 decoder: more than six variables of a check outside the staircase"); return LDPC_ERR_UNSUPPORTED; }
    const int RM = most <= 5 ? 5 : 6;
    LdsSchedule sc;
    int optimise = 1;
    if (const char *e = getenv("LDPC_HIP_FUSED_PLACE")) optimise = atoi(e);
    if (ldpc_build_lds_schedule(vc, block, RM, FQ_NDUMMY, optimise != 0, sc)) {
      ldpc_set_error("staircase Q8.8 decoder: the variables outside the staircase do not fit in LDS");
      return LDPC_ERR_UNSUPPORTED;
    }
    std::vector<uint16_t> pos(n, 0xffff);
    for (int v = 0; v < n; v++)
      if (rank[v] >= 0) pos[v] = sc.pos[rank[v]];
    for (int j = 0; j < m; j++)
      if (right[j] >= 0) pos[right[j]] = (uint16_t)(0x8000 | j);
    dummy_base = sc.dummy_base;
    words = sc.words;
    lds = (size_t)(words + 4 + 2 * (block + 2) + block) * 4;
    if (lds > 160 * 1024 || words > 0x8000) {
      ldpc_set_error("staircase Q8.8 decoder: the variables outside the staircase do not fit in LDS");
      return LDPC_ERR_UNSUPPORTED;
    }
    std::vector<uint16_t> tab((size_t)R * block * 8, 0), chainvar((size_t)R * block, 0xffff);
    std::vector<uint32_t> wextra((size_t)block / 64 * 3, 0);
    for (int t = 0; t < block; t++)
      for (int r = 0; r < R; r++) {
        const int j = t * R + r;
        uint16_t *row = &tab[((size_t)r * block + t) * 8];
        unsigned allm = 0, flags = 0;
        const int lv = (j >= 1 && j < m && !priv[j - 1]) ? right[j - 1] : -1, rv = j < m ? right[j] : -1;
        if (lv >= 0) allm |= 0x80; else flags |= 1;
        if (rv >= 0) allm |= 0x40; else flags |= 2;
        if (rv >= 0 && priv[j]) flags |= 4;
        chainvar[(size_t)r * block + t] = rv >= 0 ? (uint16_t)rv : 0xffff;
        if (flags & 6) wextra[t / 64 * 3 + 1] |= 1u << r;
        if (r == 0 && lv < 0) wextra[t / 64 * 3 + 2] |= 1u;
        for (int k = 0; k < 6; k++) {
          if (k >= RM) { row[k] = (uint16_t)dummy_base; continue; }
          row[k] = sc.colT[((size_t)r * RM + k) * block + t];
          if (sc.maskT[(size_t)r * block + t] & (1u << (RM - 1 - k))) {
            allm |= 0x20u >> k;
            if (k == 5) wextra[t / 64 * 3] |= 1u << r;
          }
        }
        row[6] = 0;
        row[7] = (uint16_t)(allm | (flags << 8));
      }
    const long pads = (long)sc.max_dummy_load * FQ_NDUMMY;
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
    if (fa.sharedSizeBytes != 0 || lds > 160 * 1024) {   // (the kernel addresses its dynamic LDS from 0)
      ldpc_set_error("staircase Q8.8 decoder: frame does not fit in LDS");
      return LDPC_ERR_UNSUPPORTED;
    }
    HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes));
    if (getenv("LDPC_HIP_VERBOSE")) {
      int nchain = 0;
      for (int v = 0; v < n; v++) nchain += is_chain[v];
      fprintf(stderr, "ldpc_hip: staircase Q8.8 decoder: %d threads x %d checks, %d staircase variables in registers, %d in LDS (%d slots per check, %ld extra LDS cycles in %ld half-wave accesses), %zu B LDS, %d VGPRs, %zu B spilled\n",
              block, R, nchain, nl, RM, sc.conflicts, sc.accesses, lds, fa.numRegs, (size_t)fa.localSizeBytes);
    }
    return LDPC_OK;
  }

  size_t q0s_bytes(long grid) const { return (size_t)grid * R * block * sizeof(short); }

  size_t nms_bytes(long grid) const { return (size_t)grid * 2 * R * block * sizeof(uint32_t); }

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
    a.timing = nullptr;
    // (cycle counters of timing builds: they allocate and synchronise, so never inside the allocation-free call)
    const bool timing = getenv("LDPC_HIP_FUSED_TIMING") != nullptr && counter == d_counter.as<int>();
    if (timing) {
      if (int rc = d_timing.ensure(16 * 8 * sizeof(unsigned long long))) return rc;
      HIPCHK(hipMemsetAsync(d_timing.p, 0, 16 * 8 * sizeof(unsigned long long), st));
      a.timing = d_timing.as<unsigned long long>();
    }
    prof_cur_frames = nframes;
    prof_begin(K_FUSED, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(block), lds, st, a);
    LAUNCHCHK("chain_q88_kernel");
    prof_end(st);
    last_iterations = max_iter;
    if (timing) {
      unsigned long long h[16 * 8];
      HIPCHK(hipStreamSynchronize(st));
      HIPCHK(hipMemcpy(h, d_timing.p, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "ldpc_hip: staircase kernel, cycles per wavefront summed over %ld workgroups: stage-in | barrier | check pass | barrier | fold | barrier | outputs | frame hand-over\n", grid);
      for (int w = 0; w < block / 64; w++) {
        fprintf(stderr, "  wave %2d:", w);
        for (int i = 0; i < 8; i++) fprintf(stderr, " %12llu", h[w * 8 + i]);
        fprintf(stderr, "\n");
      }
    }
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

  // scratch: the frame queue head (256 B), the channel rows of the staircase variables, the two sets of minima rows
  int decode_ws(void *ws, size_t ws_bytes, int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter,
                uint8_t *bits, int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out,
                hipStream_t st) override {
    if (nframes <= 0) return LDPC_OK;
    if (!llr || max_iter < 0 || max_iter > 65000 || dt < 0 || dt > 3) {
      ldpc_set_error("bad arguments to the staircase Q8.8 decoder");
      return LDPC_ERR_ARG;
    }
    // as many workgroups as the scratch has rows for (each keeps R * block * 10 bytes), at most one per CU
    long grid = std::min<long>(num_cu, nframes);
    if (ws && ws_bytes > 512) grid = std::min<long>(grid, (long)((ws_bytes - 512) / (q0s_bytes(1) + nms_bytes(1))));
    const size_t q0b = (q0s_bytes(grid) + 255) / 256 * 256;
    if (!ws || grid < 1 || ws_bytes < 256 + q0b + nms_bytes(grid)) { ldpc_set_error("scratch too small for the staircase Q8.8 decoder"); return LDPC_ERR_ARG; }
    return launch((int *)ws, (short *)((char *)ws + 256), (uint32_t *)((char *)ws + 256 + q0b), grid, nframes, llr, dt, sy, max_iter, bits, iters_out, sout,
                  status, llr_out, st);
  }
};

}  // namespace

Engine *ldpc_make_chain_q88_engine() { return new (std::nothrow) ChainQ88Engine; }
