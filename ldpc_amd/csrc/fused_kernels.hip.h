// fused_kernels.hip.h -- whole-decode kernel of the Q8.8 min-sum decoder with the
// frame resident in LDS (gfx950: 160 KiB per CU).
//
// dec() of test_code_min_sum_B_fixed.c:192-316 is integer arithmetic once q0 is
// on the 1/256 grid, and integer sums are associative.  That allows a layout
// that never materialises the 2E edge messages in memory at all:
//
//   LDS     sum[n]  int32   posterior of every variable: q0 + all its c2v
//   VGPRs   per check (one thread owns a check): the check's c2v messages in
//           compressed form -- min1, min2, position of min1, one sign bit per
//           edge -- for the previous and the current iteration
//
// One workgroup decodes one frame, start to finish.  Per iteration:
//   phase A  each check thread walks its edges: reads sum[v], rebuilds its own
//            previous message c_old (from the compressed state), forms
//            v2c = sat127(sum[v] - c_old)  (:213-229, float_to_fix :43-53),
//            accumulates sign parity (with the syndrome bit, :264-267) and
//            min1/min2 (:269-277) -> the new compressed state.  The same reads
//            give the syndrome of the current hard decisions (enc() + HamDist,
//            :308-312) for free.
//   barrier  (__syncthreads_or of "some check unsatisfied": the exit test)
//   phase B  each check thread adds (c_new - c_old) into sum[v] of its edges
//            with LDS integer atomics (ds_add_u32): sum stays q0 + sum of c2v.
//   barrier
// All reads of phase A see the sums of the previous iteration, so this is the
// flooding schedule of the reference, and every value is the reference's value
// times 256: decoded bits, iteration counts and syndromes are bit-exact.
// HBM traffic per frame is the LLRs in (2n bytes) and the decisions out.
//
// Workgroups are persistent and pull frames from a device counter, so a frame
// that converges early frees its CU at once (per-frame early exit with no idle
// lanes).  The edge->variable indices are stored transposed
// ([round][slot][thread]) so a wavefront reads them as contiguous rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace ldpc {

struct FusedArgs {
  const uint16_t *colT;   // [rounds][rmax][block]  variable of (check = round*block + tid, slot)
  const uint8_t *dcT;     // [rounds][block]        row weight (0: no such check)
  int n, m, rmax, rounds;
  int nframes, max_iter;
  const void *llr;        // [nframes][n], dtype dt (frame-major, as callers hold q0)
  int dt;                 // ldpc_dtype
  const uint8_t *synd;    // [nframes][m] or null
  uint8_t *bits;          // [nframes][n] or null
  int32_t *iters;         // [nframes] or null
  uint8_t *sout;          // [nframes][m] or null
  int32_t *status;        // [nframes] or null
  float *llr_out;         // [nframes][n] or null
  int *counter;           // frame queue head (zeroed by the host before the launch)
  int chunk;              // frames a workgroup takes from the queue per atomic (>= 1; register-cached kernel)
  // register-cached kernel only (lds_schedule.cpp): colT/dcT are then [rounds][RM][block]
  // LDS words (pads already point at dummy words)
  const uint16_t *pos;    // [n] variable -> LDS word
  const uint32_t *maskT;  // [rounds][block][nw] sign-mask bits of the real slots
  int words, dummy_base, ndummy;
  // timing builds only (LDPC_FUSED_TIMING, tools/probes): [16 wavefronts][8] cycle sums
  unsigned long long *timing;
  // generic kernel only: 1 = layered schedule, one layer = one round (blockDim.x consecutive checks)
  int layered;
};

typedef void (*FusedKern)(FusedArgs);
typedef unsigned fq_u4 __attribute__((ext_vector_type(4)));
typedef unsigned fq_u2 __attribute__((ext_vector_type(2)));

// The register-cached kernels are instantiated in fused_inst.hip, compiled in FQ_PARTS
// parts; R2 (slot pairs per check) comes from this list: every value up to 16, then even.
#define FQ_PARTS 4
inline int fq_round_r2(int r2) {
  if (r2 < 2) return 2;
  if (r2 <= 16) return r2;
  return r2 <= 32 ? (r2 + 1) / 2 * 2 : 0;
}
// shapes whose state fits the 128 VGPRs of a 16-wave workgroup without spilling (measured
// from the compiler's register reports); others are not built and take the generic kernel
constexpr bool fq_reg_shape_ok(int cpt, int r2) {
  return cpt == 1 ? r2 <= 32 : cpt == 2 ? r2 <= 16 : cpt == 3 ? r2 <= 9 : false;
}
// Largest workgroup a shape's kernel is built for: 1024 threads (16 waves x 128 VGPRs) for the
// shapes above; three checks per thread of up to 22 slot pairs get a 12-wave build (168 VGPRs:
// the few values the compiler still spills are touched outside the iteration loop or a dozen
// times per iteration, against ~2000 VALU instructions).  0: no kernel.
constexpr int fq_reg_max_block(int cpt, int r2) {
  return fq_reg_shape_ok(cpt, r2) ? 1024 : (cpt == 3 && r2 <= 22) ? 768 : 0;
}
FusedKern fused_reg_lookup_part0(int cpt, int r2, int sh);
FusedKern fused_reg_lookup_part1(int cpt, int r2, int sh);
FusedKern fused_reg_lookup_part2(int cpt, int r2, int sh);
FusedKern fused_reg_lookup_part3(int cpt, int r2, int sh);
inline FusedKern fused_reg_lookup(int cpt, int r2, int sh) {
  FusedKern k = fused_reg_lookup_part0(cpt, r2, sh);
  if (!k) k = fused_reg_lookup_part1(cpt, r2, sh);
  if (!k) k = fused_reg_lookup_part2(cpt, r2, sh);
  if (!k) k = fused_reg_lookup_part3(cpt, r2, sh);
  return k;
}

// timing experiments only (results wrong when non-zero; tools/probes/build_ablate.sh): 1 no LDS
// adds in phase B (the arithmetic stays), 2 no LDS reads in phase A (the arithmetic stays, on
// register values), 4 no barriers inside the iteration loop (and no exit test); 16 / 32 / 64 /
// 128 / 256: one instruction of the slot loops replaced by a fast-rate one or dropped
// (med3, min, the min1/min2 select, phase B's convert, the key packing)
#ifndef LDPC_FUSED_ABLATE
#define LDPC_FUSED_ABLATE 0
#endif

// timing builds (tools/probes/build_ablate.sh T): every wavefront sums the core-clock cycles it
// spends in phase A, waiting at the barrier after it, in phase B, waiting at the barrier after
// that, and outside the iteration loop, into FusedArgs::timing
#ifndef LDPC_FUSED_TIMING
#define LDPC_FUSED_TIMING 0
#endif
#if LDPC_FUSED_TIMING
#define FQ_TICK(var) const long long var = clock64()
#define FQ_TACC(i, from, to) tacc[i] += (to) - (from)
#else
#define FQ_TICK(var)
#define FQ_TACC(i, from, to)
#endif

#define FQ_SAT 32512      /* 127.0 */
#define FQ_CAP 51200      /* 200.0 */

__device__ __forceinline__ int fq_from_real(double x) {   // float_to_fix(), raw
  double r = round(x * 256.0);
  if (r > 127.0 * 256.0) r = 127.0 * 256.0;
  if (r < -128.0 * 256.0) r = -128.0 * 256.0;
  return (int)r;
}

template <typename MaskT> struct CheckState {
  unsigned mins;   // min1 | min2 << 16  (both <= FQ_CAP < 65536)
  int arg;         // slot of min1
  MaskT sg;        // bit k: c2v of slot k is negative
};

template <typename MaskT>
__device__ __forceinline__ int cs_msg(const CheckState<MaskT> &s, int k) {
  const int mag = (k == s.arg) ? (int)(s.mins >> 16) : (int)(s.mins & 0xffffu);
  return ((s.sg >> k) & 1) ? -mag : mag;
}

template <int CPT, typename MaskT>
__global__ __launch_bounds__(1024) void fused_q88_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) int sum[];   // [n] (+1 word: frame id)
  const int tid = threadIdx.x, B = blockDim.x;
  int *frame_slot = sum + a.n;

  int dc[CPT];
#pragma unroll
  for (int r = 0; r < CPT; r++) dc[r] = r < a.rounds ? a.dcT[r * B + tid] : 0;

  for (;;) {
    __syncthreads();                                   // previous frame fully written out
    if (tid == 0) *frame_slot = atomicAdd(a.counter, 1);
    __syncthreads();
    const int frame = *frame_slot;
    if (frame >= a.nframes) break;

    // q0 -> sum  (float_to_fix for real-valued inputs, as the reference's bsc() :117-120)
    {
      const size_t base = (size_t)frame * a.n;
      if (a.dt == 2) {                                 // LDPC_DT_I16 raw Q8.8
        const short *p = (const short *)a.llr + base;
        for (int i = tid; i < a.n; i += B) sum[i] = p[i];
      } else if (a.dt == 3) {                          // LDPC_DT_I32 raw
        const int *p = (const int *)a.llr + base;
        for (int i = tid; i < a.n; i += B) {
          int v = p[i];
          sum[i] = v > FQ_SAT ? FQ_SAT : (v < -32768 ? -32768 : v);
        }
      } else if (a.dt == 1) {
        const float *p = (const float *)a.llr + base;
        for (int i = tid; i < a.n; i += B) sum[i] = fq_from_real((double)p[i]);
      } else {
        const double *p = (const double *)a.llr + base;
        for (int i = tid; i < a.n; i += B) sum[i] = fq_from_real(p[i]);
      }
    }
    CheckState<MaskT> cur[CPT], nxt[CPT];
    int sbit[CPT], par[CPT];
#pragma unroll
    for (int r = 0; r < CPT; r++) {
      cur[r].mins = 0; cur[r].arg = -1; cur[r].sg = 0;
      nxt[r] = cur[r];
      const int c = r * B + tid;
      sbit[r] = (a.synd && dc[r]) ? (a.synd[(size_t)frame * a.m + c] & 1) : 0;
      par[r] = 0;
    }
    __syncthreads();

    int it_done = a.max_iter, ok = 0;
    // check pass of the thread's check of round r on the sums as they stand: new compressed
    // messages into nxt[r]; returns the parity of the hard decisions of its variables
    auto check_round = [&](const int r) -> int {
      const uint16_t *col = a.colT + (size_t)r * a.rmax * B + tid;
      const int d = dc[r];
      int m1 = FQ_CAP, m2 = FQ_CAP, arg = -1, p = 0;
      MaskT sg = 0;
      for (int k = 0; k < d; k++) {
        const int v = col[(size_t)k * B];
        const int S = sum[v];
        p ^= (S < 0);
        const int q = S - cs_msg(cur[r], k);
        const int neg = q < 0;
        int mag = neg ? -q : q;
        mag = mag > FQ_SAT ? FQ_SAT : mag;
        sg |= (MaskT)neg << k;
        if (mag < m1) { m2 = m1; m1 = mag; arg = k; }
        else if (mag < m2) { m2 = mag; }
      }
      // sgnprod = s ^ XOR(signs); c2v_k < 0 iff sgnprod != sign_k (:278-283)
      int tot = sbit[r];
      if (sizeof(MaskT) == 8) tot ^= __popcll((unsigned long long)sg) & 1;
      else tot ^= __popc((unsigned)sg) & 1;
      MaskT all = d >= (int)(8 * sizeof(MaskT)) ? ~(MaskT)0 : (((MaskT)1 << d) - 1);
      nxt[r].mins = (unsigned)m1 | ((unsigned)m2 << 16);
      nxt[r].arg = arg;
      nxt[r].sg = tot ? (sg ^ all) : sg;
      return p;
    };
    // the new messages of round r replace the old ones in the sums
    auto fold_round = [&](const int r) {
      const uint16_t *col = a.colT + (size_t)r * a.rmax * B + tid;
      const int d = dc[r];
      for (int k = 0; k < d; k++) {
        const int delta = cs_msg(nxt[r], k) - cs_msg(cur[r], k);
        if (delta) atomicAdd(&sum[col[(size_t)k * B]], delta);
      }
      cur[r] = nxt[r];
    };
    for (int t = 1; t <= a.max_iter + 1; t++) {
      int bad = 0;
      if (a.layered) {
        // ---- layered schedule (ldpc_opts.layer_checks; not in the reference): the exit test on the
        // sums of iteration t-1 first (dec()'s, :298-313), then layer after layer
#pragma unroll
        for (int r = 0; r < CPT; r++) {
          if (r < a.rounds) {
            const uint16_t *col = a.colT + (size_t)r * a.rmax * B + tid;
            int p = 0;
            for (int k = 0; k < dc[r]; k++) p ^= (sum[col[(size_t)k * B]] < 0);
            par[r] = p;
            bad |= p ^ sbit[r];
          }
        }
        const int any_bad = __syncthreads_or(bad);
        if (!any_bad) { it_done = t - 1; ok = 1; break; }
        if (t == a.max_iter + 1) break;
#pragma unroll
        for (int r = 0; r < CPT; r++) {
          if (r < a.rounds) {                      // (uniform: every thread takes part in the barriers)
            (void)check_round(r);
            __syncthreads();                       // the whole layer has read its sums
            fold_round(r);
            __syncthreads();
          }
        }
        continue;
      }
      // ---- phase A: check pass on the sums of iteration t-1
#pragma unroll
      for (int r = 0; r < CPT; r++) {
        if (r < a.rounds) {
          const int p = check_round(r);
          par[r] = p;
          bad |= p ^ sbit[r];
        }
      }
      const int any_bad = __syncthreads_or(bad);
      if (!any_bad) { it_done = t - 1; ok = 1; break; }   // H*bits == s after iteration t-1
      if (t == a.max_iter + 1) break;
      // ---- phase B: fold the new messages into the sums
#pragma unroll
      for (int r = 0; r < CPT; r++)
        if (r < a.rounds) fold_round(r);
      __syncthreads();
    }

    // ---- outputs of this frame
    if (a.bits) {
      uint8_t *o = a.bits + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = sum[i] < 0;
    }
    if (a.llr_out) {
      float *o = a.llr_out + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) {
        int s = sum[i];
        s = s > FQ_SAT ? FQ_SAT : (s < -32768 ? -32768 : s);   // sum = float_to_fix(sum), :303
        o[i] = (float)s * (1.0f / 256.0f);
      }
    }
    if (a.sout) {
#pragma unroll
      for (int r = 0; r < CPT; r++)
        if (dc[r]) a.sout[(size_t)frame * a.m + r * B + tid] = (uint8_t)par[r];
    }
    if (tid == 0) {
      if (a.iters) a.iters[frame] = it_done;
      if (a.status) a.status[frame] = ok ? 0 : 1;
    }
  }
}

// Same algorithm with the edge->variable indices of a thread's checks held in
// VGPRs for the life of the (persistent) workgroup: 2 uint16 LDS byte offsets
// per register, slot loops fully unrolled, so an iteration touches no global
// memory at all and the LDS reads of a check are issued back to back.
//   CPT    checks per thread,  R2 = ceil(rmax / 2) registers of indices per check
//
// Arithmetic: every Q8.8 value is an integer of magnitude < 2^19, which binary32
// holds exactly, and so is every sum or difference the decoder forms; the sums
// stay integers in LDS (ds_add_u32; LDS float atomics are several times slower),
// biased so that their bit pattern reads as a float (FQ_BIAS below), and a check's
// arithmetic runs on floats holding integers.  That buys the VALU shortcuts the inner loop is
// built from: |x| and -x are operand modifiers, a sign is moved with one
// v_bfi, the slot's sign bit is shifted into the mask with one v_alignbit, and
// min1 / min2 / argmin come from one v_min_u32 + one v_med3_u32 on the key
// (bits(|q|) | slot): integer-valued floats below 2^19 have at least 5 zero
// low mantissa bits, so the key orders exactly like (|q|, slot).
//
// Slots beyond a check's weight are not predicated (that would cost a wave mask
// per slot); they point at one of FQ_NDUMMY extra LDS words that hold a huge
// positive value, which makes them neutral: sign 0, never a minimum.  What phase
// B adds to the dummies is bounded per iteration; they are re-armed every
// iteration.
//
// The LDS words carry a bias: word = sum + FQ_BIAS, where FQ_BIAS is the bit pattern of the
// float 2^23 + 2^21.  Read as a float the word IS 2^23 + 2^21 + sum (integers in [2^23, 2^24)
// are consecutive bit patterns), so "int -> float" in phase A is one subtraction at the fast
// VALU rate instead of a v_cvt_f32_i32 at the slow one; phase B's integer adds do not see the
// bias at all.  Exact while -2^21 <= sum < 2^23 - 2^21: |sum| <= 128.0 + cmax * 200.0 in Q8.8
// units, which the host checks (fq_cmax_ok).  Dummy words sit 2^24 above the bias -- beyond the
// exact range, where a float moves 4 per unit of its bit pattern, which is immaterial for a
// value that only has to stay "huge and positive".
#define FQ_NDUMMY 256
#define FQ_BIAS 0x4B200000
#define FQ_BIASF 10485760.0f /* 2^23 + 2^21 */
#define FQ_BIG (FQ_BIAS + (1 << 24))
#define FQ_PADLIMF 4194304.0f /* 2^22 */
// largest variable degree for which every sum (and every sum - message) stays inside the exact range
constexpr bool fq_cmax_ok(int cmax) { return 32768 + (cmax + 1) * FQ_CAP < (1 << 21); }

__device__ __forceinline__ unsigned fq_med3u(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float fq_finish_minf(float mu) {
  // min over saturated magnitudes = saturated min; no real edge at all -> 200.0 (:270)
  return mu >= FQ_PADLIMF ? (float)FQ_CAP : (mu > (float)FQ_SAT ? (float)FQ_SAT : mu);
}
// copysign: magnitude bits of mag, sign bit of sw
__device__ __forceinline__ float fq_withsign(float mag, unsigned sw) {
  return __uint_as_float((__float_as_uint(mag) & 0x7fffffffu) | (sw & 0x80000000u));
}

// magnitude (>= 0, so no sign to clear) with the sign bit of sw
__device__ __forceinline__ float fq_signed(float mag, unsigned sw) {
  return __uint_as_float(__float_as_uint(mag) | (sw & 0x80000000u));
}

// x << 1 as an add (v_add_u32 issues at the fast VALU rate, v_lshlrev_b32 does not; the
// compiler turns x + x back into a shift, hence the asm)
__device__ __forceinline__ unsigned fq_shl1(unsigned x) {
  unsigned r;
  asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
  return r;
}
// (a & b) | c in one v_bitop3_b32 (fast rate; v_and_or_b32 is not)
#define FQ_ANDOR(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xEA)

// LDS byte address of one of the two 16-bit entries packed in `packed`, by a volatile asm:
// the unpacking cannot be hoisted out of the iteration loop (which would cost a register
// per slot) and, unlike an empty asm that "modifies" the packed word, leaves the index
// registers untouched (no copies).  SH = 0: entries are byte offsets; SH = 2: word indices.
typedef __attribute__((address_space(3))) int lds_int;
template <int HI, int SH>
__device__ __forceinline__ lds_int *fq_slot_ptr(unsigned base, unsigned packed) {
  unsigned r;
  if (SH == 0) {
    // entries already hold absolute LDS byte addresses (base folded in when they were packed)
    if (HI) asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(r) : "v"(packed));
    else asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(r) : "v"(packed));
  } else {
    if (HI) asm volatile("v_lshrrev_b32 %0, 14, %1\n\tv_and_b32 %0, 0x3fffc, %0" : "=&v"(r) : "v"(packed));
    else asm volatile("v_lshlrev_b32 %0, 2, %1\n\tv_and_b32 %0, 0x3fffc, %0" : "=&v"(r) : "v"(packed));
    r += base;
  }
  return (lds_int *)(unsigned long)r;
}

// Workgroup-wide OR with ONE barrier (__syncthreads_or compiles to three): every wavefront that
// saw a set bit raises the flag of this iteration's parity, the barrier that separates the two
// phases publishes it, and lane 0 lowers the other parity's flag for the iteration after next
// (every wavefront has read that one before it arrived here; it is raised again only behind the
// next barrier).  flags[2] must be zero when a frame starts.
__device__ __forceinline__ int fq_block_or(int bad, int *flags, int t) {
  int *f = flags + (t & 1);
  if (__ballot(bad != 0) != 0 && (threadIdx.x & 63) == 0) *f = 1;
  __syncthreads();
  const int r = *f;
  if (threadIdx.x == 0) flags[(t & 1) ^ 1] = 0;
  return r;
}

// SH: LDS byte offset = index << SH (0 when the offsets themselves fit 16 bits)
template <int CPT, int R2, int SH, int LB, int MINW>
__global__ __launch_bounds__(LB, MINW) void fused_q88_reg_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) int fsum[];   // [n] + dummies + frame id
  const int tid = threadIdx.x, B = blockDim.x;
  int *frame_slot = (int *)(fsum + a.words);
  constexpr int RM = 2 * R2;
#ifndef LDPC_FUSED_CH
#define LDPC_FUSED_CH 0   /* 0: four reads per check in flight with two or more checks per thread, else eight (measured) */
#endif
#ifndef LDPC_FUSED_BCH
#define LDPC_FUSED_BCH 8
#endif
  constexpr int CHMAX = LDPC_FUSED_CH ? LDPC_FUSED_CH : (CPT >= 2 ? 4 : 8);
  constexpr int CH = RM < CHMAX ? RM : CHMAX;  // LDS reads in flight per check
  constexpr int NW = (RM + 31) / 32;           // sign-mask words per check
  constexpr int WB = RM < 32 ? RM : 32;        // slots per mask word
  constexpr int KB = RM <= 32 ? 5 : 6;         // key bits that hold the slot
  constexpr unsigned KM = (1u << KB) - 1;
  const char *lds = (const char *)fsum;
  const unsigned lbase = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)fsum;

  int dc[CPT];
  unsigned idx[CPT][R2];
  unsigned allm[CPT][NW];                      // mask bits of the real slots
#pragma unroll
  for (int r = 0; r < CPT; r++) {
    dc[r] = r < a.rounds ? a.dcT[r * B + tid] : 0;
#pragma unroll
    for (int k2 = 0; k2 < R2; k2++) {
      unsigned lo = (unsigned)a.dummy_base, hi = (unsigned)a.dummy_base;
      if (r < a.rounds) {
        const uint16_t *col = a.colT + (size_t)r * RM * B + tid;
        lo = col[(size_t)(2 * k2) * B];
        hi = col[(size_t)(2 * k2 + 1) * B];
      }
      if (SH == 0) { lo = lo * 4u + lbase; hi = hi * 4u + lbase; }   // absolute addresses (host: words*4 + static LDS < 64 Ki)
      idx[r][k2] = lo | (hi << 16);
    }
#pragma unroll
    for (int w = 0; w < NW; w++) allm[r][w] = r < a.rounds ? a.maskT[((size_t)r * B + tid) * NW + w] : 0u;
  }

  // does this wavefront own any check of the last round?  (wave-uniform)
  const bool wave_full = __ballot(dc[CPT - 1] != 0) != 0;

  // 8 variables per thread and step on the way in (int16 LLRs) and out (bits): needs
  // 16-byte aligned frames / 8-byte aligned bit rows
  const bool vec_in = a.dt == 2 && (a.n & 7) == 0 && ((size_t)a.llr & 15) == 0;
  const bool vec_out = a.bits && (a.n & 7) == 0 && ((size_t)a.bits & 7) == 0;
  const fq_u4 *pos8 = (const fq_u4 *)a.pos;

#if LDPC_FUSED_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // Frames come from one device-wide counter, a.chunk at a time: short frames ((204,102): 80 million
  // a second) would otherwise queue up on that one L2 address
  int q_next = 0, q_lim = 0;                    // (thread 0) the rest of the chunk in hand
  auto take_frame = [&]() {
    if (q_next == q_lim) { q_next = atomicAdd(a.counter, a.chunk); q_lim = q_next + a.chunk; }
    return q_next++;
  };
  if (tid == 0) *frame_slot = take_frame();
  __syncthreads();
  for (;;) {
    const int frame = *frame_slot;
    if (frame >= a.nframes) break;
    FQ_TICK(tf0);
    // the next frame's number is fetched now and published at the end of this frame: the
    // round trip of the device atomic hides under the decoding
    int next_frame = 0;
    if (tid == 0) next_frame = take_frame();
    {
      const size_t base = (size_t)frame * a.n;
      auto put = [&](int word, int v) { fsum[word] = FQ_BIAS + v; };   // one variable's q0 (on the Q8.8 grid, within int16)
      if (vec_in) {
        const fq_u4 *p8 = (const fq_u4 *)((const short *)a.llr + base);
        for (int i = tid; i < (a.n >> 3); i += B) {
          const fq_u4 v = __builtin_nontemporal_load(p8 + i);
          const fq_u4 w = pos8[i];
          fsum[w.x & 0xffffu] = FQ_BIAS + (int)(short)(v.x & 0xffffu); fsum[w.x >> 16] = FQ_BIAS + ((int)v.x >> 16);
          fsum[w.y & 0xffffu] = FQ_BIAS + (int)(short)(v.y & 0xffffu); fsum[w.y >> 16] = FQ_BIAS + ((int)v.y >> 16);
          fsum[w.z & 0xffffu] = FQ_BIAS + (int)(short)(v.z & 0xffffu); fsum[w.z >> 16] = FQ_BIAS + ((int)v.z >> 16);
          fsum[w.w & 0xffffu] = FQ_BIAS + (int)(short)(v.w & 0xffffu); fsum[w.w >> 16] = FQ_BIAS + ((int)v.w >> 16);
        }
      } else if (a.dt == 2) {
        const short *p = (const short *)a.llr + base;
        for (int i = tid; i < a.n; i += B) put(a.pos[i], p[i]);
      } else if (a.dt == 3) {
        const int *p = (const int *)a.llr + base;
        for (int i = tid; i < a.n; i += B) {
          int v = p[i];
          put(a.pos[i], v > FQ_SAT ? FQ_SAT : (v < -32768 ? -32768 : v));
        }
      } else if (a.dt == 1) {
        const float *p = (const float *)a.llr + base;
        for (int i = tid; i < a.n; i += B) put(a.pos[i], fq_from_real((double)p[i]));
      } else {
        const double *p = (const double *)a.llr + base;
        for (int i = tid; i < a.n; i += B) put(a.pos[i], fq_from_real(p[i]));
      }
      for (int i = a.dummy_base + tid; i < a.words; i += B) fsum[i] = FQ_BIG;   // dummies
    }
    // compressed messages of the previous (o*) and the current (n*) iteration
    float om1[CPT], om2[CPT], nm1[CPT], nm2[CPT];
    unsigned oarg[CPT], narg[CPT], osg[CPT][NW], nsg[CPT][NW];
    unsigned ooff[CPT], noff[CPT];             // LDS byte offsets of the variables in slots oarg / narg
    int sbit[CPT], par[CPT];
#pragma unroll
    for (int r = 0; r < CPT; r++) {
      om1[r] = om2[r] = nm1[r] = nm2[r] = 0.0f;
      oarg[r] = narg[r] = 0u;                  // (om1 == om2 == 0: which slot it names is immaterial)
      ooff[r] = noff[r] = (unsigned)a.dummy_base * 4u;
#pragma unroll
      for (int w = 0; w < NW; w++) osg[r][w] = nsg[r][w] = 0;
      sbit[r] = (a.synd && dc[r]) ? (a.synd[(size_t)frame * a.m + r * B + tid] & 1) : 0;
      par[r] = 0;
    }
    if (tid == 0) { frame_slot[1] = 0; frame_slot[2] = 0; }   // flags of fq_block_or
    FQ_TICK(tf_in0);
    __syncthreads();
    FQ_TICK(tf_in);
    FQ_TACC(5, tf0, tf_in0);
    FQ_TACC(6, tf_in0, tf_in);

    int it_done = a.max_iter, ok = 0;
    for (int t = 1; t <= a.max_iter + 1; t++) {
      FQ_TICK(tc0);
      int bad = 0;
      // Phase A for the first NA of the thread's checks.  NA is CPT except in wavefronts whose
      // last round holds no check at all (the rounds fill up in order, so those have exactly
      // CPT - 1): they run the shorter instruction stream instead of walking dummy slots.
      // FIRST: the first iteration of a frame, where every old message is +0 -- no min1/min2 select, no
      // sign to inject (phase A: 2 of 9.5 instructions per slot less; phase B: 3 of 7)
      auto phase_a = [&](auto na_c, auto first_c) {
        constexpr int NA = decltype(na_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        // keep the offsets packed: the empty asm makes the unpacking loop-variant for the
        // compiler, which would otherwise hoist 2*R2 unpacked addresses per check
        unsigned k1[CPT], k2m[CPT], pp[CPT], sg[CPT][NW];
        // Which lanes have their previous minimum in slot k?  The slot numbers, transposed into one
        // lane mask per bit (KB compares per check and iteration); the mask of slot k is then an
        // AND of those on the scalar unit, which has nothing else to do, instead of a vector
        // compare per slot.
        unsigned long long ab[CPT][KB];
        if constexpr (!FIRST) {
#pragma unroll
          for (int r = 0; r < NA; r++)
#pragma unroll
            for (int b = 0; b < KB; b++) ab[r][b] = __ballot((oarg[r] >> b) & 1u);
        }
#pragma unroll
        for (int r = 0; r < NA; r++) {
          k1[r] = k2m[r] = 0xffffffffu;
          pp[r] = 0;
#pragma unroll
          // One register per mask word serves both directions: the old signs leave at the top
          // (the slot in turn is always bit 31) while the new ones enter at the bottom, one
          // v_alignbit per slot; after the word's last slot only new signs are left.
          for (int w = 0; w < NW; w++) sg[r][w] = FIRST ? 0u : osg[r][w] << (32 - WB);
        }
        // the thread's checks advance together, CH slots at a time: CPT*CH independent LDS
        // reads in flight and CPT independent min chains
#pragma unroll
        for (int kc = 0; kc < RM; kc += CH) {
          int S[CPT][CH];
#pragma unroll
          for (int r = 0; r < NA; r++)
#pragma unroll
            for (int j = 0; j < CH; j++) {
              const int k = kc + j;
              if (k >= RM) continue;               // (compile time: last chunk of an RM that CH does not divide)
              if (LDPC_FUSED_ABLATE & 2) S[r][j] = (int)(unsigned)(unsigned long)((k & 1) ? fq_slot_ptr<1, SH>(lbase, idx[r][k >> 1]) : fq_slot_ptr<0, SH>(lbase, idx[r][k >> 1]));
              else
              S[r][j] = (k & 1) ? *fq_slot_ptr<1, SH>(lbase, idx[r][k >> 1]) : *fq_slot_ptr<0, SH>(lbase, idx[r][k >> 1]);
            }
#pragma unroll
          for (int j = 0; j < CH; j++)
#pragma unroll
            for (int r = 0; r < NA; r++) {
              const int k = kc + j;
              if (k >= RM) continue;
              // the sums as floats (the words are biased: see FQ_BIAS), two slots at a time
              if ((j & 1) == 0) {
                S[r][j] = (int)__float_as_uint(__uint_as_float((unsigned)S[r][j]) - FQ_BIASF);
                S[r][j + 1] = (int)__float_as_uint(__uint_as_float((unsigned)S[r][j + 1]) - FQ_BIASF);
                // hard-decision parity, two slots per v_bitop3 (x ^ y ^ z); CH and RM are even
                pp[r] = __builtin_amdgcn_bitop3_b32(pp[r], (unsigned)S[r][j], (unsigned)S[r][j + 1], 0x96);
              }
              float q = __uint_as_float((unsigned)S[r][j]);     // (FIRST: sum - (+0))
              if constexpr (!FIRST) {
                float mo;
                if (LDPC_FUSED_ABLATE & 64) mo = om1[r];
                else {
                  unsigned long long in_k = (k & 1) ? ab[r][0] : ~ab[r][0];
#pragma unroll
                  for (int b = 1; b < KB; b++) in_k &= ((k >> b) & 1) ? ab[r][b] : ~ab[r][b];
                  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(mo) : "v"(om1[r]), "v"(om2[r]), "s"(in_k));
                }
                q = q - __uint_as_float(FQ_ANDOR(sg[r][k / 32], 0x80000000u, __float_as_uint(mo)));
              }
              sg[r][k / 32] = __builtin_amdgcn_alignbit(sg[r][k / 32], __float_as_uint(q), 31);
              const unsigned key = (LDPC_FUSED_ABLATE & 256) ? __float_as_uint(q) : FQ_ANDOR(__float_as_uint(q), 0x7fffffffu, (unsigned)k);
              if (LDPC_FUSED_ABLATE & 16) k2m[r] ^= key & k1[r];
              else
              k2m[r] = fq_med3u(k1[r], key, k2m[r]);
              if (LDPC_FUSED_ABLATE & 32) k1[r] &= key;
              else
              k1[r] = min(k1[r], key);
            }
        }
#pragma unroll
        for (int r = 0; r < NA; r++) {
          // the masks keep slot 32w + j of word w in bit 31 - j: a last word with fewer than 32
          // slots has been shifted in from the right and is moved up
          if constexpr (NW > 1 && RM % 32 != 0) sg[r][NW - 1] <<= 32 - RM % 32;
          const unsigned p = pp[r] >> 31;
          par[r] = (int)p;
          bad |= (int)p ^ sbit[r];
          unsigned tot = (unsigned)sbit[r];
#pragma unroll
          for (int w = 0; w < NW; w++) tot ^= (unsigned)__popc(sg[r][w]);
          nm1[r] = fq_finish_minf(__uint_as_float(k1[r] & ~KM));
          nm2[r] = fq_finish_minf(__uint_as_float(k2m[r] & ~KM));
          // A thread with no check at all (beyond m in a wavefront that has checks) sees only pad
          // slots, whose dummy words carry this phase's racing adds: 2^24 + delta is not on the
          // 5-zero-bit grid the key packing relies on, so the slot bits can come out >= RM.  Its
          // messages are all "cap" and nothing it adds matters, but the index below must stay
          // inside colT (an out-of-bounds read, up to 9 KB past the table, faulted once the
          // neighbouring pages were unmapped).
          narg[r] = min(k1[r] & KM, (unsigned)(RM - 1));
          // where slot narg's variable lives (needed by phase B's fix-up; the load hides under the barrier)
          {
            // opaque pointer: keeps the compiler from relating this load to the index registers
            const uint16_t *ct = a.colT;
            asm volatile("" : "+s"(ct));
            const int rr = r < a.rounds ? r : a.rounds - 1;
            noff[r] = (unsigned)ct[((size_t)rr * RM + narg[r]) * B + tid] * 4u;
          }
          // sgnprod = s ^ XOR(signs); c2v_k < 0 iff sgnprod != sign_k (:278-283)
#pragma unroll
          for (int w = 0; w < NW; w++) nsg[r][w] = (tot & 1) ? (sg[r][w] ^ allm[r][w]) : sg[r][w];
          // materialise the compressed state now: left alone the compiler carries the 2*R2
          // per-slot values of this check across the barrier instead (register spills)
          asm volatile("" : "+v"(nm1[r]), "+v"(nm2[r]), "+v"(narg[r]));
#pragma unroll
          for (int w = 0; w < NW; w++) asm volatile("" : "+v"(nsg[r][w]));
        }
      };
      // after the last iteration only the verdict is wanted: the parity of the hard decisions
      // (sign bits of the sums), not another set of messages
      auto parity_only = [&](auto na_c) {
        constexpr int NA = decltype(na_c)::value;
#pragma unroll
        for (int r = 0; r < NA; r++) {
          unsigned pp = 0;
#pragma unroll
          for (int k = 0; k < RM; k += 2) {
            const int s0 = *fq_slot_ptr<0, SH>(lbase, idx[r][k >> 1]) - FQ_BIAS;
            const int s1 = *fq_slot_ptr<1, SH>(lbase, idx[r][k >> 1]) - FQ_BIAS;
            pp = __builtin_amdgcn_bitop3_b32(pp, (unsigned)s0, (unsigned)s1, 0x96);
          }
          par[r] = (int)(pp >> 31);
          bad |= par[r] ^ sbit[r];
        }
      };
      if (t == a.max_iter + 1) {
        if (wave_full) parity_only(std::integral_constant<int, CPT>{});
        else parity_only(std::integral_constant<int, CPT - 1>{});
      } else if (t == 1) {
        if (wave_full) phase_a(std::integral_constant<int, CPT>{}, std::true_type{});
        else phase_a(std::integral_constant<int, CPT - 1>{}, std::true_type{});
      } else if (wave_full) phase_a(std::integral_constant<int, CPT>{}, std::false_type{});
      else phase_a(std::integral_constant<int, CPT - 1>{}, std::false_type{});
      FQ_TICK(tc1);
      const int any_bad = (LDPC_FUSED_ABLATE & 4) ? 1 : fq_block_or(bad, frame_slot + 1, t);
      FQ_TICK(tc2);
      FQ_TACC(0, tc0, tc1);
      FQ_TACC(1, tc1, tc2);
      if (!any_bad) { it_done = t - 1; ok = 1; break; }
      if (t == a.max_iter + 1) break;
      for (int i = a.dummy_base + tid; i < a.words; i += B) fsum[i] = FQ_BIG;   // re-arm (races only with this phase's bounded adds)
      // ---- phase B.  delta_k = c_new_k - c_old_k with c = +-min1 everywhere except in the
      // slot of min1 itself, where it is +-min2.  The slot loop treats every slot as
      // "+-min1" (integer arithmetic on sign masks, no per-slot selects) and the two
      // exceptional slots are put right afterwards by one extra add each.
      auto phase_b = [&](auto na_c, auto first_c) {
        constexpr int NA = decltype(na_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
        for (int r = 0; r < NA; r++) {
          // (re)derive everything from the compressed state here, not before the barrier
          asm volatile("" : "+v"(om1[r]), "+v"(om2[r]), "+v"(oarg[r]));
#pragma unroll
          for (int w = 0; w < NW; w++) asm volatile("" : "+v"(osg[r][w]), "+v"(nsg[r][w]));
          const int m1o = (int)om1[r], m1n = (int)nm1[r];
          unsigned nrot[NW], prot[NW];           // sign of the slot in turn in bit 31 (new / previous)
#pragma unroll
          for (int w = 0; w < NW; w++) { nrot[w] = nsg[r][w] << (32 - WB); prot[w] = osg[r][w] << (32 - WB); }
#pragma unroll
          for (int k = 0; k < RM; k++) {
            lds_int *dst = (k & 1) ? fq_slot_ptr<1, SH>(lbase, idx[r][k >> 1]) : fq_slot_ptr<0, SH>(lbase, idx[r][k >> 1]);
            const float cn = __uint_as_float(FQ_ANDOR(nrot[k / 32], 0x80000000u, __float_as_uint(nm1[r])));
            nrot[k / 32] = fq_shl1(nrot[k / 32]);
            float d = cn;                        // (FIRST: the old message is +0)
            if constexpr (!FIRST) {
              const float co = __uint_as_float(FQ_ANDOR(prot[k / 32], 0x80000000u, __float_as_uint(om1[r])));
              prot[k / 32] = fq_shl1(prot[k / 32]);
              d = cn - co;
            }
            const int delta = (LDPC_FUSED_ABLATE & 128) ? (int)__float_as_uint(d) : (int)d;
            if (LDPC_FUSED_ABLATE & 1) asm volatile("" ::"v"(delta), "v"(dst));
            else
            __hip_atomic_fetch_add(dst, delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((k & (LDPC_FUSED_BCH - 1)) == LDPC_FUSED_BCH - 1) __builtin_amdgcn_sched_barrier(0);   // bounds the live registers
          }
          {
            // slot oarg really carried +-min2 before; slot narg carries +-min2 now
            const int dmo = (int)om2[r] - m1o, dmn = (int)nm2[r] - m1n;
            const unsigned ao = oarg[r], an = narg[r];   // < RM: a real slot always wins the minimum
            const int so = -(int)((osg[r][ao / 32] >> (WB - 1 - (ao % 32))) & 1u);
            const int sn = -(int)((nsg[r][an / 32] >> (WB - 1 - (an % 32))) & 1u);
            if constexpr (!FIRST) atomicAdd((int *)(lds + ooff[r]), -((dmo ^ so) - so));   // (FIRST: min2 - min1 = 0)
            atomicAdd((int *)(lds + noff[r]), (dmn ^ sn) - sn);
          }
          om1[r] = nm1[r]; om2[r] = nm2[r]; oarg[r] = narg[r]; ooff[r] = noff[r];
#pragma unroll
          for (int w = 0; w < NW; w++) osg[r][w] = nsg[r][w];
        }
      };
      if (t == 1) {
        if (wave_full) phase_b(std::integral_constant<int, CPT>{}, std::true_type{});
        else phase_b(std::integral_constant<int, CPT - 1>{}, std::true_type{});
      } else if (wave_full) phase_b(std::integral_constant<int, CPT>{}, std::false_type{});
      else phase_b(std::integral_constant<int, CPT - 1>{}, std::false_type{});
      FQ_TICK(tc3);
      if (!(LDPC_FUSED_ABLATE & 4)) __syncthreads();
      FQ_TICK(tc4);
      FQ_TACC(2, tc2, tc3);
      FQ_TACC(3, tc3, tc4);
    }
    FQ_TICK(tf_out0);

    if (vec_out) {
      fq_u2 *o8 = (fq_u2 *)(a.bits + (size_t)frame * a.n);
      for (int i = tid; i < (a.n >> 3); i += B) {
        const fq_u4 w = pos8[i];
        fq_u2 b;
        auto neg = [&](unsigned word) { return (unsigned)(fsum[word] - FQ_BIAS) >> 31; };
        b.x = neg(w.x & 0xffffu) | (neg(w.x >> 16) << 8) | (neg(w.y & 0xffffu) << 16) | (neg(w.y >> 16) << 24);
        b.y = neg(w.z & 0xffffu) | (neg(w.z >> 16) << 8) | (neg(w.w & 0xffffu) << 16) | (neg(w.w >> 16) << 24);
        __builtin_nontemporal_store(b, o8 + i);
      }
    } else if (a.bits) {
      uint8_t *o = a.bits + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) o[i] = fsum[a.pos[i]] < FQ_BIAS;
    }
    if (a.llr_out) {
      float *o = a.llr_out + (size_t)frame * a.n;
      for (int i = tid; i < a.n; i += B) {
        int s = fsum[a.pos[i]] - FQ_BIAS;
        s = s > FQ_SAT ? FQ_SAT : (s < -32768 ? -32768 : s);   // float_to_fix(sum), :303
        o[i] = (float)s * (1.0f / 256.0f);
      }
    }
    if (a.sout) {
#pragma unroll
      for (int r = 0; r < CPT; r++)
        if (dc[r]) a.sout[(size_t)frame * a.m + r * B + tid] = (uint8_t)par[r];
    }
    if (tid == 0) {
      if (a.iters) a.iters[frame] = it_done;
      if (a.status) a.status[frame] = ok ? 0 : 1;
    }
    FQ_TICK(tf_out1);
    FQ_TACC(7, tf_out0, tf_out1);
    __syncthreads();                           // everyone has read this frame's number and its sums
    if (tid == 0) *frame_slot = next_frame;
    // The barrier that publishes the number sits here, in line after the write, not at the
    // loop head: there the compiler (ROCm 7.2) dropped the s_waitcnt between this ds_write
    // and the s_barrier across the back edge, and waves read the previous number.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    FQ_TICK(tf1);
    FQ_TACC(4, tf0, tf1);
  }
#if LDPC_FUSED_TIMING
  if (a.timing && (tid & 63) == 0)
    for (int i = 0; i < 8; i++) atomicAdd(a.timing + (tid >> 6) * 8 + i, (unsigned long long)tacc[i]);
#endif
}

}  // namespace ldpc
