// binary_kernels.hip.h -- gfx950 kernels of the binary flooding decoders.
//
// Data layout in HBM ("slab-interleaved"): frames are grouped in slabs of
// SF = 64*V frames, V = 16 / sizeof(message) (8 for Q8.8 int16, 4 for f32 and
// int32, 2 for f64).  Every per-node or per-edge quantity is stored as one ROW
// per (slab, node|edge): 64 lanes x 16 bytes = 1 KiB, lane l holding frames
// slab*SF + l*V .. +V-1.  A wavefront therefore always moves whole 1-KiB rows
// with one 16-byte access per lane (global_load/store_dwordx4), whatever the
// graph looks like, and node/edge indices are wave-uniform scalars.
//
//   q0  [slab][n][64] vec   channel LLRs
//   msg [slab][E][64] vec   edge messages in check-major (CSR) order, used IN
//                           PLACE: holds c2v after a check pass, v2c after a
//                           variable pass
//   hard[slab][n][64] u8    hard decisions, bit v of a byte = frame l*V+v
//   synd[slab][m][64] u8    target syndrome bits, same packing
//   fail[slab][64]   u32    per-frame "some check unsatisfied" bits (atomicOr)
//   done[slab][64]   u8     per-frame "reached its syndrome" bits
//
// One decoding iteration = vn_kernel (variable pass, also emits the previous
// iteration's hard decisions) + cn_kernel (check pass, also tests the syndrome
// of those hard decisions) + mark_kernel (per-frame bookkeeping).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace ldpc {

enum { ALG_MS = 0, ALG_MSQ = 1, ALG_SPA = 2, ALG_TAB = 3 };

template <typename T> struct MT;
template <> struct MT<short> {
  static constexpr int V = 8;
  typedef short vec __attribute__((ext_vector_type(8)));
  typedef int acc;
};
template <> struct MT<int> {
  static constexpr int V = 4;
  typedef int vec __attribute__((ext_vector_type(4)));
  typedef int acc;
};
template <> struct MT<float> {
  static constexpr int V = 4;
  typedef float vec __attribute__((ext_vector_type(4)));
  typedef float acc;
};
template <> struct MT<double> {
  static constexpr int V = 2;
  typedef double vec __attribute__((ext_vector_type(2)));
  typedef double acc;
};

// Q8.8 limits in raw units (value*256): float_to_fix() saturates to
// [-128, +127] (test_code_min_sum_B_fixed.c:43-53), min_msg starts at 200 (:270)
#define Q88_POS_MAX 32512
#define Q88_NEG_MIN (-32768)
#define Q88_CAP200 51200
// Q8.14 table size (test_code_B.c:27-31)
#define TAB_LEVELS (1 << 22)

struct Graph {
  const int32_t *row_ptr;   // [m+1]
  const int32_t *col_idx;   // [E]
  const int32_t *col_ptr;   // [n+1]
  const int32_t *edge_of;   // [E]
  int n, m, E;
};

struct Tables {           // SPA_TABLE_Q14 only
  const int32_t *flogtanh;
  const int32_t *fgallag;
};

// ---------------------------------------------------------------- helpers

template <int ALG, typename T>
__device__ __forceinline__ T vn_store_value(typename MT<T>::acc x) {
  if (ALG == ALG_MSQ) {
    // magnitude = float_to_fix(|qout|): saturate at +127.0 either sign
    int v = (int)x;
    v = v > Q88_POS_MAX ? Q88_POS_MAX : v;
    v = v < -Q88_POS_MAX ? -Q88_POS_MAX : v;
    return (T)v;
  }
  return (T)x;
}

__device__ __forceinline__ int wave_id_uniform() {
  return __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
}

// ------------------------------------------------------- variable-node pass
//
// Restates the VN loop and the posterior loop of dec()
// (test_code_min_sum_B.c:144-160 and :189-197): sum = q0 + sum_k c2v_k in
// column-slot order, v2c_k = sum - c2v_k.  Because the posterior of iteration
// t is the same sum the VN pass of iteration t+1 starts from, the hard
// decision of iteration t is produced here, one pass later.
//   FIRST: c2v are all zero (first iteration, memset at :141) -> no msg reads
//   STORE: false for the closing posterior-only pass
//   DVB:   compile-time bound on the column weight held in registers
//          (0 = any weight, two passes over the rows)
template <typename T, int ALG, int DVB, bool FIRST, bool STORE>
__global__ __launch_bounds__(256) void vn_kernel(
    Graph g, int nslab, const typename MT<T>::vec *__restrict__ q0,
    typename MT<T>::vec *__restrict__ msg, uint8_t *__restrict__ hard,
    const uint8_t *__restrict__ done, typename MT<T>::vec *__restrict__ post) {
  typedef typename MT<T>::vec vec;
  typedef typename MT<T>::acc acc;
  constexpr int V = MT<T>::V;
  const int lane = threadIdx.x & 63;
  const int wave = wave_id_uniform();
  const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  const long items = (long)g.n * nslab;

  for (long item = wave; item < items; item += nwaves) {
    const int i = (int)(item / nslab);
    const int b = (int)(item - (long)i * nslab);
    const int p0 = g.col_ptr[i];
    const int dv = g.col_ptr[i + 1] - p0;
    vec *mrow = msg + (size_t)b * g.E * 64 + lane;
    const size_t vrow = ((size_t)b * g.n + i) * 64 + lane;
    const vec q = q0[vrow];
    acc S[V];
#pragma unroll
    for (int v = 0; v < V; v++) S[v] = (acc)q[v];

    if constexpr (DVB > 0) {
      int e[DVB];
      vec c[DVB];
#pragma unroll
      for (int k = 0; k < DVB; k++) {
        if (k < dv) {
          e[k] = g.edge_of[p0 + k];
          if (!FIRST) c[k] = mrow[(size_t)e[k] * 64];
        }
      }
#pragma unroll
      for (int k = 0; k < DVB; k++) {
        if (k < dv && !FIRST) {
#pragma unroll
          for (int v = 0; v < V; v++) S[v] = S[v] + (acc)c[k][v];
        }
      }
      if (STORE) {
#pragma unroll
        for (int k = 0; k < DVB; k++) {
          if (k < dv) {
            vec o;
#pragma unroll
            for (int v = 0; v < V; v++)
              o[v] = vn_store_value<ALG, T>(FIRST ? S[v] : S[v] - (acc)c[k][v]);
            mrow[(size_t)e[k] * 64] = o;
          }
        }
      }
    } else {
      if (!FIRST) {
        for (int k = 0; k < dv; k++) {
          const vec c = mrow[(size_t)g.edge_of[p0 + k] * 64];
#pragma unroll
          for (int v = 0; v < V; v++) S[v] = S[v] + (acc)c[v];
        }
      }
      if (STORE) {
        for (int k = 0; k < dv; k++) {
          vec *p = mrow + (size_t)g.edge_of[p0 + k] * 64;
          vec o;
          if (FIRST) {
#pragma unroll
            for (int v = 0; v < V; v++) o[v] = vn_store_value<ALG, T>(S[v]);
          } else {
            const vec c = *p;
#pragma unroll
            for (int v = 0; v < V; v++) o[v] = vn_store_value<ALG, T>(S[v] - (acc)c[v]);
          }
          *p = o;
        }
      }
    }

    // hard decision (sum < 0), frozen for frames that already reached their syndrome
    unsigned hb = 0;
#pragma unroll
    for (int v = 0; v < V; v++) hb |= (S[v] < 0 ? 1u : 0u) << v;
    const unsigned dm = done[b * 64 + lane];
    if (dm) hb = (hard[vrow] & dm) | (hb & ~dm);
    hard[vrow] = (uint8_t)hb;

    if (post) {   // posterior of the frames still running
      vec o = post[vrow];
#pragma unroll
      for (int v = 0; v < V; v++) {
        if (!((dm >> v) & 1)) {
          if (ALG == ALG_MSQ) {   // sum = float_to_fix(sum), _fixed.c:303
            int s = (int)S[v];
            s = s > Q88_POS_MAX ? Q88_POS_MAX : s;
            s = s < Q88_NEG_MIN ? Q88_NEG_MIN : s;
            // an int16 row cannot hold it when T is short and s == ... it can: range fits int16
            o[v] = (T)s;
          } else {
            o[v] = (T)S[v];
          }
        }
      }
      post[vrow] = o;
    }
  }
}

// ---------------------------------------------------------- check-node math

template <int ALG, typename A> struct CnState;

// min-sum (test_code_min_sum_B.c:162-187): out_k = +-min(200, min_{l!=k}|v_l|)
template <typename A> struct CnStateMin {
  A m1, m2;
  __device__ __forceinline__ void init(A big) { m1 = big; m2 = big; }
  __device__ __forceinline__ void add(A a) {
    A hi = a > m1 ? a : m1;       // max(m1, a)
    m2 = hi < m2 ? hi : m2;
    m1 = a < m1 ? a : m1;
  }
  __device__ __forceinline__ A others(A a) const { return a == m1 ? m2 : m1; }
};

// logtanh2() / atanh2(exp()) of test_code_B_float.c:20-27.  binary64 is the reference's literal
// statement.  The literal formula in binary32 saturates at |LLR| ~ 17 (tanhf -> 1, log -> 0, the
// next atanh2(1) = inf and inf - inf = NaN in the variable pass; binary64 gets there only at
// ~37), which no DVB-S2 frame survives.  So binary32 evaluates the SAME two functions without
// the cancellation: log tanh(a/2) = -2 atanh(e^-a) by its series once a > 3, and
// log((1 - e^-a) / (1 + e^-a)) below (1 - e^-a from its series under 0.25), and
// 2 atanh(e^x) with 1 - e^x from the series of -expm1(x) above -0.1 and the atanh series
// below -3.  Worst relative error against exact arithmetic 9e-7 (binary32 rounding).  Values at -inf and NaN stay the
// reference's; a ZERO message is floored at 2^-100 (log tanh = -69.8 instead of -inf): the reference's
// -inf - (-inf) = NaN on the leave-one-out of a zero edge (test_code_B_float.c:149,168) needs an exact
// cancellation in the variable sum, which binary32 rounding produces on frames where the reference's double
// has none (seen on 1 of 530 equal-magnitude frames of 1057.244.3: NaN here, a codeword after 10 iterations
// there); with the floor the zero edge gets the product of the others and the others get ~0, as in exact arithmetic.
#define SPA_F32_MINMAG 0x1p-100f
// v_exp_f32 / v_log_f32 / v_rcp_f32 directly (1 ulp): __expf / __logf wrap them in denormal
// range handling and __fdividef compiles to the IEEE division sequence (measured: 96 v_div_fmas
// and 400 compares in the register kernel).  The arguments here never need either: log of a
// ratio in (1e-37, 2e7], exp of a non-positive number whose underflow to 0 is the right answer.
__device__ __forceinline__ float spa_fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float spa_flog(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float spa_logtanh2(float a) {
  a = a < SPA_F32_MINMAG ? SPA_F32_MINMAG : a;   // (a NaN stays a NaN)
  const float t = spa_fexp(-a);             // tanh(a/2) = (1 - t) / (1 + t): one exp, no tanhf
  if (a > 3.0f) { const float t2 = t * t; return -2.0f * t * (1.0f + t2 * (1.0f / 3.0f + t2 * 0.2f)); }
  float num = 1.0f - t;
  if (a < 0.25f)                        // 1 - e^-a by its series where the difference cancels
    num = a * (1.0f - a * (0.5f - a * (1.0f / 6.0f - a * (1.0f / 24.0f - a * (1.0f / 120.0f - a * (1.0f / 720.0f))))));
  return spa_flog(num * __builtin_amdgcn_rcpf(1.0f + t));
}
__device__ __forceinline__ float spa_atanh2exp(float x) {
  const float e = spa_fexp(x);
  if (x < -3.0f) { const float e2 = e * e; return 2.0f * e * (1.0f + e2 * (1.0f / 3.0f + e2 * 0.2f)); }
  if (x > -0.1f) return spa_flog((1.0f + e) * __builtin_amdgcn_rcpf(fabsf(x) * (1.0f + x * (0.5f + x * (1.0f / 6.0f + x * (1.0f / 24.0f + x * (1.0f / 120.0f)))))));   /* x = 0: log(2 / +0) = +inf like the reference */
  return spa_flog((1.0f + e) * __builtin_amdgcn_rcpf(1.0f - e));
}
// binary64.  The device library's tanh / exp / log and the IEEE division sequence made the literal formulas
// ~165 binary64 instructions per value; they are evaluated here in ~70 with the SAME rounding structure as the
// reference's arithmetic (test_code_B_float.c:20-27), because that structure is behaviour: the quantisation of
// tanh near 1 and of 1 - e^x near x = 0 decide where a message saturates to 0 / +inf and the inf - inf = NaN
// hazard fires (the error floor of the DVB-S2 sweeps).
//   spa_exp64:  e^x for x <= 0 as (1 + p) 2^k: k ln 2 split in two, degree-13 polynomial p = expm1(r), v_ldexp
//   spa_div64:  one v_rcp_f64, two Newton steps, one residual correction
//   spa_logq64: log(num / den) without forming the quotient: both mantissas brought within sqrt 2 of each other,
//               s = (mn - md) / (mn + md), log = k ln 2 + 2 atanh(s) by its series to s^21
//   log tanh(a/2): t = tanh(a/2) ROUNDED to binary64 first, as glibc's is: 1 - 2u / (1 + u) with u = e^-a (so t = 1,
//               the message's weight 0, exactly when u < 2^-55, a > 38.1), -p / (2 + p) where k = 0 (exact for tiny a);
//               then log t
//   atanh2(e^x): e = e^x rounded, then 1 + e and 1 - e rounded (x -> 0: the reference's cancellation, +inf at 0), then
//               the log of their quotient
// Worst relative error against exact arithmetic 4.4e-13 on [-9, -5e-13] (the literal glibc forms: 2.2e-13 / 8.3e-13:
// all of it the rounding of 1 -+ e); NaN in -> NaN out, -inf -> 0.
#ifndef LDPC_SPA_F64_LIBM
// (polynomial coefficients from constant memory: they arrive in scalar registers and enter v_fma_f64 as its
// one scalar operand; as literals each costs a 64-bit move into a vector register in front of a v_fmac -- a
// third of the old kernel's instructions were such moves)
__device__ __constant__ double SPA_EXPC[13] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0,
                                               1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0,
                                               1.0 / 24.0, 1.0 / 6.0, -6.93147180369123816490e-01, -1.90821492927058770002e-10};
__device__ __constant__ double SPA_LOGC[12] = {1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0,
                                               1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0, 1.90821492927058770002e-10, 6.93147180369123816490e-01};
__device__ __forceinline__ double spa_exp64(double x, double &p, int &k) {
  x = x < -750.0 ? -750.0 : x;
  const double kf = __builtin_rint(x * 1.4426950408889634);
  k = (int)kf;
  double r = __builtin_fma(kf, SPA_EXPC[11], x);
  r = __builtin_fma(kf, SPA_EXPC[12], r);
  double q = SPA_EXPC[0];
#pragma unroll
  for (int i = 1; i < 11; i++) q = __builtin_fma(q, r, SPA_EXPC[i]);
  q = __builtin_fma(q, r, 0.5);
  p = __builtin_fma(q * r, r, r);
  return __builtin_ldexp(1.0 + p, k);
}
__device__ __forceinline__ double spa_div64(double a, double b) {   // b finite, normal, non-zero
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  const double s = a * y;
  return __builtin_fma(__builtin_fma(-s, b, a), y, s);
}
__device__ __forceinline__ double spa_logq64(double num, double den) {   // num, den > 0
  double mn = __builtin_amdgcn_frexp_mant(num), md = __builtin_amdgcn_frexp_mant(den);
  int k2 = __builtin_amdgcn_frexp_exp(num) - __builtin_amdgcn_frexp_exp(den);
  const int an = mn < md * 0.70710678118654752440 ? 1 : 0;
  const int ad = mn > md * 1.41421356237309504880 ? 1 : 0;
  mn = __builtin_ldexp(mn, an);
  md = __builtin_ldexp(md, ad);
  k2 += ad - an;
  const double s = spa_div64(mn - md, mn + md);               // (mn + md in (0.7, 3))
  const double z = s * s;
  double t = SPA_LOGC[0];
#pragma unroll
  for (int i = 1; i < 10; i++) t = __builtin_fma(t, z, SPA_LOGC[i]);
  const double s2 = s + s, k2f = (double)k2;
  const double res = __builtin_fma(s2 * z, t, __builtin_fma(k2f, SPA_LOGC[10], s2));
  return __builtin_fma(k2f, SPA_LOGC[11], res);
}
__device__ __forceinline__ double spa_logtanh2(double a) {            // a >= 0 (or NaN)
  double p; int k;
  const double u = spa_exp64(-a, p, k);
  const double qd = spa_div64(k == 0 ? -p : u + u, k == 0 ? 2.0 + p : 1.0 + u);
  const double t = k == 0 ? qd : 1.0 - qd;                             // tanh(a/2), rounded
  double res = spa_logq64(t, 1.0);
  res = t == 0.0 ? -__builtin_huge_val() : res;
  return a != a ? a : res;
}
__device__ __forceinline__ double spa_atanh2exp(double x) {           // x <= 0 (or NaN)
  double p; int k;
  const double e = spa_exp64(x, p, k);
  const double den = 1.0 - e;
  double res = spa_logq64(1.0 + e, den);
  res = den == 0.0 ? __builtin_huge_val() : res;
  return x != x ? x : res;
}
// Round 3: the register-resident binary64 check node works in the tanh domain -- the product of the
// others' tanh(a/2) in place of exp(sum of the others' log tanh), test_code_B_float.c:159-178 with the
// logarithm and the exponential that cancel left out: per edge one e^-a, one division, one logarithm of
// a quotient instead of two, two and two.  What is behaviour stays: t = tanh(a/2) is rounded to binary64
// exactly as before (so a message's weight is exactly 1, its log exactly 0, from a > 38.1), the product of
// the others is their exact leave-one-out product (prefix x suffix, no division by the own factor), so
// e = 1 -- atanh2(1) = +inf, the reference's NaN hazard one pass later -- exactly when every other edge is
// saturated, and 1 + e, 1 - e are rounded one by one as atanh2() rounds them; t = 0 (a zero message, whose
// log tanh is -inf in the reference) sends 0 to the other edges and NaN to its own, as -inf - (-inf) does.
#define LDPC_SPA_F64_PRODUCT 1
__device__ __forceinline__ double spa_tanh_half64(double a) {            // tanh(a / 2), a >= 0 (or NaN)
  double p; int k;
  const double u = spa_exp64(-a, p, k);
  const double qd = spa_div64(k == 0 ? -p : u + u, k == 0 ? 2.0 + p : 1.0 + u);
  return k == 0 ? qd : 1.0 - qd;
}
__device__ __forceinline__ double spa_atanh2_64(double e) {              // log((1 + e) / (1 - e)), 0 <= e <= 1 (or NaN)
  const double den = 1.0 - e;
  double res = spa_logq64(1.0 + e, den);
  return den == 0.0 ? __builtin_huge_val() : res;      // (a NaN runs through spa_logq64 as a NaN; 1 - NaN != 0)
}
#else   // the literal formulas on the device library (A/B builds)
#define LDPC_SPA_F64_PRODUCT 0
__device__ __forceinline__ double spa_logtanh2(double a) { return log(tanh(a * 0.5)); }
__device__ __forceinline__ double spa_atanh2exp(double x) { double e = exp(x); return log((1.0 + e) / (1.0 - e)); }
#endif

// Flogtanh / Fgallag (test_code_B.c:76-90)
__device__ __forceinline__ int tab_logtanh(const Tables &t, int a) {
  return ((unsigned)a >= (unsigned)TAB_LEVELS) ? 0 : t.flogtanh[a];
}
__device__ __forceinline__ int tab_gallag(const Tables &t, int x) {
  // x <= 0 by construction (sum of non-positive table values)
  if (x <= -TAB_LEVELS || x > 0) return 0;
  return t.fgallag[-x];
}

// Walks the rows of a check PF at a time: the PF rows are requested before the first is used, so a wavefront
// has PF KiB in flight instead of one (a row per round trip left the two-sweep kernels latency-bound).
template <int PF, typename Vec, typename RowFn, typename Body>
__device__ __forceinline__ void cn_for_rows(int dc, RowFn row, Body body) {
  for (int k0 = 0; k0 < dc; k0 += PF) {
    Vec xb[PF];
#pragma unroll
    for (int u = 0; u < PF; u++)
      if (k0 + u < dc) xb[u] = row(k0 + u);
#pragma unroll
    for (int u = 0; u < PF; u++)
      if (k0 + u < dc) body(k0 + u, xb[u]);
  }
}

// Processes one check for one lane (V frames).  `row(k)` returns the k-th v2c
// row of this check for this lane (LDS or global), `emit(k, vec)` stores c2v.
template <typename T, int ALG, typename RowFn, typename EmitFn>
__device__ __forceinline__ void cn_compute(int dc, unsigned sbits, const Tables &tab,
                                           RowFn row, EmitFn emit) {
  typedef typename MT<T>::vec vec;
  typedef typename MT<T>::acc acc;
  constexpr int V = MT<T>::V;
  constexpr int PF = (ALG == ALG_SPA && sizeof(T) == 8) ? 4 : 8;   // (the binary64 sum-product bodies are register-hungry)
  unsigned sg = sbits;                 // bit v: s_j ^ XOR of v2c signs  (sgnprod, :163-166)
  if constexpr (ALG == ALG_MS || ALG == ALG_MSQ) {
    CnStateMin<acc> st[V];
    const acc big = ALG == ALG_MSQ ? (acc)Q88_CAP200 : (acc)200;
#pragma unroll
    for (int v = 0; v < V; v++) st[v].init(big);
    cn_for_rows<PF, vec>(dc, row, [&](int, const vec &x) {
#pragma unroll
      for (int v = 0; v < V; v++) {
        const acc xv = (acc)x[v];
        const bool neg = xv < 0;
        st[v].add(neg ? -xv : xv);
        sg ^= (neg ? 1u : 0u) << v;
      }
    });
    cn_for_rows<PF, vec>(dc, row, [&](int k, const vec &x) {
      vec o;
#pragma unroll
      for (int v = 0; v < V; v++) {
        const acc xv = (acc)x[v];
        const bool neg = xv < 0;
        const acc mag = st[v].others(neg ? -xv : xv);
        const bool flip = (((sg >> v) & 1u) != 0) != neg;   // sgnprod != Sgntin[j][k]
        o[v] = (T)(flip ? -mag : mag);
      }
      emit(k, o);
    });
  } else if constexpr (ALG == ALG_SPA) {
    // test_code_B_float.c:159-178
    // logprod is summed in binary64 whatever the message type, as the reference does: the
    // leave-one-out below is a subtraction, and a binary32 sum absorbs the confident edges'
    // tiny terms next to one weak edge (difference 0 -> atanh2(1) = inf -> NaN next pass)
    double lp[V];
#pragma unroll
    for (int v = 0; v < V; v++) lp[v] = 0;
    cn_for_rows<PF, vec>(dc, row, [&](int, const vec &x) {
#pragma unroll
      for (int v = 0; v < V; v++) {
        const acc xv = (acc)x[v];
        const bool neg = xv < 0;
        lp[v] = lp[v] + (double)spa_logtanh2(neg ? -xv : xv);
        sg ^= (neg ? 1u : 0u) << v;
      }
    });
    cn_for_rows<PF, vec>(dc, row, [&](int k, const vec &x) {
      vec o;
#pragma unroll
      for (int v = 0; v < V; v++) {
        const acc xv = (acc)x[v];
        const bool neg = xv < 0;
        const acc tout = spa_atanh2exp((acc)(lp[v] - (double)spa_logtanh2(neg ? -xv : xv)));
        const bool flip = (((sg >> v) & 1u) != 0) != neg;
        o[v] = (T)(flip ? -tout : tout);
      }
      emit(k, o);
    });
  } else {   // ALG_TAB, test_code_B.c:195-213
    int lp[V];
#pragma unroll
    for (int v = 0; v < V; v++) lp[v] = 0;
    cn_for_rows<PF, vec>(dc, row, [&](int, const vec &x) {
#pragma unroll
      for (int v = 0; v < V; v++) {
        const int xv = (int)x[v];
        const bool neg = xv < 0;
        lp[v] += tab_logtanh(tab, neg ? -xv : xv);
        sg ^= (neg ? 1u : 0u) << v;
      }
    });
    cn_for_rows<PF, vec>(dc, row, [&](int k, const vec &x) {
      vec o;
#pragma unroll
      for (int v = 0; v < V; v++) {
        const int xv = (int)x[v];
        const bool neg = xv < 0;
        const int tout = tab_gallag(tab, lp[v] - tab_logtanh(tab, neg ? -xv : xv));
        const bool flip = (((sg >> v) & 1u) != 0) != neg;
        o[v] = (T)(flip ? -tout : tout);
      }
      emit(k, o);
    });
  }
}

// ----------------------------------------------------------- check-node pass
//
// One wavefront per block; the wave owns checks c, c+WPS, ... of ONE slab so
// that its syndrome-failure bits can be accumulated in a register and pushed
// with a single atomicOr row at the end.  The dc rows of a check are staged in
// a wave-private LDS region [rmax][64 lanes][16 B] (lane-private columns, so no
// barrier is needed): either through registers (DMA = false) or with the
// gfx950 direct global->LDS load (DMA = true, global_load_lds_dwordx4: the LDS
// destination is wave-uniform base + lane*16, which is exactly this layout).
//
// Fused into the same pass: the syndrome test of the hard decisions written by
// the preceding vn_kernel (enc() + HamDist(), test_code_min_sum_B.c:200-203).
template <typename T, int ALG, bool DMA>
__global__ __launch_bounds__(64) void cn_kernel(
    Graph g, int nslab, typename MT<T>::vec *__restrict__ msg,
    const uint8_t *__restrict__ hard, const uint8_t *__restrict__ synd,
    uint32_t *__restrict__ fail, Tables tab) {
  typedef typename MT<T>::vec vec;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  vec *lds = reinterpret_cast<vec *>(lds_raw);
  const int lane = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  const int b = w % nslab;
  const int c0 = w / nslab;
  const int wps = (int)gridDim.x / nslab;
  unsigned failacc = 0;

  for (int j = c0; j < g.m; j += wps) {
    const int p0 = g.row_ptr[j];
    const int dc = g.row_ptr[j + 1] - p0;
    vec *rows = msg + ((size_t)b * g.E + p0) * 64 + lane;
    const unsigned sb = synd ? synd[((size_t)b * g.m + j) * 64 + lane] : 0u;
    unsigned par = sb;
    const uint8_t *hrow = hard + (size_t)b * g.n * 64 + lane;
    if (DMA) {
      for (int k = 0; k < dc; k++)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(rows + (size_t)k * 64),
            (__attribute__((address_space(3))) void *)(lds + k * 64), 16, 0, 0);
      for (int k = 0; k < dc; k++) par ^= hrow[(size_t)g.col_idx[p0 + k] * 64];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      for (int k = 0; k < dc; k++) lds[k * 64 + lane] = rows[(size_t)k * 64];
      for (int k = 0; k < dc; k++) par ^= hrow[(size_t)g.col_idx[p0 + k] * 64];
    }
    failacc |= par;
    cn_compute<T, ALG>(
        dc, sb, tab, [&](int k) { return lds[k * 64 + lane]; },
        [&](int k, vec o) { rows[(size_t)k * 64] = o; });
  }
  if (failacc) atomicOr(&fail[b * 64 + lane], failacc);
}

// Same pass without LDS: rows are read twice from global memory (second read
// normally served by L2).  Used when rmax KiB does not fit in LDS, and as the
// A/B baseline for the staged kernel.
template <typename T, int ALG>
__global__ __launch_bounds__(256) void cn_reload_kernel(
    Graph g, int nslab, typename MT<T>::vec *__restrict__ msg,
    const uint8_t *__restrict__ hard, const uint8_t *__restrict__ synd,
    uint32_t *__restrict__ fail, Tables tab) {
  typedef typename MT<T>::vec vec;
  const int lane = threadIdx.x & 63;
  const int w = wave_id_uniform();
  const int nw = (int)((gridDim.x * blockDim.x) >> 6);
  const int b = w % nslab;
  const int c0 = w / nslab;
  const int wps = nw / nslab;
  unsigned failacc = 0;
  if (c0 >= wps) return;
  for (int j = c0; j < g.m; j += wps) {
    const int p0 = g.row_ptr[j];
    const int dc = g.row_ptr[j + 1] - p0;
    vec *rows = msg + ((size_t)b * g.E + p0) * 64 + lane;
    const unsigned sb = synd ? synd[((size_t)b * g.m + j) * 64 + lane] : 0u;
    unsigned par = sb;
    const uint8_t *hrow = hard + (size_t)b * g.n * 64 + lane;
#pragma unroll 8
    for (int k = 0; k < dc; k++) par ^= hrow[(size_t)g.col_idx[p0 + k] * 64];
    failacc |= par;
    cn_compute<T, ALG>(
        dc, sb, tab, [&](int k) { return rows[(size_t)k * 64]; },
        [&](int k, vec o) { rows[(size_t)k * 64] = o; });
  }
  if (failacc) atomicOr(&fail[b * 64 + lane], failacc);
}

// Single-read variant for min-sum (row weights <= 64): the first sweep keeps, per frame,
// min1, min2, WHICH row gave min1 and one sign bit per row; the outputs then follow from
// that state alone (+-min2 on the arg-min row, +-min1 elsewhere; equal minima give the same
// values whichever of them is named), so the second sweep only writes.  Traffic is exactly
// the algorithmic 2*E*b per frame and iteration.
template <typename T, int ALG>
__global__ __launch_bounds__(256) void cn_single_read_kernel(
    Graph g, int nslab, typename MT<T>::vec *__restrict__ msg,
    const uint8_t *__restrict__ hard, const uint8_t *__restrict__ synd,
    uint32_t *__restrict__ fail, Tables tab) {
  typedef typename MT<T>::vec vec;
  typedef typename MT<T>::acc acc;
  constexpr int V = MT<T>::V;
  static_assert(ALG == ALG_MS || ALG == ALG_MSQ, "single-read variant is min-sum only");
  const int lane = threadIdx.x & 63;
  const int w = wave_id_uniform();
  const int nw = (int)((gridDim.x * blockDim.x) >> 6);
  const int b = w % nslab;
  const int c0 = w / nslab;
  const int wps = nw / nslab;
  unsigned failacc = 0;
  if (c0 >= wps) return;
  const acc big = ALG == ALG_MSQ ? (acc)Q88_CAP200 : (acc)200;
  for (int j = c0; j < g.m; j += wps) {
    const int p0 = g.row_ptr[j];
    const int dc = g.row_ptr[j + 1] - p0;
    vec *rows = msg + ((size_t)b * g.E + p0) * 64 + lane;
    const unsigned sb = synd ? synd[((size_t)b * g.m + j) * 64 + lane] : 0u;
    unsigned par = sb, sg = sb;
    const uint8_t *hrow = hard + (size_t)b * g.n * 64 + lane;
    acc m1[V], m2[V];
    int arg[V];
    unsigned long long neg[V];
#pragma unroll
    for (int v = 0; v < V; v++) { m1[v] = big; m2[v] = big; arg[v] = -1; neg[v] = 0; }
    // PF rows (and the hard decisions of their variables) are requested before the first is used: one
    // row per round trip, as this loop first was, left a wavefront with 1 KiB in flight -- 4 MB on the
    // whole chip, where ~12 MB are needed to cover the memory latency at the bandwidth the pass could have
    // (round 3: PEG-16000 Q8.8 check pass 4.3 -> see profiles/r03_twopass_*.json)
    constexpr int PF = 8;
    for (int k0 = 0; k0 < dc; k0 += PF) {
      vec xb[PF];
      unsigned hb[PF];
#pragma unroll
      for (int u = 0; u < PF; u++)
        if (k0 + u < dc) {
          xb[u] = rows[(size_t)(k0 + u) * 64];
          hb[u] = hrow[(size_t)g.col_idx[p0 + k0 + u] * 64];
        }
#pragma unroll
      for (int u = 0; u < PF; u++)
        if (k0 + u < dc) {
          const int k = k0 + u;
          par ^= hb[u];
#pragma unroll
          for (int v = 0; v < V; v++) {
            const acc xv = (acc)xb[u][v];
            const bool ng = xv < 0;
            const acc a = ng ? -xv : xv;
            neg[v] |= (unsigned long long)(ng ? 1u : 0u) << k;
            sg ^= (ng ? 1u : 0u) << v;
            const acc hi = a > m1[v] ? a : m1[v];
            m2[v] = hi < m2[v] ? hi : m2[v];
            if (a < m1[v]) { m1[v] = a; arg[v] = k; }
          }
        }
    }
    failacc |= par;
    for (int k = 0; k < dc; k++) {
      vec o;
#pragma unroll
      for (int v = 0; v < V; v++) {
        const acc mag = k == arg[v] ? m2[v] : m1[v];
        const bool ng = ((neg[v] >> k) & 1ull) != 0;
        const bool flip = (((sg >> v) & 1u) != 0) != ng;   // sgnprod != Sgntin[j][k]
        o[v] = (T)(flip ? -mag : mag);
      }
      rows[(size_t)k * 64] = o;
    }
  }
  if (failacc) atomicOr(&fail[b * 64 + lane], failacc);
}

// (the 16-row sum-product instantiations ask for four waves per SIMD: left alone, the scheduler spreads the
// long function bodies over 163-230 registers, two or three waves; asked, it needs 126-128 -- 1057.244 SPA_F32
// 4.65 -> 5.78 Gbit/s, Frolov 2.51 -> 3.17, binary64 +4-6 %.  The 8-row ones are left alone: DVB-S2 unchanged,
// (204,102) 4-5 % slower with the attribute)
#ifndef LDPC_CN_REG_WAVES
#define LDPC_CN_REG_WAVES __attribute__((amdgpu_waves_per_eu((ALG == ALG_SPA && (RB >= 16 || sizeof(T) == 8)) ? 4 : 1)))
#endif
// Register-resident variant for row weights <= RB: the whole check lives in
// VGPRs between the load and the store (no LDS, no second read).
template <typename T, int ALG, int RB>
__global__ __launch_bounds__(256) LDPC_CN_REG_WAVES void cn_reg_kernel(
    Graph g, int nslab, typename MT<T>::vec *__restrict__ msg,
    const uint8_t *__restrict__ hard, const uint8_t *__restrict__ synd,
    uint32_t *__restrict__ fail, Tables tab) {
  typedef typename MT<T>::vec vec;
  typedef typename MT<T>::acc acc;
  constexpr int V = MT<T>::V;
  static_assert(ALG == ALG_MS || ALG == ALG_MSQ || ALG == ALG_SPA, "register variant: min-sum and float sum-product");
  const int lane = threadIdx.x & 63;
  const int w = wave_id_uniform();
  const int nw = (int)((gridDim.x * blockDim.x) >> 6);
  const int b = w % nslab;
  const int c0 = w / nslab;
  const int wps = nw / nslab;
  unsigned failacc = 0;
  if (c0 >= wps) return;
  for (int j = c0; j < g.m; j += wps) {
    const int p0 = g.row_ptr[j];
    const int dc = g.row_ptr[j + 1] - p0;
    vec *rows = msg + ((size_t)b * g.E + p0) * 64 + lane;
    const unsigned sb = synd ? synd[((size_t)b * g.m + j) * 64 + lane] : 0u;
    unsigned par = sb, sg = sb;
    const uint8_t *hrow = hard + (size_t)b * g.n * 64 + lane;
    vec x[RB];
#pragma unroll
    for (int k = 0; k < RB; k++)
      if (k < dc) {
        x[k] = rows[(size_t)k * 64];
        par ^= hrow[(size_t)g.col_idx[p0 + k] * 64];
      }
    failacc |= par;
    if constexpr (ALG == ALG_SPA) {
      // test_code_B_float.c:159-178 with log tanh(|v2c|/2) of every row evaluated once and kept
      // (in the row's own registers; its sign goes to a bit mask): the same values, sums and
      // differences as the two-sweep kernels, half the transcendentals, one read of the rows
      static_assert(RB * V <= 64, "sign mask");
      unsigned long long negm = 0;
      if constexpr (std::is_same<T, double>::value && LDPC_SPA_F64_PRODUCT) {
        // tanh domain (see spa_tanh_half64): one frame of the lane at a time, the rows' registers take
        // tanh(|v2c| / 2) and then the outgoing message in place
#pragma unroll
        for (int v = 0; v < V; v++) {
          double pre[RB];        // product of the rows before k
          double run = 1.0;
#pragma unroll
          for (int k = 0; k < RB; k++)
            if (k < dc) {
              const double xv = x[k][v];
              const bool neg = xv < 0;
              const double t = spa_tanh_half64(neg ? -xv : xv);
              pre[k] = run;
              run = run * t;
              x[k][v] = t;
              negm |= (unsigned long long)(neg ? 1u : 0u) << (k * V + v);
              sg ^= (neg ? 1u : 0u) << v;
            }
          double suf = 1.0;      // product of the rows after k
#pragma unroll
          for (int k = RB - 1; k >= 0; k--)
            if (k < dc) {
              const double t = x[k][v];
              double tout = spa_atanh2_64(pre[k] * suf);
              tout = t == 0.0 ? __builtin_nan("") : tout;            // the reference's -inf - (-inf)
              suf = suf * t;
              const unsigned flip = ((unsigned)(negm >> (k * V + v)) ^ (sg >> v)) & 1u;       // sgnprod != the row's own sign
              x[k][v] = __hiloint2double(__double2hiint(tout) ^ (int)(flip << 31), __double2loint(tout));
            }
        }
#pragma unroll
        for (int k = 0; k < RB; k++)
          if (k < dc) rows[(size_t)k * 64] = x[k];
        continue;
      }
      double lp[V];              // (binary64 sum: see cn_compute)
#pragma unroll
      for (int v = 0; v < V; v++) lp[v] = 0;
#pragma unroll
      for (int k = 0; k < RB; k++)
        if (k < dc) {
#pragma unroll
          for (int v = 0; v < V; v++) {
            const acc xv = (acc)x[k][v];
            const bool neg = xv < 0;
            const acc l = spa_logtanh2(neg ? -xv : xv);
            lp[v] = lp[v] + (double)l;
            x[k][v] = (T)l;
            negm |= (unsigned long long)(neg ? 1u : 0u) << (k * V + v);
            sg ^= (neg ? 1u : 0u) << v;
          }
        }
#pragma unroll
      for (int k = 0; k < RB; k++)
        if (k < dc) {
          vec o;
#pragma unroll
          for (int v = 0; v < V; v++) {
            const acc tout = spa_atanh2exp((acc)(lp[v] - (double)x[k][v]));
            const bool neg = ((negm >> (k * V + v)) & 1ull) != 0;
            const bool flip = (((sg >> v) & 1u) != 0) != neg;
            o[v] = (T)(flip ? -tout : tout);
          }
          rows[(size_t)k * 64] = o;
        }
      continue;
    }
    CnStateMin<acc> st[V];
    const acc big = ALG == ALG_MSQ ? (acc)Q88_CAP200 : (acc)200;
#pragma unroll
    for (int v = 0; v < V; v++) st[v].init(big);
#pragma unroll
    for (int k = 0; k < RB; k++)
      if (k < dc) {
#pragma unroll
        for (int v = 0; v < V; v++) {
          const acc xv = (acc)x[k][v];
          const bool neg = xv < 0;
          st[v].add(neg ? -xv : xv);
          sg ^= (neg ? 1u : 0u) << v;
        }
      }
#pragma unroll
    for (int k = 0; k < RB; k++)
      if (k < dc) {
        vec o;
#pragma unroll
        for (int v = 0; v < V; v++) {
          const acc xv = (acc)x[k][v];
          const bool neg = xv < 0;
          const acc mag = st[v].others(neg ? -xv : xv);
          const bool flip = (((sg >> v) & 1u) != 0) != neg;
          o[v] = (T)(flip ? -mag : mag);
        }
        rows[(size_t)k * 64] = o;
      }
  }
  if (failacc) atomicOr(&fail[b * 64 + lane], failacc);
}

// Binary64 sum-product for row weights 17 ... 64 in the tanh domain (round 3; see spa_tanh_half64 and the
// product branch of cn_reg_kernel, which holds a whole check of up to 16 rows).  A heavier check is walked in
// chunks of CH rows:
//   1. chunks 0 .. n-2: rows read, t_k = tanh(|v2c_k| / 2) evaluated, only the chunk's product kept;
//   2. the last chunk: read, evaluated and KEPT (t_k in the rows' registers, prefix products beside them);
//   3. F_c = the product of all OTHER chunks' products; the kept chunk is finished: message of row k =
//      2 atanh((F_c * prefix_k) * suffix_k), written over the row;
//   4. chunks 0 .. n-2 again: rows read a second time, t_k evaluated a second time, finished the same way.
// (2 - 1/n) tanh evaluations and one 2 atanh per edge value, where the two-sweep reload kernel spends two
// log tanh and one 2 atanh(exp): about 155 vector instructions per edge value instead of 270.  Hazards as in
// the reference: a row whose own t is 0 gets NaN (-inf - -inf), the others of its check 0; a product of 1
// gives +inf.
template <int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void cn_spa64_chunk_kernel(
    Graph g, int nslab, MT<double>::vec *__restrict__ msg, const uint8_t *__restrict__ hard,
    const uint8_t *__restrict__ synd, uint32_t *__restrict__ fail) {
  typedef MT<double>::vec vec;
  constexpr int V = MT<double>::V;
  constexpr int NCH = 64 / CH;           // chunks of the heaviest check served
  const int lane = threadIdx.x & 63;
  const int w = wave_id_uniform();
  const int nw = (int)((gridDim.x * blockDim.x) >> 6);
  const int b = w % nslab;
  const int c0 = w / nslab;
  const int wps = nw / nslab;
  unsigned failacc = 0;
  if (c0 >= wps) return;
  for (int j = c0; j < g.m; j += wps) {
    const int p0 = g.row_ptr[j];
    const int dc = g.row_ptr[j + 1] - p0;
    vec *rows = msg + ((size_t)b * g.E + p0) * 64 + lane;
    const unsigned sb = synd ? synd[((size_t)b * g.m + j) * 64 + lane] : 0u;
    unsigned par = sb, sg = sb;
    const uint8_t *hrow = hard + (size_t)b * g.n * 64 + lane;
    const int nch = (dc + CH - 1) / CH;
    double P[NCH][V];                    // chunk products
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
      for (int v = 0; v < V; v++) P[c][v] = 1.0;
    vec x[CH];
    // rows k0 .. k0+cnt-1 -> x (all requested before the first is used); the hard decisions' parity on the first visit
    auto load = [&](int k0, int cnt, bool first) {
#pragma unroll
      for (int u = 0; u < CH; u++)
        if (u < cnt) {
          x[u] = rows[(size_t)(k0 + u) * 64];
          if (first) par ^= hrow[(size_t)g.col_idx[p0 + k0 + u] * 64];
        }
    };
    // x: v2c -> t = tanh(|v2c| / 2) with the row's sign kept in the sign bit of t (t >= 0, so the bit is free);
    // returns the chunk's product; signs into sg on the first visit
    auto forward = [&](int cnt, bool first, double prod[V]) {
#pragma unroll
      for (int v = 0; v < V; v++) {
        double run = 1.0;
#pragma unroll
        for (int u = 0; u < CH; u++)
          if (u < cnt) {
            const double xv = x[u][v];
            const bool neg = xv < 0;
            const double t = spa_tanh_half64(neg ? -xv : xv);
            run = run * t;
            if (first) sg ^= (neg ? 1u : 0u) << v;
            x[u][v] = __hiloint2double(__double2hiint(t) | (neg ? (int)0x80000000 : 0), __double2loint(t));
          }
        prod[v] = run;
      }
    };
    // x: t -> the outgoing messages, stored.  The prefix products are rebuilt here from the kept t (a multiplication
    // per row: cheaper than 2 * CH registers held across the other chunks' work)
    auto finish = [&](int k0, int cnt, const double F[V]) {
#pragma unroll
      for (int v = 0; v < V; v++) {
        double pre[CH];
        double run = F[v];
#pragma unroll
        for (int u = 0; u < CH; u++)
          if (u < cnt) {
            pre[u] = run;
            run = run * __hiloint2double(__double2hiint(x[u][v]) & 0x7fffffff, __double2loint(x[u][v]));
          }
        double suf = 1.0;
#pragma unroll
        for (int u = CH - 1; u >= 0; u--)
          if (u < cnt) {
            const unsigned hi = (unsigned)__double2hiint(x[u][v]);
            const double t = __hiloint2double((int)(hi & 0x7fffffffu), __double2loint(x[u][v]));
            double tout = spa_atanh2_64(pre[u] * suf);
            tout = t == 0.0 ? __builtin_nan("") : tout;            // the reference's -inf - (-inf)
            suf = suf * t;
            const unsigned flip = ((hi >> 31) ^ (sg >> v)) & 1u;   // sgnprod != the row's own sign
            x[u][v] = __hiloint2double(__double2hiint(tout) ^ (int)(flip << 31), __double2loint(tout));
          }
      }
#pragma unroll
      for (int u = 0; u < CH; u++)
        if (u < cnt) rows[(size_t)(k0 + u) * 64] = x[u];
    };
    // 1. products of the chunks before the last
    for (int c = 0; c + 1 < nch; c++) {
      load(c * CH, CH, true);
      double pr[V];
#pragma unroll
      for (int v = 0; v < V; v++) pr[v] = 1.0;
#pragma unroll
      for (int u = 0; u < CH; u++)
#pragma unroll
        for (int v = 0; v < V; v++) {
          const double xv = x[u][v];
          const bool neg = xv < 0;
          pr[v] = pr[v] * spa_tanh_half64(neg ? -xv : xv);
          sg ^= (neg ? 1u : 0u) << v;
        }
#pragma unroll
      for (int cc = 0; cc < NCH; cc++)
        if (cc == c) {
#pragma unroll
          for (int v = 0; v < V; v++) P[cc][v] = pr[v];
        }
    }
    // 2. the last chunk, kept
    const int lastc = nch - 1, lcnt = dc - lastc * CH;
    load(lastc * CH, lcnt, true);
    failacc |= par;
    double plast[V];
    forward(lcnt, true, plast);
#pragma unroll
    for (int cc = 0; cc < NCH; cc++)
      if (cc == lastc) {
#pragma unroll
        for (int v = 0; v < V; v++) P[cc][v] = plast[v];
      }
    // 3. / 4. every chunk finished with the product of the others
    for (int c = nch - 1; c >= 0; c--) {
      double F[V];
#pragma unroll
      for (int v = 0; v < V; v++) F[v] = 1.0;
#pragma unroll
      for (int cc = 0; cc < NCH; cc++)
        if (cc != c && cc < nch) {
#pragma unroll
          for (int v = 0; v < V; v++) F[v] = F[v] * P[cc][v];
        }
      if (c != nch - 1) {
        load(c * CH, CH, false);
        double dummy[V];
        forward(CH, false, dummy);
      }
      finish(c * CH, c == nch - 1 ? lcnt : CH, F);
    }
  }
  if (failacc) atomicOr(&fail[b * 64 + lane], failacc);
}

// Table sum-product with both tables in LDS (engine.h: kTabc* has the format, tables.cpp builds and
// checks the image).  The gathers of the other kernels go to 32 MiB of tables through the vector L1:
// 64 requests per wavefront instruction, three instructions per edge and frame.  Here a workgroup
// copies the 153 KiB image into LDS once and a table value costs three LDS reads and ~18 vector
// instructions.  Same sums, same differences, same table values: bit-exact (test_code_B.c:195-213).
#define TABC_SMALL 2048
#define TABC_SPLIT 16384
#define TABC_BLOCKS 10336
#define TABC_OFF_B 8208
#define TABC_OFF_C (TABC_OFF_B + 28688)
#define TABC_TABLE (TABC_OFF_C + TABC_BLOCKS * 4)
#define TABC_BYTES (2 * TABC_TABLE)

// |table[i]| from one table's image at t; i as unsigned, so a negative argument is "beyond the table"
// (0) like the reference's range tests.  The three arrays are read at clamped indices and hold 0 where
// they clamp, so the value is their sum: 15 vector instructions and 3 LDS reads, no branch, no select.
__device__ __forceinline__ int tabc_mag(const unsigned char *t, unsigned i) {
  const unsigned ia = i < (unsigned)TABC_SMALL ? i : (unsigned)TABC_SMALL;
  unsigned ib = i > (unsigned)TABC_SMALL - 1u ? i : (unsigned)TABC_SMALL - 1u;
  ib = ib < (unsigned)TABC_SPLIT ? ib : (unsigned)TABC_SPLIT;   // (v_med3_u32)
  unsigned j = i - (unsigned)TABC_SPLIT;
  j = j < 16u * TABC_BLOCKS - 1u ? j : 16u * TABC_BLOCKS - 1u;
  const unsigned a = *reinterpret_cast<const uint32_t *>(t + 4u * ia);
  const unsigned bq = *reinterpret_cast<const uint16_t *>(t + (TABC_OFF_B - 2 * (TABC_SMALL - 1)) + 2u * ib);
  const unsigned w = *reinterpret_cast<const uint32_t *>(t + TABC_OFF_C + ((j >> 2) & ~3u));
  return (int)(a + bq + (w >> 16)) - __popc(__builtin_amdgcn_ubfe(w, 0u, (j & 15u) + 1u));
}

// CH rows of a check are in registers at a time.  Everything is branch-free inside a chunk so that the
// compiler can interleave the CH * 4 independent lookups (an LDS read is ~100 cycles away and one
// workgroup per CU is all the tables leave room for): a slot beyond the row weight loads the check's
// LAST row again, is masked out of the sums, and stores that row's (correct) output a second time.
//   dc <= CH:  one chunk; log tanh of every input is looked up once and kept, with the sign in bit 31
//   dc >  CH:  (CH <= 16 only) two sweeps over the rows in double-buffered chunks, rows read twice
//   ONE:       the host guarantees row weights <= CH (the chunked path, and its registers, are not built)
template <int CH, int NT, bool ONE = false>
__global__ __launch_bounds__(NT) void cn_tab_lds_kernel(
    Graph g, int nslab, int wps, MT<int>::vec *__restrict__ msg, const uint8_t *__restrict__ hard,
    const uint8_t *__restrict__ synd, uint32_t *__restrict__ fail, const uint32_t *__restrict__ image) {
  typedef MT<int>::vec vec;
  constexpr int V = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 *src = reinterpret_cast<const u4 *>(image);
    u4 *dst = reinterpret_cast<u4 *>(lds_raw);
    for (int t = threadIdx.x; t < TABC_BYTES / 16; t += NT) dst[t] = src[t];
  }
  __syncthreads();
  const unsigned char *tL = lds_raw, *tG = lds_raw + TABC_TABLE;
  const int lane = threadIdx.x & 63;
  const int w = wave_id_uniform();
  const int b = w % nslab;
  const int c0 = w / nslab;
  unsigned failacc = 0;
  if (c0 >= wps) return;      // (after the only barrier)
  const uint8_t *hrow = hard + (size_t)b * g.n * 64 + lane;
  for (int j = c0; j < g.m; j += wps) {
    const int p0 = g.row_ptr[j];
    const int dc = g.row_ptr[j + 1] - p0;
    vec *rows = msg + ((size_t)b * g.E + p0) * 64 + lane;
    const unsigned sb = synd ? synd[((size_t)b * g.m + j) * 64 + lane] : 0u;
    unsigned par = sb;
    if (dc <= 0) { failacc |= par; continue; }
    int lp[V];                 // -logprod: the table values are <= 0, their magnitudes are summed
    int sg[V];                 // bit 31: s_j ^ XOR of the v2c signs (sgnprod, test_code_B.c:196-199)
#pragma unroll
    for (int v = 0; v < V; v++) { lp[v] = 0; sg[v] = (int)(sb << (31 - v)); }

    auto load = [&](int k0, vec (&x)[CH], unsigned (&hb)[CH], bool with_hard) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int k = k0 + c < dc ? k0 + c : dc - 1;
        x[c] = rows[(size_t)k * 64];
        if (with_hard) hb[c] = hrow[(size_t)g.col_idx[p0 + k] * 64];
      }
    };
    // first sweep over a chunk: sums and signs; PACK: the row's registers keep |log tanh| << 1 | sign
    auto sweep1 = [&](int k0, vec (&x)[CH], unsigned (&hb)[CH], auto pack) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const unsigned m = k0 + c < dc ? ~0u : 0u;    // wave-uniform
        par ^= hb[c] & m;
#pragma unroll
        for (int v = 0; v < V; v++) {
          const int xv = x[c][v];
          const int s = xv >> 31;
          const int l = tabc_mag(tL, (unsigned)((xv ^ s) - s));
          lp[v] += (int)((unsigned)l & m);
          sg[v] ^= (int)((unsigned)xv & m);
          if (decltype(pack)::value) x[c][v] = (int)(((unsigned)xv & 0x80000000u) | (unsigned)l);
        }
        // (four rows' lookups in flight at a time: without the fence the scheduler interleaves all CH rows
        // and needs ~220 registers, i.e. 8 waves per CU)
        if (CH > 4 && (c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto store = [&](int k0, int c, vec o) {
      const int k = k0 + c < dc ? k0 + c : dc - 1;
      rows[(size_t)k * 64] = o;
    };

    if (ONE || dc <= CH) {
      vec x[CH];
      unsigned hb[CH];
      load(0, x, hb, true);
      sweep1(0, x, hb, std::true_type());
      failacc |= par;
#pragma unroll
      for (int c = 0; c < CH; c++) {
        vec o;
#pragma unroll
        for (int v = 0; v < V; v++) {
          const int pk = x[c][v];
          const int tout = tabc_mag(tG, (unsigned)(lp[v] - (pk & 0x7fffffff)));
          const int s = (pk ^ sg[v]) >> 31;             // sgnprod != Sgntin[j][k]
          o[v] = (tout ^ s) - s;
        }
        store(0, c, o);
        if (CH > 4 && (c & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (CH <= 16 && !ONE) {
      vec xa[CH], xb[CH];
      unsigned ha[CH], hbb[CH];
      load(0, xa, ha, true);
      int ka = 0;                      // the chunk xa holds when the sweep ends (xb: the one after it, if any)
      for (int k0 = 0; k0 < dc; k0 += 2 * CH) {
        const bool second = k0 + CH < dc;
        if (second) load(k0 + CH, xb, hbb, true);
        sweep1(k0, xa, ha, std::false_type());
        if (second) {
          if (k0 + 2 * CH < dc) load(k0 + 2 * CH, xa, ha, true);
          sweep1(k0 + CH, xb, hbb, std::false_type());
        }
        ka = k0;
      }
      failacc |= par;
      auto sweep2 = [&](int k0, vec (&x)[CH]) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
          vec o;
#pragma unroll
          for (int v = 0; v < V; v++) {
            const int xv = x[c][v];
            const int sx = xv >> 31;
            const int l = tabc_mag(tL, (unsigned)((xv ^ sx) - sx));
            const int tout = tabc_mag(tG, (unsigned)(lp[v] - l));
            const int s = (xv ^ sg[v]) >> 31;
            o[v] = (tout ^ s) - s;
          }
          store(k0, c, o);
        }
      };
      // second sweep: the last one or two chunks are still in registers (no second read for them:
      // all of it for row weights <= 16); the chunks before them are read again, double-buffered
      const bool tail_b = ka + CH < dc;
      sweep2(ka, xa);
      if (ka > 0) load(0, xa, ha, false);
      if (tail_b) sweep2(ka + CH, xb);
      for (int k0 = 0; k0 < ka; k0 += 2 * CH) {        // (ka is a multiple of 2 CH)
        load(k0 + CH, xb, hbb, false);
        sweep2(k0, xa);
        if (k0 + 2 * CH < ka) load(k0 + 2 * CH, xa, ha, false);
        sweep2(k0 + CH, xb);
      }
    }
  }
  if (failacc) atomicOr(&fail[b * 64 + lane], failacc);
}

// Closing syndrome pass: sout = H * hard (the reference's tmp_s) and
// fail |= sout ^ target.  enc() + HamDist(), test_code_min_sum_B.c:200-203.
__global__ __launch_bounds__(256) void synd_kernel(Graph g, int nslab,
                                                   const uint8_t *__restrict__ hard,
                                                   const uint8_t *__restrict__ synd,
                                                   uint8_t *__restrict__ sout,
                                                   uint32_t *__restrict__ fail) {
  const int lane = threadIdx.x & 63;
  const int w = wave_id_uniform();
  const int nw = (int)((gridDim.x * blockDim.x) >> 6);
  const int b = w % nslab;
  const int c0 = w / nslab;
  const int wps = nw / nslab;
  unsigned failacc = 0;
  if (c0 >= wps) return;
  const uint8_t *hrow = hard + (size_t)b * g.n * 64 + lane;
  for (int j = c0; j < g.m; j += wps) {
    const int p0 = g.row_ptr[j];
    const int dc = g.row_ptr[j + 1] - p0;
    unsigned par = 0;
    for (int k = 0; k < dc; k++) par ^= hrow[(size_t)g.col_idx[p0 + k] * 64];
    const size_t o = ((size_t)b * g.m + j) * 64 + lane;
    if (sout) sout[o] = (uint8_t)par;
    failacc |= par ^ (synd ? synd[o] : 0u);
  }
  if (failacc) atomicOr(&fail[b * 64 + lane], failacc);
}

// ------------------------------------------------------ per-frame bookkeeping

// done/fail/iters initialisation for a pass of `nframes` frames.
__global__ void init_state_kernel(int nslab, int V, int nframes, uint32_t *fail, uint8_t *done,
                                  int32_t *iters) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nslab * 64) return;
  unsigned invalid = 0;
  for (int v = 0; v < V; v++) {
    iters[idx * V + v] = -1;
    if (idx * V + v >= nframes) invalid |= 1u << v;
  }
  fail[idx] = 0;
  done[idx] = (uint8_t)invalid;   // padding frames are frozen from the start
}

// After a syndrome test of the hard decisions of iteration `it`: frames whose
// fail bits stayed clear have converged at `it` (dec() returns 0 there).
// ignore != 0 discards the test (the float decoders have no iteration-0 check).
__global__ __launch_bounds__(1024) void mark_kernel(int nslab, int V, int nframes, int it, int ignore, uint32_t *fail,
                                                    uint8_t *done, int32_t *iters, int32_t *remaining, int32_t *remaining_host) {
  // one workgroup walks all lanes: the count of open frames is then a plain store -- to the device slot and, when
  // the host polls it (early exit), straight into its pinned word: no copy packet between two iterations' kernels
  int rem = 0;
  for (int idx = threadIdx.x; idx < nslab * 64; idx += blockDim.x) {
    const unsigned f = fail[idx];
    fail[idx] = 0;
    unsigned d = done[idx];
    const unsigned all = (1u << V) - 1u;
    if (!ignore) {
      const unsigned newly = ~f & ~d & all;
      if (newly) {
        for (int v = 0; v < V; v++)
          if ((newly >> v) & 1u) iters[idx * V + v] = it;
        d |= newly;
        done[idx] = (uint8_t)d;
      }
    }
    rem += __popc(~d & all);
  }
  for (int o = 32; o > 0; o >>= 1) rem += __shfl_down(rem, o);
  __shared__ int part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = rem;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); k++) tot += part[k];
    *remaining = tot;
    if (remaining_host) *(volatile int32_t *)remaining_host = tot;
  }
}

// ------------------------------------------------------------ message trace
//
// Single-frame read-outs for the Verilog trace of test_code_min_sum_B_fixed.c.
// trace_v2c: what :250-256 prints, the UNSATURATED sum - qin[v][slot] of every
// edge, computed from the c2v messages currently in `msg` (first != 0: they are
// all zero, memset at :196).  trace_msg: the messages as stored (:293).
template <typename T>
__global__ void trace_v2c_kernel(Graph g, const typename MT<T>::vec *__restrict__ q0,
                                 const typename MT<T>::vec *__restrict__ msg, int frame, int first,
                                 int32_t *__restrict__ out) {
  constexpr int V = MT<T>::V;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  const int b = frame / (64 * V), lane = (frame % (64 * V)) / V, v = frame % V;
  const int p0 = g.col_ptr[i], dv = g.col_ptr[i + 1] - p0;
  int sum = (int)q0[((size_t)b * g.n + i) * 64 + lane][v];
  if (!first)
    for (int k = 0; k < dv; k++) sum += (int)msg[((size_t)b * g.E + g.edge_of[p0 + k]) * 64 + lane][v];
  for (int k = 0; k < dv; k++) {
    const int e = g.edge_of[p0 + k];
    out[e] = first ? sum : sum - (int)msg[((size_t)b * g.E + e) * 64 + lane][v];
  }
}

template <typename T>
__global__ void trace_msg_kernel(Graph g, const typename MT<T>::vec *__restrict__ msg, int frame,
                                 int32_t *__restrict__ out) {
  constexpr int V = MT<T>::V;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= g.E) return;
  const int b = frame / (64 * V), lane = (frame % (64 * V)) / V, v = frame % V;
  out[e] = (int)msg[((size_t)b * g.E + e) * 64 + lane][v];
}

// ---------------------------------------------------------- ingest / egress

// float_to_fix() (test_code_min_sum_B_fixed.c:43-53) to raw Q8.8
__device__ __forceinline__ int q88_from_real(double x) {
  double r = round(x * 256.0);
  if (r > 127.0 * 256.0) r = 127.0 * 256.0;
  if (r < -128.0 * 256.0) r = -128.0 * 256.0;
  return (int)r;
}
// float2fix() (test_code_B.c:36-43) to raw Q8.14
__device__ __forceinline__ int q14_from_real(double x) {
  if (x >= 0) return (int)(x * 16384.0 + 0.5);
  return -(int)((-x) * 16384.0 + 0.5);
}

template <typename InT, typename T, int ALG>
__device__ __forceinline__ T ingest_convert(InT x) {
  if constexpr (ALG == ALG_MSQ) {
    if constexpr (std::is_integral<InT>::value) return (T)x;   // raw Q8.8
    else return (T)q88_from_real((double)x);
  } else if constexpr (ALG == ALG_TAB) {
    if constexpr (std::is_integral<InT>::value) return (T)x;   // raw Q8.14
    else return (T)q14_from_real((double)x);
  } else {
    return (T)x;
  }
}

// in [nframes][n] (frame-major, as the callers of dec() hold q0) ->
// q0 [slab][n][64] vec.  Block = 256 threads, tile = 32 variables x one slab, transposed through LDS.
// The tile's row pitch puts the 32 variables a half-wavefront writes for one frame into 32 different banks
// (pitch = one element more than a multiple of 64 dwords; the two frames of a wavefront are four apart, so
// they do not share a dword either): with the 16-byte-aligned pitch of rounds 1-2 the column writes met 4-way
// bank conflicts and the pass ran at 1.8 TB/s.  Rows are then only 4-byte aligned and are read back as dwords.
template <typename InT, typename T, int ALG>
__global__ __launch_bounds__(256) void ingest_kernel(const InT *__restrict__ in, int nframes, int n,
                                                     typename MT<T>::vec *__restrict__ q0) {
  typedef typename MT<T>::vec vec;
  constexpr int V = MT<T>::V;
  constexpr int SF = 64 * V;
  constexpr int LD = SF + (sizeof(T) < 4 ? 4 / (int)sizeof(T) : 1);
  static_assert(sizeof(vec) == 16, "a lane's row is four dwords");
  __shared__ __attribute__((aligned(16))) T tile[32 * LD + 8];
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * 32;
  const int tid = threadIdx.x;
  {
    const int ii = tid & 31;
    const int i = i0 + ii;
    // eight loads in flight per lane (one load, one LDS store at a time left the pass latency-bound: 89 % of the
    // wave cycles waiting)
    static_assert(SF % 64 == 0, "eight frames per step, eight steps per batch");
    for (int f0 = (tid >> 6) + 4 * ((tid >> 5) & 1); f0 < SF; f0 += 64) {
      InT raw[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const long frame = (long)b * SF + f0 + 8 * u;
        raw[u] = (frame < nframes && i < n) ? in[frame * n + i] : (InT)0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) tile[ii * LD + f0 + 8 * u] = ingest_convert<InT, T, ALG>(raw[u]);
    }
  }
  __syncthreads();
  {
    const int lane = tid & 63;
    for (int r = tid >> 6; r < 32; r += 4) {
      const int i = i0 + r;
      const uint32_t *src = reinterpret_cast<const uint32_t *>(&tile[r * LD + lane * V]);
      uint4 w;
      w.x = src[0]; w.y = src[1]; w.z = src[2]; w.w = src[3];
      if (i < n) q0[((size_t)b * n + i) * 64 + lane] = __builtin_bit_cast(vec, w);
    }
  }
}

// s [nframes][m] one bit per byte -> synd [slab][m][64] u8 (bit v = frame l*V+v)
template <int V>
__global__ __launch_bounds__(256) void ingest_synd_kernel(const uint8_t *__restrict__ s, int nframes,
                                                          int m, uint8_t *__restrict__ synd) {
  constexpr int SF = 64 * V;
  constexpr int LD = 68;
  __shared__ uint8_t tile[SF * LD];
  const int b = blockIdx.y;
  const int j0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  {
    const int jj = tid & 63;
    for (int f = tid >> 6; f < SF; f += 4) {
      const long frame = (long)b * SF + f;
      uint8_t val = 0;
      if (frame < nframes && j0 + jj < m) val = s[frame * m + j0 + jj] & 1;
      tile[f * LD + jj] = val;
    }
  }
  __syncthreads();
  {
    const int lane = tid & 63;
    for (int r = tid >> 6; r < 64; r += 4) {
      if (j0 + r < m) {
        unsigned byte = 0;
#pragma unroll
        for (int v = 0; v < V; v++) byte |= (unsigned)tile[(lane * V + v) * LD + r] << v;
        synd[((size_t)b * m + j0 + r) * 64 + lane] = (uint8_t)byte;
      }
    }
  }
}

// packed [slab][len][64] u8 -> out [nframes][len] one bit per byte
template <int V>
__global__ __launch_bounds__(256) void egress_bits_kernel(const uint8_t *__restrict__ packed, int nframes,
                                                          int len, uint8_t *__restrict__ out) {
  constexpr int SF = 64 * V;
  constexpr int LD = 68;
  __shared__ uint8_t tile[64 * LD];
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  {
    const int lane = tid & 63;
    for (int r = tid >> 6; r < 64; r += 4)
      tile[r * LD + lane] = (i0 + r < len) ? packed[((size_t)b * len + i0 + r) * 64 + lane] : 0;
  }
  __syncthreads();
  {
    const int ii = tid & 63;
    for (int f = tid >> 6; f < SF; f += 4) {
      const long frame = (long)b * SF + f;
      if (frame < nframes && i0 + ii < len)
        out[frame * len + i0 + ii] = (tile[ii * LD + f / V] >> (f % V)) & 1;
    }
  }
}

// post [slab][n][64] vec -> out [nframes][n] float
template <typename T, int ALG>
__global__ __launch_bounds__(256) void egress_post_kernel(const typename MT<T>::vec *__restrict__ post,
                                                          int nframes, int n, float *__restrict__ out) {
  typedef typename MT<T>::vec vec;
  constexpr int V = MT<T>::V;
  constexpr int SF = 64 * V;
  constexpr int LD = SF + V;
  __shared__ __attribute__((aligned(16))) T tile[32 * LD];
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * 32;
  const int tid = threadIdx.x;
  {
    const int lane = tid & 63;
    for (int r = tid >> 6; r < 32; r += 4) {
      const int i = i0 + r;
      vec x;
#pragma unroll
      for (int v = 0; v < V; v++) x[v] = (T)0;
      if (i < n) x = post[((size_t)b * n + i) * 64 + lane];
      *reinterpret_cast<vec *>(&tile[r * LD + lane * V]) = x;
    }
  }
  __syncthreads();
  {
    const int ii = tid & 31;
    const int i = i0 + ii;
    for (int f = tid >> 5; f < SF; f += 8) {
      const long frame = (long)b * SF + f;
      if (frame < nframes && i < n) {
        float val = (float)tile[ii * LD + f];
        if (ALG == ALG_MSQ) val *= (1.0f / 256.0f);
        if (ALG == ALG_TAB) val *= (1.0f / 16384.0f);
        out[frame * n + i] = val;
      }
    }
  }
}

// iters/status per frame from the slab state
__global__ void egress_state_kernel(int nframes, int V, int max_iter, const uint8_t *done,
                                    const int32_t *iters, int32_t *iters_out, int32_t *status_out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nframes) return;
  const int idx = f / V, v = f % V;
  const bool ok = (done[idx] >> v) & 1;
  if (iters_out) iters_out[f] = ok ? iters[f] : max_iter;
  if (status_out) status_out[f] = ok ? 0 : 1;
}

}  // namespace ldpc
