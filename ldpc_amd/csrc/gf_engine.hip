// gf_engine.hip -- non-binary engines of libldpc_hip (gfx950).
//
//   LDPC_ALG_GF2P_WHT   dec() of GFq_LDPC_NTT.c:437-544: GF(2^p), all-integer
//                       log2-domain messages, check node in the Walsh-Hadamard
//                       domain through two lookup tables.  Bit-exact.
//   LDPC_ALG_GFQ_DIRECT dec() of GFq_LDPC.c:356-467: GF(q) (prime q by mod-q
//                       arithmetic :22-24, 2^p by tables), binary64 natural-log
//                       messages, forward/backward (sigma/rho) check node.
//
// Data layout in HBM: every per-edge / per-variable pmf is stored
// [edge|variable][frame][Q] with the frame count padded to a multiple of
// G = 64 / Q.  One wavefront works on ONE node for G consecutive frames: lane
// l = g*Q + a holds symbol value a of frame g, so a wave moves G*Q contiguous
// values per access, node indices and GF coefficients are wave-uniform, and
// everything that crosses symbol values -- the coefficient permutation
// a -> a*h, the Walsh-Hadamard butterflies, the sigma/rho convolutions, the
// argmax and the normalising sums -- is a cross-lane operation inside the
// wave (ds_bpermute / shuffle-xor), with LDS holding the per-edge
// intermediates a lane needs again in the second sweep over the check.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <type_traits>

#include "engine.h"

namespace gf {

constexpr int W_DECI = 14;
constexpr int W_FMUL = 1 << W_DECI;
constexpr int W_LEVELS = 1 << 20;   // INT 6 + DECI 14 (GFq_LDPC_NTT.c:170-175)

struct GGraph {
  const int32_t *row_ptr, *col_idx, *hval, *hinv, *col_ptr, *edge_of;
  const uint8_t *mul;   // [Q*Q] GF multiplication table
  int n, m, E, Q, log2q, pow2, G, Fpad;
};

struct WTab {
  const int32_t *flog;    // [2^20]
  const uint32_t *fexp;   // centre of a [2^21] table: fexp[i], i in [-2^20, 2^20)
  const uint32_t *image;  // compact exact forms of both (tables.cpp), W_IMAGE_WORDS words: copied to LDS
};
// LDS image of the compact tables: xexp[16384] | flog[0..2047] | seg[512] of {c0 fraction, c1, c2, c0 integer}
constexpr int W_IMG_XEXP = 0, W_IMG_SMALL = kWhtFrac, W_IMG_SEG = kWhtFrac + kWhtSmall;
constexpr int W_IMAGE_WORDS = kWhtFrac + kWhtSmall + 4 * kWhtSeg;   // 80 KiB
typedef float w_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int gf_sub(const GGraph &g, int a, int b) {
  return g.pow2 ? (a ^ b) : (a - b + 2 * g.Q) % g.Q;
}
__device__ __forceinline__ int gf_add(const GGraph &g, int a, int b) {
  return g.pow2 ? (a ^ b) : (a + b) % g.Q;
}

// Flog(), GFq_LDPC_NTT.c:240-247
__device__ __forceinline__ int w_flog(const WTab &t, int x) {
  if (x <= 0) return -W_FMUL * 14;
  int i = 0;
  while (x >= W_LEVELS) { x >>= 1; i++; }
  return t.flog[x] + (i << W_DECI);
}
// Fexp(x) = fexp[x] (:236).  The reference does not check the index; an index
// outside the table is undefined behaviour there and is clamped here.
__device__ __forceinline__ int w_fexp(const WTab &t, int x) {
  x = x < -W_LEVELS ? -W_LEVELS : x;
  x = x > W_LEVELS - 1 ? W_LEVELS - 1 : x;
  return (int)t.fexp[x];
}
// The same two functions from the compact tables in LDS (img): bit-identical for every argument
// (ldpc_wht_compact_selfcheck replays this arithmetic over both full tables).
__device__ __forceinline__ int w_fexp_c(const uint32_t *img, int x) {
  x = x < -W_LEVELS ? -W_LEVELS : x;
  x = x > W_LEVELS - 1 ? W_LEVELS - 1 : x;
  if (x > 17 * kWhtFrac) return (int)(1u << 31);
  const int e = x >> W_DECI, fr = x & (kWhtFrac - 1);
  const int sh = 18 - e;                                   // >= 1
  const unsigned long long Y = (1ull << 32) | img[W_IMG_XEXP + fr];
  const unsigned v = (unsigned)((Y + (1ull << ((sh - 1) & 63))) >> (sh & 63));
  return sh >= 34 ? 0 : (int)v;
}
__device__ __forceinline__ int w_flog_c(const uint32_t *img, const WTab &t, int x) {
  if (x <= 0) return -W_FMUL * 14;
  int i = 0;
  while (x >= W_LEVELS) { x >>= 1; i++; }
  int v;
  if (x < kWhtSmall) {
    v = (int)img[W_IMG_SMALL + x];
  } else {
    const int e = 31 - __clz(x);
    const int r = x - (1 << e), sh = e - 9;
    const int h = r >> sh, lo = r & ((1 << sh) - 1);
    const float u = ldexpf((float)lo, -e) - 0x1p-10f;
    const w_f4 q = *(const w_f4 *)(img + W_IMG_SEG + 4 * h);
    const float w = fmaf(u, fmaf(u, q.z, q.y), q.x);
    const float fl = floorf(w + 0.5f);
    if (fabsf(w - fl) > 0.5f - 1e-4f) v = t.flog[x];       // too close to a rounding boundary (2e-4 of the arguments): the full table decides
    else v = __float_as_int(q.w) + W_FMUL * (e - 14) + (int)fl;
  }
  return v + (i << W_DECI);
}
__device__ __forceinline__ void w_load_image(uint32_t *img, const WTab &t) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  for (int i = threadIdx.x; i < W_IMAGE_WORDS / 4; i += blockDim.x) ((u4 *)img)[i] = ((const u4 *)t.image)[i];
  __syncthreads();
}

// ntt(), :118-135, one value per lane: stage b pairs lanes a and a ^ (1<<b)
__device__ __forceinline__ int wht_lanes(int y, int a, int log2q) {
  for (int b = 0; b < log2q; b++) {
    const int partner = __shfl_xor(y, 1 << b);
    y = (a & (1 << b)) ? partner - y : y + partner;
  }
  return y;
}

struct Lane {
  int l, g, a, base, active;
};
__device__ __forceinline__ Lane lane_of(const GGraph &g) {
  Lane L;
  L.l = threadIdx.x & 63;
  L.g = L.l / g.Q;
  L.a = L.l - L.g * g.Q;
  L.active = L.g < g.G;
  L.base = L.active ? L.g * g.Q : 0;
  if (!L.active) L.a = 0;
  return L;
}

// ------------------------------------------------------------------ ingest

// float2fix(), GFq_LDPC_NTT.c:183-190
__device__ __forceinline__ int w_float2fix(double x) {
  if (x >= 0) return (int)(x * W_FMUL + 0.5);
  return -(int)((-x) * W_FMUL + 0.5);
}

// in [F][n][Q] (frame-major, as the callers hold logfna per frame) -> fna [n][Fpad][Q]
template <typename InT, typename T, bool WHT>
__global__ void gf_ingest_kernel(const InT *__restrict__ in, int nframes, GGraph g, T *__restrict__ fna) {
  const size_t total = (size_t)g.n * g.Fpad * g.Q;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int a = (int)(idx % g.Q);
    const size_t r = idx / g.Q;
    const int f = (int)(r % g.Fpad);
    const int i = (int)(r / g.Fpad);
    T v = (T)0;
    if (f < nframes) {
      const InT x = in[((size_t)f * g.n + i) * g.Q + a];
      if constexpr (WHT && !std::is_integral<InT>::value) v = (T)w_float2fix((double)x);
      else v = (T)x;
    }
    fna[idx] = v;
  }
}

// s [F][len] u8 -> t [len][Fpad] u8 (and back)
__global__ void gf_sym_in_kernel(const uint8_t *__restrict__ s, int nframes, int len, int Fpad,
                                 uint8_t *__restrict__ t) {
  const size_t total = (size_t)len * Fpad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int f = (int)(idx % Fpad);
    const int j = (int)(idx / Fpad);
    t[idx] = f < nframes ? s[(size_t)f * len + j] : 0;
  }
}
__global__ void gf_sym_out_kernel(const uint8_t *__restrict__ t, int nframes, int len, int Fpad,
                                  uint8_t *__restrict__ s) {
  const size_t total = (size_t)len * nframes;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(idx % len);
    const int f = (int)(idx / len);
    s[idx] = t[(size_t)j * Fpad + f];
  }
}

// logqa = logfna on every edge (GFq_LDPC_NTT.c:442-446, GFq_LDPC.c:361-366)
template <typename T>
__global__ void gf_init_kernel(GGraph g, const T *__restrict__ fna, T *__restrict__ qa) {
  const size_t per = (size_t)g.Fpad * g.Q;
  const size_t total = (size_t)g.E * per;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(idx / per);
    qa[idx] = fna[(size_t)g.col_idx[e] * per + (idx - (size_t)e * per)];
  }
}

// The same, one workgroup per edge: the edge and its variable are wave-uniform (no 64-bit division, no
// gather of col_idx per element) and a variable's row goes out in 16-byte pieces.  For rows whose size is
// a multiple of 16 bytes.  528 -> see profiles/r02_gf4_wht_counters.txt
template <typename T>
__global__ __launch_bounds__(256) void gf_init_rows_kernel(GGraph g, const T *__restrict__ fna, T *__restrict__ qa) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const size_t per16 = (size_t)g.Fpad * g.Q * sizeof(T) / 16;
  for (int e = blockIdx.x; e < g.E; e += gridDim.x) {
    const u4 *src = reinterpret_cast<const u4 *>(fna) + (size_t)g.col_idx[e] * per16;
    u4 *dst = reinterpret_cast<u4 *>(qa) + (size_t)e * per16;
    for (size_t t = threadIdx.x; t < per16; t += 256) dst[t] = src[t];
  }
}

// ---------------------------------------------------- GF(2^p) WHT check node
//
// One wave per (check j, group of G frames).  GFq_LDPC_NTT.c:449-484.
__global__ __launch_bounds__(64) void gf2p_cn_kernel(GGraph g, WTab tab, int nfg, const int *__restrict__ qa,
                                                     int *__restrict__ ra, const uint8_t *__restrict__ synd) {
  extern __shared__ __attribute__((aligned(16))) int lds_i[];
  const Lane L = lane_of(g);
  const int j = blockIdx.x / nfg, fg = blockIdx.x - j * nfg;
  const int p0 = g.row_ptr[j], dc = g.row_ptr[j + 1] - p0;
  int *fq = lds_i;              // [dc][64] Flog of the scaled |WHT| value (looked up once, used twice)
  int *ng = lds_i + 64 * dc;    // [dc][64] its sign
  const int f = fg * g.G + L.g;
  const int z = (synd && L.active) ? synd[(size_t)j * g.Fpad + f] : 0;
  const bool live = L.l < g.G * g.Q;
  const int half = g.log2q / 2;
  int logprod = 0, sgnsum = 0;
  for (int k = 0; k < dc; k++) {
    const int e = p0 + k;
    const size_t off = ((size_t)e * g.Fpad + (size_t)fg * g.G) * g.Q + L.l;
    const int x = w_fexp(tab, live ? qa[off] : 0);
    // fQa[a*h] = Fexp(logqa[a]): position b receives from a = b * h^-1
    int y = __shfl(x, L.base + g.mul[L.a * g.Q + g.hinv[e]]);
    y = wht_lanes(y, L.a, g.log2q);
    int neg = 0;
    if (y < 0) { y = -y; neg = 1; }
    sgnsum ^= neg;
    y = (y + (1 << (half - 1))) >> half;
    const int lg = w_flog(tab, y);
    fq[k * 64 + L.l] = lg;
    ng[k * 64 + L.l] = neg;
    logprod += lg;
  }
  for (int k = 0; k < dc; k++) {
    const int e = p0 + k;
    const size_t off = ((size_t)e * g.Fpad + (size_t)fg * g.G) * g.Q + L.l;
    int r = w_fexp(tab, logprod - fq[k * 64 + L.l] + g.log2q * W_FMUL);
    if (ng[k * 64 + L.l] != sgnsum) r = -r;
    r = wht_lanes(r, L.a, g.log2q);
    // logra[a] = Flog(fRa[z - a*h])
    const int t = __shfl(r, L.base + (z ^ g.mul[L.a * g.Q + g.hval[e]]));
    if (live) ra[off] = w_flog(tab, t);
  }
}

// One wave per (variable i, group of G frames): GFq_LDPC_NTT.c:486-523 (sums,
// leave-one-out, argmax, and the per-edge normalisation fused in).
__global__ __launch_bounds__(256) void gf2p_vn_kernel(GGraph g, WTab tab, int nfg, const int *__restrict__ fna,
                                                      const int *__restrict__ ra, int *__restrict__ qa,
                                                      uint8_t *__restrict__ xhat,
                                                      const uint8_t *__restrict__ frozen) {
  const Lane L = lane_of(g);
  const long item = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (item >= (long)g.n * nfg) return;
  const int i = (int)(item / nfg), fg = (int)(item - (long)i * nfg);
  const int p0 = g.col_ptr[i], dv = g.col_ptr[i + 1] - p0;
  const bool live = L.l < g.G * g.Q;
  const size_t fo = (size_t)fg * g.G * g.Q + L.l;
  int logprod = live ? fna[(size_t)i * g.Fpad * g.Q + fo] : 0;
  for (int k = 0; k < dv; k++) {
    const int e = g.edge_of[p0 + k];
    logprod += live ? ra[(size_t)e * g.Fpad * g.Q + fo] : 0;
  }
  int maxp = -(1 << 30), arg = -1;
  for (int t = 0; t < g.Q; t++) {
    const int v = __shfl(logprod, L.base + t);
    if (v > maxp) { maxp = v; arg = t; }
  }
  const int f = fg * g.G + L.g;
  if (live && L.a == 0 && !frozen[f]) xhat[(size_t)i * g.Fpad + f] = (uint8_t)arg;
  for (int k = 0; k < dv; k++) {
    const int e = g.edge_of[p0 + k];
    const size_t off = (size_t)e * g.Fpad * g.Q + fo;
    int q = logprod - (live ? ra[off] : 0);
    int sum = w_fexp(tab, q);
    for (int b = 0; b < g.log2q; b++) sum += __shfl_xor(sum, 1 << b);
    if (sum <= 0) sum = g.Q;
    q -= w_flog(tab, sum);
    if (live) qa[off] = q;
  }
}

// The two kernels above with the compact tables in LDS (80 KiB per workgroup, loaded once) and
// persistent workgroups: every Fexp is one LDS read and a 64-bit shift, every Flog an LDS read
// and two fused multiply-adds, instead of a gather of 64 separate L2 lines per wavefront from
// 12 MiB of tables (what bounded the kernels above: profiles/r02_gf4_wht_counters.txt).
// fq packs Flog (<< 1) with the sign bit: [waves][dc][64] ints of LDS behind the image.
template <int PF>   // PF > 0: row weights <= PF, all of a check's inputs are loaded before any is used
__global__ __launch_bounds__(1024) void gf2p_cn_lds_kernel(GGraph g, WTab tab, int nfg, int dcmax,
                                                           const int *__restrict__ qa, int *__restrict__ ra,
                                                           const uint8_t *__restrict__ synd) {
  extern __shared__ __attribute__((aligned(16))) uint32_t w_lds[];
  w_load_image(w_lds, tab);
  const Lane L = lane_of(g);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int *fq = (int *)(w_lds + W_IMAGE_WORDS) + (size_t)wave * dcmax * 64;
  const bool live = L.l < g.G * g.Q;
  const int half = g.log2q / 2;
  const long items = (long)g.m * nfg;
  for (long item = (long)blockIdx.x * nw + wave; item < items; item += (long)gridDim.x * nw) {
    const int j = (int)(item / nfg), fg = (int)(item - (long)j * nfg);
    const int p0 = g.row_ptr[j], dc = g.row_ptr[j + 1] - p0;
    const int f = fg * g.G + L.g;
    const int z = (synd && L.active) ? synd[(size_t)j * g.Fpad + f] : 0;
    int logprod = 0, sgnsum = 0;
    int pre[PF > 0 ? PF : 1];
    if constexpr (PF > 0) {
#pragma unroll
      for (int k = 0; k < PF; k++)
        pre[k] = (k < dc && live) ? qa[((size_t)(p0 + k) * g.Fpad + (size_t)fg * g.G) * g.Q + L.l] : 0;
    }
#pragma unroll
    for (int k = 0; k < (PF > 0 ? PF : dc); k++) {
      if (PF > 0 && k >= dc) break;
      const int e = p0 + k;
      const size_t off = ((size_t)e * g.Fpad + (size_t)fg * g.G) * g.Q + L.l;
      const int x = w_fexp_c(w_lds, PF > 0 ? pre[k] : (live ? qa[off] : 0));
      int y = __shfl(x, L.base + g.mul[L.a * g.Q + g.hinv[e]]);
      y = wht_lanes(y, L.a, g.log2q);
      int neg = 0;
      if (y < 0) { y = -y; neg = 1; }
      sgnsum ^= neg;
      y = (y + (1 << (half - 1))) >> half;
      const int lg = w_flog_c(w_lds, tab, y);
      fq[k * 64 + L.l] = lg * 2 + neg;
      logprod += lg;
    }
    for (int k = 0; k < dc; k++) {
      const int e = p0 + k;
      const size_t off = ((size_t)e * g.Fpad + (size_t)fg * g.G) * g.Q + L.l;
      const int pk = fq[k * 64 + L.l];
      int r = w_fexp_c(w_lds, logprod - (pk >> 1) + g.log2q * W_FMUL);
      if ((pk & 1) != sgnsum) r = -r;
      r = wht_lanes(r, L.a, g.log2q);
      const int t = __shfl(r, L.base + (z ^ g.mul[L.a * g.Q + g.hval[e]]));
      if (live) ra[off] = w_flog_c(w_lds, tab, t);
    }
  }
}

template <int PF>   // PF > 0: column weights <= PF, a variable's messages are loaded once, up front
__global__ __launch_bounds__(1024) void gf2p_vn_lds_kernel(GGraph g, WTab tab, int nfg, const int *__restrict__ fna,
                                                           const int *__restrict__ ra, int *__restrict__ qa,
                                                           uint8_t *__restrict__ xhat,
                                                           const uint8_t *__restrict__ frozen) {
  extern __shared__ __attribute__((aligned(16))) uint32_t w_lds[];
  w_load_image(w_lds, tab);
  const Lane L = lane_of(g);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const bool live = L.l < g.G * g.Q;
  const long items = (long)g.n * nfg;
  for (long item = (long)blockIdx.x * nw + wave; item < items; item += (long)gridDim.x * nw) {
    const int i = (int)(item / nfg), fg = (int)(item - (long)i * nfg);
    const int p0 = g.col_ptr[i], dv = g.col_ptr[i + 1] - p0;
    const size_t fo = (size_t)fg * g.G * g.Q + L.l;
    int logprod = live ? fna[(size_t)i * g.Fpad * g.Q + fo] : 0;
    int pre[PF > 0 ? PF : 1], pe[PF > 0 ? PF : 1];
    if constexpr (PF > 0) {
#pragma unroll
      for (int k = 0; k < PF; k++) {
        pe[k] = k < dv ? g.edge_of[p0 + k] : 0;
        pre[k] = (k < dv && live) ? ra[(size_t)pe[k] * g.Fpad * g.Q + fo] : 0;
      }
#pragma unroll
      for (int k = 0; k < PF; k++) logprod += pre[k];
    } else {
      for (int k = 0; k < dv; k++) {
        const int e = g.edge_of[p0 + k];
        logprod += live ? ra[(size_t)e * g.Fpad * g.Q + fo] : 0;
      }
    }
    int maxp = -(1 << 30), arg = -1;
    for (int t = 0; t < g.Q; t++) {
      const int v = __shfl(logprod, L.base + t);
      if (v > maxp) { maxp = v; arg = t; }
    }
    const int f = fg * g.G + L.g;
    if (live && L.a == 0 && !frozen[f]) xhat[(size_t)i * g.Fpad + f] = (uint8_t)arg;
#pragma unroll
    for (int k = 0; k < (PF > 0 ? PF : dv); k++) {
      if (PF > 0 && k >= dv) break;
      const int e = PF > 0 ? pe[k] : g.edge_of[p0 + k];
      const size_t off = (size_t)e * g.Fpad * g.Q + fo;
      int q = logprod - (PF > 0 ? pre[k] : (live ? ra[off] : 0));
      int sum = w_fexp_c(w_lds, q);
      for (int b = 0; b < g.log2q; b++) sum += __shfl_xor(sum, 1 << b);
      if (sum <= 0) sum = g.Q;
      q -= w_flog_c(w_lds, tab, sum);
      if (live) qa[off] = q;
    }
  }
}

// ---- the LDS-table kernels again, specialised for GF(4) / GF(8) (LQ = log2 Q at compile time)
//
// Same layout, same arithmetic, same results as gf2p_*_lds_kernel; what changes is what the
// generic kernels spend their instructions on (about 280 per edge and lane, against some 60 of
// table arithmetic): the field order as a run-time value (butterfly, arg-max and lane arithmetic as
// loops and divisions), a 64-bit division per work item, two gathers into the multiplication table
// per edge, and the data-dependent branches of Flog.  Here the butterflies unroll to DPP moves, a
// lane keeps its row of the multiplication table in a register pair (a byte per coefficient), a
// work item advances by additions, rows are addressed as uniform base + lane offset, Flog / Fexp
// select instead of branching (the one branch left hands the 2e-4 of arguments next to a rounding
// boundary back to the full table), and a check's intermediate values stay in registers -- so a
// workgroup needs the 80 KiB image and nothing else, and two of them share a CU.
__device__ __forceinline__ int w_fexp_q(const uint32_t *img, int x) {
  x = x < -W_LEVELS ? -W_LEVELS : x;
  x = x > W_LEVELS - 1 ? W_LEVELS - 1 : x;
  const int e = x >> W_DECI, fr = x & (kWhtFrac - 1);
  const int sh = 18 - e;                                   // >= 1 unless x > 17 * kWhtFrac
  // (Y + 2^(sh-1)) >> sh with Y = 2^32 + mantissa, in 32-bit steps: t = Y >> (sh - 1) =
  // (2^31 + mantissa / 2) >> (sh - 2), result (t + 1) >> 1; sh < 2 only at or above 17 * kWhtFrac,
  // where the result is 2^31 either way
  const unsigned y1 = __builtin_amdgcn_alignbit(1u, img[W_IMG_XEXP + fr], 1);
  const unsigned t = y1 >> ((sh - 2) & 31);
  const unsigned v = (t >> 1) + (t & 1u);
  const int r = sh >= 34 ? 0 : (int)v;
  return x >= 17 * kWhtFrac ? (int)(1u << 31) : r;
}
__device__ __forceinline__ int w_flog_q(const uint32_t *img, const WTab &t, const int x) {
  const int xs = x < 1 ? 1 : x;
  const int e0 = 31 - __clz(xs);
  const int i = e0 > 19 ? e0 - 19 : 0;                     // while (x >= LEVELS) x >>= 1, :242
  const int xx = xs >> i;
  const bool small = xx < kWhtSmall;
  const int vs = (int)img[W_IMG_SMALL + (small ? xx : kWhtSmall - 1)];
  const int xc = small ? kWhtSmall : xx;                   // (the segment form serves 2048 and up)
  const int e = 31 - __clz(xc);
  const int r = xc - (1 << e), sh = e - 9;
  const int h = r >> sh, lo = r & ((1 << sh) - 1);
  const float u = ldexpf((float)lo, -e) - 0x1p-10f;
  const w_f4 q = *(const w_f4 *)(img + W_IMG_SEG + 4 * h);
  const float w = fmaf(u, fmaf(u, q.z, q.y), q.x);
  const float fl = floorf(w + 0.5f);
  int v = __float_as_int(q.w) + W_FMUL * (e - 14) + (int)fl;
  if (!small && fabsf(w - fl) > 0.5f - 1e-4f) v = t.flog[xx];   // too close to a rounding boundary: the full table decides
  v = small ? vs : v;
  v += i << W_DECI;
  return x <= 0 ? -W_FMUL * 14 : v;
}
// the value of lane l ^ (1 << B), B = 0..2, by DPP moves (no trip through the LDS crossbar and no
// wait for it, which is what __shfl_xor compiles to): quad permutes for 1 and 2, for 4 the two row
// shifts by four lanes, of which a lane takes the one that stays inside its group of eight
template <int B>
__device__ __forceinline__ int xor_lane(const int y, const int a) {
  if (B == 0) return __builtin_amdgcn_update_dpp(0, y, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
  if (B == 1) return __builtin_amdgcn_update_dpp(0, y, 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true);
  const int up = __builtin_amdgcn_update_dpp(0, y, 0x104 /* row_shl:4: from lane l + 4 */, 0xf, 0xf, true);
  const int dn = __builtin_amdgcn_update_dpp(0, y, 0x114 /* row_shr:4: from lane l - 4 */, 0xf, 0xf, true);
  return (a & 4) ? dn : up;
}
template <int LQ>
__device__ __forceinline__ int wht_q(int y, const int a) {
  static_assert(LQ >= 1 && LQ <= 3, "butterflies by DPP up to GF(8)");
  { const int p = xor_lane<0>(y, a); y = (a & 1) ? p - y : y + p; }
  if (LQ >= 2) { const int p = xor_lane<1>(y, a); y = (a & 2) ? p - y : y + p; }
  if (LQ >= 3) { const int p = xor_lane<2>(y, a); y = (a & 4) ? p - y : y + p; }
  return y;
}
template <int LQ>
__device__ __forceinline__ int sum_q(int y, const int a) {
  y += xor_lane<0>(y, a);
  if (LQ >= 2) y += xor_lane<1>(y, a);
  if (LQ >= 3) y += xor_lane<2>(y, a);
  return y;
}
// work items (node, frame group) of a persistent workgroup: wavefront w of block b starts at
// b * nw + w and advances by gridDim * nw, as (node, group) pairs without a division per item
struct ItemWalk {
  unsigned node, fg, dnode, dfg, nfg;
  __device__ __forceinline__ ItemWalk(unsigned first, unsigned stride, unsigned nfg_) : nfg(nfg_) {
    node = first / nfg; fg = first - node * nfg;
    dnode = stride / nfg; dfg = stride - dnode * nfg;
  }
  __device__ __forceinline__ void next() {
    node += dnode; fg += dfg;
    if (fg >= nfg) { fg -= nfg; node++; }
  }
};

template <int LQ, int PF>   // row weights <= PF
__global__ __launch_bounds__(1024) void gf2p_cn_q_kernel(GGraph g, WTab tab, int nfg, int dcmax,
                                                         const int *__restrict__ qa, int *__restrict__ ra,
                                                         const uint8_t *__restrict__ synd) {
  constexpr int Q = 1 << LQ, G = 64 / Q, half = LQ / 2;
  extern __shared__ __attribute__((aligned(16))) uint32_t w_lds[];
  w_load_image(w_lds, tab);
  const int l = threadIdx.x & 63, a = l & (Q - 1), base = l & ~(Q - 1), gi = l >> LQ;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  unsigned long long mrow = 0;                             // a * h for every coefficient h, a byte each
#pragma unroll
  for (int h = 0; h < Q; h++) mrow |= (unsigned long long)g.mul[a * Q + h] << (8 * h);
  const size_t row = (size_t)g.Fpad * Q;                   // values per edge
  for (ItemWalk it(blockIdx.x * nw + wave, gridDim.x * nw, (unsigned)nfg); it.node < (unsigned)g.m; it.next()) {
    const int j = (int)it.node;
    const unsigned lo = it.fg * 64u + (unsigned)l;         // (fg * G + gi) * Q + a
    const int p0 = g.row_ptr[j], dc = g.row_ptr[j + 1] - p0;
    const int z = synd ? synd[(size_t)j * g.Fpad + it.fg * G + gi] : 0;
    int pre[PF], pk[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) pre[k] = k < dc ? (qa + (size_t)(p0 + k) * row)[lo] : 0;
    int logprod = 0, sgnsum = 0;
#pragma unroll
    for (int k = 0; k < PF; k++) {
      if (k >= dc) break;
      const int x = w_fexp_q(w_lds, pre[k]);
      const int hi = g.hinv[p0 + k];
      int y = __shfl(x, base + (int)((mrow >> (8 * hi)) & 0xffu));
      y = wht_q<LQ>(y, a);
      const int neg = y < 0;
      y = neg ? -y : y;
      sgnsum ^= neg;
      if (half > 0) y = (y + (1 << (half > 0 ? half - 1 : 0))) >> half;
      const int lg = w_flog_q(w_lds, tab, y);
      pk[k] = lg * 2 + neg;
      logprod += lg;
    }
#pragma unroll
    for (int k = 0; k < PF; k++) {
      if (k >= dc) break;
      int r = w_fexp_q(w_lds, logprod - (pk[k] >> 1) + LQ * W_FMUL);
      if ((pk[k] & 1) != sgnsum) r = -r;
      r = wht_q<LQ>(r, a);
      const int hv = g.hval[p0 + k];
      const int t = __shfl(r, base + (z ^ (int)((mrow >> (8 * hv)) & 0xffu)));
      (ra + (size_t)(p0 + k) * row)[lo] = w_flog_q(w_lds, tab, t);
    }
  }
}

template <int LQ, int PF>   // column weights <= PF
__global__ __launch_bounds__(1024) void gf2p_vn_q_kernel(GGraph g, WTab tab, int nfg, const int *__restrict__ fna,
                                                         const int *__restrict__ ra, int *__restrict__ qa,
                                                         uint8_t *__restrict__ xhat,
                                                         const uint8_t *__restrict__ frozen) {
  constexpr int Q = 1 << LQ, G = 64 / Q;
  extern __shared__ __attribute__((aligned(16))) uint32_t w_lds[];
  w_load_image(w_lds, tab);
  const int l = threadIdx.x & 63, a = l & (Q - 1), base = l & ~(Q - 1), gi = l >> LQ;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  const size_t row = (size_t)g.Fpad * Q;
  for (ItemWalk it(blockIdx.x * nw + wave, gridDim.x * nw, (unsigned)nfg); it.node < (unsigned)g.n; it.next()) {
    const int i = (int)it.node;
    const unsigned lo = it.fg * 64u + (unsigned)l;
    const int p0 = g.col_ptr[i], dv = g.col_ptr[i + 1] - p0;
    int logprod = (fna + (size_t)i * row)[lo];
    int pre[PF], pe[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
      pe[k] = k < dv ? g.edge_of[p0 + k] : 0;
      pre[k] = k < dv ? (ra + (size_t)pe[k] * row)[lo] : 0;
    }
#pragma unroll
    for (int k = 0; k < PF; k++) logprod += pre[k];
    int maxp = -(1 << 30), arg = -1;                       // first maximum wins (strict >), :496-503
#pragma unroll
    for (int t = 0; t < Q; t++) {
      const int v = __shfl(logprod, base + t);
      if (v > maxp) { maxp = v; arg = t; }
    }
    const unsigned f = it.fg * G + (unsigned)gi;
    if (a == 0 && !frozen[f]) xhat[(size_t)i * g.Fpad + f] = (uint8_t)arg;
#pragma unroll
    for (int k = 0; k < PF; k++) {
      if (k >= dv) break;
      int q = logprod - pre[k];
      int sum = sum_q<LQ>(w_fexp_q(w_lds, q), a);
      if (sum <= 0) sum = Q;
      q -= w_flog_q(w_lds, tab, sum);
      (qa + (size_t)pe[k] * row)[lo] = q;
    }
  }
}

// ------------------------------------------------ GF(q) direct check node
//
// One wave per (check j, G frames).  GFq_LDPC.c:369-424: rho backward (kept in
// LDS, lane-private), then sigma forward and the extrinsic r in one sweep.
__global__ __launch_bounds__(64) void gfq_cn_kernel(GGraph g, int nfg, const double *__restrict__ qa,
                                                    double *__restrict__ ra, const uint8_t *__restrict__ synd) {
  extern __shared__ __attribute__((aligned(16))) double lds_d[];
  const Lane L = lane_of(g);
  const int j = blockIdx.x / nfg, fg = blockIdx.x - j * nfg;
  const int p0 = g.row_ptr[j], dc = g.row_ptr[j + 1] - p0;
  const int Q = g.Q;
  const int f = fg * g.G + L.g;
  const int z = (synd && L.active) ? synd[(size_t)j * g.Fpad + f] : 0;
  const bool live = L.l < g.G * Q;
  double *rho = lds_d;   // [dc][64]
  double nxt = 0.0;
  for (int k = dc - 1; k >= 1; k--) {
    const int e = p0 + k;
    const int h = g.hval[e];
    const double q = live ? qa[((size_t)e * g.Fpad + (size_t)fg * g.G) * Q + L.l] : 0.0;
    double r;
    if (k == dc - 1) {
      r = __shfl(q, L.base + g.mul[L.a * Q + g.hinv[e]]);   // rho[a*h] = qa[a]
    } else {
      double sum = 0;
      for (int t = 0; t < Q; t++) {
        const int s = gf_sub(g, L.a, g.mul[t * Q + h]);
        sum += exp(__shfl(nxt, L.base + s) + __shfl(q, L.base + t));
      }
      r = log(sum);
    }
    rho[k * 64 + L.l] = r;
    nxt = r;
  }
  double sig = 0.0;
  for (int k = 0; k < dc; k++) {
    const int e = p0 + k;
    const int h = g.hval[e];
    const size_t off = ((size_t)e * g.Fpad + (size_t)fg * g.G) * Q + L.l;
    const double q = live ? qa[off] : 0.0;
    const int zah = gf_sub(g, z, g.mul[L.a * Q + h]);
    double out;
    if (k == 0) {
      out = 0.0;
      if (dc > 1) out = __shfl(rho[64 + L.l], L.base + zah);
    } else if (k == dc - 1) {
      out = __shfl(sig, L.base + zah);
    } else {
      const double rn = rho[(k + 1) * 64 + L.l];
      double sum = 0;
      for (int t = 0; t < Q; t++) {
        const int s = gf_sub(g, zah, t);
        sum += exp(__shfl(sig, L.base + s) + __shfl(rn, L.base + t));
      }
      out = log(sum);
    }
    if (live) ra[off] = out;
    double ns;
    if (k == 0) {
      ns = __shfl(q, L.base + g.mul[L.a * Q + g.hinv[e]]);   // sigma[a*h] = qa[a]
    } else {
      double sum = 0;
      for (int t = 0; t < Q; t++) {
        const int s = gf_sub(g, L.a, g.mul[t * Q + h]);
        sum += exp(__shfl(sig, L.base + s) + __shfl(q, L.base + t));
      }
      ns = log(sum);
    }
    sig = ns;
  }
}

// The same check node with the sigma / rho recursions carried as probabilities.
// Every logqa row is a normalised pmf (GFq_LDPC.c:446-457), so each sigma_k and
// rho_k is the pmf of a partial check sum: the reference's
//   log(sum_t exp(logsigma[s] + logqa[t]))           (:377-382, :395-400, :415-420)
// is log(sum_t sigma[s] * qa[t]) with sigma = exp(logsigma), qa = exp(logqa).
// Keeping sigma and rho linear between the steps needs one exp per (edge, symbol)
// and one log per extrinsic value instead of q exps for each of them; the terms
// are added in the reference's order (t ascending).  Values agree with the
// literal kernel above to rounding (a few ulp of binary64), not bit for bit.
//
// The pmfs a lane convolves live in LDS rows of 64 doubles (lane l's own value at
// [l]): rho_k for every k, the running sigma and exp(logqa) of the current edge,
// so a term of a convolution is two ds_read_b64 (one of them a broadcast within
// the frame's lane group) and one fma; s = a - t*h walks down by h per term in a
// prime field and is a ^ (t*h) in GF(2^p).  QT = field order at compile time
// (0: any), which unrolls the t loops.
//
// SPILL: a check of weight dc holds dc rho rows, 512 B each, for the whole forward
// sweep; at dc = 63 that is 32 KB per wavefront and five wavefronts per CU.  With
// SPILL the backward sweep parks rho_k in the check's own output rows ra[j][k]
// instead (same shape, and ra[j][k] only receives its extrinsic value at forward
// step k, after rho_k was last needed at step k-1), LDS keeps three rows, and the
// CU fills up.  Same values either way.
template <int QT, bool SPILL>
__global__ __launch_bounds__(64) void gfq_cn_lin_kernel(GGraph g, int nfg, const double *__restrict__ qa,
                                                        double *__restrict__ ra,
                                                        const uint8_t *__restrict__ synd) {
  extern __shared__ __attribute__((aligned(16))) double lds_d[];
  const int Q = QT ? QT : g.Q;
  const int l = threadIdx.x & 63;
  const int grp = l / Q;
  const bool live = grp < g.G;
  const int a = live ? l - grp * Q : 0;
  const int base = live ? grp * Q : 0;
  const int j = blockIdx.x / nfg, fg = blockIdx.x - j * nfg;
  const int p0 = g.row_ptr[j], dc = g.row_ptr[j + 1] - p0;
  const int f = fg * g.G + grp;
  const int z = (synd && live) ? synd[(size_t)j * g.Fpad + f] : 0;
  const bool pow2 = g.pow2 != 0;
  double *rho = lds_d;                        // [dc][64], or one row (rho_{k+1}) with SPILL
  double *S = rho + (SPILL ? 64 : dc * 64);   // sigma_{k-1}
  double *P = S + 64;                         // exp(logqa) of the current edge
  const size_t fo = (size_t)fg * g.G * Q + l;
  // sum_t A[start - t*h] * B[t], t ascending
  auto conv = [&](const double *A, const double *B, int start, int h) -> double {
    double sum = 0;
    if (pow2) {
#pragma unroll
      for (int t = 0; t < Q; t++) sum += A[base + (start ^ g.mul[t * Q + h])] * B[base + t];
    } else {
      int s = start;
#pragma unroll
      for (int t = 0; t < Q; t++) {
        sum += A[base + s] * B[base + t];
        s -= h;
        s += s < 0 ? Q : 0;
      }
    }
    return sum;
  };
  for (int k = dc - 1; k >= 1; k--) {
    const int e = p0 + k;
    const int h = g.hval[e];
    const size_t off = (size_t)e * g.Fpad * Q + fo;
    P[l] = live ? exp(qa[off]) : 0.0;
    __syncthreads();
    const double r = k == dc - 1 ? P[base + g.mul[a * Q + g.hinv[e]]]      // rho[a*h] = qa[a]
                                 : conv(SPILL ? rho : rho + (k + 1) * 64, P, a, h);
    if (SPILL) {
      __syncthreads();
      rho[l] = r;
      if (live) ra[off] = r;
    } else {
      rho[k * 64 + l] = r;
    }
    __syncthreads();
  }
  for (int k = 0; k < dc; k++) {
    const int e = p0 + k;
    const int h = g.hval[e];
    const size_t off = (size_t)e * g.Fpad * Q + fo;
    P[l] = live ? exp(qa[off]) : 0.0;
    if (SPILL && k + 1 < dc) rho[l] = live ? ra[off + (size_t)g.Fpad * Q] : 0.0;   // rho_{k+1}, parked in ra[j][k+1]
    __syncthreads();
    const double *rn = SPILL ? rho : rho + (k + 1) * 64;
    const int zah = gf_sub(g, z, g.mul[a * Q + h]);
    double out;
    if (k == 0) out = dc > 1 ? log(rn[base + zah]) : 0.0;
    else if (k == dc - 1) out = log(S[base + zah]);
    else out = log(conv(S, rn, zah, 1));
    if (live) ra[off] = out;
    const double ns = k == 0 ? P[base + g.mul[a * Q + g.hinv[e]]]         // sigma[a*h] = qa[a]
                             : conv(S, P, a, h);
    __syncthreads();
    S[l] = ns;
  }
}

// GFq_LDPC.c:426-457
__global__ __launch_bounds__(256) void gfq_vn_kernel(GGraph g, int nfg, const double *__restrict__ fna,
                                                     const double *__restrict__ ra, double *__restrict__ qa,
                                                     uint8_t *__restrict__ xhat,
                                                     const uint8_t *__restrict__ frozen) {
  const Lane L = lane_of(g);
  const long item = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (item >= (long)g.n * nfg) return;
  const int i = (int)(item / nfg), fg = (int)(item - (long)i * nfg);
  const int p0 = g.col_ptr[i], dv = g.col_ptr[i + 1] - p0;
  const int Q = g.Q;
  const bool live = L.l < g.G * Q;
  const size_t fo = (size_t)fg * g.G * Q + L.l;
  double logprod = live ? fna[(size_t)i * g.Fpad * Q + fo] : 0.0;
  for (int k = 0; k < dv; k++) {
    const int e = g.edge_of[p0 + k];
    logprod += live ? ra[(size_t)e * g.Fpad * Q + fo] : 0.0;
  }
  double maxp = -9e99;
  int arg = -1;
  for (int t = 0; t < Q; t++) {
    const double v = __shfl(logprod, L.base + t);
    if (v > maxp) { maxp = v; arg = t; }
  }
  const int f = fg * g.G + L.g;
  if (live && L.a == 0 && !frozen[f]) xhat[(size_t)i * g.Fpad + f] = (uint8_t)arg;
  for (int k = 0; k < dv; k++) {
    const int e = g.edge_of[p0 + k];
    const size_t off = (size_t)e * g.Fpad * Q + fo;
    const double q = logprod - (live ? ra[off] : 0.0);
    const double ex = exp(q);
    double sum = 0;
    for (int t = 0; t < Q; t++) sum += __shfl(ex, L.base + t);
    if (live) qa[off] = q - log(sum);
  }
}

// ------------------------------------------------- syndrome + frame state

// enc() (GFq_LDPC.c:299-310) of the current estimate and the distance to the
// target syndrome: bit count for GF(2^p) (HamDist, GFq_LDPC_NTT.c:253-262),
// symbol count otherwise.  One thread per (check, frame); frames across lanes.
// A workgroup: 64 frames (one per lane) x `per` consecutive checks, dealt round-robin to its 8 wavefronts;
// the wavefronts' partial distances meet in LDS and make ONE atomic per frame and workgroup (one per
// unsatisfied check and frame queued thousands deep on a frame's word when nothing converges).  Round 2's
// first version gave a thread 32 checks in a row: ~190 dependent loads one after the other, 139 us per
// launch on the GF(4) bench where the bytes would take 15 (profiles/r02_gf4_wht_counters.txt).
#define GF_SYND_WAVES 8
__global__ __launch_bounds__(64 * GF_SYND_WAVES) void gf_synd_kernel(GGraph g, const uint8_t *__restrict__ xhat,
                                                      const uint8_t *__restrict__ synd,
                                                      uint8_t *__restrict__ zhat, int *__restrict__ dist,
                                                      const uint8_t *__restrict__ frozen, int per) {
  __shared__ int part[GF_SYND_WAVES][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + lane;
  const bool live = f < g.Fpad && !frozen[f];
  const int j0 = blockIdx.y * per, j1 = min(j0 + per, g.m);
  int dsum = 0;
  if (live)
    for (int j = j0 + grp; j < j1; j += GF_SYND_WAVES) {
      const size_t idx = (size_t)j * g.Fpad + f;
      int sum = 0;
      for (int e = g.row_ptr[j]; e < g.row_ptr[j + 1]; e++)
        sum = gf_add(g, sum, g.mul[xhat[(size_t)g.col_idx[e] * g.Fpad + f] * g.Q + g.hval[e]]);
      zhat[idx] = (uint8_t)sum;
      const int z = synd ? synd[idx] : 0;
      dsum += g.pow2 ? __popc((unsigned)(z ^ sum)) : (z != sum);
    }
  part[grp][lane] = dsum;
  __syncthreads();
  if (grp == 0 && live) {
    int tot = 0;
#pragma unroll
    for (int k = 0; k < GF_SYND_WAVES; k++) tot += part[k][lane];
    if (tot) atomicAdd(&dist[f], tot);
  }
}

// The same with FOUR frames per lane (one 32-bit access per row instead of four byte accesses, which the
// texture-address unit takes at a quarter of the dword rate) and the multiplication table in LDS (the global
// gather per edge and frame depended on the byte just loaded): fields up to GF(16), frame count a multiple of 4.
__global__ __launch_bounds__(64 * GF_SYND_WAVES) void gf_synd4_kernel(GGraph g, const uint8_t *__restrict__ xhat,
                                                       const uint8_t *__restrict__ synd,
                                                       uint8_t *__restrict__ zhat, int *__restrict__ dist,
                                                       const uint8_t *__restrict__ frozen, int per) {
  __shared__ int part[GF_SYND_WAVES][4][64];
  __shared__ uint8_t mulL[256];
  if ((int)threadIdx.x < g.Q * g.Q) mulL[threadIdx.x] = g.mul[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int f0 = (blockIdx.x * 64 + lane) * 4;
  unsigned livem = 0;                       // bit k: frame f0 + k is still being decoded
  if (f0 < g.Fpad) {
    const unsigned fr = *reinterpret_cast<const uint32_t *>(frozen + f0);
#pragma unroll
    for (int k = 0; k < 4; k++) livem |= (((fr >> (8 * k)) & 0xffu) ? 0u : 1u) << k;
  }
  const int j0 = blockIdx.y * per, j1 = min(j0 + per, g.m);
  int ds[4] = {0, 0, 0, 0};
  if (livem)
    for (int j = j0 + grp; j < j1; j += GF_SYND_WAVES) {
      const size_t idx = (size_t)j * g.Fpad + f0;
      int sum[4] = {0, 0, 0, 0};
      for (int e = g.row_ptr[j]; e < g.row_ptr[j + 1]; e++) {
        const unsigned x = *reinterpret_cast<const uint32_t *>(xhat + (size_t)g.col_idx[e] * g.Fpad + f0);
        const int h = g.hval[e];
#pragma unroll
        for (int k = 0; k < 4; k++) sum[k] = gf_add(g, sum[k], mulL[((x >> (8 * k)) & 0xffu) * g.Q + h]);
      }
      const unsigned z = synd ? *reinterpret_cast<const uint32_t *>(synd + idx) : 0u;
      unsigned out = livem == 15u ? 0u : *reinterpret_cast<const uint32_t *>(zhat + idx);   // frozen frames keep theirs
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if ((livem >> k) & 1u) {
          const int zk = (int)((z >> (8 * k)) & 0xffu);
          out = (out & ~(0xffu << (8 * k))) | ((unsigned)sum[k] << (8 * k));
          ds[k] += g.pow2 ? __popc((unsigned)(zk ^ sum[k])) : (zk != sum[k]);
        }
      }
      *reinterpret_cast<uint32_t *>(zhat + idx) = out;
    }
#pragma unroll
  for (int k = 0; k < 4; k++) part[grp][k][lane] = ds[k];
  __syncthreads();
  if (grp < 4 && ((livem >> grp) & 1u)) {   // wavefront k sums frame f0 + k of every lane
    int tot = 0;
#pragma unroll
    for (int w = 0; w < GF_SYND_WAVES; w++) tot += part[w][grp][lane];
    if (tot) atomicAdd(&dist[f0 + grp], tot);
  }
}

struct FrameState {
  int32_t *dist, *iters, *status, *iir, *prev, *nodecr;
  uint8_t *frozen;
};

__global__ void gf_state_init_kernel(FrameState s, int Fpad, int nframes) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= Fpad) return;
  s.dist[f] = 0;
  s.iters[f] = 0;
  s.status[f] = 1;
  s.iir[f] = 0;
  s.prev[f] = 999999;
  s.nodecr[f] = 0;
  s.frozen[f] = f >= nframes;
}

// After iteration `loop` (0-based): exit test (GFq_LDPC.c:459-463) and, for the
// WHT decoder, its non-convergence abort (GFq_LDPC_NTT.c:527-540).
__global__ void gf_state_kernel(FrameState s, int Fpad, int loop, int wht, int32_t *remaining) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  int rem = 0;
  if (f < Fpad && !s.frozen[f]) {
    const int d = s.dist[f];
    s.dist[f] = 0;
    s.iters[f] = loop + 1;
    bool stop = false;
    if (d == 0) {
      s.status[f] = 0;
      stop = true;
    } else if (wht) {
      int iir = s.iir[f];
      if (loop == 0) iir = d;
      else iir = (int)__dadd_rn(__dadd_rn(__dmul_rn((double)iir, 0.85), __dmul_rn((double)d, 0.15)), 0.5);
      int nd = s.nodecr[f];
      if (s.prev[f] <= d) nd++;
      else nd = 0;
      if ((double)d > __dmul_rn((double)iir, 1.1) || nd > 10) stop = true;
      s.iir[f] = iir;
      s.nodecr[f] = nd;
      s.prev[f] = d;
    }
    if (stop) s.frozen[f] = 1;
    else rem = 1;
  }
  rem = __syncthreads_count(rem);
  if (threadIdx.x == 0 && rem) atomicAdd(remaining, rem);
}

__global__ void gf_state_out_kernel(FrameState s, int nframes, int32_t *iters_out, int32_t *status_out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nframes) return;
  if (iters_out) iters_out[f] = s.iters[f];
  if (status_out) status_out[f] = s.status[f];
}

// ---- second stage of a two-stage decode (GfEngine::decode; the binary two-pass engine's BinaryEngine::decode has
// the same shape): the frames still open when the first stage stops are listed, their inputs gathered into a
// dense batch, decoded again from scratch and their results written back over the first stage's
__global__ void gf_list_open_kernel(int nframes, const uint8_t *frozen, int *count, int *idx) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < nframes && !frozen[f]) idx[atomicAdd(count, 1)] = f;
}
// rows of `row` 4-byte words (or bytes): dst[i] = src[idx[i]] (gather) or dst[idx[i]] = src[i] (scatter)
template <typename U, bool SCATTER>
__global__ void gf_move_rows_kernel(const U *__restrict__ src, U *__restrict__ dst, const int *__restrict__ idx, size_t row) {
  const size_t i = blockIdx.x, f = (size_t)idx[blockIdx.x];
  const U *sp = src + (SCATTER ? i : f) * row;
  U *dp = dst + (SCATTER ? f : i) * row;
  for (size_t k = threadIdx.x; k < row; k += blockDim.x) dp[k] = sp[k];
}
template <bool SCATTER>
static void gf_move_rows(const void *src, void *dst, const int *idx, int count, size_t row_bytes, hipStream_t st) {
  if (count <= 0 || !src || !dst) return;
  if (row_bytes % 4 == 0 && ((size_t)src | (size_t)dst) % 4 == 0)
    hipLaunchKernelGGL((gf_move_rows_kernel<uint32_t, SCATTER>), dim3((unsigned)count), dim3(256), 0, st, (const uint32_t *)src,
                       (uint32_t *)dst, idx, row_bytes / 4);
  else
    hipLaunchKernelGGL((gf_move_rows_kernel<uint8_t, SCATTER>), dim3((unsigned)count), dim3(256), 0, st, (const uint8_t *)src,
                       (uint8_t *)dst, idx, row_bytes);
}

// ------------------------------------------------------------------ engine

template <typename T, bool WHT>
struct GfEngine : Engine {
  DevBuf s2_idx, s2_llr, s2_synd, s2_bits, s2_iters, s2_sout, s2_status;
  ldpc_opts opts{};
  int n = 0, m = 0, E = 0, Q = 0, rmax = 0;
  int G = 1, Fpad_max = 0;
  GGraph g{};
  WTab tab{};
  DevBuf d_row_ptr, d_col_idx, d_hval, d_hinv, d_col_ptr, d_edge_of, d_mul, d_flog, d_fexp, d_wimg;
  bool lds_tables = false;   // GF2P_WHT: compact tables in LDS (gf2p_*_lds_kernel)
  void (*cn_lds_kern)(GGraph, WTab, int, int, const int *, int *, const uint8_t *) = nullptr;
  void (*vn_lds_kern)(GGraph, WTab, int, const int *, const int *, int *, uint8_t *, const uint8_t *) = nullptr;
  // GF(4) / GF(8) with row and column weights <= 8: the specialised kernels (gf2p_*_q_kernel), two workgroups per CU
  void (*cn_q_kern)(GGraph, WTab, int, int, const int *, int *, const uint8_t *) = nullptr;
  void (*vn_q_kern)(GGraph, WTab, int, const int *, const int *, int *, uint8_t *, const uint8_t *) = nullptr;
  int cn_waves = 0, num_cu = 256;
  DevBuf fna, qa, ra, xhat, zhat, synd, st_i32, st_frozen, remaining;
  int32_t *h_remaining = nullptr;
  int remaining_cap = 0;
  hipEvent_t ev[4] = {};
  FrameState fs{};

  ~GfEngine() override {
    if (h_remaining) (void)hipHostFree(h_remaining);
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e);
  }
  size_t llr_elems_per_frame() const override { return (size_t)n * Q; }

  template <typename U>
  int upload(DevBuf &b, const std::vector<U> &v) {
    int rc = b.ensure(sizeof(U) * (v.size() ? v.size() : 1));
    if (rc) return rc;
    HIPCHK(hipMemcpy(b.p, v.data(), sizeof(U) * v.size(), hipMemcpyHostToDevice));
    return LDPC_OK;
  }
  int ensure_remaining(int slots) {
    if (slots <= remaining_cap) return LDPC_OK;
    if (h_remaining) (void)hipHostFree(h_remaining);
    h_remaining = nullptr;
    const int cap = slots < 256 ? 256 : slots;
    HIPCHK(hipHostMalloc((void **)&h_remaining, sizeof(int32_t) * cap, hipHostMallocDefault));
    int rc = remaining.ensure(sizeof(int32_t) * cap);
    if (rc) return rc;
    remaining_cap = cap;
    return LDPC_OK;
  }

  int init(const ldpc_code &c, const ldpc_opts &o) override {
    opts = o;
    n = c.n; m = c.m; E = c.E(); Q = c.q; rmax = c.rmax;
    {
      hipDeviceProp_t prop;
      HIPCHK(hipGetDeviceProperties(&prop, o.device));
      num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const bool pow2 = (Q & (Q - 1)) == 0;
    int log2q = 0;
    while ((1 << log2q) < Q) log2q++;
    if (Q > 64 || Q < 3 || (WHT && (!pow2 || log2q < 2))) {
      ldpc_set_error(WHT ? "GF2P_WHT needs a GF(2^p) code with 4 <= q <= 64"
                         : "GFQ_DIRECT needs 3 <= q <= 64");
      return LDPC_ERR_UNSUPPORTED;
    }
    if (!pow2)
      for (int d = 2; d * d <= Q; d++)
        if (Q % d == 0) {
          ldpc_set_error("q is neither prime nor a power of two");
          return LDPC_ERR_UNSUPPORTED;
        }
    const size_t lds = (size_t)(rmax + 2) * 64 * 8;
    if (lds > 160 * 1024) {
      ldpc_set_error("row weight too large for the LDS-staged GF check node");
      return LDPC_ERR_UNSUPPORTED;
    }
    G = 64 / Q;
    const int mf = o.max_frames > 0 ? o.max_frames : 1024;
    Fpad_max = (mf + G - 1) / G * G;

    std::vector<uint8_t> mul((size_t)Q * Q);
    std::vector<int32_t> inv(Q, 0);
    for (int a = 0; a < Q; a++)
      for (int b = 0; b < Q; b++) {
        const int p = ldpc_gf_mul_host(Q, a, b);
        mul[(size_t)a * Q + b] = (uint8_t)p;
        if (p == 1) inv[a] = b;
      }
    std::vector<int32_t> hinv(E);
    for (int e = 0; e < E; e++) hinv[e] = inv[c.hval[e]];
    int rc;
    if ((rc = upload(d_row_ptr, c.row_ptr))) return rc;
    if ((rc = upload(d_col_idx, c.col_idx))) return rc;
    if ((rc = upload(d_hval, c.hval))) return rc;
    if ((rc = upload(d_hinv, hinv))) return rc;
    if ((rc = upload(d_col_ptr, c.col_ptr))) return rc;
    if ((rc = upload(d_edge_of, c.edge_of))) return rc;
    if ((rc = upload(d_mul, mul))) return rc;
    g.row_ptr = d_row_ptr.as<int32_t>();
    g.col_idx = d_col_idx.as<int32_t>();
    g.hval = d_hval.as<int32_t>();
    g.hinv = d_hinv.as<int32_t>();
    g.col_ptr = d_col_ptr.as<int32_t>();
    g.edge_of = d_edge_of.as<int32_t>();
    g.mul = d_mul.as<uint8_t>();
    g.n = n; g.m = m; g.E = E; g.Q = Q; g.log2q = log2q; g.pow2 = pow2; g.G = G; g.Fpad = Fpad_max;
    if (WHT) {
      std::vector<int32_t> fl;
      std::vector<uint32_t> fe;
      ldpc_build_wht_tables(fl, fe);
      if ((rc = upload(d_flog, fl))) return rc;
      if ((rc = upload(d_fexp, fe))) return rc;
      tab.flog = d_flog.as<int32_t>();
      tab.fexp = d_fexp.as<uint32_t>() + W_LEVELS;
      // compact forms, kept in LDS by the *_lds kernels
      WhtCompact wc;
      ldpc_build_wht_compact(wc);
      std::vector<int32_t> img((size_t)W_IMAGE_WORDS, 0);
      memcpy(&img[W_IMG_XEXP], wc.xexp.data(), sizeof(uint32_t) * kWhtFrac);
      memcpy(&img[W_IMG_SMALL], fl.data(), sizeof(int32_t) * kWhtSmall);
      for (int h = 0; h < kWhtSeg; h++) {
        memcpy(&img[W_IMG_SEG + 4 * h], &wc.lseg[(size_t)h * 4], sizeof(float) * 3);
        img[W_IMG_SEG + 4 * h + 3] = wc.lsegi[h];
      }
      if ((rc = upload(d_wimg, img))) return rc;
      tab.image = d_wimg.as<uint32_t>();
      // workgroup shape of the LDS-table kernels: as many wavefronts as the per-wave rows leave room for
      int env_c = 1;
      if (const char *e = getenv("LDPC_HIP_GF_LDS_TABLES")) env_c = atoi(e);
      const size_t img_bytes = (size_t)W_IMAGE_WORDS * 4, row_bytes = (size_t)std::max(1, rmax) * 64 * 4;
      cn_waves = (int)std::min<size_t>(16, (158 * 1024 - img_bytes) / row_bytes);
      lds_tables = env_c != 0 && cn_waves >= 4;
      if (lds_tables) {
        cn_lds_kern = rmax <= 4 ? gf2p_cn_lds_kernel<4> : rmax <= 8 ? gf2p_cn_lds_kernel<8> : gf2p_cn_lds_kernel<0>;
        vn_lds_kern = c.cmax <= 4 ? gf2p_vn_lds_kernel<4> : c.cmax <= 8 ? gf2p_vn_lds_kernel<8> : gf2p_vn_lds_kernel<0>;
        HIPCHK(hipFuncSetAttribute((const void *)cn_lds_kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(img_bytes + cn_waves * row_bytes)));
        HIPCHK(hipFuncSetAttribute((const void *)vn_lds_kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)img_bytes));
        // LDPC_HIP_GF_LDS_TABLES=2 keeps the generic LDS-table kernels for these shapes as well
        if (env_c == 1 && (log2q == 2 || log2q == 3) && rmax <= 8 && c.cmax <= 8) {
          if (log2q == 2) {
            cn_q_kern = rmax <= 4 ? gf2p_cn_q_kernel<2, 4> : gf2p_cn_q_kernel<2, 8>;
            vn_q_kern = c.cmax <= 4 ? gf2p_vn_q_kernel<2, 4> : gf2p_vn_q_kernel<2, 8>;
          } else {
            cn_q_kern = rmax <= 4 ? gf2p_cn_q_kernel<3, 4> : gf2p_cn_q_kernel<3, 8>;
            vn_q_kern = c.cmax <= 4 ? gf2p_vn_q_kernel<3, 4> : gf2p_vn_q_kernel<3, 8>;
          }
          HIPCHK(hipFuncSetAttribute((const void *)cn_q_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_bytes));
          HIPCHK(hipFuncSetAttribute((const void *)vn_q_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_bytes));
        }
      }
    }
    const size_t per = (size_t)Fpad_max * Q * sizeof(T);
    if ((rc = fna.ensure(per * n))) return rc;
    if ((rc = qa.ensure(per * E))) return rc;
    if ((rc = ra.ensure(per * E))) return rc;
    if ((rc = xhat.ensure((size_t)Fpad_max * n))) return rc;
    if ((rc = zhat.ensure((size_t)Fpad_max * m))) return rc;
    if ((rc = synd.ensure((size_t)Fpad_max * m))) return rc;
    if ((rc = st_i32.ensure((size_t)Fpad_max * 6 * sizeof(int32_t)))) return rc;
    if ((rc = st_frozen.ensure((size_t)Fpad_max))) return rc;
    if ((rc = ensure_remaining(256))) return rc;
    for (auto &e : ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (lds > 64 * 1024) {
      if (WHT)
        HIPCHK(hipFuncSetAttribute((const void *)gf2p_cn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
      else
      {
        HIPCHK(hipFuncSetAttribute((const void *)gfq_cn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
        // (the probability-domain kernel parks rho in HBM for heavy checks: it never needs > 64 KB)
      }
    }
    return LDPC_OK;
  }

  static unsigned blocks_for(size_t items, int per_block) {
    size_t b = (items + per_block - 1) / per_block;
    if (b > 65536) b = 65536;
    if (b < 1) b = 1;
    return (unsigned)b;
  }

  // Two stages when frames may leave early and the batch is large (round 3; the reference's dec() returns as
  // soon as ITS frame is done, GFq_LDPC.c:459-463, GFq_LDPC_NTT.c:527-540): the [edge][frame][Q] rows keep every
  // frame of a batch in the passes until the last one stops, so a batch in which most frames converge in a few
  // iterations and a handful need max_iter costs max_iter for all.  The first stage stops once at most 1/16 of
  // the frames are open (and >= 8 iterations remain); those frames are decoded again from their priors as a dense
  // batch.  dec() is deterministic: every frame's word, iteration count, syndrome and verdict are the one-stage
  // decode's.  LDPC_HIP_TWO_STAGE=0 turns it off.
  int decode(int nframes, const void *llr, int dt, const uint8_t *sy_in, int max_iter, uint8_t *bits,
             int32_t *iters_out, uint8_t *sout_out, int32_t *status, float *llr_out,
             hipStream_t st) override {
    const char *ts_env = getenv("LDPC_HIP_TWO_STAGE");
    const bool two = (!ts_env || atoi(ts_env) != 0) && opts.early_exit != 0 && nframes >= 16 * G && nframes >= 256 &&
                     max_iter >= 16 && (dt == LDPC_DT_F64 || dt == LDPC_DT_F32 || dt == LDPC_DT_I32);
    if (!two) return decode_once(nframes, llr, dt, sy_in, max_iter, bits, iters_out, sout_out, status, llr_out, st, 0, nullptr);
    bool stopped = false;
    int rc = decode_once(nframes, llr, dt, sy_in, max_iter, bits, iters_out, sout_out, status, llr_out, st, nframes / 16, &stopped);
    if (rc || !stopped) return rc;
    const int stage1 = last_iterations;
    if ((rc = s2_idx.ensure(sizeof(int) * ((size_t)nframes / 16 + 64)))) return rc;
    int *d_count = s2_idx.as<int>(), *d_idx = s2_idx.as<int>() + 32;
    HIPCHK(hipMemsetAsync(d_count, 0, sizeof(int), st));
    hipLaunchKernelGGL(gf_list_open_kernel, dim3((nframes + 255) / 256), dim3(256), 0, st, nframes, fs.frozen, d_count, d_idx);
    LAUNCHCHK("gf_list_open_kernel");
    int count = 0;
    HIPCHK(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (count <= 0) return LDPC_OK;
    if (count > nframes / 16) { ldpc_set_error("two-stage decode: more open frames than the first stage reported"); return LDPC_ERR_HIP; }
    const size_t in_row = (size_t)n * Q * (dt == LDPC_DT_F64 ? 8 : 4);
    if ((rc = s2_llr.ensure(in_row * count))) return rc;
    if (sy_in && (rc = s2_synd.ensure((size_t)m * count))) return rc;
    if (bits && (rc = s2_bits.ensure((size_t)n * count))) return rc;
    if (iters_out && (rc = s2_iters.ensure(sizeof(int32_t) * count))) return rc;
    if (sout_out && (rc = s2_sout.ensure((size_t)m * count))) return rc;
    if (status && (rc = s2_status.ensure(sizeof(int32_t) * count))) return rc;
    gf_move_rows<false>(llr, s2_llr.p, d_idx, count, in_row, st);
    if (sy_in) gf_move_rows<false>(sy_in, s2_synd.p, d_idx, count, (size_t)m, st);
    LAUNCHCHK("gf_move_rows_kernel (gather)");
    rc = decode_once(count, s2_llr.p, dt, sy_in ? s2_synd.as<uint8_t>() : nullptr, max_iter, bits ? s2_bits.as<uint8_t>() : nullptr,
                     iters_out ? s2_iters.as<int32_t>() : nullptr, sout_out ? s2_sout.as<uint8_t>() : nullptr,
                     status ? s2_status.as<int32_t>() : nullptr, nullptr, st, 0, nullptr);
    if (rc) return rc;
    if (bits) gf_move_rows<true>(s2_bits.p, bits, d_idx, count, (size_t)n, st);
    if (iters_out) gf_move_rows<true>(s2_iters.p, iters_out, d_idx, count, sizeof(int32_t), st);
    if (sout_out) gf_move_rows<true>(s2_sout.p, sout_out, d_idx, count, (size_t)m, st);
    if (status) gf_move_rows<true>(s2_status.p, status, d_idx, count, sizeof(int32_t), st);
    LAUNCHCHK("gf_move_rows_kernel (scatter)");
    last_iterations = std::max(last_iterations, stage1);
    return LDPC_OK;
  }

  // stop_open > 0: leave the iteration loop once at most stop_open frames are still open, provided at least 8
  // iterations remain (*stopped tells); those frames come out as "not converged" and are decoded again
  int decode_once(int nframes, const void *llr, int dt, const uint8_t *sy_in, int max_iter, uint8_t *bits,
                  int32_t *iters_out, uint8_t *sout_out, int32_t *status, float *llr_out,
                  hipStream_t st, int stop_open, bool *stopped) {
    if (nframes <= 0) return LDPC_OK;
    prof_cur_frames = nframes;
    if (llr_out) {
      ldpc_set_error("llr_out is not available for the non-binary algorithms");
      return LDPC_ERR_ARG;
    }
    const int Fpad = (nframes + G - 1) / G * G;
    if (Fpad > Fpad_max || !llr || max_iter < 0 || max_iter > 65000) {
      ldpc_set_error("nframes/max_iter out of range for this decoder");
      return LDPC_ERR_ARG;
    }
    int rc;
    if ((rc = ensure_remaining(max_iter + 2))) return rc;
    GGraph gg = g;
    gg.Fpad = Fpad;
    const int nfg = Fpad / G;
    fs.dist = st_i32.as<int32_t>();
    fs.iters = fs.dist + Fpad_max;
    fs.status = fs.iters + Fpad_max;
    fs.iir = fs.status + Fpad_max;
    fs.prev = fs.iir + Fpad_max;
    fs.nodecr = fs.prev + Fpad_max;
    fs.frozen = st_frozen.as<uint8_t>();

    hipLaunchKernelGGL(gf_state_init_kernel, dim3((Fpad + 255) / 256), dim3(256), 0, st, fs, Fpad, nframes);
    LAUNCHCHK("gf_state_init_kernel");
    HIPCHK(hipMemsetAsync(remaining.p, 0, sizeof(int32_t) * (max_iter + 2), st));
    HIPCHK(hipMemsetAsync(xhat.p, 0, (size_t)Fpad * n, st));
    HIPCHK(hipMemsetAsync(zhat.p, 0, (size_t)Fpad * m, st));

    const unsigned gi = blocks_for((size_t)n * Fpad * Q, 256);
    if (WHT) {
      if (dt == LDPC_DT_I32)
        hipLaunchKernelGGL((gf_ingest_kernel<int, T, WHT>), dim3(gi), dim3(256), 0, st, (const int *)llr,
                           nframes, gg, fna.as<T>());
      else if (dt == LDPC_DT_F64)
        hipLaunchKernelGGL((gf_ingest_kernel<double, T, WHT>), dim3(gi), dim3(256), 0, st, (const double *)llr,
                           nframes, gg, fna.as<T>());
      else if (dt == LDPC_DT_F32)
        hipLaunchKernelGGL((gf_ingest_kernel<float, T, WHT>), dim3(gi), dim3(256), 0, st, (const float *)llr,
                           nframes, gg, fna.as<T>());
      else { ldpc_set_error("GF2P_WHT takes I32 (raw Q.14), F64 or F32 log2 values"); return LDPC_ERR_ARG; }
    } else {
      if (dt == LDPC_DT_F64)
        hipLaunchKernelGGL((gf_ingest_kernel<double, T, WHT>), dim3(gi), dim3(256), 0, st, (const double *)llr,
                           nframes, gg, fna.as<T>());
      else if (dt == LDPC_DT_F32)
        hipLaunchKernelGGL((gf_ingest_kernel<float, T, WHT>), dim3(gi), dim3(256), 0, st, (const float *)llr,
                           nframes, gg, fna.as<T>());
      else { ldpc_set_error("GFQ_DIRECT takes F64 or F32 natural-log values"); return LDPC_ERR_ARG; }
    }
    LAUNCHCHK("gf_ingest_kernel");
    const uint8_t *sy = nullptr;
    if (sy_in) {
      hipLaunchKernelGGL(gf_sym_in_kernel, dim3(blocks_for((size_t)m * Fpad, 256)), dim3(256), 0, st, sy_in,
                         nframes, m, Fpad, synd.as<uint8_t>());
      LAUNCHCHK("gf_sym_in_kernel");
      sy = synd.as<uint8_t>();
    }
    if (((size_t)Fpad * Q * sizeof(T)) % 16 == 0)
      hipLaunchKernelGGL((gf_init_rows_kernel<T>), dim3((unsigned)std::min<size_t>((size_t)E, (size_t)num_cu * 64)), dim3(256), 0, st, gg,
                       fna.as<T>(), qa.as<T>());
    else
      hipLaunchKernelGGL((gf_init_kernel<T>), dim3(blocks_for((size_t)E * Fpad * Q, 256)), dim3(256), 0, st, gg,
                       fna.as<T>(), qa.as<T>());
    LAUNCHCHK("gf_init_kernel");

    const size_t lds = (size_t)(rmax + 2) * 64 * 8;
    const unsigned vn_blocks = (unsigned)(((size_t)n * nfg + 3) / 4);
    int executed = 0;
    const bool early = opts.early_exit != 0;
    // GFQ_DIRECT: cn_variant 1 = the reference's log-sum-exp recursions term by term
    // (gfq_cn_kernel); default = the same recursions on probabilities (gfq_cn_lin_kernel)
    int variant = opts.cn_variant;
    if (const char *e = getenv("LDPC_HIP_CN_VARIANT")) variant = atoi(e);
    const bool literal_cn = variant == 1;
    (void)literal_cn;
    for (int t = 0; t < max_iter; t++) {
      prof_begin(K_CN, st);
      if constexpr (WHT) {
        if (lds_tables && cn_q_kern && (size_t)std::max(m, n) * nfg < (1u << 31)) {
          const size_t want = ((size_t)m * nfg + 15) / 16;
          hipLaunchKernelGGL(cn_q_kern, dim3((unsigned)std::min<size_t>(want, (size_t)2 * num_cu)), dim3(1024),
                             (size_t)W_IMAGE_WORDS * 4, st, gg, tab, nfg, std::max(1, rmax), qa.as<int>(), ra.as<int>(), sy);
        } else if (lds_tables) {
          const size_t lb = (size_t)W_IMAGE_WORDS * 4 + (size_t)cn_waves * std::max(1, rmax) * 64 * 4;
          const size_t want = ((size_t)m * nfg + cn_waves - 1) / cn_waves;
          hipLaunchKernelGGL(cn_lds_kern, dim3((unsigned)std::min<size_t>(want, (size_t)num_cu)), dim3(cn_waves * 64),
                             lb, st, gg, tab, nfg, std::max(1, rmax), qa.as<int>(), ra.as<int>(), sy);
        } else {
          hipLaunchKernelGGL(gf2p_cn_kernel, dim3((unsigned)((size_t)m * nfg)), dim3(64), lds, st, gg, tab, nfg,
                             qa.as<int>(), ra.as<int>(), sy);
        }
      } else if (literal_cn)
        hipLaunchKernelGGL(gfq_cn_kernel, dim3((unsigned)((size_t)m * nfg)), dim3(64), lds, st, gg, nfg,
                           qa.as<double>(), ra.as<double>(), sy);
      else if (rmax > 8) {   // heavy checks: rho parked in ra, three LDS rows
        auto kern = Q == 13 ? gfq_cn_lin_kernel<13, true> : Q == 4 ? gfq_cn_lin_kernel<4, true>
                  : Q == 8  ? gfq_cn_lin_kernel<8, true>  : gfq_cn_lin_kernel<0, true>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)m * nfg)), dim3(64), 3 * 64 * 8, st, gg, nfg,
                           qa.as<double>(), ra.as<double>(), sy);
      } else {
        auto kern = Q == 13 ? gfq_cn_lin_kernel<13, false> : Q == 4 ? gfq_cn_lin_kernel<4, false>
                  : Q == 8  ? gfq_cn_lin_kernel<8, false>  : gfq_cn_lin_kernel<0, false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)m * nfg)), dim3(64), lds, st, gg, nfg, qa.as<double>(),
                           ra.as<double>(), sy);
      }
      LAUNCHCHK("gf cn kernel");
      prof_end(st);
      prof_begin(K_VN, st);
      if constexpr (WHT) {
        if (lds_tables && vn_q_kern && (size_t)std::max(m, n) * nfg < (1u << 31)) {
          const size_t want = ((size_t)n * nfg + 15) / 16;
          hipLaunchKernelGGL(vn_q_kern, dim3((unsigned)std::min<size_t>(want, (size_t)2 * num_cu)), dim3(1024),
                             (size_t)W_IMAGE_WORDS * 4, st, gg, tab, nfg, fna.as<int>(), ra.as<int>(), qa.as<int>(),
                             xhat.as<uint8_t>(), fs.frozen);
        } else if (lds_tables) {
          const size_t want = ((size_t)n * nfg + 15) / 16;
          hipLaunchKernelGGL(vn_lds_kern, dim3((unsigned)std::min<size_t>(want, (size_t)num_cu)), dim3(1024),
                             (size_t)W_IMAGE_WORDS * 4, st, gg, tab, nfg, fna.as<int>(), ra.as<int>(), qa.as<int>(),
                             xhat.as<uint8_t>(), fs.frozen);
        } else {
          hipLaunchKernelGGL(gf2p_vn_kernel, dim3(vn_blocks), dim3(256), 0, st, gg, tab, nfg, fna.as<int>(),
                             ra.as<int>(), qa.as<int>(), xhat.as<uint8_t>(), fs.frozen);
        }
      } else
        hipLaunchKernelGGL(gfq_vn_kernel, dim3(vn_blocks), dim3(256), 0, st, gg, nfg, fna.as<double>(),
                           ra.as<double>(), qa.as<double>(), xhat.as<uint8_t>(), fs.frozen);
      LAUNCHCHK("gf vn kernel");
      prof_end(st);
      if (Q <= 16 && Fpad % 4 == 0) {
        const int synd_per = (int)std::max<size_t>(GF_SYND_WAVES, std::min<size_t>(64, (size_t)m * Fpad / (256 * 1024)));
        hipLaunchKernelGGL(gf_synd4_kernel, dim3((unsigned)((Fpad + 255) / 256), (unsigned)((m + synd_per - 1) / synd_per)), dim3(64 * GF_SYND_WAVES), 0, st, gg,
                           xhat.as<uint8_t>(), sy, zhat.as<uint8_t>(), fs.dist, fs.frozen, synd_per);
      } else {
        const int synd_per = (int)std::max<size_t>(GF_SYND_WAVES, std::min<size_t>(64, (size_t)m * Fpad / (64 * 1024)));
        hipLaunchKernelGGL(gf_synd_kernel, dim3((unsigned)((Fpad + 63) / 64), (unsigned)((m + synd_per - 1) / synd_per)), dim3(64 * GF_SYND_WAVES), 0, st, gg,
                           xhat.as<uint8_t>(), sy, zhat.as<uint8_t>(), fs.dist, fs.frozen, synd_per);
      }
      LAUNCHCHK("gf_synd_kernel");
      hipLaunchKernelGGL(gf_state_kernel, dim3((Fpad + 255) / 256), dim3(256), 0, st, fs, Fpad, t, WHT ? 1 : 0,
                         remaining.as<int32_t>() + t);
      LAUNCHCHK("gf_state_kernel");
      executed = t + 1;
      if (early) {
        HIPCHK(hipMemcpyAsync(h_remaining + t, remaining.as<int32_t>() + t, sizeof(int32_t),
                              hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(ev[t % 4], st));
        if (t >= 1) {
          HIPCHK(hipEventSynchronize(ev[(t - 1) % 4]));
          if (h_remaining[t - 1] == 0) break;
          if (stop_open > 0 && h_remaining[t - 1] <= stop_open && max_iter - (t + 1) >= 8) { *stopped = true; break; }
        }
      }
    }
    last_iterations = executed;
    if (bits) {
      hipLaunchKernelGGL(gf_sym_out_kernel, dim3(blocks_for((size_t)n * nframes, 256)), dim3(256), 0, st,
                         xhat.as<uint8_t>(), nframes, n, Fpad, bits);
      LAUNCHCHK("gf_sym_out_kernel");
    }
    if (sout_out) {
      hipLaunchKernelGGL(gf_sym_out_kernel, dim3(blocks_for((size_t)m * nframes, 256)), dim3(256), 0, st,
                         zhat.as<uint8_t>(), nframes, m, Fpad, sout_out);
      LAUNCHCHK("gf_sym_out_kernel(synd)");
    }
    if (iters_out || status) {
      hipLaunchKernelGGL(gf_state_out_kernel, dim3((nframes + 255) / 256), dim3(256), 0, st, fs, nframes,
                         iters_out, status);
      LAUNCHCHK("gf_state_out_kernel");
    }
    return LDPC_OK;
  }
};

}  // namespace gf

Engine *ldpc_make_gf_engine(int alg, const ldpc_code &c) {
  if (c.q <= 2) return nullptr;
  if (alg == LDPC_ALG_GF2P_WHT) return new gf::GfEngine<int, true>;
  if (alg == LDPC_ALG_GFQ_DIRECT) return new gf::GfEngine<double, false>;
  return nullptr;
}
