// mc_frontend.hip -- the Monte-Carlo loop around dec(), on the device.
//
// The reference's *_cw_noise_gen harnesses (test_code_min_sum_B.c:288-329,
// test_code_min_sum_B_fixed.c:396-440, test_code_B_float.c:280-325,
// test_code_no_mex_B.c:321-381) run, per trial:  x = random bits, s = enc(x)
// (:69-79), y/q0 = bsc(x, p) (:41-67), dec(q0, s), then classify: not converged
// -> errors[0]++, converged to a word != x -> errors[1]++ (HamDist, :81-90).
// Here a batch of trials is one pass of four kernels that never leaves HBM:
//
//   mc_gen_kernel      source word, channel, LLRs in the decoder's input type,
//                      syndrome.  One workgroup builds FPW frames: the frames' x
//                      and error patterns live in LDS as bitmaps, the outputs are
//                      written frame-major with coalesced rows.
//   (decoder)          ldpc_decode_batch_device() on those buffers
//   mc_classify_kernel Hamming distance of every decoded frame to its x and the
//                      status, reduced into eight 64-bit counters by atomics.
//
// Random numbers: the reference seeds libc rand() from time(NULL), so its streams are
// not reproducible by design.  Here every (seed, global frame index, stream) names a
// counter-based splitmix64 sequence -- draw i = mix64(key + (i+1)*GOLDEN) -- so a
// frame's data does not depend on batch size, GPU count or launch order.  The tests hold a
// numpy restatement of this file's generators.
//
// Channels: BSC_EXACT is the reference's bsc(): exactly round(n p) distinct positions,
// found by rejection on uniform draws, LLR magnitude log((1-p')/p') with p' = flips/n.
// BSC_IID flips every bit independently with probability p (LLR magnitude from p).
// AWGN: BPSK, y = (1 - 2x) + sigma N(0,1), LLR = 2 y / sigma^2 (Box-Muller).
// QLC: the multi-read flash front-end of test_code_B_MSDP.c: four bits per cell, Gray
// mapped to one of 16 levels (:422), the read drawn from the caller's confusion matrix by
// the reference's percent-granular inverse CDF (:424-436), bit LLRs looked up in the
// table that make_p_sent_given_rec_T() / assign_llr_one_sym() (:297-395) define.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "engine.h"
#include "mc_rng.h"

namespace {

using ldpc_rng::GOLDEN;
using ldpc_rng::mix64;
using ldpc_rng::stream_key;
using ldpc_rng::draw;

struct McArgs {
  int n, m, nw;               // nw = ceil(n / 32) bitmap words per frame
  int nframes, fpw;           // frames in this pass, frames per workgroup
  int channel, source, dt;
  int flips;                  // BSC_EXACT: round(n p)
  double llr_mag;             // BSC: log((1-p)/p)
  double p, sigma;
  uint64_t seed, first_frame;
  const int32_t *row_ptr, *col_idx;
  uint8_t *x;                 // [nframes][n]
  void *llr;                  // [nframes][n] of dt
  uint8_t *synd;              // [nframes][m]
  // QLC channel: column-normalised confusion matrix [rows][16] and bit-LLR table [rows][4]
  const double *conf, *llr_tab;
  int rows;
};

// float_to_fix() of test_code_min_sum_B_fixed.c:43-53, raw (value * 256)
__device__ inline int q88_raw(double v) {
  double r = round(v * 256.0);
  if (r > 127.0 * 256.0) r = 127.0 * 256.0;
  if (r < -128.0 * 256.0) r = -128.0 * 256.0;
  return (int)r;
}
// float2fix() of test_code_B.c:36-43
__device__ inline int q14_raw(double x) {
  if (x >= 0) return (int)(x * 16384.0 + 0.5);
  return -(int)((-x) * 16384.0 + 0.5);
}
__device__ inline void store_llr(void *base, int dt, size_t at, double d) {
  switch (dt) {
    case LDPC_DT_F64: ((double *)base)[at] = d; break;
    case LDPC_DT_F32: ((float *)base)[at] = (float)d; break;
    case LDPC_DT_I16: ((int16_t *)base)[at] = (int16_t)q88_raw(d); break;
    default: ((int32_t *)base)[at] = q14_raw(d); break;
  }
}

__global__ __launch_bounds__(256) void mc_gen_kernel(McArgs a) {
  extern __shared__ uint32_t bm[];            // [fpw][2][nw]: x bitmap, error bitmap
  const int tid = threadIdx.x, B = blockDim.x;
  const int f0 = blockIdx.x * a.fpw;
  const int nf = min(a.fpw, a.nframes - f0);
  if (nf <= 0) return;
  const int nw = a.nw;
  const uint32_t tailmask = (a.n & 31) ? ((1u << (a.n & 31)) - 1u) : 0xffffffffu;

  // ---- source words (x[j] = rand() & 1, :308-311) and cleared error bitmaps
  for (int t = tid; t < nf * nw; t += B) {
    const int f = t / nw, w = t % nw;
    uint32_t xb = 0;
    if (a.source != 2) {
      const uint64_t fr = a.source == 1 ? ~0ull : a.first_frame + (uint64_t)(f0 + f);
      xb = (uint32_t)draw(stream_key(a.seed, fr, 0), (uint64_t)w);
      if (w == nw - 1) xb &= tailmask;
    }
    bm[(f * 2 + 0) * nw + w] = xb;
    bm[(f * 2 + 1) * nw + w] = 0;
  }
  __syncthreads();

  // ---- channel error patterns
  if (a.channel == 0) {
    // bsc(), :41-67: draw positions until `flips` distinct ones are marked; one thread per
    // frame walks its own bitmap, so the pattern depends on the frame's stream alone
    if (tid < nf) {
      uint32_t *eb = bm + (tid * 2 + 1) * nw;
      const uint64_t key = stream_key(a.seed, a.first_frame + (uint64_t)(f0 + tid), 1);
      int num = 0;
      for (uint64_t i = 0; num < a.flips; i++) {
        const uint32_t pos = (uint32_t)(((draw(key, i) >> 32) * (uint64_t)a.n) >> 32);
        const uint32_t bit = 1u << (pos & 31);
        const uint32_t w = eb[pos >> 5];
        if (w & bit) continue;
        eb[pos >> 5] = w | bit;
        num++;
      }
    }
  } else if (a.channel == 1) {
    const uint64_t thr = a.p >= 1.0 ? ~0ull : (uint64_t)(a.p * 18446744073709551616.0);
    for (int t = tid; t < nf * nw; t += B) {
      const int f = t / nw, w = t % nw;
      const uint64_t key = stream_key(a.seed, a.first_frame + (uint64_t)(f0 + f), 1);
      uint32_t e = 0;
      for (int b = 0; b < 32; b++)
        if (draw(key, (uint64_t)w * 32 + b) < thr) e |= 1u << b;
      if (w == nw - 1) e &= tailmask;
      bm[(f * 2 + 1) * nw + w] = e;
    }
  }
  __syncthreads();

  // ---- x, LLRs (frame-major rows, coalesced)
  // The two-valued channels (BSC) with int16 / binary32 outputs, rows of a multiple of 8 variables:
  // eight variables per thread and step -- one byte of each bitmap, one 8-byte store of x, one
  // 16- or 32-byte store of LLRs, the two possible values converted once.
  const bool two_valued = a.channel == 0 || a.channel == 1;
  const bool fast8 = two_valued && (a.n & 7) == 0 && (a.dt == LDPC_DT_I16 || a.dt == LDPC_DT_F32) &&
                     ((size_t)a.x & 7) == 0 && ((size_t)a.llr & 15) == 0;
  if (fast8) {
    const int praw = q88_raw(a.llr_mag), nraw = q88_raw(-a.llr_mag);
    const float pf = (float)a.llr_mag;
    const int n8 = a.n >> 3;
    for (int t = tid; t < nf * n8; t += B) {
      const int f = t / n8, i8 = t - f * n8;
      const uint32_t xbyte = (bm[(f * 2 + 0) * nw + (i8 >> 2)] >> ((i8 & 3) * 8)) & 0xffu;
      const uint32_t ebyte = (bm[(f * 2 + 1) * nw + (i8 >> 2)] >> ((i8 & 3) * 8)) & 0xffu;
      const uint32_t ybyte = xbyte ^ ebyte;
      const size_t at = (size_t)(f0 + f) * a.n + (size_t)i8 * 8;
      // bit k of the byte -> byte k of the word pair
      uint2 xs;
      xs.x = ((xbyte & 1u) | ((xbyte & 2u) << 7) | ((xbyte & 4u) << 14) | ((xbyte & 8u) << 21));
      xs.y = (((xbyte >> 4) & 1u) | (((xbyte >> 4) & 2u) << 7) | (((xbyte >> 4) & 4u) << 14) | (((xbyte >> 4) & 8u) << 21));
      *(uint2 *)(a.x + at) = xs;
      if (a.dt == LDPC_DT_I16) {
        uint4 v;
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int lo = ((ybyte >> (2 * k)) & 1u) ? nraw : praw, hi = ((ybyte >> (2 * k + 1)) & 1u) ? nraw : praw;
          w[k] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
        }
        v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        *(uint4 *)((int16_t *)a.llr + at) = v;
      } else {
        float4 v0, v1;
        v0.x = (ybyte & 1u) ? -pf : pf;   v0.y = (ybyte & 2u) ? -pf : pf;
        v0.z = (ybyte & 4u) ? -pf : pf;   v0.w = (ybyte & 8u) ? -pf : pf;
        v1.x = (ybyte & 16u) ? -pf : pf;  v1.y = (ybyte & 32u) ? -pf : pf;
        v1.z = (ybyte & 64u) ? -pf : pf;  v1.w = (ybyte & 128u) ? -pf : pf;
        *(float4 *)((float *)a.llr + at) = v0;
        *(float4 *)((float *)a.llr + at + 4) = v1;
      }
    }
  }
  for (int f = 0; f < (fast8 ? 0 : nf); f++) {
    const size_t row = (size_t)(f0 + f) * a.n;
    const uint32_t *xb = bm + (f * 2 + 0) * nw, *eb = bm + (f * 2 + 1) * nw;
    const uint64_t key = stream_key(a.seed, a.first_frame + (uint64_t)(f0 + f), 1);
    for (int i = tid; i < a.n; i += B) {
      const int xi = (xb[i >> 5] >> (i & 31)) & 1;
      a.x[row + i] = (uint8_t)xi;
      double d;
      if (a.channel == 3) continue;            // QLC: LLRs are written per cell below
      if (a.channel == 2) {
        const double u1 = (double)((draw(key, 2 * (uint64_t)i) >> 11) + 1) * (1.0 / 9007199254740992.0);
        const double u2 = (double)(draw(key, 2 * (uint64_t)i + 1) >> 11) * (1.0 / 9007199254740992.0);
        const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
        const double y = (double)(1 - 2 * xi) + a.sigma * g;
        d = 2.0 * y / (a.sigma * a.sigma);
      } else {
        const int yi = xi ^ (int)((eb[i >> 5] >> (i & 31)) & 1);
        d = (double)(1 - 2 * yi) * a.llr_mag;   // q0[i] = (1 - 2 y[i]) log((1-p)/p), :63-66
      }
      store_llr(a.llr, a.dt, row + i, d);
    }
  }

  if (a.channel == 3) {
    // channel() of test_code_B_MSDP.c:411-440, one thread per cell
    const int grey_code_inv[16] = {6, 5, 7, 14, 9, 12, 8, 13, 3, 4, 2, 15, 10, 11, 1, 0};   // :23
    const int cells = a.n >> 2;
    for (int t = tid; t < nf * cells; t += B) {
      const int f = t / cells, c = t % cells;
      const uint32_t *xb = bm + (f * 2 + 0) * nw;
      const int nib = (xb[c >> 3] >> ((c & 7) * 4)) & 15;      // x[4c+3..4c] as (tp up mp lp)
      const int symbol = grey_code_inv[nib];
      const uint64_t key = stream_key(a.seed, a.first_frame + (uint64_t)(f0 + f), 1);
      const int rand_select = (int)(((draw(key, (uint64_t)c) >> 32) * 101ull) >> 32);   // rand() % 101
      double temp = 0;
      int rec = 0;
      if (!rand_select)
        while (rec < a.rows && !a.conf[(rec++) * 16 + symbol]) {}
      {
        // separately rounded product and sum, as the reference's x86-64 build computes them:
        // a fused multiply-add moves the walk by one bin when the sum lands on rand_select
#pragma clang fp contract(off)
        while (temp < rand_select && rec != a.rows) {
          const double step = 100 * a.conf[(rec++) * 16 + symbol];
          temp = temp + step;
        }
      }
      rec = rec > 0 ? rec - 1 : 0;
      const size_t at = (size_t)(f0 + f) * a.n + 4 * (size_t)c;
      for (int b = 0; b < 4; b++) store_llr(a.llr, a.dt, at + b, a.llr_tab[rec * 4 + b]);
    }
  }

  // ---- syndrome, enc() :69-79
  for (int t = tid; t < nf * a.m; t += B) {
    const int f = t / a.m, j = t % a.m;
    const uint32_t *xb = bm + (f * 2 + 0) * nw;
    uint32_t k = 0;
    for (int e = a.row_ptr[j]; e < a.row_ptr[j + 1]; e++) {
      const int v = a.col_idx[e];
      k ^= xb[v >> 5] >> (v & 31);
    }
    a.synd[(size_t)(f0 + f) * a.m + j] = (uint8_t)(k & 1);
  }
}

// counters: 0 frames, 1 not converged, 2 converged to a word != x, 3 bit errors (all
// frames), 4 iterations, 5 frames with any bit error, 6 bit errors of converged frames
// One wavefront per frame at a time, many frames per workgroup, the counts carried in registers and
// merged through LDS: seven device atomics per WORKGROUP.  (One workgroup and up to six device
// atomics per FRAME, as this kernel first was, took 70 % of a (204,102) Monte-Carlo pass: every
// frame of the batch queued on the same seven L2 addresses.)
__global__ __launch_bounds__(256) void mc_classify_kernel(int n, int nframes, const uint8_t *x, const uint8_t *bits,
                                                          const int32_t *status, const int32_t *iters,
                                                          unsigned long long *counters) {
  __shared__ unsigned long long acc[8];
  if (threadIdx.x < 8) acc[threadIdx.x] = 0ull;
  __syncthreads();
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned long long c[7] = {0, 0, 0, 0, 0, 0, 0};   // (the same in every lane of the wavefront)
  for (int f = blockIdx.x * nw + wave; f < nframes; f += gridDim.x * nw) {
    const uint8_t *xr = x + (size_t)f * n, *br = bits + (size_t)f * n;
    int diff = 0;
    if (((n & 3) | ((size_t)x & 3) | ((size_t)bits & 3)) == 0) {
      const uint32_t *xw = (const uint32_t *)xr, *bw = (const uint32_t *)br;
      for (int i = lane; i < (n >> 2); i += 64) diff += __popc((xw[i] ^ bw[i]) & 0x01010101u);
    } else {
      for (int i = lane; i < n; i += 64) diff += (xr[i] ^ br[i]) & 1;
    }
    for (int o = 32; o > 0; o >>= 1) diff += __shfl_xor(diff, o);
    const int st = status[f];
    c[0] += 1ull;
    if (st) c[1] += 1ull;
    else if (diff) c[2] += 1ull;
    if (diff) {
      c[3] += (unsigned long long)diff;
      c[5] += 1ull;
      if (!st) c[6] += (unsigned long long)diff;
    }
    c[4] += (unsigned long long)iters[f];
  }
  if (lane == 0)
    for (int k = 0; k < 7; k++)
      if (c[k]) atomicAdd(&acc[k], c[k]);
  __syncthreads();
  if (threadIdx.x < 7 && acc[threadIdx.x]) atomicAdd(&counters[threadIdx.x], acc[threadIdx.x]);
}

size_t dt_size(int dt) { return dt == LDPC_DT_F64 ? 8 : dt == LDPC_DT_I16 ? 2 : 4; }

}  // namespace

struct ldpc_mc {
  const ldpc_code *code = nullptr;
  ldpc_decoder *dec = nullptr;
  int device = 0, max_frames = 0;
  DevBuf d_row_ptr, d_col_idx, d_x, d_llr, d_synd, d_bits, d_iters, d_status, d_counters;
  DevBuf d_conf, d_llr_tab;   // QLC channel tables (ldpc_mc_set_qlc_channel)
  int qlc_rows = 0;
};

extern "C" int ldpc_mc_create(const ldpc_code *c, ldpc_decoder *d, int device, ldpc_mc **out) {
  if (!c || !d || !out) return LDPC_ERR_ARG;
  if (c->q != 2) {
    ldpc_set_error("the device Monte-Carlo front-end is for binary codes");
    return LDPC_ERR_UNSUPPORTED;
  }
  HIPCHK(hipSetDevice(device));
  ldpc_mc *mc = new ldpc_mc;
  mc->code = c; mc->dec = d; mc->device = device;
  mc->max_frames = ldpc_decoder_max_frames(d);
  const size_t F = (size_t)mc->max_frames, n = (size_t)c->n, m = (size_t)c->m;
  int rc;
  if ((rc = mc->d_row_ptr.ensure((m + 1) * 4)) || (rc = mc->d_col_idx.ensure((size_t)c->E() * 4 + 4)) ||
      (rc = mc->d_x.ensure(F * n)) || (rc = mc->d_llr.ensure(F * n * 8)) || (rc = mc->d_synd.ensure(F * m + 4)) ||
      (rc = mc->d_bits.ensure(F * n)) || (rc = mc->d_iters.ensure(F * 4)) || (rc = mc->d_status.ensure(F * 4)) ||
      (rc = mc->d_counters.ensure(64))) {
    delete mc;
    return rc;
  }
  hipError_t e = hipMemcpy(mc->d_row_ptr.p, c->row_ptr.data(), (m + 1) * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(mc->d_col_idx.p, c->col_idx.data(), (size_t)c->E() * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    ldpc_set_error(std::string("ldpc_mc_create: ") + hipGetErrorString(e));
    delete mc;
    return LDPC_ERR_HIP;
  }
  *out = mc;
  return LDPC_OK;
}

extern "C" void ldpc_mc_free(ldpc_mc *mc) {
  if (!mc) return;
  (void)hipSetDevice(mc->device);
  delete mc;
}

static int mc_check(const ldpc_mc *mc, const ldpc_mc_opts *o) {
  if (!mc || !o) return LDPC_ERR_ARG;
  if (o->channel < 0 || o->channel > 3 || o->source < 0 || o->source > 2 || o->llr_dtype < 0 || o->llr_dtype > 3 ||
      o->max_iter < 0) {
    ldpc_set_error("ldpc_mc: bad channel / source / dtype / max_iter");
    return LDPC_ERR_ARG;
  }
  if (o->channel == LDPC_CH_QLC) {
    if (!mc->qlc_rows) {
      ldpc_set_error("ldpc_mc: LDPC_CH_QLC needs ldpc_mc_set_qlc_channel() first");
      return LDPC_ERR_ARG;
    }
    if (mc->code->n % 4) {
      ldpc_set_error("ldpc_mc: the QLC front-end needs n to be a multiple of 4 (test_code_B_MSDP.c:397)");
      return LDPC_ERR_UNSUPPORTED;
    }
  } else if (o->channel == LDPC_CH_AWGN ? !(o->param > 0) : !(o->param > 0 && o->param < 1)) {
    ldpc_set_error("ldpc_mc: channel parameter out of range (BSC: 0 < p < 1, AWGN: sigma > 0)");
    return LDPC_ERR_ARG;
  }
  return LDPC_OK;
}

// one pass: frames [first, first + nf) into the context's buffers
static int mc_generate_pass(ldpc_mc *mc, const ldpc_mc_opts *o, uint64_t first, int nf, hipStream_t st) {
  const ldpc_code &c = *mc->code;
  McArgs a;
  memset(&a, 0, sizeof a);
  a.n = c.n; a.m = c.m; a.nw = (c.n + 31) / 32;
  a.nframes = nf;
  a.channel = o->channel; a.source = o->source; a.dt = o->llr_dtype;
  a.p = o->param; a.sigma = o->param;
  if (o->channel == LDPC_CH_BSC_EXACT) {
    int modified = (int)(c.n * o->param + 0.5);          // bsc(), :44-47
    if (modified > c.n) modified = c.n;
    a.flips = modified;
    const double p = modified / (double)c.n;
    a.llr_mag = std::log((1.0 - p) / p);                 // +inf when no bit flips, as in the reference
  } else if (o->channel == LDPC_CH_BSC_IID) {
    a.llr_mag = std::log((1.0 - o->param) / o->param);
  }
  a.conf = mc->d_conf.as<double>(); a.llr_tab = mc->d_llr_tab.as<double>(); a.rows = mc->qlc_rows;
  a.seed = o->seed; a.first_frame = first;
  a.row_ptr = mc->d_row_ptr.as<int32_t>(); a.col_idx = mc->d_col_idx.as<int32_t>();
  a.x = mc->d_x.as<uint8_t>(); a.llr = mc->d_llr.p; a.synd = mc->d_synd.as<uint8_t>();
  const size_t per_frame = (size_t)a.nw * 8;
  // (about 32 KB of bitmaps per workgroup: several 256-thread workgroups per CU; 96 KB -- one
  // workgroup, four wavefronts per CU -- left every phase of the kernel waiting on its own latencies)
  int fpw = (int)((size_t)(32 * 1024) / per_frame);
  if (fpw > 32) fpw = 32;
  if (fpw < 1) fpw = 1;
  a.fpw = fpw;
  const size_t lds = per_frame * fpw;
  if (lds > 64 * 1024)
    HIPCHK(hipFuncSetAttribute((const void *)mc_gen_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (nf + fpw - 1) / fpw;
  hipLaunchKernelGGL(mc_gen_kernel, dim3(grid), dim3(256), lds, st, a);
  LAUNCHCHK("mc_gen_kernel");
  return LDPC_OK;
}

// Tables of the QLC front-end from a column-normalised confusion matrix
// p_rec_given_sent[rows][16] (rows = 16 for one read, 46 for three: test_code_B_MSDP.c:302-306).
extern "C" int ldpc_mc_set_qlc_channel(ldpc_mc *mc, const double *p_rec_given_sent, int num_reads) {
  if (!mc || !p_rec_given_sent || (num_reads != 1 && num_reads != 3)) {
    ldpc_set_error("ldpc_mc_set_qlc_channel: num_reads must be 1 or 3");
    return LDPC_ERR_ARG;
  }
  constexpr int Q = 16;
  const int rows = num_reads == 1 ? Q : Q * num_reads - 2;
  // bits that read 1 on the lower / middle / upper / top page, :24-27
  static const int sym_low_1[8] = {0, 4, 5, 11, 12, 13, 14, 15}, sym_mid_1[8] = {0, 1, 2, 7, 8, 13, 14, 15},
                   sym_up_1[8] = {0, 1, 8, 9, 10, 11, 12, 13}, sym_top_1[8] = {0, 1, 2, 3, 4, 10, 11, 15};
  const int *sets[4] = {sym_low_1, sym_mid_1, sym_up_1, sym_top_1};
  std::vector<double> T((size_t)rows * Q), tab((size_t)rows * 4);
  const double P_x = 1 / (1.0 * Q);
  for (int i = 0; i < rows; i++) {                       // make_p_sent_given_rec_T(), :297-323
    double P_y = 0;
    for (int j = 0; j < Q; j++) P_y += p_rec_given_sent[i * Q + j] * P_x;
    for (int j = 0; j < Q; j++) T[(size_t)i * Q + j] = P_y ? P_x * p_rec_given_sent[i * Q + j] / P_y : 0.0;
    for (int b = 0; b < 4; b++) {                        // assign_llr_one_sym(), :339-395
      double Pr_1 = 0;
      for (int j = 0; j < 8; j++) Pr_1 += T[(size_t)i * Q + sets[b][j]];
      tab[(size_t)i * 4 + b] = Pr_1 == 1 ? -100.0 : Pr_1 == 0 ? 100.0 : std::log((1.0 - Pr_1) / Pr_1);
    }
  }
  HIPCHK(hipSetDevice(mc->device));
  int rc;
  if ((rc = mc->d_conf.ensure((size_t)rows * Q * 8)) || (rc = mc->d_llr_tab.ensure((size_t)rows * 4 * 8))) return rc;
  HIPCHK(hipMemcpy(mc->d_conf.p, p_rec_given_sent, (size_t)rows * Q * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(mc->d_llr_tab.p, tab.data(), (size_t)rows * 4 * 8, hipMemcpyHostToDevice));
  mc->qlc_rows = rows;
  return LDPC_OK;
}

extern "C" int ldpc_mc_generate(ldpc_mc *mc, const ldpc_mc_opts *o, int nframes, uint8_t *x, void *llr,
                                uint8_t *synd) {
  int rc = mc_check(mc, o);
  if (rc) return rc;
  if (nframes < 0 || nframes > mc->max_frames) {
    ldpc_set_error("ldpc_mc_generate: nframes exceeds the decoder's max_frames");
    return LDPC_ERR_ARG;
  }
  if (nframes == 0) return LDPC_OK;
  HIPCHK(hipSetDevice(mc->device));
  if ((rc = mc_generate_pass(mc, o, o->first_frame, nframes, nullptr))) return rc;
  HIPCHK(hipDeviceSynchronize());
  const size_t n = (size_t)mc->code->n, m = (size_t)mc->code->m;
  if (x) HIPCHK(hipMemcpy(x, mc->d_x.p, nframes * n, hipMemcpyDeviceToHost));
  if (llr) HIPCHK(hipMemcpy(llr, mc->d_llr.p, nframes * n * dt_size(o->llr_dtype), hipMemcpyDeviceToHost));
  if (synd) HIPCHK(hipMemcpy(synd, mc->d_synd.p, nframes * m, hipMemcpyDeviceToHost));
  return LDPC_OK;
}

extern "C" int ldpc_mc_run(ldpc_mc *mc, const ldpc_mc_opts *o, int64_t nframes, uint64_t *counters) {
  int rc = mc_check(mc, o);
  if (rc) return rc;
  if (nframes < 0 || !counters) return LDPC_ERR_ARG;
  HIPCHK(hipSetDevice(mc->device));
  HIPCHK(hipMemsetAsync(mc->d_counters.p, 0, 64, nullptr));
  const int n = mc->code->n;
  for (int64_t done = 0; done < nframes; done += mc->max_frames) {
    const int nf = (int)std::min<int64_t>(mc->max_frames, nframes - done);
    if ((rc = mc_generate_pass(mc, o, o->first_frame + (uint64_t)done, nf, nullptr))) return rc;
    rc = ldpc_decode_batch_device(mc->dec, nf, mc->d_llr.p, o->llr_dtype, mc->d_synd.as<uint8_t>(), o->max_iter,
                                  mc->d_bits.as<uint8_t>(), mc->d_iters.as<int32_t>(), nullptr,
                                  mc->d_status.as<int32_t>(), nullptr, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(mc_classify_kernel, dim3((unsigned)std::min(2048, (nf + 3) / 4)), dim3(256), 0, nullptr, n, nf, mc->d_x.as<uint8_t>(),
                       mc->d_bits.as<uint8_t>(), mc->d_status.as<int32_t>(), mc->d_iters.as<int32_t>(),
                       mc->d_counters.as<unsigned long long>());
    LAUNCHCHK("mc_classify_kernel");
  }
  HIPCHK(hipMemcpy(counters, mc->d_counters.p, 64, hipMemcpyDeviceToHost));
  return LDPC_OK;
}

// the decoded words / status of the last pass of ldpc_mc_run (tests): host pointers, any may be null
extern "C" int ldpc_mc_last_pass(ldpc_mc *mc, int nframes, uint8_t *x, uint8_t *bits, int32_t *status,
                                 int32_t *iters) {
  if (!mc || nframes < 0 || nframes > mc->max_frames) return LDPC_ERR_ARG;
  HIPCHK(hipSetDevice(mc->device));
  HIPCHK(hipDeviceSynchronize());
  const size_t n = (size_t)mc->code->n;
  if (x) HIPCHK(hipMemcpy(x, mc->d_x.p, nframes * n, hipMemcpyDeviceToHost));
  if (bits) HIPCHK(hipMemcpy(bits, mc->d_bits.p, nframes * n, hipMemcpyDeviceToHost));
  if (status) HIPCHK(hipMemcpy(status, mc->d_status.p, (size_t)nframes * 4, hipMemcpyDeviceToHost));
  if (iters) HIPCHK(hipMemcpy(iters, mc->d_iters.p, (size_t)nframes * 4, hipMemcpyDeviceToHost));
  return LDPC_OK;
}


// ---------------------------------------------------------------------------------------------
// The trial loop of GFq_LDPC() (GFq_LDPC.c:540-655; WD_farm/GFq_LDPC.c:477-633) on the device.
// Per trial the reference draws x[i] = rand() % Q (:626), s = enc(x) (:632), sends x through
// channel() (:213-248: the read y walks the column of the sent symbol in the confusion matrix with a
// percent-granular uniform draw, logfna[.][i] = log p(sent | y) with the -100 floor, :199-211), runs
// dec() and classifies.  Here: gfmc_gen_kernel (one thread per symbol), gfmc_synd_kernel (one thread per
// check), the decoder on those device buffers, gfmc_classify_kernel -- eight counters cross PCIe.
namespace {

struct GfmcArgs {
  int n, m, Q, R, nframes;
  uint64_t seed, first_frame;
  const double *conf;      // [R][Q] P(read r | sent a), column-normalised
  const double *logT;      // [R][Q] log P(sent a | read r), floor -100
  const int32_t *row_ptr, *col_idx, *hval;
  const uint8_t *mul;      // [Q][Q] GF(q) multiplication
  int pow2;                // field addition: XOR (2^p) or mod q (prime)
  uint8_t *x;              // [nframes][n]
  int32_t *y;              // [nframes][n] or null
  float *sel;              // [nframes][n] or null
  double *logfna;          // [nframes][n][Q]
  uint8_t *synd;           // [nframes][m]
};

__global__ __launch_bounds__(256) void gfmc_gen_kernel(GfmcArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)a.nframes * a.n) return;
  const int f = (int)(t / a.n), i = (int)(t % a.n);
  const uint64_t fr = a.first_frame + (uint64_t)f;
  const int xi = (int)ldpc_rng::below(draw(stream_key(a.seed, fr, 0), (uint64_t)i), (uint32_t)a.Q);
  const float sel = ldpc_rng::percent(draw(stream_key(a.seed, fr, 1), (uint64_t)i));
  // channel(), :213-231.  No fused multiply-add: the walk must land in the bin the reference's doubles land in.
  double temp = 0;
  int rec = 0;
  if (!sel)
    while (rec < a.R && !a.conf[(rec++) * a.Q + xi]) {}
  while (temp < (double)sel && rec != a.R) temp = __dadd_rn(temp, __dmul_rn(100.0, a.conf[(rec++) * a.Q + xi]));
  const int yi = rec > 0 ? rec - 1 : 0;
  a.x[t] = (uint8_t)xi;
  if (a.y) a.y[t] = yi;
  if (a.sel) a.sel[t] = sel;
  double *lf = a.logfna + t * a.Q;
  const double *row = a.logT + (size_t)yi * a.Q;
  for (int q = 0; q < a.Q; q++) lf[q] = row[q];
}

// enc(), GFq_LDPC.c:299-310
__global__ __launch_bounds__(256) void gfmc_synd_kernel(GfmcArgs a) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)a.nframes * a.m) return;
  const int f = (int)(t / a.m), j = (int)(t % a.m);
  const uint8_t *x = a.x + (size_t)f * a.n;
  int acc = 0;
  for (int e = a.row_ptr[j]; e < a.row_ptr[j + 1]; e++) {
    const int prod = a.mul[x[a.col_idx[e]] * a.Q + a.hval[e]];
    acc = a.pow2 ? (acc ^ prod) : (acc + prod) % a.Q;
  }
  a.synd[t] = (uint8_t)acc;
}

// as mc_classify_kernel, on symbols: 0 frames, 1 not converged, 2 converged to a word != x, 3 symbol errors,
// 4 iterations, 5 frames with any symbol error, 6 symbol errors of converged frames
__global__ __launch_bounds__(256) void gfmc_classify_kernel(int n, int nframes, const uint8_t *x, const uint8_t *xhat,
                                                            const int32_t *status, const int32_t *iters,
                                                            unsigned long long *counters) {
  __shared__ unsigned long long acc[8];
  if (threadIdx.x < 8) acc[threadIdx.x] = 0ull;
  __syncthreads();
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned long long c[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int f = blockIdx.x * nw + wave; f < nframes; f += gridDim.x * nw) {
    const uint8_t *xr = x + (size_t)f * n, *br = xhat + (size_t)f * n;
    int diff = 0;
    for (int i = lane; i < n; i += 64) diff += xr[i] != br[i];
    for (int o = 32; o > 0; o >>= 1) diff += __shfl_xor(diff, o);
    const int st = status[f];
    c[0] += 1ull;
    if (st) c[1] += 1ull;
    else if (diff) c[2] += 1ull;
    if (diff) {
      c[3] += (unsigned long long)diff;
      c[5] += 1ull;
      if (!st) c[6] += (unsigned long long)diff;
    }
    c[4] += (unsigned long long)iters[f];
  }
  if (lane == 0)
    for (int k = 0; k < 7; k++)
      if (c[k]) atomicAdd(&acc[k], c[k]);
  __syncthreads();
  if (threadIdx.x < 7 && acc[threadIdx.x]) atomicAdd(&counters[threadIdx.x], acc[threadIdx.x]);
}

}  // namespace

struct ldpc_gfmc {
  const ldpc_code *code = nullptr;
  ldpc_decoder *dec = nullptr;
  int device = 0, max_frames = 0, R = 0;
  DevBuf d_row_ptr, d_col_idx, d_hval, d_mul, d_conf, d_logT;
  DevBuf d_x, d_y, d_sel, d_logfna, d_synd, d_xhat, d_iters, d_status, d_counters;
};

extern "C" int ldpc_gfmc_create(const ldpc_code *c, ldpc_decoder *d, int device, ldpc_gfmc **out) {
  if (!c || !d || !out) return LDPC_ERR_ARG;
  if (c->q < 3 || c->q > 256) {
    ldpc_set_error("ldpc_gfmc: the GF(q) Monte-Carlo front-end needs a non-binary code");
    return LDPC_ERR_UNSUPPORTED;
  }
  HIPCHK(hipSetDevice(device));
  std::unique_ptr<ldpc_gfmc> mc(new ldpc_gfmc);
  mc->code = c; mc->dec = d; mc->device = device;
  mc->max_frames = ldpc_decoder_max_frames(d);
  const size_t F = (size_t)mc->max_frames, n = (size_t)c->n, m = (size_t)c->m, Q = (size_t)c->q, E = (size_t)c->E();
  int rc;
  if ((rc = mc->d_row_ptr.ensure((m + 1) * 4)) || (rc = mc->d_col_idx.ensure(E * 4 + 4)) || (rc = mc->d_hval.ensure(E * 4 + 4)) ||
      (rc = mc->d_mul.ensure(Q * Q)) || (rc = mc->d_x.ensure(F * n)) || (rc = mc->d_logfna.ensure(F * n * Q * 8)) ||
      (rc = mc->d_synd.ensure(F * m + 4)) || (rc = mc->d_xhat.ensure(F * n)) || (rc = mc->d_iters.ensure(F * 4)) ||
      (rc = mc->d_status.ensure(F * 4)) || (rc = mc->d_counters.ensure(64)))
    return rc;
  std::vector<uint8_t> mul(Q * Q);
  for (size_t a = 0; a < Q; a++)
    for (size_t b = 0; b < Q; b++) mul[a * Q + b] = (uint8_t)ldpc_gf_mul_host((int)Q, (int)a, (int)b);
  HIPCHK(hipMemcpy(mc->d_row_ptr.p, c->row_ptr.data(), (m + 1) * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(mc->d_col_idx.p, c->col_idx.data(), E * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(mc->d_hval.p, c->hval.data(), E * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(mc->d_mul.p, mul.data(), Q * Q, hipMemcpyHostToDevice));
  *out = mc.release();
  return LDPC_OK;
}

extern "C" void ldpc_gfmc_free(ldpc_gfmc *mc) {
  if (!mc) return;
  (void)hipSetDevice(mc->device);
  delete mc;
}

// make_p_sent_given_rec_T() + the -100 floor of assign_llr() (GFq_LDPC.c:177-211) from the column-normalised
// confusion matrix p_rec_given_sent[q * num_reads][q]
extern "C" int ldpc_gfmc_set_channel(ldpc_gfmc *mc, const double *p_rec_given_sent, int num_reads) {
  if (!mc || !p_rec_given_sent || num_reads <= 0) return LDPC_ERR_ARG;
  const int Q = mc->code->q, R = Q * num_reads;
  std::vector<double> logT((size_t)R * Q);
  const double P_x = 1 / (1.0 * Q);
  for (int i = 0; i < R; i++) {
    double P_y = 0;
    for (int j = 0; j < Q; j++) P_y += p_rec_given_sent[i * Q + j] * P_x;
    for (int j = 0; j < Q; j++) {
      const double t = P_y ? P_x * p_rec_given_sent[i * Q + j] / P_y : 0.0;
      logT[(size_t)i * Q + j] = t == 0 ? -100.0 : std::log(t);
    }
  }
  HIPCHK(hipSetDevice(mc->device));
  int rc;
  if ((rc = mc->d_conf.ensure((size_t)R * Q * 8)) || (rc = mc->d_logT.ensure((size_t)R * Q * 8))) return rc;
  HIPCHK(hipMemcpy(mc->d_conf.p, p_rec_given_sent, (size_t)R * Q * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(mc->d_logT.p, logT.data(), (size_t)R * Q * 8, hipMemcpyHostToDevice));
  mc->R = R;
  return LDPC_OK;
}

static int gfmc_generate_pass(ldpc_gfmc *mc, uint64_t seed, uint64_t first, int nf, bool inspect, hipStream_t st) {
  const ldpc_code &c = *mc->code;
  if (!mc->R) { ldpc_set_error("ldpc_gfmc: ldpc_gfmc_set_channel() first"); return LDPC_ERR_ARG; }
  int rc;
  if (inspect && ((rc = mc->d_y.ensure((size_t)mc->max_frames * c.n * 4)) || (rc = mc->d_sel.ensure((size_t)mc->max_frames * c.n * 4)))) return rc;
  GfmcArgs a;
  memset(&a, 0, sizeof a);
  a.n = c.n; a.m = c.m; a.Q = c.q; a.R = mc->R; a.nframes = nf;
  a.seed = seed; a.first_frame = first;
  a.conf = mc->d_conf.as<double>(); a.logT = mc->d_logT.as<double>();
  a.row_ptr = mc->d_row_ptr.as<int32_t>(); a.col_idx = mc->d_col_idx.as<int32_t>(); a.hval = mc->d_hval.as<int32_t>();
  a.mul = mc->d_mul.as<uint8_t>();
  a.pow2 = (c.q & (c.q - 1)) == 0;
  a.x = mc->d_x.as<uint8_t>();
  a.y = inspect ? mc->d_y.as<int32_t>() : nullptr;
  a.sel = inspect ? mc->d_sel.as<float>() : nullptr;
  a.logfna = mc->d_logfna.as<double>(); a.synd = mc->d_synd.as<uint8_t>();
  const size_t ts = (size_t)nf * c.n, tc = (size_t)nf * c.m;
  hipLaunchKernelGGL(gfmc_gen_kernel, dim3((unsigned)((ts + 255) / 256)), dim3(256), 0, st, a);
  LAUNCHCHK("gfmc_gen_kernel");
  hipLaunchKernelGGL(gfmc_synd_kernel, dim3((unsigned)((tc + 255) / 256)), dim3(256), 0, st, a);
  LAUNCHCHK("gfmc_synd_kernel");
  return LDPC_OK;
}

extern "C" int ldpc_gfmc_generate(ldpc_gfmc *mc, uint64_t seed, uint64_t first_frame, int nframes, uint8_t *x, int32_t *y,
                                  float *sel, double *logfna, uint8_t *synd) {
  if (!mc || nframes < 0 || nframes > mc->max_frames) return LDPC_ERR_ARG;
  if (nframes == 0) return LDPC_OK;
  HIPCHK(hipSetDevice(mc->device));
  int rc = gfmc_generate_pass(mc, seed, first_frame, nframes, true, nullptr);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  const size_t n = (size_t)mc->code->n, m = (size_t)mc->code->m, Q = (size_t)mc->code->q;
  if (x) HIPCHK(hipMemcpy(x, mc->d_x.p, nframes * n, hipMemcpyDeviceToHost));
  if (y) HIPCHK(hipMemcpy(y, mc->d_y.p, nframes * n * 4, hipMemcpyDeviceToHost));
  if (sel) HIPCHK(hipMemcpy(sel, mc->d_sel.p, nframes * n * 4, hipMemcpyDeviceToHost));
  if (logfna) HIPCHK(hipMemcpy(logfna, mc->d_logfna.p, nframes * n * Q * 8, hipMemcpyDeviceToHost));
  if (synd) HIPCHK(hipMemcpy(synd, mc->d_synd.p, nframes * m, hipMemcpyDeviceToHost));
  return LDPC_OK;
}

extern "C" int ldpc_gfmc_run(ldpc_gfmc *mc, int max_iter, int64_t nframes, uint64_t seed, uint64_t first_frame,
                             uint64_t *counters) {
  if (!mc || nframes < 0 || max_iter < 0 || !counters) return LDPC_ERR_ARG;
  HIPCHK(hipSetDevice(mc->device));
  HIPCHK(hipMemsetAsync(mc->d_counters.p, 0, 64, nullptr));
  int rc;
  for (int64_t done = 0; done < nframes; done += mc->max_frames) {
    const int nf = (int)std::min<int64_t>(mc->max_frames, nframes - done);
    if ((rc = gfmc_generate_pass(mc, seed, first_frame + (uint64_t)done, nf, false, nullptr))) return rc;
    rc = ldpc_decode_batch_device(mc->dec, nf, mc->d_logfna.p, LDPC_DT_F64, mc->d_synd.as<uint8_t>(), max_iter,
                                  mc->d_xhat.as<uint8_t>(), mc->d_iters.as<int32_t>(), nullptr, mc->d_status.as<int32_t>(),
                                  nullptr, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(gfmc_classify_kernel, dim3((unsigned)std::min(2048, (nf + 3) / 4)), dim3(256), 0, nullptr, mc->code->n, nf,
                       mc->d_x.as<uint8_t>(), mc->d_xhat.as<uint8_t>(), mc->d_status.as<int32_t>(), mc->d_iters.as<int32_t>(),
                       mc->d_counters.as<unsigned long long>());
    LAUNCHCHK("gfmc_classify_kernel");
  }
  HIPCHK(hipMemcpy(counters, mc->d_counters.p, 64, hipMemcpyDeviceToHost));
  return LDPC_OK;
}

// the decoded words / verdicts of the last pass of ldpc_gfmc_run (tests); host pointers, any may be null
extern "C" int ldpc_gfmc_last_pass(ldpc_gfmc *mc, int nframes, uint8_t *x, uint8_t *xhat, int32_t *status, int32_t *iters) {
  if (!mc || nframes < 0 || nframes > mc->max_frames) return LDPC_ERR_ARG;
  HIPCHK(hipSetDevice(mc->device));
  HIPCHK(hipDeviceSynchronize());
  const size_t n = (size_t)mc->code->n;
  if (x) HIPCHK(hipMemcpy(x, mc->d_x.p, nframes * n, hipMemcpyDeviceToHost));
  if (xhat) HIPCHK(hipMemcpy(xhat, mc->d_xhat.p, nframes * n, hipMemcpyDeviceToHost));
  if (status) HIPCHK(hipMemcpy(status, mc->d_status.p, (size_t)nframes * 4, hipMemcpyDeviceToHost));
  if (iters) HIPCHK(hipMemcpy(iters, mc->d_iters.p, (size_t)nframes * 4, hipMemcpyDeviceToHost));
  return LDPC_OK;
}
