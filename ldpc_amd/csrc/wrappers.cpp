// wrappers.cpp -- the reference-shaped entry points of libldpc_hip.
//
// Same names, argument order and meaning as the C harness functions of the
// reference (see include/ldpc_hip.h for the file:line of each).  They are thin
// drivers: the binary Monte-Carlo harnesses hand the whole trial loop to the
// device front-end (mc_frontend.hip); the single-frame harnesses and GFq_LDPC
// (confusion-matrix channel) prepare their frames here and every dec() call
// goes to the GPU through ldpc_decode_batch().  There is no CPU decoder behind
// them: without a HIP device they return LDPC_ERR_HIP.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ldpc_hip.h"
#include "engine.h"

namespace {

// ------------------------------------------------------------------- RNG
// The reference seeds rand() from time(NULL) at every entry
// (test_code_min_sum_B.c:289), so its Monte-Carlo streams are not reproducible
// by design; a splitmix64 stream with an optional fixed seed stands in.
std::mutex g_mu;
uint64_t g_state = 0;
bool g_seeded = false;

uint64_t clock_seed() {
  return (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull;
}

struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); }
  int bit() { return (int)(next() >> 63); }
};

Rng take_rng() {   // one independent stream per entry-point call
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_seeded) {
    g_state = clock_seed();
    g_seeded = true;
  }
  Rng r{g_state};
  g_state = r.next() ^ 0xD1B54A32D192ED03ull;
  return r;
}

// ------------------------------------------------------ decoder cache
// "The graph/decoder for code_filename is cached per process": the reference
// re-parses the alist and leaks the previous graph on every call.
struct Cached {
  std::string file;
  int alg;
  ldpc_code *code;
  ldpc_decoder *dec;
  ldpc_mc *mc;        // device Monte-Carlo context (binary codes), made on first use
};
std::vector<Cached> g_cache;

int mc_batch() {
  if (const char *e = getenv("LDPC_HIP_MC_BATCH")) {
    int v = atoi(e);
    if (v > 0) return v;
  }
  return 1024;
}

int cache_device() {
  if (const char *e = getenv("LDPC_HIP_DEVICE")) return atoi(e);
  return 0;
}

int get_decoder(const char *file, int alg, ldpc_code **code, ldpc_decoder **dec, ldpc_mc **mc = nullptr) {
  if (!file) return LDPC_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &c : g_cache)
    if (c.alg == alg && c.file == file) {
      *code = c.code;
      *dec = c.dec;
      if (mc) {
        if (!c.mc) {
          int rc = ldpc_mc_create(c.code, c.dec, cache_device(), &c.mc);
          if (rc) return rc;
        }
        *mc = c.mc;
      }
      return LDPC_OK;
    }
  ldpc_code *c = nullptr;
  int rc = ldpc_code_load_alist(file, &c);
  if (rc) return rc;
  ldpc_opts o;
  memset(&o, 0, sizeof o);
  o.alg = alg;
  o.device = cache_device();
  o.max_frames = mc_batch();
  o.early_exit = 1;
  ldpc_decoder *d = nullptr;
  rc = ldpc_decoder_create(c, &o, &d);
  if (rc) {
    ldpc_code_free(c);
    return rc;
  }
  ldpc_mc *m = nullptr;
  if (mc) {
    rc = ldpc_mc_create(c, d, o.device, &m);
    if (rc) {
      ldpc_decoder_free(d);
      ldpc_code_free(c);
      return rc;
    }
    *mc = m;
  }
  g_cache.push_back({file, alg, c, d, m});
  *code = c;
  *dec = d;
  return LDPC_OK;
}

// CSR copy of a code for the host-side enc()
struct HostGraph {
  int n, m, q;
  std::vector<int32_t> row_ptr, col_idx, hval;
  explicit HostGraph(const ldpc_code *c) {
    n = ldpc_code_n(c);
    m = ldpc_code_m(c);
    q = ldpc_code_q(c);
    row_ptr.resize(m + 1);
    col_idx.resize(ldpc_code_edges(c));
    hval.resize(ldpc_code_edges(c));
    ldpc_code_get_csr(c, row_ptr.data(), col_idx.data(), hval.data());
  }
  // enc(), test_code_min_sum_B.c:69-79
  void enc_bin(const uint8_t *x, uint8_t *s) const {
    for (int j = 0; j < m; j++) {
      unsigned k = 0;
      for (int e = row_ptr[j]; e < row_ptr[j + 1]; e++) k ^= x[col_idx[e]];
      s[j] = (uint8_t)(k & 1);
    }
  }
};

enum Prep { PREP_NONE, PREP_Q88, PREP_Q14 };

// The Monte-Carlo loop shared by the *_cw_noise_gen harnesses and test_code_B: the whole
// trial loop (source word, enc(), bsc(), dec(), classification) runs on the device
// (mc_frontend.hip); only the counters come back.
//   fresh_x: draw x per trial (mex harnesses) or once (test_code_no_mex_B.c:339-343)
int monte_carlo_bsc(const char *file, int alg, int iteration, int trials, double p_bsc, bool fresh_x,
                    Prep prep, double *errors) {
  if (!errors) return LDPC_ERR_ARG;
  errors[0] = errors[1] = 0;
  if (trials < 0 || iteration < 0) return LDPC_ERR_ARG;
  ldpc_code *code;
  ldpc_decoder *dec;
  ldpc_mc *mc = nullptr;
  int rc = get_decoder(file, alg, &code, &dec, &mc);
  if (rc) return rc;
  Rng r = take_rng();
  ldpc_mc_opts o;
  memset(&o, 0, sizeof o);
  o.channel = LDPC_CH_BSC_EXACT;
  o.source = fresh_x ? LDPC_SRC_RANDOM : LDPC_SRC_FIXED;
  o.llr_dtype = prep == PREP_Q88 ? LDPC_DT_I16 : prep == PREP_Q14 ? LDPC_DT_I32 : LDPC_DT_F64;
  o.max_iter = iteration;
  o.param = p_bsc;
  o.seed = r.next();
  uint64_t counters[8];
  rc = ldpc_mc_run(mc, &o, trials, counters);
  if (rc) return rc;
  errors[0] = (double)counters[1];
  errors[1] = (double)counters[2];
  return LDPC_OK;
}

// The single-frame harness shape: caller's x, s, y, q0 (test_code_min_sum_B.c:331-359).
int single_frame(const char *file, int alg, int iteration, const int *x, const int *s, const double *q0,
                 double *errors, const char *trace_file) {
  if (!errors) return LDPC_ERR_ARG;
  errors[0] = errors[1] = 0;
  if (!x || !s || !q0 || iteration < 0) return LDPC_ERR_ARG;
  ldpc_code *code;
  ldpc_decoder *dec;
  int rc = get_decoder(file, alg, &code, &dec);
  if (rc) return rc;
  const int n = ldpc_code_n(code), m = ldpc_code_m(code);
  std::vector<uint8_t> sy(m), bits(n);
  for (int j = 0; j < m; j++) sy[j] = (uint8_t)(s[j] & 1);
  int32_t status = 1;
  if (trace_file) {
    rc = ldpc_decode_trace_q88(dec, q0, LDPC_DT_F64, sy.data(), iteration, bits.data(), nullptr, nullptr,
                               &status, trace_file, 128);
  } else {
    rc = ldpc_decode_batch(dec, 1, q0, LDPC_DT_F64, sy.data(), iteration, bits.data(), nullptr, nullptr,
                           &status, nullptr);
  }
  if (rc) return rc;
  if (status) {
    errors[0]++;
  } else {
    for (int i = 0; i < n; i++)
      if ((int)bits[i] != x[i]) {   // HamDist(tmp_bit, x, n) != 0
        errors[1]++;
        break;
      }
  }
  return LDPC_OK;
}

}  // namespace

extern "C" void ldpc_set_seed(uint64_t seed) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_state = seed ? seed : clock_seed();
  g_seeded = true;
}

extern "C" int test_code_min_sum_B(const char *code_filename, int iteration, int *x, int *s, int *y,
                                   double *q0, double *errors) {
  (void)y;   // unused by the reference's dec() as well
  return single_frame(code_filename, LDPC_ALG_MINSUM_F64, iteration, x, s, q0, errors, nullptr);
}

extern "C" int test_code_min_sum_B_cw_noise_gen(const char *code_filename, int iteration, int trials,
                                                double p_bsc, double *errors) {
  return monte_carlo_bsc(code_filename, LDPC_ALG_MINSUM_F64, iteration, trials, p_bsc, true, PREP_NONE,
                         errors);
}

extern "C" int test_code_min_sum_B_fixed(const char *code_filename, int iteration, int *x, int *s, int *y,
                                         double *q0, double *errors, const char *Verilog_sim_filename) {
  (void)y;
  return single_frame(code_filename, LDPC_ALG_MINSUM_Q88, iteration, x, s, q0, errors, Verilog_sim_filename);
}

extern "C" int test_code_min_sum_B_fixed_cw_noise_gen(const char *code_filename, int iteration, int trials,
                                                      double p_bsc, double *errors) {
  return monte_carlo_bsc(code_filename, LDPC_ALG_MINSUM_Q88, iteration, trials, p_bsc, true, PREP_Q88,
                         errors);
}

extern "C" int test_code_B_float(const char *code_filename, int iteration, int *x, int *s, int *y,
                                 double *q0, double *errors) {
  (void)y;
  return single_frame(code_filename, LDPC_ALG_SPA_F64, iteration, x, s, q0, errors, nullptr);
}

extern "C" int test_code_B_float_cw_noise_gen(const char *code_filename, int iteration, int trials,
                                              double p_bsc, double *errors) {
  return monte_carlo_bsc(code_filename, LDPC_ALG_SPA_F64, iteration, trials, p_bsc, true, PREP_NONE, errors);
}

extern "C" int test_code_B(int iteration, int trials, double p_bsc, const char *H_file, double *errors) {
  return monte_carlo_bsc(H_file, LDPC_ALG_SPA_TABLE_Q14, iteration, trials, p_bsc, false, PREP_Q14, errors);
}

// test_code_B_MSDP(), test_code_B_MSDP.c:442-543
extern "C" int test_code_B_MSDP(const char *code_filename, int iteration, int num_trials, int num_reads,
                                int decode_mode, double *p_rec_given_sent, double *errors) {
  if (!errors) return LDPC_ERR_ARG;
  errors[0] = errors[1] = 0;
  if (!p_rec_given_sent || (num_reads != 1 && num_reads != 3) || num_trials < 0 || iteration < 0) return LDPC_ERR_ARG;
  if (decode_mode != 0) {
    ldpc_set_error("test_code_B_MSDP: decode_mode 1 (snowbird_sym.bin page file) is not supported");
    return LDPC_ERR_UNSUPPORTED;
  }
  const int Q = 16, R = num_reads == 1 ? Q : Q * num_reads - 2;
  for (int i = 0; i < Q; i++) {                      // column normalisation in place, :455-465
    double norm = 0;
    for (int j = 0; j < R; j++) norm += p_rec_given_sent[j * Q + i];
    for (int j = 0; j < R; j++) p_rec_given_sent[j * Q + i] /= norm;
  }
  ldpc_code *code;
  ldpc_decoder *dec;
  ldpc_mc *mc = nullptr;
  int rc = get_decoder(code_filename, LDPC_ALG_SPA_F64, &code, &dec, &mc);
  if (rc) return rc;
  if ((rc = ldpc_mc_set_qlc_channel(mc, p_rec_given_sent, num_reads))) return rc;
  Rng r = take_rng();
  ldpc_mc_opts o;
  memset(&o, 0, sizeof o);
  o.channel = LDPC_CH_QLC;
  o.source = LDPC_SRC_RANDOM;
  o.llr_dtype = LDPC_DT_F64;
  o.max_iter = iteration;
  o.seed = r.next();
  uint64_t counters[8];
  if ((rc = ldpc_mc_run(mc, &o, num_trials, counters))) return rc;
  errors[0] = (double)counters[1];
  errors[1] = (double)counters[2];
  return LDPC_OK;
}

// GFq_LDPC(), GFq_LDPC.c:540-655
extern "C" int GFq_LDPC(const char *code_filename, int iteration, int num_trials, int num_strs_seek,
                        int num_reads, int decode_mode, double *p_rec_given_sent, double *errors) {
  (void)num_strs_seek;   // only used by a commented-out fseek in the reference (:587)
  if (!errors) return LDPC_ERR_ARG;
  errors[0] = errors[1] = 0;
  if (!p_rec_given_sent || num_reads <= 0 || num_trials < 0 || iteration < 0) return LDPC_ERR_ARG;
  if (decode_mode != 0) {
    // decode_mode 1 replays reads.bin / written_data_sym_13.bin (:584-585), which the
    // reference does not ship (.MISSING_LARGE_BLOBS)
    ldpc_set_error("GFq_LDPC: decode_mode 1 (page files) is not supported");
    return LDPC_ERR_UNSUPPORTED;
  }
  ldpc_code *code;
  ldpc_decoder *dec;
  int rc = get_decoder(code_filename, LDPC_ALG_GFQ_DIRECT, &code, &dec);
  if (rc) return rc;
  HostGraph g(code);
  const int Q = g.q, n = g.n, m = g.m, R = Q * num_reads;

  // column normalisation in place, :548-558
  for (int i = 0; i < Q; i++) {
    double norm = 0;
    for (int j = 0; j < R; j++) norm += p_rec_given_sent[j * Q + i];
    for (int j = 0; j < R; j++) p_rec_given_sent[j * Q + i] /= norm;
  }
  // make_p_sent_given_rec_T(), :177-197, kept as log values with the -100 floor of :204-206
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

  Rng r = take_rng();
  const int B = ldpc_decoder_max_frames(dec);
  std::vector<int32_t> x((size_t)B * n), sx(m), status(B);
  std::vector<uint8_t> s((size_t)B * m), xhat((size_t)B * n);
  std::vector<double> logfna((size_t)B * n * Q);
  for (int t0 = 0; t0 < num_trials; t0 += B) {
    const int nf = num_trials - t0 < B ? num_trials - t0 : B;
    for (int f = 0; f < nf; f++) {
      int32_t *xf = &x[(size_t)f * n];
      for (int i = 0; i < n; i++) xf[i] = (int32_t)r.below((uint32_t)Q);
      ldpc_code_syndrome(code, xf, sx.data());
      for (int j = 0; j < m; j++) s[(size_t)f * m + j] = (uint8_t)sx[j];
      // channel(), :213-248
      for (int i = 0; i < n; i++) {
        double temp = 0;
        int rec = 0;
        const float sel = (float)(r.next() >> 40) * 100.0f / (float)((1u << 24) - 1);
        if (!sel)
          while (rec < R && !p_rec_given_sent[(rec++) * Q + xf[i]]) {}
        while (temp < sel && rec != R) temp += 100 * p_rec_given_sent[(rec++) * Q + xf[i]];
        const int yi = rec > 0 ? rec - 1 : 0;
        memcpy(&logfna[((size_t)f * n + i) * Q], &logT[(size_t)yi * Q], sizeof(double) * Q);
      }
    }
    rc = ldpc_decode_batch(dec, nf, logfna.data(), LDPC_DT_F64, s.data(), iteration, xhat.data(), nullptr,
                           nullptr, status.data(), nullptr);
    if (rc) return rc;
    for (int f = 0; f < nf; f++) {
      if (status[f]) {
        errors[0]++;
      } else {
        for (int i = 0; i < n; i++)
          if ((int32_t)xhat[(size_t)f * n + i] != x[(size_t)f * n + i]) {
            errors[1]++;
            break;
          }
      }
    }
  }
  return LDPC_OK;
}
