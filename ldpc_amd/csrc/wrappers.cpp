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
#include <thread>
#include <vector>
#include <algorithm>

#include "../../include/ldpc_hip.h"
#include "engine.h"
#include "mc_rng.h"

namespace {

// ------------------------------------------------------------------- RNG
// The reference seeds rand() from time(NULL) at every entry
// (test_code_min_sum_B.c:289), so its Monte-Carlo streams are not reproducible
// by design; a splitmix64 stream with an optional fixed seed stands in.
std::mutex g_mu;
// The reference-shaped entry points share cached decoders (one per code file and algorithm), and a
// decoder is not re-entrant: calls are serialised per process, like the reference's own
// file-scope-global decoders, which are not re-entrant either.
std::recursive_mutex g_call_mu;
#define SERIALISED std::lock_guard<std::recursive_mutex> call_lock_(g_call_mu)
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
    // LDPC_HIP_SEED: what ldpc_set_seed() does, for callers that cannot call it (the C drivers)
    const char *e = getenv("LDPC_HIP_SEED");
    const uint64_t v = e ? strtoull(e, nullptr, 0) : 0;
    g_state = v ? v : clock_seed();
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
  int frames;         // pass size the decoder was made for (part of the key: the single-frame harnesses keep a small one)
  ldpc_code *code;
  ldpc_decoder *dec;
  ldpc_mc *mc;        // device Monte-Carlo context (binary codes), made on first use
  uint64_t used;      // tick of the last call that asked for it
  int slot = 0;       // position in the device list (LDPC_HIP_DEVICES): one decoder per entry of the list
  int device = 0;
  int busy = 0;       // held by a worker thread of a multi-device run: not to be evicted
};
std::vector<Cached> g_cache;
uint64_t g_cache_tick = 0;
// A cached decoder holds its workspace on the device (gigabytes for a Monte-Carlo pass of a two-pass
// engine); a sweep over many codes and algorithms must not grow without bound.  At most
// LDPC_HIP_CACHE_MAX (default 6) decoders stay; the least recently used goes first.  The entry
// points are serialised (SERIALISED), so nothing but the caller's own decoder is in use.
std::vector<int> device_list();
size_t cache_max() {
  if (const char *e = getenv("LDPC_HIP_CACHE_MAX")) { const int v = atoi(e); if (v > 0) return (size_t)v; }
  return std::max<size_t>(6, 2 * device_list().size());     // (a multi-device run keeps one decoder per entry of its list)
}
void evict_lru_locked() {
  while (g_cache.size() >= cache_max()) {
    size_t old = g_cache.size();
    for (size_t i = 0; i < g_cache.size(); i++)
      if (!g_cache[i].busy && (old == g_cache.size() || g_cache[i].used < g_cache[old].used)) old = i;
    if (old == g_cache.size()) return;     // (everything is in use by a multi-device run)
    if (g_cache[old].mc) ldpc_mc_free(g_cache[old].mc);
    ldpc_decoder_free(g_cache[old].dec);
    ldpc_code_free(g_cache[old].code);
    g_cache.erase(g_cache.begin() + (long)old);
  }
}

// Frames per device pass of the Monte-Carlo harnesses: LDPC_HIP_MC_BATCH, else about 4 GB of
// 8-byte edge messages' worth, between 1024 and 16384 frames in steps of 512 (binary codes; a
// non-binary message is q values: 1024) -- large enough that short codes fill the GPU and that the
// two-pass decoders' two-stage early exit (BinaryEngine::decode) has a batch to work with.
int mc_batch(const ldpc_code *c = nullptr) {
  if (const char *e = getenv("LDPC_HIP_MC_BATCH")) {
    int v = atoi(e);
    if (v > 0) return v;
  }
  if (!c || ldpc_code_q(c) != 2) return 1024;
  const long f = (long)(4.0e9 / (8.0 * std::max(1, ldpc_code_edges(c))));
  return (int)std::min<long>(16384, std::max<long>(1024, f / 512 * 512));
}

int cache_device() {
  if (const char *e = getenv("LDPC_HIP_DEVICE")) return atoi(e);
  return 0;
}

// The GPUs a Monte-Carlo harness call spreads its trials over: LDPC_HIP_DEVICES = "all" or a comma-separated list
// (one worker thread, decoder and generator per entry; an entry may repeat); unset: the one device of
// LDPC_HIP_DEVICE.  The reference runs one dec() at a time on one core; its MATLAB drivers get their parallelism
// from parfor workers (test_mex_B_float.m:24-30) -- this is the same idea inside one call, for callers that have
// one process and eight GPUs.  Frames are keyed by their global index, so the counters do not depend on the list.
std::vector<int> device_list() {
  std::vector<int> v;
  if (const char *e = getenv("LDPC_HIP_DEVICES")) {
    if (!strcmp(e, "all")) {
      int n = 0;
      if (hipGetDeviceCount(&n) == hipSuccess)
        for (int i = 0; i < n; i++) v.push_back(i);
    } else {
      for (const char *p = e; *p;) {
        char *end;
        const long d = strtol(p, &end, 10);
        if (end == p) break;
        v.push_back((int)d);
        p = *end == ',' ? end + 1 : end;
      }
    }
  }
  if (v.empty()) v.push_back(cache_device());
  return v;
}

// frames = 0: a Monte-Carlo pass (mc_batch); > 0: a decoder for that many frames per call (the
// single-frame harnesses ask for 64, so that one frame does not reserve a Monte-Carlo pass's workspace)
// slot / device: the entry of the device list (hold = true: the caller is a worker thread of a multi-device run and
// gives the entry back with release_decoder())
int get_decoder(const char *file, int alg, ldpc_code **code, ldpc_decoder **dec, ldpc_mc **mc = nullptr, int frames = 0,
                int slot = 0, int device = -1, bool hold = false) {
  if (!file) return LDPC_ERR_ARG;
  if (device < 0) device = cache_device();
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &c : g_cache)
    if (c.alg == alg && c.frames == frames && c.slot == slot && c.device == device && c.file == file) {
      c.used = ++g_cache_tick;
      *code = c.code;
      *dec = c.dec;
      if (mc) {
        if (!c.mc) {
          int rc = ldpc_mc_create(c.code, c.dec, device, &c.mc);
          if (rc) return rc;
        }
        *mc = c.mc;
      }
      if (hold) c.busy++;
      return LDPC_OK;
    }
  evict_lru_locked();
  ldpc_code *c = nullptr;
  int rc = ldpc_code_load_alist(file, &c);
  if (rc) return rc;
  ldpc_opts o;
  memset(&o, 0, sizeof o);
  o.alg = alg;
  o.device = device;
  o.max_frames = frames > 0 ? frames : mc_batch(c);
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
  Cached entry{file, alg, frames, c, d, m, ++g_cache_tick};
  entry.slot = slot; entry.device = device; entry.busy = hold ? 1 : 0;
  g_cache.push_back(entry);
  *code = c;
  *dec = d;
  return LDPC_OK;
}

void release_decoder(const ldpc_decoder *dec) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &c : g_cache)
    if (c.dec == dec && c.busy > 0) c.busy--;
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
  Rng r = take_rng();
  ldpc_mc_opts o;
  memset(&o, 0, sizeof o);
  o.channel = LDPC_CH_BSC_EXACT;
  o.source = fresh_x ? LDPC_SRC_RANDOM : LDPC_SRC_FIXED;
  o.llr_dtype = prep == PREP_Q88 ? LDPC_DT_I16 : prep == PREP_Q14 ? LDPC_DT_I32 : LDPC_DT_F64;
  o.max_iter = iteration;
  o.param = p_bsc;
  o.seed = r.next();
  const std::vector<int> devs = device_list();
  const int nd = (int)std::min<size_t>(devs.size(), (size_t)std::max(1, trials));
  // trials in contiguous shares, one worker thread per entry of the device list (one entry: this thread)
  std::vector<uint64_t> counters((size_t)nd * 8, 0);
  std::vector<int> rcs((size_t)nd, LDPC_OK);
  std::vector<std::string> msgs((size_t)nd);
  auto work = [&](int i) {
    const int64_t lo = (int64_t)trials * i / nd, hi = (int64_t)trials * (i + 1) / nd;
    ldpc_code *code;
    ldpc_decoder *dec;
    ldpc_mc *mc = nullptr;
    int rc = get_decoder(file, alg, &code, &dec, &mc, 0, i, devs[(size_t)i], nd > 1);
    if (rc == LDPC_OK) {
      ldpc_mc_opts oi = o;
      oi.first_frame = (uint64_t)lo;
      rc = ldpc_mc_run(mc, &oi, hi - lo, &counters[(size_t)i * 8]);
      if (nd > 1) release_decoder(dec);
    }
    rcs[(size_t)i] = rc;
    if (rc) msgs[(size_t)i] = ldpc_last_error();     // (thread-local in the library: carried back to the caller's thread)
  };
  if (nd == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int i = 0; i < nd; i++) th.emplace_back(work, i);
    for (auto &t : th) t.join();
  }
  for (int i = 0; i < nd; i++)
    if (rcs[(size_t)i]) {
      ldpc_set_error(msgs[(size_t)i]);
      return rcs[(size_t)i];
    }
  for (int i = 0; i < nd; i++) {
    errors[0] += (double)counters[(size_t)i * 8 + 1];
    errors[1] += (double)counters[(size_t)i * 8 + 2];
  }
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
  int rc = get_decoder(file, alg, &code, &dec, nullptr, 64);
  if (rc) return rc;
  const int n = ldpc_code_n(code), m = ldpc_code_m(code);
  if (alg == LDPC_ALG_MINSUM_Q88) {
    // The reference hands q0 to dec() untouched (test_code_min_sum_B_fixed.c:442-459): its sums
    // (:214-220, :299-303), the iteration-0 test (:199-207) and the trace (:250-256) see the raw
    // doubles, only the v2c magnitudes go through float_to_fix.  The int16 decoder is that
    // arithmetic exactly when q0 is on the Q8.8 grid and inside float_to_fix's range -- what the
    // reference's own bsc() (:117-120) and Verilog tester (11/256) produce.  Anything else would
    // be quantised here and could decode differently (q0 = -0.001: bit 1 there, 0 here): refused.
    for (int i = 0; i < n; i++) {
      const double r = q0[i] * 256.0;
      if (!(r == std::nearbyint(r)) || q0[i] > 127.0 || q0[i] < -128.0) {
        ldpc_set_error("test_code_min_sum_B_fixed: q0[" + std::to_string(i) + "] = " + std::to_string(q0[i]) +
                       " is not a Q8.8 value (k/256 in [-128, 127]); the fixed-point decoder takes on-grid "
                       "LLRs only (put them through float_to_fix first, as the reference's bsc() does)");
        return LDPC_ERR_UNSUPPORTED;
      }
    }
  }
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

extern "C" int64_t ldpc_wht_tables_selfcheck(int64_t *stats) {
  long st[4] = {0, 0, 0, 0};
  const long bad = ldpc_wht_compact_selfcheck(st);
  if (stats) for (int i = 0; i < 4; i++) stats[i] = st[i];
  return bad;
}

extern "C" int64_t ldpc_spa_tables_selfcheck(int64_t *stats) {
  long st[4] = {0, 0, 0, 0};
  const long bad = ldpc_spa_compact_selfcheck(st);
  if (stats) for (int i = 0; i < 4; i++) stats[i] = st[i];
  return bad;
}

extern "C" void ldpc_set_seed(uint64_t seed) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_state = seed ? seed : clock_seed();
  g_seeded = true;
}

extern "C" int test_code_min_sum_B(const char *code_filename, int iteration, int *x, int *s, int *y,
                                   double *q0, double *errors) {
  SERIALISED;
  (void)y;   // unused by the reference's dec() as well
  return single_frame(code_filename, LDPC_ALG_MINSUM_F64, iteration, x, s, q0, errors, nullptr);
}

extern "C" int test_code_min_sum_B_cw_noise_gen(const char *code_filename, int iteration, int trials,
                                                double p_bsc, double *errors) {
  SERIALISED;
  return monte_carlo_bsc(code_filename, LDPC_ALG_MINSUM_F64, iteration, trials, p_bsc, true, PREP_NONE,
                         errors);
}

extern "C" int test_code_min_sum_B_fixed(const char *code_filename, int iteration, int *x, int *s, int *y,
                                         double *q0, double *errors, const char *Verilog_sim_filename) {
  SERIALISED;
  (void)y;
  return single_frame(code_filename, LDPC_ALG_MINSUM_Q88, iteration, x, s, q0, errors, Verilog_sim_filename);
}

extern "C" int test_code_min_sum_B_fixed_cw_noise_gen(const char *code_filename, int iteration, int trials,
                                                      double p_bsc, double *errors) {
  SERIALISED;
  return monte_carlo_bsc(code_filename, LDPC_ALG_MINSUM_Q88, iteration, trials, p_bsc, true, PREP_Q88,
                         errors);
}

extern "C" int test_code_B_float(const char *code_filename, int iteration, int *x, int *s, int *y,
                                 double *q0, double *errors) {
  SERIALISED;
  (void)y;
  return single_frame(code_filename, LDPC_ALG_SPA_F64, iteration, x, s, q0, errors, nullptr);
}

extern "C" int test_code_B_float_cw_noise_gen(const char *code_filename, int iteration, int trials,
                                              double p_bsc, double *errors) {
  SERIALISED;
  return monte_carlo_bsc(code_filename, LDPC_ALG_SPA_F64, iteration, trials, p_bsc, true, PREP_NONE, errors);
}

extern "C" int test_code_B(int iteration, int trials, double p_bsc, const char *H_file, double *errors) {
  SERIALISED;
  return monte_carlo_bsc(H_file, LDPC_ALG_SPA_TABLE_Q14, iteration, trials, p_bsc, false, PREP_Q14, errors);
}

// test_code_B_MSDP(), test_code_B_MSDP.c:442-543
extern "C" int test_code_B_MSDP(const char *code_filename, int iteration, int num_trials, int num_reads,
                                int decode_mode, double *p_rec_given_sent, double *errors) {
  SERIALISED;
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

// The channel front-end of GFq_LDPC() (GFq_LDPC.c:177-248), host side: the tables once per
// call, then one frame at a time.
struct GfqChannel {
  int Q = 0, R = 0;
  const double *p = nullptr;            // p_rec_given_sent [R][Q], column-normalised
  std::vector<double> logT;             // log p_sent_given_rec_T [R][Q], floor -100 (:204-206)

  // column normalisation in place (:548-558), then make_p_sent_given_rec_T() (:177-197)
  void prepare(int q, int num_reads, double *p_rec_given_sent) {
    Q = q; R = q * num_reads; p = p_rec_given_sent;
    for (int i = 0; i < Q; i++) {
      double norm = 0;
      for (int j = 0; j < R; j++) norm += p_rec_given_sent[j * Q + i];
      for (int j = 0; j < R; j++) p_rec_given_sent[j * Q + i] /= norm;
    }
    logT.assign((size_t)R * Q, 0.0);
    const double P_x = 1 / (1.0 * Q);
    for (int i = 0; i < R; i++) {
      double P_y = 0;
      for (int j = 0; j < Q; j++) P_y += p_rec_given_sent[i * Q + j] * P_x;
      for (int j = 0; j < Q; j++) {
        const double t = P_y ? P_x * p_rec_given_sent[i * Q + j] / P_y : 0.0;
        logT[(size_t)i * Q + j] = t == 0 ? -100.0 : std::log(t);
      }
    }
  }
  // channel() (:213-248) for one symbol: the read y drawn by walking the column of x with
  // the percent-granular draw `sel` (the reference: rand() * 100 / RAND_MAX as floats)
  int draw(int xi, float sel) const {
    double temp = 0;
    int rec = 0;
    if (!sel)
      while (rec < R && !p[(rec++) * Q + xi]) {}
    while (temp < sel && rec != R) temp += 100 * p[(rec++) * Q + xi];
    return rec > 0 ? rec - 1 : 0;
  }
  // one frame: x[i] = rand() % Q (:626), y, logfna[i][*] (frame-major rows of Q); sel_out optional.
  // The draws are the device front-end's (mc_frontend.hip gfmc_gen_kernel): counter-based, keyed by
  // (seed, global frame index), so this host restatement names the very frames ldpc_gfmc_run() decodes.
  void frame(uint64_t seed, uint64_t index, int n, int32_t *xf, int32_t *yf, double *lf, float *sel_out) const {
    const uint64_t k0 = ldpc_rng::stream_key(seed, index, 0), k1 = ldpc_rng::stream_key(seed, index, 1);
    for (int i = 0; i < n; i++) xf[i] = (int32_t)ldpc_rng::below(ldpc_rng::draw(k0, (uint64_t)i), (uint32_t)Q);
    for (int i = 0; i < n; i++) {
      const float sel = ldpc_rng::percent(ldpc_rng::draw(k1, (uint64_t)i));
      const int yi = draw(xf[i], sel);
      if (yf) yf[i] = yi;
      if (sel_out) sel_out[i] = sel;
      memcpy(&lf[(size_t)i * Q], &logT[(size_t)yi * Q], sizeof(double) * Q);
    }
  }
};

// The trial loop of GFq_LDPC(), GFq_LDPC.c:540-655 (same in WD_farm/GFq_LDPC.c:477-633), on a
// given graph and decoder.
static int gfq_trials(ldpc_code *code, ldpc_decoder *dec, int iteration, int num_trials, int num_reads,
                      int decode_mode, double *p_rec_given_sent, double *errors) {
  if (!errors) return LDPC_ERR_ARG;
  errors[0] = errors[1] = 0;
  if (!p_rec_given_sent || num_reads <= 0 || num_trials < 0 || iteration < 0) return LDPC_ERR_ARG;
  if (decode_mode != 0) {
    // decode_mode 1 replays reads.bin / written_data_sym_13.bin (:584-585), which the
    // reference does not ship (.MISSING_LARGE_BLOBS)
    ldpc_set_error("GFq_LDPC: decode_mode 1 (page files) is not supported");
    return LDPC_ERR_UNSUPPORTED;
  }
  // column normalisation in place (:548-558); the whole trial loop -- source symbols, enc(), channel(),
  // dec(), classification -- then runs on the device (mc_frontend.hip, ldpc_gfmc_*): eight counters come back
  GfqChannel ch;
  ch.prepare(ldpc_code_q(code), num_reads, p_rec_given_sent);
  Rng r = take_rng();
  ldpc_gfmc *mc = nullptr;
  int rc = ldpc_gfmc_create(code, dec, cache_device(), &mc);
  if (rc) return rc;
  uint64_t counters[8] = {0};
  rc = ldpc_gfmc_set_channel(mc, p_rec_given_sent, num_reads);
  if (!rc) rc = ldpc_gfmc_run(mc, iteration, num_trials, r.next(), 0, counters);
  ldpc_gfmc_free(mc);
  if (rc) return rc;
  errors[0] = (double)counters[1];
  errors[1] = (double)counters[2];
  return LDPC_OK;
}

// The inputs GFq_LDPC() would decode for a given seed (inspection; host only, needs no GPU)
extern "C" int GFq_LDPC_experiment(const char *code_filename, int num_reads, double *p_rec_given_sent,
                                   uint64_t seed, int nframes, int32_t *x, int32_t *s, int32_t *y,
                                   double *logfna, float *rand_select) {
  if (!code_filename || !p_rec_given_sent || num_reads <= 0 || nframes < 0 || !x || !logfna) return LDPC_ERR_ARG;
  ldpc_code *code = nullptr;
  int rc = ldpc_code_load_alist(code_filename, &code);
  if (rc) return rc;
  const int Q = ldpc_code_q(code), n = ldpc_code_n(code), m = ldpc_code_m(code);
  if (Q < 3) {
    ldpc_code_free(code);
    ldpc_set_error("GFq_LDPC_experiment: not a non-binary alist");
    return LDPC_ERR_UNSUPPORTED;
  }
  GfqChannel ch;
  ch.prepare(Q, num_reads, p_rec_given_sent);
  for (int f = 0; f < nframes; f++) {
    ch.frame(seed, (uint64_t)f, n, x + (size_t)f * n, y ? y + (size_t)f * n : nullptr, logfna + (size_t)f * n * Q,
             rand_select ? rand_select + (size_t)f * n : nullptr);
    if (s) ldpc_code_syndrome(code, x + (size_t)f * n, s + (size_t)f * m);
  }
  ldpc_code_free(code);
  return LDPC_OK;
}

// GFq_LDPC(), GFq_LDPC.c:540-655
extern "C" int GFq_LDPC(const char *code_filename, int iteration, int num_trials, int num_strs_seek,
                        int num_reads, int decode_mode, double *p_rec_given_sent, double *errors) {
  SERIALISED;
  (void)num_strs_seek;   // only used by a commented-out fseek in the reference (:587)
  if (errors) errors[0] = errors[1] = 0;
  ldpc_code *code;
  ldpc_decoder *dec;
  int rc = get_decoder(code_filename, LDPC_ALG_GFQ_DIRECT, &code, &dec);
  if (rc) return rc;
  return gfq_trials(code, dec, iteration, num_trials, num_reads, decode_mode, p_rec_given_sent, errors);
}

// GFq_LDPC() of WD_farm/GFq_LDPC.c:477: the graph comes in as the flat arrays that
// InitSumProductDecoderSyndrome.m builds instead of a file name.
extern "C" int GFq_LDPC_graph(int q, int iteration, int num_trials, int num_strs_seek, int num_reads,
                              int decode_mode, double *p_rec_given_sent, int nvar, int nchk, int cmax_in,
                              int rmax_in, const int *col_weight_in, const int *row_weight_in,
                              const int *row_list_col_in, const int *col_list_row_in, const int *col_list_N_in,
                              const int *H_in, double *errors) {
  SERIALISED;
  (void)num_strs_seek;
  if (errors) errors[0] = errors[1] = 0;
  if (q < 2 || nvar <= 0 || nchk <= 0 || cmax_in <= 0 || rmax_in <= 0 || !row_weight_in || !row_list_col_in || !H_in)
    return LDPC_ERR_ARG;
  std::vector<int32_t> row_ptr((size_t)nchk + 1, 0), col_idx, hval, colw((size_t)nvar, 0);
  for (int j = 0; j < nchk; j++) {
    const int w = row_weight_in[j];
    if (w < 0 || w > rmax_in) {
      ldpc_set_error("GFq_LDPC_graph: row_weight[" + std::to_string(j) + "] = " + std::to_string(w) + " outside 0..rmax");
      return LDPC_ERR_ARG;
    }
    row_ptr[j + 1] = row_ptr[j] + w;
    for (int k = 0; k < w; k++) {
      const int v = row_list_col_in[(size_t)j * rmax_in + k], h = H_in[(size_t)j * rmax_in + k];
      if (v < 0 || v >= nvar || h <= 0 || h >= q) {
        ldpc_set_error("GFq_LDPC_graph: row " + std::to_string(j) + " entry " + std::to_string(k) + ": column " + std::to_string(v) + " / value " + std::to_string(h) + " out of range");
        return LDPC_ERR_ARG;
      }
      // the view from the variable's side must be the one the rows imply (slot = order of
      // appearance over ascending rows, InitSumProductDecoderSyndrome.m / GFq_LDPC.c:505-515)
      if (col_list_row_in && col_list_N_in) {
        const int slot = colw[v];
        if (slot >= cmax_in || col_list_row_in[(size_t)v * cmax_in + slot] != j ||
            col_list_N_in[(size_t)v * cmax_in + slot] != k) {
          ldpc_set_error("GFq_LDPC_graph: col_list_row / col_list_N disagree with the rows at variable " + std::to_string(v) + " slot " + std::to_string(slot));
          return LDPC_ERR_ARG;
        }
      }
      colw[v]++;
      col_idx.push_back(v);
      hval.push_back(h);
    }
  }
  if (col_weight_in)
    for (int v = 0; v < nvar; v++)
      if (col_weight_in[v] != colw[v]) {
        ldpc_set_error("GFq_LDPC_graph: col_weight[" + std::to_string(v) + "] = " + std::to_string(col_weight_in[v]) + " but the rows hold it " + std::to_string(colw[v]) + " times");
        return LDPC_ERR_ARG;
      }
  ldpc_code *code = nullptr;
  int rc = ldpc_code_from_csr(nvar, nchk, q, row_ptr.data(), col_idx.data(), hval.data(), &code);
  if (rc) return rc;
  ldpc_opts o;
  memset(&o, 0, sizeof o);
  o.alg = LDPC_ALG_GFQ_DIRECT;
  o.device = cache_device();
  o.max_frames = mc_batch();
  o.early_exit = 1;
  ldpc_decoder *dec = nullptr;
  rc = ldpc_decoder_create(code, &o, &dec);
  if (!rc) {
    rc = gfq_trials(code, dec, iteration, num_trials, num_reads, decode_mode, p_rec_given_sent, errors);
    ldpc_decoder_free(dec);
  }
  ldpc_code_free(code);
  return rc;
}

// ---------------------------------------------------------------------------
// GFq_LDPC_NTT.c's program: main() :641-703 runs three experiments, SRand(j) for
// j = 1..3, on a 31-bit LCG (:155-167) that is the same on every platform, so
// unlike the rand()-seeded harnesses above its inputs can be regenerated exactly.
namespace {

struct Lcg31 {   // Rand(), GFq_LDPC_NTT.c:163-167
  uint32_t r;
  uint32_t next() { return r = (77u * r + 1243u) & 0x7fffffffu; }
};

// the program's own log2 (:153, :194): a quotient of natural logarithms, not libm's log2()
inline double ntt_log2(double v) { return std::log(v) / std::log(2.0); }

// float2fix(), GFq_LDPC_NTT.c:177-184, DECI = 14
inline int32_t ntt_fix(double v) {
  return v >= 0 ? (int32_t)(v * 16384.0 + 0.5) : -(int32_t)((-v) * 16384.0 + 0.5);
}

// bsc(), GFq_LDPC_NTT.c:278-315: exactly round(p * n * log2 q) distinct bit flips
void ntt_bsc(Lcg31 &g, int n, int Q, int log2q, const int32_t *x, int32_t *y, double p, int32_t *logfna) {
  const int len = log2q * n;
  int todo = (int)(len * p + 0.5);
  p = todo / (double)len;
  std::vector<int32_t> err(n, 0);
  while (todo) {
    const int i = (int)(g.next() % (uint32_t)len);
    const int bit = 1 << (i % log2q);
    if (err[i / log2q] & bit) continue;
    err[i / log2q] |= bit;
    todo--;
  }
  for (int i = 0; i < n; i++) y[i] = x[i] ^ err[i];
  const double lp = ntt_log2(p), l1p = ntt_log2(1 - p);
  for (int i = 0; i < n; i++)
    for (int a = 0; a < Q; a++) {
      double logprod = 0;
      for (int j = 1; j < Q; j <<= 1) logprod += ((a & j) == (y[i] & j)) ? l1p : lp;
      logfna[(size_t)i * Q + a] = ntt_fix(logprod);
    }
}

// lap(), GFq_LDPC_NTT.c:320-356 (its statistics printout, :357-364, is not reproduced)
void ntt_lap(Lcg31 &g, int n, int Q, const int32_t *x, int32_t *y, double stddev, int32_t *logfna) {
  std::vector<double> lf(Q);
  for (int i = 0; i < n; i++) {
    const double u2 = (g.next() + 1) * (1.0 / (0x7fffffffUL + 1.0));
    const int logu2 = (int)std::floor(ntt_log2(u2) * stddev + .5);
    y[i] = (g.next() & 1) == 0 ? x[i] - logu2 : x[i] + logu2;
    if (y[i] < 0) y[i] = 0;
    else if (y[i] >= Q) y[i] = Q - 1;
    double sum = 0;
    for (int a = 0; a < Q; a++) {
      lf[a] = y[i] == a ? -0.5 / stddev : -(std::abs(y[i] - a) - 0.0413) / stddev;
      sum += std::pow(2.0, lf[a]);
    }
    sum = ntt_log2(sum);
    for (int a = 0; a < Q; a++) logfna[(size_t)i * Q + a] = ntt_fix(lf[a] - sum);
  }
}

int ntt_channel_id(const char *channel) {
  if (channel && strcmp(channel, "bsc") == 0) return 0;
  if (channel && strcmp(channel, "lap") == 0) return 1;
  ldpc_set_error("GFq_LDPC_NTT: please specify 'lap' or 'bsc'");   // :666
  return -1;
}

int ntt_experiment(const ldpc_code *code, int seed, int ch, double level, int32_t *x, int32_t *s, int32_t *y,
                   int32_t *logfna) {
  const int n = ldpc_code_n(code), Q = ldpc_code_q(code);
  int log2q = 0;
  while ((1 << log2q) < Q) log2q++;
  if ((1 << log2q) != Q) {
    ldpc_set_error("GFq_LDPC_NTT: the field order must be a power of two");
    return LDPC_ERR_UNSUPPORTED;
  }
  Lcg31 g{(uint32_t)seed};                                       // SRand(j), :683
  for (int i = 0; i < n; i++) x[i] = (int32_t)(g.next() % (uint32_t)Q);   // :684-685
  int rc = ldpc_code_syndrome(code, x, s);                       // enc(x, s), :686
  if (rc) return rc;
  if (ch == 0) ntt_bsc(g, n, Q, log2q, x, y, level, logfna);     // channel(x, y, error, logfna), :687
  else ntt_lap(g, n, Q, x, y, level, logfna);
  return LDPC_OK;
}

}  // namespace

extern "C" int GFq_LDPC_NTT_experiment(const char *matrix_file, int seed, const char *channel, double level,
                                       int32_t *x, int32_t *s, int32_t *y, int32_t *logfna) {
  if (!matrix_file || !x || !s || !y || !logfna) return LDPC_ERR_ARG;
  const int ch = ntt_channel_id(channel);
  if (ch < 0) return LDPC_ERR_ARG;
  ldpc_code *code = nullptr;
  int rc = ldpc_code_load_alist(matrix_file, &code);
  if (rc) return rc;
  rc = ntt_experiment(code, seed, ch, level, x, s, y, logfna);
  ldpc_code_free(code);
  return rc;
}

extern "C" int GFq_LDPC_NTT(int iteration, const char *channel, double level, const char *matrix_file,
                            int32_t *verdict, int32_t *iters) {
  SERIALISED;
  if (!verdict || iteration < 0) return LDPC_ERR_ARG;
  const int ch = ntt_channel_id(channel);
  if (ch < 0) return LDPC_ERR_ARG;
  ldpc_code *code;
  ldpc_decoder *dec;
  int rc = get_decoder(matrix_file, LDPC_ALG_GF2P_WHT, &code, &dec);
  if (rc) return rc;
  const int n = ldpc_code_n(code), m = ldpc_code_m(code), Q = ldpc_code_q(code);
  std::vector<int32_t> x(n), s(m), y(n), logfna((size_t)3 * n * Q), status(3), it(3);
  std::vector<uint8_t> synd((size_t)3 * m);
  for (int j = 1; j <= 3; j++) {                                 // :681
    rc = ntt_experiment(code, j, ch, level, x.data(), s.data(), y.data(), &logfna[(size_t)(j - 1) * n * Q]);
    if (rc) return rc;
    for (int k = 0; k < m; k++) synd[(size_t)(j - 1) * m + k] = (uint8_t)s[k];
  }
  // the three dec(logfna, s, iteration, x) calls of :692 as one batch
  rc = ldpc_decode_batch(dec, 3, logfna.data(), LDPC_DT_I32, synd.data(), iteration, nullptr, it.data(), nullptr,
                         status.data(), nullptr);
  if (rc) return rc;
  for (int j = 0; j < 3; j++) {
    verdict[j] = status[j] ? -1 : 0;                             // dec() returns 0 (:531) or -1 (:543)
    if (iters) iters[j] = it[j];
  }
  return LDPC_OK;
}
