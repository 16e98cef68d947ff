// engine.h -- shared host-side plumbing of libldpc_hip: error reporting, a
// device buffer holder and the Engine interface every algorithm implements.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/ldpc_hip.h"
#include "graph.h"

void ldpc_set_error(const std::string &s);

#define HIPCHK(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      ldpc_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
      return LDPC_ERR_HIP;                                                        \
    }                                                                             \
  } while (0)

// LDPC_HIP_SYNC=1 (debugging): wait for every kernel right after its launch, so that a fault is
// reported against the kernel that caused it instead of at the next copy
inline bool ldpc_sync_launches() {
  static const bool on = [] { const char *e = getenv("LDPC_HIP_SYNC"); return e && atoi(e) != 0; }();
  return on;
}

#define LAUNCHCHK(name)                                                           \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ == hipSuccess && ldpc_sync_launches()) e_ = hipDeviceSynchronize();    \
    if (e_ != hipSuccess) {                                                       \
      ldpc_set_error(std::string("launch ") + name + ": " + hipGetErrorString(e_)); \
      return LDPC_ERR_HIP;                                                        \
    }                                                                             \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int ensure(size_t want) {
    if (want <= bytes) return LDPC_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    if (hipMalloc(&p, want) != hipSuccess) {
      p = nullptr;
      ldpc_set_error("hipMalloc of " + std::to_string(want) + " bytes failed");
      return LDPC_ERR_NOMEM;
    }
    bytes = want;
    return LDPC_OK;
  }
  template <typename U> U *as() const { return reinterpret_cast<U *>(p); }
};

// Points a DevBuf at caller-provided device memory for the life of the guard (the
// allocation-free entry point carves an engine's workspace out of one array this way).
struct DevBufBorrow {
  DevBuf &b;
  void *p0;
  size_t n0;
  DevBufBorrow(DevBuf &buf, void *p, size_t n) : b(buf), p0(buf.p), n0(buf.bytes) { b.p = p; b.bytes = n; }
  ~DevBufBorrow() { b.p = p0; b.bytes = n0; }
  DevBufBorrow(const DevBufBorrow &) = delete;
  DevBufBorrow &operator=(const DevBufBorrow &) = delete;
};

struct Engine {
  virtual ~Engine() { profile_free(); }
  virtual int init(const ldpc_code &c, const ldpc_opts &o) = 0;
  // device pointers
  virtual int decode(int nframes, const void *llr, int dt, const uint8_t *synd, int max_iter,
                     uint8_t *bits, int32_t *iters, uint8_t *sout, int32_t *status, float *llr_out,
                     hipStream_t st) = 0;
  virtual size_t llr_elems_per_frame() const = 0;   // n (binary) or n*q
  // single frame, device pointers, synchronous; writes the Verilog trace (Q8.8 engines only)
  virtual int decode_traced(const void *, int, const uint8_t *, int, uint8_t *, int32_t *, uint8_t *,
                            int32_t *, FILE *, int) {
    ldpc_set_error("message trace is only available for MINSUM_Q88");
    return LDPC_ERR_UNSUPPORTED;
  }
  // decode() with every byte of device scratch the call needs (work-queue head, message
  // store, per-frame state) taken from ws[0 .. ws_bytes) instead of the engine's own
  // buffers; allocates nothing.  ws is 256-byte aligned device memory.
  virtual int decode_ws(void *, size_t, int, const void *, int, const uint8_t *, int, uint8_t *, int32_t *,
                        uint8_t *, int32_t *, float *, hipStream_t) {
    ldpc_set_error("the allocation-free entry point serves the binary decoders only");
    return LDPC_ERR_UNSUPPORTED;
  }
  int last_iterations = 0;

  // Optional per-kernel timing (bench.py's roofline leg): hipEvent pairs around the
  // launches of each kernel class, read back with profile_read().
  enum { K_VN = 0, K_CN = 1, K_FUSED = 2, K_OTHER = 3, K_KINDS = 4 };
  struct ProfRec { int kind; hipEvent_t a, b; int frames; };
  bool profiling = false;
  int prof_cur_frames = 0;            // frames the launches being recorded work on (set by decode())
  long long prof_frames[4] = {};      // per kind: sum over the launches of the last profile_read() of those frames
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> prof_pool;
  hipEvent_t prof_event() {
    hipEvent_t e = nullptr;
    if (!prof_pool.empty()) { e = prof_pool.back(); prof_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  }
  void prof_begin(int kind, hipStream_t st) {
    if (!profiling) return;
    ProfRec r{kind, prof_event(), prof_event(), prof_cur_frames};
    (void)hipEventRecord(r.a, st);
    prof_recs.push_back(r);
  }
  void prof_end(hipStream_t st) {
    if (!profiling) return;
    (void)hipEventRecord(prof_recs.back().b, st);
  }
  // sums elapsed ms and launch counts per kind since the last read; synchronises
  void profile_read(double ms[K_KINDS], int count[K_KINDS]) {
    for (int k = 0; k < K_KINDS; k++) { ms[k] = 0; count[k] = 0; prof_frames[k] = 0; }
    for (auto &r : prof_recs) {
      (void)hipEventSynchronize(r.b);
      float t = 0;
      if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.kind] += t; count[r.kind]++; prof_frames[r.kind] += r.frames; }
      prof_pool.push_back(r.a);
      prof_pool.push_back(r.b);
    }
    prof_recs.clear();
  }
  void profile_free() {
    double ms[K_KINDS]; int c[K_KINDS];
    profile_read(ms, c);
    for (auto e : prof_pool) (void)hipEventDestroy(e);
    prof_pool.clear();
  }
};


// host-built lookup tables (tables.cpp)
void ldpc_build_spa_tables(std::vector<int32_t> &flogtanh, std::vector<int32_t> &fgallag);
void ldpc_build_wht_tables(std::vector<int32_t> &flog, std::vector<uint32_t> &fexp);
// compact exact forms of the WHT tables (tables.cpp)
constexpr int kWhtFrac = 1 << 14;   // fraction cells of fexp
constexpr int kWhtSeg = 512;        // log2 segments of flog
constexpr int kWhtSmall = 2048;     // flog entries kept verbatim (the last entry that is not round(16384 log2 i) - 14 * 16384 is 1955)
struct WhtCompact {
  std::vector<uint32_t> xexp;       // [kWhtFrac]
  std::vector<float> lseg;          // [kWhtSeg][4]: c0 fraction, c1, c2, unused
  std::vector<int32_t> lsegi;       // [kWhtSeg]: c0 integer part
};
void ldpc_build_wht_compact(WhtCompact &c);
uint32_t ldpc_wht_fexp_compact(const WhtCompact &c, int x);
int ldpc_wht_flog_compact(const WhtCompact &c, int i, bool *defer);
long ldpc_wht_compact_selfcheck(long stats[4]);
// compact exact form of the two Q8.14 sum-product tables (tables.cpp; kernels: binary_kernels.hip.h
// cn_tab_lds_kernel).  Both tables are monotone, zero from entry 181 705 on, step by at most one per
// entry from entry 16 384 on and fit 16 bits from entry 2 048 on.  Per table (kTabcTable bytes) three
// arrays whose values ADD UP to |t[i]|: each is read at a clamped index and holds 0 where it clamps.
//   u32 a[0..2048]:       |t[i]| below 2 048, a[2048] = 0
//   u16 b[2047..16384]:   |t[i]| in [2 048, 16 384), b[2047] = b[16384] = 0
//   u32 c[kTabcBlocks]:   from 16 384 on, per 16 entries: |t| of the entry before the block << 16 | one bit per
//                         entry that is one smaller than its predecessor; the last entry (181 759) and all
//                         beyond are 0, and arguments below 16 384 wrap to it
// image: Flogtanh's three arrays, then Fgallag's = kTabcBytes (153 KiB, one workgroup's LDS).
constexpr int kTabcSmall = 2048;
constexpr int kTabcSplit = 16384;
constexpr int kTabcBlocks = 10336;
constexpr int kTabcEnd = kTabcSplit + 16 * kTabcBlocks;          // 181 760
constexpr int kTabcOffB = 8208;                                  // (2049 * 4, rounded up to 16)
constexpr int kTabcOffC = kTabcOffB + 28688;                     // (14 338 * 2, rounded up to 16) = 36 896
constexpr int kTabcTable = kTabcOffC + kTabcBlocks * 4;          // 78 240
constexpr int kTabcBytes = 2 * kTabcTable;                       // 156 480
bool ldpc_build_spa_compact(const std::vector<int32_t> &flogtanh, const std::vector<int32_t> &fgallag,
                            std::vector<uint32_t> &image);
int ldpc_spa_compact_mag(const uint32_t *image, int table, unsigned i);   // the kernels' arithmetic, restated
long ldpc_spa_compact_selfcheck(long stats[4]);
void ldpc_gf_tables_host(int q, int *logq, int *expq);
int ldpc_gf_mul_host(int q, int a, int b);

// gf_engine.hip
Engine *ldpc_make_gf_engine(int alg, const ldpc_code &c);
// fused_float.hip: frame-resident floating-point min-sum (prec 4: binary32, 8: binary64);
// nullptr when the code does not fit in LDS
Engine *ldpc_make_fused_float_engine(int prec, const ldpc_code &c);
// chain_q88.hip: frame-resident Q8.8 min-sum for staircase (IRA / DVB-S2) codes too large for the
// fused decoder's LDS sums; init() returns LDPC_ERR_UNSUPPORTED for codes it does not serve
Engine *ldpc_make_chain_q88_engine();
// the staircase binary32 min-sum decoder (stair_f32.hip); init() returns LDPC_ERR_UNSUPPORTED for codes it does not serve
Engine *ldpc_make_stair_f32_engine(const ldpc_opts &o);
