// ldpc_hip.hip -- C ABI (include/ldpc_hip.h) and host orchestration of the
// gfx950 decoders.  One ldpc_decoder = one GPU, one resident workspace sized
// for opts.max_frames frames; a decode call is a fixed sequence of kernel
// launches on the caller's stream (see run_pass()).
#include "../../include/ldpc_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "binary_kernels.hip.h"
#include "fused_kernels.hip.h"
#include "engine.h"
#include "graph.h"
#include "lds_schedule.h"

// ------------------------------------------------------------------ errors

static thread_local std::string g_last_error;

void ldpc_set_error(const std::string &s) { g_last_error = s; }

extern "C" const char *ldpc_last_error(void) { return g_last_error.c_str(); }

extern "C" const char *ldpc_strerror(int code) {
  switch (code) {
    case LDPC_OK: return "ok";
    case LDPC_ERR_FORMAT: return "malformed alist / graph";
    case LDPC_ERR_IO: return "cannot open file";
    case LDPC_ERR_NOMEM: return "out of memory";
    case LDPC_ERR_ARG: return "bad argument";
    case LDPC_ERR_HIP: return "HIP runtime error";
    case LDPC_ERR_UNSUPPORTED: return "unsupported algorithm/code combination";
    default: return "unknown error";
  }
}

extern "C" const char *ldpc_version(void) { return "libldpc_hip 0.1 gfx950"; }

// ------------------------------------------------------------------- codes

extern "C" int ldpc_code_load_alist(const char *path, ldpc_code **out) {
  if (!path || !out) return LDPC_ERR_ARG;
  std::unique_ptr<ldpc_code> c(new (std::nothrow) ldpc_code);
  if (!c) return LDPC_ERR_NOMEM;
  std::string err;
  int rc = ldpc_parse_alist(path, *c, err);
  if (rc != LDPC_OK) {
    ldpc_set_error(err);
    return rc;
  }
  *out = c.release();
  return LDPC_OK;
}

extern "C" int ldpc_code_from_csr(int n, int m, int q, const int32_t *row_ptr, const int32_t *col_idx,
                                  const int32_t *hval, ldpc_code **out) {
  if (!row_ptr || !col_idx || !out || n <= 0 || m <= 0 || q < 2 || q > 256) return LDPC_ERR_ARG;
  std::unique_ptr<ldpc_code> c(new (std::nothrow) ldpc_code);
  if (!c) return LDPC_ERR_NOMEM;
  c->n = n;
  c->m = m;
  c->q = q;
  // nothing is read through row_ptr before it has been checked
  if (row_ptr[0] != 0) { ldpc_set_error("row_ptr[0] is not 0"); return LDPC_ERR_FORMAT; }
  for (int j = 0; j < m; j++)
    if (row_ptr[j + 1] < row_ptr[j]) { ldpc_set_error("row_ptr not monotone"); return LDPC_ERR_FORMAT; }
  c->row_ptr.assign(row_ptr, row_ptr + m + 1);
  const int E = row_ptr[m];
  c->col_idx.assign(col_idx, col_idx + E);
  if (hval) c->hval.assign(hval, hval + E);
  std::string err;
  int rc = ldpc_finish_code(*c, err);
  if (rc != LDPC_OK) {
    ldpc_set_error(err);
    return rc;
  }
  c->source = "<csr>";
  *out = c.release();
  return LDPC_OK;
}

extern "C" void ldpc_code_free(ldpc_code *c) { delete c; }
extern "C" int ldpc_code_n(const ldpc_code *c) { return c ? c->n : LDPC_ERR_ARG; }
extern "C" int ldpc_code_m(const ldpc_code *c) { return c ? c->m : LDPC_ERR_ARG; }
extern "C" int ldpc_code_q(const ldpc_code *c) { return c ? c->q : LDPC_ERR_ARG; }
extern "C" int ldpc_code_edges(const ldpc_code *c) { return c ? c->E() : LDPC_ERR_ARG; }
extern "C" int ldpc_code_cmax(const ldpc_code *c) { return c ? c->cmax : LDPC_ERR_ARG; }
extern "C" int ldpc_code_rmax(const ldpc_code *c) { return c ? c->rmax : LDPC_ERR_ARG; }

extern "C" int ldpc_code_get_csr(const ldpc_code *c, int32_t *row_ptr, int32_t *col_idx, int32_t *hval) {
  if (!c) return LDPC_ERR_ARG;
  if (row_ptr) memcpy(row_ptr, c->row_ptr.data(), sizeof(int32_t) * (c->m + 1));
  if (col_idx) memcpy(col_idx, c->col_idx.data(), sizeof(int32_t) * c->E());
  if (hval) memcpy(hval, c->hval.data(), sizeof(int32_t) * c->E());
  return LDPC_OK;
}

extern "C" int ldpc_code_get_csc(const ldpc_code *c, int32_t *col_ptr, int32_t *edge_of) {
  if (!c) return LDPC_ERR_ARG;
  if (col_ptr) memcpy(col_ptr, c->col_ptr.data(), sizeof(int32_t) * (c->n + 1));
  if (edge_of) memcpy(edge_of, c->edge_of.data(), sizeof(int32_t) * c->E());
  return LDPC_OK;
}

int ldpc_gf_mul_host(int q, int a, int b);   // gf_engine.hip

extern "C" int ldpc_code_syndrome(const ldpc_code *c, const int32_t *x, int32_t *s) {
  if (!c || !x || !s) return LDPC_ERR_ARG;
  const bool pow2 = (c->q & (c->q - 1)) == 0;
  for (int j = 0; j < c->m; j++) {
    int acc = 0;
    for (int e = c->row_ptr[j]; e < c->row_ptr[j + 1]; e++) {
      if (c->q == 2) {
        acc ^= x[c->col_idx[e]] & 1;
      } else {
        int t = ldpc_gf_mul_host(c->q, x[c->col_idx[e]], c->hval[e]);
        acc = pow2 ? (acc ^ t) : (acc + t) % c->q;
      }
    }
    s[j] = acc;
  }
  return LDPC_OK;
}

// ----------------------------------------------------------------- engines

static size_t dtype_size(int dt) {
  switch (dt) {
    case LDPC_DT_F64: return 8;
    case LDPC_DT_F32: return 4;
    case LDPC_DT_I16: return 2;
    case LDPC_DT_I32: return 4;
    default: return 0;
  }
}

namespace {
using namespace ldpc;

constexpr int EV_RING = 4;

// ---- second stage of a two-stage decode (BinaryEngine::decode): the frames that have not met their
// syndrome when the first stage stops are listed, their inputs gathered into a dense batch, and
// their results written back over the first stage's
// (first: number of the span's first frame in the caller's batch)
__global__ void list_open_frames_kernel(int nframes, int V, const uint8_t *done, int *count, int *idx, int first) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nframes) return;
  if (!((done[f / V] >> (f % V)) & 1)) idx[atomicAdd(count, 1)] = first + f;
}
// rows of `row` bytes: dst[i] = src[idx[i]] (gather) or dst[idx[i]] = src[i] (scatter); one workgroup per row
template <typename U, bool SCATTER>
__global__ void move_rows_kernel(const U *__restrict__ src, U *__restrict__ dst, const int *__restrict__ idx, size_t row) {
  const size_t i = blockIdx.x, f = (size_t)idx[blockIdx.x];
  const U *s = src + (SCATTER ? i : f) * row;
  U *d = dst + (SCATTER ? f : i) * row;
  for (size_t k = threadIdx.x; k < row; k += blockDim.x) d[k] = s[k];
}
template <bool SCATTER>
static void move_rows(const void *src, void *dst, const int *idx, int count, size_t row_bytes, hipStream_t st) {
  if (count <= 0 || !src || !dst) return;
  const bool w4 = row_bytes % 4 == 0 && ((size_t)src | (size_t)dst) % 4 == 0;
  if (w4)
    hipLaunchKernelGGL((move_rows_kernel<uint32_t, SCATTER>), dim3((unsigned)count), dim3(256), 0, st, (const uint32_t *)src,
                       (uint32_t *)dst, idx, row_bytes / 4);
  else
    hipLaunchKernelGGL((move_rows_kernel<uint8_t, SCATTER>), dim3((unsigned)count), dim3(256), 0, st, (const uint8_t *)src,
                       (uint8_t *)dst, idx, row_bytes);
}

template <typename T, int ALG>
struct BinaryEngine : Engine {
  typedef typename MT<T>::vec vec;
  static constexpr int V = MT<T>::V;
  static constexpr int SF = 64 * V;

  int tab_nt = 0, tab_rb = -1;   // cn_tab_lds_kernel: threads per workgroup (one workgroup per CU), rows in registers at a time (env overrides)
  ldpc_opts opts{};
  int n = 0, m = 0, E = 0, cmax = 0, rmax = 0;
  int nslab_max = 0;
  int num_cu = 256;
  int cn_variant = 0;
  int vn_bucket = 0;
  DevBuf g_row_ptr, g_col_idx, g_col_ptr, g_edge_of, trace_buf;
  std::vector<int32_t> h_row_ptr, h_col_idx, h_col_ptr, h_edge_of;   // host copy (trace ordering)
  DevBuf q0, msg, post, hard, synd, sout, fail, done, iters, remaining, tab_lt, tab_g, tab_c;
  // two-stage decode: list of open frames (+ count), their gathered inputs and their results
  DevBuf s2_idx, s2_llr, s2_synd, s2_bits, s2_iters, s2_sout, s2_status, s2_post;
  bool one_stage_only = false;   // set by the allocation-free entry (the second stage allocates its batch)
  int32_t *h_remaining = nullptr;
  int remaining_cap = 0;
  hipEvent_t ev[EV_RING] = {};
  Graph g{};
  Tables tab{};

  ~BinaryEngine() override {
    if (h_remaining) (void)hipHostFree(h_remaining);
    for (auto &e : ev)
      if (e) (void)hipEventDestroy(e);
  }

  size_t llr_elems_per_frame() const override { return (size_t)n; }

  int upload(DevBuf &b, const std::vector<int32_t> &v) {
    int rc = b.ensure(sizeof(int32_t) * (v.size() ? v.size() : 1));
    if (rc) return rc;
    HIPCHK(hipMemcpy(b.p, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice));
    return LDPC_OK;
  }

  int ensure_remaining(int slots) {
    if (slots <= remaining_cap) return LDPC_OK;
    if (h_remaining) (void)hipHostFree(h_remaining);
    h_remaining = nullptr;
    int cap = slots < 256 ? 256 : slots;
    HIPCHK(hipHostMalloc((void **)&h_remaining, sizeof(int32_t) * cap, hipHostMallocDefault));
    int rc = remaining.ensure(sizeof(int32_t) * cap);
    if (rc) return rc;
    remaining_cap = cap;
    return LDPC_OK;
  }

  int init(const ldpc_code &c, const ldpc_opts &o) override {
    opts = o;
    n = c.n; m = c.m; E = c.E(); cmax = c.cmax; rmax = c.rmax;
    h_row_ptr = c.row_ptr; h_col_idx = c.col_idx; h_col_ptr = c.col_ptr; h_edge_of = c.edge_of;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, o.device));
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int mf = o.max_frames > 0 ? o.max_frames : 4096;
    nslab_max = (mf + SF - 1) / SF;
    int rc;
    if ((rc = upload(g_row_ptr, c.row_ptr))) return rc;
    if ((rc = upload(g_col_idx, c.col_idx))) return rc;
    if ((rc = upload(g_col_ptr, c.col_ptr))) return rc;
    if ((rc = upload(g_edge_of, c.edge_of))) return rc;
    g.row_ptr = g_row_ptr.as<int32_t>();
    g.col_idx = g_col_idx.as<int32_t>();
    g.col_ptr = g_col_ptr.as<int32_t>();
    g.edge_of = g_edge_of.as<int32_t>();
    g.n = n; g.m = m; g.E = E;
    const size_t rows = (size_t)nslab_max * 64;
    if ((rc = q0.ensure(rows * n * sizeof(vec)))) return rc;
    if ((rc = msg.ensure(rows * E * sizeof(vec)))) return rc;
    if (o.want_llr && (rc = post.ensure(rows * n * sizeof(vec)))) return rc;
    if ((rc = hard.ensure(rows * n))) return rc;
    if ((rc = synd.ensure(rows * m))) return rc;
    if ((rc = sout.ensure(rows * m))) return rc;
    if ((rc = fail.ensure(rows * sizeof(uint32_t)))) return rc;
    if ((rc = done.ensure(rows))) return rc;
    if ((rc = iters.ensure(rows * V * sizeof(int32_t)))) return rc;
    if ((rc = ensure_remaining(256))) return rc;
    for (auto &e : ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    vn_bucket = cmax <= 4 ? 4 : cmax <= 8 ? 8 : cmax <= 16 ? 16 : 0;

    // check-node kernel selection
    //   0 auto, 1 LDS staged through registers, 2 two-pass reload, 3 register
    //   resident (min-sum, rmax <= 32), 4 LDS staged by global->LDS DMA, 5 single read
    //   with compressed per-frame state (min-sum, rmax <= 64), 6 both tables in LDS (table
    //   sum-product; the default there, 4 when the tables do not have the structure it needs), 7 chunked
    //   tanh-domain kernel (binary64 sum-product, row weights 17 ... 64; the default there)
    cn_variant = o.cn_variant;
    if (const char *e = getenv("LDPC_HIP_CN_VARIANT")) cn_variant = atoi(e);
    const int cn_requested = cn_variant;
    if (cn_variant == 6) cn_variant = 0;   // (chosen below, once the image is built)
    // measured on PEG-16000 (tools/bench_codes.py): the single-read kernel wins for min-sum
    // (+11 % over the reload kernel), the reload kernel for the float sum-product (+48 % over
    // the LDS-staged one), the DMA-staged one for the table sum-product
    // (float sum-product with row weights <= 16, e.g. the DVB-S2 codes: register-resident kernel)
    if (cn_variant == 0)
      cn_variant = ((ALG == ALG_MS || ALG == ALG_MSQ) && rmax <= 64) ? 5 : (ALG == ALG_TAB || (ALG == ALG_MS && sizeof(T) == 8)) ? 4 :
                   (ALG == ALG_SPA && rmax <= 16) ? 3 : (ALG == ALG_SPA && sizeof(T) == 8 && rmax <= 64 && LDPC_SPA_F64_PRODUCT) ? 7 : 2;
    // 7: binary64 sum-product, row weights 17 ... 64, chunked tanh-domain kernel (cn_spa64_chunk_kernel)
    if (cn_variant == 7 && !(ALG == ALG_SPA && sizeof(T) == 8 && rmax > 16 && rmax <= 64 && LDPC_SPA_F64_PRODUCT)) cn_variant = (ALG == ALG_SPA && rmax <= 16) ? 3 : 2;
    const size_t lds_need = (size_t)rmax * 1024;
    if ((cn_variant == 1 || cn_variant == 4) && lds_need > 160 * 1024) cn_variant = 2;
    if (cn_variant == 3 && ALG == ALG_SPA && rmax > 16) cn_variant = 2;
    if (cn_variant == 3 && (rmax > 32 || !(ALG == ALG_MS || ALG == ALG_MSQ || ALG == ALG_SPA))) cn_variant = 1;
    if (cn_variant == 5 && (rmax > 64 || !(ALG == ALG_MS || ALG == ALG_MSQ))) cn_variant = 2;
    if (cn_variant == 1 && lds_need > 64 * 1024)
      HIPCHK(hipFuncSetAttribute((const void *)cn_kernel<T, ALG, false>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    if (cn_variant == 4 && lds_need > 64 * 1024)
      HIPCHK(hipFuncSetAttribute((const void *)cn_kernel<T, ALG, true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));

    if (ALG == ALG_TAB) {
      std::vector<int32_t> lt, ga;
      ldpc_build_spa_tables(lt, ga);
      if ((rc = upload(tab_lt, lt))) return rc;
      if ((rc = upload(tab_g, ga))) return rc;
      tab.flogtanh = tab_lt.as<int32_t>();
      tab.fgallag = tab_g.as<int32_t>();
      // PEG-16000, 5 184 frames: 2.7 Gbit/s with the gathers (variant 4) -> see profiles/r02_bench_codes_table.txt
      std::vector<uint32_t> image;
      if ((cn_requested == 0 || cn_requested == 6) && ldpc_build_spa_compact(lt, ga, image)) {
        static_assert(TABC_SMALL == kTabcSmall && TABC_SPLIT == kTabcSplit && TABC_BLOCKS == kTabcBlocks &&
                      TABC_OFF_B == kTabcOffB && TABC_OFF_C == kTabcOffC && TABC_TABLE == kTabcTable &&
                      TABC_BYTES == kTabcBytes && kTabcOffB >= (kTabcSmall + 1) * 4 &&
                      kTabcOffC - kTabcOffB >= (kTabcSplit - kTabcSmall + 2) * 2 && kTabcBytes <= 160 * 1024,
                      "image format");
        if ((rc = tab_c.ensure(kTabcBytes))) return rc;
        HIPCHK(hipMemcpy(tab_c.p, image.data(), kTabcBytes, hipMemcpyHostToDevice));
        if constexpr (ALG == ALG_TAB) {
          if (const char *e = getenv("LDPC_HIP_TAB_NT")) tab_nt = atoi(e) == 1024 ? 1024 : 512;
          if (const char *e = getenv("LDPC_HIP_TAB_CH")) tab_rb = atoi(e);
#define TAB_ATTR(RB, NT) HIPCHK(hipFuncSetAttribute((const void *)cn_tab_lds_kernel<RB, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, kTabcBytes))
#define TAB_ATTR3(RB) TAB_ATTR(RB, 512); TAB_ATTR(RB, 1024)
          TAB_ATTR3(8); TAB_ATTR(16, 512); TAB_ATTR3(4);
          HIPCHK(hipFuncSetAttribute((const void *)cn_tab_lds_kernel<8, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kTabcBytes));
          HIPCHK(hipFuncSetAttribute((const void *)cn_tab_lds_kernel<8, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kTabcBytes));
#undef TAB_ATTR3
#undef TAB_ATTR
        }
        cn_variant = 6;
      }
    }
    return LDPC_OK;
  }

  template <bool FIRST, bool STORE>
  int launch_vn(int nslab, hipStream_t st) {
    const long items = (long)n * nslab;
    long blocks = (items + 3) / 4;
    const long cap = (long)num_cu * 8;
    if (blocks > cap) blocks = cap;
    vec *p = opts.want_llr ? post.as<vec>() : nullptr;
    dim3 grid((unsigned)blocks), block(256);
#define VN_LAUNCH(B)                                                                         \
  hipLaunchKernelGGL((vn_kernel<T, ALG, B, FIRST, STORE>), grid, block, 0, st, g, nslab,      \
                     q0.as<vec>(), msg.as<vec>(), hard.as<uint8_t>(), done.as<uint8_t>(), p)
    switch (vn_bucket) {
      case 4: VN_LAUNCH(4); break;
      case 8: VN_LAUNCH(8); break;
      case 16: VN_LAUNCH(16); break;
      default: VN_LAUNCH(0); break;
    }
#undef VN_LAUNCH
    LAUNCHCHK("vn_kernel");
    return LDPC_OK;
  }

  // waves per slab so that every wave gets the same number of checks
  int balanced_wps(int nslab, long target_waves) const {
    long wps = target_waves / nslab;
    if (wps < 1) wps = 1;
    if (wps > m) wps = m;
    const long per = (m + wps - 1) / wps;
    wps = (m + per - 1) / per;
    return (int)wps;
  }

  int launch_cn(int nslab, const uint8_t *sy, hipStream_t st) {
    int oversub = 1;
    if (const char *e = getenv("LDPC_HIP_CN_OVERSUB")) oversub = atoi(e) > 0 ? atoi(e) : 1;
    if (cn_variant == 1 || cn_variant == 4) {
      const size_t lds = (size_t)rmax * 1024;
      long per_cu = (160 * 1024) / (long)lds;
      if (per_cu > 32) per_cu = 32;
      if (per_cu < 1) per_cu = 1;
      const int wps = balanced_wps(nslab, (long)num_cu * per_cu * oversub);
      dim3 grid((unsigned)(wps * nslab)), block(64);
      if (cn_variant == 1)
        hipLaunchKernelGGL((cn_kernel<T, ALG, false>), grid, block, lds, st, g, nslab, msg.as<vec>(),
                           hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab);
      else
        hipLaunchKernelGGL((cn_kernel<T, ALG, true>), grid, block, lds, st, g, nslab, msg.as<vec>(),
                           hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab);
      LAUNCHCHK("cn_kernel");
      return LDPC_OK;
    }
    if (cn_variant == 6) {
      // one workgroup per CU (its LDS holds the tables), TAB_NT / 64 waves each
      if constexpr (ALG == ALG_TAB) {
        // rows of a check in registers at a time / waves per CU (tools/probes/tab_sweep.sh, PEG-16000 Gbit/s):
        // 4 / 16: 6.88, 8 / 8: 6.50, 4 / 8: 6.52, 16 / 8: 5.21, whole checks of 32 rows / 8: 5.30, 8 / 16 (spills): 5.30
        // -- the double-buffered chunks overlap the row loads with the lookups, a whole check in registers does
        // not.  Row weights <= 8 (DVB-S2): one chunk of eight holds the check and keeps its log tanh values; built
        // without the chunked path and with the scheduler fenced every four rows it fits 16 waves per CU
        // (DVB-S2 0.5: 2.82 -> 3.09 Gbit/s).
        int ch = rmax <= 8 ? 8 : 4;
        if (tab_rb == 4 || tab_rb == 8 || tab_rb == 16) ch = tab_rb;
        int nt = (ch == 4 || rmax <= 8) ? 1024 : 512;   // (the single-chunk build for row weights <= 8 needs 103 registers)
        if (tab_nt > 0 && ch <= 8) nt = tab_nt;
        const int wpb = nt / 64;
        const int wps = balanced_wps(nslab, (long)num_cu * wpb * oversub);
        const long waves = (long)wps * nslab;
        dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(nt);
#define TAB_LAUNCH(RB, NT)                                                                                 \
  hipLaunchKernelGGL((cn_tab_lds_kernel<RB, NT>), grid, block, kTabcBytes, st, g, nslab, wps, msg.as<vec>(), \
                     hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab_c.as<uint32_t>())
#define TAB_LAUNCH3(RB) do { if (nt == 512) TAB_LAUNCH(RB, 512); else TAB_LAUNCH(RB, 1024); } while (0)
        if (ch == 16) TAB_LAUNCH(16, 512);
        else if (ch == 4) TAB_LAUNCH3(4);
        else if (rmax <= 8 && nt == 1024)
          hipLaunchKernelGGL((cn_tab_lds_kernel<8, 1024, true>), grid, block, kTabcBytes, st, g, nslab, wps, msg.as<vec>(),
                             hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab_c.as<uint32_t>());
        else if (rmax <= 8)
          hipLaunchKernelGGL((cn_tab_lds_kernel<8, 512, true>), grid, block, kTabcBytes, st, g, nslab, wps, msg.as<vec>(),
                             hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab_c.as<uint32_t>());
        else TAB_LAUNCH3(8);
#undef TAB_LAUNCH3
#undef TAB_LAUNCH
      }
      LAUNCHCHK("cn_tab_lds_kernel");
      return LDPC_OK;
    }
    // 256-thread blocks, one wave per (check, slab) item
    const int wps = balanced_wps(nslab, (long)num_cu * 16 * oversub);
    const long waves = (long)wps * nslab;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    if (cn_variant == 3) {
      if constexpr (ALG == ALG_MS || ALG == ALG_MSQ || ALG == ALG_SPA) {
        if (rmax <= 8)
          hipLaunchKernelGGL((cn_reg_kernel<T, ALG, 8>), grid, block, 0, st, g, nslab, msg.as<vec>(),
                             hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab);
        else if (rmax <= 16)
          hipLaunchKernelGGL((cn_reg_kernel<T, ALG, 16>), grid, block, 0, st, g, nslab, msg.as<vec>(),
                             hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab);
        else if constexpr (ALG != ALG_SPA)
          hipLaunchKernelGGL((cn_reg_kernel<T, ALG, 32>), grid, block, 0, st, g, nslab, msg.as<vec>(),
                             hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab);
      }
      LAUNCHCHK("cn_reg_kernel");
      return LDPC_OK;
    }
    if (cn_variant == 7) {
      if constexpr (ALG == ALG_SPA && sizeof(T) == 8)
        hipLaunchKernelGGL((cn_spa64_chunk_kernel<8>), grid, block, 0, st, g, nslab, msg.as<vec>(), hard.as<uint8_t>(), sy,
                           fail.as<uint32_t>());
      LAUNCHCHK("cn_spa64_chunk_kernel");
      return LDPC_OK;
    }
    if (cn_variant == 5) {
      if constexpr (ALG == ALG_MS || ALG == ALG_MSQ)
        hipLaunchKernelGGL((cn_single_read_kernel<T, ALG>), grid, block, 0, st, g, nslab, msg.as<vec>(),
                           hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab);
      LAUNCHCHK("cn_single_read_kernel");
      return LDPC_OK;
    }
    hipLaunchKernelGGL((cn_reload_kernel<T, ALG>), grid, block, 0, st, g, nslab, msg.as<vec>(),
                       hard.as<uint8_t>(), sy, fail.as<uint32_t>(), tab);
    LAUNCHCHK("cn_reload_kernel");
    return LDPC_OK;
  }

  // to_host: the count of open frames also goes straight into the pinned word the host polls (h_remaining[slot])
  int launch_mark(int nslab, int nframes, int it, int ignore, int slot, hipStream_t st, bool to_host = false) {
    const int threads = std::min(1024, (nslab * 64 + 63) / 64 * 64);
    hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(threads), 0, st, nslab, V, nframes, it,
                       ignore, fail.as<uint32_t>(), done.as<uint8_t>(), iters.as<int32_t>(),
                       remaining.as<int32_t>() + slot, to_host ? h_remaining + slot : nullptr);
    LAUNCHCHK("mark_kernel");
    return LDPC_OK;
  }

  template <typename InT>
  int launch_ingest(const void *llr, int nframes, int nslab, hipStream_t st) {
    dim3 grid((n + 31) / 32, nslab), block(256);
    hipLaunchKernelGGL((ingest_kernel<InT, T, ALG>), grid, block, 0, st, (const InT *)llr, nframes, n,
                       q0.as<vec>());
    LAUNCHCHK("ingest_kernel");
    return LDPC_OK;
  }

  // Two stages when frames may leave early (opts.early_exit) and the batch is large: the slab layout
  // keeps every frame of a batch in the passes until the last one is done, so a batch in which 95 %
  // of the frames converge in 15 iterations and the rest need 50 costs 50 iterations for all.  The
  // first stage therefore stops once at most 1/16 of the frames are still open (and at least 8
  // iterations remain); the open frames are decoded again, from their channel values, as a dense
  // batch of their own.  dec() is deterministic, so a frame's bits, iteration count, syndrome and
  // verdict are the ones one stage gives (LDPC_HIP_TWO_STAGE=0 turns the second stage off).
  // Frames per pass over the message store.  The min-sum passes are bound by memory traffic, and a message store that
  // stays in the 256 MB Infinity Cache moves 6.0 - 6.4 TB/s in their access shape against 4.9 TB/s from HBM
  // (tools/hbm_ceiling with 64 - 192 MiB buffers against 1 - 4 GiB): a large call is cut into
  // spans of about 240 MB of messages, each decoded to the end before the next starts.  PEG-16000: Q8.8 14.8 -> 17.5
  // Gbit/s, binary32 7.7 -> 9.6, binary64 3.9 -> 4.9.  The sum-product passes are bound by arithmetic and gain
  // less (+2 ... +15 %), provided a span still fills the GPU several times over (the floors below).
  // LDPC_HIP_PASS_MB overrides (0: never cut).
  int frames_per_span(int nframes) const {
    double mb = 240.0;
    if (const char *e = getenv("LDPC_HIP_PASS_MB")) mb = atof(e);
    if (mb <= 0.0) return nframes;
    const double per_frame = (double)E * sizeof(T);
    // whole slabs (64 lanes x V frames each), at least two: DVB-S2 0.5, whose slab alone is 232 MB, gains nothing from
    // one-slab spans and 13 - 17 % from two (tools/probes/span_sweep.sh, profiles/r03_twopass_spans.txt)
    long per0 = std::max(2L, (long)(mb * 1e6 / per_frame / SF + 0.5)) * SF;
    // the sum-product check passes are bound by arithmetic: a span must still fill the GPU several times over
    // (32000.2240, 2 240 checks: 256-frame spans cost SPA_F64 8 %; PEG-16000, 1 600 checks: 640-frame spans gain 7 %),
    // so at least 10^6 check-frames per launch
    if (ALG != ALG_MS && ALG != ALG_MSQ) {
      // (the table kernel runs one persistent workgroup per CU: 512-frame spans cost 32000.2240 17 %; from 4 x 10^6
      // check-frames on nothing is lost and DVB-S2 0.5 / 4095.737 gain 8 %)
      const long least = ALG == ALG_TAB ? 4000000L : 1000000L;
      per0 = std::max(per0, ((least + m - 1) / m + SF - 1) / SF * SF);
    }
    if (const char *e = getenv("LDPC_HIP_PASS_SLABS")) per0 = std::max(1L, (long)atoi(e)) * SF;   // (tuning)
    // (spans of exactly that size and a shorter last one: evening them out rounds up to whole slabs, and the measured
    // optimum is narrow -- PEG-16000 Q8.8: 2 560 frames per span 17.5 Gbit/s, 3 072 frames 16.5)
    return 2 * (long)nframes >= 3 * per0 ? (int)per0 : nframes;
  }

  // One span after the other through decode_once(); f0 = first frame of the span in the caller's arrays.
  // stage1: stop a span's iterations once at most 1/16 of its frames are still open and list those (numbered within the
  // caller's batch) at d_idx / d_count.
  int decode_spans(int nframes, const void *llr, int dt, const uint8_t *sy_in, int max_iter, uint8_t *bits, int32_t *iters_out,
                   uint8_t *sout_out, int32_t *status, float *llr_out, hipStream_t st, bool stage1, int *d_count, int *d_idx) {
    static const size_t elt[4] = {8, 4, 2, 4};
    int per = frames_per_span(nframes);
    if (dt < 0 || dt > 3 || !llr || nframes > nslab_max * SF) per = nframes;   // (bad arguments: decode_once says so)
    int most = 0;
    for (int f0 = 0, cnt = 0; f0 < nframes; f0 += cnt) {
      cnt = std::min(per, nframes - f0);
      if (nframes - f0 - cnt < per / 4) cnt = nframes - f0;      // (a short tail rides with the last span)
      bool stopped = false;
      const int rc = decode_once(cnt, per < nframes ? (const void *)((const char *)llr + (size_t)f0 * n * elt[dt]) : llr, dt,
                                 sy_in ? sy_in + (size_t)f0 * m : nullptr, max_iter, bits ? bits + (size_t)f0 * n : nullptr,
                                 iters_out ? iters_out + f0 : nullptr, sout_out ? sout_out + (size_t)f0 * m : nullptr,
                                 status ? status + f0 : nullptr, llr_out ? llr_out + (size_t)f0 * n : nullptr, st,
                                 stage1 ? cnt / 16 : 0, stage1 ? &stopped : nullptr);
      if (rc) return rc;
      most = std::max(most, last_iterations);
      if (stopped) {
        hipLaunchKernelGGL(list_open_frames_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, cnt, V, done.as<uint8_t>(), d_count,
                           d_idx, f0);
        LAUNCHCHK("list_open_frames_kernel");
      }
    }
    last_iterations = most;
    return LDPC_OK;
  }

  int decode(int nframes, const void *llr, int dt, const uint8_t *sy_in, int max_iter, uint8_t *bits,
             int32_t *iters_out, uint8_t *sout_out, int32_t *status, float *llr_out,
             hipStream_t st) override {
    if (nframes <= 0) return LDPC_OK;
    const char *ts_env = getenv("LDPC_HIP_TWO_STAGE");
    const bool two_stage_env = !ts_env || atoi(ts_env) != 0;
    const bool two = two_stage_env && !one_stage_only && opts.early_exit != 0 && nframes >= 8 * SF && max_iter >= 16 &&
                     dt >= 0 && dt <= 3;
    if (!two) return decode_spans(nframes, llr, dt, sy_in, max_iter, bits, iters_out, sout_out, status, llr_out, st, false, nullptr, nullptr);
    int rc;
    if ((rc = s2_idx.ensure(sizeof(int) * ((size_t)nframes / 16 + 64)))) return rc;
    int *d_count = s2_idx.as<int>(), *d_idx = s2_idx.as<int>() + 32;
    HIPCHK(hipMemsetAsync(d_count, 0, sizeof(int), st));
    // the first stage, span by span; the frames it leaves open, in whatever order the lanes report them
    if ((rc = decode_spans(nframes, llr, dt, sy_in, max_iter, bits, iters_out, sout_out, status, llr_out, st, true, d_count, d_idx))) return rc;
    const int stage1_iterations = last_iterations;
    int count = 0;
    HIPCHK(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (count <= 0) return LDPC_OK;
    if (count > nframes / 16) { ldpc_set_error("two-stage decode: more open frames than the first stage reported"); return LDPC_ERR_HIP; }
    static const size_t elt[4] = {8, 4, 2, 4};
    const size_t in_row = (size_t)n * elt[dt];
    if ((rc = s2_llr.ensure(in_row * count))) return rc;
    if (sy_in && (rc = s2_synd.ensure((size_t)m * count))) return rc;
    if (bits && (rc = s2_bits.ensure((size_t)n * count))) return rc;
    if (iters_out && (rc = s2_iters.ensure(sizeof(int32_t) * count))) return rc;
    if (sout_out && (rc = s2_sout.ensure((size_t)m * count))) return rc;
    if (status && (rc = s2_status.ensure(sizeof(int32_t) * count))) return rc;
    if (llr_out && (rc = s2_post.ensure(sizeof(float) * (size_t)n * count))) return rc;
    move_rows<false>(llr, s2_llr.p, d_idx, count, in_row, st);
    if (sy_in) move_rows<false>(sy_in, s2_synd.p, d_idx, count, (size_t)m, st);
    LAUNCHCHK("move_rows_kernel (gather)");
    rc = decode_spans(count, s2_llr.p, dt, sy_in ? s2_synd.as<uint8_t>() : nullptr, max_iter, bits ? s2_bits.as<uint8_t>() : nullptr,
                      iters_out ? s2_iters.as<int32_t>() : nullptr, sout_out ? s2_sout.as<uint8_t>() : nullptr,
                      status ? s2_status.as<int32_t>() : nullptr, llr_out ? s2_post.as<float>() : nullptr, st, false, nullptr, nullptr);
    if (rc) return rc;
    // d_idx lives in s2_idx, which decode_once does not touch
    if (bits) move_rows<true>(s2_bits.p, bits, d_idx, count, (size_t)n, st);
    if (iters_out) move_rows<true>(s2_iters.p, iters_out, d_idx, count, sizeof(int32_t), st);
    if (sout_out) move_rows<true>(s2_sout.p, sout_out, d_idx, count, (size_t)m, st);
    if (status) move_rows<true>(s2_status.p, status, d_idx, count, sizeof(int32_t), st);
    if (llr_out) move_rows<true>(s2_post.p, llr_out, d_idx, count, sizeof(float) * (size_t)n, st);
    LAUNCHCHK("move_rows_kernel (scatter)");
    last_iterations = std::max(last_iterations, stage1_iterations);
    return LDPC_OK;
  }

  // stop_open > 0: leave the iteration loop once at most stop_open frames are still open, provided
  // at least 8 iterations remain (*stopped tells); those frames come out as "not converged"
  int decode_once(int nframes, const void *llr, int dt, const uint8_t *sy_in, int max_iter, uint8_t *bits,
                  int32_t *iters_out, uint8_t *sout_out, int32_t *status, float *llr_out,
                  hipStream_t st, int stop_open, bool *stopped) {
    if (nframes <= 0) return LDPC_OK;
    if (nframes > nslab_max * SF || !llr || max_iter < 0 || max_iter > 65000) {
      ldpc_set_error("nframes/max_iter out of range for this decoder");
      return LDPC_ERR_ARG;
    }
    if (llr_out && !opts.want_llr) {
      ldpc_set_error("llr_out needs opts.want_llr");
      return LDPC_ERR_ARG;
    }
    const int nslab = (nframes + SF - 1) / SF;
    const bool early = opts.early_exit != 0;
    int rc;
    if ((rc = ensure_remaining(max_iter + 2))) return rc;
    prof_cur_frames = nframes;

    hipLaunchKernelGGL(init_state_kernel, dim3((nslab * 64 + 255) / 256), dim3(256), 0, st, nslab, V,
                       nframes, fail.as<uint32_t>(), done.as<uint8_t>(), iters.as<int32_t>());
    LAUNCHCHK("init_state_kernel");
    HIPCHK(hipMemsetAsync(remaining.p, 0, sizeof(int32_t) * (max_iter + 2), st));
    HIPCHK(hipMemsetAsync(hard.p, 0, (size_t)nslab * 64 * n, st));
    if (opts.want_llr) HIPCHK(hipMemsetAsync(post.p, 0, (size_t)nslab * 64 * n * sizeof(vec), st));

    const uint8_t *sy = nullptr;
    if ((rc = ingest_all(nframes, nslab, llr, dt, sy_in, &sy, st))) return rc;

    // iteration 1's variable pass: c2v = 0, also emits sign(q0) as "iteration 0" decisions
    prof_begin(K_VN, st);
    if ((rc = launch_vn<true, true>(nslab, st))) return rc;
    prof_end(st);
    bool all_done = false;
    int executed = 0;
    for (int t = 1; t <= max_iter; t++) {
      if (t > 1) {
        prof_begin(K_VN, st);
        if ((rc = launch_vn<false, true>(nslab, st))) return rc;
        prof_end(st);
      }
      prof_begin(K_CN, st);
      if ((rc = launch_cn(nslab, sy, st))) return rc;
      prof_end(st);
      // the check pass tested the decisions of iteration t-1; only the Q8.8
      // decoder has an iteration-0 test (_fixed.c:198-207)
      const int ignore = (t == 1 && ALG != ALG_MSQ) ? 1 : 0;
      if ((rc = launch_mark(nslab, nframes, t - 1, ignore, t - 1, st, early))) return rc;
      executed = t;
      if (early) {
        HIPCHK(hipEventRecord(ev[(t - 1) % EV_RING], st));
        if (t >= 2) {
          HIPCHK(hipEventSynchronize(ev[(t - 2) % EV_RING]));
          if (h_remaining[t - 2] == 0) { all_done = true; break; }
          if (stop_open > 0 && h_remaining[t - 2] <= stop_open && max_iter - t >= 8) {
            // (the open frames' decisions of iteration t - 1 have been tested and stay as they are:
            // no closing variable pass, as when every frame is done)
            all_done = true;
            *stopped = true;
            break;
          }
        }
      }
    }
    if (!all_done && max_iter > 0) {
      if ((rc = launch_vn<false, false>(nslab, st))) return rc;   // decisions of the last iteration
    }
    {
      const int wps = balanced_wps(nslab, (long)num_cu * 16);
      const long waves = (long)wps * nslab;
      hipLaunchKernelGGL(synd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, g, nslab,
                         hard.as<uint8_t>(), sy, sout.as<uint8_t>(), fail.as<uint32_t>());
      LAUNCHCHK("synd_kernel");
      const int ignore = (max_iter == 0 && ALG != ALG_MSQ) ? 1 : 0;
      if ((rc = launch_mark(nslab, nframes, max_iter, ignore, max_iter + 1, st))) return rc;
    }
    last_iterations = executed;

    return emit_outputs(nframes, nslab, max_iter, bits, iters_out, sout_out, status, llr_out, st);
  }

  // Workspace of one decode call of nslab slabs, as (offset, bytes) into a 256-byte aligned
  // array: q0, msg, post, hard, synd, sout, fail, done, iters, remaining -- the buffers init()
  // allocates for nslab_max slabs.  Returns the total.
  enum { WS_Q0, WS_MSG, WS_POST, WS_HARD, WS_SYND, WS_SOUT, WS_FAIL, WS_DONE, WS_ITERS, WS_REM, WS_N };
  size_t ws_layout(int nslab, int rem_slots, size_t off[WS_N], size_t len[WS_N]) const {
    const size_t rows = (size_t)nslab * 64;
    len[WS_Q0] = rows * n * sizeof(vec);
    len[WS_MSG] = rows * E * sizeof(vec);
    len[WS_POST] = opts.want_llr ? rows * n * sizeof(vec) : 0;
    len[WS_HARD] = rows * n;
    len[WS_SYND] = rows * m;
    len[WS_SOUT] = rows * m;
    len[WS_FAIL] = rows * sizeof(uint32_t);
    len[WS_DONE] = rows;
    len[WS_ITERS] = rows * V * sizeof(int32_t);
    len[WS_REM] = sizeof(int32_t) * (size_t)rem_slots;
    size_t at = 0;
    for (int i = 0; i < WS_N; i++) { off[i] = at; at += (len[i] + 255) & ~(size_t)255; }
    return at;
  }

  int decode_ws(void *ws, size_t ws_bytes, int nframes, const void *llr, int dt, const uint8_t *sy_in, int max_iter,
                uint8_t *bits, int32_t *iters_out, uint8_t *sout_out, int32_t *status, float *llr_out,
                hipStream_t st) override {
    if (nframes <= 0) return LDPC_OK;
    if (!ws || nframes > nslab_max * SF || max_iter < 0) {
      ldpc_set_error("nframes/max_iter out of range for this decoder");
      return LDPC_ERR_ARG;
    }
    // the host-side poll slots were pinned by init(); growing them would allocate
    if (max_iter + 2 > remaining_cap) {
      ldpc_set_error("allocation-free decode: max_iter " + std::to_string(max_iter) + " beyond the " +
                     std::to_string(remaining_cap - 2) + " iterations this entry point serves");
      return LDPC_ERR_ARG;
    }
    const int nslab = (nframes + SF - 1) / SF;
    size_t off[WS_N], len[WS_N];
    const size_t need = ws_layout(nslab, remaining_cap, off, len);
    if (need > ws_bytes) {
      ldpc_set_error("allocation-free decode: " + std::to_string(nframes) + " frames of this code need " +
                     std::to_string(need) + " bytes of message store, the static scratch holds " +
                     std::to_string(ws_bytes) + " (use ldpc_decode_batch_device, or fewer frames per call)");
      return LDPC_ERR_ARG;
    }
    char *w = (char *)ws;
    DevBufBorrow b0(q0, w + off[WS_Q0], len[WS_Q0]), b1(msg, w + off[WS_MSG], len[WS_MSG]),
        b2(post, opts.want_llr ? w + off[WS_POST] : nullptr, len[WS_POST]), b3(hard, w + off[WS_HARD], len[WS_HARD]),
        b4(synd, w + off[WS_SYND], len[WS_SYND]), b5(sout, w + off[WS_SOUT], len[WS_SOUT]),
        b6(fail, w + off[WS_FAIL], len[WS_FAIL]), b7(done, w + off[WS_DONE], len[WS_DONE]),
        b8(iters, w + off[WS_ITERS], len[WS_ITERS]), b9(remaining, w + off[WS_REM], len[WS_REM]);
    one_stage_only = true;
    const int rc = decode(nframes, llr, dt, sy_in, max_iter, bits, iters_out, sout_out, status, llr_out, st);
    one_stage_only = false;
    return rc;
  }

  int emit_outputs(int nframes, int nslab, int max_iter, uint8_t *bits, int32_t *iters_out,
                   uint8_t *sout_out, int32_t *status, float *llr_out, hipStream_t st) {
    if (bits) {
      hipLaunchKernelGGL((egress_bits_kernel<V>), dim3((n + 63) / 64, nslab), dim3(256), 0, st,
                         hard.as<uint8_t>(), nframes, n, bits);
      LAUNCHCHK("egress_bits_kernel");
    }
    if (sout_out) {
      hipLaunchKernelGGL((egress_bits_kernel<V>), dim3((m + 63) / 64, nslab), dim3(256), 0, st,
                         sout.as<uint8_t>(), nframes, m, sout_out);
      LAUNCHCHK("egress_bits_kernel(synd)");
    }
    if (iters_out || status) {
      hipLaunchKernelGGL(egress_state_kernel, dim3((nframes + 255) / 256), dim3(256), 0, st, nframes, V,
                         max_iter, done.as<uint8_t>(), iters.as<int32_t>(), iters_out, status);
      LAUNCHCHK("egress_state_kernel");
    }
    if (llr_out) {
      hipLaunchKernelGGL((egress_post_kernel<T, ALG>), dim3((n + 31) / 32, nslab), dim3(256), 0, st,
                         post.as<vec>(), nframes, n, llr_out);
      LAUNCHCHK("egress_post_kernel");
    }
    return LDPC_OK;
  }

  int ingest_all(int nframes, int nslab, const void *llr, int dt, const uint8_t *sy_in, const uint8_t **sy,
                 hipStream_t st) {
    int rc;
    switch (dt) {
      case LDPC_DT_F64: rc = launch_ingest<double>(llr, nframes, nslab, st); break;
      case LDPC_DT_F32: rc = launch_ingest<float>(llr, nframes, nslab, st); break;
      case LDPC_DT_I16:
        if (ALG != ALG_MSQ) { ldpc_set_error("I16 input is for MINSUM_Q88"); return LDPC_ERR_ARG; }
        rc = launch_ingest<short>(llr, nframes, nslab, st);
        break;
      case LDPC_DT_I32:
        if (ALG != ALG_TAB && ALG != ALG_MSQ) { ldpc_set_error("I32 input is for SPA_TABLE_Q14 / MINSUM_Q88"); return LDPC_ERR_ARG; }
        rc = launch_ingest<int>(llr, nframes, nslab, st);
        break;
      default: ldpc_set_error("bad llr_dtype"); return LDPC_ERR_ARG;
    }
    if (rc) return rc;
    *sy = nullptr;
    if (sy_in) {
      hipLaunchKernelGGL((ingest_synd_kernel<V>), dim3((m + 63) / 64, nslab), dim3(256), 0, st, sy_in,
                         nframes, m, synd.as<uint8_t>());
      LAUNCHCHK("ingest_synd_kernel");
      *sy = synd.as<uint8_t>();
    }
    return LDPC_OK;
  }

  int launch_synd(int nslab, const uint8_t *sy, hipStream_t st) {
    const int wps = balanced_wps(nslab, (long)num_cu * 16);
    const long waves = (long)wps * nslab;
    hipLaunchKernelGGL(synd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, g, nslab,
                       hard.as<uint8_t>(), sy, sout.as<uint8_t>(), fail.as<uint32_t>());
    LAUNCHCHK("synd_kernel");
    return LDPC_OK;
  }

  // Single frame with the reference's Verilog message trace (_fixed.c:205, :245-260,
  // :288-296).  The passes are the batch kernels; after each one the frame's messages
  // are read back and the host only orders and prints them.
  int decode_traced(const void *llr, int dt, const uint8_t *sy_in, int max_iter, uint8_t *bits,
                    int32_t *iters_out, uint8_t *sout_out, int32_t *status, FILE *fp,
                    int circ) override {
    if constexpr (ALG != ALG_MSQ) {
      return Engine::decode_traced(llr, dt, sy_in, max_iter, bits, iters_out, sout_out, status, fp, circ);
    } else {
      if (!llr || max_iter < 0 || max_iter > 65000 || circ <= 0) return LDPC_ERR_ARG;
      // the c2v dump prints row_weight[circ*k] entries of every row of block k (:290): on a
      // code where a row of the block is lighter the reference reads unset pointers
      for (int k = 0; k < m / circ; k++) {
        const int w0 = h_row_ptr[circ * k + 1] - h_row_ptr[circ * k];
        for (int j = 0; j < circ; j++)
          if (h_row_ptr[circ * k + j + 1] - h_row_ptr[circ * k + j] < w0) {
            ldpc_set_error("trace: rows of one circulant block differ in weight (the reference's dump is "
                           "for quasi-cyclic codes only)");
            return LDPC_ERR_UNSUPPORTED;
          }
      }
      hipStream_t st = nullptr;
      int rc;
      if ((rc = ensure_remaining(max_iter + 3))) return rc;
      if ((rc = trace_buf.ensure(sizeof(int32_t) * (size_t)E))) return rc;
      std::vector<int32_t> hbuf(E), row_of(E), shown(n);
      for (int j = 0; j < m; j++)
        for (int e = h_row_ptr[j]; e < h_row_ptr[j + 1]; e++) row_of[e] = j;
      const int nslab = 1, nframes = 1;
      hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(256), 0, st, nslab, V, nframes, fail.as<uint32_t>(),
                         done.as<uint8_t>(), iters.as<int32_t>());
      LAUNCHCHK("init_state_kernel");
      HIPCHK(hipMemsetAsync(remaining.p, 0, sizeof(int32_t) * (max_iter + 3), st));
      HIPCHK(hipMemsetAsync(hard.p, 0, (size_t)64 * n, st));
      if (opts.want_llr) HIPCHK(hipMemsetAsync(post.p, 0, (size_t)64 * n * sizeof(vec), st));
      const uint8_t *sy = nullptr;
      if ((rc = ingest_all(nframes, nslab, llr, dt, sy_in, &sy, st))) return rc;
      auto converged_after = [&](int it, bool *yes) -> int {
        int r;
        if ((r = launch_synd(nslab, sy, st))) return r;
        if ((r = launch_mark(nslab, nframes, it, 0, it, st))) return r;
        HIPCHK(hipMemcpyAsync(h_remaining + it, remaining.as<int32_t>() + it, sizeof(int32_t),
                              hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        *yes = h_remaining[it] == 0;
        return LDPC_OK;
      };
      bool ok = false;
      if ((rc = launch_vn<true, true>(nslab, st))) return rc;       // hard(0) = (q0 < 0)
      if ((rc = converged_after(0, &ok))) return rc;
      if (ok) {
        fprintf(fp, "already cw\n");                               // :205
      } else {
        for (int t = 1; t <= max_iter && !ok; t++) {
          hipLaunchKernelGGL((trace_v2c_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, st, g, q0.as<vec>(),
                             msg.as<vec>(), 0, t == 1 ? 1 : 0, trace_buf.as<int32_t>());
          LAUNCHCHK("trace_v2c_kernel");
          if (t > 1 && (rc = launch_vn<false, true>(nslab, st))) return rc;
          HIPCHK(hipMemcpyAsync(hbuf.data(), trace_buf.p, sizeof(int32_t) * E, hipMemcpyDeviceToHost, st));
          HIPCHK(hipStreamSynchronize(st));
          std::fill(shown.begin(), shown.end(), 0);
          for (int j = 0; j < circ; j++)                             // :245-260
            for (int i = 0; i < m / circ; i++)
              for (int k = 0; k < n / circ; k++) {
                const int v = circ * k + j;
                bool hit = false;
                for (int l = h_col_ptr[v]; l < h_col_ptr[v + 1] && !hit; l++) {
                  const int r = row_of[h_edge_of[l]];
                  hit = r >= circ * i && r <= circ * (i + 1) - 1;
                }
                if (hit) fprintf(fp, "%08x\n", (unsigned)hbuf[h_edge_of[h_col_ptr[v] + shown[v]++]]);
              }
          if ((rc = launch_cn(nslab, sy, st))) return rc;
          if ((rc = launch_mark(nslab, nframes, 0, 1, max_iter + 2, st))) return rc;   // drop its syndrome test
          hipLaunchKernelGGL((trace_msg_kernel<T>), dim3((E + 255) / 256), dim3(256), 0, st, g, msg.as<vec>(), 0,
                             trace_buf.as<int32_t>());
          LAUNCHCHK("trace_msg_kernel");
          HIPCHK(hipMemcpyAsync(hbuf.data(), trace_buf.p, sizeof(int32_t) * E, hipMemcpyDeviceToHost, st));
          HIPCHK(hipStreamSynchronize(st));
          for (int j = 0; j < circ; j++)                             // :288-296
            for (int k = 0; k < m / circ; k++) {
              const int w0 = h_row_ptr[circ * k + 1] - h_row_ptr[circ * k];
              for (int i = 0; i < w0; i++) fprintf(fp, "%08x\n", (unsigned)hbuf[h_row_ptr[circ * k + j] + i]);
            }
          if ((rc = launch_vn<false, false>(nslab, st))) return rc;   // hard(t)
          if ((rc = converged_after(t, &ok))) return rc;
          last_iterations = t;
        }
      }
      return emit_outputs(nframes, nslab, max_iter, bits, iters_out, sout_out, status, nullptr, st);
    }
  }
};


// Q8.8 min-sum with the frame resident in LDS (fused_kernels.hip.h): one launch
// per decode call, no workspace in HBM besides the frame queue head.
struct FusedQ88Engine : Engine {
  ldpc_opts opts{};
  int n = 0, m = 0, rmax = 0;
  int block = 0, rounds = 0, wg_per_cu = 1, num_cu = 256;
  size_t lds = 0;
  DevBuf d_colT, d_dcT, d_counter, d_pos, d_maskT, d_timing;
  LdsSchedule sched;
  bool use_reg = false, layered = false;
  void (*kern)(FusedArgs) = nullptr;

  static bool fits(const ldpc_code &c) {
    // (fq_cmax_ok: the sums of the heaviest variable stay inside the exact range of the biased LDS words)
    return c.q == 2 && c.n <= 65535 && c.rmax <= 64 && (size_t)(c.n + FQ_NDUMMY + 4) * 4 <= 160 * 1024 &&
           (c.m + 63) / 64 <= 8 * 16 && fq_cmax_ok(c.cmax);
  }
  size_t llr_elems_per_frame() const override { return (size_t)n; }

  // Is there a register-cached kernel for this code (some split into wavefronts whose state
  // fits the register file)?  Without one the generic fused kernel re-reads the indices from
  // L2 every iteration and the two-pass HBM engine is faster (32000.2240.3.105: 13.9 against
  // 7.8 Gbit/s), so the decoder only goes fused when this holds.
  static bool reg_kernel_available(const ldpc_code &c) {
    if (c.rmax > 64) return false;
    const int need = (c.m + 63) / 64, r2 = fq_round_r2((c.rmax + 1) / 2);
    for (int W = 1; W <= 16; W++)
      if (W * 64 <= fq_reg_max_block((need + W - 1) / W, r2)) return true;
    return false;
  }

  template <typename MaskT>
  static void (*pick(int cpt))(FusedArgs) {
    switch (cpt) {
      case 1: return fused_q88_kernel<1, MaskT>;
      case 2: return fused_q88_kernel<2, MaskT>;
      case 3: return fused_q88_kernel<3, MaskT>;
      case 4: return fused_q88_kernel<4, MaskT>;
      case 5: return fused_q88_kernel<5, MaskT>;
      case 6: return fused_q88_kernel<6, MaskT>;
      default: return fused_q88_kernel<8, MaskT>;
    }
  }

  int init(const ldpc_code &c, const ldpc_opts &o) override {
    opts = o;
    n = c.n; m = c.m; rmax = c.rmax;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, o.device));
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    lds = (size_t)(n + FQ_NDUMMY + 4) * 4;
    wg_per_cu = (int)((160 * 1024) / lds);
    if (wg_per_cu > 16) wg_per_cu = 16;   // (16 wavefronts of 128 VGPRs fill a CU)
    if (wg_per_cu < 1) wg_per_cu = 1;
    // Workgroup shape.  First choice: the register-cached kernel (<= 3 checks per thread,
    // <= 16 waves per CU so that every wave gets 128 VGPRs); waves should own equal
    // numbers of checks.  Otherwise the generic kernel (indices re-read every iteration).
    const int need = (m + 63) / 64;
    const int r2 = fq_round_r2((rmax + 1) / 2);   // slot pairs per check the kernels come in
    int env_w = 0, env_g = 0, env_reg = 1;
    if (const char *e = getenv("LDPC_HIP_FUSED_WAVES")) env_w = atoi(e);
    if (const char *e = getenv("LDPC_HIP_FUSED_WG_PER_CU")) env_g = atoi(e);
    if (const char *e = getenv("LDPC_HIP_FUSED_REG")) env_reg = atoi(e);
    int bestW = 0, bestG = 0;
    double best = -1;
    use_reg = false;
    layered = o.layer_checks > 0;
    if (layered) {
      // layered schedule: one layer = one round of the generic kernel, so the workgroup has
      // layer_checks threads (ldpc_hip.h, ldpc_opts.layer_checks)
      const int L = o.layer_checks;
      if (L % 64 || L < 64 || L > 1024 || (m + L - 1) / L > 8) {
        ldpc_set_error("layer_checks must be a multiple of 64 with m/8 <= layer_checks <= 1024");
        return LDPC_ERR_UNSUPPORTED;
      }
      env_reg = 0;
      env_w = L / 64;
      env_g = std::max(1, std::min(wg_per_cu, 1024 / L));
    }
    if (env_reg && rmax <= 64) {
      for (int G = 1; G <= wg_per_cu; G++)
        for (int W = 1; W * G <= 16; W++) {
          if (env_w && W != env_w) continue;
          if (env_g && G != env_g) continue;
          const int r = (need + W - 1) / W;
          if (W * 64 > fq_reg_max_block(r, r2)) continue;
          // wavefront w owns r groups of 64 checks if w < full, else r - 1 (and runs the shorter
          // instruction stream); wavefronts go to the 4 SIMDs in turn, the busiest SIMD sets the pace
          const int full = need - (r - 1) * W;
          int load[4] = {0, 0, 0, 0};
          for (int w = 0; w < W; w++) load[w & 3] += w < full ? r : r - 1;
          const int worst = std::max(std::max(load[0], load[1]), std::max(load[2], load[3]));
          const double score = (double)need / (4.0 * worst) + 0.002 * W * G + 0.001 * G;
          if (score > best) { best = score; bestW = W; bestG = G; }
        }
      use_reg = bestW > 0;
    }

    if (!use_reg) {
      bestG = env_g >= 1 && env_g <= wg_per_cu ? env_g : std::min(wg_per_cu, 4);
      const int maxW = bestG >= 2 ? 8 : 16;
      for (int W = maxW; W >= 1; W--) {
        if (env_w && W != env_w) continue;
        const int r = (need + W - 1) / W;
        if (r > 8) break;
        const double score = (double)need / ((double)r * W) + 1e-3 * W;
        if (score > best) { best = score; bestW = W; }
      }
      if (!bestW) bestW = 16;
    }
    wg_per_cu = bestG;
    block = bestW * 64;
    rounds = (m + block - 1) / block;
    int rc;
    if (use_reg) {
      const int RM = 2 * r2;
      int optimise = 1;
      if (const char *e = getenv("LDPC_HIP_FUSED_PLACE")) optimise = atoi(e);
      if (ldpc_build_lds_schedule(c, block, RM, FQ_NDUMMY, optimise != 0, sched)) {
        ldpc_set_error("LDS schedule: code too large");
        return LDPC_ERR_UNSUPPORTED;
      }
      // A dummy word starts 2^24 above the bias and has to stay a pad ("huge, positive") through a
      // phase B in which every pad slot pointing at it adds up to +-200.0 (FQ_CAP): at most 10^7 / FQ_CAP
      // pad slots per word.  Very irregular codes (many pads) go to the two-pass engine instead.
      if ((long)sched.max_dummy_load * FQ_CAP > 10000000L) {
        ldpc_set_error("fused Q8.8 decoder: too many padded slots per dummy word for this code's row weights");
        return LDPC_ERR_UNSUPPORTED;
      }
      lds = (size_t)(sched.words + 4) * 4;
      // SH = 0 packs absolute 16-bit LDS byte addresses: the dynamic array starts after the
      // kernel's static LDS (the scratch of __syncthreads_or), so that must fit as well
      kern = fused_reg_lookup(rounds, r2, 0);
      bool small = false;
      if (kern) {
        hipFuncAttributes fa;
        HIPCHK(hipFuncGetAttributes(&fa, (const void *)kern));
        small = (size_t)(sched.words - 1) * 4 + fa.sharedSizeBytes <= 65535;
      }
      if (!small) kern = fused_reg_lookup(rounds, r2, 2);
      if ((rc = d_colT.ensure(sched.colT.size() * 2))) return rc;
      if ((rc = d_dcT.ensure(sched.dcT.size()))) return rc;
      if ((rc = d_pos.ensure(sched.pos.size() * 2))) return rc;
      if ((rc = d_maskT.ensure(sched.maskT.size() * 4))) return rc;
      HIPCHK(hipMemcpy(d_colT.p, sched.colT.data(), sched.colT.size() * 2, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_dcT.p, sched.dcT.data(), sched.dcT.size(), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_pos.p, sched.pos.data(), sched.pos.size() * 2, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_maskT.p, sched.maskT.data(), sched.maskT.size() * 4, hipMemcpyHostToDevice));
    } else {
      int cpt = rounds;
      if (cpt == 7) cpt = 8;
      kern = rmax <= 32 ? pick<uint32_t>(cpt) : pick<unsigned long long>(cpt);
      std::vector<uint16_t> colT((size_t)rounds * rmax * block, 0);
      std::vector<uint8_t> dcT((size_t)rounds * block, 0);
      for (int j = 0; j < m; j++) {
        const int r = j / block, t = j % block;
        const int d = c.row_ptr[j + 1] - c.row_ptr[j];
        dcT[(size_t)r * block + t] = (uint8_t)d;
        for (int k = 0; k < d; k++)
          colT[((size_t)r * rmax + k) * block + t] = (uint16_t)c.col_idx[c.row_ptr[j] + k];
      }
      if ((rc = d_colT.ensure(colT.size() * 2))) return rc;
      if ((rc = d_dcT.ensure(dcT.size()))) return rc;
      HIPCHK(hipMemcpy(d_colT.p, colT.data(), colT.size() * 2, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_dcT.p, dcT.data(), dcT.size(), hipMemcpyHostToDevice));
    }
    if (!kern) { ldpc_set_error("no fused kernel for this code shape"); return LDPC_ERR_UNSUPPORTED; }
    if (lds > 160 * 1024) { ldpc_set_error("fused decoder: frame does not fit in LDS"); return LDPC_ERR_UNSUPPORTED; }
    {
      // (the attribute belongs to the kernel, not to this decoder: a later decoder of a smaller
      // code with the same instantiation must not lower it, so it is always the maximum)
      hipFuncAttributes fa;
      HIPCHK(hipFuncGetAttributes(&fa, (const void *)kern));
      HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - (int)fa.sharedSizeBytes));
      if (lds + fa.sharedSizeBytes > 160 * 1024) { ldpc_set_error("fused decoder: frame does not fit in LDS"); return LDPC_ERR_UNSUPPORTED; }
    }
    if ((rc = d_counter.ensure(256))) return rc;
    if (getenv("LDPC_HIP_VERBOSE"))
      fprintf(stderr, "ldpc_hip: fused Q8.8 decoder: %d threads x %d workgroup(s) per CU, %d check(s) per thread, %s, %zu B LDS\n",
              block, wg_per_cu, rounds, use_reg ? "register-cached kernel" : layered ? "generic kernel, layered schedule" : "generic kernel", lds);
    return LDPC_OK;
  }

  int decode(int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter, uint8_t *bits,
             int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out, hipStream_t st) override {
    if (nframes <= 0) return LDPC_OK;
    if (!llr || max_iter < 0 || max_iter > 65000 || dt < 0 || dt > 3) {
      ldpc_set_error("bad arguments to the fused Q8.8 decoder");
      return LDPC_ERR_ARG;
    }
    int *counter = ws_counter ? ws_counter : d_counter.as<int>();
    HIPCHK(hipMemsetAsync(counter, 0, 4, st));
    if (sout) HIPCHK(hipMemsetAsync(sout, 0, (size_t)nframes * m, st));
    prof_cur_frames = nframes;
    FusedArgs a;
    a.colT = d_colT.as<uint16_t>();
    a.dcT = d_dcT.as<uint8_t>();
    a.n = n; a.m = m; a.rmax = rmax; a.rounds = rounds;
    a.nframes = nframes; a.max_iter = max_iter;
    a.llr = llr; a.dt = dt; a.synd = sy;
    a.bits = bits; a.iters = iters_out; a.sout = sout; a.status = status; a.llr_out = llr_out;
    a.counter = counter;
    a.pos = d_pos.as<uint16_t>();
    a.maskT = d_maskT.as<uint32_t>();
    a.words = sched.words; a.dummy_base = sched.dummy_base; a.ndummy = sched.ndummy;
    a.timing = nullptr;
    a.layered = layered ? 1 : 0;
    // (only timing builds of the kernel write it; it allocates and synchronises, so never inside the allocation-free call)
    const bool timing = getenv("LDPC_HIP_FUSED_TIMING") != nullptr && !ws_counter;
    if (timing) {
      if (int rc = d_timing.ensure(16 * 8 * sizeof(unsigned long long))) return rc;
      HIPCHK(hipMemsetAsync(d_timing.p, 0, 16 * 8 * sizeof(unsigned long long), st));
      a.timing = d_timing.as<unsigned long long>();
    }
    long grid = (long)num_cu * wg_per_cu;
    if (grid > nframes) grid = nframes;
    // short frames are taken from the queue several at a time (one atomic per chunk), while every
    // workgroup still makes at least 16 trips so that the tail stays short
    a.chunk = (int)std::max<long>(1, std::min<long>(std::min<long>(4096 / n, nframes / (grid * 16)), 16));
    if (const char *e = getenv("LDPC_HIP_FUSED_CHUNK")) a.chunk = std::max(1, atoi(e));
    prof_begin(K_FUSED, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(block), lds, st, a);
    LAUNCHCHK("fused_q88_kernel");
    prof_end(st);
    last_iterations = max_iter;
    if (timing) {
      unsigned long long h[16 * 8];
      HIPCHK(hipStreamSynchronize(st));
      HIPCHK(hipMemcpy(h, d_timing.p, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "ldpc_hip: fused kernel cycles per wavefront slot, summed over %ld workgroups: phase A | wait | phase B | wait | whole frames | stage-in | its barrier | outputs\n", grid);
      for (int w = 0; w < block / 64; w++)
        fprintf(stderr, "  wave %2d: %12llu %12llu %12llu %12llu %12llu %12llu %12llu %12llu\n", w, h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3],
                h[w * 8 + 4], h[w * 8 + 5], h[w * 8 + 6], h[w * 8 + 7]);
    }
    return LDPC_OK;
  }

  // the kernel's only device scratch is the frame queue head: one word of ws
  int *ws_counter = nullptr;
  int decode_ws(void *ws, size_t ws_bytes, int nframes, const void *llr, int dt, const uint8_t *sy, int max_iter,
                uint8_t *bits, int32_t *iters_out, uint8_t *sout, int32_t *status, float *llr_out,
                hipStream_t st) override {
    if (!ws || ws_bytes < 256) { ldpc_set_error("scratch too small for the frame queue head"); return LDPC_ERR_ARG; }
    ws_counter = (int *)ws;
    const int rc = decode(nframes, llr, dt, sy, max_iter, bits, iters_out, sout, status, llr_out, st);
    ws_counter = nullptr;
    return rc;
  }
};

}  // namespace

// ------------------------------------------------ the allocation-free entry point
// Device scratch of ldpc_decode_batch_device_noalloc(): a file-scope array instead of
// hipMalloc'ed buffers.  Every GPU the library runs on holds its own copy, at its own
// address, until the process ends; `scratch` below is only the host-side stand-in for
// hipGetSymbolAddress -- host code uses the per-device pointer it returns.
#define LDPC_NOALLOC_SCRATCH_BYTES (32u << 20)
static __device__ __align__(256) unsigned char scratch[LDPC_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[MAX_DEVICES] = {};
// One call at a time per GPU: every call on a device works in that device's one array.  `busy` is a
// try-lock held while a call is on the host; `last_ev` (created at set-up, not in the call) marks the
// end of the previous call's work on the device, so that a call on ANOTHER stream while that work is
// still running is refused instead of racing with it (the same stream orders itself).
struct ScratchUse {
  std::atomic<bool> busy{false};
  hipEvent_t last_ev = nullptr;
  hipStream_t last_stream = nullptr;
  bool have_last = false;
};
ScratchUse g_scratch_use[MAX_DEVICES];

// Address of `scratch` on `device` (which must be the current device); resolved once per device.
int scratch_on_device(int device, void **out) {
  if (device < 0 || device >= MAX_DEVICES) { ldpc_set_error("device ordinal beyond the scratch table"); return LDPC_ERR_ARG; }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[device]) {
    void *p = nullptr;
    HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) { ldpc_set_error("hipGetSymbolAddress(scratch) returned null"); return LDPC_ERR_HIP; }
    HIPCHK(hipEventCreateWithFlags(&g_scratch_use[device].last_ev, hipEventDisableTiming));
    g_scratch_ptr[device] = p;
  }
  *out = g_scratch_ptr[device];
  return LDPC_OK;
}
}  // namespace

// ---------------------------------------------------------------- decoder

struct ldpc_host_pipe;
void ldpc_host_pipe_free(ldpc_host_pipe *p);

struct ldpc_decoder {
  ldpc_code code;
  ldpc_opts opts{};
  std::unique_ptr<Engine> eng;
  std::unique_ptr<Engine> trace_eng;   // two-pass Q8.8 engine, made on the first trace call
  // staging for the host-pointer entry point
  DevBuf s_llr, s_synd, s_bits, s_sout, s_iters, s_status, s_post;   // single-frame trace call
  ldpc_host_pipe *pipe = nullptr;      // streams and double-buffered staging of ldpc_decode_batch
  int max_frames = 0;
  ~ldpc_decoder() { ldpc_host_pipe_free(pipe); }
};

extern "C" int ldpc_decoder_create(const ldpc_code *c, const ldpc_opts *opts, ldpc_decoder **out) {
  if (!c || !opts || !out) return LDPC_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    ldpc_set_error("no HIP device available: libldpc_hip has no CPU fallback");
    return LDPC_ERR_HIP;
  }
  if (opts->device < 0 || opts->device >= ndev) {
    ldpc_set_error("bad device ordinal");
    return LDPC_ERR_ARG;
  }
  HIPCHK(hipSetDevice(opts->device));
  {
    // set-up of the allocation-free entry point: this device's address of the static scratch
    void *ws = nullptr;
    int rc = scratch_on_device(opts->device, &ws);
    if (rc) return rc;
  }
  std::unique_ptr<ldpc_decoder> d(new (std::nothrow) ldpc_decoder);
  if (!d) return LDPC_ERR_NOMEM;
  d->code = *c;
  d->opts = *opts;
  if (d->opts.max_frames <= 0) d->opts.max_frames = 4096;
  d->max_frames = d->opts.max_frames;
  const bool binary = c->q == 2;
  if (opts->layer_checks != 0 && (opts->alg != LDPC_ALG_MINSUM_Q88 || !binary || opts->layer_checks < 0)) {
    ldpc_set_error("layer_checks (layered schedule) is available for MINSUM_Q88 on binary codes only");
    return LDPC_ERR_UNSUPPORTED;
  }
  const char *fe = getenv("LDPC_HIP_FUSED");
  const int fmode = fe ? atoi(fe) : 1;
  switch (opts->alg) {
    case LDPC_ALG_MINSUM_F32:
    case LDPC_ALG_MINSUM_F64:
      if (binary) {
        // cn_variant 0: frame-resident fused kernel when the frame fits in LDS (fused_float.hip);
        // any other value (or LDPC_HIP_FUSED=0) selects the two-pass HBM engine
        const bool f32 = opts->alg == LDPC_ALG_MINSUM_F32;
        if (opts->cn_variant == 0 && fmode != 0) {
          // (LDPC_HIP_CHAIN=2: the staircase decoder wherever it applies -- tests on small codes)
          if (const char *ce2 = getenv("LDPC_HIP_CHAIN"); f32 && ce2 && atoi(ce2) == 2) {
            d->eng.reset(ldpc_make_stair_f32_engine(d->opts));
            if (d->eng && d->eng->init(d->code, d->opts) == LDPC_OK) { *out = d.release(); return LDPC_OK; }
            d->eng.reset();
          }
          d->eng.reset(ldpc_make_fused_float_engine(f32 ? 4 : 8, *c));
          if (d->eng && d->eng->init(d->code, d->opts) != LDPC_OK) d->eng.reset();
          if (d->eng) { *out = d.release(); return LDPC_OK; }
          // binary32 on an accumulator-chain code whose frame fits neither LDS layout (DVB-S2 long frames): the
          // staircase decoder of stair_f32.hip; LDPC_HIP_CHAIN=0 turns it off
          const char *ce = getenv("LDPC_HIP_CHAIN");
          if (f32 && !(ce && atoi(ce) == 0)) {
            d->eng.reset(ldpc_make_stair_f32_engine(d->opts));
            if (d->eng) {
              const int rc = d->eng->init(d->code, d->opts);
              if (rc == LDPC_OK) { *out = d.release(); return LDPC_OK; }
              d->eng.reset();
              if (rc != LDPC_ERR_UNSUPPORTED) return rc;
            }
          }
        }
        if (f32) d->eng.reset(new BinaryEngine<float, ALG_MS>);
        else d->eng.reset(new BinaryEngine<double, ALG_MS>);
      }
      break;
    case LDPC_ALG_MINSUM_Q88:
      if (binary) {
        // cn_variant 0: frame-resident fused kernel when the code fits in LDS and a
        // register-cached kernel exists for its shape; any other value (or
        // LDPC_HIP_FUSED=0) selects the two-pass HBM engine
        // (LDPC_HIP_FUSED=2 also allows the generic fused kernel where no register-cached one exists)
        bool fused = opts->cn_variant == 0 && FusedQ88Engine::fits(*c);
        if (opts->layer_checks > 0) {
          // the layered schedule lives in the frame-resident decoder only
          if (!FusedQ88Engine::fits(*c)) {
            ldpc_set_error("layer_checks: the layered schedule needs a frame that fits LDS (n <= ~40000, row weight <= 64)");
            return LDPC_ERR_UNSUPPORTED;
          }
          d->eng.reset(new FusedQ88Engine);
          break;
        }
        fused = fused && fmode != 0 && (fmode == 2 || FusedQ88Engine::reg_kernel_available(*c));
        // codes whose frame is too large for that decoder but which have a staircase (DVB-S2 long
        // frames): frame-resident with the staircase variables in registers (chain_q88.hip);
        // LDPC_HIP_CHAIN=0 turns it off, =2 prefers it wherever it applies (tests on small codes)
        const char *ce = getenv("LDPC_HIP_CHAIN");
        const int cmode = ce ? atoi(ce) : 1;
        if (opts->cn_variant == 0 && fmode != 0 && ((cmode == 1 && !fused) || cmode == 2)) {
          std::unique_ptr<Engine> ch(ldpc_make_chain_q88_engine());
          const int crc = ch ? ch->init(d->code, d->opts) : LDPC_ERR_NOMEM;
          if (crc == LDPC_OK) {
            d->eng = std::move(ch);
            *out = d.release();
            return LDPC_OK;
          }
          // "not a code for this kernel" falls through to the other engines; anything else is an error
          if (crc != LDPC_ERR_UNSUPPORTED) return crc;
        }
        if (fused) d->eng.reset(new FusedQ88Engine);
        // a weight-1 check sends +-200 (min over nothing, _fixed.c:270), which
        // does not fit an int16 Q8.8 message: such codes use int32 messages
        else if (c->min_row_weight >= 2) d->eng.reset(new BinaryEngine<short, ALG_MSQ>);
        else d->eng.reset(new BinaryEngine<int, ALG_MSQ>);
      }
      break;
    case LDPC_ALG_SPA_F32: if (binary) d->eng.reset(new BinaryEngine<float, ALG_SPA>); break;
    case LDPC_ALG_SPA_F64: if (binary) d->eng.reset(new BinaryEngine<double, ALG_SPA>); break;
    case LDPC_ALG_SPA_TABLE_Q14: if (binary) d->eng.reset(new BinaryEngine<int, ALG_TAB>); break;
    case LDPC_ALG_GF2P_WHT:
    case LDPC_ALG_GFQ_DIRECT:
      d->eng.reset(ldpc_make_gf_engine(opts->alg, *c));
      break;
    default: break;
  }
  if (!d->eng) {
    ldpc_set_error("algorithm not available for this code (binary algorithms need a binary alist, "
                   "GF algorithms a non-binary one of matching field)");
    return LDPC_ERR_UNSUPPORTED;
  }
  int rc = d->eng->init(d->code, d->opts);
  if (rc == LDPC_ERR_UNSUPPORTED && opts->alg == LDPC_ALG_MINSUM_Q88 && opts->layer_checks == 0 &&
      dynamic_cast<FusedQ88Engine *>(d->eng.get())) {
    // a shape the frame-resident decoder turns down after all (schedule limits): two-pass engine
    if (c->min_row_weight >= 2) d->eng.reset(new BinaryEngine<short, ALG_MSQ>);
    else d->eng.reset(new BinaryEngine<int, ALG_MSQ>);
    rc = d->eng->init(d->code, d->opts);
  }
  if (rc) return rc;
  *out = d.release();
  return LDPC_OK;
}

extern "C" void ldpc_decoder_free(ldpc_decoder *d) {
  if (!d) return;
  (void)hipSetDevice(d->opts.device);
  (void)hipDeviceSynchronize();
  delete d;
}

extern "C" int ldpc_code_lds_schedule_stats(const ldpc_code *c, int block, int optimise, int64_t *stats) {
  if (!c || !stats || block <= 0 || block % 64 || c->rmax > 64) return LDPC_ERR_ARG;
  const int RM = 2 * ldpc::fq_round_r2((c->rmax + 1) / 2);
  LdsSchedule s;
  if (ldpc_build_lds_schedule(*c, block, RM, FQ_NDUMMY, optimise != 0, s)) return LDPC_ERR_UNSUPPORTED;
  stats[0] = s.conflicts; stats[1] = s.accesses; stats[2] = s.words; stats[3] = s.RM;
  // self-check: every edge appears exactly once in its check's slots
  std::vector<int> inv(s.words, -1);
  for (int v = 0; v < c->n; v++) {
    if (inv[s.pos[v]] >= 0) return LDPC_ERR_FORMAT;
    inv[s.pos[v]] = v;
  }
  int64_t bad = 0;
  for (int j = 0; j < c->m; j++) {
    std::vector<int> want(c->col_idx.begin() + c->row_ptr[j], c->col_idx.begin() + c->row_ptr[j + 1]), got;
    const int r = j / block, t = j % block;
    for (int k = 0; k < RM; k++) {
      const int w = s.colT[((size_t)r * RM + k) * block + t];
      const bool real = (s.maskT[((size_t)r * block + t) * s.nw + k / 32] >> ((RM < 32 ? RM : 32) - 1 - (k % 32))) & 1u;
      if (real != (w < s.dummy_base)) bad++;
      if (w < s.dummy_base) got.push_back(inv[w]);
    }
    std::sort(want.begin(), want.end());
    std::sort(got.begin(), got.end());
    if (want != got) bad++;
  }
  stats[4] = bad;
  return LDPC_OK;
}

extern "C" int ldpc_code_vpass_place_stats(const ldpc_code *c, int block, int64_t *stats) {
  if (!c || !stats || block <= 0 || block % 64 || c->rmax > 64 || c->q != 2) return LDPC_ERR_ARG;
  const int RM = std::max(2, 2 * ((c->rmax + 1) / 2));
  LdsSchedule s0, s;
  if (ldpc_build_lds_schedule(*c, block, RM, 64, true, s0)) return LDPC_ERR_UNSUPPORTED;
  s = s0;
  long before = 0, after = 0;
  ldpc_place_variable_pass(*c, s, &before, &after);
  stats[0] = before; stats[1] = after; stats[2] = (int64_t)((s.var_words + 63) / 64) * 4 * c->cmax;
  int64_t bad = 0;
  std::vector<int> inv(s.words, -1);
  for (int v = 0; v < c->n; v++) {
    if (s.pos[v] >= s.var_words || inv[s.pos[v]] >= 0) bad++;
    else inv[s.pos[v]] = v;
    if ((s.pos[v] & 31) != (s0.pos[v] & 31)) bad++;          // left its bank
  }
  for (int j = 0; j < c->m && !bad; j++) {
    std::vector<int> want(c->col_idx.begin() + c->row_ptr[j], c->col_idx.begin() + c->row_ptr[j + 1]), got;
    const int r = j / block, t = j % block;
    for (int k = 0; k < RM; k++) {
      const size_t at = ((size_t)r * RM + k) * block + t;
      if ((s.colT[at] < s.dummy_base) != (s0.colT[at] < s0.dummy_base)) bad++;
      if (s.colT[at] < s.dummy_base) got.push_back(inv[s.colT[at]]);
      else if (s.colT[at] != s0.colT[at]) bad++;              // pad slots keep their dummy word
    }
    std::sort(want.begin(), want.end());
    std::sort(got.begin(), got.end());
    if (want != got) bad++;
  }
  stats[3] = bad;
  return LDPC_OK;
}

extern "C" int ldpc_decoder_profile(ldpc_decoder *d, int enable) {
  if (!d) return LDPC_ERR_ARG;
  d->eng->profiling = enable != 0;
  return LDPC_OK;
}

extern "C" int ldpc_decoder_profile_read(ldpc_decoder *d, double *ms, int32_t *launches) {
  if (!d || !ms || !launches) return LDPC_ERR_ARG;
  HIPCHK(hipSetDevice(d->opts.device));
  double t[Engine::K_KINDS];
  int c[Engine::K_KINDS];
  d->eng->profile_read(t, c);
  for (int k = 0; k < Engine::K_KINDS; k++) { ms[k] = t[k]; launches[k] = c[k]; }
  return LDPC_OK;
}

extern "C" int ldpc_decoder_profile_frames(ldpc_decoder *d, int64_t *frame_launches) {
  if (!d || !frame_launches) return LDPC_ERR_ARG;
  for (int k = 0; k < Engine::K_KINDS; k++) frame_launches[k] = d->eng->prof_frames[k];
  return LDPC_OK;
}

extern "C" int ldpc_decoder_max_frames(const ldpc_decoder *d) { return d ? d->max_frames : LDPC_ERR_ARG; }
extern "C" int ldpc_decoder_last_iterations(const ldpc_decoder *d) {
  return d ? d->eng->last_iterations : LDPC_ERR_ARG;
}

extern "C" int ldpc_decode_batch_device(ldpc_decoder *d, int nframes, const void *llr, int llr_dtype,
                                        const uint8_t *syndrome, int max_iter, uint8_t *bits_out,
                                        int32_t *iters_out, uint8_t *synd_out, int32_t *status_out,
                                        float *llr_out, void *stream) {
  if (!d) return LDPC_ERR_ARG;
  if (nframes > d->max_frames) {
    ldpc_set_error("nframes exceeds the decoder's max_frames");
    return LDPC_ERR_ARG;
  }
  HIPCHK(hipSetDevice(d->opts.device));
  return d->eng->decode(nframes, llr, llr_dtype, syndrome, max_iter, bits_out, iters_out, synd_out,
                        status_out, llr_out, (hipStream_t)stream);
}

extern "C" int ldpc_decode_batch_device_noalloc(ldpc_decoder *d, int nframes, const void *llr, int llr_dtype,
                                                const uint8_t *syndrome, int max_iter, uint8_t *bits_out,
                                                int32_t *iters_out, uint8_t *synd_out, int32_t *status_out,
                                                float *llr_out, void *stream) {
  if (!d) return LDPC_ERR_ARG;
  if (nframes > d->max_frames) {
    ldpc_set_error("nframes exceeds the decoder's max_frames");
    return LDPC_ERR_ARG;
  }
  HIPCHK(hipSetDevice(d->opts.device));
  void *ws = nullptr;
  int rc = scratch_on_device(d->opts.device, &ws);   // resolved by ldpc_decoder_create(); a lookup here
  if (rc) return rc;
  // one call at a time on this GPU's array
  ScratchUse &use = g_scratch_use[d->opts.device];
  if (use.busy.exchange(true, std::memory_order_acquire)) {
    ldpc_set_error("allocation-free decode: another call is using this GPU's static scratch (calls on one GPU must not overlap)");
    return LDPC_ERR_ARG;
  }
  struct Release { std::atomic<bool> &b; ~Release() { b.store(false, std::memory_order_release); } } release{use.busy};
  hipStream_t st = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  if (use.have_last && use.last_stream != st && hipEventQuery(use.last_ev) == hipErrorNotReady) {
    ldpc_set_error("allocation-free decode: the previous call's work on another stream of this GPU is still running "
                   "in the static scratch (order the streams, or use ldpc_decode_batch_device)");
    return LDPC_ERR_ARG;
  }
  (void)hipGetLastError();
  // (per-kernel timing creates events on demand: not inside this call)
  const bool prof = d->eng->profiling;
  d->eng->profiling = false;
  rc = d->eng->decode_ws(ws, LDPC_NOALLOC_SCRATCH_BYTES, nframes, llr, llr_dtype, syndrome, max_iter, bits_out,
                         iters_out, synd_out, status_out, llr_out, st);
  d->eng->profiling = prof;
  // where this call's work ends on the device (a stream under graph capture has no such point yet)
  use.have_last = false;
  if (cap == hipStreamCaptureStatusNone && hipEventRecord(use.last_ev, st) == hipSuccess) {
    use.last_stream = st;
    use.have_last = true;
  }
  return rc;
}

// ---- host-pointer entry point: a chunked pipeline over PCIe
// A call is cut into chunks; chunk i+1 travels to the GPU (copy-in stream) and chunk i-1 travels
// back (copy-out stream; PCIe is full duplex) while chunk i decodes (compute stream).  Decodes stay
// serialised on the one compute stream: an engine's workspace serves one decode at a time.
// Pinned caller memory (hipHostMalloc / hipHostRegister, torch pin_memory) is copied from and to
// directly.  Pageable caller memory goes through pinned staging buffers that a few host threads
// fill and drain with memcpy (a pageable hipMemcpy moves ~16 GB/s, a fraction of the link).
namespace {

bool host_ptr_is_pinned(const void *p) {
  if (!p) return true;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

int copy_threads() {
  static const int n = [] {
    int v = 8;
    if (const char *e = getenv("LDPC_HIP_COPY_THREADS")) v = atoi(e);
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && v > hw) v = hw;
    return v < 1 ? 1 : v;
  }();
  return n;
}

// memcpy split over host threads (large blocks only)
void par_memcpy(void *dst, const void *src, size_t bytes) {
  const int T = bytes >= (8u << 20) ? copy_threads() : 1;
  if (T <= 1) { memcpy(dst, src, bytes); return; }
  std::vector<std::thread> th;
  const size_t per = ((bytes + T - 1) / T + 4095) & ~(size_t)4095;
  for (int t = 0; t < T; t++) {
    const size_t lo = (size_t)t * per;
    if (lo >= bytes) break;
    const size_t len = std::min(per, bytes - lo);
    th.emplace_back([=] { memcpy((char *)dst + lo, (const char *)src + lo, len); });
  }
  for (auto &x : th) x.join();
}

struct PinBuf {
  void *p = nullptr;
  size_t bytes = 0;
  PinBuf() = default;
  PinBuf(const PinBuf &) = delete;
  PinBuf &operator=(const PinBuf &) = delete;
  ~PinBuf() { if (p) (void)hipHostFree(p); }
  int ensure(size_t want) {
    if (want <= bytes) return LDPC_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; bytes = 0;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
      p = nullptr;
      ldpc_set_error("hipHostMalloc of " + std::to_string(want) + " bytes failed");
      return LDPC_ERR_NOMEM;
    }
    bytes = want;
    return LDPC_OK;
  }
};

}  // namespace

struct ldpc_host_pipe {
  enum { IN_LLR, IN_SYND, OUT_BITS, OUT_SOUT, OUT_ITERS, OUT_STATUS, OUT_POST, NBUF };
  struct Slot {
    DevBuf dev[NBUF];
    PinBuf pin[NBUF];
    hipEvent_t ev_in = nullptr, ev_cmp = nullptr, ev_out = nullptr;
    int frames = 0, first = 0;      // chunk in flight (frames == 0: none)
  } slot[2];
  hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
  bool ready = false;
  int init() {
    if (ready) return LDPC_OK;
    HIPCHK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&s_cmp, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    for (auto &sl : slot) {
      HIPCHK(hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&sl.ev_cmp, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
    }
    ready = true;
    return LDPC_OK;
  }
  ~ldpc_host_pipe() {
    for (auto &sl : slot) {
      if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
      if (sl.ev_cmp) (void)hipEventDestroy(sl.ev_cmp);
      if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
    }
    if (s_in) (void)hipStreamDestroy(s_in);
    if (s_cmp) (void)hipStreamDestroy(s_cmp);
    if (s_out) (void)hipStreamDestroy(s_out);
  }
};

void ldpc_host_pipe_free(ldpc_host_pipe *p) { delete p; }

extern "C" int ldpc_decode_batch(ldpc_decoder *d, int nframes, const void *llr, int llr_dtype,
                                 const uint8_t *syndrome, int max_iter, uint8_t *bits_out,
                                 int32_t *iters_out, uint8_t *synd_out, int32_t *status_out,
                                 float *llr_out) {
  if (!d || !llr || nframes < 0) return LDPC_ERR_ARG;
  const size_t es = dtype_size(llr_dtype);
  if (!es) return LDPC_ERR_ARG;
  HIPCHK(hipSetDevice(d->opts.device));
  if (!d->pipe) d->pipe = new (std::nothrow) ldpc_host_pipe;
  if (!d->pipe) return LDPC_ERR_NOMEM;
  ldpc_host_pipe &P = *d->pipe;
  int rc;
  if ((rc = P.init())) return rc;
  typedef ldpc_host_pipe HP;
  const size_t per = d->eng->llr_elems_per_frame();
  const int n = d->code.n, m = d->code.m, mf = d->max_frames;

  // bytes per frame of each stream of data, the caller's pointer, and whether it is pinned
  const size_t fb[HP::NBUF] = {per * es, (size_t)m, (size_t)n, (size_t)m, 4, 4, (size_t)n * 4};
  void *const user[HP::NBUF] = {(void *)llr, (void *)syndrome, bits_out, synd_out, iters_out, status_out, llr_out};
  bool pinned[HP::NBUF];
  for (int k = 0; k < HP::NBUF; k++) pinned[k] = host_ptr_is_pinned(user[k]);

  // chunk: up to ~256 MB of input (measured on PEG-16000 Q8.8, 65 536 frames on pinned buffers:
  // 18.8 Gbit/s with 32 and 128 MB chunks, 20.9 with 256 MB), at least four chunks per call where
  // the call is large enough to overlap anything, a multiple of 256 frames, at most the decoder's
  // pass size; small calls are one chunk
  long chunk = (long)((256u << 20) / (fb[HP::IN_LLR] ? fb[HP::IN_LLR] : 1));
  if (chunk > (nframes + 3) / 4) chunk = (nframes + 3) / 4;
  if (const char *e = getenv("LDPC_HIP_HOST_CHUNK")) { const long v = atol(e); if (v > 0) chunk = v; }
  chunk = (chunk + 255) / 256 * 256;
  if (chunk < 256) chunk = 256;
  if (chunk > mf) chunk = mf;
  if (chunk > nframes) chunk = nframes > 0 ? nframes : 1;

  for (auto &sl : P.slot) {
    sl.frames = 0;
    for (int k = 0; k < HP::NBUF; k++) {
      if (!user[k]) continue;
      if ((rc = sl.dev[k].ensure((size_t)chunk * fb[k]))) return rc;
      if (!pinned[k] && (rc = sl.pin[k].ensure((size_t)chunk * fb[k]))) return rc;
    }
  }

  int worst = 0;
  // hand a finished chunk's outputs to the caller (pageable outputs: out of the staging buffers)
  auto retire = [&](HP::Slot &sl) -> int {
    if (!sl.frames) return LDPC_OK;
    HIPCHK(hipEventSynchronize(sl.ev_out));
    for (int k = HP::OUT_BITS; k < HP::NBUF; k++)
      if (user[k] && !pinned[k])
        par_memcpy((char *)user[k] + (size_t)sl.first * fb[k], sl.pin[k].p, (size_t)sl.frames * fb[k]);
    sl.frames = 0;
    return LDPC_OK;
  };
  auto fail = [&](int code) {   // leave nothing in flight behind an error
    (void)hipStreamSynchronize(P.s_in); (void)hipStreamSynchronize(P.s_cmp); (void)hipStreamSynchronize(P.s_out);
    for (auto &sl : P.slot) sl.frames = 0;
    return code;
  };

  int i = 0;
  for (int f0 = 0; f0 < nframes; f0 += (int)chunk, i++) {
    HP::Slot &sl = P.slot[i & 1];
    const int nf = nframes - f0 < chunk ? nframes - f0 : (int)chunk;
    if ((rc = retire(sl))) return fail(rc);                  // chunk i-2 has left this slot
    // copy-in
    for (int k = HP::IN_LLR; k <= HP::IN_SYND; k++) {
      if (!user[k]) continue;
      const char *src = (const char *)user[k] + (size_t)f0 * fb[k];
      if (!pinned[k]) { par_memcpy(sl.pin[k].p, src, (size_t)nf * fb[k]); src = (const char *)sl.pin[k].p; }
      if (hipMemcpyAsync(sl.dev[k].p, src, (size_t)nf * fb[k], hipMemcpyHostToDevice, P.s_in) != hipSuccess) {
        ldpc_set_error("hipMemcpyAsync (host to device) failed");
        return fail(LDPC_ERR_HIP);
      }
    }
    if (hipEventRecord(sl.ev_in, P.s_in) != hipSuccess || hipStreamWaitEvent(P.s_cmp, sl.ev_in, 0) != hipSuccess)
      return fail(LDPC_ERR_HIP);
    // decode (engines that poll for early exit return when the chunk is done; the others at once)
    rc = d->eng->decode(nf, sl.dev[HP::IN_LLR].p, llr_dtype, syndrome ? sl.dev[HP::IN_SYND].as<uint8_t>() : nullptr,
                        max_iter, bits_out ? sl.dev[HP::OUT_BITS].as<uint8_t>() : nullptr,
                        iters_out ? sl.dev[HP::OUT_ITERS].as<int32_t>() : nullptr,
                        synd_out ? sl.dev[HP::OUT_SOUT].as<uint8_t>() : nullptr,
                        status_out ? sl.dev[HP::OUT_STATUS].as<int32_t>() : nullptr,
                        llr_out ? sl.dev[HP::OUT_POST].as<float>() : nullptr, P.s_cmp);
    if (rc) return fail(rc);
    if (d->eng->last_iterations > worst) worst = d->eng->last_iterations;
    if (hipEventRecord(sl.ev_cmp, P.s_cmp) != hipSuccess || hipStreamWaitEvent(P.s_out, sl.ev_cmp, 0) != hipSuccess)
      return fail(LDPC_ERR_HIP);
    // copy-out
    for (int k = HP::OUT_BITS; k < HP::NBUF; k++) {
      if (!user[k]) continue;
      void *dst = pinned[k] ? (void *)((char *)user[k] + (size_t)f0 * fb[k]) : sl.pin[k].p;
      if (hipMemcpyAsync(dst, sl.dev[k].p, (size_t)nf * fb[k], hipMemcpyDeviceToHost, P.s_out) != hipSuccess) {
        ldpc_set_error("hipMemcpyAsync (device to host) failed");
        return fail(LDPC_ERR_HIP);
      }
    }
    if (hipEventRecord(sl.ev_out, P.s_out) != hipSuccess) return fail(LDPC_ERR_HIP);
    sl.frames = nf;
    sl.first = f0;
  }
  // drain, oldest chunk first
  if ((rc = retire(P.slot[i & 1]))) return fail(rc);
  if ((rc = retire(P.slot[(i + 1) & 1]))) return fail(rc);
  d->eng->last_iterations = worst;
  return LDPC_OK;
}

extern "C" int ldpc_decode_trace_q88(ldpc_decoder *d, const void *llr, int llr_dtype,
                                     const uint8_t *syndrome, int max_iter, uint8_t *bits_out,
                                     int32_t *iters_out, uint8_t *synd_out, int32_t *status_out,
                                     const char *trace_path, int circ_size) {
  if (!d || !llr || !trace_path || circ_size <= 0) return LDPC_ERR_ARG;
  const size_t es = dtype_size(llr_dtype);
  if (!es) return LDPC_ERR_ARG;
  HIPCHK(hipSetDevice(d->opts.device));
  const int n = d->code.n, m = d->code.m;
  int rc;
  if ((rc = d->s_llr.ensure((size_t)d->max_frames * d->eng->llr_elems_per_frame() * es))) return rc;
  if ((rc = d->s_synd.ensure((size_t)d->max_frames * m))) return rc;
  if ((rc = d->s_bits.ensure((size_t)d->max_frames * n))) return rc;
  if ((rc = d->s_sout.ensure((size_t)d->max_frames * m))) return rc;
  if ((rc = d->s_iters.ensure((size_t)d->max_frames * 4))) return rc;
  if ((rc = d->s_status.ensure((size_t)d->max_frames * 4))) return rc;
  HIPCHK(hipMemcpy(d->s_llr.p, llr, (size_t)n * es, hipMemcpyHostToDevice));
  if (syndrome) HIPCHK(hipMemcpy(d->s_synd.p, syndrome, (size_t)m, hipMemcpyHostToDevice));
  FILE *fp = fopen(trace_path, "wt");   // test_code_min_sum_B_fixed.c:454
  if (!fp) {
    ldpc_set_error(std::string("cannot open trace file ") + trace_path);
    return LDPC_ERR_IO;
  }
  Engine *te = d->eng.get();
  if (d->opts.alg == LDPC_ALG_MINSUM_Q88 && dynamic_cast<FusedQ88Engine *>(te)) {
    if (!d->trace_eng) {
      std::unique_ptr<Engine> e;
      if (d->code.min_row_weight >= 2) e.reset(new BinaryEngine<short, ALG_MSQ>);
      else e.reset(new BinaryEngine<int, ALG_MSQ>);
      ldpc_opts o = d->opts;
      o.max_frames = 64;
      o.want_llr = 0;
      if ((rc = e->init(d->code, o))) { fclose(fp); return rc; }
      d->trace_eng = std::move(e);
    }
    te = d->trace_eng.get();
  }
  rc = te->decode_traced(d->s_llr.p, llr_dtype, syndrome ? d->s_synd.as<uint8_t>() : nullptr, max_iter,
                             d->s_bits.as<uint8_t>(), d->s_iters.as<int32_t>(), d->s_sout.as<uint8_t>(),
                             d->s_status.as<int32_t>(), fp, circ_size);
  fclose(fp);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  if (bits_out) HIPCHK(hipMemcpy(bits_out, d->s_bits.p, (size_t)n, hipMemcpyDeviceToHost));
  if (synd_out) HIPCHK(hipMemcpy(synd_out, d->s_sout.p, (size_t)m, hipMemcpyDeviceToHost));
  if (iters_out) HIPCHK(hipMemcpy(iters_out, d->s_iters.p, 4, hipMemcpyDeviceToHost));
  if (status_out) HIPCHK(hipMemcpy(status_out, d->s_status.p, 4, hipMemcpyDeviceToHost));
  return LDPC_OK;
}
