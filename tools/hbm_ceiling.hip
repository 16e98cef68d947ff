// hbm_ceiling.hip -- what a hand-written streaming kernel reaches on this GPU's HBM, for the access
// shapes of the two-pass decoders: read-only, write-only, copy (read A, write B) and in-place
// read-modify-write, 16 bytes per lane (global_load / global_store_dwordx4), grid-stride loops with
// the grid sized to the CUs.  Replaces round 1's torch elementwise kernels (tools/hbm_ceiling.py),
// which undersold the ceiling the two-pass engine is priced against.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_ceiling tools/hbm_ceiling.hip && tools/hbm_ceiling
// Prints GB/s of bytes moved (read + written).  UNROLL independent 16-byte accesses in flight per
// lane; variants with non-temporal loads / stores (__builtin_nontemporal_*: glc slc / nt bits).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { READ = 0, WRITE = 1, COPY = 2, RMW = 3 };

template <int MODE, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const u4 *__restrict__ a, u4 *__restrict__ b, size_t n16, unsigned *sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u4 v[UNROLL];
    if (MODE != WRITE) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        const u4 *p = (MODE == RMW ? (const u4 *)b : a) + i + u * stride;
        v[u] = NT ? __builtin_nontemporal_load(p) : *p;
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if (MODE == READ) acc ^= v[u];
      else {
        u4 w = MODE == WRITE ? (u4){(unsigned)i, 1u, 2u, 3u} : (MODE == RMW ? v[u] + 1u : v[u]);
        if (NT) __builtin_nontemporal_store(w, b + i + u * stride);
        else b[i + u * stride] = w;
      }
    }
  }
  for (; i < n16; i += stride) {            // tail
    if (MODE == READ) acc ^= a[i];
    else if (MODE == WRITE) b[i] = (u4){(unsigned)i, 1u, 2u, 3u};
    else if (MODE == COPY) b[i] = a[i];
    else b[i] = b[i] + 1u;
  }
  if (MODE == READ && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;   // keeps the loads alive
}

// the same traffic with every workgroup walking ONE contiguous region (what an elementwise library
// kernel does, and what a wavefront of the decoders does with its 1-KiB rows) instead of a grid stride
template <int MODE, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void blocked_kernel(const u4 *__restrict__ a, u4 *__restrict__ b, size_t n16, unsigned *sink) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
  u4 acc = {0, 0, 0, 0};
  size_t i = lo + threadIdx.x;
  for (; i + (UNROLL - 1) * 256 < hi; i += UNROLL * 256) {
    u4 v[UNROLL];
    if (MODE != WRITE) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        const u4 *p = (MODE == RMW ? (const u4 *)b : a) + i + u * 256;
        v[u] = NT ? __builtin_nontemporal_load(p) : *p;
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if (MODE == READ) acc ^= v[u];
      else {
        u4 w = MODE == WRITE ? (u4){(unsigned)i, 1u, 2u, 3u} : (MODE == RMW ? v[u] + 1u : v[u]);
        if (NT) __builtin_nontemporal_store(w, b + i + u * 256);
        else b[i + u * 256] = w;
      }
    }
  }
  for (; i < hi; i += 256) {
    if (MODE == READ) acc ^= a[i];
    else if (MODE == WRITE) b[i] = (u4){(unsigned)i, 1u, 2u, 3u};
    else if (MODE == COPY) b[i] = a[i];
    else b[i] = b[i] + 1u;
  }
  if (MODE == READ && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

// the two-pass decoders' own shape: every wavefront (ROWL = 64 lanes) or workgroup (ROWL = 256) reads NR rows of
// ROWL x 16 bytes that lie far apart (a variable's edges), and writes them back -- rows in a scrambled order, so
// that neighbouring wavefronts do not walk neighbouring rows
template <int ROWL, int NR>
__global__ __launch_bounds__(256) void rows_kernel(u4 *__restrict__ b, size_t nrows) {
  const size_t units = nrows / NR;
  const size_t nunit = ROWL == 64 ? (size_t)gridDim.x * 4 : gridDim.x;
  size_t unit = ROWL == 64 ? (size_t)blockIdx.x * 4 + (threadIdx.x >> 6) : blockIdx.x;
  const int lane = ROWL == 64 ? (threadIdx.x & 63) : threadIdx.x;
  for (; unit < units; unit += nunit) {
    u4 v[NR];
    size_t row[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
      row[k] = ((unit * NR + k) * 2654435761ull) % nrows;     // (odd multiplier: a permutation when nrows is a power of two)
      v[k] = b[row[k] * ROWL + lane];
    }
#pragma unroll
    for (int k = 0; k < NR; k++) b[row[k] * ROWL + lane] = v[k] + 1u;
  }
}

template <int ROWL, int NR>
static double run_rows(u4 *b, size_t bytes, int blocks) {
  size_t nrows = 1;
  while (nrows * 2 * ROWL * 16 <= bytes) nrows *= 2;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((rows_kernel<ROWL, NR>), dim3(blocks), dim3(256), 0, 0, b, nrows);
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((rows_kernel<ROWL, NR>), dim3(blocks), dim3(256), 0, 0, b, nrows);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return (double)nrows * ROWL * 16 * 2 * reps / (ms * 1e-3) / 1e9;
}

template <int MODE, int UNROLL, bool NT, bool BLOCKED = false>
static double run(const u4 *a, u4 *b, size_t bytes, int blocks, unsigned *sink) {
  const size_t n16 = bytes / 16;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto kern = BLOCKED ? blocked_kernel<MODE, UNROLL, NT> : stream_kernel<MODE, UNROLL, NT>;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, a, b, n16, sink);
  CHECK(hipDeviceSynchronize());
  const int reps = 10;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, a, b, n16, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double moved = (double)bytes * (MODE == COPY || MODE == RMW ? 2 : 1) * reps;
  return moved / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv) {
  const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 4096) << 20;   // MiB per buffer, default 4 GiB
  u4 *a, *b;
  unsigned *sink;
  CHECK(hipMalloc(&a, bytes));
  CHECK(hipMalloc(&b, bytes));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(a, 1, bytes));
  CHECK(hipMemset(b, 2, bytes));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, buffers of %zu MiB, 16 B per lane, 256-thread workgroups, GB/s of bytes moved (read + written)\n", prop.name, cus,
         bytes >> 20);
  printf("%-10s %6s %6s %3s %10s\n", "shape", "wg/CU", "unroll", "nt", "GB/s");
  double best[4] = {0, 0, 0, 0};
  const char *names[4] = {"read", "write", "copy", "in-place"};
#define ROW(MODE, U, NT)                                                                         \
  do {                                                                                           \
    const double g = run<MODE, U, NT>(a, b, bytes, cus * wg, sink);                              \
    printf("%-10s %6d %6d %3d %10.1f\n", names[MODE], wg, U, (int)NT, g);                        \
    if (g > best[MODE]) best[MODE] = g;                                                          \
  } while (0)
  for (int wg : {4, 8, 16, 32}) {
    ROW(READ, 4, false); ROW(READ, 8, false); ROW(READ, 4, true);
    ROW(WRITE, 4, false); ROW(WRITE, 4, true);
    ROW(COPY, 2, false); ROW(COPY, 4, false); ROW(COPY, 8, false); ROW(COPY, 4, true);
    ROW(RMW, 2, false); ROW(RMW, 4, false); ROW(RMW, 8, false); ROW(RMW, 4, true);
  }
  printf("# one contiguous region per workgroup (blocks = wg/CU x CUs x 16):\n");
#define BROW(MODE, U, NT)                                                                        \
  do {                                                                                           \
    const double g = run<MODE, U, NT, true>(a, b, bytes, cus * wg * 16, sink);                   \
    printf("%-10s %6d %6d %3d %10.1f  blocked\n", names[MODE], wg, U, (int)NT, g);               \
    if (g > best[MODE]) best[MODE] = g;                                                          \
  } while (0)
  for (int wg : {8, 32, 128}) {
    BROW(READ, 4, false); BROW(READ, 4, true); BROW(WRITE, 4, false); BROW(COPY, 4, false); BROW(COPY, 4, true);
    BROW(RMW, 4, false); BROW(RMW, 4, true); BROW(RMW, 1, false);
  }
  printf("# scattered rows, read then written back in place (the two-pass decoders' access shape): rows of 1 KiB per wavefront / 4 KiB per workgroup, 3 or 8 rows in flight\n");
  for (int wg : {8, 16, 32}) {
    printf("rows 1 KiB x3  %6d %10.1f\n", wg, run_rows<64, 3>(b, bytes, cus * wg));
    printf("rows 1 KiB x8  %6d %10.1f\n", wg, run_rows<64, 8>(b, bytes, cus * wg));
    printf("rows 4 KiB x3  %6d %10.1f\n", wg, run_rows<256, 3>(b, bytes, cus * wg));
    printf("rows 4 KiB x8  %6d %10.1f\n", wg, run_rows<256, 8>(b, bytes, cus * wg));
  }
  printf("# best: read %.0f, write %.0f, copy %.0f, in-place %.0f GB/s = %.2f / %.2f / %.2f / %.2f of the 8 TB/s peak\n", best[0], best[1],
         best[2], best[3], best[0] / 8000, best[1] / 8000, best[2] / 8000, best[3] / 8000);
  return 0;
}
