// lds_rates.hip -- micro-benchmark: per-CU throughput of the LDS operations the fused
// decoder is built from (conflict-free addresses).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(1024) void k(int *out, int iters) {
  __shared__ int lds[16384];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += blockDim.x) lds[i] = i;
  __syncthreads();
  int acc = 0;
  unsigned a0 = (tid * 4) & 0xffff;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const unsigned off = (a0 + j * 4096 + it * 256) & 0xfffc;
      int *p = (int *)((char *)lds + off);
      if (MODE == 0) acc += *p;                         // ds_read_b32
      else if (MODE == 1) *p = it + j;                  // ds_write_b32
      else if (MODE == 2) atomicAdd(p, it + j);         // ds_add_u32
      else if (MODE == 3) *p = *p + it + j;             // read + add + write
      else if (MODE == 4) atomicAdd((float *)p, 1.0f);  // ds_add_f32
      else if (MODE == 5) acc += atomicAdd(p, it + j);  // ds_add_rtn_u32
    }
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + tid] = acc + lds[tid];
}

template <int MODE> void run(const char *name, int threads) {
  int *out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, threads>>>(out, 10);
  hipEventRecord(a);
  k<MODE><<<256, threads>>>(out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double wave_instr_per_cu = (double)iters * 16 * (threads / 64);
  printf("%-22s threads %4d: %.3f ms, %.1f ns per wave-instr per CU (~%.1f cycles @2.4GHz)\n", name, threads, ms,
         ms * 1e6 / wave_instr_per_cu, ms * 1e6 / wave_instr_per_cu * 2.4);
  hipFree(out);
}

int main() {
  for (int t : {256, 1024}) {
    if (t == 256) { run<0>("ds_read_b32", 256); run<1>("ds_write_b32", 256); run<2>("ds_add_u32", 256); run<3>("read+add+write", 256); run<4>("ds_add_f32", 256); run<5>("ds_add_rtn_u32", 256); }
    else { run<0>("ds_read_b32", 1024); run<1>("ds_write_b32", 1024); run<2>("ds_add_u32", 1024); run<3>("read+add+write", 1024); run<4>("ds_add_f32", 1024); run<5>("ds_add_rtn_u32", 1024); }
  }
  return 0;
}
