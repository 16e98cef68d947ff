// glds_probe.hip -- does __builtin_amdgcn_global_load_lds(16 B) land where the fused decoder's
// LLR prefetch expects it?  One 1024-thread block copies `bytes` (multiple of 16) from global
// to LDS with a partially active last wave-instruction, then copies LDS back out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ __launch_bounds__(1024) void k(const unsigned char *src, unsigned char *dst, int bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n16 = bytes / 16;
  for (int c = wave; c * 64 < n16; c += 16) {       // one KiB per wave-instruction
    const int i = c * 64 + lane;
    if (i < n16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)i * 16),
                                       (lds_void *)(lds + 4096 + (size_t)c * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < bytes; i += 1024) dst[i] = lds[4096 + i];
}
int main() {
  const int bytes = 32000;
  std::vector<unsigned char> h(bytes), o(bytes);
  for (int i = 0; i < bytes; i++) h[i] = (unsigned char)(i * 7 + i / 251);
  unsigned char *s, *d;
  hipMalloc(&s, bytes); hipMalloc(&d, bytes);
  hipMemcpy(s, h.data(), bytes, hipMemcpyHostToDevice);
  hipMemset(d, 0, bytes);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  k<<<1, 1024, 4096 + 32768>>>(s, d, bytes);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(o.data(), d, bytes, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < bytes; i++) bad += o[i] != h[i];
  printf("glds probe: %s, %d of %d bytes differ\n", hipGetErrorString(e), bad, bytes);
  return bad != 0;
}
