// Does the LDS ignore address bits above its size?  (If it did, the low half of a packed pair of
// 16-bit LDS addresses could be used as an address as it stands.)  Prints what a ds_read_b32 returns
// for addresses with extra high bits set, next to the in-range value.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) int lds_int;
__global__ void probe(int *out) {
  extern __shared__ int buf[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = 1000 + i;
  __syncthreads();
  const unsigned base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char *)buf;
  const unsigned a = base + threadIdx.x * 4;
  const unsigned hi[6] = {0u, 1u << 16, 1u << 18, 1u << 20, 1u << 24, 1u << 31};
  for (int k = 0; k < 6; k++) out[k * 64 + threadIdx.x] = *(lds_int *)(unsigned long)(a | hi[k]);
}
int main() {
  int *d, h[6 * 64];
  hipMalloc(&d, sizeof(h));
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 65536, 0, d);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int k = 0; k < 6; k++) printf("high bits pattern %d: lane 0 -> %d, lane 5 -> %d\n", k, h[k * 64], h[k * 64 + 5]);
  return 0;
}
