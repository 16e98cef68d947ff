#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) {
  const int v = threadIdx.x + 100;
  o[threadIdx.x] = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
  o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);
}
int main() {
  int *d, h[128];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("wave_shr:1 "); for (int i = 0; i < 64; i++) printf("%d ", h[i]); printf("\n");
  printf("wave_shl:1 "); for (int i = 0; i < 64; i++) printf("%d ", h[64 + i]); printf("\n");
  return 0;
}
