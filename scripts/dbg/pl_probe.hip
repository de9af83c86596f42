#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  unsigned v = threadIdx.x, s = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(v, s, false, false);
  o[threadIdx.x] = r[0]; o[threadIdx.x + 64] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int row = 0; row < 4; ++row) printf("r[0] row%d: lane%d -> %u ... ; r[1] row%d: lane%d -> %u\n", row, row*16, h[row*16], row, row*16, h[64 + row*16]);
  return 0;
}
