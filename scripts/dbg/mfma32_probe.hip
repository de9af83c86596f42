// Verify the operand / accumulator lane layouts of v_mfma_f32_32x32x16_bf16 and the semantics of
// v_permlane32_swap on gfx950 against the formulas the kernels assume.
//   hipcc --offload-arch=gfx950 -O3 -w -o mfma32_probe mfma32_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(const float* A, const float* B, float* D, unsigned* sw) {
  // assumed: A operand: lane l holds A[m = l%32][k = 8*(l/32) + e], e = 0..7;  B operand: B[k = 8*(l/32) + e][n = l%32]
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[(l % 32) * 16 + 8 * (l / 32) + e]; b[e] = (__bf16)B[(8 * (l / 32) + e) * 32 + (l % 32)]; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  // assumed: D[m = 8*(r/4) + 4*(l/32) + (r%4)][n = l%32]
  for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (l / 32) + (r % 4)) * 32 + (l % 32)] = acc[r];
  unsigned v = 1000 + l, s = 2000 + l;
  auto res = __builtin_amdgcn_permlane32_swap(v, s, false, false);
  sw[l] = res[0]; sw[64 + l] = res[1];
}
int main() {
  float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
  for (int i = 0; i < 32 * 16; ++i) hA[i] = (float)((i * 7) % 13 - 6);
  for (int i = 0; i < 16 * 32; ++i) hB[i] = (float)((i * 5) % 11 - 5);
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[m * 16 + kk] * hB[kk * 32 + n]; ref[m * 32 + n] = s; }
  float *dA, *dB, *dD; unsigned* dS; unsigned hS[128];
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD)); hipMalloc(&dS, sizeof(hS));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD, dS);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost); hipMemcpy(hS, dS, sizeof(hS), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) if (fabsf(hD[i] - ref[i]) > 1e-3f) ++bad;
  printf("mfma_f32_32x32x16_bf16 layout assumption: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  printf("permlane32_swap(v=1000+l, s=2000+l): r0[0]=%u r0[31]=%u r0[32]=%u r0[63]=%u | r1[0]=%u r1[31]=%u r1[32]=%u r1[63]=%u\n",
         hS[0], hS[31], hS[32], hS[63], hS[64], hS[95], hS[96], hS[127]);
  return 0;
}
