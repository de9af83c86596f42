// Cost of the dropout bit generators per 128 random bits (one call per 8 elements), VALU only.
//   hipcc --offload-arch=gfx950 -O3 -w -I../../vit-vs-raw-iq_amd/csrc -I../../include -o rng_probe rng_probe.hip
#include "common.h"
#include <stdio.h>
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, uint64_t seed) {
  uint32_t acc = 0;
  const uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * (uint64_t)iters;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      IqRng r = {seed, 3, 5, nullptr};
      acc ^= dropout_keep8(r, base + i, 6554);
    } else {
      const uint32_t c = (uint32_t)(base + i) * 4u, key = (uint32_t)seed;
      uint32_t w0 = lowbias32(c ^ key), w1 = lowbias32((c + 1) ^ key), w2 = lowbias32((c + 2) ^ key), w3 = lowbias32((c + 3) ^ key);
      uint32_t keep = 0;
      keep |= ((w0 & 0xffff) >= 6554u) << 0; keep |= ((w0 >> 16) >= 6554u) << 1;
      keep |= ((w1 & 0xffff) >= 6554u) << 2; keep |= ((w1 >> 16) >= 6554u) << 3;
      keep |= ((w2 & 0xffff) >= 6554u) << 4; keep |= ((w2 >> 16) >= 6554u) << 5;
      keep |= ((w3 & 0xffff) >= 6554u) << 6; keep |= ((w3 >> 16) >= 6554u) << 7;
      acc ^= keep;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int KIND> void run(const char* name, unsigned* d) {
  const int grid = 2048, iters = 2048;
  k<KIND><<<grid, 256>>>(d, iters, 1234);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); k<KIND><<<grid, 256>>>(d, iters, 1234); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double calls = (double)grid * 256 * iters;           // per-lane calls
  const double simd_clk = ms * 1e-3 * 2.4e9 * 1024;           // SIMD-cycles available at 2.4 GHz
  printf("%-28s %8.1f us  %6.1f SIMD-clk per wave-call (64 lanes x 8 decisions)\n", name, ms * 1e3, simd_clk / (calls / 64));
}
int main() { unsigned* d; hipMalloc(&d, 2048 * 256 * 4); run<0>("Philox4x32-7 (dropout_keep8)", d); run<1>("4 x lowbias32 hash", d); return 0; }
