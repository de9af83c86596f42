// MFMA issue rate of the two bf16 shapes on gfx950, operands in registers, no memory traffic in the loop.
// Replaces the MFMA part of clk_probe.hip, whose 16x16x32 loop the compiler had polluted with v_accvgpr copies (the
// accumulators were split over VGPRs and AGPRs and rotated every iteration: its "16x16x32 tops out at 1.4-1.6 PFLOP/s"
// was an artefact).  Here: __launch_bounds__(256, 2) keeps the accumulators in VGPRs (checked in the ISA: the loop is
// MFMAs, one s_add, one compare, one branch), same output tile per wave for both shapes (64 x 64 fp32 = 64 VGPRs),
// data = zeros or random bf16 (the chip holds a lower clock on random data: MI355X_MICROARCH.md, DVFS give-back).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe mfma_rate_probe.hip && ./mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256, 2) void k16(const bf16x8* __restrict__ src, unsigned long long* out, float* sink, int iters) {
  const bf16x8 a = src[threadIdx.x], b = src[256 + threadIdx.x];
  f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {}, c8 = {}, c9 = {}, c10 = {}, c11 = {}, c12 = {}, c13 = {}, c14 = {}, c15 = {};
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#define M16(c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
    M16(c0); M16(c1); M16(c2); M16(c3); M16(c4); M16(c5); M16(c6); M16(c7);
    M16(c8); M16(c9); M16(c10); M16(c11); M16(c12); M16(c13); M16(c14); M16(c15);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  const f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + c8 + c9 + c10 + c11 + c12 + c13 + c14 + c15;
  if (s[0] == 12345.678f) sink[0] = s[1];
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256, 2) void k32(const bf16x8* __restrict__ src, unsigned long long* out, float* sink, int iters) {
  const bf16x8 a = src[threadIdx.x], b = src[256 + threadIdx.x];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#define M32(c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
    M32(c0); M32(c1); M32(c2); M32(c3); M32(c0); M32(c1); M32(c2); M32(c3);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  const f32x16 s = c0 + c1 + c2 + c3;
  if (s[0] == 12345.678f) sink[0] = s[1];
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  unsigned short h[512 * 8];
  bf16x8* src; unsigned long long* out; float* sink;
  hipMalloc(&src, sizeof(h)); hipMalloc(&out, 4096 * 16); hipMalloc(&sink, 16);
  unsigned long long ho[4096 * 2];
  const int iters = 40000;
  for (int data = 0; data < 2; ++data) {
    srand(1);
    for (int i = 0; i < 512 * 8; ++i) h[i] = data ? (unsigned short)(((rand() & 1) << 15) | (0x3F00 + (rand() & 0xFF))) : 0;   // +-[0.5, 2) | 0
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape)
      for (int grid : {256, 512}) {
        float best = 1e30f; double clk = 0;
        for (int rep = 0; rep < 4; ++rep) {
          hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
          hipEventRecord(a);
          if (shape == 0) k16<<<grid, 256>>>(src, out, sink, iters); else k32<<<grid, 256>>>(src, out, sink, iters);
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
          hipMemcpy(ho, out, grid * 16, hipMemcpyDeviceToHost);
          double st = 0, sr = 0; for (int i = 0; i < grid; ++i) { st += ho[i * 2]; sr += ho[i * 2 + 1]; }
          clk = st / sr * 100.0;
        }
        const double flops = (double)grid * 4 * iters * (shape == 0 ? 16.0 * 16384 : 8.0 * 32768);
        const double cyc = best * 1e-3 * clk * 1e6 / ((double)iters * (shape == 0 ? 16 : 8)) / (grid / 256);
        printf("%s data, %s, %d wave(s)/SIMD: %7.1f TFLOP/s, shader clock %6.0f MHz, %5.1f clk per MFMA per SIMD\n", data ? "random" : "zero  ",
               shape == 0 ? "16x16x32" : "32x32x16", grid / 256, flops / (best * 1e-3) / 1e12, clk, cyc);
      }
  }
  return 0;
}
