// What does the shader clock actually run at?  s_memtime counts shader clocks, s_memrealtime a constant 100 MHz.
// Each wave spins on MFMAs (and optionally streams memory) for a while and reports both deltas.
//   hipcc --offload-arch=gfx950 -O3 -w -o clk_probe clk_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, int mem, const u32x4* src, u32x4* dst) {
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)1.0f; }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  u32x4 m = {0, 0, 0, 0};
  size_t off = ((size_t)blockIdx.x * 256 + threadIdx.x);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    if (mem) { m ^= src[off & ((1u << 24) - 1)]; dst[off & ((1u << 24) - 1)] = m; off += 256 * 2048; }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (unsigned long long)s + m[0]; }
}

__global__ __launch_bounds__(256) void k32(unsigned long long* out, int iters) {
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)1.0f; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (unsigned long long)s; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 4096 * 24);
  u32x4 *src, *dst; hipMalloc(&src, 1u << 28); hipMalloc(&dst, 1u << 28); hipMemset(src, 1, 1u << 28);
  unsigned long long h[4096 * 3];
  for (int mem = 0; mem < 1; ++mem)
    for (int grid : {256, 512, 1024, 2048}) {
      const int iters = 20000;
      k<<<grid, 256>>>(d, iters, mem, src, dst);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a); k<<<grid, 256>>>(d, iters, mem, src, dst); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      hipMemcpy(h, d, grid * 24, hipMemcpyDeviceToHost);
      double st = 0, sr = 0; for (int i = 0; i < grid; ++i) { st += h[i * 3]; sr += h[i * 3 + 1]; }
      const double mhz = st / sr * 100.0;
      const double flops = (double)grid * 4 * iters * 8 * 16384;
      printf("mem %d grid %5d: kernel %8.1f us; shader clock %7.1f MHz (memtime/memrealtime); MFMA rate %7.1f TFLOP/s; per-wave MFMA issue interval %.1f shader clk\n",
             mem, grid, ms * 1e3, mhz, flops / (ms * 1e-3) / 1e12, (st / grid) / (iters * 8.0));
    }
  for (int grid : {256, 512, 1024, 2048}) {
    const int iters = 20000;
    k32<<<grid, 256>>>(d, iters);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k32<<<grid, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = (double)grid * 4 * iters * 4 * 32768;
    printf("32x32x16: grid %5d (%d waves/SIMD): kernel %8.1f us; MFMA rate %7.1f TFLOP/s\n", grid, grid / 256, ms * 1e3, flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}
