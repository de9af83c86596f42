// Probe: LDS-DMA (global_load_lds_dwordx4) ingest rate of one 8-wave workgroup per CU for the two piece shapes a GEMM
// operand ring can use on a row-major [rows][K] bf16 matrix (pitch 6144 B):
//   shape A: 16 rows x 64 B per wave-instruction, the k offset advancing 64 B per stage (BK = 32 stages: every 128 B line
//            is fetched by two instructions, two stages apart)
//   shape B:  8 rows x 128 B per wave-instruction (BK = 64 units: whole lines)
// with 12 pieces in flight per wave (vmcnt(8) after every 4), private rows (HBM stream) or rows shared by all workgroups (L2).
//   hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
constexpr size_t PITCH = 6144;

template <int SHAPE, int SHARED>
__global__ __launch_bounds__(512, 1) void probe(const unsigned char* __restrict__ src, size_t total_rows, int tiles, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int slot = 0;
  for (int t = 0; t < tiles; ++t) {
    const size_t row0 = (((SHARED ? 0 : (size_t)blockIdx.x) + (size_t)t * gridDim.x) * 512) % (total_rows - 512);
    for (int it = 0; it < 96; ++it) {                    // 96 x 32 KB = 512 rows x 6144 B
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;                  // 0..31
        size_t off;
        if (SHAPE == 0) off = (row0 + piece * 16 + (lane >> 2)) * PITCH + (size_t)it * 64 + (lane & 3) * 16;
        else off = (row0 + (it & 1) * 256 + piece * 8 + (lane >> 3)) * PITCH + (size_t)(it >> 1) * 128 + (lane & 7) * 16;
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(src + off), (lds_void_t*)(lds + slot * 32768 + piece * 1024), 16, 0, 0);
      }
      slot = (slot + 1) & 3;
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (*reinterpret_cast<unsigned*>(lds + tid * 4) == 0x12345678u) sink[0] = 1;
}

template <int SHAPE, int SHARED>
void run(const char* name, unsigned char* src, size_t total_rows, int tiles, unsigned* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  auto k = probe<SHAPE, SHARED>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int w = 0; w < 2; ++w) k<<<256, 512, 131072>>>(src, total_rows, tiles, sink);
  hipEventRecord(a);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) k<<<256, 512, 131072>>>(src, total_rows, tiles, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps, bytes = 256.0 * tiles * 96 * 32768;
  printf("%-52s %8.1f us  %6.2f TB/s  (%5.1f B/clk/CU @2.4 GHz)\n", name, us, bytes / us / 1e6, bytes / us / 1e3 / 256 / 2.4);
}

int main() {
  const size_t bytes = 1ull << 30, total_rows = bytes / PITCH;
  unsigned char* src; unsigned* sink;
  hipMalloc(&src, bytes); hipMalloc(&sink, 64); hipMemset(src, 1, bytes);
  run<0, 0>("16 rows x 64 B pieces, private rows (HBM)", src, total_rows, 4, sink);
  run<1, 0>(" 8 rows x 128 B pieces, private rows (HBM)", src, total_rows, 4, sink);
  run<0, 1>("16 rows x 64 B pieces, rows shared by all WGs (L2)", src, total_rows, 4, sink);
  run<1, 1>(" 8 rows x 128 B pieces, rows shared by all WGs (L2)", src, total_rows, 4, sink);
  return 0;
}
