// Probe 3: store bandwidth as a function of the per-instruction footprint.
//   mode 0: wave writes 1 KB contiguous per instruction (64 lanes x 16 B)
//   mode 1: the GEMM epilogue pattern: an instruction writes 16 rows x 64 B (row stride = ld bytes); the second half of
//           each 128 B line comes from the NEXT instruction of the same wave
//   mode 2: 8 rows x 128 B per instruction (full lines)
//   mode 3: 4 rows x 256 B
//   hipcc --offload-arch=gfx950 -O3 -w -o st_probe st_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned char* dst, int ld, int rows_per_wg, int ncolblk) {
  // the WG owns rows [blockIdx.x/ncolblk * 128 ...) and a 256 B (128-col bf16) column block, like a 128x128 GEMM tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = (long)(blockIdx.x / ncolblk) * 128 + wave * 32;   // wave: 32 rows x 256 B
  const long col0 = (long)(blockIdx.x % ncolblk) * 256;
  u32x4 v = {(unsigned)lane, (unsigned)wave, blockIdx.x, 7u};
  if (MODE == 0) {       // not a tile: contiguous 8 KB per wave
    unsigned char* p = dst + ((long)blockIdx.x * 4 + wave) * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(p + i * 1024) = v;
  } else if (MODE == 1) {
    // 8 instrs: (16-row group rg = 0,1) x (64 B quarter qd = 0..3): lane -> row = lane & 15, 16 B chunk = lane >> 4
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
        *reinterpret_cast<u32x4*>(dst + (row0 + rg * 16 + (lane & 15)) * ld + col0 + qd * 64 + (lane >> 4) * 16) = v;
  } else if (MODE == 2) {
    // 8 instrs: (8-row group 0..3) x (128 B half 0..1): lane -> row = lane >> 3, chunk = lane & 7
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<u32x4*>(dst + (row0 + rg * 8 + (lane >> 3)) * ld + col0 + h * 128 + (lane & 7) * 16) = v;
  } else if (MODE == 4) {
    // 8 instrs: one 32-row group x (32 B eighth 0..7): lane -> row = lane & 31, 16 B chunk = lane >> 5   (32x32 MFMA epilogue)
#pragma unroll
    for (int e8 = 0; e8 < 8; ++e8)
      *reinterpret_cast<u32x4*>(dst + (row0 + (lane & 31)) * ld + col0 + e8 * 32 + (lane >> 5) * 16) = v;
  } else {
    // 8 instrs: 4-row groups, 256 B rows: lane -> row = lane >> 4, chunk = lane & 15
#pragma unroll
    for (int rg = 0; rg < 8; ++rg)
      *reinterpret_cast<u32x4*>(dst + (row0 + rg * 4 + (lane >> 4)) * ld + col0 + (lane & 15) * 16) = v;
  }
}
template <int MODE> void run(const char* name, unsigned char* dst, int M, int N) {
  const int ld = N * 2, ncolblk = N / 128, grid = (M / 128) * ncolblk;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) k<MODE><<<grid, 256>>>(dst, ld, 128, ncolblk);
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) k<MODE><<<grid, 256>>>(dst, ld, 128, ncolblk);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 100, bytes = (double)M * N * 2;
  printf("%-44s M %6d N %4d: %7.1f us  %6.2f TB/s\n", name, M, N, us, bytes / us / 1e6);
}
int main() {
  unsigned char* dst; hipMalloc(&dst, 1ull << 30); hipMemset(dst, 0, 1ull << 30);
  for (int M : {50432, 131072}) {
    for (int N : {768, 128 * 5}) {
      run<0>("contiguous 1 KB / instruction", dst, M, N);
      run<1>("16 rows x 64 B / instruction (epilogue)", dst, M, N);
      run<2>("8 rows x 128 B / instruction", dst, M, N);
      run<3>("4 rows x 256 B / instruction", dst, M, N);
      run<4>("32 rows x 32 B / instruction (32x32 epi)", dst, M, N);
    }
  }
  return 0;
}
