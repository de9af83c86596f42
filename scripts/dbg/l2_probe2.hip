// Probe 2: G workgroups on the SAME XCD stream the SAME region at the same time (the way the tiles of one wgrad
// split / the column tiles of one GEMM row block re-read their operand): what is the L2-level ingest rate when
// 1/G of the requests miss and the rest hit lines that were just (or are being) filled?
// Also: 256 B row segments with a 1536 B stride (a 128-column slice of a [M][768] bf16 matrix) vs contiguous 4 KB.
//   hipcc --offload-arch=gfx950 -O3 -w -o l2_probe2 l2_probe2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <bool STRIDED, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const unsigned char* __restrict__ src, size_t region, int iters, int G,
                                                 size_t src_bytes, unsigned* sink) {
  const int tid = threadIdx.x;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const size_t rid = (size_t)(idx / G) * 8 + xcd;
  const size_t base = (rid * region) % (src_bytes - region);
  u32x4 acc = {0, 0, 0, 0};
  if (STRIDED) {
    // thread -> (row = tid/16, 16 B chunk = tid%16) of a 256 B segment; rows 1536 B apart
    const int rows_per_it = THREADS / 16;
    size_t off = (size_t)(tid >> 4) * 1536 + (size_t)(tid & 15) * 16;
#pragma unroll 4
    for (int i = 0; i < iters; ++i) {
      acc ^= *reinterpret_cast<const u32x4*>(src + base + off);
      off += (size_t)rows_per_it * 1536;
    }
  } else {
    size_t off = (size_t)tid * 16;
#pragma unroll 4
    for (int i = 0; i < iters; ++i) {
      acc ^= *reinterpret_cast<const u32x4*>(src + base + off);
      off += (size_t)THREADS * 16;
    }
  }
  if (acc[0] == 0x12345678u && acc[1] == 77u) sink[0] = acc[2] ^ acc[3];
}

template <bool STRIDED, int THREADS>
void run(const char* name, unsigned char* src, size_t src_bytes, int grid, int iters, int G, unsigned* sink) {
  const size_t region = STRIDED ? (size_t)iters * (THREADS / 16) * 1536 : (size_t)iters * THREADS * 16;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) probe<STRIDED, THREADS><<<grid, THREADS>>>(src, region, iters, G, src_bytes, sink);
  hipEventRecord(a);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) probe<STRIDED, THREADS><<<grid, THREADS>>>(src, region, iters, G, src_bytes, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps;
  const double ld = (double)grid * iters * THREADS * 16;
  printf("%-34s thr %3d grid %5d G %2d: %8.1f us  L2-level %6.2f TB/s (%5.1f B/clk/CU)  distinct %6.2f TB/s\n", name, THREADS, grid, G, us,
         ld / us / 1e6, ld / us / 1e3 / 256 / 2.4, ld / G / us / 1e6);
}

int main() {
  const size_t src_bytes = 2048ull << 20;
  unsigned char* src; unsigned* sink;
  hipMalloc(&src, src_bytes); hipMalloc(&sink, 64);
  hipMemset(src, 1, src_bytes);
  for (int G : {1, 2, 3, 6, 18}) {
    // 512-thread workgroups, 2 per CU (the wgrad configuration): 16 waves/CU; each WG streams 2 MB
    run<false, 512>("contiguous 8 KB/iter", src, src_bytes, 512 / (8 * G) * (8 * G), 256, G, sink);
    run<true, 512>("256 B rows, 1536 B stride", src, src_bytes, 512 / (8 * G) * (8 * G), 256, G, sink);
    // 256-thread workgroups, 4 per CU
    run<false, 256>("contiguous 4 KB/iter", src, src_bytes, 1024 / (8 * G) * (8 * G), 256, G, sink);
    run<true, 256>("256 B rows, 1536 B stride", src, src_bytes, 1024 / (8 * G) * (8 * G), 256, G, sink);
  }
  return 0;
}
