// Probe: how many bytes/clk a CU can ingest (a) from an L2-resident region shared by every workgroup (the weight
// operand of the GEMMs), (b) from distinct streaming regions (the activation operand), through
// global_load_dwordx4 (registers) and global_load_lds_dwordx4 (LDS DMA), and how that combines with stores.
//   hipcc --offload-arch=gfx950 -O3 -o l2_probe l2_probe.hip && ./l2_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// mode bit0: 1 = every WG reads the same `region` bytes (L2 hits), 0 = WG-private streaming regions
// mode bit1: 1 = LDS DMA, 0 = register loads
// mode bit2: also store 16 B per lane per `st_every` loads (streaming writes)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                             size_t region, int iters, size_t src_bytes, int st_every, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t wg_base = (MODE & 1) ? 0 : ((size_t)blockIdx.x * region) % (src_bytes - region);
  u32x4 acc = {0, 0, 0, 0};
  size_t off = (size_t)tid * 16;
  size_t wr = ((size_t)blockIdx.x * iters / (st_every > 0 ? st_every : 1)) * 4096 + (size_t)tid * 16;
  for (int i = 0; i < iters; ++i) {
    const unsigned char* p = src + wg_base + off;
    if (MODE & 2) {
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)p, (lds_void_t*)(lds + (i & 3) * 4096 + wave * 1024), 16, 0, 0);
    } else {
      const u32x4 v = *reinterpret_cast<const u32x4*>(p);
      acc ^= v;
    }
    if ((MODE & 4) && (i % st_every) == 0) {
      *reinterpret_cast<u32x4*>(dst + (wr & ((1024ull << 20) - 1))) = acc;
      wr += 4096;
    }
    off += 4096;
    if (off >= region) off -= region;
  }
  if (MODE & 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc[0] ^= *reinterpret_cast<unsigned*>(lds + tid * 4);
  }
  if (acc[0] == 0x12345678u && acc[1] == 77u) sink[0] = acc[2] ^ acc[3];
}

template <int MODE>
void run(const char* name, unsigned char* src, unsigned char* dst, size_t region, size_t src_bytes, int grid, int iters,
         int st_every, unsigned* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) probe<MODE><<<grid, 256>>>(src, dst, region, iters, src_bytes, st_every, sink);
  hipEventRecord(a);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) probe<MODE><<<grid, 256>>>(src, dst, region, iters, src_bytes, st_every, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps;
  const double ld = (double)grid * iters * 4096, stb = (MODE & 4) ? (double)grid * (iters / st_every) * 4096 : 0;
  printf("%-44s grid %5d iters %5d: %8.1f us  load %7.2f TB/s (%5.1f B/clk/CU @2.4GHz)  store %6.2f TB/s\n", name, grid, iters,
         us, ld / us / 1e6, ld / us / 1e3 / 256 / 2.4, stb / us / 1e6);
}

int main() {
  const size_t src_bytes = 512ull << 20, dst_bytes = 1024ull << 20;
  unsigned char *src, *dst; unsigned* sink;
  hipMalloc(&src, src_bytes); hipMalloc(&dst, dst_bytes); hipMalloc(&sink, 64);
  hipMemset(src, 1, src_bytes); hipMemset(dst, 0, dst_bytes);
  for (int wgs_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * wgs_per_cu;
    const int iters = 2048 / wgs_per_cu;   // 8 MB per CU in total
    printf("---- %d WG/CU (4 waves each)\n", wgs_per_cu);
    run<1>("shared 48 KB region, register loads", src, dst, 49152, src_bytes, grid, iters, 0, sink);
    run<3>("shared 48 KB region, LDS DMA", src, dst, 49152, src_bytes, grid, iters, 0, sink);
    run<1>("shared 2 MB region, register loads", src, dst, 2 << 20, src_bytes, grid, iters, 0, sink);
    run<0>("private streaming, register loads", src, dst, (size_t)iters * 4096, src_bytes, grid, iters, 0, sink);
    run<2>("private streaming, LDS DMA", src, dst, (size_t)iters * 4096, src_bytes, grid, iters, 0, sink);
    run<5>("shared 48 KB reg loads + store every 3", src, dst, 49152, src_bytes, grid, iters, 3, sink);
    run<4>("private streaming + store every 1", src, dst, (size_t)iters * 4096, src_bytes, grid, iters, 1, sink);
    run<7>("shared 48 KB LDS DMA + store every 3", src, dst, 49152, src_bytes, grid, iters, 3, sink);
  }
  return 0;
}
