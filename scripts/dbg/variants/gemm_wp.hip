// Wave-private, weight-in-registers bf16 MFMA GEMM for the wide-N, K <= 192 instances of
//   C[M,N] = epilogue(A[M,K] * W[N,K]^T)     (FFN1 forward, QKV forward, FFN2 data-grad: N = 576/768, K = 192).
//
// Why a third NT kernel.  Phase stamps and the M-sweep of the tiled kernel (gemm_nt.hip) on these shapes show neither
// HBM (~4 TB/s marginal), nor the L2 (probe: 35 TB/s of hits available), nor LDS, nor the MFMA pipe (24 %) saturated:
// a workgroup lives through load-wait, K loop, epilogue and drain in sequence, with a workgroup barrier every 32-deep
// K step and every operand (activations AND the same 48 KB weight tile) pulled through LDS by every workgroup.
// The wave-private weight-gradient kernel (gemm_wgrad.hip) showed what removing the barriers is worth.  Same idea:
//   * a WAVE is the unit of work: it owns a 64-column block of W for its whole life and keeps that block's MFMA
//     fragments in REGISTERS (64 x 192 bf16 = 96 VGPRs) -- the weight never touches LDS, LDS read traffic halves;
//   * it walks down M in 64-row tiles (tile rg, rg + RG, ...): the activation tile streams through a PRIVATE 3-slot
//     global_load_lds ring (4 KB stages, swizzled on the DMA source address) that keeps running across tile
//     boundaries, so there is no pipeline refill per tile and NO workgroup barrier anywhere: only counted vmcnt waits
//     on the wave's own DMA, and 2 waves per SIMD cover each other's LDS latency and epilogues;
//   * accumulators are C^T (weight fragment as the A operand), so the register-only fused epilogue of gemm_common.h
//     (bias, ReLU, Philox dropout, gate, residual, one 16 B store per lane) is reused unchanged.
// MEASURED (cfg B shapes, M = 50432): equal to the tiled kernel within 3 % either way (FFN1 31.3 vs 32.2 us, QKV 28.0 vs
// 27.8 us; whole step 6.66 vs 6.59 ms) -- opt-in (IQ_GEMM_WP=1).  The ablation builds of this kernel are what located
// the time: with DMA, MFMA and stores ALL compiled out it still takes 16.5 of its 31 us (launch, weight load, LDS
// fragment reads + waits, epilogue ALU); dropping only the stores saves 8 us, only the MFMAs 5 us, only the DMA 3 us.
// The four waves of a workgroup are consecutive workers: neighbouring column blocks of the same row group, so their
// (duplicated) activation reads hit L1 / the XCD's L2.
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int WP_THREADS = 256, WP_ROWS = 64, WP_COLS = 64, WP_BK = 32, WP_NS = 3;
constexpr int WP_STAGE = WP_ROWS * WP_BK * 2;       // 4 KiB
constexpr int WP_WAVE_LDS = WP_NS * WP_STAGE;       // 12 KiB per wave

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ int wp_swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}

template <int KS, int EPI>
__global__ __launch_bounds__(WP_THREADS, 2) void gemm_nt_wp_kernel(const GemmParams p, int ncb, int RG, int tiles) {
  constexpr int MT = 4, NT = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* ring = smem + wave * WP_WAVE_LDS;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int worker = lid * 4 + wave;
  const int cb = worker % ncb, rg = worker / ncb;
  if (rg >= RG) return;                      // left-over workers (no barrier in this kernel: a wave may leave)
  const int n0 = cb * WP_COLS;

  // ---- this wave's weight block, once, straight into MFMA fragment layout ------------------------
  bf16x8 bfr[NT][KS];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const bf16* wrow = p.B + (long)(n0 + j * 16 + (lane & 15)) * p.ldb + 8 * (lane >> 4);
#pragma unroll
    for (int s = 0; s < KS; ++s) bfr[j][s] = *reinterpret_cast<const bf16x8*>(wrow + s * WP_BK);
  }
  // Retire these loads HERE: otherwise the compiler's own vmcnt(0) for the first use of bfr lands inside the tile loop
  // (it cannot see past the back edge) and drains the activation ring at the top of every tile.
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(bfr[j][s]));

  const int ntile = rg < tiles ? (tiles - rg + RG - 1) / RG : 0;
  const int Q = ntile * KS;
  // DMA piece i of a stage: rows 16 i .. 16 i + 15 (64 B each): lane -> row 16 i + (lane >> 2), 16 B chunk lane & 3
  const int prow = lane >> 2, pch = lane & 3;
  auto issue = [&](int q) {
    const int ti = q / KS, s = q - ti * KS;
    const int row0 = (rg + ti * RG) * WP_ROWS;
    unsigned char* dst = ring + (q % WP_NS) * WP_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = i * 16 + prow;
      const int gm = min(row0 + r, p.M - 1);
      const bf16* src = p.A + (long)gm * p.lda + s * WP_BK + ((pch ^ wp_swz(r)) << 3);
#ifndef IQ_WP_NO_DMA
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)(dst + i * 1024), 16, 0, 0);
#else
      asm volatile("" :: "v"(src), "v"(dst));
#endif
    }
  };
  if (Q > 0) issue(0);
  if (Q > 1) issue(1);

  int q = 0;
  for (int ti = 0; ti < ntile; ++ti) {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s, ++q) {
      // Stage q must have landed.  In vmcnt order it is followed only by stage q+1 (4 DMA instructions): the
      // epilogue of the previous tile ended with vmcnt(0) before its stores, which already retired the first two
      // stages of this tile (they were requested during that tile's last two steps), so steps 0 and 1 need no wait
      // and the stores drain under them; from step 2 on the stores are older than the stage being waited for.
      if (ti == 0 || s >= 2) {
        if (q + 1 < Q) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_wave_barrier();
      const bf16* As = reinterpret_cast<const bf16*>(ring + (q % WP_NS) * WP_STAGE);
      bf16x8 af[MT];
      const int ch = lane >> 4;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const bf16x8*>(As + row * WP_BK + ((ch ^ wp_swz(row)) << 3));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (q + 2 < Q) issue(q + 2);       // refills the slot stage q-1 vacated (its fragments were consumed last step)
#if defined(IQ_WP_NO_MFMA)   // ablation builds (scripts/dbg/ablate.py): timing only, results are wrong
      asm volatile("" :: "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]));
#else
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j][s], af[i], acc[i][j], 0, 0, 0);   // C^T tile
#endif
    }
    gemm_epilogue<MT, NT, EPI>(p, acc, (rg + ti * RG) * WP_ROWS, n0, lane);
  }
}

}  // namespace

// Returns IQ_OK after launching, or -1 ("not applicable") so the caller falls back to the tiled kernel.
int iq_gemm_wp_try_launch(const GemmParams& p0, int epi_mode, hipStream_t st) {
  GemmParams p = p0;
  const int K = p.K, N = p.N, M = p.M;
  if (!(K == 128 || K == 192)) return -1;
  if (N < 512 || (N % WP_COLS)) return -1;             // wide outputs only: a worker must see several row tiles
  if (M < 64 * 256) return -1;
  if ((((uintptr_t)p.A | (uintptr_t)p.B) % 16) || (p.lda % 8) || (p.ldb % 8)) return -1;
  if (epi_mode & EPI_PE) return -1;
  if (epi_mode == (EPI_RES | EPI_GATE)) return -1;
  const int ncb = N / WP_COLS;
  const int tiles = (M + WP_ROWS - 1) / WP_ROWS;
  const int grid = 512;                                // 2 workgroups (8 waves) per CU
  int RG = grid * 4 / ncb;
  if (RG > tiles) RG = tiles;
  const size_t lds = (size_t)4 * WP_WAVE_LDS;          // 48 KiB
#define IQ_WP_LAUNCH(KS_, EPI_) gemm_nt_wp_kernel<KS_, EPI_><<<grid, WP_THREADS, lds, st>>>(p, ncb, RG, tiles)
#define IQ_WP_EPI(KS_)                                 \
  switch (epi_mode) {                                  \
    case 0: IQ_WP_LAUNCH(KS_, 0); break;               \
    case EPI_RES: IQ_WP_LAUNCH(KS_, EPI_RES); break;   \
    case EPI_GATE: IQ_WP_LAUNCH(KS_, EPI_GATE); break; \
    default: return -1;                                \
  }
  if (K == 192) { IQ_WP_EPI(6) } else { IQ_WP_EPI(4) }
#undef IQ_WP_EPI
#undef IQ_WP_LAUNCH
  return IQ_OK;
}
