// Weight-stationary, persistent bf16 MFMA GEMM for the K <= 256 instances of
//   C[M,N] = epilogue(A[M,K] * W[N,K]^T)        (M = frames*tokens ~ 5e4..1e5, N,K <= a few hundred).
//
// Why a second NT kernel: on these shapes the tiled kernel in gemm_nt.hip is bounded by the bytes each CU
// must INGEST (measured ~13 B/cycle/CU, L2 hits included), and more than half of what it ingests is
// re-fetched operands: the weight tile once per 128-row tile and the A rows once per column tile
// (FFN1, M=50432,N=768,K=192: 227 MB through the CUs for 97 MB of algorithmic traffic -> 32 us, where the
// HBM floor is 15 us).  Here a workgroup (8 waves, one per CU) keeps its [BN x K] weight tile RESIDENT in
// LDS for its whole life and streams 128-row A tiles through a 3-slot global_load_lds ring that keeps
// running across tile boundaries; BN is 192/256, so A is re-read N/BN (1..3) times instead of N/128 or
// N/64.  Same shape: 83 MB ingested.
//
// Wave grid 4 (rows) x 2 (cols): a wave owns 32 x BN/2 of the 128 x BN tile (2 x BN/32 MFMA tiles of
// mfma_f32_16x16x32_bf16, weight fragment as the A operand -> C^T accumulators, see gemm_common.h).
// LDS: W image [BN][K] (16 B chunk c of row r at c ^ f(r); f = r&15 when a row is a multiple of 256 B,
// else (r>>1)&7 -- conflict-free for the 16-lane groups of ds_read_b128, which mix chunks c and c+1),
// A ring 3 x [128][32] (64 B rows, chunk ^ {0,2,3,1}[(r>>2)&3], swizzle applied to the DMA source address).
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int WS_THREADS = 512, WS_BM = 128, WS_BK = 32, WS_NS = 3;
constexpr int WS_A_STAGE = WS_BM * WS_BK * 2;   // 8 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ int swz64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

template <int BN, int EPI>
__global__ __launch_bounds__(WS_THREADS) void gemm_nt_ws_kernel(const GemmParams p, int n_ct, int row_stride) {
  constexpr int NT = BN / 32;   // 16-col tiles per wave
  constexpr int MT = 2;         // 16-row tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = p.K, RB = 2 * K, CPRW = K / 8, NKS = K / WS_BK;
  const bool pow2row = (RB & 255) == 0;
  unsigned char* Ws = smem;
  unsigned char* ring = smem + BN * RB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);   // the column tiles of one row group share an XCD's L2
  const int ct = lid % n_ct, rgrp = lid / n_ct;
  const int n0 = ct * BN;
  auto wswz = [&](int row) { return pow2row ? (row & 15) : ((row >> 1) & 7); };

  // ---- resident weight tile (once) ---------------------------------------------------------------
  for (int id = tid; id < BN * CPRW; id += WS_THREADS) {
    const int row = id / CPRW, c = id - row * CPRW;
    const int gn = min(n0 + row, p.N - 1);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.B + (long)gn * p.ldb + c * 8);
    *reinterpret_cast<bf16x8*>(Ws + row * RB + ((c ^ wswz(row)) << 4)) = v;
  }

  const int ntile = rgrp < p.tiles_m ? (p.tiles_m - rgrp + row_stride - 1) / row_stride : 0;
  const int Q = ntile * NKS;     // A stages this workgroup consumes
  // this wave's DMA piece of an A stage: rows 16w .. 16w+15, 64 B each
  const int prow = wave * 16 + (lane >> 2), pch = lane & 3;
  const int pcol = (pch ^ swz64(prow)) * 8;
  auto issue = [&](int q) {
    const int ti = q / NKS, ks = q - ti * NKS;
    const int gm = min((rgrp + ti * row_stride) * WS_BM + prow, p.M - 1);
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(p.A + (long)gm * p.lda + ks * WS_BK + pcol),
                                     (lds_void_t*)(ring + (q % WS_NS) * WS_A_STAGE + wave * 1024), 16, 0, 0);
  };
  if (Q > 0) issue(0);
  if (Q > 1) issue(1);
  __syncthreads();               // weight image visible (also drains the two DMAs: one-time)

  int q = 0;
  for (int ti = 0; ti < ntile; ++ti) {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < NKS; ++ks, ++q) {
      // one DMA per wave per stage: all but the youngest one done == stage q landed (conservative when the
      // previous tile's stores are still in flight: vmcnt is one in-order counter for loads and stores)
      if (q + 1 < Q) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (q + 2 < Q) issue(q + 2);       // refills the slot stage q-1 vacated; runs ahead across tile boundaries
      const bf16* As = reinterpret_cast<const bf16*>(ring + (q % WS_NS) * WS_A_STAGE);
      bf16x8 af[MT], bfr[NT];
      const int ch = lane >> 4;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * 32 + i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const bf16x8*>(As + row * WS_BK + (ch ^ swz64(row)) * 8);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * (BN / 2) + j * 16 + (lane & 15);
        bfr[j] = *reinterpret_cast<const bf16x8*>(Ws + row * RB + (((ks * 4 + ch) ^ wswz(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile
    }
    const int m0 = (rgrp + ti * row_stride) * WS_BM;
    gemm_epilogue<MT, NT, EPI>(p, acc, m0 + wm * 32, n0 + wn * (BN / 2), lane);
  }
}

}  // namespace

// Returns IQ_OK after launching, or a negative "not applicable" code (-1) so the caller can fall back.
int iq_gemm_ws_try_launch(const GemmParams& p0, int epi_mode, hipStream_t st) {
  GemmParams p = p0;
  const int K = p.K, N = p.N, M = p.M;
  if (!(K == 64 || K == 128 || K == 192 || K == 256)) return -1;
  if (N < 128 || (N % 8)) return -1;
  if ((((uintptr_t)p.A | (uintptr_t)p.B) % 16) || (p.lda % 8) || (p.ldb % 8)) return -1;
  if (epi_mode == (EPI_RES | EPI_GATE)) return -1;
  // column tile: least padding, ties to the wider tile
  int bn = 256, best = (N + 255) / 256 * 256 - N;
  for (int c : {192, 128}) {
    const int w = (N + c - 1) / c * c - N;
    if (w < best) { best = w; bn = c; }
  }
  const size_t lds = (size_t)bn * K * 2 + (size_t)WS_NS * WS_A_STAGE;
  if (lds > 160 * 1024) return -1;
  p.tiles_m = (M + WS_BM - 1) / WS_BM;
  const int n_ct = (N + bn - 1) / bn;
  int R = 256 / n_ct;
  if (R < 1) R = 1;
  if (R > p.tiles_m) R = p.tiles_m;
  const int grid = n_ct * R;
#define IQ_WS_LAUNCH(BN_, EPI_)                                                                                 \
  do {                                                                                                          \
    auto k = gemm_nt_ws_kernel<BN_, EPI_>;                                                                      \
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    k<<<grid, WS_THREADS, lds, st>>>(p, n_ct, R);                                                               \
  } while (0)
#define IQ_WS_EPI(BN_)                                            \
  switch (epi_mode) {                                             \
    case 0: IQ_WS_LAUNCH(BN_, 0); break;                          \
    case EPI_RES: IQ_WS_LAUNCH(BN_, EPI_RES); break;              \
    case EPI_GATE: IQ_WS_LAUNCH(BN_, EPI_GATE); break;            \
    case EPI_PE: IQ_WS_LAUNCH(BN_, EPI_PE); break;                \
    default: return -1;                                           \
  }
  if (bn == 256) { IQ_WS_EPI(256) } else if (bn == 192) { IQ_WS_EPI(192) } else { IQ_WS_EPI(128) }
#undef IQ_WS_EPI
#undef IQ_WS_LAUNCH
  return IQ_OK;
}
