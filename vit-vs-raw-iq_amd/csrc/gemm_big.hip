// bf16 MFMA GEMM for the MFMA-bound shapes, C[M,N] = epilogue(A[M,K] * B[N,K]^T) with N % 256 == 0, K >= 256 (gfx950):
// every Linear of ViT-Base (V/models/amc_transformer.py:9 at D768 / F3072; multi_head_attention.py:18,28,
// position_wise_feed_forward.py:13-16) and its data gradients.  The 128 x 128 tiles of gemm_nt.hip are built for
// K = 192 (three independent workgroups per CU hide a 6-stage loop's fill and drain); at K = 768..3072 they reach 0.28 of
// the MFMA peak, because a 128 x 128 x 32 stage is only 16 MFMAs per wave behind 8 LDS fragment reads and a barrier.
//
//   tile 256 x 256, 8 waves as 2 (M) x 4 (N), 128 x 64 per wave = 8 x 4 accumulator tiles (128 VGPRs),
//   32 MFMAs per wave per 32-deep stage behind 12 fragment reads (0.375 reads / MFMA, was 0.5),
//   operands by global_load_lds into a 4-slot ring of [A 256 rows | B 256 rows] x 64 B stages (128 KiB: one workgroup per
//   CU), three stages in flight across ONE raw s_barrier per stage, counted vmcnt (never 0 inside the loop),
//   rows XOR-swizzled on the global side exactly as in gemm_nt.hip (conflict-free ds_read_b128),
//   the shared register-only epilogue (gemm_common.h: bias, ReLU, dropout, gate, residual) over two row tiles at a time.
// Persistent over tiles: a workgroup walks its share of the tile list and issues the next tile's first three stages
// BEFORE the current tile's epilogue, so the ring never drains at a tile boundary and the epilogue's loads and stores
// run under operand traffic -- with one workgroup per CU nothing else would hide them.
#include "common.h"
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

#ifndef BG_NS_
#define BG_NS_ 4
#endif
constexpr int BG_THREADS = 512, BG_BM = 256, BG_BN = 256, BG_BK = 32, BG_NS = BG_NS_, BG_DIST = BG_NS - 1;
constexpr int BG_STAGE = (BG_BM + BG_BN) * BG_BK * 2;     // 32 KiB
constexpr int BG_PS = BG_STAGE / 1024 / 8;                // DMA pieces (16 rows x 64 B) per wave per stage: 4
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ int bswz64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}: gemm_nt.hip

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 layout: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
constexpr int bg_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

template <int EPI>
__global__ __launch_bounds__(BG_THREADS, 1) void gemm_big_kernel(const GemmParams p, int ntiles) {
  constexpr int MT = 8, NT = 4, PS = BG_PS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nk = p.K / BG_BK;
  const int prow = lane >> 2, pch = lane & 3, ch = lane >> 4;

  // Tile list: logical id -> (row block, column block), column fastest, so that the workgroups of one XCD (contiguous
  // logical ids after the remap) share A row blocks and sweep the whole weight through that XCD's L2.
  const int first = xcd_remap(blockIdx.x, gridDim.x);
  // stage q of this workgroup's stream = tile (q / nk) of its list, k-slice (q % nk)
  const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nstage = my_tiles * nk;
  auto tile_rc = [&](int ti, int& m0, int& n0) {
    const int t = first + ti * (int)gridDim.x;
    m0 = (t / p.tiles_n) * BG_BM;
    n0 = (t % p.tiles_n) * BG_BN;
  };
  const IqRng rng = p.drop_on ? rng_resolve(p.rng) : p.rng;     // oldest entry of the vector-memory queue

  // The issue stream runs three stages ahead of the compute stream and crosses tile boundaries: its own tile / k-slice /
  // ring-slot state, row pointers recomputed once per tile (4 per wave: pieces 4w..4w+3; 0..15 are A rows, 16..31 B rows).
  int iq = 0, iks = 0, iti = 0, islot = 0;
  const bf16* isrc[PS];
  auto issue_setup = [&]() {
    int m0, n0;
    tile_rc(iti, m0, n0);
#pragma unroll
    for (int i = 0; i < PS; ++i) {
      const int piece = wave * PS + i;
      const int row = (piece & 15) * 16 + prow;
      isrc[i] = (piece < 16 ? p.A + (long)min(m0 + row, p.M - 1) * p.lda : p.B + (long)min(n0 + row, p.N - 1) * p.ldb) +
                (pch ^ bswz64(row)) * 8;
    }
  };
  auto issue_piece = [&](int i) {
#ifdef BG_NO_DMA    // ablation build: timing only
    return;
#endif
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(isrc[i] + iks * BG_BK),
                                     (lds_void_t*)(smem + islot * BG_STAGE + (wave * PS + i) * 1024), 16, 0, 0);
  };
  auto issue_advance = [&]() {
    ++iq;
    islot = islot + 1 == BG_NS ? 0 : islot + 1;
    if (++iks == nk) {
      iks = 0;
      if (++iti < my_tiles) issue_setup();
    }
  };
  issue_setup();
#pragma unroll
  for (int s = 0; s < BG_DIST; ++s) {                            // (nstage >= nk >= 8)
#pragma unroll
    for (int i = 0; i < PS; ++i) issue_piece(i);
    issue_advance();
  }

  // Ping-pong: the four waves of row half 1 run one barrier behind those of row half 0 (a workgroup's waves go to SIMDs
  // round-robin, so every SIMD holds one wave of each half).  Between two barriers one half reads its 12 fragments while
  // the other half runs its 32 MFMAs with its 4 DMA pieces of stage q+3 issued between them (a piece costs ~60 cycles
  // of issue among bare MFMAs, 100-185 in a phase that also carries the fragment reads), then they swap: LDS latency
  // and barrier skew hide under the partner's MFMAs.  Two barriers per stage, all eight waves at each.
  //   barrier 2q   .. 2q+1 : half 0 reads stage q                  | half 1 MFMAs stage q-1, issues q+2
  //   barrier 2q+1 .. 2q+2 : half 0 MFMAs stage q, issues q+3      | half 1 reads stage q
  // Stage q+1 is waited for (counted vmcnt) by every wave before barrier 2q+2: half 0 after its MFMAs, half 1 after its
  // reads.  A ring slot is refilled (stage q+3 over stage q-1) only after barrier 2q+1, behind which both halves' reads
  // of stage q-1 have returned (lgkmcnt(0) sits before the barrier that ends a read phase).
  __builtin_amdgcn_s_waitcnt(bg_vmcnt((BG_DIST - 1) * PS));
  __builtin_amdgcn_s_barrier();                                  // barrier 0: stage 0 is visible
  if (wm == 1) __builtin_amdgcn_s_barrier();                     // half 1: one phase behind

  f32x4 acc[MT][NT];
  int q = 0, cslot = 0;
  auto epilogue = [&](int ti) {
    // every load of the tail first (the fragment registers are dead here: the 64 residual / gate registers fit), one
    // wait -- which also retires the next tile's first stages -- then 16 stores back to back
    int m0, n0;
    tile_rc(ti, m0, n0);
    const int row0 = m0 + wm * 128, col0 = n0 + wn * 64;
    EpiRegs<MT, NT, EPI> R;
    R.rng = rng;
    epi_load_early<MT, NT, EPI>(p, R, row0, col0, lane);
    __builtin_amdgcn_s_waitcnt(bg_vmcnt(0));
    epi_finish<MT, NT, EPI>(p, acc, R, row0, col0, lane);
  };
  // Stage q+1 must have landed before the barrier that follows.  Entries of this wave's queue younger than its pieces:
  // those of the stages issued after it (iq - q - 2 of them).  fresh = the first two stages behind a tile boundary:
  // stage q+1 was issued before the epilogue, whose vmcnt(0) retired it, and a wait here would wait for its stores.
  auto wait_next = [&](bool fresh) {
#ifdef BG_NO_WAIT   // ablation build: timing only
    return;
#endif
    if (fresh || q + 1 >= nstage) return;
    const int y = iq - q - 2;
    if (y >= 3) __builtin_amdgcn_s_waitcnt(bg_vmcnt(3 * PS));
    else if (y == 2) __builtin_amdgcn_s_waitcnt(bg_vmcnt(2 * PS));
    else if (y == 1) __builtin_amdgcn_s_waitcnt(bg_vmcnt(PS));
    else __builtin_amdgcn_s_waitcnt(bg_vmcnt(0));
  };
  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < nk; ++ks, ++q) {
      const bool fresh = ti > 0 && ks < BG_DIST - 1;
      // ---- read phase ----------------------------------------------------------------------------------------------
      const bf16* As = reinterpret_cast<const bf16*>(smem + cslot * BG_STAGE);
      cslot = cslot + 1 == BG_NS ? 0 : cslot + 1;
      const bf16* Bs = As + BG_BM * BG_BK;
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = wn * 64 + j * 16 + (lane & 15);
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BG_BK + (ch ^ bswz64(row)) * 8);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = wm * 128 + i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const bf16x8*>(As + row * BG_BK + (ch ^ bswz64(row)) * 8);
      }
      if (iq < nstage) {                                          // wave-uniform: stage q+3, into the slot of stage q-1
#pragma unroll
        for (int i = 0; i < PS; ++i) issue_piece(i);
        issue_advance();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (wm == 1) wait_next(fresh);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA phase ----------------------------------------------------------------------------------------------
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile: gemm_common.h
      __builtin_amdgcn_s_setprio(0);
      if (wm == 0) wait_next(fresh);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    // In program order the epilogue follows the tile's last barrier: it runs in the partner half's MFMA phase of that
    // stage (half 0) or of the next tile's first stage (half 1), with the next tile's first three stages in flight.
    epilogue(ti);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();                     // matches half 1's extra barrier at the start
}

}  // namespace

// Called by iq_gemm_bf16_nt with its resolved parameters.  Returns false when the shape / epilogue is not this kernel's.
bool gemm_big_try(const GemmParams& p0, int epi_mode, hipStream_t st) {
  if (epi_mode != 0 && epi_mode != EPI_RES && epi_mode != EPI_GATE) return false;
  if (p0.N % BG_BN != 0 || p0.K % BG_BK != 0 || p0.K < 256 || p0.M < 2048) return false;
  if ((((uintptr_t)p0.A | (uintptr_t)p0.B | (uintptr_t)p0.C) % 16) || (p0.ldc % 8)) return false;
  GemmParams p = p0;
  p.tiles_m = (p.M + BG_BM - 1) / BG_BM;
  p.tiles_n = p.N / BG_BN;
  const int ntiles = p.tiles_m * p.tiles_n;
  if (ntiles < 512) return false;                          // fewer than two rounds of 256 CUs: the 128 x 128 tiles fill the chip better
  const int grid = ntiles < 256 ? ntiles : 256;            // one workgroup per CU, persistent over its share of the tiles
  const size_t lds = (size_t)BG_NS * BG_STAGE;             // 128 KiB
#define IQ_BIG_LAUNCH(E)                                                                                              \
  do {                                                                                                                \
    auto k = gemm_big_kernel<E>;                                                                                      \
    static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    (void)attr;                                                                                                       \
    k<<<grid, BG_THREADS, lds, st>>>(p, ntiles);                                                                      \
  } while (0)
  if (epi_mode == EPI_RES) IQ_BIG_LAUNCH(EPI_RES);
  else if (epi_mode == EPI_GATE) IQ_BIG_LAUNCH(EPI_GATE);
  else IQ_BIG_LAUNCH(0);
#undef IQ_BIG_LAUNCH
  return true;
}
