// bf16 MFMA GEMM for the MFMA-bound shapes, C[M,N] = epilogue(A[M,K] * B[N,K]^T) with N % 256 == 0, K % 64 == 0, K >= 256
// (gfx950): every Linear of ViT-Base (V/models/amc_transformer.py:9 at D768 / F3072; multi_head_attention.py:18,28,
// position_wise_feed_forward.py:13-16) and its data gradients.  The 128 x 128 tiles of gemm_nt.hip are built for
// K = 192 (three independent workgroups per CU hide a 6-stage loop's fill and drain); at K = 768..3072 they reach 0.28 of
// the MFMA peak.
//
// Structure (the 256 x 256 "quadrant" schedule of the CDNA4 playbook, cdna_hip_programming.md section 5, rebuilt here as a
// persistent kernel with this library's epilogue):
//   * tile 256 x 256, K-tile 64, 8 waves as 2 (M) x 4 (N), 128 x 64 per wave = 8 x 4 accumulator tiles (128 VGPRs);
//   * operands arrive by global_load_lds in WHOLE 128-byte lines (8 rows x 128 B per wave-instruction: 42 B/clk/CU from
//     L2 against 27 for the 16 rows x 64 B pieces of a 32-deep stage, scripts/dbg/dma_probe.hip -- a 256 x 256 tile needs
//     32 B/clk/CU at the MFMA rate), rows XOR-swizzled on the global side (chunk ^ (row >> 1) & 7): conflict-free
//     ds_read_b128 on 128-byte rows;
//   * a K-tile is four 16 KiB units: A0 / A1 = every wave's first / second 64 rows, B0 / B1 = every wave's first / second
//     32 columns; two sets of four (128 KiB, one workgroup per CU).  Two phases per K-tile: X = quadrants (A0,B0) (A0,B1)
//     behind 16 fragment reads (A0, B0, B1), Y = (A1,B1) (A1,B0) behind 8 (A1; the B fragments stay in registers):
//     24 reads per 64 MFMAs.  Two units are requested per phase, each two or three phases before its first read and
//     as soon as the slot it overwrites has been read for the last time: X(g) requests B1 A1 of K-tile g+1, Y(g) requests
//     A0 B0 of K-tile g+2 (the slots of A0 B0 of K-tile g, which X(g) moved to registers); counted vmcnt, never 0 in
//     the loop; the loop carries no address arithmetic (uniform base + per-lane offsets that change only at a ragged row block);
//   * ping-pong: the four waves of row half 1 run one barrier behind those of row half 0 (every SIMD holds one wave of
//     each half): between two barriers one half reads fragments and issues its DMA pieces while the other half issues
//     MFMAs, then they swap -- LDS latency, DMA issue and barrier skew sit under the partner's MFMAs;
//   * persistent over tiles: the unit stream crosses tile boundaries, so the next tile's first K-tile is in flight under
//     the epilogue; the epilogue (gemm_common.h: bias, ReLU, dropout, gate, residual; register-only) issues all its loads,
//     waits once and stores 16 x 16 B per lane back to back.
#include "common.h"
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int BG_THREADS = 512, BG_BM = 256, BG_BN = 256, BG_KT = 64, BG_UNIT = 128 * 128;   // unit: 128 rows x 128 B
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 layout: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
constexpr int bg_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

template <int EPI>
__global__ __launch_bounds__(BG_THREADS, 1) void gemm_big_kernel(const GemmParams p, int ntiles) {
  constexpr int MT = 8, NT = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nkt = p.K / BG_KT;
  const int ch = lane >> 4, c16 = lane & 15;

  // Tile list: logical id -> (row block, column block), column fastest, so that the workgroups of one XCD (contiguous
  // logical ids after the remap) share A row blocks and sweep the whole weight through that XCD's L2.
  const int first = xcd_remap(blockIdx.x, gridDim.x);
  const int my_tiles = (ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total_kt = my_tiles * nkt;
  auto tile_rc = [&](int ti, int& m0, int& n0) {
    const int t = first + ti * (int)gridDim.x;
    m0 = (t / p.tiles_n) * BG_BM;
    n0 = (t % p.tiles_n) * BG_BN;
  };
  const IqRng rng = p.drop_on ? rng_resolve(p.rng) : p.rng;     // oldest entry of the vector-memory queue

  // ---- the unit stream ------------------------------------------------------------------------------------------------------
  // Request order per K-tile: A0 B0 | B1 A1 (the bar = a phase boundary).  LDS slot of a unit: set * 4 + {A0: 0, A1: 1,
  // B0: 2, B1: 3}, set = K-tile parity.  A wave's two pieces of a unit are unit rows 16 w .. 16 w + 15; a lane's source =
  // (uniform tile / K-tile base) + (per-lane offset that only changes with the tile's row count: the rows of a ragged
  // last row block are clamped to its last valid row, their outputs are never stored).
  unsigned offA[2][2], offB[2][2];                               // [half][piece], bytes
  int irows = BG_BM;                                             // valid rows of the tile being requested
  // (recomputed behind every epilogue from an opaque copy of the lane id: kept live across the epilogue they are what the
  //  register allocator spills -- and a scratch reload in the loop is a vmcnt entry that drains the DMA queue)
  auto lane_offsets = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ur = (wave * 2 + i) * 8 + (l >> 3);              // row inside the unit
      const int gch = ((l & 7) ^ ((ur >> 1) & 7)) * 16;          // source chunk of this lane's LDS position, bytes
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        offA[half][i] = (unsigned)(min((ur >> 6) * 128 + half * 64 + (ur & 63), irows - 1) * p.lda * 2 + gch);
        offB[half][i] = (unsigned)(((ur >> 5) * 64 + half * 32 + (ur & 31)) * p.ldb * 2 + gch);
      }
    }
  };
  int iti = 0, ikt = 0, iset = 0;                                // the K-tile whose units are being requested
  const char *ibaseA, *ibaseB;
  auto issue_base = [&]() {
    int m0, n0;
    tile_rc(iti, m0, n0);
    ibaseA = reinterpret_cast<const char*>(p.A + (long)m0 * p.lda);
    ibaseB = reinterpret_cast<const char*>(p.B + (long)n0 * p.ldb);
    const int rows = min(BG_BM, p.M - m0);
    if (rows != irows) {                                         // uniform; entering or leaving the ragged row block
      irows = rows;
      lane_offsets();
    }
  };
  lane_offsets();
  auto issue_a = [&](int half) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(ibaseA + ikt * (BG_KT * 2) + offA[half][i]),
                                       (lds_void_t*)(smem + (iset * 4 + half) * BG_UNIT + (wave * 2 + i) * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int half) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(ibaseB + ikt * (BG_KT * 2) + offB[half][i]),
                                       (lds_void_t*)(smem + (iset * 4 + 2 + half) * BG_UNIT + (wave * 2 + i) * 1024), 16, 0, 0);
  };
  auto issue_advance = [&]() {                                   // after B1, A1: the next K-tile, other set
    iset ^= 1;
    if (++ikt == nkt) {
      ikt = 0;
      if (++iti < my_tiles) issue_base();
    }
  };
  // prologue: K-tile 0 whole, A0 B0 of K-tile 1 (total_kt >= nkt >= 4)
  issue_base();
  issue_a(0); issue_b(0); issue_b(1); issue_a(1);
  issue_advance();
  issue_a(0); issue_b(0);

  __builtin_amdgcn_s_waitcnt(bg_vmcnt(6));                       // A0, B0, B1 of K-tile 0 landed (A1, A0', B0' in flight)
  __builtin_amdgcn_s_barrier();                                  // barrier 0
  if (wr == 1) __builtin_amdgcn_s_barrier();                     // half 1 runs one phase-half behind

  f32x4 acc[MT][NT];
  bf16x8 af[8], bfr[8];          // A fragments [row tile rt][k-step h] of one A unit; B fragments [half j][col tile ct][h]
  int cset = 0;
  auto read_a = [&](int half) {
    const unsigned char* U = smem + (cset * 4 + half) * BG_UNIT;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int ur = wr * 64 + rt * 16 + c16;
#pragma unroll
      for (int h = 0; h < 2; ++h) af[rt * 2 + h] = *reinterpret_cast<const bf16x8*>(U + ur * 128 + (((h * 4 + ch) ^ ((ur >> 1) & 7)) << 4));
    }
  };
  auto read_b = [&](int half) {
    const unsigned char* U = smem + (cset * 4 + 2 + half) * BG_UNIT;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int ur = wc * 32 + ct * 16 + c16;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        bfr[half * 4 + ct * 2 + h] = *reinterpret_cast<const bf16x8*>(U + ur * 128 + (((h * 4 + ch) ^ ((ur >> 1) & 7)) << 4));
    }
  };
  // The two halves of a phase: [fragment reads, two units requested, reads returned | barrier | 32 MFMAs | barrier].  What
  // the NEXT phase reads must have landed before the second barrier: half 1 waits for it in its read part, half 0 after
  // its MFMAs (both sit before the same barrier).  `left` = entries of this wave's queue that may stay outstanding.
  auto sync_a = [&](int left) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wr == 1) {
      if (left == 8) __builtin_amdgcn_s_waitcnt(bg_vmcnt(8));
      else if (left == 6) __builtin_amdgcn_s_waitcnt(bg_vmcnt(6));
      else if (left == 2) __builtin_amdgcn_s_waitcnt(bg_vmcnt(2));
      else if (left == 0) __builtin_amdgcn_s_waitcnt(bg_vmcnt(0));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto sync_b = [&](int left) {
    if (wr == 0) {
      if (left == 8) __builtin_amdgcn_s_waitcnt(bg_vmcnt(8));
      else if (left == 6) __builtin_amdgcn_s_waitcnt(bg_vmcnt(6));
      else if (left == 2) __builtin_amdgcn_s_waitcnt(bg_vmcnt(2));
      else if (left == 0) __builtin_amdgcn_s_waitcnt(bg_vmcnt(0));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
#define BG_MFMA(AI, BJ)                                                                                                       \
  do {                                                                                                                        \
    _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                          \
      _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                                        \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                         \
          acc[(AI) * 4 + rt][(BJ) * 2 + ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                        \
              bfr[(BJ) * 4 + ct * 2 + h], af[rt * 2 + h], acc[(AI) * 4 + rt][(BJ) * 2 + ct], 0, 0, 0);                        \
  } while (0)

  int g = 0;                     // K-tiles done (this workgroup's stream)
  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; ++kt, ++g) {
      // Queue of this wave at the two wait points of K-tile g (2 entries per unit), oldest first:
      //   end of X(g): A1[g] | A0 B0 B1 A1 of g+1   -> A1[g] landed <=> at most 8 left (fewer near the end of the stream)
      //   end of Y(g): B1 A1 of g+1 | A0 B0 of g+2  -> B1[g+1] landed <=> at most 6 left
      // Behind an epilogue (vmcnt(0): everything requested before it has landed) the first X of a tile does not wait: a
      // counted wait there would wait for the epilogue's stores.
      const bool fresh = ti > 0 && kt == 0;
      // phase X: quadrants (A0, B0), (A0, B1); requests B1, A1 of K-tile g+1
      read_b(0);
      read_b(1);
      read_a(0);
      const bool more1 = g + 1 < total_kt;
      if (more1) { issue_b(1); issue_a(1); issue_advance(); }
      sync_a(fresh ? -1 : more1 ? 8 : 0);
      __builtin_amdgcn_s_setprio(1);
      BG_MFMA(0, 0);
      BG_MFMA(0, 1);
      __builtin_amdgcn_s_setprio(0);
      sync_b(fresh ? -1 : more1 ? 8 : 0);
      // phase Y: (A1, B1), (A1, B0); requests A0, B0 of K-tile g+2
      read_a(1);
      const bool more2 = g + 2 < total_kt;
      if (more2) { issue_a(0); issue_b(0); }
      sync_a(more2 ? 6 : more1 ? 2 : -1);
      __builtin_amdgcn_s_setprio(1);
      BG_MFMA(1, 1);
      BG_MFMA(1, 0);
      __builtin_amdgcn_s_setprio(0);
      sync_b(more2 ? 6 : more1 ? 2 : -1);
      cset ^= 1;
    }
    // Epilogue, in program order behind the tile's last barrier: it runs in the partner half's MFMA phase (half 0) or
    // under the partner's first phase of the next tile (half 1).  Every load of the tail first (the fragment registers
    // are dead here), one wait -- which also retires the units already requested for the next tile -- then 16 stores.
    {
      int m0, n0;
      tile_rc(ti, m0, n0);
      const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
      EpiRegs<MT, NT, EPI> R;
      R.rng = rng;
      epi_load_early<MT, NT, EPI>(p, R, row0, col0, lane);
      __builtin_amdgcn_s_waitcnt(bg_vmcnt(0));
      epi_finish<MT, NT, EPI>(p, acc, R, row0, col0, lane);
    }
    lane_offsets();
  }
#undef BG_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();                     // matches half 1's extra barrier at the start
}

}  // namespace

// Called by iq_gemm_bf16_nt with its resolved parameters (bias non-null).  Returns false when the shape / epilogue is not
// this kernel's; true after a launch.
bool gemm_big_try(const GemmParams& p0, int epi_mode, hipStream_t st) {
  if (epi_mode != 0 && epi_mode != EPI_RES && epi_mode != EPI_GATE) return false;
  if (p0.N % BG_BN != 0 || p0.K % BG_KT != 0 || p0.K < 256 || p0.M < 2048) return false;
  if ((((uintptr_t)p0.A | (uintptr_t)p0.B | (uintptr_t)p0.C) % 16) || (p0.ldc % 8)) return false;
  GemmParams p = p0;
  p.tiles_m = (p.M + BG_BM - 1) / BG_BM;
  p.tiles_n = p.N / BG_BN;
  const int ntiles = p.tiles_m * p.tiles_n;
  if (ntiles < 512) return false;                          // fewer than two rounds of an MI355X's 256 CUs: the 128 x 128 tiles fill the chip better
  // one workgroup per CU (128 KiB of LDS each), persistent over its share of the tiles
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int grid = cus;
  const size_t lds = (size_t)8 * BG_UNIT;                  // 128 KiB
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
