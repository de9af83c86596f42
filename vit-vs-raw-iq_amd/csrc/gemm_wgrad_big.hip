// Weight-gradient GEMM for the MFMA-bound shapes: slab[split][N,K] = dY[rows of the split, N]^T * X[rows, K] (+ column
// sums of dY), N % 256 == 0, K % 256 == 0, M % 64 == 0 (gfx950) -- the autograd weight / bias gradients of every Linear
// of ViT-Base (multi_head_attention.py:11-14, position_wise_feed_forward.py:7-8 at D768 / F3072).  Same schedule as
// gemm_big.hip, with the contraction index (tokens) as the ROW index of both operands:
//   * output tile 256 (n) x 256 (k) per workgroup, 8 waves as 2 (n) x 4 (k), 128 x 64 per wave, one (tile, M-split) per
//     workgroup, one workgroup per CU (grid = tiles x splits ~ 256); fp32 partial tiles go to slabs that
//     wgrad_reduce_kernel sums in fixed order (bit-reproducible, no atomics), as for the smaller shapes;
//   * a 64-row step of the contraction is four 16 KiB units in whole 128-byte lines: A0 / A1 = 64 rows x the first /
//     second 64 dY columns of each wave row (two 128 B segments per row), B0 / B1 = rows 0-31 / 32-63 x all 256 X
//     columns (512 B rows); global_load_lds pieces of 4 rows x 256 B resp. 2 rows x 512 B;
//   * both MFMA operands need "8 consecutive tokens at a fixed column": fragments come from the row-major images through
//     ds_read_b64_tr_b16; 16-byte chunks are XOR-swizzled on the global side with 2 * (row & 7), which keeps the 32-byte
//     blocks a transposing read fetches together and spreads the 8 rows of a 32-lane group over all 64 banks;
//   * two phases per step (X: A0 x B, Y: A1 x B with the B fragments kept in registers), ping-pong wave halves, units
//     requested 2-3 phases ahead with counted vmcnt, no address arithmetic in the loop (gemm_big.hip has the details);
//   * bias gradient: one extra MFMA per phase and k-step against an all-ones fragment, the four waves of a row sharing
//     the work (wave wc takes n-tile wc of the phase's A half), only in the workgroups of the first k-tile column.
#include "common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int WB_THREADS = 512, WB_T = 256, WB_MT = 64, WB_UNIT = 16384;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int wb_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

struct WbParams {
  const bf16* Y; const bf16* X;
  int ldy, ldx, M, N, K;
  float* slab;       // [splits][N*K]
  float* bslab;      // [splits][N] or null
  int tiles_n, tiles_k, splits, rows_per_split;
};

__global__ __launch_bounds__(WB_THREADS, 1) void wgrad_big_kernel(const WbParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntile = p.tiles_n * p.tiles_k;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);      // the tiles of one split are neighbours on one XCD: they share rows
  const int split = lid / ntile, tile = lid % ntile;
  const int n0 = (tile / p.tiles_k) * WB_T, k0 = (tile % p.tiles_k) * WB_T;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);
  const int total = (mend - mbeg) / WB_MT;               // 64-row steps of this workgroup (whole: M % 64 == 0)
  const bool do_bias = p.bslab != nullptr && (tile % p.tiles_k) == 0;

  // ---- unit requests: uniform base (split rows, tile columns) + per-lane offsets that do not change ---------------------
  // A unit (half h): LDS row m = [64 cols of wave row 0 | 64 cols of wave row 1] (256 B); piece = 4 rows, lane -> (row, chunk)
  // B unit (half j): rows 32 j .. 32 j + 31, all 256 X columns (512 B); piece = 2 rows
  unsigned offA[2][2], offB[2][2];                       // [half][piece], bytes from the step's first row
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = wave * 2 + i;                      // 0..15
    {
      const int row = piece * 4 + (lane >> 4);           // 0..63
      const int cs = (lane & 15) ^ (2 * (row & 7));      // source chunk of this lane's LDS position
#pragma unroll
      for (int h = 0; h < 2; ++h)
        offA[h][i] = (unsigned)(row * p.ldy * 2 + ((cs >> 3) * 128 + h * 64 + (cs & 7) * 8) * 2);
    }
    {
      const int row = piece * 2 + (lane >> 5);           // 0..31 inside the half
      const int cs = (lane & 31) ^ (2 * (row & 7));
#pragma unroll
      for (int j = 0; j < 2; ++j) offB[j][i] = (unsigned)((j * 32 + row) * p.ldx * 2 + cs * 16);
    }
  }
  const char* baseA = reinterpret_cast<const char*>(p.Y + (long)mbeg * p.ldy + n0);
  const char* baseB = reinterpret_cast<const char*>(p.X + (long)mbeg * p.ldx + k0);
  const long stepA = (long)WB_MT * p.ldy * 2, stepB = (long)WB_MT * p.ldx * 2;
  int iset = 0;
  auto issue_a = [&](int h) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(baseA + offA[h][i]), (lds_void_t*)(smem + (iset * 4 + h) * WB_UNIT + (wave * 2 + i) * 1024),
                                       16, 0, 0);
  };
  auto issue_b = [&](int j) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(baseB + offB[j][i]),
                                       (lds_void_t*)(smem + (iset * 4 + 2 + j) * WB_UNIT + (wave * 2 + i) * 1024), 16, 0, 0);
  };
  auto issue_advance = [&]() { iset ^= 1; baseA += stepA; baseB += stepB; };

  f32x4 acc[8][4], accb[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
  if (total <= 0) return;                                // (uniform; the host never creates an empty split)

  // prologue: step 0 whole, A0 B0 of step 1
  issue_a(0); issue_b(0); issue_b(1); issue_a(1);
  issue_advance();
  if (total > 1) { issue_a(0); issue_b(0); }
  if (total > 1) __builtin_amdgcn_s_waitcnt(wb_vmcnt(6)); else __builtin_amdgcn_s_waitcnt(wb_vmcnt(2));
  __builtin_amdgcn_s_barrier();                          // barrier 0: A0 B0 B1 of step 0 visible
  if (wr == 1) __builtin_amdgcn_s_barrier();             // half 1 runs one phase-half behind

  const int i16 = lane & 15, g4 = lane >> 4;
  bf16x8 af[8], bfr[8];          // A fragments [n-tile rt][m-step s]; B fragments [k-tile ct][m-step s]
  int cset = 0;
  // transposing fragment: 8 tokens {4 g + q} U {16 + 4 g + q} of m-step s at column `col` + (lane & 15); rows of `rowb` bytes
  auto tr = [&](const unsigned char* U, int rowb, int r0, int col) -> bf16x8 {
    const int row = r0 + 4 * g4 + (i16 >> 2), c = col + 4 * (i16 & 3);
    const int chunk = (c >> 3) ^ (2 * (row & 7));        // (row + 16) & 7 == row & 7: one swizzle for both halves
    const unsigned char* a = U + row * rowb + chunk * 16 + (c & 7) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * rowb));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto read_a = [&](int h) {
    const unsigned char* U = smem + (cset * 4 + h) * WB_UNIT;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int s = 0; s < 2; ++s) af[rt * 2 + s] = tr(U, 256, s * 32, wr * 64 + rt * 16);
  };
  auto read_b = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {                        // m-step s lives in unit B_s
      const unsigned char* U = smem + (cset * 4 + 2 + s) * WB_UNIT;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bfr[ct * 2 + s] = tr(U, 512, 0, wc * 64 + ct * 16);
    }
  };
  auto sync_a = [&](int left) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wr == 1) {
      if (left == 8) __builtin_amdgcn_s_waitcnt(wb_vmcnt(8));
      else if (left == 6) __builtin_amdgcn_s_waitcnt(wb_vmcnt(6));
      else if (left == 2) __builtin_amdgcn_s_waitcnt(wb_vmcnt(2));
      else if (left == 0) __builtin_amdgcn_s_waitcnt(wb_vmcnt(0));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto sync_b = [&](int left) {
    if (wr == 0) {
      if (left == 8) __builtin_amdgcn_s_waitcnt(wb_vmcnt(8));
      else if (left == 6) __builtin_amdgcn_s_waitcnt(wb_vmcnt(6));
      else if (left == 2) __builtin_amdgcn_s_waitcnt(wb_vmcnt(2));
      else if (left == 0) __builtin_amdgcn_s_waitcnt(wb_vmcnt(0));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
#define WB_MFMA(H)                                                                                                           \
  do {                                                                                                                       \
    _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                         \
      _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)                                                                       \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                        \
          acc[(H) * 4 + rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt * 2 + s], bfr[ct * 2 + s], acc[(H) * 4 + rt][ct], 0, 0, 0); \
    if (do_bias) {                                                                                                           \
      _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                       \
        if (rt == wc) {                                                                                                      \
          _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                      \
            accb[H] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt * 2 + s], ones, accb[H], 0, 0, 0);                       \
        }                                                                                                                    \
    }                                                                                                                        \
  } while (0)

  for (int g = 0; g < total; ++g) {
    // queue of this wave at the two wait points of step g (2 entries per unit), oldest first (gemm_big.hip):
    //   end of X(g): A1[g] | A0 B0 B1 A1 of g+1    -> A1[g] landed <=> at most 8 left
    //   end of Y(g): B1 A1 of g+1 | A0 B0 of g+2   -> B1[g+1] landed <=> at most 6 left
    const bool more1 = g + 1 < total, more2 = g + 2 < total;
    // phase X: n-half 0 of every wave; requests B1, A1 of step g+1
    read_b();
    read_a(0);
    if (more1) { issue_b(1); issue_a(1); issue_advance(); }
    sync_a(more1 ? 8 : 0);
    __builtin_amdgcn_s_setprio(1);
    WB_MFMA(0);
    __builtin_amdgcn_s_setprio(0);
    sync_b(more1 ? 8 : 0);
    // phase Y: n-half 1; requests A0, B0 of step g+2
    read_a(1);
    if (more2) { issue_a(0); issue_b(0); }
    sync_a(more2 ? 6 : more1 ? 2 : -1);
    __builtin_amdgcn_s_setprio(1);
    WB_MFMA(1);
    __builtin_amdgcn_s_setprio(0);
    sync_b(more2 ? 6 : more1 ? 2 : -1);
    cset ^= 1;
  }
#undef WB_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();             // matches half 1's extra barrier at the start

  // ---- partial tile -> slab: lane (g4, i16) holds C[n = 4 g4 + r][k = i16] of every 16 x 16 tile ------------------------
  float* out = p.slab + (long)split * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wr * 128 + i * 16 + g4 * 4 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) out[(long)n * p.K + k0 + wc * 64 + j * 16 + i16] = acc[i][j][r];
    }
  if (do_bias && i16 == 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) p.bslab[(long)split * p.N + n0 + wr * 128 + h * 64 + wc * 16 + g4 * 4 + r] = accb[h][r];
  }
}

}  // namespace

// ---- host side (used by gemm_wgrad.hip) --------------------------------------------------------------------------------------
struct WbPlan { int tiles_n, tiles_k, splits, rows_per_split; };

bool wgrad_big_eligible(int M, int N, int K) {
  return N % WB_T == 0 && K % WB_T == 0 && M % WB_MT == 0 && M >= 64 * WB_MT && (long)N * K > 512 * 1024;
}
static WbPlan wb_plan(int M, int N, int K) {
  WbPlan w;
  w.tiles_n = N / WB_T; w.tiles_k = K / WB_T;
  const int tiles = w.tiles_n * w.tiles_k;
  int splits = 256 / tiles;                                // one workgroup per CU, the chip filled once
  if (splits < 1) splits = 1;
  const int steps = M / WB_MT;
  if (splits > steps / 8) splits = steps / 8 > 0 ? steps / 8 : 1;      // at least 8 steps per workgroup
  const int sps = (steps + splits - 1) / splits;
  w.rows_per_split = sps * WB_MT;
  w.splits = (steps + sps - 1) / sps;
  return w;
}
size_t wgrad_big_ws_floats(int M, int N, int K) {
  const WbPlan w = wb_plan(M, N, K);
  return (size_t)w.splits * (((size_t)N * K + 3) / 4 * 4 + ((size_t)N + 3) / 4 * 4);
}
// launches the partial-tile kernel; *splits / *bslab tell the caller what to reduce
void wgrad_big_launch(const void* dY, int ldy, const void* X, int ldx, int M, int N, int K, float* ws, bool with_bias, int* splits,
                      float** bslab, hipStream_t st) {
  const WbPlan w = wb_plan(M, N, K);
  WbParams q;
  q.Y = (const bf16*)dY; q.X = (const bf16*)X; q.ldy = ldy; q.ldx = ldx; q.M = M; q.N = N; q.K = K;
  q.slab = ws;
  q.bslab = with_bias ? ws + (size_t)w.splits * (((size_t)N * K + 3) / 4 * 4) : nullptr;
  q.tiles_n = w.tiles_n; q.tiles_k = w.tiles_k; q.splits = w.splits; q.rows_per_split = w.rows_per_split;
  static const hipError_t attr = hipFuncSetAttribute((const void*)wgrad_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WB_UNIT);
  (void)attr;
  wgrad_big_kernel<<<w.tiles_n * w.tiles_k * w.splits, WB_THREADS, 8 * WB_UNIT, st>>>(q);
  *splits = w.splits;
  *bslab = q.bslab;
}
