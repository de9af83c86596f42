// Weight-gradient GEMMs with LDS-shared 256-row output tiles: slab[split][N,K] = dY[rows of the split, N]^T * X[rows, K]
// (+ column sums of dY) for up to four problems that share M, in one launch (gfx950) -- the autograd weight / bias
// gradients of the four Linear layers of an encoder layer (multi_head_attention.py:11-14,
// position_wise_feed_forward.py:7-8).  Same schedule as gemm_big.hip, with the contraction index (tokens) as the ROW
// index of both operands:
//   * a problem is oriented so that its output is n' x k' with k' a multiple of the group's column tile TK (128 | 192 |
//     256: the model width D of the raw-IQ encoder / ViT-Tiny, or a third of ViT-Base's) and n' the longer side: the
//     N x K gradient itself, or its transpose (operands swapped, stores transposed) for the Linear whose N is the width;
//   * output tile 256 (n') x TK per workgroup, 8 waves as 2 (n') x 4 (k'), 128 x TK/4 per wave; one (tile, M-split) per
//     workgroup, one workgroup per CU (grid = tiles x splits ~ 256); fp32 partial tiles go to slabs that
//     wgrad_reduce_kernel sums in fixed order (bit-reproducible, no atomics);
//   * a 64-row step of the contraction is four units in whole 128-byte lines: A0 / A1 = 64 rows x the first / second 64
//     A columns of each wave row (two 128 B segments per row, 16 KiB), B0 / B1 = rows 0-31 / 32-63 x all TK B columns
//     (256 / 384 / 512 B rows); both MFMA operands need "8 consecutive tokens at a fixed column": fragments come from
//     the row-major images through ds_read_b64_tr_b16, 16-byte chunks XOR-swizzled on the global side by an even number
//     derived from the row (keeps the 32-byte blocks a transposing read fetches together, spreads the 8 rows of a
//     32-lane group over all 64 banks);
//   * two phases per step (X: A0 x B, Y: A1 x B with the B fragments kept in registers), ping-pong wave halves, units
//     requested 2-3 phases ahead with counted vmcnt, no address arithmetic in the loop (gemm_big.hip has the details);
//   * bias gradient: extra MFMAs against an all-ones fragment (A side: one per phase and k-step, the four waves of a row
//     sharing the n-tiles; B side for a transposed problem: the two wave rows sharing the k-tiles), only in the
//     workgroups of the first tile column / row.
// Against the wave-private kernel of gemm_wgrad.hip (64..128-wide private tiles, 384 B of L2 traffic per MFMA) a 256 x 192
// tile moves 155 B per MFMA, by DMA instead of register staging.
#include "common.h"
#include <string.h>
#include <type_traits>
#include "iqvit.h"
#include "prof.h"
#include "gemm_wgrad_big.h"

namespace {

constexpr int WB_THREADS = 512, WB_TN = 256, WB_MT = 64, WB_AUNIT = 16384, WB_MAXP = 4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int wb_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

struct WbProb {
  const bf16* A; const bf16* B;      // A: its columns are the output rows n', B: the output columns k'
  int lda, ldb, NR, NC;              // valid output rows (multiple of 64) / columns (multiple of TK)
  int ldo, transposed;               // slab element (n', k') at n' * ldo + k', or at k' * ldo + n' when transposed
  int bias_side;                     // 0 none | 1 column sums of A (indexed by n') | 2 column sums of B (indexed by k')
  float* slab;                       // [splits][NR * NC]
  float* bslab;                      // [splits][bias length]
  long slab_stride, bslab_stride;    // floats per split
  int tiles_k, tile0;
};
struct WbGroup { WbProb pr[WB_MAXP]; int nprob, M, ntile, rows_per_split; };

// chunk swizzle of a row-major image with `rowb`-byte rows (see the header): an even number < 8 or < 16
template <int ROWB> __device__ __forceinline__ int wb_swz(int row) { return ROWB == 384 ? 2 * ((row >> 1) & 3) : 2 * (row & 7); }

template <int TK>
__global__ __launch_bounds__(WB_THREADS, 1) void wgrad_big_kernel(const WbGroup g) {
  constexpr int CT = TK / 64;                         // 16-column k'-tiles per wave (wave = 128 n' x TK/4 k')
  constexpr int BROWB = TK * 2, BCPR = TK / 8;        // B image: bytes / 16-byte chunks per row
  constexpr int BUNIT = 32 * BROWB;                   // 8 / 12 / 16 KiB
  constexpr int BPIECES = BUNIT / 1024;               // 8 / 12 / 16
  constexpr int PB = BPIECES > 8 ? 2 : 1;             // DMA instructions per wave per B unit (TK = 192: 4 waves repeat a piece)
  constexpr int SET = 2 * WB_AUNIT + 2 * BUNIT;
  constexpr int L_X = 4 + 2 * PB, L_Y = 4 + PB;       // queue entries that may stay outstanding at the two wait points
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);   // the tiles of one split are neighbours on one XCD: they share rows
  const int split = lid / g.ntile, gt = lid % g.ntile;
  WbProb p = g.pr[0];
#pragma unroll
  for (int i = 1; i < WB_MAXP; ++i)
    if (i < g.nprob && gt >= g.pr[i].tile0) p = g.pr[i];
  const int tile = gt - p.tile0;
  const int tn = tile / p.tiles_k, tk = tile % p.tiles_k;
  const int n0 = tn * WB_TN, k0 = tk * TK;
  const int mbeg = split * g.rows_per_split;
  const int mend = min(g.M, mbeg + g.rows_per_split);
  const int total = (mend - mbeg) / WB_MT;            // 64-row steps of this workgroup (whole: M % 64 == 0)
  const bool bias_a = p.bias_side == 1 && tk == 0, bias_b = p.bias_side == 2 && tn == 0;

  // ---- unit requests: uniform base (split rows, tile columns) + per-lane offsets that do not change ---------------------
  // A unit (half h): image row m = [64 columns of wave row 0 | 64 columns of wave row 1] (256 B); piece = 4 rows.  A
  // 64-column segment that starts past the problem's last row (ragged n' tile) is moved onto the last valid one: those
  // accumulator rows are never stored.
  unsigned offA[2][2], offB[2][PB];                   // [half][piece], bytes from the step's first row
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = wave * 2 + i;                   // 0..15
    const int row = piece * 4 + (lane >> 4);          // 0..63
    const int cs = (lane & 15) ^ wb_swz<256>(row);    // source chunk of this lane's LDS position
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int seg = min(n0 + (cs >> 3) * 128 + h * 64, p.NR - 64);
      offA[h][i] = (unsigned)(row * p.lda * 2 + (seg + (cs & 7) * 8) * 2);
    }
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    int piece = wave + 8 * i;
    if (piece >= BPIECES) piece = wave;               // TK = 192: waves 4-7 request their first piece twice (uniform counts)
    const int c = piece * 64 + lane;                  // chunk of the unit, row-major
    const int row = c / BCPR, pos = c - row * BCPR;
    const int cs = pos ^ wb_swz<BROWB>(row);
#pragma unroll
    for (int j = 0; j < 2; ++j) offB[j][i] = (unsigned)((j * 32 + row) * p.ldb * 2 + (k0 + cs * 8) * 2);
  }
  const char* baseA = reinterpret_cast<const char*>(p.A + (long)mbeg * p.lda);
  const char* baseB = reinterpret_cast<const char*>(p.B + (long)mbeg * p.ldb);
  const long stepA = (long)WB_MT * p.lda * 2, stepB = (long)WB_MT * p.ldb * 2;
  int iset = 0;
  auto issue_a = [&](int h) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(baseA + offA[h][i]), (lds_void_t*)(smem + iset * SET + h * WB_AUNIT + (wave * 2 + i) * 1024),
                                       16, 0, 0);
  };
  auto issue_b = [&](int j) {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int piece = wave + 8 * i;
      if (piece >= BPIECES) piece = wave;
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(baseB + offB[j][i]),
                                       (lds_void_t*)(smem + iset * SET + 2 * WB_AUNIT + j * BUNIT + piece * 1024), 16, 0, 0);
    }
  };
  auto issue_advance = [&]() { iset ^= 1; baseA += stepA; baseB += stepB; };

  f32x4 acc[8][CT], accb[2], accbt[CT];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < CT; ++j) accbt[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
  if (total <= 0) return;                             // (uniform; the host never creates an empty split)

  // prologue: step 0 whole, A0 B0 of step 1
  issue_a(0); issue_b(0); issue_b(1); issue_a(1);
  issue_advance();
  if (total > 1) { issue_a(0); issue_b(0); }
  if (total > 1) __builtin_amdgcn_s_waitcnt(wb_vmcnt(L_Y)); else __builtin_amdgcn_s_waitcnt(wb_vmcnt(2));
  __builtin_amdgcn_s_barrier();                       // barrier 0: A0 B0 B1 of step 0 visible
  if (wr == 1) __builtin_amdgcn_s_barrier();          // half 1 runs one phase-half behind

  const int i16 = lane & 15, g4 = lane >> 4;
  bf16x8 af[8], bfr[CT * 2];     // A fragments [n'-tile rt][m-step s]; B fragments [k'-tile ct][m-step s]
  int cset = 0;
  // transposing fragment: 8 tokens {4 g + q} U {16 + 4 g + q} of a 32-row m-step at column `col` + (lane & 15)
  auto tr = [&](const unsigned char* U, auto rowb_tag, int r0, int col) -> bf16x8 {
    constexpr int ROWB = decltype(rowb_tag)::value;
    const int row = r0 + 4 * g4 + (i16 >> 2), c = col + 4 * (i16 & 3);
    const int chunk = (c >> 3) ^ wb_swz<ROWB>(row);   // swz(row + 16) == swz(row): one swizzle for both halves
    const unsigned char* a = U + row * ROWB + chunk * 16 + (c & 7) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 16 * ROWB));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto read_a = [&](int h) {
    const unsigned char* U = smem + cset * SET + h * WB_AUNIT;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int s = 0; s < 2; ++s) af[rt * 2 + s] = tr(U, std::integral_constant<int, 256>{}, s * 32, wr * 64 + rt * 16);
  };
  auto read_b = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {                     // m-step s lives in unit B_s
      const unsigned char* U = smem + cset * SET + 2 * WB_AUNIT + s * BUNIT;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bfr[ct * 2 + s] = tr(U, std::integral_constant<int, BROWB>{}, 0, wc * (TK / 4) + ct * 16);
    }
  };
  auto wait_left = [&](int left) {
    if (left == L_X) __builtin_amdgcn_s_waitcnt(wb_vmcnt(L_X));
    else if (left == L_Y) __builtin_amdgcn_s_waitcnt(wb_vmcnt(L_Y));
    else if (left == 2) __builtin_amdgcn_s_waitcnt(wb_vmcnt(2));
    else if (left == 0) __builtin_amdgcn_s_waitcnt(wb_vmcnt(0));
  };
  auto sync_a = [&](int left) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wr == 1) wait_left(left);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto sync_b = [&](int left) {
    if (wr == 0) wait_left(left);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
#define WB_MFMA(H)                                                                                                           \
  do {                                                                                                                       \
    _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                         \
      _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                                                      \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                        \
          acc[(H) * 4 + rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt * 2 + s], bfr[ct * 2 + s], acc[(H) * 4 + rt][ct], 0, 0, 0); \
    if (bias_a) {                                                                                                            \
      _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                                       \
        if (rt == wc) {                                                                                                      \
          _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                      \
            accb[H] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt * 2 + s], ones, accb[H], 0, 0, 0);                       \
        }                                                                                                                    \
    }                                                                                                                        \
  } while (0)

  for (int q = 0; q < total; ++q) {
    // queue of this wave at the two wait points of step q, oldest first (gemm_big.hip):
    //   end of X(q): A1[q] | A0 B0 B1 A1 of q+1    -> A1[q] landed <=> at most L_X left
    //   end of Y(q): B1 A1 of q+1 | A0 B0 of q+2   -> B1[q+1] landed <=> at most L_Y left
    const bool more1 = q + 1 < total, more2 = q + 2 < total;
    // phase X: n'-half 0 of every wave; requests B1, A1 of step q+1
    read_b();
    read_a(0);
    if (more1) { issue_b(1); issue_a(1); issue_advance(); }
    sync_a(more1 ? L_X : 0);
    __builtin_amdgcn_s_setprio(1);
    WB_MFMA(0);
    if (bias_b) {                                      // column sums of B: the two wave rows share the k'-tiles
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        if ((ct & 1) == wr) {
#pragma unroll
          for (int s = 0; s < 2; ++s) accbt[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[ct * 2 + s], accbt[ct], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    sync_b(more1 ? L_X : 0);
    // phase Y: n'-half 1; requests A0, B0 of step q+2
    read_a(1);
    if (more2) { issue_a(0); issue_b(0); }
    sync_a(more2 ? L_Y : more1 ? 2 : -1);
    __builtin_amdgcn_s_setprio(1);
    WB_MFMA(1);
    __builtin_amdgcn_s_setprio(0);
    sync_b(more2 ? L_Y : more1 ? 2 : -1);
    cset ^= 1;
  }
#undef WB_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();          // matches half 1's extra barrier at the start

  // ---- partial tile -> slab: lane (g4, i16) holds C[n' = 4 g4 + r][k' = i16] of every 16 x 16 tile ----------------------
  float* out = p.slab + (long)split * p.slab_stride;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int nb = n0 + wr * 128 + i * 16 + g4 * 4;   // first of this lane's 4 rows; NR % 64 == 0: all 4 valid or none
    if (nb < p.NR) {
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int k = k0 + wc * (TK / 4) + j * 16 + i16;
        if (p.transposed) {
          *reinterpret_cast<f32x4*>(out + (long)k * p.ldo + nb) = acc[i][j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) out[(long)(nb + r) * p.ldo + k] = acc[i][j][r];
        }
      }
    }
  }
  if (bias_a && i16 == 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int nb = n0 + wr * 128 + h * 64 + wc * 16 + g4 * 4;
      if (nb < p.NR) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p.bslab[(long)split * p.bslab_stride + nb + r] = accb[h][r];
      }
    }
  }
  if (bias_b && g4 == 0) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      if ((ct & 1) == wr) p.bslab[(long)split * p.bslab_stride + k0 + wc * (TK / 4) + ct * 16 + i16] = accbt[ct][0];
  }
}

inline size_t wb_pad4(size_t v) { return (v + 3) / 4 * 4; }

// padded work of one orientation: n' rounded up to whole 256-row tiles
inline long wb_area(int nr, int nc) { return (long)((nr + WB_TN - 1) / WB_TN) * WB_TN * nc; }

struct WbOrient { bool ok, transposed; };
inline WbOrient wb_orient(int N, int K, int tk) {
  const bool a = (K % tk) == 0 && (N % 64) == 0;       // as is: n' = N, k' = K
  const bool b = (N % tk) == 0 && (K % 64) == 0;       // transposed: n' = K, k' = N
  WbOrient o = {a || b, false};
  if (a && b) o.transposed = wb_area(K, N) < wb_area(N, K);
  else if (b) o.transposed = true;
  return o;
}

}  // namespace

// ---- host side (used by gemm_wgrad.hip) --------------------------------------------------------------------------------------

// false when the group does not fit this kernel (then the wave-private / shared-tile kernels of gemm_wgrad.hip run)
bool wgrad_big_plan(const iq_wgrad_problem_t* pr, int nprob, int M, WbPlan* out) {
  if (nprob < 1 || nprob > WB_MAXP || M % WB_MT != 0 || M < 64 * WB_MT) return false;
  long best = -1;
  WbPlan w;
  memset(&w, 0, sizeof(w));
  for (int tk : {256, 192, 128}) {
    long area = 0;
    bool ok = true;
    bool tr[WB_MAXP];
    for (int i = 0; i < nprob && ok; ++i) {
      const WbOrient o = wb_orient(pr[i].N, pr[i].K, tk);
      ok = o.ok && (pr[i].ldy % 64) == 0 && (pr[i].ldx % 64) == 0 && (((uintptr_t)pr[i].dY | (uintptr_t)pr[i].X) % 128) == 0;
      tr[i] = o.transposed;
      area += o.transposed ? wb_area(pr[i].K, pr[i].N) : wb_area(pr[i].N, pr[i].K);
    }
    if (ok && (best < 0 || area < best)) {
      best = area;
      w.tk = tk;
      for (int i = 0; i < nprob; ++i) w.transposed[i] = tr[i];
    }
  }
  if (best < 0) return false;
  w.ntile = 0;
  for (int i = 0; i < nprob; ++i) {
    const int nr = w.transposed[i] ? pr[i].K : pr[i].N, nc = w.transposed[i] ? pr[i].N : pr[i].K;
    w.ntile += ((nr + WB_TN - 1) / WB_TN) * (nc / w.tk);
  }
  int splits = 256 / w.ntile;                              // one workgroup per CU, the chip filled once
  if (splits < 1) splits = 1;
  const int steps = M / WB_MT;
  if (splits > steps / 8) splits = steps / 8 > 0 ? steps / 8 : 1;      // at least 8 steps per workgroup
  const int sps = (steps + splits - 1) / splits;
  w.rows_per_split = sps * WB_MT;
  w.splits = (steps + sps - 1) / sps;
  w.floats = 0;
  for (int i = 0; i < nprob; ++i) w.floats += (size_t)w.splits * (wb_pad4((size_t)pr[i].N * pr[i].K) + wb_pad4(pr[i].N));
  *out = w;
  return true;
}

// Launches the partial-tile kernel for the group; slab[i] / bslab[i] (splits rows of N*K resp. N floats, strides in
// *stride_w / *stride_b) tell the caller what to reduce.
void wgrad_big_launch(const iq_wgrad_problem_t* pr, int nprob, int M, const WbPlan& w, float* ws, float** slab, float** bslab,
                      hipStream_t st) {
  WbGroup g;
  memset(&g, 0, sizeof(g));
  g.nprob = nprob; g.M = M; g.ntile = w.ntile; g.rows_per_split = w.rows_per_split;
  float* cur = ws;
  int tile0 = 0;
  for (int i = 0; i < nprob; ++i) {
    WbProb& q = g.pr[i];
    const bool t = w.transposed[i];
    q.A = (const bf16*)(t ? pr[i].X : pr[i].dY); q.lda = t ? pr[i].ldx : pr[i].ldy;
    q.B = (const bf16*)(t ? pr[i].dY : pr[i].X); q.ldb = t ? pr[i].ldy : pr[i].ldx;
    q.NR = t ? pr[i].K : pr[i].N; q.NC = t ? pr[i].N : pr[i].K;
    q.ldo = pr[i].K; q.transposed = t ? 1 : 0;
    q.bias_side = pr[i].dbias ? (t ? 2 : 1) : 0;
    q.tiles_k = q.NC / w.tk;
    q.tile0 = tile0;
    tile0 += ((q.NR + WB_TN - 1) / WB_TN) * q.tiles_k;
    q.slab_stride = (long)wb_pad4((size_t)pr[i].N * pr[i].K);
    q.slab = cur; cur += (size_t)w.splits * q.slab_stride;
    slab[i] = q.slab;
    q.bslab_stride = (long)wb_pad4(pr[i].N);
    q.bslab = nullptr;
    if (pr[i].dbias) { q.bslab = cur; cur += (size_t)w.splits * q.bslab_stride; }
    bslab[i] = q.bslab;
  }
  const int grid = w.ntile * w.splits;
#define IQ_WB_LAUNCH(TK_)                                                                                               \
  do {                                                                                                                  \
    constexpr int lds = 2 * (2 * WB_AUNIT + 2 * 32 * TK_ * 2);                                                          \
    auto k = wgrad_big_kernel<TK_>;                                                                                     \
    static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    (void)attr;                                                                                                         \
    k<<<grid, WB_THREADS, lds, st>>>(g);                                                                                \
  } while (0)
  if (w.tk == 256) IQ_WB_LAUNCH(256);
  else if (w.tk == 192) IQ_WB_LAUNCH(192);
  else IQ_WB_LAUNCH(128);
#undef IQ_WB_LAUNCH
}
