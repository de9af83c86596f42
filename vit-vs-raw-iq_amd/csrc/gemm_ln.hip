// bf16 MFMA GEMM with the post-norm tail of an encoder sub-layer fused into its epilogue (gfx950):
//
//     Z[M,D] = dropout(A[M,K] * W[D,K]^T + bias) + R          (bf16, kept for backward)
//     X[M,D] = gamma * (Z - mean) * rstd + beta                (bf16),  mean / rstd fp32 [M] (kept for backward)
//
// Replaces   x = norm1(dropout1(attention(x)) + x)   and   x = norm2(dropout2(ffn(x)) + x)
// (V/models/blocks/encoder_layer.py:24-25, 32-33 with LayerNorm.forward, V/models/layers/layers_norm.py:11-19:
// biased variance, eps inside the square root): the out-projection / second FFN GEMM and the LayerNorm that follows
// it are ONE launch.  The separate LayerNorm kernel re-read Z from HBM (M*D*2 bytes per call, 24 calls per cfg B step)
// and cost a launch; here the row statistics come from the tile that is already in registers.
//
// Whole-row tile: one workgroup owns BMT rows x all D columns (D = BN in {128, 192, 256}), 4 waves as 2 (rows) x 2
// (column halves).  Main loop = the global_load_lds ring of gemm_nt.hip (3 slots, 32-deep stages, counted vmcnt, raw
// s_barrier).  The tail's global loads (bias, residual rows) are issued when the LAST operand stage has been issued,
// so they travel under the last two stages' MFMAs instead of being exposed after the loop (they are the youngest
// entries of the in-order vmcnt queue, so the stage waits simply leave them outstanding).
// LayerNorm statistics are taken from the bf16-ROUNDED Z (what backward re-reads), two-pass (mean, then centred sum
// of squares) in fp32: lane-local sums -> 2 shuffles across the 4 lane groups that share a row -> one LDS exchange
// between the two column-half waves.  Same arithmetic as ln_fwd_kernel; results agree to fp32 summation order.
#include <stdlib.h>

#include "common.h"
#include "gemm_common.h"
#include "iqvit.h"
#include "prof.h"

namespace {

constexpr int LNG_THREADS = 256;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

struct GemmLnParams {
  const bf16* A; const bf16* B;            // [M,K], [D,K]
  bf16* Z; bf16* X;                        // [M,D]
  int lda, ldb, M, K;
  const float* bias;                       // [D]
  int drop_on; IqRng rng; uint32_t thresh; float dscale;
  const bf16* residual; int ldr;           // [M,D]
  const float* gamma; const float* beta;   // [D]
  float* mean; float* rstd;                // [M]
  float eps;
};

__device__ __forceinline__ int lswz64(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}: gemm_nt.hip

template <int BMT, int BN>
__global__ __launch_bounds__(LNG_THREADS, 2) void gemm_ln_kernel(const GemmLnParams p) {
  constexpr int BK2 = 32, NS = 3;
  constexpr int WN = BN / 2, NT = WN / 16, NP = NT / 2, MT = BMT / 32;   // wave tile = BMT/2 rows x BN/2 columns
  constexpr int STAGE_BYTES = (BMT + BN) * BK2 * 2;
  constexpr int A_LD = BMT * BK2 * 2 / (4 * 1024);     // 1 KiB DMA pieces per wave per stage
  constexpr int B_LD = BN * BK2 * 2 / (4 * 1024);
  constexpr int PER_STAGE = A_LD + B_LD;
  constexpr int TAIL_LOADS = MT * NP + 2 * NP;         // residual rows + bias vectors, per lane
  static_assert(NT % 2 == 0 && A_LD >= 1 && B_LD >= 1, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, c16 = lane & 15;
  const bool odd = (g & 1) != 0;
  const int m0 = blockIdx.x * BMT;
  constexpr int N = BN;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int prow = lane >> 2, pch = lane & 3;
  const bf16* a_src[A_LD];
  const bf16* b_src[B_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int row = (wave * A_LD + i) * 16 + prow;
    const int gm = min(m0 + row, p.M - 1);
    a_src[i] = p.A + (long)gm * p.lda + (pch ^ lswz64(row)) * 8;
  }
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int row = (wave * B_LD + i) * 16 + prow;
    b_src[i] = p.B + (long)row * p.ldb + (pch ^ lswz64(row)) * 8;
  }
  const int nk = p.K / BK2;          // >= 2 (host checks)
  auto issue = [&](int ks) {
    unsigned char* st = smem + (ks % NS) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(a_src[i] + ks * BK2), (lds_void_t*)(st + (wave * A_LD + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(b_src[i] + ks * BK2),
                                       (lds_void_t*)(st + BMT * BK2 * 2 + (wave * B_LD + i) * 1024), 16, 0, 0);
  };

  // the tail's operands: column of this lane's 8-wide group jp, rows i*16 + c16 of the wave tile
  const int row0 = m0 + wm * (BMT / 2), col0 = wn * WN;
  f32x4 bias_lo[NP], bias_hi[NP];
  bf16x8 res[MT][NP];
  auto col_of = [&](int jp) { return col0 + (odd ? (2 * jp + 1) * 16 + 4 * (g - 1) : (2 * jp) * 16 + 4 * g); };
  auto issue_tail = [&]() {
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int col = col_of(jp);
      bias_lo[jp] = *reinterpret_cast<const f32x4*>(p.bias + col);
      bias_hi[jp] = *reinterpret_cast<const f32x4*>(p.bias + col + 4);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int gm = min(row0 + i * 16 + c16, p.M - 1);
        res[i][jp] = *reinterpret_cast<const bf16x8*>(p.residual + (long)gm * p.ldr + col);
      }
    }
  };

  // the device-resident dropout step: loaded first, so it is the OLDEST entry of the vmcnt queue and never waited for
  const IqRng rng = p.drop_on ? rng_resolve(p.rng) : p.rng;
  { // de-phase co-resident workgroups (gemm_nt.hip)
    const int d = (int)(((unsigned)blockIdx.x * 2654435761u) >> 30) * 2;
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(8);
  }
  issue(0);
  issue(1);
  const int ch = lane >> 4;
  auto compute = [&](int ks) {
    const bf16* As = reinterpret_cast<const bf16*>(smem + (ks % NS) * STAGE_BYTES);
    const bf16* Bs = As + BMT * BK2;
    bf16x8 af[MT], bfr[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * (BMT / 2) + i * 16 + (lane & 15);
      af[i] = *reinterpret_cast<const bf16x8*>(As + row * BK2 + (ch ^ lswz64(row)) * 8);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * WN + j * 16 + (lane & 15);
      bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + row * BK2 + (ch ^ lswz64(row)) * 8);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile: gemm_common.h
  };
  // head of every stage: this wave's pieces of stage ks have landed (the youngest stage stays in flight), everyone's
  // have (barrier, which also says stage ks-1 is no longer read), the slot ks-1 vacated is refilled.
  // The last two stages are peeled: a loop body that also held the tail's register loads made the compiler drain the
  // whole queue (vmcnt(0)) on every iteration.
  for (int ks = 0; ks + 2 < nk; ++ks) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_STAGE) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(ks + 2);
    compute(ks);
  }
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_STAGE) : "memory");          // stage nk-2 landed, nk-1 in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  issue_tail();                                                               // every stage is issued: the tail's loads go
  asm volatile("" ::: "memory");                                              // ... now, not after this stage's MFMAs
  compute(nk - 2);
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TAIL_LOADS) : "memory");         // stage nk-1 landed, only the tail's loads fly
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  compute(nk - 1);

  // ---- tail -----------------------------------------------------------------------------------------------------
  // pass 1: z = dropout(acc + bias) + residual, rounded to bf16, stored; the rounded values replace the accumulators
  float rsum[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int gm = row0 + i * 16 + c16;
    rsum[i] = 0.f;
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      float w[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float va = acc[i][2 * jp][r], vb = acc[i][2 * jp + 1][r];
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
        w[r] = __uint_as_float(sw[0]) + bias_lo[jp][r];
        w[4 + r] = __uint_as_float(sw[1]) + bias_hi[jp][r];
      }
      const int col = col_of(jp);
      if (p.drop_on) {
        const uint32_t keep = dropout_keep8(rng, (uint64_t)((long)gm * N + col) >> 3, p.thresh);
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = ((keep >> e) & 1u) ? w[e] * p.dscale : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] += (float)res[i][jp][e];
      const bf16x8 zb = pack8(w);
      if (gm < p.M) *reinterpret_cast<bf16x8*>(p.Z + (long)gm * N + col) = zb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float lo = (float)zb[r], hi = (float)zb[4 + r];
        acc[i][2 * jp][r] = lo;
        acc[i][2 * jp + 1][r] = hi;
        rsum[i] += lo + hi;
      }
    }
    rsum[i] += __shfl_xor(rsum[i], 16, 64);
    rsum[i] += __shfl_xor(rsum[i], 32, 64);      // this wave's half of the row, on all 4 lanes that share it
  }
  // exchange between the two column-half waves of a row block: red[pass][wn][row in block]
  float* red = reinterpret_cast<float*>(smem);
  __builtin_amdgcn_s_barrier();                  // every wave has left the operand ring
  asm volatile("" ::: "memory");
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) red[wn * BMT + wm * (BMT / 2) + i * 16 + c16] = rsum[i];
  }
  // raw barrier + lgkmcnt only: __syncthreads() would also wait (vmcnt(0)) for the Z stores just issued
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float mean[MT], rstd[MT];
  const float invD = 1.0f / (float)N;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    mean[i] = (rsum[i] + red[(1 - wn) * BMT + wm * (BMT / 2) + i * 16 + c16]) * invD;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[i]; q += d * d; }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    rsum[i] = q;
  }
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) red[2 * BMT + wn * BMT + wm * (BMT / 2) + i * 16 + c16] = rsum[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const float var = (rsum[i] + red[2 * BMT + (1 - wn) * BMT + wm * (BMT / 2) + i * 16 + c16]) * invD;
    rstd[i] = 1.0f / sqrtf(var + p.eps);
    const int gm = row0 + i * 16 + c16;
    if (wn == 0 && g == 0 && gm < p.M) { p.mean[gm] = mean[i]; p.rstd[gm] = rstd[i]; }
  }
  // pass 2: x = gamma * (z - mean) * rstd + beta
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) {
    const int col = col_of(jp);
    const f32x4 g_lo = *reinterpret_cast<const f32x4*>(p.gamma + col), g_hi = *reinterpret_cast<const f32x4*>(p.gamma + col + 4);
    const f32x4 b_lo = *reinterpret_cast<const f32x4*>(p.beta + col), b_hi = *reinterpret_cast<const f32x4*>(p.beta + col + 4);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int gm = row0 + i * 16 + c16;
      float y[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[r] = g_lo[r] * ((acc[i][2 * jp][r] - mean[i]) * rstd[i]) + b_lo[r];
        y[4 + r] = g_hi[r] * ((acc[i][2 * jp + 1][r] - mean[i]) * rstd[i]) + b_hi[r];
      }
      if (gm < p.M) *reinterpret_cast<bf16x8*>(p.X + (long)gm * N + col) = pack8(y);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Band variant: ONE workgroup per CU, each owning an equal band of rows (cfg B: 50432 rows / 256 CUs = 197 = one frame),
// 8 waves as 4 (rows) x 2 (column halves), the main loop of gemm_big.hip (whole-line operand units, two phases per
// 64-deep K-tile, ping-pong wave halves, counted vmcnt, no address arithmetic in the loop), then the tail above.
// Why: with 128-row blocks the 394 workgroups of cfg B land two or one to a CU, and a 24-stage loop in a 4-wave
// workgroup spends most of its life between a barrier, 8 fragment reads and 12 MFMAs.  Here every CU gets the same
// rows, the 13..16 row groups of a band go 4 / 3 / 3 / 3 to the wave rows (a missing fourth group is not multiplied),
// and a phase is 24 MFMAs behind the partner half's reads.
// LDS: A0 / A1 = the first / second 32 rows of every wave row (128 rows x 128 B, 16 KiB), B0 / B1 = the W rows of column
// half 0 / 1 (96 x 128 B, 12 KiB; 12 DMA pieces for 8 waves: waves 4..7 request one of theirs twice), two sets: 112 KiB.
constexpr int LNB_THREADS = 512, LNB_AUNIT = 16384;
constexpr int lnb_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

template <int BN>      // D = 192
__global__ __launch_bounds__(LNB_THREADS, 1) void gemm_ln_band_kernel(const GemmLnParams p, int nwg) {
  constexpr int WN = BN / 2, NT = WN / 16, NP = NT / 2, MT = 4, BMT = 256;
  constexpr int BUNIT = WN * 128, BPIECES = BUNIT / 1024;          // 12 KiB, 12 pieces
  constexpr int SET = 2 * LNB_AUNIT + 2 * BUNIT;
  static_assert(BN == 192, "12 B pieces over 8 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, half = wave >> 2;       // half: the ping-pong group (wave rows 0,1 | 2,3)
  const int g = lane >> 4, c16 = lane & 15, ch = lane >> 4;
  const bool odd = (g & 1) != 0;
  constexpr int N = BN;
  // this workgroup's band and its row groups: 4 / 3 / 3 / 3 (13) ... 4 / 4 / 4 / 4 (16)
  const int r0 = (int)((long)blockIdx.x * p.M / nwg), r1 = (int)((long)(blockIdx.x + 1) * p.M / nwg);
  const int ngroups = (r1 - r0 + 15) >> 4;
  const int gbase = ngroups >> 2, grem = ngroups & 3;
  const int gcount = gbase + (wm < grem ? 1 : 0);                  // row groups of this wave row (3 or 4)
  const int gstart = wm * gbase + min(wm, grem);
  const int row0 = r0 + gstart * 16;                               // first row of this wave's tile
  const IqRng rng = p.drop_on ? rng_resolve(p.rng) : p.rng;       // oldest entry of the vector-memory queue

  // ---- unit requests: uniform base + per-lane offsets (bytes) -----------------------------------------------------------------
  unsigned offA[2][2], offB[2][2];                                 // [half-unit][piece]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    {
      const int ur = (wave * 2 + i) * 8 + (lane >> 3);             // unit row 0..127: wave row ur >> 5 (= this wave's), row ur & 31
      const int gch = ((lane & 7) ^ ((ur >> 1) & 7)) * 16;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        offA[h][i] = (unsigned)(min(row0 + h * 32 + (ur & 31), r1 - 1) * p.lda * 2 + gch);
    }
    {
      int piece = wave + 8 * i;
      if (piece >= BPIECES) piece = wave;                          // waves 4..7: their first piece again (uniform counts)
      const int ur = piece * 8 + (lane >> 3);                      // 0..95
      const int gch = ((lane & 7) ^ ((ur >> 1) & 7)) * 16;
#pragma unroll
      for (int j = 0; j < 2; ++j) offB[j][i] = (unsigned)((j * WN + ur) * p.ldb * 2 + gch);
    }
  }
  const char* baseA = reinterpret_cast<const char*>(p.A);
  const char* baseB = reinterpret_cast<const char*>(p.B);
  int iset = 0, ikt = 0;
  auto issue_a = [&](int h) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(baseA + ikt * 128 + offA[h][i]),
                                       (lds_void_t*)(smem + iset * SET + h * LNB_AUNIT + (wave * 2 + i) * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int j) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int piece = wave + 8 * i;
      if (piece >= BPIECES) piece = wave;
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)(baseB + ikt * 128 + offB[j][i]),
                                       (lds_void_t*)(smem + iset * SET + 2 * LNB_AUNIT + j * BUNIT + piece * 1024), 16, 0, 0);
    }
  };
  const int total = p.K / 64;                                      // K-tiles (>= 2: host)
  // prologue: K-tile 0 whole, A0 B0 of K-tile 1
  issue_a(0); issue_b(0); issue_b(1); issue_a(1);
  iset = 1; ikt = 1;
  issue_a(0); issue_b(0);
  __builtin_amdgcn_s_waitcnt(lnb_vmcnt(6));                        // A0 B0 B1 of K-tile 0 landed
  __builtin_amdgcn_s_barrier();
  if (half == 1) __builtin_amdgcn_s_barrier();                     // half 1 runs one phase-half behind

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4], bfr[NT * 2];     // A fragments [row group of the half-unit][k-step]; B fragments [column tile][k-step]
  int cset = 0;
  auto read_a = [&](int h) {
    const unsigned char* U = smem + cset * SET + h * LNB_AUNIT;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ur = wm * 32 + it * 16 + c16;
#pragma unroll
      for (int s = 0; s < 2; ++s) af[it * 2 + s] = *reinterpret_cast<const bf16x8*>(U + ur * 128 + (((s * 4 + ch) ^ ((ur >> 1) & 7)) << 4));
    }
  };
  auto read_b = [&]() {
    const unsigned char* U = smem + cset * SET + 2 * LNB_AUNIT + wn * BUNIT;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const int ur = ct * 16 + c16;
#pragma unroll
      for (int s = 0; s < 2; ++s) bfr[ct * 2 + s] = *reinterpret_cast<const bf16x8*>(U + ur * 128 + (((s * 4 + ch) ^ ((ur >> 1) & 7)) << 4));
    }
  };
  auto wait_left = [&](int left) {
    if (left == 8) __builtin_amdgcn_s_waitcnt(lnb_vmcnt(8));
    else if (left == 6) __builtin_amdgcn_s_waitcnt(lnb_vmcnt(6));
    else if (left == 2) __builtin_amdgcn_s_waitcnt(lnb_vmcnt(2));
    else if (left == 0) __builtin_amdgcn_s_waitcnt(lnb_vmcnt(0));
  };
  auto sync_a = [&](int left) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (half == 1) wait_left(left);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto sync_b = [&](int left) {
    if (half == 0) wait_left(left);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  const bool four = gcount > 3;                                    // wave-uniform: this wave row has a fourth row group
  for (int q = 0; q < total; ++q) {
    const bool more1 = q + 1 < total, more2 = q + 2 < total;
    // phase X: row groups 0, 1 of every wave; requests B1, A1 of K-tile q+1
    read_b();
    read_a(0);
    if (more1) { issue_b(1); issue_a(1); iset ^= 1; ++ikt; }
    sync_a(more1 ? 8 : 0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc[it][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ct * 2 + s], af[it * 2 + s], acc[it][ct], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    sync_b(more1 ? 8 : 0);
    // phase Y: row groups 2, (3); requests A0, B0 of K-tile q+2
    read_a(1);
    if (more2) { issue_a(0); issue_b(0); }
    sync_a(more2 ? 6 : more1 ? 2 : -1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        acc[2][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ct * 2 + s], af[s], acc[2][ct], 0, 0, 0);
    if (four) {
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc[3][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ct * 2 + s], af[2 + s], acc[3][ct], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    sync_b(more2 ? 6 : more1 ? 2 : -1);
    cset ^= 1;
  }
  if (half == 0) __builtin_amdgcn_s_barrier();                     // matches half 1's extra barrier: every wave has left the ring

  // ---- tail (the arithmetic of gemm_ln_kernel above, 4 wave rows) --------------------------------------------------------------
  const int col0 = wn * WN;
  auto col_of = [&](int jp) { return col0 + (odd ? (2 * jp + 1) * 16 + 4 * (g - 1) : (2 * jp) * 16 + 4 * g); };
  bool rowok[MT];
  f32x4 bias_lo[NP], bias_hi[NP];
  bf16x8 res[MT][NP];
#pragma unroll
  for (int i = 0; i < MT; ++i) rowok[i] = i < gcount && row0 + i * 16 + c16 < r1;
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) {
    const int col = col_of(jp);
    bias_lo[jp] = *reinterpret_cast<const f32x4*>(p.bias + col);
    bias_hi[jp] = *reinterpret_cast<const f32x4*>(p.bias + col + 4);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int gm = min(row0 + i * 16 + c16, r1 - 1);
      res[i][jp] = *reinterpret_cast<const bf16x8*>(p.residual + (long)gm * p.ldr + col);
    }
  }
  // pass 1: z = dropout(acc + bias) + residual, rounded to bf16, stored; the rounded values replace the accumulators
  float rsum[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int gm = row0 + i * 16 + c16;
    rsum[i] = 0.f;
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      float w[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float va = acc[i][2 * jp][r], vb = acc[i][2 * jp + 1][r];
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
        w[r] = __uint_as_float(sw[0]) + bias_lo[jp][r];
        w[4 + r] = __uint_as_float(sw[1]) + bias_hi[jp][r];
      }
      const int col = col_of(jp);
      if (p.drop_on) {
        const uint32_t keep = dropout_keep8(rng, (uint64_t)((long)gm * N + col) >> 3, p.thresh);
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = ((keep >> e) & 1u) ? w[e] * p.dscale : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] += (float)res[i][jp][e];
      const bf16x8 zb = pack8(w);
      if (rowok[i]) *reinterpret_cast<bf16x8*>(p.Z + (long)gm * N + col) = zb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float lo = (float)zb[r], hi = (float)zb[4 + r];
        acc[i][2 * jp][r] = lo;
        acc[i][2 * jp + 1][r] = hi;
        rsum[i] += lo + hi;
      }
    }
    rsum[i] += __shfl_xor(rsum[i], 16, 64);
    rsum[i] += __shfl_xor(rsum[i], 32, 64);      // this wave's half of the row, on all 4 lanes that share it
  }
  // exchange between the two column-half waves of a wave row: red[pass][wn][wave row][row]
  float* red = reinterpret_cast<float*>(smem);
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) red[wn * BMT + wm * 64 + i * 16 + c16] = rsum[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (raw barrier + lgkmcnt only: the Z stores stay in flight)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float mean[MT], rstd[MT];
  const float invD = 1.0f / (float)N;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    mean[i] = (rsum[i] + red[(1 - wn) * BMT + wm * 64 + i * 16 + c16]) * invD;
    float qv = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[i]; qv += d * d; }
    qv += __shfl_xor(qv, 16, 64);
    qv += __shfl_xor(qv, 32, 64);
    rsum[i] = qv;
  }
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) red[2 * BMT + wn * BMT + wm * 64 + i * 16 + c16] = rsum[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const float var = (rsum[i] + red[2 * BMT + (1 - wn) * BMT + wm * 64 + i * 16 + c16]) * invD;
    rstd[i] = 1.0f / sqrtf(var + p.eps);
    const int gm = row0 + i * 16 + c16;
    if (wn == 0 && g == 0 && rowok[i]) { p.mean[gm] = mean[i]; p.rstd[gm] = rstd[i]; }
  }
  // pass 2: x = gamma * (z - mean) * rstd + beta
#pragma unroll
  for (int jp = 0; jp < NP; ++jp) {
    const int col = col_of(jp);
    const f32x4 g_lo = *reinterpret_cast<const f32x4*>(p.gamma + col), g_hi = *reinterpret_cast<const f32x4*>(p.gamma + col + 4);
    const f32x4 b_lo = *reinterpret_cast<const f32x4*>(p.beta + col), b_hi = *reinterpret_cast<const f32x4*>(p.beta + col + 4);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int gm = row0 + i * 16 + c16;
      float y[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[r] = g_lo[r] * ((acc[i][2 * jp][r] - mean[i]) * rstd[i]) + b_lo[r];
        y[4 + r] = g_hi[r] * ((acc[i][2 * jp + 1][r] - mean[i]) * rstd[i]) + b_hi[r];
      }
      if (rowok[i]) *reinterpret_cast<bf16x8*>(p.X + (long)gm * N + col) = pack8(y);
    }
  }
}

// one workgroup per CU (a multiple of 256 workgroups), 13..16 row groups per band; 0 = not this kernel's shape
static int band_workgroups(int M, int D, int K) {
  if (D != 192 || K % 64 != 0 || K < 512) return 0;     // (K = 192: 24.7 us against the 4-wave kernel's 22.4 -- three K-tiles leave nothing to overlap)
  const int nwg = 256 * ((M + 65535) / 65536);
  const int lo = M / nwg, hi = (M + nwg - 1) / nwg;
  return (lo > 192 && hi <= 256) ? nwg : 0;
}

template <int BMT, int BN>
int launch(const GemmLnParams& p, hipStream_t st) {
  const size_t lds = (size_t)3 * (BMT + BN) * 32 * 2;
  static_assert(3 * (BMT + BN) * 32 * 2 >= 4 * BMT * 4, "the reduction scratch fits in the ring");
  auto k = gemm_ln_kernel<BMT, BN>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<<<(p.M + BMT - 1) / BMT, LNG_THREADS, lds, st>>>(p);
  return iq_launch_status();
}

}  // namespace

extern "C" int iq_gemm_ln_supported(int D, int K) {
  return ((D == 128 || D == 192 || D == 256) && K >= 64 && K % 32 == 0) ? 1 : 0;
}

extern "C" int iq_gemm_bf16_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual,
                               int ldr, const iq_dropout_t* drop, const float* gamma, const float* beta, float eps,
                               void* Z, void* X, float* mean, float* rstd, int M, int D, int K, iq_stream_t stream) {
  if (M <= 0) return IQ_OK;
  if (!A || !W || !bias || !residual || !gamma || !beta || !Z || !X || !mean || !rstd) return IQ_ERR_ARG;
  if (!iq_gemm_ln_supported(D, K) || (lda % 8) || (ldw % 8) || (ldr % 8)) return IQ_ERR_UNSUPPORTED;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)residual | (uintptr_t)Z | (uintptr_t)X | (uintptr_t)bias |
       (uintptr_t)gamma | (uintptr_t)beta) % 16) return IQ_ERR_ARG;
  GemmLnParams p = {};
  p.A = (const bf16*)A; p.B = (const bf16*)W; p.Z = (bf16*)Z; p.X = (bf16*)X;
  p.lda = lda; p.ldb = ldw; p.M = M; p.K = K;
  p.bias = bias; p.residual = (const bf16*)residual; p.ldr = ldr;
  p.gamma = gamma; p.beta = beta; p.mean = mean; p.rstd = rstd; p.eps = eps;
  if (drop && drop->p > 0.f) {
    if (drop->p >= 1.f) return IQ_ERR_ARG;
    p.drop_on = 1;
    p.rng.seed = drop->seed; p.rng.step = drop->step; p.rng.site = drop->site; p.rng.step_dev = drop->step_dev;
    p.thresh = dropout_thresh(drop->p);
    p.dscale = dropout_scale(drop->p);
  }
  hipStream_t st = (hipStream_t)stream;
  IQ_PROF(IQ_FAM_GEMM_NT, st);
  // A + W + residual read, Z + X written (+ bias, gamma, beta, statistics)
  const double work_bytes = 2.0 * ((double)M * K + (double)D * K + 3.0 * (double)M * D) + 12.0 * D + 8.0 * M;
  const double work_flops = 2.0 * (double)M * D * K;
  if (const int nwg = band_workgroups(M, D, K)) {
    if ((lda % 64) == 0 && (ldw % 64) == 0 && (((uintptr_t)A | (uintptr_t)W) % 128) == 0) {
      constexpr int lds = 2 * (2 * LNB_AUNIT + 2 * 96 * 128);
      auto k = gemm_ln_band_kernel<192>;
      static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)attr;
      k<<<nwg, LNB_THREADS, lds, st>>>(p, nwg);
      IQ_PROF_K(work_bytes, work_flops, "gemm_ln_band_kernel<192>");
      return iq_launch_status();
    }
  }
  // Row block: 128, or 64 where 128-row blocks would leave CUs without a workgroup (cfg C: M = 16,640 = 130 blocks of 128
  // on 256 CUs).  Same K order per row either way: Z / X / statistics are bit-identical.  IQ_TUNE_LN_ROWS forces it (probes).
  static const int tune_rows = [] { const char* e = getenv("IQ_TUNE_LN_ROWS"); return e ? atoi(e) : 0; }();
  const bool rows64 = D == 256 || tune_rows == 64 || (tune_rows == 0 && (M + 127) / 128 <= 320);
  IQ_PROF_K(work_bytes, work_flops, "gemm_ln_kernel<%d, %d>", rows64 ? 64 : 128, D);
  switch (D) {
    case 128: return rows64 ? launch<64, 128>(p, st) : launch<128, 128>(p, st);
    case 192: return rows64 ? launch<64, 192>(p, st) : launch<128, 192>(p, st);
    case 256: return launch<64, 256>(p, st);      // 64-row blocks: 128 accumulator + 64 residual registers would spill
    default: return IQ_ERR_UNSUPPORTED;
  }
}
